// vh_tracking.cpp -- CUDACameraTrackingMultiRes (DSC/CUDACameraTrackingMultiRes.{h,cpp}) over the vh_icp_* steps,
// and the reader of zParametersTracking*.txt (GlobalCameraTrackingState).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>

#include "../../include/vh.hpp"
#include "vh_host_util.hpp"

namespace {
inline void check(int code, const char* what)
{
    if (code != 0) throw vh::Error(code, std::string(what) + ": " + vh_error_string(code));
}
inline void checkHip(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw vh::Error(-(int)e, std::string(what) + ": " + hipGetErrorString(e));
}
float* allocFloats(size_t n, const char* what)
{
    float* p = nullptr;
    checkHip(hipMalloc((void**)&p, sizeof(float) * (n ? n : 1)), what);
    return p;
}
} // namespace

CUDACameraTrackingMultiRes::CUDACameraTrackingMultiRes(unsigned int imageWidth, unsigned int imageHeight, unsigned int levels, vhStream_t stream)
    : m_levels(levels), m_stream(stream), d_partials(nullptr), d_state(nullptr), d_deltaEstimate(nullptr)
{
    if (levels == 0 || levels > VH_TRACKING_MAX_LEVELS || (imageWidth >> (levels - 1)) < 2 || (imageHeight >> (levels - 1)) < 2)
        throw vh::Error(VH_ERR_BAD_ARGUMENT, "CUDACameraTrackingMultiRes: bad pyramid");
    std::memset(&m_lastState, 0, sizeof(m_lastState));
    unsigned int fac = 1;
    for (unsigned int i = 0; i < levels; i++) { // :39-95
        m_imageWidth.push_back(imageWidth / fac);
        m_imageHeight.push_back(imageHeight / fac);
        const size_t n = 4 * (size_t)m_imageWidth[i] * m_imageHeight[i];
        d_correspondence.push_back(allocFloats(n, "d_correspondence"));
        d_correspondenceNormal.push_back(allocFloats(n, "d_correspondenceNormal"));
        d_input.push_back(i ? allocFloats(n, "d_input") : nullptr); // the finest level is the caller's maps
        d_inputNormal.push_back(i ? allocFloats(n, "d_inputNormal") : nullptr);
        d_model.push_back(i ? allocFloats(n, "d_model") : nullptr);
        d_modelNormal.push_back(i ? allocFloats(n, "d_modelNormal") : nullptr);
        fac *= 2;
    }
    d_partials = allocFloats(30 * (size_t)vh_icp_num_partials(imageWidth, imageHeight), "d_partials");
    checkHip(hipMalloc((void**)&d_state, sizeof(VhIcpState)), "VhIcpState");
    d_deltaEstimate = allocFloats(16, "deltaEstimate");
}

CUDACameraTrackingMultiRes::~CUDACameraTrackingMultiRes()
{
    (void)hipStreamSynchronize((hipStream_t)m_stream);
    for (auto* v : { &d_correspondence, &d_correspondenceNormal, &d_input, &d_inputNormal, &d_model, &d_modelNormal })
        for (float* p : *v)
            if (p) (void)hipFree(p);
    if (d_partials) (void)hipFree(d_partials);
    if (d_state) (void)hipFree(d_state);
    if (d_deltaEstimate) (void)hipFree(d_deltaEstimate);
}

bool CUDACameraTrackingMultiRes::isTrackingLost(const vh::mat4f& m) { return m.m[0] == -std::numeric_limits<float>::infinity(); }

vh::mat4f CUDACameraTrackingMultiRes::applyCT(float* dInput, float* dInputNormals, float* dModel, float* dModelNormals, const vh::mat4f& lastTransform,
                                              const VhTrackingState& ts, const vh::mat4f& deltaTransformEstimate, const DepthCameraParams& cp)
{
    if (!dInput || !dInputNormals || !dModel || !dModelNormals) throw vh::Error(VH_ERR_BAD_ARGUMENT, "applyCT: null map");
    hipStream_t s = (hipStream_t)m_stream;
    d_input[0] = dInput; d_inputNormal[0] = dInputNormals;
    d_model[0] = dModel; d_modelNormal[0] = dModelNormals;
    // the pyramids, :256-263
    for (unsigned int i = 0; i + 1 < m_levels; i++) {
        check(vh_resample_float4_map(d_input[i + 1], m_imageWidth[i + 1], m_imageHeight[i + 1], d_input[i], m_imageWidth[i], m_imageHeight[i], m_stream), "resampleFloat4Map");
        check(vh_compute_normals(d_inputNormal[i + 1], d_input[i + 1], m_imageWidth[i + 1], m_imageHeight[i + 1], m_stream), "computeNormals");
        check(vh_resample_float4_map(d_model[i + 1], m_imageWidth[i + 1], m_imageHeight[i + 1], d_model[i], m_imageWidth[i], m_imageHeight[i], m_stream), "resampleFloat4Map");
        check(vh_compute_normals(d_modelNormal[i + 1], d_model[i + 1], m_imageWidth[i + 1], m_imageHeight[i + 1], m_stream), "computeNormals");
    }
    checkHip(hipMemcpyAsync(d_deltaEstimate, deltaTransformEstimate.m, sizeof(float) * 16, hipMemcpyHostToDevice, s), "deltaEstimate");
    check(vh_icp_begin(d_state, d_deltaEstimate, m_stream), "vh_icp_begin");
    // coarse to fine, :265-279; align :291-321 with the loop exits taken on the device
    for (int level = (int)m_levels - 1; level >= 0; level--) {
        const unsigned int W = m_imageWidth[level], H = m_imageHeight[level];
        const float levelFactor = std::pow(2.0f, (float)level);
        check(vh_icp_begin_level(d_state, m_stream), "vh_icp_begin_level");
        for (unsigned int outer = 0; outer < ts.s_maxOuterIter[level]; outer++) {
            check(vh_icp_projective_correspondences(d_input[level], d_inputNormal[level], d_model[level], d_modelNormal[level], d_correspondence[level],
                                                    d_correspondenceNormal[level], W, H, ts.s_distThres[level], ts.s_normalThres[level], levelFactor,
                                                    d_state, &cp, m_stream), "projectiveCorrespondences");
            const unsigned int inner = ts.s_maxInnerIter[level];
            for (unsigned int i = 0; i < inner; i++) {
                check(vh_icp_build_linear_system(W, H, d_partials, d_input[level], d_correspondence[level], d_correspondenceNormal[level], d_state, m_stream), "buildLinearSystem");
                check(vh_icp_solve(d_state, d_partials, vh_icp_num_partials(W, H), ts.s_angleTransThres[level], ts.s_distTransThres[level],
                                   ts.s_residualEarlyOut[level], i + 1 == inner, m_stream), "vh_icp_solve");
            }
        }
    }
    checkHip(hipMemcpyAsync(&m_lastState, d_state, sizeof(VhIcpState), hipMemcpyDeviceToHost, s), "VhIcpState");
    checkHip(hipStreamSynchronize(s), "applyCT");
    d_input[0] = d_inputNormal[0] = d_model[0] = d_modelNormal[0] = nullptr;
    vh::mat4f out;
    if (m_lastState.lost) {
        for (float& v : out.m) v = -std::numeric_limits<float>::infinity();
        return out;
    }
    vh::mat4f delta;
    std::memcpy(delta.m, m_lastState.delta, sizeof(delta.m));
    return lastTransform * delta;
}

// ---------------------------------------------------------------------------
// zParametersTracking*.txt: the ParameterFile rules of vh_params.cpp, members s_name[level]
// ---------------------------------------------------------------------------

namespace {
void stripT(std::string& s)
{
    const std::string junk = " \t\";";
    while (!s.empty() && junk.find(s.front()) != std::string::npos) s.erase(s.begin());
    while (!s.empty() && junk.find(s.back()) != std::string::npos) s.pop_back();
}
void parseTracking(std::istream& in, VhTrackingState* out)
{
    std::map<std::string, std::string> values;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        for (const char* c : { "//", "#", ";" }) {
            const size_t at = line.find(c);
            if (at != std::string::npos) line = line.substr(0, at);
        }
        stripT(line);
        const size_t sep = line.find('=');
        if (line.empty() || sep == std::string::npos) continue;
        std::string name = line.substr(0, sep), value = line.substr(sep + 1);
        stripT(name); stripT(value);
        if (!name.empty()) values[name] = value;
    }
    std::memset(out, 0, sizeof(*out));
    auto u32 = [&](const std::string& k, uint32_t& v) { auto it = values.find(k); if (it == values.end()) return false; try { v = (uint32_t)std::stoi(it->second); } catch (...) { v = 0; } return true; };
    auto f32 = [&](const std::string& k, float& v) { auto it = values.find(k); if (it == values.end()) return false; try { v = std::stof(it->second); } catch (...) { v = 0.0f; } return true; };
    u32("s_maxLevels", out->s_maxLevels);
    for (unsigned int i = 0; i < VH_TRACKING_MAX_LEVELS; i++) { // readParameter(name, std::vector<U>&): name[0], name[1], ... until one is missing
        const std::string idx = "[" + std::to_string(i) + "]";
        if (!u32("s_maxOuterIter" + idx, out->s_maxOuterIter[i])) break;
        out->numLevelsFound = i + 1;
        u32("s_maxInnerIter" + idx, out->s_maxInnerIter[i]);
        f32("s_distThres" + idx, out->s_distThres[i]);
        f32("s_normalThres" + idx, out->s_normalThres[i]);
        f32("s_angleTransThres" + idx, out->s_angleTransThres[i]);
        f32("s_distTransThres" + idx, out->s_distTransThres[i]);
        f32("s_residualEarlyOut" + idx, out->s_residualEarlyOut[i]);
    }
}
} // namespace

extern "C" {

int vh_tracking_state_read(const char* filename, VhTrackingState* out)
{
    if (!filename || !out) return VH_ERR_BAD_ARGUMENT;
    std::ifstream f(filename);
    if (!f.is_open()) return VH_ERR_IO;
    parseTracking(f, out);
    return VH_OK;
}

int vh_tracking_state_parse(const char* text, VhTrackingState* out)
{
    if (!text || !out) return VH_ERR_BAD_ARGUMENT;
    std::istringstream in(text);
    parseTracking(in, out);
    return VH_OK;
}

} // extern "C"
