// vh_sensor.cpp -- the per-frame image path between a depth sensor and integrate(): CUDARGBDAdapter::process
// (DSC/CUDARGBDAdapter.cpp:93-137) followed by CUDARGBDSensor::process (DSC/CUDARGBDSensor.cpp:147-257), as one
// host class over the kernels of vh_kernels.hip ("sensor pre-processing").  The D3D11 remapping branch
// (s_bUseCameraCalibration, :198-217) and the disabled erosion loop (:224-237) are not part of it.
#include <hip/hip_runtime.h>

#include <cstring>

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
template <class T> void devAlloc(T*& p, size_t n, const char* what) { checkHip(hipMalloc((void**)&p, sizeof(T) * (n ? n : 1)), what); }
} // namespace

CUDARGBDSensor::CUDARGBDSensor(const Config& c, vhStream_t stream) : m_cfg(c), m_stream(stream), m_frameNumber(0)
{
    if (c.depthWidth < 2 || c.depthHeight < 2 || c.colorWidth < 2 || c.colorHeight < 2 || c.adapterWidth < 2 || c.adapterHeight < 2)
        throw vh::Error(VH_ERR_BAD_ARGUMENT, "CUDARGBDSensor: image sizes must be at least 2x2");
    m_bFilterDepthValues = c.filterDepth; m_fBilateralFilterSigmaD = c.sigmaD; m_fBilateralFilterSigmaR = c.sigmaR;
    m_bFilterIntensityValues = c.filterIntensity; m_fBilateralFilterSigmaDIntensity = c.sigmaDIntensity; m_fBilateralFilterSigmaRIntensity = c.sigmaRIntensity;
    // adapt intrinsics, DSC/CUDARGBDAdapter.cpp:56-62
    std::memset(&m_depthCameraParams, 0, sizeof(m_depthCameraParams));
    m_depthCameraParams.fx = c.fx * ((float)c.adapterWidth / (float)c.depthWidth);
    m_depthCameraParams.fy = c.fy * ((float)c.adapterHeight / (float)c.depthHeight);
    m_depthCameraParams.mx = c.mx * ((float)(c.adapterWidth - 1) / (float)(c.depthWidth - 1));
    m_depthCameraParams.my = c.my * ((float)(c.adapterHeight - 1) / (float)(c.depthHeight - 1));
    m_depthCameraParams.m_sensorDepthWorldMin = c.sensorDepthMin;
    m_depthCameraParams.m_sensorDepthWorldMax = c.sensorDepthMax;
    m_depthCameraParams.m_imageWidth = c.adapterWidth;
    m_depthCameraParams.m_imageHeight = c.adapterHeight;

    const size_t nDepthIn = (size_t)c.depthWidth * c.depthHeight, nColorIn = (size_t)c.colorWidth * c.colorHeight;
    const size_t nOut = (size_t)c.adapterWidth * c.adapterHeight;
    d_depthMapFloat = d_depthMapResampledFloat = d_depthMapFilteredFloat = d_intensityMapFilteredFloat = nullptr;
    d_colorMapRaw = nullptr;
    d_colorMapFloat4 = d_colorMapResampledFloat4 = d_cameraSpaceFloat4 = d_normalMapFloat4 = nullptr;
    std::memset(&m_depthCameraData, 0, sizeof(m_depthCameraData));
    devAlloc(d_depthMapFloat, nDepthIn, "d_depthMapFloat");
    devAlloc(d_depthMapResampledFloat, nOut, "d_depthMapResampledFloat");
    devAlloc(d_colorMapRaw, 4 * nColorIn, "d_colorMapRaw");
    devAlloc(d_colorMapFloat4, 4 * nColorIn, "d_colorMapFloat4");
    devAlloc(d_colorMapResampledFloat4, 4 * nOut, "d_colorMapResampledFloat4");
    devAlloc(d_depthMapFilteredFloat, nOut, "d_depthMapFilteredFloat");
    devAlloc(d_cameraSpaceFloat4, 4 * nOut, "d_cameraSpaceFloat4");
    devAlloc(d_normalMapFloat4, 4 * nOut, "d_normalMapFloat4");
    devAlloc(d_intensityMapFilteredFloat, nOut, "d_intensityMapFilteredFloat");
    d_depthData = d_colorData = nullptr;
    devAlloc(d_depthData, nOut, "DepthCameraData::d_depthData");
    devAlloc(d_colorData, 4 * nOut, "DepthCameraData::d_colorData");
    m_depthCameraData.d_depthData = d_depthData;
    m_depthCameraData.d_colorData = d_colorData;
    // a resampled pixel whose nearest source pixel is outside the source is left untouched by the reference:
    // start from "invalid" instead of from uninitialised memory
    check(vh_set_invalid_float_map(d_depthMapResampledFloat, c.adapterWidth, c.adapterHeight, m_stream), "setInvalidFloatMap");
    checkHip(hipMemsetAsync(d_colorMapResampledFloat4, 0, sizeof(float) * 4 * nOut, (hipStream_t)m_stream), "clear colour");
}

CUDARGBDSensor::~CUDARGBDSensor()
{
    (void)hipStreamSynchronize((hipStream_t)m_stream);
    void* all[] = { d_depthMapFloat, d_depthMapResampledFloat, d_colorMapRaw, d_colorMapFloat4, d_colorMapResampledFloat4, d_depthMapFilteredFloat,
                    d_cameraSpaceFloat4, d_normalMapFloat4, d_intensityMapFilteredFloat, d_depthData, d_colorData };
    for (void* p : all)
        if (p) (void)hipFree(p);
}

void CUDARGBDSensor::setFiterDepthValues(bool b, float sigmaD, float sigmaR)
{
    m_bFilterDepthValues = b; m_fBilateralFilterSigmaD = sigmaD; m_fBilateralFilterSigmaR = sigmaR;
}
void CUDARGBDSensor::setFiterIntensityValues(bool b, float sigmaD, float sigmaR)
{
    m_bFilterIntensityValues = b; m_fBilateralFilterSigmaDIntensity = sigmaD; m_fBilateralFilterSigmaRIntensity = sigmaR;
}

void CUDARGBDSensor::process(const float* h_depthFloat, const unsigned char* h_colorRGBX)
{
    if (!h_depthFloat || !h_colorRGBX) throw vh::Error(VH_ERR_BAD_ARGUMENT, "CUDARGBDSensor::process: null frame");
    const Config& c = m_cfg;
    const unsigned int W = c.adapterWidth, H = c.adapterHeight;
    hipStream_t s = (hipStream_t)m_stream;
    // ---- CUDARGBDAdapter::process :107-131
    checkHip(hipMemcpyAsync(d_colorMapRaw, h_colorRGBX, 4 * (size_t)c.colorWidth * c.colorHeight, hipMemcpyHostToDevice, s), "upload colour");
    check(vh_convert_color_raw_to_float4(d_colorMapFloat4, d_colorMapRaw, c.colorWidth, c.colorHeight, m_stream), "convertColorRawToFloat4");
    if (c.colorWidth == W && c.colorHeight == H) check(vh_copy_float4_map(d_colorMapResampledFloat4, d_colorMapFloat4, W, H, m_stream), "copyFloat4Map");
    else check(vh_resample_float4_map(d_colorMapResampledFloat4, W, H, d_colorMapFloat4, c.colorWidth, c.colorHeight, m_stream), "resampleFloat4Map");
    checkHip(hipMemcpyAsync(d_depthMapFloat, h_depthFloat, sizeof(float) * (size_t)c.depthWidth * c.depthHeight, hipMemcpyHostToDevice, s), "upload depth");
    check(vh_resample_float_map(d_depthMapResampledFloat, W, H, d_depthMapFloat, c.depthWidth, c.depthHeight, m_stream), "resampleFloatMap");
    // ---- CUDARGBDSensor::process :159-248
    if (m_bFilterIntensityValues) check(vh_gauss_filter_float4_map(d_colorData, d_colorMapResampledFloat4, m_fBilateralFilterSigmaDIntensity, m_fBilateralFilterSigmaRIntensity, W, H, m_stream), "gaussFilterFloat4Map");
    else check(vh_copy_float4_map(d_colorData, d_colorMapResampledFloat4, W, H, m_stream), "copyFloat4Map");
    if (m_bFilterDepthValues) check(vh_gauss_filter_float_map(d_depthMapFilteredFloat, d_depthMapResampledFloat, m_fBilateralFilterSigmaD, m_fBilateralFilterSigmaR, W, H, m_stream), "gaussFilterFloatMap");
    else check(vh_copy_float_map(d_depthMapFilteredFloat, d_depthMapResampledFloat, W, H, m_stream), "copyFloatMap");
    // (the reference also calls setInvalidFloatMap on d_depthData here and overwrites it right away, :188-219)
    check(vh_copy_float_map(d_depthData, d_depthMapFilteredFloat, W, H, m_stream), "copyFloatMap");
    check(vh_convert_color_to_intensity_float(d_intensityMapFilteredFloat, d_colorData, W, H, m_stream), "convertColorToIntensityFloat");
    check(vh_convert_depth_float_to_camera_space_float4(d_cameraSpaceFloat4, d_depthData, &m_depthCameraParams, W, H, m_stream), "convertDepthFloatToCameraSpaceFloat4");
    check(vh_compute_normals(d_normalMapFloat4, d_cameraSpaceFloat4, W, H, m_stream), "computeNormals");
    // the source buffers are the caller's: they may be reused as soon as this returns
    checkHip(hipStreamSynchronize(s), "CUDARGBDSensor::process");
    m_frameNumber++;
}
