// vh_params.cpp -- zParameters*.txt -> VhAppState -> the parameter structs of the host classes.
// Restates mLib's ParameterFile as vendored by the reference (DSCroot/Include/mLib/include/core-util/parameterFile.h,
// stringUtilConvert.h) and the parametersFromGlobalAppState builders.  No HIP in here.
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>

#include "../../include/vh_api.h"

namespace {

// removeSpecialCharacters, parameterFile.h:136-152: blanks, tabs, quotes and semicolons off both ends
void strip(std::string& s)
{
    const std::string junk = " \t\";";
    while (!s.empty() && junk.find(s.front()) != std::string::npos) s.erase(s.begin());
    while (!s.empty() && junk.find(s.back()) != std::string::npos) s.pop_back();
}

// removeComments, parameterFile.h:155-165: everything from the first "//", then "#", then ";" on
void uncomment(std::string& s)
{
    for (const char* c : { "//", "#", ";" }) {
        const size_t at = s.find(c);
        if (at != std::string::npos) s = s.substr(0, at);
    }
}

typedef std::map<std::string, std::string> Values;

// addParameterFile, parameterFile.h:22-60 (a later line overrides an earlier one)
void parseStream(std::istream& in, Values& values)
{
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back(); // the reference's files have CRLF line ends
        uncomment(line);
        strip(line);
        if (line.empty()) continue;
        const size_t sep = line.find('=');
        if (sep == std::string::npos) continue;
        std::string name = line.substr(0, sep), value = line.substr(sep + 1);
        strip(name);
        strip(value);
        if (name.empty()) continue;
        values[name] = value;
    }
}

struct Reader {
    const Values& v;
    uint32_t found;
    bool get(const char* name, std::string& out)
    {
        const auto it = v.find(name);
        if (it == v.end()) return false;
        out = it->second;
        found++;
        return true;
    }
    // convert::to<...>, stringUtilConvert.h:78-127
    void u32(const char* name, uint32_t& out)
    {
        std::string s;
        out = 0;
        if (!get(name, s)) return;
        try { out = (uint32_t)std::stoi(s); } catch (...) { out = 0; }
    }
    void f32(const char* name, float& out)
    {
        std::string s;
        out = 0.0f;
        if (!get(name, s)) return;
        try { out = std::stof(s); } catch (...) { out = 0.0f; }
    }
    void boolean(const char* name, uint32_t& out)
    {
        std::string s;
        out = 0;
        if (!get(name, s)) return;
        out = (s == "false" || s == "False" || s == "0") ? 0u : 1u;
    }
    void text(const char* name, char* out, size_t cap)
    {
        std::string s;
        out[0] = 0;
        if (!get(name, s)) return;
        std::strncpy(out, s.c_str(), cap - 1);
        out[cap - 1] = 0;
    }
    template <class T> void vec3(const char* name, T out[3])
    {
        std::string s;
        out[0] = out[1] = out[2] = T();
        if (!get(name, s)) return;
        std::string t;
        for (char c : s)
            if (c != 'f') t.push_back(c); // util::removeChar(s, 'f')
        std::stringstream ss(t);
        ss >> out[0] >> out[1] >> out[2];
    }
};

void fill(const Values& values, VhAppState* out)
{
    std::memset(out, 0, sizeof(*out));
    Reader r{ values, 0 };
    r.u32("s_sensorIdx", out->s_sensorIdx);
    r.u32("s_adapterWidth", out->s_adapterWidth);
    r.u32("s_adapterHeight", out->s_adapterHeight);
    r.f32("s_sensorDepthMax", out->s_sensorDepthMax);
    r.f32("s_sensorDepthMin", out->s_sensorDepthMin);
    r.f32("s_SDFVoxelSize", out->s_SDFVoxelSize);
    r.f32("s_SDFMarchingCubeThreshFactor", out->s_SDFMarchingCubeThreshFactor);
    r.f32("s_SDFTruncation", out->s_SDFTruncation);
    r.f32("s_SDFTruncationScale", out->s_SDFTruncationScale);
    r.f32("s_SDFMaxIntegrationDistance", out->s_SDFMaxIntegrationDistance);
    r.u32("s_SDFIntegrationWeightSample", out->s_SDFIntegrationWeightSample);
    r.u32("s_SDFIntegrationWeightMax", out->s_SDFIntegrationWeightMax);
    r.u32("s_hashNumBuckets", out->s_hashNumBuckets);
    r.u32("s_hashNumSDFBlocks", out->s_hashNumSDFBlocks);
    r.u32("s_hashMaxCollisionLinkedListSize", out->s_hashMaxCollisionLinkedListSize);
    r.f32("s_SDFRayIncrementFactor", out->s_SDFRayIncrementFactor);
    r.f32("s_SDFRayThresSampleDistFactor", out->s_SDFRayThresSampleDistFactor);
    r.f32("s_SDFRayThresDistFactor", out->s_SDFRayThresDistFactor);
    r.boolean("s_SDFUseGradients", out->s_SDFUseGradients);
    r.f32("s_depthSigmaD", out->s_depthSigmaD);
    r.f32("s_depthSigmaR", out->s_depthSigmaR);
    r.boolean("s_depthFilter", out->s_depthFilter);
    r.f32("s_colorSigmaD", out->s_colorSigmaD);
    r.f32("s_colorSigmaR", out->s_colorSigmaR);
    r.boolean("s_colorFilter", out->s_colorFilter);
    r.boolean("s_integrationEnabled", out->s_integrationEnabled);
    r.boolean("s_trackingEnabled", out->s_trackingEnabled);
    r.boolean("s_timingsDetailledEnabled", out->s_timingsDetailledEnabled);
    r.boolean("s_timingsTotalEnabled", out->s_timingsTotalEnabled);
    r.boolean("s_garbageCollectionEnabled", out->s_garbageCollectionEnabled);
    r.u32("s_garbageCollectionStarve", out->s_garbageCollectionStarve);
    r.u32("s_marchingCubesMaxNumTriangles", out->s_marchingCubesMaxNumTriangles);
    r.boolean("s_streamingEnabled", out->s_streamingEnabled);
    r.vec3("s_streamingVoxelExtents", out->s_streamingVoxelExtents);
    r.vec3("s_streamingGridDimensions", out->s_streamingGridDimensions);
    r.vec3("s_streamingMinGridPos", out->s_streamingMinGridPos);
    r.u32("s_streamingInitialChunkListSize", out->s_streamingInitialChunkListSize);
    r.f32("s_streamingRadius", out->s_streamingRadius);
    r.vec3("s_streamingPos", out->s_streamingPos);
    r.u32("s_streamingOutParts", out->s_streamingOutParts);
    r.boolean("s_offlineProcessing", out->s_offlineProcessing);
    r.boolean("s_binaryDumpSensorUseTrajectory", out->s_binaryDumpSensorUseTrajectory);
    r.boolean("s_binaryDumpSensorUseTrajectoryOnlyInit", out->s_binaryDumpSensorUseTrajectoryOnlyInit);
    r.boolean("s_playData", out->s_playData);
    r.boolean("s_recordData", out->s_recordData);
    r.boolean("s_recordCompression", out->s_recordCompression);
    r.boolean("s_reconstructionEnabled", out->s_reconstructionEnabled);
    r.text("s_recordDataFile", out->s_recordDataFile, sizeof(out->s_recordDataFile));
    // readParameter(name, std::vector<U>&), parameterFile.h:92-108: name[0], name[1], ... until one is missing
    const uint32_t before = r.found;
    for (uint32_t i = 0;; i++) {
        std::string s;
        if (!r.get(("s_binaryDumpSensorFile[" + std::to_string(i) + "]").c_str(), s)) break;
        if (i < 8) std::strncpy(out->s_binaryDumpSensorFile[i], s.c_str(), sizeof(out->s_binaryDumpSensorFile[i]) - 1);
        out->s_numBinaryDumpSensorFiles = i + 1;
    }
    if (out->s_numBinaryDumpSensorFiles > 8) out->s_numBinaryDumpSensorFiles = 8;
    r.found = before + (out->s_numBinaryDumpSensorFiles ? 1u : 0u);
    out->numKeysFound = r.found;
}

const float kIdentity[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };

} // namespace

extern "C" {

int vh_app_state_read(const char* filename, VhAppState* out)
{
    if (!filename || !out) return VH_ERR_BAD_ARGUMENT;
    std::ifstream f(filename);
    if (!f.is_open()) return VH_ERR_IO;
    Values values;
    parseStream(f, values);
    fill(values, out);
    return VH_OK;
}

int vh_app_state_parse(const char* text, VhAppState* out)
{
    if (!text || !out) return VH_ERR_BAD_ARGUMENT;
    std::istringstream in(text);
    Values values;
    parseStream(in, values);
    fill(values, out);
    return VH_OK;
}

void vh_hash_params_from_app_state(const VhAppState* gas, VhHashParams* p)
{
    std::memset(p, 0, sizeof(*p));
    std::memcpy(p->m_rigidTransform, kIdentity, sizeof(kIdentity));
    std::memcpy(p->m_rigidTransformInverse, kIdentity, sizeof(kIdentity));
    p->m_hashNumBuckets = gas->s_hashNumBuckets;
    p->m_hashBucketSize = VH_HASH_BUCKET_SIZE;
    p->m_hashMaxCollisionLinkedListSize = gas->s_hashMaxCollisionLinkedListSize;
    p->m_SDFBlockSize = VH_SDF_BLOCK_SIZE;
    p->m_numSDFBlocks = gas->s_hashNumSDFBlocks;
    p->m_virtualVoxelSize = gas->s_SDFVoxelSize;
    p->m_maxIntegrationDistance = gas->s_SDFMaxIntegrationDistance;
    p->m_truncation = gas->s_SDFTruncation;
    p->m_truncScale = gas->s_SDFTruncationScale;
    p->m_integrationWeightSample = gas->s_SDFIntegrationWeightSample;
    p->m_integrationWeightMax = gas->s_SDFIntegrationWeightMax;
    for (int i = 0; i < 3; i++) {
        p->m_streamingVoxelExtents[i] = gas->s_streamingVoxelExtents[i];
        p->m_streamingGridDimensions[i] = gas->s_streamingGridDimensions[i];
        p->m_streamingMinGridPos[i] = gas->s_streamingMinGridPos[i];
    }
    p->m_streamingInitialChunkListSize = gas->s_streamingInitialChunkListSize;
}

void vh_raycast_params_from_app_state(const VhAppState* gas, const float intrinsics[16], const float intrinsicsInv[16], VhRayCastParams* p)
{
    std::memset(p, 0, sizeof(*p));
    std::memcpy(p->m_viewMatrix, kIdentity, sizeof(kIdentity));
    std::memcpy(p->m_viewMatrixInverse, kIdentity, sizeof(kIdentity));
    std::memcpy(p->m_intrinsics, intrinsics ? intrinsics : kIdentity, sizeof(kIdentity));
    std::memcpy(p->m_intrinsicsInverse, intrinsicsInv ? intrinsicsInv : kIdentity, sizeof(kIdentity));
    p->m_width = gas->s_adapterWidth;
    p->m_height = gas->s_adapterHeight;
    p->m_minDepth = gas->s_sensorDepthMin;
    p->m_maxDepth = gas->s_sensorDepthMax;
    p->m_rayIncrement = gas->s_SDFRayIncrementFactor * gas->s_SDFTruncation;
    p->m_thresSampleDist = gas->s_SDFRayThresSampleDistFactor * p->m_rayIncrement;
    p->m_thresDist = gas->s_SDFRayThresDistFactor * p->m_rayIncrement;
    p->m_useGradients = gas->s_SDFUseGradients ? 1 : 0;
    p->m_maxNumVertices = gas->s_hashNumSDFBlocks * 6;
}

void vh_marching_cubes_params_from_app_state(const VhAppState* gas, VhMarchingCubesParams* p)
{
    std::memset(p, 0, sizeof(*p));
    p->m_maxNumTriangles = gas->s_marchingCubesMaxNumTriangles;
    p->m_threshMarchingCubes = gas->s_SDFMarchingCubeThreshFactor * gas->s_SDFVoxelSize;
    p->m_threshMarchingCubes2 = gas->s_SDFMarchingCubeThreshFactor * gas->s_SDFVoxelSize;
    p->m_sdfBlockSize = VH_SDF_BLOCK_SIZE;
    p->m_hashBucketSize = VH_HASH_BUCKET_SIZE;
    p->m_hashNumBuckets = gas->s_hashNumBuckets;
}

void vh_scene_options_from_app_state(const VhAppState* gas, VhSceneOptions* o)
{
    std::memset(o, 0, sizeof(*o));
    o->s_offlineProcessing = gas->s_offlineProcessing ? 1 : 0;
    o->s_garbageCollectionEnabled = gas->s_garbageCollectionEnabled ? 1 : 0;
    o->s_timingsDetailledEnabled = gas->s_timingsDetailledEnabled ? 1 : 0;
    o->s_garbageCollectionStarve = gas->s_garbageCollectionStarve;
    o->s_streamingOutParts = gas->s_streamingOutParts;
}

} // extern "C"
