// vh_c_api.cpp -- handle-level C ABI over the C++ host classes (include/vh.hpp):
// what an FFI binding (ctypes, cgo, JNI ...) of the reference's host-class
// interface would call.  Exceptions become error codes here.
#include "vh_handles.hpp"

char* vh_last_error_buffer() { extern thread_local char vh_g_lastError[512]; return vh_g_lastError; }
thread_local char vh_g_lastError[512] = "";

namespace {

template <class F> int guarded(F&& f) { return vh_guarded(static_cast<F&&>(f)); }

inline vh::mat4f toMat(const float m[16])
{
    vh::mat4f r;
    std::memcpy(r.m, m, sizeof(r.m));
    return r;
}
inline vh::vec3f toVec(const float v[3]) { return { v[0], v[1], v[2] }; }

} // namespace

extern "C" {

const char* vh_last_error_message(void) { return vh_g_lastError; }

// ---- CUDASceneRepHashSDF ----------------------------------------------------

int vh_scene_rep_create(const VhHashParams* hp, const VhSceneOptions* opt, vhStream_t stream, VhSceneRep** out)
{
    if (!hp || !out) return VH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    return guarded([&] {
        const VhSceneOptions o = opt ? *opt : CUDASceneRepHashSDF::defaultOptions();
        *out = new VhSceneRep(*hp, o, stream);
    });
}
void vh_scene_rep_destroy(VhSceneRep* s) { delete s; }

int vh_scene_rep_integrate(VhSceneRep* s, const float rigidTransform[16], const VhDepthCameraData* cam,
                           const VhDepthCameraParams* cp, const uint32_t* d_bitMask)
{
    if (!s || !rigidTransform || !cam || !cp) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { s->impl.integrate(toMat(rigidTransform), *cam, *cp, d_bitMask); });
}
int vh_scene_rep_integrate_ahead(VhSceneRep* s, const float rigidTransform[16], const VhDepthCameraData* cam,
                                 const VhDepthCameraParams* cp, const uint32_t* d_bitMask, VhFrameJob** job)
{
    if (!s || !rigidTransform || !cam || !cp) return VH_ERR_BAD_ARGUMENT;
    if (job) *job = nullptr;
    return guarded([&] {
        VhFrameJob* j = s->impl.integrateAhead(toMat(rigidTransform), *cam, *cp, d_bitMask);
        if (job) *job = j;
    });
}
int vh_scene_rep_integrate_finish(VhSceneRep* s, const VhDepthCameraData* cam, const VhDepthCameraParams* cp)
{
    if (!s || !cam || !cp) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { s->impl.integrateFinish(*cam, *cp); });
}
int vh_scene_rep_set_last_rigid_transform_and_compactify(VhSceneRep* s, const float rigidTransform[16], const VhDepthCameraParams* cp)
{
    if (!s || !rigidTransform || !cp) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { s->impl.setLastRigidTransformAndCompactify(toMat(rigidTransform), *cp); });
}
int vh_scene_rep_reset(VhSceneRep* s)
{
    if (!s) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { s->impl.reset(); });
}
int vh_scene_rep_get_hash_data(VhSceneRep* s, VhHashData* out)
{
    if (!s || !out) return VH_ERR_BAD_ARGUMENT;
    *out = s->impl.getHashData();
    return VH_OK;
}
int vh_scene_rep_get_hash_params(VhSceneRep* s, VhHashParams* out)
{
    if (!s || !out) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { *out = s->impl.getHashParams(); });
}
int vh_scene_rep_get_heap_free_count(VhSceneRep* s, uint32_t* out)
{
    if (!s || !out) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { *out = s->impl.getHeapFreeCount(); });
}
int vh_scene_rep_get_num_occupied_blocks(VhSceneRep* s, uint32_t* out)
{
    if (!s || !out) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { *out = s->impl.getNumOccupiedBlocks(); });
}
int vh_scene_rep_debug_hash(VhSceneRep* s, uint32_t report[4])
{
    if (!s) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { s->impl.debugHash(report); });
}
int vh_scene_rep_get_state(VhSceneRep* s, uint32_t* out)
{
    if (!s || !out) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { s->impl.getState(out); });
}
int vh_scene_rep_get_timings(VhSceneRep* s, double out[4])
{
    if (!s || !out) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { s->impl.getTimings(out); });
}
int vh_scene_rep_set_options(VhSceneRep* s, const VhSceneOptions* opt)
{
    if (!s || !opt) return VH_ERR_BAD_ARGUMENT;
    s->impl.setOptions(*opt);
    return VH_OK;
}

// ---- CUDARayCastSDF -----------------------------------------------------------

int vh_raycast_create(const VhRayCastParams* rp, vhStream_t stream, VhRayCast** out)
{
    if (!rp || !out) return VH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    return guarded([&] { *out = new VhRayCast(*rp, stream); });
}
void vh_raycast_destroy(VhRayCast* r) { delete r; }
int vh_raycast_render(VhRayCast* r, const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp,
                      const float lastRigidTransform[16])
{
    if (!r || !hd || !hp || !cp || !lastRigidTransform) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { r->impl.render(*hd, *hp, *cp, toMat(lastRigidTransform)); });
}
int vh_raycast_render_co(VhRayCast* r, const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp,
                         const float lastRigidTransform[16], VhFrameJob* job)
{
    if (!r || !hd || !hp || !cp || !lastRigidTransform) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { r->impl.render(*hd, *hp, *cp, toMat(lastRigidTransform), job); });
}
int vh_raycast_get_data(VhRayCast* r, VhRayCastData* out)
{
    if (!r || !out) return VH_ERR_BAD_ARGUMENT;
    *out = r->impl.getRayCastData();
    return VH_OK;
}
int vh_raycast_get_params(VhRayCast* r, VhRayCastParams* out)
{
    if (!r || !out) return VH_ERR_BAD_ARGUMENT;
    *out = r->impl.getRayCastParams();
    return VH_OK;
}
int vh_raycast_get_timings(VhRayCast* r, double out[4])
{
    if (!r || !out) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { r->impl.getTimings(out); });
}
int vh_raycast_get_event_pair_overhead(VhRayCast* r, double* ms)
{
    if (!r || !ms) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { *ms = r->impl.getEventPairOverheadMs(); });
}
int vh_raycast_set_interval_splatting(VhRayCast* r, int enabled)
{
    if (!r) return VH_ERR_BAD_ARGUMENT;
    r->impl.setIntervalSplatting(enabled != 0);
    return VH_OK;
}
int vh_raycast_set_timing(VhRayCast* r, int enabled)
{
    if (!r) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { r->impl.setTiming(enabled != 0, enabled == 2, 1); });
}
int vh_raycast_set_timing_stride(VhRayCast* r, int enabled, uint32_t stride)
{
    if (!r) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { r->impl.setTiming(enabled != 0, enabled == 2, stride); });
}

// ---- CUDASceneRepChunkGrid ----------------------------------------------------

int vh_chunk_grid_create(VhSceneRep* s, const float voxelExtents[3], const int32_t gridDimensions[3],
                         const int32_t minGridPos[3], uint32_t initialChunkListSize, int streamingEnabled,
                         uint32_t streamOutParts, VhChunkGrid** out)
{
    if (!s || !voxelExtents || !gridDimensions || !minGridPos || !out) return VH_ERR_BAD_ARGUMENT;
    if (gridDimensions[0] <= 0 || gridDimensions[1] <= 0 || gridDimensions[2] <= 0) return VH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    return guarded([&] {
        *out = new VhChunkGrid(&s->impl, toVec(voxelExtents), { gridDimensions[0], gridDimensions[1], gridDimensions[2] },
                               { minGridPos[0], minGridPos[1], minGridPos[2] }, initialChunkListSize, streamingEnabled != 0, streamOutParts);
    });
}
void vh_chunk_grid_destroy(VhChunkGrid* g) { delete g; }

int vh_chunk_grid_stream_out_to_cpu_pass0_gpu(VhChunkGrid* g, const float posCamera[3], float radius, int useParts, int multiThreaded)
{
    if (!g || !posCamera) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { g->impl.streamOutToCPUPass0GPU(toVec(posCamera), radius, useParts != 0, multiThreaded != 0); });
}
int vh_chunk_grid_stream_out_to_cpu_pass1_cpu(VhChunkGrid* g, int multiThreaded)
{
    if (!g) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { g->impl.streamOutToCPUPass1CPU(multiThreaded != 0); });
}
int vh_chunk_grid_stream_in_to_gpu_pass0_cpu(VhChunkGrid* g, const float posCamera[3], float radius, int useParts, int multiThreaded)
{
    if (!g || !posCamera) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { g->impl.streamInToGPUPass0CPU(toVec(posCamera), radius, useParts != 0, multiThreaded != 0); });
}
int vh_chunk_grid_stream_in_to_gpu_pass1_gpu(VhChunkGrid* g, int multiThreaded)
{
    if (!g) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { g->impl.streamInToGPUPass1GPU(multiThreaded != 0); });
}
int vh_chunk_grid_stream_out_to_cpu(VhChunkGrid* g, const float posCamera[3], float radius, int useParts, uint32_t* nStreamedBlocks)
{
    if (!g || !posCamera) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] {
        unsigned int n = 0;
        g->impl.streamOutToCPU(toVec(posCamera), radius, useParts != 0, n);
        if (nStreamedBlocks) *nStreamedBlocks = n;
    });
}
int vh_chunk_grid_stream_in_to_gpu(VhChunkGrid* g, const float posCamera[3], float radius, int useParts, uint32_t* nStreamedBlocks)
{
    if (!g || !posCamera) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] {
        unsigned int n = 0;
        g->impl.streamInToGPU(toVec(posCamera), radius, useParts != 0, n);
        if (nStreamedBlocks) *nStreamedBlocks = n;
    });
}
int vh_chunk_grid_stream_out_to_cpu_all(VhChunkGrid* g)
{
    if (!g) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { g->impl.streamOutToCPUAll(); });
}
int vh_chunk_grid_stream_in_to_gpu_all(VhChunkGrid* g, const float posCamera[3], float radius, int useParts, uint32_t* nStreamedBlocks)
{
    if (!g || !posCamera) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] {
        unsigned int n = 0;
        g->impl.streamInToGPUAll(toVec(posCamera), radius, useParts != 0, n);
        if (nStreamedBlocks) *nStreamedBlocks = n;
    });
}
int vh_chunk_grid_get_bit_mask_gpu(VhChunkGrid* g, const uint32_t** d_bitMask)
{
    if (!g || !d_bitMask) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { *d_bitMask = g->impl.getBitMaskGPU(); });
}
int vh_chunk_grid_reset(VhChunkGrid* g)
{
    if (!g) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { g->impl.reset(); });
}
int vh_chunk_grid_debug_check_for_duplicates(VhChunkGrid* g)
{
    if (!g) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { g->impl.debugCheckForDuplicates(); });
}
int vh_chunk_grid_get_num_failed_inserts(VhChunkGrid* g, uint32_t* out)
{
    if (!g || !out) return VH_ERR_BAD_ARGUMENT;
    *out = g->impl.getNumFailedInserts();
    return VH_OK;
}
int vh_chunk_grid_get_statistics(VhChunkGrid* g, uint32_t out[3])
{
    if (!g || !out) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { g->impl.getStatistics(out); });
}
int vh_chunk_grid_download_host_blocks(VhChunkGrid* g, VhSDFBlockDesc* descs, VhVoxel* blocks, uint32_t capacity, uint32_t* n)
{
    if (!g || !n) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] {
        std::vector<SDFBlockDesc> d;
        std::vector<vh::SDFBlock> b;
        g->impl.downloadHostBlocks(d, b);
        *n = (uint32_t)d.size();
        if (descs && blocks) {
            if (d.size() > capacity) throw vh::Error(VH_ERR_STAGING_OVERFLOW, "download buffer too small");
            if (!d.empty()) {
                std::memcpy(descs, d.data(), sizeof(SDFBlockDesc) * d.size());
                std::memcpy(blocks, b.data(), sizeof(vh::SDFBlock) * b.size());
            }
        }
    });
}
int vh_chunk_grid_save_to_file(VhChunkGrid* g, const char* filename, const float camPos[3], float radius)
{
    if (!g || !filename || !camPos) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { g->impl.saveToFile(filename, toVec(camPos), radius); });
}
int vh_chunk_grid_load_from_file(VhChunkGrid* g, const char* filename, const float camPos[3], float radius)
{
    if (!g || !filename || !camPos) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { g->impl.loadFromFile(filename, toVec(camPos), radius); });
}

// ---- CUDAMarchingCubesHashSDF ---------------------------------------------------

int vh_marching_cubes_create(const VhMarchingCubesParams* params, vhStream_t stream, VhMarchingCubes** out)
{
    if (!params || !out) return VH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    return guarded([&] { *out = new VhMarchingCubes(*params, stream); });
}
void vh_marching_cubes_destroy(VhMarchingCubes* mc) { delete mc; }
int vh_marching_cubes_parameters(uint32_t maxNumTriangles, float threshFactor, float voxelSize, uint32_t hashNumBuckets,
                                 VhMarchingCubesParams* out)
{
    if (!out) return VH_ERR_BAD_ARGUMENT;
    *out = CUDAMarchingCubesHashSDF::parameters(maxNumTriangles, threshFactor, voxelSize, hashNumBuckets);
    return VH_OK;
}
int vh_marching_cubes_set_offline_processing(VhMarchingCubes* mc, int enabled)
{
    if (!mc) return VH_ERR_BAD_ARGUMENT;
    mc->impl.setOfflineProcessing(enabled != 0);
    return VH_OK;
}
int vh_marching_cubes_extract_iso_surface(VhMarchingCubes* mc, const VhHashData* hd, const VhHashParams* hp,
                                          const float minCorner[3], const float maxCorner[3], int boxEnabled, int copy)
{
    if (!mc || !hd || !hp) return VH_ERR_BAD_ARGUMENT;
    const vh::vec3f lo = minCorner ? toVec(minCorner) : vh::vec3f{ 0, 0, 0 }, hi = maxCorner ? toVec(maxCorner) : vh::vec3f{ 0, 0, 0 };
    return guarded([&] {
        if (copy) mc->impl.extractIsoSurface(*hd, *hp, lo, hi, boxEnabled != 0);
        else mc->impl.extractIsoSurfaceWithoutCopy(*hd, *hp, lo, hi, boxEnabled != 0);
    });
}
int vh_marching_cubes_extract_iso_surface_chunk_grid(VhMarchingCubes* mc, VhChunkGrid* grid, const float camPos[3], float radius)
{
    if (!mc || !grid || !camPos) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { mc->impl.extractIsoSurface(grid->impl, toVec(camPos), radius); });
}
int vh_marching_cubes_copy_triangles_to_cpu(VhMarchingCubes* mc)
{
    if (!mc) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { mc->impl.copyTrianglesToCPU(); });
}
int vh_marching_cubes_clear_mesh_buffer(VhMarchingCubes* mc)
{
    if (!mc) return VH_ERR_BAD_ARGUMENT;
    mc->impl.clearMeshBuffer();
    return VH_OK;
}
int vh_marching_cubes_get_counts(VhMarchingCubes* mc, uint32_t out[2])
{
    if (!mc || !out) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { out[0] = mc->impl.getNumTriangles(); out[1] = mc->impl.getNumOccupiedBlocks(); });
}
int vh_marching_cubes_download_triangles(VhMarchingCubes* mc, VhTriangle* out, uint32_t n)
{
    if (!mc) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { mc->impl.downloadTriangles(out, n); });
}
int vh_marching_cubes_get_mesh_size(VhMarchingCubes* mc, uint64_t out[2])
{
    if (!mc || !out) return VH_ERR_BAD_ARGUMENT;
    out[0] = mc->impl.getMeshData().m_Vertices.size();
    out[1] = mc->impl.getMeshData().m_FaceIndicesVertices.size();
    return VH_OK;
}
int vh_marching_cubes_get_mesh(VhMarchingCubes* mc, float* vertices3, float* colors4, uint32_t* faceIndices)
{
    if (!mc) return VH_ERR_BAD_ARGUMENT;
    const vh::MeshData& m = mc->impl.getMeshData();
    if (vertices3 && !m.m_Vertices.empty()) std::memcpy(vertices3, m.m_Vertices.data(), sizeof(vh::vec3f) * m.m_Vertices.size());
    if (colors4 && !m.m_Colors.empty()) std::memcpy(colors4, m.m_Colors.data(), sizeof(float) * m.m_Colors.size());
    if (faceIndices && !m.m_FaceIndicesVertices.empty()) std::memcpy(faceIndices, m.m_FaceIndicesVertices.data(), sizeof(uint32_t) * m.m_FaceIndicesVertices.size());
    return VH_OK;
}
int vh_marching_cubes_save_mesh(VhMarchingCubes* mc, const char* filename, const float transform[16], int overwriteExistingFile)
{
    if (!mc || !filename) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] {
        if (transform) { const vh::mat4f t = toMat(transform); mc->impl.saveMesh(filename, &t, overwriteExistingFile != 0); }
        else mc->impl.saveMesh(filename, nullptr, overwriteExistingFile != 0);
    });
}

// ---- SensorData / SensorDataReader ------------------------------------------------

namespace {
void fillInfo(const vh::SensorData& d, VhSensorDataInfo* out)
{
    std::memset(out, 0, sizeof(*out));
    out->m_versionNumber = d.m_versionNumber;
    out->m_colorCompressionType = d.m_colorCompressionType; out->m_depthCompressionType = d.m_depthCompressionType;
    out->m_colorWidth = d.m_colorWidth; out->m_colorHeight = d.m_colorHeight;
    out->m_depthWidth = d.m_depthWidth; out->m_depthHeight = d.m_depthHeight;
    out->m_depthShift = d.m_depthShift;
    out->m_numFrames = d.m_frames.size(); out->m_numIMUFrames = d.m_IMUFrames.size();
    std::memcpy(out->m_colorIntrinsic, d.m_colorIntrinsic.m, 64); std::memcpy(out->m_colorExtrinsic, d.m_colorExtrinsic.m, 64);
    std::memcpy(out->m_depthIntrinsic, d.m_depthIntrinsic.m, 64); std::memcpy(out->m_depthExtrinsic, d.m_depthExtrinsic.m, 64);
    std::strncpy(out->m_sensorName, d.m_sensorName.c_str(), sizeof(out->m_sensorName) - 1);
}
} // namespace

int vh_sensor_data_create(const VhSensorDataInfo* h, VhSensorData** out)
{
    if (!h || !out) return VH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    return guarded([&] {
        std::unique_ptr<VhSensorData> s(new VhSensorData);
        vh::SensorData& d = s->impl;
        if (h->m_versionNumber != 0 && h->m_versionNumber != vh::SensorData::kVersion) throw vh::Error(VH_ERR_VERSION_MISMATCH, "only version 4 sequences can be written");
        d.m_sensorName = std::string(h->m_sensorName, strnlen(h->m_sensorName, sizeof(h->m_sensorName)));
        d.m_colorCompressionType = h->m_colorCompressionType; d.m_depthCompressionType = h->m_depthCompressionType;
        if (d.m_colorCompressionType < 0 || d.m_colorCompressionType > 2 || d.m_depthCompressionType < 0 || d.m_depthCompressionType > 1)
            throw vh::Error(VH_ERR_BAD_ARGUMENT, "compression type not supported for writing");
        d.m_colorWidth = h->m_colorWidth; d.m_colorHeight = h->m_colorHeight; d.m_depthWidth = h->m_depthWidth; d.m_depthHeight = h->m_depthHeight;
        d.m_depthShift = h->m_depthShift;
        d.m_colorIntrinsic = toMat(h->m_colorIntrinsic); d.m_colorExtrinsic = toMat(h->m_colorExtrinsic);
        d.m_depthIntrinsic = toMat(h->m_depthIntrinsic); d.m_depthExtrinsic = toMat(h->m_depthExtrinsic);
        *out = s.release();
    });
}
int vh_sensor_data_load(const char* filename, VhSensorData** out)
{
    if (!filename || !out) return VH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    return guarded([&] {
        std::unique_ptr<VhSensorData> s(new VhSensorData);
        s->impl.loadFromFile(filename);
        *out = s.release();
    });
}
void vh_sensor_data_destroy(VhSensorData* s) { delete s; }
int vh_sensor_data_save(const VhSensorData* s, const char* filename)
{
    if (!s || !filename) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { s->impl.saveToFile(filename); });
}
int vh_sensor_data_info(const VhSensorData* s, VhSensorDataInfo* out)
{
    if (!s || !out) return VH_ERR_BAD_ARGUMENT;
    fillInfo(s->impl, out);
    return VH_OK;
}
int vh_sensor_data_add_frame(VhSensorData* s, const uint8_t* colorRGB, const uint16_t* depth, const float cameraToWorld[16], uint64_t tsColor, uint64_t tsDepth)
{
    if (!s) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { s->impl.addFrame(colorRGB, depth, cameraToWorld ? toMat(cameraToWorld) : vh::mat4f::identity(), tsColor, tsDepth); });
}
int vh_sensor_data_add_frame_compressed(VhSensorData* s, const uint8_t* colorBytes, uint64_t numColorBytes, const uint16_t* depth,
                                        const float cameraToWorld[16], uint64_t tsColor, uint64_t tsDepth)
{
    if (!s || (!colorBytes && numColorBytes)) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] {
        s->impl.addFrame(nullptr, depth, cameraToWorld ? toMat(cameraToWorld) : vh::mat4f::identity(), tsColor, tsDepth);
        s->impl.m_frames.back().m_colorCompressed.assign(colorBytes, colorBytes + numColorBytes);
    });
}
int vh_sensor_data_add_imu_frame(VhSensorData* s, const double v[15], uint64_t timeStamp)
{
    if (!s || !v) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] {
        vh::SensorData::IMUFrame f;
        std::memcpy(&f, v, 15 * sizeof(double));
        f.timeStamp = timeStamp;
        s->impl.m_IMUFrames.push_back(f);
    });
}
int vh_sensor_data_get_frame(const VhSensorData* s, uint64_t idx, uint16_t* depth, uint8_t* colorRGB, float cameraToWorld[16], uint64_t timeStamps[2])
{
    if (!s) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] {
        if (idx >= s->impl.m_frames.size()) throw vh::Error(VH_ERR_BAD_ARGUMENT, "out of bounds");
        const vh::SensorData::RGBDFrame& f = s->impl.m_frames[(size_t)idx];
        if (depth) s->impl.decompressDepth((size_t)idx, depth);
        if (colorRGB) s->impl.decompressColor((size_t)idx, colorRGB);
        if (cameraToWorld) std::memcpy(cameraToWorld, f.m_cameraToWorld.m, 64);
        if (timeStamps) { timeStamps[0] = f.m_timeStampColor; timeStamps[1] = f.m_timeStampDepth; }
    });
}

int vh_sensor_data_reader_create(const char* filename, VhSensorDataReader** out)
{
    if (!filename || !out) return VH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    return guarded([&] {
        std::unique_ptr<VhSensorDataReader> r(new VhSensorDataReader);
        r->impl.createFirstConnected(filename);
        *out = r.release();
    });
}
void vh_sensor_data_reader_destroy(VhSensorDataReader* r) { delete r; }
int vh_sensor_data_reader_info(const VhSensorDataReader* r, VhSensorDataInfo* out)
{
    if (!r || !out) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { fillInfo(r->impl.getSensorData(), out); });
}
int vh_sensor_data_reader_process_depth(VhSensorDataReader* r, int* gotFrame, const float** depthFloat, const uint8_t** colorRGBX)
{
    if (!r || !gotFrame) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] {
        *gotFrame = r->impl.processDepth() ? 1 : 0;
        if (depthFloat) *depthFloat = r->impl.getDepthFloat();
        if (colorRGBX) *colorRGBX = r->impl.getColorRGBX();
    });
}
int vh_sensor_data_reader_get_rigid_transform(const VhSensorDataReader* r, int offset, float out[16])
{
    if (!r || !out) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { std::memcpy(out, r->impl.getRigidTransform(offset).m, 64); });
}
int vh_sensor_data_reader_get_curr_frame(const VhSensorDataReader* r, uint32_t* currFrame, uint32_t* numFrames)
{
    if (!r) return VH_ERR_BAD_ARGUMENT;
    if (currFrame) *currFrame = r->impl.getCurrFrame();
    if (numFrames) *numFrames = r->impl.getNumFrames();
    return VH_OK;
}

// ---- CUDARGBDSensor ---------------------------------------------------------------

int vh_rgbd_sensor_create(const uint32_t sizes[6], const float intrinsics[6], vhStream_t stream, VhRGBDSensor** out)
{
    if (!sizes || !intrinsics || !out) return VH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    CUDARGBDSensor::Config c;
    std::memset(&c, 0, sizeof(c));
    c.depthWidth = sizes[0]; c.depthHeight = sizes[1]; c.colorWidth = sizes[2]; c.colorHeight = sizes[3]; c.adapterWidth = sizes[4]; c.adapterHeight = sizes[5];
    c.fx = intrinsics[0]; c.fy = intrinsics[1]; c.mx = intrinsics[2]; c.my = intrinsics[3]; c.sensorDepthMin = intrinsics[4]; c.sensorDepthMax = intrinsics[5];
    c.sigmaD = c.sigmaR = c.sigmaDIntensity = c.sigmaRIntensity = 1.0f;
    return guarded([&] { *out = new VhRGBDSensor(c, stream); });
}
void vh_rgbd_sensor_destroy(VhRGBDSensor* s) { delete s; }
int vh_rgbd_sensor_set_filter_depth_values(VhRGBDSensor* s, int enabled, float sigmaD, float sigmaR)
{
    if (!s) return VH_ERR_BAD_ARGUMENT;
    s->impl.setFiterDepthValues(enabled != 0, sigmaD, sigmaR);
    return VH_OK;
}
int vh_rgbd_sensor_set_filter_intensity_values(VhRGBDSensor* s, int enabled, float sigmaD, float sigmaR)
{
    if (!s) return VH_ERR_BAD_ARGUMENT;
    s->impl.setFiterIntensityValues(enabled != 0, sigmaD, sigmaR);
    return VH_OK;
}
int vh_rgbd_sensor_process(VhRGBDSensor* s, const float* h_depthFloat, const uint8_t* h_colorRGBX)
{
    if (!s) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { s->impl.process(h_depthFloat, h_colorRGBX); });
}
int vh_rgbd_sensor_get_depth_camera_data(VhRGBDSensor* s, VhDepthCameraData* out)
{
    if (!s || !out) return VH_ERR_BAD_ARGUMENT;
    *out = s->impl.getDepthCameraData();
    return VH_OK;
}
int vh_rgbd_sensor_get_depth_camera_params(VhRGBDSensor* s, VhDepthCameraParams* out)
{
    if (!s || !out) return VH_ERR_BAD_ARGUMENT;
    *out = s->impl.getDepthCameraParams();
    return VH_OK;
}
int vh_rgbd_sensor_get_maps(VhRGBDSensor* s, float** d_cameraSpace4, float** d_normals4, float** d_intensity)
{
    if (!s) return VH_ERR_BAD_ARGUMENT;
    if (d_cameraSpace4) *d_cameraSpace4 = s->impl.getCameraSpacePositionsFloat4();
    if (d_normals4) *d_normals4 = s->impl.getNormalMapFloat4();
    if (d_intensity) *d_intensity = s->impl.getIntensityMapFilteredFloat();
    return VH_OK;
}

// ---- CUDACameraTrackingMultiRes -----------------------------------------------------

int vh_camera_tracking_create(uint32_t imageWidth, uint32_t imageHeight, uint32_t levels, vhStream_t stream, VhCameraTracking** out)
{
    if (!out) return VH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    return guarded([&] { *out = new VhCameraTracking(imageWidth, imageHeight, levels, stream); });
}
void vh_camera_tracking_destroy(VhCameraTracking* t) { delete t; }
int vh_camera_tracking_apply_ct(VhCameraTracking* t, float* d_input4, float* d_inputNormals4, float* d_model4, float* d_modelNormals4,
                                const float lastTransform[16], const VhTrackingState* settings, const float deltaTransformEstimate[16],
                                const VhDepthCameraParams* cp, float transformOut[16], int* trackingLost, VhIcpState* state)
{
    if (!t || !lastTransform || !settings || !cp || !transformOut) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] {
        const vh::mat4f est = deltaTransformEstimate ? toMat(deltaTransformEstimate) : vh::mat4f::identity();
        const vh::mat4f r = t->impl.applyCT(d_input4, d_inputNormals4, d_model4, d_modelNormals4, toMat(lastTransform), *settings, est, *cp);
        std::memcpy(transformOut, r.m, sizeof(r.m));
        if (trackingLost) *trackingLost = CUDACameraTrackingMultiRes::isTrackingLost(r) ? 1 : 0;
        if (state) *state = t->impl.getLastState();
    });
}

} // extern "C"
