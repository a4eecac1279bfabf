// vh_host.cpp -- host side of the engine: memory ownership, the
// CUDASceneRepHashSDF and CUDARayCastSDF classes (include/vh.hpp) and the
// small utility entry points of the C ABI.
//
// Behavioural contract: DSC/CUDASceneRepHashSDF.h, DSC/CUDARayCastSDF.{h,cpp},
// DSC/VoxelUtilHashSDF.h:113-181 (DSC/ = /root/reference/DepthSensingCUDA/Source/).
// Unlike the reference frame loop there is no blocking host<->device round
// trip per frame in online mode: block counts stay on the device (persistent
// grid in the fused integrate kernel), bucket locks use a running epoch instead
// of a per-pass mutex reset, parameters are kernel arguments.
#include <hip/hip_runtime.h>

#include <array>
#include <climits>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/vh.hpp"
#include "vh_host_util.hpp"
#include "vh_stage_timer.hpp"

namespace {

inline void check(int code, const char* what)
{
    if (code != 0) throw vh::Error(code, std::string(what) + ": " + vh_error_string(code));
}
inline void checkHip(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw vh::Error(-(int)e, std::string(what) + ": " + hipGetErrorString(e));
}

} // namespace

// ---------------------------------------------------------------------------
// utility C ABI
// ---------------------------------------------------------------------------

namespace {
thread_local hipEvent_t t_launchStart = nullptr, t_launchStop = nullptr;
}
bool vh_take_launch_events(hipEvent_t* start, hipEvent_t* stop)
{
    if (!t_launchStart || !t_launchStop) return false;
    *start = t_launchStart; *stop = t_launchStop;
    t_launchStart = t_launchStop = nullptr;
    return true;
}

extern "C" {

const char* vh_version(void) { return "voxelhashing_amd 0.1.0 (gfx950)"; }

const char* vh_error_string(int code)
{
    if (code < 0) return hipGetErrorString((hipError_t)(-code));
    switch (code) {
    case VH_OK: return "ok";
    case VH_ERR_HEAP_EXHAUSTED: return "SDF block heap exhausted";
    case VH_ERR_STAGING_OVERFLOW: return "streaming staging buffer overflow";
    case VH_ERR_INSERT_FAILED: return "hash insert failed";
    case VH_ERR_BAD_ARGUMENT: return "bad argument";
    case VH_ERR_VERSION_MISMATCH: return "hashgrid version mismatch";
    case VH_ERR_TIMEOUT: return "the device made no progress (time-out)";
    case VH_ERR_IO: return "file i/o error";
    default: return "unknown error";
    }
}

int vh_malloc(void** devPtr, size_t bytes)
{
    if (!devPtr) return VH_ERR_BAD_ARGUMENT;
    VH_HIP(hipMalloc(devPtr, bytes));
    return VH_OK;
}
int vh_free(void* devPtr)
{
    if (!devPtr) return VH_ERR_BAD_ARGUMENT;
    VH_HIP(hipFree(devPtr));
    return VH_OK;
}
int vh_malloc_host(void** hostPtr, size_t bytes)
{
    if (!hostPtr) return VH_ERR_BAD_ARGUMENT;
    VH_HIP(hipHostMalloc(hostPtr, bytes ? bytes : 1, hipHostMallocDefault));
    return VH_OK;
}
int vh_free_host(void* hostPtr)
{
    if (!hostPtr) return VH_ERR_BAD_ARGUMENT;
    VH_HIP(hipHostFree(hostPtr));
    return VH_OK;
}
int vh_memcpy_h2d(void* dst, const void* src, size_t bytes, vhStream_t stream)
{
    VH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    VH_HIP(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_memcpy_d2h(void* dst, const void* src, size_t bytes, vhStream_t stream)
{
    VH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    VH_HIP(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_memset(void* dst, int value, size_t bytes, vhStream_t stream)
{
    VH_HIP(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
    return VH_OK;
}
int vh_time_next_launch(void* startEvent, void* stopEvent)
{
    if (!startEvent || !stopEvent) return VH_ERR_BAD_ARGUMENT;
    t_launchStart = (hipEvent_t)startEvent;
    t_launchStop = (hipEvent_t)stopEvent;
    return VH_OK;
}
int vh_stream_create(vhStream_t* out)
{
    if (!out) return VH_ERR_BAD_ARGUMENT;
    hipStream_t s = nullptr;
    VH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (vhStream_t)s;
    return VH_OK;
}
int vh_stream_destroy(vhStream_t stream)
{
    if (!stream) return VH_ERR_BAD_ARGUMENT;
    VH_HIP(hipStreamDestroy((hipStream_t)stream));
    return VH_OK;
}
int vh_stream_synchronize(vhStream_t stream)
{
    VH_HIP(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_device_synchronize(void)
{
    VH_HIP(hipDeviceSynchronize());
    return VH_OK;
}

// HashData::allocate, DSC/VoxelUtilHashSDF.h:113-139
int vh_hash_data_alloc(VhHashData* hd, const VhHashParams* hp)
{
    if (!hd || !hp || hp->m_hashNumBuckets == 0 || hp->m_numSDFBlocks == 0) return VH_ERR_BAD_ARGUMENT;
    if (hp->m_hashBucketSize != VH_HASH_BUCKET_SIZE || hp->m_SDFBlockSize != VH_SDF_BLOCK_SIZE) return VH_ERR_BAD_ARGUMENT;
    // ptr = blockId*512 must fit an int; entry indices must fit 32 bits
    if ((uint64_t)hp->m_numSDFBlocks * VH_SDF_BLOCK_VOXELS > (uint64_t)INT_MAX) return VH_ERR_BAD_ARGUMENT;
    if ((uint64_t)hp->m_hashNumBuckets * VH_HASH_BUCKET_SIZE > (uint64_t)UINT_MAX / 2) return VH_ERR_BAD_ARGUMENT;
    std::memset(hd, 0, sizeof(*hd));
    const size_t nb = hp->m_hashNumBuckets, ne = nb * VH_HASH_BUCKET_SIZE, nblk = hp->m_numSDFBlocks;
    int err = 0;
#define VH_ALLOC(field, bytes)                                              \
    if (!err) {                                                             \
        hipError_t e_ = hipMalloc((void**)&hd->field, (bytes));             \
        if (e_ != hipSuccess) { hd->field = nullptr; err = -(int)e_; }      \
    }
    VH_ALLOC(d_heap, sizeof(uint32_t) * nblk)
    VH_ALLOC(d_heapCounter, sizeof(uint32_t))
    VH_ALLOC(d_hash, sizeof(VhHashEntry) * ne)
    VH_ALLOC(d_hashDecision, sizeof(int32_t) * ne)
    VH_ALLOC(d_hashDecisionPrefix, sizeof(int32_t) * ne)
    VH_ALLOC(d_hashCompactified, sizeof(VhHashEntry) * ne)
    VH_ALLOC(d_hashCompactifiedCounter, sizeof(int32_t))
    VH_ALLOC(d_SDFBlocks, sizeof(VhVoxel) * nblk * VH_SDF_BLOCK_VOXELS)
    VH_ALLOC(d_hashBucketMutex, sizeof(int32_t) * nb)
    VH_ALLOC(d_bucketCount, sizeof(uint32_t) * nb)
    VH_ALLOC(d_bucketBits, sizeof(uint32_t) * ((nb + 31) / 32))
    VH_ALLOC(d_state, sizeof(uint32_t) * VH_STATE_WORDS)
#undef VH_ALLOC
    if (err) {
        vh_hash_data_free(hd);
        return err;
    }
    hd->m_bIsOnGPU = 1;
    return VH_OK;
}

// HashData::free, DSC/VoxelUtilHashSDF.h:148-181
int vh_hash_data_free(VhHashData* hd)
{
    if (!hd) return VH_ERR_BAD_ARGUMENT;
    void* ptrs[] = { hd->d_heap, hd->d_heapCounter, hd->d_hash, hd->d_hashDecision, hd->d_hashDecisionPrefix,
                     hd->d_hashCompactified, hd->d_hashCompactifiedCounter, hd->d_SDFBlocks, hd->d_hashBucketMutex,
                     hd->d_bucketCount, hd->d_bucketBits, hd->d_state };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    std::memset(hd, 0, sizeof(*hd));
    return VH_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------
// mat4f
// ---------------------------------------------------------------------------

namespace vh {

mat4f mat4f::identity()
{
    mat4f r;
    for (int i = 0; i < 16; i++) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    return r;
}

// float4x4::getInverse (DSC/cuda_SimpleMatrixUtil.h:944-1069): cofactors as
// six signed triple products summed in the reference's order, then scaled by
// 1/det.  Host code is compiled with -ffp-contract=off like the kernels.
mat4f mat4f::getInverse() const
{
    static const signed char T[16][6][4] = {
        { {+1,5,10,15}, {-1,5,11,14}, {-1,9,6,15}, {+1,9,7,14}, {+1,13,6,11}, {-1,13,7,10} },
        { {-1,1,10,15}, {+1,1,11,14}, {+1,9,2,15}, {-1,9,3,14}, {-1,13,2,11}, {+1,13,3,10} },
        { {+1,1,6,15}, {-1,1,7,14}, {-1,5,2,15}, {+1,5,3,14}, {+1,13,2,7}, {-1,13,3,6} },
        { {-1,1,6,11}, {+1,1,7,10}, {+1,5,2,11}, {-1,5,3,10}, {-1,9,2,7}, {+1,9,3,6} },
        { {-1,4,10,15}, {+1,4,11,14}, {+1,8,6,15}, {-1,8,7,14}, {-1,12,6,11}, {+1,12,7,10} },
        { {+1,0,10,15}, {-1,0,11,14}, {-1,8,2,15}, {+1,8,3,14}, {+1,12,2,11}, {-1,12,3,10} },
        { {-1,0,6,15}, {+1,0,7,14}, {+1,4,2,15}, {-1,4,3,14}, {-1,12,2,7}, {+1,12,3,6} },
        { {+1,0,6,11}, {-1,0,7,10}, {-1,4,2,11}, {+1,4,3,10}, {+1,8,2,7}, {-1,8,3,6} },
        { {+1,4,9,15}, {-1,4,11,13}, {-1,8,5,15}, {+1,8,7,13}, {+1,12,5,11}, {-1,12,7,9} },
        { {-1,0,9,15}, {+1,0,11,13}, {+1,8,1,15}, {-1,8,3,13}, {-1,12,1,11}, {+1,12,3,9} },
        { {+1,0,5,15}, {-1,0,7,13}, {-1,4,1,15}, {+1,4,3,13}, {+1,12,1,7}, {-1,12,3,5} },
        { {-1,0,5,11}, {+1,0,7,9}, {+1,4,1,11}, {-1,4,3,9}, {-1,8,1,7}, {+1,8,3,5} },
        { {-1,4,9,14}, {+1,4,10,13}, {+1,8,5,14}, {-1,8,6,13}, {-1,12,5,10}, {+1,12,6,9} },
        { {+1,0,9,14}, {-1,0,10,13}, {-1,8,1,14}, {+1,8,2,13}, {+1,12,1,10}, {-1,12,2,9} },
        { {-1,0,5,14}, {+1,0,6,13}, {+1,4,1,14}, {-1,4,2,13}, {-1,12,1,6}, {+1,12,2,5} },
        { {+1,0,5,10}, {-1,0,6,9}, {-1,4,1,10}, {+1,4,2,9}, {+1,8,1,6}, {-1,8,2,5} },
    };
    float inv[16];
    for (int n = 0; n < 16; n++) {
        float acc = 0.0f;
        for (int t = 0; t < 6; t++) {
            const float p = m[T[n][t][1]] * m[T[n][t][2]] * m[T[n][t][3]];
            if (t == 0) acc = (T[n][t][0] > 0) ? p : -p;
            else acc = (T[n][t][0] > 0) ? acc + p : acc - p;
        }
        inv[n] = acc;
    }
    const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    const float detr = 1.0f / det;
    mat4f r;
    for (int n = 0; n < 16; n++) r.m[n] = inv[n] * detr;
    return r;
}

// float4x4::operator*, DSC/cuda_SimpleMatrixUtil.h:861-885
mat4f mat4f::operator*(const mat4f& o) const
{
    mat4f r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[4 * i + j] = m[4 * i + 0] * o.m[j] + m[4 * i + 1] * o.m[4 + j] + m[4 * i + 2] * o.m[8 + j] + m[4 * i + 3] * o.m[12 + j];
    return r;
}

vec3f mat4f::transformPoint(const vec3f& v) const
{
    vec3f r;
    r.x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * 1.0f;
    r.y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * 1.0f;
    r.z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * 1.0f;
    return r;
}

} // namespace vh

// ---------------------------------------------------------------------------
// CUDASceneRepHashSDF
// ---------------------------------------------------------------------------

enum { ST_ALLOC = 0, ST_COMPACTIFY = 1, ST_INTEGRATE = 2, ST_RAYCAST = 0, ST_NORMALS = 1, ST_SPLAT = 2, ST_EMPTY = 3 };

VhSceneOptions CUDASceneRepHashSDF::defaultOptions()
{
    // DSC reference defaults (zParametersDefault.txt:64-67) except GC, which the
    // measurement contract enables
    VhSceneOptions o;
    std::memset(&o, 0, sizeof(o));
    o.s_offlineProcessing = 0;
    o.s_garbageCollectionEnabled = 1;
    o.s_timingsDetailledEnabled = 0;
    o.s_useReferenceLaunchSequence = 0;
    o.s_garbageCollectionStarve = 15;
    o.s_streamingOutParts = 80;
    return o;
}

CUDASceneRepHashSDF::CUDASceneRepHashSDF(const HashParams& params)
    : m_options(defaultOptions()), m_stream(nullptr)
{
    create(params);
}

CUDASceneRepHashSDF::CUDASceneRepHashSDF(const HashParams& params, const VhSceneOptions& options, vhStream_t stream)
    : m_options(options), m_stream(stream)
{
    create(params);
}

CUDASceneRepHashSDF::~CUDASceneRepHashSDF() { destroy(); }

void CUDASceneRepHashSDF::create(const HashParams& params)
{
    m_hashParams = params;
    m_numIntegratedFrames = 0;
    m_lockEpoch = 0;
    h_occupied = nullptr;
    m_occupiedEvent = nullptr;
    m_occupiedPending = false;
    m_counterCleared = false;
    m_timer = new VhStageTimer(3);
    m_aheadPending = 0;
    m_tableEpoch = 0;
    d_packedFrame = nullptr;
    m_packedPixels = 0;
    d_riderDone = nullptr;
    std::memset(m_riderTotals, 0, sizeof(m_riderTotals));
    // (scenes of more blocks in view than this keep the pass over the voxels in a launch of its own.  Up to 2048 blocks the
    // rider has a workgroup for every block and saves 4.8 us a frame at 350 blocks, 5 us at 1200; with more a workgroup takes
    // several in turn, which costs 1.4 us a frame at 2500 - 5800 blocks and saves 2.3 at 8600 (measured).
    // 0 switches the rider off.)
    const char* riderMost = std::getenv("VH_INTEGRATE_RIDER_MAX_BLOCKS");
    m_riderMostBlocks = riderMost ? (unsigned int)std::strtoul(riderMost, nullptr, 10) : 2048u;
    std::memset(&m_job, 0, sizeof(m_job));
    std::memset(&m_hashData, 0, sizeof(m_hashData));
    check(vh_hash_data_alloc(&m_hashData, &m_hashParams), "HashData::allocate");
    // the counters the riders of computeNormals' launch count themselves off on (never reset: the totals are compared)
    checkHip(hipMalloc((void**)&d_riderDone, VH_RIDER_DONE_WORDS * sizeof(uint32_t)), "rider counters");
    checkHip(hipMemsetAsync(d_riderDone, 0, VH_RIDER_DONE_WORDS * sizeof(uint32_t), (hipStream_t)m_stream), "hipMemsetAsync");
    // mapped pinned word the fused integrate kernel mirrors the in-frustum block count into
    checkHip(hipHostMalloc((void**)&h_occupied, 2 * sizeof(uint32_t), hipHostMallocMapped), "hipHostMalloc");
    h_occupied[0] = h_occupied[1] = 0;
    void* dptr = nullptr;
    checkHip(hipHostGetDevicePointer(&dptr, h_occupied, 0), "hipHostGetDevicePointer");
    m_occupiedEvent = dptr; // device alias of h_occupied
    reset();
}

void CUDASceneRepHashSDF::destroy()
{
    (void)hipStreamSynchronize((hipStream_t)m_stream);
    delete m_timer;
    m_timer = nullptr;
    if (d_packedFrame) { (void)hipFree(d_packedFrame); d_packedFrame = nullptr; }
    if (d_riderDone) { (void)hipFree(d_riderDone); d_riderDone = nullptr; }
    if (h_occupied) (void)hipHostFree(h_occupied);
    vh_hash_data_free(&m_hashData);
}

// DSC/CUDASceneRepHashSDF.h:101-109
void CUDASceneRepHashSDF::reset()
{
    m_numIntegratedFrames = 0;
    const vh::mat4f id = vh::mat4f::identity();
    std::memcpy(m_hashParams.m_rigidTransform, id.m, sizeof(id.m));
    std::memcpy(m_hashParams.m_rigidTransformInverse, id.m, sizeof(id.m));
    m_aheadPending = 0;
    m_tableEpoch++;
    m_occupiedPending = true; // reset() waits for the device before it clears the mapped words
    pollOccupiedCount(true);
    h_occupied[0] = h_occupied[1] = 0;
    m_hashParams.m_numOccupiedBlocks = 0;
    m_counterCleared = false;
    m_lockEpoch = 0;
    check(vh_reset(&m_hashData, &m_hashParams, m_stream), "resetCUDA");
}

int32_t CUDASceneRepHashSDF::nextLockToken()
{
    // tokens are positive, so they never equal FREE_ENTRY(-2) / LOCK_ENTRY(-1)
    if (m_lockEpoch == INT32_MAX - 1) {
        check(vh_reset_bucket_mutex(&m_hashData, &m_hashParams, m_stream), "resetHashBucketMutexCUDA");
        m_lockEpoch = 0;
    }
    return ++m_lockEpoch;
}

// DSC/CUDASceneRepHashSDF.h:85-88
void CUDASceneRepHashSDF::setLastRigidTransform(const vh::mat4f& t)
{
    std::memcpy(m_hashParams.m_rigidTransform, t.m, sizeof(t.m));
    const vh::mat4f inv = t.getInverse();
    std::memcpy(m_hashParams.m_rigidTransformInverse, inv.m, sizeof(inv.m));
}

// DSC/CUDASceneRepHashSDF.h:90-93
void CUDASceneRepHashSDF::setLastRigidTransformAndCompactify(const vh::mat4f& t, const DepthCameraParams& cp)
{
    setLastRigidTransform(t);
    compactifyHashEntries(cp);
}

const vh::mat4f CUDASceneRepHashSDF::getLastRigidTransform() const
{
    vh::mat4f r;
    std::memcpy(r.m, m_hashParams.m_rigidTransform, sizeof(r.m));
    return r;
}

void CUDASceneRepHashSDF::pollOccupiedCount(bool block)
{
    if (!m_occupiedPending) return;
    if (block) {
        checkHip(hipStreamSynchronize((hipStream_t)m_stream), "hipStreamSynchronize");
        m_occupiedPending = false;
    }
    // the fused integrate kernel stores the count of its frame into the mapped word; without blocking this is
    // the count of the most recent frame whose kernel has run
    m_hashParams.m_numOccupiedBlocks = ((volatile uint32_t*)h_occupied)[0];
}

const HashParams& CUDASceneRepHashSDF::getHashParams()
{
    pollOccupiedCount(false);
    return m_hashParams;
}

unsigned int CUDASceneRepHashSDF::getNumOccupiedBlocks()
{
    // exact: the device counter of the last compaction
    uint32_t n = 0;
    check(vh_memcpy_d2h(&n, m_hashData.d_hashCompactifiedCounter, sizeof(n), m_stream), "getNumOccupiedBlocks");
    m_hashParams.m_numOccupiedBlocks = n;
    m_occupiedPending = false;
    return n;
}

// DSC/CUDASceneRepHashSDF.h:122-126
unsigned int CUDASceneRepHashSDF::getHeapFreeCount()
{
    unsigned int count = 0;
    check(vh_memcpy_d2h(&count, m_hashData.d_heapCounter, sizeof(count), m_stream), "getHeapFreeCount");
    return count + 1;
}

unsigned int CUDASceneRepHashSDF::getNumFramesStartedOnDevice() const { return ((volatile uint32_t*)h_occupied)[1]; }

// the alloc + compactify passes of the frame whose pose has just been set, as a job
void CUDASceneRepHashSDF::prepareJob(const DepthCameraData& cam, const DepthCameraParams& cp, const unsigned int* d_bitMask)
{
    const size_t pixels = (size_t)cp.m_imageWidth * cp.m_imageHeight;
    if (pixels > m_packedPixels) { // the packed frame follows the image size (first frame, or a larger adapter image)
        if (d_packedFrame) {
            checkHip(hipStreamSynchronize((hipStream_t)m_stream), "hipStreamSynchronize");
            checkHip(hipFree(d_packedFrame), "hipFree");
            d_packedFrame = nullptr;
        }
        checkHip(hipMalloc(&d_packedFrame, 8 * pixels), "packed frame");
        m_packedPixels = pixels;
    }
    std::memset(&m_job, 0, sizeof(m_job));
    m_job.hashData = m_hashData;
    m_job.hashParams = m_hashParams;
    m_job.cam = cam;
    m_job.cp = cp;
    m_job.d_bitMask = d_bitMask;
    m_job.d_packedFrame = d_packedFrame;
    m_job.lockToken = nextLockToken();
    m_job.frameNumber = m_numIntegratedFrames;
    m_job.tableEpoch = m_tableEpoch;
    m_job.d_riderDone = d_riderDone;
    m_job.listDoneTotal = m_riderTotals[0]; m_job.listClassTotal = m_riderTotals[1];
}

uint32_t CUDASceneRepHashSDF::fusedFlags() const
{
    uint32_t flags = 0;
    if (m_options.s_garbageCollectionEnabled) {
        flags |= VH_FUSED_GC;
        if (m_numIntegratedFrames > 0 && m_options.s_garbageCollectionStarve != 0 &&
            m_numIntegratedFrames % m_options.s_garbageCollectionStarve == 0)
            flags |= VH_FUSED_STARVE;
    }
    return flags;
}

// DSC/CUDASceneRepHashSDF.h:64-83
void CUDASceneRepHashSDF::integrate(const vh::mat4f& lastRigidTransform, const DepthCameraData& cam,
                                    const DepthCameraParams& cp, const unsigned int* d_bitMask)
{
    if (m_aheadPending) throw vh::Error(VH_ERR_BAD_ARGUMENT, "integrate(): integrateAhead() is waiting for its integrateFinish()");
    setLastRigidTransform(lastRigidTransform);
    alloc(cam, cp, d_bitMask, false);
    compactifyHashEntries(cp);
    if (m_options.s_useReferenceLaunchSequence) {
        integrateDepthMap(cam, cp);
        garbageCollect(cp);
    } else {
        integrateFused(cam, cp);
    }
    m_numIntegratedFrames++;
}

// integrate -> [starve] -> identify -> free in one pass over the voxels
void CUDASceneRepHashSDF::integrateFused(const DepthCameraData& cam, const DepthCameraParams& cp)
{
    const uint32_t flags = fusedFlags();
    const bool timed = m_options.s_timingsDetailledEnabled;
    if (timed) m_timer->arm(ST_INTEGRATE); // (the kernel's own dispatch time stamps)
    // the packed frame is this frame's only if this frame's alloc pass wrote it (same image, same size)
    const bool packedIsCurrent = m_job.allocLaunched && m_job.d_packedFrame && m_job.cam.d_depthData == cam.d_depthData &&
                                 m_job.cam.d_colorData == cam.d_colorData && m_job.cp.m_imageWidth == cp.m_imageWidth &&
                                 m_job.cp.m_imageHeight == cp.m_imageHeight;
    check(vh_integrate_fused(&m_hashData, &m_hashParams, &cam, &cp, flags, nextLockToken(), (uint32_t*)m_occupiedEvent,
                             m_numIntegratedFrames + 1u, packedIsCurrent ? m_job.d_packedFrame : nullptr, m_stream), "integrate (fused)");
    m_occupiedPending = true;
}

// first half of integrate(): the pose, and alloc + compactify as a job for a co-launch (see include/vh.hpp)
VhFrameJob* CUDASceneRepHashSDF::integrateAhead(const vh::mat4f& lastRigidTransform, const DepthCameraData& cam,
                                                const DepthCameraParams& cp, const unsigned int* d_bitMask)
{
    if (m_aheadPending) throw vh::Error(VH_ERR_BAD_ARGUMENT, "integrateAhead(): the previous one has not been finished");
    setLastRigidTransform(lastRigidTransform);
    prepareJob(cam, cp, d_bitMask);
    m_aheadPending = 1;
    // offline mode loops over alloc with read-backs, the reference sequence reads the compactify count: not for a co-launch
    if (m_options.s_offlineProcessing || m_options.s_useReferenceLaunchSequence || m_options.s_timingsDetailledEnabled) return nullptr;
    // the pass over the voxels may ride behind compactify in computeNormals' launch (vh_compute_normals_co2 decides)
    pollOccupiedCount(false);
    if (m_riderMostBlocks != 0u && m_hashParams.m_numOccupiedBlocks <= m_riderMostBlocks) {
        m_job.fusedFlags = fusedFlags();
        m_job.fusedLockToken = nextLockToken();
        m_job.d_countMirror = (uint32_t*)m_occupiedEvent;
        m_job.mirrorTag = m_numIntegratedFrames + 1u;
        m_job.fusedPrepared = 1;
    }
    return &m_job;
}

void CUDASceneRepHashSDF::keepRiderTotals()
{
    m_riderTotals[0] = m_job.listDoneTotal; m_riderTotals[1] = m_job.listClassTotal;
}

void CUDASceneRepHashSDF::abortAhead()
{
    if (m_aheadPending && m_job.fusedLaunched) { // (the frame's pass over the voxels is in the queue already: the frame counts)
        keepRiderTotals();
        m_counterCleared = false;
        m_occupiedPending = true;
        m_numIntegratedFrames++;
    }
    m_aheadPending = 0;
}

void CUDASceneRepHashSDF::integrateFinish(const DepthCameraData& cam, const DepthCameraParams& cp)
{
    if (!m_aheadPending) throw vh::Error(VH_ERR_BAD_ARGUMENT, "integrateFinish() without integrateAhead()");
    m_aheadPending = 0;
    if (m_job.allocLaunched) m_counterCleared = true;
    else alloc(cam, cp, m_job.d_bitMask, true);
    if (m_job.compactifyLaunched) m_counterCleared = false;
    else compactifyHashEntries(cp);
    keepRiderTotals();
    if (m_job.fusedLaunched) {
        m_occupiedPending = true; // (it rode in computeNormals' launch)
    } else if (m_options.s_useReferenceLaunchSequence) {
        integrateDepthMap(cam, cp);
        garbageCollect(cp);
    } else {
        integrateFused(cam, cp);
    }
    m_numIntegratedFrames++;
}

// DSC/CUDASceneRepHashSDF.h:247-279
void CUDASceneRepHashSDF::alloc(const DepthCameraData& cam, const DepthCameraParams& cp, const unsigned int* d_bitMask, bool jobPrepared)
{
    const bool timed = m_options.s_timingsDetailledEnabled;
    if (timed) m_timer->start(ST_ALLOC, (hipStream_t)m_stream);
    if (!jobPrepared) prepareJob(cam, cp, d_bitMask);
    if (m_options.s_offlineProcessing) {
        // allocate until all blocks are allocated (one blocking read-back per pass, as the reference)
        unsigned int prevFree = getHeapFreeCount();
        while (true) {
            m_job.allocLaunched = 0;
            check(vh_alloc_job(&m_job, m_stream), "allocCUDA");
            unsigned int currFree = getHeapFreeCount();
            if (prevFree != currFree) prevFree = currFree;
            else break;
            m_job.lockToken = nextLockToken();
        }
    } else {
        check(vh_alloc_job(&m_job, m_stream), "allocCUDA");
    }
    m_counterCleared = true; // k_alloc clears d_hashCompactifiedCounter
    if (timed) m_timer->stop(ST_ALLOC, (hipStream_t)m_stream);
}

// DSC/CUDASceneRepHashSDF.h:282-315
void CUDASceneRepHashSDF::compactifyHashEntries(const DepthCameraParams& cp)
{
    vhStream_t stream = m_stream;
    const bool timed = m_options.s_timingsDetailledEnabled;
    if (timed) m_timer->start(ST_COMPACTIFY, (hipStream_t)stream);
    const bool needHostCount = m_options.s_offlineProcessing || m_options.s_useReferenceLaunchSequence;
    // alloc() leaves the counter cleared; the stand-alone path (setLastRigidTransformAndCompactify) must clear it
    const uint32_t flags = m_counterCleared ? VH_COMPACT_COUNTER_IS_ZERO : 0u;
    m_counterCleared = false;
    if (needHostCount) {
        uint32_t n = 0;
        check(vh_compactify(&m_hashData, &m_hashParams, &cp, &n, flags, stream), "compactifyHashAllInOneCUDA");
        m_hashParams.m_numOccupiedBlocks = n;
        m_occupiedPending = false;
    } else {
        check(vh_compactify(&m_hashData, &m_hashParams, &cp, nullptr, flags, stream), "compactifyHashAllInOneCUDA");
    }
    if (timed) m_timer->stop(ST_COMPACTIFY, (hipStream_t)stream);
}

// DSC/CUDASceneRepHashSDF.h:317-325
void CUDASceneRepHashSDF::integrateDepthMap(const DepthCameraData& cam, const DepthCameraParams& cp)
{
    const bool timed = m_options.s_timingsDetailledEnabled;
    if (timed) m_timer->start(ST_INTEGRATE, (hipStream_t)m_stream);
    check(vh_integrate(&m_hashData, &m_hashParams, &cam, &cp, m_stream), "integrateDepthMapCUDA");
    if (timed) m_timer->stop(ST_INTEGRATE, (hipStream_t)m_stream);
}

// DSC/CUDASceneRepHashSDF.h:327-339
void CUDASceneRepHashSDF::garbageCollect(const DepthCameraParams& cp)
{
    if (!m_options.s_garbageCollectionEnabled) return;
    if (m_numIntegratedFrames > 0 && m_options.s_garbageCollectionStarve != 0 &&
        m_numIntegratedFrames % m_options.s_garbageCollectionStarve == 0) {
        check(vh_starve(&m_hashData, &m_hashParams, m_stream), "starveVoxelsKernelCUDA");
    }
    check(vh_gc_identify(&m_hashData, &m_hashParams, &cp, m_stream), "garbageCollectIdentifyCUDA");
    check(vh_reset_bucket_mutex(&m_hashData, &m_hashParams, m_stream), "resetHashBucketMutexCUDA");
    m_lockEpoch = 0; // the array was reset: epochs may restart
    check(vh_gc_free(&m_hashData, &m_hashParams, nextLockToken(), m_stream), "garbageCollectFreeCUDA");
}

void CUDASceneRepHashSDF::getState(uint32_t out[VH_STATE_WORDS])
{
    check(vh_memcpy_d2h(out, m_hashData.d_state, sizeof(uint32_t) * VH_STATE_WORDS, m_stream), "getState");
}

void CUDASceneRepHashSDF::getTimings(double out[4])
{
    m_timer->resolve((hipStream_t)m_stream);
    out[0] = m_timer->totalMs[ST_ALLOC];
    out[1] = m_timer->totalMs[ST_COMPACTIFY];
    out[2] = m_timer->totalMs[ST_INTEGRATE];
    out[3] = (double)m_timer->count[ST_INTEGRATE];
}

// DSC/CUDASceneRepHashSDF.h:129-233
void CUDASceneRepHashSDF::debugHash(unsigned int report[4])
{
    const size_t ne = (size_t)m_hashParams.m_hashBucketSize * m_hashParams.m_hashNumBuckets;
    const unsigned int nblk = m_hashParams.m_numSDFBlocks;
    std::vector<HashEntry> hashCPU(ne);
    std::vector<unsigned int> heapCPU(nblk);
    unsigned int heapCounterCPU = 0;
    check(vh_memcpy_d2h(&heapCounterCPU, m_hashData.d_heapCounter, sizeof(unsigned int), m_stream), "debugHash");
    heapCounterCPU++; // points to the first free entry: number of blocks is one more
    check(vh_memcpy_d2h(heapCPU.data(), m_hashData.d_heap, sizeof(unsigned int) * nblk, m_stream), "debugHash");
    check(vh_memcpy_d2h(hashCPU.data(), m_hashData.d_hash, sizeof(HashEntry) * ne, m_stream), "debugHash");

    if (heapCounterCPU > nblk) throw vh::Error(VH_ERR_HEAP_EXHAUSTED, "ERROR: heap counter out of range");
    std::vector<int> pointersFreeVec(nblk, 0);
    for (unsigned int i = 0; i < heapCounterCPU; i++) {
        if (heapCPU[i] >= nblk) throw vh::Error(VH_ERR_BAD_ARGUMENT, "ERROR: heap holds an out-of-range block id");
        if (pointersFreeVec[heapCPU[i]] == VH_FREE_ENTRY) throw vh::Error(VH_ERR_BAD_ARGUMENT, "ERROR: duplicate free pointers in heap array");
        pointersFreeVec[heapCPU[i]] = VH_FREE_ENTRY;
    }

    struct PosHash {
        size_t operator()(const std::array<int, 3>& v) const
        {
            return ((size_t)v[0] * 73856093u) ^ ((size_t)v[1] * 19349669u) ^ ((size_t)v[2] * 83492791u);
        }
    };
    std::unordered_set<std::array<int, 3>, PosHash> seen;
    unsigned int numOccupied = 0, numMinusOne = 0, duplicates = 0;
    for (size_t i = 0; i < ne; i++) {
        if (hashCPU[i].ptr == VH_LOCK_ENTRY) { numMinusOne++; continue; }
        if (hashCPU[i].ptr != VH_FREE_ENTRY) {
            numOccupied++;
            if (!seen.insert({ hashCPU[i].pos[0], hashCPU[i].pos[1], hashCPU[i].pos[2] }).second) duplicates++;
            const unsigned int blk = (unsigned int)hashCPU[i].ptr / VH_SDF_BLOCK_VOXELS;
            if (blk >= nblk) throw vh::Error(VH_ERR_BAD_ARGUMENT, "ERROR: entry points outside the SDF block array");
            if (pointersFreeVec[blk] == VH_FREE_ENTRY)
                throw vh::Error(VH_ERR_BAD_ARGUMENT, "ERROR: ptr is on free heap, but also marked as an allocated entry");
            if (pointersFreeVec[blk] == VH_LOCK_ENTRY)
                throw vh::Error(VH_ERR_BAD_ARGUMENT, "ERROR: two entries share one SDF block");
            pointersFreeVec[blk] = VH_LOCK_ENTRY;
        }
    }
    unsigned int numHeapFree = 0, numHeapOccupied = 0;
    for (unsigned int i = 0; i < nblk; i++) {
        if (pointersFreeVec[i] == VH_FREE_ENTRY) numHeapFree++;
        else if (pointersFreeVec[i] == VH_LOCK_ENTRY) numHeapOccupied++;
        else throw vh::Error(VH_ERR_BAD_ARGUMENT, "memory leak detected: neither free nor allocated");
    }
    if (numHeapFree + numHeapOccupied != nblk) throw vh::Error(VH_ERR_BAD_ARGUMENT, "HEAP CORRUPTED");
    if (report) {
        report[0] = numOccupied;
        report[1] = heapCounterCPU;
        report[2] = duplicates;
        report[3] = numMinusOne;
    }
    if (duplicates) throw vh::Error(VH_ERR_BAD_ARGUMENT, "ERROR: duplicate block positions in hash");
}

// ---------------------------------------------------------------------------
// CUDARayCastSDF
// ---------------------------------------------------------------------------

CUDARayCastSDF::CUDARayCastSDF(const RayCastParams& params, vhStream_t stream)
    : m_params(params), m_stream(stream), m_timer(nullptr), m_timeMarchOnly(false), m_timeStride(1), m_renderCalls(0)
{
    std::memset(&m_data, 0, sizeof(m_data));
    const size_t n = (size_t)params.m_width * params.m_height;
    // RayCastData::allocate, DSC/RayCastSDFUtil.h:56-61
    checkHip(hipMalloc((void**)&m_data.d_depth, sizeof(float) * n), "RayCastData::allocate");
    checkHip(hipMalloc((void**)&m_data.d_depth4, sizeof(float) * 4 * n), "RayCastData::allocate");
    checkHip(hipMalloc((void**)&m_data.d_normals, sizeof(float) * 4 * n), "RayCastData::allocate");
    checkHip(hipMalloc((void**)&m_data.d_colors, sizeof(float) * 4 * n), "RayCastData::allocate");
    d_tileHeads = nullptr;
    d_tileBlocks = nullptr;
    m_useIntervals = true;
    const size_t tiles = (size_t)((params.m_width + 7) / 8) * ((params.m_height + 7) / 8);
    checkHip(hipMalloc((void**)&d_tileHeads, sizeof(uint32_t) * 4 * (tiles ? tiles : 1)), "tile heads");
    checkHip(hipMalloc((void**)&d_tileBlocks, sizeof(VhTileBlock) * VH_TILE_LIST_CAPACITY_LARGE * (tiles ? tiles : 1)), "tile block lists");
    h_longestList = nullptr;
    d_longestList = nullptr;
    m_largeTables = false;
    m_quietFrames = 0;
    checkHip(hipHostMalloc((void**)&h_longestList, sizeof(uint32_t), hipHostMallocMapped), "hipHostMalloc");
    *h_longestList = 0;
    checkHip(hipHostGetDevicePointer((void**)&d_longestList, h_longestList, 0), "hipHostGetDevicePointer");
    check(vh_ray_interval_clear(d_tileHeads, params.m_width, params.m_height, m_stream), "vh_ray_interval_clear");
    d_schedule = nullptr;
    m_phase = 0;
    m_preSplatsUsed = 0;
    std::memset(&m_preSplat, 0, sizeof(m_preSplat));
    const size_t schedBytes = vh_render_schedule_bytes(params.m_width, params.m_height);
    checkHip(hipMalloc((void**)&d_schedule, schedBytes), "render schedule");
    checkHip(hipMemsetAsync(d_schedule, 0, schedBytes, (hipStream_t)m_stream), "render schedule");
}

CUDARayCastSDF::~CUDARayCastSDF()
{
    (void)hipStreamSynchronize((hipStream_t)m_stream);
    delete m_timer;
    if (m_data.d_depth) (void)hipFree(m_data.d_depth);
    if (m_data.d_depth4) (void)hipFree(m_data.d_depth4);
    if (m_data.d_normals) (void)hipFree(m_data.d_normals);
    if (m_data.d_colors) (void)hipFree(m_data.d_colors);
    if (d_tileHeads) (void)hipFree(d_tileHeads);
    if (d_tileBlocks) (void)hipFree(d_tileBlocks);
    if (d_schedule) (void)hipFree(d_schedule);
    if (h_longestList) (void)hipHostFree(h_longestList);
}

void CUDARayCastSDF::setTiming(bool on, bool marchOnly, unsigned int stride)
{
    m_timeMarchOnly = marchOnly;
    m_timeStride = stride ? stride : 1u;
    if (on && !m_timer) m_timer = new VhStageTimer(4);
    if (!on && m_timer) { delete m_timer; m_timer = nullptr; }
}

double CUDARayCastSDF::getEventPairOverheadMs()
{
    if (!m_timer) return 0.0;
    m_timer->resolve((hipStream_t)m_stream);
    return m_timer->count[ST_EMPTY] ? m_timer->totalMs[ST_EMPTY] / (double)m_timer->count[ST_EMPTY] : 0.0;
}

void CUDARayCastSDF::getTimings(double out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0.0;
    if (!m_timer) return;
    m_timer->resolve((hipStream_t)m_stream);
    out[0] = m_timer->totalMs[ST_RAYCAST];
    out[1] = m_timer->totalMs[ST_NORMALS];
    out[2] = (double)m_timer->count[ST_RAYCAST];
    out[3] = m_timer->totalMs[ST_SPLAT];
}

// DSC/CUDARayCastSDF.cpp:38-72 with rayIntervalSplatting :84-100 (view matrices only)
void CUDARayCastSDF::render(const HashData& hashData, const HashParams& hashParams, const DepthCameraParams& cp,
                            const vh::mat4f& lastRigidTransform, VhFrameJob* coLaunch)
{
    m_params.m_numOccupiedSDFBlocks = hashParams.m_numOccupiedBlocks;
    const vh::mat4f view = lastRigidTransform.getInverse();
    std::memcpy(m_params.m_viewMatrix, view.m, sizeof(view.m));
    std::memcpy(m_params.m_viewMatrixInverse, lastRigidTransform.m, sizeof(lastRigidTransform.m));

    // an event record idles the queue for a few microseconds: with a stride only every n-th call is timed
    const bool timed = m_timer && (m_renderCalls++ % m_timeStride) == 0;
    const bool timedAll = timed && !m_timeMarchOnly;
    // Was this render's splat made ahead, inside the previous render's computeNormals launch?  Only if it is for this
    // very pose and table, the scene has integrated exactly the one frame it was made before, and nothing else has
    // edited the table since.
    const bool preSplatted = m_useIntervals && m_preSplat.valid && coLaunch && m_preSplat.table == (const void*)hashData.d_hash &&
                             coLaunch->frameNumber == m_preSplat.frameNumber + 1u && coLaunch->tableEpoch == m_preSplat.tableEpoch &&
                             std::memcmp(m_preSplat.pose, lastRigidTransform.m, sizeof(m_preSplat.pose)) == 0;
    if (m_preSplat.valid && !preSplatted) {
        // a splat made ahead that this call cannot use: its tile heads and lists must not leak into the fresh one
        check(vh_ray_interval_clear(d_tileHeads, m_params.m_width, m_params.m_height, m_stream), "vh_ray_interval_clear");
        m_phase = m_preSplat.phase;
    }
    m_preSplat.valid = false;
    if (preSplatted) {
        m_preSplatsUsed++;
        m_phase = m_preSplat.phase;
        m_tileCapacity = m_preSplat.capacity;
    } else if (m_useIntervals) {
        if (timedAll) m_timer->start(ST_SPLAT, (hipStream_t)m_stream);
        ++m_phase;
        // Table size for this frame, from what the ray caster reported two frames ago (mapped host word, no
        // synchronisation): large tables as soon as a list outgrew the small ones, back after 30 frames without one.
        const uint32_t longest = *(volatile uint32_t*)h_longestList;
        if (longest > (uint32_t)VH_TILE_LIST_CAPACITY) { m_largeTables = true; m_quietFrames = 0; }
        else if (m_largeTables && ++m_quietFrames > 30) m_largeTables = false;
        m_tileCapacity = m_largeTables ? VH_TILE_LIST_CAPACITY_LARGE : VH_TILE_LIST_CAPACITY;
        check(vh_ray_interval_splat(&hashData, &hashParams, &cp, &m_params, d_tileHeads, d_tileBlocks, m_tileCapacity, d_schedule, m_phase, d_longestList, m_stream), "rayIntervalSplatCUDA");
        if (timedAll) m_timer->stop(ST_SPLAT, (hipStream_t)m_stream);
    }
    if (timedAll) { // what an event pair reads with nothing between its two records: the share of a bracketed launch that is not the kernel
        m_timer->start(ST_EMPTY, (hipStream_t)m_stream);
        m_timer->stop(ST_EMPTY, (hipStream_t)m_stream);
    }
    if (timed && m_useIntervals) m_timer->arm(ST_RAYCAST); // the march kernel alone, by its own dispatch time stamps
    else if (timed) m_timer->start(ST_RAYCAST, (hipStream_t)m_stream);
    // Without gradients computeNormals rewrites every pixel of the normal map right after (MINF or a normal,
    // DSC/CameraUtil.cu:669-697), so the march does not write its MINF there first: a null map is skipped.
    RayCastData out = m_data;
#ifndef VH_RENDER_STATS // (the instrumented build of scratch/ writes its counters into the maps)
    if (!m_params.m_useGradients) out.d_normals = nullptr;
#endif
    if (m_useIntervals) {
        check(vh_render_intervals_co(&hashData, &hashParams, &out, &cp, &m_params, d_tileHeads, d_tileBlocks, m_tileCapacity, d_schedule, m_phase, coLaunch, m_stream), "renderCS");
    } else {
        check(vh_render(&hashData, &hashParams, &out, &cp, &m_params, m_stream), "renderCS");
    }
    if (timed && !m_useIntervals) m_timer->stop(ST_RAYCAST, (hipStream_t)m_stream);
    if (!m_params.m_useGradients) {
        if (timedAll) m_timer->arm(ST_NORMALS);
        // With a job riding along, the NEXT render's splat rides too: that render's pose is the job's (the pose of
        // the frame being integrated), and the table it will see differs from the one listed here only by what the
        // frame's pass over the voxels frees and what the next alloc adds (vh_compute_normals_co2).
        const bool ahead = m_useIntervals && coLaunch && coLaunch->allocLaunched && !coLaunch->compactifyLaunched &&
                           coLaunch->hashData.d_hash == hashData.d_hash;
        if (ahead) {
            RayCastParams next = m_params;
            std::memcpy(next.m_viewMatrix, coLaunch->hashParams.m_rigidTransformInverse, sizeof(next.m_viewMatrix));
            std::memcpy(next.m_viewMatrixInverse, coLaunch->hashParams.m_rigidTransform, sizeof(next.m_viewMatrixInverse));
            const uint32_t longest = *(volatile uint32_t*)h_longestList;
            if (longest > (uint32_t)VH_TILE_LIST_CAPACITY) { m_largeTables = true; m_quietFrames = 0; }
            else if (m_largeTables && ++m_quietFrames > 30) m_largeTables = false;
            const uint32_t capacity = m_largeTables ? VH_TILE_LIST_CAPACITY_LARGE : VH_TILE_LIST_CAPACITY;
            check(vh_compute_normals_co2(m_data.d_normals, m_data.d_depth4, m_params.m_width, m_params.m_height, coLaunch, &next, d_tileHeads, d_tileBlocks,
                                         capacity, d_schedule, m_phase + 1u, d_longestList, m_stream), "computeNormals");
            m_preSplat.valid = true;
            std::memcpy(m_preSplat.pose, coLaunch->hashParams.m_rigidTransform, sizeof(m_preSplat.pose));
            m_preSplat.table = (const void*)hashData.d_hash;
            m_preSplat.frameNumber = coLaunch->frameNumber;
            m_preSplat.tableEpoch = coLaunch->tableEpoch;
            m_preSplat.phase = m_phase + 1u;
            m_preSplat.capacity = capacity;
        } else {
            check(vh_compute_normals_co(m_data.d_normals, m_data.d_depth4, m_params.m_width, m_params.m_height, coLaunch, m_stream), "computeNormals");
        }
    }
}
