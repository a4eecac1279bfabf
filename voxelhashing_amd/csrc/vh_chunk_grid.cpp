// vh_chunk_grid.cpp -- CUDASceneRepChunkGrid: GPU <-> host streaming of SDF
// blocks that leave / enter a sphere around the camera, the host chunk grid,
// its bit mask and the .hashgrid file format.
//
// Behavioural contract: DSC/CUDASceneRepChunkGrid.{h,cpp}, DSC/BitArray.h
// (DSC/ = /root/reference/DepthSensingCUDA/Source/).  Re-designed host side:
//   - std::thread + condition variables replace the Win32 thread/event pairs;
//   - the chunk grid is a sparse map (the reference keeps 257^3 pointers);
//   - the worker's host->device copies run on their own HIP stream from
//     pinned staging; the bit mask is uploaded only when it changed (the
//     reference re-uploads all of it every frame, DSC/CUDASceneRepChunkGrid.h:306-309);
//   - bucket locks use the scene's lock epoch instead of a mutex-array reset.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <unordered_set>

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
inline int signi(float v) { return (0.0f < v) - (v < 0.0f); }
inline float length3(float x, float y, float z) { return std::sqrt(x * x + y * y + z * z); }

} // namespace

// ---------------------------------------------------------------------------
// events
// ---------------------------------------------------------------------------

void CUDASceneRepChunkGrid::AutoResetEvent::set()
{
    {
        std::lock_guard<std::mutex> l(mtx);
        signaled = true;
    }
    cv.notify_one();
}
void CUDASceneRepChunkGrid::AutoResetEvent::wait()
{
    // Frames come every 100-250 us and each hands over to the other thread four times: a condition variable's wake-up
    // (tens of microseconds) would be most of the frame's slack.  So the waiter looks at the flag for a frame or two first
    // and only then goes to sleep.
    const auto t0 = std::chrono::steady_clock::now();
    unsigned int spins = 0;
    while (!signaled.load(std::memory_order_acquire)) {
        if ((++spins & 0xffu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
    }
    std::unique_lock<std::mutex> l(mtx);
    cv.wait(l, [this] { return signaled.load(std::memory_order_acquire); });
    signaled.store(false, std::memory_order_release);
}
void CUDASceneRepChunkGrid::AutoResetEvent::reset(bool state)
{
    std::lock_guard<std::mutex> l(mtx);
    signaled = state;
}

// ---------------------------------------------------------------------------
// construction
// ---------------------------------------------------------------------------

CUDASceneRepChunkGrid::CUDASceneRepChunkGrid(CUDASceneRepHashSDF* sceneRepHashSDF, const vh::vec3f& voxelExtends,
                                             const vh::vec3i& gridDimensions, const vh::vec3i& minGridPos,
                                             unsigned int initialChunkListSize, bool streamingEnabled,
                                             unsigned int streamOutParts)
{
    m_sceneRepHashSDF = sceneRepHashSDF;
    m_currentPart = 0;
    m_numFailedInserts = 0;
    m_streamOutParts = streamOutParts ? streamOutParts : 1;
    m_maxNumberOfSDFBlocksIntegrateFromGlobalHash = 100000; // DSC/CUDASceneRepChunkGrid.h:162
    h_SDFBlockDescOutput = nullptr; h_SDFBlockOutput = nullptr;
    h_SDFBlockDescInput = nullptr; h_SDFBlockInput = nullptr; h_counter = nullptr;
    h_mirror = nullptr; d_mirror = nullptr; m_mirrorTag = 0;
    h_probe = nullptr; d_probe = nullptr; m_probeTag = 0; d_probeCounter = nullptr;
    d_SDFBlockDescOutput = nullptr; d_SDFBlockDescInput = nullptr;
    d_SDFBlockOutput = nullptr; d_SDFBlockInput = nullptr;
    d_SDFBlockCounter = nullptr; d_insertFailed = nullptr; d_bitMask = nullptr; m_copyStream = nullptr;
    s_terminateThread = true; // by default the thread is disabled
    s_nStreamdInBlocks = 0; s_nStreamdOutBlocks = 0;
    s_posCamera = { 0.0f, 0.0f, 0.0f };
    s_radius = 0.0f;
    m_bitMaskDirty = true;
    m_plStarted = false; m_plQuit = false;
    m_plPosted = 0; m_plDone = 0; m_plError = 0;
    m_plDecision = { 0u, 0xffffffffu, 0 };
    m_plDecisionValid = false;
    m_plDecisionPos = { 0.0f, 0.0f, 0.0f };
    m_plDecisionRadius = 0.0f;
    m_plFrame = 0; m_plOutThisFrame = false; m_plOutTag = 0; m_plOutMost = 0;
    for (int i = 0; i < 2; i++) { m_plInsert[i].pending = false; m_plInsert[i].tag = 0; m_plInsert[i].nIn = 0; m_plInsert[i].chunkBit = 0xffffffffu; h_plInMirror[i] = nullptr; hd_plInMirror[i] = nullptr; }
    m_plBlocksOut = 0; m_plBlocksIn = 0; m_plTag = 0;
    for (int i = 0; i < 2; i++) {
        d_plOutDesc[i] = nullptr; h_plOutDesc[i] = nullptr; h_plOutBlocks[i] = nullptr; hd_plOutDesc[i] = nullptr; hd_plOutBlocks[i] = nullptr;
        h_plOutMirror[i] = nullptr; hd_plOutMirror[i] = nullptr;
        h_plInDesc[i] = nullptr; h_plInBlocks[i] = nullptr; d_plInDesc[i] = nullptr; d_plInBlocks[i] = nullptr;
    }
    create(voxelExtends, gridDimensions, minGridPos, initialChunkListSize, streamingEnabled);
}

CUDASceneRepChunkGrid::~CUDASceneRepChunkGrid() { destroy(); }

// DSC/CUDASceneRepChunkGrid.h:366-393
void CUDASceneRepChunkGrid::create(const vh::vec3f& voxelExtends, const vh::vec3i& gridDimensions, const vh::vec3i& minGridPos,
                                   unsigned int initialChunkListSize, bool streamingEnabled)
{
    m_voxelExtents = voxelExtends;
    m_gridDimensions = gridDimensions;
    m_initialChunkDescListSize = initialChunkListSize;
    m_minGridPos = minGridPos;
    m_maxGridPos = { minGridPos.x + gridDimensions.x, minGridPos.y + gridDimensions.y, minGridPos.z + gridDimensions.z };

    const size_t nBits = (size_t)gridDimensions.x * gridDimensions.y * gridDimensions.z;
    m_bitMask.assign((nBits + 31) / 32, 0u);

    const size_t n = m_maxNumberOfSDFBlocksIntegrateFromGlobalHash;
    checkHip(hipHostMalloc((void**)&h_SDFBlockDescOutput, sizeof(SDFBlockDesc) * n, hipHostMallocDefault), "hipHostMalloc");
    checkHip(hipHostMalloc((void**)&h_SDFBlockOutput, sizeof(vh::SDFBlock) * n, hipHostMallocDefault), "hipHostMalloc");
    checkHip(hipHostMalloc((void**)&h_SDFBlockDescInput, sizeof(SDFBlockDesc) * n, hipHostMallocDefault), "hipHostMalloc");
    checkHip(hipHostMalloc((void**)&h_SDFBlockInput, sizeof(vh::SDFBlock) * n, hipHostMallocDefault), "hipHostMalloc");
    checkHip(hipHostMalloc((void**)&h_counter, sizeof(uint32_t) * 2, hipHostMallocDefault), "hipHostMalloc");
    checkHip(hipHostMalloc((void**)&h_mirror, sizeof(uint32_t) * 4, hipHostMallocMapped), "hipHostMalloc");
    h_mirror[0] = h_mirror[1] = h_mirror[2] = h_mirror[3] = 0u;
    checkHip(hipHostGetDevicePointer((void**)&d_mirror, h_mirror, 0), "hipHostGetDevicePointer");
    checkHip(hipHostMalloc((void**)&h_probe, sizeof(uint32_t) * 4, hipHostMallocMapped), "hipHostMalloc");
    h_probe[0] = h_probe[1] = h_probe[2] = h_probe[3] = 0u;
    checkHip(hipHostGetDevicePointer((void**)&d_probe, h_probe, 0), "hipHostGetDevicePointer");
    checkHip(hipMalloc((void**)&d_probeCounter, sizeof(unsigned int)), "hipMalloc");
    checkHip(hipMemset(d_probeCounter, 0, sizeof(unsigned int)), "hipMemset");
    checkHip(hipMalloc((void**)&d_SDFBlockDescOutput, sizeof(SDFBlockDesc) * n), "hipMalloc");
    checkHip(hipMalloc((void**)&d_SDFBlockDescInput, sizeof(SDFBlockDesc) * n), "hipMalloc");
    checkHip(hipMalloc((void**)&d_SDFBlockOutput, sizeof(vh::SDFBlock) * n), "hipMalloc");
    checkHip(hipMalloc((void**)&d_SDFBlockInput, sizeof(vh::SDFBlock) * n), "hipMalloc");
    checkHip(hipMalloc((void**)&d_SDFBlockCounter, sizeof(unsigned int)), "hipMalloc");
    checkHip(hipMalloc((void**)&d_insertFailed, sizeof(unsigned int) * (1 + (size_t)m_maxNumberOfSDFBlocksIntegrateFromGlobalHash)), "hipMalloc");
    checkHip(hipMalloc((void**)&d_bitMask, sizeof(unsigned int) * m_bitMask.size()), "hipMalloc");
    checkHip(hipGetDevice(&m_device), "hipGetDevice"); // one instance is bound to one device
    hipStream_t cs;
    checkHip(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking), "hipStreamCreate");
    m_copyStream = cs;

    if (streamingEnabled) startMultiThreading();
}

void CUDASceneRepChunkGrid::destroy()
{
    try { pipelineDrain(); } catch (...) {}
    pipelineStop();
    stopMultiThreading();
    clearGrid();
    if (m_sceneRepHashSDF) (void)hipStreamSynchronize((hipStream_t)m_sceneRepHashSDF->getStream());
    if (m_copyStream) { (void)hipStreamSynchronize((hipStream_t)m_copyStream); (void)hipStreamDestroy((hipStream_t)m_copyStream); }
    (void)hipHostFree(h_SDFBlockDescOutput); (void)hipHostFree(h_SDFBlockOutput);
    (void)hipHostFree(h_SDFBlockDescInput); (void)hipHostFree(h_SDFBlockInput); (void)hipHostFree(h_counter); (void)hipHostFree(h_mirror); (void)hipHostFree(h_probe); (void)hipFree(d_probeCounter);
    (void)hipFree(d_SDFBlockDescOutput); (void)hipFree(d_SDFBlockDescInput);
    (void)hipFree(d_SDFBlockOutput); (void)hipFree(d_SDFBlockInput);
    (void)hipFree(d_SDFBlockCounter); (void)hipFree(d_insertFailed); (void)hipFree(d_bitMask);
}

// ---------------------------------------------------------------------------
// worker thread (StreamingFunc, DSC/CUDASceneRepChunkGrid.cpp:8-29)
// ---------------------------------------------------------------------------

void CUDASceneRepChunkGrid::workerLoop()
{
    (void)hipSetDevice(m_device); // the current device is per thread: bind the worker to the scene's GPU
    while (true) {
        streamOutToCPUPass1CPU(true);
        if (!m_sceneRepHashSDF->getOptions().s_offlineProcessing) {
            // offline mode streams in from the main thread
            try {
                streamInToGPUPass0CPU(getPosCamera(), getRadius(), s_useParts, true);
            } catch (const vh::Error&) {
                // staging overflow: surfaced to the main thread through s_nStreamdInBlocks = 0
                s_nStreamdInBlocks = 0;
                hEventInConsume.set();
            }
        }
        if (getTerminatedThread()) return;
    }
}

// DSC/CUDASceneRepChunkGrid.h:248-260 + initializeCriticalSection :617-629
void CUDASceneRepChunkGrid::startMultiThreading()
{
    if (!s_terminateThread) return;
    hEventOutProduce.reset(true);
    hEventOutConsume.reset(false);
    hEventInProduce.reset(true);
    hEventInConsume.reset(false);
    s_terminateThread = false;
    m_thread = std::thread(&CUDASceneRepChunkGrid::workerLoop, this);
}

// DSC/CUDASceneRepChunkGrid.h:262-289
void CUDASceneRepChunkGrid::stopMultiThreading()
{
    pipelineDrain();
    if (!s_terminateThread) {
        s_terminateThread = true;
        hEventOutProduce.set();
        hEventOutConsume.set();
        hEventInProduce.set();
        hEventInConsume.set();
        if (m_thread.joinable()) m_thread.join();
    }
}

void CUDASceneRepChunkGrid::clearGrid()
{
    std::lock_guard<std::mutex> l(m_gridMutex);
    m_grid.clear();
}

// DSC/CUDASceneRepChunkGrid.h:297-304
void CUDASceneRepChunkGrid::reset()
{
    const bool wasRunning = !s_terminateThread;
    stopMultiThreading();
    clearGrid();
    {
        std::lock_guard<std::mutex> l(m_gridMutex);
        std::fill(m_bitMask.begin(), m_bitMask.end(), 0u);
        m_bitMaskDirty = true;
    }
    m_currentPart = 0;
    if (wasRunning) startMultiThreading();
}

// ---------------------------------------------------------------------------
// helpers (DSC/CUDASceneRepChunkGrid.h:560-614)
// ---------------------------------------------------------------------------

bool CUDASceneRepChunkGrid::isValidChunk(const vh::vec3i& c) const
{
    // the reference accepts c == maxGridPos (one past the last chunk), which indexes
    // outside its grid; here a chunk is valid iff it lies inside the grid
    if (c.x < m_minGridPos.x || c.y < m_minGridPos.y || c.z < m_minGridPos.z) return false;
    if (c.x >= m_maxGridPos.x || c.y >= m_maxGridPos.y || c.z >= m_maxGridPos.z) return false;
    return true;
}

vh::vec3i CUDASceneRepChunkGrid::worldToChunks(const vh::vec3f& posWorld) const
{
    const float px = posWorld.x / m_voxelExtents.x, py = posWorld.y / m_voxelExtents.y, pz = posWorld.z / m_voxelExtents.z;
    vh::vec3i r;
    r.x = (int)(px + (float)signi(px) * 0.5f);
    r.y = (int)(py + (float)signi(py) * 0.5f);
    r.z = (int)(pz + (float)signi(pz) * 0.5f);
    return r;
}

vh::vec3f CUDASceneRepChunkGrid::chunkToWorld(const vh::vec3i& c) const
{
    return { (float)c.x * m_voxelExtents.x, (float)c.y * m_voxelExtents.y, (float)c.z * m_voxelExtents.z };
}

vh::vec3i CUDASceneRepChunkGrid::delinearizeChunkIndex(unsigned int idx) const
{
    const unsigned int x = idx % (unsigned int)m_gridDimensions.x;
    const unsigned int y = (idx % (unsigned int)(m_gridDimensions.x * m_gridDimensions.y)) / (unsigned int)m_gridDimensions.x;
    const unsigned int z = idx / (unsigned int)(m_gridDimensions.x * m_gridDimensions.y);
    return { m_minGridPos.x + (int)x, m_minGridPos.y + (int)y, m_minGridPos.z + (int)z };
}

unsigned int CUDASceneRepChunkGrid::linearizeChunkPos(const vh::vec3i& c) const
{
    const unsigned int px = (unsigned int)(c.x - m_minGridPos.x), py = (unsigned int)(c.y - m_minGridPos.y), pz = (unsigned int)(c.z - m_minGridPos.z);
    return pz * (unsigned int)m_gridDimensions.x * (unsigned int)m_gridDimensions.y + py * (unsigned int)m_gridDimensions.x + px;
}

vh::vec3i CUDASceneRepChunkGrid::meterToNumberOfChunksCeil(float f) const
{
    return { (int)std::ceil(f / m_voxelExtents.x), (int)std::ceil(f / m_voxelExtents.y), (int)std::ceil(f / m_voxelExtents.z) };
}

float CUDASceneRepChunkGrid::getChunkRadiusInMeter() const
{
    return length3(m_voxelExtents.x, m_voxelExtents.y, m_voxelExtents.z) / 2.0f;
}

float CUDASceneRepChunkGrid::getGridRadiusInMeter() const
{
    const vh::vec3f a = chunkToWorld(m_minGridPos), b = chunkToWorld(m_maxGridPos);
    const float hx = m_voxelExtents.x / 2.0f, hy = m_voxelExtents.y / 2.0f, hz = m_voxelExtents.z / 2.0f;
    return length3((a.x - hx) - (b.x + hx), (a.y - hy) - (b.y + hy), (a.z - hz) - (b.z + hz)) / 2.0f;
}

// conservative test: the *entire* chunk is within the sphere, DSC/CUDASceneRepChunkGrid.h:317-346
bool CUDASceneRepChunkGrid::isChunkInSphere(const vh::vec3i& chunk, const vh::vec3f& center, float radius) const
{
    const vh::vec3f posWorld = chunkToWorld(chunk);
    const float chunkExt = std::max(std::max(m_voxelExtents.x, m_voxelExtents.y), m_voxelExtents.z);
    const float chunkRadius = 0.5f * chunkExt * std::sqrt(3.0f);
    const float l = length3(posWorld.x - center.x, posWorld.y - center.y, posWorld.z - center.z);
    return l <= std::abs(radius - chunkRadius);
}

bool CUDASceneRepChunkGrid::containsSDFBlocksChunk(const vh::vec3i& chunk) const
{
    if (!isValidChunk(chunk)) return false;
    std::lock_guard<std::mutex> l(m_gridMutex);
    auto it = m_grid.find(linearizeChunkPos(chunk));
    return it != m_grid.end() && it->second->isStreamedOut();
}

bool CUDASceneRepChunkGrid::containsSDFBlocksChunkInRadius(const vh::vec3i& chunk, int r) const
{
    const vh::vec3i s = { std::max(chunk.x - r, m_minGridPos.x), std::max(chunk.y - r, m_minGridPos.y), std::max(chunk.z - r, m_minGridPos.z) };
    const vh::vec3i e = { std::min(chunk.x + r, m_maxGridPos.x), std::min(chunk.y + r, m_maxGridPos.y), std::min(chunk.z + r, m_maxGridPos.z) };
    for (int x = s.x; x <= e.x; x++)
        for (int y = s.y; y <= e.y; y++)
            for (int z = s.z; z <= e.z; z++)
                if (containsSDFBlocksChunk({ x, y, z })) return true;
    return false;
}

void CUDASceneRepChunkGrid::setBit(unsigned int index)
{
    m_bitMask[index / 32] |= (1u << (index % 32));
    m_bitMaskDirty = true;
}
void CUDASceneRepChunkGrid::resetBit(unsigned int index)
{
    m_bitMask[index / 32] &= ~(1u << (index % 32));
    m_bitMaskDirty = true;
}

// DSC/CUDASceneRepChunkGrid.h:306-309
unsigned int* CUDASceneRepChunkGrid::getBitMaskGPU()
{
    pipelineDrain(); // (the host's copy is then complete, and the same as the device's)
    std::lock_guard<std::mutex> l(m_gridMutex);
    if (m_bitMaskDirty) {
        hipStream_t s = (hipStream_t)m_sceneRepHashSDF->getStream();
        // pageable source: the copy is staged before the call returns, so the lock can be dropped afterwards
        checkHip(hipMemcpyAsync(d_bitMask, m_bitMask.data(), sizeof(unsigned int) * m_bitMask.size(), hipMemcpyHostToDevice, s), "getBitMaskGPU");
        m_bitMaskDirty = false;
    }
    return d_bitMask;
}

void CUDASceneRepChunkGrid::getStatistics(unsigned int out[3]) const
{
    const_cast<CUDASceneRepChunkGrid*>(this)->pipelineDrain();
    std::lock_guard<std::mutex> l(m_gridMutex);
    unsigned int blocks = 0, chunks = 0;
    for (auto& kv : m_grid) { chunks++; blocks += kv.second->getNElements(); }
    unsigned int bits = 0;
    for (unsigned int w : m_bitMask) bits += (unsigned int)__builtin_popcount(w);
    out[0] = chunks; out[1] = blocks; out[2] = bits;
}

void CUDASceneRepChunkGrid::downloadHostBlocks(std::vector<SDFBlockDesc>& descs, std::vector<vh::SDFBlock>& blocks) const
{
    const_cast<CUDASceneRepChunkGrid*>(this)->pipelineDrain();
    std::lock_guard<std::mutex> l(m_gridMutex);
    std::vector<unsigned int> keys;
    for (auto& kv : m_grid) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    descs.clear(); blocks.clear();
    for (unsigned int k : keys) {
        const ChunkDesc& c = *m_grid.at(k);
        descs.insert(descs.end(), c.getSDFBlockDescs().begin(), c.getSDFBlockDescs().end());
        blocks.insert(blocks.end(), c.getSDFBlocks().begin(), c.getSDFBlocks().end());
    }
}

// ---------------------------------------------------------------------------
// stream out: GPU -> host
// ---------------------------------------------------------------------------

// DSC/CUDASceneRepChunkGrid.cpp:31-42
void CUDASceneRepChunkGrid::streamOutToCPUAll()
{
    unsigned int nStreamedBlocksSum = 1;
    while (nStreamedBlocksSum != 0) {
        nStreamedBlocksSum = 0;
        for (unsigned int i = 0; i < m_streamOutParts; i++) {
            unsigned int nStreamedBlocks = 0;
            // radius 0: every block is "outside"
            const vh::vec3i far = worldToChunks({ (float)(m_minGridPos.x - 1), (float)(m_minGridPos.y - 1), (float)(m_minGridPos.z - 1) });
            streamOutToCPU({ (float)far.x, (float)far.y, (float)far.z }, 0.0f, s_useParts, nStreamedBlocks);
            nStreamedBlocksSum += nStreamedBlocks;
        }
    }
}

// DSC/CUDASceneRepChunkGrid.cpp:44-53
void CUDASceneRepChunkGrid::streamOutToCPU(const vh::vec3f& posCamera, float radius, bool useParts, unsigned int& nStreamedBlocks)
{
    s_posCamera = posCamera;
    s_radius = radius;
    streamOutToCPUPass0GPU(posCamera, radius, useParts, false);
    streamOutToCPUPass1CPU(false);
    nStreamedBlocks = s_nStreamdOutBlocks;
}

// The reference reads its counters back with a blocking cudaMemcpy (DSC/CUDASceneRepChunkGrid.cu:88, :140).  Here a
// one-thread kernel publishes them to mapped host memory behind the work already in the stream and the host polls the
// tag: no synchronisation call, no copy.  (After two seconds without the tag it synchronises and copies after all.)
void CUDASceneRepChunkGrid::readBack(const unsigned int* d_word0, const unsigned int* d_word1, unsigned int* out0, unsigned int* out1)
{
    vhStream_t stream = m_sceneRepHashSDF->getStream();
    const uint32_t tag = ++m_mirrorTag ? m_mirrorTag : ++m_mirrorTag; // never 0
    check(vh_publish_words(d_word0, d_word1, d_mirror, tag, stream), "vh_publish_words");
    volatile uint32_t* m = h_mirror;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned int spins = 0;
    while (m[2] != tag) {
        if ((++spins & 0x3ffu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            checkHip(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
            if (m[2] != tag) throw vh::Error(-(int)hipErrorUnknown, "read-back: the device did not publish its counters");
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (out0) *out0 = m[0];
    if (out1) *out1 = m[1];
}

// DSC/CUDASceneRepChunkGrid.cpp:55-105
void CUDASceneRepChunkGrid::streamOutToCPUPass0GPU(const vh::vec3f& posCamera, float radius, bool useParts, bool multiThreaded)
{
    pipelineDrain();
    std::unique_lock<std::mutex> lock(hMutexOut, std::defer_lock);
    if (multiThreaded && s_terminateThread)
        throw vh::Error(VH_ERR_BAD_ARGUMENT, "streamOutToCPUPass0GPU(multiThreaded): the streaming thread is not running");
    if (multiThreaded) {
        hEventOutProduce.wait();
        lock.lock();
    }
    s_posCamera = posCamera;
    s_radius = radius;

    const HashParams& hp = m_sceneRepHashSDF->getHashParams();
    HashData& hd = m_sceneRepHashSDF->getHashData();
    vhStream_t stream = m_sceneRepHashSDF->getStream();
    const int32_t token = m_sceneRepHashSDF->nextLockToken(); // = resetHashBucketMutexCUDA
    m_sceneRepHashSDF->noteTableEdited();
    check(vh_memset(d_SDFBlockCounter, 0, sizeof(unsigned int), stream), "clearSDFBlockCounter");

    const unsigned int numEntries = hp.m_hashNumBuckets * hp.m_hashBucketSize;
    unsigned int threadsPerPart = (numEntries + m_streamOutParts - 1) / m_streamOutParts;
    if (!useParts) threadsPerPart = numEntries;
    const unsigned int start = useParts ? m_currentPart * threadsPerPart : 0;

    const float cam[3] = { posCamera.x, posCamera.y, posCamera.z };
    check(vh_stream_out_pass1(&hd, &hp, threadsPerPart, start, radius, cam, d_SDFBlockCounter, d_SDFBlockDescOutput,
                              m_maxNumberOfSDFBlocksIntegrateFromGlobalHash, token, stream), "integrateFromGlobalHashPass1CUDA");
    unsigned int nSDFBlockDescs = 0;
    readBack(d_SDFBlockCounter, nullptr, &nSDFBlockDescs, nullptr);
    if (nSDFBlockDescs >= m_maxNumberOfSDFBlocksIntegrateFromGlobalHash) {
        if (multiThreaded) hEventOutProduce.set();
        throw vh::Error(VH_ERR_STAGING_OVERFLOW,
                        "not enough memory allocated for intermediate GPU buffer (wants to stream out more blocks than m_maxNumberOfSDFBlocksIntegrateFromGlobalHash)");
    }
    if (useParts) m_currentPart = (m_currentPart + 1) % m_streamOutParts;

    if (nSDFBlockDescs != 0) {
        check(vh_stream_out_pass2(&hd, &hp, d_SDFBlockDescOutput, (VhVoxel*)d_SDFBlockOutput, nSDFBlockDescs, stream), "integrateFromGlobalHashPass2CUDA");
        hipStream_t s = (hipStream_t)stream;
        checkHip(hipMemcpyAsync(h_SDFBlockDescOutput, d_SDFBlockDescOutput, sizeof(SDFBlockDesc) * nSDFBlockDescs, hipMemcpyDeviceToHost, s), "D2H descs");
        checkHip(hipMemcpyAsync(h_SDFBlockOutput, d_SDFBlockOutput, sizeof(vh::SDFBlock) * nSDFBlockDescs, hipMemcpyDeviceToHost, s), "D2H blocks");
        checkHip(hipStreamSynchronize(s), "hipStreamSynchronize");
    }
    s_nStreamdOutBlocks = nSDFBlockDescs;

    if (multiThreaded) hEventOutConsume.set();
}

// ---- a frame in which nothing streams (see vh.hpp) ----------------------------------------------------------------

void CUDASceneRepChunkGrid::probeStreamOut(const vh::vec3f& posCamera, float radius, bool useParts)
{
    const HashParams& hp = m_sceneRepHashSDF->getHashParams();
    HashData& hd = m_sceneRepHashSDF->getHashData();
    const unsigned int numEntries = hp.m_hashNumBuckets * hp.m_hashBucketSize;
    unsigned int threadsPerPart = (numEntries + m_streamOutParts - 1) / m_streamOutParts;
    if (!useParts) threadsPerPart = numEntries;
    const unsigned int start = useParts ? m_currentPart * threadsPerPart : 0; // the part the NEXT pass 0 scans
    const float cam[3] = { posCamera.x, posCamera.y, posCamera.z };
    m_probeTag = ++m_probeTag ? m_probeTag : 1u;
    check(vh_stream_out_probe(&hd, &hp, threadsPerPart, start, radius, cam, d_probeCounter, d_probe, m_probeTag, m_sceneRepHashSDF->getStream()),
          "vh_stream_out_probe");
}

unsigned int CUDASceneRepChunkGrid::probeResult()
{
    volatile uint32_t* m = h_probe;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned int spins = 0;
    while (m[2] != m_probeTag) {
        if ((++spins & 0x3ffu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            checkHip(hipStreamSynchronize((hipStream_t)m_sceneRepHashSDF->getStream()), "hipStreamSynchronize");
            if (m[2] != m_probeTag) throw vh::Error(-(int)hipErrorUnknown, "stream-out probe: the device did not publish its count");
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return m[0];
}

// streamOutToCPUPass0GPU(multiThreaded = true) of a part the probe found nothing to move out of
void CUDASceneRepChunkGrid::streamOutNothing(const vh::vec3f& posCamera, float radius, bool useParts)
{
    pipelineDrain();
    if (s_terminateThread) throw vh::Error(VH_ERR_BAD_ARGUMENT, "streamOutNothing: the streaming thread is not running");
    hEventOutProduce.wait();
    {
        std::lock_guard<std::mutex> lock(hMutexOut);
        s_posCamera = posCamera;
        s_radius = radius;
        (void)m_sceneRepHashSDF->nextLockToken(); // (the pass draws one: keep the sequence of tokens the same)
        if (useParts) m_currentPart = (m_currentPart + 1) % m_streamOutParts;
        s_nStreamdOutBlocks = 0;
    }
    hEventOutConsume.set();
}

unsigned int CUDASceneRepChunkGrid::streamInWait()
{
    if (s_terminateThread || m_sceneRepHashSDF->getOptions().s_offlineProcessing)
        throw vh::Error(VH_ERR_BAD_ARGUMENT, "streamInWait: no worker thread prepares the pass (offline processing, or streaming thread stopped)");
    hEventInConsume.wait();
    m_streamInLock = std::unique_lock<std::mutex>(hMutexIn);
    return s_nStreamdInBlocks;
}

void CUDASceneRepChunkGrid::streamInFinish()
{
    if (!m_streamInLock.owns_lock()) throw vh::Error(VH_ERR_BAD_ARGUMENT, "streamInFinish without streamInWait");
    try {
        streamInLaunches();
    } catch (...) {
        m_streamInLock.unlock();
        hEventInProduce.set();
        throw;
    }
    m_streamInLock.unlock();
    hEventInProduce.set();
}

void CUDASceneRepChunkGrid::streamInAbort()
{
    if (!m_streamInLock.owns_lock()) return;
    try {
        // what the worker took out of the grid for this pass is still in its staging buffers (integrateInHash)
        if (s_nStreamdInBlocks != 0) integrateInChunkGrid(h_SDFBlockDescInput, h_SDFBlockInput, s_nStreamdInBlocks);
    } catch (...) {
    }
    s_nStreamdInBlocks = 0;
    m_streamInLock.unlock();
    hEventInProduce.set();
}

// DSC/CUDASceneRepChunkGrid.cpp:107-124
void CUDASceneRepChunkGrid::streamOutToCPUPass1CPU(bool multiThreaded)
{
    std::unique_lock<std::mutex> lock(hMutexOut, std::defer_lock);
    if (multiThreaded) {
        hEventOutConsume.wait();
        lock.lock();
        if (s_terminateThread) return; // avoid duplicate insertions when stop multi-threading is called
    }
    if (s_nStreamdOutBlocks != 0) integrateInChunkGrid(h_SDFBlockDescOutput, h_SDFBlockOutput, s_nStreamdOutBlocks);
    if (multiThreaded) hEventOutProduce.set();
}

// DSC/CUDASceneRepChunkGrid.cpp:126-153
void CUDASceneRepChunkGrid::integrateInChunkGrid(const SDFBlockDesc* desc, const vh::SDFBlock* block, unsigned int nSDFBlocks)
{
    const float voxelSize = m_sceneRepHashSDF->getHashParams().m_virtualVoxelSize;
    std::lock_guard<std::mutex> l(m_gridMutex);
    for (unsigned int i = 0; i < nSDFBlocks; i++) {
        const vh::vec3f posWorld = { (float)(desc[i].pos[0] * VH_SDF_BLOCK_SIZE) * voxelSize, (float)(desc[i].pos[1] * VH_SDF_BLOCK_SIZE) * voxelSize,
                                     (float)(desc[i].pos[2] * VH_SDF_BLOCK_SIZE) * voxelSize };
        const vh::vec3i chunk = worldToChunks(posWorld);
        if (!isValidChunk(chunk)) {
            std::fprintf(stderr, "Chunk out of bounds\n");
            continue;
        }
        const unsigned int index = linearizeChunkPos(chunk);
        auto it = m_grid.find(index);
        if (it == m_grid.end()) it = m_grid.emplace(index, std::unique_ptr<ChunkDesc>(new ChunkDesc(m_initialChunkDescListSize))).first;
        it->second->addSDFBlock(desc[i], block[i]);
        setBit(index);
    }
}

// ---------------------------------------------------------------------------
// stream in: host -> GPU
// ---------------------------------------------------------------------------

// DSC/CUDASceneRepChunkGrid.cpp:155-161
void CUDASceneRepChunkGrid::streamInToGPUAll()
{
    unsigned int nStreamedBlocks = 1;
    while (nStreamedBlocks != 0) {
        streamInToGPU(chunkToWorld({ 0, 0, 0 }), 1.1f * getGridRadiusInMeter(), s_useParts, nStreamedBlocks);
    }
}

// DSC/CUDASceneRepChunkGrid.cpp:163-172
void CUDASceneRepChunkGrid::streamInToGPUAll(const vh::vec3f& posCamera, float radius, bool useParts, unsigned int& nStreamedBlocks)
{
    unsigned int nStreamedBlocksSum = 0;
    unsigned int nBlock = 1;
    while (nBlock != 0) {
        streamInToGPU(posCamera, radius, useParts, nBlock);
        nStreamedBlocksSum += nBlock;
    }
    nStreamedBlocks = nStreamedBlocksSum;
}

// DSC/CUDASceneRepChunkGrid.cpp:174-181
void CUDASceneRepChunkGrid::streamInToGPUChunk(const vh::vec3i& chunkPos)
{
    unsigned int nStreamedBlocks = 1;
    while (nStreamedBlocks != 0) streamInToGPU(chunkToWorld(chunkPos), 1.1f * getChunkRadiusInMeter(), true, nStreamedBlocks);
}

// DSC/CUDASceneRepChunkGrid.cpp:183-195
void CUDASceneRepChunkGrid::streamInToGPUChunkNeighborhood(const vh::vec3i& c, int r)
{
    const vh::vec3i s = { std::max(c.x - r, m_minGridPos.x), std::max(c.y - r, m_minGridPos.y), std::max(c.z - r, m_minGridPos.z) };
    const vh::vec3i e = { std::min(c.x + r, m_maxGridPos.x), std::min(c.y + r, m_maxGridPos.y), std::min(c.z + r, m_maxGridPos.z) };
    for (int x = s.x; x < e.x; x++)
        for (int y = s.y; y < e.y; y++)
            for (int z = s.z; z < e.z; z++) streamInToGPUChunk({ x, y, z });
}

// DSC/CUDASceneRepChunkGrid.cpp:197-206
void CUDASceneRepChunkGrid::streamInToGPU(const vh::vec3f& posCamera, float radius, bool useParts, unsigned int& nStreamedBlocks)
{
    s_posCamera = posCamera;
    s_radius = radius;
    streamInToGPUPass0CPU(posCamera, radius, useParts, false);
    streamInToGPUPass1GPU(false);
    nStreamedBlocks = s_nStreamdInBlocks;
}

// DSC/CUDASceneRepChunkGrid.cpp:208-225
void CUDASceneRepChunkGrid::streamInToGPUPass0CPU(const vh::vec3f& posCamera, float radius, bool useParts, bool multiThreaded)
{
    if (!multiThreaded) pipelineDrain(); // (called by the worker otherwise: the main thread's entry points have drained)
    std::unique_lock<std::mutex> lock(hMutexIn, std::defer_lock);
    if (multiThreaded) {
        hEventInProduce.wait();
        lock.lock();
        if (s_terminateThread) return; // avoid duplicate insertions when stop multi-threading is called
    }
    s_nStreamdInBlocks = integrateInHash(posCamera, radius, useParts);
    if (multiThreaded) hEventInConsume.set();
}

// DSC/CUDASceneRepChunkGrid.cpp:227-266
void CUDASceneRepChunkGrid::streamInToGPUPass1GPU(bool multiThreaded)
{
    std::unique_lock<std::mutex> lock(hMutexIn, std::defer_lock);
    // the worker prepares stream-in passes only in online mode (workerLoop; DSC/CUDASceneRepChunkGrid.cpp:13): waiting
    // for it in offline mode, or while it is stopped, would never end
    if (multiThreaded && (s_terminateThread || m_sceneRepHashSDF->getOptions().s_offlineProcessing))
        throw vh::Error(VH_ERR_BAD_ARGUMENT, "streamInToGPUPass1GPU(multiThreaded): no worker thread prepares the pass (offline processing, or streaming thread stopped)");
    if (multiThreaded) {
        hEventInConsume.wait();
        lock.lock();
    }
    try {
        streamInLaunches();
    } catch (...) {
        if (multiThreaded) hEventInProduce.set();
        throw;
    }
    if (multiThreaded) hEventInProduce.set();
}

// the device side of a stream-in pass the worker (or streamInToGPUPass0CPU) has prepared
void CUDASceneRepChunkGrid::streamInLaunches()
{
    if (s_nStreamdInBlocks != 0) {
        const HashParams& hp = m_sceneRepHashSDF->getHashParams();
        HashData& hd = m_sceneRepHashSDF->getHashData();
        vhStream_t stream = m_sceneRepHashSDF->getStream();
        unsigned int heapCountPrev = 0; // index of the top free block
        readBack(hd.d_heapCounter, nullptr, &heapCountPrev, nullptr);
        if (s_nStreamdInBlocks > heapCountPrev + 1u) throw vh::Error(VH_ERR_HEAP_EXHAUSTED, "stream-in: not enough free SDF blocks");
        const int32_t token = m_sceneRepHashSDF->nextLockToken();
        m_sceneRepHashSDF->noteTableEdited();
        hipStream_t hs = (hipStream_t)stream;
        checkHip(hipMemsetAsync(d_insertFailed, 0, sizeof(unsigned int), hs), "clear failed inserts");
        check(vh_stream_in_pass1_report(&hd, &hp, s_nStreamdInBlocks, heapCountPrev, d_SDFBlockDescInput, token, d_insertFailed, stream), "chunkToGlobalHashPass1CUDA");
        check(vh_stream_in_pass2(&hd, &hp, s_nStreamdInBlocks, heapCountPrev, d_SDFBlockDescInput, (const VhVoxel*)d_SDFBlockInput, stream), "chunkToGlobalHashPass2CUDA");
        // update heap counter (pinned source: stays valid until the copy ran)
        h_counter[0] = heapCountPrev - s_nStreamdInBlocks;
        checkHip(hipMemcpyAsync(hd.d_heapCounter, &h_counter[0], sizeof(unsigned int), hipMemcpyHostToDevice, hs), "heapCounter");
        unsigned int nFailed = 0;
        readBack(d_insertFailed, nullptr, &nFailed, nullptr); // (also: the copy above has read h_counter[0] by then)
        if (nFailed != 0) takeBackFailedInserts(nFailed, heapCountPrev);
    }
}

// Blocks of the last stream-in pass that found no slot (their bucket and its list full, or a second overflow of one
// bucket within the pass): the reference has no defined behaviour for them (DSC/VoxelUtilHashSDF.h:682-713).  Here they
// go back to where they came from -- the host chunk grid -- and their SDF blocks back to the heap, cleared, so that no
// block is lost and the pool still is the union of heap and table.  They come in again with a later pass.
void CUDASceneRepChunkGrid::takeBackFailedInserts(unsigned int nFailed, unsigned int heapCountPrev)
{
    takeBackFailedInserts(nFailed, heapCountPrev, h_SDFBlockDescInput, h_SDFBlockInput, s_nStreamdInBlocks);
}

void CUDASceneRepChunkGrid::takeBackFailedInserts(unsigned int nFailed, unsigned int heapCountPrev, const SDFBlockDesc* descs, const vh::SDFBlock* blocks, unsigned int& nIn)
{
    HashData& hd = m_sceneRepHashSDF->getHashData();
    hipStream_t hs = (hipStream_t)m_sceneRepHashSDF->getStream();
    std::vector<unsigned int> idx(nFailed), blockIds(nFailed);
    checkHip(hipMemcpy(idx.data(), d_insertFailed + 1, sizeof(unsigned int) * nFailed, hipMemcpyDeviceToHost), "failed insert list");
    unsigned int counter = 0;
    checkHip(hipMemcpy(&counter, hd.d_heapCounter, sizeof(unsigned int), hipMemcpyDeviceToHost), "heapCounter");
    for (unsigned int k = 0; k < nFailed; k++) {
        const unsigned int i = idx[k];
        if (i >= nIn) throw vh::Error(VH_ERR_INSERT_FAILED, "stream-in: corrupt list of failed inserts");
        // the heap slot the pass took for this block (chunkToGlobalHashPass1Kernel: heap[heapCountPrev - i])
        checkHip(hipMemcpy(&blockIds[k], hd.d_heap + (heapCountPrev - i), sizeof(unsigned int), hipMemcpyDeviceToHost), "heap slot");
        checkHip(hipMemsetAsync(hd.d_SDFBlocks + (size_t)blockIds[k] * VH_SDF_BLOCK_VOXELS, 0, sizeof(vh::SDFBlock), hs), "clear block");
        integrateInChunkGrid(&descs[i], &blocks[i], 1); // the staging copy is still there
    }
    // appendHeap, DSC/VoxelUtilHashSDF.h:525-529, nFailed times
    checkHip(hipMemcpyAsync(hd.d_heap + counter + 1, blockIds.data(), sizeof(unsigned int) * nFailed, hipMemcpyHostToDevice, hs), "heap");
    counter += nFailed;
    checkHip(hipMemcpyAsync(hd.d_heapCounter, &counter, sizeof(unsigned int), hipMemcpyHostToDevice, hs), "heapCounter");
    checkHip(hipStreamSynchronize(hs), "hipStreamSynchronize");
    nIn -= nFailed;
    m_numFailedInserts += nFailed;
}

// DSC/CUDASceneRepChunkGrid.cpp:268-311
unsigned int CUDASceneRepChunkGrid::integrateInHash(const vh::vec3f& posCamera, float radius, bool useParts)
{
    return integrateInHash(posCamera, radius, useParts, h_SDFBlockDescInput, h_SDFBlockInput, d_SDFBlockDescInput, d_SDFBlockInput,
                           m_maxNumberOfSDFBlocksIntegrateFromGlobalHash, nullptr);
}

// (the staging buffers as arguments: the pipeline has two sets of its own; chunkBit: the bit of the LAST chunk taken -- with
// useParts the only one)
unsigned int CUDASceneRepChunkGrid::integrateInHash(const vh::vec3f& posCamera, float radius, bool useParts, SDFBlockDesc* hDescs, vh::SDFBlock* hBlocks,
                                                    SDFBlockDesc* dDescs, vh::SDFBlock* dBlocks, unsigned int capacity, unsigned int* chunkBit)
{
    const vh::vec3i camChunk = worldToChunks(posCamera);
    const vh::vec3i chunkRadius = meterToNumberOfChunksCeil(radius);
    const vh::vec3i startChunk = { std::max(camChunk.x - chunkRadius.x, m_minGridPos.x), std::max(camChunk.y - chunkRadius.y, m_minGridPos.y),
                                   std::max(camChunk.z - chunkRadius.z, m_minGridPos.z) };
    const vh::vec3i endChunk = { std::min(camChunk.x + chunkRadius.x, m_maxGridPos.x - 1), std::min(camChunk.y + chunkRadius.y, m_maxGridPos.y - 1),
                                 std::min(camChunk.z + chunkRadius.z, m_maxGridPos.z - 1) };
    hipStream_t cs = (hipStream_t)m_copyStream;

    unsigned int nSDFBlocks = 0;
    std::lock_guard<std::mutex> l(m_gridMutex);
    for (int x = startChunk.x; x <= endChunk.x; x++) {
        for (int y = startChunk.y; y <= endChunk.y; y++) {
            for (int z = startChunk.z; z <= endChunk.z; z++) {
                const unsigned int index = linearizeChunkPos({ x, y, z });
                auto it = m_grid.find(index);
                if (it == m_grid.end() || !it->second->isStreamedOut()) continue; // has been allocated and has streamed out blocks
                if (!isChunkInSphere(delinearizeChunkIndex(index), posCamera, radius)) continue; // is in camera range
                ChunkDesc& c = *it->second;
                const unsigned int nBlock = c.getNElements();
                if (nBlock + nSDFBlocks > capacity) {
                    throw vh::Error(VH_ERR_STAGING_OVERFLOW,
                                    "not enough memory allocated for intermediate GPU buffer (wants to stream in more blocks than m_maxNumberOfSDFBlocksIntegrateFromGlobalHash)");
                }
                // copy data to GPU through pinned staging, on the worker's own stream
                std::memcpy(hDescs + nSDFBlocks, c.getSDFBlockDescs().data(), sizeof(SDFBlockDesc) * nBlock);
                std::memcpy(hBlocks + nSDFBlocks, c.getSDFBlocks().data(), sizeof(vh::SDFBlock) * nBlock);
                checkHip(hipMemcpyAsync(dDescs + nSDFBlocks, hDescs + nSDFBlocks, sizeof(SDFBlockDesc) * nBlock, hipMemcpyHostToDevice, cs), "H2D descs");
                checkHip(hipMemcpyAsync(dBlocks + nSDFBlocks, hBlocks + nSDFBlocks, sizeof(vh::SDFBlock) * nBlock, hipMemcpyHostToDevice, cs), "H2D blocks");
                // remove data from CPU
                c.clear();
                resetBit(index);
                if (chunkBit) *chunkBit = index;
                nSDFBlocks += nBlock;
                if (useParts) {
                    checkHip(hipStreamSynchronize(cs), "hipStreamSynchronize");
                    return nSDFBlocks; // only one chunk per frame
                }
            }
        }
    }
    checkHip(hipStreamSynchronize(cs), "hipStreamSynchronize");
    return nSDFBlocks;
}

// DSC/CUDASceneRepChunkGrid.cpp:313-341
void CUDASceneRepChunkGrid::debugCheckForDuplicates() const
{
    const_cast<CUDASceneRepChunkGrid*>(this)->pipelineDrain();
    struct PosHash {
        size_t operator()(const std::array<int, 3>& v) const
        {
            return ((size_t)v[0] * 73856093u) ^ ((size_t)v[1] * 19349669u) ^ ((size_t)v[2] * 83492791u);
        }
    };
    std::unordered_set<std::array<int, 3>, PosHash> seen;
    const HashParams& hp = m_sceneRepHashSDF->getHashParams();
    const size_t ne = (size_t)hp.m_hashBucketSize * hp.m_hashNumBuckets;
    std::vector<HashEntry> hashCPU(ne);
    check(vh_memcpy_d2h(hashCPU.data(), m_sceneRepHashSDF->getHashData().d_hash, sizeof(HashEntry) * ne, m_sceneRepHashSDF->getStream()), "debugCheckForDuplicates");
    for (size_t i = 0; i < ne; i++) {
        if (hashCPU[i].ptr != VH_FREE_ENTRY) {
            if (!seen.insert({ hashCPU[i].pos[0], hashCPU[i].pos[1], hashCPU[i].pos[2] }).second)
                throw vh::Error(VH_ERR_BAD_ARGUMENT, "Duplicate found in streaming hash data (in hash)");
        }
    }
    std::lock_guard<std::mutex> l(m_gridMutex);
    for (auto& kv : m_grid) {
        for (const SDFBlockDesc& d : kv.second->getSDFBlockDescs()) {
            if (!seen.insert({ d.pos[0], d.pos[1], d.pos[2] }).second)
                throw vh::Error(VH_ERR_BAD_ARGUMENT, "Duplicate found in streaming hash data (in grid)");
        }
    }
}

// ---------------------------------------------------------------------------
// the streaming step without host waits (see vh.hpp)
// ---------------------------------------------------------------------------

void CUDASceneRepChunkGrid::pipelineStart()
{
    if (m_plStarted) return;
    const size_t n = kPipelineBlocks;
    for (int i = 0; i < 2; i++) {
        checkHip(hipMalloc((void**)&d_plOutDesc[i], sizeof(SDFBlockDesc) * n), "hipMalloc");
        checkHip(hipHostMalloc((void**)&h_plOutDesc[i], sizeof(SDFBlockDesc) * n, hipHostMallocMapped), "hipHostMalloc");
        checkHip(hipHostMalloc((void**)&h_plOutBlocks[i], sizeof(vh::SDFBlock) * n, hipHostMallocMapped), "hipHostMalloc");
        checkHip(hipHostGetDevicePointer((void**)&hd_plOutDesc[i], h_plOutDesc[i], 0), "hipHostGetDevicePointer");
        checkHip(hipHostGetDevicePointer((void**)&hd_plOutBlocks[i], h_plOutBlocks[i], 0), "hipHostGetDevicePointer");
        checkHip(hipHostMalloc((void**)&h_plOutMirror[i], sizeof(uint32_t) * 4, hipHostMallocMapped), "hipHostMalloc");
        h_plOutMirror[i][0] = h_plOutMirror[i][1] = h_plOutMirror[i][2] = h_plOutMirror[i][3] = 0u;
        checkHip(hipHostGetDevicePointer((void**)&hd_plOutMirror[i], h_plOutMirror[i], 0), "hipHostGetDevicePointer");
        checkHip(hipHostMalloc((void**)&h_plInDesc[i], sizeof(SDFBlockDesc) * n, hipHostMallocDefault), "hipHostMalloc");
        checkHip(hipHostMalloc((void**)&h_plInBlocks[i], sizeof(vh::SDFBlock) * n, hipHostMallocDefault), "hipHostMalloc");
        checkHip(hipMalloc((void**)&d_plInDesc[i], sizeof(SDFBlockDesc) * n), "hipMalloc");
        checkHip(hipMalloc((void**)&d_plInBlocks[i], sizeof(vh::SDFBlock) * n), "hipMalloc");
    }
    for (int i = 0; i < 2; i++) {
        checkHip(hipHostMalloc((void**)&h_plInMirror[i], sizeof(uint32_t) * 4, hipHostMallocMapped), "hipHostMalloc");
        h_plInMirror[i][0] = h_plInMirror[i][1] = h_plInMirror[i][2] = h_plInMirror[i][3] = 0u;
        checkHip(hipHostGetDevicePointer((void**)&hd_plInMirror[i], h_plInMirror[i], 0), "hipHostGetDevicePointer");
    }
    m_plQuit = false;
    m_plThread = std::thread(&CUDASceneRepChunkGrid::pipelineWorker, this);
    m_plStarted = true;
}

void CUDASceneRepChunkGrid::pipelineStop()
{
    if (!m_plStarted) return;
    {
        std::lock_guard<std::mutex> l(m_plMutex);
        m_plQuit = true;
    }
    m_plCv.notify_all();
    if (m_plThread.joinable()) m_plThread.join();
    m_plStarted = false;
    for (int i = 0; i < 2; i++) {
        (void)hipFree(d_plOutDesc[i]); (void)hipHostFree(h_plOutDesc[i]); (void)hipHostFree(h_plOutBlocks[i]); (void)hipHostFree(h_plOutMirror[i]);
        (void)hipHostFree(h_plInDesc[i]); (void)hipHostFree(h_plInBlocks[i]); (void)hipFree(d_plInDesc[i]); (void)hipFree(d_plInBlocks[i]);
        d_plOutDesc[i] = nullptr; h_plOutDesc[i] = nullptr; h_plOutBlocks[i] = nullptr; h_plOutMirror[i] = nullptr;
        h_plInDesc[i] = nullptr; h_plInBlocks[i] = nullptr; d_plInDesc[i] = nullptr; d_plInBlocks[i] = nullptr;
    }
    for (int i = 0; i < 2; i++) { (void)hipHostFree(h_plInMirror[i]); h_plInMirror[i] = nullptr; }
}

// the worker: one job per frame
void CUDASceneRepChunkGrid::pipelineWorker()
{
    (void)hipSetDevice(m_device);
    unsigned int done = 0;
    for (;;) {
        PipelineJob job;
        {
            // frames come every 100-200 us: look at the counter for a while before going to sleep (AutoResetEvent::wait)
            const auto t0 = std::chrono::steady_clock::now();
            unsigned int spins = 0;
            while (m_plPosted.load(std::memory_order_acquire) == done && !m_plQuit) {
                if ((++spins & 0xffu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
            }
            std::unique_lock<std::mutex> l(m_plMutex);
            m_plCv.wait(l, [&] { return m_plQuit || m_plPosted.load(std::memory_order_acquire) != done; });
            if (m_plQuit) return;
            job = m_plJob;
        }
        try {
            if (job.haveOut) {
                // the blocks that left: in the mapped staging buffer once the device has published the pass's tag
                volatile uint32_t* m = h_plOutMirror[job.outSlot];
                const auto t0 = std::chrono::steady_clock::now();
                unsigned int spins = 0;
                while (m[2] != job.outTag) {
                    if ((++spins & 0xfffu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
                        throw vh::Error(VH_ERR_TIMEOUT, "streaming pipeline: the device did not publish its stream-out pass for 30 s");
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                const unsigned int n = m[0];
                if (n > job.outMost)
                    throw vh::Error(VH_ERR_STAGING_OVERFLOW, "streaming pipeline: the stream-out pass found more blocks than its probe (blocks are lost)");
                if (n != 0) integrateInChunkGrid(h_plOutDesc[job.outSlot], h_plOutBlocks[job.outSlot], n);
                m_plBlocksOut += n;
            }
            StreamDecision d = { 0u, 0xffffffffu, job.inSlot };
            if (job.haveNext) {
                d.nIn = integrateInHash(job.nextPos, job.nextRadius, true, h_plInDesc[job.inSlot], h_plInBlocks[job.inSlot], d_plInDesc[job.inSlot],
                                        d_plInBlocks[job.inSlot], kPipelineBlocks, &d.chunkBit);
            }
            m_plDecision = d;
        } catch (const vh::Error& e) {
            m_plDecision = { 0u, 0xffffffffu, job.inSlot };
            m_plError.store(e.code != 0 ? e.code : VH_ERR_BAD_ARGUMENT, std::memory_order_release);
        }
        done++;
        m_plDone.store(done, std::memory_order_release);
    }
}

bool CUDASceneRepChunkGrid::pipelineHasDecision(const vh::vec3f& posCamera, float radius) const
{
    return m_plStarted && m_plDecisionValid && m_plDecisionPos.x == posCamera.x && m_plDecisionPos.y == posCamera.y && m_plDecisionPos.z == posCamera.z &&
           m_plDecisionRadius == radius;
}

// the outcome of the previous frame's insert (published by k_stream_in_commit): failures are repaired here, by the main thread
void CUDASceneRepChunkGrid::pipelineCheckInsert(int slot, bool block)
{
    if (!m_plInsert[slot].pending) return;
    volatile uint32_t* m = h_plInMirror[slot];
    if (m[2] != m_plInsert[slot].tag) {
        if (!block) return;
        checkHip(hipStreamSynchronize((hipStream_t)m_sceneRepHashSDF->getStream()), "hipStreamSynchronize");
        if (m[2] != m_plInsert[slot].tag) throw vh::Error(-(int)hipErrorUnknown, "streaming pipeline: the device did not publish its stream-in pass");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    m_plInsert[slot].pending = false;
    const unsigned int nFailed = m[0], heapCountPrev = m[1], exhausted = m[3];
    unsigned int nIn = m_plInsert[slot].nIn;
    if (exhausted) {
        // the heap held too few free blocks: nothing was inserted, the chunk goes back into the grid (its bit is set again there;
        // the device's copy of the mask was not touched)
        integrateInChunkGrid(h_plInDesc[slot], h_plInBlocks[slot], nIn);
        m_numFailedInserts += nIn;
        nIn = 0;
    } else if (nFailed != 0) {
        checkHip(hipStreamSynchronize((hipStream_t)m_sceneRepHashSDF->getStream()), "hipStreamSynchronize");
        takeBackFailedInserts(nFailed, heapCountPrev, h_plInDesc[slot], h_plInBlocks[slot], nIn);
        // The blocks that went back set their chunk's bit in the host's copy; the device's copy had it cleared by the pass: set
        // there too.  (That one bit, not an upload of the host's copy: a stream-out pass enqueued since may have set bits on
        // the device that the host will only learn of when its blocks arrive.  The device is idle: the stream was synchronised.)
        const unsigned int bit = m_plInsert[slot].chunkBit;
        if (bit != 0xffffffffu) {
            unsigned int word = 0;
            checkHip(hipMemcpy(&word, d_bitMask + (bit >> 5), sizeof(word), hipMemcpyDeviceToHost), "bit mask word");
            word |= 1u << (bit & 31u);
            checkHip(hipMemcpy(d_bitMask + (bit >> 5), &word, sizeof(word), hipMemcpyHostToDevice), "bit mask word");
        }
    }
    m_plBlocksIn += nIn;
}

CUDASceneRepChunkGrid::StreamDecision CUDASceneRepChunkGrid::pipelineDecision()
{
    if (!m_plStarted || !m_plDecisionValid) throw vh::Error(VH_ERR_BAD_ARGUMENT, "pipelineDecision(): pipelineAsk() has not asked for one");
    // the worker has had the rest of the previous frame's device time
    const unsigned int posted = m_plPosted.load(std::memory_order_acquire);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned int spins = 0;
    while (m_plDone.load(std::memory_order_acquire) != posted) {
        if ((++spins & 0xfffu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(40))
            throw vh::Error(VH_ERR_TIMEOUT, "streaming pipeline: the worker thread did not answer for 40 s");
    }
    const int err = m_plError.exchange(0, std::memory_order_acq_rel);
    if (err != 0) {
        m_plDecisionValid = false;
        throw vh::Error(err, std::string("streaming pipeline (worker): ") + vh_error_string(err));
    }
    m_plDecisionValid = false; // (consumed)
    pipelineCheckInsert(0, false);
    pipelineCheckInsert(1, false);
    return m_plDecision;
}

bool CUDASceneRepChunkGrid::pipelineStreamOut(const vh::vec3f& posCamera, float radius, bool useParts, unsigned int mostBlocks)
{
    if (mostBlocks > (unsigned int)kPipelineBlocks) return false;
    pipelineStart();
    s_posCamera = posCamera;
    s_radius = radius;
    const int slot = (int)(m_plFrame & 1u);
    m_plOutThisFrame = false;
    if (mostBlocks != 0u) {
        const HashParams& hp = m_sceneRepHashSDF->getHashParams();
        HashData& hd = m_sceneRepHashSDF->getHashData();
        vhStream_t stream = m_sceneRepHashSDF->getStream();
        const int32_t token = m_sceneRepHashSDF->nextLockToken(); // = resetHashBucketMutexCUDA
        m_sceneRepHashSDF->noteTableEdited();
        const unsigned int numEntries = hp.m_hashNumBuckets * hp.m_hashBucketSize;
        unsigned int threadsPerPart = (numEntries + m_streamOutParts - 1) / m_streamOutParts;
        if (!useParts) threadsPerPart = numEntries;
        const unsigned int start = useParts ? m_currentPart * threadsPerPart : 0;
        const float cam[3] = { posCamera.x, posCamera.y, posCamera.z };
        // pass 1 lists into device memory (pass 2 reads the list back); pass 2 writes blocks AND descriptors to the host
        check(vh_stream_out_device(&hd, &hp, threadsPerPart, start, radius, cam, d_SDFBlockCounter, d_plOutDesc[slot], (VhVoxel*)hd_plOutBlocks[slot],
                                   mostBlocks, token, d_bitMask, stream), "vh_stream_out_device");
        checkHip(hipMemcpyAsync(h_plOutDesc[slot], d_plOutDesc[slot], sizeof(SDFBlockDesc) * mostBlocks, hipMemcpyDeviceToHost, (hipStream_t)stream), "descs to host");
        m_plOutTag = ++m_plTag ? m_plTag : ++m_plTag;
        m_plOutMost = mostBlocks;
        check(vh_publish_count(d_SDFBlockCounter, hd_plOutMirror[slot], m_plOutTag, stream), "vh_publish_count");
        m_plOutThisFrame = true;
    } else {
        (void)m_sceneRepHashSDF->nextLockToken(); // (the pass draws one: keep the sequence of tokens the same)
    }
    if (useParts) m_currentPart = (m_currentPart + 1) % m_streamOutParts;
    s_nStreamdOutBlocks = 0; // (known to the worker only)
    return true;
}

void CUDASceneRepChunkGrid::pipelineStreamIn(const StreamDecision& d)
{
    s_nStreamdInBlocks = d.nIn;
    if (d.nIn == 0u) return;
    pipelineCheckInsert(d.slot, true); // (the slot's previous insert, two frames back: long done)
    const HashParams& hp = m_sceneRepHashSDF->getHashParams();
    HashData& hd = m_sceneRepHashSDF->getHashData();
    const int32_t token = m_sceneRepHashSDF->nextLockToken();
    m_sceneRepHashSDF->noteTableEdited();
    const uint32_t tag = ++m_plTag ? m_plTag : ++m_plTag;
    check(vh_stream_in_device(&hd, &hp, d.nIn, d_plInDesc[d.slot], (const VhVoxel*)d_plInBlocks[d.slot], token, d_insertFailed, d_bitMask, d.chunkBit,
                              hd_plInMirror[d.slot], tag, m_sceneRepHashSDF->getStream()), "vh_stream_in_device");
    m_plInsert[d.slot].pending = true; m_plInsert[d.slot].tag = tag; m_plInsert[d.slot].nIn = d.nIn; m_plInsert[d.slot].chunkBit = d.chunkBit;
}

void CUDASceneRepChunkGrid::pipelineAsk(bool haveNext, const vh::vec3f& nextPosCamera, float nextRadius)
{
    pipelineStart();
    if (!haveNext && !m_plOutThisFrame) { m_plFrame++; return; } // nothing for the worker to do
    // The worker will stage the next frame's chunk in the buffer an insert of two frames back was made from: if that insert
    // failed, its repair needs the buffer as it is -- look at its outcome first (the device passed it most of a frame ago)
    if (haveNext) pipelineCheckInsert((int)((m_plFrame + 1u) & 1u), true);
    {
        std::lock_guard<std::mutex> l(m_plMutex);
        m_plJob.haveOut = m_plOutThisFrame;
        m_plJob.outTag = m_plOutTag;
        m_plJob.outMost = m_plOutMost;
        m_plJob.outSlot = (int)(m_plFrame & 1u);
        m_plJob.haveNext = haveNext;
        m_plJob.inSlot = (int)((m_plFrame + 1u) & 1u);
        m_plJob.nextPos = nextPosCamera;
        m_plJob.nextRadius = nextRadius;
        m_plPosted.fetch_add(1u, std::memory_order_release);
    }
    m_plCv.notify_one();
    m_plOutThisFrame = false;
    m_plDecisionValid = haveNext;
    m_plDecisionPos = nextPosCamera;
    m_plDecisionRadius = nextRadius;
    m_plFrame++;
}

void CUDASceneRepChunkGrid::pipelineReturn(const StreamDecision& d, const vh::vec3f& posCamera, float radius)
{
    m_plDecision = d;
    m_plDecisionValid = true;
    m_plDecisionPos = posCamera;
    m_plDecisionRadius = radius;
}

void CUDASceneRepChunkGrid::pipelineDrain(bool undo)
{
    if (!m_plStarted) return;
    const unsigned int posted = m_plPosted.load(std::memory_order_acquire);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned int spins = 0;
    while (m_plDone.load(std::memory_order_acquire) != posted) {
        if ((++spins & 0xfffu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(40))
            throw vh::Error(VH_ERR_TIMEOUT, "streaming pipeline: the worker thread did not finish for 40 s");
    }
    if (undo && m_plDecisionValid) {
        // a choice nobody will use: the chunk goes back into the grid (its bit with it; the device's copy still has it set)
        m_plDecisionValid = false;
        if (m_plDecision.nIn != 0u) integrateInChunkGrid(h_plInDesc[m_plDecision.slot], h_plInBlocks[m_plDecision.slot], m_plDecision.nIn);
    }
    pipelineCheckInsert(0, undo);
    pipelineCheckInsert(1, undo);
    const int err = m_plError.exchange(0, std::memory_order_acq_rel);
    if (err != 0) throw vh::Error(err, std::string("streaming pipeline (worker): ") + vh_error_string(err));
}

void CUDASceneRepChunkGrid::pipelineTotals(unsigned long long* blocksOut, unsigned long long* blocksIn)
{
    pipelineDrain(false);
    pipelineCheckInsert(0, true);
    pipelineCheckInsert(1, true);
    if (blocksOut) *blocksOut = m_plBlocksOut.load();
    if (blocksIn) *blocksIn = m_plBlocksIn.load();
}

// ---------------------------------------------------------------------------
// .hashgrid (DSC/CUDASceneRepChunkGrid.h:456-548; mLib BinaryDataStreamFile:
// raw little-endian PODs, std::vector<T> = UINT64 count + raw elements,
// binaryDataStream.h:156-163,273-281; vec3 = 3 raw scalars)
// ---------------------------------------------------------------------------

namespace {
const unsigned int kHashGridVersion = 1;
template <class T> bool wr(FILE* f, const T& v) { return std::fwrite(&v, sizeof(T), 1, f) == 1; }
template <class T> bool rd(FILE* f, T& v) { return std::fread(&v, sizeof(T), 1, f) == 1; }
} // namespace

void CUDASceneRepChunkGrid::saveToFile(const std::string& filename, const vh::vec3f& camPos, float radius)
{
    const bool wasRunning = !s_terminateThread;
    stopMultiThreading();
    struct Restart {
        CUDASceneRepChunkGrid* g; bool on;
        ~Restart() { if (on) g->startMultiThreading(); }
    } restart{ this, wasRunning };
    streamOutToCPUAll();

    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) throw vh::Error(VH_ERR_IO, "cannot open " + filename);
    bool ok = true;
    {
        std::lock_guard<std::mutex> l(m_gridMutex);
        const float voxelSize = m_sceneRepHashSDF->getHashParams().m_virtualVoxelSize;
        ok = ok && wr(f, kHashGridVersion) && wr(f, voxelSize) && wr(f, m_voxelExtents) && wr(f, m_gridDimensions) && wr(f, m_minGridPos) &&
             wr(f, m_maxGridPos) && wr(f, m_initialChunkDescListSize);
        std::vector<unsigned int> keys;
        for (auto& kv : m_grid) keys.push_back(kv.first);
        std::sort(keys.begin(), keys.end());
        const unsigned int numOccupiedChunks = (unsigned int)keys.size();
        ok = ok && wr(f, numOccupiedChunks);
        for (unsigned int k : keys) {
            const ChunkDesc& c = *m_grid.at(k);
            const uint64_t nb = c.getSDFBlocks().size(), nd = c.getSDFBlockDescs().size();
            ok = ok && wr(f, k) && wr(f, nb);
            if (nb) ok = ok && std::fwrite(c.getSDFBlocks().data(), sizeof(vh::SDFBlock), nb, f) == nb;
            ok = ok && wr(f, nd);
            if (nd) ok = ok && std::fwrite(c.getSDFBlockDescs().data(), sizeof(SDFBlockDesc), nd, f) == nd;
        }
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) throw vh::Error(VH_ERR_IO, "write error on " + filename);

    unsigned int nStreamedBlocks;
    streamInToGPUAll(camPos, radius, true, nStreamedBlocks);
}

void CUDASceneRepChunkGrid::loadFromFile(const std::string& filename, const vh::vec3f& camPos, float radius)
{
    (void)camPos; (void)radius;
    const bool wasRunning = !s_terminateThread;
    stopMultiThreading();
    // whatever happens below, the worker thread runs again afterwards if it ran before
    struct Restart {
        CUDASceneRepChunkGrid* g; bool on;
        ~Restart() { if (on) g->startMultiThreading(); }
    } restart{ this, wasRunning };
    streamOutToCPUAll();
    clearGrid();
    {
        std::lock_guard<std::mutex> l(m_gridMutex);
        std::fill(m_bitMask.begin(), m_bitMask.end(), 0u);
        m_bitMaskDirty = true;
    }

    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) throw vh::Error(VH_ERR_IO, "cannot open " + filename);
    struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{ f };
    // no count in the file is trusted beyond what the file can hold
    if (std::fseek(f, 0, SEEK_END) != 0) throw vh::Error(VH_ERR_IO, "cannot seek in " + filename);
    const long fileSizeL = std::ftell(f);
    if (fileSizeL < 0 || std::fseek(f, 0, SEEK_SET) != 0) throw vh::Error(VH_ERR_IO, "cannot seek in " + filename);
    const uint64_t fileSize = (uint64_t)fileSizeL;
    auto remaining = [&]() -> uint64_t {
        const long at = std::ftell(f);
        return (at < 0 || (uint64_t)at > fileSize) ? 0u : fileSize - (uint64_t)at;
    };

    unsigned int version = 0, listSize = 0, numOccupiedChunks = 0;
    float voxelSize = 0.0f;
    vh::vec3f ext = { 0, 0, 0 };
    vh::vec3i dims = { 0, 0, 0 }, minPos = { 0, 0, 0 }, maxPos = { 0, 0, 0 };
    if (!(rd(f, version) && rd(f, voxelSize) && rd(f, ext) && rd(f, dims) && rd(f, minPos) && rd(f, maxPos) && rd(f, listSize)))
        throw vh::Error(VH_ERR_IO, "invalid read; probably wrong file name (" + filename + ")?");
    if (version != kHashGridVersion)
        throw vh::Error(VH_ERR_VERSION_MISMATCH, "hashgrid versions don't match - found " + std::to_string(version) + " should be " + std::to_string(kHashGridVersion));
    if (!rd(f, numOccupiedChunks)) throw vh::Error(VH_ERR_IO, "invalid read");
    if (ext.x != m_voxelExtents.x || ext.y != m_voxelExtents.y || ext.z != m_voxelExtents.z) throw vh::Error(VH_ERR_BAD_ARGUMENT, "voxel extends don't match");
    if (dims.x != m_gridDimensions.x || dims.y != m_gridDimensions.y || dims.z != m_gridDimensions.z) throw vh::Error(VH_ERR_BAD_ARGUMENT, "grid dimensions don't match");
    if (minPos.x != m_minGridPos.x || minPos.y != m_minGridPos.y || minPos.z != m_minGridPos.z) throw vh::Error(VH_ERR_BAD_ARGUMENT, "minGridPos doesn't match");
    if (maxPos.x != m_maxGridPos.x || maxPos.y != m_maxGridPos.y || maxPos.z != m_maxGridPos.z) throw vh::Error(VH_ERR_BAD_ARGUMENT, "maxGridPos doesn't match");
    if (listSize != m_initialChunkDescListSize) throw vh::Error(VH_ERR_BAD_ARGUMENT, "initial chunkListSize doesn't match");

    {
        std::lock_guard<std::mutex> l(m_gridMutex);
        const size_t nChunks = (size_t)m_gridDimensions.x * m_gridDimensions.y * m_gridDimensions.z;
        try {
            for (unsigned int i = 0; i < numOccupiedChunks; i++) {
                unsigned int index = 0;
                uint64_t nb = 0, nd = 0;
                if (!rd(f, index) || index >= nChunks) throw vh::Error(VH_ERR_IO, "invalid chunk index");
                if (m_grid.count(index)) throw vh::Error(VH_ERR_IO, "chunk listed twice");
                std::unique_ptr<ChunkDesc> c(new ChunkDesc(m_initialChunkDescListSize));
                if (!rd(f, nb) || nb > remaining() / sizeof(vh::SDFBlock)) throw vh::Error(VH_ERR_IO, "invalid read: more blocks than the file holds");
                c->getSDFBlocks().resize(nb);
                if (nb && std::fread(c->getSDFBlocks().data(), sizeof(vh::SDFBlock), nb, f) != nb) throw vh::Error(VH_ERR_IO, "invalid read");
                if (!rd(f, nd) || nd > remaining() / sizeof(SDFBlockDesc)) throw vh::Error(VH_ERR_IO, "invalid read: more descriptors than the file holds");
                // the two vectors of a chunk are parallel (ChunkDesc::addSDFBlock pushes to both): everything
                // downstream copies nb descriptors
                if (nd != nb) throw vh::Error(VH_ERR_IO, "invalid chunk: " + std::to_string(nb) + " blocks but " + std::to_string(nd) + " descriptors");
                c->getSDFBlockDescs().resize(nd);
                if (nd && std::fread(c->getSDFBlockDescs().data(), sizeof(SDFBlockDesc), nd, f) != nd) throw vh::Error(VH_ERR_IO, "invalid read");
                // the reference leaves the bit mask cleared after a load (alloc may then re-create blocks
                // of streamed-out chunks); here the mask follows the grid content
                if (c->isStreamedOut()) setBit(index);
                m_grid[index] = std::move(c);
            }
        } catch (...) { // a refused file leaves an empty grid, not half of one
            m_grid.clear();
            std::fill(m_bitMask.begin(), m_bitMask.end(), 0u);
            m_bitMaskDirty = true;
            throw;
        }
    }
}
