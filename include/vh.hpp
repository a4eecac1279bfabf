// vh.hpp -- C++ host classes of the MI355X voxel-hashing engine.
//
// Same class names, constructors and public methods as the reference's
// CUDASceneRepHashSDF (DSC/CUDASceneRepHashSDF.h:28-350), CUDARayCastSDF
// (DSC/CUDARayCastSDF.h:13-69) and CUDASceneRepChunkGrid
// (DSC/CUDASceneRepChunkGrid.h:152-753), so the reference's frame loop
// (DSC/DepthSensing.cpp:720-924) compiles against them after swapping the
// includes.  Differences, all forced by dropping Windows/D3D/mLib:
//   - mat4f / vec3f / vec3i are small PODs defined here (row-major 4x4);
//   - GlobalAppState is gone: the five flags it supplied are a VhSceneOptions;
//   - DepthCameraParams is passed explicitly where the reference read the
//     process-global __constant__ c_depthCameraParams;
//   - errors throw vh::Error (the reference aborts via cutilSafeCall/exit);
//   - a HIP stream can be given; all work of one instance is issued on it.
//
// This header needs no HIP headers; link against libvoxelhashing_amd.so.
//   DSC/ = /root/reference/DepthSensingCUDA/Source/
#ifndef VH_HPP
#define VH_HPP

#include <condition_variable>
#include <cstdint>
#include <memory>
#include <atomic>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "vh_api.h"

typedef VhHashEntry HashEntry;
typedef VhVoxel Voxel;
typedef VhHashParams HashParams;
typedef VhHashData HashData;
typedef VhDepthCameraParams DepthCameraParams;
typedef VhDepthCameraData DepthCameraData;
typedef VhRayCastParams RayCastParams;
typedef VhRayCastData RayCastData;
typedef VhSDFBlockDesc SDFBlockDesc;

namespace vh {

struct Error : public std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

struct vec3f { float x, y, z; };
struct vec3i { int x, y, z; };

// Row-major 4x4, entries m[0..15] = m11,m12,...,m44 (the layout of the
// reference's float4x4 / mLib mat4f).
struct mat4f {
    float m[16];
    static mat4f identity();
    mat4f getInverse() const; // float4x4::getInverse, DSC/cuda_SimpleMatrixUtil.h:944-1069
    mat4f operator*(const mat4f& o) const;
    vec3f transformPoint(const vec3f& p) const; // float4x4 * float3, :900-907
    float& operator()(int r, int c) { return m[4 * r + c]; }
    float operator()(int r, int c) const { return m[4 * r + c]; }
};

struct SDFBlock { // DSC/CUDASceneRepChunkGrid.h:11-21
    Voxel data[VH_SDF_BLOCK_VOXELS];
};

} // namespace vh

struct VhStageTimer; // opaque: per-stage HIP event pairs

// ---------------------------------------------------------------------------
class CUDASceneRepHashSDF {
public:
    explicit CUDASceneRepHashSDF(const HashParams& params);
    CUDASceneRepHashSDF(const HashParams& params, const VhSceneOptions& options, vhStream_t stream = nullptr);
    ~CUDASceneRepHashSDF();
    CUDASceneRepHashSDF(const CUDASceneRepHashSDF&) = delete;
    CUDASceneRepHashSDF& operator=(const CUDASceneRepHashSDF&) = delete;

    static VhSceneOptions defaultOptions();

    // no-op kept for source compatibility (DSC/CUDASceneRepHashSDF.h:60-62)
    void bindDepthCameraTextures(const DepthCameraData&) {}

    // DSC/CUDASceneRepHashSDF.h:64-83
    void integrate(const vh::mat4f& lastRigidTransform, const DepthCameraData& depthCameraData,
                   const DepthCameraParams& depthCameraParams, const unsigned int* d_bitMask);
    void setLastRigidTransform(const vh::mat4f& lastRigidTransform);                       // :85
    void setLastRigidTransformAndCompactify(const vh::mat4f& lastRigidTransform,
                                            const DepthCameraParams& depthCameraParams);   // :90
    const vh::mat4f getLastRigidTransform() const;                                         // :96
    void reset();                                                                          // :101
    HashData& getHashData() { return m_hashData; }                                         // :112
    // m_numOccupiedBlocks is exact in offline mode; in online mode it is the
    // most recent count whose asynchronous read-back has completed.
    const HashParams& getHashParams();                                                     // :116
    unsigned int getHeapFreeCount();                                                       // :122 (blocking)
    unsigned int getNumOccupiedBlocks();                                                   // blocking, exact
    // :129-233; throws vh::Error on a violated invariant.  report = {numOccupied, numFree, duplicates, lockEntries}
    void debugHash(unsigned int report[4] = nullptr);

    // additions
    // The two halves of integrate() for a frame loop that knows the pose of a frame before it ray-casts the previous
    // one (a recorded trajectory: s_binaryDumpSensorUseTrajectory, DSC/DepthSensing.cpp:733-747).  integrateAhead()
    // sets the pose and returns the frame's alloc + compactify passes as a job; CUDARayCastSDF::render(..., job)
    // launches them INSIDE its own two launches (the alloc pass behind the ray caster's workgroups, where it fills the
    // tail the dearest tiles leave; the compactify pass behind computeNormals'); integrateFinish() launches whatever
    // of the job is still open and then the pass over the voxels.  The maps stay the same: a block allocated while
    // rays are marched holds only unobserved voxels (weight 0), which a sample treats exactly like an absent block
    // (DESIGN.md section 3).  Offline mode and the reference launch sequence need their blocking read-backs: there the
    // job is never handed out.
    VhFrameJob* integrateAhead(const vh::mat4f& lastRigidTransform, const DepthCameraData& depthCameraData,
                               const DepthCameraParams& depthCameraParams, const unsigned int* d_bitMask);
    void integrateFinish(const DepthCameraData& depthCameraData, const DepthCameraParams& depthCameraParams);
    // gives up a job of integrateAhead() that will not be finished (the caller is unwinding): the scene accepts
    // integrate() / integrateAhead() again.  Blocks an alloc pass that already ran has added stay: they are empty, and
    // garbage collection takes them like any other unobserved block.
    void abortAhead();
    // frames whose pass over the voxels has started on the device (read from mapped host memory: no synchronisation)
    unsigned int getNumFramesStartedOnDevice() const;
    // whoever edits the table outside integrate() (streaming) says so: work prepared from the table before is void
    void noteTableEdited() { m_tableEpoch++; }
    unsigned int getTableEpoch() const { return m_tableEpoch; }
    void setOptions(const VhSceneOptions& o) { m_options = o; }
    const VhSceneOptions& getOptions() const { return m_options; }
    vhStream_t getStream() const { return m_stream; }
    int32_t nextLockToken(); // fresh bucket-lock epoch (replaces resetHashBucketMutexCUDA)
    void getState(uint32_t out[VH_STATE_WORDS]);
    void getTimings(double out[4]); // ms: alloc, compactify, integrate(+gc), frames
    unsigned int getNumIntegratedFrames() const { return m_numIntegratedFrames; }

private:
    void create(const HashParams& params);
    void destroy();
    void alloc(const DepthCameraData&, const DepthCameraParams&, const unsigned int* d_bitMask, bool jobPrepared); // :247
    void compactifyHashEntries(const DepthCameraParams&);                                        // :282
    void integrateFused(const DepthCameraData&, const DepthCameraParams&);
    void integrateDepthMap(const DepthCameraData&, const DepthCameraParams&);                    // :317
    void garbageCollect(const DepthCameraParams&);                                               // :327
    void pollOccupiedCount(bool block);

    HashParams m_hashParams;
    HashData m_hashData;
    VhSceneOptions m_options;
    vhStream_t m_stream;
    unsigned int m_numIntegratedFrames;
    int32_t m_lockEpoch;
    uint32_t* h_occupied;     // mapped pinned words: the fused integrate kernel mirrors {block count, frame number} here
    void* m_occupiedEvent;    // device alias of h_occupied
    bool m_occupiedPending;   // a frame was enqueued since the host value was last known exact
    bool m_counterCleared;    // d_hashCompactifiedCounter is known to be 0 (k_alloc clears it)
    VhStageTimer* m_timer;
    unsigned int m_tableEpoch;
    VhFrameJob m_job;         // alloc + compactify of the frame in progress
    int m_aheadPending;       // 0 none, 1 job prepared by integrateAhead()
    void* d_packedFrame;      // the frame as the alloc pass packs it for the pass over the voxels (8 bytes per pixel)
    size_t m_packedPixels;
    uint32_t* d_riderDone;    // VH_RIDER_DONE_WORDS words (see VhFrameJob::d_riderDone)
    uint32_t m_riderTotals[2]; // VhFrameJob::listDoneTotal, listClassTotal as of the last launch
    void keepRiderTotals();
    unsigned int m_riderMostBlocks;
    uint32_t fusedFlags() const;
    void prepareJob(const DepthCameraData&, const DepthCameraParams&, const unsigned int* d_bitMask);
};

// ---------------------------------------------------------------------------
class CUDARayCastSDF {
public:
    explicit CUDARayCastSDF(const RayCastParams& params, vhStream_t stream = nullptr);
    ~CUDARayCastSDF();
    CUDARayCastSDF(const CUDARayCastSDF&) = delete;
    CUDARayCastSDF& operator=(const CUDARayCastSDF&) = delete;

    // DSC/CUDARayCastSDF.cpp:38-72.  The reference skips the view-matrix
    // update while hashParams.m_numOccupiedBlocks == 0 (stale matrices); here
    // the given transform is always used.
    // coLaunch (not in the reference): a job of CUDASceneRepHashSDF::integrateAhead, whose alloc and compactify
    // passes then ride in this call's two launches
    void render(const HashData& hashData, const HashParams& hashParams, const DepthCameraParams& cameraParams,
                const vh::mat4f& lastRigidTransform, VhFrameJob* coLaunch = nullptr);
    const RayCastData& getRayCastData() { return m_data; }        // :42
    const RayCastParams& getRayCastParams() const { return m_params; } // :45

    // marchOnly: events around the march kernel only; stride n: only every n-th render() is timed
    void setTiming(bool on, bool marchOnly = false, unsigned int stride = 1);
    void getTimings(double out[4]); // ms: raycast (the march kernel), normals, frames, interval splat
    double getEventPairOverheadMs(); // average reading of an empty HIP event pair (taken while every stage is timed)
    // ray-interval splatting (DSC/CUDARayCastSDF.cpp:84-100, disabled in the reference fork): on by default here,
    // as a conservative compute pass that leaves every output bit unchanged
    void setIntervalSplatting(bool on) { m_useIntervals = on; }
    // render() calls so far that ran on an interval splat made ahead (inside the previous call's computeNormals launch)
    unsigned int getNumSplatsMadeAheadUsed() const { return m_preSplatsUsed; }

private:
    RayCastParams m_params;
    RayCastData m_data;
    vhStream_t m_stream;
    VhStageTimer* m_timer;
    bool m_timeMarchOnly;
    unsigned int m_timeStride, m_renderCalls;
    uint32_t* d_tileHeads;     // {min, max camera depth, block count, 0} per 8x8-pixel tile
    VhTileBlock* d_tileBlocks; // up to VH_TILE_LIST_CAPACITY_LARGE blocks per tile
    uint32_t *h_longestList, *d_longestList; // mapped host word: longest tile list the ray caster met lately
    bool m_largeTables;        // current choice of table size
    uint32_t m_quietFrames, m_tileCapacity;
    uint32_t* d_schedule;      // tiles by cost class, for the launch order of the next render()
    uint32_t m_phase;          // render() calls with intervals so far
    unsigned int m_preSplatsUsed;
    bool m_useIntervals;
    // the interval splat of the next render, made ahead inside this render's computeNormals launch (vh_compute_normals_co2)
    struct PreSplat {
        bool valid;
        float pose[16];         // the pose it was made for
        const void* table;      // d_hash of the scene
        uint32_t frameNumber, tableEpoch, phase, capacity;
    } m_preSplat;
};

// ---------------------------------------------------------------------------
class ChunkDesc { // DSC/CUDASceneRepChunkGrid.h:68-121
public:
    explicit ChunkDesc(unsigned int initialChunkListSize)
    {
        m_SDFBlocks.reserve(initialChunkListSize);
        m_ChunkDesc.reserve(initialChunkListSize);
    }
    void addSDFBlock(const SDFBlockDesc& desc, const vh::SDFBlock& data)
    {
        m_ChunkDesc.push_back(desc);
        m_SDFBlocks.push_back(data);
    }
    unsigned int getNElements() const { return (unsigned int)m_SDFBlocks.size(); }
    void clear() { m_ChunkDesc.clear(); m_SDFBlocks.clear(); }
    bool isStreamedOut() const { return !m_SDFBlocks.empty(); }
    std::vector<SDFBlockDesc>& getSDFBlockDescs() { return m_ChunkDesc; }
    std::vector<vh::SDFBlock>& getSDFBlocks() { return m_SDFBlocks; }
    const std::vector<SDFBlockDesc>& getSDFBlockDescs() const { return m_ChunkDesc; }
    const std::vector<vh::SDFBlock>& getSDFBlocks() const { return m_SDFBlocks; }

private:
    std::vector<vh::SDFBlock> m_SDFBlocks;
    std::vector<SDFBlockDesc> m_ChunkDesc;
};

class CUDASceneRepChunkGrid {
public:
    // DSC/CUDASceneRepChunkGrid.h:155
    CUDASceneRepChunkGrid(CUDASceneRepHashSDF* sceneRepHashSDF, const vh::vec3f& voxelExtends,
                          const vh::vec3i& gridDimensions, const vh::vec3i& minGridPos,
                          unsigned int initialChunkListSize, bool streamingEnabled, unsigned int streamOutParts);
    ~CUDASceneRepChunkGrid();
    CUDASceneRepChunkGrid(const CUDASceneRepChunkGrid&) = delete;
    CUDASceneRepChunkGrid& operator=(const CUDASceneRepChunkGrid&) = delete;

    // stream out (GPU -> host), DSC/CUDASceneRepChunkGrid.cpp:31-153
    void streamOutToCPUAll();
    void streamOutToCPU(const vh::vec3f& posCamera, float radius, bool useParts, unsigned int& nStreamedBlocks);
    void streamOutToCPUPass0GPU(const vh::vec3f& posCamera, float radius, bool useParts, bool multiThreaded = true);
    void streamOutToCPUPass1CPU(bool multiThreaded = true);
    void integrateInChunkGrid(const SDFBlockDesc* desc, const vh::SDFBlock* block, unsigned int nSDFBlocks);

    // stream in (host -> GPU), DSC/CUDASceneRepChunkGrid.cpp:155-311
    void streamInToGPUAll();
    void streamInToGPUAll(const vh::vec3f& posCamera, float radius, bool useParts, unsigned int& nStreamedBlocks);
    void streamInToGPUChunk(const vh::vec3i& chunkPos);
    void streamInToGPUChunkNeighborhood(const vh::vec3i& chunkPos, int kernelRadius);
    void streamInToGPU(const vh::vec3f& posCamera, float radius, bool useParts, unsigned int& nStreamedBlocks);
    void streamInToGPUPass0CPU(const vh::vec3f& posCamera, float radius, bool useParts, bool multiThreaded = true);
    void streamInToGPUPass1GPU(bool multiThreaded = true);

    // ---- the streaming step of a frame in which nothing streams, kept out of the frame's launches (no reference twin;
    // used by Reconstruction when it knows the next frame's pose).  probeStreamOut() enqueues the scan the NEXT stream-out
    // pass will make, counting only; probeResult() waits for its answer.  With the answer 0, streamOutNothing() is that
    // pass without its launches (same protocol with the worker thread, same part counter), and streamInWait() /
    // streamInFinish() are the two halves of streamInToGPUPass1GPU(true): the worker's answer first, the launches -- if
    // any block comes in -- wherever the caller's launch order wants them.
    void probeStreamOut(const vh::vec3f& posCamera, float radius, bool useParts);
    unsigned int probeResult();
    void streamOutNothing(const vh::vec3f& posCamera, float radius, bool useParts);
    unsigned int streamInWait();
    void streamInFinish();
    // after streamInWait(), instead of streamInFinish(): the blocks the worker has staged go back into the host grid,
    // nothing is launched, the worker gets its buffers and its event (for a caller that is unwinding)
    void streamInAbort();
    bool bitMaskDirty() const { return m_bitMaskDirty; }
    unsigned int integrateInHash(const vh::vec3f& posCamera, float radius, bool useParts);

    // ---- the streaming step of a frame without a host wait (not in the reference; used by Reconstruction for a sequence
    // whose poses are known a frame ahead).  The reference's step is a chain of four host <-> device round trips: the
    // stream-out count is read back before the blocks are copied, the host puts the blocks into its grid and only then knows
    // the bit mask the frame's alloc pass needs, the worker picks the chunk that comes in, the heap counter is read back
    // before it is inserted.  Here
    //   * the counts stay on the device (vh_stream_out_device / vh_stream_in_device), and the stream-out pass writes the
    //     blocks straight into mapped host memory;
    //   * the DEVICE's copy of the bit mask is kept by the passes themselves (a block that leaves sets its chunk's bit, a
    //     chunk that comes in clears it): what the frame's alloc pass reads is in place without the host;
    //   * the chunk that comes in at frame k+1 is chosen and uploaded by a worker thread while the device is still working
    //     on frame k.  It may be chosen that early: a chunk gains blocks at frame k+1 only from blocks OUTSIDE that frame's
    //     sphere, and a chunk with such a block does not lie entirely inside the sphere, so what leaves in a frame cannot
    //     be what comes in in the same frame (isChunkInSphere is the conservative test of the chunk's bounding sphere);
    //   * the blocks that left at frame k are put into the host grid by that worker when their copy has arrived, before it
    //     chooses for frame k+1 -- the one order the choice does depend on.
    // Per frame: pipelineDecision() (the worker's answer for this frame) -> [the ray cast] -> pipelineStreamOut() ->
    // pipelineStreamIn() -> [integrate with getBitMaskDevice()] -> pipelineAsk() (the job for the next frame).
    // Every other entry point of this class drains the pipeline first (pipelineDrain()).
    struct StreamDecision {
        unsigned int nIn;      // blocks of the chunk that comes in (0: none)
        unsigned int chunkBit; // its bit in the bit mask
        int slot;              // which staging buffer holds them
    };
    bool pipelineHasDecision(const vh::vec3f& posCamera, float radius) const; // was pipelineAsk() called for this frame's sphere?
    StreamDecision pipelineDecision();  // waits for the worker (it has had most of a frame)
    // enqueues the frame's stream-out pass for at most `mostBlocks` blocks (vh_stream_out_probe's count; 0: nothing is
    // launched) and advances the part counter; false if mostBlocks exceeds the pipeline's staging (the caller then takes the
    // reference's order of calls: pipelineDrain() and the classic methods)
    bool pipelineStreamOut(const vh::vec3f& posCamera, float radius, bool useParts, unsigned int mostBlocks);
    void pipelineStreamIn(const StreamDecision& d); // enqueues the insert of what the worker has uploaded
    // the job for the worker: take in what this frame's stream-out pass copies, then (haveNext) choose for the next frame
    void pipelineAsk(bool haveNext, const vh::vec3f& nextPosCamera, float nextRadius);
    // waits for the worker's job and looks at the outcome of finished inserts (repairs failures).  undo: a choice that has
    // been made for a coming frame is put back into the grid (whoever looks at the host grid or takes the reference's order
    // of calls next must find it as the reference would have it at this point; the next frame then chooses again)
    void pipelineDrain(bool undo = true);
    void pipelineReturn(const StreamDecision& d, const vh::vec3f& posCamera, float radius); // hands an unused pipelineDecision() back
    unsigned int pipelineCapacity() const { return (unsigned int)kPipelineBlocks; }
    unsigned int* getBitMaskDevice() { return d_bitMask; } // (current while the pipeline runs: no upload)
    void pipelineTotals(unsigned long long* blocksOut, unsigned long long* blocksIn); // drains; blocks moved by the pipeline so far
    // blocks that stream-in passes could not insert and that went back to the host grid (not in the reference)
    unsigned int getNumFailedInserts() const { return m_numFailedInserts; }

    void debugCheckForDuplicates() const; // .cpp:313-341; throws vh::Error
    void startMultiThreading();           // .h:248
    void stopMultiThreading();            // .h:262
    void clearGrid();                     // .h:291
    void reset();                         // .h:297
    // .h:306 -- uploads the bit mask only if it changed since the last call
    unsigned int* getBitMaskGPU();
    bool containsSDFBlocksChunk(const vh::vec3i& chunk) const;                               // .h:311
    bool isChunkInSphere(const vh::vec3i& chunk, const vh::vec3f& center, float radius) const; // .h:317
    bool containsSDFBlocksChunkInRadius(const vh::vec3i& chunk, int chunkRadius) const;      // .h:348

    HashData& getHashData() { return m_sceneRepHashSDF->getHashData(); }               // .h:287
    const HashParams& getHashParams() { return m_sceneRepHashSDF->getHashParams(); }   // .h:283
    const vh::vec3i& getMinGridPos() const { return m_minGridPos; }
    const vh::vec3i& getMaxGridPos() const { return m_maxGridPos; }
    const vh::vec3f& getVoxelExtends() const { return m_voxelExtents; }
    vh::vec3f getWorldPosChunk(const vh::vec3i& chunk) const { return chunkToWorld(chunk); }
    void getStatistics(unsigned int out[3]) const; // chunks, blocks on host, bits set

    // .hashgrid version 1, DSC/CUDASceneRepChunkGrid.h:456-548 (mLib BinaryDataStreamFile layout)
    void saveToFile(const std::string& filename, const vh::vec3f& camPos, float radius);
    void loadFromFile(const std::string& filename, const vh::vec3f& camPos, float radius);

    unsigned int getNumStreamedOutBlocks() const { return s_nStreamdOutBlocks; } // of the last pass 0
    unsigned int getNumStreamedInBlocks() const { return s_nStreamdInBlocks; }   // of the last pass 1
    const vh::vec3f& getPosCamera() const { return s_posCamera; }
    float getRadius() const { return s_radius; }
    bool getTerminatedThread() const { return s_terminateThread; }
    static const bool s_useParts = true;

    // host-side content (sorted by chunk index, insertion order inside a chunk)
    void downloadHostBlocks(std::vector<SDFBlockDesc>& descs, std::vector<vh::SDFBlock>& blocks) const;

    // helpers (private in the reference, .h:560-640)
    bool isValidChunk(const vh::vec3i& chunk) const;
    vh::vec3i worldToChunks(const vh::vec3f& posWorld) const;
    vh::vec3f chunkToWorld(const vh::vec3i& posChunk) const;
    vh::vec3i delinearizeChunkIndex(unsigned int idx) const;
    unsigned int linearizeChunkPos(const vh::vec3i& chunkPos) const;
    vh::vec3i meterToNumberOfChunksCeil(float f) const;
    float getChunkRadiusInMeter() const;
    float getGridRadiusInMeter() const;

private:
    struct AutoResetEvent { // Win32 auto-reset event (CreateEvent(NULL, FALSE, initial, NULL))
        std::mutex mtx;
        std::condition_variable cv;
        std::atomic<bool> signaled{ false };
        void set();
        void wait();
        void reset(bool state);
    };
    void workerLoop();
    void create(const vh::vec3f& voxelExtends, const vh::vec3i& gridDimensions, const vh::vec3i& minGridPos,
                unsigned int initialChunkListSize, bool streamingEnabled);
    void destroy();
    void setBit(unsigned int index);
    void resetBit(unsigned int index);
    void takeBackFailedInserts(unsigned int nFailed, unsigned int heapCountPrev);
    void takeBackFailedInserts(unsigned int nFailed, unsigned int heapCountPrev, const SDFBlockDesc* descs, const vh::SDFBlock* blocks, unsigned int& nIn);
    unsigned int integrateInHash(const vh::vec3f& posCamera, float radius, bool useParts, SDFBlockDesc* hDescs, vh::SDFBlock* hBlocks,
                                 SDFBlockDesc* dDescs, vh::SDFBlock* dBlocks, unsigned int capacity, unsigned int* chunkBit);
    // the pipeline (see above)
    enum { kPipelineBlocks = 4096 }; // staging capacity of a pipelined pass (16 MB per buffer)
    struct PipelineJob {
        bool haveOut, haveNext;
        uint32_t outTag, outMost;
        int outSlot, inSlot;
        vh::vec3f nextPos;
        float nextRadius;
    };
    void pipelineStart();
    void pipelineStop();
    void pipelineWorker();
    void pipelineCheckInsert(int slot, bool block);
    std::thread m_plThread;
    std::mutex m_plMutex;
    std::condition_variable m_plCv;
    bool m_plStarted, m_plQuit;
    PipelineJob m_plJob;
    std::atomic<unsigned int> m_plPosted, m_plDone; // jobs handed to / finished by the worker
    std::atomic<int> m_plError;                     // a vh error code the worker ran into (reported by the next call)
    StreamDecision m_plDecision;                    // the finished job's choice
    bool m_plDecisionValid;                         // ... if it was asked for
    vh::vec3f m_plDecisionPos;                      // ... and the sphere it was asked for
    float m_plDecisionRadius;
    unsigned int m_plFrame;                         // frames the pipeline has run (slot = frame & 1)
    bool m_plOutThisFrame;                          // pipelineStreamOut() of the current frame launched a pass
    uint32_t m_plOutTag, m_plOutMost;
    struct { bool pending; uint32_t tag; unsigned int nIn, chunkBit; } m_plInsert[2]; // per staging slot: an insert whose outcome has not been looked at
    std::atomic<unsigned long long> m_plBlocksOut, m_plBlocksIn;
    SDFBlockDesc* d_plOutDesc[2];      // pass 1 -> pass 2 (device)
    SDFBlockDesc* h_plOutDesc[2];      // mapped pinned: pass 2 writes the blocks it moves straight to the host
    vh::SDFBlock* h_plOutBlocks[2];
    SDFBlockDesc* hd_plOutDesc[2];     // their device aliases
    vh::SDFBlock* hd_plOutBlocks[2];
    uint32_t* h_plOutMirror[2];        // mapped pinned {count, 0, tag}
    uint32_t* hd_plOutMirror[2];
    SDFBlockDesc* h_plInDesc[2];       // pinned staging of the worker's upload
    vh::SDFBlock* h_plInBlocks[2];
    SDFBlockDesc* d_plInDesc[2];
    vh::SDFBlock* d_plInBlocks[2];
    uint32_t* h_plInMirror[2];         // mapped pinned {failed, heap counter before, tag, exhausted}, per staging slot
    uint32_t* hd_plInMirror[2];
    uint32_t m_plTag;
    void streamInLaunches();
    unsigned int m_numFailedInserts;

    unsigned int m_maxNumberOfSDFBlocksIntegrateFromGlobalHash;

    SDFBlockDesc* h_SDFBlockDescOutput; // pinned
    vh::SDFBlock* h_SDFBlockOutput;     // pinned
    SDFBlockDesc* h_SDFBlockDescInput;  // pinned staging for the worker's H2D
    vh::SDFBlock* h_SDFBlockInput;      // pinned
    uint32_t* h_counter;                // pinned
    uint32_t* h_mirror;                 // mapped pinned: {word 0, word 1, tag} published by the device (vh_publish_words)
    uint32_t* d_mirror;                 // its device alias
    uint32_t m_mirrorTag;
    uint32_t* h_probe;                  // mapped pinned: {count, 0, tag} of the stream-out probe
    uint32_t* d_probe;                  // its device alias
    uint32_t m_probeTag;
    unsigned int* d_probeCounter;
    std::unique_lock<std::mutex> m_streamInLock; // held between streamInWait() and streamInFinish()
    // values of up to two device words once the stream has reached this point, without a blocking driver call
    void readBack(const unsigned int* d_word0, const unsigned int* d_word1, unsigned int* out0, unsigned int* out1);
    SDFBlockDesc* d_SDFBlockDescOutput;
    SDFBlockDesc* d_SDFBlockDescInput;
    vh::SDFBlock* d_SDFBlockOutput;
    vh::SDFBlock* d_SDFBlockInput;
    unsigned int* d_SDFBlockCounter;
    unsigned int* d_insertFailed; // {count, indices ...} of the blocks a stream-in pass could not insert
    unsigned int* d_bitMask;
    void* m_copyStream; // hipStream_t of the worker thread
    int m_device;       // HIP device the scene lives on (the worker thread binds to it)

    vh::vec3f m_voxelExtents;
    vh::vec3i m_gridDimensions;
    vh::vec3i m_minGridPos;
    vh::vec3i m_maxGridPos;
    unsigned int m_initialChunkDescListSize;

    std::unordered_map<unsigned int, std::unique_ptr<ChunkDesc>> m_grid; // sparse: chunk index -> chunk
    std::vector<unsigned int> m_bitMask;                                 // BitArray<unsigned int>, DSC/BitArray.h
    bool m_bitMaskDirty;
    mutable std::mutex m_gridMutex;

    unsigned int m_currentPart;
    unsigned int m_streamOutParts;

    std::thread m_thread;
    std::mutex hMutexOut, hMutexIn;
    AutoResetEvent hEventOutProduce, hEventOutConsume, hEventInProduce, hEventInConsume;

    vh::vec3f s_posCamera;
    float s_radius;
    unsigned int s_nStreamdInBlocks;
    unsigned int s_nStreamdOutBlocks;
    volatile bool s_terminateThread;

    CUDASceneRepHashSDF* m_sceneRepHashSDF;
};


// ---------------------------------------------------------------------------
// The frame loop: reconstruction() of DSC/DepthSensing.cpp:720-924 for a recorded sequence at given poses, as a
// class over the three host classes above (the reference keeps them in globals: g_sceneRep, g_rayCast, g_chunkGrid).
typedef VhReconstructionOptions ReconstructionOptions;
typedef VhSequenceFrame SequenceFrame;
typedef VhReconstructionStats ReconstructionStats;

class Reconstruction {
public:
    Reconstruction(CUDASceneRepHashSDF* sceneRep, CUDARayCastSDF* rayCast, CUDASceneRepChunkGrid* chunkGrid,
                   const DepthCameraParams& depthCameraParams, const ReconstructionOptions& options);
    ~Reconstruction();
    Reconstruction(const Reconstruction&) = delete;
    Reconstruction& operator=(const Reconstruction&) = delete;

    static ReconstructionOptions defaultOptions();
    // next: the frame that will follow frames[n-1] in a later call, if the caller knows it (only its pose is read)
    void run(const SequenceFrame* frames, unsigned int n, const SequenceFrame* next = nullptr);
    void synchronize();
    void reset();
    const ReconstructionStats& getStats();
    // test hook: the n-th ray cast from now throws instead of running (0: off) -- the loop's unwinding is tested with it
    void debugFailRender(unsigned int nthRenderFromNow) { m_debugFailRender = nthRenderFromNow; }

private:
    unsigned int m_debugFailRender;
    unsigned long long m_pipelineOutSeen, m_pipelineInSeen; // the grid's pipeline totals already in m_stats
    void frame(const SequenceFrame& f, const SequenceFrame* next);
    bool m_probePending;        // a stream-out probe for the frame with pose m_probePose is in the stream
    float m_probePose[16];
    DepthCameraData upload(const SequenceFrame& f);

    CUDASceneRepHashSDF* m_sceneRep;
    CUDARayCastSDF* m_rayCast;
    CUDASceneRepChunkGrid* m_chunkGrid;
    DepthCameraParams m_cp;
    ReconstructionOptions m_opt;
    ReconstructionStats m_stats;
    unsigned int m_frameNumber;
    // frames on the host: a ring of staging slots fed by a copy stream
    enum { kStagingSlots = 4 };
    void* m_copyStream;
    float* d_stageDepth[kStagingSlots];
    unsigned char* d_stageColorRaw[kStagingSlots];
    float* d_stageColor[kStagingSlots];
    void* m_copyStream2;                          // the depth copy runs on a stream (a copy engine) of its own
    void* m_slotReady[kStagingSlots];             // copy stream -> main stream
    void* m_slotReady2[kStagingSlots];
    unsigned int m_slotSceneFrame[kStagingSlots]; // the scene's frame count when the slot's frame was enqueued, + 1 (0: never used)
    unsigned int m_uploads;                       // frames uploaded so far (the slot is m_uploads % kStagingSlots)
    std::vector<std::pair<void*, void*>> m_uploadTimers; // event pairs on the copy stream, not yet read
    std::vector<void*> m_timerPool;
};

// ---------------------------------------------------------------------------
// CUDAMarchingCubesHashSDF (DSC/CUDAMarchingCubesHashSDF.h:8-67, .cpp:16-224): iso-surface extraction of the
// voxel hash into a triangle mesh, and the mesh file.
typedef VhMarchingCubesParams MarchingCubesParams;
typedef VhMarchingCubesData MarchingCubesData;

namespace vh {
// what the reference keeps in an mLib MeshDataf: vertices, per-vertex colours (r,g,b,1), index triples
struct MeshData {
    std::vector<vec3f> m_Vertices;
    std::vector<float> m_Colors;                      // 4 per vertex
    std::vector<unsigned int> m_FaceIndicesVertices;  // 3 per face; empty = triangle soup (face i = 3i, 3i+1, 3i+2)
    void clear() { m_Vertices.clear(); m_Colors.clear(); m_FaceIndicesVertices.clear(); }
    bool hasVertexIndices() const { return !m_FaceIndicesVertices.empty(); }
    void makeTriangleSoupIndices();
    // MeshData::mergeCloseVertices(thresh, approx = true) / removeDuplicateFaces() / merge() / applyTransform() of the
    // mLib revision the reference vendors (DepthSensingCUDA/Include/mLib/include/core-mesh/meshData.{h,cpp})
    void mergeCloseVertices(float thresh);
    void removeDuplicateFaces();
    void merge(const MeshData& other);
    void applyTransform(const mat4f& t);
    void saveToPLY(const std::string& filename) const; // binary little endian, x y z red green blue alpha + faces
};
} // namespace vh

class CUDAMarchingCubesHashSDF {
public:
    explicit CUDAMarchingCubesHashSDF(const MarchingCubesParams& params, vhStream_t stream = nullptr);
    ~CUDAMarchingCubesHashSDF();
    CUDAMarchingCubesHashSDF(const CUDAMarchingCubesHashSDF&) = delete;
    CUDAMarchingCubesHashSDF& operator=(const CUDAMarchingCubesHashSDF&) = delete;

    // parametersFromGlobalAppState, .h:19-28: the four GlobalAppState values it reads, as arguments
    static MarchingCubesParams parameters(unsigned int marchingCubesMaxNumTriangles, float SDFMarchingCubeThreshFactor,
                                          float SDFVoxelSize, unsigned int hashNumBuckets);

    void clearMeshBuffer() { m_meshData.clear(); }
    // copies the result of the last extraction to the host and appends it to the mesh (.cpp:31-86); throws when the
    // triangle buffer overflowed.  offlineProcessing (GlobalAppState::s_offlineProcessing): merge each batch first.
    void copyTrianglesToCPU();
    void setOfflineProcessing(bool on) { m_offline = on; }
    void saveMesh(const std::string& filename, const vh::mat4f* transform = nullptr, bool overwriteExistingFile = false);

    // per chunk of the grid: stream the chunk and its neighbours in, extract inside the chunk's box (.cpp:149-192)
    void extractIsoSurface(CUDASceneRepChunkGrid& chunkGrid, const vh::vec3f& camPos, float radius);
    void extractIsoSurface(const HashData& hashData, const HashParams& hashParams, const vh::vec3f& minCorner = { 0, 0, 0 },
                           const vh::vec3f& maxCorner = { 0, 0, 0 }, bool boxEnabled = false);
    void extractIsoSurfaceWithoutCopy(const HashData& hashData, const HashParams& hashParams, const vh::vec3f& minCorner = { 0, 0, 0 },
                                      const vh::vec3f& maxCorner = { 0, 0, 0 }, bool boxEnabled = false);

    const vh::MeshData& getMeshData() const { return m_meshData; }
    const MarchingCubesData& getMarchingCubesData() const { return m_data; }
    unsigned int getNumTriangles();       // of the last extraction (blocks); may exceed m_maxNumTriangles on overflow
    unsigned int getNumOccupiedBlocks();  // of the last extraction
    void downloadTriangles(VhTriangle* out, unsigned int n); // the device triangle buffer of the last extraction

private:
    MarchingCubesParams m_params;
    MarchingCubesData m_data;
    vh::MeshData m_meshData;
    vhStream_t m_stream;
    bool m_offline;
};


// ---------------------------------------------------------------------------
// Recorded sequences: ml::SensorData, the `.sens` container (DSC/sensorData/sensorData.h), and
// SensorDataReader (DSC/SensorDataReader.{h,cpp}).  Host side; see voxelhashing_amd/csrc/vh_sensor_data.cpp.
namespace vh {

class SensorData {
public:
    static const uint32_t kVersion = 4; // M_SENSOR_DATA_VERSION, sensorData.h:607
    enum COMPRESSION_TYPE_COLOR { TYPE_RAW = 0, TYPE_PNG = 1, TYPE_JPEG = 2 };                     // :217-221
    enum COMPRESSION_TYPE_DEPTH { TYPE_RAW_USHORT = 0, TYPE_ZLIB_USHORT = 1, TYPE_OCCI_USHORT = 2 }; // :222-226
    struct RGBDFrame { // :229-551
        std::vector<uint8_t> m_colorCompressed, m_depthCompressed;
        uint64_t m_timeStampColor = 0, m_timeStampDepth = 0; // microseconds by convention
        mat4f m_cameraToWorld;
    };
    struct IMUFrame { // :553-605: 15 doubles and a time stamp, 128 bytes in the file
        double rotationRate[3], acceleration[3], magneticField[3], attitude[3], gravity[3];
        uint64_t timeStamp;
    };

    SensorData();
    static mat4f makeIntrinsicMatrix(float fx, float fy, float mx, float my);
    void loadFromFile(const std::string& filename); // throws vh::Error: VH_ERR_IO, VH_ERR_VERSION_MISMATCH
    void saveToFile(const std::string& filename) const;
    // compresses with the file's compression types: depth raw / zlib; colour raw only (NULL = frame without colour)
    void addFrame(const uint8_t* colorRGB, const uint16_t* depth, const mat4f& cameraToWorld = mat4f::identity(),
                  uint64_t timeStampColor = 0, uint64_t timeStampDepth = 0);
    void decompressDepth(size_t frameIdx, uint16_t* out) const;   // depthWidth*depthHeight samples
    void decompressColor(size_t frameIdx, uint8_t* outRGB) const; // colorWidth*colorHeight*3 bytes; raw, PNG, baseline JPEG

    uint32_t m_versionNumber;
    std::string m_sensorName;
    mat4f m_colorIntrinsic, m_colorExtrinsic, m_depthIntrinsic, m_depthExtrinsic; // CalibrationData :150-215
    int32_t m_colorCompressionType, m_depthCompressionType;
    uint32_t m_colorWidth, m_colorHeight, m_depthWidth, m_depthHeight;
    float m_depthShift; // depth in metres = sample / m_depthShift
    std::vector<RGBDFrame> m_frames;
    std::vector<IMUFrame> m_IMUFrames;
};

class SensorDataReader { // the RGBDSensor the frame loop polls when s_sensorIdx selects a recorded sequence
public:
    SensorDataReader();
    void createFirstConnected(const std::string& filename);
    bool processDepth(); // decodes the next frame; false once the sequence is complete
    const float* getDepthFloat() const { return m_depthFloat.data(); }    // metres, 0 = no measurement
    const uint8_t* getColorRGBX() const { return m_colorRGBX.data(); }    // {r, g, b, 1}
    mat4f getRigidTransform(int offset = 0) const;                        // recorded pose of the frame last decoded
    unsigned int getNumFrames() const { return m_numFrames; }
    unsigned int getCurrFrame() const { return m_currFrame; }
    bool hasColorData() const { return m_bHasColorData; }
    const SensorData& getSensorData() const;

private:
    std::unique_ptr<SensorData> m_sensorData;
    std::vector<float> m_depthFloat;
    std::vector<uint8_t> m_colorRGBX, m_colorRGB;
    std::vector<uint16_t> m_depthShorts;
    unsigned int m_numFrames, m_currFrame;
    bool m_bHasColorData;
};

} // namespace vh

// ---------------------------------------------------------------------------
// CUDARGBDSensor over CUDARGBDAdapter (DSC/CUDARGBDSensor.{h,cpp}, DSC/CUDARGBDAdapter.{h,cpp}): the image path from
// a sensor frame (float depth in metres + RGBX bytes, host memory, as RGBDSensor::getDepthFloat / getColorRGBX
// deliver them) to the DepthCameraData that integrate() consumes, plus the camera-space and normal maps tracking uses.
class CUDARGBDSensor {
public:
    struct Config {
        unsigned int depthWidth, depthHeight, colorWidth, colorHeight; // sensor images
        unsigned int adapterWidth, adapterHeight;                      // s_adapterWidth / s_adapterHeight: working resolution
        float fx, fy, mx, my;                                          // depth intrinsics at sensor resolution
        float sensorDepthMin, sensorDepthMax;                          // s_sensorDepthMin / s_sensorDepthMax
        bool filterDepth; float sigmaD, sigmaR;                        // s_depthFilter, s_depthSigmaD, s_depthSigmaR
        bool filterIntensity; float sigmaDIntensity, sigmaRIntensity;  // s_colorFilter, s_colorSigmaD, s_colorSigmaR
    };
    explicit CUDARGBDSensor(const Config& config, vhStream_t stream = nullptr);
    ~CUDARGBDSensor();
    CUDARGBDSensor(const CUDARGBDSensor&) = delete;
    CUDARGBDSensor& operator=(const CUDARGBDSensor&) = delete;

    void process(const float* h_depthFloat, const unsigned char* h_colorRGBX); // CUDARGBDSensor::process :147 (blocks until done)
    void setFiterDepthValues(bool b = true, float sigmaD = 1.0f, float sigmaR = 1.0f);     // (sic) .h:37
    void setFiterIntensityValues(bool b = true, float sigmaD = 1.0f, float sigmaR = 1.0f); // (sic) .h:40

    const DepthCameraData& getDepthCameraData() const { return m_depthCameraData; }       // .h:78
    const DepthCameraParams& getDepthCameraParams() const { return m_depthCameraParams; } // .h:82
    float* getCameraSpacePositionsFloat4() { return d_cameraSpaceFloat4; }                // .h:50
    float* getNormalMapFloat4() { return d_normalMapFloat4; }                             // .h:53
    float* getDepthMapColorSpaceFloat() { return d_depthData; }                           // .h:44
    float* getColorMapFilteredFloat4() { return d_colorData; }                            // .h:47
    float* getIntensityMapFilteredFloat() { return d_intensityMapFilteredFloat; }
    unsigned int getDepthWidth() const { return m_cfg.adapterWidth; }
    unsigned int getDepthHeight() const { return m_cfg.adapterHeight; }
    unsigned int getFrameNumber() const { return m_frameNumber; }

private:
    Config m_cfg;
    vhStream_t m_stream;
    unsigned int m_frameNumber;
    bool m_bFilterDepthValues, m_bFilterIntensityValues;
    float m_fBilateralFilterSigmaD, m_fBilateralFilterSigmaR, m_fBilateralFilterSigmaDIntensity, m_fBilateralFilterSigmaRIntensity;
    DepthCameraParams m_depthCameraParams;
    DepthCameraData m_depthCameraData;
    float *d_depthMapFloat, *d_depthMapResampledFloat, *d_depthMapFilteredFloat, *d_intensityMapFilteredFloat;
    unsigned char* d_colorMapRaw;
    float *d_colorMapFloat4, *d_colorMapResampledFloat4, *d_cameraSpaceFloat4, *d_normalMapFloat4;
    float *d_depthData, *d_colorData; // what m_depthCameraData points to
};


// ---------------------------------------------------------------------------
// CUDACameraTrackingMultiRes (DSC/CUDACameraTrackingMultiRes.{h,cpp}): coarse-to-fine projective point-to-plane ICP of
// the sensor frame (camera-space positions + normals) against the ray-cast model maps.  The per-level settings the
// reference takes from GlobalCameraTrackingState (and passes in five vectors) travel as one VhTrackingState.
class CUDACameraTrackingMultiRes {
public:
    CUDACameraTrackingMultiRes(unsigned int imageWidth, unsigned int imageHeight, unsigned int levels, vhStream_t stream = nullptr);
    ~CUDACameraTrackingMultiRes();
    CUDACameraTrackingMultiRes(const CUDACameraTrackingMultiRes&) = delete;
    CUDACameraTrackingMultiRes& operator=(const CUDACameraTrackingMultiRes&) = delete;

    // applyCT :241-289.  Returns lastTransform * delta; a matrix of -inf when tracking was lost (isTrackingLost).
    vh::mat4f applyCT(float* dInput, float* dInputNormals, float* dModel, float* dModelNormals, const vh::mat4f& lastTransform,
                      const VhTrackingState& settings, const vh::mat4f& deltaTransformEstimate, const DepthCameraParams& depthCameraParams);
    static bool isTrackingLost(const vh::mat4f& m);
    const VhIcpState& getLastState() const { return m_lastState; } // LinearSystemConfidence of the last solve + iteration count
    unsigned int getLevels() const { return m_levels; }

private:
    unsigned int m_levels;
    vhStream_t m_stream;
    std::vector<unsigned int> m_imageWidth, m_imageHeight;
    std::vector<float*> d_correspondence, d_correspondenceNormal, d_input, d_inputNormal, d_model, d_modelNormal;
    float* d_partials;
    VhIcpState* d_state;
    float* d_deltaEstimate;
    VhIcpState m_lastState;
};

#endif // VH_HPP
