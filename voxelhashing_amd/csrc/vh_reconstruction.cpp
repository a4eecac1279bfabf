// vh_reconstruction.cpp -- the frame loop of the reference application, reconstruction()
// (DSC/DepthSensing.cpp:720-924), for a recorded sequence at given poses, behind the C ABI: one host call enqueues any
// number of frames (the build's headless driver).  DSC/ = /root/reference/DepthSensingCUDA/Source/.
//
// Per frame, in the reference's order:
//   render(pose of the previous frame)                         :750-763
//   [stream out around the camera; stream in]                  :881-900
//   integrate(pose, depth, colour, bit mask)                   :903
// What is not in the reference:
//   * s_allocAhead: the pose of frame k is known before pose k-1 is ray-cast (it comes from the file), so the frame's
//     alloc pass rides in the ray caster's launch (its last workgroups) and its compactify pass, with the next pose's
//     interval splat, in computeNormals' (CUDASceneRepHashSDF::integrateAhead hands out the job) -- and, up to 2048 blocks in view, the pass over the
//     voxels there too: two launches, else three, per
//     frame on ONE stream, no event.  With streaming on this happens in the frames whose streaming step is known a
//     frame ahead to be a no-op (vh_stream_out_probe; frame() below);
//   * s_framesOnHost: float depth + RGBX colour in host memory (what RGBDSensor::getDepthFloat / getColorRGBX hand
//     to CUDARGBDAdapter::process, DSC/CUDARGBDAdapter.cpp:107-131) are uploaded by two copy streams (one copy engine
//     each: depth, colour) into a ring of kStagingSlots = 4 staging slots, beside the previous frames' work, and the
//     colour is converted there (convertColorRawToFloat4);
//   * s_maxFramesInFlight: the host stays at most that many frames ahead of the device (polled through the mapped
//     frame counter the fused integrate pass writes: no event).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>

#include "vh_handles.hpp"
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
inline double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline hipEvent_t newEvent(bool timing)
{
    hipEvent_t e = nullptr;
    // device-scope release: these events order streams of one device (or time them); nothing on the host reads memory
    // behind them (a default event makes the queue write back its caches: ~6 us of idle queue per record)
    if (hipEventCreateWithFlags(&e, (timing ? 0u : hipEventDisableTiming) | hipEventReleaseToDevice) != hipSuccess) {
        (void)hipGetLastError();
        checkHip(timing ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    }
    return e;
}

} // namespace

ReconstructionOptions Reconstruction::defaultOptions()
{
    ReconstructionOptions o;
    std::memset(&o, 0, sizeof(o));
    o.s_streamingEnabled = 0;
    o.s_integrationEnabled = 1;
    o.s_offlineProcessing = 0;
    o.s_renderEnabled = 1;
    o.s_allocAhead = 1;
    o.s_framesOnHost = 0;
    o.s_maxFramesInFlight = 16;
    o.s_streamingPos[0] = o.s_streamingPos[1] = 0.0f;
    o.s_streamingPos[2] = 3.0f; // zParametersDefault.txt: s_streamingPos
    o.s_streamingRadius = 4.0f;
    return o;
}

Reconstruction::Reconstruction(CUDASceneRepHashSDF* sceneRep, CUDARayCastSDF* rayCast, CUDASceneRepChunkGrid* chunkGrid,
                               const DepthCameraParams& cp, const ReconstructionOptions& options)
    : m_sceneRep(sceneRep), m_rayCast(rayCast), m_chunkGrid(chunkGrid), m_cp(cp), m_opt(options), m_frameNumber(0), m_copyStream(nullptr), m_copyStream2(nullptr)
{
    m_debugFailRender = 0;
    m_pipelineOutSeen = m_pipelineInSeen = 0;
    if (chunkGrid) chunkGrid->pipelineTotals(&m_pipelineOutSeen, &m_pipelineInSeen);
    if (!sceneRep) throw vh::Error(VH_ERR_BAD_ARGUMENT, "Reconstruction: no scene");
    if (options.s_streamingEnabled && !chunkGrid) throw vh::Error(VH_ERR_BAD_ARGUMENT, "Reconstruction: streaming needs a chunk grid");
    if (options.s_renderEnabled && !rayCast) throw vh::Error(VH_ERR_BAD_ARGUMENT, "Reconstruction: rendering needs a ray caster");
    std::memset(&m_stats, 0, sizeof(m_stats));
    for (int i = 0; i < kStagingSlots; i++) {
        d_stageDepth[i] = nullptr; d_stageColorRaw[i] = nullptr; d_stageColor[i] = nullptr;
        m_slotReady[i] = m_slotReady2[i] = nullptr;
        m_slotSceneFrame[i] = 0;
    }
    m_uploads = 0;
    m_probePending = false;
    std::memset(m_probePose, 0, sizeof(m_probePose));
    if (m_opt.s_framesOnHost) {
        const size_t n = (size_t)cp.m_imageWidth * cp.m_imageHeight;
        hipStream_t cs = nullptr;
        checkHip(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking), "hipStreamCreate");
        m_copyStream = (void*)cs;
        hipStream_t cs2 = nullptr;
        checkHip(hipStreamCreateWithFlags(&cs2, hipStreamNonBlocking), "hipStreamCreate");
        m_copyStream2 = (void*)cs2;
        for (int i = 0; i < kStagingSlots; i++) {
            checkHip(hipMalloc((void**)&d_stageDepth[i], sizeof(float) * (n ? n : 1)), "staging depth");
            checkHip(hipMalloc((void**)&d_stageColorRaw[i], 4 * (n ? n : 1)), "staging colour (raw)");
            checkHip(hipMalloc((void**)&d_stageColor[i], sizeof(float) * 4 * (n ? n : 1)), "staging colour");
            m_slotReady[i] = (void*)newEvent(false);
            m_slotReady2[i] = (void*)newEvent(false);
        }
        m_stats.uploadBytes = (sizeof(float) + 4) * n;
    }
}

Reconstruction::~Reconstruction()
{
    try { synchronize(); } catch (...) {}
    for (auto& p : m_uploadTimers) { (void)hipEventDestroy((hipEvent_t)p.first); (void)hipEventDestroy((hipEvent_t)p.second); }
    for (void* e : m_timerPool) (void)hipEventDestroy((hipEvent_t)e);
    for (int i = 0; i < kStagingSlots; i++) {
        if (m_slotReady[i]) (void)hipEventDestroy((hipEvent_t)m_slotReady[i]);
        if (m_slotReady2[i]) (void)hipEventDestroy((hipEvent_t)m_slotReady2[i]);
        if (d_stageDepth[i]) (void)hipFree(d_stageDepth[i]);
        if (d_stageColorRaw[i]) (void)hipFree(d_stageColorRaw[i]);
        if (d_stageColor[i]) (void)hipFree(d_stageColor[i]);
    }
    if (m_copyStream) (void)hipStreamDestroy((hipStream_t)m_copyStream);
    if (m_copyStream2) (void)hipStreamDestroy((hipStream_t)m_copyStream2);
}

void Reconstruction::synchronize()
{
    if (m_copyStream) checkHip(hipStreamSynchronize((hipStream_t)m_copyStream), "hipStreamSynchronize");
    if (m_copyStream2) checkHip(hipStreamSynchronize((hipStream_t)m_copyStream2), "hipStreamSynchronize");
    // the scene's side stream joins the main stream in integrateFinish(): the main stream is the last to finish
    checkHip(hipStreamSynchronize((hipStream_t)m_sceneRep->getStream()), "hipStreamSynchronize");
    if (m_chunkGrid) m_chunkGrid->pipelineDrain(false); // (the grid's worker has taken in what the last frame moved out; its choice for the next frame stands)
}

void Reconstruction::reset()
{
    synchronize();
    (void)getStats(); // returns the pending timer events to the pool
    const uint64_t bytes = m_stats.uploadBytes;
    std::memset(&m_stats, 0, sizeof(m_stats));
    m_stats.uploadBytes = bytes;
    m_frameNumber = 0;
    m_probePending = false;
    for (int i = 0; i < kStagingSlots; i++) m_slotSceneFrame[i] = 0;
}

const ReconstructionStats& Reconstruction::getStats()
{
    if (!m_uploadTimers.empty()) {
        checkHip(hipStreamSynchronize((hipStream_t)m_copyStream), "hipStreamSynchronize");
        for (auto& p : m_uploadTimers) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, (hipEvent_t)p.first, (hipEvent_t)p.second) == hipSuccess) { m_stats.uploadMs += ms; m_stats.uploadsTimed++; }
            m_timerPool.push_back(p.first);
            m_timerPool.push_back(p.second);
        }
        m_uploadTimers.clear();
    }
    // the scene's status words: an empty voxel pool or a failed stream-in insert is raised on the device; this is where a
    // caller of the loop gets to see it (a blocking 64-byte read-back: get_stats() is not for the inside of a timed region)
    if (m_chunkGrid) { // what the streaming pipeline moved (its worker counts the blocks that left when they arrive)
        unsigned long long out = 0, in = 0;
        m_chunkGrid->pipelineTotals(&out, &in);
        m_stats.blocksStreamedOut += out - m_pipelineOutSeen;
        m_stats.blocksStreamedIn += in - m_pipelineInSeen;
        m_pipelineOutSeen = out;
        m_pipelineInSeen = in;
    }
    uint32_t state[VH_STATE_WORDS];
    m_sceneRep->getState(state);
    m_stats.heapUnderflows = state[VH_STATE_HEAP_UNDERFLOW];
    m_stats.failedInserts = state[VH_STATE_INSERT_FAILED];
    return m_stats;
}

// CUDARGBDAdapter::process :107-131 for a frame at adapter resolution: upload, colour bytes -> float4.  The copy
// stream runs beside the frame loop's stream; the only thing the main stream does for an upload is to wait for its
// "ready" event (a record on the main stream would idle it for ~6 us per frame).  A staging slot is reused once the
// frame that read it last has been integrated, which the host sees in the scene's mapped frame counter: the pass over
// the voxels of the NEXT frame has started.
DepthCameraData Reconstruction::upload(const SequenceFrame& f)
{
    const unsigned int slot = m_uploads % kStagingSlots;
    const size_t n = (size_t)m_cp.m_imageWidth * m_cp.m_imageHeight;
    hipStream_t cs = (hipStream_t)m_copyStream, ms = (hipStream_t)m_sceneRep->getStream();
    if (m_slotSceneFrame[slot] != 0) {
        // the slot's last frame was the scene's frame number m_slotSceneFrame[slot]: done once a later frame's pass has started
        const unsigned int need = m_slotSceneFrame[slot] + 1u;
        const double w0 = now();
        bool waited = false;
        const VhSceneOptions& so = m_sceneRep->getOptions();
        const bool mirrored = m_opt.s_integrationEnabled && !so.s_useReferenceLaunchSequence; // only the fused pass keeps the counter
        while (m_sceneRep->getNumFramesStartedOnDevice() < need) {
            if (!mirrored || m_sceneRep->getNumIntegratedFrames() < need) { // nothing later has been enqueued: only a synchronisation tells
                checkHip(hipStreamSynchronize(ms), "hipStreamSynchronize");
                break;
            }
            std::this_thread::yield();
            waited = true;
            if (now() - w0 > 30.0) throw vh::Error(VH_ERR_TIMEOUT, "Reconstruction: the device made no progress for 30 s");
        }
        if (waited) m_stats.hostWaitSeconds += now() - w0;
    }
    auto timerEvent = [&]() {
        if (!m_timerPool.empty()) { void* e = m_timerPool.back(); m_timerPool.pop_back(); return e; }
        return (void*)newEvent(true);
    };
    const bool timed = (m_uploads % 8u) == 0u; // (a timed pair idles the copy stream twice)
    void *t0 = nullptr, *t1 = nullptr;
    // The frame travels by the copy engines (hipMemcpyAsync: depth on one stream, colour on another, so that each gets
    // an engine), then the colour is converted as in the sensor path.  Reading the pinned frame from a kernel instead
    // (vh_upload_frame: one pass, no raw-colour staging) was measured slower for the loop as a whole: while uncached
    // reads of host memory are in flight every other kernel's memory accesses queue behind them, and k_render takes
    // two to three times as long (vh_kernels.hip, k_upload_frame; VH_UPLOAD_KERNEL=1 selects that path for measurement).
    static const bool useKernel = std::getenv("VH_UPLOAD_KERNEL") != nullptr;
    void *devDepth = nullptr, *devColor = nullptr;
    const bool mapped = useKernel && (n % 4u) == 0u && hipHostGetDevicePointer(&devDepth, const_cast<float*>(f.depth), 0) == hipSuccess &&
                        (!f.color || hipHostGetDevicePointer(&devColor, const_cast<void*>(f.color), 0) == hipSuccess);
    if (useKernel && !mapped) (void)hipGetLastError();
    if (timed) {
        t0 = timerEvent();
        t1 = timerEvent();
        checkHip(hipEventRecord((hipEvent_t)t0, cs), "hipEventRecord");
    }
    if (mapped) {
        check(vh_upload_frame((const float*)devDepth, (const uint8_t*)devColor, d_stageDepth[slot], d_stageColor[slot], m_cp.m_imageWidth, m_cp.m_imageHeight, m_copyStream), "vh_upload_frame");
    } else {
        // two copies, two streams: each gets a copy engine of its own
        hipStream_t cs2 = (hipStream_t)m_copyStream2;
        checkHip(hipMemcpyAsync(d_stageDepth[slot], f.depth, sizeof(float) * n, hipMemcpyHostToDevice, cs2), "upload depth");
        checkHip(hipEventRecord((hipEvent_t)m_slotReady2[slot], cs2), "hipEventRecord");
        checkHip(hipStreamWaitEvent(ms, (hipEvent_t)m_slotReady2[slot], 0), "hipStreamWaitEvent");
        if (f.color) {
            checkHip(hipMemcpyAsync(d_stageColorRaw[slot], f.color, 4 * n, hipMemcpyHostToDevice, cs), "upload colour");
            check(vh_convert_color_raw_to_float4(d_stageColor[slot], d_stageColorRaw[slot], m_cp.m_imageWidth, m_cp.m_imageHeight, m_copyStream), "convertColorRawToFloat4");
        }
    }
    if (timed) {
        // the pair spans the whole upload: the depth copy runs on the other stream, so this one waits for it first
        // (t0 was recorded before either copy was enqueued; both streams were idle or busy with earlier uploads)
        if (!mapped) checkHip(hipStreamWaitEvent(cs, (hipEvent_t)m_slotReady2[slot], 0), "hipStreamWaitEvent");
        checkHip(hipEventRecord((hipEvent_t)t1, cs), "hipEventRecord");
        m_uploadTimers.emplace_back(t0, t1);
    }
    checkHip(hipEventRecord((hipEvent_t)m_slotReady[slot], cs), "hipEventRecord");
    checkHip(hipStreamWaitEvent(ms, (hipEvent_t)m_slotReady[slot], 0), "hipStreamWaitEvent");
    m_slotSceneFrame[slot] = m_sceneRep->getNumIntegratedFrames() + 1u; // the scene frame this upload feeds
    m_uploads++;
    DepthCameraData cam;
    std::memset(&cam, 0, sizeof(cam));
    cam.d_depthData = d_stageDepth[slot];
    cam.d_colorData = f.color ? d_stageColor[slot] : nullptr;
    return cam;
}

namespace {
bool poseValid(const float* m) { return !(m[0] == -std::numeric_limits<float>::infinity() || std::isnan(m[0])); }
}

void Reconstruction::frame(const SequenceFrame& f, const SequenceFrame* next)
{
    // :733-747
    vh::mat4f transformation;
    std::memcpy(transformation.m, f.rigidTransform, sizeof(transformation.m));
    if (!poseValid(transformation.m)) {
        m_stats.invalidFrames++;
        return; // "INVALID FRAME"
    }
    if (!f.depth) throw vh::Error(VH_ERR_BAD_ARGUMENT, "Reconstruction: frame without a depth map");

    DepthCameraData cam;
    if (m_opt.s_framesOnHost) cam = upload(f);
    else {
        std::memset(&cam, 0, sizeof(cam));
        cam.d_depthData = const_cast<float*>(f.depth);
        cam.d_colorData = const_cast<float*>(static_cast<const float*>(f.color));
    }

    const bool streaming = m_opt.s_streamingEnabled && m_chunkGrid;
    const bool threaded = streaming && !m_opt.s_offlineProcessing && !m_chunkGrid->getTerminatedThread();
    const vh::vec3f p = transformation.transformPoint({ m_opt.s_streamingPos[0], m_opt.s_streamingPos[1], m_opt.s_streamingPos[2] });

    // The streaming step of this frame (:881-900).  The reference runs it between the ray cast of the previous pose and this
    // frame's alloc as a chain of host <-> device round trips (two counters read back, the blocks that left put into the host
    // grid before the bit mask for alloc is known, the chunk that comes in chosen after that), so alloc could not ride in the
    // ray caster's launch and the host could not enqueue past it.  With the next pose known a frame ahead (the sequence
    // comes from a file) the step runs without a host wait (CUDASceneRepChunkGrid's pipeline, vh.hpp):
    //   * how many blocks leave at most was asked of the device a frame early (a count-only run of the stream-out scan for
    //     THIS frame's sphere and part, behind the previous frame's alloc: nothing adds blocks between there and here);
    //   * the chunk that comes in was chosen and uploaded by the grid's worker while the device worked on the previous frame;
    //   * the counts stay on the device, the device keeps its own copy of the bit mask.
    // A frame in which nothing leaves and nothing comes in is then two or three launches, like a frame without streaming (alloc
    // rides in the ray caster's launch, reading the device's bit mask); a frame with traffic is the reference's order of
    // launches, enqueued without waiting.  Without the answers (first frame, next pose unknown, a pass too large for the
    // pipeline's staging) the frame takes the reference's order of calls.
    enum Step { kFull, kPipelined } step = kFull;
    unsigned int mostOut = 0;
    CUDASceneRepChunkGrid::StreamDecision choice = { 0u, 0xffffffffu, 0 };
    struct Unwind { // (integrateAhead() ... integrateFinish() with the ray cast in between: a throw must not leave the scene refusing every integrate())
        CUDASceneRepHashSDF* scene;
        bool ahead = false;
        ~Unwind() { if (ahead) scene->abortAhead(); }
    } unwind{ m_sceneRep };
    const bool pipelined = threaded && m_opt.s_allocAhead && m_opt.s_integrationEnabled;
    if (pipelined && m_probePending && std::memcmp(m_probePose, f.rigidTransform, sizeof(m_probePose)) == 0 &&
        m_chunkGrid->pipelineHasDecision(p, m_opt.s_streamingRadius)) {
        const double t0 = now();
        mostOut = m_chunkGrid->probeResult();
        if (mostOut <= m_chunkGrid->pipelineCapacity()) {
            choice = m_chunkGrid->pipelineDecision();
            step = kPipelined;
        }
        m_stats.hostWaitSeconds += now() - t0;
    }
    m_probePending = false;
    const bool quiet = step == kPipelined && mostOut == 0u && choice.nIn == 0u;

    const bool ahead = m_opt.s_allocAhead && m_opt.s_integrationEnabled && (!streaming || quiet);
    const unsigned int* d_bitMask = nullptr;
    if (streaming && step == kPipelined) d_bitMask = m_chunkGrid->getBitMaskDevice(); // (kept by the passes themselves: no upload)
    // :750-751 (the pose the scene holds is the previous frame's)
    const vh::mat4f renderTransform = m_sceneRep->getLastRigidTransform();
    VhFrameJob* job = nullptr;
    if (ahead) {
        job = m_sceneRep->integrateAhead(transformation, cam, m_cp, d_bitMask);
        unwind.ahead = true;
    }
    if (m_frameNumber > 0 && m_opt.s_renderEnabled) { // :750 "getFrameNumber() > 1" with frames counted from 1
        if (m_debugFailRender && --m_debugFailRender == 0) {
            // (the choice for this frame has been taken off the worker: hand it back before leaving)
            if (step == kPipelined) m_chunkGrid->pipelineReturn(choice, p, m_opt.s_streamingRadius);
            throw vh::Error(VH_ERR_BAD_ARGUMENT, "Reconstruction: injected failure of the ray cast (vh_reconstruction_debug_fail_render)");
        }
        const unsigned int used0 = m_rayCast->getNumSplatsMadeAheadUsed();
        try {
            m_rayCast->render(m_sceneRep->getHashData(), m_sceneRep->getHashParams(), m_cp, renderTransform, job); // :763
        } catch (...) {
            if (step == kPipelined) m_chunkGrid->pipelineReturn(choice, p, m_opt.s_streamingRadius);
            throw;
        }
        m_stats.splatsMadeAheadUsed += m_rayCast->getNumSplatsMadeAheadUsed() - used0;
        if (job && job->allocLaunched && job->compactifyLaunched) m_stats.framesWithRiders++;
        if (job && job->fusedLaunched) m_stats.framesInTwoLaunches++;
    }

    if (streaming && step == kPipelined) {
        try {
            (void)m_chunkGrid->pipelineStreamOut(p, m_opt.s_streamingRadius, CUDASceneRepChunkGrid::s_useParts, mostOut);
            m_chunkGrid->pipelineStreamIn(choice);
        } catch (...) { // (the chunk that was to come in is still in the worker's staging buffer: it goes back into the grid with the next drain)
            m_chunkGrid->pipelineReturn(choice, p, m_opt.s_streamingRadius);
            throw;
        }
        m_stats.streamingFramesPipelined++;
        if (quiet) m_stats.streamingStepsSkipped++;
    } else if (streaming) { // :881-900
        const double t0 = now();
        unsigned int nStreamedBlocks = 0;
        if (m_opt.s_offlineProcessing) {
            for (unsigned int i = 0; i < m_sceneRep->getOptions().s_streamingOutParts; i++) {
                m_chunkGrid->streamOutToCPU(p, m_opt.s_streamingRadius, CUDASceneRepChunkGrid::s_useParts, nStreamedBlocks);
                m_stats.blocksStreamedOut += nStreamedBlocks;
            }
            m_chunkGrid->streamInToGPUAll(p, m_opt.s_streamingRadius, CUDASceneRepChunkGrid::s_useParts, nStreamedBlocks);
            m_stats.blocksStreamedIn += nStreamedBlocks;
        } else if (threaded) {
            m_chunkGrid->streamOutToCPUPass0GPU(p, m_opt.s_streamingRadius, CUDASceneRepChunkGrid::s_useParts, true);
            m_stats.blocksStreamedOut += m_chunkGrid->getNumStreamedOutBlocks();
            m_chunkGrid->streamInToGPUPass1GPU(true);
            m_stats.blocksStreamedIn += m_chunkGrid->getNumStreamedInBlocks();
        } else {
            m_chunkGrid->streamOutToCPU(p, m_opt.s_streamingRadius, CUDASceneRepChunkGrid::s_useParts, nStreamedBlocks);
            m_stats.blocksStreamedOut += nStreamedBlocks;
            m_chunkGrid->streamInToGPU(p, m_opt.s_streamingRadius, CUDASceneRepChunkGrid::s_useParts, nStreamedBlocks);
            m_stats.blocksStreamedIn += nStreamedBlocks;
        }
        d_bitMask = m_chunkGrid->getBitMaskGPU();
        m_stats.hostWaitSeconds += now() - t0; // read-backs of the streaming counters: the host waits for the device here
    }

    // the question for the next frame, behind this frame's alloc
    const bool ask = pipelined && next && next->depth && poseValid(next->rigidTransform);
    vh::vec3f np = { 0.0f, 0.0f, 0.0f };
    if (ask) {
        vh::mat4f nt;
        std::memcpy(nt.m, next->rigidTransform, sizeof(nt.m));
        np = nt.transformPoint({ m_opt.s_streamingPos[0], m_opt.s_streamingPos[1], m_opt.s_streamingPos[2] });
    }
    auto askNow = [&]() {
        m_chunkGrid->probeStreamOut(np, m_opt.s_streamingRadius, CUDASceneRepChunkGrid::s_useParts);
        std::memcpy(m_probePose, next->rigidTransform, sizeof(m_probePose));
        m_probePending = true;
    };
    const bool allocIsIn = ahead && job && job->allocLaunched; // (it rode in the ray caster's launch)
    if (ask && allocIsIn) askNow();

    if (m_opt.s_integrationEnabled) { // :903
        unwind.ahead = false; // (integrateFinish() closes the job first thing)
        if (ahead) m_sceneRep->integrateFinish(cam, m_cp);
        else m_sceneRep->integrate(transformation, cam, m_cp, d_bitMask);
    } else {
        m_sceneRep->setLastRigidTransformAndCompactify(transformation, m_cp); // :907
    }
    if (ask && !allocIsIn) askNow();
    // the worker's job: take in what this frame's stream-out pass moves, choose and upload what comes in at the next frame
    if (pipelined && (ask || step == kPipelined)) m_chunkGrid->pipelineAsk(ask, np, m_opt.s_streamingRadius);
    m_frameNumber++;
    m_stats.frames++;
}

void Reconstruction::run(const SequenceFrame* frames, unsigned int n, const SequenceFrame* after)
{
    if (n && !frames) throw vh::Error(VH_ERR_BAD_ARGUMENT, "Reconstruction::run: null frames");
    const double t0 = now();
    double waited = 0.0;
    const double streamWait0 = m_stats.hostWaitSeconds;
    // Run-ahead bound without an event (a record idles the queue for ~6 us on this machine): the pass over the voxels
    // mirrors the scene's frame counter into mapped host memory when it starts.  Only that (fused) pass does so.
    const VhSceneOptions& so = m_sceneRep->getOptions();
    const bool bounded = m_opt.s_maxFramesInFlight && m_opt.s_integrationEnabled && !so.s_useReferenceLaunchSequence;
    for (unsigned int i = 0; i < n; i++) {
        if (bounded && m_sceneRep->getNumIntegratedFrames() - m_sceneRep->getNumFramesStartedOnDevice() >= m_opt.s_maxFramesInFlight) {
            const double w0 = now();
            while (m_sceneRep->getNumIntegratedFrames() - m_sceneRep->getNumFramesStartedOnDevice() >= m_opt.s_maxFramesInFlight) {
                std::this_thread::yield();
                if (now() - w0 > 30.0) throw vh::Error(VH_ERR_TIMEOUT, "Reconstruction::run: the device made no progress for 30 s");
            }
            waited += now() - w0;
        }
        frame(frames[i], i + 1 < n ? &frames[i + 1] : after);
    }
    const double total = now() - t0, streamWait = m_stats.hostWaitSeconds - streamWait0;
    m_stats.hostWaitSeconds += waited;
    m_stats.hostEnqueueSeconds += total - waited - streamWait;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

namespace {
template <class F> int guarded(F&& f) { return vh_guarded(static_cast<F&&>(f)); }
} // namespace

extern "C" {

void vh_reconstruction_default_options(VhReconstructionOptions* out)
{
    if (out) *out = Reconstruction::defaultOptions();
}

int vh_reconstruction_create(VhSceneRep* scene, VhRayCast* rayCast, VhChunkGrid* chunkGrid, const VhDepthCameraParams* cp,
                             const VhReconstructionOptions* opt, VhReconstruction** out)
{
    if (!scene || !cp || !out) return VH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    return guarded([&] {
        const ReconstructionOptions o = opt ? *opt : Reconstruction::defaultOptions();
        *out = new VhReconstruction(&scene->impl, rayCast ? &rayCast->impl : nullptr, chunkGrid ? &chunkGrid->impl : nullptr, *cp, o);
    });
}
void vh_reconstruction_destroy(VhReconstruction* r) { delete r; }
int vh_reconstruction_run(VhReconstruction* r, const VhSequenceFrame* frames, uint32_t n)
{
    if (!r || (n && !frames)) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { r->impl.run(frames, n); });
}
int vh_reconstruction_run_ahead(VhReconstruction* r, const VhSequenceFrame* frames, uint32_t n, const VhSequenceFrame* next)
{
    if (!r || (n && !frames)) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { r->impl.run(frames, n, next); });
}
int vh_reconstruction_synchronize(VhReconstruction* r)
{
    if (!r) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { r->impl.synchronize(); });
}
int vh_reconstruction_debug_fail_render(VhReconstruction* r, uint32_t nthRenderFromNow)
{
    if (!r) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { r->impl.debugFailRender(nthRenderFromNow); });
}
int vh_reconstruction_get_stats(VhReconstruction* r, VhReconstructionStats* out)
{
    if (!r || !out) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { *out = r->impl.getStats(); });
}
int vh_reconstruction_reset(VhReconstruction* r)
{
    if (!r) return VH_ERR_BAD_ARGUMENT;
    return guarded([&] { r->impl.reset(); });
}

} // extern "C"
