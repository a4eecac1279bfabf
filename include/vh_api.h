/*
 * vh_api.h -- C ABI of the MI355X voxel-hashing TSDF fusion + raycast engine.
 *
 * Drop-in boundary: the reference's host classes call `extern "C"` launchers
 * defined in its .cu files (C++ references, by-value structs -- not C-ABI
 * clean).  This header is their pointer-based twin; every entry point cites the
 * reference interface it replaces.
 *
 *   DSC/ = /root/reference/DepthSensingCUDA/Source/
 *
 * Conventions
 *  - all pointers inside VhHashData / VhDepthCameraData / VhRayCastData are
 *    DEVICE pointers; parameter structs are host pointers, read at call time
 *    and passed to the kernels as arguments (no __constant__ singletons, so
 *    several scenes per process are possible);
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *    launcher is asynchronous unless it documents a read-back;
 *  - return value: 0 ok, <0 = -(hipError_t), >0 = VH_ERR_* (vh_types.h);
 *  - `lockToken`: value written into d_hashBucketMutex to take a bucket for
 *    the rest of the pass.  VH_LOCK_ENTRY reproduces the reference (caller
 *    resets the mutex array before the pass with vh_reset_bucket_mutex); any
 *    other value that differs from every token used since the last reset
 *    makes that reset unnecessary (the host classes use a running epoch).
 */
#ifndef VH_API_H
#define VH_API_H

#include "vh_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vhStream_t;

/* ---- library -------------------------------------------------------------- */
const char* vh_version(void);
const char* vh_error_string(int code);
/* text of the last error raised by a handle-level call on this thread */
const char* vh_last_error_message(void);

/* ---- device memory helpers (thin hipMalloc/hipMemcpy wrappers for FFI users) */
int vh_malloc(void** devPtr, size_t bytes);
int vh_free(void* devPtr);
/* pinned, device-visible host memory (hipHostMalloc): what vh_upload_frame and the frame loop's host-fed mode read
 * straight over the link */
int vh_malloc_host(void** hostPtr, size_t bytes);
int vh_free_host(void* hostPtr);
int vh_memcpy_h2d(void* dst, const void* src, size_t bytes, vhStream_t stream);
int vh_memcpy_d2h(void* dst, const void* src, size_t bytes, vhStream_t stream); /* synchronises the stream */
int vh_memset(void* dst, int value, size_t bytes, vhStream_t stream);
/* measurement: the next kernel the calling thread launches through vh_render_intervals[_co], vh_compute_normals[_co, _co2] or
 * vh_integrate_fused is launched with hipExtLaunchKernel's start / stop events (two hipEvent_t created with timing):
 * hipEventElapsedTime(start, stop) is then that kernel's own duration -- the dispatch's begin and end time stamps, what
 * rocprofv3's kernel trace reports -- with no event record in the stream.  The launch consumes the pair. */
int vh_time_next_launch(void* startEvent, void* stopEvent);
int vh_stream_create(vhStream_t* out);   /* a non-blocking HIP stream, for FFI users without a HIP binding */
int vh_stream_destroy(vhStream_t stream);
int vh_stream_synchronize(vhStream_t stream);
int vh_device_synchronize(void);

/* ---- HashData ownership: HashData::allocate / free, DSC/VoxelUtilHashSDF.h:113-181 */
int vh_hash_data_alloc(VhHashData* hd, const VhHashParams* hp);
int vh_hash_data_free(VhHashData* hd);

/* ---- scene-rep launchers: DSC/CUDASceneRepHashSDF.h:15-26 ------------------- */
/* resetCUDA(HashData&, const HashParams&)                       DSC/CUDASceneRepHashSDF.cu:63 */
int vh_reset(const VhHashData* hd, const VhHashParams* hp, vhStream_t stream);
/* resetHashBucketMutexCUDA(HashData&, const HashParams&)        DSC/CUDASceneRepHashSDF.cu:109 */
int vh_reset_bucket_mutex(const VhHashData* hd, const VhHashParams* hp, vhStream_t stream);
/* allocCUDA(HashData&, const HashParams&, const DepthCameraData&, const DepthCameraParams&,
 *           const unsigned int* d_bitMask)                      DSC/CUDASceneRepHashSDF.cu:245
 * d_bitMask may be NULL (streaming disabled).  Also clears d_hashCompactifiedCounter for the
 * compaction that follows. */
int vh_alloc(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraData* cam,
             const VhDepthCameraParams* cp, const uint32_t* d_bitMask, int32_t lockToken, vhStream_t stream);
/* unsigned compactifyHashAllInOneCUDA(HashData&, const HashParams&)  DSC/CUDASceneRepHashSDF.cu:361
 * The count lands in d_hashCompactifiedCounter.  numOccupied != NULL: blocking
 * read-back as the reference does; NULL: fully asynchronous.
 * flags: VH_COMPACT_COUNTER_IS_ZERO = the caller guarantees the counter is already 0 (vh_alloc leaves it
 * cleared), which saves the memset the reference issues (DSC/CUDASceneRepHashSDF.cu:367). */
enum { VH_COMPACT_COUNTER_IS_ZERO = 1 };
int vh_compactify(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp,
                  uint32_t* numOccupied, uint32_t flags, vhStream_t stream);
/* integrateDepthMapCUDA(...)                                     DSC/CUDASceneRepHashSDF.cu:495
 * integrates hp->m_numOccupiedBlocks compactified blocks. */
int vh_integrate(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraData* cam,
                 const VhDepthCameraParams* cp, vhStream_t stream);
/* starveVoxelsKernelCUDA(HashData&, const HashParams&)          DSC/CUDASceneRepHashSDF.cu:523 */
int vh_starve(const VhHashData* hd, const VhHashParams* hp, vhStream_t stream);
/* garbageCollectIdentifyCUDA(HashData&, const HashParams&)      DSC/CUDASceneRepHashSDF.cu:592 */
int vh_gc_identify(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp, vhStream_t stream);
/* garbageCollectFreeCUDA(HashData&, const HashParams&)          DSC/CUDASceneRepHashSDF.cu:631 */
int vh_gc_free(const VhHashData* hd, const VhHashParams* hp, int32_t lockToken, vhStream_t stream);
/* bindInputDepthColorTextures(const DepthCameraData&)           DSC/CUDASceneRepHashSDF.cu:15
 * images are read with plain loads: kept as a no-op for source compatibility. */
int vh_bind_input_depth_color_textures(const VhDepthCameraData* cam);

/* Fused integrate -> [starve] -> GC identify -> GC free in ONE pass over the
 * voxels (one read + one write per voxel instead of up to four kernels);
 * same results as the four launchers above in the reference's order
 * (CUDASceneRepHashSDF::integrateDepthMap + garbageCollect, DSC/CUDASceneRepHashSDF.h:317-339).
 * The block count is read on the device from d_hashCompactifiedCounter; if d_countMirror != NULL
 * (a device pointer, e.g. of mapped pinned host memory) the count is also stored there. */
enum { VH_FUSED_GC = 1, VH_FUSED_STARVE = 2 };
/* d_countMirror (may be NULL) receives two words: the block count and `mirrorTag`, a number of the caller's choice
 * (the host class passes its frame counter: a host that maps the words can follow the device without an event).
 * d_packedFrame (may be NULL): the frame as vh_alloc_job packed it, 8 bytes per pixel {depth, colour bytes + sample
 * weight}; the pass then gathers 8 bytes per voxel instead of 20 from the depth and colour maps (same results). */
int vh_integrate_fused(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraData* cam,
                       const VhDepthCameraParams* cp, uint32_t flags, int32_t lockToken, uint32_t* d_countMirror,
                       uint32_t mirrorTag, const void* d_packedFrame, vhStream_t stream);
/* the alloc / compactify pass of a prepared frame (VhFrameJob, vh_types.h); vh_alloc_job also packs the frame into
 * job->d_packedFrame when that is not NULL.  Each marks the job. */
int vh_alloc_job(VhFrameJob* job, vhStream_t stream);
int vh_compactify_job(VhFrameJob* job, vhStream_t stream);

/* ---- ray-cast launchers: DSC/CUDARayCastSDF.cpp:10-21 ----------------------- */
/* renderCS(const HashData&, const RayCastData&, const DepthCameraData&, const RayCastParams&)
 *                                                               DSC/CUDARayCastSDF.cu:59 */
/* d_normals of the VhRayCastData may be NULL for vh_render / vh_render_intervals: the map is then not written (the
 * host class does so when vh_compute_normals overwrites it right after). */
int vh_render(const VhHashData* hd, const VhHashParams* hp, const VhRayCastData* rd,
              const VhDepthCameraParams* cp, const VhRayCastParams* rp, vhStream_t stream);
/* Ray-interval splatting as a compute pass: resetRayIntervalSplatCUDA / rayIntervalSplatCUDA
 * (DSC/CUDARayCastSDF.cu:88,169) + the D3D11 min/max rasterisation (DSC/DX11RayIntervalSplatting.cpp:150-220) that
 * this fork leaves disabled.  Per 8x8-pixel tile (ceil(W/8)*ceil(H/8) of them, row-major):
 *   d_tileHeads   4 words: {min, max} camera depth (float bits) of the allocated blocks the tile's rays can read --
 *                 kept only when no lists are (d_tileBlocks NULL); with lists the ray caster forms the range from the
 *                 listed blocks --, their number, 0;
 *   d_tileBlocks  tileCapacity entries: those blocks (may be NULL: intervals only).  tileCapacity also picks the
 *                 ray caster's table size: up to VH_TILE_LIST_CAPACITY (64) small tables, above it large ones
 *                 (VH_TILE_LIST_CAPACITY_LARGE, 128).  A longer list is used as far as it fits; the blocks it could
 *                 not hold are looked up in the hash table.
 *   d_longestList (splat, with a schedule; may be NULL) receives the longest list the previous render met if it came
 *                 within 16 of the small capacity, else 0: what a host needs to choose the capacity.
 * Both are conservative (every allocated block, grown by the reach of a sample), so rendering with them gives
 * bit-identical maps.  vh_render_intervals consumes and re-arms the heads; vh_ray_interval_clear arms them once. */
int vh_ray_interval_clear(uint32_t* d_tileHeads, uint32_t width, uint32_t height, vhStream_t stream);
int vh_ray_interval_splat(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp, const VhRayCastParams* rp,
                          uint32_t* d_tileHeads, VhTileBlock* d_tileBlocks, uint32_t tileCapacity, uint32_t* d_schedule, uint32_t phase,
                          uint32_t* d_longestList, vhStream_t stream);
/* d_schedule (may be NULL) is vh_render_schedule_bytes() of device memory, zeroed once, that belongs to one sequence
 * of splat + render calls; phase is that sequence's call counter (1, 2, 3, ...; the same value for the splat and the
 * render of one frame).  The ray caster stores the cost every tile had; the next splat sorts the tiles by it and
 * deals them to the workgroups so that the compute units get even loads.  The maps do not depend on it, and a render
 * whose splat was given no schedule uses raster order. */
size_t vh_render_schedule_bytes(uint32_t width, uint32_t height);
/* how many tiles of an image of this size a scheduled render marches with two waves each (the dearest ones; 0 for
 * small images): lets a test make sure it exercises that path */
uint32_t vh_render_split_tiles(uint32_t width, uint32_t height);
int vh_render_intervals(const VhHashData* hd, const VhHashParams* hp, const VhRayCastData* rd, const VhDepthCameraParams* cp,
                        const VhRayCastParams* rp, uint32_t* d_tileHeads, const VhTileBlock* d_tileBlocks, uint32_t tileCapacity,
                        uint32_t* d_schedule, uint32_t phase, vhStream_t stream);
/* computeNormals(float4* d_output, float4* d_input, width, height)  DSC/CameraUtil.cu:699 */
int vh_compute_normals(float* d_output4, const float* d_input4, uint32_t width, uint32_t height, vhStream_t stream);
/* The same two launches with the passes of a frame job riding along as extra workgroups (job may be NULL, or already
 * launched: then they equal the plain calls): the alloc pass behind the ray caster's workgroups, where it fills the
 * tail the dearest tiles leave, and the compactify pass behind computeNormals'.  Legal because a block allocated while
 * rays are marched holds only unobserved voxels, which a sample treats like an absent block. */
int vh_render_intervals_co(const VhHashData* hd, const VhHashParams* hp, const VhRayCastData* rd, const VhDepthCameraParams* cp,
                           const VhRayCastParams* rp, uint32_t* d_tileHeads, const VhTileBlock* d_tileBlocks, uint32_t tileCapacity,
                           uint32_t* d_schedule, uint32_t phase, VhFrameJob* job, vhStream_t stream);
int vh_compute_normals_co(float* d_output4, const float* d_input4, uint32_t width, uint32_t height, VhFrameJob* job, vhStream_t stream);
/* ... and, when the job's compactify pass rides along and nextView != NULL, the interval splat of the NEXT render as
 * well (arguments as vh_ray_interval_splat; nextView holds that render's view matrices: the pose of the job's frame).
 * It lists the table as it stands BEFORE the job's frame is integrated: blocks that pass frees stay listed with all-zero
 * voxels (read like absent ones), blocks allocated later are empty.  Whoever uses it must make sure nothing else edits
 * the table in between (the host class CUDARayCastSDF checks the job's frame number and table epoch). */
int vh_compute_normals_co2(float* d_output4, const float* d_input4, uint32_t width, uint32_t height, VhFrameJob* job,
                           const VhRayCastParams* nextView, uint32_t* d_tileHeads, VhTileBlock* d_tileBlocks, uint32_t tileCapacity,
                           uint32_t* d_schedule, uint32_t phase, uint32_t* d_longestList, vhStream_t stream);

/* Read-back without a blocking call: one thread writes {*d_src0, *d_src1 (0 where NULL), tag} -- the tag last, with
 * system scope -- to d_mapped, the device alias of three words of mapped pinned host memory (vh_malloc_host memory is
 * mapped).  The host polls word 2 for the tag.  Replaces the reference's blocking cudaMemcpy of the streaming counters
 * (DSC/CUDASceneRepChunkGrid.cu:88, :140). */
int vh_publish_words(const uint32_t* d_src0, const uint32_t* d_src1, uint32_t* d_mapped, uint32_t tag, vhStream_t stream);

/* ---- streaming launchers: DSC/CUDASceneRepChunkGrid.h:142-146 --------------- */
/* integrateFromGlobalHashPass1CUDA(params, hashData, threadsPerPart, start, radius, camPos,
 *                                  d_outputCounter, d_output)   DSC/CUDASceneRepChunkGrid.cu:76 */
int vh_stream_out_pass1(const VhHashData* hd, const VhHashParams* hp, uint32_t threadsPerPart, uint32_t start,
                        float radius, const float camPos[3], uint32_t* d_outputCounter, VhSDFBlockDesc* d_output,
                        uint32_t outputCapacity, int32_t lockToken, vhStream_t stream);
/* The scan of vh_stream_out_pass1 without its deletes: the number of blocks the pass would move out, published as
 * {count, 0, tag} to d_mapped (device alias of mapped host memory, see vh_publish_words); *d_counter must be zero and is
 * zero again afterwards.  No reference twin: it lets a frame loop that knows its poses ahead skip the streaming step of a
 * frame in which nothing would stream (CUDASceneRepChunkGrid::probeStreamOut). */
int vh_stream_out_probe(const VhHashData* hd, const VhHashParams* hp, uint32_t threadsPerPart, uint32_t start, float radius,
                        const float camPos[3], uint32_t* d_counter, uint32_t* d_mapped, uint32_t tag, vhStream_t stream);
/* integrateFromGlobalHashPass2CUDA(params, hashData, threadsPerPart, descs, d_output, n)  :115 */
int vh_stream_out_pass2(const VhHashData* hd, const VhHashParams* hp, const VhSDFBlockDesc* d_descs,
                        VhVoxel* d_output, uint32_t nSDFBlocks, vhStream_t stream);
/* ---- the streaming passes for a caller that does not wait for the device (not in the reference: its host reads a counter back
 * before every second launch, DSC/CUDASceneRepChunkGrid.cu:88,140).
 * vh_stream_out_device: integrateFromGlobalHashPass1CUDA + Pass2CUDA in one call.  The counter is cleared, pass 1 lists at most
 * `mostBlocks` blocks (an upper bound the caller has from vh_stream_out_probe) and sets, in d_bitMask (may be NULL), the bit of
 * every listed block's chunk -- what the host's integrateInChunkGrid does when the block arrives; pass 2 is launched for
 * mostBlocks blocks and reads the count on the device.
 * vh_publish_count: {*d_counter, 0, tag} into mapped host memory (tag last, system scope), to be enqueued behind the copies of
 * the pass's output.
 * vh_stream_in_device: chunkToGlobalHashPass1CUDA + Pass2CUDA + the heap counter's update with the counter read on the device;
 * clears bit `chunkBit` of d_bitMask (0xffffffff: none); publishes {blocks that found no slot (listed in d_failed[1..]), heap
 * counter before the pass, tag, 1 if the heap held too few free blocks -- then nothing was inserted} to d_mapped. */
int vh_stream_out_device(const VhHashData* hd, const VhHashParams* hp, uint32_t threadsPerPart, uint32_t start, float radius,
                         const float camPos[3], uint32_t* d_outputCounter, VhSDFBlockDesc* d_descs, VhVoxel* d_blocks,
                         uint32_t mostBlocks, int32_t lockToken, uint32_t* d_bitMask, vhStream_t stream);
int vh_publish_count(const uint32_t* d_counter, uint32_t* d_mapped, uint32_t tag, vhStream_t stream);
int vh_stream_in_device(const VhHashData* hd, const VhHashParams* hp, uint32_t n, const VhSDFBlockDesc* d_descs, const VhVoxel* d_blocks,
                        int32_t lockToken, uint32_t* d_failed, uint32_t* d_bitMask, uint32_t chunkBit, uint32_t* d_mapped, uint32_t tag,
                        vhStream_t stream);
/* chunkToGlobalHashPass1CUDA(params, hashData, n, heapCountPrev, descs, blocks)           :162 */
int vh_stream_in_pass1(const VhHashData* hd, const VhHashParams* hp, uint32_t n, uint32_t heapCountPrev,
                       const VhSDFBlockDesc* d_descs, int32_t lockToken, vhStream_t stream);
/* the same, reporting which blocks could not be inserted (their bucket and its list were full, or two blocks of the
 * pass overflowed one bucket): d_failed[0], zeroed by the caller, counts them, d_failed[1 ..] lists their indices into
 * d_descs (room for n).  The reference has no such case handling (its overflow branch is an unported remnant). */
int vh_stream_in_pass1_report(const VhHashData* hd, const VhHashParams* hp, uint32_t n, uint32_t heapCountPrev,
                              const VhSDFBlockDesc* d_descs, int32_t lockToken, uint32_t* d_failed, vhStream_t stream);
/* chunkToGlobalHashPass2CUDA(params, hashData, n, heapCountPrev, descs, blocks)           :192 */
int vh_stream_in_pass2(const VhHashData* hd, const VhHashParams* hp, uint32_t n, uint32_t heapCountPrev,
                       const VhSDFBlockDesc* d_descs, const VhVoxel* d_blocks, vhStream_t stream);

/* ---- utilities that are not in the reference -------------------------------- */
/* Synthetic analytic-sphere depth+colour frame,
 * generated on the device so benchmark inputs are HBM-resident. */
int vh_synth_frame(const double* h_spheres, int nSpheres, int inside, const float camToWorld[16],
                   const VhDepthCameraParams* cp, float* d_depth, float* d_color4, vhStream_t stream);
/* Executes a list of hash operations one after the other in a single thread:
 * deterministic exercise of allocBlock / deleteHashEntryElement /
 * insertHashEntry / getHashEntryForSDFBlockPos including collision lists.
 * ops: n x {op, x, y, z, arg}; results: n ints.  op 0 = alloc, 1 = delete,
 * 2 = insert(ptr = arg), 3 = lookup (result = ptr), 4 = new lock pass. */
enum { VH_OP_ALLOC = 0, VH_OP_DELETE = 1, VH_OP_INSERT = 2, VH_OP_LOOKUP = 3, VH_OP_NEW_PASS = 4 };
int vh_debug_hash_ops(const VhHashData* hd, const VhHashParams* hp, const int32_t* d_ops, int32_t* d_results,
                      uint32_t n, vhStream_t stream);

/* Self-check of the two exact shortcuts the ray caster uses: division by the
 * voxel size through a reciprocal with two correction steps, and modulo by the
 * bucket count through a multiply-shift.  d_mismatches[0] / [1] = number of
 * operands (of n pseudo-random ones) where the shortcut differs from `/` / `%`. */
int vh_debug_check_fast_math(float divisor, uint32_t modulus, uint32_t n, uint32_t seed, uint32_t* d_mismatches, vhStream_t stream);

/* Self-check of the division the fused integrate pass uses for blocks it has certified (one refined reciprocal shared by
 * the two perspective divisions of a voxel; the same for the blend's division by the weight sum): n pseudo-random
 * operand pairs inside the certified ranges against `/`.  d_mismatches[0]: projection range, [1]: blend range. */
int vh_debug_check_refined_division(uint32_t n, uint32_t seed, uint32_t* d_mismatches, vhStream_t stream);
/* measurement, not part of the path: every SIMD of the device runs `wavesPerSimd` waves (1..8), each a chain-free stream of
 * 32 * iters vector instructions (mode 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_add_u32, 3 v_mul_lo_u32); per wave
 * {start, end in 100 MHz ticks, s_memtime ticks spent, HW_ID[19:0] | XCC_ID << 20} into d_stamps (4 words per wave, *numWaves waves: room for
 * 8 * 4 * the device's compute units).  tools/valu_issue_probe.py turns that into cycles per wave-instruction per SIMD */
int vh_debug_valu_probe(uint32_t mode, uint32_t wavesPerSimd, uint32_t iters, uint32_t* d_stamps, uint32_t* numWaves, vhStream_t stream);

/* ---- host classes (opaque handles over the C++ classes of include/vh.hpp) ---- */
typedef struct VhSceneRep VhSceneRep;   /* CUDASceneRepHashSDF,   DSC/CUDASceneRepHashSDF.h:28 */
typedef struct VhRayCast VhRayCast;     /* CUDARayCastSDF,        DSC/CUDARayCastSDF.h:13 */
typedef struct VhChunkGrid VhChunkGrid; /* CUDASceneRepChunkGrid, DSC/CUDASceneRepChunkGrid.h:152 */

/* CUDASceneRepHashSDF(const HashParams&) :31 ; the five GlobalAppState flags arrive as options */
int vh_scene_rep_create(const VhHashParams* hp, const VhSceneOptions* opt, vhStream_t stream, VhSceneRep** out);
void vh_scene_rep_destroy(VhSceneRep* s);
/* integrate(lastRigidTransform, depthCameraData, depthCameraParams, d_bitMask) :64 */
int vh_scene_rep_integrate(VhSceneRep* s, const float rigidTransform[16], const VhDepthCameraData* cam,
                           const VhDepthCameraParams* cp, const uint32_t* d_bitMask);
/* setLastRigidTransformAndCompactify :90 */
int vh_scene_rep_set_last_rigid_transform_and_compactify(VhSceneRep* s, const float rigidTransform[16],
                                                         const VhDepthCameraParams* cp);
/* reset() :101 */
int vh_scene_rep_reset(VhSceneRep* s);
/* getHashData() :112 / getHashParams() :116 / getLastRigidTransform() :96 */
int vh_scene_rep_get_hash_data(VhSceneRep* s, VhHashData* out);
int vh_scene_rep_get_hash_params(VhSceneRep* s, VhHashParams* out);
/* getHeapFreeCount() :122 (blocking read-back) */
int vh_scene_rep_get_heap_free_count(VhSceneRep* s, uint32_t* out);
/* blocking, exact count of in-frustum blocks of the last compactify */
int vh_scene_rep_get_num_occupied_blocks(VhSceneRep* s, uint32_t* out);
/* debugHash() :129-233: 0 if every invariant holds; report = {numOccupied, numFree, duplicates, lockEntries} */
int vh_scene_rep_debug_hash(VhSceneRep* s, uint32_t report[4]);
/* device-side status words (heap underflow, failed inserts, lost lock races): copies VH_STATE_WORDS words */
int vh_scene_rep_get_state(VhSceneRep* s, uint32_t* out);
/* per-stage device time in ms accumulated while s_timingsDetailledEnabled:
 * {alloc, compactify, integrate(+gc), count} (TimingLog of the reference) */
int vh_scene_rep_get_timings(VhSceneRep* s, double out[4]);
int vh_scene_rep_set_options(VhSceneRep* s, const VhSceneOptions* opt);

/* integrateAhead / integrateFinish: the two halves of integrate() (include/vh.hpp).  *job receives the frame's alloc +
 * compactify passes for vh_raycast_render_co (NULL when the scene's options rule a co-launch out); it belongs to the
 * scene and is valid until vh_scene_rep_integrate_finish */
int vh_scene_rep_integrate_ahead(VhSceneRep* s, const float rigidTransform[16], const VhDepthCameraData* cam,
                                 const VhDepthCameraParams* cp, const uint32_t* d_bitMask, VhFrameJob** job);
int vh_scene_rep_integrate_finish(VhSceneRep* s, const VhDepthCameraData* cam, const VhDepthCameraParams* cp);

/* CUDARayCastSDF(const RayCastParams&) :16 */
int vh_raycast_create(const VhRayCastParams* rp, vhStream_t stream, VhRayCast** out);
void vh_raycast_destroy(VhRayCast* r);
/* render(hashData, hashParams, cameraData, lastRigidTransform), DSC/CUDARayCastSDF.cpp:38 */
int vh_raycast_render(VhRayCast* r, const VhHashData* hd, const VhHashParams* hp,
                      const VhDepthCameraParams* cp, const float lastRigidTransform[16]);
/* the same with a frame job riding along (may be NULL) */
int vh_raycast_render_co(VhRayCast* r, const VhHashData* hd, const VhHashParams* hp,
                         const VhDepthCameraParams* cp, const float lastRigidTransform[16], VhFrameJob* job);
/* getRayCastData() :42 / getRayCastParams() :45 */
int vh_raycast_get_data(VhRayCast* r, VhRayCastData* out);
int vh_raycast_get_params(VhRayCast* r, VhRayCastParams* out);
/* device time in ms of render() accumulated while timing is enabled: {raycast (march kernel), normals, count,
 * interval splat} */
int vh_raycast_get_timings(VhRayCast* r, double out[4]);
/* ms an event pair reads with nothing between its two records (sampled while every stage is timed): what a bracketed
 * launch's reading holds beside the kernel */
int vh_raycast_get_event_pair_overhead(VhRayCast* r, double* ms);
int vh_raycast_set_timing(VhRayCast* r, int enabled); /* 0 off, 1 every stage, 2 the march kernel only */
/* same, timing only every stride-th render() (an event record idles the queue for a few microseconds) */
int vh_raycast_set_timing_stride(VhRayCast* r, int enabled, uint32_t stride);
/* 1 (default): render() splats ray intervals first; 0: march the full depth range as this fork of the reference does */
int vh_raycast_set_interval_splatting(VhRayCast* r, int enabled);

/* CUDASceneRepChunkGrid(sceneRep, voxelExtends, gridDimensions, minGridPos, initialChunkListSize,
 *                       streamingEnabled, streamOutParts)         DSC/CUDASceneRepChunkGrid.h:155 */
int vh_chunk_grid_create(VhSceneRep* s, const float voxelExtents[3], const int32_t gridDimensions[3],
                         const int32_t minGridPos[3], uint32_t initialChunkListSize, int streamingEnabled,
                         uint32_t streamOutParts, VhChunkGrid** out);
void vh_chunk_grid_destroy(VhChunkGrid* g);
/* streamOutToCPUPass0GPU(posCamera, radius, useParts, multiThreaded)  DSC/CUDASceneRepChunkGrid.cpp:55 */
int vh_chunk_grid_stream_out_to_cpu_pass0_gpu(VhChunkGrid* g, const float posCamera[3], float radius, int useParts, int multiThreaded);
/* streamOutToCPUPass1CPU(multiThreaded) :107 */
int vh_chunk_grid_stream_out_to_cpu_pass1_cpu(VhChunkGrid* g, int multiThreaded);
/* streamInToGPUPass0CPU(posCamera, radius, useParts, multiThreaded) :208 */
int vh_chunk_grid_stream_in_to_gpu_pass0_cpu(VhChunkGrid* g, const float posCamera[3], float radius, int useParts, int multiThreaded);
/* streamInToGPUPass1GPU(multiThreaded) :227 */
int vh_chunk_grid_stream_in_to_gpu_pass1_gpu(VhChunkGrid* g, int multiThreaded);
/* streamOutToCPU / streamInToGPU (both passes, single-threaded) :44 / :197 ; nStreamedBlocks out */
int vh_chunk_grid_stream_out_to_cpu(VhChunkGrid* g, const float posCamera[3], float radius, int useParts, uint32_t* nStreamedBlocks);
int vh_chunk_grid_stream_in_to_gpu(VhChunkGrid* g, const float posCamera[3], float radius, int useParts, uint32_t* nStreamedBlocks);
/* streamOutToCPUAll() :31 / streamInToGPUAll(posCamera, radius, useParts, n) :164 */
int vh_chunk_grid_stream_out_to_cpu_all(VhChunkGrid* g);
int vh_chunk_grid_stream_in_to_gpu_all(VhChunkGrid* g, const float posCamera[3], float radius, int useParts, uint32_t* nStreamedBlocks);
/* getBitMaskGPU() DSC/CUDASceneRepChunkGrid.h:306 (uploads only when the mask changed) */
int vh_chunk_grid_get_bit_mask_gpu(VhChunkGrid* g, const uint32_t** d_bitMask);
/* reset() :297 */
int vh_chunk_grid_reset(VhChunkGrid* g);
/* debugCheckForDuplicates() DSC/CUDASceneRepChunkGrid.cpp:313: 0 if no block is present twice */
int vh_chunk_grid_debug_check_for_duplicates(VhChunkGrid* g);
/* host-side statistics: {chunks allocated, blocks on the host, bits set} */
int vh_chunk_grid_get_statistics(VhChunkGrid* g, uint32_t out[3]);
/* blocks that stream-in passes could not insert (bucket and list full, or a second overflow of one bucket in a pass) and
 * that went back to the host grid for a later pass; not in the reference, which has no defined behaviour there */
int vh_chunk_grid_get_num_failed_inserts(VhChunkGrid* g, uint32_t* out);
/* copies the host chunk grid content: descs[n], blocks[n*512] (pass NULL to query n) */
int vh_chunk_grid_download_host_blocks(VhChunkGrid* g, VhSDFBlockDesc* descs, VhVoxel* blocks, uint32_t capacity, uint32_t* n);
/* saveToFile / loadFromFile (.hashgrid v1) DSC/CUDASceneRepChunkGrid.h:459-548 */
int vh_chunk_grid_save_to_file(VhChunkGrid* g, const char* filename, const float camPos[3], float radius);
int vh_chunk_grid_load_from_file(VhChunkGrid* g, const char* filename, const float camPos[3], float radius);

/* ---- the frame loop: reconstruction(), DSC/DepthSensing.cpp:720-924, headless over a recorded sequence at given
 * poses ("build's headless vh_bench / vh_replay driver").  Per frame, in the reference's order:
 * render(pose of the previous frame) -> [stream out / stream in around the camera] -> integrate(pose, depth, colour,
 * bit mask).  One call enqueues any number of frames; nothing in it waits for the device unless streaming is on (its
 * read-backs) or the run-ahead bound is reached.  The scene, the ray caster and the chunk grid stay the caller's. */
typedef struct VhReconstruction VhReconstruction;
void vh_reconstruction_default_options(VhReconstructionOptions* out);
int vh_reconstruction_create(VhSceneRep* scene, VhRayCast* rayCast, VhChunkGrid* chunkGrid /* may be NULL */,
                             const VhDepthCameraParams* cp, const VhReconstructionOptions* opt, VhReconstruction** out);
void vh_reconstruction_destroy(VhReconstruction* r);
/* processes frames[0..n): frame numbers continue from the previous call (the first frame of all is not ray-cast) */
int vh_reconstruction_run(VhReconstruction* r, const VhSequenceFrame* frames, uint32_t n);
/* the same for a caller that feeds the loop a few frames at a time and knows what comes next: `next` is the frame that
 * will follow frames[n-1] (only its pose is read; NULL: unknown).  With streaming on, the loop then asks the device
 * about that frame's streaming step behind the last frame's alloc pass, as it does inside a call (not in the reference) */
int vh_reconstruction_run_ahead(VhReconstruction* r, const VhSequenceFrame* frames, uint32_t n, const VhSequenceFrame* next);
/* waits for everything the loop has enqueued (all its streams) */
int vh_reconstruction_synchronize(VhReconstruction* r);
int vh_reconstruction_get_stats(VhReconstruction* r, VhReconstructionStats* out);
/* test hook: the n-th ray cast from now fails with an error instead of running (0: off); the loop must unwind cleanly */
int vh_reconstruction_debug_fail_render(VhReconstruction* r, uint32_t nthRenderFromNow);
/* frame counter and statistics back to zero (the scene is the caller's to reset) */
int vh_reconstruction_reset(VhReconstruction* r);

/* ---- sensor pre-processing: the image kernels of DSC/CameraUtil.cu that CUDARGBDAdapter::process
 * (DSC/CUDARGBDAdapter.cpp:93-137) and CUDARGBDSensor::process (DSC/CUDARGBDSensor.cpp:147-257) run on every frame.
 * Device pointers; float4 maps are passed as float* (4 per pixel); MINF marks an invalid pixel.  The filters read
 * their whole neighbourhood, so they do not work in place. */
int vh_convert_color_raw_to_float4(float* d_output4, const uint8_t* d_inputRGBX, uint32_t width, uint32_t height, vhStream_t stream); /* :154 */
/* not in the reference: a sensor frame read by a kernel straight from pinned, device-visible host memory (the pointers
 * as hipHostGetDevicePointer returns them): depth copied, RGBX colour converted to float4 on the way.  width*height
 * must be a multiple of 4.  hostRGBX / d_color4 may be NULL. */
int vh_upload_frame(const float* hostDepth, const uint8_t* hostRGBX, float* d_depth, float* d_color4, uint32_t width, uint32_t height, vhStream_t stream);
int vh_resample_float_map(float* d_output, uint32_t outputWidth, uint32_t outputHeight, const float* d_input,
                          uint32_t inputWidth, uint32_t inputHeight, vhStream_t stream);                                       /* :1120 */
int vh_resample_float4_map(float* d_output4, uint32_t outputWidth, uint32_t outputHeight, const float* d_input4,
                           uint32_t inputWidth, uint32_t inputHeight, vhStream_t stream);                                      /* :1188 */
int vh_copy_float_map(float* d_output, const float* d_input, uint32_t width, uint32_t height, vhStream_t stream);              /* :37  */
int vh_copy_float4_map(float* d_output4, const float* d_input4, uint32_t width, uint32_t height, vhStream_t stream);           /* :120 */
int vh_set_invalid_float_map(float* d_output, uint32_t width, uint32_t height, vhStream_t stream);                             /* :348 */
int vh_convert_color_to_intensity_float(float* d_output, const float* d_input4, uint32_t width, uint32_t height, vhStream_t stream); /* :269 */
int vh_convert_depth_float_to_camera_space_float4(float* d_output4, const float* d_input, const VhDepthCameraParams* cp,
                                                  uint32_t width, uint32_t height, vhStream_t stream);                         /* :409 */
int vh_gauss_filter_float_map(float* d_output, const float* d_input, float sigmaD, float sigmaR, uint32_t width, uint32_t height, vhStream_t stream);    /* :595 */
int vh_gauss_filter_float4_map(float* d_output4, const float* d_input4, float sigmaD, float sigmaR, uint32_t width, uint32_t height, vhStream_t stream); /* :653 */
int vh_bilateral_filter_float_map(float* d_output, const float* d_input, float sigmaD, float sigmaR, uint32_t width, uint32_t height, vhStream_t stream); /* :485 */
int vh_erode_depth_map(float* d_output, const float* d_input, int32_t structureSize, uint32_t width, uint32_t height, float dThresh,
                       float fracReq, vhStream_t stream);                                                                       /* :1672 */

/* ---- parameter files: zParameters*.txt as mLib's ParameterFile reads them
 * (DSCroot/Include/mLib/include/core-util/parameterFile.h:22-60,136-172: per line, cut at the first "//", "#" or ";",
 * strip blanks / quotes / semicolons, split at the first "="; numbers by stoi / stof, bool false iff "false", "False" or
 * "0") and the parametersFromGlobalAppState builders of the host classes. */
int vh_app_state_read(const char* filename, VhAppState* out);
int vh_app_state_parse(const char* text, VhAppState* out); /* the same on a string */
void vh_hash_params_from_app_state(const VhAppState* gas, VhHashParams* out);        /* DSC/CUDASceneRepHashSDF.h:38-58 */
void vh_raycast_params_from_app_state(const VhAppState* gas, const float intrinsics[16], const float intrinsicsInv[16],
                                      VhRayCastParams* out);                          /* DSC/CUDARayCastSDF.h:24-40 */
void vh_marching_cubes_params_from_app_state(const VhAppState* gas, VhMarchingCubesParams* out); /* DSC/CUDAMarchingCubesHashSDF.h:19-28 */
void vh_scene_options_from_app_state(const VhAppState* gas, VhSceneOptions* out);

/* handle level: CUDARGBDSensor over CUDARGBDAdapter (include/vh.hpp).  config = {depthW, depthH, colorW, colorH, adapterW,
 * adapterH} and {fx, fy, mx, my, sensorDepthMin, sensorDepthMax}. */
typedef struct VhRGBDSensor VhRGBDSensor;
int vh_rgbd_sensor_create(const uint32_t sizes[6], const float intrinsics[6], vhStream_t stream, VhRGBDSensor** out);
void vh_rgbd_sensor_destroy(VhRGBDSensor* s);
int vh_rgbd_sensor_set_filter_depth_values(VhRGBDSensor* s, int enabled, float sigmaD, float sigmaR);
int vh_rgbd_sensor_set_filter_intensity_values(VhRGBDSensor* s, int enabled, float sigmaD, float sigmaR);
int vh_rgbd_sensor_process(VhRGBDSensor* s, const float* h_depthFloat, const uint8_t* h_colorRGBX);
int vh_rgbd_sensor_get_depth_camera_data(VhRGBDSensor* s, VhDepthCameraData* out);
int vh_rgbd_sensor_get_depth_camera_params(VhRGBDSensor* s, VhDepthCameraParams* out);
/* device maps at adapter resolution: {camera space float4, normals float4, intensity float} */
int vh_rgbd_sensor_get_maps(VhRGBDSensor* s, float** d_cameraSpace4, float** d_normals4, float** d_intensity);

/* ---- recorded sequences: the `.sens` container and its reader.
 *   VhSensorData        ml::SensorData (load / save / frames)          DSC/sensorData/sensorData.h:608-830
 *   VhSensorDataReader  SensorDataReader (the sensor the loop polls)    DSC/SensorDataReader.cpp:39-179
 * Host side only; depth raw / zlib, colour raw / PNG / baseline JPEG are decoded (see vh_sensor_data.cpp). */
typedef struct VhSensorData VhSensorData;
int vh_sensor_data_create(const VhSensorDataInfo* header, VhSensorData** out); /* empty sequence with this header */
int vh_sensor_data_load(const char* filename, VhSensorData** out);             /* loadFromFile :789-830 */
void vh_sensor_data_destroy(VhSensorData* s);
int vh_sensor_data_save(const VhSensorData* s, const char* filename);          /* saveToFile :756-787 */
int vh_sensor_data_info(const VhSensorData* s, VhSensorDataInfo* out);
/* addFrame :657-667.  colorRGB (3 bytes / pixel) or depth may be NULL; stored with the header's compression types
 * (depth raw / zlib, colour raw). */
int vh_sensor_data_add_frame(VhSensorData* s, const uint8_t* colorRGB, const uint16_t* depth, const float cameraToWorld[16],
                             uint64_t timeStampColor, uint64_t timeStampDepth);
/* stores an already compressed colour frame (PNG / JPEG bytes as another tool produced them) with the frame's depth */
int vh_sensor_data_add_frame_compressed(VhSensorData* s, const uint8_t* colorBytes, uint64_t numColorBytes, const uint16_t* depth,
                                        const float cameraToWorld[16], uint64_t timeStampColor, uint64_t timeStampDepth);
int vh_sensor_data_add_imu_frame(VhSensorData* s, const double values15[15], uint64_t timeStamp);
/* decompressDepthAlloc / decompressColorAlloc :687-705 and the frame's pose and time stamps; any output may be NULL */
int vh_sensor_data_get_frame(const VhSensorData* s, uint64_t frameIdx, uint16_t* depth, uint8_t* colorRGB, float cameraToWorld[16],
                             uint64_t timeStamps[2]);

typedef struct VhSensorDataReader VhSensorDataReader;
int vh_sensor_data_reader_create(const char* filename, VhSensorDataReader** out); /* createFirstConnected */
void vh_sensor_data_reader_destroy(VhSensorDataReader* r);
int vh_sensor_data_reader_info(const VhSensorDataReader* r, VhSensorDataInfo* out);
/* processDepth: decodes the next frame.  *gotFrame = 0 once the sequence is complete.  The pointers stay valid until
 * the next call: depth in metres (depthWidth*depthHeight floats), colour {r, g, b, 1} (colorWidth*colorHeight*4). */
int vh_sensor_data_reader_process_depth(VhSensorDataReader* r, int* gotFrame, const float** depthFloat, const uint8_t** colorRGBX);
int vh_sensor_data_reader_get_rigid_transform(const VhSensorDataReader* r, int offset, float out[16]); /* getRigidTransform :172-179 */
int vh_sensor_data_reader_get_curr_frame(const VhSensorDataReader* r, uint32_t* currFrame, uint32_t* numFrames);

/* ---- projective ICP camera tracking.  Launcher level: the steps of one alignment, each a kernel
 * that reads and updates a VhIcpState in device memory (a step returns at once when the state says "lost" or "level
 * done"), so that a whole multi-resolution solve runs without a host round trip.
 *   vh_icp_projective_correspondences  projectiveCorrespondences           DSC/CUDAImageHelper.cu:70-145
 *   vh_icp_build_linear_system         buildLinearSystem (per-wave terms)  DSC/CUDABuildLinearSystem.cu:130-204
 *   vh_icp_solve                       reductionSystemCPU + computeBestRigidAlignment + delinearizeTransformation +
 *                                      the early-out of align               DSC/CUDABuildLinearSystem.cpp:52-92,
 *                                                                           DSC/CUDACameraTrackingMultiRes.cpp:186-253,306-318 */
int vh_icp_begin(VhIcpState* d_state, const float* d_deltaEstimate16, vhStream_t stream);
int vh_icp_begin_level(VhIcpState* d_state, vhStream_t stream);
int vh_icp_projective_correspondences(const float* d_input4, const float* d_inputNormals4, const float* d_target4, const float* d_targetNormals4,
                                      float* d_output4, float* d_outputNormals4, uint32_t width, uint32_t height, float distThres, float normalThres,
                                      float levelFactor, const VhIcpState* d_state, const VhDepthCameraParams* cp, vhStream_t stream);
uint32_t vh_icp_num_partials(uint32_t width, uint32_t height); /* rows of 30 floats vh_icp_build_linear_system writes */
int vh_icp_build_linear_system(uint32_t width, uint32_t height, float* d_partials, const float* d_input4, const float* d_corr4,
                               const float* d_corrNormals4, const VhIcpState* d_state, vhStream_t stream);
int vh_icp_solve(VhIcpState* d_state, const float* d_partials, uint32_t numPartials, float angleThres, float distThres, float earlyOutResidual,
                 int lastInnerIteration, vhStream_t stream);
/* GlobalCameraTrackingState::readMembers on zParametersTracking*.txt (DSC/GlobalCameraTrackingState.h:14-60) */
int vh_tracking_state_read(const char* filename, VhTrackingState* out);
int vh_tracking_state_parse(const char* text, VhTrackingState* out);

/* handle level: CUDACameraTrackingMultiRes (DSC/CUDACameraTrackingMultiRes.h:17-78) */
typedef struct VhCameraTracking VhCameraTracking;
int vh_camera_tracking_create(uint32_t imageWidth, uint32_t imageHeight, uint32_t levels, vhStream_t stream, VhCameraTracking** out);
void vh_camera_tracking_destroy(VhCameraTracking* t);
/* applyCT(dInput, dInputNormals, -, dModel, dModelNormals, -, lastTransform, <per-level settings>, condThres, angleThres,
 * deltaTransformEstimate, ...) :241-289: returns lastTransform * delta in transformOut, every entry -inf if tracking was lost
 * (trackingLost = 1).  state (may be NULL) receives the final VhIcpState. */
int vh_camera_tracking_apply_ct(VhCameraTracking* t, float* d_input4, float* d_inputNormals4, float* d_model4, float* d_modelNormals4,
                                const float lastTransform[16], const VhTrackingState* settings, const float deltaTransformEstimate[16],
                                const VhDepthCameraParams* cp, float transformOut[16], int* trackingLost, VhIcpState* state);

/* ---- marching cubes -------------------------------------------------------------------------
 * launcher level: resetMarchingCubesCUDA / extractIsoSurfacePass1CUDA / extractIsoSurfacePass2CUDA
 * (DSC/CUDAMarchingCubesSDF.cu:29-40, 94-105, 132-143).  The reference passes a RayCastData only for its member
 * function trilinearInterpolationSimpleFastFast; no buffer of it is read, so it is not a parameter here.
 * d_numTriangles counts every triangle produced; those beyond m_maxNumTriangles are dropped (the reference clamps
 * the counter instead). */
int vh_marching_cubes_data_alloc(VhMarchingCubesData* data, const VhMarchingCubesParams* params); /* MarchingCubesData::allocate */
void vh_marching_cubes_data_free(VhMarchingCubesData* data);
int vh_marching_cubes_update_params(const VhMarchingCubesData* data, const VhMarchingCubesParams* params, vhStream_t stream);
int vh_reset_marching_cubes(const VhMarchingCubesData* data, vhStream_t stream);
int vh_extract_iso_surface_pass1(const VhHashData* hd, const VhHashParams* hp, const VhMarchingCubesData* data, vhStream_t stream);
int vh_extract_iso_surface_pass2(const VhHashData* hd, const VhHashParams* hp, const VhMarchingCubesData* data,
                                 uint32_t numOccupiedBlocks, vhStream_t stream);

/* handle level: CUDAMarchingCubesHashSDF (DSC/CUDAMarchingCubesHashSDF.h:8-67) */
typedef struct VhMarchingCubes VhMarchingCubes;
int vh_marching_cubes_create(const VhMarchingCubesParams* params, vhStream_t stream, VhMarchingCubes** out);
void vh_marching_cubes_destroy(VhMarchingCubes* mc);
/* parametersFromGlobalAppState :19-28 */
int vh_marching_cubes_parameters(uint32_t maxNumTriangles, float threshFactor, float voxelSize, uint32_t hashNumBuckets,
                                 VhMarchingCubesParams* out);
int vh_marching_cubes_set_offline_processing(VhMarchingCubes* mc, int enabled);
/* extractIsoSurface(hashData, hashParams, rayCastData, minCorner, maxCorner, boxEnabled) .cpp:194-209 (copy = 1)
 * / extractIsoSurfaceWithoutCopy :211-224 (copy = 0) */
int vh_marching_cubes_extract_iso_surface(VhMarchingCubes* mc, const VhHashData* hd, const VhHashParams* hp,
                                          const float minCorner[3], const float maxCorner[3], int boxEnabled, int copy);
/* extractIsoSurface(chunkGrid, rayCastData, camPos, radius) .cpp:149-192 */
int vh_marching_cubes_extract_iso_surface_chunk_grid(VhMarchingCubes* mc, VhChunkGrid* grid, const float camPos[3], float radius);
int vh_marching_cubes_copy_triangles_to_cpu(VhMarchingCubes* mc);
int vh_marching_cubes_clear_mesh_buffer(VhMarchingCubes* mc);
/* counts of the last extraction: {triangles produced, occupied blocks} */
int vh_marching_cubes_get_counts(VhMarchingCubes* mc, uint32_t out[2]);
int vh_marching_cubes_download_triangles(VhMarchingCubes* mc, VhTriangle* out, uint32_t n);
/* the host mesh (getMetaDataf): sizes {vertices, face indices (3 per face; 0 = triangle soup)}, then the arrays */
int vh_marching_cubes_get_mesh_size(VhMarchingCubes* mc, uint64_t out[2]);
int vh_marching_cubes_get_mesh(VhMarchingCubes* mc, float* vertices3, float* colors4, uint32_t* faceIndices);
/* saveMesh(filename, transform, overwriteExistingFile) .cpp:89-145; transform may be NULL */
int vh_marching_cubes_save_mesh(VhMarchingCubes* mc, const char* filename, const float transform[16], int overwriteExistingFile);

#ifdef __cplusplus
}
#endif
#endif /* VH_API_H */
