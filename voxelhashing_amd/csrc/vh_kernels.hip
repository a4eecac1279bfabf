// vh_kernels.hip -- hand-written gfx950 kernels of the voxel-hash TSDF frame
// loop and their launcher-level C ABI (include/vh_api.h).
//
// Reference behaviour: DSC/CUDASceneRepHashSDF.cu, DSC/CUDARayCastSDF.cu,
// DSC/RayCastSDFUtil.h, DSC/CameraUtil.cu:669-711, DSC/CUDASceneRepChunkGrid.cu
// (DSC/ = /root/reference/DepthSensingCUDA/Source/).  Nothing here is derived
// from those kernels' structure: launch shapes, data movement and intra-wave
// cooperation are designed for CDNA4 (wave64, 16-byte lanes, no textures, no
// __constant__ singletons, no host round trips).
//
// MUST be compiled with -ffp-contract=off (see vh_device.hpp).
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/vh_api.h"
#include "vh_device.hpp"
#include "vh_host_util.hpp"

using namespace vhd;

namespace {

constexpr int kWave = 64;

VHD uint32_t lane_id() { return threadIdx.x & (kWave - 1); }
VHD uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// ---------------------------------------------------------------------------
// reset (resetHeapKernel / resetHashKernel / resetHashBucketMutexKernel,
// DSC/CUDASceneRepHashSDF.cu:23-61).  Voxels and summaries are cleared with
// hipMemsetAsync; these kernels write the non-zero patterns, 16 B per lane.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_reset_heap(uint32_t* heap, uint32_t* heapCounter, uint32_t n)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) heapCounter[0] = n - 1;
    if (idx < n) heap[idx] = n - idx - 1;
}

__global__ __launch_bounds__(256) void k_reset_hash(VhHashEntry* a, VhHashEntry* b, uint32_t ne)
{
    // one 16-byte half-entry per lane: even lanes {0,0,0,FREE}, odd lanes {offset=0,pad}
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 2ull * ne) {
        const int4 v = (idx & 1) ? make_int4(0, 0, 0, 0) : make_int4(0, 0, 0, VH_FREE_ENTRY);
        reinterpret_cast<int4*>(a)[idx] = v;
        reinterpret_cast<int4*>(b)[idx] = v;
    }
}

__global__ __launch_bounds__(256) void k_fill_i32(int32_t* p, int32_t v, uint32_t n)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) p[idx] = v;
}

// ---------------------------------------------------------------------------
// alloc (allocKernel, DSC/CUDASceneRepHashSDF.cu:158-243)
//
// One wave per 8x8 pixel tile.  Each lane walks its ray's SDF blocks with the
// reference's DDA; at every step the wave de-duplicates the block ids its
// lanes ask for (neighbouring rays hit the same 8^3 block) and ONE lane probes
// the table per distinct id, so the table sees ~1/64 of the reference's probes.
// ---------------------------------------------------------------------------

// worldToChunks :133-146, linearizeChunkPos :124-130, isSDFBlockStreamedOut :149-156
VHD bool block_streamed_out(const VhHashParams& hp, I3 blk, const uint32_t* bitMask)
{
    if (!bitMask) return false;
    F3 pw = block_to_world(hp.m_virtualVoxelSize, blk);
    F3 p = mk3(pw.x / hp.m_streamingVoxelExtents[0], pw.y / hp.m_streamingVoxelExtents[1], pw.z / hp.m_streamingVoxelExtents[2]);
    I3 c = mki3(f2i(p.x + (float)signi(p.x) * 0.5f), f2i(p.y + (float)signi(p.y) * 0.5f), f2i(p.z + (float)signi(p.z) * 0.5f));
    I3 q = mki3(c.x - hp.m_streamingMinGridPos[0], c.y - hp.m_streamingMinGridPos[1], c.z - hp.m_streamingMinGridPos[2]);
    uint32_t index = (uint32_t)(q.z * hp.m_streamingGridDimensions[0] * hp.m_streamingGridDimensions[1] +
                                q.y * hp.m_streamingGridDimensions[0] + q.x);
    return (bitMask[index >> 5] & (1u << (index & 31))) != 0u;
}

// What integrateDepthMapKernel reads of a pixel (DSC/CUDASceneRepHashSDF.cu:436-470), formed once per pixel instead
// of once per voxel that projects onto it: the depth, the colour as the bytes the kernel would make of it
// (uchar(255 * c), :466) and the weight of the sample (:463-464, a function of the depth alone).  Weight 0 marks a
// pixel that integrates nothing: invalid depth or colour, or beyond the integration distance (:443-445); a valid
// sample weighs at least 1.
VHD uint2 pack_pixel(const VhHashParams& hp, const VhDepthCameraParams& cp, float depth, bool hasColor, float4 c)
{
    uint32_t cw = 0u;
    if (hasColor && c.x != minf() && depth != minf() && depth < hp.m_maxIntegrationDistance) {
        const float depthZeroOne = cam_to_proj_z(cp, depth);
        const float weightUpdate = fmaxf((float)hp.m_integrationWeightSample * 1.5f * (1.0f - depthZeroOne), 1.0f);
        cw = pack_cw(f2uc(255.0f * c.x), f2uc(255.0f * c.y), f2uc(255.0f * c.z), f2uc(weightUpdate));
    }
    return make_uint2(__float_as_uint(depth), cw);
}

// one wave, one 8x8 pixel tile
VHD void alloc_tile(const VhHashData& hd, const VhHashParams& hp, const VhDepthCameraData& cam, const VhDepthCameraParams& cp,
                    const uint32_t* bitMask, int32_t lockToken, HashMod hm, uint2* packed, uint32_t tile)
{
    const uint32_t lane = lane_id();
    const uint32_t W = cp.m_imageWidth, H = cp.m_imageHeight;
    const uint32_t tilesX = (W + 7) / 8, tilesY = (H + 7) / 8;
    if (tile >= tilesX * tilesY) return; // wave-uniform
    const uint32_t x = (tile % tilesX) * 8 + (lane & 7), y = (tile / tilesX) * 8 + (lane >> 3);
    const float vs = hp.m_virtualVoxelSize;

    bool active = (x < W) && (y < H);
    float d = active ? cam.d_depthData[y * W + x] : minf();
    if (packed && active) { // the frame as the integrate pass reads it (a 128-byte row segment per 8 lanes)
        float4 c = make_float4(minf(), minf(), minf(), minf());
        if (cam.d_colorData) c = reinterpret_cast<const float4*>(cam.d_colorData)[y * W + x];
        packed[y * W + x] = pack_pixel(hp, cp, d, cam.d_colorData != nullptr, c);
    }
    if (d == minf() || d == 0.0f) active = false;
    if (d >= hp.m_maxIntegrationDistance) active = false;
    const float t = get_truncation(hp, d);
    const float minDepth = fminf(hp.m_maxIntegrationDistance, d - t);
    const float maxDepth = fminf(hp.m_maxIntegrationDistance, d + t);
    if (minDepth >= maxDepth) active = false;
    if (!__any(active)) return; // a tile without a measurement

    const F3 rayMin = mat_mul_p(hp.m_rigidTransform, depth_to_skeleton(cp, x, y, minDepth));
    const F3 rayMax = mat_mul_p(hp.m_rigidTransform, depth_to_skeleton(cp, x, y, maxDepth));
    const F3 rayDir = normalize3(mk3(rayMax.x - rayMin.x, rayMax.y - rayMin.y, rayMax.z - rayMin.z));

    // (world_to_block with the division by the voxel size as div_exact: five vector instructions instead of eleven, the same
    // quotient bit for bit -- tests/test_gpu_parity.py::test_exact_shortcuts; the ray caster's tap coordinates do the same)
    const float rvs = 1.0f / vs;
    I3 id = vvp_to_block(mki3(world_to_vvp1_rb(rayMin.x, vs, rvs), world_to_vvp1_rb(rayMin.y, vs, rvs), world_to_vvp1_rb(rayMin.z, vs, rvs)));
    const I3 idEnd = vvp_to_block(mki3(world_to_vvp1_rb(rayMax.x, vs, rvs), world_to_vvp1_rb(rayMax.y, vs, rvs), world_to_vvp1_rb(rayMax.z, vs, rvs)));

    const F3 step = mk3((float)signi(rayDir.x), (float)signi(rayDir.y), (float)signi(rayDir.z));
    const I3 cl = mki3(f2i(fmaxf(0.0f, fminf(step.x, 1.0f))), f2i(fmaxf(0.0f, fminf(step.y, 1.0f))), f2i(fmaxf(0.0f, fminf(step.z, 1.0f))));
    F3 boundaryPos = block_to_world(vs, mki3(id.x + cl.x, id.y + cl.y, id.z + cl.z));
    const float half = 0.5f * vs;
    boundaryPos.x -= half; boundaryPos.y -= half; boundaryPos.z -= half;
    F3 tMax = mk3((boundaryPos.x - rayMin.x) / rayDir.x, (boundaryPos.y - rayMin.y) / rayDir.y, (boundaryPos.z - rayMin.z) / rayDir.z);
    F3 tDelta = mk3((step.x * (float)VH_SDF_BLOCK_SIZE * vs) / rayDir.x, (step.y * (float)VH_SDF_BLOCK_SIZE * vs) / rayDir.y,
                    (step.z * (float)VH_SDF_BLOCK_SIZE * vs) / rayDir.z);
    const I3 idBound = mki3(f2i((float)idEnd.x + step.x), f2i((float)idEnd.y + step.y), f2i((float)idEnd.z + step.z));

    if (rayDir.x == 0.0f) { tMax.x = pinf(); tDelta.x = pinf(); }
    if (boundaryPos.x - rayMin.x == 0.0f) { tMax.x = pinf(); tDelta.x = pinf(); }
    if (rayDir.y == 0.0f) { tMax.y = pinf(); tDelta.y = pinf(); }
    if (boundaryPos.y - rayMin.y == 0.0f) { tMax.y = pinf(); tDelta.y = pinf(); }
    if (rayDir.z == 0.0f) { tMax.z = pinf(); tDelta.z = pinf(); }
    if (boundaryPos.z - rayMin.z == 0.0f) { tMax.z = pinf(); tDelta.z = pinf(); }

    uint32_t iter = 0;
    while (__any(active)) {
        // Steady state: the block exists and sits in the first slot of its bucket.  Every lane checks that for
        // its own block with ONE load (all lanes in flight together); only the rest -- new blocks and blocks
        // further down a bucket -- goes through the serial allocBlock below.  The reference asks "in the frustum?
        // not streamed out?" first; the tests are independent and a block that exists needs nothing, so the cheap
        // one goes first and the projection runs only for a block that is not where it is expected.
        bool want = active;
        if (want) {
            const int4 q0 = load_quad(&hd.d_hash[hash_pos_fast(hm, id) * VH_HASH_BUCKET_SIZE]);
            want = !quad_matches(q0, id);
        }
        if (want) want = block_in_frustum(hp, cp, id) && !block_streamed_out(hp, id, bitMask);
        // wave-level de-duplication of the requested block ids: one lane per distinct id, and those lanes allocate
        // side by side -- the chain lock -> heap counter -> heap slot -> entry is four trips to memory, walked once
        // for all the new blocks of this step instead of once per block
        uint64_t pending = __ballot(want);
        bool leads = false;
        while (pending) {
            const int leader = __ffsll((unsigned long long)pending) - 1;
            const int bx = __builtin_amdgcn_readlane(id.x, leader);
            const int by = __builtin_amdgcn_readlane(id.y, leader);
            const int bz = __builtin_amdgcn_readlane(id.z, leader);
            const bool same = want && id.x == bx && id.y == by && id.z == bz;
            pending &= ~__ballot(same);
            leads = leads || (int)lane == leader;
        }
        if (leads) alloc_block(hd, hp, id, lockToken);
        if (active) {
            if (tMax.x < tMax.y && tMax.x < tMax.z) {
                id.x = f2i((float)id.x + step.x);
                if (id.x == idBound.x) active = false;
                tMax.x += tDelta.x;
            } else if (tMax.z < tMax.y) {
                id.z = f2i((float)id.z + step.z);
                if (id.z == idBound.z) active = false;
                tMax.z += tDelta.z;
            } else {
                id.y = f2i((float)id.y + step.y);
                if (id.y == idBound.y) active = false;
                tMax.y += tDelta.y;
            }
            iter++;
            if (iter >= 1024u) active = false;
        }
    }
}

__global__ __launch_bounds__(512) void k_alloc(VhHashData hd, VhHashParams hp, VhDepthCameraData cam,
                                               VhDepthCameraParams cp, const uint32_t* bitMask, int32_t lockToken, HashMod hm, uint2* packed)
{
    // the compaction that follows needs its counter at zero: cleared here instead of by a separate memset
    if (blockIdx.x == 0 && threadIdx.x == 0) hd.d_hashCompactifiedCounter[0] = 0;
    alloc_tile(hd, hp, cam, cp, bitMask, lockToken, hm, packed, blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave));
}

// ---------------------------------------------------------------------------
// compactify (compactifyHashAllInOneKernel, DSC/CUDASceneRepHashSDF.cu:317-359)
//
// The reference scans all Ne entries (32 B each) every frame.  Here the
// 1-bit-per-bucket summary is scanned instead (Nb/8 bytes) and only non-empty
// buckets are opened; kept entries queue up in LDS and the workgroup appends
// them to the list with ONE atomic (compactify_group below).
// ---------------------------------------------------------------------------

// v_cvt_i32_f32 truncates, saturates and sends NaN to 0 by itself: f2i() without the v_trunc_f32 the compiler puts in front
VHD int cvt_rz(float v)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

constexpr uint32_t kIntegrateTile = 32;        // a block's screen footprint of up to 32 x 29 pixels is staged in LDS,
constexpr uint32_t kIntegrateTileRows = 29;    // rows 34 pixels apart (272 B: vertical neighbours fall into different banks):
constexpr uint32_t kIntegrateTileStride = 34;  // 31.6 KB per workgroup = 25 allocation units of 1280 B, five workgroups per compute
                                               // unit (30 rows are 26 units: per-wave time stamps showed four resident, the fifth waiting)


// What the fused integrate pass wants to know of a block before it touches a voxel, worked out once per block by the
// compactify pass (one lane per block there; in the integrate pass it would be ~160 instructions of every wave) and
// kept in the three spare words of the block's entry in the compactified list:
//   box: the block's screen footprint for the frame being integrated -- the hull of its eight corner voxels' pixels, one
//     pixel wider all round (a perspective projection maps the block into the hull of its corners when all of them lie
//     in front of the camera; the extra pixel covers rounding), clipped to the image, starting at an even pixel.  The
//     corners are projected with a hardware reciprocal: the box only decides WHERE a pixel is read from (the tile
//     staged in LDS or the frame itself), never which pixel;
//   VH_BOX_STAGED: the box is worth staging (in front of the camera, at most 32 x 29 pixels, even image width);
//   VH_BOX_CERTIFIED: the range certificate of integrate_block_certified holds at all eight corners (exact arithmetic,
//     the same expressions the voxels go through);
//   tag: which transform / camera / voxel size that was worked out for.  A pass that runs with another one works it
//     out again itself.
constexpr uint32_t VH_BOX_STAGED = 1u, VH_BOX_CERTIFIED = 2u;
struct BlockBox {
    uint32_t x0, y0, w, h, flags;
};
VHD uint32_t frame_tag(const VhHashParams& hp, const VhDepthCameraParams& cp)
{
    uint32_t t = 0x9E3779B9u;
#pragma unroll
    for (int i = 0; i < 12; i++) t = (t ^ __float_as_uint(hp.m_rigidTransformInverse[i])) * 0x01000193u + (t >> 15);
    const uint32_t more[7] = { __float_as_uint(hp.m_virtualVoxelSize), __float_as_uint(cp.fx), __float_as_uint(cp.fy), __float_as_uint(cp.mx),
                               __float_as_uint(cp.my), cp.m_imageWidth, cp.m_imageHeight };
#pragma unroll
    for (int i = 0; i < 7; i++) t = (t ^ more[i]) * 0x01000193u + (t >> 15);
    return (t & 0x7fffffffu) | 1u; // never 0: a zeroed entry carries no box; bit 31 is the mark of a rider_tag
}
// The tag of a list made and read within ONE launch (the pass over the voxels as a rider): the frame's number.  The reader
// takes an entry for this frame's when it carries this tag (CoIntegrate), so the tag must not repeat from one frame to the
// next as frame_tag does when the camera stands still, and must not collide with one by chance.
VHD uint32_t rider_tag(uint32_t frameNumber) { return 0x80000000u | frameNumber; }
struct BoxCorners { // what the eight corners add up to
    int bx0, bx1, by0, by1;
    float cz;
    bool fine;
};
VHD BoxCorners box_no_corner()
{
    BoxCorners a;
    a.bx0 = 0x7fffffff; a.bx1 = (int)0x80000000; a.by0 = 0x7fffffff; a.by1 = (int)0x80000000;
    a.cz = pinf();
    a.fine = true;
    return a;
}
VHD void box_add_corner(const VhHashParams& hp, const VhDepthCameraParams& cp, int ex, int ey, int ez, uint32_t c, BoxCorners& a)
{
    const I3 pc = mki3(ex * VH_SDF_BLOCK_SIZE + ((c & 1u) ? 7 : 0), ey * VH_SDF_BLOCK_SIZE + ((c & 2u) ? 7 : 0), ez * VH_SDF_BLOCK_SIZE + ((c & 4u) ? 7 : 0));
    const F3 pf = mat_mul_p(hp.m_rigidTransformInverse, vvp_to_world(hp.m_virtualVoxelSize, pc));
    const float rz = __builtin_amdgcn_rcpf(pf.z);
    const int cx = cvt_rz((pf.x * cp.fx * rz + cp.mx) + 0.5f), cy = cvt_rz((pf.y * cp.fy * rz + cp.my) + 0.5f); // (saturating; NaN -> 0)
    a.bx0 = min(a.bx0, cx); a.bx1 = max(a.bx1, cx);
    a.by0 = min(a.by0, cy); a.by1 = max(a.by1, cy);
    a.cz = fminf(a.cz, pf.z);
    a.fine = a.fine && pf.z >= 0x1p-20f && pf.z <= 0x1p20f && fabsf(pf.x * cp.fx) <= 0x1p60f && fabsf(pf.y * cp.fy) <= 0x1p60f;
}
VHD BlockBox box_of_corners(const VhDepthCameraParams& cp, const BoxCorners& a)
{
    const bool inFront = a.cz > 1e-3f;
    // (coordinates beyond +-2^30 would overflow the arithmetic below: such a block is simply not staged)
    const bool sane = a.bx0 > -(1 << 30) && a.bx1 < (1 << 30) && a.by0 > -(1 << 30) && a.by1 < (1 << 30);
    const int x0i = max(a.bx0 - 1, 0) & ~1, x1i = min(a.bx1 + 1, (int)cp.m_imageWidth - 1);
    const int y0i = max(a.by0 - 1, 0), y1i = min(a.by1 + 1, (int)cp.m_imageHeight - 1);
    const bool staged = inFront && sane && x0i <= x1i && y0i <= y1i && (x1i - x0i) < (int)kIntegrateTile && (y1i - y0i) < (int)kIntegrateTileRows &&
                        (cp.m_imageWidth & 1u) == 0u && cp.m_imageWidth <= 0xffffu && cp.m_imageHeight <= 0xffffu;
    BlockBox b;
    b.x0 = staged ? (uint32_t)x0i : 0u;
    b.y0 = staged ? (uint32_t)y0i : 0u;
    b.w = staged ? (uint32_t)(x1i - x0i + 1) : 0u;
    b.h = staged ? (uint32_t)(y1i - y0i + 1) : 0u;
    b.flags = (staged ? VH_BOX_STAGED : 0u) | (a.fine ? VH_BOX_CERTIFIED : 0u);
    return b;
}
VHD BlockBox block_box(const VhHashParams& hp, const VhDepthCameraParams& cp, int ex, int ey, int ez)
{
    BoxCorners a = box_no_corner();
#pragma unroll
    for (uint32_t c = 0; c < 8u; c++) box_add_corner(hp, cp, ex, ey, ez, c, a);
    return box_of_corners(cp, a);
}
// the same by eight neighbouring lanes, a corner each (every lane gets the box)
VHD BlockBox block_box_by_eight_lanes(const VhHashParams& hp, const VhDepthCameraParams& cp, int ex, int ey, int ez)
{
    BoxCorners a = box_no_corner();
    box_add_corner(hp, cp, ex, ey, ez, lane_id() & 7u, a);
    int fine = a.fine ? 1 : 0;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        a.bx0 = min(a.bx0, __shfl_xor(a.bx0, o)); a.bx1 = max(a.bx1, __shfl_xor(a.bx1, o));
        a.by0 = min(a.by0, __shfl_xor(a.by0, o)); a.by1 = max(a.by1, __shfl_xor(a.by1, o));
        a.cz = fminf(a.cz, __shfl_xor(a.cz, o));
        fine &= __shfl_xor(fine, o);
    }
    a.fine = fine != 0;
    return box_of_corners(cp, a);
}
// the spare words of a compactified entry: {x0 | y0 << 16, w | h << 8 | flags << 16, tag}
VHD uint4 pack_box(uint32_t offset, const BlockBox& b, uint32_t tag) { return make_uint4(offset, b.x0 | (b.y0 << 16), b.w | (b.h << 8) | (b.flags << 16), tag); }

// A workgroup takes 256 words of 32 occupancy bits.  It first gathers its non-empty buckets (a lane per word), then reads
// their slots a lane per slot, four slots in flight per lane: the loads do not depend on each other, so the workgroup's life
// is a few trips to memory -- with a lane per word and a loop over the word's bits it was one trip per set bit of the fullest
// word of the wave, 5-6 us where the launch's other riders need 2 (and the pass over the voxels, when it rides in the same
// launch, waits for the list).  What the workgroup keeps is queued in LDS and appended to the list with ONE atomic per
// workgroup (agent-scope atomics on one address are served one after the other at the memory side of the eight L2s, ~12 ns
// each: with one per wave and slot the dense scene's 8 700 blocks cost 40 us).  The queued entries then get their boxes, one
// lane each.  Entries beyond the queue's capacity take the list directly.
constexpr uint32_t kCompactQueue = 768;
struct CompactShared {
    int4 q[kCompactQueue];
    uint32_t off[kCompactQueue];
    uint32_t n, base, nBuckets;
    uint16_t buckets[256 * 32]; // the non-empty buckets, relative to the workgroup's first
};
// An entry of the compactified list, written and read within ONE launch (the pass over the voxels as a rider of the launch
// that makes its list, k_compute_normals): the L2s of the eight XCDs are not coherent with each other, so such an entry is
// written through and read at the agent's coherence point (relaxed agent-scope atomics: the sc1 forms of the instructions)
// instead of the writers writing their whole L2 back and the readers dropping theirs (a release / acquire pair at agent
// scope costs exactly that, buffer_wbl2 / buffer_inv: measured, +30 us a frame).
template <bool COHERENT> VHD void list_store(VhHashEntry* o, const int4 q, const uint4 box)
{
    if (COHERENT) {
        uint64_t* const w = reinterpret_cast<uint64_t*>(o);
        __hip_atomic_store(w + 0, (uint64_t)(uint32_t)q.x | ((uint64_t)(uint32_t)q.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(w + 1, (uint64_t)(uint32_t)q.z | ((uint64_t)(uint32_t)q.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(w + 2, (uint64_t)box.x | ((uint64_t)box.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the word with the tag goes last, when the others have arrived: a reader that finds the tag finds the entry
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(w + 3, (uint64_t)box.z | ((uint64_t)box.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *(reinterpret_cast<uint4*>(o) + 1) = box;
        store_quad(o, q);
    }
}
template <bool COHERENT> VHD int4 list_quad(const VhHashEntry* e)
{
    if (COHERENT) {
        const uint64_t* const w = reinterpret_cast<const uint64_t*>(e);
        const uint64_t a = __hip_atomic_load(w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_int4((int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)(uint32_t)b, (int)(uint32_t)(b >> 32));
    }
    return load_quad(e);
}
template <bool COHERENT> VHD uint4 list_box(const VhHashEntry* e)
{
    if (COHERENT) {
        const uint64_t* const w = reinterpret_cast<const uint64_t*>(e);
        const uint64_t a = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(w + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    }
    return *(reinterpret_cast<const uint4*>(e) + 1);
}

template <bool COHERENT = false>
__device__ void compactify_group(const VhHashData& hd, const VhHashParams& hp, const VhDepthCameraParams& cp, uint32_t wordIdx, CompactShared& sh, const uint32_t riderTag = 0u)
{
    const uint32_t nWords = (hp.m_hashNumBuckets + 31) / 32;
    const uint32_t tag = COHERENT ? riderTag : frame_tag(hp, cp);
    const uint32_t lane = lane_id();
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 42 // measurement build: the phases of a compactify workgroup (tools/riders_stamps.py)
    uint32_t phase[6];
    phase[0] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#define VH_COMPACT_PHASE(I) phase[I] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#define VH_COMPACT_PHASES_OUT { __syncthreads(); if (threadIdx.x == 0) { uint4* const at = reinterpret_cast<uint4*>(hd.d_hashCompactified) + (hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE) / 2u + 8192u + 2u * (wordIdx / 256u); \
        at[0] = make_uint4(phase[0], phase[1], phase[2], 0x5743u); at[1] = make_uint4(phase[3], phase[4], (uint32_t)__builtin_amdgcn_s_memrealtime(), sh.n | (sh.nBuckets << 16)); } }
#else
#define VH_COMPACT_PHASE(I)
#define VH_COMPACT_PHASES_OUT
#endif
    if (threadIdx.x == 0) { sh.n = 0u; sh.nBuckets = 0u; }
    __syncthreads();
    // the non-empty buckets, gathered (their order does not matter)
    uint32_t bits = (wordIdx < nWords) ? hd.d_bucketBits[wordIdx] : 0u;
    {
        const uint32_t mine = (uint32_t)__popc(bits);
        uint32_t incl = mine;
#pragma unroll
        for (uint32_t d = 1; d < kWave; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += t;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, kWave - 1);
        uint32_t at = 0;
        if (lane == 0 && total != 0u) at = atomicAdd(&sh.nBuckets, total);
        at = (uint32_t)__shfl((int)at, 0) + incl - mine;
        while (bits != 0u) {
            sh.buckets[at++] = (uint16_t)(threadIdx.x * 32u + (uint32_t)(__ffs((int)bits) - 1));
            bits &= bits - 1u;
        }
    }
    __syncthreads();
    VH_COMPACT_PHASE(1)
    const uint32_t nSlots = sh.nBuckets * VH_HASH_BUCKET_SIZE;
    const uint64_t firstBucket = (uint64_t)(wordIdx - threadIdx.x) * 32u;
    constexpr uint32_t kInFlight = 4; // (eight: 20 registers more for the launch's every wave, and no sooner done)
    for (uint32_t s0 = threadIdx.x; s0 < nSlots; s0 += 256u * kInFlight) { // (s0 < nSlots for all or none of a wave's lanes but in its last trip)
        int4 qs[kInFlight];
        uint32_t offs[kInFlight];
#pragma unroll
        for (uint32_t j = 0; j < kInFlight; j++) {
            const uint32_t slot = s0 + 256u * j;
            qs[j] = make_int4(0, 0, 0, VH_FREE_ENTRY);
            offs[j] = 0;
            if (slot < nSlots) {
                const VhHashEntry* e = &hd.d_hash[(firstBucket + sh.buckets[slot / VH_HASH_BUCKET_SIZE]) * VH_HASH_BUCKET_SIZE + slot % VH_HASH_BUCKET_SIZE];
                qs[j] = load_quad(e);
                offs[j] = e->offset;
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < kInFlight; j++) {
            const int4 q = qs[j];
            const bool keep = q.w != VH_FREE_ENTRY && block_in_frustum(hp, cp, mki3(q.x, q.y, q.z));
            const uint64_t kept = __ballot(keep); // one LDS atomic per wave
            if (kept == 0ull) continue;
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&sh.n, (uint32_t)__popcll(kept));
            at = (uint32_t)__shfl((int)at, 0) + (uint32_t)__popcll(kept & lanemask_lt());
            if (keep) {
                if (at < kCompactQueue) {
                    sh.q[at] = q;
                    sh.off[at] = offs[j];
                } else { // the queue is full: straight to the list
                    VhHashEntry* o = &hd.d_hashCompactified[(uint32_t)atomicAdd(hd.d_hashCompactifiedCounter, 1)];
                    list_store<COHERENT>(o, q, pack_box(offs[j], block_box(hp, cp, q.x, q.y, q.z), tag));
                }
            }
        }
    }
    __syncthreads();
    VH_COMPACT_PHASE(2)
    const uint32_t n = min(sh.n, kCompactQueue);
    if (n == 0u) return; // (the same for every thread)
    // the workgroup's place in the list: asked for now, needed when the first boxes are done (a trip to memory)
    uint32_t place = 0;
    if (threadIdx.x == 0) place = (uint32_t)atomicAdd(hd.d_hashCompactifiedCounter, (int)n);
    if (COHERENT) {
        // (the pass that reads this list in the same launch gives a workgroup to every block and never looks at a box: the
        // entries go out as soon as the place is known, the tag in their last word)
        if (threadIdx.x == 0) sh.base = place;
        __syncthreads();
        VH_COMPACT_PHASE(3)
        const uint32_t base = sh.base;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) list_store<true>(&hd.d_hashCompactified[base + i], sh.q[i], make_uint4(sh.off[i], 0u, 0u, tag));
        VH_COMPACT_PHASE(4)
        VH_COMPACT_PHASES_OUT
        return;
    }
    // the boxes, eight lanes per entry (a corner each: one lane per entry is ~300 instructions in a row, 1 us on a busy unit)
    constexpr uint32_t kBoxRounds = 2; // (rounds of 32 entries whose boxes wait in registers for the place; more entries: after it)
    uint4 boxes[kBoxRounds];
    const uint32_t mine = threadIdx.x >> 3;
#pragma unroll
    for (uint32_t r = 0; r < kBoxRounds; r++) {
        const uint32_t i = min(r * 32u + mine, n - 1u); // (whole groups of eight lanes compute or idle; an idle group repeats the last entry)
        if (r * 32u < n) {
            const int4 q = sh.q[i];
            boxes[r] = pack_box(sh.off[i], block_box_by_eight_lanes(hp, cp, q.x, q.y, q.z), tag);
        }
    }
    if (threadIdx.x == 0) sh.base = place;
    __syncthreads();
    VH_COMPACT_PHASE(3)
    const uint32_t base = sh.base;
#pragma unroll
    for (uint32_t r = 0; r < kBoxRounds; r++) {
        const uint32_t i = r * 32u + mine;
        if (i < n && (threadIdx.x & 7u) == 0u) list_store<COHERENT>(&hd.d_hashCompactified[base + i], sh.q[i], boxes[r]);
    }
    for (uint32_t r = kBoxRounds; r * 32u < n; r++) {
        const uint32_t i = r * 32u + mine;
        const int4 q = sh.q[min(i, n - 1u)];
        const BlockBox b = block_box_by_eight_lanes(hp, cp, q.x, q.y, q.z);
        if (i < n && (threadIdx.x & 7u) == 0u) list_store<COHERENT>(&hd.d_hashCompactified[base + i], q, pack_box(sh.off[i], b, tag));
    }
    VH_COMPACT_PHASE(4)
    VH_COMPACT_PHASES_OUT
}

__global__ __launch_bounds__(256) void k_compactify(VhHashData hd, VhHashParams hp, VhDepthCameraParams cp)
{
    __shared__ CompactShared sh;
    compactify_group(hd, hp, cp, blockIdx.x * blockDim.x + threadIdx.x, sh);
}

// ---------------------------------------------------------------------------
// integrate / starve / garbage collection
// (integrateDepthMapKernel :412-492, starveVoxelsKernel :512-521,
//  garbageCollectIdentifyKernel :543-590, garbageCollectFreeKernel :608-628)
//
// One 256-thread workgroup per SDF block, two x-adjacent voxels (16 B) per
// lane: a block is 4 waves x 1 KiB fully coalesced.  The fused kernel keeps the
// block in registers across integrate -> starve -> identify -> free, so every
// voxel is read once and written once per frame; min/max are reduced with
// wave64 shuffles and a 4-entry LDS combine.
// ---------------------------------------------------------------------------

VHD Vox integrate_voxel(const VhHashParams& hp, const VhDepthCameraParams& cp, const VhDepthCameraData& cam, I3 pi, Vox stored)
{
    F3 pf = mat_mul_p(hp.m_rigidTransformInverse, vvp_to_world(hp.m_virtualVoxelSize, pi));
    // cameraToKinectScreenInt, DSC/DepthCameraUtil.h:74-85 -> uint2
    const uint32_t sx = (uint32_t)f2i((pf.x * cp.fx / pf.z + cp.mx) + 0.5f);
    const uint32_t sy = (uint32_t)f2i((pf.y * cp.fy / pf.z + cp.my) + 0.5f);
    if (sx < cp.m_imageWidth && sy < cp.m_imageHeight) {
        const uint32_t pix = sy * cp.m_imageWidth + sx;
        const float depth = cam.d_depthData[pix];
        float cr = minf(), cg = minf(), cb = minf();
        if (cam.d_colorData) {
            const float4 c = reinterpret_cast<const float4*>(cam.d_colorData)[pix];
            cr = c.x; cg = c.y; cb = c.z;
        }
        if (cr != minf() && depth != minf()) {
            if (depth < hp.m_maxIntegrationDistance) {
                const float depthZeroOne = cam_to_proj_z(cp, depth);
                float sdf = depth - pf.z;
                const float truncation = get_truncation(hp, depth);
                if (sdf > -truncation) {
                    if (sdf >= 0.0f) sdf = fminf(truncation, sdf);
                    else sdf = fmaxf(-truncation, sdf);
                    const float weightUpdate = fmaxf((float)hp.m_integrationWeightSample * 1.5f * (1.0f - depthZeroOne), 1.0f);
                    Vox curr;
                    curr.sdf = sdf;
                    if (cam.d_colorData) curr.cw = pack_cw(f2uc(255.0f * cr), f2uc(255.0f * cg), f2uc(255.0f * cb), f2uc(weightUpdate));
                    else curr.cw = pack_cw(0, 255, 0, f2uc(weightUpdate));
                    return combine_voxel(hp, stored, curr);
                }
            }
        }
    }
    return stored;
}

VHD Vox starve_voxel(Vox v)
{
    uint32_t w = v.weight();
    w = (w > 0u) ? w - 1u : 0u;
    v.cw = (v.cw & 0x00ffffffu) | (w << 24);
    return v;
}

// block-wide min(|sdf| of weighted voxels) / max(weight): wave shuffle tree,
// then one LDS slot per wave.  Every lane returns the block result.
VHD void block_min_max(float& minSdf, uint32_t& maxW, float* sMin, uint32_t* sMax)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        minSdf = fminf(minSdf, __shfl_xor(minSdf, o));
        maxW = max(maxW, (uint32_t)__shfl_xor((int)maxW, o));
    }
    const uint32_t wave = threadIdx.x / kWave;
    if (lane_id() == 0) { sMin[wave] = minSdf; sMax[wave] = maxW; }
    __syncthreads();
    float m = sMin[0];
    uint32_t w = sMax[0];
    for (uint32_t i = 1; i < blockDim.x / kWave; i++) { m = fminf(m, sMin[i]); w = max(w, sMax[i]); }
    minSdf = m; maxW = w;
}

VHD float gc_key(Vox v) { return (v.weight() == 0u) ? pinf() : fabsf(v.sdf); }

template <bool FUSED>
__global__ __launch_bounds__(256) void k_integrate(VhHashData hd, VhHashParams hp, VhDepthCameraData cam,
                                                   VhDepthCameraParams cp, uint32_t flags, int32_t lockToken, uint32_t* countMirror)
{
    __shared__ float sMin[4];
    __shared__ uint32_t sMax[4];
    __shared__ int sFreed;

    const uint32_t count = FUSED ? (uint32_t)hd.d_hashCompactifiedCounter[0] : hp.m_numOccupiedBlocks;
    // host-visible copy of the block count (mapped pinned memory): replaces a per-frame device->host copy
    if (FUSED && countMirror && blockIdx.x == 0 && threadIdx.x == 0) *countMirror = count;
    const uint32_t t = threadIdx.x;
    // voxel pair (2t, 2t+1): x = (2t)%8 (+1), y = (2t%64)/8, z = 2t/64  (delinearizeVoxelIndex, DSC/VoxelUtilHashSDF.h:313-318)
    const int lx = (int)((2u * t) & 7u), ly = (int)(((2u * t) & 63u) >> 3), lz = (int)((2u * t) >> 6);

    for (uint32_t b = blockIdx.x; b < count; b += gridDim.x) {
        const int4 q = load_quad(&hd.d_hashCompactified[b]);
        const int ex = __builtin_amdgcn_readfirstlane(q.x), ey = __builtin_amdgcn_readfirstlane(q.y);
        const int ez = __builtin_amdgcn_readfirstlane(q.z), ptr = __builtin_amdgcn_readfirstlane(q.w);

        uint4* vp = reinterpret_cast<uint4*>(&hd.d_SDFBlocks[(uint32_t)ptr]) + t;
        const uint4 raw = *vp;
        Vox v0 = unpack_vox(make_uint2(raw.x, raw.y)), v1 = unpack_vox(make_uint2(raw.z, raw.w));

        const I3 p0 = mki3(ex * VH_SDF_BLOCK_SIZE + lx, ey * VH_SDF_BLOCK_SIZE + ly, ez * VH_SDF_BLOCK_SIZE + lz);
        v0 = integrate_voxel(hp, cp, cam, p0, v0);
        v1 = integrate_voxel(hp, cp, cam, mki3(p0.x + 1, p0.y, p0.z), v1);
        // Pin the four result words in VGPRs here.  Without this, hipcc (ROCm 7.2, gfx950, -O3) merges the
        // "not integrated" path of the second voxel with a register that the colour fetch has already
        // overwritten (found by the parity tests: sdf of skipped odd voxels came back as the colour's MINF).
        asm volatile("" : "+v"(v0.sdf), "+v"(v0.cw), "+v"(v1.sdf), "+v"(v1.cw));

        bool freed = false;
        if (FUSED && (flags & VH_FUSED_GC)) {
            if (flags & VH_FUSED_STARVE) { v0 = starve_voxel(v0); v1 = starve_voxel(v1); }
            float minSdf = fminf(gc_key(v0), gc_key(v1));
            uint32_t maxW = max(v0.weight(), v1.weight());
            __syncthreads(); // LDS slots of the previous block iteration are consumed
            block_min_max(minSdf, maxW, sMin, sMax);
            const float thr = get_truncation(hp, cp.m_sensorDepthWorldMax);
            const bool decide = (minSdf >= thr) || (maxW == 0u);
            if (t == 0) {
                hd.d_hashDecision[b] = decide ? 1 : 0;
                sFreed = (decide && delete_hash_entry_element(hd, hp, mki3(ex, ey, ez), lockToken)) ? 1 : 0;
            }
            __syncthreads();
            freed = sFreed != 0;
        }
        if (freed) {
            *vp = make_uint4(0u, 0u, 0u, 0u);
        } else {
            const uint2 a = pack_vox(v0), c = pack_vox(v1);
            *vp = make_uint4(a.x, a.y, c.x, c.y);
        }
    }
}

// integrate_voxel on a frame packed by alloc_tile / pack_pixel, in two steps: where the voxel projects to, and what
// the pixel found there makes of the stored voxel
VHD bool project_voxel(const VhHashParams& hp, const VhDepthCameraParams& cp, I3 pi, uint32_t& sx, uint32_t& sy, float& pz)
{
    F3 pf = mat_mul_p(hp.m_rigidTransformInverse, vvp_to_world(hp.m_virtualVoxelSize, pi));
    sx = (uint32_t)f2i((pf.x * cp.fx / pf.z + cp.mx) + 0.5f);
    sy = (uint32_t)f2i((pf.y * cp.fy / pf.z + cp.my) + 0.5f);
    pz = pf.z;
    return sx < cp.m_imageWidth && sy < cp.m_imageHeight;
}
VHD Vox apply_pixel(const VhHashParams& hp, uint2 px, float pz, Vox stored)
{
    // Without branches: every lane forms the blend, a select keeps or drops it (the lanes of a wave rarely agree, and
    // straight-line code lets the eight voxels of a lane overlap: measured 6 % on the dense scene).  The arithmetic of a
    // kept blend is integrateDepthMapKernel's (:447-472); a dropped one may be anything, NaN included.
    const float depth = __uint_as_float(px.x);
    float sdf = depth - pz;
    const float truncation = get_truncation(hp, depth);
    // a sample: valid depth within the integration distance, valid colour (weight byte, pack_pixel), not behind the band
    const bool use = ((px.y >> 24) != 0u) && (sdf > -truncation);
    const float lo = fmaxf(-truncation, sdf), hi = fminf(truncation, sdf);
    Vox curr;
    curr.sdf = (sdf >= 0.0f) ? hi : lo;
    curr.cw = px.y;
    const Vox c = combine_voxel(hp, stored, curr);
    Vox out;
    out.sdf = use ? c.sdf : stored.sdf;
    out.cw = use ? c.cw : stored.cw;
    return out;
}
VHD Vox integrate_voxel_packed(const VhHashParams& hp, const VhDepthCameraParams& cp, const uint2* packed, I3 pi, Vox stored)
{
    uint32_t sx, sy;
    float pz;
    if (project_voxel(hp, cp, pi, sx, sy, pz)) return apply_pixel(hp, packed[sy * cp.m_imageWidth + sx], pz, stored);
    return stored;
}

// The fused pass: integrate -> starve -> identify -> free with every voxel read once and written once.  The block count
// lives on the device (no host round trip), so the grid is fixed and the kernel picks its shape from the count:
//   * few blocks (count <= workgroups): one WORKGROUP per block, two x-adjacent voxels (16 B) per lane -- the frame is
//     a latency chain (entry -> voxels -> gather -> table edit), and four waves per block keep it short;
//   * many blocks: one WAVE per block (4 KB = four 16-byte loads per lane), at most 5120 waves (five per SIMD; the
//     active waves fill whole workgroups); wave w takes blocks w, w + 5120, ... -- dealt statically (a ticket counter
//     was measured: 165 us, agent-scope atomics on one address are served one after the other).  min |sdf| / max weight are a
//     wave reduction (no LDS, no barrier); lane 0 edits the table; the block's screen footprint is staged in LDS and
//     blocks whose corners pass a range certificate take the packed arithmetic of integrate_block_certified (below).
// Load j of lane l of a wave that holds a whole block: voxels 128 j + 2 l and + 1, i.e. x = 2l mod 8 (+1),
// y = (l / 4) mod 8, z = 2j + l / 32 (delinearizeVoxelIndex, DSC/VoxelUtilHashSDF.h:313-318).
template <bool PACKED>
VHD void integrate_pair(const VhHashParams& hp, const VhDepthCameraParams& cp, const VhDepthCameraData& cam, const uint2* packed,
                        uint32_t flags, I3 p0, uint4& raw, float& minSdf, uint32_t& maxW)
{
    Vox v0 = unpack_vox(make_uint2(raw.x, raw.y)), v1 = unpack_vox(make_uint2(raw.z, raw.w));
    if (PACKED) {
        v0 = integrate_voxel_packed(hp, cp, packed, p0, v0);
        v1 = integrate_voxel_packed(hp, cp, packed, mki3(p0.x + 1, p0.y, p0.z), v1);
    } else {
        v0 = integrate_voxel(hp, cp, cam, p0, v0);
        v1 = integrate_voxel(hp, cp, cam, mki3(p0.x + 1, p0.y, p0.z), v1);
    }
    // Pin the result words in VGPRs here.  Without this, hipcc (ROCm 7.2, gfx950, -O3) merges the "not integrated" path
    // of the second voxel with a register that the gather has already overwritten (found by the parity tests: sdf of
    // skipped odd voxels came back as the colour's MINF).
    asm volatile("" : "+v"(v0.sdf), "+v"(v0.cw), "+v"(v1.sdf), "+v"(v1.cw));
    if (flags & VH_FUSED_STARVE) { v0 = starve_voxel(v0); v1 = starve_voxel(v1); }
    minSdf = fminf(minSdf, fminf(gc_key(v0), gc_key(v1)));
    maxW = max(maxW, max(v0.weight(), v1.weight()));
    const uint2 a = pack_vox(v0), c = pack_vox(v1);
    raw = make_uint4(a.x, a.y, c.x, c.y);
}

// ---- the voxel pair of a lane as a two-vector: packed fp32 instructions (v_pk_mul/add/fma_f32: two IEEE operations per
// issue slot; each component is the scalar instruction's result)
typedef float f32x2 __attribute__((ext_vector_type(2)));
VHD f32x2 both(float v) { return (f32x2){ v, v }; }
VHD f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// The reciprocal the compiler's own expansion of an fp32 division forms when v_div_scale_f32 leaves its operands alone:
// v_rcp_f32 and one Newton step.
VHD f32x2 rcp_refined2(f32x2 d)
{
    const f32x2 y0 = (f32x2){ __builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y) };
    const f32x2 e = fma2(-d, y0, both(1.0f));
    return fma2(e, y0, y0);
}
// n / d with y = rcp_refined2(d): the rest of that expansion (product, two residual corrections), instruction for
// instruction, minus v_div_scale / v_div_fmas' scaling / v_div_fixup.  Those three only act when an operand is zero,
// infinite, NaN or denormal, when |n| < 2^-103, when |d| >= 2^126 or when the exponents of n and d are 96 or more apart
// (the ISA's v_div_scale_f32 cases); outside of that the result is the division's, bit for bit.  Callers guard the range.
VHD f32x2 div_refined2(f32x2 n, f32x2 d, f32x2 y)
{
    f32x2 q = n * y;
    f32x2 r = fma2(-d, q, n);
    q = fma2(r, y, q);
    r = fma2(-d, q, n);
    return fma2(r, y, q);
}

// One block by one wave, the hot shape of the dense case: integrate_voxel_packed for the eight voxels of a lane (four
// x-adjacent pairs) with ~60 instead of ~110 vector instructions per voxel.  Only for blocks the caller has certified:
//   * the block's footprint is staged in LDS (tile, x0, y0, w, h);
//   * at all eight CORNER voxels 2^-20 <= pf.z <= 2^20 and |pf.x fx|, |pf.y fy| <= 2^60.  Every voxel of the block then
//     satisfies the same: each float operation of the transform is monotone in each of its operands, so pf.x, pf.y,
//     pf.z and the two products, as computed, take their extremes over the block at corner voxels;
//   * m_truncation > 0, m_truncScale >= 0 (so that a pixel that integrates has truncation > 0: the reference's
//     two-sided clamp is then the median of {sdf, -truncation, truncation}).
// What changes against the plain code, none of it in the results:
//   * the two divisions by pf.z share one refined reciprocal (div_refined2; a numerator below 2^-60, where the
//     argument above stops, gives a quotient below 2^-38 by either route, which the following "+ mx, + 0.5, truncate"
//     cannot see);
//   * the blend's division by the weight sum (1..510) takes the same route, with the exact division for any numerator
//     outside [2^-100, 2^90) (never seen with voxels this library wrote; a crafted .hashgrid could hold one);
//   * the pixel comes from the staged tile by an LDS read; a voxel that projects outside the staged box (not seen: the
//     box is the hull of the corners' pixels and a margin) is gathered from the frame as before;
//   * everything is written for the pair of a lane, so that the compiler can issue packed fp32.
VHD void integrate_block_certified(const VhHashParams& hp, const VhDepthCameraParams& cp, const uint2* packed, const uint2* tile, uint32_t tileStride,
                                   uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, int ex, int ey, int ez, uint32_t lane, uint32_t flags,
                                   uint4 (&raw)[4], float& minSdf, uint32_t& maxW)
{
    // (The transform as values of this function's own: packed instructions want their uniform operands in vector registers,
    // and copies shared with the kernel's other code paths live from the top of the kernel to its end -- the compiler
    // parks them in scratch memory, a store per lane of every wave of the launch.)
    float vs = hp.m_virtualVoxelSize;
    float m[12];
#pragma unroll
    for (int i = 0; i < 12; i++) m[i] = hp.m_rigidTransformInverse[i];
    asm volatile("" : "+s"(vs), "+s"(m[0]), "+s"(m[1]), "+s"(m[2]), "+s"(m[3]), "+s"(m[4]), "+s"(m[5]), "+s"(m[6]), "+s"(m[7]), "+s"(m[8]), "+s"(m[9]),
                      "+s"(m[10]), "+s"(m[11]));
    const int lx = (int)((2u * lane) & 7u), ly = (int)((lane >> 2) & 7u), lz0 = (int)(lane >> 5);
    // vvp_to_world, mat_mul_p: ((m0 x + m1 y) + m2 z) + m3 per row; the first sum does not depend on z
    const f32x2 xw = (f32x2){ (float)(ex * VH_SDF_BLOCK_SIZE + lx), (float)(ex * VH_SDF_BLOCK_SIZE + lx + 1) } * vs;
    const float yw = (float)(ey * VH_SDF_BLOCK_SIZE + ly) * vs;
    f32x2 ab[3];
#pragma unroll
    for (int r = 0; r < 3; r++) ab[r] = m[4 * r] * xw + both(m[4 * r + 1] * yw);
    uint32_t maxCw = 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float zw = (float)(ez * VH_SDF_BLOCK_SIZE + 2 * j + lz0) * vs;
        f32x2 pf[3];
#pragma unroll
        for (int r = 0; r < 3; r++) pf[r] = (ab[r] + both(m[4 * r + 2] * zw)) + both(m[4 * r + 3] * 1.0f);
        // cameraToKinectScreenInt: int((pf.x fx / pf.z + mx) + 0.5f), likewise y
        const f32x2 y = rcp_refined2(pf[2]);
        const f32x2 fsx = (div_refined2(pf[0] * cp.fx, pf[2], y) + cp.mx) + 0.5f;
        const f32x2 fsy = (div_refined2(pf[1] * cp.fy, pf[2], y) + cp.my) + 0.5f;
        uint2 px[2]; // the pixel of each voxel; weight byte 0: nothing to integrate (pack_pixel), or no pixel at all
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t sx = (uint32_t)cvt_rz(fsx[k]), sy = (uint32_t)cvt_rz(fsy[k]);
            const uint32_t tx = sx - x0, ty = sy - y0; // unsigned: a pixel left of / above the box wraps to a huge number
            const bool inBox = (tx < w) & (ty < h);    // the box lies inside the image
            // (one ballot per comparison: the ballot of their conjunction goes through a register and back)
            const bool allInBox = (__builtin_amdgcn_ballot_w64(tx < w) & __builtin_amdgcn_ballot_w64(ty < h)) == ~0ull; // (every lane is active here)
            // (as one indivisible 8-byte read: left alone the compiler splits it in two)
            // (no clamp of the index: an LDS read beyond the workgroup's allocation returns zero, one inside it some other
            // pixel, and the weight byte of a voxel outside the box is cleared below either way)
            const unsigned long long t = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&tile[__umul24(ty, tileStride) + tx]),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            px[k] = make_uint2((uint32_t)t, inBox ? (uint32_t)(t >> 32) : 0u);
            if (__builtin_expect(!allInBox, 0)) {
                if (!inBox && sx < cp.m_imageWidth && sy < cp.m_imageHeight) px[k] = packed[sy * cp.m_imageWidth + sx];
            }
        }
        // integrateDepthMapKernel :447-472 and combineVoxel :229-250 (apply_pixel / combine_voxel above) for the pair
        const Vox s0 = unpack_vox(make_uint2(raw[j].x, raw[j].y)), s1 = unpack_vox(make_uint2(raw[j].z, raw[j].w));
        const f32x2 depth = (f32x2){ __uint_as_float(px[0].x), __uint_as_float(px[1].x) };
        const f32x2 sdf = depth - pf[2];
        const f32x2 truncation = both(hp.m_truncation) + hp.m_truncScale * depth;
        const bool use0 = ((px[0].y >> 24) != 0u) & (sdf.x > -truncation.x);
        const bool use1 = ((px[1].y >> 24) != 0u) & (sdf.y > -truncation.y);
        const f32x2 clamped = (f32x2){ __builtin_amdgcn_fmed3f(sdf.x, -truncation.x, truncation.x), __builtin_amdgcn_fmed3f(sdf.y, -truncation.y, truncation.y) };
        const f32x2 wOld = (f32x2){ (float)(s0.cw >> 24), (float)(s1.cw >> 24) }, wNew = (f32x2){ (float)(px[0].y >> 24), (float)(px[1].y >> 24) };
        const f32x2 num = (f32x2){ s0.sdf, s1.sdf } * wOld + clamped * wNew;
        const f32x2 den = wOld + wNew; // exact: integers up to 510
        f32x2 q = div_refined2(num, den, rcp_refined2(den));
        const bool odd0 = use0 & !((fabsf(num.x) >= 0x1p-100f) & (fabsf(num.x) < 0x1p90f));
        const bool odd1 = use1 & !((fabsf(num.y) >= 0x1p-100f) & (fabsf(num.y) < 0x1p90f));
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd0 | odd1) != 0ull, 0)) {
            if (odd0) q.x = num.x / den.x;
            if (odd1) q.y = num.y / den.y;
        }
        Vox v[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const Vox st = k ? s1 : s0;
            // colour: the byte-wise average rounded up (combine_voxel) is v_lerp_u8 with the rounding bit set; the weight
            // byte comes in by v_perm_b32 (the top byte of the average is not used)
            const uint32_t rgb = __builtin_amdgcn_lerp(st.cw, px[k].y, 0x01010101u);
            const uint32_t wSum = min(hp.m_integrationWeightMax, (st.cw >> 24) + (px[k].y >> 24));
            const bool use = k ? use1 : use0;
            v[k].sdf = use ? q[k] : st.sdf;
            v[k].cw = use ? __builtin_amdgcn_perm(wSum, rgb, 0x04020100u) : st.cw;
        }
        if (flags & VH_FUSED_STARVE) { v[0] = starve_voxel(v[0]); v[1] = starve_voxel(v[1]); }
        minSdf = fminf(minSdf, fminf(gc_key(v[0]), gc_key(v[1])));
        maxCw = max(maxCw, max(v[0].cw, v[1].cw)); // the weight is the top byte: the largest word has the largest weight
        const uint2 a = pack_vox(v[0]), c = pack_vox(v[1]);
        raw[j] = make_uint4(a.x, a.y, c.x, c.y);
    }
    maxW = max(maxW, maxCw >> 24);
}

constexpr uint32_t kIntegrateWavesMost = 5120; // waves that take blocks when there are many: five per SIMD
// The kernel's one argument.  The pass that frees a block (a handful of blocks per frame, one lane each) wants a dozen
// table pointers and sizes that nothing else in the kernel needs; as ordinary kernel arguments they are loaded at the top
// of the kernel and kept in scalar registers throughout -- or, as measured, parked in scratch memory by every wave of
// the launch (4.5 MB of stores per launch at cfg2).  cold_args() hands that pass the argument block as memory the
// compiler knows nothing about, so it loads what it needs where it needs it.
struct FusedArgs {
    VhHashData hd;
    VhHashParams hp;
    VhDepthCameraData cam;
    VhDepthCameraParams cp;
    uint32_t flags;
    int32_t lockToken;
    uint32_t* countMirror;
    uint32_t mirrorTag;
    const uint2* packed;
};
// A rider waits for a flag another rider of the launch raises (rider_done): VH_RIDER_DONE_COUNTERS copies 128 bytes apart, set
// to a number that only grows (compared modulo 2^32).  An exit every wave reaches: after ~1 s of polling the wait gives up
// (never seen; the caller says so in the status words and the grid drains whatever happens).  Every lane of the wave calls it.
VHD bool rider_wait(const uint32_t* flags, uint32_t expected)
{
    const uint32_t* const flag = flags + (blockIdx.x % VH_RIDER_DONE_COUNTERS) * 32u;
    for (uint32_t polls = 0; polls < (1u << 22); polls++) {
        const uint32_t v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (the same word for every lane: one request)
        if ((int32_t)(v - expected) >= 0) return true;
        __builtin_amdgcn_s_sleep(4);
    }
    return false;
}
template <uint32_t KERNARG_OFFSET> // where the kernel's argument block holds the FusedArgs (0: it is the kernel's only argument)
VHD const FusedArgs* cold_args()
{
    typedef const char __attribute__((address_space(4))) * KernargPtr;
    KernargPtr p = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr() + KERNARG_OFFSET;
    asm volatile("" : "+s"(p));
    return (const FusedArgs*)p;
}
template <uint32_t KERNARG_OFFSET>
VHD bool free_block_cold(int ex, int ey, int ez, uint32_t lane) // every lane of the wave calls it
{
    const FusedArgs* a = cold_args<KERNARG_OFFSET>();
    const VhHashData hd = a->hd;
    const VhHashParams hp = a->hp;
    // (As a rider of k_compute_normals the pass frees blocks while that launch's splat workgroups read the table.  That needs no
    // order: the delete never moves an entry -- it overwrites the freed one with one 16-byte store, clears counters and unlinks --
    // and the splat reads every slot once, so it lists every live block, and the freed one or not.  Either is what the splat made
    // BEFORE the pass, in a launch of its own, always was for the ray cast that uses it: a freed block that is still listed
    // has all-zero voxels by then (weight 0: no sample reads it), see CoSplat.)
    return delete_hash_entry_element_wave(hd, hp, mki3(ex, ey, ez), a->lockToken, lane);
}

// the fused pass's shared memory (a kernel that carries the pass as a rider overlays it with its other riders')
template <bool PACKED>
struct FusedShared {
    float sMin[4];
    uint32_t sMax[4];
    int sFreed;
    __attribute__((aligned(16))) uint2 sTile[PACKED ? 256 / kWave : 1][PACKED ? kIntegrateTileRows * kIntegrateTileStride : 2];
};

// One block by one workgroup, a voxel pair per thread: the pass's shape for up to 2048 blocks (integrate_fused_body), and the
// pass as a rider of k_compute_normals (pass_rider_group).  q: the block's entry, b: its place in the list.
template <bool PACKED, uint32_t KERNARG_OFFSET, class Shared> // Shared: a FusedShared (its sMin, sMax, sFreed)
VHD void integrate_block_by_workgroup(const FusedArgs& args, Shared& sh, const int4 q, const uint32_t b, const float thr)
{
    const VhHashData& hd = args.hd;
    const VhHashParams& hp = args.hp;
    const uint32_t flags = args.flags;
    const uint32_t lane = lane_id();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 43 // measurement build: every workgroup's life in this shape, read by tools/integrate_stamps.py
    const uint32_t stampA = (uint32_t)__builtin_amdgcn_s_memrealtime();
    uint32_t stampB = 0u, stampFreed = 0u;
#endif
    const uint32_t t = threadIdx.x;
    const int ex = __builtin_amdgcn_readfirstlane(q.x), ey = __builtin_amdgcn_readfirstlane(q.y);
    const int ez = __builtin_amdgcn_readfirstlane(q.z), ptr = __builtin_amdgcn_readfirstlane(q.w);
    uint4* vp = reinterpret_cast<uint4*>(&hd.d_SDFBlocks[(uint32_t)ptr]) + t;
    uint4 raw = *vp;
    // voxel pair (2t, 2t+1): x = (2t)%8 (+1), y = (2t%64)/8, z = 2t/64
    const I3 p0 = mki3(ex * VH_SDF_BLOCK_SIZE + (int)((2u * t) & 7u), ey * VH_SDF_BLOCK_SIZE + (int)(((2u * t) & 63u) >> 3), ez * VH_SDF_BLOCK_SIZE + (int)((2u * t) >> 6));
    float minSdf = pinf();
    uint32_t maxW = 0u;
    integrate_pair<PACKED>(hp, args.cp, args.cam, args.packed, flags, p0, raw, minSdf, maxW);
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 43
    asm volatile("" : "+v"(raw.x));
    stampB = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
    if (flags & VH_FUSED_GC) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            minSdf = fminf(minSdf, __shfl_xor(minSdf, o));
            maxW = max(maxW, (uint32_t)__shfl_xor((int)maxW, o));
        }
        if (lane == 0) { sh.sMin[wave] = minSdf; sh.sMax[wave] = maxW; }
        __syncthreads();
        minSdf = fminf(fminf(sh.sMin[0], sh.sMin[1]), fminf(sh.sMin[2], sh.sMin[3]));
        maxW = max(max(sh.sMax[0], sh.sMax[1]), max(sh.sMax[2], sh.sMax[3]));
        const bool decide = (minSdf >= thr) || (maxW == 0u); // the same in every thread of the workgroup
        if (t == 0) hd.d_hashDecision[b] = decide ? 1 : 0;
        if (decide) {
            if (wave == 0u) {
                const bool f = free_block_cold<KERNARG_OFFSET>(ex, ey, ez, lane);
                if (lane == 0u) sh.sFreed = f ? 1 : 0;
            }
            __syncthreads();
            if (sh.sFreed != 0) raw = make_uint4(0u, 0u, 0u, 0u);
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 43
            stampFreed = 1u + (uint32_t)sh.sFreed;
#endif
        }
    }
    *vp = raw;
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 43
    if (t == 0) reinterpret_cast<uint4*>(hd.d_hashCompactified)[(hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE) / 2u + b] = make_uint4(stampA, stampB, (uint32_t)__builtin_amdgcn_s_memrealtime(), 0x57430000u | stampFreed);
#endif
}

// The pass in a launch of its own (k_integrate_fused), as a function of (workgroup index, number of workgroups).  As a rider of
// k_compute_normals it is pass_rider_group (below), which shares the workgroup-per-block code above.
template <bool PACKED, uint32_t KERNARG_OFFSET>
VHD void integrate_fused_body(const FusedArgs& args, const uint32_t groupIdx, const uint32_t numGroups, FusedShared<PACKED>& sh)
{
    const VhHashData& hd = args.hd;
    const VhHashParams& hp = args.hp;
    const VhDepthCameraData& cam = args.cam;
    const VhDepthCameraParams& cp = args.cp;
    const uint32_t flags = args.flags;
    uint32_t* const countMirror = args.countMirror;
    const uint32_t mirrorTag = args.mirrorTag;
    const uint2* const packed = args.packed;
    float (&sMin)[4] = sh.sMin;
    uint32_t (&sMax)[4] = sh.sMax;
    int& sFreed = sh.sFreed;
    auto& sTile = sh.sTile;
    const uint32_t lane = lane_id();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const uint32_t nEntries = hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
    // the count and the entry this workgroup / wave would start with in one trip (the list is Ne entries long: reading
    // beyond the count is reading stale entries, which are not used)
    const uint32_t count = (uint32_t)hd.d_hashCompactifiedCounter[0];
    // first block of this wave when waves take blocks: the active waves fill whole workgroups (the others leave at
    // once and free their slot: a workgroup's LDS and registers are held until its last wave is done)
    const uint32_t wFirst = groupIdx * (256u / kWave) + wave;
    int4 qg = make_int4(0, 0, 0, 0), qw = make_int4(0, 0, 0, 0);
    uint4 boxw = make_uint4(0u, 0u, 0u, 0u); // the entry's second half: {offset, box, box, tag} (compactify_group)
    if (groupIdx < nEntries) qg = load_quad(&hd.d_hashCompactified[groupIdx]);
    if (wFirst < nEntries) {
        qw = load_quad(&hd.d_hashCompactified[wFirst]);
        if (PACKED) boxw = list_box<false>(&hd.d_hashCompactified[wFirst]);
    }
    // host-visible copy of the block count and the caller's tag (mapped pinned memory): replaces a per-frame
    // device->host copy, and lets the host see how far the device has come
    if (countMirror && groupIdx == 0 && threadIdx.x == 0) *reinterpret_cast<uint2*>(countMirror) = make_uint2(count, mirrorTag);
    const float thr = get_truncation(hp, cp.m_sensorDepthWorldMax);

    if (count <= numGroups) {
        // ---- one workgroup per block
        if (groupIdx < count) integrate_block_by_workgroup<PACKED, KERNARG_OFFSET>(args, sh, qg, groupIdx, thr);
        return;
    }

    // ---- one wave per block: the first min(count, 5120) waves (five per SIMD; the hardware deals workgroups to the
    // compute units evenly) take blocks w, w + 5120, ...: no SIMD gets more than a block or two above the mean.
    // Measured and dropped: ceil(count / rounds) waves with `rounds` blocks each (SIMDs with five waves finished 3 us
    // after those with four: 27 us for 8 600 blocks), and a ticket counter (agent-scope atomics on one address are
    // served at the memory side of the eight L2s, ~12 ns each: 165 us).
    const uint32_t nActive = min(min(count, kIntegrateWavesMost), numGroups * (256u / kWave));
    if (wFirst >= nActive) return;
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 9 // measurement build: shader clock over one wave's lifetime
    const uint64_t stampC0 = __builtin_amdgcn_s_memtime(), stampR0 = __builtin_amdgcn_s_memrealtime();
    uint32_t stamps[6] = { 0u, 0u, 0u, 0u, 0u, 0u }, stampN = 0u; // per block: staged, voxels arrived, computed (two blocks)
#endif
    uint32_t b = wFirst, round = 0u;
    // this wave's place in the later rounds: the SIMD it sits on (unit g mod 256, wave of the workgroup) bit-reversed, then
    // the workgroup's turn on that unit -- a permutation of 0 .. 5119 whose every prefix is spread evenly over the SIMDs
    // (used as is when all 5 120 waves are active; with fewer waves, which happens below 5 120 blocks, there is one round)
    const uint32_t simdOfMachine = ((groupIdx & 255u) << 2) | wave; // (as a rider the workgroups sit elsewhere: the order is then only a permutation)
    const uint32_t scattered = nActive == kIntegrateWavesMost ? (__brev(simdOfMachine) >> 22) + (groupIdx >> 8) * 1024u : wFirst;
    int4 q = qw;
    uint4 qbox = boxw;
    uint2* tile = sTile[wave];
    const uint32_t tagNow = frame_tag(hp, cp);
    for (;;) {
        // What depends on the lane alone is formed again for every block (a dozen cheap instructions): hoisted out of the
        // loop it is a dozen registers that live through everything, and the compiler parks them in scratch memory.
        uint32_t lane = lane_id();
        asm volatile("" : "+v"(lane));
        const int lx = (int)((2u * lane) & 7u), ly = (int)((lane >> 2) & 7u), lz0 = (int)(lane >> 5);
        const int ex = __builtin_amdgcn_readfirstlane(q.x), ey = __builtin_amdgcn_readfirstlane(q.y);
        const int ez = __builtin_amdgcn_readfirstlane(q.z), ptr = __builtin_amdgcn_readfirstlane(q.w);
        uint4* vp = reinterpret_cast<uint4*>(&hd.d_SDFBlocks[(uint32_t)ptr]) + lane;
        uint4 raw[4];
#pragma unroll
        for (int j = 0; j < 4; j++) raw[j] = vp[j * kWave];
        const uint32_t boxA = (uint32_t)__builtin_amdgcn_readfirstlane((int)qbox.y), boxB = (uint32_t)__builtin_amdgcn_readfirstlane((int)qbox.z);
        const uint32_t boxTag = (uint32_t)__builtin_amdgcn_readfirstlane((int)qbox.w);
        // The wave's next block.  Rounds of nActive blocks; within a round the blocks go to the waves in an order that
        // scatters a partial last round over the compute units (workgroup g sits on unit g mod 256: with the plain order
        // w + round * nActive the first units get all of the last round's blocks -- 36 blocks against 32 at 8 600).
        round += 1u;
        const uint32_t bNext = round * nActive + scattered;
        const bool hasNext = scattered < nActive && bNext < count;
        if (hasNext) { // its entry (both halves) behind this block's voxels
            q = load_quad(&hd.d_hashCompactified[bNext]);
            if (PACKED) qbox = list_box<false>(&hd.d_hashCompactified[bNext]);
        }
        // (Requesting the next block's voxels here as well was measured: slower.  The waves do not wait for the stream --
        // the kernel is bound by instruction issue, ~1000 vector instructions per block at 4 cycles each.)

        float minSdf = pinf();
        uint32_t maxW = 0u;
        bool wrote = false; // the plain code has put the block back itself
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 1 // measurement build: the voxel stream alone (read, write back)
        if (false) {
#else
        if (PACKED) {
#endif
            // A voxel's gather from the frame is one cache access per LANE -- 64 tag look-ups per wave instruction, the
            // bound of this shape before staging (1 cm voxels: 13 us of a 33 us launch).  So the block's screen
            // footprint is staged in LDS first, four image rows per load instruction (whole lines), while the voxel
            // loads are in flight.  The box comes with the block's entry (block_box, worked out by the compactify pass
            // for this very transform: the tag says so; a list made for another one goes through the plain code, every
            // voxel gathered from the frame).  A voxel that projects outside the staged box all the same is gathered
            // from the frame itself: the box decides where a pixel is read, never which.
            BlockBox bb;
            bb.x0 = boxA & 0xffffu; bb.y0 = boxA >> 16;
            bb.w = boxB & 0xffu; bb.h = (boxB >> 8) & 0xffu;
            bb.flags = boxTag == tagNow ? boxB >> 16 : 0u; // a box made for another transform: neither staged nor certified
            // (a box that does not fit this image -- it cannot come from block_box for this frame -- is not used)
            const bool staged = (bb.flags & VH_BOX_STAGED) != 0u && bb.w <= kIntegrateTile && bb.h <= kIntegrateTileRows && (bb.x0 & 1u) == 0u &&
                                bb.x0 + bb.w <= cp.m_imageWidth && bb.y0 + bb.h <= cp.m_imageHeight && (cp.m_imageWidth & 1u) == 0u;
            const uint32_t x0 = bb.x0, y0 = bb.y0;
            const uint32_t w = staged ? bb.w : 0u, h = staged ? bb.h : 0u;
            if (staged) {
                // lane -> (row r + lane / 16, pixels 2 (lane % 16) and + 1): four rows per load instruction.  An even image
                // width and an even x0 keep a pixel pair inside its row.
                const uint32_t col2 = (lane & 15u) * 2u, rowInQuad = lane >> 4;
                const uint4* src = reinterpret_cast<const uint4*>(packed + (size_t)(y0 + rowInQuad) * cp.m_imageWidth + x0 + col2);
                uint4* dst = reinterpret_cast<uint4*>(tile + rowInQuad * kIntegrateTileStride + col2);
                __builtin_amdgcn_wave_barrier(); // the previous block's reads of the tile are done (they fed its stores)
#pragma unroll 4
                for (uint32_t r = 0; r < h; r += 4u) {
                    if (r + rowInQuad < h && col2 < w) dst[(size_t)r * (kIntegrateTileStride / 2u)] = src[(size_t)r * (cp.m_imageWidth / 2u)];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#if !defined(VH_INTEGRATE_PLAIN) // (measurement / test builds: every block through the plain code)
            const bool certified = staged && (bb.flags & VH_BOX_CERTIFIED) != 0u && hp.m_truncation > 0.0f && hp.m_truncScale >= 0.0f;
#else
            const bool certified = false;
#endif
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 9
            if (stampN < 6u) stamps[stampN++] = (uint32_t)__builtin_amdgcn_s_memrealtime(); // staged
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (stampN < 6u) stamps[stampN++] = (uint32_t)__builtin_amdgcn_s_memrealtime(); // voxels here
#endif
            if (__builtin_amdgcn_readfirstlane(certified ? 1 : 0) != 0) {
                integrate_block_certified(hp, cp, packed, tile, kIntegrateTileStride, x0, y0, w, h, ex, ey, ez, lane, flags, raw, minSdf, maxW);
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 9
                asm volatile("" : "+v"(raw[3].x), "+v"(raw[0].x));
                if (stampN < 6u) stamps[stampN++] = (uint32_t)__builtin_amdgcn_s_memrealtime(); // computed
#endif
            } else {
                // The plain code, for the blocks without a certificate or a staged footprint (near the camera, behind it,
                // on the image's edge): a compact loop that takes its voxels from memory again and puts them back itself.
                // Unrolled and on the registers of the loads above it costs the whole kernel 30 registers and its
                // fifth wave per SIMD (measured: 99 against 65); the voxels are in the cache.
                wrote = true;
#pragma unroll 1
                for (int j = 0; j < 4; j++) {
                    const uint4 r = vp[j * kWave];
                    Vox v[2] = { unpack_vox(make_uint2(r.x, r.y)), unpack_vox(make_uint2(r.z, r.w)) };
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        uint32_t sx, sy;
                        float pz;
                        if (project_voxel(hp, cp, mki3(ex * VH_SDF_BLOCK_SIZE + lx + k, ey * VH_SDF_BLOCK_SIZE + ly, ez * VH_SDF_BLOCK_SIZE + 2 * j + lz0), sx, sy, pz)) {
                            const uint32_t tx = sx - x0, ty = sy - y0; // unsigned: a pixel left of / above the box wraps to a huge number
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 2 // measurement build: projection only
                            v[k].cw ^= (tx + ty + __float_as_uint(pz)) & 1u;
#elif defined(VH_KNOCKOUT) && VH_KNOCKOUT == 3 // measurement build: projection + the pixel, no blend
                            const uint2 px = (tx < w && ty < h) ? tile[ty * kIntegrateTileStride + tx] : packed[sy * cp.m_imageWidth + sx];
                            v[k].cw ^= (px.x + px.y + __float_as_uint(pz)) & 1u;
#else
                            const uint2 px = (tx < w && ty < h) ? tile[ty * kIntegrateTileStride + tx] : packed[sy * cp.m_imageWidth + sx];
                            v[k] = apply_pixel(hp, px, pz, v[k]);
#endif
                        }
                    }
                    asm volatile("" : "+v"(v[0].sdf), "+v"(v[0].cw), "+v"(v[1].sdf), "+v"(v[1].cw)); // see integrate_pair
                    if (flags & VH_FUSED_STARVE) { v[0] = starve_voxel(v[0]); v[1] = starve_voxel(v[1]); }
                    minSdf = fminf(minSdf, fminf(gc_key(v[0]), gc_key(v[1])));
                    maxW = max(maxW, max(v[0].weight(), v[1].weight()));
                    const uint2 a = pack_vox(v[0]), c = pack_vox(v[1]);
                    vp[j * kWave] = make_uint4(a.x, a.y, c.x, c.y);
                }
            }
        } else {
#if !(defined(VH_KNOCKOUT) && VH_KNOCKOUT == 1)
#pragma unroll
            for (int j = 0; j < 4; j++)
                integrate_pair<PACKED>(hp, cp, cam, packed, flags, mki3(ex * VH_SDF_BLOCK_SIZE + lx, ey * VH_SDF_BLOCK_SIZE + ly, ez * VH_SDF_BLOCK_SIZE + 2 * j + lz0),
                                       raw[j], minSdf, maxW);
#endif
        }
        bool freed = false;
        if (flags & VH_FUSED_GC) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                minSdf = fminf(minSdf, __shfl_xor(minSdf, o));
                maxW = max(maxW, (uint32_t)__shfl_xor((int)maxW, o));
            }
            const bool decide = (minSdf >= thr) || (maxW == 0u);
            if (lane == 0) hd.d_hashDecision[b] = decide ? 1 : 0;
            // (the same decision in every lane: the reduction left every lane with the block's minimum and maximum)
            if (__builtin_amdgcn_readfirstlane(decide ? 1 : 0) != 0) freed = free_block_cold<KERNARG_OFFSET>(ex, ey, ez, lane);
        }
        if (freed) {
#pragma unroll
            for (int j = 0; j < 4; j++) vp[j * kWave] = make_uint4(0u, 0u, 0u, 0u);
        } else if (!wrote) {
#pragma unroll
            for (int j = 0; j < 4; j++) vp[j * kWave] = raw[j];
        }
        if (!hasNext) break;
        b = bNext;
    }
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 9
    if (wFirst == 0u && lane == 0u) {
        const uint64_t c = __builtin_amdgcn_s_memtime() - stampC0, r = __builtin_amdgcn_s_memrealtime() - stampR0;
        hd.d_state[8] = (uint32_t)c; hd.d_state[9] = (uint32_t)r; hd.d_state[10] = (count + nActive - 1u) / nActive; hd.d_state[11] = nActive;
    }
    if (lane == 0u) { // every wave's life in the unused upper half of the compactified list: {start, end (100 MHz ticks), hardware id, xcc id}
        const uint32_t hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        reinterpret_cast<uint4*>(hd.d_hashCompactified)[nEntries / 2u + 3u * wFirst] =
            make_uint4((uint32_t)stampR0, (uint32_t)__builtin_amdgcn_s_memrealtime(), hwid, xcc);
        reinterpret_cast<uint4*>(hd.d_hashCompactified)[nEntries / 2u + 3u * wFirst + 1u] = make_uint4(stamps[0], stamps[1], stamps[2], stamps[3]);
        reinterpret_cast<uint4*>(hd.d_hashCompactified)[nEntries / 2u + 3u * wFirst + 2u] = make_uint4(stamps[4], stamps[5], 0u, 0u);
    }
#endif
}

template <bool PACKED>
__global__ __launch_bounds__(256)
void k_integrate_fused(FusedArgs args)
{
    __shared__ FusedShared<PACKED> sh;
    integrate_fused_body<PACKED, 0u>(args, blockIdx.x, gridDim.x, sh);
}

__global__ __launch_bounds__(256) void k_starve(VhHashData hd, VhHashParams hp)
{
    const uint32_t b = blockIdx.x;
    if (b >= hp.m_numOccupiedBlocks) return;
    const int ptr = __builtin_amdgcn_readfirstlane(load_quad(&hd.d_hashCompactified[b]).w);
    uint4* vp = reinterpret_cast<uint4*>(&hd.d_SDFBlocks[(uint32_t)ptr]) + threadIdx.x;
    uint4 raw = *vp;
    Vox v0 = starve_voxel(unpack_vox(make_uint2(raw.x, raw.y))), v1 = starve_voxel(unpack_vox(make_uint2(raw.z, raw.w)));
    raw.y = v0.cw; raw.w = v1.cw;
    *vp = raw;
}

__global__ __launch_bounds__(256) void k_gc_identify(VhHashData hd, VhHashParams hp, VhDepthCameraParams cp)
{
    __shared__ float sMin[4];
    __shared__ uint32_t sMax[4];
    const uint32_t b = blockIdx.x;
    if (b >= hp.m_numOccupiedBlocks) return;
    const int ptr = __builtin_amdgcn_readfirstlane(load_quad(&hd.d_hashCompactified[b]).w);
    const uint4 raw = *(reinterpret_cast<const uint4*>(&hd.d_SDFBlocks[(uint32_t)ptr]) + threadIdx.x);
    const Vox v0 = unpack_vox(make_uint2(raw.x, raw.y)), v1 = unpack_vox(make_uint2(raw.z, raw.w));
    float minSdf = fminf(gc_key(v0), gc_key(v1));
    uint32_t maxW = max(v0.weight(), v1.weight());
    block_min_max(minSdf, maxW, sMin, sMax);
    if (threadIdx.x == 0) {
        const float thr = get_truncation(hp, cp.m_sensorDepthWorldMax);
        hd.d_hashDecision[b] = ((minSdf >= thr) || (maxW == 0u)) ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void k_gc_free(VhHashData hd, VhHashParams hp, int32_t lockToken)
{
    __shared__ int sFreed;
    const uint32_t b = blockIdx.x;
    if (b >= hp.m_numOccupiedBlocks) return;
    if (hd.d_hashDecision[b] == 0) return; // block-uniform
    const int4 q = load_quad(&hd.d_hashCompactified[b]);
    if (threadIdx.x == 0) sFreed = delete_hash_entry_element(hd, hp, mki3(q.x, q.y, q.z), lockToken) ? 1 : 0;
    __syncthreads();
    if (sFreed) {
        const int ptr = __builtin_amdgcn_readfirstlane(q.w);
        *(reinterpret_cast<uint4*>(&hd.d_SDFBlocks[(uint32_t)ptr]) + threadIdx.x) = make_uint4(0u, 0u, 0u, 0u);
    }
}

// ---------------------------------------------------------------------------
// ray-interval splatting as a compute pass (the reference rasterises block quads with D3D11 into min/max depth
// textures, DSC/DX11RayIntervalSplatting.cpp:150-220 + rayIntervalSplatKernel DSC/CUDARayCastSDF.cu:101-167, and
// this fork then ignores them: DSC/CUDARayCastSDF.cu:36-42).  Here every ALLOCATED block (not only the ones the
// approximate frustum test keeps) is projected, one wave per block, and folded into the 8x8-pixel tiles it can
// matter to:
//   * head {min depth, max depth, count}: camera-depth range of those blocks (positive float bits: uint order =
//     float order) and how many there are;
//   * list: their {block position, voxel pointer}, up to `cap` per tile.
// Both are conservative supersets, so rendering with them cannot change a result: a sample reads voxels within one
// voxel of its position (two for the gradient), the box of a block is grown by more than that, and a perspective
// projection maps a box in front of the camera into the hull of its projected corners.  A tile whose list is
// complete therefore knows every block its rays can read: k_render resolves block -> pointer in LDS instead of
// probing the hash table in HBM (and a block missing from the list is unallocated, as a failed probe would say).
// ---------------------------------------------------------------------------

constexpr uint32_t kSplatWordsPerGroup = 32; // occupancy words (x32 buckets) per workgroup
constexpr uint32_t kSplatQueue = 128;        // blocks a workgroup queues per round

// (i / nx, i % nx) for i < 2^22 without the integer-division sequence: float estimate, corrected by one step
VHD void divmod_small(uint32_t i, uint32_t nx, float rnx, uint32_t& q, uint32_t& r)
{
    q = (uint32_t)((float)i * rnx);
    int rem = (int)i - (int)(q * nx);
    if (rem < 0) { q--; rem += (int)nx; }
    if (rem >= (int)nx) { q++; rem -= (int)nx; }
    r = (uint32_t)rem;
}

// Three steps per workgroup, each one trip to memory: occupancy words -> bucket queue (LDS); slots of the queued
// buckets -> block queue (LDS); one wave per queued block folds it into its tiles.  The queues balance the waves: a
// block costs a wave a few thousand cycles, and the kernel is as long as its busiest wave.
// Cost of a tile in units of one empty-space step; a full sample (8 taps) weighs kCostSample of them.  The wave's
// cost is its busiest lane's: k_render stores the tile's cost class for the next frame's launch order.
constexpr uint32_t kCostSample = 6;
constexpr uint32_t kCostClasses = 32;
constexpr uint32_t kCostClassWidth = 8;
// The dearest tiles of a frame are marched by TWO waves of one workgroup, each one half of the tile's depth interval:
// a wave alone on its SIMD runs at the pace of its own chain of dependent loads, and the few tiles with the longest
// rays (a silhouette grazing the truncation band) are the tail of the kernel.  A sample's state is the previous
// sample's alone, so the far half starts at the last sample before the middle with nothing remembered and arrives at
// the middle in the state the whole march would have; the near half's hit, if any, is the first along the ray.
#ifndef VH_SPLIT_TILES
#define VH_SPLIT_TILES 256
#endif
constexpr uint32_t kSplitTiles = VH_SPLIT_TILES;     // at most (even: two split tiles fill a workgroup); measured at 640x480: 64 -> 37.8 us, 256 -> 37.0, 512 -> 37.8
constexpr uint32_t kSplitMinTiles = 1024; // smaller images are not split
__host__ __device__ inline uint32_t split_tiles(uint32_t nTiles) // one tile in sixteen, dearest first
{
#ifndef VH_SPLIT_DIV
#define VH_SPLIT_DIV 16
#endif
    const uint32_t n = (nTiles / VH_SPLIT_DIV) & ~1u;
    return nTiles >= kSplitMinTiles ? (n < kSplitTiles ? n : kSplitTiles) : 0u;
}

// Launch order of the next k_render (one workgroup of k_interval_splat, beside the others): tiles sorted by the cost
// class the previous k_render stored, dearest first, dealt to the workgroups (4 tiles each) in rows of numCUs that
// alternate direction.  The hardware places workgroup g on compute unit g mod numCUs (all of them are resident), so
// a compute unit receives one workgroup of every row: the dearest of one row with the cheapest of the next.
// The sort was ONE workgroup's and the longest chain of its launch (per-workgroup time stamps, tools/riders_stamps.py: it
// ended at 8.4 us of a 9.9 us launch at 640x480, at 50 of 50 us at 1920x1080, everything else long done).  Now kSchedGroups
// workgroups sort a share of the tiles each -- the quads of four tiles q = part (mod parts): shares that look alike --
// and deal their sorted shares into the launch order in turns (the tile of rank i in share w comes after rank i of the
// shares before it): the dearest first as before, to within the difference between the shares.
constexpr uint32_t kSchedGroups = 8;
__host__ __device__ inline uint32_t sched_groups(uint32_t nTiles) { return nTiles >= kSplitMinTiles ? kSchedGroups : 1u; }

__device__ void schedule_tiles(uint32_t* sched, uint32_t* feedback, uint32_t nTiles, uint32_t phase, uint32_t numCUs, uint32_t part, uint32_t parts,
                               uint32_t* sCount /*[2 * 256]*/)
{
    // layout: {phase the slots were made for, -, -, -}, cost class per tile, {tile, phase} per launch slot
    const uint32_t* cls = sched + 4;
    uint2* slots = reinterpret_cast<uint2*>(sched + 4 + 4u * ((nTiles + 3u) / 4u));
    const uint32_t nSplit = split_tiles(nTiles);
    if (threadIdx.x == 0 && part == 0u) {
        sched[0] = phase;
        if (feedback) *feedback = sched[1]; // longest tile list the previous k_render met (0: none near the small capacity)
        sched[1] = 0u;
    }
    // counting sort with kSub sub-bins per class (keyed by the thread): 64 lanes adding to one LDS word serialise
    constexpr uint32_t kSub = 8, kBins = kCostClasses * kSub;
    static_assert(kBins == 4u * kWave, "the scan below gives four bins to each lane of one wave");
    for (uint32_t i = threadIdx.x; i < 2u * kBins; i += blockDim.x) sCount[i] = 0u;
    __syncthreads();
    const uint32_t sub = threadIdx.x % kSub;
    // four tiles per load, kBatch loads in flight: this workgroup is alone with its trips to memory
    constexpr uint32_t kBatch = 5;
    const uint32_t nQuads = (nTiles + 3u) / 4u; // the cost array is padded to whole quads (vh_render_schedule_bytes)
    const uint4* cls4 = reinterpret_cast<const uint4*>(cls);
    const uint32_t myQuads = part < nQuads ? (nQuads - part + parts - 1u) / parts : 0u; // quads part, part + parts, ...
    // tiles in each share (the image's last quad may be short), for the turn-taking below
    uint32_t nShare[kSchedGroups];
#pragma unroll
    for (uint32_t w = 0; w < kSchedGroups; w++) {
        const uint32_t nq = (w < parts && w < nQuads) ? (nQuads - w + parts - 1u) / parts : 0u;
        nShare[w] = 4u * nq - ((nq != 0u && (nQuads - 1u) % parts == w) ? 4u * nQuads - nTiles : 0u);
    }
    for (uint32_t q0 = threadIdx.x; q0 < myQuads; q0 += blockDim.x * kBatch) {
        uint4 c[kBatch];
#pragma unroll
        for (uint32_t j = 0; j < kBatch; j++) {
            const uint32_t q = q0 + j * blockDim.x;
            c[j] = q < myQuads ? cls4[q * parts + part] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (uint32_t j = 0; j < kBatch; j++) {
            const uint32_t t = ((q0 + j * blockDim.x) * parts + part) * 4u;
            if (t + 0u < nTiles) atomicAdd(&sCount[min(c[j].x, kCostClasses - 1u) * kSub + sub], 1u);
            if (t + 1u < nTiles) atomicAdd(&sCount[min(c[j].y, kCostClasses - 1u) * kSub + sub], 1u);
            if (t + 2u < nTiles) atomicAdd(&sCount[min(c[j].z, kCostClasses - 1u) * kSub + sub], 1u);
            if (t + 3u < nTiles) atomicAdd(&sCount[min(c[j].w, kCostClasses - 1u) * kSub + sub], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < kWave) { // start of each bin in the sorted order, dearest class first: one wave scans the 256 bins
        const uint32_t lane = threadIdx.x;
        uint32_t n[4], mine = 0u;
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) { n[j] = sCount[kBins - 1u - (lane * 4u + j)]; mine += n[j]; } // bins in descending order
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < (int)kWave; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
            if ((int)lane >= off) incl += up;
        }
        uint32_t start = incl - mine;
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) { sCount[kBins + kBins - 1u - (lane * 4u + j)] = start; start += n[j]; }
    }
    __syncthreads();
    // launch slots: the nSplit dearest tiles take two each (near half, far half; two tiles to a workgroup), the others one
    const uint32_t nGroups = (nTiles + nSplit + 3u) / 4u, fullRows = nGroups / numCUs;
    for (uint32_t q0 = threadIdx.x; q0 < myQuads; q0 += blockDim.x * kBatch) {
        uint4 c[kBatch];
#pragma unroll
        for (uint32_t j = 0; j < kBatch; j++) {
            const uint32_t q = q0 + j * blockDim.x;
            c[j] = q < myQuads ? cls4[q * parts + part] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (uint32_t j = 0; j < kBatch; j++) {
            const uint32_t cc[4] = { c[j].x, c[j].y, c[j].z, c[j].w };
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) {
                const uint32_t t = ((q0 + j * blockDim.x) * parts + part) * 4u + k;
                if (t < nTiles) {
                    const uint32_t mine = atomicAdd(&sCount[kBins + min(cc[k], kCostClasses - 1u) * kSub + sub], 1u); // rank of tile t in this share
                    // its rank overall: the shares take turns, a share that has run out is skipped
                    uint32_t i = 0u;
#pragma unroll
                    for (uint32_t w = 0; w < kSchedGroups; w++) i += min(mine, nShare[w]) + ((w < part && nShare[w] > mine) ? 1u : 0u);
                    if (i < nSplit) {
                        const uint32_t at = (i / 2u) * 4u + (i & 1u) * 2u;
                        slots[at] = make_uint2(t | (1u << 24), phase);
                        slots[at + 1u] = make_uint2(t | (2u << 24), phase);
                    } else {
                        const uint32_t j = i + nSplit;
                        uint32_t g = j / 4u;
                        const uint32_t row = g / numCUs, col = g % numCUs;
                        // (a row that also holds split tiles keeps its order: their slots are fixed)
                        if ((row & 1u) && row < fullRows && row * numCUs >= nSplit / 2u) g = row * numCUs + (numCUs - 1u - col);
                        slots[g * 4u + (j & 3u)] = make_uint2(t, phase);
                    }
                }
            }
        }
    }
}

struct SplatShared {
    uint32_t sBuckets[kSplatWordsPerGroup * 32];
    int4 sBlocks[kSplatQueue];
    uint32_t sNumBuckets, sNumBlocks;
};

// one workgroup of the splat: `group` < nSplatGroups takes 32 occupancy words, group == nSplatGroups makes the schedule
__device__ void interval_splat_group(const VhHashData& hd, const VhHashParams& hp, const VhDepthCameraParams& cp, const VhRayCastParams& rp,
                                     uint4* heads, int4* lists, uint32_t cap, uint32_t* sched, uint32_t phase, uint32_t numCUs,
                                     uint32_t nSplatGroups, uint32_t* feedback, uint32_t group, SplatShared& sh)
{
    uint32_t (&sBuckets)[kSplatWordsPerGroup * 32] = sh.sBuckets;
    int4 (&sBlocks)[kSplatQueue] = sh.sBlocks;
    uint32_t& sNumBuckets = sh.sNumBuckets;
    uint32_t& sNumBlocks = sh.sNumBlocks;
    const uint32_t nWords = (hp.m_hashNumBuckets + 31) / 32;
    const uint32_t lane = lane_id(), wave = threadIdx.x / kWave;
    const int tilesX = (int)((rp.m_width + 7) / 8), tilesY = (int)((rp.m_height + 7) / 8);
    const float vs = hp.m_virtualVoxelSize;
    // voxel indices a sample at p can read along one axis: floor(p/vs) and floor(p/vs)+1 (+-1/2 more for the
    // gradient's offset samples); block b holds indices 8b..8b+7  =>  p/vs in [8b-1, 8b+8) (+-1/2).  A quarter
    // voxel on top covers every rounding on the way (|p/vs| < 2^16 where the quotient is resolved to 2^-8).
    const float growLo = (rp.m_useGradients ? 1.75f : 1.25f) * vs, growHi = (rp.m_useGradients ? 0.75f : 0.25f) * vs;

    if (group >= nSplatGroups) { // the extra workgroups (launched only with a schedule)
        schedule_tiles(sched, feedback, (uint32_t)(tilesX * tilesY), phase, numCUs, group - nSplatGroups, sched_groups((uint32_t)(tilesX * tilesY)), sBuckets);
        return;
    }
    if (threadIdx.x == 0) { sNumBuckets = 0u; sNumBlocks = 0u; }
    __syncthreads();
    // 1: occupied buckets of this group's words
    if (threadIdx.x < kSplatWordsPerGroup) {
        const uint32_t wordIdx = group * kSplatWordsPerGroup + threadIdx.x;
        uint32_t bits = wordIdx < nWords ? hd.d_bucketBits[wordIdx] : 0u;
        while (bits) {
            const uint32_t bucket = wordIdx * 32u + (uint32_t)(__ffs((int)bits) - 1);
            bits &= bits - 1u;
            sBuckets[atomicAdd(&sNumBuckets, 1u)] = bucket;
        }
    }
    __syncthreads();
    const uint32_t nBuckets = sNumBuckets;
    // 2+3 in rounds of 12 buckets (120 slots, within the block queue)
    for (uint32_t b0 = 0; b0 < nBuckets; b0 += 12u) {
        const uint32_t bi = b0 + threadIdx.x / 16u, sl = threadIdx.x % 16u;
        if (threadIdx.x < 12u * 16u && bi < nBuckets && sl < VH_HASH_BUCKET_SIZE) {
            const int4 q = load_quad(&hd.d_hash[(uint64_t)sBuckets[bi] * VH_HASH_BUCKET_SIZE + sl]);
            if (q.w != VH_FREE_ENTRY) sBlocks[atomicAdd(&sNumBlocks, 1u)] = q;
        }
        __syncthreads();
        const uint32_t nBlocks = sNumBlocks;
        for (uint32_t k = wave; k < nBlocks; k += 256 / kWave) {
            const int4 q = sBlocks[k];
            const int bx = q.x, by = q.y, bz = q.z, ptr = q.w;
            const float lox = (float)(bx * VH_SDF_BLOCK_SIZE) * vs - growLo, hix = (float)(bx * VH_SDF_BLOCK_SIZE + VH_SDF_BLOCK_SIZE) * vs + growHi;
            const float loy = (float)(by * VH_SDF_BLOCK_SIZE) * vs - growLo, hiy = (float)(by * VH_SDF_BLOCK_SIZE + VH_SDF_BLOCK_SIZE) * vs + growHi;
            const float loz = (float)(bz * VH_SDF_BLOCK_SIZE) * vs - growLo, hiz = (float)(bz * VH_SDF_BLOCK_SIZE + VH_SDF_BLOCK_SIZE) * vs + growHi;
            float zmin = pinf(), zmax = minf(), xmin = pinf(), xmax = minf(), ymin = pinf(), ymax = minf();
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const F3 pc = mat_mul_p(rp.m_viewMatrix, mk3((c & 1) ? hix : lox, (c & 2) ? hiy : loy, (c & 4) ? hiz : loz));
                zmin = fminf(zmin, pc.z); zmax = fmaxf(zmax, pc.z);
                const float iz = 1.0f / fmaxf(pc.z, 1e-6f);
                const float sx = pc.x * cp.fx * iz + cp.mx, sy = pc.y * cp.fy * iz + cp.my;
                xmin = fminf(xmin, sx); xmax = fmaxf(xmax, sx);
                ymin = fminf(ymin, sy); ymax = fmaxf(ymax, sy);
            }
            if (!(zmax > 0.0f)) continue; // entirely behind the camera: no sample (depth > 0) lies in it
            int tx0 = 0, ty0 = 0, tx1 = tilesX - 1, ty1 = tilesY - 1;
            if (zmin > 0.05f) { // box in front of the camera; otherwise it may project anywhere: every tile
                const float slop = 1.0f + 1e-3f * fmaxf(fmaxf(fabsf(xmin), fabsf(xmax)), fmaxf(fabsf(ymin), fabsf(ymax)));
                // clamp in float first: the float -> int conversion of a huge coordinate is not defined
                tx0 = (int)fminf(fmaxf(floorf((xmin - slop) * 0.125f), 0.0f), (float)tilesX);
                ty0 = (int)fminf(fmaxf(floorf((ymin - slop) * 0.125f), 0.0f), (float)tilesY);
                tx1 = (int)fminf(fmaxf(floorf((xmax + slop) * 0.125f), -1.0f), (float)(tilesX - 1));
                ty1 = (int)fminf(fmaxf(floorf((ymax + slop) * 0.125f), -1.0f), (float)(tilesY - 1));
            }
            if (tx1 < tx0 || ty1 < ty0) continue; // off screen
            const float zs = 1e-3f * fabsf(zmax) + 0.5f * vs;
            const uint32_t lo = __float_as_uint(fmaxf(zmin - zs, 0.0f)), hi = __float_as_uint(fmaxf(zmax + zs, 0.0f));
            const uint32_t nx = (uint32_t)(tx1 - tx0 + 1), n = nx * (uint32_t)(ty1 - ty0 + 1);
            const float rnx = 1.0f / (float)nx;
            constexpr uint32_t kInFlight = 4; // tiles per lane in flight: the list slots come back from L2 together
            for (uint32_t base = 0; base < n; base += kInFlight * kWave) {
                uint32_t t[kInFlight], slot[kInFlight];
#pragma unroll
                for (uint32_t j = 0; j < kInFlight; j++) {
                    const uint32_t i = base + j * kWave + lane;
                    uint32_t qy, qx;
                    divmod_small(i, nx, rnx, qy, qx);
                    t[j] = (uint32_t)(ty0 + (int)qy) * (uint32_t)tilesX + (uint32_t)(tx0 + (int)qx);
                    slot[j] = 0xffffffffu;
                    if (i < n) {
                        slot[j] = atomicAdd(&heads[t[j]].z, 1u);
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < kInFlight; j++) {
                    if (slot[j] < cap) {
                        lists[(size_t)t[j] * cap + slot[j]] = make_int4(bx, by, bz, ptr);
                    } else if (slot[j] != 0xffffffffu) {
                        // the head's depth range covers the blocks that are NOT listed (all of them without lists);
                        // k_render forms the range of the listed ones itself: two atomics per tile and block less
                        atomicMin(&heads[t[j]].x, lo);
                        atomicMax(&heads[t[j]].y, hi);
                    }
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) sNumBlocks = 0u;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_interval_splat(VhHashData hd, VhHashParams hp, VhDepthCameraParams cp,
                                                        VhRayCastParams rp, uint4* heads, int4* lists, uint32_t cap,
                                                        uint32_t* sched, uint32_t phase, uint32_t numCUs, uint32_t nSplatGroups, uint32_t* feedback)
{
    __shared__ SplatShared sh;
    interval_splat_group(hd, hp, cp, rp, heads, lists, cap, sched, phase, numCUs, nSplatGroups, feedback, blockIdx.x, sh);
}

__global__ __launch_bounds__(256) void k_interval_clear(uint4* heads, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) heads[i] = make_uint4(0x7f800000u, 0u, 0u, 0u); // empty: {+inf, 0, no blocks}
}

// ---------------------------------------------------------------------------
// ray caster (renderKernel DSC/CUDARayCastSDF.cu:18-57,
// traverseCoarseGridSimpleSampleAll DSC/RayCastSDFUtil.h:198-262)
//
// One wave per 8x8 pixel tile (the rays of a tile stay in the same few blocks).  The reference's march is a chain
// of dependent gathers (8 hash probes x up to 11 entry loads + 8 voxel loads per sample, one after the other); here
// the work is re-ordered without changing an arithmetic operation, its order, or an early-out:
//   * k_render (with the tile heads / block lists of k_interval_splat): block -> pointer through a per-wave table in
//     LDS, the march confined to the tile's depth interval, one probe and one round of LDS reads per sample;
//   * k_render_hash (no intervals: the reference fork's behaviour): block -> pointer through the hash table behind
//     a 2-entry per-ray cache, empty space skipped on the bucket occupancy bit;
//   * both: division-free tap coordinates where they can be certified, the eight voxels of a sample in flight
//     together as 8-byte loads, march-until-sign-change / bisect / resume so that the lanes of a wave bisect together.
// ---------------------------------------------------------------------------

struct BlockCache {
    int bx0, by0, bz0, p0;
    int bx1, by1, bz1, p1;
    bool v0, v1;
};

VHD void cache_init(BlockCache& bc)
{
    bc.v0 = bc.v1 = false;
    bc.bx0 = bc.by0 = bc.bz0 = bc.bx1 = bc.by1 = bc.bz1 = 0;
    bc.p0 = bc.p1 = VH_FREE_ENTRY;
}

// getHashEntryForSDFBlockPos (DSC/VoxelUtilHashSDF.h:424-468) behind a 2-entry per-ray cache.  A miss loads the
// bucket's occupancy word and its first slot together (one round trip): a block sits in slot 0 unless two blocks
// share the bucket, and an empty bucket proves absence; only the remaining cases walk the bucket.
VHD int cached_lookup(const VhHashData& hd, const VhHashParams& hp, HashMod hm, BlockCache& bc, int bx, int by, int bz)
{
    if (bc.v0 && bc.bx0 == bx && bc.by0 == by && bc.bz0 == bz) return bc.p0;
    if (bc.v1 && bc.bx1 == bx && bc.by1 == by && bc.bz1 == bz) return bc.p1;
    const uint32_t h = hash_pos_fast(hm, mki3(bx, by, bz));
    const uint32_t word = hd.d_bucketBits[h >> 5];
    const int4 q = load_quad(&hd.d_hash[h * VH_HASH_BUCKET_SIZE]);
    int p;
    if (quad_matches(q, mki3(bx, by, bz))) p = q.w;
    else if (!((word >> (h & 31)) & 1u)) p = VH_FREE_ENTRY;
    else p = lookup_ptr(hd, hp, mki3(bx, by, bz));
    bc.bx1 = bc.bx0; bc.by1 = bc.by0; bc.bz1 = bc.bz0; bc.p1 = bc.p0; bc.v1 = bc.v0;
    bc.bx0 = bx; bc.by0 = by; bc.bz0 = bz; bc.p0 = p; bc.v0 = true;
    return p;
}

constexpr int kPtrUnknown = -3; // "first tap not resolved yet" (never a block pointer, VH_FREE_ENTRY or VH_LOCK_ENTRY)

// block -> voxel pointer through the hash table in HBM
struct HashLookup {
    const VhHashData& hd;
    const VhHashParams& hp;
    HashMod hm;
    BlockCache bc;
    VHD int find(int bx, int by, int bz) { return cached_lookup(hd, hp, hm, bc, bx, by, bz); }
    // may the sample whose first tap lies in this block be valid?  (occupancy bit of the bucket: one cached dword)
    VHD bool first_tap(int bx, int by, int bz, int& handle)
    {
        handle = kPtrUnknown;
        return bucket_maybe_occupied(hd, hash_pos_fast(hm, mki3(bx, by, bz)));
    }
    // voxel pointers of the eight taps (p[j]: bit0 = x1, bit1 = y1, bit2 = z1), one probe per DISTINCT block;
    // false if one of the blocks does not exist.  bit a of `straddle`: the tap pair along axis a lies in two blocks.
    VHD bool resolve(int, int bxa, int bya, int bza, int bxb, int byb, int bzb, uint32_t straddle, int (&p)[8])
    {
        const int p0 = find(bxa, bya, bza);
        if (p0 == VH_FREE_ENTRY) return false; // the first tap reads the zero voxel (weight 0)
#pragma unroll
        for (int j = 0; j < 8; j++) p[j] = p0;
        if (straddle) {
            uint32_t need = 0xfeu; // combos whose block differs from combo 0 still need a pointer
#pragma unroll
            for (uint32_t j = 1; j < 8u; j++)
                if ((j & straddle) == 0u) need &= ~(1u << j);
#pragma unroll 1
            while (need) {
                const uint32_t k = (uint32_t)__ffs((int)need) - 1u;
                const int pk = find((k & 1u) ? bxb : bxa, (k & 2u) ? byb : bya, (k & 4u) ? bzb : bza);
                if (pk == VH_FREE_ENTRY) return false;
                const uint32_t km = k & straddle;
                uint32_t same = 0u;
#pragma unroll
                for (uint32_t j = 1; j < 8u; j++) {
                    const bool eq = (j & straddle) == km;
                    p[j] = eq ? pk : p[j];
                    same |= (eq ? 1u : 0u) << j;
                }
                need &= ~same;
            }
        }
        return true;
    }
};

// block -> voxel pointers through the tile's own table in LDS, built from the tile's block list (open addressing).
// A slot is kTileSlotWords words: {x, y, z, p000, p100, p010, p110, p001, p101, p011, p111, -}: the block, its voxel
// pointer and the pointers of its seven +x/+y/+z neighbours (VH_FREE_ENTRY where there is none), so that a sample
// resolves its eight taps with one probe and one round of LDS reads, however many blocks they straddle.
// k_interval_splat lists every block the tile's rays can read, so a block that is not in a COMPLETE table is not
// allocated; when the list overflowed, the table holds a part of it and a miss falls back to the hash table.
constexpr uint32_t kTileSlotWords = 12;

template <uint32_t kTileTabSlots> // 2 x the list capacity (load factor <= 1/2), a power of two
struct TileLookup {
    const int* tab;
    bool complete;
    const VhHashData& hd;
    const VhHashParams& hp;
    VHD static uint32_t slot_of(int bx, int by, int bz)
    {
        // Multiply-shift hash, three full-rate 24-bit multiplies.  The blocks of a tile are a dense slab along its beam;
        // a linear form like x + 5y + 24z packs such a slab into ONE contiguous run of slots, and a probe for a block
        // that is not in the table (every empty-space step of a ray) then walks the whole run: 100 probes instead
        // of one on the long lists of fine voxels.
        return ((__umul24((uint32_t)bx, 0x9E3779u) + __umul24((uint32_t)by, 0x7F4A7Du) + __umul24((uint32_t)bz, 0x85EBCBu)) >> 10) & (kTileTabSlots - 1u);
    }
    VHD static int slot_find(const int* tab, int bx, int by, int bz)
    {
        uint32_t h = slot_of(bx, by, bz);
        for (;;) {
            const int4 e = *reinterpret_cast<const int4*>(&tab[h * kTileSlotWords]); // one 16-byte LDS read
            if (e.w == VH_FREE_ENTRY) return -1;
            if ((int)(e.x == bx) & (int)(e.y == by) & (int)(e.z == bz)) return (int)h;
            h = (h + 1u) & (kTileTabSlots - 1u);
        }
    }
    VHD int find_any(int bx, int by, int bz) const
    {
        const int sl = slot_find(tab, bx, by, bz);
        if (sl >= 0) return tab[(uint32_t)sl * kTileSlotWords + 3u];
        return complete ? VH_FREE_ENTRY : lookup_ptr(hd, hp, mki3(bx, by, bz));
    }
    // handle: the slot of the block (>= 0), or kPtrUnknown
    VHD bool first_tap(int bx, int by, int bz, int& handle) const
    {
        handle = slot_find(tab, bx, by, bz);
        if (handle >= 0) return true;
        handle = kPtrUnknown;
        return complete ? false : lookup_ptr(hd, hp, mki3(bx, by, bz)) != VH_FREE_ENTRY;
    }
    VHD bool resolve(int handle, int bxa, int bya, int bza, int bxb, int byb, int bzb, uint32_t straddle, int (&p)[8]) const
    {
        if (complete) {
            const int sl = handle >= 0 ? handle : slot_find(tab, bxa, bya, bza);
            if (sl < 0) return false; // the first tap reads the zero voxel (weight 0)
            // the tap pair of an axis lies in one block or in two adjacent ones: combo j reads neighbour (j & straddle)
            const int* e = &tab[(uint32_t)sl * kTileSlotWords + 3u];
            int all = 0;
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) {
                p[j] = e[j & straddle];
                all |= p[j];
            }
            return all >= 0; // voxel pointers are >= 0, VH_FREE_ENTRY is not
        }
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) p[k] = 0;
#pragma unroll 1
        for (uint32_t j = 0; j < 8u; j++) {
            const int pj = find_any((j & 1u) ? bxb : bxa, (j & 2u) ? byb : bya, (j & 4u) ? bzb : bza);
            if (pj == VH_FREE_ENTRY) return false;
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) p[k] = (k == j) ? pj : p[k];
        }
        return true;
    }
};

// One 8-byte load per voxel.  As an (indivisible) relaxed atomic: an ordinary load is split by the compiler into its
// two words, and the sdf word is then fetched only after the weight test -- a second trip to memory per sample.
VHD uint2 load_voxel(const VhHashData& hd, int ptr, int lx, int ly, int lz)
{
    const unsigned long long v = __hip_atomic_load(
        reinterpret_cast<const unsigned long long*>(&hd.d_SDFBlocks[(uint32_t)ptr + (uint32_t)(lz * 64 + ly * 8 + lx)]),
        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

struct Taps {
    int x0, y0, z0, x1, y1, z1;       // voxel coordinates of the lower / upper tap per axis
    int bxa, bya, bza, bxb, byb, bzb; // their SDF blocks (virtualVoxelPosToSDFBlock)
};

// trilinearInterpolationSimpleFastFast, DSC/RayCastSDFUtil.h:97-116, for tap
// voxel coordinates (x0..z1) the caller has already resolved.
// Returns false exactly when the reference does (some tap, in tap order, has
// weight 0 -- a missing block yields the zero voxel); `dist` is then undefined
// (the reference leaves a partial sum that only gradientForPoint can observe:
// see trilinear_partial below).  The colour is accumulated only on request:
// the reference computes it for every sample but reads it only from the last
// bisection sample (RayCastSDFUtil.h:231,241).
template <bool COLOR, class LK>
VHD bool trilinear(const VhHashData& hd, float vs, LK& lk, int p0in, const Taps& tp,
                   F3 pos, float rvs, float& dist, uint32_t& colorOut)
{
    const int x0 = tp.x0, y0 = tp.y0, z0 = tp.z0, x1 = tp.x1, y1 = tp.y1, z1 = tp.z1;
    const int bxa = tp.bxa, bya = tp.bya, bza = tp.bza, bxb = tp.bxb, byb = tp.byb, bzb = tp.bzb;
    // bit a of `straddle`: the tap pair along axis a lies in two different blocks
    const uint32_t straddle = (bxb != bxa ? 1u : 0u) | (byb != bya ? 2u : 0u) | (bzb != bza ? 4u : 0u);

    // voxel pointer per tap combo (bit0 = x1, bit1 = y1, bit2 = z1)
    int p[8];
    if (!lk.resolve(p0in, bxa, bya, bza, bxb, byb, bzb, straddle, p)) return false;
    const int p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4], p5 = p[5], p6 = p[6], p7 = p[7];

    const int lx0 = x0 & 7, ly0 = y0 & 7, lz0 = z0 & 7; // = local1(): two's complement & 7 is the non-negative remainder
    const int lx1 = x1 & 7, ly1 = y1 & 7, lz1 = z1 & 7;
    // the eight voxels, in flight together (reference tap order 000,100,010,001,110,011,101,111)
    uint2 r000 = load_voxel(hd, p0, lx0, ly0, lz0), r100 = load_voxel(hd, p1, lx1, ly0, lz0);
    uint2 r010 = load_voxel(hd, p2, lx0, ly1, lz0), r001 = load_voxel(hd, p4, lx0, ly0, lz1);
    uint2 r110 = load_voxel(hd, p3, lx1, ly1, lz0), r011 = load_voxel(hd, p6, lx0, ly1, lz1);
    uint2 r101 = load_voxel(hd, p5, lx1, ly0, lz1), r111 = load_voxel(hd, p7, lx1, ly1, lz1);
    // one trip to memory for the eight: left alone the compiler waits for the first pair before it issues the rest
    // (three trips), and the dearest waves of a frame run at the pace of their own chain of loads
    asm volatile("" : "+v"(r000.x), "+v"(r000.y), "+v"(r100.x), "+v"(r100.y), "+v"(r010.x), "+v"(r010.y), "+v"(r001.x), "+v"(r001.y),
                      "+v"(r110.x), "+v"(r110.y), "+v"(r011.x), "+v"(r011.y), "+v"(r101.x), "+v"(r101.y), "+v"(r111.x), "+v"(r111.y));
    const Vox v000 = unpack_vox(r000), v100 = unpack_vox(r100), v010 = unpack_vox(r010), v001 = unpack_vox(r001);
    const Vox v110 = unpack_vox(r110), v011 = unpack_vox(r011), v101 = unpack_vox(r101), v111 = unpack_vox(r111);
    // weight byte == 0 for any tap <=> min over the taps of (cw >> 24) == 0
    const uint32_t wmin = min(min(min(v000.cw, v100.cw), min(v010.cw, v001.cw)), min(min(v110.cw, v011.cw), min(v101.cw, v111.cw)));
    if ((wmin >> 24) == 0u) return false;

    const float fx = div_exact(pos.x, vs, rvs), fy = div_exact(pos.y, vs, rvs), fz = div_exact(pos.z, vs, rvs);
    const float wx = fx - floorf(fx), wy = fy - floorf(fy), wz = fz - floorf(fz);
    const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
    float d = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
#define VH_ACC(V, WX, WY, WZ)                                           \
    {                                                                   \
        const float s = (WX) * (WY) * (WZ);                             \
        d += s * (V).sdf;                                               \
        if (COLOR) { cr += s * (float)(V).r(); cg += s * (float)(V).g(); cb += s * (float)(V).b(); } \
    }
    VH_ACC(v000, ux, uy, uz)
    VH_ACC(v100, wx, uy, uz)
    VH_ACC(v010, ux, wy, uz)
    VH_ACC(v001, ux, uy, wz)
    VH_ACC(v110, wx, wy, uz)
    VH_ACC(v011, ux, wy, wz)
    VH_ACC(v101, wx, uy, wz)
    VH_ACC(v111, wx, wy, wz)
#undef VH_ACC
    dist = d;
    if (COLOR) colorOut = (uint32_t)f2uc(cr) | ((uint32_t)f2uc(cg) << 8) | ((uint32_t)f2uc(cb) << 16);
    return true;
}

// trilinear<false> in two steps, for a march that asks for the NEXT sample's voxels before it blends this one's
// (march_ray, PIPELINED): taps_issue resolves the eight pointers and puts the eight loads in flight (false: a block is
// missing, the sample is invalid and nothing was asked for), taps_finish is the weight test and the blend.
struct TapLoads {
    uint2 r000, r100, r010, r001, r110, r011, r101, r111;
};
template <class LK>
VHD bool taps_issue(const VhHashData& hd, LK& lk, int p0in, const Taps& tp, TapLoads& L)
{
    const uint32_t straddle = (tp.bxb != tp.bxa ? 1u : 0u) | (tp.byb != tp.bya ? 2u : 0u) | (tp.bzb != tp.bza ? 4u : 0u);
    int p[8];
    if (!lk.resolve(p0in, tp.bxa, tp.bya, tp.bza, tp.bxb, tp.byb, tp.bzb, straddle, p)) return false;
    const int lx0 = tp.x0 & 7, ly0 = tp.y0 & 7, lz0 = tp.z0 & 7;
    const int lx1 = tp.x1 & 7, ly1 = tp.y1 & 7, lz1 = tp.z1 & 7;
    L.r000 = load_voxel(hd, p[0], lx0, ly0, lz0); L.r100 = load_voxel(hd, p[1], lx1, ly0, lz0);
    L.r010 = load_voxel(hd, p[2], lx0, ly1, lz0); L.r001 = load_voxel(hd, p[4], lx0, ly0, lz1);
    L.r110 = load_voxel(hd, p[3], lx1, ly1, lz0); L.r011 = load_voxel(hd, p[6], lx0, ly1, lz1);
    L.r101 = load_voxel(hd, p[5], lx1, ly0, lz1); L.r111 = load_voxel(hd, p[7], lx1, ly1, lz1);
    return true;
}
VHD bool taps_finish(TapLoads& L, float vs, float rvs, F3 pos, float& dist)
{
    // (all eight have arrived together: see trilinear)
    asm volatile("" : "+v"(L.r000.x), "+v"(L.r000.y), "+v"(L.r100.x), "+v"(L.r100.y), "+v"(L.r010.x), "+v"(L.r010.y), "+v"(L.r001.x), "+v"(L.r001.y),
                      "+v"(L.r110.x), "+v"(L.r110.y), "+v"(L.r011.x), "+v"(L.r011.y), "+v"(L.r101.x), "+v"(L.r101.y), "+v"(L.r111.x), "+v"(L.r111.y));
    const Vox v000 = unpack_vox(L.r000), v100 = unpack_vox(L.r100), v010 = unpack_vox(L.r010), v001 = unpack_vox(L.r001);
    const Vox v110 = unpack_vox(L.r110), v011 = unpack_vox(L.r011), v101 = unpack_vox(L.r101), v111 = unpack_vox(L.r111);
    const uint32_t wmin = min(min(min(v000.cw, v100.cw), min(v010.cw, v001.cw)), min(min(v110.cw, v011.cw), min(v101.cw, v111.cw)));
    if ((wmin >> 24) == 0u) return false;
    const float fx = div_exact(pos.x, vs, rvs), fy = div_exact(pos.y, vs, rvs), fz = div_exact(pos.z, vs, rvs);
    const float wx = fx - floorf(fx), wy = fy - floorf(fy), wz = fz - floorf(fz);
    const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
    float d = 0.0f;
    // (the reference's tap order and arithmetic: trilinear)
    d += ((ux * uy) * uz) * v000.sdf;
    d += ((wx * uy) * uz) * v100.sdf;
    d += ((ux * wy) * uz) * v010.sdf;
    d += ((ux * uy) * wz) * v001.sdf;
    d += ((wx * wy) * uz) * v110.sdf;
    d += ((ux * wy) * wz) * v011.sdf;
    d += ((wx * uy) * wz) * v101.sdf;
    d += ((wx * wy) * wz) * v111.sdf;
    dist = d;
    return true;
}

// The same function with the reference's tap-by-tap early-out, leaving the
// partial sum in `dist` on failure: gradientForPoint (:174-195) ignores the
// return value and uses that partial sum.
__device__ __noinline__ float trilinear_partial(const VhHashData hd, const VhHashParams hp, float px, float py, float pz)
{
    const float vs = hp.m_virtualVoxelSize;
    const float oSet = vs;
    const float h = oSet / 2.0f;
    const F3 pd = mk3(px - h, py - h, pz - h);
    const float fx = px / vs, fy = py / vs, fz = pz / vs;
    const float wx = fx - floorf(fx), wy = fy - floorf(fy), wz = fz - floorf(fz);
    float d = 0.0f;
#pragma unroll 1
    for (uint32_t k = 0; k < 8u; k++) {
        // reference tap order 000,100,010,001,110,011,101,111 as (x,y,z) bit triples
        const uint32_t combo = (0x75634210u >> (4u * k)) & 7u; // k-th nibble: 0,1,2,4,3,6,5,7
        const bool bx = combo & 1u, by = combo & 2u, bz = combo & 4u;
        const int vx = world_to_vvp1(bx ? pd.x + oSet : pd.x, vs);
        const int vy = world_to_vvp1(by ? pd.y + oSet : pd.y, vs);
        const int vz = world_to_vvp1(bz ? pd.z + oSet : pd.z, vs);
        const int p = lookup_ptr(hd, hp, mki3(vvp_to_block1(vx), vvp_to_block1(vy), vvp_to_block1(vz)));
        if (p == VH_FREE_ENTRY) return d;
        const Vox v = unpack_vox(load_voxel(hd, p, local1(vx), local1(vy), local1(vz)));
        if (v.weight() == 0u) return d;
        const float s = (bx ? wx : 1.0f - wx) * (by ? wy : 1.0f - wy) * (bz ? wz : 1.0f - wz);
        d += s * v.sdf;
    }
    return d;
}

// gradientForPoint :174-195
VHD F3 gradient_for_point(const VhHashData& hd, const VhHashParams& hp, F3 pos)
{
    const float vs = hp.m_virtualVoxelSize;
    const float dp00 = trilinear_partial(hd, hp, pos.x - 0.5f * vs, pos.y - 0.0f, pos.z - 0.0f);
    const float d0p0 = trilinear_partial(hd, hp, pos.x - 0.0f, pos.y - 0.5f * vs, pos.z - 0.0f);
    const float d00p = trilinear_partial(hd, hp, pos.x - 0.0f, pos.y - 0.0f, pos.z - 0.5f * vs);
    const float d100 = trilinear_partial(hd, hp, pos.x + 0.5f * vs, pos.y + 0.0f, pos.z + 0.0f);
    const float d010 = trilinear_partial(hd, hp, pos.x + 0.0f, pos.y + 0.5f * vs, pos.z + 0.0f);
    const float d001 = trilinear_partial(hd, hp, pos.x + 0.0f, pos.y + 0.0f, pos.z + 0.5f * vs);
    const F3 g = mk3((dp00 - d100) / vs, (d0p0 - d010) / vs, (d00p - d001) / vs);
    const float l = sqrtf(dot3(g, g));
    if (l == 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    return mk3(-g.x / l, -g.y / l, -g.z / l);
}

// Tap voxel coordinates of the sample at ray parameter t.
// Division-free when possible: q ~ pos/voxel is evaluated approximately (one fma per axis).  The lower tap
// of an axis is round_half_away((pos - voxel/2)/voxel) = floor(pos/voxel) and the upper tap is that + 1
// whenever pos/voxel is not within rounding distance of an integer.  With u = 2^-24 and Q a bound on
// |pos/voxel| along the ray, the approximate q is off by <= 3uQ and the reference's own quotient (two
// products, two sums, a division and the +-0.5) by <= 5uQ; a fractional part at least 32uQ away from 0
// and 1 therefore settles both taps.  Anything closer takes the exact path with the reference's
// arithmetic.  The decision is made per WAVE-iteration in practice (one uncertain lane sends the whole
// wave through the exact code), hence the tight margin: ~0.3 % per lane at Q = 256.
struct RayQ {
    F3 cam, dir;   // exact ray (world)
    F3 camq, dirq; // ~ cam/voxel, dir/voxel
    float vs, rvs, halfVoxel;
    float certLim; // 0.5 - margin, negative if the approximation must not be used
};

VHD void tap_coords(const RayQ& rq, float t, Taps& tp)
{
    const float qx = __fmaf_rn(t, rq.dirq.x, rq.camq.x), qy = __fmaf_rn(t, rq.dirq.y, rq.camq.y), qz = __fmaf_rn(t, rq.dirq.z, rq.camq.z);
    const float gx = floorf(qx), gy = floorf(qy), gz = floorf(qz);
    const float dev = fmaxf(fmaxf(fabsf((qx - gx) - 0.5f), fabsf((qy - gy) - 0.5f)), fabsf((qz - gz) - 0.5f));
    if (dev < rq.certLim) {
        tp.x0 = (int)gx; tp.y0 = (int)gy; tp.z0 = (int)gz;
        tp.x1 = tp.x0 + 1; tp.y1 = tp.y0 + 1; tp.z1 = tp.z0 + 1;
        // |coordinates| < 2^16 here: the arithmetic shift is the reference's floor division by 8
        tp.bxa = tp.x0 >> 3; tp.bya = tp.y0 >> 3; tp.bza = tp.z0 >> 3;
        tp.bxb = tp.x1 >> 3; tp.byb = tp.y1 >> 3; tp.bzb = tp.z1 >> 3;
    } else {
        const F3 pd = mk3((rq.cam.x + t * rq.dir.x) - rq.halfVoxel, (rq.cam.y + t * rq.dir.y) - rq.halfVoxel, (rq.cam.z + t * rq.dir.z) - rq.halfVoxel);
        tp.x0 = world_to_vvp1_rb(pd.x, rq.vs, rq.rvs); tp.y0 = world_to_vvp1_rb(pd.y, rq.vs, rq.rvs); tp.z0 = world_to_vvp1_rb(pd.z, rq.vs, rq.rvs);
        tp.x1 = world_to_vvp1_rb(pd.x + rq.vs, rq.vs, rq.rvs); tp.y1 = world_to_vvp1_rb(pd.y + rq.vs, rq.vs, rq.rvs); tp.z1 = world_to_vvp1_rb(pd.z + rq.vs, rq.vs, rq.rvs);
        tp.bxa = vvp_to_block1(tp.x0); tp.bya = vvp_to_block1(tp.y0); tp.bza = vvp_to_block1(tp.z0);
        tp.bxb = vvp_to_block1(tp.x1); tp.byb = vvp_to_block1(tp.y1); tp.bzb = vvp_to_block1(tp.z1);
    }
}


#ifndef VH_PIPELINE_SMALL // (measurement builds: the pipelined march with the small tables too)
#define VH_PIPELINE_SMALL 0
#endif

struct RayHit {
    float alpha;    // ray parameter of the accepted intersection; NaN-free flag in `hit`
    uint32_t color; // packed colour of the last bisection sample
    bool hit;
    float depthToRayLength;
    F3 normal;      // GRADIENTS only (camera space)
};

// traverseCoarseGridSimpleSampleAll, DSC/RayCastSDFUtil.h:198-262, for the ray of pixel (x, y), restricted to the
// tile's conservative depth interval [tileZmin, tileZmax].  Written as march-until-sign-change / bisect / resume so
// that the lanes of a wave run their bisections together instead of interleaving them with other lanes' marching;
// each ray's own sequence of samples is the reference's.
template <bool GRADIENTS, bool PIPELINED = false, class LK>
VHD void march_ray(LK& lk, const VhHashData& hd, const VhHashParams& hp, const VhDepthCameraParams& cp, const VhRayCastParams& rp,
                   uint32_t x, uint32_t y, float tileZmin, float tileZmax, uint32_t half, float zMid, RayHit& out, uint32_t& cost
#ifdef VH_RENDER_STATS
                   , float& statTri, float& statIter, float (&statCyc)[3]
#endif
)
{
    const float mi = minf();
    const F3 camDir = normalize3(depth_to_skeleton(cp, x, y, proj_to_cam_z(cp, 1.0f)));
    const F3 worldCamPos = mat_mul_p(rp.m_viewMatrixInverse, mk3(0.0f, 0.0f, 0.0f));
    const F3 worldDir = normalize3(mat_mul_d(rp.m_viewMatrixInverse, camDir));

    const float minInterval = rp.m_minDepth, maxInterval = rp.m_maxDepth;
    const bool run = !(minInterval == 0.0f || minInterval == mi) && !(maxInterval == 0.0f || maxInterval == mi);
    if (!run) return;
    const float depthToRayLength = 1.0f / camDir.z;
    out.depthToRayLength = depthToRayLength;
    const float rayEnd = depthToRayLength * fminf(rp.m_maxDepth, maxInterval);
    const float inc = rp.m_rayIncrement;
    RayQ rq;
    rq.cam = worldCamPos; rq.dir = worldDir;
    rq.vs = hp.m_virtualVoxelSize;
    rq.rvs = 1.0f / rq.vs; // IEEE reciprocal for div_exact
    rq.halfVoxel = rq.vs / 2.0f;
    rq.camq = mk3(worldCamPos.x * rq.rvs, worldCamPos.y * rq.rvs, worldCamPos.z * rq.rvs);
    rq.dirq = mk3(worldDir.x * rq.rvs, worldDir.y * rq.rvs, worldDir.z * rq.rvs);
    {
        // Q bounds |pos/voxel| for every sample of this ray (march and bisection parameters are <= rayEnd)
        const float Q = 1.0f + (fabsf(rq.camq.x) + fabsf(rq.camq.y) + fabsf(rq.camq.z)) +
                        fabsf(rayEnd) * (fabsf(rq.dirq.x) + fabsf(rq.dirq.y) + fabsf(rq.dirq.z));
        // Margin: 16 u Q with u = 2^-24.  Per axis, with S = (|cam| + t |dir|) / voxel <= Q - 1: the fused route
        // q = fma(t, dir * r, cam * r), r = fl(1 / voxel), is within 3 u S of the real (cam + t dir) / voxel (r, the
        // two scaled operands, the fma); the reference's route -- fl(t dir), + cam, - half, / voxel, +- 0.5, and for
        // the upper tap + voxel before the division -- within u (6 S + 5.5) of the same number.  Both take the floor:
        // they agree when q is further than u (9 S + 5.5) < 16 u Q from an integer.
        const float margin = Q * (16.0f / 16777216.0f);
        rq.certLim = (Q < 65536.0f) ? 0.5f - margin : -1.0f; // NaN/inf Q compare false: exact path
    }

    float rcur = depthToRayLength * fmaxf(rp.m_minDepth, minInterval); // rayCurrent
    float lastSdf = 0.0f, lastAlpha = 0.0f;
    int lastValid = 0; // flags live in VGPRs: on this kernel the scalar unit (one per CU) is the scarce resource

    // Interval of the tile in ray-parameter units.  Samples before it have their first tap in no allocated block
    // (they are invalid: lastValid = 0), samples after it likewise, so nothing can be hit there.  rcur still
    // advances by the same sequence of additions, one VALU op per skipped sample.
    const float tSkip = depthToRayLength * tileZmin;
    float tStop = fminf(rayEnd, depthToRayLength * tileZmax);
#pragma unroll 1
    while (rcur < tSkip && rcur < rayEnd) rcur += inc;
    if (half != 0u) { // half of a split tile (wave-uniform): the near one samples before tMid, the far one from the last sample before it
        const float tMid = depthToRayLength * zMid;
        if (half == 1u) {
            tStop = fminf(tStop, tMid);
        } else {
#pragma unroll 1
            for (;;) {
                const float nxt = rcur + inc; // the same additions the march makes
                if (!(nxt < tMid) || !(nxt < rayEnd)) break;
                rcur = nxt;
            }
        }
    }

    if constexpr (PIPELINED) {
        // The same march with the NEXT sample's voxels asked for before this sample's are blended.  With three waves on a SIMD
        // (the large tables) and, in a frame's long tail, one, a wave otherwise runs at the pace of its own chain: probe,
        // eight loads, a trip to memory, blend, probe, ...  The next sample is the one the march would take next whatever this
        // one turns out to be (also after a bisection that did not end the ray: the march resumes one increment on), so what
        // is asked for early is never asked for in vain but at the end of the ray.  Every ray's sequence of samples, and what
        // is computed from each, is the plain march's.
        struct Pending {
            float r;      // the sample's ray parameter
            int skipped;  // samples without a block lay between the previous sample and this one
            int state;    // 0: the ray has left the range, 1: the eight voxels are in flight, 2: a block is missing (invalid)
            TapLoads L;
        };
        auto fetch = [&](float r0) -> Pending {
            Pending f;
            f.skipped = 0;
            float r = r0;
            Taps tp;
            int p0 = kPtrUnknown;
#pragma unroll 1
            while (r < tStop) { // (A: samples whose first tap has no block)
                cost += 1u;
                tap_coords(rq, r, tp);
                if (lk.first_tap(tp.bxa, tp.bya, tp.bza, p0)) break;
                f.skipped = 1;
                r += inc;
            }
            f.r = r;
            f.state = 0;
            if (r < tStop) {
                cost += kCostSample;
                f.state = taps_issue(hd, lk, p0, tp, f.L) ? 1 : 2;
            }
            return f;
        };
        Pending P = fetch(rcur);
#pragma unroll 1
        for (;;) {
            bool candidate = false;
            float dist = 0.0f;
            Pending N;
            N.state = 0; N.r = 0.0f; N.skipped = 0;
#pragma unroll 1
            for (;;) {
                if (P.state == 0) break; // ray left the depth range (or the range in which blocks exist)
                N = fetch(P.r + inc);
                rcur = P.r;
                lastValid = P.skipped ? 0 : lastValid;
                bool ok = false;
                if (P.state == 1) ok = taps_finish(P.L, rq.vs, rq.rvs, mk3(worldCamPos.x + rcur * worldDir.x, worldCamPos.y + rcur * worldDir.y, worldCamPos.z + rcur * worldDir.z), dist);
                if (ok & (lastValid != 0) & (lastSdf > 0.0f) & (dist < 0.0f)) { candidate = true; break; }
                lastSdf = ok ? dist : lastSdf;
                lastAlpha = ok ? rcur : lastAlpha;
                lastValid = ok ? 1 : 0;
                P = N;
            }
            if (!candidate) break;
            // ---- findIntersectionBisection :149-170 on [lastAlpha, rcur] (as below)
            float a = lastAlpha, aDist = lastSdf, b = rcur, bDist = dist, c = 0.0f;
            uint32_t color2 = 0u;
            bool success = true;
#pragma unroll 1
            for (int i = 0; i < 3; i++) {
                cost += kCostSample;
                c = a + (aDist / (aDist - bDist)) * (b - a); // findIntersectionLinear :140-143
                Taps ctp;
                tap_coords(rq, c, ctp);
                const F3 cpos = mk3(worldCamPos.x + c * worldDir.x, worldCamPos.y + c * worldDir.y, worldCamPos.z + c * worldDir.z);
                float cDist = 0.0f;
                if (!trilinear<true>(hd, rq.vs, lk, kPtrUnknown, ctp, cpos, rq.rvs, cDist, color2)) { success = false; break; }
                if (aDist * cDist > 0.0f) { a = c; aDist = cDist; }
                else { b = c; bDist = cDist; }
            }
            if (success && fabsf(lastSdf - dist) < rp.m_thresSampleDist && fabsf(dist) < rp.m_thresDist) {
                out.hit = true;
                out.alpha = c;
                out.color = color2;
                if (GRADIENTS) {
                    const F3 iso = mk3(worldCamPos.x + c * worldDir.x, worldCamPos.y + c * worldDir.y, worldCamPos.z + c * worldDir.z);
                    const F3 g = gradient_for_point(hd, hp, iso);
                    out.normal = mat_mul_d(rp.m_viewMatrix, mk3(-g.x, -g.y, -g.z));
                }
                break;
            }
            // no accepted hit: the (valid) march sample becomes the last sample and the march goes on (:248-252) -- at N
            lastSdf = dist;
            lastAlpha = rcur;
            lastValid = 1;
            P = N;
        }
        return;
    }

#pragma unroll 1
    for (;;) {
        // ---- march until a sign change (or the end of the range).  The lanes of a wave leave this loop together:
        // whichever lane finds its sign change first waits here, so that the wave runs ONE bisection phase for all
        // of them instead of one per march step.  Each ray's own sequence of samples is the reference's.
        bool candidate = false;
        float dist = 0.0f;
#pragma unroll 1
        for (;;) {
            // ---- A: skip samples whose first tap has no block (they are invalid: weight 0 at the first tap)
            Taps tp;
            int p0 = kPtrUnknown;
            int skipped = 0;
#ifdef VH_RENDER_STATS
            const long long tA0 = clock64();
#endif
#pragma unroll 1
            while (rcur < tStop) {
#ifdef VH_RENDER_STATS
                statIter += 1024.0f / (float)__popcll(__ballot(1));
#endif
                cost += 1u;
                tap_coords(rq, rcur, tp);
                if (lk.first_tap(tp.bxa, tp.bya, tp.bza, p0)) break;
                skipped = 1;
                rcur += inc;
            }
#ifdef VH_RENDER_STATS
            statCyc[0] += (float)(clock64() - tA0);
            const long long tB0 = clock64();
#endif
            if (!(rcur < tStop)) break; // ray left the depth range (or the range in which blocks exist)
            lastValid = skipped ? 0 : lastValid;

            // ---- B: full sample at rcur
#ifdef VH_RENDER_STATS
            statTri += 1.0f;
            statIter += 1.0f / (float)__popcll(__ballot(1));
#endif
            cost += kCostSample;
            uint32_t colorUnused = 0u;
            const F3 pos = mk3(worldCamPos.x + rcur * worldDir.x, worldCamPos.y + rcur * worldDir.y, worldCamPos.z + rcur * worldDir.z);
            const bool ok = trilinear<false>(hd, rq.vs, lk, p0, tp, pos, rq.rvs, dist, colorUnused);
#ifdef VH_RENDER_STATS
            statCyc[1] += (float)(clock64() - tB0);
#endif
            if (ok & (lastValid != 0) & (lastSdf > 0.0f) & (dist < 0.0f)) { candidate = true; break; }
            lastSdf = ok ? dist : lastSdf;
            lastAlpha = ok ? rcur : lastAlpha;
            lastValid = ok ? 1 : 0;
            rcur += inc;
        }
        if (!candidate) break;

        // ---- findIntersectionBisection :149-170 on [lastAlpha, rcur]
        float a = lastAlpha, aDist = lastSdf, b = rcur, bDist = dist, c = 0.0f;
        uint32_t color2 = 0u;
        bool success = true;
#ifdef VH_RENDER_STATS
        const long long tC0 = clock64();
#endif
#pragma unroll 1
        for (int i = 0; i < 3; i++) {
#ifdef VH_RENDER_STATS
            statTri += 1.0f;
            statIter += 1.0f / (float)__popcll(__ballot(1));
#endif
            cost += kCostSample;
            c = a + (aDist / (aDist - bDist)) * (b - a); // findIntersectionLinear :140-143
            Taps ctp;
            tap_coords(rq, c, ctp);
            const F3 cpos = mk3(worldCamPos.x + c * worldDir.x, worldCamPos.y + c * worldDir.y, worldCamPos.z + c * worldDir.z);
            float cDist = 0.0f;
            if (!trilinear<true>(hd, rq.vs, lk, kPtrUnknown, ctp, cpos, rq.rvs, cDist, color2)) { success = false; break; }
            if (aDist * cDist > 0.0f) { a = c; aDist = cDist; }
            else { b = c; bDist = cDist; }
        }
#ifdef VH_RENDER_STATS
        statCyc[2] += (float)(clock64() - tC0);
#endif
        if (success && fabsf(lastSdf - dist) < rp.m_thresSampleDist && fabsf(dist) < rp.m_thresDist) {
            out.hit = true;
            out.alpha = c;
            out.color = color2;
            if (GRADIENTS) {
                const F3 iso = mk3(worldCamPos.x + c * worldDir.x, worldCamPos.y + c * worldDir.y, worldCamPos.z + c * worldDir.z);
                const F3 g = gradient_for_point(hd, hp, iso);
                out.normal = mat_mul_d(rp.m_viewMatrix, mk3(-g.x, -g.y, -g.z));
            }
            break;
        }
        // no accepted hit: the (valid) march sample becomes the last sample and the march goes on (:248-252)
        lastSdf = dist;
        lastAlpha = rcur;
        lastValid = 1;
        rcur += inc;
    }
}

// the maps of one pixel (renderKernel DSC/CUDARayCastSDF.cu:18-57; outputs of traverseCoarseGridSimpleSampleAll :236-247)
VHD void store_ray(const VhRayCastData& rd, const VhDepthCameraParams& cp, size_t pix, uint32_t x, uint32_t y, const RayHit& r, bool gradients)
{
    const float mi = minf();
    float depth = mi;
    float4 depth4 = make_float4(mi, mi, mi, mi), normal = depth4, color = depth4;
    if (r.hit) {
        depth = r.alpha / r.depthToRayLength;
        const F3 sk = depth_to_skeleton(cp, x, y, depth);
        depth4 = make_float4(sk.x, sk.y, sk.z, 1.0f);
        color = make_float4((float)(r.color & 0xffu) / 255.f, (float)((r.color >> 8) & 0xffu) / 255.f, (float)((r.color >> 16) & 0xffu) / 255.f, 1.0f);
        if (gradients) normal = make_float4(r.normal.x, r.normal.y, r.normal.z, 1.0f);
    }
    rd.d_depth[pix] = depth;
    reinterpret_cast<float4*>(rd.d_depth4)[pix] = depth4;
    if (rd.d_normals) reinterpret_cast<float4*>(rd.d_normals)[pix] = normal; // a null map is not written (vh_api.h)
    reinterpret_cast<float4*>(rd.d_colors)[pix] = color;
}

#ifdef VH_RENDER_STATS
#define VH_STAT_DECL const long long statT0 = clock64(); const long long statR0 = wall_clock64(); float statTri = 0.0f, statIter = 0.0f; float statCyc[3] = { 0.0f, 0.0f, 0.0f };
#define VH_STAT_ARGS , statTri, statIter, statCyc
#define VH_STAT_STORE                                                                                                                   \
    {                                                                                                                                   \
        const uint32_t hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);              \
        reinterpret_cast<float4*>(rd.d_depth4)[pix] = make_float4(statCyc[0], statCyc[1], statCyc[2], 0.0f);                             \
        reinterpret_cast<float4*>(rd.d_normals)[pix] = make_float4((float)(clock64() - statT0), statTri, statIter, (float)(wall_clock64() - statR0)); \
        reinterpret_cast<float4*>(rd.d_colors)[pix] = make_float4((float)(uint32_t)(statR0 & 0xffffffll), (float)((hwid >> 8) & 0xfu),  \
            (float)((hwid >> 13) & 0x7u) + 8.0f * (float)((hwid >> 12) & 1u), (float)(xcc & 0xfu) * 4.0f + (float)((hwid >> 4) & 3u));   \
    }
#else
#define VH_STAT_DECL
#define VH_STAT_ARGS
#define VH_STAT_STORE
#endif

// One wave per 8x8-pixel tile, block pointers from the hash table: the reference fork's ray caster (no intervals).
template <bool GRADIENTS>
__global__ __launch_bounds__(256) void k_render_hash(VhHashData hd, VhHashParams hp, VhRayCastData rd,
                                                     VhDepthCameraParams cp, VhRayCastParams rp, HashMod hm)
{
    const uint32_t lane = lane_id();
    const uint32_t W = rp.m_width, H = rp.m_height;
    const uint32_t tilesX = (W + 7) / 8, tilesY = (H + 7) / 8;
    const uint32_t tile = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (tile >= tilesX * tilesY) return;
    const uint32_t x = (tile % tilesX) * 8 + (lane & 7), y = (tile / tilesX) * 8 + (lane >> 3);
    if (x >= W || y >= H) return;
    const size_t pix = (size_t)y * W + x;
    VH_STAT_DECL
    RayHit out;
    out.hit = false;
    HashLookup lk{ hd, hp, hm, {} };
    cache_init(lk.bc);
    uint32_t cost = 0u;
    march_ray<GRADIENTS>(lk, hd, hp, cp, rp, x, y, 0.0f, pinf(), 0u, 0.0f, out, cost VH_STAT_ARGS);
    store_ray(rd, cp, pix, x, y, out, GRADIENTS);
    VH_STAT_STORE
}

// One wave per 8x8-pixel tile with the tile's head and block list from k_interval_splat: the wave builds its block
// table in LDS, forms the tile's depth interval from the listed blocks and marches inside it.  The kernel is bound by
// VALU issue per SIMD: what counts is instructions per sample and even loads of the SIMDs.
template <bool GRADIENTS, uint32_t CAP>
VHD void render_tile(const VhHashData& hd, const VhHashParams& hp, const VhRayCastData& rd, const VhDepthCameraParams& cp, const VhRayCastParams& rp,
                     uint4* heads, const int4* lists, uint32_t cap, uint32_t* sched, uint32_t phase,
                     int (*tileTab)[2u * CAP * 12u])
{
    // CAP = blocks a tile's table takes: 64 (6 KB of LDS per wave, every tile of a 640x480 frame resident) or 128 (12 KB:
    // three workgroups per compute unit; for fine voxels, where a few tiles see more than 64 blocks and would
    // otherwise probe the hash table for every block the table could not hold)
    constexpr uint32_t kTileTabSlots = 2u * CAP;
    typedef TileLookup<kTileTabSlots> Lookup;
    static_assert(kTileSlotWords == 12u, "tileTab row size");
    const uint32_t lane = lane_id();
    const uint32_t W = rp.m_width, H = rp.m_height;
    const uint32_t tilesX = (W + 7) / 8, tilesY = (H + 7) / 8;
    const uint32_t nTiles = tilesX * tilesY;
    const uint32_t waveIdx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave)));
    // Launch order.  The kernel is as long as its busiest SIMD, and a SIMD's waves are the workgroups i, i+numCUs,
    // i+2 numCUs, ... in launch order.  k_interval_splat sorts the tiles by the cost the previous frame measured
    // (it differs little) and deals them to the workgroups so that the sums come out even.  The schedule carries the
    // phase it was made for: one that was not refreshed for this call is ignored (raster order).
    uint32_t tile = waveIdx, half = 0u; // half: 0 whole tile, 1 / 2 near / far half of a split tile (all four waves of a workgroup or none)
    if (sched && sched[0] == phase) {
        const uint2 e = reinterpret_cast<const uint2*>(sched + 4 + 4u * ((nTiles + 3u) / 4u))[waveIdx];
        tile = e.y == phase ? (e.x & 0xffffffu) : nTiles; // a slot the dealing left empty (last, partial workgroup)
        half = e.y == phase ? (e.x >> 24) : 0u;
    }
    if (tile >= nTiles) return;
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 41 // measurement build: every wave's life, read by tools/render_stamps.py
    const uint32_t stamp0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
    uint32_t stamp1 = 0u;
    uint4* const stampAt = reinterpret_cast<uint4*>(hd.d_hashCompactified) + (hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE) / 2u + 3u * waveIdx;
    uint32_t stampList = 0u;
#define VH_WAVE_STAMP(COST)                                                                                                                     \
    if (lane == 0) {                                                                                                                             \
        stampAt[0] = make_uint4(stamp0, stamp1, (uint32_t)__builtin_amdgcn_s_memrealtime(), tile | (half << 24));                               \
        stampAt[1] = make_uint4((COST), __builtin_amdgcn_s_getreg((31 << 11) | 4), __builtin_amdgcn_s_getreg((31 << 11) | 20), 0x57410000u | waveIdx); \
        stampAt[2] = make_uint4(stampList, 0u, 0u, 0u);                                                                                           \
    }
#else
#define VH_WAVE_STAMP(COST)
#endif
    int* tab = tileTab[threadIdx.x / kWave];
    // consume the head and re-arm it, so that no separate clear pass is needed
    const uint4 head = heads[tile];
    // With the large tables (three waves per SIMD: a trip to memory is not hidden by five other waves) the tile's list entries
    // are asked for at once, whatever the head will say of their number -- the list's CAP places exist whether they are used or
    // not: one trip instead of two before the table can be built (cfg3: 87.7 -> 84.6 us).  With the small tables the extra
    // 3 MB at the start of the launch cost more than the trip (cfg2 +1.0 us, cfg4 +1.3): there the entries wait for the head.
    constexpr uint32_t kPerLane = CAP / kWave; // list entries per lane
    constexpr bool kListAhead = CAP > (uint32_t)VH_TILE_LIST_CAPACITY;
    int4 ahead[kPerLane];
    if (kListAhead) {
#pragma unroll
        for (uint32_t k = 0; k < kPerLane; k++) {
            ahead[k] = make_int4(0, 0, 0, VH_FREE_ENTRY);
            if (lane + k * kWave < min(cap, CAP)) ahead[k] = lists[(size_t)tile * cap + lane + k * kWave];
        }
    }
    float tileZmin = __uint_as_float(head.x), tileZmax = __uint_as_float(head.y); // as splatted (lists == nullptr)
    if (lane == 0 && half == 0u) heads[tile] = make_uint4(0x7f800000u, 0u, 0u, 0u); // (a split tile: once both halves have read it)
    const uint32_t listed = min(head.z, min(cap, CAP));
    const bool complete = listed == head.z;
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 41
    stampList = head.z | (complete ? 0u : 0x10000u);
#endif
    // feedback for the host's choice of CAP: the longest list of the frame (only lists near the small capacity report)
    if (sched && lane == 0 && head.z > (uint32_t)VH_TILE_LIST_CAPACITY - 16u) atomicMax(&sched[1], head.z);
    for (uint32_t i = lane; i < kTileTabSlots; i += kWave) tab[i * kTileSlotWords + 3u] = VH_FREE_ENTRY;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int4 mine[kPerLane];
    uint32_t slot[kPerLane];
#pragma unroll
    for (uint32_t k = 0; k < kPerLane; k++) {
        const uint32_t i = lane + k * kWave;
        mine[k] = make_int4(0, 0, 0, VH_FREE_ENTRY);
        slot[k] = 0u;
        if (i < listed) {
            mine[k] = kListAhead ? ahead[k] : lists[(size_t)tile * cap + i];
            uint32_t h = Lookup::slot_of(mine[k].x, mine[k].y, mine[k].z);
            // claim a slot through its pointer word, then fill in the position (nobody reads it before the barrier)
            while (atomicCAS(&tab[h * kTileSlotWords + 3u], VH_FREE_ENTRY, mine[k].w) != VH_FREE_ENTRY) h = (h + 1u) & (kTileTabSlots - 1u);
            tab[h * kTileSlotWords + 0u] = mine[k].x; tab[h * kTileSlotWords + 1u] = mine[k].y; tab[h * kTileSlotWords + 2u] = mine[k].z;
            slot[k] = h;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (a short list: its 7 x listed probes dealt to the lanes, a probe each per round, instead of seven in a row in the lanes that
    // hold an entry while the others idle -- two rounds for the median list of 13 blocks)
    constexpr uint32_t kSpreadRounds = 3;
    const bool spread = kPerLane == 1u && complete && listed * 7u <= kSpreadRounds * kWave;
    if (spread) {
#pragma unroll
        for (uint32_t r = 0; r < kSpreadRounds; r++) {
            if (r * kWave < listed * 7u) { // (wave-uniform)
                const uint32_t t = lane + r * kWave;
                const uint32_t e = (t * 9363u) >> 16; // t / 7 for t < 448
                const uint32_t j = t - 7u * e + 1u;
                const uint32_t src = min(e, kWave - 1u);
                const int bx = __shfl(mine[0].x, (int)src), by = __shfl(mine[0].y, (int)src), bz = __shfl(mine[0].z, (int)src);
                const uint32_t home = (uint32_t)__shfl((int)slot[0], (int)src);
                if (e < listed) {
                    const int sl = Lookup::slot_find(tab, bx + (int)(j & 1u), by + (int)((j >> 1) & 1u), bz + (int)((j >> 2) & 1u));
                    tab[home * kTileSlotWords + 3u + j] = sl >= 0 ? tab[(uint32_t)sl * kTileSlotWords + 3u] : VH_FREE_ENTRY;
                }
            }
        }
    } else if (complete) {
        // pointers of the seven neighbours a sample in this block can straddle into
#pragma unroll
        for (uint32_t k = 0; k < kPerLane; k++) {
            if (lane + k * kWave < listed) {
                // the seven first probes in flight together (the slots are scattered: one probe nearly always settles it)
                int nb[8];
#pragma unroll
                for (uint32_t j = 1; j < 8u; j++) {
                    const int sl = Lookup::slot_find(tab, mine[k].x + (int)(j & 1u), mine[k].y + (int)((j >> 1) & 1u), mine[k].z + (int)((j >> 2) & 1u));
                    nb[j] = sl >= 0 ? tab[(uint32_t)sl * kTileSlotWords + 3u] : VH_FREE_ENTRY;
                }
#pragma unroll
                for (uint32_t j = 1; j < 8u; j++) tab[slot[k] * kTileSlotWords + 3u + j] = nb[j];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    if (cap != 0u) {
        // With lists the splat keeps the depth range of the blocks a full list could not take (head.x, head.y); the
        // range of the listed blocks is formed here (two atomics per tile and block less).  Camera depth is linear in
        // the world position, so over a (grown) box its extremes are sums of per-axis extremes; the same margins as
        // k_interval_splat.
        {
            const float vs = hp.m_virtualVoxelSize;
            const float growLo = (GRADIENTS ? 1.75f : 1.25f) * vs, growHi = (GRADIENTS ? 0.75f : 0.25f) * vs;
            const float* vm = rp.m_viewMatrix; // camera z of a world point: row 2
            float zlo = pinf(), zhi = minf();
#pragma unroll
            for (uint32_t k = 0; k < kPerLane; k++) {
                if (lane + k * kWave < listed) {
                    const int b3[3] = { mine[k].x, mine[k].y, mine[k].z };
                    float a = vm[11], b = vm[11];
#pragma unroll
                    for (int ax = 0; ax < 3; ax++) {
                        const float lo = (float)(b3[ax] * VH_SDF_BLOCK_SIZE) * vs - growLo, hi = (float)(b3[ax] * VH_SDF_BLOCK_SIZE + VH_SDF_BLOCK_SIZE) * vs + growHi;
                        const float p = vm[8 + ax] * lo, q = vm[8 + ax] * hi;
                        a += fminf(p, q);
                        b += fmaxf(p, q);
                    }
                    const float zs = 2e-3f * fmaxf(fabsf(a), fabsf(b)) + 0.5f * vs;
                    zlo = fminf(zlo, a - zs);
                    zhi = fmaxf(zhi, b + zs);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                zlo = fminf(zlo, __shfl_xor(zlo, off));
                zhi = fmaxf(zhi, __shfl_xor(zhi, off));
            }
            if (!complete) { // {+inf, 0} if nothing overflowed
                zlo = fminf(zlo, __uint_as_float(head.x));
                zhi = fmaxf(zhi, __uint_as_float(head.y));
            }
            tileZmin = fmaxf(zlo, 0.0f);  // an empty list leaves {+inf, -inf}: nothing to march
            tileZmax = zhi;
        }
    }
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 41
    stamp1 = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t x = (tile % tilesX) * 8 + (lane & 7), y = (tile / tilesX) * 8 + (lane >> 3);
    const bool inImage = x < W && y < H;
    const size_t pix = (size_t)y * W + x;
    VH_STAT_DECL
    RayHit out;
    out.hit = false;
    uint32_t cost = 0u;
    if (inImage && tileZmin <= tileZmax) { // else: no allocated block can be read by this tile's rays, every sample is invalid
        Lookup lk{ tab, complete, hd, hp };
        march_ray<GRADIENTS, (CAP > (uint32_t)VH_TILE_LIST_CAPACITY) || VH_PIPELINE_SMALL>(lk, hd, hp, cp, rp, x, y, tileZmin, tileZmax, half, 0.5f * (tileZmin + tileZmax), out, cost VH_STAT_ARGS);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cost = max(cost, (uint32_t)__shfl_xor((int)cost, off));
    if (half != 0u) {
        // the far half leaves its result in its (spent) table, the near half takes it where it found nothing itself
        int* mine = tab;
        int* other = tileTab[(threadIdx.x / kWave) ^ 1u];
        if (half == 2u) {
            mine[lane * 8u + 0u] = out.hit ? 1 : 0;
            mine[lane * 8u + 1u] = __float_as_int(out.alpha);
            mine[lane * 8u + 2u] = (int)out.color;
            if (GRADIENTS) { mine[lane * 8u + 3u] = __float_as_int(out.normal.x); mine[lane * 8u + 4u] = __float_as_int(out.normal.y); mine[lane * 8u + 5u] = __float_as_int(out.normal.z); }
            if (lane == 0) mine[64u * 8u] = (int)cost;
        }
        __syncthreads(); // all four waves of this workgroup are halves of split tiles (schedule_tiles)
        if (half == 2u) { VH_WAVE_STAMP(cost) return; }
        if (!out.hit && other[lane * 8u + 0u] != 0) {
            out.hit = true;
            out.alpha = __int_as_float(other[lane * 8u + 1u]);
            out.color = (uint32_t)other[lane * 8u + 2u];
            if (GRADIENTS) out.normal = mk3(__int_as_float(other[lane * 8u + 3u]), __int_as_float(other[lane * 8u + 4u]), __int_as_float(other[lane * 8u + 5u]));
        }
        cost += (uint32_t)other[64u * 8u];
        if (lane == 0) heads[tile] = make_uint4(0x7f800000u, 0u, 0u, 0u);
    }
    if (inImage) {
        store_ray(rd, cp, pix, x, y, out, GRADIENTS);
        VH_STAT_STORE
    }
    VH_WAVE_STAMP(cost)
    if (sched && lane == 0) sched[4 + tile] = min(cost / kCostClassWidth, kCostClasses - 1u); // plain store: nobody waits for it
}
#undef VH_WAVE_STAMP
#undef VH_STAT_DECL
#undef VH_STAT_ARGS
#undef VH_STAT_STORE

// Co-launch (CUDASceneRepHashSDF::integrateAhead): the workgroups behind the ray caster's own, `firstGroup` onwards,
// run the alloc pass of the NEXT frame (four 8x8 pixel tiles each), and the workgroups behind computeNormals' own run
// its compactify pass.  One launch instead of two streams: the extra workgroups are dispatched as the ray caster's
// waves retire, so they fill the tail its few dearest tiles leave, and no event or second queue is involved.  A block
// allocated while rays are marched holds only unobserved voxels (weight 0), which a sample treats exactly like an
// absent block, and the ray caster reads the table through k_interval_splat's lists, made before this launch.
struct CoAlloc {
    VhHashData hd;
    VhHashParams hp; // of the frame being allocated (its pose)
    VhDepthCameraData cam;
    VhDepthCameraParams cp;
    const uint32_t* bitMask;
    uint2* packed;
    HashMod hm;
    int32_t lockToken;
    uint32_t firstGroup; // 0: nothing to co-launch
};

VHD bool co_alloc(const CoAlloc& job)
{
    if (job.firstGroup == 0u || blockIdx.x < job.firstGroup) return false;
    const uint32_t g = blockIdx.x - job.firstGroup;
    if (g == 0u && threadIdx.x == 0) job.hd.d_hashCompactifiedCounter[0] = 0; // as k_alloc
    alloc_tile(job.hd, job.hp, job.cam, job.cp, job.bitMask, job.lockToken, job.hm, job.packed, g * (256u / kWave) + (threadIdx.x / kWave));
    return true;
}

// small tables: all tiles of a 640x480 frame resident at once (4800 + 256 waves <= 6 x 4 x 256) -- the march is a
// latency chain, and a second round of waves costs as much as the first -- and the riders behind them find free
// slots sooner.  Six waves per SIMD (80 registers, 16 bytes of spill) measured against five (96): +1.5 % frames/s at
// 640x480, +5 % at 1920x1080; four (128 registers): -9 %.
#ifndef VH_RENDER_WAVES
#define VH_RENDER_WAVES 6
#endif
template <bool GRADIENTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VH_RENDER_WAVES, VH_RENDER_WAVES)))
void k_render(VhHashData hd, VhHashParams hp, VhRayCastData rd, VhDepthCameraParams cp, VhRayCastParams rp,
              uint4* heads, const int4* lists, uint32_t cap, uint32_t* sched, uint32_t phase, CoAlloc job)
{
    __shared__ int tileTab[256 / kWave][2u * VH_TILE_LIST_CAPACITY * kTileSlotWords];
    if (co_alloc(job)) return;
    render_tile<GRADIENTS, VH_TILE_LIST_CAPACITY>(hd, hp, rd, cp, rp, heads, lists, cap, sched, phase, tileTab);
}

// large tables (48 KB of LDS per workgroup: three workgroups per compute unit)
template <bool GRADIENTS>
__global__ __launch_bounds__(256)
void k_render_large(VhHashData hd, VhHashParams hp, VhRayCastData rd, VhDepthCameraParams cp, VhRayCastParams rp,
                    uint4* heads, const int4* lists, uint32_t cap, uint32_t* sched, uint32_t phase, CoAlloc job)
{
    __shared__ int tileTab[256 / kWave][2u * VH_TILE_LIST_CAPACITY_LARGE * kTileSlotWords];
    if (co_alloc(job)) return;
    render_tile<GRADIENTS, VH_TILE_LIST_CAPACITY_LARGE>(hd, hp, rd, cp, rp, heads, lists, cap, sched, phase, tileTab);
}

// Riders of computeNormals' launch.  They come FIRST in the grid (the schedule workgroup, the splat, the compactify
// pass, then the image): each of them is a chain of dependent trips to memory, and what starts first ends first.
struct CoCompactify {
    VhHashData hd;
    VhHashParams hp;
    VhDepthCameraParams cp;
    uint32_t groups; // 0: nothing to co-launch
};

// The interval splat of the NEXT render, for the pose of the frame being integrated.  It lists the table as it stands
// before that frame's pass over the voxels: what the pass frees stays listed with all-zero voxels (weight 0: read
// like an absent block), what the next alloc adds is not listed and is empty anyway.
struct CoSplat {
    VhRayCastParams rp; // view of the next render
    VhDepthCameraParams cp;
    uint4* heads;
    int4* lists;
    uint32_t* sched;
    uint32_t* feedback;
    uint32_t cap, phase, numCUs, nSplatGroups;
    uint32_t groups; // nSplatGroups (+ sched_groups() with a schedule); 0: nothing to co-launch
};

// The frame's pass over the voxels (integrate + starve + GC) as a third rider: the LAST workgroups of the launch, two launches
// a frame instead of three.  The pass needs the compactified list, which this launch's compactify workgroups make, and it
// frees blocks, which edits the table this launch's splat workgroups read.
//   * A workgroup of the pass polls ITS entry of the list (pass_rider_group): the entry is there as soon as the compactify
//     workgroup that found the block has written it, and its last word says so (rider_tag).  The compactify workgroups also
//     count themselves off when their part of the list is out (rider_done), and the last one raises flags (rider_wait): that
//     is how the pass learns that the list is complete -- for its length, which it needs afterwards -- and how a workgroup
//     beyond the end of the list learns that it has nothing to do.  (The first version waited for the flags before it read
//     anything: 1.4 us a frame more.)
//   * The frees and the splat need no order (free_block_cold): the splat lists every live block whatever happens, and a freed
//     one or not -- which is all the same to the ray cast it is made for.  (The first version had the splat workgroups count
//     themselves off too and the frees wait for them: 0.4 us a frame for nothing.)
// A wait cannot be in vain: the hardware starts a launch's workgroups in grid order, so what a workgroup of the pass waits
// for is running or done when it starts (and every wait gives up after ~1 s: VH_STATE_RIDER_GAVE_UP, never seen).
// Between workgroups of ONE launch the eight XCDs' L2s are not coherent, and a release / acquire pair at agent scope is the
// writer writing its whole L2 back and the reader dropping its own (buffer_wbl2 / buffer_inv per workgroup: measured, +30 us
// a frame).  The list is the only data that crosses: it is written through and read at the coherence point (list_store), the
// counters and flags are relaxed agent-scope atomics, and the writers wait for their stores before they count.
// What it buys (cfg2, tools/riders_stamps.py): the compactify workgroups are done 1.6 - 4.9 us into the launch (four trips to
// memory each, ~1 us on a busy machine), the pass's workgroups 6.4 - 8.9 us, beside the splat's 8.1; the launch alone ended at
// ~8 us and a launch of the pass would then ramp up and take its own ~6 us: 4.8 us less per frame at 640x480 (350 blocks in
// view), 5 us at 1080p (1200 blocks).  With more blocks than the launch has workgroups for the pass (2048) a workgroup takes
// several, one after the other: at 2500 - 5800 blocks (cfg3) that costs 1.4 us a frame more than it saves, at 8600 (the dense
// scene) it saves 2.3 -- the scene keeps the pass in a launch of its own, whose wave-per-block shape is made for such lists.
struct CoIntegrate {
    FusedArgs args;
    uint32_t* done;          // VH_RIDER_DONE_WORDS words: flags, class counters, top counter (rider_done; never reset)
    uint32_t listExpected, listClassExpected;   // what the compactify workgroups' top counter / class counters read once this launch's are done: the flags are set to the first then
    uint32_t riderTag;       // the tag of this launch's list entries (rider_tag)
    uint32_t first;          // the pass's first workgroup in the grid
    uint32_t groups;         // 0: nothing to co-launch
};
struct NormalsKernargs { // (the argument block of k_compute_normals, for the offset of the pass's arguments in it)
    float4* out; const float4* in; uint32_t width, height; CoCompactify job; CoSplat splat; CoIntegrate integ;
};
// A compactify workgroup, the i-th of n, has made its part of the list and counts itself off.  Counting on one word would
// not do: same-address atomics are served one after the other, ~12 ns each, and a launch has up to 1 000 of them (measured
// with the splat's 2 000 workgroups counting too: the launch twice as long).  So the workgroups count in
// VH_RIDER_DONE_COUNTERS classes (i mod 32, a counter each, 128 bytes apart), the last of a class counts the class off on a
// top counter, and the last class raises the flags.  Nothing is ever reset -- the host keeps what each word reads when all
// launches so far are done -- so every class must grow by the same amount in every launch: n rounded up to a multiple of 32,
// the workgroups whose i + 32 falls into the padding counting for two (n > 32 there).  A stage of few
// workgroups counts on the top counter alone.
constexpr uint32_t kRiderFewGroups = 128; // a stage of up to this many workgroups counts on one word (the launcher's totals follow: rider_totals)
VHD void rider_done(const CoIntegrate& integ, const uint32_t i, const uint32_t n)
{
    if (integ.groups == 0u) return;
    // every thread's stores to the list have arrived (written through, list_store: waiting for them is all a release has to
    // do here)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x < kWave) {
        uint32_t* const words = integ.done; // flags, class counters, top counter
        const uint32_t padded = (n + VH_RIDER_DONE_COUNTERS - 1u) / VH_RIDER_DONE_COUNTERS * VH_RIDER_DONE_COUNTERS;
        const uint32_t add = (i + VH_RIDER_DONE_COUNTERS >= n && i + VH_RIDER_DONE_COUNTERS < padded) ? 2u : 1u;
        uint32_t last = 0;
        if (threadIdx.x == 0) {
            if (n <= kRiderFewGroups) { // few enough to count on the top counter itself: a trip to memory less before the flags go up
                last = __hip_atomic_fetch_add(&words[2u * VH_RIDER_DONE_COUNTERS * 32u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == integ.listExpected ? 1u : 0u;
            } else {
                const uint32_t before = __hip_atomic_fetch_add(&words[(VH_RIDER_DONE_COUNTERS + i % VH_RIDER_DONE_COUNTERS) * 32u], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + add == integ.listClassExpected)
                    last = __hip_atomic_fetch_add(&words[2u * VH_RIDER_DONE_COUNTERS * 32u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == integ.listExpected ? 1u : 0u;
            }
        }
        last = (uint32_t)__shfl((int)last, 0);
        if (last && threadIdx.x < (uint32_t)VH_RIDER_DONE_COUNTERS)
            __hip_atomic_store(&words[threadIdx.x * 32u], integ.listExpected, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// A workgroup of the pass as a rider: block `group` of the list, then (with more than `groups` blocks) group + groups, ...
// It does not wait for the list to be complete before it starts: it polls ITS entry, which is there as soon as the compactify
// workgroup that found the block has written it (the entry's tag says so: rider_tag, the last word list_store writes) -- on
// average a microsecond before the slowest compactify workgroup is done, and without the trip for the count and the flag's own
// way through memory (0.9 us).  The count it needs only afterwards: to know whether the list goes on beyond the launch's
// workgroups, and, in workgroup 0, for the host's mirror.
struct PassRiderShared {
    FusedShared<false> fused;
    int4 q;
    uint32_t have, count;
};
template <uint32_t KERNARG_OFFSET>
VHD void pass_rider_group(const CoIntegrate& integ, const uint32_t group, PassRiderShared& sh)
{
    const FusedArgs& args = integ.args;
    const VhHashEntry* const list = args.hd.d_hashCompactified;
    const uint32_t nEntries = args.hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
    const float thr = get_truncation(args.hp, args.cp.m_sensorDepthWorldMax);
    // 1. this workgroup's entry, or the end of the list (every lane of the wave reads the same words: one request each)
    if (threadIdx.x < kWave) {
        uint32_t have = 0u;
        if (group < nEntries) {
            const uint32_t* const flag = integ.done + (blockIdx.x % VH_RIDER_DONE_COUNTERS) * 32u;
            const uint64_t* const tagWord = reinterpret_cast<const uint64_t*>(&list[group]) + 3;
            for (uint32_t polls = 0; polls < (1u << 22); polls++) { // (an exit every wave reaches, as rider_wait)
                // both words in flight together: a poll is one trip to memory
                const uint64_t w = __hip_atomic_load(tagWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(w >> 32) == integ.riderTag) { have = 1u; break; }
                if ((int32_t)(f - integ.listExpected) >= 0) break; // the list is complete (step 2 looks again)
                __builtin_amdgcn_s_sleep(2);
            }
        }
        int4 q = make_int4(0, 0, 0, 0);
        if (have) q = list_quad<true>(&list[group]);
        if (threadIdx.x == 0) { sh.q = q; sh.have = have; }
    }
    __syncthreads();
    const bool early = sh.have != 0u;
    if (early) integrate_block_by_workgroup<true, KERNARG_OFFSET>(args, sh.fused, sh.q, group, thr);
    // 2. the whole list
    if (threadIdx.x < kWave) {
        uint32_t count = 0u;
        if (rider_wait(integ.done, integ.listExpected)) count = (uint32_t)__hip_atomic_load(args.hd.d_hashCompactifiedCounter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (threadIdx.x == 0) atomicAdd(&args.hd.d_state[VH_STATE_RIDER_GAVE_UP], 1u); // (what is left of its share stays undone)
        if (threadIdx.x == 0) sh.count = count;
    }
    __syncthreads();
    const uint32_t count = sh.count;
    // host-visible copy of the block count and the caller's tag (as integrate_fused_body)
    if (args.countMirror && group == 0u && threadIdx.x == 0) *reinterpret_cast<uint2*>(args.countMirror) = make_uint2(count, args.mirrorTag);
    for (uint32_t b = group; b < count; b += integ.groups) {
        if (b == group && early) continue;
        __syncthreads(); // (the previous block's reads of the shared words are done)
        integrate_block_by_workgroup<true, KERNARG_OFFSET>(args, sh.fused, list_quad<true>(&list[b]), b, thr);
    }
}

// computeNormalsDevice, DSC/CameraUtil.cu:669-697
__global__ __launch_bounds__(256) void k_compute_normals(float4* out, const float4* in, uint32_t width, uint32_t height, CoCompactify job, CoSplat splat, CoIntegrate integ)
{
    __shared__ union RiderShared { SplatShared splat; CompactShared compact; PassRiderShared pass; } shared;
    SplatShared& sh = shared.splat;
    uint32_t g = blockIdx.x;
    if (integ.groups != 0u && g >= integ.first) {
        static_assert(kRiderFewGroups >= VH_RIDER_DONE_COUNTERS && VH_RIDER_DONE_COUNTERS <= kWave, "one lane per flag");
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 42
        const uint32_t riderStamp0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
        uint4* const riderStampAt = reinterpret_cast<uint4*>(job.hd.d_hashCompactified) + (job.hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE) / 2u + blockIdx.x;
#endif
        pass_rider_group<(uint32_t)(offsetof(NormalsKernargs, integ) + offsetof(CoIntegrate, args))>(integ, g - integ.first, shared.pass);
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 42
        __syncthreads();
        if (threadIdx.x == 0) *riderStampAt = make_uint4(riderStamp0, (uint32_t)__builtin_amdgcn_s_memrealtime(), 5u, 0x5742u);
#endif
        return;
    }
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 42 // measurement build: every workgroup's life by kind, read by tools/riders_stamps.py
    const uint32_t stamp0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
    uint4* const stampAt = reinterpret_cast<uint4*>(job.hd.d_hashCompactified) + (job.hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE) / 2u + blockIdx.x;
#define VH_GROUP_STAMP(KIND) { __syncthreads(); if (threadIdx.x == 0) *stampAt = make_uint4(stamp0, (uint32_t)__builtin_amdgcn_s_memrealtime(), (KIND), 0x5742u); }
#else
#define VH_GROUP_STAMP(KIND)
#endif
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT >= 31 && VH_KNOCKOUT <= 34 // measurement builds: the riders of this launch, one at a time
    {
        const uint32_t nSchedK = splat.groups - splat.nSplatGroups;
        const bool isSched = g < nSchedK, isCompact = !isSched && g - nSchedK < job.groups, isSplat = !isSched && !isCompact && g < splat.groups + job.groups;
        if (VH_KNOCKOUT == 31 && isSched) return;               // no schedule workgroups
        if (VH_KNOCKOUT == 32 && (isSched || isSplat)) return;  // no splat at all
        if (VH_KNOCKOUT == 33 && isCompact) return;             // no compactify
        if (VH_KNOCKOUT == 34 && !isSched && !isCompact && !isSplat) return; // no normals
    }
#endif
    // the schedule workgroups first (the longest chains), then compactify (the pass over the voxels, if it rides, waits for
    // its list), then the table's slices
    const uint32_t nSched = splat.groups - splat.nSplatGroups;
    if (g >= nSched && g - nSched < job.groups) {
        g -= nSched;
        if (integ.groups != 0u) compactify_group<true>(job.hd, job.hp, job.cp, g * blockDim.x + threadIdx.x, shared.compact, integ.riderTag);
        else compactify_group<false>(job.hd, job.hp, job.cp, g * blockDim.x + threadIdx.x, shared.compact);
        VH_GROUP_STAMP(3u)
        rider_done(integ, g, job.groups);
        return;
    }
    if (g >= nSched) g -= job.groups;
    if (g < splat.groups) {
        const uint32_t group = g < nSched ? splat.nSplatGroups + g : g - nSched;
        interval_splat_group(job.hd, job.hp, splat.cp, splat.rp, splat.heads, splat.lists, splat.cap, splat.sched, splat.phase, splat.numCUs,
                             splat.nSplatGroups, splat.feedback, group, sh);
        VH_GROUP_STAMP(group >= splat.nSplatGroups ? 1u : 2u)
        return;
    }
    g -= splat.groups;
    const uint32_t idx = g * blockDim.x + threadIdx.x;
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 42
    if (job.groups != 0u && idx < width * height) {
#else
    if (idx >= width * height) return;
    {
#endif
    const uint32_t x = idx % width, y = idx / width;
    const float mi = minf();
    float4 o = make_float4(mi, mi, mi, mi);
    if (x > 0 && x < width - 1 && y > 0 && y < height - 1) {
        const float4 CC = in[idx], PC = in[idx + width], CP = in[idx + 1], MC = in[idx - width], CM = in[idx - 1];
        if (CC.x != mi && PC.x != mi && CP.x != mi && MC.x != mi && CM.x != mi) {
            const F3 a = mk3(PC.x - MC.x, PC.y - MC.y, PC.z - MC.z), b = mk3(CP.x - CM.x, CP.y - CM.y, CP.z - CM.z);
            const F3 n = mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
            const float l = sqrtf(dot3(n, n));
            if (l > 0.0f) o = make_float4(n.x / -l, n.y / -l, n.z / -l, 1.0f);
        }
    }
    out[idx] = o;
    }
#if defined(VH_KNOCKOUT) && VH_KNOCKOUT == 42
    if (job.groups != 0u) VH_GROUP_STAMP(4u)
#endif
}
#undef VH_GROUP_STAMP

// ---------------------------------------------------------------------------
// marching cubes (DSC/CUDAMarchingCubesSDF.cu:65-143, DSC/MarchingCubesSDFUtil.h:154-311)
//
// Pass 1 lists the allocated hash entries (from the occupancy bits, not by reading all Ne entries); pass 2 runs one
// 512-thread workgroup per listed block, one voxel per thread.  A voxel's eight corner samples are trilinear
// interpolations whose 64 taps all lie in the block and a one-voxel shell around it, so the workgroup first stages
// that 10x10x10 neighbourhood (27 block look-ups, 8 KB) in LDS and every tap is an LDS read.  Tap coordinates and
// weights are computed with the reference's arithmetic (they decide bits of the output); a tap outside the staged
// shell -- it cannot happen for finite coordinates, but nothing here relies on that -- takes the global path.
// Triangles are appended with one atomic per wave; their order in the buffer differs from the reference's (and is
// not deterministic there either).
// ---------------------------------------------------------------------------

namespace mc_tables {
#define VH_MC_QUAL __device__ const
#include "../../include/vh_mc_tables.h"
#undef VH_MC_QUAL
} // namespace mc_tables

// getHashEntryForSDFBlockPos :424-468 with the ten slots of the bucket in flight together (lookup_ptr walks them one
// trip at a time, which is what a block that is NOT there costs whenever its bucket holds something else)
VHD int lookup_ptr_wide(const VhHashData& hd, const VhHashParams& hp, I3 blk)
{
    const uint32_t ne = hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
    const uint32_t h = hash_pos(hp.m_hashNumBuckets, blk);
    if (!bucket_maybe_occupied(hd, h)) return VH_FREE_ENTRY;
    const uint32_t base = h * VH_HASH_BUCKET_SIZE, idxLast = base + VH_HASH_BUCKET_SIZE - 1;
    int4 qs[VH_HASH_BUCKET_SIZE];
#pragma unroll
    for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) qs[j] = load_quad(&hd.d_hash[base + j]);
    uint32_t off = hd.d_hash[idxLast].offset;
#pragma unroll
    for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++)
        if (quad_matches(qs[j], blk)) return qs[j].w;
    uint32_t maxIter = 0;
#pragma unroll 1
    while (maxIter < hp.m_hashMaxCollisionLinkedListSize) { // the list behind the last slot
        if (off == 0) break;
        const uint32_t i = (idxLast + off) % ne;
        const int4 q = load_quad(&hd.d_hash[i]);
        if (quad_matches(q, blk)) return q.w;
        off = hd.d_hash[i].offset;
        maxIter++;
    }
    return VH_FREE_ENTRY;
}

__global__ void k_mc_reset(VhMarchingCubesData d)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { d.d_numTriangles[0] = 0u; d.d_numOccupiedBlocks[0] = 0u; }
}

// extractIsoSurfacePass1Kernel :65-92 (its box test is commented out in the reference)
__global__ __launch_bounds__(256) void k_mc_pass1(VhHashData hd, VhHashParams hp, VhMarchingCubesData d)
{
    const uint32_t nWords = (hp.m_hashNumBuckets + 31) / 32;
    const uint32_t wordIdx = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bits = (wordIdx < nWords) ? hd.d_bucketBits[wordIdx] : 0u;
    while (__any(bits != 0u)) {
        const bool has = bits != 0u;
        const uint32_t bucket = wordIdx * 32u + (has ? (uint32_t)(__ffs((int)bits) - 1) : 0u);
        bits &= bits - 1u;
        int ptrs[VH_HASH_BUCKET_SIZE];
#pragma unroll
        for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) ptrs[j] = has ? hd.d_hash[(uint64_t)bucket * VH_HASH_BUCKET_SIZE + j].ptr : VH_FREE_ENTRY;
#pragma unroll
        for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) {
            const bool keep = ptrs[j] != VH_FREE_ENTRY;
            const uint64_t m = __ballot(keep);
            if (m) {
                const int leader = __ffsll((unsigned long long)m) - 1;
                uint32_t base = 0;
                if ((int)lane_id() == leader) base = atomicAdd(d.d_numOccupiedBlocks, (uint32_t)__popcll(m));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                if (keep) d.d_occupiedBlocks[base + (uint32_t)__popcll(m & lanemask_lt())] = bucket * VH_HASH_BUCKET_SIZE + j;
            }
        }
    }
}

constexpr int kMcTile = VH_SDF_BLOCK_SIZE + 2; // block plus a one-voxel shell

struct McTile {
    const uint2* vox; // LDS, kMcTile^3
    I3 base;          // voxel coordinates of tile cell (0,0,0) = block base - 1
    const VhHashData& hd;
    const VhHashParams& hp;
    // getVoxel(float3) :390-400: the zero voxel where there is no block
    VHD Vox voxel_at(F3 worldPos) const
    {
        const I3 v = world_to_vvp(hp.m_virtualVoxelSize, worldPos);
        const int tx = v.x - base.x, ty = v.y - base.y, tz = v.z - base.z;
        if ((unsigned)tx < (unsigned)kMcTile && (unsigned)ty < (unsigned)kMcTile && (unsigned)tz < (unsigned)kMcTile)
            return unpack_vox(vox[(tz * kMcTile + ty) * kMcTile + tx]);
        const int ptr = lookup_ptr(hd, hp, vvp_to_block(v));
        if (ptr == VH_FREE_ENTRY) return unpack_vox(make_uint2(0u, 0u));
        const VhVoxel* g = &hd.d_SDFBlocks[(uint32_t)ptr + (uint32_t)(local1(v.z) * 64 + local1(v.y) * 8 + local1(v.x))];
        return unpack_vox(*reinterpret_cast<const uint2*>(g));
    }
    // trilinearInterpolationSimpleFastFast, DSC/RayCastSDFUtil.h:97-116 (the colour it also forms is not used here)
    VHD bool trilinear(F3 pos, float& dist) const
    {
        const float oSet = hp.m_virtualVoxelSize;
        const F3 pd = mk3(pos.x - oSet / 2.0f, pos.y - oSet / 2.0f, pos.z - oSet / 2.0f);
        const float fx = pos.x / oSet, fy = pos.y / oSet, fz = pos.z / oSet;
        const float wx = fx - floorf(fx), wy = fy - floorf(fy), wz = fz - floorf(fz);
        float d = 0.0f;
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) {
            const uint32_t combo = (0x75634210u >> (4u * k)) & 7u; // reference tap order 000,100,010,001,110,011,101,111
            const bool bx = combo & 1u, by = combo & 2u, bz = combo & 4u;
            const Vox v = voxel_at(mk3(bx ? pd.x + oSet : pd.x + 0.0f, by ? pd.y + oSet : pd.y + 0.0f, bz ? pd.z + oSet : pd.z + 0.0f));
            if (v.weight() == 0u) return false;
            const float s = (bx ? wx : 1.0f - wx) * (by ? wy : 1.0f - wy) * (bz ? wz : 1.0f - wz);
            d += s * v.sdf;
        }
        dist = d;
        return true;
    }
};

// vertexInterp, DSC/MarchingCubesSDFUtil.h:237-262, with c1 == c2 == the voxel's colour as at every call site
VHD VhVertex mc_vertex(F3 p1, F3 p2, float d1, float d2, uint32_t cw)
{
    const float isolevel = 0.0f;
    const float cr = (float)(cw & 0xffu), cg = (float)((cw >> 8) & 0xffu), cb = (float)((cw >> 16) & 0xffu);
    VhVertex r;
    const bool first = fabsf(isolevel - d1) < 0.00001f, second = fabsf(isolevel - d2) < 0.00001f, flat = fabsf(d1 - d2) < 0.00001f;
    if (first || second || flat) {
        const F3 p = first ? p1 : (second ? p2 : p1);
        r.p[0] = p.x; r.p[1] = p.y; r.p[2] = p.z;
        r.c[0] = cr / 255.f; r.c[1] = cg / 255.f; r.c[2] = cb / 255.f;
        return r;
    }
    const float mu = (isolevel - d1) / (d2 - d1);
    r.p[0] = p1.x + mu * (p2.x - p1.x);
    r.p[1] = p1.y + mu * (p2.y - p1.y);
    r.p[2] = p1.z + mu * (p2.z - p1.z);
    r.c[0] = (cr + mu * 0.0f) / 255.f; // (float)(c2 - c1) is 0
    r.c[1] = (cg + mu * 0.0f) / 255.f;
    r.c[2] = (cb + mu * 0.0f) / 255.f;
    return r;
}

// extractIsoSurfacePass2Kernel :107-129 + extractIsoSurfaceAtPosition
__global__ __launch_bounds__(512) void k_mc_pass2(VhHashData hd, VhHashParams hp, VhMarchingCubesData data, uint32_t numBlocks)
{
    __shared__ uint2 sVox[kMcTile * kMcTile * kMcTile];
    __shared__ int sPtr[27];
    const uint32_t t = threadIdx.x;
    if (blockIdx.x >= numBlocks) return;
    const uint32_t idx = data.d_occupiedBlocks[blockIdx.x];
    const int4 q = load_quad(&hd.d_hash[idx]);
    if (q.w == VH_FREE_ENTRY) return; // block-uniform
    const I3 blk = mki3(q.x, q.y, q.z);
    const I3 base = mki3(blk.x * VH_SDF_BLOCK_SIZE - 1, blk.y * VH_SDF_BLOCK_SIZE - 1, blk.z * VH_SDF_BLOCK_SIZE - 1);
    if (t < 27u) {
        const int dx = (int)(t % 3u) - 1, dy = (int)((t / 3u) % 3u) - 1, dz = (int)(t / 9u) - 1;
        sPtr[t] = (dx == 0 && dy == 0 && dz == 0) ? q.w : lookup_ptr_wide(hd, hp, mki3(blk.x + dx, blk.y + dy, blk.z + dz));
    }
    __syncthreads();
    for (uint32_t i = t; i < (uint32_t)(kMcTile * kMcTile * kMcTile); i += blockDim.x) {
        const int tx = (int)(i % kMcTile), ty = (int)((i / kMcTile) % kMcTile), tz = (int)(i / (kMcTile * kMcTile));
        // shell cells belong to the neighbour block on that side
        const int nx = tx == 0 ? 0 : (tx == kMcTile - 1 ? 2 : 1), ny = ty == 0 ? 0 : (ty == kMcTile - 1 ? 2 : 1), nz = tz == 0 ? 0 : (tz == kMcTile - 1 ? 2 : 1);
        const int ptr = sPtr[(nz * 3 + ny) * 3 + nx];
        uint2 v = make_uint2(0u, 0u);
        if (ptr != VH_FREE_ENTRY) {
            const int lx = (tx + 7) & 7, ly = (ty + 7) & 7, lz = (tz + 7) & 7; // (t - 1) mod 8
            v = *reinterpret_cast<const uint2*>(&hd.d_SDFBlocks[(uint32_t)ptr + (uint32_t)(lz * 64 + ly * 8 + lx)]);
        }
        sVox[i] = v;
    }
    __syncthreads();

    const VhMarchingCubesParams mp = *data.d_params;
    const McTile tile{ sVox, base, hd, hp };
    // threadIdx of the reference's 8x8x8 block: x fastest
    const I3 pi = mki3(base.x + 1 + (int)(t & 7u), base.y + 1 + (int)((t >> 3) & 7u), base.z + 1 + (int)(t >> 6));
    const F3 worldPos = vvp_to_world(hp.m_virtualVoxelSize, pi);

    uint32_t nTri = 0;
    uint64_t triList = ~0ull;
    float dist[8] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }; // reference corner order 000,100,010,001,110,011,101,111
    const float P = hp.m_virtualVoxelSize / 2.0f, M = -P;
    bool ok = true;
    if ((mp.m_boxEnabled & 0xffu) == 1u) { // isInBoxAA :264-271
        if (worldPos.x < mp.m_minCorner[0] || worldPos.x > mp.m_maxCorner[0]) ok = false;
        if (worldPos.y < mp.m_minCorner[1] || worldPos.y > mp.m_maxCorner[1]) ok = false;
        if (worldPos.z < mp.m_minCorner[2] || worldPos.z > mp.m_maxCorner[2]) ok = false;
    }
    // The voxel itself is a tap of each of its eight corner samples (they lie half a voxel away, so their taps are
    // the voxel and its neighbours) whenever the coordinates are far from the float -> int cliffs: an unobserved
    // voxel (weight 0) then fails all eight, and most voxels of an allocated block are unobserved.
    const bool tame = abs(pi.x) < (1 << 20) && abs(pi.y) < (1 << 20) && abs(pi.z) < (1 << 20);
    if (ok && tame && (sVox[((int)(t >> 6) + 1) * kMcTile * kMcTile + ((int)((t >> 3) & 7u) + 1) * kMcTile + (int)(t & 7u) + 1].y >> 24) == 0u) ok = false;
#pragma unroll 1
    for (uint32_t k = 0; k < 8u; k++) {
        if (!__any(ok)) break; // wave-uniform: nothing left to sample in this 8x8 slab of voxels
        const uint32_t combo = (0x75634210u >> (4u * k)) & 7u;
        float dk = 0.0f;
        bool v = false;
        if (ok) v = tile.trilinear(mk3(worldPos.x + ((combo & 1u) ? P : M), worldPos.y + ((combo & 2u) ? P : M), worldPos.z + ((combo & 4u) ? P : M)), dk);
        ok = ok && v;
        // dist[k] with a run-time k would put the array in scratch
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) dist[j] = (j == k) ? dk : dist[j];
    }
    uint32_t cubeindex = 0;
    if (ok) {
        const float isolevel = 0.0f;
        if (dist[2] < isolevel) cubeindex += 1;   // 010
        if (dist[4] < isolevel) cubeindex += 2;   // 110
        if (dist[1] < isolevel) cubeindex += 4;   // 100
        if (dist[0] < isolevel) cubeindex += 8;   // 000
        if (dist[5] < isolevel) cubeindex += 16;  // 011
        if (dist[7] < isolevel) cubeindex += 32;  // 111
        if (dist[6] < isolevel) cubeindex += 64;  // 101
        if (dist[3] < isolevel) cubeindex += 128; // 001
        const float thres = mp.m_threshMarchingCubes;
        // the reference tests all 64 ordered pairs; the test is symmetric and a value passes against itself
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++)
#pragma unroll
            for (uint32_t l = k + 1u; l < 8u; l++) {
                if (dist[k] * dist[l] < 0.0f) { if (fabsf(dist[k]) + fabsf(dist[l]) > thres) ok = false; }
                else { if (fabsf(dist[k] - dist[l]) > thres) ok = false; }
            }
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) if (fabsf(dist[k]) > mp.m_threshMarchingCubes2) ok = false;
        const uint32_t edges = mc_tables::VH_MC_EDGE[cubeindex];
        if (edges == 0u || edges == 255u) ok = false;
    }
    if (ok) {
        triList = mc_tables::VH_MC_TRI[cubeindex];
        uint64_t l = triList;
        while ((l & 0xFull) != 0xFull) { nTri++; l >>= 12; }
    }

    // one atomic per wave: exclusive scan of the lanes' triangle counts
    const uint32_t lane = lane_id();
    uint32_t incl = nTri;
#pragma unroll
    for (int off = 1; off < (int)kWave; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
        if ((int)lane >= off) incl += up;
    }
    const uint32_t waveTotal = (uint32_t)__shfl((int)incl, (int)kWave - 1);
    if (waveTotal == 0u) return; // wave-uniform
    uint32_t waveBase = 0u;
    if (lane == kWave - 1u) waveBase = atomicAdd(data.d_numTriangles, waveTotal);
    waveBase = (uint32_t)__shfl((int)waveBase, (int)kWave - 1);
    uint32_t at = waveBase + incl - nTri;
    if (nTri == 0u) return;

    const Vox own = tile.voxel_at(worldPos);
    // Bourke's edges in the reference's corner names (vertlist, :205-216): endpoints as x|y<<1|z<<2, three bits each
    constexpr uint64_t kEdgeA = 0ull | (2ull << 0) | (3ull << 3) | (1ull << 6) | (0ull << 9) | (6ull << 12) | (7ull << 15) | (5ull << 18) | (4ull << 21) | (2ull << 24) | (3ull << 27) | (1ull << 30) | (0ull << 33);
    constexpr uint64_t kEdgeB = 0ull | (3ull << 0) | (1ull << 3) | (0ull << 6) | (2ull << 9) | (7ull << 12) | (5ull << 15) | (4ull << 18) | (6ull << 21) | (6ull << 24) | (7ull << 27) | (5ull << 30) | (4ull << 33);
    // corner bits x|y<<1|z<<2  ->  index in the reference's sample order 000,100,010,001,110,011,101,111
    constexpr uint32_t kOrder = 0u | (1u << 4) | (2u << 8) | (4u << 12) | (3u << 16) | (6u << 20) | (5u << 24) | (7u << 28);
    auto corner_dist = [&](uint32_t c) {
        const uint32_t k = (kOrder >> (4u * c)) & 7u;
        float r = dist[0];
#pragma unroll
        for (uint32_t j = 1; j < 8u; j++) r = (k == j) ? dist[j] : r;
        return r;
    };
    auto corner_pos = [&](uint32_t c) {
        return mk3(worldPos.x + ((c & 1u) ? P : M), worldPos.y + ((c & 2u) ? P : M), worldPos.z + ((c & 4u) ? P : M));
    };
    auto edge_vertex = [&](uint32_t e) {
        const uint32_t a = (uint32_t)(kEdgeA >> (3u * e)) & 7u, b = (uint32_t)(kEdgeB >> (3u * e)) & 7u;
        return mc_vertex(corner_pos(a), corner_pos(b), corner_dist(a), corner_dist(b), own.cw);
    };
#pragma unroll 1
    for (uint64_t l = triList; (l & 0xFull) != 0xFull; l >>= 12, at++) {
        if (at >= mp.m_maxNumTriangles) break; // appendTriangle :283-309 drops what does not fit; the host sees the full count
        VhTriangle tri;
        tri.v0 = edge_vertex((uint32_t)(l & 0xFull));
        tri.v1 = edge_vertex((uint32_t)((l >> 4) & 0xFull));
        tri.v2 = edge_vertex((uint32_t)((l >> 8) & 0xFull));
        data.d_triangles[at] = tri;
    }
}

// ---------------------------------------------------------------------------
// sensor pre-processing (DSC/CameraUtil.cu): the image kernels CUDARGBDAdapter::process and
// CUDARGBDSensor::process run between the sensor and integrate().  One pixel per lane, rows contiguous across the
// wave (the reference uses 16x16 tiles); all of them stream the image once.
// ---------------------------------------------------------------------------

// convertColorRawToFloatDevice :137-152 (RGBX bytes; black means "no colour")
__global__ __launch_bounds__(256) void k_convert_color_raw_to_float4(float4* out, const uint32_t* in, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = in[i];
    const uint32_t r = c & 0xffu, g = (c >> 8) & 0xffu, b = (c >> 16) & 0xffu, w = c >> 24;
    const float mi = minf();
    out[i] = (r == 0u && g == 0u && b == 0u) ? make_float4(mi, mi, mi, mi)
                                             : make_float4((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f, (float)(w / 255u));
}

// A sensor frame straight from (pinned, device-visible) host memory: the depth copied, the colour converted on the way
// (convertColorRawToFloatDevice above) -- one pass over the PCIe link, four pixels per 16-byte read, no staging copy of
// the raw colour.  Takes the place of the two uploads + the conversion of CUDARGBDAdapter::process
// (DSC/CUDARGBDAdapter.cpp:107-131) for a frame at adapter resolution.  The link, not the machine, sets the time
// (2.4 MB at ~45 GB/s = 55 us), so the grid is small and each lane keeps four reads in flight: the kernel runs beside
// the frame loop's own launches on another stream.
constexpr uint32_t kUploadInFlight = 4;
#ifndef VH_UPLOAD_GROUPS
#define VH_UPLOAD_GROUPS 16
#endif
// Measured, frames/s of the host-fed loop at 640x480 (the frame loop runs beside this kernel): 1 workgroup 1 860,
// 2: 3 310, 4: 5 710, 8: 9 440, 16: 11 700, 300: 9 300.  A read over the link takes ~3.3 us to come back, so few lanes
// cannot fill it (one workgroup moves 4.9 GB/s); and however small the kernel is, k_render runs two to three times
// slower while reads of host memory are in flight (36 -> 63 us beside 16 workgroups, 116 us beside ONE that takes
// 500 us): it is the uncached reads themselves, not the wave slots, that get in the way.  Sixteen workgroups is the
// best trade found; a copy engine (hipMemcpyAsync from pinned memory) would not touch the shader's memory path at all,
// but needs an event pair per frame on the frame loop's stream.
constexpr uint32_t kUploadGroups = VH_UPLOAD_GROUPS;
__global__ __launch_bounds__(256) void k_upload_frame(const uint4* hostDepth, const uint4* hostRGBX, uint4* depth, float4* color, uint32_t nQuads)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const float mi = minf();
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < nQuads; i0 += stride * kUploadInFlight) {
        u32x4 d[kUploadInFlight], c[kUploadInFlight];
#pragma unroll
        for (uint32_t k = 0; k < kUploadInFlight; k++) {
            const uint32_t i = i0 + k * stride;
            d[k] = u32x4{ 0u, 0u, 0u, 0u };
            c[k] = u32x4{ 0u, 0u, 0u, 0u };
            if (i < nQuads) {
                d[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(&hostDepth[i]));
                if (hostRGBX) c[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(&hostRGBX[i]));
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < kUploadInFlight; k++) {
            const uint32_t i = i0 + k * stride;
            if (i >= nQuads) continue;
            depth[i] = make_uint4(d[k].x, d[k].y, d[k].z, d[k].w);
            if (!hostRGBX) continue;
            const uint32_t cv[4] = { c[k].x, c[k].y, c[k].z, c[k].w };
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) {
                const uint32_t px = cv[j];
                const uint32_t r = px & 0xffu, g = (px >> 8) & 0xffu, b = (px >> 16) & 0xffu, w = px >> 24;
                color[4u * i + j] = (r == 0u && g == 0u && b == 0u) ? make_float4(mi, mi, mi, mi)
                                                                    : make_float4((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f, (float)(w / 255u));
            }
        }
    }
}

// bilinearInterpolationFloat :1071-1098 (invalid taps drop out of the weights)
VHD float bilinear_float(float x, float y, const float* in, uint32_t W, uint32_t H)
{
    const int px = (int)floorf(x), py = (int)floorf(y);
    const float alpha = x - (float)px, beta = y - (float)py;
    const float mi = minf();
    float s0 = 0.0f, w0 = 0.0f, s1 = 0.0f, w1 = 0.0f;
    if ((uint32_t)px < W && (uint32_t)py < H) { const float v = in[(uint32_t)py * W + (uint32_t)px]; if (v != mi) { s0 += (1.0f - alpha) * v; w0 += (1.0f - alpha); } }
    if ((uint32_t)(px + 1) < W && (uint32_t)py < H) { const float v = in[(uint32_t)py * W + (uint32_t)(px + 1)]; if (v != mi) { s0 += alpha * v; w0 += alpha; } }
    if ((uint32_t)px < W && (uint32_t)(py + 1) < H) { const float v = in[(uint32_t)(py + 1) * W + (uint32_t)px]; if (v != mi) { s1 += (1.0f - alpha) * v; w1 += (1.0f - alpha); } }
    if ((uint32_t)(px + 1) < W && (uint32_t)(py + 1) < H) { const float v = in[(uint32_t)(py + 1) * W + (uint32_t)(px + 1)]; if (v != mi) { s1 += alpha * v; w1 += alpha; } }
    const float p0 = s0 / w0, p1 = s1 / w1;
    float ss = 0.0f, ww = 0.0f;
    if (w0 > 0.0f) { ss += (1.0f - beta) * p0; ww += (1.0f - beta); }
    if (w1 > 0.0f) { ss += beta * p1; ww += beta; }
    return ww > 0.0f ? ss / ww : mi;
}

VHD float4 f4_scale(float a, float4 v) { return make_float4(a * v.x, a * v.y, a * v.z, a * v.w); }
VHD float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
VHD float4 f4_div(float4 a, float b) { return make_float4(a.x / b, a.y / b, a.z / b, a.w / b); }

// bilinearInterpolationFloat4 :1136-1166
VHD float4 bilinear_float4(float x, float y, const float4* in, uint32_t W, uint32_t H)
{
    const int px = (int)floorf(x), py = (int)floorf(y);
    const float alpha = x - (float)px, beta = y - (float)py;
    const float mi = minf();
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    float w0 = 0.0f, w1 = 0.0f;
    auto tap = [&](int tx, int ty, float wgt, float4& s, float& w) {
        if ((uint32_t)tx < W && (uint32_t)ty < H) {
            const float4 v = in[(uint32_t)ty * W + (uint32_t)tx];
            if (v.x != mi && v.y != mi && v.z != mi) { s = f4_add(s, f4_scale(wgt, v)); w += wgt; }
        }
    };
    tap(px, py, 1.0f - alpha, s0, w0);
    tap(px + 1, py, alpha, s0, w0);
    tap(px, py + 1, 1.0f - alpha, s1, w1);
    tap(px + 1, py + 1, alpha, s1, w1);
    const float4 p0 = f4_div(s0, w0), p1 = f4_div(s1, w1);
    float4 ss = make_float4(0.f, 0.f, 0.f, 0.f);
    float ww = 0.0f;
    if (w0 > 0.0f) { ss = f4_add(ss, f4_scale(1.0f - beta, p0)); ww += (1.0f - beta); }
    if (w1 > 0.0f) { ss = f4_add(ss, f4_scale(beta, p1)); ww += beta; }
    return ww > 0.0f ? f4_div(ss, ww) : make_float4(mi, mi, mi, mi);
}

// resampleFloatMapDevice :1100-1118 / resampleFloat4MapDevice :1168-1186 (pixels whose nearest source pixel lies
// outside the source keep their old value, as in the reference)
template <class T>
__global__ __launch_bounds__(256) void k_resample(T* out, const T* in, uint32_t inW, uint32_t inH, uint32_t outW, uint32_t outH)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= outW * outH) return;
    const int x = (int)(i % outW), y = (int)(i / outW);
    const float scaleWidth = (float)(inW - 1) / (float)(outW - 1), scaleHeight = (float)(inH - 1) / (float)(outH - 1);
    const uint32_t xInput = (uint32_t)((float)x * scaleWidth + 0.5f), yInput = (uint32_t)((float)y * scaleHeight + 0.5f);
    if (xInput < inW && yInput < inH) {
        if constexpr (sizeof(T) == 4) out[i] = bilinear_float((float)x * scaleWidth, (float)y * scaleHeight, in, inW, inH);
        else out[i] = bilinear_float4((float)x * scaleWidth, (float)y * scaleHeight, in, inW, inH);
    }
}

// setInvalidFloatMapDevice :338-346
__global__ __launch_bounds__(256) void k_set_invalid_float(float* out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = minf();
}

// convertColorToIntensityFloatDevice :258-267
__global__ __launch_bounds__(256) void k_color_to_intensity(float* out, const float4* in, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = in[i];
    out[i] = 0.299f * c.x + 0.587f * c.y + 0.114f * c.z;
}

// convertDepthFloatToCameraSpaceFloat4Device :390-407
__global__ __launch_bounds__(256) void k_depth_to_camera_space(float4* out, const float* in, VhDepthCameraParams cp, uint32_t W, uint32_t H)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const float mi = minf();
    const float depth = in[i];
    float4 o = make_float4(mi, mi, mi, mi);
    if (depth != mi) {
        const F3 p = depth_to_skeleton(cp, i % W, i / W, depth);
        o = make_float4(p.x, p.y, p.z, 1.0f);
    }
    out[i] = o;
}

// gaussD :436-439 (float exp), gaussR :426-429 (double arithmetic as written)
VHD float gauss_d(float sigma, int x, int y) { return expf(-((float)(x * x + y * y) / (2.0f * sigma * sigma))); }
VHD double gauss_r(float sigma, float dist) { return exp(-(double)(dist * dist) / (2.0 * (double)sigma * (double)sigma)); }

// gaussFilterFloatMapDevice :555-593
__global__ __launch_bounds__(256) void k_gauss_filter_float(float* out, const float* in, float sigmaD, float sigmaR, uint32_t W, uint32_t H)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const int x = (int)(i % W), y = (int)(i / W);
    const int kernelRadius = (int)ceil(2.0 * (double)sigmaD);
    const float mi = minf();
    float sum = 0.0f, sumWeight = 0.0f;
    const float center = in[i];
    if (center != mi) {
        for (int m = x - kernelRadius; m <= x + kernelRadius; m++)
            for (int n = y - kernelRadius; n <= y + kernelRadius; n++)
                if (m >= 0 && n >= 0 && m < (int)W && n < (int)H) {
                    const float cur = in[(uint32_t)n * W + (uint32_t)m];
                    if (cur != mi && fabsf(center - cur) < sigmaR) {
                        const float weight = gauss_d(sigmaD, m - x, n - y);
                        sumWeight += weight;
                        sum += weight * cur;
                    }
                }
    }
    out[i] = sumWeight > 0.0f ? sum / sumWeight : mi;
}

// The same filter for radii up to kGaussMaxRadius with the neighbourhood and the weights in LDS: a workgroup owns a
// 32x8 tile of pixels, stages the tile plus its halo once and computes the (2r+1)^2 weights once instead of once per
// pixel (the expf is most of the per-pixel kernel's work).  Same taps in the same order, same weights: same sums.
constexpr int kGaussMaxRadius = 8, kGaussTileW = 32, kGaussTileH = 8;

__global__ __launch_bounds__(256) void k_gauss_filter_float_tiled(float* out, const float* in, float sigmaD, float sigmaR, int W, int H, int r)
{
    extern __shared__ float sGauss[];
    const int tw = kGaussTileW + 2 * r, th = kGaussTileH + 2 * r, side = 2 * r + 1;
    float* sTile = sGauss;              // tw x th, rows contiguous
    float* sWeight = sGauss + tw * th;  // side x side, [dx + r][dy + r]
    const int x0 = (int)blockIdx.x * kGaussTileW, y0 = (int)blockIdx.y * kGaussTileH;
    const float mi = minf();
    for (int i = (int)threadIdx.x; i < tw * th; i += 256) {
        const int gx = x0 - r + i % tw, gy = y0 - r + i / tw;
        sTile[i] = (gx >= 0 && gy >= 0 && gx < W && gy < H) ? in[(size_t)gy * W + gx] : mi; // outside the image: skipped like an invalid pixel
    }
    for (int i = (int)threadIdx.x; i < side * side; i += 256) sWeight[i] = gauss_d(sigmaD, i / side - r, i % side - r);
    __syncthreads();
    const int lx = (int)threadIdx.x % kGaussTileW, ly = (int)threadIdx.x / kGaussTileW;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= W || y >= H) return;
    float sum = 0.0f, sumWeight = 0.0f;
    const float center = sTile[(ly + r) * tw + lx + r];
    if (center != mi) {
        for (int dx = -r; dx <= r; dx++)      // m = x + dx outer, n = y + dy inner: the reference's order of summation
            for (int dy = -r; dy <= r; dy++) {
                const float cur = sTile[(ly + r + dy) * tw + lx + r + dx];
                // a tap outside the image holds MINF here; the reference skips it by its bounds test
                if (cur != mi && fabsf(center - cur) < sigmaR) {
                    const float weight = sWeight[(dx + r) * side + dy + r];
                    sumWeight += weight;
                    sum += weight * cur;
                }
            }
    }
    out[(size_t)y * W + x] = sumWeight > 0.0f ? sum / sumWeight : mi;
}

// gaussFilterFloat4MapDevice :611-651
__global__ __launch_bounds__(256) void k_gauss_filter_float4(float4* out, const float4* in, float sigmaD, float sigmaR, uint32_t W, uint32_t H)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const int x = (int)(i % W), y = (int)(i / W);
    const int kernelRadius = (int)ceil(2.0 * (double)sigmaD);
    const float mi = minf();
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    float sumWeight = 0.0f;
    const float4 center = in[i];
    if (center.x != mi) {
        for (int m = x - kernelRadius; m <= x + kernelRadius; m++)
            for (int n = y - kernelRadius; n <= y + kernelRadius; n++)
                if (m >= 0 && n >= 0 && m < (int)W && n < (int)H) {
                    const float4 cur = in[(uint32_t)n * W + (uint32_t)m];
                    if (cur.x != mi) {
                        const float dx = center.x - cur.x, dy = center.y - cur.y, dz = center.z - cur.z, dw = center.w - cur.w;
                        if (sqrtf(dx * dx + dy * dy + dz * dz + dw * dw) < sigmaR) { // length(float4), cutil_math.h
                            const float weight = gauss_d(sigmaD, m - x, n - y);
                            sumWeight += weight;
                            sum = f4_add(sum, f4_scale(weight, cur));
                        }
                    }
                }
    }
    out[i] = sumWeight > 0.0f ? f4_div(sum, sumWeight) : make_float4(mi, mi, mi, mi);
}

// bilateralFilterFloatMapDevice :446-483
__global__ __launch_bounds__(256) void k_bilateral_filter_float(float* out, const float* in, float sigmaD, float sigmaR, uint32_t W, uint32_t H)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const int x = (int)(i % W), y = (int)(i / W);
    const int kernelRadius = (int)ceil(2.0 * (double)sigmaD);
    const float mi = minf();
    float sum = 0.0f, sumWeight = 0.0f;
    const float center = in[i];
    float o = mi;
    if (center != mi) {
        for (int m = x - kernelRadius; m <= x + kernelRadius; m++)
            for (int n = y - kernelRadius; n <= y + kernelRadius; n++)
                if (m >= 0 && n >= 0 && m < (int)W && n < (int)H) {
                    const float cur = in[(uint32_t)n * W + (uint32_t)m];
                    if (cur != mi) {
                        const float weight = (float)((double)gauss_d(sigmaD, m - x, n - y) * gauss_r(sigmaR, cur - center));
                        sumWeight += weight;
                        sum += weight * cur;
                    }
                }
        if (sumWeight > 0.0f) o = sum / sumWeight;
    }
    out[i] = o;
}

// erodeDepthMapDevice :1632-1670
__global__ __launch_bounds__(256) void k_erode_depth(float* out, const float* in, int structureSize, int W, int H, float dThresh, float fracReq)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint32_t)(W * H)) return;
    const int x = (int)(idx % (uint32_t)W), y = (int)(idx / (uint32_t)W);
    const float mi = minf();
    uint32_t count = 0;
    const float oldDepth = in[idx];
    for (int i = -structureSize; i <= structureSize; i++)
        for (int j = -structureSize; j <= structureSize; j++)
            if (x + j >= 0 && x + j < W && y + i >= 0 && y + i < H) {
                const float depth = in[(y + i) * W + (x + j)];
                if (depth == mi || depth == 0.0f || fabsf(depth - oldDepth) > dThresh) count++;
            }
    const uint32_t sum = (uint32_t)((2 * structureSize + 1) * (2 * structureSize + 1));
    out[idx] = ((float)count / (float)sum >= fracReq) ? mi : oldDepth;
}

// ---------------------------------------------------------------------------
// projective ICP camera tracking: projectiveCorrespondencesKernel (DSC/CUDAImageHelper.cu:70-125),
// scanScanElementsCS + reductionSystemCPU (DSC/CUDABuildLinearSystem.cu:130-188, .cpp:52-92) and the 6x6 solve /
// delinearisation the reference does on the host with Eigen (DSC/CUDACameraTrackingMultiRes.cpp:186-253).
//
// The reference copies every linear system to the host, solves it there and uploads the next transform: up to 18
// blocking round trips per frame.  Here the transform, the residual history and the lost / early-out flags live in
// a VhIcpState on the device; every step is a kernel on the stream that reads and updates it, a step whose level has
// finished returns at once, and the host reads the result once per frame.
// ---------------------------------------------------------------------------

constexpr uint32_t kIcpWindow = 12;   // pixels a lane sums before the wave reduces (localWindowSize, .cpp:41)
constexpr uint32_t kIcpTerms = 30;    // 21 upper-triangle terms of A^T A, 6 of A^T b, residual, weight, count (ARRAY_SIZE)

__global__ void k_icp_begin(VhIcpState* st, const float* d_deltaEstimate)
{
    const uint32_t t = threadIdx.x;
    if (t < 16u) st->delta[t] = d_deltaEstimate[t];
    if (t == 0u) { st->lost = 0u; st->done = 0u; st->lastError = -1.0f; st->iterations = 0u; st->sumRegError = 0.0f; st->sumRegWeight = 0.0f; st->numCorr = 0u; st->matrixCondition = 0.0f; }
}

__global__ void k_icp_begin_level(VhIcpState* st)
{
    if (threadIdx.x == 0u) { st->done = 0u; st->lastError = -1.0f; }
}

// projectiveCorrespondencesKernel :70-125 (getBestCorrespondence1x1 = the target pixel itself)
__global__ __launch_bounds__(256) void k_icp_correspondences(const float4* input, const float4* inputNormals, const float4* target, const float4* targetNormals,
                                                             float4* outCorr, float4* outCorrNormals, uint32_t W, uint32_t H, float distThres,
                                                             float normalThres, float levelFactor, const VhIcpState* st, VhDepthCameraParams cp)
{
    if (st->lost || st->done) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const float mi = minf();
    float4 oc = make_float4(mi, mi, mi, mi), on = oc;
    const float4 p = input[i], n = inputNormals[i];
    if (p.x != mi && n.x != mi) {
        const F3 pt = mat_mul_p(st->delta, mk3(p.x, p.y, p.z)), nt = mat_mul_d(st->delta, mk3(n.x, n.y, n.z));
        // cameraToKinectScreenInt, DSC/DepthCameraUtil.h:74-85, then the division by the level factor (both truncate)
        int sx = f2i((pt.x * cp.fx / pt.z + cp.mx) + 0.5f), sy = f2i((pt.y * cp.fy / pt.z + cp.my) + 0.5f);
        sx = f2i((float)sx / levelFactor); sy = f2i((float)sy / levelFactor);
        if (sx >= 0 && sy >= 0 && sx < (int)W && sy < (int)H) {
            const float4 tp = target[(uint32_t)sy * W + (uint32_t)sx];
            float4 tn = targetNormals[(uint32_t)sy * W + (uint32_t)sx];
            if (tp.x != mi && tn.x != mi) {
                const float dx = pt.x - tp.x, dy = pt.y - tp.y, dz = pt.z - tp.z;
                const float d = sqrtf(dx * dx + dy * dy + dz * dz);
                const float dNormal = nt.x * tn.x + nt.y * tn.y + nt.z * tn.z;
                if (d <= distThres && dNormal >= normalThres) {
                    oc = tp;
                    tn.w = fmaxf(0.0f, 0.5f * ((1.0f - d / distThres) + (1.0f - cam_to_proj_z(cp, pt.z)))); // weight of the pair
                    on = tn;
                }
            }
        }
    }
    outCorr[i] = oc;
    outCorrNormals[i] = on;
}

// scanScanElementsCS :130-188: lane x sums pixels [12x, 12x+12) in order, the 64 lanes of a wave are reduced with the
// reference's tree (+32, +16, ... +1) and lane 0 writes the wave's 30 terms.
__global__ __launch_bounds__(64) void k_icp_build_system(uint32_t W, uint32_t H, float* partials, const float4* input, const float4* corr,
                                                         const float4* corrNormals, const VhIcpState* st)
{
    if (st->lost || st->done) return;
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const float mi = minf();
    float acc[kIcpTerms];
#pragma unroll
    for (uint32_t k = 0; k < kIcpTerms; k++) acc[k] = 0.0f;
    for (uint32_t w = 0; w < kIcpWindow; w++) {
        const uint32_t idx = kIcpWindow * x + w;
        if (idx % W < W && idx / W < H) {
            const float4 tp = corr[idx], ip = input[idx], tn = corrNormals[idx];
            if (tp.x != mi && ip.x != mi && tn.x != mi) {
                const F3 q = mat_mul_p(st->delta, mk3(ip.x, ip.y, ip.z)); // moving point
                const F3 pT = mk3(tp.x, tp.y, tp.z), n = mk3(tn.x, tn.y, tn.z);
                const float weight = tn.w;
                // buildRowSystemMatrixPlane :70-82, buildRowRHSPlane :85-88
                const float row[6] = { n.x * q.y - n.y * q.x, n.z * q.x - n.x * q.z, n.y * q.z - n.z * q.y, -n.x, -n.y, -n.z };
                const float b = n.x * (q.x - pT.x) + n.y * (q.y - pT.y) + n.z * (q.z - pT.z);
                uint32_t at = 0;
#pragma unroll
                for (uint32_t r = 0; r < 6u; r++) {
#pragma unroll
                    for (uint32_t c = r; c < 6u; c++) acc[at + c - r] += weight * row[r] * row[c];
                    at += 6u - r;
                    acc[21u + r] += weight * row[r] * b;
                }
                const float dN = (pT.x - q.x) * n.x + (pT.y - q.y) * n.y + (pT.z - q.z) * n.z;
                acc[27] += weight * dN * dN;
                acc[28] += weight;
                acc[29] += 1.0f;
            }
        }
    }
    const uint32_t lane = lane_id();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (uint32_t k = 0; k < kIcpTerms; k++) {
            const float other = __shfl_down(acc[k], off);
            if ((int)lane < off) acc[k] += other;
        }
    }
    if (lane == 0u) {
#pragma unroll
        for (uint32_t k = 0; k < kIcpTerms; k++) partials[(size_t)blockIdx.x * kIcpTerms + k] = acc[k];
    }
}

// One wave: reductionSystemCPU (.cpp:52-92) over the wave partials in their order, then what computeBestRigidAlignment,
// delinearizeTransformation and align do with the system on the host (DSC/CUDACameraTrackingMultiRes.cpp:186-253,
// 306-318).  The 6x6 symmetric system is solved through its eigen-decomposition (cyclic Jacobi, double precision):
// x = V diag(1/l_i) V^T b with eigenvalues below 6 eps * l_max dropped, which is what Eigen's JacobiSVD::solve returns
// for a symmetric positive semi-definite matrix; the condition number is l_max / l_min.
__global__ __launch_bounds__(64) void k_icp_solve(VhIcpState* st, const float* partials, uint32_t nPartials, float angleThres, float distThres, float earlyOut, uint32_t lastInner)
{
    __shared__ float sTerms[kIcpTerms];
    if (st->lost || st->done) return;
    const uint32_t t = threadIdx.x;
    if (t < kIcpTerms) {
        // one term per lane, summed over the waves in their order (as reductionSystemCPU does); eight loads in flight
        float sum = 0.0f;
        uint32_t k = 0;
        for (; k + 8u <= nPartials; k += 8u) {
            float v[8];
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) v[j] = partials[(size_t)(k + j) * kIcpTerms + t];
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) sum += v[j];
        }
        for (; k < nPartials; k++) sum += partials[(size_t)k * kIcpTerms + t];
        sTerms[t] = sum;
    }
    __syncthreads();
    if (t != 0u) return;
    double A[6][6], b[6];
    {
        uint32_t at = 0;
        bool zero = true;
        for (uint32_t r = 0; r < 6u; r++) {
            for (uint32_t c = r; c < 6u; c++) {
                A[r][c] = A[c][r] = (double)sTerms[at + c - r];
                if (sTerms[at + c - r] != 0.0f) zero = false;
            }
            at += 6u - r;
            b[r] = (double)sTerms[21u + r];
        }
        st->sumRegError = sTerms[27];
        st->sumRegWeight = sTerms[28];
        st->numCorr = (uint32_t)sTerms[29];
        st->iterations += 1u;
        if (zero) { st->lost = 1u; return; } // ATA.isZero(): no correspondence at all
    }
    // cyclic Jacobi on A (symmetric): A -> diag, V accumulates the rotations
    double V[6][6];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < 6; i++) {
            diag += A[i][i] * A[i][i];
            for (int j = i + 1; j < 6; j++) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-26 * diag) break; // eigenvalues to ~1e-13 relative: far below what the float results can show
        for (int p = 0; p < 5; p++)
            for (int q = p + 1; q < 6; q++) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < 6; k++) { // columns p, q
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; k++) { // rows p, q
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    double lmax = 0.0, lmin = 1e300;
    for (int i = 0; i < 6; i++) { const double l = fabs(A[i][i]); lmax = l > lmax ? l : lmax; lmin = l < lmin ? l : lmin; }
    st->matrixCondition = (float)(lmax / lmin);
    double xs[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < 6; i++) {
        const double l = fabs(A[i][i]);
        if (l <= 6.0 * 1.1920928955078125e-7 * lmax) continue; // rank decision of JacobiSVD (threshold = diagSize * epsilon)
        double proj = 0.0;
        for (int k = 0; k < 6; k++) proj += V[k][i] * b[k];
        proj /= A[i][i];
        for (int k = 0; k < 6; k++) xs[k] += V[k][i] * proj;
    }
    // delinearizeTransformation :186-207: R = Rz(x0) Ry(x1) Rx(x2), t = x[3..5]; mean 0, meanStDev 1
    const float x0 = (float)xs[0], x1 = (float)xs[1], x2 = (float)xs[2];
    const float tx = (float)xs[3], ty = (float)xs[4], tz = (float)xs[5];
    const float cz = cosf(x0), sz = sinf(x0), cy = cosf(x1), sy = sinf(x1), cx = cosf(x2), sx = sinf(x2);
    float R[9] = { cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                   sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                   -sy, cy * sx, cy * cx };
    // checkRigidTransformation :176-185: angle of the rotation (Eigen::AngleAxisf) and length of the translation
    const float trace = R[0] + R[4] + R[8];
    const float angle = acosf(fminf(1.0f, fmaxf(-1.0f, 0.5f * (trace - 1.0f))));
    const float tnorm = sqrtf(tx * tx + ty * ty + tz * tz);
    if (!(angle <= angleThres) || !(tnorm <= distThres)) { st->lost = 1u; return; }
    // deltaTransform = t * deltaTransform
    float M[16] = { R[0], R[1], R[2], tx, R[3], R[4], R[5], ty, R[6], R[7], R[8], tz, 0.0f, 0.0f, 0.0f, 1.0f };
    float D[16];
    for (int k = 0; k < 16; k++) D[k] = st->delta[k];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            float acc = 0.0f;
            for (int k = 0; k < 4; k++) acc += M[4 * r + k] * D[4 * k + c];
            st->delta[4 * r + c] = acc;
        }
    // align :306-318, after the last inner iteration: leave the level when the residual stops changing
    if (lastInner) {
        if (fabsf(st->lastError - st->sumRegError) < earlyOut) st->done = 1u;
        st->lastError = st->sumRegError;
    }
}

// ---------------------------------------------------------------------------
// streaming (DSC/CUDASceneRepChunkGrid.cu)
// ---------------------------------------------------------------------------

// integrateFromGlobalHashPass1Kernel :27-74.  The double heap push of the
// reference's list branch (:58-64) is not reproduced: the element delete is
// the only push (DESIGN.md "Fenced reference defects").
__global__ __launch_bounds__(64) void k_stream_out_pass1(VhHashData hd, VhHashParams hp, uint32_t start, float radius,
                                                         float cx, float cy, float cz, uint32_t* outCounter,
                                                         VhSDFBlockDesc* out, uint32_t capacity, int32_t lockToken)
{
    const uint32_t ne = hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x + start;
    if (idx >= ne) return;
    VhHashEntry* e = &hd.d_hash[idx];
    const int4 q = load_quad(e);
    const uint32_t off = e->offset;
    const I3 pos = mki3(q.x, q.y, q.z);
    const F3 pw = block_to_world(hp.m_virtualVoxelSize, pos);
    const F3 df = mk3(pw.x - cx, pw.y - cy, pw.z - cz);
    const float d = sqrtf(dot3(df, df));
    if (q.w != VH_FREE_ENTRY && d >= radius) {
        bool emit = false;
        if (off != 0u || hash_pos(hp.m_hashNumBuckets, pos) != idx / VH_HASH_BUCKET_SIZE) {
            emit = delete_hash_entry_element(hd, hp, pos, lockToken);
        } else {
            append_heap(hd, (uint32_t)q.w / VH_SDF_BLOCK_VOXELS);
            delete_hash_entry(e);
            bucket_dec(hd, idx);
            emit = true;
        }
        if (emit) {
            const uint32_t addr = atomicAdd(outCounter, 1u);
            if (addr < capacity) {
                VhSDFBlockDesc dsc;
                dsc.pos[0] = q.x; dsc.pos[1] = q.y; dsc.pos[2] = q.z; dsc.ptr = q.w;
                out[addr] = dsc;
            }
        }
    }
}

// The chunk of a block, as the host's integrateInChunkGrid works it out (worldToChunks of the block's world position,
// linearizeChunkPos; DSC/CUDASceneRepChunkGrid.cpp:126-153, .h:570-598): the index of its bit in the bit mask, or
// 0xffffffff for a chunk outside the grid (the host drops such a block: "Chunk out of bounds")
VHD uint32_t chunk_bit_of_block(const VhHashParams& hp, I3 blk)
{
    const F3 pw = block_to_world(hp.m_virtualVoxelSize, blk);
    const F3 p = mk3(pw.x / hp.m_streamingVoxelExtents[0], pw.y / hp.m_streamingVoxelExtents[1], pw.z / hp.m_streamingVoxelExtents[2]);
    const I3 c = mki3(f2i(p.x + (float)signi(p.x) * 0.5f), f2i(p.y + (float)signi(p.y) * 0.5f), f2i(p.z + (float)signi(p.z) * 0.5f));
    const int qx = c.x - hp.m_streamingMinGridPos[0], qy = c.y - hp.m_streamingMinGridPos[1], qz = c.z - hp.m_streamingMinGridPos[2];
    if (qx < 0 || qy < 0 || qz < 0 || qx >= hp.m_streamingGridDimensions[0] || qy >= hp.m_streamingGridDimensions[1] || qz >= hp.m_streamingGridDimensions[2])
        return 0xffffffffu;
    return (uint32_t)(qz * hp.m_streamingGridDimensions[0] * hp.m_streamingGridDimensions[1] + qy * hp.m_streamingGridDimensions[0] + qx);
}

// k_stream_out_pass1 that also keeps the DEVICE's copy of the bit mask: the bit of every block's chunk is set here, where
// the block leaves, instead of by the host a round trip later (the host sets the same bit in its own copy when the block
// arrives; the frame's alloc pass, which reads the mask, then need not wait for the host)
__global__ __launch_bounds__(64) void k_stream_out_pass1_bits(VhHashData hd, VhHashParams hp, uint32_t start, float radius,
                                                              float cx, float cy, float cz, uint32_t* outCounter,
                                                              VhSDFBlockDesc* out, uint32_t capacity, int32_t lockToken, uint32_t* bitMask)
{
    const uint32_t ne = hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x + start;
    if (idx >= ne) return;
    VhHashEntry* e = &hd.d_hash[idx];
    const int4 q = load_quad(e);
    const uint32_t off = e->offset;
    const I3 pos = mki3(q.x, q.y, q.z);
    const F3 pw = block_to_world(hp.m_virtualVoxelSize, pos);
    const F3 df = mk3(pw.x - cx, pw.y - cy, pw.z - cz);
    const float d = sqrtf(dot3(df, df));
    if (q.w != VH_FREE_ENTRY && d >= radius) {
        bool emit = false;
        if (off != 0u || hash_pos(hp.m_hashNumBuckets, pos) != idx / VH_HASH_BUCKET_SIZE) {
            emit = delete_hash_entry_element(hd, hp, pos, lockToken);
        } else {
            append_heap(hd, (uint32_t)q.w / VH_SDF_BLOCK_VOXELS);
            delete_hash_entry(e);
            bucket_dec(hd, idx);
            emit = true;
        }
        if (emit) {
            const uint32_t addr = atomicAdd(outCounter, 1u);
            if (addr < capacity) {
                VhSDFBlockDesc dsc;
                dsc.pos[0] = q.x; dsc.pos[1] = q.y; dsc.pos[2] = q.z; dsc.ptr = q.w;
                out[addr] = dsc;
                const uint32_t bit = chunk_bit_of_block(hp, pos);
                if (bitMask && bit != 0xffffffffu) atomicOr(&bitMask[bit >> 5], 1u << (bit & 31u));
            }
        }
    }
}

// The same scan without the deletes: how many blocks of the part would the pass move out?  (A frame loop that knows its
// poses ahead asks this a frame early -- after that frame's alloc, the last pass that adds blocks -- and keeps the
// whole streaming step out of the next frame's launches when the answer is none: Reconstruction::frame.)
__global__ __launch_bounds__(256) void k_stream_out_probe(VhHashData hd, VhHashParams hp, uint32_t start, uint32_t n, float radius,
                                                          float cx, float cy, float cz, uint32_t* counter)
{
    const uint32_t ne = hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, idx = t + start;
    bool would = false;
    if (t < n && idx < ne) {
        const int4 q = load_quad(&hd.d_hash[idx]);
        const F3 pw = block_to_world(hp.m_virtualVoxelSize, mki3(q.x, q.y, q.z));
        const F3 df = mk3(pw.x - cx, pw.y - cy, pw.z - cz);
        would = q.w != VH_FREE_ENTRY && sqrtf(dot3(df, df)) >= radius;
    }
    const unsigned long long m = __ballot(would);
    if (m != 0ull && lane_id() == 0u) atomicAdd(counter, (uint32_t)__popcll(m));
}

// {*src, tag} into mapped host memory like k_publish_words, and the device word back to zero for the next use
__global__ void k_publish_and_clear(uint32_t* src, uint32_t* mapped, uint32_t tag)
{
    mapped[0] = *src;
    mapped[1] = 0u;
    *src = 0u;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    __hip_atomic_store(&mapped[2], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// integrateFromGlobalHashPass2Kernel :97-113 (copy block out, clear source)
__global__ __launch_bounds__(256) void k_stream_out_pass2(VhHashData hd, const VhSDFBlockDesc* descs, VhVoxel* out, uint32_t n)
{
    const uint32_t b = blockIdx.x;
    if (b >= n) return;
    const int ptr = __builtin_amdgcn_readfirstlane(descs[b].ptr);
    uint4* src = reinterpret_cast<uint4*>(&hd.d_SDFBlocks[(uint32_t)ptr]) + threadIdx.x;
    reinterpret_cast<uint4*>(out)[(size_t)b * 256 + threadIdx.x] = *src;
    *src = make_uint4(0u, 0u, 0u, 0u);
}

// k_stream_out_pass2 for a caller that has not read the count: as many workgroups as blocks there can be at most, each
// looks the count up
__global__ __launch_bounds__(256) void k_stream_out_pass2_counted(VhHashData hd, const VhSDFBlockDesc* descs, VhVoxel* out, const uint32_t* counter, uint32_t capacity)
{
    const uint32_t b = blockIdx.x, n = min(*counter, capacity);
    if (b >= n) return;
    const int ptr = __builtin_amdgcn_readfirstlane(descs[b].ptr);
    uint4* src = reinterpret_cast<uint4*>(&hd.d_SDFBlocks[(uint32_t)ptr]) + threadIdx.x;
    reinterpret_cast<uint4*>(out)[(size_t)b * 256 + threadIdx.x] = *src;
    *src = make_uint4(0u, 0u, 0u, 0u);
}

// {count of the pass, 0, tag} into mapped host memory, behind the pass's copies in the stream: a host thread that sees the
// tag finds the copied blocks in its staging buffer
__global__ void k_publish_count(const uint32_t* counter, uint32_t* mapped, uint32_t tag)
{
    mapped[0] = *counter;
    mapped[1] = 0u;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    __hip_atomic_store(&mapped[2], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The stream-in pass for a caller that does not read the heap counter back: chunkToGlobalHashPass1Kernel / Pass2Kernel
// with the counter looked up on the device, the chunk's bit cleared in the device's copy of the bit mask, and a third
// launch that takes the blocks off the heap and tells the host (mapped memory) how it went:
//   {blocks that found no slot, heap counter before the pass, 1 if the heap held too few free blocks (nothing was done), tag}
__global__ __launch_bounds__(64) void k_stream_in_pass1_dev(VhHashData hd, VhHashParams hp, uint32_t n, const VhSDFBlockDesc* descs, int32_t lockToken,
                                                            uint32_t* failed, uint32_t* bitMask, uint32_t chunkBit)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t heapCountPrev = hd.d_heapCounter[0];
    if (n > heapCountPrev + 1u) return; // (k_stream_in_commit reports it; the host puts the blocks back into its grid)
    if (i == 0u && bitMask && chunkBit != 0xffffffffu) atomicAnd(&bitMask[chunkBit >> 5], ~(1u << (chunkBit & 31u)));
    const uint32_t ptr = hd.d_heap[heapCountPrev - i] * VH_SDF_BLOCK_VOXELS;
    const VhSDFBlockDesc dsc = descs[i];
    if (!insert_hash_entry(hd, hp, mki3(dsc.pos[0], dsc.pos[1], dsc.pos[2]), (int)ptr, lockToken)) {
        atomicAdd(&hd.d_state[VH_STATE_INSERT_FAILED], 1u);
        failed[1u + atomicAdd(&failed[0], 1u)] = i;
    }
}
__global__ __launch_bounds__(256) void k_stream_in_pass2_dev(VhHashData hd, uint32_t n, const VhVoxel* blocks)
{
    const uint32_t b = blockIdx.x;
    if (b >= n) return;
    const uint32_t heapCountPrev = hd.d_heapCounter[0];
    if (n > heapCountPrev + 1u) return;
    const uint32_t ptr = hd.d_heap[heapCountPrev - b] * VH_SDF_BLOCK_VOXELS;
    *(reinterpret_cast<uint4*>(&hd.d_SDFBlocks[ptr]) + threadIdx.x) = reinterpret_cast<const uint4*>(blocks)[(size_t)b * 256 + threadIdx.x];
}
__global__ void k_stream_in_commit(VhHashData hd, uint32_t n, const uint32_t* failed, uint32_t* mapped, uint32_t tag)
{
    const uint32_t heapCountPrev = hd.d_heapCounter[0];
    const bool exhausted = n > heapCountPrev + 1u;
    if (!exhausted) hd.d_heapCounter[0] = heapCountPrev - n;
    else atomicAdd(&hd.d_state[VH_STATE_HEAP_UNDERFLOW], 1u);
    mapped[0] = failed[0];
    mapped[1] = heapCountPrev;
    mapped[3] = exhausted ? 1u : 0u;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    __hip_atomic_store(&mapped[2], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// chunkToGlobalHashPass1Kernel :143-160
__global__ __launch_bounds__(64) void k_stream_in_pass1(VhHashData hd, VhHashParams hp, uint32_t n, uint32_t heapCountPrev,
                                                        const VhSDFBlockDesc* descs, int32_t lockToken, uint32_t* failed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t ptr = hd.d_heap[heapCountPrev - i] * VH_SDF_BLOCK_VOXELS;
    const VhSDFBlockDesc dsc = descs[i];
    if (!insert_hash_entry(hd, hp, mki3(dsc.pos[0], dsc.pos[1], dsc.pos[2]), (int)ptr, lockToken)) {
        atomicAdd(&hd.d_state[VH_STATE_INSERT_FAILED], 1u);
        // which ones: failed[0] counts them, failed[1 ...] lists their indices (the caller takes them back)
        if (failed) failed[1u + atomicAdd(&failed[0], 1u)] = i;
    }
}

// chunkToGlobalHashPass2Kernel :181-189
__global__ __launch_bounds__(256) void k_stream_in_pass2(VhHashData hd, uint32_t n, uint32_t heapCountPrev, const VhVoxel* blocks)
{
    const uint32_t b = blockIdx.x;
    if (b >= n) return;
    const uint32_t ptr = hd.d_heap[heapCountPrev - b] * VH_SDF_BLOCK_VOXELS;
    *(reinterpret_cast<uint4*>(&hd.d_SDFBlocks[ptr]) + threadIdx.x) = reinterpret_cast<const uint4*>(blocks)[(size_t)b * 256 + threadIdx.x];
}

// ---------------------------------------------------------------------------
// utilities
// ---------------------------------------------------------------------------

struct SynthArgs {
    double spheres[4 * 8];
    int nSpheres;
    int inside;
    float T[16];
};

// analytic sphere scene in double, rounded once to float
__global__ __launch_bounds__(256) void k_synth(SynthArgs a, VhDepthCameraParams cp, float* depth, float4* color)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cp.m_imageWidth * cp.m_imageHeight) return;
    const uint32_t u = idx % cp.m_imageWidth, v = idx / cp.m_imageWidth;
    const double ox = (double)a.T[3], oy = (double)a.T[7], oz = (double)a.T[11];
    const double dx = ((double)u - (double)cp.mx) / (double)cp.fx;
    const double dy = ((double)v - (double)cp.my) / (double)cp.fy;
    const double wx = (double)a.T[0] * dx + (double)a.T[1] * dy + (double)a.T[2];
    const double wy = (double)a.T[4] * dx + (double)a.T[5] * dy + (double)a.T[6];
    const double wz = (double)a.T[8] * dx + (double)a.T[9] * dy + (double)a.T[10];
    const double aa = wx * wx + wy * wy + wz * wz;
    double bestT = 0.0;
    int best = -1;
    for (int s = 0; s < a.nSpheres; s++) {
        const double cx = a.spheres[4 * s + 0], cy = a.spheres[4 * s + 1], cz = a.spheres[4 * s + 2], r = a.spheres[4 * s + 3];
        const double ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
        const double b = ocx * wx + ocy * wy + ocz * wz;
        const double c = ocx * ocx + ocy * ocy + ocz * ocz - r * r;
        const double disc = b * b - aa * c;
        if (disc < 0.0) continue;
        const double sq = sqrt(disc);
        const double t = a.inside ? (-b + sq) / aa : (-b - sq) / aa;
        if (t > 0.0 && (best < 0 || t < bestT)) { bestT = t; best = s; }
    }
    const float mi = minf();
    if (best < 0) {
        depth[idx] = mi;
        color[idx] = make_float4(mi, mi, mi, mi);
    } else {
        const double cx = a.spheres[4 * best + 0], cy = a.spheres[4 * best + 1], cz = a.spheres[4 * best + 2], r = a.spheres[4 * best + 3];
        const double px = ox + bestT * wx, py = oy + bestT * wy, pz = oz + bestT * wz;
        double nx = (px - cx) / r, ny = (py - cy) / r, nz = (pz - cz) / r;
        if (a.inside) { nx = -nx; ny = -ny; nz = -nz; }
        depth[idx] = (float)bestT;
        color[idx] = make_float4((float)(0.5 + 0.5 * nx), (float)(0.5 + 0.5 * ny), (float)(0.5 + 0.5 * nz), 1.0f);
    }
}

// serial hash-operation interpreter (tests of the collision paths)
__global__ void k_debug_hash_ops(VhHashData hd, VhHashParams hp, const int32_t* ops, int32_t* results, uint32_t n)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int32_t token = 1;
    for (uint32_t i = 0; i < n; i++) {
        const int32_t op = ops[5 * i + 0];
        const I3 p = mki3(ops[5 * i + 1], ops[5 * i + 2], ops[5 * i + 3]);
        const int32_t arg = ops[5 * i + 4];
        int32_t r = 0;
        switch (op) {
        case VH_OP_ALLOC: r = alloc_block(hd, hp, p, token); break;
        case VH_OP_DELETE: r = delete_hash_entry_element(hd, hp, p, token) ? 1 : 0; break;
        case VH_OP_INSERT: r = insert_hash_entry(hd, hp, p, arg, token) ? 1 : 0; break;
        case VH_OP_LOOKUP: r = lookup_ptr(hd, hp, p); break;
        case VH_OP_NEW_PASS: token++; break;
        default: break;
        }
        // ops are "passes" of one launch here: drop this CU's L1 so that plain loads of the next op see what the
        // atomics of this one did at L2 (between real launches the hardware does that)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        results[i] = r;
    }
}

// checks div_exact against `/` and umod_fast against `%` on pseudo-random operands
__global__ __launch_bounds__(256) void k_check_fast_math(float b, HashMod hm, uint32_t n, uint32_t seed, uint32_t* mismatches)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // xorshift-multiply scramble of (seed, i)
    uint32_t s = (i + 1u) * 2654435761u ^ seed;
    s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
    uint32_t u = s * 747796405u + 2891336453u;
    // dividend: mostly scene-scale positions, some raw bit patterns (any finite magnitude)
    float a;
    if ((i & 7u) == 7u) {
        a = __uint_as_float(u);
        const uint32_t ex = (u >> 23) & 0xffu;
        if (ex == 0xffu || ex < 0x10u || ex > 0xe8u) a = (float)(int)u * 1.0e-6f; // keep a and a/b normal
    } else {
        a = ((float)(int)u) * (1.0f / 2147483648.0f) * (((i >> 3) & 1u) ? 400.0f : 8.0f);
    }
    const float rb = 1.0f / b;
    const float q0 = a / b, q1 = div_exact(a, b, rb);
    if (__float_as_uint(q0) != __float_as_uint(q1) && !(q0 == 0.0f && q1 == 0.0f)) atomicAdd(&mismatches[0], 1u);
    if ((s % hm.d) != umod_fast(s, hm)) atomicAdd(&mismatches[1], 1u);
    if ((u % hm.d) != umod_fast(u, hm)) atomicAdd(&mismatches[1], 1u);
    if (i < 64u) { // extremes of the unsigned range
        const uint32_t e = 0xffffffffu - i;
        if ((e % hm.d) != umod_fast(e, hm)) atomicAdd(&mismatches[1], 1u);
        if ((i % hm.d) != umod_fast(i, hm)) atomicAdd(&mismatches[1], 1u);
    }
}

// checks div_refined2 (integrate_block_certified's division) against `/` inside the ranges its callers certify:
// [0] the projection: divisor in [2^-20, 2^20], |numerator| in [2^-100, 2^60] or zero; [1] the blend: divisor an integer
// in 1..510, |numerator| in [2^-100, 2^90)
__global__ __launch_bounds__(256) void k_check_refined_division(uint32_t n, uint32_t seed, uint32_t* mismatches)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = (i + 1u) * 2654435761u ^ seed;
    s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
    const uint32_t u = s * 747796405u + 2891336453u, t = u * 2654435761u + 40503u;
    // sign and mantissa from the random words, exponents spread evenly over the certified ranges
    const uint32_t dExp = 127u - 20u + (s >> 8) % 41u, nExpP = 127u - 100u + (u >> 8) % 161u, nExpB = 127u - 100u + (t >> 8) % 190u;
    float d = __uint_as_float((dExp << 23) | (s & 0x7fffffu));
    if (dExp == 127u + 20u) d = 0x1p20f; // the closed end of the range
    float nP = __uint_as_float((u & 0x80000000u) | (nExpP << 23) | (u & 0x7fffffu));
    if (nExpP == 127u + 60u) nP = copysignf(0x1p60f, nP);
    if ((i & 1023u) == 0u) nP = (i & 1024u) ? 0.0f : -0.0f;
    if ((i & 1023u) == 1u) { d = 1.0f + (float)(i >> 10) * 0x1p-23f; nP = __uint_as_float(0x3fffffffu - (s & 0xffu)); } // mantissas near all-ones
    const float dB = (float)(1u + (s >> 3) % 510u);
    const float nB = __uint_as_float((t & 0x80000000u) | (nExpB << 23) | (t & 0x7fffffu));
    const f32x2 dd = (f32x2){ d, dB }, nn = (f32x2){ nP, nB };
    const f32x2 q = div_refined2(nn, dd, rcp_refined2(dd));
    const float wantP = nP / d, wantB = nB / dB;
    // (a zero numerator: the projection takes either zero -- "+ mx, + 0.5" cannot tell them apart)
    if (__float_as_uint(q.x) != __float_as_uint(wantP) && !(wantP == 0.0f && q.x == 0.0f)) atomicAdd(&mismatches[0], 1u);
    if (__float_as_uint(q.y) != __float_as_uint(wantB)) atomicAdd(&mismatches[1], 1u);
}

inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// multiply-shift constants of umod_fast for divisor d >= 2
inline HashMod make_hash_mod(uint32_t d)
{
    HashMod k;
    k.d = d;
    uint32_t l = 0;
    while ((1ull << l) < (uint64_t)d) l++;
    k.m = (uint32_t)((((1ull << 32) * ((1ull << l) - (uint64_t)d)) / d) + 1ull);
    k.sh = l - 1;
    return k;
}

} // namespace

// ---------------------------------------------------------------------------
// launcher-level C ABI
// ---------------------------------------------------------------------------

extern "C" {

int vh_reset(const VhHashData* hd, const VhHashParams* hp, vhStream_t stream)
{
    if (!hd || !hp || !hd->d_hash) return VH_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nb = hp->m_hashNumBuckets, ne = nb * VH_HASH_BUCKET_SIZE, nblk = hp->m_numSDFBlocks;
    VH_HIP(hipMemsetAsync(hd->d_SDFBlocks, 0, sizeof(VhVoxel) * (size_t)nblk * VH_SDF_BLOCK_VOXELS, s));
    VH_HIP(hipMemsetAsync(hd->d_bucketCount, 0, sizeof(uint32_t) * nb, s));
    VH_HIP(hipMemsetAsync(hd->d_bucketBits, 0, sizeof(uint32_t) * ((nb + 31) / 32), s));
    VH_HIP(hipMemsetAsync(hd->d_state, 0, sizeof(uint32_t) * VH_STATE_WORDS, s));
    VH_HIP(hipMemsetAsync(hd->d_hashDecision, 0, sizeof(int32_t) * ne, s));
    VH_HIP(hipMemsetAsync(hd->d_hashCompactifiedCounter, 0, sizeof(int32_t), s));
    k_reset_heap<<<cdiv(nblk, 256), 256, 0, s>>>(hd->d_heap, hd->d_heapCounter, nblk);
    k_reset_hash<<<cdiv(2ull * ne, 256), 256, 0, s>>>(hd->d_hash, hd->d_hashCompactified, ne);
    k_fill_i32<<<cdiv(nb, 256), 256, 0, s>>>(hd->d_hashBucketMutex, VH_FREE_ENTRY, nb);
    return vh_last_launch_error();
}

int vh_reset_bucket_mutex(const VhHashData* hd, const VhHashParams* hp, vhStream_t stream)
{
    if (!hd || !hp) return VH_ERR_BAD_ARGUMENT;
    k_fill_i32<<<cdiv(hp->m_hashNumBuckets, 256), 256, 0, (hipStream_t)stream>>>(hd->d_hashBucketMutex, VH_FREE_ENTRY, hp->m_hashNumBuckets);
    return vh_last_launch_error();
}

int vh_alloc(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraData* cam,
             const VhDepthCameraParams* cp, const uint32_t* d_bitMask, int32_t lockToken, vhStream_t stream)
{
    if (!hd || !hp || !cam || !cp || !cam->d_depthData) return VH_ERR_BAD_ARGUMENT;
    const uint32_t tiles = cdiv(cp->m_imageWidth, 8) * cdiv(cp->m_imageHeight, 8);
    if (tiles == 0) return VH_OK;
    if (hp->m_hashNumBuckets < 2) return VH_ERR_BAD_ARGUMENT;
    k_alloc<<<cdiv(tiles, 8), 512, 0, (hipStream_t)stream>>>(*hd, *hp, *cam, *cp, d_bitMask, lockToken, make_hash_mod(hp->m_hashNumBuckets), nullptr);
    return vh_last_launch_error();
}

int vh_alloc_job(VhFrameJob* job, vhStream_t stream)
{
    if (!job || !job->cam.d_depthData || !job->hashData.d_hash) return VH_ERR_BAD_ARGUMENT;
    const uint32_t tiles = cdiv(job->cp.m_imageWidth, 8) * cdiv(job->cp.m_imageHeight, 8);
    if (job->hashParams.m_hashNumBuckets < 2) return VH_ERR_BAD_ARGUMENT;
    job->allocLaunched = 1;
    if (tiles == 0) return VH_OK;
    k_alloc<<<cdiv(tiles, 8), 512, 0, (hipStream_t)stream>>>(job->hashData, job->hashParams, job->cam, job->cp, job->d_bitMask, job->lockToken,
                                                             make_hash_mod(job->hashParams.m_hashNumBuckets), reinterpret_cast<uint2*>(job->d_packedFrame));
    return vh_last_launch_error();
}

int vh_compactify_job(VhFrameJob* job, vhStream_t stream)
{
    if (!job || !job->hashData.d_hash) return VH_ERR_BAD_ARGUMENT;
    job->compactifyLaunched = 1;
    const uint32_t nWords = (job->hashParams.m_hashNumBuckets + 31) / 32;
    k_compactify<<<cdiv(nWords, 256), 256, 0, (hipStream_t)stream>>>(job->hashData, job->hashParams, job->cp);
    return vh_last_launch_error();
}

int vh_compactify(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp,
                  uint32_t* numOccupied, uint32_t flags, vhStream_t stream)
{
    if (!hd || !hp || !cp) return VH_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    if (!(flags & VH_COMPACT_COUNTER_IS_ZERO)) VH_HIP(hipMemsetAsync(hd->d_hashCompactifiedCounter, 0, sizeof(int32_t), s));
    const uint32_t nWords = (hp->m_hashNumBuckets + 31) / 32;
    k_compactify<<<cdiv(nWords, 256), 256, 0, s>>>(*hd, *hp, *cp);
    int err = vh_last_launch_error();
    if (err) return err;
    if (numOccupied) {
        VH_HIP(hipMemcpyAsync(numOccupied, hd->d_hashCompactifiedCounter, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        VH_HIP(hipStreamSynchronize(s));
    }
    return VH_OK;
}

int vh_integrate(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraData* cam,
                 const VhDepthCameraParams* cp, vhStream_t stream)
{
    if (!hd || !hp || !cam || !cp || !cam->d_depthData) return VH_ERR_BAD_ARGUMENT;
    if (hp->m_numOccupiedBlocks == 0) return VH_OK; // DSC/CUDASceneRepHashSDF.cu:501
    k_integrate<false><<<hp->m_numOccupiedBlocks, 256, 0, (hipStream_t)stream>>>(*hd, *hp, *cam, *cp, 0u, VH_LOCK_ENTRY, nullptr);
    return vh_last_launch_error();
}

static uint32_t device_num_cus();

int vh_integrate_fused(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraData* cam,
                       const VhDepthCameraParams* cp, uint32_t flags, int32_t lockToken, uint32_t* d_countMirror, uint32_t mirrorTag,
                       const void* d_packedFrame, vhStream_t stream)
{
    if (!hd || !hp || !cam || !cp || !cam->d_depthData) return VH_ERR_BAD_ARGUMENT;
    if (cp->m_imageWidth > 0xffffu || cp->m_imageHeight > 0xffffu) return VH_ERR_BAD_ARGUMENT;
    // persistent grid: the block count lives on the device, so no host read-back is needed (the kernel picks its shape
    // from the count)
    const uint32_t want = cdiv(hp->m_numSDFBlocks, 4), most = device_num_cus() * 8u;
    const uint32_t grid = want < most ? want : most;
    const uint2* packed = reinterpret_cast<const uint2*>(d_packedFrame);
    FusedArgs args;
    args.hd = *hd; args.hp = *hp; args.cam = *cam; args.cp = *cp;
    args.flags = flags; args.lockToken = lockToken; args.countMirror = d_countMirror; args.mirrorTag = mirrorTag;
    args.packed = reinterpret_cast<const uint2*>(packed);
    if (packed) VH_LAUNCH_TIMED(k_integrate_fused<true>, grid, 256, (hipStream_t)stream, args);
    else VH_LAUNCH_TIMED(k_integrate_fused<false>, grid, 256, (hipStream_t)stream, args);
    return vh_last_launch_error();
}

int vh_starve(const VhHashData* hd, const VhHashParams* hp, vhStream_t stream)
{
    if (!hd || !hp) return VH_ERR_BAD_ARGUMENT;
    if (hp->m_numOccupiedBlocks == 0) return VH_OK;
    k_starve<<<hp->m_numOccupiedBlocks, 256, 0, (hipStream_t)stream>>>(*hd, *hp);
    return vh_last_launch_error();
}

int vh_gc_identify(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp, vhStream_t stream)
{
    if (!hd || !hp || !cp) return VH_ERR_BAD_ARGUMENT;
    if (hp->m_numOccupiedBlocks == 0) return VH_OK;
    k_gc_identify<<<hp->m_numOccupiedBlocks, 256, 0, (hipStream_t)stream>>>(*hd, *hp, *cp);
    return vh_last_launch_error();
}

int vh_gc_free(const VhHashData* hd, const VhHashParams* hp, int32_t lockToken, vhStream_t stream)
{
    if (!hd || !hp) return VH_ERR_BAD_ARGUMENT;
    if (hp->m_numOccupiedBlocks == 0) return VH_OK;
    k_gc_free<<<hp->m_numOccupiedBlocks, 256, 0, (hipStream_t)stream>>>(*hd, *hp, lockToken);
    return vh_last_launch_error();
}

int vh_bind_input_depth_color_textures(const VhDepthCameraData* cam)
{
    (void)cam;
    return VH_OK;
}

int vh_render(const VhHashData* hd, const VhHashParams* hp, const VhRayCastData* rd,
              const VhDepthCameraParams* cp, const VhRayCastParams* rp, vhStream_t stream)
{
    if (!hd || !hp || !rd || !cp || !rp || !rd->d_depth) return VH_ERR_BAD_ARGUMENT;
    const uint32_t tiles = cdiv(rp->m_width, 8) * cdiv(rp->m_height, 8);
    if (tiles == 0) return VH_OK;
    if (hp->m_hashNumBuckets < 2) return VH_ERR_BAD_ARGUMENT;
    const HashMod hm = make_hash_mod(hp->m_hashNumBuckets);
    if (rp->m_useGradients) k_render_hash<true><<<cdiv(tiles, 4), 256, 0, (hipStream_t)stream>>>(*hd, *hp, *rd, *cp, *rp, hm);
    else k_render_hash<false><<<cdiv(tiles, 4), 256, 0, (hipStream_t)stream>>>(*hd, *hp, *rd, *cp, *rp, hm);
    return vh_last_launch_error();
}

static uint32_t device_num_cus() // of the current device (one device per process: INTEGRATION.md)
{
    static int numCUs = 0;
    if (numCUs == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&numCUs, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || numCUs <= 0)
            numCUs = 256;
    }
    return (uint32_t)numCUs;
}

size_t vh_render_schedule_bytes(uint32_t width, uint32_t height)
{
    const size_t tiles = (size_t)cdiv(width, 8) * cdiv(height, 8);
    // header, cost classes, launch slots (one per wave: the tiles, and a second one for each split tile)
    return (4u + 4u * ((tiles + 3u) / 4u) + 2u * 4u * ((tiles + kSplitTiles + 3u) / 4u)) * sizeof(uint32_t);
}

uint32_t vh_render_split_tiles(uint32_t width, uint32_t height) { return split_tiles(cdiv(width, 8) * cdiv(height, 8)); }

int vh_render_intervals_co(const VhHashData* hd, const VhHashParams* hp, const VhRayCastData* rd, const VhDepthCameraParams* cp,
                           const VhRayCastParams* rp, uint32_t* d_tileHeads, const VhTileBlock* d_tileBlocks, uint32_t tileCapacity,
                           uint32_t* d_schedule, uint32_t phase, VhFrameJob* fj, vhStream_t stream)
{
    if (!hd || !hp || !rd || !cp || !rp || !rd->d_depth || !d_tileHeads) return VH_ERR_BAD_ARGUMENT;
    const uint32_t tiles = cdiv(rp->m_width, 8) * cdiv(rp->m_height, 8);
    if (tiles == 0) return VH_OK;
    uint4* h = reinterpret_cast<uint4*>(d_tileHeads);
    const int4* l = reinterpret_cast<const int4*>(d_tileBlocks);
    const uint32_t cap = d_tileBlocks ? tileCapacity : 0u;
    // the capacity of the lists picks the table size: up to VH_TILE_LIST_CAPACITY the small tables, beyond it the large ones
    const bool large = cap > (uint32_t)VH_TILE_LIST_CAPACITY;
    uint32_t groups = cdiv(tiles + (d_schedule ? split_tiles(tiles) : 0u), 4);
    CoAlloc job;
    std::memset(&job, 0, sizeof(job));
    if (fj && !fj->allocLaunched && fj->cam.d_depthData && fj->hashData.d_hash && fj->hashParams.m_hashNumBuckets >= 2) {
        const uint32_t allocTiles = cdiv(fj->cp.m_imageWidth, 8) * cdiv(fj->cp.m_imageHeight, 8);
        job.hd = fj->hashData; job.hp = fj->hashParams; job.cam = fj->cam; job.cp = fj->cp;
        job.bitMask = fj->d_bitMask;
        job.packed = reinterpret_cast<uint2*>(fj->d_packedFrame);
        job.hm = make_hash_mod(fj->hashParams.m_hashNumBuckets);
        job.lockToken = fj->lockToken;
        job.firstGroup = groups;
        groups += cdiv(allocTiles, 4);
        fj->allocLaunched = 1;
    }
    const dim3 grid(groups);
    hipStream_t st = (hipStream_t)stream;
    if (rp->m_useGradients) {
        if (large) VH_LAUNCH_TIMED(k_render_large<true>, grid, 256, st, *hd, *hp, *rd, *cp, *rp, h, l, cap, d_schedule, phase, job);
        else VH_LAUNCH_TIMED(k_render<true>, grid, 256, st, *hd, *hp, *rd, *cp, *rp, h, l, cap, d_schedule, phase, job);
    } else {
        if (large) VH_LAUNCH_TIMED(k_render_large<false>, grid, 256, st, *hd, *hp, *rd, *cp, *rp, h, l, cap, d_schedule, phase, job);
        else VH_LAUNCH_TIMED(k_render<false>, grid, 256, st, *hd, *hp, *rd, *cp, *rp, h, l, cap, d_schedule, phase, job);
    }
    return vh_last_launch_error();
}

int vh_render_intervals(const VhHashData* hd, const VhHashParams* hp, const VhRayCastData* rd, const VhDepthCameraParams* cp,
                        const VhRayCastParams* rp, uint32_t* d_tileHeads, const VhTileBlock* d_tileBlocks, uint32_t tileCapacity,
                        uint32_t* d_schedule, uint32_t phase, vhStream_t stream)
{
    return vh_render_intervals_co(hd, hp, rd, cp, rp, d_tileHeads, d_tileBlocks, tileCapacity, d_schedule, phase, nullptr, stream);
}

int vh_ray_interval_clear(uint32_t* d_tileHeads, uint32_t width, uint32_t height, vhStream_t stream)
{
    if (!d_tileHeads) return VH_ERR_BAD_ARGUMENT;
    const uint32_t tiles = cdiv(width, 8) * cdiv(height, 8);
    if (tiles == 0) return VH_OK;
    k_interval_clear<<<cdiv(tiles, 256), 256, 0, (hipStream_t)stream>>>(reinterpret_cast<uint4*>(d_tileHeads), tiles);
    return vh_last_launch_error();
}

int vh_ray_interval_splat(const VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp, const VhRayCastParams* rp,
                          uint32_t* d_tileHeads, VhTileBlock* d_tileBlocks, uint32_t tileCapacity, uint32_t* d_schedule, uint32_t phase,
                          uint32_t* d_longestList, vhStream_t stream)
{
    if (!hd || !hp || !cp || !rp || !d_tileHeads) return VH_ERR_BAD_ARGUMENT;
    if (rp->m_width == 0 || rp->m_height == 0) return VH_OK;
    const uint32_t numCUs = d_schedule ? device_num_cus() : 256u;
    const uint32_t nWords = (hp->m_hashNumBuckets + 31) / 32;
    const uint32_t groups = cdiv(nWords, kSplatWordsPerGroup);
    const uint32_t nSched = d_schedule ? sched_groups(cdiv(rp->m_width, 8) * cdiv(rp->m_height, 8)) : 0u;
    k_interval_splat<<<groups + nSched, 256, 0, (hipStream_t)stream>>>(*hd, *hp, *cp, *rp, reinterpret_cast<uint4*>(d_tileHeads),
                                                                                 reinterpret_cast<int4*>(d_tileBlocks), d_tileBlocks ? tileCapacity : 0u,
                                                                                 d_schedule, phase, numCUs, groups, d_longestList);
    return vh_last_launch_error();
}

int vh_compute_normals_co2(float* d_output4, const float* d_input4, uint32_t width, uint32_t height, VhFrameJob* fj,
                           const VhRayCastParams* nextView, uint32_t* d_tileHeads, VhTileBlock* d_tileBlocks, uint32_t tileCapacity,
                           uint32_t* d_schedule, uint32_t phase, uint32_t* d_longestList, vhStream_t stream)
{
    if (!d_output4 || !d_input4) return VH_ERR_BAD_ARGUMENT;
    if (width * height == 0) return VH_OK;
    uint32_t groups = cdiv((uint64_t)width * height, 256);
    CoCompactify job;
    std::memset(&job, 0, sizeof(job));
    CoSplat sp;
    std::memset(&sp, 0, sizeof(sp));
    if (fj && fj->allocLaunched && !fj->compactifyLaunched && fj->hashData.d_hash) {
        job.hd = fj->hashData; job.hp = fj->hashParams; job.cp = fj->cp;
        job.groups = cdiv((fj->hashParams.m_hashNumBuckets + 31) / 32, 256);
        groups += job.groups;
        fj->compactifyLaunched = 1;
        if (nextView && d_tileHeads && nextView->m_width != 0 && nextView->m_height != 0) { // needs the job's table (job.hd / job.hp)
            sp.rp = *nextView; sp.cp = fj->cp;
            sp.heads = reinterpret_cast<uint4*>(d_tileHeads);
            sp.lists = reinterpret_cast<int4*>(d_tileBlocks);
            sp.sched = d_schedule; sp.feedback = d_longestList;
            sp.cap = d_tileBlocks ? tileCapacity : 0u;
            sp.phase = phase;
            sp.numCUs = d_schedule ? device_num_cus() : 256u;
            sp.nSplatGroups = cdiv((fj->hashParams.m_hashNumBuckets + 31) / 32, kSplatWordsPerGroup);
            sp.groups = sp.nSplatGroups + (d_schedule ? sched_groups(cdiv(nextView->m_width, 8) * cdiv(nextView->m_height, 8)) : 0u);
            groups += sp.groups;
        }
    }
    // the frame's pass over the voxels as a third rider (the last workgroups of the grid), if the scene has prepared it and its
    // list is made in this very launch
    CoIntegrate integ;
    std::memset(&integ, 0, sizeof(integ));
    if (fj && job.groups != 0u &&
        fj->fusedPrepared && !fj->fusedLaunched && fj->d_riderDone && fj->d_packedFrame && fj->cam.d_depthData &&
        fj->cp.m_imageWidth <= 0xffffu && fj->cp.m_imageHeight <= 0xffffu) {
        integ.args.hd = fj->hashData; integ.args.hp = fj->hashParams; integ.args.cam = fj->cam; integ.args.cp = fj->cp;
        integ.args.flags = fj->fusedFlags; integ.args.lockToken = fj->fusedLockToken;
        integ.args.countMirror = fj->d_countMirror; integ.args.mirrorTag = fj->mirrorTag;
        integ.args.packed = reinterpret_cast<const uint2*>(fj->d_packedFrame);
        integ.done = fj->d_riderDone;
        integ.riderTag = 0x80000000u | fj->mirrorTag; // (rider_tag of the frame's number)
        // (rider_done: a class of a stage grows by the stage's workgroups rounded up to 32s, over 32, and the top counter by 32;
        // a stage of few workgroups counts on the top counter alone)
        if (job.groups <= kRiderFewGroups) {
            fj->listDoneTotal += job.groups;
        } else {
            fj->listClassTotal += cdiv(job.groups, VH_RIDER_DONE_COUNTERS);
            fj->listDoneTotal += VH_RIDER_DONE_COUNTERS;
        }
        integ.listExpected = fj->listDoneTotal;
        integ.listClassExpected = fj->listClassTotal;
        integ.first = groups;
        const uint32_t want = cdiv(fj->hashParams.m_numSDFBlocks, 4), most = device_num_cus() * 8u;
        integ.groups = want < most ? want : most; // (as vh_integrate_fused)
        groups += integ.groups;
        fj->fusedLaunched = 1;
    }
    VH_LAUNCH_TIMED(k_compute_normals, groups, 256, (hipStream_t)stream, reinterpret_cast<float4*>(d_output4), reinterpret_cast<const float4*>(d_input4), width, height, job, sp, integ);
    return vh_last_launch_error();
}

int vh_compute_normals_co(float* d_output4, const float* d_input4, uint32_t width, uint32_t height, VhFrameJob* fj, vhStream_t stream)
{
    return vh_compute_normals_co2(d_output4, d_input4, width, height, fj, nullptr, nullptr, nullptr, 0u, nullptr, 0u, nullptr, stream);
}

int vh_compute_normals(float* d_output4, const float* d_input4, uint32_t width, uint32_t height, vhStream_t stream)
{
    return vh_compute_normals_co(d_output4, d_input4, width, height, nullptr, stream);
}

int vh_reset_marching_cubes(const VhMarchingCubesData* data, vhStream_t stream)
{
    if (!data || !data->d_numTriangles || !data->d_numOccupiedBlocks) return VH_ERR_BAD_ARGUMENT;
    k_mc_reset<<<1, 64, 0, (hipStream_t)stream>>>(*data);
    return vh_last_launch_error();
}

int vh_extract_iso_surface_pass1(const VhHashData* hd, const VhHashParams* hp, const VhMarchingCubesData* data, vhStream_t stream)
{
    if (!hd || !hp || !data || !data->d_occupiedBlocks) return VH_ERR_BAD_ARGUMENT;
    const uint32_t nWords = (hp->m_hashNumBuckets + 31) / 32;
    k_mc_pass1<<<cdiv(nWords, 256), 256, 0, (hipStream_t)stream>>>(*hd, *hp, *data);
    return vh_last_launch_error();
}

int vh_extract_iso_surface_pass2(const VhHashData* hd, const VhHashParams* hp, const VhMarchingCubesData* data,
                                 uint32_t numOccupiedBlocks, vhStream_t stream)
{
    if (!hd || !hp || !data || !data->d_params || !data->d_triangles) return VH_ERR_BAD_ARGUMENT;
    if (numOccupiedBlocks == 0) return VH_OK;
    k_mc_pass2<<<numOccupiedBlocks, 512, 0, (hipStream_t)stream>>>(*hd, *hp, *data, numOccupiedBlocks);
    return vh_last_launch_error();
}

#define VH_IMG_LAUNCH(n) cdiv((uint32_t)(n), 256u), 256, 0, (hipStream_t)stream

int vh_convert_color_raw_to_float4(float* d_output4, const uint8_t* d_inputRGBX, uint32_t width, uint32_t height, vhStream_t stream)
{
    if (!d_output4 || !d_inputRGBX) return VH_ERR_BAD_ARGUMENT;
    if (width * height == 0) return VH_OK;
    k_convert_color_raw_to_float4<<<VH_IMG_LAUNCH(width * height)>>>(reinterpret_cast<float4*>(d_output4), reinterpret_cast<const uint32_t*>(d_inputRGBX), width * height);
    return vh_last_launch_error();
}
int vh_upload_frame(const float* hostDepth, const uint8_t* hostRGBX, float* d_depth, float* d_color4, uint32_t width, uint32_t height, vhStream_t stream)
{
    if (!hostDepth || !d_depth || (hostRGBX && !d_color4)) return VH_ERR_BAD_ARGUMENT;
    const uint32_t n = width * height;
    if (n == 0) return VH_OK;
    // whole 16-byte reads: the last one may reach up to 12 bytes beyond n pixels, so the images must be padded to a
    // multiple of four pixels (640x480 and every even-by-even size is)
    if (n % 4u) return VH_ERR_BAD_ARGUMENT;
    const uint32_t nQuads = n / 4u;
    const uint32_t want = cdiv(nQuads, 256u * kUploadInFlight);
    k_upload_frame<<<want < kUploadGroups ? want : kUploadGroups, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const uint4*>(hostDepth), reinterpret_cast<const uint4*>(hostRGBX),
                                                                            reinterpret_cast<uint4*>(d_depth), reinterpret_cast<float4*>(d_color4), nQuads);
    return vh_last_launch_error();
}

int vh_resample_float_map(float* d_output, uint32_t outputWidth, uint32_t outputHeight, const float* d_input, uint32_t inputWidth, uint32_t inputHeight, vhStream_t stream)
{
    if (!d_output || !d_input || inputWidth == 0 || inputHeight == 0) return VH_ERR_BAD_ARGUMENT;
    if (outputWidth * outputHeight == 0) return VH_OK;
    k_resample<float><<<VH_IMG_LAUNCH(outputWidth * outputHeight)>>>(d_output, d_input, inputWidth, inputHeight, outputWidth, outputHeight);
    return vh_last_launch_error();
}
int vh_resample_float4_map(float* d_output4, uint32_t outputWidth, uint32_t outputHeight, const float* d_input4, uint32_t inputWidth, uint32_t inputHeight, vhStream_t stream)
{
    if (!d_output4 || !d_input4 || inputWidth == 0 || inputHeight == 0) return VH_ERR_BAD_ARGUMENT;
    if (outputWidth * outputHeight == 0) return VH_OK;
    k_resample<float4><<<VH_IMG_LAUNCH(outputWidth * outputHeight)>>>(reinterpret_cast<float4*>(d_output4), reinterpret_cast<const float4*>(d_input4), inputWidth, inputHeight, outputWidth, outputHeight);
    return vh_last_launch_error();
}
int vh_copy_float_map(float* d_output, const float* d_input, uint32_t width, uint32_t height, vhStream_t stream)
{
    if (!d_output || !d_input) return VH_ERR_BAD_ARGUMENT;
    VH_HIP(hipMemcpyAsync(d_output, d_input, sizeof(float) * (size_t)width * height, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return VH_OK;
}
int vh_copy_float4_map(float* d_output4, const float* d_input4, uint32_t width, uint32_t height, vhStream_t stream)
{
    if (!d_output4 || !d_input4) return VH_ERR_BAD_ARGUMENT;
    VH_HIP(hipMemcpyAsync(d_output4, d_input4, sizeof(float) * 4 * (size_t)width * height, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return VH_OK;
}
int vh_set_invalid_float_map(float* d_output, uint32_t width, uint32_t height, vhStream_t stream)
{
    if (!d_output) return VH_ERR_BAD_ARGUMENT;
    if (width * height == 0) return VH_OK;
    k_set_invalid_float<<<VH_IMG_LAUNCH(width * height)>>>(d_output, width * height);
    return vh_last_launch_error();
}
int vh_convert_color_to_intensity_float(float* d_output, const float* d_input4, uint32_t width, uint32_t height, vhStream_t stream)
{
    if (!d_output || !d_input4) return VH_ERR_BAD_ARGUMENT;
    if (width * height == 0) return VH_OK;
    k_color_to_intensity<<<VH_IMG_LAUNCH(width * height)>>>(d_output, reinterpret_cast<const float4*>(d_input4), width * height);
    return vh_last_launch_error();
}
int vh_convert_depth_float_to_camera_space_float4(float* d_output4, const float* d_input, const VhDepthCameraParams* cp, uint32_t width, uint32_t height, vhStream_t stream)
{
    if (!d_output4 || !d_input || !cp) return VH_ERR_BAD_ARGUMENT;
    if (width * height == 0) return VH_OK;
    k_depth_to_camera_space<<<VH_IMG_LAUNCH(width * height)>>>(reinterpret_cast<float4*>(d_output4), d_input, *cp, width, height);
    return vh_last_launch_error();
}
int vh_gauss_filter_float_map(float* d_output, const float* d_input, float sigmaD, float sigmaR, uint32_t width, uint32_t height, vhStream_t stream)
{
    if (!d_output || !d_input || d_output == d_input) return VH_ERR_BAD_ARGUMENT;
    if (width * height == 0) return VH_OK;
    const int r = (int)ceil(2.0 * (double)sigmaD);
    if (r >= 0 && r <= kGaussMaxRadius) {
        const size_t lds = sizeof(float) * ((size_t)(kGaussTileW + 2 * r) * (kGaussTileH + 2 * r) + (size_t)(2 * r + 1) * (2 * r + 1));
        const dim3 grid(cdiv(width, (uint32_t)kGaussTileW), cdiv(height, (uint32_t)kGaussTileH));
        k_gauss_filter_float_tiled<<<grid, 256, lds, (hipStream_t)stream>>>(d_output, d_input, sigmaD, sigmaR, (int)width, (int)height, r);
    } else {
        k_gauss_filter_float<<<VH_IMG_LAUNCH(width * height)>>>(d_output, d_input, sigmaD, sigmaR, width, height);
    }
    return vh_last_launch_error();
}
int vh_gauss_filter_float4_map(float* d_output4, const float* d_input4, float sigmaD, float sigmaR, uint32_t width, uint32_t height, vhStream_t stream)
{
    if (!d_output4 || !d_input4 || d_output4 == d_input4) return VH_ERR_BAD_ARGUMENT;
    if (width * height == 0) return VH_OK;
    k_gauss_filter_float4<<<VH_IMG_LAUNCH(width * height)>>>(reinterpret_cast<float4*>(d_output4), reinterpret_cast<const float4*>(d_input4), sigmaD, sigmaR, width, height);
    return vh_last_launch_error();
}
int vh_bilateral_filter_float_map(float* d_output, const float* d_input, float sigmaD, float sigmaR, uint32_t width, uint32_t height, vhStream_t stream)
{
    if (!d_output || !d_input || d_output == d_input) return VH_ERR_BAD_ARGUMENT;
    if (width * height == 0) return VH_OK;
    k_bilateral_filter_float<<<VH_IMG_LAUNCH(width * height)>>>(d_output, d_input, sigmaD, sigmaR, width, height);
    return vh_last_launch_error();
}
int vh_erode_depth_map(float* d_output, const float* d_input, int32_t structureSize, uint32_t width, uint32_t height, float dThresh, float fracReq, vhStream_t stream)
{
    if (!d_output || !d_input || d_output == d_input || structureSize < 0) return VH_ERR_BAD_ARGUMENT;
    if (width * height == 0) return VH_OK;
    k_erode_depth<<<VH_IMG_LAUNCH(width * height)>>>(d_output, d_input, structureSize, (int)width, (int)height, dThresh, fracReq);
    return vh_last_launch_error();
}
#undef VH_IMG_LAUNCH

int vh_icp_begin(VhIcpState* d_state, const float* d_deltaEstimate, vhStream_t stream)
{
    if (!d_state || !d_deltaEstimate) return VH_ERR_BAD_ARGUMENT;
    k_icp_begin<<<1, 64, 0, (hipStream_t)stream>>>(d_state, d_deltaEstimate);
    return vh_last_launch_error();
}
int vh_icp_begin_level(VhIcpState* d_state, vhStream_t stream)
{
    if (!d_state) return VH_ERR_BAD_ARGUMENT;
    k_icp_begin_level<<<1, 64, 0, (hipStream_t)stream>>>(d_state);
    return vh_last_launch_error();
}
int vh_icp_projective_correspondences(const float* d_input4, const float* d_inputNormals4, const float* d_target4, const float* d_targetNormals4,
                                      float* d_output4, float* d_outputNormals4, uint32_t width, uint32_t height, float distThres, float normalThres,
                                      float levelFactor, const VhIcpState* d_state, const VhDepthCameraParams* cp, vhStream_t stream)
{
    if (!d_input4 || !d_inputNormals4 || !d_target4 || !d_targetNormals4 || !d_output4 || !d_outputNormals4 || !d_state || !cp) return VH_ERR_BAD_ARGUMENT;
    if (width * height == 0) return VH_OK;
    k_icp_correspondences<<<cdiv(width * height, 256u), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(d_input4), reinterpret_cast<const float4*>(d_inputNormals4), reinterpret_cast<const float4*>(d_target4),
        reinterpret_cast<const float4*>(d_targetNormals4), reinterpret_cast<float4*>(d_output4), reinterpret_cast<float4*>(d_outputNormals4), width, height,
        distThres, normalThres, levelFactor, d_state, *cp);
    return vh_last_launch_error();
}
uint32_t vh_icp_num_partials(uint32_t width, uint32_t height) { return cdiv(width * height, 64u * kIcpWindow); }
int vh_icp_build_linear_system(uint32_t width, uint32_t height, float* d_partials, const float* d_input4, const float* d_corr4, const float* d_corrNormals4,
                               const VhIcpState* d_state, vhStream_t stream)
{
    if (!d_partials || !d_input4 || !d_corr4 || !d_corrNormals4 || !d_state) return VH_ERR_BAD_ARGUMENT;
    if (width * height == 0) return VH_OK;
    k_icp_build_system<<<vh_icp_num_partials(width, height), 64, 0, (hipStream_t)stream>>>(
        width, height, d_partials, reinterpret_cast<const float4*>(d_input4), reinterpret_cast<const float4*>(d_corr4),
        reinterpret_cast<const float4*>(d_corrNormals4), d_state);
    return vh_last_launch_error();
}
int vh_icp_solve(VhIcpState* d_state, const float* d_partials, uint32_t numPartials, float angleThres, float distThres, float earlyOutResidual,
                 int lastInnerIteration, vhStream_t stream)
{
    if (!d_state || !d_partials) return VH_ERR_BAD_ARGUMENT;
    k_icp_solve<<<1, 64, 0, (hipStream_t)stream>>>(d_state, d_partials, numPartials, angleThres, distThres, earlyOutResidual, lastInnerIteration ? 1u : 0u);
    return vh_last_launch_error();
}

int vh_stream_out_pass1(const VhHashData* hd, const VhHashParams* hp, uint32_t threadsPerPart, uint32_t start,
                        float radius, const float camPos[3], uint32_t* d_outputCounter, VhSDFBlockDesc* d_output,
                        uint32_t outputCapacity, int32_t lockToken, vhStream_t stream)
{
    if (!hd || !hp || !camPos || !d_outputCounter || !d_output) return VH_ERR_BAD_ARGUMENT;
    if (threadsPerPart == 0) return VH_OK; // DSC/CUDASceneRepChunkGrid.cu:81
    k_stream_out_pass1<<<cdiv(threadsPerPart, 64), 64, 0, (hipStream_t)stream>>>(*hd, *hp, start, radius, camPos[0], camPos[1], camPos[2],
                                                                                   d_outputCounter, d_output, outputCapacity, lockToken);
    return vh_last_launch_error();
}

int vh_stream_out_probe(const VhHashData* hd, const VhHashParams* hp, uint32_t threadsPerPart, uint32_t start, float radius,
                        const float camPos[3], uint32_t* d_counter, uint32_t* d_mapped, uint32_t tag, vhStream_t stream)
{
    if (!hd || !hp || !camPos || !d_counter || !d_mapped) return VH_ERR_BAD_ARGUMENT;
    if (threadsPerPart != 0) {
        // (the pass itself runs whole workgroups of 64: it looks at up to 63 entries beyond its part, and so must its probe --
        // the count is used as an upper bound)
        const uint32_t scanned = cdiv(threadsPerPart, 64) * 64u;
        k_stream_out_probe<<<cdiv(scanned, 256), 256, 0, (hipStream_t)stream>>>(*hd, *hp, start, scanned, radius, camPos[0], camPos[1], camPos[2], d_counter);
    }
    k_publish_and_clear<<<1, 1, 0, (hipStream_t)stream>>>(d_counter, d_mapped, tag);
    return vh_last_launch_error();
}

int vh_stream_out_pass2(const VhHashData* hd, const VhHashParams* hp, const VhSDFBlockDesc* d_descs,
                        VhVoxel* d_output, uint32_t nSDFBlocks, vhStream_t stream)
{
    (void)hp;
    if (!hd || !d_descs || !d_output) return VH_ERR_BAD_ARGUMENT;
    if (nSDFBlocks == 0) return VH_OK;
    k_stream_out_pass2<<<nSDFBlocks, 256, 0, (hipStream_t)stream>>>(*hd, d_descs, d_output, nSDFBlocks);
    return vh_last_launch_error();
}

int vh_stream_out_device(const VhHashData* hd, const VhHashParams* hp, uint32_t threadsPerPart, uint32_t start, float radius,
                         const float camPos[3], uint32_t* d_outputCounter, VhSDFBlockDesc* d_descs, VhVoxel* d_blocks,
                         uint32_t mostBlocks, int32_t lockToken, uint32_t* d_bitMask, vhStream_t stream)
{
    if (!hd || !hp || !camPos || !d_outputCounter || !d_descs || !d_blocks) return VH_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    VH_HIP(hipMemsetAsync(d_outputCounter, 0, sizeof(uint32_t), s));
    if (threadsPerPart == 0 || mostBlocks == 0) return VH_OK;
    k_stream_out_pass1_bits<<<cdiv(threadsPerPart, 64), 64, 0, s>>>(*hd, *hp, start, radius, camPos[0], camPos[1], camPos[2], d_outputCounter, d_descs,
                                                                     mostBlocks, lockToken, d_bitMask);
    k_stream_out_pass2_counted<<<mostBlocks, 256, 0, s>>>(*hd, d_descs, d_blocks, d_outputCounter, mostBlocks);
    return vh_last_launch_error();
}

int vh_publish_count(const uint32_t* d_counter, uint32_t* d_mapped, uint32_t tag, vhStream_t stream)
{
    if (!d_counter || !d_mapped) return VH_ERR_BAD_ARGUMENT;
    k_publish_count<<<1, 1, 0, (hipStream_t)stream>>>(d_counter, d_mapped, tag);
    return vh_last_launch_error();
}

int vh_stream_in_device(const VhHashData* hd, const VhHashParams* hp, uint32_t n, const VhSDFBlockDesc* d_descs, const VhVoxel* d_blocks,
                        int32_t lockToken, uint32_t* d_failed, uint32_t* d_bitMask, uint32_t chunkBit, uint32_t* d_mapped, uint32_t tag,
                        vhStream_t stream)
{
    if (!hd || !hp || !d_descs || !d_blocks || !d_failed || !d_mapped) return VH_ERR_BAD_ARGUMENT;
    if (hp->m_hashNumBuckets < 2) return VH_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    VH_HIP(hipMemsetAsync(d_failed, 0, sizeof(uint32_t), s));
    if (n != 0) {
        k_stream_in_pass1_dev<<<cdiv(n, 64), 64, 0, s>>>(*hd, *hp, n, d_descs, lockToken, d_failed, d_bitMask, chunkBit);
        k_stream_in_pass2_dev<<<n, 256, 0, s>>>(*hd, n, d_blocks);
    }
    k_stream_in_commit<<<1, 1, 0, s>>>(*hd, n, d_failed, d_mapped, tag);
    return vh_last_launch_error();
}

int vh_stream_in_pass1_report(const VhHashData* hd, const VhHashParams* hp, uint32_t n, uint32_t heapCountPrev,
                              const VhSDFBlockDesc* d_descs, int32_t lockToken, uint32_t* d_failed, vhStream_t stream)
{
    if (!hd || !hp || !d_descs) return VH_ERR_BAD_ARGUMENT;
    if (n == 0) return VH_OK;
    if (n > heapCountPrev + 1u) return VH_ERR_HEAP_EXHAUSTED;
    if (hp->m_hashNumBuckets < 2) return VH_ERR_BAD_ARGUMENT;
    k_stream_in_pass1<<<cdiv(n, 64), 64, 0, (hipStream_t)stream>>>(*hd, *hp, n, heapCountPrev, d_descs, lockToken, d_failed);
    return vh_last_launch_error();
}

int vh_stream_in_pass1(const VhHashData* hd, const VhHashParams* hp, uint32_t n, uint32_t heapCountPrev,
                       const VhSDFBlockDesc* d_descs, int32_t lockToken, vhStream_t stream)
{
    return vh_stream_in_pass1_report(hd, hp, n, heapCountPrev, d_descs, lockToken, nullptr, stream);
}

int vh_stream_in_pass2(const VhHashData* hd, const VhHashParams* hp, uint32_t n, uint32_t heapCountPrev,
                       const VhSDFBlockDesc* d_descs, const VhVoxel* d_blocks, vhStream_t stream)
{
    (void)hp; (void)d_descs;
    if (!hd || !d_blocks) return VH_ERR_BAD_ARGUMENT;
    if (n == 0) return VH_OK;
    if (n > heapCountPrev + 1u) return VH_ERR_HEAP_EXHAUSTED;
    k_stream_in_pass2<<<n, 256, 0, (hipStream_t)stream>>>(*hd, n, heapCountPrev, d_blocks);
    return vh_last_launch_error();
}

int vh_synth_frame(const double* h_spheres, int nSpheres, int inside, const float camToWorld[16],
                   const VhDepthCameraParams* cp, float* d_depth, float* d_color4, vhStream_t stream)
{
    if (!h_spheres || !camToWorld || !cp || !d_depth || !d_color4 || nSpheres < 0 || nSpheres > 8) return VH_ERR_BAD_ARGUMENT;
    SynthArgs a;
    for (int i = 0; i < 4 * nSpheres; i++) a.spheres[i] = h_spheres[i];
    a.nSpheres = nSpheres;
    a.inside = inside;
    for (int i = 0; i < 16; i++) a.T[i] = camToWorld[i];
    const uint64_t n = (uint64_t)cp->m_imageWidth * cp->m_imageHeight;
    if (n == 0) return VH_OK;
    k_synth<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(a, *cp, d_depth, reinterpret_cast<float4*>(d_color4));
    return vh_last_launch_error();
}

int vh_debug_hash_ops(const VhHashData* hd, const VhHashParams* hp, const int32_t* d_ops, int32_t* d_results,
                      uint32_t n, vhStream_t stream)
{
    if (!hd || !hp || !d_ops || !d_results) return VH_ERR_BAD_ARGUMENT;
    if (n == 0) return VH_OK;
    k_debug_hash_ops<<<1, 64, 0, (hipStream_t)stream>>>(*hd, *hp, d_ops, d_results, n);
    return vh_last_launch_error();
}

// {*src0, *src1, tag} into mapped host memory, the tag last and with system scope: a host that polls the tag reads
// the two words without a stream synchronisation or a copy (each costs a blocking driver call; the streaming passes of
// a frame need two such read-backs)
__global__ void k_publish_words(const uint32_t* src0, const uint32_t* src1, uint32_t* mapped, uint32_t tag)
{
    mapped[0] = src0 ? *src0 : 0u;
    mapped[1] = src1 ? *src1 : 0u;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    __hip_atomic_store(&mapped[2], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int vh_publish_words(const uint32_t* d_src0, const uint32_t* d_src1, uint32_t* d_mapped, uint32_t tag, vhStream_t stream)
{
    if (!d_mapped) return VH_ERR_BAD_ARGUMENT;
    k_publish_words<<<1, 1, 0, (hipStream_t)stream>>>(d_src0, d_src1, d_mapped, tag);
    return vh_last_launch_error();
}

int vh_debug_check_refined_division(uint32_t n, uint32_t seed, uint32_t* d_mismatches, vhStream_t stream)
{
    if (!d_mismatches) return VH_ERR_BAD_ARGUMENT;
    VH_HIP(hipMemsetAsync(d_mismatches, 0, 2 * sizeof(uint32_t), (hipStream_t)stream));
    if (n == 0) return VH_OK;
    k_check_refined_division<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(n, seed, d_mismatches);
    return vh_last_launch_error();
}

int vh_debug_check_fast_math(float divisor, uint32_t modulus, uint32_t n, uint32_t seed, uint32_t* d_mismatches, vhStream_t stream)
{
    if (!d_mismatches || modulus < 2 || !(divisor > 0.0f)) return VH_ERR_BAD_ARGUMENT;
    VH_HIP(hipMemsetAsync(d_mismatches, 0, 2 * sizeof(uint32_t), (hipStream_t)stream));
    if (n == 0) return VH_OK;
    k_check_fast_math<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(divisor, make_hash_mod(modulus), n, seed, d_mismatches);
    return vh_last_launch_error();
}

} // extern "C"
