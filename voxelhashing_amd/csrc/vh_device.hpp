// vh_device.hpp -- device-side functions of the voxel-hash TSDF path (gfx950).
//
// Behavioural contract: DSC/VoxelUtilHashSDF.h, DSC/DepthCameraUtil.h
// (DSC/ = /root/reference/DepthSensingCUDA/Source/); each function cites the
// lines whose semantics it must reproduce.  Numerics are IEEE fp32 with one
// rounding per operation: this translation unit MUST be compiled with
// -ffp-contract=off and without fast-math (block ids and pixel ids are
// float->int cliffs; see DESIGN.md "Numerics").
//
// Parameters travel as kernel arguments (wave-uniform, SGPR-resident), not as
// process-global __constant__ symbols.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vh_types.h"

namespace vhd {

struct F3 { float x, y, z; };
struct I3 { int x, y, z; };

#define VHD __device__ __forceinline__

VHD float minf() { return __int_as_float((int)0xff800000u); }
VHD float pinf() { return __int_as_float((int)0x7f800000u); }

VHD F3 mk3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
VHD I3 mki3(int x, int y, int z) { I3 r; r.x = x; r.y = y; r.z = z; return r; }
VHD float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// float -> int, round toward zero, saturating, NaN -> 0: v_cvt_i32_f32 has
// exactly the semantics of the cvt.rzi.s32.f32 the reference executes.
VHD int f2i(float v) { return __float2int_rz(v); }
// float -> uchar, saturating (cvt.rzi.u8.f32)
VHD uint8_t f2uc(float v) { return (uint8_t)__float2int_rz(fminf(fmaxf(v, 0.0f), 255.0f)); }

// cutil_math.h:31-33
VHD int signi(float v) { return (0.0f < v) - (v < 0.0f); }

// normalize, cutil_math.h:1207-1211 with rsqrtf = 1/sqrtf (cutil_math.h:81-84)
VHD F3 normalize3(F3 v)
{
    float invLen = 1.0f / sqrtf(dot3(v, v));
    return mk3(v.x * invLen, v.y * invLen, v.z * invLen);
}

// float4x4 * float3 (w = 1), DSC/cuda_SimpleMatrixUtil.h:900-907
VHD F3 mat_mul_p(const float* m, F3 v)
{
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * 1.0f,
               m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * 1.0f,
               m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * 1.0f);
}
// xyz of float4x4 * float4(v, 0), DSC/cuda_SimpleMatrixUtil.h:888-896
VHD F3 mat_mul_d(const float* m, F3 v)
{
    const float w = 0.0f;
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * w,
               m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * w,
               m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * w);
}

// ---------------------------------------------------------------------------
// coordinate helpers (DSC/VoxelUtilHashSDF.h)
// ---------------------------------------------------------------------------

// computeHashPos :217-225 (wrapping int products, unsigned modulo)
VHD uint32_t hash_pos(uint32_t numBuckets, I3 p)
{
    const uint32_t p0 = 73856093u, p1 = 19349669u, p2 = 83492791u;
    uint32_t r = ((uint32_t)p.x * p0) ^ ((uint32_t)p.y * p1) ^ ((uint32_t)p.z * p2);
    return r % numBuckets;
}

// worldToVirtualVoxelPos :266-270, one component
VHD int world_to_vvp1(float pos, float voxelSize)
{
    float p = pos / voxelSize;
    return f2i(p + (float)signi(p) * 0.5f);
}
VHD I3 world_to_vvp(float voxelSize, F3 pos)
{
    return mki3(world_to_vvp1(pos.x, voxelSize), world_to_vvp1(pos.y, voxelSize), world_to_vvp1(pos.z, voxelSize));
}
// virtualVoxelPosToSDFBlock :273-282 (floor division by 8)
VHD int vvp_to_block1(int v)
{
    if (v < 0) v -= VH_SDF_BLOCK_SIZE - 1;
    return v / VH_SDF_BLOCK_SIZE;
}
VHD I3 vvp_to_block(I3 v) { return mki3(vvp_to_block1(v.x), vvp_to_block1(v.y), vvp_to_block1(v.z)); }
// SDFBlockToVirtualVoxelPos :286 / virtualVoxelPosToWorld :291 / SDFBlockToWorld :296
VHD F3 vvp_to_world(float voxelSize, I3 v) { return mk3((float)v.x * voxelSize, (float)v.y * voxelSize, (float)v.z * voxelSize); }
VHD F3 block_to_world(float voxelSize, I3 b)
{
    return vvp_to_world(voxelSize, mki3(b.x * VH_SDF_BLOCK_SIZE, b.y * VH_SDF_BLOCK_SIZE, b.z * VH_SDF_BLOCK_SIZE));
}
VHD I3 world_to_block(float voxelSize, F3 p) { return vvp_to_block(world_to_vvp(voxelSize, p)); }
// virtualVoxelPosToLocalSDFBlockIndex :330-341
VHD int local1(int v)
{
    int l = v % VH_SDF_BLOCK_SIZE;
    if (l < 0) l += VH_SDF_BLOCK_SIZE;
    return l;
}
VHD int vvp_to_local_index(I3 v)
{
    return local1(v.z) * VH_SDF_BLOCK_SIZE * VH_SDF_BLOCK_SIZE + local1(v.y) * VH_SDF_BLOCK_SIZE + local1(v.x);
}
// getTruncation :255-257
VHD float get_truncation(const VhHashParams& hp, float z) { return hp.m_truncation + hp.m_truncScale * z; }

// ---------------------------------------------------------------------------
// cheaper exact arithmetic for the hot loops
// ---------------------------------------------------------------------------

// a / b, correctly rounded, for a loop-invariant divisor b with rb = 1.0f / b
// (IEEE).  Two Markstein correction steps on q = a*rb, the same refinement the
// compiler's division expansion performs after its reciprocal -- 5 VALU ops
// instead of ~11.  Exact for finite operands whose quotient is a normal
// number (positions / voxel size); verified against `/` by
// vh_debug_check_fast_math and by the bit-exact raycast parity tests.
VHD float div_exact(float a, float b, float rb)
{
    float q = a * rb;
    float e = __fmaf_rn(-b, q, a);
    q = __fmaf_rn(e, rb, q);
    e = __fmaf_rn(-b, q, a);
    q = __fmaf_rn(e, rb, q);
    return q;
}
VHD int world_to_vvp1_rb(float pos, float voxelSize, float rVoxelSize)
{
    float p = div_exact(pos, voxelSize, rVoxelSize);
    return f2i(p + (float)signi(p) * 0.5f);
}

// n % d for an invariant d through a host-computed multiply-shift
// (Granlund-Montgomery round-up method, exact for every 32-bit n; d >= 2)
struct HashMod {
    uint32_t d, m, sh;
};
VHD uint32_t umod_fast(uint32_t n, HashMod k)
{
    const uint32_t q0 = __umulhi(k.m, n);
    const uint32_t t = ((n - q0) >> 1) + q0;
    const uint32_t q = t >> k.sh;
    return n - q * k.d;
}
VHD uint32_t hash_pos_fast(HashMod k, I3 p)
{
    const uint32_t p0 = 73856093u, p1 = 19349669u, p2 = 83492791u;
    return umod_fast(((uint32_t)p.x * p0) ^ ((uint32_t)p.y * p1) ^ ((uint32_t)p.z * p2), k);
}

// ---------------------------------------------------------------------------
// camera helpers (DSC/DepthCameraUtil.h)
// ---------------------------------------------------------------------------

// kinectDepthToSkeleton :117-122
VHD F3 depth_to_skeleton(const VhDepthCameraParams& cp, uint32_t ux, uint32_t uy, float depth)
{
    const float x = ((float)ux - cp.mx) / cp.fx;
    const float y = ((float)uy - cp.my) / cp.fy;
    return mk3(depth * x, depth * y, depth);
}
// cameraToKinectProjZ :94-96
VHD float cam_to_proj_z(const VhDepthCameraParams& cp, float z)
{
    return (z - cp.m_sensorDepthWorldMin) / (cp.m_sensorDepthWorldMax - cp.m_sensorDepthWorldMin);
}
// kinectProjToCameraZ :129-131
VHD float proj_to_cam_z(const VhDepthCameraParams& cp, float z)
{
    return z * (cp.m_sensorDepthWorldMax - cp.m_sensorDepthWorldMin) + cp.m_sensorDepthWorldMin;
}
// isInCameraFrustumApprox :141-147 over cameraToKinectProj :99-110
VHD bool in_frustum_approx(const VhDepthCameraParams& cp, const float* viewMatrixInverse, F3 pos)
{
    F3 pc = mat_mul_p(viewMatrixInverse, pos);
    float px = pc.x * cp.fx / pc.z + cp.mx;
    float py = pc.y * cp.fy / pc.z + cp.my;
    float wm1 = (float)cp.m_imageWidth - 1.0f, hm1 = (float)cp.m_imageHeight - 1.0f;
    float x = (2.0f * px - wm1) / wm1;
    float y = (hm1 - 2.0f * py) / hm1;
    float z = cam_to_proj_z(cp, pc.z);
    const float s = 0.95f;
    x *= s; y *= s; z *= s;
    return !(x < -1.0f || x > 1.0f || y < -1.0f || y > 1.0f || z < 0.0f || z > 1.0f);
}
// isSDFBlockInCameraFrustumApprox, DSC/VoxelUtilHashSDF.h:306-309
VHD bool block_in_frustum(const VhHashParams& hp, const VhDepthCameraParams& cp, I3 blk)
{
    F3 pw = block_to_world(hp.m_virtualVoxelSize, blk);
    float off = hp.m_virtualVoxelSize * 0.5f * ((float)VH_SDF_BLOCK_SIZE - 1.0f);
    pw.x += off; pw.y += off; pw.z += off;
    return in_frustum_approx(cp, hp.m_rigidTransformInverse, pw);
}

// ---------------------------------------------------------------------------
// hash table access.  A VhHashEntry is 32 B: {pos.xyz, ptr} is one aligned
// 16-byte quad (a single dwordx4 load/store), offset is the dword behind it.
// ---------------------------------------------------------------------------

VHD int4 load_quad(const VhHashEntry* e) { return *reinterpret_cast<const int4*>(e); }
VHD void store_quad(VhHashEntry* e, int4 q) { *reinterpret_cast<int4*>(e) = q; }
VHD bool quad_matches(int4 q, I3 p) { return q.x == p.x && q.y == p.y && q.z == p.z && q.w != VH_FREE_ENTRY; }

// deleteHashEntry :365-369
VHD void delete_hash_entry(VhHashEntry* e)
{
    e->offset = 0;
    store_quad(e, make_int4(0, 0, 0, VH_FREE_ENTRY));
}
// HashEntry::operator= :64-73 (20-byte payload)
VHD void copy_entry(VhHashEntry* dst, const VhHashEntry* src)
{
    int4 q = load_quad(src);
    uint32_t o = src->offset;
    dst->offset = o;
    store_quad(dst, q);
}

// extension: per-bucket occupancy summary (count + 1 bit per bucket)
VHD void bucket_inc(const VhHashData& hd, uint32_t slotIdx)
{
    uint32_t b = slotIdx / VH_HASH_BUCKET_SIZE;
    uint32_t old = atomicAdd(&hd.d_bucketCount[b], 1u);
    if (old == 0u) atomicOr(&hd.d_bucketBits[b >> 5], 1u << (b & 31));
}
VHD void bucket_dec(const VhHashData& hd, uint32_t slotIdx)
{
    uint32_t b = slotIdx / VH_HASH_BUCKET_SIZE;
    uint32_t old = atomicSub(&hd.d_bucketCount[b], 1u);
    if (old == 1u) atomicAnd(&hd.d_bucketBits[b >> 5], ~(1u << (b & 31)));
}
VHD bool bucket_maybe_occupied(const VhHashData& hd, uint32_t h)
{
    return (hd.d_bucketBits[h >> 5] >> (h & 31)) & 1u;
}

// consumeHeap :519-523 (+ underflow guard the reference lacks: on an empty
// heap the counter is restored, the status word raised and block 0 returned
// is NOT handed out -- the caller must check `ok`)
VHD uint32_t consume_heap(const VhHashData& hd, uint32_t numSDFBlocks, bool& ok)
{
    uint32_t addr = atomicSub(&hd.d_heapCounter[0], 1u);
    if (addr >= numSDFBlocks) { // counter was already "-1": heap empty
        atomicAdd(&hd.d_heapCounter[0], 1u);
        atomicAdd(&hd.d_state[VH_STATE_HEAP_UNDERFLOW], 1u);
        ok = false;
        return 0;
    }
    ok = true;
    return hd.d_heap[addr];
}
// appendHeap :525-529
VHD void append_heap(const VhHashData& hd, uint32_t blockId)
{
    uint32_t addr = atomicAdd(&hd.d_heapCounter[0], 1u);
    hd.d_heap[addr + 1] = blockId;
}

// getHashEntryForSDFBlockPos :424-468, returning only what callers use:
// the ptr (VH_FREE_ENTRY on a miss).  The occupancy bit lets a miss in an
// empty bucket return without touching d_hash (an empty bucket has a free
// last slot with offset 0, so the list walk would stop at once).
VHD int lookup_ptr(const VhHashData& hd, const VhHashParams& hp, I3 blk)
{
    const uint32_t ne = hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
    const uint32_t h = hash_pos(hp.m_hashNumBuckets, blk);
    if (!bucket_maybe_occupied(hd, h)) return VH_FREE_ENTRY;
    const uint32_t base = h * VH_HASH_BUCKET_SIZE;
#pragma unroll 1
    for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) {
        int4 q = load_quad(&hd.d_hash[base + j]);
        if (quad_matches(q, blk)) return q.w;
    }
    const uint32_t idxLast = base + VH_HASH_BUCKET_SIZE - 1;
    uint32_t i = idxLast;
    uint32_t maxIter = 0;
#pragma unroll 1
    while (maxIter < hp.m_hashMaxCollisionLinkedListSize) {
        int4 q = load_quad(&hd.d_hash[i]);
        if (quad_matches(q, blk)) return q.w;
        if (q.w == VH_FREE_ENTRY && i != idxLast) {
            // A list never holds a free entry.  This one is a copy from before an alloc_block that runs beside this
            // kernel (CUDASceneRepHashSDF::integrateAhead) put a new head here: it stores the entry, fences, then
            // stores the link that led here, so the entry is in memory; drop this compute unit's stale lines and
            // read it again (what follows the new head is the list as it was).
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            q = load_quad(&hd.d_hash[i]);
            if (quad_matches(q, blk)) return q.w;
        }
        uint32_t off = hd.d_hash[i].offset;
        if (off == 0) break;
        i = (idxLast + off) % ne;
        maxIter++;
    }
    return VH_FREE_ENTRY;
}

// allocBlock :533-638.  Returns 0 = already there, 1 = allocated, 2 = lost a
// lock race (retry next pass), 3 = no room / heap empty.
VHD int alloc_block(const VhHashData& hd, const VhHashParams& hp, I3 pos, int32_t lockToken)
{
    const uint32_t ne = hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
    uint32_t h = hash_pos(hp.m_hashNumBuckets, pos);
    const uint32_t base = h * VH_HASH_BUCKET_SIZE;

    int firstEmpty = -1;
    const uint32_t idxLast = base + VH_HASH_BUCKET_SIZE - 1;
    const uint32_t maxLoop = hp.m_hashMaxCollisionLinkedListSize;
    uint32_t i = idxLast;
    uint32_t maxIter = 0;
    if (!bucket_maybe_occupied(hd, h)) {
        // empty bucket (no entry, hence no list hanging off its last slot): the first slot is the first free one.
        // A stale bit can only hide an entry written earlier in THIS pass, whose writer still holds the bucket
        // lock, so the lock attempt below fails exactly as it would after a full scan.
        firstEmpty = (int)base;
    } else {
        // the ten slots and the list head in flight together
        int4 qs[VH_HASH_BUCKET_SIZE];
#pragma unroll
        for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) qs[j] = load_quad(&hd.d_hash[base + j]);
        uint32_t off = hd.d_hash[idxLast].offset;
#pragma unroll
        for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) {
            if (quad_matches(qs[j], pos)) return 0;
            if (firstEmpty == -1 && qs[j].w == VH_FREE_ENTRY) firstEmpty = (int)(base + j);
        }
        // collision list (the reference's first hop re-reads the last slot: no match, see above)
#pragma unroll 1
        while (maxIter < maxLoop) {
            if (off == 0) break;
            i = (idxLast + off) % ne;
            maxIter++;
            if (!(maxIter < maxLoop)) break;
            int4 q = load_quad(&hd.d_hash[i]);
            if (quad_matches(q, pos)) return 0;
            off = hd.d_hash[i].offset;
        }
    }

    if (firstEmpty != -1) {
        int prev = atomicExch(&hd.d_hashBucketMutex[h], lockToken);
        if (prev != lockToken) {
            bool ok;
            uint32_t blk = consume_heap(hd, hp.m_numSDFBlocks, ok);
            if (!ok) return 3;
            VhHashEntry* e = &hd.d_hash[firstEmpty];
            e->offset = VH_NO_OFFSET;
            store_quad(e, make_int4(pos.x, pos.y, pos.z, (int)(blk * VH_SDF_BLOCK_VOXELS)));
            bucket_inc(hd, (uint32_t)firstEmpty);
            return 1;
        }
        atomicAdd(&hd.d_state[VH_STATE_ALLOC_LOCK_LOST], 1u);
        return 2;
    }

    int offset = 0;
    maxIter = 0;
#pragma unroll 1
    while (maxIter < maxLoop) {
        offset++;
        i = (idxLast + (uint32_t)offset) % ne;
        if ((offset % VH_HASH_BUCKET_SIZE) == 0) continue; // never a bucket's last slot
        int4 q = load_quad(&hd.d_hash[i]);
        if (q.w == VH_FREE_ENTRY) {
            int prev = atomicExch(&hd.d_hashBucketMutex[h], lockToken);
            if (prev != lockToken) {
                uint32_t lastOffset = hd.d_hash[idxLast].offset;
                uint32_t h2 = i / VH_HASH_BUCKET_SIZE;
                prev = atomicExch(&hd.d_hashBucketMutex[h2], lockToken);
                if (prev != lockToken) {
                    bool ok;
                    uint32_t blk = consume_heap(hd, hp.m_numSDFBlocks, ok);
                    if (!ok) return 3;
                    VhHashEntry* e = &hd.d_hash[i];
                    e->offset = lastOffset;
                    store_quad(e, make_int4(pos.x, pos.y, pos.z, (int)(blk * VH_SDF_BLOCK_VOXELS)));
                    // the new head is in memory before the link to it: a ray caster that runs beside this pass
                    // (integrateAhead) and follows the link finds the entry (lookup_ptr)
                    __threadfence();
                    hd.d_hash[idxLast].offset = (uint32_t)offset;
                    bucket_inc(hd, i);
                    return 1;
                }
            }
            atomicAdd(&hd.d_state[VH_STATE_ALLOC_LOCK_LOST], 1u);
            return 2;
        }
        maxIter++;
    }
    return 3;
}

// insertHashEntry :643-717.  In-bucket part as the reference (CAS on ptr).
// The reference's overflow branch (:682-713) is an unported HLSL remnant
// (3-int indexing) and undefined on this layout; defined behaviour here:
// take the home bucket, claim the first free non-last slot behind it by CAS
// and insert at the head of the home bucket's list.
VHD bool insert_hash_entry(const VhHashData& hd, const VhHashParams& hp, I3 pos, int ptr, int32_t lockToken)
{
    const uint32_t ne = hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
    const uint32_t h = hash_pos(hp.m_hashNumBuckets, pos);
    const uint32_t base = h * VH_HASH_BUCKET_SIZE;
#pragma unroll 1
    for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) {
        VhHashEntry* e = &hd.d_hash[base + j];
        int prev = atomicCAS(&e->ptr, VH_FREE_ENTRY, VH_LOCK_ENTRY);
        if (prev == VH_FREE_ENTRY) {
            e->offset = 0;
            store_quad(e, make_int4(pos.x, pos.y, pos.z, ptr));
            bucket_inc(hd, base + j);
            return true;
        }
    }
    const uint32_t idxLast = base + VH_HASH_BUCKET_SIZE - 1;
    int prevLock = atomicExch(&hd.d_hashBucketMutex[h], lockToken);
    if (prevLock == lockToken) return false;
    uint32_t maxIter = 0;
    int offset = 0;
#pragma unroll 1
    while (maxIter < hp.m_hashMaxCollisionLinkedListSize) {
        offset++;
        uint32_t i = (idxLast + (uint32_t)offset) % ne;
        if ((offset % VH_HASH_BUCKET_SIZE) == 0) continue;
        VhHashEntry* e = &hd.d_hash[i];
        int prev = atomicCAS(&e->ptr, VH_FREE_ENTRY, VH_LOCK_ENTRY);
        if (prev == VH_FREE_ENTRY) {
            e->offset = hd.d_hash[idxLast].offset;
            store_quad(e, make_int4(pos.x, pos.y, pos.z, ptr));
            hd.d_hash[idxLast].offset = (uint32_t)offset;
            bucket_inc(hd, i);
            return true;
        }
        maxIter++;
    }
    return false;
}

// deleteHashEntryElement :723-809
VHD bool delete_hash_entry_element(const VhHashData& hd, const VhHashParams& hp, I3 blk, int32_t lockToken)
{
    const uint32_t ne = hp.m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
    const uint32_t h = hash_pos(hp.m_hashNumBuckets, blk);
    const uint32_t base = h * VH_HASH_BUCKET_SIZE;

#pragma unroll 1
    for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) {
        const uint32_t i = base + j;
        int4 q = load_quad(&hd.d_hash[i]);
        if (quad_matches(q, blk)) {
            uint32_t off = hd.d_hash[i].offset;
            if (off != 0) {
                int prev = atomicExch(&hd.d_hashBucketMutex[h], lockToken);
                if (prev == lockToken) return false;
                append_heap(hd, (uint32_t)q.w / VH_SDF_BLOCK_VOXELS);
                uint32_t nextIdx = (i + off) % ne;
                copy_entry(&hd.d_hash[i], &hd.d_hash[nextIdx]);
                delete_hash_entry(&hd.d_hash[nextIdx]);
                bucket_dec(hd, nextIdx);
                return true;
            } else {
                append_heap(hd, (uint32_t)q.w / VH_SDF_BLOCK_VOXELS);
                delete_hash_entry(&hd.d_hash[i]);
                bucket_dec(hd, i);
                return true;
            }
        }
    }
    const uint32_t idxLast = base + VH_HASH_BUCKET_SIZE - 1;
    uint32_t prevIdx = idxLast;
    uint32_t i = (idxLast + hd.d_hash[idxLast].offset) % ne;
    uint32_t maxIter = 0;
#pragma unroll 1
    while (maxIter < hp.m_hashMaxCollisionLinkedListSize) {
        int4 q = load_quad(&hd.d_hash[i]);
        uint32_t off = hd.d_hash[i].offset;
        if (quad_matches(q, blk)) {
            int prev = atomicExch(&hd.d_hashBucketMutex[h], lockToken);
            if (prev == lockToken) return false;
            append_heap(hd, (uint32_t)q.w / VH_SDF_BLOCK_VOXELS);
            delete_hash_entry(&hd.d_hash[i]);
            bucket_dec(hd, i);
            hd.d_hash[prevIdx].offset = off;
            return true;
        }
        if (off == 0) return false;
        prevIdx = i;
        i = (idxLast + off) % ne;
        maxIter++;
    }
    return false;
}

// deleteHashEntryElement for a whole wave that frees ONE block (every lane calls it with the same block): ten lanes read
// the bucket's ten slots together instead of one lane reading them one after the other (up to ten dependent trips to
// memory at the end of a wave's life), and the lane that finds the block frees it with its two counter atomics in flight
// together.  An element of a collision list (a head with an offset, or a block that is not in its bucket) goes through
// the one-lane code.  Returns the same value in every lane.
VHD bool delete_hash_entry_element_wave(const VhHashData& hd, const VhHashParams& hp, I3 blk, int32_t lockToken, uint32_t lane)
{
    const uint32_t h = hash_pos(hp.m_hashNumBuckets, blk);
    const uint32_t i = h * VH_HASH_BUCKET_SIZE + lane;
    int4 q = make_int4(0, 0, 0, VH_FREE_ENTRY);
    uint32_t off = 0u;
    if (lane < VH_HASH_BUCKET_SIZE) { q = load_quad(&hd.d_hash[i]); off = hd.d_hash[i].offset; }
    const bool match = lane < VH_HASH_BUCKET_SIZE && quad_matches(q, blk);
    const unsigned long long m = __ballot(match);
    bool ok = false;
    if (m == 0ull || __ballot(match && off != 0u) != 0ull) {
        if (lane == 0u) ok = delete_hash_entry_element(hd, hp, blk, lockToken);
    } else if (match) { // (a block sits in at most one slot)
        const uint32_t addr = atomicAdd(&hd.d_heapCounter[0], 1u);          // append_heap
        const uint32_t old = atomicSub(&hd.d_bucketCount[h], 1u);           // bucket_dec
        hd.d_heap[addr + 1] = (uint32_t)q.w / VH_SDF_BLOCK_VOXELS;
        delete_hash_entry(&hd.d_hash[i]);
        if (old == 1u) atomicAnd(&hd.d_bucketBits[h >> 5], ~(1u << (h & 31)));
        ok = true;
    }
    return __ballot(ok) != 0ull;
}

// ---------------------------------------------------------------------------
// voxels.  A voxel is moved as one 64-bit word (DSC/VoxelUtilHashSDF.h:82-86):
// low dword = sdf bits, high dword = r | g<<8 | b<<16 | weight<<24.
// ---------------------------------------------------------------------------

struct Vox {
    float sdf;
    uint32_t cw; // colour + weight
    VHD uint32_t weight() const { return cw >> 24; }
    VHD uint32_t r() const { return cw & 0xffu; }
    VHD uint32_t g() const { return (cw >> 8) & 0xffu; }
    VHD uint32_t b() const { return (cw >> 16) & 0xffu; }
};
VHD Vox unpack_vox(uint2 w) { Vox v; v.sdf = __uint_as_float(w.x); v.cw = w.y; return v; }
VHD uint2 pack_vox(Vox v) { return make_uint2(__float_as_uint(v.sdf), v.cw); }
VHD uint32_t pack_cw(uint32_t r, uint32_t g, uint32_t b, uint32_t w) { return r | (g << 8) | (b << 16) | (w << 24); }

// combineVoxel :229-250
VHD Vox combine_voxel(const VhHashParams& hp, Vox v0, Vox v1)
{
    Vox out;
    // colour :236-240: uchar(0.5f * c0 + 0.5f * c1 + 0.5f) per channel.  For bytes c0, c1 every term and sum is exact in
    // fp32 (half-integers up to 255.5), so the value is floor((c0 + c1) / 2 + 1/2) = (c0 + c1 + 1) >> 1, the byte-wise
    // average rounded up: (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7f) on the packed word, all channels at once
    // (tests/test_oracle_math.py checks the identity for all 65 536 pairs against the float formula).
    const uint32_t a = v0.cw & 0x00ffffffu, c = v1.cw & 0x00ffffffu;
    const uint32_t rgb = (a | c) - (((a ^ c) >> 1) & 0x7f7f7f7fu);
    float w0 = (float)v0.weight(), w1 = (float)v1.weight();
    out.sdf = (v0.sdf * w0 + v1.sdf * w1) / (w0 + w1);
    uint32_t w = min(hp.m_integrationWeightMax, v0.weight() + v1.weight());
    out.cw = rgb | ((w & 0xffu) << 24);
    return out;
}

} // namespace vhd
