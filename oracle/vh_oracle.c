/*
 * vh_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see vh_oracle.h).
 *
 * Serial, IEEE-fp32 restatement of the reference's voxel-hashing hot path.
 * "Kernels" run their threads one after the other in launch order, atomics are
 * plain read-modify-writes.  Build with -O2 -ffp-contract=off -fwrapv on
 * x86-64 (SSE2 scalar float: no excess precision, no FMA contraction).
 *
 * Citations:  DSC/ = /root/reference/DepthSensingCUDA/Source/
 *             CUTIL = /root/reference/DepthSensingCUDA/Include/cutil/inc/cutil_math.h
 *
 * Float->integer conversions follow the CUDA device semantics the reference
 * runs with (cvt.rzi: truncate, saturate, NaN -> 0), not x86 cvttss2si.
 *
 * Two reference defects on the collision + streaming path are fenced, not
 * reproduced (hard part 4): see vho_insert_hash_entry and
 * vho_stream_out_pass1.
 *
 * The `#pragma omp` lines and `#ifdef _OPENMP` sections are inert in the
 * checker (libvh_oracle.so, built without -fopenmp).  libvh_oracle_omp.so, the
 * same file built with -fopenmp, is bench.py's all-core CPU baseline
 * (SURVEY.md 8(d)); a test holds it to the serial build's results.
 *
 * PARITY UNPINNED (see vh_oracle.h).
 */
#include "vh_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP /* only libvh_oracle_omp.so, the bench's all-core CPU baseline; the checker is built without */
#include <omp.h>
#endif

/* caps the threads of the parallel sections (no-op in the checker's build) */
void vho_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* threads the parallel sections use: 1 in the checker's build */
int vho_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* small vector helpers                                                      */
/* ------------------------------------------------------------------------- */

typedef struct { float x, y, z; } f3;
typedef struct { int x, y, z; } i3;

static const union { uint32_t u; float f; } kMinf = { 0xff800000u }, kPinf = { 0x7f800000u };
#define MINF (kMinf.f)
#define PINF (kPinf.f)

static inline f3 mk3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline i3 mki3(int x, int y, int z) { i3 r = { x, y, z }; return r; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* CUTIL:1145 */

/* cvt.rzi.s32.f32 */
static inline int f2i(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
/* cvt.rzi.u8.f32 */
static inline uint8_t f2uc(float v)
{
    if (v != v) return 0;
    if (v >= 255.0f) return 255;
    if (v <= 0.0f) return 0;
    return (uint8_t)(int)v;
}

/* CUTIL:31-33 */
static inline int signf_i(float v) { return (0.0f < v) - (v < 0.0f); }

/* CUTIL:1207-1211 with the host rsqrtf of CUTIL:81-84 */
static inline f3 normalize3(f3 v)
{
    float invLen = 1.0f / sqrtf(dot3(v, v));
    return scale3(v, invLen);
}

/* float4x4 * float3, DSC/cuda_SimpleMatrixUtil.h:900-907 */
static inline f3 mat_mul_p(const float* m, f3 v)
{
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * 1.0f,
               m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * 1.0f,
               m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * 1.0f);
}
/* float4x4 * float4 with w = 0 (xyz part), DSC/cuda_SimpleMatrixUtil.h:888-896 */
static inline f3 mat_mul_d(const float* m, f3 v)
{
    const float w = 0.0f;
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * w,
               m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * w,
               m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * w);
}

/* float4x4::getInverse, DSC/cuda_SimpleMatrixUtil.h:944-1069: cofactor
 * expansion; each cofactor is six signed triple products summed left to right
 * in the reference's term order (table below: {sign, i, j, k} on entries[]). */
void vho_mat4_inverse(const float e[16], float out[16])
{
    static const signed char T[16][6][4] = {
        /* inv[0]  */ { {+1,5,10,15}, {-1,5,11,14}, {-1,9,6,15}, {+1,9,7,14}, {+1,13,6,11}, {-1,13,7,10} },
        /* inv[1]  */ { {-1,1,10,15}, {+1,1,11,14}, {+1,9,2,15}, {-1,9,3,14}, {-1,13,2,11}, {+1,13,3,10} },
        /* inv[2]  */ { {+1,1,6,15}, {-1,1,7,14}, {-1,5,2,15}, {+1,5,3,14}, {+1,13,2,7}, {-1,13,3,6} },
        /* inv[3]  */ { {-1,1,6,11}, {+1,1,7,10}, {+1,5,2,11}, {-1,5,3,10}, {-1,9,2,7}, {+1,9,3,6} },
        /* inv[4]  */ { {-1,4,10,15}, {+1,4,11,14}, {+1,8,6,15}, {-1,8,7,14}, {-1,12,6,11}, {+1,12,7,10} },
        /* inv[5]  */ { {+1,0,10,15}, {-1,0,11,14}, {-1,8,2,15}, {+1,8,3,14}, {+1,12,2,11}, {-1,12,3,10} },
        /* inv[6]  */ { {-1,0,6,15}, {+1,0,7,14}, {+1,4,2,15}, {-1,4,3,14}, {-1,12,2,7}, {+1,12,3,6} },
        /* inv[7]  */ { {+1,0,6,11}, {-1,0,7,10}, {-1,4,2,11}, {+1,4,3,10}, {+1,8,2,7}, {-1,8,3,6} },
        /* inv[8]  */ { {+1,4,9,15}, {-1,4,11,13}, {-1,8,5,15}, {+1,8,7,13}, {+1,12,5,11}, {-1,12,7,9} },
        /* inv[9]  */ { {-1,0,9,15}, {+1,0,11,13}, {+1,8,1,15}, {-1,8,3,13}, {-1,12,1,11}, {+1,12,3,9} },
        /* inv[10] */ { {+1,0,5,15}, {-1,0,7,13}, {-1,4,1,15}, {+1,4,3,13}, {+1,12,1,7}, {-1,12,3,5} },
        /* inv[11] */ { {-1,0,5,11}, {+1,0,7,9}, {+1,4,1,11}, {-1,4,3,9}, {-1,8,1,7}, {+1,8,3,5} },
        /* inv[12] */ { {-1,4,9,14}, {+1,4,10,13}, {+1,8,5,14}, {-1,8,6,13}, {-1,12,5,10}, {+1,12,6,9} },
        /* inv[13] */ { {+1,0,9,14}, {-1,0,10,13}, {-1,8,1,14}, {+1,8,2,13}, {+1,12,1,10}, {-1,12,2,9} },
        /* inv[14] */ { {-1,0,5,14}, {+1,0,6,13}, {+1,4,1,14}, {-1,4,2,13}, {-1,12,1,6}, {+1,12,2,5} },
        /* inv[15] */ { {+1,0,5,10}, {-1,0,6,9}, {-1,4,1,10}, {+1,4,2,9}, {+1,8,1,6}, {-1,8,2,5} },
    };
    float inv[16];
    for (int n = 0; n < 16; n++) {
        float acc = 0.0f;
        for (int t = 0; t < 6; t++) {
            float p = e[T[n][t][1]] * e[T[n][t][2]] * e[T[n][t][3]];
            if (t == 0) acc = (T[n][t][0] > 0) ? p : -p;
            else acc = (T[n][t][0] > 0) ? acc + p : acc - p;
        }
        inv[n] = acc;
    }
    float det = e[0] * inv[0] + e[1] * inv[4] + e[2] * inv[8] + e[3] * inv[12];
    float detr = 1.0f / det;
    for (int n = 0; n < 16; n++) out[n] = inv[n] * detr;
}

/* ------------------------------------------------------------------------- */
/* memory                                                                    */
/* ------------------------------------------------------------------------- */

static inline uint32_t num_entries(const VhHashParams* hp)
{
    return hp->m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
}

/* HashData::allocate(params, false), DSC/VoxelUtilHashSDF.h:126-136 */
int vho_hash_data_alloc(VhHashData* hd, const VhHashParams* hp)
{
    memset(hd, 0, sizeof(*hd));
    size_t ne = num_entries(hp);
    hd->d_heap = (uint32_t*)malloc(sizeof(uint32_t) * hp->m_numSDFBlocks);
    hd->d_heapCounter = (uint32_t*)malloc(sizeof(uint32_t));
    hd->d_hash = (VhHashEntry*)malloc(sizeof(VhHashEntry) * ne);
    hd->d_hashDecision = (int32_t*)malloc(sizeof(int32_t) * ne);
    hd->d_hashDecisionPrefix = (int32_t*)malloc(sizeof(int32_t) * ne);
    hd->d_hashCompactified = (VhHashEntry*)malloc(sizeof(VhHashEntry) * ne);
    hd->d_hashCompactifiedCounter = (int32_t*)malloc(sizeof(int32_t));
    hd->d_SDFBlocks = (VhVoxel*)malloc(sizeof(VhVoxel) * (size_t)hp->m_numSDFBlocks * VH_SDF_BLOCK_VOXELS);
    hd->d_hashBucketMutex = (int32_t*)malloc(sizeof(int32_t) * hp->m_hashNumBuckets);
    hd->m_bIsOnGPU = 0;
    if (!hd->d_heap || !hd->d_heapCounter || !hd->d_hash || !hd->d_hashDecision || !hd->d_hashDecisionPrefix ||
        !hd->d_hashCompactified || !hd->d_hashCompactifiedCounter || !hd->d_SDFBlocks || !hd->d_hashBucketMutex) {
        vho_hash_data_free(hd);
        return 1;
    }
    memset(hd->d_hashDecision, 0, sizeof(int32_t) * ne);
    memset(hd->d_hashDecisionPrefix, 0, sizeof(int32_t) * ne);
    return 0;
}

void vho_hash_data_free(VhHashData* hd)
{
    free(hd->d_heap); free(hd->d_heapCounter); free(hd->d_hash); free(hd->d_hashDecision);
    free(hd->d_hashDecisionPrefix); free(hd->d_hashCompactified); free(hd->d_hashCompactifiedCounter);
    free(hd->d_SDFBlocks); free(hd->d_hashBucketMutex);
    memset(hd, 0, sizeof(*hd));
}

/* ------------------------------------------------------------------------- */
/* HashData device functions (DSC/VoxelUtilHashSDF.h)                        */
/* ------------------------------------------------------------------------- */

/* computeHashPos :217-225.  int products wrap (-fwrapv); the modulo is
 * unsigned because m_hashNumBuckets is unsigned. */
static inline uint32_t hash_pos(const VhHashParams* hp, i3 p)
{
    const int p0 = 73856093, p1 = 19349669, p2 = 83492791;
    uint32_t res = (uint32_t)((p.x * p0) ^ (p.y * p1) ^ (p.z * p2)) % hp->m_hashNumBuckets;
    return res;
}

/* combineVoxel :229-250 */
static inline VhVoxel combine_voxel(const VhHashParams* hp, VhVoxel v0, VhVoxel v1)
{
    VhVoxel out;
    for (int c = 0; c < 3; c++) {
        float res = 0.5f * (float)v0.color[c] + 0.5f * (float)v1.color[c];
        out.color[c] = f2uc(res + 0.5f);
    }
    out.sdf = (v0.sdf * (float)v0.weight + v1.sdf * (float)v1.weight) / ((float)v0.weight + (float)v1.weight);
    uint32_t w = (uint32_t)v0.weight + (uint32_t)v1.weight;
    if (hp->m_integrationWeightMax < w) w = hp->m_integrationWeightMax;
    out.weight = (uint8_t)w;
    return out;
}

/* getTruncation :255-257 */
static inline float get_truncation(const VhHashParams* hp, float z)
{
    return hp->m_truncation + hp->m_truncScale * z;
}

/* worldToVirtualVoxelPos :266-270 */
static inline i3 world_to_vvp(const VhHashParams* hp, f3 pos)
{
    f3 p = mk3(pos.x / hp->m_virtualVoxelSize, pos.y / hp->m_virtualVoxelSize, pos.z / hp->m_virtualVoxelSize);
    return mki3(f2i(p.x + (float)signf_i(p.x) * 0.5f),
                f2i(p.y + (float)signf_i(p.y) * 0.5f),
                f2i(p.z + (float)signf_i(p.z) * 0.5f));
}

/* virtualVoxelPosToSDFBlock :273-282 */
static inline i3 vvp_to_block(i3 v)
{
    if (v.x < 0) v.x -= VH_SDF_BLOCK_SIZE - 1;
    if (v.y < 0) v.y -= VH_SDF_BLOCK_SIZE - 1;
    if (v.z < 0) v.z -= VH_SDF_BLOCK_SIZE - 1;
    return mki3(v.x / VH_SDF_BLOCK_SIZE, v.y / VH_SDF_BLOCK_SIZE, v.z / VH_SDF_BLOCK_SIZE);
}

/* SDFBlockToVirtualVoxelPos :286, virtualVoxelPosToWorld :291, SDFBlockToWorld :296 */
static inline i3 block_to_vvp(i3 b) { return mki3(b.x * VH_SDF_BLOCK_SIZE, b.y * VH_SDF_BLOCK_SIZE, b.z * VH_SDF_BLOCK_SIZE); }
static inline f3 vvp_to_world(const VhHashParams* hp, i3 v)
{
    return mk3((float)v.x * hp->m_virtualVoxelSize, (float)v.y * hp->m_virtualVoxelSize, (float)v.z * hp->m_virtualVoxelSize);
}
static inline f3 block_to_world(const VhHashParams* hp, i3 b) { return vvp_to_world(hp, block_to_vvp(b)); }
/* worldToSDFBlock :301 */
static inline i3 world_to_block(const VhHashParams* hp, f3 p) { return vvp_to_block(world_to_vvp(hp, p)); }

/* virtualVoxelPosToLocalSDFBlockIndex :330-341 with linearizeVoxelPos :322-327 */
static inline int vvp_to_local_index(i3 v)
{
    int lx = v.x % VH_SDF_BLOCK_SIZE, ly = v.y % VH_SDF_BLOCK_SIZE, lz = v.z % VH_SDF_BLOCK_SIZE;
    if (lx < 0) lx += VH_SDF_BLOCK_SIZE;
    if (ly < 0) ly += VH_SDF_BLOCK_SIZE;
    if (lz < 0) lz += VH_SDF_BLOCK_SIZE;
    return lz * VH_SDF_BLOCK_SIZE * VH_SDF_BLOCK_SIZE + ly * VH_SDF_BLOCK_SIZE + lx;
}

/* DepthCameraData::cameraToKinectScreenFloat, DSC/DepthCameraUtil.h:74-79 */
static inline void cam_to_screen_float(const VhDepthCameraParams* cp, f3 pos, float* sx, float* sy)
{
    *sx = pos.x * cp->fx / pos.z + cp->mx;
    *sy = pos.y * cp->fy / pos.z + cp->my;
}
/* cameraToKinectProjZ :94-96 */
static inline float cam_to_proj_z(const VhDepthCameraParams* cp, float z)
{
    return (z - cp->m_sensorDepthWorldMin) / (cp->m_sensorDepthWorldMax - cp->m_sensorDepthWorldMin);
}
/* kinectDepthToSkeleton :117-122 */
static inline f3 depth_to_skeleton(const VhDepthCameraParams* cp, uint32_t ux, uint32_t uy, float depth)
{
    const float x = ((float)ux - cp->mx) / cp->fx;
    const float y = ((float)uy - cp->my) / cp->fy;
    return mk3(depth * x, depth * y, depth);
}
/* kinectProjToCameraZ :129-131 */
static inline float proj_to_cam_z(const VhDepthCameraParams* cp, float z)
{
    return z * (cp->m_sensorDepthWorldMax - cp->m_sensorDepthWorldMin) + cp->m_sensorDepthWorldMin;
}

/* isInCameraFrustumApprox, DSC/DepthCameraUtil.h:141-147 with cameraToKinectProj :99-110 */
static inline int in_frustum_approx(const VhDepthCameraParams* cp, const float* viewMatrixInverse, f3 pos)
{
    f3 pc = mat_mul_p(viewMatrixInverse, pos);
    float px, py;
    cam_to_screen_float(cp, pc, &px, &py);
    float wm1 = (float)cp->m_imageWidth - 1.0f, hm1 = (float)cp->m_imageHeight - 1.0f;
    f3 pr;
    pr.x = (2.0f * px - wm1) / wm1;
    pr.y = (hm1 - 2.0f * py) / hm1;
    pr.z = cam_to_proj_z(cp, pc.z);
    const float s = (float)0.95;
    pr.x *= s; pr.y *= s; pr.z *= s;
    return !(pr.x < -1.0f || pr.x > 1.0f || pr.y < -1.0f || pr.y > 1.0f || pr.z < 0.0f || pr.z > 1.0f);
}

/* isSDFBlockInCameraFrustumApprox, DSC/VoxelUtilHashSDF.h:306-309 */
static inline int block_in_frustum(const VhHashParams* hp, const VhDepthCameraParams* cp, i3 blk)
{
    f3 pw = block_to_world(hp, blk);
    float off = hp->m_virtualVoxelSize * 0.5f * ((float)VH_SDF_BLOCK_SIZE - 1.0f);
    pw.x += off; pw.y += off; pw.z += off;
    return in_frustum_approx(cp, hp->m_rigidTransformInverse, pw);
}

static inline int entry_matches(const VhHashEntry* e, i3 p)
{
    return e->pos[0] == p.x && e->pos[1] == p.y && e->pos[2] == p.z && e->ptr != VH_FREE_ENTRY;
}

/* deleteHashEntry :365-369 */
static inline void delete_hash_entry(VhHashEntry* e)
{
    e->pos[0] = e->pos[1] = e->pos[2] = 0;
    e->offset = 0;
    e->ptr = VH_FREE_ENTRY;
}

/* HashEntry::operator= copies the 20-byte payload :64-73 */
static inline void copy_entry(VhHashEntry* dst, const VhHashEntry* src)
{
    dst->pos[0] = src->pos[0]; dst->pos[1] = src->pos[1]; dst->pos[2] = src->pos[2];
    dst->ptr = src->ptr; dst->offset = src->offset;
}

/* getHashEntryForSDFBlockPos :424-468 */
static VhHashEntry get_hash_entry_for_block(const VhHashData* hd, const VhHashParams* hp, i3 blk)
{
    const uint32_t ne = num_entries(hp);
    uint32_t h = hash_pos(hp, blk);
    uint32_t hpn = h * VH_HASH_BUCKET_SIZE;

    VhHashEntry entry;
    memset(&entry, 0, sizeof(entry));
    entry.pos[0] = blk.x; entry.pos[1] = blk.y; entry.pos[2] = blk.z;
    entry.offset = 0;
    entry.ptr = VH_FREE_ENTRY;

    for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) {
        const VhHashEntry* curr = &hd->d_hash[hpn + j];
        if (entry_matches(curr, blk)) return *curr;
    }
    const uint32_t idxLast = (h + 1) * VH_HASH_BUCKET_SIZE - 1;
    int i = (int)idxLast;
    uint32_t maxIter = 0;
    while (maxIter < hp->m_hashMaxCollisionLinkedListSize) {
        const VhHashEntry* curr = &hd->d_hash[i];
        if (entry_matches(curr, blk)) return *curr;
        if (curr->offset == 0) break;
        i = (int)(((uint32_t)idxLast + curr->offset) % ne);
        maxIter++;
    }
    return entry;
}

/* consumeHeap :519-523 / appendHeap :525-529 */
static inline uint32_t consume_heap(VhHashData* hd)
{
    uint32_t addr = hd->d_heapCounter[0];
    hd->d_heapCounter[0] = addr - 1;
    return hd->d_heap[addr];
}
static inline void append_heap(VhHashData* hd, uint32_t ptr)
{
    uint32_t addr = hd->d_heapCounter[0];
    hd->d_heapCounter[0] = addr + 1;
    hd->d_heap[addr + 1] = ptr;
}
static inline int atomic_exch(int32_t* p, int32_t v) { int32_t o = *p; *p = v; return o; }

/* allocBlock :533-638 */
static void alloc_block(VhHashData* hd, const VhHashParams* hp, i3 pos)
{
    const uint32_t ne = num_entries(hp);
    uint32_t h = hash_pos(hp, pos);
    uint32_t hpn = h * VH_HASH_BUCKET_SIZE;

    int firstEmpty = -1;
    for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) {
        uint32_t i = j + hpn;
        const VhHashEntry* curr = &hd->d_hash[i];
        if (entry_matches(curr, pos)) return;
        if (firstEmpty == -1 && curr->ptr == VH_FREE_ENTRY) firstEmpty = (int)i;
    }

    const uint32_t idxLast = (h + 1) * VH_HASH_BUCKET_SIZE - 1;
    uint32_t i = idxLast;
    uint32_t maxIter = 0;
    const uint32_t maxLoop = hp->m_hashMaxCollisionLinkedListSize;
    while (maxIter < maxLoop) {
        VhHashEntry curr = hd->d_hash[i];
        if (entry_matches(&curr, pos)) return;
        if (curr.offset == 0) break;
        i = (idxLast + curr.offset) % ne;
        maxIter++;
    }

    if (firstEmpty != -1) {
        int prev = atomic_exch(&hd->d_hashBucketMutex[h], VH_LOCK_ENTRY);
        if (prev != VH_LOCK_ENTRY) {
            VhHashEntry* e = &hd->d_hash[firstEmpty];
            e->pos[0] = pos.x; e->pos[1] = pos.y; e->pos[2] = pos.z;
            e->offset = VH_NO_OFFSET;
            e->ptr = (int32_t)(consume_heap(hd) * VH_SDF_BLOCK_VOXELS);
        }
        return;
    }

    int offset = 0;
    maxIter = 0;
    while (maxIter < maxLoop) {
        offset++;
        i = (idxLast + (uint32_t)offset) % ne;
        if ((offset % VH_HASH_BUCKET_SIZE) == 0) continue; /* never a bucket's last slot */
        VhHashEntry curr = hd->d_hash[i];
        if (curr.ptr == VH_FREE_ENTRY) {
            int prev = atomic_exch(&hd->d_hashBucketMutex[h], VH_LOCK_ENTRY);
            if (prev != VH_LOCK_ENTRY) {
                VhHashEntry last = hd->d_hash[idxLast];
                h = i / VH_HASH_BUCKET_SIZE;
                prev = atomic_exch(&hd->d_hashBucketMutex[h], VH_LOCK_ENTRY);
                if (prev != VH_LOCK_ENTRY) {
                    VhHashEntry* e = &hd->d_hash[i];
                    e->pos[0] = pos.x; e->pos[1] = pos.y; e->pos[2] = pos.z;
                    e->offset = last.offset;
                    e->ptr = (int32_t)(consume_heap(hd) * VH_SDF_BLOCK_VOXELS);
                    last.offset = (uint32_t)offset;
                    copy_entry(&hd->d_hash[idxLast], &last);
                }
            }
            return;
        }
        maxIter++;
    }
}

/* insertHashEntry :643-717.  In-bucket part as the reference (CAS on ptr).
 * FENCED: the reference's overflow branch (:682-713) indexes the table as
 * 3-int entries and packs (offset<<16)|z, an HLSL remnant that is undefined on
 * the 5-int layout.  Defined behaviour here: take the home bucket's mutex,
 * claim the first free non-last slot after the bucket by CAS, insert it at
 * the head of the home bucket's list (as allocBlock's overflow branch does). */
static int insert_hash_entry(VhHashData* hd, const VhHashParams* hp, VhHashEntry entry)
{
    const uint32_t ne = num_entries(hp);
    i3 p = mki3(entry.pos[0], entry.pos[1], entry.pos[2]);
    uint32_t h = hash_pos(hp, p);
    uint32_t hpn = h * VH_HASH_BUCKET_SIZE;
    for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) {
        uint32_t i = j + hpn;
        if (hd->d_hash[i].ptr == VH_FREE_ENTRY) { /* atomicCAS(ptr, FREE, LOCK) succeeded */
            copy_entry(&hd->d_hash[i], &entry);
            return 1;
        }
    }
    const uint32_t idxLast = (h + 1) * VH_HASH_BUCKET_SIZE - 1;
    int prev = atomic_exch(&hd->d_hashBucketMutex[h], VH_LOCK_ENTRY);
    if (prev == VH_LOCK_ENTRY) return 0;
    uint32_t maxIter = 0;
    int offset = 0;
    while (maxIter < hp->m_hashMaxCollisionLinkedListSize) {
        offset++;
        uint32_t i = (idxLast + (uint32_t)offset) % ne;
        if ((offset % VH_HASH_BUCKET_SIZE) == 0) continue;
        if (hd->d_hash[i].ptr == VH_FREE_ENTRY) {
            entry.offset = hd->d_hash[idxLast].offset;
            copy_entry(&hd->d_hash[i], &entry);
            hd->d_hash[idxLast].offset = (uint32_t)offset;
            return 1;
        }
        maxIter++;
    }
    return 0;
}

/* deleteHashEntryElement :723-809 */
static int delete_hash_entry_element(VhHashData* hd, const VhHashParams* hp, i3 blk)
{
    const uint32_t ne = num_entries(hp);
    uint32_t h = hash_pos(hp, blk);
    uint32_t hpn = h * VH_HASH_BUCKET_SIZE;

    for (uint32_t j = 0; j < VH_HASH_BUCKET_SIZE; j++) {
        uint32_t i = j + hpn;
        const VhHashEntry* curr = &hd->d_hash[i];
        if (entry_matches(curr, blk)) {
            if (curr->offset != 0) {
                int prev = atomic_exch(&hd->d_hashBucketMutex[h], VH_LOCK_ENTRY);
                if (prev == VH_LOCK_ENTRY) return 0;
                append_heap(hd, (uint32_t)curr->ptr / VH_SDF_BLOCK_VOXELS);
                int nextIdx = (int)((i + curr->offset) % ne);
                copy_entry(&hd->d_hash[i], &hd->d_hash[nextIdx]);
                delete_hash_entry(&hd->d_hash[nextIdx]);
                return 1;
            } else {
                append_heap(hd, (uint32_t)curr->ptr / VH_SDF_BLOCK_VOXELS);
                delete_hash_entry(&hd->d_hash[i]);
                return 1;
            }
        }
    }
    const uint32_t idxLast = (h + 1) * VH_HASH_BUCKET_SIZE - 1;
    int i = (int)idxLast;
    VhHashEntry curr = hd->d_hash[i];
    int prevIdx = i;
    i = (int)((idxLast + curr.offset) % ne);

    uint32_t maxIter = 0;
    while (maxIter < hp->m_hashMaxCollisionLinkedListSize) {
        curr = hd->d_hash[i];
        if (entry_matches(&curr, blk)) {
            int prev = atomic_exch(&hd->d_hashBucketMutex[h], VH_LOCK_ENTRY);
            if (prev == VH_LOCK_ENTRY) return 0;
            append_heap(hd, (uint32_t)curr.ptr / VH_SDF_BLOCK_VOXELS);
            delete_hash_entry(&hd->d_hash[i]);
            VhHashEntry pe = hd->d_hash[prevIdx];
            pe.offset = curr.offset;
            copy_entry(&hd->d_hash[prevIdx], &pe);
            return 1;
        }
        if (curr.offset == 0) return 0;
        prevIdx = i;
        i = (int)((idxLast + curr.offset) % ne);
        maxIter++;
    }
    return 0;
}

/* getVoxel(float3) :390-400 */
static inline VhVoxel get_voxel_world(const VhHashData* hd, const VhHashParams* hp, f3 worldPos)
{
    VhHashEntry e = get_hash_entry_for_block(hd, hp, world_to_block(hp, worldPos));
    VhVoxel v;
    if (e.ptr == VH_FREE_ENTRY) {
        memset(&v, 0, sizeof(v));
    } else {
        i3 vv = world_to_vvp(hp, worldPos);
        v = hd->d_SDFBlocks[e.ptr + vvp_to_local_index(vv)];
    }
    return v;
}

/* ------------------------------------------------------------------------- */
/* scene-rep launchers (DSC/CUDASceneRepHashSDF.cu)                          */
/* ------------------------------------------------------------------------- */

/* resetCUDA :63-107 = resetHeapKernel :23-41, resetHashKernel :43-51,
 * resetHashBucketMutexKernel :54-61 */
void vho_reset(VhHashData* hd, const VhHashParams* hp)
{
    const uint32_t n = hp->m_numSDFBlocks;
    hd->d_heapCounter[0] = n - 1;
    for (uint32_t idx = 0; idx < n; idx++) hd->d_heap[idx] = n - idx - 1;
    memset(hd->d_SDFBlocks, 0, sizeof(VhVoxel) * (size_t)n * VH_SDF_BLOCK_VOXELS);
    const uint32_t ne = num_entries(hp);
    memset(hd->d_hash, 0, sizeof(VhHashEntry) * ne);
    memset(hd->d_hashCompactified, 0, sizeof(VhHashEntry) * ne);
    for (uint32_t i = 0; i < ne; i++) {
        delete_hash_entry(&hd->d_hash[i]);
        delete_hash_entry(&hd->d_hashCompactified[i]);
    }
    vho_reset_bucket_mutex(hd, hp);
}

/* resetHashBucketMutexCUDA :109-120 */
void vho_reset_bucket_mutex(VhHashData* hd, const VhHashParams* hp)
{
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < hp->m_hashNumBuckets; i++) hd->d_hashBucketMutex[i] = VH_FREE_ENTRY;
}

/* worldToChunks :133-146, linearizeChunkPos :124-130, isSDFBlockStreamedOut :149-156 */
static inline i3 world_to_chunks(const VhHashParams* hp, f3 posWorld)
{
    f3 p = mk3(posWorld.x / hp->m_streamingVoxelExtents[0], posWorld.y / hp->m_streamingVoxelExtents[1],
               posWorld.z / hp->m_streamingVoxelExtents[2]);
    f3 s = mk3((float)signf_i(p.x), (float)signf_i(p.y), (float)signf_i(p.z));
    return mki3(f2i(p.x + s.x * 0.5f), f2i(p.y + s.y * 0.5f), f2i(p.z + s.z * 0.5f));
}
static inline uint32_t linearize_chunk_pos(const VhHashParams* hp, i3 c)
{
    i3 p = mki3(c.x - hp->m_streamingMinGridPos[0], c.y - hp->m_streamingMinGridPos[1], c.z - hp->m_streamingMinGridPos[2]);
    return (uint32_t)(p.z * hp->m_streamingGridDimensions[0] * hp->m_streamingGridDimensions[1] +
                      p.y * hp->m_streamingGridDimensions[0] + p.x);
}
static inline int block_streamed_out(const VhHashParams* hp, i3 blk, const uint32_t* bitMask)
{
    if (!bitMask) return 0; /* streaming disabled: the host always passes an all-zero mask */
    f3 pw = block_to_world(hp, blk);
    uint32_t index = linearize_chunk_pos(hp, world_to_chunks(hp, pw));
    return (bitMask[index / 32] & (0x1u << (index % 32))) != 0;
}

/* Blocks a pass of the multi-threaded baseline build wants and does not find (see vho_alloc). */
typedef struct { i3* v; size_t n, cap; } BlockList;

static void block_list_push(BlockList* l, i3 b)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 256;
        l->v = (i3*)realloc(l->v, l->cap * sizeof(i3));
        if (!l->v) abort();
    }
    l->v[l->n++] = b;
}

/* allocKernel :158-243, one thread = one pixel.  `missing` == NULL: allocate on the spot (the restatement proper).
 * Otherwise the table is only read and the blocks that are wanted and absent are listed. */
static void alloc_pixel(VhHashData* hd, const VhHashParams* hp, const VhDepthCameraData* cam,
                        const VhDepthCameraParams* cp, const uint32_t* bitMask, uint32_t x, uint32_t y, BlockList* missing)
{
    const float vs = hp->m_virtualVoxelSize;
    {
        float d = cam->d_depthData[y * cp->m_imageWidth + x];
        if (d == MINF || d == 0.0f) return;
        if (d >= hp->m_maxIntegrationDistance) return;

        float t = get_truncation(hp, d);
        float minDepth = fminf(hp->m_maxIntegrationDistance, d - t);
        float maxDepth = fminf(hp->m_maxIntegrationDistance, d + t);
        if (minDepth >= maxDepth) return;

        f3 rayMin = mat_mul_p(hp->m_rigidTransform, depth_to_skeleton(cp, x, y, minDepth));
        f3 rayMax = mat_mul_p(hp->m_rigidTransform, depth_to_skeleton(cp, x, y, maxDepth));
        f3 rayDir = normalize3(sub3(rayMax, rayMin));

        i3 id = world_to_block(hp, rayMin);
        i3 idEnd = world_to_block(hp, rayMax);

        f3 step = mk3((float)signf_i(rayDir.x), (float)signf_i(rayDir.y), (float)signf_i(rayDir.z));
        i3 cl = mki3(f2i(fmaxf(0.0f, fminf(step.x, 1.0f))), f2i(fmaxf(0.0f, fminf(step.y, 1.0f))),
                     f2i(fmaxf(0.0f, fminf(step.z, 1.0f))));
        f3 boundaryPos = block_to_world(hp, mki3(id.x + cl.x, id.y + cl.y, id.z + cl.z));
        float half = 0.5f * vs;
        boundaryPos.x -= half; boundaryPos.y -= half; boundaryPos.z -= half;
        f3 tMax = mk3((boundaryPos.x - rayMin.x) / rayDir.x, (boundaryPos.y - rayMin.y) / rayDir.y,
                      (boundaryPos.z - rayMin.z) / rayDir.z);
        f3 tDelta = mk3((step.x * (float)VH_SDF_BLOCK_SIZE * vs) / rayDir.x,
                        (step.y * (float)VH_SDF_BLOCK_SIZE * vs) / rayDir.y,
                        (step.z * (float)VH_SDF_BLOCK_SIZE * vs) / rayDir.z);
        i3 idBound = mki3(f2i((float)idEnd.x + step.x), f2i((float)idEnd.y + step.y), f2i((float)idEnd.z + step.z));

        if (rayDir.x == 0.0f) { tMax.x = PINF; tDelta.x = PINF; }
        if (boundaryPos.x - rayMin.x == 0.0f) { tMax.x = PINF; tDelta.x = PINF; }
        if (rayDir.y == 0.0f) { tMax.y = PINF; tDelta.y = PINF; }
        if (boundaryPos.y - rayMin.y == 0.0f) { tMax.y = PINF; tDelta.y = PINF; }
        if (rayDir.z == 0.0f) { tMax.z = PINF; tDelta.z = PINF; }
        if (boundaryPos.z - rayMin.z == 0.0f) { tMax.z = PINF; tDelta.z = PINF; }

        uint32_t iter = 0;
        while (iter < 1024) {
            if (block_in_frustum(hp, cp, id) && !block_streamed_out(hp, id, bitMask)) {
                if (!missing) alloc_block(hd, hp, id);
                else if (get_hash_entry_for_block(hd, hp, id).ptr == VH_FREE_ENTRY) block_list_push(missing, id);
            }

            if (tMax.x < tMax.y && tMax.x < tMax.z) {
                id.x = f2i((float)id.x + step.x);
                if (id.x == idBound.x) break;
                tMax.x += tDelta.x;
            } else if (tMax.z < tMax.y) {
                id.z = f2i((float)id.z + step.z);
                if (id.z == idBound.z) break;
                tMax.z += tDelta.z;
            } else {
                id.y = f2i((float)id.y + step.y);
                if (id.y == idBound.y) break;
                tMax.y += tDelta.y;
            }
            iter++;
        }
    }
}

/* Threads in raster order.  Built with OpenMP (the bench's all-core CPU baseline, never the checker) the pass has
 * two phases: every pixel walks its ray against the unchanged table and lists the blocks it misses, rows split into
 * contiguous bands; then the lists are allocated serially in band order -- the raster order of the serial pass,
 * since allocBlock on a block that exists changes nothing -- so both builds leave the same table. */
void vho_alloc(VhHashData* hd, const VhHashParams* hp, const VhDepthCameraData* cam,
               const VhDepthCameraParams* cp, const uint32_t* bitMask)
{
#ifdef _OPENMP
    const int nt = omp_get_max_threads();
    BlockList* lists = (BlockList*)calloc((size_t)nt, sizeof(BlockList));
    if (!lists) abort();
#pragma omp parallel num_threads(nt)
    {
        const int tid = omp_get_thread_num();
        const uint32_t H = cp->m_imageHeight;
        const uint32_t y0 = (uint32_t)((uint64_t)H * (uint32_t)tid / (uint32_t)nt), y1 = (uint32_t)((uint64_t)H * ((uint32_t)tid + 1) / (uint32_t)nt);
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = 0; x < cp->m_imageWidth; x++) alloc_pixel(hd, hp, cam, cp, bitMask, x, y, &lists[tid]);
    }
    for (int t = 0; t < nt; t++) {
        for (size_t k = 0; k < lists[t].n; k++) alloc_block(hd, hp, lists[t].v[k]);
        free(lists[t].v);
    }
    free(lists);
#else
    for (uint32_t y = 0; y < cp->m_imageHeight; y++)
        for (uint32_t x = 0; x < cp->m_imageWidth; x++) alloc_pixel(hd, hp, cam, cp, bitMask, x, y, NULL);
#endif
}

/* compactifyHashAllInOneCUDA :361-377 (kernel :317-359; output order is
 * arbitrary in the reference, entry order here). */
uint32_t vho_compactify(VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp)
{
    const uint32_t ne = num_entries(hp);
    int32_t count = 0;
#ifdef _OPENMP
    /* all-core baseline build: contiguous slices of the table, counted then copied, so the output keeps entry order */
    const int nt = omp_get_max_threads();
    int32_t* first = (int32_t*)calloc((size_t)nt + 1, sizeof(int32_t));
    if (!first) abort();
#pragma omp parallel num_threads(nt)
    {
        const int tid = omp_get_thread_num();
        const uint32_t i0 = (uint32_t)((uint64_t)ne * (uint32_t)tid / (uint32_t)nt), i1 = (uint32_t)((uint64_t)ne * ((uint32_t)tid + 1) / (uint32_t)nt);
        int32_t n = 0;
        for (uint32_t idx = i0; idx < i1; idx++) {
            const VhHashEntry* e = &hd->d_hash[idx];
            if (e->ptr != VH_FREE_ENTRY && block_in_frustum(hp, cp, mki3(e->pos[0], e->pos[1], e->pos[2]))) n++;
        }
        first[tid + 1] = n;
#pragma omp barrier
#pragma omp single
        for (int t = 0; t < nt; t++) first[t + 1] += first[t];
        int32_t at = first[tid];
        for (uint32_t idx = i0; idx < i1 && at < first[tid + 1]; idx++) {
            const VhHashEntry* e = &hd->d_hash[idx];
            if (e->ptr != VH_FREE_ENTRY && block_in_frustum(hp, cp, mki3(e->pos[0], e->pos[1], e->pos[2]))) copy_entry(&hd->d_hashCompactified[at++], e);
        }
    }
    count = first[nt];
    free(first);
#else
    for (uint32_t idx = 0; idx < ne; idx++) {
        const VhHashEntry* e = &hd->d_hash[idx];
        if (e->ptr != VH_FREE_ENTRY) {
            if (block_in_frustum(hp, cp, mki3(e->pos[0], e->pos[1], e->pos[2]))) {
                copy_entry(&hd->d_hashCompactified[count], e);
                count++;
            }
        }
    }
#endif
    hd->d_hashCompactifiedCounter[0] = count;
    return (uint32_t)count;
}

/* integrateDepthMapKernel :412-492 */
void vho_integrate(VhHashData* hd, const VhHashParams* hp, const VhDepthCameraData* cam,
                   const VhDepthCameraParams* cp)
{
#pragma omp parallel for schedule(dynamic, 4) /* blocks are independent; the pragma is inert in the checker's build */
    for (uint32_t b = 0; b < hp->m_numOccupiedBlocks; b++) {
        const VhHashEntry entry = hd->d_hashCompactified[b];
        i3 base = block_to_vvp(mki3(entry.pos[0], entry.pos[1], entry.pos[2]));
        for (uint32_t i = 0; i < VH_SDF_BLOCK_VOXELS; i++) {
            /* delinearizeVoxelIndex, DSC/VoxelUtilHashSDF.h:313-318 */
            i3 pi = mki3(base.x + (int)(i % 8), base.y + (int)((i % 64) / 8), base.z + (int)(i / 64));
            f3 pf = mat_mul_p(hp->m_rigidTransformInverse, vvp_to_world(hp, pi));
            float sxf, syf;
            cam_to_screen_float(cp, pf, &sxf, &syf);
            uint32_t sx = (uint32_t)f2i(sxf + 0.5f), sy = (uint32_t)f2i(syf + 0.5f);
            if (sx < cp->m_imageWidth && sy < cp->m_imageHeight) {
                float depth = cam->d_depthData[sy * cp->m_imageWidth + sx];
                float cr = MINF, cg = MINF, cb = MINF;
                if (cam->d_colorData) {
                    const float* c = &cam->d_colorData[4 * (size_t)(sy * cp->m_imageWidth + sx)];
                    cr = c[0]; cg = c[1]; cb = c[2];
                }
                if (cr != MINF && depth != MINF) {
                    if (depth < hp->m_maxIntegrationDistance) {
                        float depthZeroOne = cam_to_proj_z(cp, depth);
                        float sdf = depth - pf.z;
                        float truncation = get_truncation(hp, depth);
                        if (sdf > -truncation) {
                            if (sdf >= 0.0f) sdf = fminf(truncation, sdf);
                            else sdf = fmaxf(-truncation, sdf);
                            float weightUpdate = fmaxf((float)hp->m_integrationWeightSample * 1.5f * (1.0f - depthZeroOne), 1.0f);
                            VhVoxel curr;
                            curr.sdf = sdf;
                            curr.weight = f2uc(weightUpdate);
                            if (cam->d_colorData) {
                                curr.color[0] = f2uc(255.0f * cr);
                                curr.color[1] = f2uc(255.0f * cg);
                                curr.color[2] = f2uc(255.0f * cb);
                            } else {
                                curr.color[0] = 0; curr.color[1] = 255; curr.color[2] = 0;
                            }
                            uint32_t idx = (uint32_t)entry.ptr + i;
                            hd->d_SDFBlocks[idx] = combine_voxel(hp, hd->d_SDFBlocks[idx], curr);
                        }
                    }
                }
            }
        }
    }
}

/* starveVoxelsKernel :512-521 */
void vho_starve(VhHashData* hd, const VhHashParams* hp)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t b = 0; b < hp->m_numOccupiedBlocks; b++) {
        const VhHashEntry* e = &hd->d_hashCompactified[b];
        for (uint32_t i = 0; i < VH_SDF_BLOCK_VOXELS; i++) {
            int w = hd->d_SDFBlocks[e->ptr + i].weight;
            w = (w - 1 > 0) ? w - 1 : 0;
            hd->d_SDFBlocks[e->ptr + i].weight = (uint8_t)w;
        }
    }
}

/* garbageCollectIdentifyKernel :543-590 (min/max tree: order-independent) */
void vho_gc_identify(VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t b = 0; b < hp->m_numOccupiedBlocks; b++) {
        const VhHashEntry* e = &hd->d_hashCompactified[b];
        float minSDF = PINF;
        uint32_t maxWeight = 0;
        for (uint32_t i = 0; i < VH_SDF_BLOCK_VOXELS; i++) {
            VhVoxel v = hd->d_SDFBlocks[e->ptr + i];
            float s = (v.weight == 0) ? PINF : v.sdf;
            minSDF = fminf(minSDF, fabsf(s));
            if (v.weight > maxWeight) maxWeight = v.weight;
        }
        float t = get_truncation(hp, cp->m_sensorDepthWorldMax);
        hd->d_hashDecision[b] = (minSDF >= t || maxWeight == 0) ? 1 : 0;
    }
}

/* garbageCollectFreeKernel :608-628 */
void vho_gc_free(VhHashData* hd, const VhHashParams* hp)
{
    for (uint32_t b = 0; b < hp->m_numOccupiedBlocks; b++) {
        if (hd->d_hashDecision[b] != 0) {
            const VhHashEntry e = hd->d_hashCompactified[b];
            if (delete_hash_entry_element(hd, hp, mki3(e.pos[0], e.pos[1], e.pos[2]))) {
                memset(&hd->d_SDFBlocks[e.ptr], 0, sizeof(VhVoxel) * VH_SDF_BLOCK_VOXELS);
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* ray caster (DSC/RayCastSDFUtil.h, DSC/CUDARayCastSDF.cu)                  */
/* ------------------------------------------------------------------------- */

static inline float fracf_(float v) { return v - floorf(v); } /* RayCastSDFUtil.h:88-95 */

/* trilinearInterpolationSimpleFastFast, DSC/RayCastSDFUtil.h:97-116 */
static int trilinear(const VhHashData* hd, const VhHashParams* hp, f3 pos, float* dist, uint8_t color[3])
{
    const float oSet = hp->m_virtualVoxelSize;
    const f3 posDual = mk3(pos.x - oSet / 2.0f, pos.y - oSet / 2.0f, pos.z - oSet / 2.0f);
    f3 w = mk3(fracf_(pos.x / hp->m_virtualVoxelSize), fracf_(pos.y / hp->m_virtualVoxelSize), fracf_(pos.z / hp->m_virtualVoxelSize));

    /* tap order of the reference: 000,100,010,001,110,011,101,111 */
    static const int tap[8][3] = { {0,0,0},{1,0,0},{0,1,0},{0,0,1},{1,1,0},{0,1,1},{1,0,1},{1,1,1} };
    float d = 0.0f;
    f3 cf = mk3(0.0f, 0.0f, 0.0f);
    for (int k = 0; k < 8; k++) {
        f3 q = mk3(posDual.x + (tap[k][0] ? oSet : 0.0f), posDual.y + (tap[k][1] ? oSet : 0.0f), posDual.z + (tap[k][2] ? oSet : 0.0f));
        VhVoxel v = get_voxel_world(hd, hp, q);
        if (v.weight == 0) { *dist = d; return 0; }
        float wx = tap[k][0] ? w.x : (1.0f - w.x);
        float wy = tap[k][1] ? w.y : (1.0f - w.y);
        float wz = tap[k][2] ? w.z : (1.0f - w.z);
        float s = wx * wy * wz;
        d += s * v.sdf;
        cf.x += s * (float)v.color[0];
        cf.y += s * (float)v.color[1];
        cf.z += s * (float)v.color[2];
    }
    *dist = d;
    color[0] = f2uc(cf.x); color[1] = f2uc(cf.y); color[2] = f2uc(cf.z);
    return 1;
}

/* findIntersectionLinear :140-143 */
static inline float intersect_linear(float tNear, float tFar, float dNear, float dFar)
{
    return tNear + (dNear / (dNear - dFar)) * (tFar - tNear);
}

/* findIntersectionBisection :149-170 */
static int intersect_bisection(const VhHashData* hd, const VhHashParams* hp, f3 camPos, f3 dir,
                               float d0, float r0, float d1, float r1, float* alpha, uint8_t color[3])
{
    float a = r0, aDist = d0, b = r1, bDist = d1, c = 0.0f;
    for (int i = 0; i < 3; i++) {
        c = intersect_linear(a, b, aDist, bDist);
        float cDist;
        if (!trilinear(hd, hp, add3(camPos, scale3(dir, c)), &cDist, color)) return 0;
        if (aDist * cDist > 0.0f) { a = c; aDist = cDist; }
        else { b = c; bDist = cDist; }
    }
    *alpha = c;
    return 1;
}

/* gradientForPoint :174-195 */
static f3 gradient_for_point(const VhHashData* hd, const VhHashParams* hp, f3 pos)
{
    const float vs = hp->m_virtualVoxelSize;
    float dp00 = 0, d0p0 = 0, d00p = 0, d100 = 0, d010 = 0, d001 = 0;
    uint8_t c[3];
    trilinear(hd, hp, mk3(pos.x - 0.5f * vs, pos.y - 0.0f, pos.z - 0.0f), &dp00, c);
    trilinear(hd, hp, mk3(pos.x - 0.0f, pos.y - 0.5f * vs, pos.z - 0.0f), &d0p0, c);
    trilinear(hd, hp, mk3(pos.x - 0.0f, pos.y - 0.0f, pos.z - 0.5f * vs), &d00p, c);
    trilinear(hd, hp, mk3(pos.x + 0.5f * vs, pos.y + 0.0f, pos.z + 0.0f), &d100, c);
    trilinear(hd, hp, mk3(pos.x + 0.0f, pos.y + 0.5f * vs, pos.z + 0.0f), &d010, c);
    trilinear(hd, hp, mk3(pos.x + 0.0f, pos.y + 0.0f, pos.z + 0.5f * vs), &d001, c);
    f3 grad = mk3((dp00 - d100) / vs, (d0p0 - d010) / vs, (d00p - d001) / vs);
    float l = sqrtf(dot3(grad, grad));
    if (l == 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    return mk3(-grad.x / l, -grad.y / l, -grad.z / l);
}

/* renderKernel, DSC/CUDARayCastSDF.cu:18-57 + traverseCoarseGridSimpleSampleAll,
 * DSC/RayCastSDFUtil.h:198-262 */
void vho_render(const VhHashData* hd, const VhHashParams* hp, const VhRayCastData* rd,
                const VhDepthCameraParams* cp, const VhRayCastParams* rp)
{
#pragma omp parallel for schedule(dynamic, 2) /* rays are independent */
    for (uint32_t y = 0; y < rp->m_height; y++)
    for (uint32_t x = 0; x < rp->m_width; x++) {
        size_t pix = (size_t)y * rp->m_width + x;
        rd->d_depth[pix] = MINF;
        for (int k = 0; k < 4; k++) {
            rd->d_depth4[4 * pix + k] = MINF;
            rd->d_normals[4 * pix + k] = MINF;
            rd->d_colors[4 * pix + k] = MINF;
        }
        f3 camDir = normalize3(depth_to_skeleton(cp, x, y, proj_to_cam_z(cp, 1.0f)));
        f3 worldCamPos = mat_mul_p(rp->m_viewMatrixInverse, mk3(0.0f, 0.0f, 0.0f));
        f3 worldDir = normalize3(mat_mul_d(rp->m_viewMatrixInverse, camDir));

        float minInterval = rp->m_minDepth, maxInterval = rp->m_maxDepth;
        if (minInterval == 0 || minInterval == MINF) continue;
        if (maxInterval == 0 || maxInterval == MINF) continue;

        float lastSdf = 0.0f, lastAlpha = 0.0f;
        uint32_t lastWeight = 0;
        const float depthToRayLength = 1.0f / camDir.z;
        float rayCurrent = depthToRayLength * fmaxf(rp->m_minDepth, minInterval);
        float rayEnd = depthToRayLength * fminf(rp->m_maxDepth, maxInterval);

        while (rayCurrent < rayEnd) {
            f3 p = add3(worldCamPos, scale3(worldDir, rayCurrent));
            float dist;
            uint8_t color[3];
            if (trilinear(hd, hp, p, &dist, color)) {
                if (lastWeight > 0 && lastSdf > 0.0f && dist < 0.0f) {
                    float alpha = 0.0f;
                    uint8_t color2[3] = { 0, 0, 0 };
                    int b = intersect_bisection(hd, hp, worldCamPos, worldDir, lastSdf, lastAlpha, dist, rayCurrent, &alpha, color2);
                    f3 currentIso = add3(worldCamPos, scale3(worldDir, alpha));
                    if (b && fabsf(lastSdf - dist) < rp->m_thresSampleDist) {
                        if (fabsf(dist) < rp->m_thresDist) {
                            float depth = alpha / depthToRayLength;
                            rd->d_depth[pix] = depth;
                            f3 sk = depth_to_skeleton(cp, x, y, depth);
                            rd->d_depth4[4 * pix + 0] = sk.x; rd->d_depth4[4 * pix + 1] = sk.y;
                            rd->d_depth4[4 * pix + 2] = sk.z; rd->d_depth4[4 * pix + 3] = 1.0f;
                            rd->d_colors[4 * pix + 0] = (float)color2[0] / 255.f;
                            rd->d_colors[4 * pix + 1] = (float)color2[1] / 255.f;
                            rd->d_colors[4 * pix + 2] = (float)color2[2] / 255.f;
                            rd->d_colors[4 * pix + 3] = 1.0f;
                            if (rp->m_useGradients) {
                                f3 g = gradient_for_point(hd, hp, currentIso);
                                f3 normal = mk3(-g.x, -g.y, -g.z);
                                f3 n = mat_mul_d(rp->m_viewMatrix, normal);
                                rd->d_normals[4 * pix + 0] = n.x; rd->d_normals[4 * pix + 1] = n.y;
                                rd->d_normals[4 * pix + 2] = n.z; rd->d_normals[4 * pix + 3] = 1.0f;
                            }
                            break;
                        }
                    }
                }
                lastSdf = dist;
                lastAlpha = rayCurrent;
                lastWeight = 1;
                rayCurrent += rp->m_rayIncrement;
            } else {
                lastWeight = 0;
                rayCurrent += rp->m_rayIncrement;
            }
        }
    }
}

/* computeNormalsDevice, DSC/CameraUtil.cu:669-697 */
void vho_compute_normals(float* out4, const float* in4, uint32_t width, uint32_t height)
{
#pragma omp parallel for schedule(static)
    for (uint32_t y = 0; y < height; y++)
    for (uint32_t x = 0; x < width; x++) {
        float* o = &out4[4 * ((size_t)y * width + x)];
        o[0] = o[1] = o[2] = o[3] = MINF;
        if (x > 0 && x < width - 1 && y > 0 && y < height - 1) {
            const float* CC = &in4[4 * ((size_t)(y + 0) * width + (x + 0))];
            const float* PC = &in4[4 * ((size_t)(y + 1) * width + (x + 0))];
            const float* CP = &in4[4 * ((size_t)(y + 0) * width + (x + 1))];
            const float* MC = &in4[4 * ((size_t)(y - 1) * width + (x + 0))];
            const float* CM = &in4[4 * ((size_t)(y + 0) * width + (x - 1))];
            if (CC[0] != MINF && PC[0] != MINF && CP[0] != MINF && MC[0] != MINF && CM[0] != MINF) {
                f3 a = mk3(PC[0] - MC[0], PC[1] - MC[1], PC[2] - MC[2]);
                f3 b = mk3(CP[0] - CM[0], CP[1] - CM[1], CP[2] - CM[2]);
                f3 n = mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); /* CUTIL:1318-1321 */
                float l = sqrtf(dot3(n, n));
                if (l > 0.0f) {
                    o[0] = n.x / -l; o[1] = n.y / -l; o[2] = n.z / -l; o[3] = 1.0f;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* streaming launchers (DSC/CUDASceneRepChunkGrid.cu)                        */
/* ------------------------------------------------------------------------- */

/* integrateFromGlobalHashPass1Kernel :27-74.  The launch rounds the part up to
 * whole 64-thread groups and the kernel only guards against the table end, so
 * up to 63 entries of the next part are visited too (:78-82, :35).
 * FENCED: for list / displaced entries the reference pushes the heap a second
 * time after deleteHashEntryElement already did, reading an entry that may
 * have been overwritten (:58-64).  Here the element delete is the only push.
 * Returns the number of descriptors written (the reference's d_outputCounter). */
uint32_t vho_stream_out_pass1(VhHashData* hd, const VhHashParams* hp, uint32_t threadsPerPart,
                              uint32_t start, float radius, const float camPos[3],
                              VhSDFBlockDesc* out, uint32_t outCapacity)
{
    const uint32_t ne = num_entries(hp);
    uint32_t count = 0;
    if (threadsPerPart == 0) return 0;
    uint32_t nthreads = ((threadsPerPart + 63) / 64) * 64;
    for (uint32_t t = 0; t < nthreads; t++) {
        uint32_t bucketID = t + start;
        if (bucketID >= ne) continue;
        VhHashEntry* entry = &hd->d_hash[bucketID];
        i3 epos = mki3(entry->pos[0], entry->pos[1], entry->pos[2]);
        f3 pw = block_to_world(hp, epos);
        f3 df = mk3(pw.x - camPos[0], pw.y - camPos[1], pw.z - camPos[2]);
        float d = sqrtf(dot3(df, df));
        if (entry->ptr != VH_FREE_ENTRY && d >= radius) {
            VhSDFBlockDesc desc;
            desc.pos[0] = epos.x; desc.pos[1] = epos.y; desc.pos[2] = epos.z;
            desc.ptr = entry->ptr;
            if (entry->offset != 0 || hash_pos(hp, epos) != bucketID / VH_HASH_BUCKET_SIZE) {
                if (delete_hash_entry_element(hd, hp, epos)) {
                    if (count < outCapacity) out[count] = desc;
                    count++;
                }
            } else {
                if (count < outCapacity) out[count] = desc;
                count++;
                append_heap(hd, (uint32_t)entry->ptr / VH_SDF_BLOCK_VOXELS);
                delete_hash_entry(entry);
            }
        }
    }
    return count;
}

/* integrateFromGlobalHashPass2Kernel :97-113 */
void vho_stream_out_pass2(VhHashData* hd, const VhHashParams* hp, const VhSDFBlockDesc* descs,
                          VhVoxel* out, uint32_t n)
{
    (void)hp;
    for (uint32_t b = 0; b < n; b++) {
        memcpy(&out[(size_t)b * VH_SDF_BLOCK_VOXELS], &hd->d_SDFBlocks[descs[b].ptr], sizeof(VhVoxel) * VH_SDF_BLOCK_VOXELS);
        memset(&hd->d_SDFBlocks[descs[b].ptr], 0, sizeof(VhVoxel) * VH_SDF_BLOCK_VOXELS);
    }
}

/* chunkToGlobalHashPass1Kernel :143-160; returns the number of failed inserts */
uint32_t vho_stream_in_pass1(VhHashData* hd, const VhHashParams* hp, uint32_t n,
                             uint32_t heapCountPrev, const VhSDFBlockDesc* descs)
{
    uint32_t failed = 0;
    for (uint32_t b = 0; b < n; b++) {
        uint32_t ptr = hd->d_heap[heapCountPrev - b] * VH_SDF_BLOCK_VOXELS;
        VhHashEntry e;
        memset(&e, 0, sizeof(e));
        e.pos[0] = descs[b].pos[0]; e.pos[1] = descs[b].pos[1]; e.pos[2] = descs[b].pos[2];
        e.offset = 0;
        e.ptr = (int32_t)ptr;
        if (!insert_hash_entry(hd, hp, e)) failed++;
    }
    return failed;
}

/* chunkToGlobalHashPass2Kernel :181-189 */
void vho_stream_in_pass2(VhHashData* hd, const VhHashParams* hp, uint32_t n,
                         uint32_t heapCountPrev, const VhSDFBlockDesc* descs, const VhVoxel* blocks)
{
    (void)hp; (void)descs;
    for (uint32_t b = 0; b < n; b++) {
        uint32_t ptr = hd->d_heap[heapCountPrev - b] * VH_SDF_BLOCK_VOXELS;
        memcpy(&hd->d_SDFBlocks[ptr], &blocks[(size_t)b * VH_SDF_BLOCK_VOXELS], sizeof(VhVoxel) * VH_SDF_BLOCK_VOXELS);
    }
}

/* ------------------------------------------------------------------------- */
/* exported single operations / scalar helpers                               */
/* ------------------------------------------------------------------------- */

void vho_alloc_block(VhHashData* hd, const VhHashParams* hp, const int32_t pos[3])
{
    alloc_block(hd, hp, mki3(pos[0], pos[1], pos[2]));
}
int vho_delete_hash_entry_element(VhHashData* hd, const VhHashParams* hp, const int32_t pos[3])
{
    return delete_hash_entry_element(hd, hp, mki3(pos[0], pos[1], pos[2]));
}
int vho_insert_hash_entry(VhHashData* hd, const VhHashParams* hp, const VhHashEntry* e)
{
    return insert_hash_entry(hd, hp, *e);
}
VhHashEntry vho_get_hash_entry(const VhHashData* hd, const VhHashParams* hp, const int32_t pos[3])
{
    return get_hash_entry_for_block(hd, hp, mki3(pos[0], pos[1], pos[2]));
}
uint32_t vho_compute_hash_pos(const VhHashParams* hp, const int32_t pos[3])
{
    return hash_pos(hp, mki3(pos[0], pos[1], pos[2]));
}
void vho_world_to_virtual_voxel_pos(const VhHashParams* hp, const float p[3], int32_t out[3])
{
    i3 r = world_to_vvp(hp, mk3(p[0], p[1], p[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void vho_virtual_voxel_pos_to_sdf_block(const int32_t v[3], int32_t out[3])
{
    i3 r = vvp_to_block(mki3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int vho_is_block_in_frustum(const VhHashParams* hp, const VhDepthCameraParams* cp, const int32_t blk[3])
{
    return block_in_frustum(hp, cp, mki3(blk[0], blk[1], blk[2]));
}
void vho_camera_to_screen_int(const VhDepthCameraParams* cp, const float p[3], int32_t out[2])
{
    float sx, sy;
    cam_to_screen_float(cp, mk3(p[0], p[1], p[2]), &sx, &sy);
    out[0] = f2i(sx + 0.5f); out[1] = f2i(sy + 0.5f);
}
VhVoxel vho_combine_voxel(const VhHashParams* hp, VhVoxel v0, VhVoxel v1) { return combine_voxel(hp, v0, v1); }

/* ------------------------------------------------------------------------- */
/* host classes                                                              */
/* ------------------------------------------------------------------------- */

/* getHeapFreeCount, DSC/CUDASceneRepHashSDF.h:122-126 */
static inline uint32_t heap_free_count(const VhHashData* hd) { return hd->d_heapCounter[0] + 1; }

/* CUDASceneRepHashSDF::integrate :64-83 = setLastRigidTransform :85-88,
 * alloc :247-279, compactifyHashEntries :282-315, integrateDepthMap :317-325,
 * garbageCollect :327-339 */
void vho_scene_integrate(VhHashData* hd, VhHashParams* hp, const VhSceneOptions* opt,
                         uint32_t* numIntegratedFrames, const float rigidTransform[16],
                         const VhDepthCameraData* cam, const VhDepthCameraParams* cp,
                         const uint32_t* bitMask)
{
    memcpy(hp->m_rigidTransform, rigidTransform, sizeof(float) * 16);
    vho_mat4_inverse(hp->m_rigidTransform, hp->m_rigidTransformInverse);

    if (opt->s_offlineProcessing) {
        uint32_t prevFree = heap_free_count(hd);
        for (;;) {
            vho_reset_bucket_mutex(hd, hp);
            vho_alloc(hd, hp, cam, cp, bitMask);
            uint32_t currFree = heap_free_count(hd);
            if (prevFree != currFree) prevFree = currFree;
            else break;
        }
    } else {
        vho_reset_bucket_mutex(hd, hp);
        vho_alloc(hd, hp, cam, cp, bitMask);
    }

    hp->m_numOccupiedBlocks = vho_compactify(hd, hp, cp);
    vho_integrate(hd, hp, cam, cp);

    if (opt->s_garbageCollectionEnabled) {
        if (*numIntegratedFrames > 0 && opt->s_garbageCollectionStarve != 0 &&
            *numIntegratedFrames % opt->s_garbageCollectionStarve == 0) {
            vho_starve(hd, hp);
        }
        vho_gc_identify(hd, hp, cp);
        vho_reset_bucket_mutex(hd, hp);
        vho_gc_free(hd, hp);
    }
    (*numIntegratedFrames)++;
}

/* CUDARayCastSDF::render, DSC/CUDARayCastSDF.cpp:38-72 with
 * rayIntervalSplatting :84-100 (view matrices only; params untouched while no
 * block is in the frustum) */
void vho_raycast_render(const VhHashData* hd, const VhHashParams* hp, const VhRayCastData* rd,
                        const VhDepthCameraParams* cp, VhRayCastParams* rp,
                        const float lastRigidTransform[16])
{
    if (hp->m_numOccupiedBlocks != 0) {
        rp->m_numOccupiedSDFBlocks = hp->m_numOccupiedBlocks;
        vho_mat4_inverse(lastRigidTransform, rp->m_viewMatrix);
        memcpy(rp->m_viewMatrixInverse, lastRigidTransform, sizeof(float) * 16);
    }
    vho_render(hd, hp, rd, cp, rp);
    if (!rp->m_useGradients) vho_compute_normals(rd->d_normals, rd->d_depth4, rp->m_width, rp->m_height);
}

/* ------------------------------------------------------------------------- */
/* synthetic scenes                                 */
/* ------------------------------------------------------------------------- */

void vho_synth_frame(const double* spheres, int nSpheres, int inside, const float T[16],
                     const VhDepthCameraParams* cp, float* depth, float* color4)
{
    const double ox = (double)T[3], oy = (double)T[7], oz = (double)T[11];
    for (uint32_t v = 0; v < cp->m_imageHeight; v++)
    for (uint32_t u = 0; u < cp->m_imageWidth; u++) {
        double dx = ((double)u - (double)cp->mx) / (double)cp->fx;
        double dy = ((double)v - (double)cp->my) / (double)cp->fy;
        double wx = (double)T[0] * dx + (double)T[1] * dy + (double)T[2];
        double wy = (double)T[4] * dx + (double)T[5] * dy + (double)T[6];
        double wz = (double)T[8] * dx + (double)T[9] * dy + (double)T[10];
        double a = wx * wx + wy * wy + wz * wz;
        double bestT = 0.0;
        int best = -1;
        for (int s = 0; s < nSpheres; s++) {
            double cx = spheres[4 * s + 0], cy = spheres[4 * s + 1], cz = spheres[4 * s + 2], r = spheres[4 * s + 3];
            double ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
            double b = ocx * wx + ocy * wy + ocz * wz;
            double c = ocx * ocx + ocy * ocy + ocz * ocz - r * r;
            double disc = b * b - a * c;
            if (disc < 0.0) continue;
            double sq = sqrt(disc);
            double t = inside ? (-b + sq) / a : (-b - sq) / a;
            if (t > 0.0 && (best < 0 || t < bestT)) { bestT = t; best = s; }
        }
        size_t pix = (size_t)v * cp->m_imageWidth + u;
        if (best < 0) {
            depth[pix] = MINF;
            color4[4 * pix + 0] = color4[4 * pix + 1] = color4[4 * pix + 2] = color4[4 * pix + 3] = MINF;
        } else {
            double cx = spheres[4 * best + 0], cy = spheres[4 * best + 1], cz = spheres[4 * best + 2], r = spheres[4 * best + 3];
            double px = ox + bestT * wx, py = oy + bestT * wy, pz = oz + bestT * wz;
            double nx = (px - cx) / r, ny = (py - cy) / r, nz = (pz - cz) / r;
            if (inside) { nx = -nx; ny = -ny; nz = -nz; }
            depth[pix] = (float)bestT;
            color4[4 * pix + 0] = (float)(0.5 + 0.5 * nx);
            color4[4 * pix + 1] = (float)(0.5 + 0.5 * ny);
            color4[4 * pix + 2] = (float)(0.5 + 0.5 * nz);
            color4[4 * pix + 3] = 1.0f;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* marching cubes (DSC/MarchingCubesSDFUtil.h, DSC/CUDAMarchingCubesSDF.cu)  */
/* ------------------------------------------------------------------------- */

#include "../include/vh_mc_tables.h"

/* vertexInterp, DSC/MarchingCubesSDFUtil.h:237-262 */
static VhVertex mc_vertex_interp(float isolevel, f3 p1, f3 p2, float d1, float d2, const uint8_t c1[3], const uint8_t c2[3])
{
    VhVertex r1, r2, res;
    r1.p[0] = p1.x; r1.p[1] = p1.y; r1.p[2] = p1.z;
    r1.c[0] = (float)c1[0] / 255.f; r1.c[1] = (float)c1[1] / 255.f; r1.c[2] = (float)c1[2] / 255.f;
    r2.p[0] = p2.x; r2.p[1] = p2.y; r2.p[2] = p2.z;
    r2.c[0] = (float)c2[0] / 255.f; r2.c[1] = (float)c2[1] / 255.f; r2.c[2] = (float)c2[2] / 255.f;
    if (fabsf(isolevel - d1) < 0.00001f) return r1;
    if (fabsf(isolevel - d2) < 0.00001f) return r2;
    if (fabsf(d1 - d2) < 0.00001f) return r1;
    const float mu = (isolevel - d1) / (d2 - d1);
    res.p[0] = p1.x + mu * (p2.x - p1.x);
    res.p[1] = p1.y + mu * (p2.y - p1.y);
    res.p[2] = p1.z + mu * (p2.z - p1.z);
    res.c[0] = (float)((float)c1[0] + mu * (float)((int)c2[0] - (int)c1[0])) / 255.f;
    res.c[1] = (float)((float)c1[1] + mu * (float)((int)c2[1] - (int)c1[1])) / 255.f;
    res.c[2] = (float)((float)c1[2] + mu * (float)((int)c2[2] - (int)c1[2])) / 255.f;
    return res;
}

/* extractIsoSurfaceAtPosition, DSC/MarchingCubesSDFUtil.h:154-235.  Returns the number of triangles written to t[5]. */
static int mc_at_position(const VhHashData* hd, const VhHashParams* hp, const VhMarchingCubesParams* mp, f3 worldPos, VhTriangle t[5])
{
    if ((mp->m_boxEnabled & 0xffu) == 1u) { /* isInBoxAA :264-271 */
        if (worldPos.x < mp->m_minCorner[0] || worldPos.x > mp->m_maxCorner[0]) return 0;
        if (worldPos.y < mp->m_minCorner[1] || worldPos.y > mp->m_maxCorner[1]) return 0;
        if (worldPos.z < mp->m_minCorner[2] || worldPos.z > mp->m_maxCorner[2]) return 0;
    }
    const float isolevel = 0.0f;
    const float P = hp->m_virtualVoxelSize / 2.0f;
    const float M = -P;
    /* corner order of the reference: 000,100,010,001,110,011,101,111 */
    static const int cs[8][3] = { {0,0,0},{1,0,0},{0,1,0},{0,0,1},{1,1,0},{0,1,1},{1,0,1},{1,1,1} };
    f3 p[8];
    float d[8];
    int valid[8];
    uint8_t cdummy[3];
    for (int k = 0; k < 8; k++) { /* the reference evaluates all eight before testing any */
        p[k] = mk3(worldPos.x + (cs[k][0] ? P : M), worldPos.y + (cs[k][1] ? P : M), worldPos.z + (cs[k][2] ? P : M));
        d[k] = 0.0f;
        valid[k] = trilinear(hd, hp, p[k], &d[k], cdummy);
    }
    for (int k = 0; k < 8; k++) if (!valid[k]) return 0;
    const f3 p000 = p[0], p100 = p[1], p010 = p[2], p001 = p[3], p110 = p[4], p011 = p[5], p101 = p[6], p111 = p[7];
    const float dist000 = d[0], dist100 = d[1], dist010 = d[2], dist001 = d[3], dist110 = d[4], dist011 = d[5], dist101 = d[6], dist111 = d[7];

    uint32_t cubeindex = 0;
    if (dist010 < isolevel) cubeindex += 1;
    if (dist110 < isolevel) cubeindex += 2;
    if (dist100 < isolevel) cubeindex += 4;
    if (dist000 < isolevel) cubeindex += 8;
    if (dist011 < isolevel) cubeindex += 16;
    if (dist111 < isolevel) cubeindex += 32;
    if (dist101 < isolevel) cubeindex += 64;
    if (dist001 < isolevel) cubeindex += 128;

    const float thres = mp->m_threshMarchingCubes;
    for (int k = 0; k < 8; k++)
        for (int l = 0; l < 8; l++) {
            if (d[k] * d[l] < 0.0f) {
                if (fabsf(d[k]) + fabsf(d[l]) > thres) return 0;
            } else {
                if (fabsf(d[k] - d[l]) > thres) return 0;
            }
        }
    for (int k = 0; k < 8; k++) if (fabsf(d[k]) > mp->m_threshMarchingCubes2) return 0;

    const uint32_t edges = VH_MC_EDGE[cubeindex];
    if (edges == 0 || edges == 255) return 0;

    const VhVoxel v = get_voxel_world(hd, hp, worldPos);
    VhVertex vl[12];
    memset(vl, 0, sizeof(vl));
    if (edges & 1)    vl[0]  = mc_vertex_interp(isolevel, p010, p110, dist010, dist110, v.color, v.color);
    if (edges & 2)    vl[1]  = mc_vertex_interp(isolevel, p110, p100, dist110, dist100, v.color, v.color);
    if (edges & 4)    vl[2]  = mc_vertex_interp(isolevel, p100, p000, dist100, dist000, v.color, v.color);
    if (edges & 8)    vl[3]  = mc_vertex_interp(isolevel, p000, p010, dist000, dist010, v.color, v.color);
    if (edges & 16)   vl[4]  = mc_vertex_interp(isolevel, p011, p111, dist011, dist111, v.color, v.color);
    if (edges & 32)   vl[5]  = mc_vertex_interp(isolevel, p111, p101, dist111, dist101, v.color, v.color);
    if (edges & 64)   vl[6]  = mc_vertex_interp(isolevel, p101, p001, dist101, dist001, v.color, v.color);
    if (edges & 128)  vl[7]  = mc_vertex_interp(isolevel, p001, p011, dist001, dist011, v.color, v.color);
    if (edges & 256)  vl[8]  = mc_vertex_interp(isolevel, p010, p011, dist010, dist011, v.color, v.color);
    if (edges & 512)  vl[9]  = mc_vertex_interp(isolevel, p110, p111, dist110, dist111, v.color, v.color);
    if (edges & 1024) vl[10] = mc_vertex_interp(isolevel, p100, p101, dist100, dist101, v.color, v.color);
    if (edges & 2048) vl[11] = mc_vertex_interp(isolevel, p000, p001, dist000, dist001, v.color, v.color);

    int n = 0;
    unsigned long long tri = VH_MC_TRI[cubeindex];
    while ((tri & 0xFull) != 0xFull) {
        t[n].v0 = vl[tri & 0xF];
        t[n].v1 = vl[(tri >> 4) & 0xF];
        t[n].v2 = vl[(tri >> 8) & 0xF];
        tri >>= 12;
        n++;
    }
    return n;
}

uint32_t vho_extract_iso_surface(const VhHashData* hd, const VhHashParams* hp, const VhMarchingCubesParams* mp,
                                 VhTriangle* out, uint32_t maxTriangles)
{
    const uint32_t ne = hp->m_hashNumBuckets * VH_HASH_BUCKET_SIZE;
    uint32_t count = 0;
    for (uint32_t idx = 0; idx < ne; idx++) {
        const VhHashEntry* e = &hd->d_hash[idx];
        if (e->ptr == VH_FREE_ENTRY) continue;
        const i3 base = block_to_vvp(mki3(e->pos[0], e->pos[1], e->pos[2]));
        for (int z = 0; z < VH_SDF_BLOCK_SIZE; z++)
            for (int y = 0; y < VH_SDF_BLOCK_SIZE; y++)
                for (int x = 0; x < VH_SDF_BLOCK_SIZE; x++) {
                    VhTriangle t[5];
                    const int n = mc_at_position(hd, hp, mp, vvp_to_world(hp, mki3(base.x + x, base.y + y, base.z + z)), t);
                    for (int k = 0; k < n; k++) {
                        if (count < maxTriangles) out[count] = t[k];
                        count++;
                    }
                }
    }
    return count;
}

/* ------------------------------------------------------------------------- */
/* sensor pre-processing (DSC/CameraUtil.cu)                                 */
/* ------------------------------------------------------------------------- */

/* convertColorRawToFloatDevice :137-152 */
void vho_convert_color_raw_to_float4(float* out4, const uint8_t* in, uint32_t width, uint32_t height)
{
    for (uint32_t i = 0; i < width * height; i++) {
        const uint8_t r = in[4 * i + 0], g = in[4 * i + 1], b = in[4 * i + 2], w = in[4 * i + 3];
        if (r == 0 && g == 0 && b == 0) {
            out4[4 * i + 0] = out4[4 * i + 1] = out4[4 * i + 2] = out4[4 * i + 3] = MINF;
        } else {
            out4[4 * i + 0] = r / 255.0f; out4[4 * i + 1] = g / 255.0f; out4[4 * i + 2] = b / 255.0f; out4[4 * i + 3] = (float)(w / 255);
        }
    }
}

/* bilinearInterpolationFloat :1071-1098; the int < unsigned comparisons of the reference reject negative coordinates */
static float bilinear_float(float x, float y, const float* in, uint32_t W, uint32_t H)
{
    const int px = (int)floorf(x), py = (int)floorf(y);
    const float alpha = x - (float)px, beta = y - (float)py;
    float s0 = 0.0f, w0 = 0.0f, s1 = 0.0f, w1 = 0.0f;
    if ((uint32_t)px < W && (uint32_t)py < H) { float v = in[(uint32_t)py * W + (uint32_t)px]; if (v != MINF) { s0 += (1.0f - alpha) * v; w0 += (1.0f - alpha); } }
    if ((uint32_t)(px + 1) < W && (uint32_t)py < H) { float v = in[(uint32_t)py * W + (uint32_t)(px + 1)]; if (v != MINF) { s0 += alpha * v; w0 += alpha; } }
    if ((uint32_t)px < W && (uint32_t)(py + 1) < H) { float v = in[(uint32_t)(py + 1) * W + (uint32_t)px]; if (v != MINF) { s1 += (1.0f - alpha) * v; w1 += (1.0f - alpha); } }
    if ((uint32_t)(px + 1) < W && (uint32_t)(py + 1) < H) { float v = in[(uint32_t)(py + 1) * W + (uint32_t)(px + 1)]; if (v != MINF) { s1 += alpha * v; w1 += alpha; } }
    const float p0 = s0 / w0, p1 = s1 / w1;
    float ss = 0.0f, ww = 0.0f;
    if (w0 > 0.0f) { ss += (1.0f - beta) * p0; ww += (1.0f - beta); }
    if (w1 > 0.0f) { ss += beta * p1; ww += beta; }
    return ww > 0.0f ? ss / ww : MINF;
}

/* resampleFloatMapDevice :1100-1118 */
void vho_resample_float_map(float* out, uint32_t outW, uint32_t outH, const float* in, uint32_t inW, uint32_t inH)
{
    const float scaleWidth = (float)(inW - 1) / (float)(outW - 1), scaleHeight = (float)(inH - 1) / (float)(outH - 1);
    for (uint32_t y = 0; y < outH; y++)
        for (uint32_t x = 0; x < outW; x++) {
            const uint32_t xInput = (uint32_t)((float)(int)x * scaleWidth + 0.5f), yInput = (uint32_t)((float)(int)y * scaleHeight + 0.5f);
            if (xInput < inW && yInput < inH) out[y * outW + x] = bilinear_float((float)(int)x * scaleWidth, (float)(int)y * scaleHeight, in, inW, inH);
        }
}

/* bilinearInterpolationFloat4 :1136-1166 */
static void bilinear_float4(float x, float y, const float* in4, uint32_t W, uint32_t H, float out[4])
{
    const int px = (int)floorf(x), py = (int)floorf(y);
    const float alpha = x - (float)px, beta = y - (float)py;
    float s0[4] = { 0, 0, 0, 0 }, s1[4] = { 0, 0, 0, 0 }, w0 = 0.0f, w1 = 0.0f;
    const int tx[4] = { px, px + 1, px, px + 1 }, ty[4] = { py, py, py + 1, py + 1 };
    const float wg[4] = { 1.0f - alpha, alpha, 1.0f - alpha, alpha };
    for (int k = 0; k < 4; k++) {
        if ((uint32_t)tx[k] < W && (uint32_t)ty[k] < H) {
            const float* v = &in4[4 * ((size_t)(uint32_t)ty[k] * W + (uint32_t)tx[k])];
            if (v[0] != MINF && v[1] != MINF && v[2] != MINF) {
                float* s = k < 2 ? s0 : s1;
                for (int c = 0; c < 4; c++) s[c] += wg[k] * v[c];
                if (k < 2) w0 += wg[k]; else w1 += wg[k];
            }
        }
    }
    float ss[4] = { 0, 0, 0, 0 }, ww = 0.0f;
    if (w0 > 0.0f) { for (int c = 0; c < 4; c++) ss[c] += (1.0f - beta) * (s0[c] / w0); ww += (1.0f - beta); }
    if (w1 > 0.0f) { for (int c = 0; c < 4; c++) ss[c] += beta * (s1[c] / w1); ww += beta; }
    for (int c = 0; c < 4; c++) out[c] = ww > 0.0f ? ss[c] / ww : MINF;
}

/* resampleFloat4MapDevice :1168-1186 */
void vho_resample_float4_map(float* out4, uint32_t outW, uint32_t outH, const float* in4, uint32_t inW, uint32_t inH)
{
    const float scaleWidth = (float)(inW - 1) / (float)(outW - 1), scaleHeight = (float)(inH - 1) / (float)(outH - 1);
    for (uint32_t y = 0; y < outH; y++)
        for (uint32_t x = 0; x < outW; x++) {
            const uint32_t xInput = (uint32_t)((float)(int)x * scaleWidth + 0.5f), yInput = (uint32_t)((float)(int)y * scaleHeight + 0.5f);
            if (xInput < inW && yInput < inH) bilinear_float4((float)(int)x * scaleWidth, (float)(int)y * scaleHeight, in4, inW, inH, &out4[4 * ((size_t)y * outW + x)]);
        }
}

/* convertColorToIntensityFloatDevice :258-267 */
void vho_convert_color_to_intensity_float(float* out, const float* in4, uint32_t width, uint32_t height)
{
    for (uint32_t i = 0; i < width * height; i++) out[i] = 0.299f * in4[4 * i] + 0.587f * in4[4 * i + 1] + 0.114f * in4[4 * i + 2];
}

/* convertDepthFloatToCameraSpaceFloat4Device :390-407 */
void vho_convert_depth_float_to_camera_space_float4(float* out4, const float* in, const VhDepthCameraParams* cp, uint32_t width, uint32_t height)
{
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            float* o = &out4[4 * ((size_t)y * width + x)];
            const float depth = in[y * width + x];
            o[0] = o[1] = o[2] = o[3] = MINF;
            if (depth != MINF) {
                const f3 p = depth_to_skeleton(cp, x, y, depth);
                o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = 1.0f;
            }
        }
}

static inline float gauss_d(float sigma, int x, int y) { return expf(-((float)(x * x + y * y) / (2.0f * sigma * sigma))); } /* :436-439 */
static inline double gauss_r(float sigma, float dist) { return exp(-(double)(dist * dist) / (2.0 * (double)sigma * (double)sigma)); } /* :426-429 */

/* gaussFilterFloatMapDevice :555-593 */
void vho_gauss_filter_float_map(float* out, const float* in, float sigmaD, float sigmaR, uint32_t W, uint32_t H)
{
    const int kernelRadius = (int)ceil(2.0 * (double)sigmaD);
    for (int y = 0; y < (int)H; y++)
        for (int x = 0; x < (int)W; x++) {
            float sum = 0.0f, sumWeight = 0.0f;
            const float center = in[y * (int)W + x];
            if (center != MINF)
                for (int m = x - kernelRadius; m <= x + kernelRadius; m++)
                    for (int n = y - kernelRadius; n <= y + kernelRadius; n++)
                        if (m >= 0 && n >= 0 && m < (int)W && n < (int)H) {
                            const float cur = in[n * (int)W + m];
                            if (cur != MINF && fabsf(center - cur) < sigmaR) {
                                const float weight = gauss_d(sigmaD, m - x, n - y);
                                sumWeight += weight;
                                sum += weight * cur;
                            }
                        }
            out[y * (int)W + x] = sumWeight > 0.0f ? sum / sumWeight : MINF;
        }
}

/* gaussFilterFloat4MapDevice :611-651 */
void vho_gauss_filter_float4_map(float* out4, const float* in4, float sigmaD, float sigmaR, uint32_t W, uint32_t H)
{
    const int kernelRadius = (int)ceil(2.0 * (double)sigmaD);
    for (int y = 0; y < (int)H; y++)
        for (int x = 0; x < (int)W; x++) {
            float sum[4] = { 0, 0, 0, 0 }, sumWeight = 0.0f;
            const float* center = &in4[4 * ((size_t)y * W + x)];
            if (center[0] != MINF)
                for (int m = x - kernelRadius; m <= x + kernelRadius; m++)
                    for (int n = y - kernelRadius; n <= y + kernelRadius; n++)
                        if (m >= 0 && n >= 0 && m < (int)W && n < (int)H) {
                            const float* cur = &in4[4 * ((size_t)n * W + m)];
                            if (cur[0] != MINF) {
                                const float dx = center[0] - cur[0], dy = center[1] - cur[1], dz = center[2] - cur[2], dw = center[3] - cur[3];
                                if (sqrtf(dx * dx + dy * dy + dz * dz + dw * dw) < sigmaR) {
                                    const float weight = gauss_d(sigmaD, m - x, n - y);
                                    sumWeight += weight;
                                    for (int c = 0; c < 4; c++) sum[c] += weight * cur[c];
                                }
                            }
                        }
            float* o = &out4[4 * ((size_t)y * W + x)];
            for (int c = 0; c < 4; c++) o[c] = sumWeight > 0.0f ? sum[c] / sumWeight : MINF;
        }
}

/* bilateralFilterFloatMapDevice :446-483 */
void vho_bilateral_filter_float_map(float* out, const float* in, float sigmaD, float sigmaR, uint32_t W, uint32_t H)
{
    const int kernelRadius = (int)ceil(2.0 * (double)sigmaD);
    for (int y = 0; y < (int)H; y++)
        for (int x = 0; x < (int)W; x++) {
            float sum = 0.0f, sumWeight = 0.0f, o = MINF;
            const float center = in[y * (int)W + x];
            if (center != MINF) {
                for (int m = x - kernelRadius; m <= x + kernelRadius; m++)
                    for (int n = y - kernelRadius; n <= y + kernelRadius; n++)
                        if (m >= 0 && n >= 0 && m < (int)W && n < (int)H) {
                            const float cur = in[n * (int)W + m];
                            if (cur != MINF) {
                                const float weight = (float)((double)gauss_d(sigmaD, m - x, n - y) * gauss_r(sigmaR, cur - center));
                                sumWeight += weight;
                                sum += weight * cur;
                            }
                        }
                if (sumWeight > 0.0f) o = sum / sumWeight;
            }
            out[y * (int)W + x] = o;
        }
}

/* erodeDepthMapDevice :1632-1670 */
void vho_erode_depth_map(float* out, const float* in, int structureSize, uint32_t width, uint32_t height, float dThresh, float fracReq)
{
    const int W = (int)width, H = (int)height;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            unsigned int count = 0;
            const float oldDepth = in[y * W + x];
            for (int i = -structureSize; i <= structureSize; i++)
                for (int j = -structureSize; j <= structureSize; j++)
                    if (x + j >= 0 && x + j < W && y + i >= 0 && y + i < H) {
                        const float depth = in[(y + i) * W + (x + j)];
                        if (depth == MINF || depth == 0.0f || fabsf(depth - oldDepth) > dThresh) count++;
                    }
            const unsigned int sum = (unsigned int)((2 * structureSize + 1) * (2 * structureSize + 1));
            out[y * W + x] = ((float)count / (float)sum >= fracReq) ? MINF : oldDepth;
        }
}
