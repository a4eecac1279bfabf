// vh_probe.hip -- measurement kernels that are not part of the path: how fast one SIMD of this machine issues vector
// instructions, as a function of the number of waves it holds.  bench.py's second ceiling for the ray caster (vector
// issue per SIMD) is priced with the figure measured here (tools/valu_issue_probe.py).
#include <hip/hip_runtime.h>

#include "../../include/vh_api.h"
#include "vh_host_util.hpp"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// 32 instructions per loop trip in eight independent chains (a dependent instruction is eight issues behind its source)
template <int MODE>
__global__ __launch_bounds__(256) void k_valu_probe(uint32_t iters, uint4* stamps, float* sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const float m = 1.0f + 1e-7f * (float)lane, c = 1e-9f;
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    float out = 0.0f;
    if (MODE == 0) { // v_fma_f32
        float a0 = 1.0f, a1 = 1.1f, a2 = 1.2f, a3 = 1.3f, a4 = 1.4f, a5 = 1.5f, a6 = 1.6f, a7 = 1.7f;
#pragma unroll 1
        for (uint32_t i = 0; i < iters; i++) {
#define VH_ROUND "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
            asm volatile(VH_ROUND VH_ROUND VH_ROUND VH_ROUND
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
#undef VH_ROUND
        }
        out = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    } else if (MODE == 1) { // v_pk_fma_f32: two fp32 fmas per lane and instruction
        f32x2 a0 = { 1.0f, 1.05f }, a1 = { 1.1f, 1.15f }, a2 = { 1.2f, 1.25f }, a3 = { 1.3f, 1.35f }, a4 = { 1.4f, 1.45f }, a5 = { 1.5f, 1.55f },
              a6 = { 1.6f, 1.65f }, a7 = { 1.7f, 1.75f };
        const f32x2 m2 = { m, m }, c2 = { c, c };
#pragma unroll 1
        for (uint32_t i = 0; i < iters; i++) {
#define VH_ROUND "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n" \
                 "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
            asm volatile(VH_ROUND VH_ROUND VH_ROUND VH_ROUND
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m2), "v"(c2));
#undef VH_ROUND
        }
        out = ((a0.x + a1.y) + (a2.x + a3.y)) + ((a4.x + a5.y) + (a6.x + a7.y));
    } else if (MODE == 2) { // v_add_u32 (integer, full rate)
        uint32_t a0 = lane, a1 = 1u, a2 = 2u, a3 = 3u, a4 = 4u, a5 = 5u, a6 = 6u, a7 = 7u;
        const uint32_t k = lane | 1u;
#pragma unroll 1
        for (uint32_t i = 0; i < iters; i++) {
#define VH_ROUND "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n" \
                 "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
            asm volatile(VH_ROUND VH_ROUND VH_ROUND VH_ROUND
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
#undef VH_ROUND
        }
        out = (float)(((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)));
    } else { // v_mul_lo_u32 (quarter rate on older parts)
        uint32_t a0 = lane | 1u, a1 = 3u, a2 = 5u, a3 = 7u, a4 = 9u, a5 = 11u, a6 = 13u, a7 = 15u;
        const uint32_t k = 2u * lane + 3u;
#pragma unroll 1
        for (uint32_t i = 0; i < iters; i++) {
#define VH_ROUND "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n" \
                 "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
            asm volatile(VH_ROUND VH_ROUND VH_ROUND VH_ROUND
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
#undef VH_ROUND
        }
        out = (float)(((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)));
    }
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0u) {
        const uint32_t w = blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u;
        // {start, end in 100 MHz ticks, s_memtime ticks spent, where: HW_ID[19:0] (wave, SIMD, pipe, CU, SH, SE, TG) | XCC_ID << 20}
        const uint32_t hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        stamps[w] = make_uint4((uint32_t)r0, (uint32_t)r1, (uint32_t)(c1 - c0), (hwid & 0xfffffu) | ((xcc & 0xfu) << 20));
    }
    if (out == 123.456f) *sink = out; // keeps the chains alive
}

} // namespace

extern "C" int vh_debug_valu_probe(uint32_t mode, uint32_t wavesPerSimd, uint32_t iters, uint32_t* d_stamps /* 4 words per wave */, uint32_t* numWaves,
                                   vhStream_t stream)
{
    if (!d_stamps || !numWaves || wavesPerSimd == 0u || wavesPerSimd > 8u || mode > 3u || iters == 0u) return VH_ERR_BAD_ARGUMENT;
    int dev = 0, numCUs = 0;
    VH_HIP(hipGetDevice(&dev));
    VH_HIP(hipDeviceGetAttribute(&numCUs, hipDeviceAttributeMultiprocessorCount, dev));
    // one workgroup = four waves = one wave on each SIMD of a compute unit; wavesPerSimd workgroups per unit, all resident
    const uint32_t groups = (uint32_t)numCUs * wavesPerSimd;
    *numWaves = groups * 4u;
    uint4* st = reinterpret_cast<uint4*>(d_stamps);
    float* sink = reinterpret_cast<float*>(d_stamps); // (never written)
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0u) k_valu_probe<0><<<groups, 256, 0, s>>>(iters, st, sink);
    else if (mode == 1u) k_valu_probe<1><<<groups, 256, 0, s>>>(iters, st, sink);
    else if (mode == 2u) k_valu_probe<2><<<groups, 256, 0, s>>>(iters, st, sink);
    else k_valu_probe<3><<<groups, 256, 0, s>>>(iters, st, sink);
    return vh_last_launch_error();
}
