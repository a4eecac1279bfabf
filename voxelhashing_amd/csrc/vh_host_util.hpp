// vh_host_util.hpp -- error plumbing shared by the launchers and host classes.
#pragma once

#include <hip/hip_runtime.h>

// C-ABI convention: 0 ok, <0 = -(hipError_t), >0 = VH_ERR_*
#define VH_HIP(expr)                                 \
    do {                                             \
        hipError_t vh_e_ = (expr);                   \
        if (vh_e_ != hipSuccess) return -(int)vh_e_; \
    } while (0)

#define VH_TRY(expr)                 \
    do {                             \
        int vh_r_ = (expr);          \
        if (vh_r_ != 0) return vh_r_; \
    } while (0)

static inline int vh_last_launch_error()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// Timing one launch by the dispatch's own begin / end time stamps (what rocprofv3's kernel trace reports), instead of a
// pair of event records around it (each record costs the queue a few microseconds and a pair reads 5-7 us more than the
// kernel it brackets: 8.3 against 5.9 us for the fused integrate pass at cfg2).  vh_time_next_launch() arms the calling
// thread; the next launch that goes through VH_LAUNCH_TIMED uses hipExtLaunchKernel with the two events, and
// hipEventElapsedTime(start, stop) is then that kernel's duration.  Declared in include/vh_api.h.
bool vh_take_launch_events(hipEvent_t* start, hipEvent_t* stop);

#include <hip/hip_ext.h>
#define VH_LAUNCH_TIMED(kernel, grid, block, stream, ...)                                            \
    do {                                                                                             \
        hipEvent_t vh_e0_ = nullptr, vh_e1_ = nullptr;                                               \
        if (vh_take_launch_events(&vh_e0_, &vh_e1_))                                                 \
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, vh_e0_, vh_e1_, 0, __VA_ARGS__); \
        else                                                                                         \
            kernel<<<grid, block, 0, stream>>>(__VA_ARGS__);                                         \
    } while (0)
