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
