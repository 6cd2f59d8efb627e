// Shared helpers for libm355.so (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/m355.h"

namespace m355 {

void set_error(const char *fmt, ...);
void note_kernel(const char *name);  // records which kernel family an entry point dispatched to (m355_last_kernel)

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return M355_ERR_LAUNCH;
    }
    return M355_OK;
}

#define M355_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            m355::set_error(__VA_ARGS__);    \
            return M355_ERR_BAD_ARG;         \
        }                                    \
    } while (0)

// in-bounds test of utils/trilinear_interpolation.py:24 -- thresholds are the python doubles
// 0.5-1e-6 / -0.5+1e-6 cast to fp32 (torch casts the wrapped scalar to the tensor dtype)
__device__ __forceinline__ bool in_bounds3(float c0, float c1, float c2)
{
    const float hi = (float)(0.5 - 1e-6), lo = (float)(-0.5 + 1e-6);
    return c0 < hi && c0 > lo && c1 < hi && c1 > lo && c2 < hi && c2 > lo;
}

}  // namespace m355
