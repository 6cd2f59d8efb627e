// Shared device helpers of the MFMA conv kernels (gfx950): LDS-DMA wrapper, vector typedefs, bf16 conversions.
#pragma once
#include "common.h"

namespace m355 {

typedef __attribute__((ext_vector_type(8))) short bf16x8;  // 8 bf16 = one 16-byte granule
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

// one 16-byte-per-lane LDS-DMA: LDS destination = lds_dst (wave-uniform) + 16*lane; source = rsrc base + voff + soff.
// Lanes whose offset lies beyond the descriptor's num_records deposit ZEROS (scripts/probes/blds_oob.hip).
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char *lds_dst, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)lds_dst, 16, voff, soff, 0, 0);
}

constexpr unsigned OOB = 0x80000000u;  // beyond num_records of every descriptor (tensors are < 2 GiB, checked on the host)

__device__ __forceinline__ unsigned short f2bf(float f)
{
    // round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

}  // namespace m355
