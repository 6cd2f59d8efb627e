// What does the 1400 W package limit leave of the bf16 MFMA peak, as a function of the LDS operand traffic per MFMA?
// One kernel, 4 waves per workgroup, 2 workgroups per CU resident (8 waves per CU = the conv kernels' occupancy); every wave runs
// MFMA 32x32x16 bf16 back to back on 16 independent accumulators and, per group of G MFMAs, R ds_read_b128 (1 KB each per wave)
// of fresh operands from LDS filled with random bf16 -- the register-blocking ratios a conv kernel can be built with:
//   64x64 wave tile:   4 MFMAs per 4 reads  = 1 KB / MFMA   (k_conv_halo, k_wgrad_halo, k_conv_c8 today)
//   128x64 wave tile:  8 MFMAs per 6 reads  = 0.75 KB / MFMA
//   128x128 wave tile: 16 MFMAs per 8 reads = 0.5 KB / MFMA
//   no LDS at all:     the MFMA pipe alone
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/probes/power_roof.so scripts/probes/power_roof.hip
#include <hip/hip_runtime.h>
#include "../../2dimageto3dmodel_amd/csrc/conv_dma.h"
using namespace m355;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;

// operand data: 0 = random sign and mantissa, FIXED exponent (|v| in [0.5, 1)); 1 = the seed words as they are (the driver passes
// bf16 pairs drawn from N(0, 1): random exponents, what a conv's activations look like)
__device__ int g_raw = 0;
extern "C" int roof_set_raw(int raw) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_raw), &raw, sizeof(int)); }
__device__ __forceinline__ unsigned init_word(unsigned h)
{
    if (g_raw) return h;   // (1 or 2)
    const unsigned lo = (h & 0x807fu) | 0x3f00u, hi = ((h >> 16) & 0x807fu) | 0x3f00u;
    return lo | (hi << 16);
}

template <int NREAD>
__device__ __forceinline__ void reads(u32x4 (&r)[8], unsigned addr)
{
#pragma unroll
    for (int i = 0; i < NREAD; ++i)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(i * 4096));
}
__device__ __forceinline__ void wait8(u32x4 (&r)[8])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
}

// NA x NB operand fragments -> NA*NB MFMAs per group, NA+NB reads per group (NREAD = 0: operands stay in registers)
template <int NA, int NB, bool LDS, int DMAI>
__global__ __launch_bounds__(256, (NA * NB > 8 ? 1 : 2)) void k_roof(float *out, const unsigned *seed, int iters, const unsigned char *src, unsigned srcbytes)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 4096 + 4096 + 4 * 4096];
    const int tid = threadIdx.x, lane = tid & 63;
    // random bf16 in (-2, 2): sign + exponent 126..127 region + random mantissa
    for (int i = tid; i < (8 * 4096 + 4096) / 4; i += 256) {
        unsigned h = seed[(i + blockIdx.x * 977) & 4095];
        reinterpret_cast<unsigned *>(lds)[i] = init_word(h);
    }
    __syncthreads();
    f32x16 acc[NA * NB];
#pragma unroll
    for (int i = 0; i < NA * NB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    u32x4 ra[8], rb[8];
    const unsigned base = (unsigned)(size_t)lds;   // LDS byte address of the array (address space 3 offsets start at the symbol)
    unsigned addr = lane * 16;
    (void)base;
    reads<8>(ra, addr);
    wait8(ra);
#pragma unroll
    for (int i = 0; i < 8; ++i) rb[i] = ra[i];
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, srcbytes, 0x00020000);
    unsigned char *const dst = lds + 9 * 4096 + wave * 4096;
    unsigned goff = (blockIdx.x * 4 + wave) * 4096u + lane * 16u;
    for (int it = 0; it < iters; ++it) {
        addr = (unsigned)((lane ^ (it & 63)) << 4);   // fresh data every iteration (a lane permutation: still conflict free)
        if (DMAI > 0) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#pragma unroll
            for (int q = 0; q < DMAI; ++q) dma16(rs, dst + (q & 3) * 1024, (goff + q * 1024u) & (srcbytes - 1), 0u);
            goff += 4096u * 2048u + DMAI * 1024u;   // the next pass of this wave is somewhere else in the (L2 resident) source
        }
        // two groups per iteration: compute on ra while rb loads, then the reverse (the software pipeline of the real kernels)
        if (LDS) reads<NA + NB>(rb, addr);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[i * NB + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[i]), __builtin_bit_cast(bf16x8, ra[NA + j]),
                                                                         acc[i * NB + j], 0, 0, 0);
        if (LDS) wait8(rb);
        if (LDS) reads<NA + NB>(ra, addr ^ 0x200);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[i * NB + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[i]), __builtin_bit_cast(bf16x8, rb[NA + j]),
                                                                         acc[i * NB + j], 0, 0, 0);
        if (LDS) wait8(ra);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NA * NB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;   // (keeps the accumulators alive)
}

// the 64x64 wave tile with ONE operand (the weights, pre-packed in fragment order: 1 KB contiguous per fragment) loaded straight
// from global / L2 into registers, two iterations ahead -- 0.5 KB of LDS reads + 0.5 KB of vector-memory loads per MFMA
template <int DMAI>
__global__ __launch_bounds__(256, 2) void k_roof_gb(float *out, const unsigned *seed, int iters, const unsigned char *src, unsigned srcbytes)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 4096 + 4096 + 4 * 4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < (8 * 4096 + 4096) / 4; i += 256) {
        unsigned h = seed[(i + blockIdx.x * 977) & 4095];
        reinterpret_cast<unsigned *>(lds)[i] = init_word(h);
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, srcbytes, 0x00020000);
    unsigned char *const dst = lds + 9 * 4096 + wave * 4096;
    unsigned goff = (blockIdx.x * 4 + wave) * 4096u + lane * 16u;
    // weight fragments: every workgroup streams the same sequence (a conv's weights); waves 0,2 / 1,3 share a co half
    unsigned woff = (wave & 1) * 2048u + lane * 16u;
    auto wload = [&](u32x4 (&g)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned o = (woff + (q >> 1) * 4096u + (q & 1) * 1024u) & (srcbytes / 2 - 1);
            g[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0));
        }
        woff += 8192u;
    };
    u32x4 ra[8], rb[8], g0[4], g1[4], g2[4];
    unsigned addr = lane * 16;
    reads<8>(ra, addr);
    wait8(ra);
#pragma unroll
    for (int i = 0; i < 8; ++i) rb[i] = ra[i];
    wload(g0);
    wload(g1);
    int it = 0;
    auto body = [&](u32x4 (&cur)[4], u32x4 (&nxt2)[4]) {
        addr = (unsigned)((lane ^ (it & 63)) << 4);
        if (DMAI > 0) {
#pragma unroll
            for (int q = 0; q < DMAI; ++q) dma16(rs, dst + (q & 3) * 1024, srcbytes / 2 + ((goff + q * 1024u) & (srcbytes / 2 - 1)), 0u);
            goff += 4096u * 2048u + DMAI * 1024u;
        }
        wload(nxt2);
        reads<2>(rb, addr);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[i]), __builtin_bit_cast(bf16x8, cur[j]), acc[i * 2 + j], 0, 0, 0);
        wait8(rb);
        reads<2>(ra, addr ^ 0x200);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[i]), __builtin_bit_cast(bf16x8, cur[2 + j]), acc[i * 2 + j], 0, 0, 0);
        wait8(ra);
        ++it;
    };
    while (it + 3 <= iters) {
        body(g0, g2);
        body(g1, g0);
        body(g2, g1);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

// The schedule a 128x64-wave-tile conv kernel would run (8 waves = 4 pixel groups x 2 channel halves, one workgroup per CU,
// 512 pixels x 128 output channels): per (class, 32-channel chunk) SEGMENT one drain + one barrier, then the DMAs of the next
// segment (halo 36 KB from a streaming source, four taps' weights 32 KB from an L2-resident one) into the other half of a double
// buffered LDS, then 4 taps x 2 k-groups x (6 fragment reads + 8 MFMAs) with no barrier inside.
template <int BAR>
__global__ __launch_bounds__(512, 2) void k_roof_wt(float *out, const unsigned *seed, int iters, const unsigned char *src, unsigned srcbytes,
                                                    const unsigned char *big, unsigned bigbytes)
{
    constexpr int HB = 36 * 1024, WB = 8 * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * HB + 2 * 4 * WB];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < (2 * HB + 8 * WB) / 4; i += 512) {
        unsigned h = seed[(i + blockIdx.x * 977) & 4095];
        reinterpret_cast<unsigned *>(lds)[i] = init_word(h);
    }
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, srcbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)big, 0, bigbytes, 0x00020000);
    unsigned hoff = blockIdx.x * (unsigned)HB + wave * 1024u + lane * 16u;   // streaming: every segment 36 KB further
    unsigned woff = wave * 1024u + lane * 16u;
    u32x4 ra[8], rbb[8];
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        if (g_raw == 2 && (it & 15) == 0) {   // a real kernel starts a new output tile every 16 segments: accumulators of the
#pragma unroll                            // products' own magnitude, not ones that have grown for the whole launch
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = (float)(e & 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // next segment: halo rows (4.5 per wave) + one weight row block per tap
        unsigned char *const hn = lds + (buf ^ 1) * HB, *const wnx = lds + 2 * HB + (buf ^ 1) * 4 * WB;
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (k < 4 || wave < 4) dma16(rb, hn + (8 * k + wave) * 1024, (hoff + k * 8192u) & (bigbytes - 1), 0u);
        hoff += gridDim.x * (unsigned)HB;
#pragma unroll
        for (int t = 0; t < 4; ++t) dma16(rs, wnx + t * WB + wave * 1024, (woff + t * 8192u) & (srcbytes - 1), 0u);
        woff += 32768u;
        // 64-byte pixel / weight-row pitch: the 16-byte k slot is XOR-swizzled with bits 2..3 of the row (16 consecutive rows of
        // one slot then cover all 64 banks once -- the layout the DMA of a real kernel produces by permuting its source chunks)
        const unsigned ha = (unsigned)(buf * HB + wm * 8448 + ((lane & 31) << 6));
        const unsigned wa = (unsigned)(2 * HB + buf * 4 * WB + wn * 4096 + ((lane & 31) << 6));
        const unsigned sw = (lane >> 2) & 3, half = lane >> 5;
        auto rd = [&](u32x4 (&r)[8], int g) {   // fragment reads of group g = 2 * tap + kk
            const int tap = g >> 1, kk = g & 1;
            const unsigned slot = ((kk * 2 + half) ^ sw) << 4;
            const unsigned hx = ha + (tap >> 1) * 2112 + (tap & 1) * 64 + slot, wx = wa + tap * WB + slot;
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(hx), "n"(i * 2112));
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[4 + j]) : "v"(wx), "n"(j * 2048));
        };
        auto wt = [&](u32x4 (&r)[8]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5])); };
        auto mm = [&](u32x4 (&r)[8]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r[4 + j]), __builtin_bit_cast(bf16x8, r[i]), acc[j * 4 + i], 0, 0, 0);
        };
        rd(ra, 0);
        wt(ra);
#pragma unroll
        for (int g = 0; g < 8; g += 2) {
            rd(rbb, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(ra);
            wt(rbb);
            if (g + 2 < 8) rd(ra, g + 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(rbb);
            if (g + 2 < 8) wt(ra);
            if (BAR) __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sacc = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc += acc[i][e];
    if (sacc == 12345.678f) out[blockIdx.x * 512 + tid] = sacc;
}

extern "C" int roof_launch_wt(int bar, float *out, const unsigned *seed, int iters, int grid, hipStream_t st, const void *src, unsigned srcbytes,
                              const void *big, unsigned bigbytes)
{
    if (bar) hipLaunchKernelGGL((k_roof_wt<1>), dim3(grid), dim3(512), 0, st, out, seed, iters, (const unsigned char *)src, srcbytes, (const unsigned char *)big, bigbytes);
    else hipLaunchKernelGGL((k_roof_wt<0>), dim3(grid), dim3(512), 0, st, out, seed, iters, (const unsigned char *)src, srcbytes, (const unsigned char *)big, bigbytes);
    return (int)hipGetLastError();
}

extern "C" int roof_launch(int variant, float *out, const unsigned *seed, int iters, int grid, hipStream_t st, const void *src, unsigned srcbytes)
{
    const unsigned char *sp = (const unsigned char *)src;
#define GO(NA_, NB_, L_, D_) hipLaunchKernelGGL((k_roof<NA_, NB_, L_, D_>), dim3(grid), dim3(256), 0, st, out, seed, iters, sp, srcbytes)
    switch (variant) {
    case 0: GO(2, 2, false, 0); break;
    case 1: GO(2, 2, true, 0); break;
    case 2: GO(4, 2, true, 0); break;
    case 3: GO(4, 4, true, 0); break;
    case 4: GO(4, 4, false, 0); break;
    case 5: GO(2, 2, true, 1); break;    // 8 MFMAs / iteration + 1 KB of DMA  = 0.125 KB / MFMA
    case 6: GO(2, 2, true, 2); break;    // 0.25 KB / MFMA (k_conv_halo's weight + halo traffic)
    case 7: GO(4, 2, true, 2); break;    // 16 MFMAs / iteration + 2 KB = 0.125
    case 8: GO(4, 2, true, 4); break;    // 0.25
    case 9: GO(2, 2, false, 2); break;   // MFMA + DMA, no LDS reads
    case 10: hipLaunchKernelGGL((k_roof_gb<0>), dim3(grid), dim3(256), 0, st, out, seed, iters, sp, srcbytes); break;
    case 11: hipLaunchKernelGGL((k_roof_gb<1>), dim3(grid), dim3(256), 0, st, out, seed, iters, sp, srcbytes); break;
    default: return -1;
    }
#undef GO
    return (int)hipGetLastError();
}
extern "C" int roof_mfmas_per_iter(int variant) { const int t[12] = {8, 8, 16, 32, 32, 8, 8, 16, 16, 8, 8, 8}; return t[variant]; }
