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

// XCD-contiguous workgroup id: the dispatcher deals consecutive workgroups to the 8 XCDs in turn (each with its own L2), so
// neighbours in blockIdx.x never share an L2.  This bijection of [0, n) gives XCD x the ids [start_x, start_x + count_x): what
// is neighbouring in the result (the output-channel blocks of one pixel tile, adjacent pixel tiles with overlapping halos) meets
// in one L2.  (k_conv_glds has had it since round 1; the persistent halo kernels fetched 1.4-1.7x their compulsory bytes.)
#ifndef M355_HALO_XCD
#define M355_HALO_XCD 1
#endif
__device__ __forceinline__ int xcd_contiguous_id(int id, int n)
{
    if (!M355_HALO_XCD) return id;
    const int q = n >> 3, r8 = n & 7, xcd = id & 7, idx = id >> 3;
    return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
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

// two fp32 -> packed bf16x2 (lo in bits 15:0), round-to-nearest-even: ONE v_cvt_pk_bf16_f32 instead of ~10 integer ops
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

// launch description shared by the implicit-GEMM kernels (csrc/conv_mfma.hip, csrc/conv_halo.hip)
struct ConvArgs {
    const unsigned short *x;  // bf16 NHWC [N,H,W,Cin]
    const unsigned short *w;  // bf16 [Cout_p][KH][KW][Cin]
    const float *bias;        // [Cout] or null
    void *y;
    int N, H, W, Cin;    // stored input
    int Hl, Wl;          // logical input extent seen by the taps (after the optional x2 upsample)
    int ups;             // 0/1: logical (h,w) reads stored (h>>ups, w>>ups)
    int Ho, Wo, Cout;    // GEMM pixel grid, real output channels
    int CoutP;           // weight rows (Cout rounded up to the N tile)
    int KH, KW, stride, pad_h, pad_w, pad_w_mode;  // W mode 0 zero, 1 replicate, 2 circular; H always zero
    int OH, OW;          // physical output extent
    int oy_mul, oy_off, ox_mul, ox_off;  // physical (oh,ow) = (ho*oy_mul+oy_off, wo*ox_mul+ox_off)
    int Kp;              // flattened K = KH*KW*Cin rounded up to a multiple of 32 (weight row length)
    int y_f32_nchw;      // 0: bf16 NHWC with channel stride Cs; 1: fp32 NCHW
    int Cs;
    float slope;         // epilogue LeakyReLU slope (1 = identity)
    // stride-2 dgrad in one launch (k_conv_glds only): blockIdx.y = output-parity class with its own weight view,
    // pads and output offsets
    const unsigned short *mask_x;  // dgrad only (k_conv_glds): tensor of the output's shape; y *= (mask_x > 0 ? 1 : mask_slope)
    float mask_slope;              //   = the backward of the LeakyReLU that produced this conv's input, folded in
    int fold2;
    // bit-packed LeakyReLU masks (1 bit per activation, [pixel][C/64][2] uint32; word = lane half, bit = 16 j + 4 g + e of the
    // epilogue's register layout -- producer and consumer are the same kernel family): written by a forward with an
    // activation epilogue, read by the consumer's dgrad instead of 2 bytes per element of the activation itself
    unsigned *bits_out;
    const unsigned *bits_in;           // k_conv_halo only: sum each 2x2 block of output pixels before the store (adjoint of the nearest
                         //   x2 upsample folded into the dgrad of an upsample+conv layer); OH, OW are the LOW-res extents
    int lgWo, lgHo;      // log2 of the GEMM pixel grid sides when both are powers of two (else -1): shift/mask decode
    int ncls;
    int cpad_h[4], cpad_w[4], coy[4], cox[4];
    unsigned cls_w_elems;
    // k_conv_halo, plain unguarded epilogue only: [workgroups along the pixel axis][2][Cout] fp32 partial sums of the results
    // and of their squares (batch-norm statistics of the output without a pass over it), see conv_halo_stats_rows
    float *stats;
    // k_conv_glds split-K (small layers, conv_mfma.hip splitk_plan): blockIdx.y = K slice of `sk`; the slices' fp32 accumulators go
    // to skws[slice][M][CoutP] and a finishing pass (k_splitk_finish / k_fold_pad) adds them in slice order
    int sk;
    float *skws;
#ifdef M355_DBG_STAMP
    unsigned *stamp;   // debug build (scripts/stamp_halo.py): where workgroup (0,0) of k_conv_halo dumps its shader-clock stamps
#endif
};

// launch description of the weight-gradient kernels (csrc/conv_mfma.hip, csrc/conv_halo.hip)
struct WgradArgs {
    const unsigned short *x;   // bf16 NHWC [N,H,W,Cin]
    const unsigned short *dy;  // bf16 NHWC [N,Ho,Wo,Cy]
    float *dw;                 // fp32 [Cout][KH][KW][Cin], pre-zeroed
    float *db;                 // k_wgrad_dma / k_wgrad_halo, nullable: fp32 [Cout] += column sums of dy (bias gradient), pre-zeroed
    int N, H, W, Cin, Hl, Wl, ups;
    int Ho, Wo, Cout, Cy;
    int KH, KW, stride, pad_h, pad_w, pad_w_mode;
    int chunk;  // pixels per z-slice (multiple of WK)
    // deterministic form (m355_conv2d_wgrad_det): the partial tiles are accumulated as 64-bit FIXED-POINT integers (integer adds
    // are associative, so the order in which the workgroups' atomics land does not matter); fix = [flag | dw (Cout*K) | db (Cout)],
    // every cell kFixCell integers (wg_accum below)
    long long *fix;
    // k_wgrad_halo, non-deterministic instantiations: per-workgroup PARTIAL ROWS instead of atomics -- part[blockIdx.x][Cout*K + Cout] fp32
    // (every cell of a row has exactly one writer: the split-K replicas of a (co, ci, class) block differ in blockIdx.x only), added in row
    // order by k_wgrad_part_sum: no contention on the weight tile (the 64-way same-address fp32 atomics of D.conv2's classes), no pre-zeroing,
    // and the same bits on every run.  null: accumulate into dw / db (or fix).
    float *part;
    size_t part_stride;
};

// ---- split-K accumulation of the weight-gradient kernels.  Default: fp32 atomics into the zeroed dw (the result depends on
// the order the workgroups finish in: two runs differ in the last bits).  DET: the partial is added to the cell's fixed-point
// integers with 64-bit INTEGER atomics -- integer adds are associative, so the order in which the workgroups' atomics land does not
// matter.  A cell is THREE integers on the grids 1, 2^-50 and 2^-100, and a partial v goes where its bits are:
//   |v| < 1 (every gradient of a healthy run): mid += round(v 2^50); the remainder (exact in fp32) is non-zero only below
//   2^-27 = 7.5e-9, and only then small += round(remainder 2^50) -- ONE atomic per contribution in the common case;
//   |v| >= 1: big += round(v), mid += the (exact) remainder when there is one.
// The triple holds the EXACT sum of every partial above 2^-77 (its last mantissa bit is on the finest grid; below that an absolute
// 8e-31 per contribution) and is converted with one rounding: the result has no preferred gradient magnitude -- Adam is scale free,
// a layer whose gradients are 1e-12 matters as much as any other (rounds 4-5a used ONE cell on the grid 2^-36: everything under
// 1e-4 was quantised at an absolute 1.5e-11).  4096 contributions per cell stay inside int64 (2^50 2^12); a non-finite partial, or
// one beyond 1e8, raises the flag word and the conversion pass then writes NaN (nothing is hidden).
constexpr float kFixMid = 1125899906842624.0f;        // 2^50
constexpr float kFixSmall = 1125899906842624.0f;      // 2^50 (of the remainder, in units of 2^-50: the finest grid is 2^-100)
constexpr double kFixMidInv = 1.0 / 1125899906842624.0;
constexpr double kFixSmallInv = 1.0 / 1125899906842624.0 / 1125899906842624.0;
constexpr int kFixCell = 3;                           // 64-bit integers per cell: (big, mid, small)
// the value of a cell (the caller's cast to float is the one rounding)
__device__ __forceinline__ double fix_value(long long big, long long mid, long long small)
{
    return (double)big + (double)mid * kFixMidInv + (double)small * kFixSmallInv;
}
// workspace layout: [flag | mid of the ncell cells | big of the ncell cells | small of the ncell cells] -- PLANES, the common case's one
// atomic per contribution lands on consecutive 8-byte words for consecutive cells (interleaved triples measured 8 % of the whole
// deterministic step slower: three times the cache lines per wave)
__device__ __forceinline__ double fix_value(const long long *__restrict__ fix, size_t i, size_t ncell)   // cell i behind the flag word
{
    return fix_value(fix[1 + ncell + i], fix[1 + i], fix[1 + 2 * ncell + i]);
}
// ncell: the cells of the launch's workspace (weights + the Cout bias cells behind them)
template <bool DET>
__device__ __forceinline__ void wg_accum(float *dst, long long *fix, size_t idx, float v, size_t ncell)
{
    if constexpr (DET) {
        if (!(fabsf(v) < 1.0e8f)) {
            atomicOr(reinterpret_cast<unsigned long long *>(fix), 1ull);
            return;
        }
        unsigned long long *c = reinterpret_cast<unsigned long long *>(fix) + 1 + idx;
        if (fabsf(v) >= 1.0f) {
            const float b = rintf(v);
            atomicAdd(c + ncell, (unsigned long long)(long long)b);
            v -= b;                   // exact: |v| <= 0.5, a multiple of the original's last bit (>= 2^-23)
        }
        const float t = v * kFixMid, h = rintf(t);
        if (h != 0.0f) atomicAdd(c, (unsigned long long)(long long)h);
        const float r = t - h;        // exact
        if (r != 0.0f) atomicAdd(c + 2 * ncell, (unsigned long long)__float2ll_rn(r * kFixSmall));
    } else {
        atomicAdd(dst + idx, v);
    }
}

}  // namespace m355
