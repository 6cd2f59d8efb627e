// G4 / G5 / G6 heads: the convolutions with 1..4 output channels (conv_final / conv_mesh 64->3 5x5 gan.py:359,364;
// TextureDiscriminator.conv5 512->1 5x5 gan.py:177; MeshDiscriminator.conv4 256->1 5x5 gan.py:65).
//
// As implicit GEMMs these are N = 3 columns wide: an MFMA tile wastes > 80 % of its columns and, worse, the
// gather re-reads every input pixel kh*kw = 25 times through the L1/TA path (measured: 890 us for conv_final at
// batch 64 against a 50 us HBM floor).  They are HBM-bound operators, so this kernel is built around input reuse:
//   * a workgroup owns a 16x16 output tile; per 64-channel chunk the (16+kh-1) x (16+kw-1) input halo is brought
//     into LDS ONCE by the LDS-DMA (W replicate / circular pads resolved on the source index, zero pads and image
//     borders as out-of-range descriptor offsets = zeros), together with the chunk's weights [Cout][taps][64];
//   * all taps are then served from LDS: v_mfma_f32_16x16x32_bf16 with the weights as the A operand (rows = output
//     channels, rows >= Cout are zero registers) and 16 consecutive pixels of a tile row as the B operand;
//     the pixel's 16-byte chunk c sits at slot c ^ ((halo_x >> 1) & 7) so that the ds_read_b128 of 16 neighbouring
//     pixels is bank-conflict free, and every fragment address is a per-lane base + an immediate;
//   * accumulators stay in registers across the channel chunks; fp32 NCHW output (what the heads return) with
//     bias and optional LeakyReLU.
// Algorithmic traffic: input once (x 1.56 halo overlap) + output; LDS-read bound (1 KiB pixel fragment per MFMA).
#include <stdlib.h>

#include <type_traits>

#include "conv_dma.h"

namespace m355 {

constexpr int ST = 16;       // output tile side
constexpr int SMAXCO = 8;    // output channels served (weights chunk <= 8 * 25 * 128 B = 25 KiB of LDS)

struct SmallArgs {
    const unsigned short *x;  // bf16 NHWC [N,H,W,Cin], Cin % 64 == 0
    const unsigned short *w;  // bf16 forward view [rows][Kp], row = output channel, K ordered (kh, kw, ci)
    const float *bias;        // [Cout] or null
    float *y;                 // fp32 NCHW [N,Cout,H,W]   (heads)            -- exactly one of y / yb
    unsigned short *yb;       // bf16 NHWC [N,H,W,Cs]     (dgrad of an 8-channel-input conv: Cout = 8 here)
    int Cs;
    int N, H, W, Cin, Cout, Kp;
    int tiles_x;
    float slope;
    unsigned xbytes, wbytes;
};

template <int KS, int MODE>
__global__ __launch_bounds__(256, 2) void k_conv_smallco(SmallArgs a)
{
    constexpr int HS = ST + KS - 1;               // halo side (20 for 5x5)
    constexpr int HPIX = HS * HS;                 // halo pixels
    constexpr int HINS = (HPIX + 7) / 8;          // DMA instructions for the halo (8 pixels x 128 B each)
    constexpr int HALO_B = HINS * 1024;
    constexpr int TAPS = KS * KS;
    constexpr int WCH = SMAXCO * TAPS * 8;        // 16-byte weight chunks per channel chunk
    constexpr int WINS = (WCH + 63) / 64;
    __shared__ __attribute__((aligned(16))) unsigned char lds[HALO_B + WINS * 1024];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int y0 = ty * ST - KS / 2, x0 = tx * ST - KS / 2;  // image coordinates of halo pixel (0,0)

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);

    // ---- DMA roles.  Halo instruction q = 4i + wave covers halo pixels 8q .. 8q+7; lane l -> pixel 8q + (l>>3), slot l&7.
    constexpr int HI = (HINS + 3) / 4;
    unsigned hoff[HI];
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int P = 8 * (4 * i + wave) + (lane >> 3);
        const int hy = P / HS, hx = P - hy * HS;
        const int gy = y0 + hy;
        int gx = x0 + hx;
        bool ok = (4 * i + wave) < HINS && P < HPIX && (unsigned)gy < (unsigned)a.H;
        if (MODE == 1) gx = min(max(gx, 0), a.W - 1);
        else if (MODE == 2) gx = gx < 0 ? gx + a.W : (gx >= a.W ? gx - a.W : gx);
        ok = ok && (unsigned)gx < (unsigned)a.W;   // (also drops the columns of a tile that overhangs a narrow image)
        const int chunk = (lane & 7) ^ ((hx >> 1) & 7);
        hoff[i] = ok ? (unsigned)(((n * a.H + gy) * a.W + gx) * a.Cin * 2 + chunk * 16) : OOB;
    }
    // weights: chunk index e = 64 j + lane  ->  (co, tap, c8) = (e / (TAPS*8), (e / 8) % TAPS, e % 8)
    unsigned woff[WINS];
#pragma unroll
    for (int j = 0; j < WINS; ++j) {
        const int e = 64 * j + lane;
        const int co = e / (TAPS * 8), r = e - co * (TAPS * 8);
        const int tap = r >> 3, c8 = r & 7;
        woff[j] = (e < WCH && co < a.Cout) ? (unsigned)((co * a.Kp + tap * a.Cin) * 2 + c8 * 16) : OOB;
    }

    // ---- fragment roles (16x16x32: lane -> pixel / out-channel lane&15, k group lane>>4 = 8 channels)
    const int px = lane & 15, kg = lane >> 4;
    int boff[KS][2];  // per (kw, k half): byte offset of this lane's 16-byte chunk inside a halo row
#pragma unroll
    for (int kw = 0; kw < KS; ++kw)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int hx = px + kw;
            boff[kw][h] = hx * 128 + (((h * 4 + kg) ^ ((hx >> 1) & 7)) << 4);
        }
    const unsigned char *hb = lds + (4 * wave) * HS * 128;          // halo row of this wave's first output row
    const unsigned char *wb = lds + HALO_B + (px * TAPS) * 128 + kg * 16;  // weights of out-channel px (valid if px < Cout)
    const bool wok = px < a.Cout;

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.Cin >> 6;
    for (int cc = 0; cc < nchunks; ++cc) {
        if (cc) __syncthreads();  // everyone is done reading the previous chunk
#pragma unroll
        for (int i = 0; i < HI; ++i)
            if (4 * i + wave < HINS) dma16(rx, lds + (4 * i + wave) * 1024, hoff[i], (unsigned)cc * 128u);
#pragma unroll
        for (int j = 0; j < WINS; ++j)
            if ((j & 3) == wave) dma16(rw, lds + HALO_B + j * 1024, woff[j], (unsigned)cc * 128u);
        __syncthreads();  // (drains the DMA)
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
            for (int kw = 0; kw < KS; ++kw)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x8 wf = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (wok) wf = *reinterpret_cast<const bf16x8 *>(wb + (kh * KS + kw) * 128 + h * 64);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const bf16x8 xf = *reinterpret_cast<const bf16x8 *>(hb + (g + kh) * HS * 128 + boff[kw][h]);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[g], 0, 0, 0);
                    }
                }
    }

    // ---- epilogue: acc[g][r] = out channel 4*kg + r, pixel (row 4*wave + g, column px) of the tile
    const int ox = tx * ST + px;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int oy = ty * ST + 4 * wave + g;
        if (oy < a.H && ox < a.W) {
            if (a.yb) {  // 4 consecutive channels of this lane's pixel = one 8-byte store
                if (4 * kg < a.Cout) {
                    uint2 o;
                    o.x = pack_bf16(acc[g][0], acc[g][1]);
                    o.y = pack_bf16(acc[g][2], acc[g][3]);
                    *reinterpret_cast<uint2 *>(a.yb + (((size_t)n * a.H + oy) * a.W + ox) * a.Cs + 4 * kg) = o;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 4 * kg + r;
                    if (co < a.Cout) {
                        float v = acc[g][r] + (a.bias ? a.bias[co] : 0.0f);
                        v = v >= 0.0f ? v : v * a.slope;
                        a.y[(((size_t)n * a.Cout + co) * a.H + oy) * a.W + ox] = v;
                    }
                }
            }
        }
    }
}

// ---- conv_final / conv_mesh (64 -> 1..4 channels, 5x5): the "scatter" form, no LDS at all.
// k_conv_smallco serves every tap of every pixel with its own 1 KiB fragment read from LDS (25 taps x 2 k-halves per 16
// pixels) for an MFMA that uses 3 of its 16 columns: LDS-read bound at 1.7 TB/s of input (147-177 us at batch 64 against a
// 50 us HBM floor).  Here the GEMM runs the other way round: for a ROW of 32 input pixels (one wave; 28 of them are the
// strip's output columns, 2 + 2 halo) ONE pass over the pixel's 64 channels -- 4 B fragments loaded straight from global
// memory, 16 bytes per lane -- is multiplied with ALL 25 taps x Cout weight rows (the A operand: 128 (slot, half) rows
// resident in 64 registers), giving z[tap][co][pixel] for the whole row in 16 MFMAs.  The convolution is then
//     y[r][c][co] = sum_{kh,kw} z_{row r + kh - 2}[kh,kw][co][c + kw - 2]:
// kw - 2 is a LANE shift (pixels are lanes: DPP wave_shl / wave_shr, the same instruction for both halves of the wave because
// the weight rows are ordered so that a register holds the same kw in both) and kh selects which of five rolling output-row
// accumulators takes the value; the wave walks down its strip, and output row q - 2 is complete after input row q.
// Lane halves hold different output channels (co = 2 cp + half), so nothing has to be exchanged between them.
// Per input row: 4 x 16-byte loads, 16 MFMA 32x32x16, ~75 VALU ops, 2 stores -- the kernel is bound by its reads of x.
struct Head5Args {
    const unsigned short *x;  // bf16 NHWC [N,H,W,64]
    const unsigned short *w;  // bf16 forward view [rows][Kp], row = output channel, K ordered (kh, kw, ci)
    const float *bias;        // [Cout] or null
    float *y;                 // fp32 NCHW [N,Cout,H,W]
    unsigned short *yb;       // OUT = 1: bf16 NHWC with 8 channels per pixel, channels 0 .. Cout-1 written, the rest zero
    int N, H, W, Cout, Kp;
    int nsx, nsy, rs;         // strips of 28 columns, row segments of rs rows
    float slope;
    unsigned xbytes, wbytes;
};

template <int I, int N, typename F>
__device__ __forceinline__ void head5_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        head5_for<I + 1, N>(f);
    }
}

template <int S>
__device__ __forceinline__ float lane_from(float v)
{   // value of lane (l + S) for S in -2..2, across the whole wave (the two lanes next to a half boundary receive the other half's)
    if (S == 0) return v;
    int x = __float_as_int(v);
    if (S > 0) {
        x = __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, true);            // wave_shl:1
        if (S == 2) x = __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, true);
    } else {
        x = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true);            // wave_shr:1
        if (S == -2) x = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true);
    }
    return __int_as_float(x);
}

// OUT = 1 (round 5): the same kernel as the INPUT GRADIENT of TextureDiscriminator.conv1 (gan.py:204-213) in the G step: a 5x5 conv of
// the 64-channel dy with the flipped, transposed weights onto the first <= 4 input channels -- the only ones whose gradient
// anything reads (the other four are the batch-constant positional planes of gan.py:9-20) -- written as the 8-channel bf16 NHWC
// pixel the loaders' backward expects (m355_conv2d_dgrad_lead).  k_conv_smallco produced all 8 with half of every MFMA's columns
// unused: 302 us at batch 64 against ~2x the 88 us this form takes for the half-size conv_final.
template <int MODE, int OUT = 0>
__global__ __launch_bounds__(64) void k_head5(Head5Args a)
{
    constexpr int SW = 28;   // output columns of a strip (32 lanes - 2 - 2)
    const int lane = threadIdx.x & 63, c = lane & 31, half = lane >> 5;
    const int item = blockIdx.x, per_img = a.nsx * a.nsy;   // one wave per workgroup: no LDS, nothing shared -- the dispatcher
    if (item >= a.N * per_img) return;                       // spreads the ~2000 waves evenly over the SIMDs
    const int n = item / per_img, rem = item - n * per_img, sy = rem / a.nsx, sx = rem - sy * a.nsx;
    const int r0 = sy * a.rs, r1 = min(a.H, r0 + a.rs);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);

    // ---- weights: A rows m = 32 b + (lane & 31) = 32 b + 8 g + 4 h' + e  <->  D register i = 4 g + e of block b in lane half h'.
    // value index u = 16 b + i: kw = u % 5, slot j = u / 5 (< 10), kh = j % 5, channel pair cp = j / 5, co = 2 cp + h'
    bf16x8 wf[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int g = c >> 3, hp = (c >> 2) & 1, e = c & 3;
        const int u = 16 * b + 4 * g + e, j = u / 5, kw = u - 5 * j, kh = j % 5, co = 2 * (j / 5) + hp;
        const bool ok = j < 10 && co < a.Cout;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const unsigned off = ok ? (unsigned)((co * a.Kp + (kh * 5 + kw) * 64 + 16 * kk + 8 * half) * 2) : OOB;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0);
            wf[b][kk] = __builtin_bit_cast(bf16x8, v);
        }
    }

    // ---- this lane's pixel column (W pad resolved on the index; zero pad = out-of-range offset = zeros)
    const int gx = sx * SW + c - 2;
    int gxc = gx;
    if (MODE == 1) gxc = min(max(gx, 0), a.W - 1);
    else if (MODE == 2) gxc = gx < 0 ? gx + a.W : (gx >= a.W ? gx - a.W : gx);
    const bool col_ok = (unsigned)gxc < (unsigned)a.W;
    const unsigned colb = (unsigned)gxc * 128u + 16u * half;
    auto load_row = [&](int q, bf16x8 (&xf)[4]) {
        const bool ok = col_ok && (unsigned)q < (unsigned)a.H;   // (rows outside the image: zero padding)
        const unsigned base = ok ? (unsigned)((n * a.H + q) * a.W) * 128u + colb : OOB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xf[kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, base, 32 * kk, 0));
    };

    float R[5][2];   // R[k][cp]: output row (current input row) - 2 + k, channel 2 cp + half, column gx
#pragma unroll
    for (int k = 0; k < 5; ++k) R[k][0] = R[k][1] = 0.0f;
    const bool st_col = c >= 2 && c < 2 + SW && gx < a.W;
    float bv[2] = {0.0f, 0.0f};
#pragma unroll
    for (int cp = 0; cp < 2; ++cp)
        if (a.bias && 2 * cp + half < a.Cout) bv[cp] = a.bias[2 * cp + half];

    // rows in flight: a row costs ~0.4 us of MFMA + VALU but ~2 us of memory latency, and a wave has only its own loads to
    // hide it behind -- PD rows (PD x 4 KiB per wave) are requested ahead of the one being multiplied
    constexpr int PD = 4;
    bf16x8 buf[PD][4];
    const int q0 = r0 - 2, qend = r1 + 2;
#pragma unroll
    for (int k = 0; k < PD; ++k) load_row(q0 + k, buf[k]);   // (rows past the segment are rows past the image or unused)
    for (int qb = q0; qb < qend; qb += PD) {
        head5_for<0, PD>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const int q = qb + k;
            if (q < qend) {
                if (q >= 0 && q < a.H) {   // (a zero row contributes nothing)
                    head5_for<0, 4>([&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        f32x16 d;
#pragma unroll
                        for (int i = 0; i < 16; ++i) d[i] = 0.0f;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[b][kk], buf[k][kk], d, 0, 0, 0);
                        head5_for<0, 16>([&](auto ic) {
                            constexpr int i = decltype(ic)::value, u = 16 * b + i;
                            if constexpr (u < 50) {
                                constexpr int j = u / 5, kw = u - 5 * j, kh = j % 5, cp = j / 5;
                                R[4 - kh][cp] += lane_from<kw - 2>(d[i]);
                            }
                        });
                    });
                }
                if (q + PD < qend) load_row(q + PD, buf[k]);
                // output row q - 2 is complete
                const int r = q - 2;
                if (OUT == 1) {
                    // lane c of half 0 holds channels 0 / 2, its partner in half 1 channels 1 / 3 of the same pixel: one 16-byte store
                    float v0 = R[0][0] + bv[0], v1 = R[0][1] + bv[1];
                    v0 = v0 >= 0.0f ? v0 : v0 * a.slope;
                    v1 = v1 >= 0.0f ? v1 : v1 * a.slope;
                    const float p0 = __shfl_xor(v0, 32), p1 = __shfl_xor(v1, 32);
                    if (r >= r0 && r < r1 && st_col && half == 0) {
                        uint4 o;
                        o.x = pack_bf16(v0, p0);
                        o.y = pack_bf16(v1, p1);
                        o.z = 0u; o.w = 0u;
                        *reinterpret_cast<uint4 *>(a.yb + (((size_t)n * a.H + r) * a.W + gx) * 8) = o;
                    }
                } else if (r >= r0 && r < r1 && st_col) {
#pragma unroll
                    for (int cp = 0; cp < 2; ++cp) {
                        const int co = 2 * cp + half;
                        if (co < a.Cout) {
                            float v = R[0][cp] + bv[cp];
                            v = v >= 0.0f ? v : v * a.slope;
                            a.y[(((size_t)n * a.Cout + co) * a.H + r) * a.W + gx] = v;
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) { R[t][0] = R[t + 1][0]; R[t][1] = R[t + 1][1]; }
                R[4][0] = R[4][1] = 0.0f;
            }
        });
    }
}

// ---- weight gradient of the same heads: dw[co][kh][kw][ci] = sum over pixels dy[p][co] * xpad[p + (kh,kw)][ci].
// M = Cout <= 4 rows: as a GEMM over 2 M pixels it is a long skinny reduction; the implicit-GEMM wgrad pads the rows
// to 64 and re-gathers x per tap (1460 us for conv_final at batch 64).  Here, per 16x16 pixel tile and 64-channel
// chunk, the x halo and the dy tile are DMA'd into LDS once; wave w owns input channels 16w..16w+15 of the chunk and
// keeps all kh*kw accumulators (16x16 fp32 tiles, rows = out channels) in registers across the tiles the workgroup
// visits; both operands have the pixel axis as K, so their fragments are ds_read_b64_tr_b16 transpose reads
// (dy^T: [4 px][16 co] blocks, x: [4 px][16 ci] blocks).  One atomicAdd pass at the end.
struct SmallWgArgs {
    const unsigned short *x;   // bf16 NHWC [N,H,W,Cin]
    const unsigned short *dy;  // bf16 NHWC [N,H,W,Cy]
    float *dw;                 // fp32 [Cout][KS][KS][Cin], pre-zeroed
    long long *fix;            // deterministic form: [flag | fixed-point dw] (conv_dma.h wg_accum), else null
    float *part;               // per-workgroup partial sums [gridDim.x][Cout*KS*KS*Cin] (summed in order by k_wgrad_part_sum), else null
    int N, H, W, Cin, Cout, Cy;
    int tiles_x, tiles_y;
    unsigned xbytes, ybytes;
};

typedef short s4w __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4w lds_s4w;

__device__ __forceinline__ bf16x8 tr8(const unsigned char *p0, const unsigned char *p1)
{
    const s4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4w *)p0);
    const s4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4w *)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// halo swizzle of the wgrad kernel: 8 pixels {x..x+3, x+8..x+11} must land on 8 distinct (parity, slot pair)
__device__ __forceinline__ int wg_swz(int hx) { return (((hx >> 1) & 1) << 1) | (((hx >> 3) & 1) << 2); }

template <int KS, int MODE, bool DET = false>
__global__ __launch_bounds__(256, 2) void k_wgrad_smallco(SmallWgArgs a)
{
    constexpr int HS = ST + KS - 1, HPIX = HS * HS, HINS = (HPIX + 7) / 8, HALO_B = HINS * 1024;
    constexpr int TAPS = KS * KS;
    constexpr int DY_B = ST * ST * 32;  // dy tile: 32 bytes (16 channel slots) per pixel
    __shared__ __attribute__((aligned(16))) unsigned char lds[HALO_B + DY_B];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cc = blockIdx.z;  // 64-channel chunk of the input
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, a.ybytes, 0x00020000);

    // ---- fragment roles (16x16x32, K = 32 pixels = 2 tile rows): lane group kgp = lane>>4 takes row kgp>>1,
    // columns 8*(kgp&1) .. +7 of the pair of rows; inside the group lane q supplies pixel q>>2 of a 4-pixel run,
    // channel quad q&3; the group's 16 lanes receive channel (lane&15) of those 4 pixels.
    const int q = lane & 15, kgp = lane >> 4;
    const int frow = kgp >> 1, fx = 8 * (kgp & 1) + (q >> 2);  // + 4*rd
    // dy tile position of pixel (r, x): (r*16 + (x ^ ((x>>3)&1)<<2)) * 32 bytes
    int dyo[2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int x = fx + 4 * rd;
        dyo[rd] = HALO_B + (frow * 16 + (x ^ (((x >> 3) & 1) << 2))) * 32 + (q & 3) * 8;
    }
    // x halo offsets per (kw, rd): pixel column fx + 4rd + kw, chunk = 2*wave + ((q&3)>>1), 8-byte half q&1
    int xo[KS][2];
#pragma unroll
    for (int kw = 0; kw < KS; ++kw)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int hx = fx + 4 * rd + kw;
            xo[kw][rd] = (frow * HS + hx) * 128 + (((2 * wave + ((q & 3) >> 1)) ^ wg_swz(hx)) << 4) + (q & 1) * 8;
        }

    f32x4 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = a.tiles_x * a.tiles_y * a.N;
    constexpr int HI = (HINS + 3) / 4;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (a.tiles_x * a.tiles_y), tr_ = tile - n * (a.tiles_x * a.tiles_y);
        const int ty = tr_ / a.tiles_x, tx = tr_ - ty * a.tiles_x;
        const int y0 = ty * ST - KS / 2, x0 = tx * ST - KS / 2;
        if (tile != (int)blockIdx.x) __syncthreads();  // the previous tile's fragments are consumed
        // halo (as in k_conv_smallco, with this kernel's swizzle)
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            if (4 * i + wave < HINS) {
                const int P = 8 * (4 * i + wave) + (lane >> 3);
                const int hy = P / HS, hx = P - hy * HS;
                const int gy = y0 + hy;
                int gx = x0 + hx;
                bool ok = P < HPIX && (unsigned)gy < (unsigned)a.H;
                if (MODE == 1) gx = min(max(gx, 0), a.W - 1);
                else if (MODE == 2) gx = gx < 0 ? gx + a.W : (gx >= a.W ? gx - a.W : gx);
                ok = ok && (unsigned)gx < (unsigned)a.W;
                const int chunk = (lane & 7) ^ wg_swz(hx);
                const unsigned off = (unsigned)(((n * a.H + gy) * a.W + gx) * a.Cin * 2 + cc * 128 + chunk * 16);
                dma16(rx, lds + (4 * i + wave) * 1024, ok ? off : OOB, 0u);
            }
        }
        // dy tile: 512 16-byte slots (pixel position pp = e>>1, channel half e&1); 8 DMA instructions, 2 per wave
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = 64 * (4 * j + wave) + lane;
            const int pp = e >> 1, hf = e & 1;
            const int r = pp >> 4, xs = pp & 15, x = xs ^ (((xs >> 3) & 1) << 2);  // involution: position -> pixel column
            const int gy = ty * ST + r, gx = tx * ST + x;
            const bool ok = gy < a.H && gx < a.W && hf * 8 < a.Cy;
            const unsigned off = (unsigned)(((n * a.H + gy) * a.W + gx) * a.Cy * 2 + hf * 16);
            dma16(ry, lds + HALO_B + (4 * j + wave) * 1024, ok ? off : OOB, 0u);
        }
        __syncthreads();
#pragma unroll 1
        for (int kb = 0; kb < ST / 2; ++kb) {  // K block = tile rows 2kb, 2kb+1
            const bf16x8 dyf = tr8(lds + dyo[0] + kb * 2 * 16 * 32, lds + dyo[1] + kb * 2 * 16 * 32);
            const unsigned char *hrow = lds + kb * (2 * HS * 128);
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) {
                    const unsigned char *hb = hrow + kh * HS * 128;
                    const bf16x8 xf = tr8(hb + xo[kw][0], hb + xo[kw][1]);
                    acc[kh * KS + kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dyf, xf, acc[kh * KS + kw], 0, 0, 0);
                }
        }
    }
    // acc[tap][r]: out channel 4*kgp + r, input channel cc*64 + 16*wave + (lane&15)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 4 * kgp + r;
            if (co < a.Cout) {
                const size_t e = ((size_t)co * TAPS + t) * a.Cin + cc * 64 + 16 * wave + (lane & 15);
                if (a.part) a.part[(size_t)blockIdx.x * ((size_t)a.Cout * TAPS * a.Cin) + e] = acc[t][r];
                else wg_accum<DET>(a.dw, a.fix, e, acc[t][r], (size_t)a.Cout * TAPS * a.Cin + a.Cout);
            }
        }
}

// dw[e] = sum over the rows of part[row][e], e < n; optionally the same for a bias tail.  A FIXED summation tree, the same on every run: a
// workgroup owns 256 / RG outputs; RG groups of its threads take 1 / RG of the rows each (four interleaved accumulators per thread, added
// in a fixed order), the groups are added in order through LDS.  RG = 4 for many rows (round 6: one thread per output walking all rows
// -- RG = 1, n / 256 workgroups -- had 51 workgroups and four loads in flight per thread for D.conv1's 256 rows of 12.9 k floats: 22 us
// for 13 MB; per-launch constant of that layer 36 -> 22 us, of the sub-pixel fold 31 -> 17 us); RG = 1 below 16 rows (D.conv4's 4 rows
// of 2 M floats are bandwidth, not latency).
template <int RG>
__global__ __launch_bounds__(256) void k_wgrad_part_sum(const float *__restrict__ part, int nwg, size_t stride, float *__restrict__ dw, size_t n,
                                                        float *__restrict__ db, int nb)
{
    constexpr int OUT = 256 / RG;
    __shared__ float red[RG][OUT];
    const int o = threadIdx.x % OUT, rg = threadIdx.x / OUT;
    const size_t e = blockIdx.x * (size_t)OUT + o, total = n + (db ? (size_t)nb : 0);
    const int q = (nwg + RG - 1) / RG, g0 = rg * q, g1 = min(nwg, g0 + q);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (e < total) {
        int g = g0;
        for (; g + 4 <= g1; g += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += part[(size_t)(g + u) * stride + e];
        }
        for (; g < g1; ++g) s[0] += part[(size_t)g * stride + e];
    }
    float t = (s[0] + s[1]) + (s[2] + s[3]);
    if (RG > 1) {
        red[rg][o] = t;
        __syncthreads();
        if (rg != 0) return;
        t = red[0][o];
#pragma unroll
        for (int k = 1; k < RG; ++k) t += red[k][o];
    }
    if (e < total) {
        if (e < n) dw[e] = t;
        else db[e - n] = t;
    }
}
// (the choice is a function of the row count only: the same tree on every run)
static void part_sum_launch(const float *part, int rows, size_t stride, float *dw, size_t n, float *db, int nb, hipStream_t st)
{
    const size_t total = n + (db ? (size_t)nb : 0);
    if (rows >= 16) hipLaunchKernelGGL(k_wgrad_part_sum<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, part, rows, stride, dw, n, db, nb);
    else hipLaunchKernelGGL(k_wgrad_part_sum<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, rows, stride, dw, n, db, nb);
}

bool wgrad_small_eligible(const m355_conv_desc *d, int Cy)
{
    return d->Cout <= SMAXCO && d->stride == 1 && d->upsample == 0 && d->kh == d->kw && (d->kh == 5 || d->kh == 3) &&
           d->pad_h == d->kh / 2 && d->pad_w == d->kw / 2 && d->Cin % 64 == 0 && Cy % 8 == 0 &&
           (size_t)d->N * d->H * d->W * d->Cin * 2 < (1ull << 31) && (size_t)d->N * d->H * d->W * Cy * 2 < (1ull << 31);
}

// workgroups of the small-Cout wgrad along the tile axis (each a row of the partial-sum workspace)
static int wgrad_small_gx(const m355_conv_desc *d)
{
    const int ntiles = ((d->W + ST - 1) / ST) * ((d->H + ST - 1) / ST) * d->N, nchunks = d->Cin / 64;
    int gx = (512 + nchunks - 1) / nchunks;  // ~512 workgroups (2 per CU)
    return gx > ntiles ? ntiles : gx;
}
size_t wgrad_small_ws_floats(const m355_conv_desc *d) { return (size_t)wgrad_small_gx(d) * d->Cout * d->kh * d->kw * d->Cin; }

int wgrad_small_launch(const m355_conv_desc *d, const void *x, const void *dy, int Cy, float *dw, hipStream_t st, long long *fix,
                       float *part)
{
    SmallWgArgs a = {};
    a.x = (const unsigned short *)x;
    a.dy = (const unsigned short *)dy;
    a.dw = dw;
    a.fix = fix;
    a.part = part;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Cy = Cy;
    a.tiles_x = (d->W + ST - 1) / ST;
    a.tiles_y = (d->H + ST - 1) / ST;
    a.xbytes = (unsigned)((size_t)d->N * d->H * d->W * d->Cin * 2);
    a.ybytes = (unsigned)((size_t)d->N * d->H * d->W * Cy * 2);
    const int nchunks = d->Cin / 64, gx = wgrad_small_gx(d);
    const dim3 grid(gx, 1, nchunks);
#define M355_SW(KS_, DET_)                                                                                               \
    do {                                                                                                                 \
        if (d->pad_w_mode == 0) hipLaunchKernelGGL((k_wgrad_smallco<KS_, 0, DET_>), grid, dim3(256), 0, st, a);          \
        else if (d->pad_w_mode == 1) hipLaunchKernelGGL((k_wgrad_smallco<KS_, 1, DET_>), grid, dim3(256), 0, st, a);     \
        else hipLaunchKernelGGL((k_wgrad_smallco<KS_, 2, DET_>), grid, dim3(256), 0, st, a);                             \
    } while (0)
    if (fix) {
        if (d->kh == 5) M355_SW(5, true);
        else M355_SW(3, true);
    } else if (d->kh == 5) M355_SW(5, false);
    else M355_SW(3, false);
#undef M355_SW
    if (part) {
        const size_t n = (size_t)d->Cout * d->kh * d->kw * d->Cin;
        part_sum_launch((const float *)part, gx, n, dw, n, nullptr, 0, st);
    }
    note_kernel("k_wgrad_smallco");
    return check_launch("conv2d_wgrad (small Cout)");
}

// ---- the OTHER thin layer: 8 input channels -> 64*k output channels, 5x5 (TextureDiscriminator.conv1 on the
// [texture(3) | alpha(1) | positional(4)] image, gan.py:163).  K = 25 taps x 8 channels = 200: as an implicit GEMM the
// k_conv_glds launch spends its time in per-tile prologues / epilogues (4 K steps per tile) and runs at 260 TF,
// 3.5x off the HBM floor (1.07 GB of bf16 output per launch at batch 128).  Here:
//   * a pixel is ONE 16-byte chunk; the (8+4) x (32+4) halo of an 8 x 32 output tile is 6.9 KB, double buffered and
//     prefetched one tile ahead; the whole weight matrix [64][26 taps x 8] stays in LDS (row pitch 432 B = 16 x odd:
//     conflict-free fragment reads) for the lifetime of the persistent workgroup;
//   * one MFMA 32x32x16 K step = two taps: lane half h reads the pixel chunk of tap 2ks+h -- a single ds_read_b128;
//     13 K steps, 52 MFMAs per wave and tile; epilogue = k_conv_glds's (bias, LeakyReLU, permlane32 16-byte stores).
struct C8Args {
    const unsigned short *x;  // bf16 NHWC [N,H,W,8]
    const unsigned short *w;  // bf16 forward view [rows][Kp], K ordered (kh, kw, ci): 200 real columns
    const float *bias;
    unsigned short *y;        // bf16 NHWC [N,H,W,Cout]
    unsigned *bits;           // optional activation-sign bits [pixel][Cout/64][2] (conv_dma.h: ConvArgs::bits_out)
    const unsigned short *mask;  // optional bf16 NHWC [N,H,W,Cout]: y *= (mask > 0 ? 1 : mask_slope) -- the LeakyReLU backward of
    float mask_slope;            // the tensor this conv's output is the gradient of (dgrad of a head, HeadConvFn in_slope)
    int N, H, W, Cout, Kp;
    float slope;
    unsigned xbytes, wbytes;
#ifdef M355_DBG_STAMP_C8
    unsigned *stamp;   // debug build (scripts/stamp_c8.py): [4 waves][64 tiles][4] shader-clock stamps of workgroup 0
#endif
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_conv_c8(C8Args a)
{
    constexpr int TH = 8, TW = 32, KS = 5, HS_Y = TH + KS - 1, HS_X = TW + KS - 1;  // 12 x 36 halo
    constexpr int HPIX = HS_Y * HS_X, HINS = (HPIX + 63) / 64;                      // 432 chunks, 7 DMA instructions
    constexpr int HBUF = HINS * 1024;
    constexpr int WPITCH = 432, WCH = 26;                                            // weight row: 26 chunks (taps 0..25)
    constexpr int WINS = (64 * 27 + 63) / 64;                                        // 27 DMA instructions ([co][27-chunk] image)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * HBUF + WINS * 1024 + 4 * 2048];
    unsigned char *const ldsW = lds + 2 * HBUF;
    unsigned char *const ldsS = ldsW + WINS * 1024;   // epilogue stage: 2 KB per wave

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nN = a.Cout / 64, tn = blockIdx.x % nN, bp = blockIdx.x / nN, PS = gridDim.x / nN;
    const int tpx = a.W / TW, tpy = a.H / TH, tiles = a.N * tpx * tpy;
    if (bp >= tiles) return;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);

    // weights: chunk e = 64 j + lane -> (co, c) = (e / 26, e % 26) -> LDS co*432 + c*16.  The DMA destination is
    // lane-linear (1 KB per instruction), so the 432-byte pitch is produced by giving every (co, c) its own instruction
    // slot in a [co][27-chunk] image: e' = co*27 + c, chunk 26 of each row is padding (never read).
    for (int j = wave; j < WINS; j += 4) {
        const int e = 64 * j + lane, co = e / 27, c = e - co * 27;
        const bool ok = co < 64 && c < WCH;
        dma16(rw, ldsW + j * 1024, ok ? (unsigned)(((tn * 64 + co) * a.Kp) * 2 + c * 16) : OOB, 0u);
    }

    // Tile coordinates (image, tile row, tile column) are carried along and stepped by the workgroup's stride: three run-time integer
    // divisions per tile (there is no divide instruction: ~100 cycles each, twice per tile here) were a measurable part of the
    // ~10 k cycles a tile takes (scripts/stamp_c8.py).
    struct TileAt {
        int n, ty, tx;
    };
    TileAt stepv;   // the stride PS as (images, rows, columns)
    {
        const int per_img = tpx * tpy;
        stepv.n = PS / per_img;
        const int r = PS - stepv.n * per_img;
        stepv.ty = r / tpx;
        stepv.tx = r - stepv.ty * tpx;
    }
    auto at = [&](int tp) {
        TileAt t;
        t.n = tp / (tpx * tpy);
        const int r = tp - t.n * (tpx * tpy);
        t.ty = r / tpx;
        t.tx = r - t.ty * tpx;
        return t;
    };
    auto advance = [&](TileAt t) {
        t.tx += stepv.tx;
        const int cx = t.tx >= tpx ? 1 : 0;
        t.tx -= cx * tpx;
        t.ty += stepv.ty + cx;
        const int cy = t.ty >= tpy ? 1 : 0;
        t.ty -= cy * tpy;
        t.n += stepv.n + cy;
        return t;
    };
    // halo chunk P = 64 q + lane of a tile -> image pixel; instructions q = wave, wave + 4
    auto issue_halo = [&](const TileAt &t, int buf) {
        const int n = t.n, oy0 = t.ty * TH, ox0 = t.tx * TW;
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const int q = wave + 4 * qi;
            if (q < HINS) {
                const int P = 64 * q + lane;
                const int hy = P / HS_X, hx = P - hy * HS_X;
                const int gy = oy0 - KS / 2 + hy;
                int gx = ox0 - KS / 2 + hx;
                bool ok = P < HPIX && (unsigned)gy < (unsigned)a.H;
                if (MODE == 1) gx = min(max(gx, 0), a.W - 1);
                else if (MODE == 2) gx = gx < 0 ? gx + a.W : (gx >= a.W ? gx - a.W : gx);
                ok = ok && (unsigned)gx < (unsigned)a.W;
                dma16(rx, lds + buf * HBUF + q * 1024, ok ? (unsigned)(((n * a.H + gy) * a.W + gx) * 16) : OOB, 0u);
            }
        }
    };

    // fragment roles: lane -> pixel column tx / output channel row (lane & 31), half -> which of the step's two taps
    const int tx = lane & 31, half = lane >> 5;
    int poff[13];  // byte offset of this lane's tap inside the halo, relative to its pixel, per K step
#pragma unroll
    for (int ks = 0; ks < 13; ++ks) {
        const int tap = min(2 * ks + half, 24);  // (tap 25 multiplies zero weights)
        poff[ks] = ((tap / KS) * HS_X + tap % KS) * 16;
    }
    const unsigned char *wfrag = ldsW + (lane & 31) * WPITCH + half * 16;

    f32x16 acc[2][2];  // [co block j][pixel row i]
    // The bias lives in registers for the whole kernel and is the accumulators' initial value.  (A global load inside the tile
    // loop -- the epilogue used to fetch the bias per tile -- is awaited with an in-order vmcnt: behind the NEXT tile's halo
    // DMA issued just before it, i.e. one full HBM round trip per tile with nothing to overlap it; the counters showed the
    // waves parked on s_waitcnt half of the time, profiles/r02_pmc_sq_c8.txt.)
    float4 bias_r[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bias_r[j][g] = a.bias ? *reinterpret_cast<const float4 *>(a.bias + tn * 64 + 32 * j + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef M355_DBG_STAMP_C8
    int dbg_i = 0;
    const bool dbg_on = a.stamp != nullptr && blockIdx.x == 0;
#define C8_STAMP(slot) do { if (dbg_on && dbg_i < 64 && lane == 0) a.stamp[(wave * 64 + dbg_i) * 4 + (slot)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define C8_STAMP(slot) do { } while (0)
#endif
    int tp = bp, buf = 0;
    TileAt cur = at(tp), nxt = advance(cur);
    issue_halo(cur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // weights + first halo (the in-loop wait only covers later halos)
    for (;;) {
        const int n = cur.n, oy0 = cur.ty * TH, ox0 = cur.tx * TW;
        const int tp_next = tp + PS;
        // the halo of this tile (issued one tile ago) and, first time round, the weights have landed; the 8 epilogue
        // stores of the previous tile are younger and may still be in flight
        // (vmcnt retires in order: everything but the previous tile's epilogue stores -- 8, or 10 with the bit masks)
        C8_STAMP(0);
        if (a.bits) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        C8_STAMP(1);
        // (dgrad of a head) the mask vectors of all eight store units are requested NOW, ahead of the next tile's halo DMA:
        // they are older than it in the in-order vmcnt, and have the whole MFMA phase to arrive
        uint4 mk[2][2][2];
        if (a.mask) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int u = lane + 64 * k, p = u >> 2, lc = (u & 3) ^ ((p >> 1) & 3);
                        const size_t pix0m = ((size_t)n * a.H + (oy0 + 2 * wave + i)) * a.W + ox0;
                        mk[i][j][k] = *reinterpret_cast<const uint4 *>(a.mask + (pix0m + p) * a.Cout + tn * 64 + 32 * j + 8 * lc);
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tp_next < tiles) issue_halo(nxt, buf ^ 1);
        else issue_halo(cur, buf ^ 1);  // keep the DMA count per tile constant (harmless re-load into the idle buffer)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[j][i][4 * g] = bias_r[j][g].x; acc[j][i][4 * g + 1] = bias_r[j][g].y;
                    acc[j][i][4 * g + 2] = bias_r[j][g].z; acc[j][i][4 * g + 3] = bias_r[j][g].w;
                }
        const unsigned char *hb = lds + buf * HBUF + ((2 * wave) * HS_X + tx) * 16;
#pragma unroll
        for (int ks = 0; ks < 13; ++ks) {
            const bf16x8 w0 = *reinterpret_cast<const bf16x8 *>(wfrag + ks * 32);
            const bf16x8 w1 = *reinterpret_cast<const bf16x8 *>(wfrag + 32 * WPITCH + ks * 32);
            const bf16x8 p0 = *reinterpret_cast<const bf16x8 *>(hb + poff[ks]);
            const bf16x8 p1 = *reinterpret_cast<const bf16x8 *>(hb + HS_X * 16 + poff[ks]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, p0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, p1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, p0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, p1, acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        C8_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        // epilogue: acc[j][i][r] = channel 64 tn + 32 j + 8 (r>>2) + 4 half + (r&3), pixel (oy0 + 2 wave + i, ox0 + tx).
        // A lane owns ONE pixel (two lanes per pixel), so straight from the registers a store instruction touches 32
        // different 128-byte rows with 32 bytes each -- the layer is store-issue bound that way (2.8 TB/s of the 1.07 GB it
        // writes at batch 128).  Each (row i, channel half j) piece -- 32 pixels x 64 bytes -- goes through a wave-private
        // 2 KB LDS stage instead and leaves as two store instructions of 16 pixels x 64 contiguous bytes.
        unsigned char *const stg = ldsS + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t pix0 = ((size_t)n * a.H + (oy0 + 2 * wave + i)) * a.W + ox0;   // pixel of column 0 of this tile row
            unsigned wbits = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cbase = tn * 64 + 32 * j;
                uint2 pk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4] = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                    if (a.bits) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) wbits |= (v[e] > 0.0f ? 1u : 0u) << (16 * j + 4 * g + e);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.0f ? v[e] : v[e] * a.slope;
                    pk[g].x = pack_bf16(v[0], v[1]);
                    pk[g].y = pack_bf16(v[2], v[3]);
                }
                // the previous piece's reads of the stage have returned (their data was stored from registers)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    auto sx = __builtin_amdgcn_permlane32_swap(pk[g].x, pk[g + 1].x, false, false);
                    auto sy = __builtin_amdgcn_permlane32_swap(pk[g].y, pk[g + 1].y, false, false);
                    uint4 o;
                    o.x = sx[0]; o.y = sy[0]; o.z = sx[1]; o.w = sy[1];
                    // this lane: channels 32 j + 8 (g + half) ..+7 of pixel tx = 16-byte chunk lc = g + half of the pixel's
                    // 64-byte piece; slot lc ^ ((tx >> 1) & 3): eight consecutive lanes hit eight different 16-byte banks
                    *reinterpret_cast<uint4 *>(stg + tx * 64 + (((g + half) ^ ((tx >> 1) & 3)) << 4)) = o;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int u = lane + 64 * k, p = u >> 2, lc = (u & 3) ^ ((p >> 1) & 3);   // unit u of the stage -> (pixel, chunk)
                    uint4 o = *reinterpret_cast<const uint4 *>(stg + u * 16);
                    const size_t off = (pix0 + p) * a.Cout + cbase + 8 * lc;
                    if (a.mask) {
                        const uint4 m = mk[i][j][k];
                        const unsigned mw[4] = {m.x, m.y, m.z, m.w};
                        unsigned ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float lo = __uint_as_float(ow[q] << 16), hi = __uint_as_float(ow[q] & 0xffff0000u);
                            if (!(__uint_as_float(mw[q] << 16) > 0.0f)) lo *= a.mask_slope;
                            if (!(__uint_as_float(mw[q] & 0xffff0000u) > 0.0f)) hi *= a.mask_slope;
                            ow[q] = pack_bf16(lo, hi);
                        }
                        o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                    }
                    *reinterpret_cast<uint4 *>(a.y + off) = o;
                }
            }
            if (a.bits) a.bits[((pix0 + tx) * (size_t)(a.Cout >> 6) + tn) * 2 + half] = wbits;
        }
        __builtin_amdgcn_sched_barrier(0);
        C8_STAMP(3);
#ifdef M355_DBG_STAMP_C8
        ++dbg_i;
#endif
        if (tp_next >= tiles) break;
        tp = tp_next;
        cur = nxt;
        nxt = advance(nxt);
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// rows of per-workgroup partial sums [rows][n + nb] -> dw[n] (+ db[nb]), added in row order (used by the halo weight gradients too)
int wgrad_part_sum_launch(const float *part, int rows, size_t stride, float *dw, size_t n, float *db, int nb, hipStream_t st)
{
    part_sum_launch(part, rows, stride, dw, n, db, nb, st);
    return check_launch("conv2d_wgrad (ordered sum of the partial rows)");
}

bool conv_c8_eligible(const m355_conv_desc *d, int y_f32_nchw)
{
    return !y_f32_nchw && d->Cin == 8 && d->kh == 5 && d->kw == 5 && d->stride == 1 && d->upsample == 0 && d->pad_h == 2 &&
           d->pad_w == 2 && d->Cout % 64 == 0 && d->W % 32 == 0 && d->H % 8 == 0 &&
           (size_t)d->N * d->H * d->W * 16 < (1ull << 31);
}

int conv_c8_launch(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, float slope, int Kp,
                   size_t wbytes, unsigned *bits, hipStream_t st, const void *mask = nullptr, float mask_slope = 1.0f)
{
    C8Args a = {};
    a.mask = (const unsigned short *)mask;
    a.mask_slope = mask_slope;
    a.x = (const unsigned short *)x;
    a.w = (const unsigned short *)w_fwd;
    a.bias = bias;
    a.y = (unsigned short *)y;
    a.bits = bits;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.Kp = Kp;
    a.slope = slope;
    a.xbytes = (unsigned)((size_t)d->N * d->H * d->W * 16);
    a.wbytes = (unsigned)wbytes;
#ifdef M355_DBG_STAMP_C8
    if (const char *sp = getenv("M355_STAMP_PTR")) a.stamp = reinterpret_cast<unsigned *>(strtoull(sp, nullptr, 0));
#endif
    const int tiles = d->N * (d->H / 8) * (d->W / 32), nN = d->Cout / 64;
    // TWO resident workgroups per CU: 50 KB of LDS would allow three, but the kernel holds 215 registers (the bias lives in 32 of
    // them) = two waves per SIMD.  With 768 workgroups a third of them ran as a second round on a third-empty chip: the stamps
    // (scripts/stamp_c8.py) show 10.2 k cycles per tile and wave, 43 tiles per workgroup = 194 us per round, two rounds = the
    // measured 384 us.  512 workgroups x 64 tiles is one round.
    static const int c8_wgs = getenv("M355_C8_WGS") ? atoi(getenv("M355_C8_WGS")) : 512;
    int per = c8_wgs / nN;
    if (per > tiles) per = tiles;
    if (per < 1) per = 1;
    const dim3 grid(per * nN);
    if (d->pad_w_mode == 0) hipLaunchKernelGGL((k_conv_c8<0>), grid, dim3(256), 0, st, a);
    else if (d->pad_w_mode == 1) hipLaunchKernelGGL((k_conv_c8<1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_c8<2>), grid, dim3(256), 0, st, a);
    note_kernel("k_conv_c8");
    return check_launch("conv2d_fwd (8 input channels)");
}

// ---- dgrad of a replicate-padded 5x5 head (conv_final / conv_mesh of the symmetric generator, gan.py:328-329,359-368):
// Cout <= 8 so dy is ONE 16-byte chunk per pixel and the gradient is "a conv with 8 input channels" = k_conv_c8 on dy with
// the flipped weight view -- for the interior term (zero-padded dy).  The replicate pad adds the gradient of the two pad
// columns on each side, which lands on image columns 0 and W-1: dx[:, y, 0] += dxp[y, -2] + dxp[y, -1] (and mirrored), where
// dxp is the same conv evaluated at the pad columns.  Only taps reaching back into the image contribute (1 + 2 of the 5 kw
// per pad column), so this is 2/W of the layer's work: one thread per (row, side, input channel).
__global__ __launch_bounds__(256) void k_dgrad_edge5(const unsigned short *__restrict__ dy, const unsigned short *__restrict__ wd,
                                                     unsigned short *__restrict__ dx, int N, int H, int W, int Cin, int Kp,
                                                     const unsigned short *__restrict__ mask, float mask_slope)
{
    const size_t total = (size_t)N * H * 2 * Cin;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        size_t t = i / Cin;
        const int side = (int)(t & 1);
        t >>= 1;
        const int y = (int)(t % H), n = (int)(t / H);
        const unsigned short *wrow = wd + (size_t)ci * Kp;
        float acc = 0.0f;
        for (int pp = 0; pp < 2; ++pp) {
            const int p = side == 0 ? -2 + pp : W + pp;          // pad column (frame coordinate of the dgrad output)
            for (int kh = 0; kh < 5; ++kh) {
                const int yy = y + kh - 2;
                if ((unsigned)yy >= (unsigned)H) continue;
                for (int kw = 0; kw < 5; ++kw) {
                    const int xx = p + kw - 2;
                    if ((unsigned)xx >= (unsigned)W) continue;
                    const bf16x8 dv = *reinterpret_cast<const bf16x8 *>(dy + (((size_t)n * H + yy) * W + xx) * 8);
                    const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wrow + (kh * 5 + kw) * 8);
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc += bf2f((unsigned short)dv[c]) * bf2f((unsigned short)wv[c]);
                }
            }
        }
        const size_t off = (((size_t)n * H + y) * W + (side == 0 ? 0 : W - 1)) * Cin + ci;
        if (mask && !(bf2f(mask[off]) > 0.0f)) acc *= mask_slope;   // (the interior term was masked by k_conv_c8's epilogue)
        dx[off] = f2bf(bf2f(dx[off]) + acc);
    }
}

bool dgrad_c8_replicate_eligible(const m355_conv_desc *d, int Cy)
{
    return Cy == 8 && d->pad_w_mode == 1 && d->stride == 1 && d->upsample == 0 && d->kh == 5 && d->kw == 5 && d->pad_h == 2 &&
           d->pad_w == 2 && d->Cin % 64 == 0 && d->W % 32 == 0 && d->H % 8 == 0 && d->W >= 4 &&
           (size_t)d->N * d->H * d->W * 16 < (1ull << 31) && !getenv("M355_NO_C8_DGRAD");
}

int dgrad_c8_replicate_launch(const m355_conv_desc *d, const void *dy, const void *w_dgrad, int Kp, size_t wbytes, void *dx,
                              hipStream_t st, const void *mask_x, float mask_slope)
{
    m355_conv_desc t = *d;
    t.Cin = 8;             // the "input" of this conv is dy (3 real channels in an 8-channel chunk)
    t.Cout = d->Cin;
    t.pad_w_mode = 0;      // interior term: zero-padded dy
    if (int rc = conv_c8_launch(&t, dy, w_dgrad, nullptr, dx, 1.0f, Kp, wbytes, nullptr, st, mask_x, mask_slope)) return rc;
    const size_t total = (size_t)d->N * d->H * 2 * d->Cin;
    const size_t g = (total + 255) / 256;
    hipLaunchKernelGGL(k_dgrad_edge5, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, st, (const unsigned short *)dy,
                       (const unsigned short *)w_dgrad, (unsigned short *)dx, d->N, d->H, d->W, d->Cin, Kp,
                       (const unsigned short *)mask_x, mask_slope);
    note_kernel("k_conv_c8");
    return check_launch("conv2d_dgrad (replicate 5x5 head)");
}

// host side: eligibility + launch (called from m355_conv2d_fwd)
bool conv_small_eligible(const m355_conv_desc *d, int y_f32_nchw)
{
    return y_f32_nchw && d->Cout <= 4 && d->stride == 1 && d->upsample == 0 && d->kh == d->kw &&
           (d->kh == 5 || d->kh == 3) && d->pad_h == d->kh / 2 && d->pad_w == d->kw / 2 && d->Cin % 64 == 0 &&
           (size_t)d->N * d->H * d->W * d->Cin * 2 < (1ull << 31);
}

// dgrad of a stride-1, "same"-padded conv with <= 8 INPUT channels (TextureDiscriminator.conv1): a conv of dy (Cy % 64 == 0
// channels) with the flipped, transposed weights onto 8 channels -- the same halo kernel with bf16 NHWC output
bool dgrad_small_eligible(const m355_conv_desc *d, int Cy)
{
    return d->Cin == 8 && d->stride == 1 && d->upsample == 0 && d->pad_w_mode != 1 && d->kh == d->kw &&
           (d->kh == 5 || d->kh == 3) && d->pad_h == d->kh / 2 && d->pad_w == d->kw / 2 && Cy % 64 == 0 &&
           (size_t)d->N * d->H * d->W * Cy * 2 < (1ull << 31);
}

int dgrad_small_launch(const m355_conv_desc *d, const void *dy, int Cy, const void *w_dgrad, int Kp, size_t wbytes, void *dx,
                       hipStream_t st, int lead)
{
    if (lead >= 1 && lead <= 4 && Cy == 64 && d->kh == 5 && !getenv("M355_NO_HEAD5")) {
        // only the first `lead` input channels' gradient is wanted: the scatter form (k_head5<MODE, 1>), rows 0 .. lead-1 of the
        // dgrad view (= a forward view of a 64 -> Cin conv with the taps already flipped)
        Head5Args h = {};
        h.x = (const unsigned short *)dy;
        h.w = (const unsigned short *)w_dgrad;
        h.yb = (unsigned short *)dx;
        h.N = d->N; h.H = d->H; h.W = d->W; h.Cout = lead; h.Kp = Kp;
        h.nsx = (d->W + 27) / 28;
        long nsy = 2048 / ((long)d->N * h.nsx);
        if (nsy < 1) nsy = 1;
        if (nsy > (d->H + 7) / 8) nsy = (d->H + 7) / 8;
        h.rs = (int)((d->H + nsy - 1) / nsy);
        h.nsy = (d->H + h.rs - 1) / h.rs;
        h.slope = 1.0f;
        h.xbytes = (unsigned)((size_t)d->N * d->H * d->W * Cy * 2);
        h.wbytes = (unsigned)wbytes;
        const dim3 grid((unsigned)((long)d->N * h.nsx * h.nsy));
        if (d->pad_w_mode == 0) hipLaunchKernelGGL((k_head5<0, 1>), grid, dim3(64), 0, st, h);
        else hipLaunchKernelGGL((k_head5<2, 1>), grid, dim3(64), 0, st, h);
        note_kernel("k_head5");
        return check_launch("conv2d_dgrad (8 input channels, leading channels only)");
    }
    m355_conv_desc t = *d;
    t.Cin = Cy;   // the "input" of this conv is dy
    t.Cout = 8;
    SmallArgs a = {};
    a.x = (const unsigned short *)dy;
    a.w = (const unsigned short *)w_dgrad;
    a.yb = (unsigned short *)dx;
    a.Cs = 8;
    a.N = t.N; a.H = t.H; a.W = t.W; a.Cin = Cy; a.Cout = 8; a.Kp = Kp;
    a.tiles_x = (t.W + ST - 1) / ST;
    a.slope = 1.0f;
    a.xbytes = (unsigned)((size_t)t.N * t.H * t.W * Cy * 2);
    a.wbytes = (unsigned)wbytes;
    const dim3 grid(a.tiles_x * ((t.H + ST - 1) / ST), t.N);
    if (t.kh == 5) {
        if (t.pad_w_mode == 0) hipLaunchKernelGGL((k_conv_smallco<5, 0>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_conv_smallco<5, 2>), grid, dim3(256), 0, st, a);
    } else {
        if (t.pad_w_mode == 0) hipLaunchKernelGGL((k_conv_smallco<3, 0>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_conv_smallco<3, 2>), grid, dim3(256), 0, st, a);
    }
    note_kernel("k_conv_smallco");
    return check_launch("conv2d_dgrad (8 input channels)");
}

int conv_small_launch(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, float slope,
                      int Kp, size_t wbytes, hipStream_t st)
{
    if (d->Cin == 64 && d->kh == 5 && !getenv("M355_NO_HEAD5")) {   // conv_final / conv_mesh: the scatter form (k_head5)
        Head5Args h = {};
        h.x = (const unsigned short *)x;
        h.w = (const unsigned short *)w_fwd;
        h.bias = bias;
        h.y = (float *)y;
        h.N = d->N; h.H = d->H; h.W = d->W; h.Cout = d->Cout; h.Kp = Kp;
        h.nsx = (d->W + 27) / 28;
        // row segments: (rs + 4) / rs of the input is read, so as few as fill the chip ONCE (2 waves per SIMD = 2048 waves:
        // a second, partial round of waves would double the kernel's time), but at least 8 rows each
        long nsy = 2048 / ((long)d->N * h.nsx);
        if (nsy < 1) nsy = 1;
        if (nsy > (d->H + 7) / 8) nsy = (d->H + 7) / 8;
        h.rs = (int)((d->H + nsy - 1) / nsy);
        h.nsy = (d->H + h.rs - 1) / h.rs;
        h.slope = slope;
        h.xbytes = (unsigned)((size_t)d->N * d->H * d->W * d->Cin * 2);
        h.wbytes = (unsigned)wbytes;
        const dim3 grid((unsigned)((long)d->N * h.nsx * h.nsy));
        if (d->pad_w_mode == 0) hipLaunchKernelGGL((k_head5<0>), grid, dim3(64), 0, st, h);
        else if (d->pad_w_mode == 1) hipLaunchKernelGGL((k_head5<1>), grid, dim3(64), 0, st, h);
        else hipLaunchKernelGGL((k_head5<2>), grid, dim3(64), 0, st, h);
        note_kernel("k_head5");
        return check_launch("conv2d_fwd (5x5 head)");
    }
    SmallArgs a = {};
    a.x = (const unsigned short *)x;
    a.w = (const unsigned short *)w_fwd;
    a.bias = bias;
    a.y = (float *)y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Kp = Kp;
    a.tiles_x = (d->W + ST - 1) / ST;
    a.slope = slope;
    a.xbytes = (unsigned)((size_t)d->N * d->H * d->W * d->Cin * 2);
    a.wbytes = (unsigned)wbytes;
    const dim3 grid(a.tiles_x * ((d->H + ST - 1) / ST), d->N);
#define M355_SM(KS_)                                                                                              \
    do {                                                                                                          \
        if (d->pad_w_mode == 0) hipLaunchKernelGGL((k_conv_smallco<KS_, 0>), grid, dim3(256), 0, st, a);          \
        else if (d->pad_w_mode == 1) hipLaunchKernelGGL((k_conv_smallco<KS_, 1>), grid, dim3(256), 0, st, a);     \
        else hipLaunchKernelGGL((k_conv_smallco<KS_, 2>), grid, dim3(256), 0, st, a);                             \
    } while (0)
    if (d->kh == 5) M355_SM(5);
    else M355_SM(3);
#undef M355_SM
    note_kernel("k_conv_smallco");
    return check_launch("conv2d_fwd (small Cout)");
}

// ---- weight (and bias) gradient of the 8-input-channel 5x5 layer (D.conv1): dw[co][kh][kw][ci] = sum_p dy[p][co] x[p + (kh,kw)][ci].
// As a GEMM: M = 64 output channels, N = 200 (tap, ci) columns, K = all pixels -- 215 GFLOP per launch at N = 128 against
// 1.07 GB of dy: HBM-side.  k_wgrad_dma gathers every pixel of x 25 times through L2 into its 256-column tile (19 KB of
// LDS-DMA per MFLOP; 0.55 ms).  Here a workgroup walks 8 x 32-pixel tiles: the dy tile (32 KB) and the 12 x 36 x 16-byte
// halo of x (6.9 KB) are DMA'd once, three stages deep; both MFMA operands come from transpose reads with the pixel axis
// as K.  An x "row" for ds_read_b64_tr_b16 is 32 contiguous bytes = the 8 channels of halo pixels (px + kw), (px + kw + 1):
// a 32-column MFMA block = 4 consecutive kw taps x 8 ci of one kh; 2 blocks per kh (kw 0..3 | 4..7, taps 5..7 are
// discarded columns), 10 blocks in all.  8 waves = 2 pixel halves x 2 co blocks x 2 groups of 5 column blocks; the 5
// accumulators of a wave stay in registers across all tiles; fp32 atomics at the end (split K over workgroups).
struct C8WgArgs {
    const unsigned short *x;   // bf16 NHWC [N,H,W,8]
    const unsigned short *dy;  // bf16 NHWC [N,H,W,Cy], Cy = 64 * k
    float *dw, *db;            // fp32 [Cout][5][5][8] (+=), [Cout] (+=) or null
    long long *fix;            // deterministic form: [flag | fixed-point dw | fixed-point db] (conv_dma.h wg_accum), else null
    float *part;               // k_wgrad_c8p: per-workgroup partial sums [gridDim.x][Cout*200 + Cout] (k_wgrad_part_sum), else null
    int N, H, W, Cout, Cy;
    unsigned xbytes, ybytes;
};

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// ---- k_wgrad_c8p (round 4): that weight gradient with the x halo held CHANNEL-MAJOR ("planar") in LDS.
// Its round-2 predecessor (k_wgrad_c8, the design of the paragraph above; removed from the library in round 4, git history has it)
// read every x pixel 25 times from LDS through transpose reads (96 ds_read_b64_tr per 40 MFMAs and wave: the LDS
// pipe is as busy as the matrix pipe, and only 200 of the 320 MFMA columns it computes are taps) and sat at 3.0-3.2 TB/s of its
// 1.2 GB stream at 1.4 kW.  Here:
//   * the 12 x 36 x 8-channel halo of a tile (6.9 KB) is fetched by plain 16-byte global loads one tile ahead and written into
//     LDS as eight planes [ci][12 rows][40 columns] of bf16 (row pitch 80 B, plane pitch 1056 B = 32 mod 256: the 16-byte window
//     reads of a 16-lane group are conflict free);
//   * MFMA 16x16x32 (K = one tile row of 32 pixels): a B column is (kh = 2 kp + a, ci) for a lane's a = (lane >> 3) & 1 and
//     ci = lane & 7, and its eight k values are eight CONSECUTIVE pixels of one channel = one 16-byte run of a plane.  A lane
//     reads a 12-pixel window of halo row (ty + kh) ONCE (ds_read_b128 + ds_read_b64) and derives the five kw taps from it in
//     registers (kw even: a register offset; kw odd: v_alignbit_b32) -- 6 LDS reads + 24 VALU per 30 MFMAs instead of 60 reads;
//   * column blocks: 3 kh pairs x 5 kw = 15 blocks of 16 = 240 columns for 200 taps (83 %; was 62 %); the A operand (dy^T) keeps the
//     transpose reads of the DMA'd pixel-major dy tile; the bias gradient is one more MFMA per co block against an all-ones B;
//   * waves = 2 co halves x 4 tile-row pairs; a wave keeps 2 x 15 (+ 2) 16 x 16 accumulators for the whole kernel; at the end the
//     four row-pair waves of a co half are summed through LDS and leave as ONE atomic per element and workgroup;
//   * the dy tiles (32 KB) are DMA'd four stages deep (three tiles = 96 KB in flight per CU).
constexpr int C8P_SC = 1056, C8P_RP = 80, C8P_XB = 8 * C8P_SC;   // plane / row pitch, bytes per planar halo buffer
constexpr int C8P_NST = 4, C8P_YB = 256 * 128;

template <int MODE, bool DET = false>
__global__ __launch_bounds__(512, 2) void k_wgrad_c8p(C8WgArgs a)
{
    constexpr int TH = 8, TW = 32, KS = 5, HS_X = TW + KS - 1, HS_Y = TH + KS - 1;   // 12 x 36 halo
    constexpr int NST = C8P_NST;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * C8P_YB + 2 * C8P_XB];
    unsigned char *const ldsX = lds + NST * C8P_YB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.y * 64;
    const int tpx = a.W / TW, tpy = a.H / TH, tiles = a.N * tpx * tpy, G = gridDim.x;
    if ((int)blockIdx.x >= tiles) return;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, a.ybytes, 0x00020000);

    // ---- tile cursors (image, tile row, tile column), stepped by the grid stride; the LAST tile is re-fetched when the list is
    // exhausted so that every iteration issues the same number of memory instructions (constant s_waitcnt counts)
    struct Cur { int n, ty, tx, left; };
    int sg_n, sg_ty, sg_tx;
    auto cur_at = [&](int tile) {
        Cur c;
        const int per_img = tpx * tpy;
        c.n = tile / per_img;
        const int r = tile - c.n * per_img;
        c.ty = r / tpx; c.tx = r - c.ty * tpx;
        c.left = (tiles - 1 - tile) / G;     // further tiles of this workgroup after `tile`
        return c;
    };
    {
        const int per_img = tpx * tpy;
        sg_n = G / per_img;
        const int r = G - sg_n * per_img;
        sg_ty = r / tpx; sg_tx = r - sg_ty * tpx;
    }
    auto advance = [&](Cur &c) {
        if (c.left <= 0) return;             // stay on the last tile
        --c.left;
        c.tx += sg_tx;
        const int cx = c.tx >= tpx ? 1 : 0;
        c.tx -= cx * tpx;
        c.ty += sg_ty + cx;
        const int cy = c.ty >= tpy ? 1 : 0;
        c.ty -= cy * tpy;
        c.n += sg_n + cy;
    };
    Cur cdy = cur_at((int)blockIdx.x), cx = cdy;   // next tile whose dy DMA / whose x loads are to be issued

    // ---- dy DMA: instruction qi = 8 k + wave fills tile pixels 8 qi .. 8 qi + 7 (pixel = LDS row of 128 B = 64 co); slot l & 7 of
    // row r holds source chunk (l & 7) ^ 4 ((r >> 1) & 1) ^ 2 ((r >> 3) & 1): the transpose reads of four pixel groups 8 apart
    // then hit distinct banks
    const int csrc = (lane & 7) ^ (((lane >> 4) & 1) << 2) ^ ((wave & 1) << 1);
    auto issue_dy = [&](int st) {
        const int n = cdy.n, oy0 = cdy.ty * TH, ox0 = cdy.tx * TW;
        advance(cdy);
        unsigned char *dY = lds + st * C8P_YB;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = 8 * (8 * k + wave) + (lane >> 3);
            const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31);
            dma16(ry, dY + (8 * k + wave) * 1024, (unsigned)((((n * a.H + oy) * a.W + ox) * a.Cy + co0) * 2 + csrc * 16), 0u);
        }
    };
    // ---- x halo: thread t < 216 owns halo pixels (hy, 2 hp), (hy, 2 hp + 1), t = 18 hy + hp: two 16-byte loads, eight 4-byte
    // plane writes
    const bool has_x = wave < 4;                    // (wave-uniform: threads 216..255 of wave 3 idle inside)
    const int xhy = tid / 18, xhp = tid - 18 * xhy;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 xr0 = {0, 0, 0, 0}, xr1 = {0, 0, 0, 0};
    auto load_x = [&]() {
        const int n = cx.n, oy0 = cx.ty * TH, ox0 = cx.tx * TW;
        advance(cx);
        const int gy = oy0 - KS / 2 + xhy;
        unsigned off[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int gx = ox0 - KS / 2 + 2 * xhp + e;
            bool ok = tid < 216 && (unsigned)gy < (unsigned)a.H;
            if (MODE == 1) gx = min(max(gx, 0), a.W - 1);
            else if (MODE == 2) gx = gx < 0 ? gx + a.W : (gx >= a.W ? gx - a.W : gx);
            ok = ok && (unsigned)gx < (unsigned)a.W;
            off[e] = ok ? (unsigned)(((n * a.H + gy) * a.W + gx) * 16) : OOB;
        }
        xr0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off[0], 0, 0));
        xr1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off[1], 0, 0));
    };
    auto store_x = [&](int buf) {
        if (tid < 216) {
            unsigned char *dst = ldsX + buf * C8P_XB + xhy * C8P_RP + xhp * 4;
#pragma unroll
            for (int w = 0; w < 4; ++w) {   // word w of a pixel = channels 2w (low half), 2w + 1
                *reinterpret_cast<unsigned *>(dst + (2 * w) * C8P_SC) = __builtin_amdgcn_perm(xr1[w], xr0[w], 0x05040100u);
                *reinterpret_cast<unsigned *>(dst + (2 * w + 1) * C8P_SC) = __builtin_amdgcn_perm(xr1[w], xr0[w], 0x07060302u);
            }
        }
    };

    // ---- fragment roles
    const int ch = wave & 1, pq = wave >> 1;                 // co half (32 channels), tile rows 2 pq, 2 pq + 1
    const int q = lane & 15, g = lane >> 4;
    const int swz = (((q >> 3) & 1) << 2) ^ ((g & 1) << 1);
    int ya[2][2];                                            // [co block i][pixels 4 rd ..]: byte offset inside a dy stage, tile row 0
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd)
            ya[i][rd] = (8 * g + 4 * rd + (q >> 2)) * 128 + (((2 * (2 * ch + i) + ((q & 3) >> 1)) ^ swz) << 4) + (q & 1) * 8;
    const int bci = lane & 7, ba = (lane >> 3) & 1;
    int xw[3];                                               // window base per kh pair, tile row 0: plane ci, halo row kh, column 8 g
#pragma unroll
    for (int kp = 0; kp < 3; ++kp) xw[kp] = bci * C8P_SC + min(2 * kp + ba, 4) * C8P_RP + 16 * g;

    f32x4 acc[2][15], accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 15; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_db = a.db != nullptr;
    const bf16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};

    // ---- prologue: x of tile 0 into planar buffer 0, dy of tiles 0 .. NST-2 in flight
    if (has_x) load_x();
#pragma unroll
    for (int k = 0; k < NST - 1; ++k) issue_dy(k);
    if (has_x) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NST - 1)) : "memory");
        store_x(0);
    }
    int st = 0;
    const int my_tiles = (tiles - 1 - (int)blockIdx.x) / G + 1;
    for (int it = 0; it < my_tiles; ++it) {
        // dy of this tile was issued NST-1 tiles ago: the DMAs of the NST-2 younger tiles may stay in flight
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (NST - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (has_x) load_x();                                 // x of the NEXT tile (older than the DMAs below: vmcnt(4) awaits it alone)
        issue_dy(st == 0 ? NST - 1 : st - 1);                // into the stage consumed one tile ago
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char *by = lds + st * C8P_YB, *bx = ldsX + (it & 1) * C8P_XB;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            const int ty = 2 * pq + kg;
            bf16x8 yf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const s4w y0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4w *)(by + ty * 4096 + ya[i][0]));
                const s4w y1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4w *)(by + ty * 4096 + ya[i][1]));
                yf[i] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            if (do_db) {
#pragma unroll
                for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], ones, accb[i], 0, 0, 0);
            }
#pragma unroll
            for (int kp = 0; kp < 3; ++kp) {
                const unsigned char *wp = bx + ty * C8P_RP + xw[kp];
                const u32x4 wa = *reinterpret_cast<const u32x4 *>(wp);                                     // pixels 0..7 of the window
                const uint2 wb = *reinterpret_cast<const uint2 *>(wp + 16);                                // pixels 8..11
                const unsigned w6[6] = {wa[0], wa[1], wa[2], wa[3], wb.x, wb.y};
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) {
                    u32x4 f;
                    if (kw & 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) f[j] = __builtin_amdgcn_alignbit(w6[kw / 2 + j + 1], w6[kw / 2 + j], 16);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) f[j] = w6[kw / 2 + j];
                    }
                    const bf16x8 xf = __builtin_bit_cast(bf16x8, f);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][kp * 5 + kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf, acc[i][kp * 5 + kw], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has_x) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // the next tile's x (its 4 younger dy DMAs stay in flight)
            store_x((it + 1) & 1);                             // (that buffer was last read one tile ago, behind this tile's barrier)
        }
        st = st == NST - 1 ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- sum the four row-pair waves of each co half through LDS (the stages are free now), one atomic per element
    float *red = reinterpret_cast<float *>(lds);              // [co half][slot 0..1][128 values][64 lanes] = 128 KB of the free stages
    float *const mine = red + (size_t)ch * 2 * 128 * 64 + lane;
    auto put = [&](int s) {
        float *o = mine + s * 128 * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int t = 0; t < 15; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[((i * 16 + t) * 4 + r) * 64] = acc[i][t][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[((i * 16 + 15) * 4 + r) * 64] = accb[i][r];
        }
    };
    auto add = [&](int s) {
        const float *o = mine + s * 128 * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int t = 0; t < 15; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][t][r] += o[((i * 16 + t) * 4 + r) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) accb[i][r] += o[((i * 16 + 15) * 4 + r) * 64];
        }
    };
    if (pq >= 2) put(pq - 2);
    __syncthreads();
    if (pq < 2) add(pq);
    __syncthreads();
    if (pq == 1) put(0);
    __syncthreads();
    if (pq != 0) return;
    add(0);
    // acc[i][kp*5+kw][r]: co = co0 + 16 (2 ch + i) + 4 g + r; column lane & 15 = (a, ci): kh = 2 kp + a
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + 16 * (2 * ch + i) + 4 * g + r;
            if (co >= a.Cout) continue;
            float *const prow = a.part ? a.part + (size_t)blockIdx.x * ((size_t)a.Cout * 201) : nullptr;
            if (do_db && (lane & 15) == 0) {
                if (prow) prow[(size_t)a.Cout * 200 + co] = accb[i][r];
                else wg_accum<DET>(a.db, a.fix, (DET ? (size_t)a.Cout * 200 : 0) + co, accb[i][r], (size_t)a.Cout * 201);
            }
#pragma unroll
            for (int kp = 0; kp < 3; ++kp) {
                const int kh = 2 * kp + ba;
                if (kh >= KS) continue;
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) {
                    const size_t e = ((size_t)co * 25 + kh * 5 + kw) * 8 + bci;
                    if (prow) prow[e] = acc[i][kp * 5 + kw][r];
                    else wg_accum<DET>(a.dw, a.fix, e, acc[i][kp * 5 + kw][r], (size_t)a.Cout * 201);
                }
            }
        }
    }
}

bool wgrad_c8_eligible(const m355_conv_desc *d, int Cy)
{
    return d->Cin == 8 && d->kh == 5 && d->kw == 5 && d->stride == 1 && d->upsample == 0 && d->pad_h == 2 && d->pad_w == 2 &&
           Cy % 64 == 0 && d->W % 32 == 0 && d->H % 8 == 0 && (size_t)d->N * d->H * d->W * Cy * 2 < (1ull << 31) &&
           !getenv("M355_NO_C8");
}

static int wgrad_c8_gx(const m355_conv_desc *d, int Cy)
{
    const int tiles = d->N * (d->H / 8) * (d->W / 32), ny = Cy / 64;
    int gx = 256 / ny;  // one 8-wave workgroup per CU (120 / 145 KB of LDS)
    if (gx < 1) gx = 1;
    return gx > tiles ? tiles : gx;
}
size_t wgrad_c8_ws_floats(const m355_conv_desc *d, int Cy)
{
    return (size_t)wgrad_c8_gx(d, Cy) * ((size_t)d->Cout * 201);
}

int wgrad_c8_launch(const m355_conv_desc *d, const void *x, const void *dy, int Cy, float *dw, float *db, hipStream_t st,
                    long long *fix, float *part)
{
    C8WgArgs a = {};
    a.x = (const unsigned short *)x;
    a.dy = (const unsigned short *)dy;
    a.dw = dw; a.db = db;
    a.fix = fix;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.Cy = Cy;
    a.xbytes = (unsigned)((size_t)d->N * d->H * d->W * 16);
    a.ybytes = (unsigned)((size_t)d->N * d->H * d->W * Cy * 2);
    const int ny = Cy / 64, gx = wgrad_c8_gx(d, Cy);
    const dim3 grid(gx, ny);
    a.part = part;
#define M355_C8P(MD_)                                                                              \
    do {                                                                                          \
        if (fix) hipLaunchKernelGGL((k_wgrad_c8p<MD_, true>), grid, dim3(512), 0, st, a);          \
        else hipLaunchKernelGGL((k_wgrad_c8p<MD_>), grid, dim3(512), 0, st, a);                    \
    } while (0)
    if (d->pad_w_mode == 0) M355_C8P(0);
    else if (d->pad_w_mode == 1) M355_C8P(1);
    else M355_C8P(2);
#undef M355_C8P
    if (part) {
        // every (pixel-axis workgroup) row holds the partial of ALL output channels (blockIdx.y writes its own 64): sum the rows
        const size_t n = (size_t)d->Cout * 200;
        part_sum_launch((const float *)part, gx, (size_t)d->Cout * 201, dw, n, db, d->Cout, st);
    }
    note_kernel("k_wgrad_c8");
    return check_launch("conv2d_wgrad (8 input channels, planar)");
}

}  // namespace m355
