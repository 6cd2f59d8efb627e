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
#include "conv_dma.h"

namespace m355 {

constexpr int ST = 16;       // output tile side
constexpr int SMAXCO = 4;    // output channels served (weights chunk <= 4 * 25 * 128 B = 12.5 KiB of LDS)

struct SmallArgs {
    const unsigned short *x;  // bf16 NHWC [N,H,W,Cin], Cin % 64 == 0
    const unsigned short *w;  // bf16 forward view [rows][Kp], row = output channel, K ordered (kh, kw, ci)
    const float *bias;        // [Cout] or null
    float *y;                 // fp32 NCHW [N,Cout,H,W]
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

// host side: eligibility + launch (called from m355_conv2d_fwd)
bool conv_small_eligible(const m355_conv_desc *d, int y_f32_nchw)
{
    return y_f32_nchw && d->Cout <= SMAXCO && d->stride == 1 && d->upsample == 0 && d->kh == d->kw &&
           (d->kh == 5 || d->kh == 3) && d->pad_h == d->kh / 2 && d->pad_w == d->kw / 2 && d->Cin % 64 == 0 &&
           (size_t)d->N * d->H * d->W * d->Cin * 2 < (1ull << 31);
}

int conv_small_launch(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, float slope,
                      int Kp, size_t wbytes, hipStream_t st)
{
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
    return check_launch("conv2d_fwd (small Cout)");
}

}  // namespace m355
