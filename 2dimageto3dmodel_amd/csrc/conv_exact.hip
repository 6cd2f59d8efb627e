// EXACT build (-DM355_EXACT, lib/libm355_exact.so; SURVEY.md 8c "an fp32-accumulate exact mode for 1e-4 checks").
//
// The same C-ABI, the same Python orchestration (gan.py / gan_ops.py / train.py), but every ACTIVATION tensor is fp32 instead of
// bf16 and the convolutions are these plain fp32 kernels with fp64 accumulation -- F.conv2d of models/gan.py:57-65,163-177,
// 294-302,359,364 with the pads (F.pad replicate gan.py:329, circpad rendering/utils.py:29-33) and the nearest x2 upsample
// (gan.py:319,386-404) evaluated literally, one thread per output element, no tiling, no MFMA.  Speed is irrelevant here: the
// build exists so that what sits ABOVE a conv layer (conditional batch-norm statistics and eps, spectral-norm iteration order,
// hinge masking, Adam(0, 0.9), the running-average ramp) can be held to the reference's fp32 CPU results at 1e-4 instead of
// through bf16 noise (tests/test_exact_mode_gpu.py).  The bf16 product kernels are compiled into this build too but are never
// dispatched to: the conv entry points of conv_mfma.hip return through exact::* first.
//
// Layouts: x [N,H,W,Cin] fp32, dy [N,Ho,Wo,Cy] fp32 (Cy = m355_conv2d_dy_channels(Cout)), weights fp32 [Cout][kh][kw][Cin]
// (the same array serves forward and dgrad; channels >= cin_w are zero), dw fp32 [Cout][kh][kw][Cin] -- the layout
// m355_sn_wgrad_finish reads in every build.  Deterministic: every output element is one thread's sequential fp64 sum.
#include "conv_dma.h"

namespace m355 {
namespace exact {

struct Geo {
    int N, H, W, Cin, Cout, kh, kw, stride, pad_h, pad_w, mode, ups, Ho, Wo, Hl, Wl;
};

// stored pixel read by logical padded-frame position (hp, xp) (row / column of the padded, upsampled image), or -1 (zero)
__device__ __forceinline__ int src_row(const Geo &g, int hp)
{
    const int hl = hp - g.pad_h;
    return (hl < 0 || hl >= g.Hl) ? -1 : (hl >> g.ups);
}
__device__ __forceinline__ int src_col(const Geo &g, int xp)
{
    int wl = xp - g.pad_w;
    if (g.mode == 1) wl = min(max(wl, 0), g.Wl - 1);
    else if (g.mode == 2) wl = wl < 0 ? wl + g.Wl : (wl >= g.Wl ? wl - g.Wl : wl);
    return (wl < 0 || wl >= g.Wl) ? -1 : (wl >> g.ups);
}

__global__ __launch_bounds__(256) void k_fwd(Geo g, const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                             float *__restrict__ y, int nchw, float slope)
{
    const size_t total = (size_t)g.N * g.Ho * g.Wo * g.Cout;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i % g.Cout);
        size_t t = i / g.Cout;
        const int wo = (int)(t % g.Wo);
        t /= g.Wo;
        const int ho = (int)(t % g.Ho), n = (int)(t / g.Ho);
        double acc = bias ? (double)bias[co] : 0.0;
        for (int a = 0; a < g.kh; ++a) {
            const int sh = src_row(g, ho * g.stride + a);
            if (sh < 0) continue;
            for (int b = 0; b < g.kw; ++b) {
                const int sw = src_col(g, wo * g.stride + b);
                if (sw < 0) continue;
                const float *xp = x + (((size_t)n * g.H + sh) * g.W + sw) * g.Cin;
                const float *wp = w + (((size_t)co * g.kh + a) * g.kw + b) * g.Cin;
                double s = 0.0;
                for (int c = 0; c < g.Cin; ++c) s += (double)xp[c] * (double)wp[c];
                acc += s;
            }
        }
        float v = (float)acc;
        v = v >= 0.0f ? v : v * slope;
        if (nchw) y[(((size_t)n * g.Cout + co) * g.Ho + ho) * g.Wo + wo] = v;
        else y[i] = v;
    }
}

// gather form of the input gradient: stored pixel (h, w) collects every padded-frame position that reads it
__global__ __launch_bounds__(256) void k_dgrad(Geo g, const float *__restrict__ dy, int Cy, const float *__restrict__ w, float *__restrict__ dx,
                                               const float *__restrict__ mask_x, float mask_slope)
{
    const size_t total = (size_t)g.N * g.H * g.W * g.Cin;
    const int pw = g.pad_w;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % g.Cin);
        size_t t = i / g.Cin;
        const int ws = (int)(t % g.W);
        t /= g.W;
        const int hs = (int)(t % g.H), n = (int)(t / g.H);
        double acc = 0.0;
        for (int uy = 0; uy <= g.ups; ++uy) {
            const int hp = ((hs << g.ups) + uy) + g.pad_h;
            for (int ux = 0; ux <= g.ups; ++ux) {
                const int wl = (ws << g.ups) + ux;
                // padded-frame columns whose source is logical column wl: itself, and the pad columns mapped onto it
                for (int xp = 0; xp < g.Wl + 2 * pw; ++xp) {
                    bool hit = xp == wl + pw;
                    if (!hit && g.mode == 1) hit = (wl == 0 && xp < pw) || (wl == g.Wl - 1 && xp >= g.Wl + pw);
                    if (!hit && g.mode == 2) hit = (xp < pw && xp + g.Wl - pw == wl) || (xp >= g.Wl + pw && xp - g.Wl - pw == wl);
                    if (!hit) continue;
                    for (int a = 0; a < g.kh; ++a) {
                        const int th = hp - a;
                        if (th < 0 || th % g.stride) continue;
                        const int ho = th / g.stride;
                        if (ho >= g.Ho) continue;
                        for (int b = 0; b < g.kw; ++b) {
                            const int tw = xp - b;
                            if (tw < 0 || tw % g.stride) continue;
                            const int wo = tw / g.stride;
                            if (wo >= g.Wo) continue;
                            const float *dp = dy + (((size_t)n * g.Ho + ho) * g.Wo + wo) * Cy;
                            const float *wp = w + ((size_t)a * g.kw + b) * g.Cin + ci;
                            const size_t ws_co = (size_t)g.kh * g.kw * g.Cin;
                            double s = 0.0;
                            for (int co = 0; co < g.Cout; ++co) s += (double)dp[co] * (double)wp[co * ws_co];
                            acc += s;
                        }
                    }
                }
            }
        }
        float v = (float)acc;
        if (mask_x) v = mask_x[i] > 0.0f ? v : v * mask_slope;
        dx[i] = v;
    }
}

// thread = one weight (co, kh, kw, ci): sequential fp64 sum over all output pixels
__global__ __launch_bounds__(256) void k_wgrad(Geo g, const float *__restrict__ x, const float *__restrict__ dy, int Cy, float *__restrict__ dw,
                                               int accumulate)
{
    const size_t total = (size_t)g.Cout * g.kh * g.kw * g.Cin;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % g.Cin);
        size_t t = i / g.Cin;
        const int b = (int)(t % g.kw);
        t /= g.kw;
        const int a = (int)(t % g.kh), co = (int)(t / g.kh);
        double acc = 0.0;
        for (int n = 0; n < g.N; ++n)
            for (int ho = 0; ho < g.Ho; ++ho) {
                const int sh = src_row(g, ho * g.stride + a);
                if (sh < 0) continue;
                double s = 0.0;
                for (int wo = 0; wo < g.Wo; ++wo) {
                    const int sw = src_col(g, wo * g.stride + b);
                    if (sw < 0) continue;
                    s += (double)dy[(((size_t)n * g.Ho + ho) * g.Wo + wo) * Cy + co] * (double)x[(((size_t)n * g.H + sh) * g.W + sw) * g.Cin + ci];
                }
                acc += s;
            }
        dw[i] = (accumulate ? dw[i] : 0.0f) + (float)acc;
    }
}

// bias gradient: column sums of dy; 64 threads per channel, fixed-order combine
__global__ __launch_bounds__(64) void k_dbias(const float *__restrict__ dy, int Cy, size_t P, float *__restrict__ db, int accumulate)
{
    __shared__ double red[64];
    const int co = blockIdx.x;
    double s = 0.0;
    for (size_t p = threadIdx.x; p < P; p += 64) s += (double)dy[p * Cy + co];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
        for (int k = 0; k < 64; ++k) tsum += red[k];
        db[co] = (accumulate ? db[co] : 0.0f) + (float)tsum;
    }
}

// adjoint of the nearest x2 upsample: dx[n,h,w,:] = sum of the 2x2 block of g
__global__ __launch_bounds__(256) void k_fold2x2(const float *__restrict__ gr, float *__restrict__ dx, int N, int H, int W, int C)
{
    const size_t total = (size_t)N * H * W * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t t = i / C;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H), n = (int)(t / H);
        const float *p = gr + (((size_t)n * 2 * H + 2 * h) * 2 * W + 2 * w) * C + c;
        dx[i] = (p[0] + p[C]) + (p[(size_t)2 * W * C] + p[(size_t)2 * W * C + C]);
    }
}

static Geo geo(const m355_conv_desc *d)
{
    Geo g;
    g.N = d->N; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.Cout = d->Cout; g.kh = d->kh; g.kw = d->kw; g.stride = d->stride;
    g.pad_h = d->pad_h; g.pad_w = d->pad_w; g.mode = d->pad_w_mode; g.ups = d->upsample;
    g.Hl = d->H << d->upsample; g.Wl = d->W << d->upsample;
    g.Ho = (g.Hl + 2 * d->pad_h - d->kh) / d->stride + 1;
    g.Wo = (g.Wl + 2 * d->pad_w - d->kw) / d->stride + 1;
    return g;
}

static unsigned blocks_for(size_t total) { return (unsigned)((total + 255) / 256 > 262144 ? 262144 : (total + 255) / 256); }

int conv_fwd(const m355_conv_desc *d, const void *x, const void *w, const float *bias, void *y, int nchw, float slope, hipStream_t st)
{
    const Geo g = geo(d);
    hipLaunchKernelGGL(k_fwd, dim3(blocks_for((size_t)g.N * g.Ho * g.Wo * g.Cout)), dim3(256), 0, st, g, (const float *)x, (const float *)w,
                       bias, (float *)y, nchw, slope);
    note_kernel("k_exact_fwd");
    return check_launch("conv2d_fwd (exact)");
}

int conv_dgrad(const m355_conv_desc *d, const void *dy, int Cy, const void *w, void *dx, const void *mask_x, float mask_slope, hipStream_t st)
{
    const Geo g = geo(d);
    hipLaunchKernelGGL(k_dgrad, dim3(blocks_for((size_t)g.N * g.H * g.W * g.Cin)), dim3(256), 0, st, g, (const float *)dy, Cy, (const float *)w,
                       (float *)dx, (const float *)mask_x, mask_slope);
    note_kernel("k_exact_dgrad");
    return check_launch("conv2d_dgrad (exact)");
}

int conv_wgrad(const m355_conv_desc *d, const void *x, const void *dy, int Cy, float *dw, float *dbias, int accumulate, hipStream_t st)
{
    const Geo g = geo(d);
    hipLaunchKernelGGL(k_wgrad, dim3(blocks_for((size_t)g.Cout * g.kh * g.kw * g.Cin)), dim3(256), 0, st, g, (const float *)x, (const float *)dy,
                       Cy, dw, accumulate);
    if (dbias)
        hipLaunchKernelGGL(k_dbias, dim3(g.Cout), dim3(64), 0, st, (const float *)dy, Cy, (size_t)g.N * g.Ho * g.Wo, dbias, accumulate);
    note_kernel("k_exact_wgrad");
    return check_launch("conv2d_wgrad (exact)");
}

int fold2x2(const void *gr, void *dx, int N, int H, int W, int C, hipStream_t st)
{
    hipLaunchKernelGGL(k_fold2x2, dim3(blocks_for((size_t)N * H * W * C)), dim3(256), 0, st, (const float *)gr, (float *)dx, N, H, W, C);
    return check_launch("fold2x2 (exact)");
}

}  // namespace exact
}  // namespace m355
