// P7/P8: silhouette loss.  SupervisedLoss.forward (models/supervised_part.py:68-72) and the per-cloud
// squared error that UnsupervisedLoss.forward (models/unsupervised_part.py:108-126) builds on.
//
//   m   = F.interpolate(mask[None], scale_factor=1/2, mode="bilinear", align_corners=True)   (sup:70)
//   diff = proj - m ; sse[b] = sum diff^2 ; total = sum_b sse[b]
// SupervisedLoss = total / (2B); its gradient w.r.t. proj is diff / B, which the render backward applies
// through its `gmul` argument, so `diff` doubles as the saved tensor.
//
// Bilinear source index follows ATen's upsample_bilinear2d with align_corners: src = dst*(in-1)/(out-1),
// lambda in fp32.  HBM-bound: 4 B proj + 16 B mask taps in, 4 B diff out per pixel.  Deterministic
// two-stage reduction (per-chunk fp64 partials, then one block).
#include "common.h"

namespace m355 {

constexpr int kChunk = 4096;  // pixels per workgroup

__global__ __launch_bounds__(256) void k_sil_loss(const float *__restrict__ proj, const float *__restrict__ mask,
                                                   int Hin, int Win, int mask_repeat,
                                                   float *__restrict__ diff, double *__restrict__ part, int S,
                                                   int nchunks)
{
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int npix = S * S;
    const float sh = S > 1 ? (float)(Hin - 1) / (float)(S - 1) : 0.0f;
    const float sw = S > 1 ? (float)(Win - 1) / (float)(S - 1) : 0.0f;
    const float *mb = mask + (size_t)(b / mask_repeat) * Hin * Win;
    double acc = 0.0;
    const int end = min(npix, (chunk + 1) * kChunk);
    for (int p = chunk * kChunk + tid; p < end; p += 256) {
        const int y = p / S, x = p - y * S;
        const float fy = sh * (float)y, fx = sw * (float)x;
        const int yi = (int)fy, xi = (int)fx;
        const int y1 = yi + (yi < Hin - 1 ? 1 : 0), x1 = xi + (xi < Win - 1 ? 1 : 0);
        const float ly = fy - (float)yi, lx = fx - (float)xi;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const float m = hy * (hx * mb[(size_t)yi * Win + xi] + lx * mb[(size_t)yi * Win + x1]) +
                        ly * (hx * mb[(size_t)y1 * Win + xi] + lx * mb[(size_t)y1 * Win + x1]);
        const float d = proj[(size_t)b * npix + p] - m;
        diff[(size_t)b * npix + p] = d;
        acc += (double)(d * d);
    }
    __shared__ double red[256];
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) part[(size_t)b * nchunks + chunk] = red[0];
}

__global__ __launch_bounds__(256) void k_sil_loss_final(const double *__restrict__ part, int nchunks,
                                                         float *__restrict__ sse, float *__restrict__ total, int B)
{
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int b = tid; b < B; b += 256) {
        double s = 0.0;
        for (int c = 0; c < nchunks; ++c) s += part[(size_t)b * nchunks + c];
        if (sse) sse[b] = (float)s;
        acc += s;
    }
    __shared__ double red[256];
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0 && total) total[0] = (float)red[0];
}

}  // namespace m355

extern "C" size_t m355_sil_loss_ws_bytes(int B, int S)
{
    if (B <= 0 || S <= 0) return 0;
    const size_t nchunks = ((size_t)S * S + m355::kChunk - 1) / m355::kChunk;
    return sizeof(double) * nchunks * (size_t)B;
}

extern "C" int m355_sil_loss_fwd(const float *proj, const float *mask, int Hin, int Win, int mask_repeat, float *diff,
                                 float *sse, float *total, void *ws, int B, int S, void *stream)
{
    M355_REQUIRE(proj && mask && diff && ws, "sil_loss_fwd: null pointer");
    M355_REQUIRE(B >= 1 && S >= 1 && Hin >= 1 && Win >= 1, "sil_loss_fwd: bad size B=%d S=%d Hin=%d Win=%d", B, S, Hin,
                 Win);
    M355_REQUIRE(Hin / 2 == S && Win / 2 == S, "sil_loss_fwd: mask %dx%d does not halve to %d", Hin, Win, S);
    M355_REQUIRE(B <= 65535, "sil_loss_fwd: B=%d exceeds grid.y", B);
    M355_REQUIRE(mask_repeat >= 1 && B % mask_repeat == 0, "sil_loss_fwd: mask_repeat=%d does not divide B=%d", mask_repeat, B);
    const int nchunks = (S * S + m355::kChunk - 1) / m355::kChunk;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(m355::k_sil_loss, dim3(nchunks, B), dim3(256), 0, st, proj, mask, Hin, Win, mask_repeat, diff,
                       (double *)ws, S, nchunks);
    int rc = m355::check_launch("sil_loss_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(m355::k_sil_loss_final, dim3(1), dim3(256), 0, st, (const double *)ws, nchunks, sse, total, B);
    return m355::check_launch("sil_loss_final");
}
