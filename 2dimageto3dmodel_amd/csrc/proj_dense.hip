// Dense (volume-materialising) forms of the projection stages, for callers that use the reference's stage API
// directly instead of EffectiveLossFunction.forward:
//   P3  TrilinearInterpolation.trilinear_interpolation   utils/trilinear_interpolation.py:62-74
//   P4  VoxelsSmooth.smooth (one axis per call)           utils/smooth_voxels.py:44-84
//   P5  EffectiveLossFunction.termination_probs           utils/effective_loss_function.py:18-56
// Each is HBM bound: the S^3 volume is written / read once per stage (SURVEY 8d's operator-granular byte count is
// exactly this dataflow).  The fused renderer (proj_render21.hip) is what the training path uses.
#include "common.h"
#include "tiles.h"

namespace m355 {

// ---------------------------------------------------------------------------------------- P3 forward
// grid (tiles, B): splat the tile's records into an LDS tile [rays][S], then write clamp(V,0,1) for the tile's
// rays: every voxel of the volume is written exactly once (empty tiles write zeros), no global atomics.
template <int TH, int TW>
__global__ __launch_bounds__(256) void k_trilinear_fwd(const int *__restrict__ tile_start, const float4 *__restrict__ tile_pts,
                                                       float *__restrict__ vol, float *__restrict__ raw_out, int N, int S,
                                                       int tiles_x, int ntiles, int fixed_weights)
{
    extern __shared__ float tile[];  // [TH*TW][S]
    constexpr int RAYS = TH * TW;
    const int tid = threadIdx.x, b = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x, y0 = ty * TH, x0 = tx * TW;
    const float sm1 = (float)S - 1.0f;
    for (int i = tid; i < RAYS * S; i += 256) tile[i] = 0.0f;
    __syncthreads();
    const int *ts = tile_start + (size_t)b * (ntiles + 1) + blockIdx.x;
    const float4 *pts = tile_pts + (size_t)b * 4 * N;
    for (int pi = ts[0] + tid; pi < ts[1]; pi += 256) {
        const float4 r = pts[pi];
        const float g0 = sm1 * (r.x + 0.5f), g1 = sm1 * (r.y + 0.5f), g2 = sm1 * (r.z + 0.5f);  // tri:34
        const float fl0 = floorf(g0), fl1 = floorf(g1), fl2 = floorf(g2);
        const int f0 = (int)fl0, f1 = (int)fl1, f2 = (int)fl2;
        float w0[2], w1[2], w2[2];
        w0[1] = g0 - fl0; w1[1] = g1 - fl1; w2[1] = g2 - fl2;
        if (fixed_weights) { w0[0] = 1.0f - w0[1]; w1[0] = 1.0f - w1[1]; w2[0] = 1.0f - w2[1]; }
        else { w0[0] = (1.0f - g0) - fl0; w1[0] = (1.0f - g1) - fl1; w2[0] = (1.0f - g2) - fl2; }  // tri:66 literal
        for (int j = 0; j < 2; ++j) {
            const int ry = f1 + j - y0;
            if (ry < 0 || ry >= TH) continue;
            for (int k = 0; k < 2; ++k) {
                const int rx = f2 + k - x0;
                if (rx < 0 || rx >= TW) continue;
                for (int i = 0; i < 2; ++i) atomicAdd(&tile[(ry * TW + rx) * S + f0 + i], w0[i] * w1[j] * w2[k]);
            }
        }
    }
    __syncthreads();
    // vol[b][d][y][x]: consecutive threads -> consecutive x of the tile, then y, then d
    for (int i = tid; i < RAYS * S; i += 256) {
        const int rx = i % TW, ry = (i / TW) % TH, d = i / RAYS;
        const int yy = y0 + ry, xx = x0 + rx;
        if (yy >= S || xx >= S) continue;
        const float v = tile[(ry * TW + rx) * S + d];
        const size_t o = (((size_t)b * S + d) * S + yy) * S + xx;
        vol[o] = fminf(fmaxf(v, 0.0f), 1.0f);  // tri:74
        if (raw_out) raw_out[o] = v;
    }
}

// ---------------------------------------------------------------------------------------- P3 backward
// dvol -> dcam[B,N,3] (pre-zeroed, fp32 atomics: a point straddling tiles is visited by up to 4 workgroups).
// The clamp mask needs the raw splat sum, which is recomputed into LDS.
template <int TH, int TW>
__global__ __launch_bounds__(256) void k_trilinear_bwd(const int *__restrict__ tile_start, const float4 *__restrict__ tile_pts,
                                                       const float *__restrict__ dvol, float *__restrict__ dcam, int N,
                                                       int S, int tiles_x, int ntiles, int fixed_weights)
{
    extern __shared__ float tile[];
    constexpr int RAYS = TH * TW;
    const int tid = threadIdx.x, b = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x, y0 = ty * TH, x0 = tx * TW;
    const float sm1 = (float)S - 1.0f;
    const int *ts = tile_start + (size_t)b * (ntiles + 1) + blockIdx.x;
    const int beg = ts[0], end = ts[1];
    if (beg == end) return;
    const float4 *pts = tile_pts + (size_t)b * 4 * N;
    for (int i = tid; i < RAYS * S; i += 256) tile[i] = 0.0f;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        for (int pi = beg + tid; pi < end; pi += 256) {
            const float4 r = pts[pi];
            const int n = __float_as_int(r.w);
            const float g0 = sm1 * (r.x + 0.5f), g1 = sm1 * (r.y + 0.5f), g2 = sm1 * (r.z + 0.5f);
            const float fl0 = floorf(g0), fl1 = floorf(g1), fl2 = floorf(g2);
            const int f0 = (int)fl0, f1 = (int)fl1, f2 = (int)fl2;
            float w0[2], w1[2], w2[2];
            w0[1] = g0 - fl0; w1[1] = g1 - fl1; w2[1] = g2 - fl2;
            if (fixed_weights) { w0[0] = 1.0f - w0[1]; w1[0] = 1.0f - w1[1]; w2[0] = 1.0f - w2[1]; }
            else { w0[0] = (1.0f - g0) - fl0; w1[0] = (1.0f - g1) - fl1; w2[0] = (1.0f - g2) - fl2; }
            const float dw[2] = {-1.0f, 1.0f};
            float dg0 = 0.f, dg1 = 0.f, dg2 = 0.f;
            for (int j = 0; j < 2; ++j) {
                const int ry = f1 + j - y0;
                if (ry < 0 || ry >= TH) continue;
                for (int k = 0; k < 2; ++k) {
                    const int rx = f2 + k - x0;
                    if (rx < 0 || rx >= TW) continue;
                    for (int i = 0; i < 2; ++i) {
                        float *cell = &tile[(ry * TW + rx) * S + f0 + i];
                        if (pass == 0) {
                            atomicAdd(cell, w0[i] * w1[j] * w2[k]);
                        } else {
                            const float raw = *cell;
                            if (!(raw >= 0.0f && raw <= 1.0f)) continue;  // clamp(0,1) passes min <= x <= max
                            const float gv = dvol[(((size_t)b * S + f0 + i) * S + (y0 + ry)) * S + (x0 + rx)];
                            dg0 += gv * (dw[i] * w1[j] * w2[k]);
                            dg1 += gv * (w0[i] * dw[j] * w2[k]);
                            dg2 += gv * (w0[i] * w1[j] * dw[k]);
                        }
                    }
                }
            }
            if (pass == 1) {
                float *o = dcam + ((size_t)b * N + n) * 3;
                atomicAdd(o, dg0 * sm1);
                atomicAdd(o + 1, dg1 * sm1);
                atomicAdd(o + 2, dg2 * sm1);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------- P4
// out[b][..p..] = sum_t taps[t] * in[b][.. p + t - half ..] along `axis` (zero padded); transpose applies the adjoint.
// Optional epilogue: * scale[b], clamp(0,1) (sm:80-82).
__global__ __launch_bounds__(256) void k_smooth_axis(const float *__restrict__ in, float *__restrict__ out,
                                                     const float *__restrict__ taps, int ntaps, int axis,
                                                     const float *__restrict__ scale, int transpose, int S)
{
    const int b = blockIdx.y;
    const size_t vol = (size_t)S * S * S;
    const size_t stride = axis == 0 ? (size_t)S * S : (axis == 1 ? (size_t)S : 1);
    const int half = ntaps >> 1;
    const float sc = scale ? scale[b] : 1.0f;
    const float *ib = in + (size_t)b * vol;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < vol; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % S), y = (int)((i / S) % S), z = (int)(i / ((size_t)S * S));
        const int pos = axis == 0 ? z : (axis == 1 ? y : x);
        float acc = 0.0f;
        for (int t = 0; t < ntaps; ++t) {
            const int q = transpose ? pos - t + half : pos + t - half;
            if (q < 0 || q >= S) continue;
            acc = fmaf(taps[t], ib[i + (ptrdiff_t)(q - pos) * (ptrdiff_t)stride], acc);
        }
        if (scale) acc = fminf(fmaxf(acc * sc, 0.0f), 1.0f);
        out[(size_t)b * vol + i] = acc;
    }
}

// backward of the epilogue y = clamp(pre*scale[b], 0, 1): dpre = dout*scale*mask ; dscale[b] += sum dout*pre*mask
__global__ __launch_bounds__(256) void k_scale_clamp_bwd(const float *__restrict__ pre, const float *__restrict__ scale,
                                                         const float *__restrict__ dout, float *__restrict__ dpre,
                                                         float *__restrict__ dscale, size_t per_sample)
{
    const int b = blockIdx.y;
    const float sc = scale[b];
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (size_t)gridDim.x * 256) {
        const size_t o = (size_t)b * per_sample + i;
        const float p = pre[o], v = p * sc;
        const float g = (v >= 0.0f && v <= 1.0f) ? dout[o] : 0.0f;
        dpre[o] = g * sc;
        acc += g * p;
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(dscale + b, red[0]);
}

// ---------------------------------------------------------------------------------------- P5
// one thread per ray (b,h,w), sequential over depth, coalesced along w.  The cumulative sum accumulates in fp64 and
// is rounded to fp32 at every prefix, as ATen's CPU cumsum does for float tensors.
__global__ __launch_bounds__(256) void k_termination_fwd(const float *__restrict__ V, float *__restrict__ T, int D,
                                                         size_t hw, float eps)
{
    const int b = blockIdx.y;
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= hw) return;
    const float hi = (float)(1.0 - (double)eps);
    const float *col = V + (size_t)b * D * hw + r;
    float *out = T + (size_t)b * (D + 1) * hw + r;
    double acc = 0.0;
    float prevL = eps;  // the "zeros_matrix" is epsilon-filled (elf:40-41,48)
    for (int d = 0; d < D; ++d) {
        const float v = col[(size_t)d * hw];
        const float o = fminf(fmaxf(v, eps), hi);  // elf:32
        out[(size_t)d * hw] = expf(prevL + logf(o));  // elf:54-56
        acc += (double)logf(1.0f - o);                // elf:34,37
        prevL = (float)acc;
    }
    out[(size_t)D * hw] = expf(prevL + eps);
}

// dT[B,D+1,H,W] -> dV[B,D,H,W]; dV doubles as the scratch for dsum[d] = dT[d]*T[d] between the two sweeps.
__global__ __launch_bounds__(256) void k_termination_bwd(const float *__restrict__ V, const float *__restrict__ dT,
                                                         float *__restrict__ dV, int D, size_t hw, float eps)
{
    const int b = blockIdx.y;
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= hw) return;
    const float hi = (float)(1.0 - (double)eps);
    const float *col = V + (size_t)b * D * hw + r;
    const float *g = dT + (size_t)b * (D + 1) * hw + r;
    float *out = dV + (size_t)b * D * hw + r;
    double acc = 0.0;
    float prevL = eps;
    for (int d = 0; d < D; ++d) {
        const float o = fminf(fmaxf(col[(size_t)d * hw], eps), hi);
        const float t = expf(prevL + logf(o));
        out[(size_t)d * hw] = g[(size_t)d * hw] * t;
        acc += (double)logf(1.0f - o);
        prevL = (float)acc;
    }
    float suffix = g[(size_t)D * hw] * expf(prevL + eps);  // sum_{m>d} dsum[m], starting with the background slice
    for (int d = D - 1; d >= 0; --d) {
        const float v = col[(size_t)d * hw];
        const float o = fminf(fmaxf(v, eps), hi);
        const float ds = out[(size_t)d * hw];
        const float d_o = ds / o - suffix / (1.0f - o);
        suffix += ds;
        out[(size_t)d * hw] = (v >= eps && v <= hi) ? d_o : 0.0f;
    }
}

}  // namespace m355

using namespace m355;

template <typename F>
static int tile_dispatch(int S, F f)
{
    TileShape c;
    if (!tile_shape(S, c)) {
        set_error("S=%d not supported (2..512)", S);
        return M355_ERR_UNSUPPORTED;
    }
    f(c);
    return 0;
}

extern "C" int m355_trilinear_fwd(const int32_t *tile_start, const float *tile_pts, float *vol, float *raw, int B, int N,
                                  int S, int flags, void *stream)
{
    M355_REQUIRE(tile_start && vol && (tile_pts || N == 0), "trilinear_fwd: null pointer");
    M355_REQUIRE(B >= 0 && N >= 0 && B <= 65535, "trilinear_fwd: bad size B=%d N=%d", B, N);
    if (B == 0) return M355_OK;
    hipStream_t st = (hipStream_t)stream;
    const int fw = (flags & M355_FIXED_WEIGHTS) ? 1 : 0;
    int rc = tile_dispatch(S, [&](const TileShape &c) {
        const int tiles_x = (S + c.tw - 1) / c.tw, ntiles = tile_count(S);
        const size_t lds = sizeof(float) * c.th * c.tw * S;
        dim3 grid(ntiles, B);
        if (c.th == 8) hipLaunchKernelGGL((k_trilinear_fwd<8, 8>), grid, dim3(256), lds, st, (const int *)tile_start, (const float4 *)tile_pts, vol, raw, N, S, tiles_x, ntiles, fw);
        else if (c.tw == 8) hipLaunchKernelGGL((k_trilinear_fwd<4, 8>), grid, dim3(256), lds, st, (const int *)tile_start, (const float4 *)tile_pts, vol, raw, N, S, tiles_x, ntiles, fw);
        else hipLaunchKernelGGL((k_trilinear_fwd<4, 4>), grid, dim3(256), lds, st, (const int *)tile_start, (const float4 *)tile_pts, vol, raw, N, S, tiles_x, ntiles, fw);
    });
    return rc ? rc : check_launch("trilinear_fwd");
}

extern "C" int m355_trilinear_bwd(const int32_t *tile_start, const float *tile_pts, const float *dvol, float *dcam, int B,
                                  int N, int S, int flags, void *stream)
{
    M355_REQUIRE(B >= 0 && N >= 0 && B <= 65535, "trilinear_bwd: bad size B=%d N=%d", B, N);
    if (B == 0 || N == 0) return M355_OK;
    M355_REQUIRE(tile_start && tile_pts && dvol && dcam, "trilinear_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int fw = (flags & M355_FIXED_WEIGHTS) ? 1 : 0;
    if (hipMemsetAsync(dcam, 0, sizeof(float) * (size_t)B * N * 3, st) != hipSuccess) {
        set_error("trilinear_bwd: memset failed");
        return M355_ERR_LAUNCH;
    }
    int rc = tile_dispatch(S, [&](const TileShape &c) {
        const int tiles_x = (S + c.tw - 1) / c.tw, ntiles = tile_count(S);
        const size_t lds = sizeof(float) * c.th * c.tw * S;
        dim3 grid(ntiles, B);
        if (c.th == 8) hipLaunchKernelGGL((k_trilinear_bwd<8, 8>), grid, dim3(256), lds, st, (const int *)tile_start, (const float4 *)tile_pts, dvol, dcam, N, S, tiles_x, ntiles, fw);
        else if (c.tw == 8) hipLaunchKernelGGL((k_trilinear_bwd<4, 8>), grid, dim3(256), lds, st, (const int *)tile_start, (const float4 *)tile_pts, dvol, dcam, N, S, tiles_x, ntiles, fw);
        else hipLaunchKernelGGL((k_trilinear_bwd<4, 4>), grid, dim3(256), lds, st, (const int *)tile_start, (const float4 *)tile_pts, dvol, dcam, N, S, tiles_x, ntiles, fw);
    });
    return rc ? rc : check_launch("trilinear_bwd");
}

extern "C" int m355_smooth_axis(const float *in, float *out, const float *taps, int ntaps, int axis, const float *scale,
                                int transpose, int B, int S, void *stream)
{
    M355_REQUIRE(in && out && taps && in != out, "smooth_axis: null pointer / in-place");
    M355_REQUIRE(B >= 1 && S >= 1 && B <= 65535 && axis >= 0 && axis <= 2, "smooth_axis: bad size B=%d S=%d axis=%d", B, S, axis);
    M355_REQUIRE(ntaps >= 1 && ntaps <= 63 && (ntaps & 1), "smooth_axis: ntaps=%d must be odd and <= 63", ntaps);
    const size_t vol = (size_t)S * S * S;
    const unsigned gx = (unsigned)(vol / 256 > 2048 ? 2048 : (vol + 255) / 256);
    hipLaunchKernelGGL(k_smooth_axis, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, in, out, taps, ntaps, axis, scale,
                       transpose, S);
    return check_launch("smooth_axis");
}

extern "C" int m355_scale_clamp_bwd(const float *pre, const float *scale, const float *dout, float *dpre, float *dscale,
                                    int B, size_t per_sample, void *stream)
{
    M355_REQUIRE(pre && scale && dout && dpre && dscale && B >= 1 && B <= 65535, "scale_clamp_bwd: null pointer / bad B");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dscale, 0, sizeof(float) * B, st) != hipSuccess) {
        set_error("scale_clamp_bwd: memset failed");
        return M355_ERR_LAUNCH;
    }
    const unsigned gx = (unsigned)(per_sample / 256 > 1024 ? 1024 : (per_sample + 255) / 256);
    hipLaunchKernelGGL(k_scale_clamp_bwd, dim3(gx, B), dim3(256), 0, st, pre, scale, dout, dpre, dscale, per_sample);
    return check_launch("scale_clamp_bwd");
}

extern "C" int m355_termination_fwd(const float *vol, float *T, int B, int D, int H, int W, float eps, void *stream)
{
    M355_REQUIRE(vol && T && B >= 1 && D >= 1 && H >= 1 && W >= 1 && B <= 65535, "termination_fwd: null pointer / bad size");
    const size_t hw = (size_t)H * W;
    hipLaunchKernelGGL(k_termination_fwd, dim3((unsigned)((hw + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, vol, T, D,
                       hw, eps);
    return check_launch("termination_fwd");
}

extern "C" int m355_termination_bwd(const float *vol, const float *dT, float *dvol, int B, int D, int H, int W, float eps,
                                    void *stream)
{
    M355_REQUIRE(vol && dT && dvol && B >= 1 && D >= 1 && H >= 1 && W >= 1 && B <= 65535, "termination_bwd: null pointer / bad size");
    const size_t hw = (size_t)H * W;
    hipLaunchKernelGGL(k_termination_bwd, dim3((unsigned)((hw + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, vol, dT,
                       dvol, D, hw, eps);
    return check_launch("termination_bwd");
}
