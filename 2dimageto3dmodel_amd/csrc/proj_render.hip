// P3..P6 fused: EffectiveLossFunction.forward / backward without ever materialising the S^3 volume.
//
// Reference dataflow (utils/effective_loss_function.py:58-81, literal semantics + shims S0/S1):
//   trilinear splat (trilinear_interpolation.py:37-74) -> clamp(0,1) -> 1-D smoothing along DEPTH only
//   (smooth_voxels.py:66-73 overwrites instead of chaining, so only the last = depth kernel survives)
//   -> * scale[b], clamp(0,1) (sm:80-82) -> termination_probs (elf:18-56) -> sum over depth, flip y (elf:81).
// After the splat every step acts on one depth column ("ray") at a time, so a workgroup owns a tile of
// TH x TW rays, keeps their S-deep columns in a 32 KiB LDS tile, and produces TH*TW silhouette pixels.
// HBM traffic is the compulsory I/O only (cam/raykey in, proj out); the ~50 volume-sized passes of the
// reference dataflow become LDS traffic.
//
// Work decomposition inside a workgroup (256 threads = 4 waves):
//   phase 1  zero the LDS tile
//   phase 2  scan the cloud's per-point ray keys (4 B/point); points touching the tile ds_add_f32 their
//            8 corner weights into the tile (products evaluated left to right as tri:40-41 does)
//   phase 3  one wave per ray, lanes along depth (lane owns R consecutive depths, R = ceil(S/64)):
//            untouched rays take a precomputed constant; touched rays run
//              clamp -> sparse broadcast convolution (only non-zero voxels emit taps) -> scale/clamps ->
//              prefix PRODUCT of (1-o) in fp64 (== exp(cumsum(log(1-o))) of elf:34-37, whose CPU cumsum
//              accumulates in double) -> T_d = P_{d-1} * o_d -> wave sum.
//   backward adds: suffix sums of T, the clamp masks, dscale, the transposed depth convolution through
//   the LDS row, and phase 4: points touching the tile gather dV at their 8 corners and write one
//   gradient slot per ray (j,k) -- exactly one writer per slot, no atomics, deterministic.
//
// Numerics: T_d = exp(L_{d-1} + log o_d) in the reference; here P_{d-1} * o_d with P the running product of
// the SAME fp32 values float(1 - o_j) promoted to fp64.  Agreement with the reference ~5e-7 relative per
// pixel (its own fp32 rounding of L + log o), far inside the 1e-4 contract on the loss.
#include "common.h"
#include "proj_render21.h"
#include "tiles.h"

namespace m355 {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxTaps = 63;

__device__ __forceinline__ double wave_sum_d(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__device__ __forceinline__ float wave_sum_f(float x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// inclusive scans across the 64 lanes of a wave
__device__ __forceinline__ double wave_incl_prod_d(double x, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double y = __shfl_up(x, off, 64);
        if (lane >= off) x *= y;
    }
    return x;
}

__device__ __forceinline__ double wave_incl_sum_d(double x, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    return x;
}

struct RenderArgs {
    const int32_t *tile_start;  // [B, ntiles+1]   (m355_proj_bin_fwd)
    const float *tile_pts;      // [B, 4N] float4 records (c0,c1,c2,n)
    const float *scale;         // nullable
    const float *taps;
    int ntaps;
    float *proj;           // fwd
    const float *dproj;    // bwd
    float gmul;            // bwd
    float *dcam_slots;     // bwd [B,N,4,3]
    float *dscale_part;    // bwd [B,nparts]
    int N, S, tiles_x, tiles_y;
    int fixed_weights;
    int taps_from_sigma;  // a.taps points at the sigma scalar; taps are built in-kernel (sm:24-31)
    int true_gaussian;    // exp(-x^2/2s^2) instead of the literal exp(+x^2/2s^2) (defect D4)
};

// VoxelsSmooth.separate_kernels (utils/smooth_voxels.py:14-42): x = -k//2+1 .. k//2, literal sign as written.
// den = 2*pow(std_dev,2) is evaluated in double on the python float and rounded to fp32 by the tensor op.
__device__ __forceinline__ float tap_unnormalised(int t, int half, float sigma, int true_gaussian)
{
    const float den = (float)(2.0 * (double)sigma * (double)sigma);
    const float x = (float)(t - half);
    const float e = (x * x) / den;
    return expf(true_gaussian ? -e : e);
}

__global__ void k_smooth_taps(const float *sigma, int ntaps, int true_gaussian, float *taps)
{
    __shared__ float raw[64];
    const int t = threadIdx.x;
    if (t < ntaps) raw[t] = tap_unnormalised(t, ntaps >> 1, sigma[0], true_gaussian);
    __syncthreads();
    float s = 0.0f;
    for (int i = 0; i < ntaps; ++i) s += raw[i];
    if (t < ntaps) taps[t] = raw[t] / s;
}

template <int R>
struct Ray {
    float vraw[R], vc[R], sm[R], o[R], qf[R];
    bool live[R];  // depth < S
};

// Everything the forward needs for one ray; returns the silhouette value (valid in all lanes).
// When BWD, also leaves T-related state in registers via the out params.
template <int R, bool BWD>
__device__ __forceinline__ float ray_forward(Ray<R> &ry, float *row, const float *tapP, int SP, int S, int ntaps,
                                             bool has_scale, float scale, int lane, bool force_empty,
                                             double (&pex)[R], double &tsum_lane)
{
    const int half = ntaps >> 1;
    const float eps = 1e-5f, hi = (float)(1.0 - 1e-5);  // elf:18,32  clamp(epsilon, 1.0 - epsilon)
    const double E = 1.0000100000500002;                 // exp(1e-5f): T_0 = exp(eps + log o_0), elf:40-41,48
    // ---- load + clamp (tri:74)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int d = lane * R + r;
        ry.live[r] = d < S;
        float v = force_empty ? 0.0f : row[d];
        ry.vraw[r] = v;
        ry.vc[r] = fminf(fmaxf(v, 0.0f), 1.0f);
        ry.sm[r] = 0.0f;
    }
    // ---- depth convolution, scatter form: every non-zero voxel z adds vc[z]*tap[d-z+half] to its
    //      neighbours d (smooth_voxels.py:72 with the depth kernel; zero padding is implicit)
    if (!force_empty) {
#pragma unroll
        for (int rz = 0; rz < R; ++rz) {
            unsigned long long m = __ballot(ry.vc[rz] != 0.0f);
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const float val = __shfl(ry.vc[rz], l, 64);
                const int z = l * R + rz;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int idx = (lane * R + r) - z + half + SP;  // padded table: zero outside the taps
                    ry.sm[r] = fmaf(val, tapP[idx], ry.sm[r]);
                }
            }
        }
    }
    // ---- scale/clamp (sm:80-82), occupancy clamp (elf:32), q = 1 - o in fp32 exactly as elf:34
    double run = 1.0;
    double lp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float c = ry.sm[r];
        if (has_scale) c = fminf(fmaxf(c * scale, 0.0f), 1.0f);
        float o = fminf(fmaxf(c, eps), hi);
        float qf = 1.0f - o;
        if (!ry.live[r]) {
            o = 0.0f;
            qf = 1.0f;
        }
        ry.o[r] = o;
        ry.qf[r] = qf;
        run *= (double)qf;
        lp[r] = run;
    }
    // ---- exclusive prefix product over depth (== exp of the exclusive cumsum of log(1-o))
    const double incl = wave_incl_prod_d(run, lane);
    double ex = __shfl_up(incl, 1, 64);
    if (lane == 0) ex = 1.0;
    double tsum = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double p = (r == 0) ? ex : ex * lp[r - 1];
        if (lane == 0 && r == 0) p = E;
        pex[r] = p;
        tsum += p * (double)ry.o[r];
    }
    tsum_lane = tsum;
    return (float)wave_sum_d(tsum);
}

template <int R, int TH, int TW, bool BWD>
__global__ __launch_bounds__(kThreads) void k_render(RenderArgs a)
{
    constexpr int RAYS = TH * TW;
    constexpr int SP = 64 * R;
    constexpr int TAPN = 2 * SP + 64;
    __shared__ __attribute__((aligned(16))) float tile[RAYS * SP];
    __shared__ float tapP[TAPN];
    __shared__ int rayflag[RAYS];
    __shared__ int tile_touched;
    __shared__ float wave_ds[kWaves];
    __shared__ float tapraw[64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x % a.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int S = a.S, N = a.N, ntaps = a.ntaps, half = ntaps >> 1;
    const float sm1 = (float)S - 1.0f;  // tri:34
    const bool has_scale = a.scale != nullptr;
    const float scale = has_scale ? a.scale[b] : 1.0f;

    // ---- phase 1: zero tile, build the zero-padded tap table
    {
        float4 *t4 = reinterpret_cast<float4 *>(tile);
        for (int i = tid; i < RAYS * SP / 4; i += kThreads) t4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        float tsum = 1.0f;
        if (a.taps_from_sigma) {
            if (tid < ntaps) tapraw[tid] = tap_unnormalised(tid, half, a.taps[0], a.true_gaussian);
            __syncthreads();
            tsum = 0.0f;
            for (int i = 0; i < ntaps; ++i) tsum += tapraw[i];
        }
        for (int i = tid; i < TAPN; i += kThreads) {
            const int t = i - SP;
            float v = 0.0f;
            if (t >= 0 && t < ntaps) v = a.taps_from_sigma ? tapraw[t] / tsum : a.taps[t];
            tapP[i] = v;
        }
        for (int i = tid; i < RAYS; i += kThreads) rayflag[i] = 0;
        if (tid == 0) tile_touched = 0;
    }
    __syncthreads();

    // ---- phase 2: splat the records the binning kernel filed under this tile
    const int ntiles = a.tiles_x * a.tiles_y;
    const int *ts = a.tile_start + (size_t)b * (ntiles + 1) + blockIdx.x;
    const int beg = ts[0], end = ts[1];
    const float4 *pts = reinterpret_cast<const float4 *>(a.tile_pts) + (size_t)b * 4 * N;
    for (int pi = beg + tid; pi < end; pi += kThreads) {
        const float4 rec = pts[pi];
        const float c0 = rec.x, c1 = rec.y, c2 = rec.z;
        const int f1 = (int)floorf(sm1 * (c1 + 0.5f)), f2 = (int)floorf(sm1 * (c2 + 0.5f));
        const float g0 = sm1 * (c0 + 0.5f), g1 = sm1 * (c1 + 0.5f), g2 = sm1 * (c2 + 0.5f);  // tri:34
        const float fl0 = floorf(g0), fl1 = floorf(g1), fl2 = floorf(g2);
        const int f0 = (int)fl0;
        float w0[2], w1[2], w2[2];  // tri:66  [1.0 - grid - floor, grid - floor]
        w0[1] = g0 - fl0;
        w1[1] = g1 - fl1;
        w2[1] = g2 - fl2;
        if (a.fixed_weights) {
            w0[0] = 1.0f - w0[1];
            w1[0] = 1.0f - w1[1];
            w2[0] = 1.0f - w2[1];
        } else {
            w0[0] = (1.0f - g0) - fl0;
            w1[0] = (1.0f - g1) - fl1;
            w2[0] = (1.0f - g2) - fl2;
        }
        tile_touched = 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ry = f1 + j - y0;
            if (ry < 0 || ry >= TH) continue;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int rx = f2 + k - x0;
                if (rx < 0 || rx >= TW) continue;
                const int ray = ry * TW + rx;
                rayflag[ray] = 1;
                float *col = tile + ray * SP + f0;
#pragma unroll
                for (int i = 0; i < 2; ++i) atomicAdd(col + i, w0[i] * w1[j] * w2[k]);  // tri:40-41,58
            }
        }
    }
    __syncthreads();

    // ---- phase 3: rays
    Ray<R> ry;
    double pex[R], tsum_lane;
    const bool any = tile_touched != 0;
    float proj_empty = 0.0f;
    if (!BWD) proj_empty = ray_forward<R, false>(ry, tile, tapP, SP, S, ntaps, has_scale, scale, lane, true, pex,
                                                 tsum_lane);
    float ds_acc = 0.0f;
    for (int ray = wave; ray < RAYS; ray += kWaves) {
        const int yy = y0 + ray / TW, xx = x0 + ray % TW;
        if (yy >= S || xx >= S) continue;  // wave-uniform
        const size_t pix = ((size_t)b * S + (S - 1 - yy)) * S + xx;  // elf:81 flip(1)
        if (!any || rayflag[ray] == 0) {
            if (!BWD && lane == 0) a.proj[pix] = proj_empty;
            continue;
        }
        float *row = tile + ray * SP;
        const float pr = ray_forward<R, BWD>(ry, row, tapP, SP, S, ntaps, has_scale, scale, lane, false, pex,
                                             tsum_lane);
        if (!BWD) {
            if (lane == 0) a.proj[pix] = pr;
            continue;
        }
        // ================= backward of one ray =================
        const double g = (double)(a.dproj[pix] * a.gmul);
        const float eps = 1e-5f, hi = (float)(1.0 - 1e-5);
        // inclusive prefix sums of T over depth -> suffix sums (reverse cumsum of elf:37's autograd)
        double Tl[R], ls = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            ls += pex[r] * (double)ry.o[r];
            Tl[r] = ls;
        }
        const double inclT = wave_incl_sum_d(ls, lane);
        const double exT = inclT - ls;
        const double Ttot = __shfl(inclT, 63, 64);
        float dsm[R];
        float ds_lane = 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double suf = Ttot - (exT + Tl[r]);
            // d/do [ log o ] * dl + d/do [ log(1-o) ] * da  with dl = g*T, da = g*suffix (autograd of elf:34-56)
            float d_o = (float)(g * (pex[r] - suf / (double)ry.qf[r]));
            float c = ry.sm[r];
            bool pass = true;
            if (has_scale) {
                const float cs = ry.sm[r] * scale;
                pass = (cs >= 0.0f && cs <= 1.0f);       // clamp(0,1) of sm:82
                c = fminf(fmaxf(cs, 0.0f), 1.0f);
            }
            pass = pass && (c >= eps && c <= hi) && ry.live[r];  // clamp(eps,1-eps) of elf:32
            const float dsp = pass ? d_o : 0.0f;
            ds_lane += dsp * ry.sm[r];
            dsm[r] = has_scale ? dsp * scale : dsp;
        }
        if (has_scale) ds_acc += wave_sum_f(ds_lane);
        // transposed depth convolution through the LDS row: dV[z] = sum_t tap[t] * dsm[z - t + half]
#pragma unroll
        for (int r = 0; r < R; ++r) row[lane * R + r] = dsm[r];
        float dv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) dv[r] = 0.0f;
        for (int t = 0; t < ntaps; ++t) {
            const float tp = tapP[SP + t];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int d = lane * R + r - t + half;
                const float v = (d >= 0 && d < S) ? row[d] : 0.0f;
                dv[r] = fmaf(tp, v, dv[r]);
            }
        }
        // clamp(0,1) mask of tri:74 on the raw splat sum, then park dV in the tile for phase 4
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool pass = ry.vraw[r] >= 0.0f && ry.vraw[r] <= 1.0f;
            row[lane * R + r] = pass ? dv[r] : 0.0f;
        }
    }
    if (!BWD) return;

    if (lane == 0) wave_ds[wave] = ds_acc;
    __syncthreads();
    if (tid == 0 && a.dscale_part) {
        float s = 0.0f;
        for (int w = 0; w < kWaves; ++w) s += wave_ds[w];
        a.dscale_part[(size_t)b * (a.tiles_x * a.tiles_y) + blockIdx.x] = s;
    }
    if (!any) return;

    // ---- phase 4: gather dV at the 8 corners of every point touching the tile
    for (int pi = beg + tid; pi < end; pi += kThreads) {
        const float4 rec = pts[pi];
        const float c0 = rec.x, c1 = rec.y, c2 = rec.z;
        const int n = __float_as_int(rec.w);
        const int f1 = (int)floorf(sm1 * (c1 + 0.5f)), f2 = (int)floorf(sm1 * (c2 + 0.5f));
        const float g0 = sm1 * (c0 + 0.5f), g1 = sm1 * (c1 + 0.5f), g2 = sm1 * (c2 + 0.5f);
        const float fl0 = floorf(g0), fl1 = floorf(g1), fl2 = floorf(g2);
        const int f0 = (int)fl0;
        float w0[2], w1[2], w2[2];
        w0[1] = g0 - fl0;
        w1[1] = g1 - fl1;
        w2[1] = g2 - fl2;
        if (a.fixed_weights) {
            w0[0] = 1.0f - w0[1];
            w1[0] = 1.0f - w1[1];
            w2[0] = 1.0f - w2[1];
        } else {
            w0[0] = (1.0f - g0) - fl0;
            w1[0] = (1.0f - g1) - fl1;
            w2[0] = (1.0f - g2) - fl2;
        }
        const float dw[2] = {-1.0f, 1.0f};  // d w[0]/dg = -1, d w[1]/dg = +1 (floor has zero gradient)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ryy = f1 + j - y0;
            if (ryy < 0 || ryy >= TH) continue;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int rx = f2 + k - x0;
                if (rx < 0 || rx >= TW) continue;
                const float *col = tile + (ryy * TW + rx) * SP + f0;
                float dg0 = 0.f, dg1 = 0.f, dg2 = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float gv = col[i];
                    dg0 += gv * (dw[i] * w1[j] * w2[k]);
                    dg1 += gv * (w0[i] * dw[j] * w2[k]);
                    dg2 += gv * (w0[i] * w1[j] * dw[k]);
                }
                float *out = a.dcam_slots + (((size_t)b * N + n) * 4 + (j * 2 + k)) * 3;
                out[0] = dg0 * sm1;  // g = (S-1)(c+0.5)
                out[1] = dg1 * sm1;
                out[2] = dg2 * sm1;
            }
        }
    }
}

template <bool BWD>
static int launch_render(RenderArgs a, int B, hipStream_t st)
{
    TileShape c;
    if (!tile_shape(a.S, c)) {
        set_error("proj_render: S=%d not supported by the fused kernel (max 512)", a.S);
        return M355_ERR_UNSUPPORTED;
    }
    const int R = (a.S + 63) / 64 > 4 ? 8 : ((a.S + 63) / 64 > 2 ? 4 : (a.S + 63) / 64);
    a.tiles_x = (a.S + c.tw - 1) / c.tw;
    a.tiles_y = (a.S + c.th - 1) / c.th;
    dim3 grid(a.tiles_x * a.tiles_y, B), block(kThreads);
    switch (R) {
        case 1: hipLaunchKernelGGL((k_render<1, 8, 8, BWD>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((k_render<2, 8, 8, BWD>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((k_render<4, 4, 8, BWD>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((k_render<8, 4, 4, BWD>), grid, block, 0, st, a); break;
    }
    return check_launch(BWD ? "proj_render_bwd" : "proj_render_fwd");
}

}  // namespace m355

extern "C" int m355_smooth_taps(const float *sigma, int ntaps, int flags, float *taps, void *stream)
{
    M355_REQUIRE(sigma && taps, "smooth_taps: null pointer");
    M355_REQUIRE(ntaps >= 1 && ntaps <= m355::kMaxTaps && (ntaps & 1), "smooth_taps: ntaps=%d must be odd and <= %d", ntaps,
                 m355::kMaxTaps);
    hipLaunchKernelGGL(m355::k_smooth_taps, dim3(1), dim3(64), 0, (hipStream_t)stream, sigma, ntaps,
                       (flags & M355_TRUE_GAUSSIAN) ? 1 : 0, taps);
    return m355::check_launch("smooth_taps");
}

extern "C" int m355_proj_render_fwd(const int32_t *tile_start, const float *tile_pts, const float *scale,
                                    const float *taps, int ntaps, float *proj, int B, int N, int S, int flags,
                                    void *stream)
{
    M355_REQUIRE(tile_start && (tile_pts || N == 0) && taps && proj, "proj_render_fwd: null pointer");
    M355_REQUIRE(B >= 0 && N >= 0 && S >= 2, "proj_render_fwd: bad size B=%d N=%d S=%d", B, N, S);
    M355_REQUIRE(ntaps >= 1 && ntaps <= m355::kMaxTaps && (ntaps & 1), "proj_render_fwd: ntaps=%d must be odd and <= %d",
                 ntaps, m355::kMaxTaps);
    M355_REQUIRE(B <= 65535, "proj_render_fwd: B=%d exceeds grid.y", B);
    if (B == 0) return M355_OK;
    if (ntaps == 21 && !(flags & M355_TAPS_FROM_SIGMA)) {
        m355::Render21Args r = {};
        r.tile_start = tile_start;
        r.tile_pts = tile_pts;
        r.scale = scale;
        r.taps = taps;
        r.proj = proj;
        r.empty_val = m355::render_empty_value(S);
        r.N = N;
        r.S = S;
        r.fixed_weights = (flags & M355_FIXED_WEIGHTS) ? 1 : 0;
        r.det_scale = (flags & M355_DET_SPLAT) ? 1.0f : 0.0f;   // (the launcher picks the actual scale)
        return m355::launch_render21<false>(r, B, (hipStream_t)stream);
    }
    m355::RenderArgs a = {};
    a.tile_start = tile_start;
    a.tile_pts = tile_pts;
    a.scale = scale;
    a.taps = taps;
    a.ntaps = ntaps;
    a.proj = proj;
    a.N = N;
    a.S = S;
    a.fixed_weights = (flags & M355_FIXED_WEIGHTS) ? 1 : 0;
    a.taps_from_sigma = (flags & M355_TAPS_FROM_SIGMA) ? 1 : 0;
    a.true_gaussian = (flags & M355_TRUE_GAUSSIAN) ? 1 : 0;
    return m355::launch_render<false>(a, B, (hipStream_t)stream);
}

extern "C" int m355_proj_render_bwd(const int32_t *tile_start, const float *tile_pts, const float *scale,
                                    const float *taps, int ntaps, const float *dproj, float gmul, float *dcam_slots,
                                    float *dscale_part, int B, int N, int S, int flags, void *stream)
{
    M355_REQUIRE(tile_start && ((tile_pts && dcam_slots) || N == 0) && taps && dproj, "proj_render_bwd: null pointer");
    M355_REQUIRE((scale == nullptr) == (dscale_part == nullptr), "proj_render_bwd: scale/dscale_part mismatch");
    M355_REQUIRE(B >= 0 && N >= 0 && S >= 2, "proj_render_bwd: bad size B=%d N=%d S=%d", B, N, S);
    M355_REQUIRE(ntaps >= 1 && ntaps <= m355::kMaxTaps && (ntaps & 1), "proj_render_bwd: ntaps=%d must be odd and <= %d",
                 ntaps, m355::kMaxTaps);
    M355_REQUIRE(B <= 65535, "proj_render_bwd: B=%d exceeds grid.y", B);
    if (B == 0) return M355_OK;
    if (ntaps == 21 && !(flags & M355_TAPS_FROM_SIGMA)) {
        m355::Render21Args r = {};
        r.tile_start = tile_start;
        r.tile_pts = tile_pts;
        r.scale = scale;
        r.taps = taps;
        r.dproj = dproj;
        r.gmul = gmul;
        r.dcam_slots = dcam_slots;
        r.dscale_part = dscale_part;
        r.N = N;
        r.S = S;
        r.fixed_weights = (flags & M355_FIXED_WEIGHTS) ? 1 : 0;
        r.det_scale = (flags & M355_DET_SPLAT) ? 1.0f : 0.0f;
        return m355::launch_render21<true>(r, B, (hipStream_t)stream);
    }
    m355::RenderArgs a = {};
    a.tile_start = tile_start;
    a.tile_pts = tile_pts;
    a.scale = scale;
    a.taps = taps;
    a.ntaps = ntaps;
    a.dproj = dproj;
    a.gmul = gmul;
    a.dcam_slots = dcam_slots;
    a.dscale_part = dscale_part;
    a.N = N;
    a.S = S;
    a.fixed_weights = (flags & M355_FIXED_WEIGHTS) ? 1 : 0;
    a.taps_from_sigma = (flags & M355_TAPS_FROM_SIGMA) ? 1 : 0;
    a.true_gaussian = (flags & M355_TRUE_GAUSSIAN) ? 1 : 0;
    return m355::launch_render<true>(a, B, (hipStream_t)stream);
}
