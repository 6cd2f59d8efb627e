// G9 / G3 glue of the GAN stacks: everything that sits BETWEEN the heavy kernels and used to be hundreds of 3-5 us
// torch launches per forward (measured: 3400 launches = 19.7 ms of a 112 ms cycle).
//   * spectral norm (torch.nn.utils.spectral_norm on 18 generator + 9 discriminator convs, gan.py:57-65,163-177,
//     294-302): one power iteration per forward in train mode,  v = normalize(W^T u), u = normalize(W v),
//     sigma = u . (W v),  W_sn = W / sigma.   Here ALL layers of a network advance in three launches
//     (grid.y = layer) driven by a device-resident layer table; the division by sigma is folded into the bf16
//     weight-view kernel and the backward through sigma into the kernel that lays out the weight gradient.
//   * conditional batch norm (gan.py:264-286): from the per-workgroup partial sums to mean / rstd / running stats /
//     the per-(sample, channel) affine coefficients in one launch; likewise the backward's coefficient algebra.
#include "common.h"

namespace m355 {

// ------------------------------------------------------------------------------------------- spectral norm, forward
// Three launches per training forward (two in eval mode), no atomics on floats: every cross-workgroup sum is a per-workgroup
// partial in `scratch` added in a FIXED order by the next launch, so sigma / u / v have the same bits on every run (the first
// version accumulated the two norms with fp32 atomics: run-to-run differences in the last bit of every weight of the network).
//   scratch = [L][T1] partials of |W^T u|^2 per 64-column tile | [L][T2] partials per 4-row block;
//   T1 = ceil(max_cols / 64), T2 = ceil(max_rows / 4).
// (Tried in round 4: the third launch folded into the second behind a per-layer ticket -- the last row block to finish does the
// final step.  The release / acquire fences that makes correct across the 8 XCDs' L2s are a full `buffer_wbl2` + `buffer_inv`
// per workgroup: 65 -> 87 us per call.  Kernel boundaries are the cheaper fence.)
__device__ __forceinline__ float wave_sum_fixed(float v)   // butterfly: the same additions in the same order in every wave
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float ordered_sum(const float *p, int n)   // by one full wave; identical wherever it is evaluated
{
    const int lane = threadIdx.x & 63;
    float s = 0.0f;
    for (int j = lane; j < n; j += 64) s += p[j];
    return wave_sum_fixed(s);
}

// t = W^T u   (train only).  Block = 64 columns x 16 row groups, 4 independent loads in flight per thread (the
// 4-group version walked 128 rows of the 512-row layers one dependent load at a time).
__global__ __launch_bounds__(1024) void k_sn_wtu(const m355_sn_layer *__restrict__ tab, float *__restrict__ p1, int T1)
{
    __shared__ float red[16][64];
    const m355_sn_layer L = tab[blockIdx.y];
    const int c0 = blockIdx.x * 64;
    if (c0 >= L.cols) return;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = c0 + cl;
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < L.cols)
        for (int i = rg; i < L.rows; i += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + 16 * u < L.rows) a4[u] += L.w[(size_t)(i + 16 * u) * L.cols + c] * L.u[i + 16 * u];
        }
    red[rg][cl] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    __syncthreads();
    if (rg == 0) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        if (c < L.cols) L.t[c] = t;
        const float sq = wave_sum_fixed(c < L.cols ? t * t : 0.0f);
        if (cl == 0) p1[(size_t)blockIdx.y * T1 + blockIdx.x] = sq;
    }
}

// s = W vhat, one wave per row (4 rows per block).  train: vhat = t / max(||t||, eps) (and v <- vhat, written by the
// layer's first block); eval: vhat = v.   p2[l][block] = sum over the block's rows of s_i^2 (train) or u_i * s_i (eval).
__global__ __launch_bounds__(256) void k_sn_wv(const m355_sn_layer *__restrict__ tab, const float *__restrict__ p1, int T1,
                                               float *__restrict__ p2, int T2, int training, float eps)
{
    const int l = blockIdx.y;
    const m355_sn_layer L = tab[l];
    const int r0 = blockIdx.x * 4;
    if (r0 >= L.rows) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = r0 + wave;
    float inv = 1.0f;
    const float *vv = L.v;
    if (training) {
        inv = 1.0f / fmaxf(sqrtf(ordered_sum(p1 + (size_t)l * T1, (L.cols + 63) / 64)), eps);
        vv = L.t;
    }
    float acc = 0.0f;
    if (i < L.rows) {
        const float *wr = L.w + (size_t)i * L.cols;
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
        if ((L.cols & 3) == 0 && ((size_t)wr & 15) == 0 && ((size_t)vv & 15) == 0) {
            // 16-byte loads, four of them in flight per lane: a 4608-column row (18 KB) is two round trips of the wave instead
            // of eighteen (one wave per row is latency bound: this kernel was 54 us per call)
            const float4 *w4 = reinterpret_cast<const float4 *>(wr), *v4 = reinterpret_cast<const float4 *>(vv);
            const int n4 = L.cols >> 2;
            for (int j = lane; j < n4; j += 256) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j + 64 * u < n4) {
                        const float4 a = w4[j + 64 * u], b = v4[j + 64 * u];
                        a4[u] += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
                    }
            }
        } else {
            for (int j = lane; j < L.cols; j += 256) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j + 64 * u < L.cols) a4[u] += wr[j + 64 * u] * vv[j + 64 * u];
            }
        }
        acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    acc = wave_sum_fixed(acc) * inv;
    __shared__ float wsum[4];
    if (lane == 0) {
        float contrib = 0.0f;
        if (i < L.rows) {
            L.s[i] = acc;
            contrib = training ? acc * acc : acc * L.u[i];
        }
        wsum[wave] = contrib;
    }
    if (training && blockIdx.x == 0)
        for (int j = threadIdx.x; j < L.cols; j += 256) {
            const float vn = L.t[j] * inv;
            L.v[j] = vn;
            if (L.v_snap) L.v_snap[j] = vn;
        }
    __syncthreads();
    if (threadIdx.x == 0) p2[(size_t)l * T2 + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// train: u = s / max(||s||, eps), sigma = ||s||^2 / max(||s||, eps);  eval: sigma = u . s.   ||s||^2 (u . s) = the layer's
// row-block partials added in block order.
__global__ __launch_bounds__(256) void k_sn_final(const m355_sn_layer *__restrict__ tab, const float *__restrict__ p2, int T2,
                                                  float *__restrict__ sigma, int training, float eps)
{
    const int l = blockIdx.x;
    const m355_sn_layer L = tab[l];
    const float n1 = ordered_sum(p2 + (size_t)l * T2, (L.rows + 3) / 4);
    if (training) {
        const float inv = 1.0f / fmaxf(sqrtf(n1), eps);
        for (int i = threadIdx.x; i < L.rows; i += 256) {
            const float un = L.s[i] * inv;
            L.u[i] = un;
            if (L.u_snap) L.u_snap[i] = un;
        }
        if (threadIdx.x == 0) sigma[l] = n1 * inv;
    } else {
        if (L.u_snap)
            for (int i = threadIdx.x; i < L.rows; i += 256) L.u_snap[i] = L.u[i];
        if (L.v_snap)
            for (int j = threadIdx.x; j < L.cols; j += 256) L.v_snap[j] = L.v[j];
        if (threadIdx.x == 0) sigma[l] = n1;
    }
}

// ------------------------------------------------------------------------------------------- spectral norm, backward
// G = dL/dW_sn arrives from the wgrad kernel as [Cout][kh][kw][CinP]; the parameter is [Cout][Cin][kh][kw].
//   dL/dW_orig = G / sigma - (<G, W_orig> / sigma^2) u v^T         (u, v constants of the power iteration)
// stage 1: partial dot products; stage 2: every block sums the partials, then applies and re-lays-out.
__global__ __launch_bounds__(256) void k_sn_bwd_dot(const float *__restrict__ g, const float *__restrict__ w, float *__restrict__ part,
                                                    int Cout, int Cin, int CinP, int KK)
{
    __shared__ float red[4];
    const size_t total = (size_t)Cout * Cin * KK;
    float acc = 0.0f;
    for (size_t idx = blockIdx.x * (size_t)256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        // idx enumerates the parameter layout [co][ci][tap]
        const int tap = (int)(idx % KK);
        const size_t r = idx / KK;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        acc += w[idx] * g[((size_t)co * KK + tap) * CinP + ci];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_sn_bwd_apply(const float *__restrict__ g, const float *__restrict__ u,
                                                      const float *__restrict__ v, const float *__restrict__ sigma,
                                                      const float *__restrict__ part, int npart, float *__restrict__ dw, int Cout,
                                                      int Cin, int CinP, int KK)
{
    __shared__ float sh;
    float inv = 1.0f, coef = 0.0f;
    if (sigma) {
        if (threadIdx.x < 64) {
            float d = 0.0f;
            for (int i = threadIdx.x; i < npart; i += 64) d += part[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
            if (threadIdx.x == 0) sh = d;
        }
        __syncthreads();
        inv = 1.0f / sigma[0];
        coef = sh * inv * inv;
    }
    const size_t total = (size_t)Cout * Cin * KK;
    for (size_t idx = blockIdx.x * (size_t)256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int tap = (int)(idx % KK);
        const size_t r = idx / KK;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        float val = g[((size_t)co * KK + tap) * CinP + ci] * inv;
        if (sigma) val -= coef * u[co] * v[(size_t)ci * KK + tap];
        dw[idx] = val;
    }
}

// ------------------------------------------------------------------------------------------- conditional batch norm
// part[nblk][2][C] -> mean, rstd (biased variance, eps inside the sqrt =
// F.batch_norm, code/sync_batchnorm/batchnorm.py:71-73), running stats (unbiased variance, momentum), and
// a[n,c] = rstd * (1 + gamma[n,c]),  b[n,c] = beta[n,c] - mean * a[n,c].
// CW channels per workgroup, 1024 / CW row lanes: with 32 channels per workgroup a C = 64 layer ran on TWO workgroups, each
// thread walking 32+ dependent-latency rounds of the up-to-1024 partial rows (~10 us per call, 42 calls per cycle); 8
// channels per workgroup give 4x the workgroups and a quarter of the rows per thread.
// accumulator type of the batch-norm finalising kernels: fp32 in the product build, fp64 in the EXACT build (csrc/gan_elem.hip acc_t)
#ifdef M355_EXACT
typedef double gacc_t;
#else
typedef float gacc_t;
#endif
template <int CW>
__global__ __launch_bounds__(1024) void k_bn_finalize(const float *__restrict__ part, int nblk, float count_h,
                                                     const float *__restrict__ count_dev, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, int gstride, int N, int C, float eps,
                                                     float momentum, float *__restrict__ rmean, float *__restrict__ rvar,
                                                     float *__restrict__ mean_o, float *__restrict__ rstd_o, float *__restrict__ a,
                                                     float *__restrict__ b)
{
    constexpr int RL = 1024 / CW;
    __shared__ gacc_t red[2][RL][CW];
    __shared__ float stat[2][CW];
    const int cl = threadIdx.x % CW, l = threadIdx.x / CW, c = blockIdx.x * CW + cl;
    const float count = count_dev ? *count_dev : count_h;   // SyncBN: the all-reduced pixel count, no host round trip
    gacc_t s0 = 0, s1 = 0;
    if (c < C) {
        // 8 rows (16 loads) in flight per thread: a conv with fused statistics hands over one row per workgroup (up to 2048)
        gacc_t p0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, p1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = l; k < nblk; k += 8 * RL) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k + RL * u;
                if (kk < nblk) {
                    p0[u] += part[((size_t)kk * 2) * C + c];
                    p1[u] += part[((size_t)kk * 2 + 1) * C + c];
                }
            }
        }
        s0 = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p0[4] + p0[5]) + (p0[6] + p0[7]));
        s1 = ((p1[0] + p1[1]) + (p1[2] + p1[3])) + ((p1[4] + p1[5]) + (p1[6] + p1[7]));
    }
    red[0][l][cl] = s0;
    red[1][l][cl] = s1;
    __syncthreads();
#pragma unroll
    for (int s = RL / 2; s > 0; s >>= 1) {   // fixed-order tree over the row lanes
        if (l < s) {
            red[0][l][cl] += red[0][l + s][cl];
            red[1][l][cl] += red[1][l + s][cl];
        }
        __syncthreads();
    }
    if (l == 0 && c < C) {
#ifdef M355_EXACT
        // (fp64 from the partial rows on: mean, E[x^2] - mean^2 and 1 / sqrt(var + eps) as ATen's CPU batch norm evaluates them)
        const double t0 = red[0][0][cl], t1 = red[1][0][cl];
        const double mean_d = t0 / (double)count, var_d = fmax(t1 / (double)count - mean_d * mean_d, 0.0);
        const float mean = (float)mean_d, var = (float)var_d, rstd = (float)(1.0 / sqrt(var_d + (double)eps));
#else
        const float t0 = red[0][0][cl], t1 = red[1][0][cl];
        const float mean = t0 / count;
        const float var = fmaxf(t1 / count - mean * mean, 0.0f);
        const float rstd = rsqrtf(var + eps);
#endif
        stat[0][cl] = mean;
        stat[1][cl] = rstd;
        mean_o[c] = mean;
        rstd_o[c] = rstd;
        if (rmean) {
            const float unb = count > 1.0f ? var * (count / (count - 1.0f)) : var;
            rmean[c] = rmean[c] * (1.0f - momentum) + mean * momentum;
            rvar[c] = rvar[c] * (1.0f - momentum) + unb * momentum;
        }
    }
    __syncthreads();
    if (c < C) {
        const float mean = stat[0][cl], rstd = stat[1][cl];
        for (int n = l; n < N; n += RL) {
            const float av = rstd * (1.0f + gamma[(size_t)n * gstride + c]);
            a[(size_t)n * C + c] = av;
            b[(size_t)n * C + c] = beta[(size_t)n * gstride + c] - mean * av;
        }
    }
}

// vec[0 .. C2) = column sums of part[nblk][C2], vec[C2] = count (the forward message of SyncBN: m355_bn_sync_pack)
__global__ __launch_bounds__(256) void k_bn_sync_pack(const float *__restrict__ part, int nblk, int C2, float count,
                                                      float *__restrict__ vec)
{
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C2)
        for (int k = rl; k < nblk; k += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k + 4 * u < nblk) a4[u] += part[(size_t)(k + 4 * u) * C2 + c];
        }
    red[rl][cl] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    __syncthreads();
    if (rl == 0 && c < C2) vec[c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    if (blockIdx.x == 0 && threadIdx.x == 0) vec[C2] = count;
}

// backward: part[N][nblk][2][C] = per-workgroup (sum dz, sum dz*x)  ->
//   dgamma[n,c] = rstd (s2 - mean s1),  dbeta[n,c] = s1,  A[n,c] = rstd (1+gamma),
//   m1 = sum_n (1+gamma) s1 / count,  m2 = sum_n (1+gamma) dgamma / count,
//   Bc = -rstd^2 m2,  Cc = -rstd m1 + rstd^2 mean m2            (zeros when the statistics are not batch statistics)
__global__ __launch_bounds__(1024) void k_bn_bwd_finalize(const float *__restrict__ part, int nblk, float count,
                                                         const float *__restrict__ gamma, int gstride, int N, int C,
                                                         const float *__restrict__ mean_i, const float *__restrict__ rstd_i,
                                                         int batch_stats, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                         float *__restrict__ A, float *__restrict__ Bc, float *__restrict__ Cc,
                                                         float *__restrict__ m_out)
{
    __shared__ gacc_t red[2][32][32];
    const int cl = threadIdx.x & 31, l = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    gacc_t m1 = 0, m2 = 0;
    gacc_t mean = 0, rstd = 1;
    if (c < C) {
        mean = mean_i[c];
        rstd = rstd_i[c];
        for (int n = l; n < N; n += 32) {
            gacc_t q1[4] = {0, 0, 0, 0}, q2[4] = {0, 0, 0, 0};
            const float *p = part + (size_t)n * nblk * 2 * C;
            for (int k = 0; k < nblk; k += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k + u < nblk) {
                        q1[u] += p[((size_t)(k + u) * 2) * C + c];
                        q2[u] += p[((size_t)(k + u) * 2 + 1) * C + c];
                    }
            }
            const gacc_t s1 = (q1[0] + q1[1]) + (q1[2] + q1[3]), s2 = (q2[0] + q2[1]) + (q2[2] + q2[3]);
            const gacc_t sc = (gacc_t)1 + gamma[(size_t)n * gstride + c];
            const gacc_t dg = rstd * (s2 - mean * s1);
            dgamma[(size_t)n * C + c] = (float)dg;
            dbeta[(size_t)n * C + c] = (float)s1;
            A[(size_t)n * C + c] = (float)(rstd * sc);
            m1 += sc * s1;
            m2 += sc * dg;
        }
    }
    red[0][l][cl] = m1;
    red[1][l][cl] = m2;
    __syncthreads();
    if (l == 0 && c < C) {
        gacc_t t1 = 0, t2 = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            t1 += red[0][k][cl];
            t2 += red[1][k][cl];
        }
        if (m_out) {  // SyncBN: the sums over the LOCAL samples; all-reduced, then m355_bn_bwd_coeffs
            m_out[c] = (float)t1;
            m_out[C + c] = (float)t2;
        } else if (batch_stats) {
            t1 /= count;
            t2 /= count;
            Bc[c] = (float)(-rstd * rstd * t2);
            Cc[c] = (float)(-rstd * t1 + rstd * rstd * mean * t2);
        } else {
            Bc[c] = 0.0f;
            Cc[c] = 0.0f;
        }
    }
}

// Bc, Cc from the (all-reduced) moment sums m[2][C] and the global pixel count
__global__ void k_bn_bwd_coeffs(const float *__restrict__ m, float count_h, const float *__restrict__ count_dev,
                                const float *__restrict__ mean_i, const float *__restrict__ rstd_i, int C, float *__restrict__ Bc,
                                float *__restrict__ Cc)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float count = count_dev ? *count_dev : count_h;
    const float t1 = m[c] / count, t2 = m[C + c] / count, rstd = rstd_i[c], mean = mean_i[c];
    Bc[c] = -rstd * rstd * t2;
    Cc[c] = -rstd * t1 + rstd * rstd * mean * t2;
}

}  // namespace m355

using namespace m355;

/* 4-byte words of scratch m355_sn_power_iter needs for L layers of at most max_rows x max_cols */
extern "C" size_t m355_sn_scratch_words(int L, int max_rows, int max_cols)
{
    if (L <= 0 || max_rows <= 0 || max_cols <= 0) return 0;
    return (size_t)L * ((size_t)(max_cols + 63) / 64 + (size_t)(max_rows + 3) / 4);
}

extern "C" int m355_sn_power_iter(const m355_sn_layer *table_dev, int L, int max_rows, int max_cols, float *scratch,
                                  float *sigma, int training, float eps, void *stream)
{
    M355_REQUIRE(table_dev && scratch && sigma && L > 0 && max_rows > 0 && max_cols > 0, "sn_power_iter: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int T1 = (max_cols + 63) / 64, T2 = (max_rows + 3) / 4;
    float *p1 = scratch, *p2 = p1 + (size_t)L * T1;
    if (training) hipLaunchKernelGGL(k_sn_wtu, dim3(T1, L), dim3(1024), 0, st, table_dev, p1, T1);
    hipLaunchKernelGGL(k_sn_wv, dim3(T2, L), dim3(256), 0, st, table_dev, (const float *)p1, T1, p2, T2, training, eps);
    hipLaunchKernelGGL(k_sn_final, dim3(L), dim3(256), 0, st, table_dev, (const float *)p2, T2, sigma, training, eps);
    return check_launch("sn_power_iter");
}

extern "C" int m355_sn_wgrad_finish(const float *g_khwc, const float *w_orig, const float *u, const float *v,
                                    const float *sigma, float *part /*[256]*/, float *dw, int Cout, int Cin, int CinP,
                                    int kh, int kw, void *stream)
{
    M355_REQUIRE(g_khwc && dw && Cout > 0 && Cin > 0 && CinP >= Cin, "sn_wgrad_finish: bad argument");
    M355_REQUIRE(!sigma || (w_orig && u && v && part), "sn_wgrad_finish: spectral-norm state missing");
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)Cout * Cin * kh * kw;
    const int nblk = (int)((total + 255) / 256 > 256 ? 256 : (total + 255) / 256);
    if (sigma) hipLaunchKernelGGL(k_sn_bwd_dot, dim3(nblk), dim3(256), 0, st, g_khwc, w_orig, part, Cout, Cin, CinP, kh * kw);
    const int nb2 = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(k_sn_bwd_apply, dim3(nb2), dim3(256), 0, st, g_khwc, u, v, sigma, part, nblk, dw, Cout, Cin, CinP,
                       kh * kw);
    return check_launch("sn_wgrad_finish");
}

// All weight-gradient epilogues of ONE backward pass in two launches (was two launches per layer, 76 per GAN cycle: latency,
// and a re-layout whose reads were strided by CinP floats).  One workgroup per (layer, output channel): the channel's raw
// gradient g[co][tap][ci] is read ONCE per launch, coalesced, into LDS (row stride CinP + 1: conflict-free transposed reads) and
// leaves as dw[co][ci][tap] in coalesced stores; w_orig and v are read in their own (linear) order.  The entries travel as a
// kernel argument (<= 24 layers): no device table to upload, capturable.  Deterministic: <g, w_orig> = a fixed-order sum of one
// partial per output channel.
struct SnFinBatch {
    m355_snfin_entry e[M355_SNFIN_MAX];
    int row0[M355_SNFIN_MAX + 1];   // first workgroup of every layer (prefix sum of Cout)
    int L;
};
constexpr int kSnFinLds = 12832;    // floats: (CinP + 1) * kh * kw of the widest row (512 channels x 5 x 5)

__device__ __forceinline__ int snfin_layer(const SnFinBatch &b, int row)
{
    int l = 0;
    while (l + 1 < b.L && row >= b.row0[l + 1]) ++l;
    return l;
}

template <bool APPLY>
__global__ __launch_bounds__(256) void k_sn_fin(SnFinBatch b)
{
    __shared__ float gl[kSnFinLds];
    __shared__ float red[4];
    const int l = snfin_layer(b, blockIdx.x);
    const m355_snfin_entry &e = b.e[l];
    if (!APPLY && !e.sigma) return;
    const int co = blockIdx.x - b.row0[l], KK = e.kh * e.kw, n = e.Cin * KK, ld = e.CinP + 1;
    const float *g = e.g_khwc + (size_t)co * KK * e.CinP;
    for (int t = threadIdx.x; t < KK * e.CinP; t += 256) {
        const int tap = t / e.CinP, ci = t - tap * e.CinP;
        gl[tap * ld + ci] = g[t];
    }
    float inv = 1.0f, coef = 0.0f;
    if (APPLY && e.sigma) {   // <g, w_orig> = the per-channel partials of the first launch, summed in a fixed order
        float d = 0.0f;
        for (int i = threadIdx.x; i < e.Cout; i += 256) d += e.part[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    }
    __syncthreads();
    if (APPLY && e.sigma) {
        inv = 1.0f / e.sigma[0];
        coef = (red[0] + red[1] + red[2] + red[3]) * inv * inv;
    }
    if (APPLY) {
        const float uc = e.sigma ? e.u[co] : 0.0f;
        float *dw = e.dw + (size_t)co * n;
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            const int ci = idx / KK, tap = idx - ci * KK;
            float val = gl[tap * ld + ci] * inv;
            if (e.sigma) val -= coef * uc * e.v[idx];
            dw[idx] = val;
        }
    } else {
        const float *w = e.w_orig + (size_t)co * n;
        float acc = 0.0f;
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            const int ci = idx / KK, tap = idx - ci * KK;
            acc += w[idx] * gl[tap * ld + ci];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) e.part[co] = red[0] + red[1] + red[2] + red[3];
    }
}

extern "C" int m355_sn_wgrad_finish_batched(const m355_snfin_entry *entries_host, int L, void *stream)
{
    M355_REQUIRE(entries_host && L > 0 && L <= M355_SNFIN_MAX, "sn_wgrad_finish_batched: 1..%d layers per call", M355_SNFIN_MAX);
    SnFinBatch b;
    bool any_sn = false;
    int rows = 0;
    for (int i = 0; i < L; ++i) {
        const m355_snfin_entry &e = entries_host[i];
        M355_REQUIRE(e.g_khwc && e.dw && e.Cout > 0 && e.Cin > 0 && e.CinP >= e.Cin && e.kh > 0 && e.kw > 0,
                     "sn_wgrad_finish_batched: bad entry %d", i);
        M355_REQUIRE((e.CinP + 1) * e.kh * e.kw <= kSnFinLds, "sn_wgrad_finish_batched: entry %d: %d x %d x %d exceeds the LDS row", i,
                     e.CinP, e.kh, e.kw);
        M355_REQUIRE(!e.sigma || (e.w_orig && e.u && e.v && e.part), "sn_wgrad_finish_batched: spectral-norm state missing in entry %d", i);
        b.e[i] = e;
        b.row0[i] = rows;
        rows += e.Cout;
        any_sn = any_sn || e.sigma;
    }
    b.row0[L] = rows;
    b.L = L;
    hipStream_t st = (hipStream_t)stream;
    if (any_sn) hipLaunchKernelGGL(k_sn_fin<false>, dim3(rows), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_sn_fin<true>, dim3(rows), dim3(256), 0, st, b);
    return check_launch("sn_wgrad_finish_batched");
}

extern "C" int m355_bn_finalize(const float *part, int nblk, float count, const float *count_dev, const float *gamma, const float *beta, int gstride,
                                int N, int C, float eps, float momentum, float *running_mean, float *running_var,
                                float *mean, float *rstd, float *a, float *b, void *stream)
{
    M355_REQUIRE(part && gamma && beta && mean && rstd && a && b && nblk > 0 && N > 0 && C > 0, "bn_finalize: bad argument");
    hipLaunchKernelGGL(k_bn_finalize<8>, dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)stream, part, nblk, count, count_dev,
                       gamma, beta, gstride, N, C, eps, momentum, running_mean, running_var, mean, rstd, a, b);
    return check_launch("bn_finalize");
}

extern "C" int m355_bn_bwd_finalize(const float *part, int nblk, float count, const float *gamma, int gstride, int N, int C,
                                    const float *mean, const float *rstd, int batch_stats, float *dgamma, float *dbeta,
                                    float *A, float *Bc, float *Cc, float *m_out, void *stream)
{
    M355_REQUIRE(part && gamma && mean && rstd && dgamma && dbeta && A && Bc && Cc && nblk > 0 && N > 0 && C > 0,
                 "bn_bwd_finalize: bad argument");
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, part, nblk, count, gamma,
                       gstride, N, C, mean, rstd, batch_stats, dgamma, dbeta, A, Bc, Cc, m_out);
    return check_launch("bn_bwd_finalize");
}

/* SyncBN: the message of the forward all-reduce, vec[2C + 1] = [ sum over the partial rows of (sum x | sum x^2) | pixel count ],
 * in one launch (was a row reduction and a fill) */
extern "C" int m355_bn_sync_pack(const float *part, int nblk, int C, float count, float *vec, void *stream)
{
    M355_REQUIRE(part && vec && nblk > 0 && C > 0, "bn_sync_pack: bad argument");
    hipLaunchKernelGGL(k_bn_sync_pack, dim3((2 * C + 63) / 64), dim3(256), 0, (hipStream_t)stream, part, nblk, 2 * C, count, vec);
    return check_launch("bn_sync_pack");
}

extern "C" int m355_bn_bwd_coeffs(const float *m, float count, const float *count_dev, const float *mean, const float *rstd,
                                  int C, float *Bc, float *Cc, void *stream)
{
    M355_REQUIRE(m && mean && rstd && Bc && Cc && C > 0, "bn_bwd_coeffs: bad argument");
    hipLaunchKernelGGL(k_bn_bwd_coeffs, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, m, count, count_dev, mean, rstd, C,
                       Bc, Cc);
    return check_launch("bn_bwd_coeffs");
}
