// Chamfer nearest neighbour (BASELINE configs[4]; NEW capability: the reference has no Chamfer code, SURVEY 0.3,
// so there is no reference oracle -- parity is against the brute-force restatement oracle/p_oracle.c:orc_chamfer_nn).
//   dist[b,i] = min_j |a[b,i] - b[b,j]|^2,  idx[b,i] = argmin (lowest j on ties)
// fp32-VALU bound (8 flops per pair, inputs are 2 x 196 KB per cloud).  A workgroup owns 64 * QPL query points (QPL per
// lane); its NW waves sweep disjoint slices of every LDS-staged tile of 1024 target points.  Round 2 (the first version
// issued 11 VALU + 3 LDS instructions per pair and ran at 6-18 % of the fp32 vector peak):
//   * targets are staged as x[1024] | y[1024] | z[1024]: ONE ds_read_b128 broadcasts a coordinate of FOUR targets;
//   * the distance arithmetic runs two targets at a time on the packed fp32 pipe (v_pk_add / v_pk_mul / v_pk_fma_f32: each
//     lane of a packed op rounds on its own; the file is compiled with -ffp-contract=off, the two fused operations are written
//     out): every distance has the oracle's bits, fma(dz, dz, fma(dy, dy, dx*dx)) -- round 4: 3 + 3 packed operations per two
//     pairs instead of 3 + 5, i.e. 3.5 instead of 4.5 VALU instructions per pair;
//   * the sweep only tracks the running MINIMUM (one v_min3_f32 per two pairs) and, per 64-target chunk, whether it
//     improved; the index is recovered afterwards by re-scanning that one chunk for the first target attaining the
//     minimum (lowest j on ties, as the oracle): 3.5 VALU + 0.19 LDS instructions per pair (round 1: 11 + 3);
//   * small problems (B = 1) get 16 waves per workgroup over 64 queries instead of a quarter-filled chip.
#include <stdlib.h>
#include "common.h"

namespace m355 {

constexpr int kTile = 1024;
typedef float f2 __attribute__((ext_vector_type(2)));

// squared distance, DEFINED (round 4) as the fused chain fma(dz, dz, fma(dy, dy, dx * dx)) -- what the oracle evaluates too
// (oracle/p_oracle.c orc_chamfer_nn): 6 instead of 8 vector operations per pair, one rounding less
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// SPLIT (round 3, small batches): gridDim.z workgroups share one block of queries, each sweeping a contiguous slice of the
// targets [z * mslice, (z+1) * mslice) (whole LDS tiles).  Every slice stores its key (distance bits << 32 | chunk start) for
// every query into ITS OWN row keys[z][B*N] (all-ones when it found nothing); k_chamfer_unpack takes the minimum over the rows:
// distances are >= 0, so their float bits order like the values, and among equal distances the lowest chunk wins -- exactly the
// oracle's tie rule.  (Round 3 merged with a 64-bit atomicMin into one row pre-filled by a memset: one launch and B*N*slices
// atomics more.)
template <int QPL, int NW, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64) void k_chamfer_nn(const float *__restrict__ a, const float *__restrict__ b,
                                                       float *__restrict__ dist, int32_t *__restrict__ idx, int N, int M,
                                                       unsigned long long *__restrict__ keys = nullptr, int mslice = 0)
{
    constexpr int SL = kTile / NW;            // targets of a tile one wave sweeps (a multiple of 64)
    static_assert(SL % 64 == 0, "wave slices are whole chunks");
    __shared__ __attribute__((aligned(16))) float tb[3][kTile];
    __shared__ float red_d[NW][64 * QPL];
    __shared__ int red_i[NW][64 * QPL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bi = blockIdx.y;
    const float *ab = a + (size_t)bi * N * 3, *bb = b + (size_t)bi * M * 3;
    float ax[QPL], ay[QPL], az[QPL], best[QPL];
    int cstart[QPL];                          // first target of the chunk in which `best` last improved
#pragma unroll
    for (int k = 0; k < QPL; ++k) {
        const int i = (blockIdx.x * QPL + k) * 64 + lane;
        const bool live = i < N;
        ax[k] = live ? ab[3 * i] : 0.f;
        ay[k] = live ? ab[3 * i + 1] : 0.f;
        az[k] = live ? ab[3 * i + 2] : 0.f;
        best[k] = INFINITY;
        cstart[k] = -1;
    }
    const int jbeg = SPLIT ? (int)blockIdx.z * mslice : 0, jend = SPLIT ? min(M, jbeg + mslice) : M;
    for (int j0 = jbeg; j0 < jend; j0 += kTile) {
        const int cnt = min(kTile, jend - j0);
        __syncthreads();
        for (int t = tid; t < kTile; t += NW * 64) {   // structure of arrays; the tail is padded with far-away points (d = inf)
            const bool in = t < cnt;
            tb[0][t] = in ? bb[(size_t)(j0 + t) * 3] : 1e30f;
            tb[1][t] = in ? bb[(size_t)(j0 + t) * 3 + 1] : 1e30f;
            tb[2][t] = in ? bb[(size_t)(j0 + t) * 3 + 2] : 1e30f;
        }
        __syncthreads();
        const int lo = wave * SL;
        if (lo < cnt) {
            for (int c = 0; c < SL; c += 64) {   // one chunk: 16 groups of 4 targets
                float prev[QPL];
#pragma unroll
                for (int k = 0; k < QPL; ++k) prev[k] = best[k];
#pragma unroll 4
                for (int g = 0; g < 64; g += 4) {
                    const float4 bx = *reinterpret_cast<const float4 *>(&tb[0][lo + c + g]);
                    const float4 by = *reinterpret_cast<const float4 *>(&tb[1][lo + c + g]);
                    const float4 bz = *reinterpret_cast<const float4 *>(&tb[2][lo + c + g]);
                    const f2 bx01 = {bx.x, bx.y}, bx23 = {bx.z, bx.w}, by01 = {by.x, by.y}, by23 = {by.z, by.w},
                             bz01 = {bz.x, bz.y}, bz23 = {bz.z, bz.w};
#pragma unroll
                    for (int k = 0; k < QPL; ++k) {
                        const f2 qx = {ax[k], ax[k]}, qy = {ay[k], ay[k]}, qz = {az[k], az[k]};
                        const f2 dx0 = qx - bx01, dy0 = qy - by01, dz0 = qz - bz01;
                        const f2 dx1 = qx - bx23, dy1 = qy - by23, dz1 = qz - bz23;
                        const f2 d0 = __builtin_elementwise_fma(dz0, dz0, __builtin_elementwise_fma(dy0, dy0, dx0 * dx0));   // v_pk_fma_f32
                        const f2 d1 = __builtin_elementwise_fma(dz1, dz1, __builtin_elementwise_fma(dy1, dy1, dx1 * dx1));
                        best[k] = fminf(fminf(best[k], d0.x), d0.y);
                        best[k] = fminf(fminf(best[k], d1.x), d1.y);
                    }
                }
#pragma unroll
                for (int k = 0; k < QPL; ++k)
                    if (best[k] < prev[k]) cstart[k] = j0 + lo + c;
            }
        }
    }
    // ---- recover the index: the first target of the recorded chunk that attains the minimum (same arithmetic).  SPLIT: the
    // slices are merged on (distance, chunk start) -- among equal distances the lowest chunk holds the lowest index -- and the
    // ONE re-scan per query happens in k_chamfer_unpack, not once per slice
    int besti[QPL];
#pragma unroll
    for (int k = 0; k < QPL; ++k) {
        besti[k] = SPLIT ? cstart[k] : -1;
        if (!SPLIT && cstart[k] >= 0) {
            const int hi = min(cstart[k] + 64, M);
            for (int j = cstart[k]; j < hi; ++j) {
                const float d = dist2(ax[k], ay[k], az[k], bb[(size_t)j * 3], bb[(size_t)j * 3 + 1], bb[(size_t)j * 3 + 2]);
                if (d == best[k]) {
                    besti[k] = j;
                    break;
                }
            }
        }
        red_d[wave][k * 64 + lane] = best[k];
        red_i[wave][k * 64 + lane] = besti[k];
    }
    __syncthreads();
    if (wave == 0) {
        // merge the NW slice sweeps: lower distance wins, equal distance -> lower index
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            const int i = (blockIdx.x * QPL + k) * 64 + lane;
            float d0 = red_d[0][k * 64 + lane];
            int i0 = red_i[0][k * 64 + lane];
            for (int w = 1; w < NW; ++w) {
                const float d1 = red_d[w][k * 64 + lane];
                const int i1 = red_i[w][k * 64 + lane];
                if (i1 >= 0 && (d1 < d0 || (d1 == d0 && i1 < i0) || i0 < 0)) {
                    d0 = d1;
                    i0 = i1;
                }
            }
            if (i < N) {
                if (SPLIT) {
                    keys[((size_t)blockIdx.z * gridDim.y + bi) * N + i] =
                        i0 >= 0 ? (((unsigned long long)__float_as_uint(d0) << 32) | (unsigned)i0) : ~0ull;
                } else {
                    dist[(size_t)bi * N + i] = d0;
                    idx[(size_t)bi * N + i] = i0;
                }
            }
        }
    }
}

// key = (distance bits << 32 | start of the winning 64-target chunk), one row per slice -> the minimum over the rows -> distance,
// and the first target of that chunk attaining it
__global__ __launch_bounds__(256) void k_chamfer_unpack(const unsigned long long *__restrict__ keys, int nz, const float *__restrict__ a,
                                                        const float *__restrict__ b, float *__restrict__ dist,
                                                        int32_t *__restrict__ idx, int N, int M)
{
    const int i = blockIdx.x * 256 + threadIdx.x, bi = blockIdx.y;
    if (i >= N) return;
    unsigned long long k = ~0ull;
    for (int z = 0; z < nz; ++z) {
        const unsigned long long kz = keys[((size_t)z * gridDim.y + bi) * N + i];
        k = kz < k ? kz : k;
    }
    if (k == ~0ull) {   // no slice found a finite distance (NaN / overflowing input): what the one-pass form returns, no rescan
        dist[(size_t)bi * N + i] = INFINITY;
        idx[(size_t)bi * N + i] = -1;
        return;
    }
    const float d = __uint_as_float((unsigned)(k >> 32));
    const int cs = (int)(unsigned)(k & 0xffffffffull);
    const float *q = a + ((size_t)bi * N + i) * 3, *bb = b + (size_t)bi * M * 3;
    const float ax = q[0], ay = q[1], az = q[2];
    int best = -1;
    const int hi = min(cs + 64, M);
    for (int j = cs; j < hi; ++j)
        if (dist2(ax, ay, az, bb[(size_t)j * 3], bb[(size_t)j * 3 + 1], bb[(size_t)j * 3 + 2]) == d) {
            best = j;
            break;
        }
    dist[(size_t)bi * N + i] = d;
    idx[(size_t)bi * N + i] = best;
}

// target slices per query block for the split form (0: the one-pass form fills the chip already)
static int chamfer_slices(int B, int N, int M)
{
    // Measured at N = M = 16384 (scripts/chamfer_rate.py, TFLOP/s of 8 B N M, host clock over three launches): one pass 26 / 33 /
    // 43.5 at B = 1 / 4 / 8; sliced so that the launch has ~2048 workgroups, at most 8 slices: 32.8 / 40.6 / 45.5 / 49.8 / 52.3 at
    // B = 1 / 2 / 4 / 8 / 16 -- a slice's targets stay in L1 and the tail of the launch is shorter.  More than 8 slices gain
    // nothing (every slice pays a merge and 256 atomics per block, the unpack rescans one chunk per query).
    static const long target = getenv("M355_CHAMFER_WGS") ? atol(getenv("M355_CHAMFER_WGS")) : 2048;
    // (round 4: with per-slice key rows a slice costs a plain store per query, not an atomic: 16 slices measure 1-2 % ahead of 8)
    static const long max_ts = getenv("M355_CHAMFER_MAXTS") ? atol(getenv("M355_CHAMFER_MAXTS")) : 16;
    const long blocks = (long)B * ((N + 255) / 256);   // 4 queries per lane: the most reuse of a broadcast target
    if (blocks >= target || M < 2 * kTile) return 0;
    long ts = (target + blocks - 1) / blocks;
    if (ts > max_ts) ts = max_ts;
    const long tiles = (M + kTile - 1) / kTile;
    if (ts > tiles) ts = tiles;
    while (ts & (ts - 1)) ts &= ts - 1;   // a power of two: equal slices of the (usually power-of-two) tile count -- 3 or 6 slices measured
                                          // 20 % behind 4 (the last slice's workgroups finish alone)
    return ts >= 2 ? (int)ts : 0;
}

}  // namespace m355

extern "C" size_t m355_chamfer_nn_ws_bytes(int B, int N, int M)
{
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    return (size_t)m355::chamfer_slices(B, N, M) * B * N * sizeof(unsigned long long);   // one key row per slice
}

extern "C" int m355_chamfer_nn_fwd_ws(const float *a, const float *b, float *dist, int32_t *idx, int B, int N, int M, void *ws,
                                      void *stream)
{
    M355_REQUIRE(B >= 0 && N >= 0 && M >= 1, "chamfer_nn_fwd_ws: bad size B=%d N=%d M=%d", B, N, M);
    if (B == 0 || N == 0) return M355_OK;
    const int ts = m355::chamfer_slices(B, N, M);
    if (!ts || !ws) return m355_chamfer_nn_fwd(a, b, dist, idx, B, N, M, stream);
    M355_REQUIRE(a && b && dist && idx, "chamfer_nn_fwd_ws: null pointer");
    M355_REQUIRE(B <= 65535, "chamfer_nn_fwd_ws: B=%d exceeds grid.y", B);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *keys = (unsigned long long *)ws;
    const int tiles = (M + m355::kTile - 1) / m355::kTile;
    const int mslice = ((tiles + ts - 1) / ts) * m355::kTile;          // whole tiles per slice
    const int nz = (M + mslice - 1) / mslice;
    hipLaunchKernelGGL((m355::k_chamfer_nn<4, 4, true>), dim3((N + 255) / 256, B, nz), dim3(256), 0, st, a, b, dist, idx, N, M, keys, mslice);
    hipLaunchKernelGGL(m355::k_chamfer_unpack, dim3((N + 255) / 256, B), dim3(256), 0, st, keys, nz, a, b, dist, idx, N, M);
    return m355::check_launch("chamfer_nn_fwd_ws");
}

extern "C" int m355_chamfer_nn_fwd(const float *a, const float *b, float *dist, int32_t *idx, int B, int N, int M,
                                   void *stream)
{
    M355_REQUIRE(B >= 0 && N >= 0 && M >= 1, "chamfer_nn_fwd: bad size B=%d N=%d M=%d", B, N, M);
    if (B == 0 || N == 0) return M355_OK;
    M355_REQUIRE(a && b && dist && idx, "chamfer_nn_fwd: null pointer");
    M355_REQUIRE(B <= 65535, "chamfer_nn_fwd: B=%d exceeds grid.y", B);
    hipStream_t st = (hipStream_t)stream;
    // most queries per lane (= most reuse of a broadcast target) that still gives every CU a workgroup; the smaller shapes
    // make up for it with more waves per workgroup over the same target tile
    if ((long)B * ((N + 255) / 256) >= 256)
        hipLaunchKernelGGL((m355::k_chamfer_nn<4, 4>), dim3((N + 255) / 256, B), dim3(256), 0, st, a, b, dist, idx, N, M);
    else if ((long)B * ((N + 127) / 128) >= 256)
        hipLaunchKernelGGL((m355::k_chamfer_nn<2, 8>), dim3((N + 127) / 128, B), dim3(512), 0, st, a, b, dist, idx, N, M);
    else
        hipLaunchKernelGGL((m355::k_chamfer_nn<1, 16>), dim3((N + 63) / 64, B), dim3(1024), 0, st, a, b, dist, idx, N, M);
    return m355::check_launch("chamfer_nn_fwd");
}
