// Chamfer nearest neighbour (BASELINE configs[4]; NEW capability: the reference has no Chamfer code, SURVEY 0.3,
// so there is no reference oracle -- parity is against the brute-force restatement oracle/p_oracle.c:orc_chamfer_nn).
//   dist[b,i] = min_j |a[b,i] - b[b,j]|^2,  idx[b,i] = argmin (lowest j on ties)
// fp32-VALU bound (8 flops per pair, inputs are 2 x 196 KB per cloud): a workgroup owns 64 query points; its 4 waves
// sweep disjoint quarters of every LDS-staged tile of 1024 target points (each lane reads the same target point: an
// LDS broadcast), keep a running (min, argmin) per lane, and the 4 partial results are merged through LDS.
// The squared distance is evaluated exactly as the oracle does (dx*dx + dy*dy + dz*dz, left to right, no FMA:
// this file is compiled with -ffp-contract=off) so that distances and tie-breaking are bit-identical.
#include "common.h"

namespace m355 {

constexpr int kTile = 1024;

// QPL query points per lane: every LDS broadcast of a target point is reused for QPL distance evaluations (the
// single-query version spent a third of its issue slots on the three ds_reads per pair).
template <int QPL>
__global__ __launch_bounds__(256) void k_chamfer_nn(const float *__restrict__ a, const float *__restrict__ b,
                                                    float *__restrict__ dist, int32_t *__restrict__ idx, int N, int M)
{
    __shared__ float tb[kTile * 3];
    __shared__ float red_d[4][64 * QPL];
    __shared__ int red_i[4][64 * QPL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bi = blockIdx.y;
    const float *ab = a + (size_t)bi * N * 3, *bb = b + (size_t)bi * M * 3;
    float ax[QPL], ay[QPL], az[QPL], best[QPL];
    int besti[QPL];
#pragma unroll
    for (int k = 0; k < QPL; ++k) {
        const int i = (blockIdx.x * QPL + k) * 64 + lane;
        const bool live = i < N;
        ax[k] = live ? ab[3 * i] : 0.f;
        ay[k] = live ? ab[3 * i + 1] : 0.f;
        az[k] = live ? ab[3 * i + 2] : 0.f;
        best[k] = INFINITY;
        besti[k] = -1;
    }
    for (int j0 = 0; j0 < M; j0 += kTile) {
        const int cnt = min(kTile, M - j0);
        __syncthreads();
        for (int t = tid; t < cnt * 3; t += 256) tb[t] = bb[(size_t)j0 * 3 + t];
        __syncthreads();
        // wave w sweeps points [w*256, w*256+256) of the tile, in index order (ties keep the lowest j)
        const int lo = wave * (kTile / 4), hi = min(cnt, lo + kTile / 4);
#pragma unroll 2
        for (int j = lo; j < hi; ++j) {
            const float bx = tb[3 * j], by = tb[3 * j + 1], bz = tb[3 * j + 2];
#pragma unroll
            for (int k = 0; k < QPL; ++k) {
                const float dx = ax[k] - bx, dy = ay[k] - by, dz = az[k] - bz;
                const float d = dx * dx + dy * dy + dz * dz;
                if (d < best[k]) {
                    best[k] = d;
                    besti[k] = j0 + j;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < QPL; ++k) {
        red_d[wave][k * 64 + lane] = best[k];
        red_i[wave][k * 64 + lane] = besti[k];
    }
    __syncthreads();
    if (wave == 0) {
        // merge the 4 quarter-sweeps: lower distance wins, equal distance -> lower index
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            const int i = (blockIdx.x * QPL + k) * 64 + lane;
            float d0 = red_d[0][k * 64 + lane];
            int i0 = red_i[0][k * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float d1 = red_d[w][k * 64 + lane];
                const int i1 = red_i[w][k * 64 + lane];
                if (i1 >= 0 && (d1 < d0 || (d1 == d0 && i1 < i0) || i0 < 0)) {
                    d0 = d1;
                    i0 = i1;
                }
            }
            if (i < N) {
                dist[(size_t)bi * N + i] = d0;
                idx[(size_t)bi * N + i] = i0;
            }
        }
    }
}

}  // namespace m355

extern "C" int m355_chamfer_nn_fwd(const float *a, const float *b, float *dist, int32_t *idx, int B, int N, int M,
                                   void *stream)
{
    M355_REQUIRE(B >= 0 && N >= 0 && M >= 1, "chamfer_nn_fwd: bad size B=%d N=%d M=%d", B, N, M);
    if (B == 0 || N == 0) return M355_OK;
    M355_REQUIRE(a && b && dist && idx, "chamfer_nn_fwd: null pointer");
    M355_REQUIRE(B <= 65535, "chamfer_nn_fwd: B=%d exceeds grid.y", B);
    // 4 queries per lane once that still leaves >= 1 workgroup per CU
    if ((long)B * ((N + 255) / 256) >= 256)
        hipLaunchKernelGGL(m355::k_chamfer_nn<4>, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, a, b, dist, idx, N, M);
    else
        hipLaunchKernelGGL(m355::k_chamfer_nn<1>, dim3((N + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, a, b, dist, idx, N, M);
    return m355::check_launch("chamfer_nn_fwd");
}
