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

__global__ __launch_bounds__(256) void k_chamfer_nn(const float *__restrict__ a, const float *__restrict__ b,
                                                    float *__restrict__ dist, int32_t *__restrict__ idx, int N, int M)
{
    __shared__ float tb[kTile * 3];
    __shared__ float red_d[4][64];
    __shared__ int red_i[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bi = blockIdx.y;
    const int i = blockIdx.x * 64 + lane;
    const float *ab = a + (size_t)bi * N * 3, *bb = b + (size_t)bi * M * 3;
    const bool live = i < N;
    const float ax = live ? ab[3 * i] : 0.f, ay = live ? ab[3 * i + 1] : 0.f, az = live ? ab[3 * i + 2] : 0.f;
    float best = INFINITY;
    int besti = -1;
    for (int j0 = 0; j0 < M; j0 += kTile) {
        const int cnt = min(kTile, M - j0);
        __syncthreads();
        for (int t = tid; t < cnt * 3; t += 256) tb[t] = bb[(size_t)j0 * 3 + t];
        __syncthreads();
        // wave w sweeps points [w*256, w*256+256) of the tile, in index order (ties keep the lowest j)
        const int lo = wave * (kTile / 4), hi = min(cnt, lo + kTile / 4);
        for (int j = lo; j < hi; ++j) {
            const float dx = ax - tb[3 * j], dy = ay - tb[3 * j + 1], dz = az - tb[3 * j + 2];
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < best) {
                best = d;
                besti = j0 + j;
            }
        }
    }
    red_d[wave][lane] = best;
    red_i[wave][lane] = besti;
    __syncthreads();
    if (wave == 0 && live) {
        // merge the 4 quarter-sweeps: lower distance wins, equal distance -> lower index
        float d0 = red_d[0][lane];
        int i0 = red_i[0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float d1 = red_d[w][lane];
            const int i1 = red_i[w][lane];
            if (i1 >= 0 && (d1 < d0 || (d1 == d0 && i1 < i0) || i0 < 0)) {
                d0 = d1;
                i0 = i1;
            }
        }
        dist[(size_t)bi * N + i] = d0;
        idx[(size_t)bi * N + i] = i0;
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
    hipLaunchKernelGGL(m355::k_chamfer_nn, dim3((N + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, a, b, dist, idx, N, M);
    return m355::check_launch("chamfer_nn_fwd");
}
