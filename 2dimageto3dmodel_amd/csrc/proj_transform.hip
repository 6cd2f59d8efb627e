// P1+P2: quaternion rotation + perspective divide of a point cloud (and its backward).
//
// Replaces CameraUtilities.transformation_3d_coord_to_camera_coord
// (camera/coordinate_system_transformation.py:20-39), PointsQuaternionsRotator.rotate_points
// (quaternions/points_quaternions.py:41-81) and QuaternionOperations.quaternion_multiplication /
// quaternion_conjugate (quaternions/operations.py:68-97,120-136).
//
// Parity contract: cam[] is BIT-EXACT with the torch-CPU reference, because the projection bin of a
// point is floor((S-1)(cam+0.5)) after ~60 dependent fp32 ops.  That requires
//   * this file compiled with -ffp-contract=off (no FMA contraction), IEEE div/sqrt (hipcc default
//     -fhip-fp32-correctly-rounded-divide-sqrt), no fast-math;
//   * the reference's operation order: Hamilton product terms left to right, the norm as the
//     sequential sum ((q0^2+q1^2)+q2^2)+q3^2.
// Memory-bound and tiny (36 B/point): one thread per point, 12 B coalesced in, 12(+4) B out.
#include "common.h"
#include "tiles.h"

namespace m355 {

struct Quat {
    float w, x, y, z;
};

__device__ __forceinline__ Quat hamilton(const Quat a, const Quat b)
{
    // operations.py:82-85, left to right, each op rounded
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

__device__ __forceinline__ Quat normalize_quat(const float *q, float *norm_out)
{
    // F.normalize (points_quaternions.py:52-55): q / max(||q||, 1e-12)
    float s = q[0] * q[0];
    s = s + q[1] * q[1];
    s = s + q[2] * q[2];
    s = s + q[3] * q[3];
    float n = sqrtf(s);
    *norm_out = n;
    float d = n < 1e-12f ? 1e-12f : n;
    Quat r;
    r.w = q[0] / d;
    r.x = q[1] / d;
    r.y = q[2] / d;
    r.z = q[3] / d;
    return r;
}

// one point through P1+P2; the reference's operation order, every op rounded (see header)
__device__ __forceinline__ void transform_point(const Quat qn, const Quat qs, const float *__restrict__ p, float fov,
                                                float dist, float &z_out, float &y_out, float &x_out)
{
    Quat p4;  // points_quaternions.py:33: pad -> (0, p0, p1, p2)
    p4.w = 0.0f;
    p4.x = p[0];
    p4.y = p[1];
    p4.z = p[2];
    const Quat t = hamilton(qn, p4);  // points_quaternions.py:72-75
    const Quat r = hamilton(t, qs);
    const float z = r.x, y = r.y, x = r.z;  // cam:25  z,y,x = unbind(dim=2)
    const float den = z + dist;
    x_out = x * fov / den;  // cam:33
    y_out = y * fov / den;  // cam:34
    z_out = z;
}

__device__ __forceinline__ Quat conj_quat(const Quat qn)
{
    Quat qs;  // operations.py:131-136: q * (1,-1,-1,-1)
    qs.w = qn.w * 1.0f;
    qs.x = qn.x * -1.0f;
    qs.y = qn.y * -1.0f;
    qs.z = qn.z * -1.0f;
    return qs;
}

// P1+P2 fused with the per-tile binning the renderer consumes.  One workgroup per cloud:
//   pass 1: transform (or read) every point, count it into each tile its 2x2 ray footprint touches (LDS histogram)
//   scan  : exclusive prefix over the tiles -> tile_start[b][0..ntiles]
//   pass 2: scatter (c0,c1,c2,n) records into tile_pts[b][...] (<= 4N records)
// The order of records inside a tile is whatever the LDS atomics produce; the renderer accumulates with LDS
// atomics anyway.
template <bool XFORM>
__global__ __launch_bounds__(1024) void k_bin(const float *__restrict__ pc, const float *__restrict__ q,
                                              float *__restrict__ cam, int32_t *__restrict__ raykey,
                                              int *__restrict__ tile_start, float4 *__restrict__ tile_pts, int N, int S,
                                              int TH, int TW, int tiles_x, int ntiles, float fov, float dist)
{
    extern __shared__ int cnt[];  // ntiles + 1
    __shared__ int wave_tot[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float sm1 = (float)S - 1.0f;  // tri:34
    for (int i = tid; i <= ntiles; i += 1024) cnt[i] = 0;
    Quat qn = {1.f, 0.f, 0.f, 0.f}, qs = qn;
    if (XFORM) {
        float nrm;
        qn = normalize_quat(q + 4 * b, &nrm);
        qs = conj_quat(qn);
    }
    __syncthreads();
    float *camb = cam + (size_t)b * N * 3;
    auto tiles_of = [&](float c1, float c2, int (&t)[4]) -> int {
        const int f1 = (int)floorf(sm1 * (c1 + 0.5f)), f2 = (int)floorf(sm1 * (c2 + 0.5f));
        const int ty0 = f1 / TH, ty1 = (f1 + 1) / TH, tx0 = f2 / TW, tx1 = (f2 + 1) / TW;
        int k = 0;
        t[k++] = ty0 * tiles_x + tx0;
        if (tx1 != tx0) t[k++] = ty0 * tiles_x + tx1;
        if (ty1 != ty0) {
            t[k++] = ty1 * tiles_x + tx0;
            if (tx1 != tx0) t[k++] = ty1 * tiles_x + tx1;
        }
        return k;
    };
    for (int n = tid; n < N; n += 1024) {
        float c0, c1, c2;
        if (XFORM) {
            transform_point(qn, qs, pc + ((size_t)b * N + n) * 3, fov, dist, c0, c1, c2);
            camb[3 * n] = c0;
            camb[3 * n + 1] = c1;
            camb[3 * n + 2] = c2;
        } else {
            c0 = camb[3 * n];
            c1 = camb[3 * n + 1];
            c2 = camb[3 * n + 2];
        }
        const bool inb = in_bounds3(c0, c1, c2);
        if (raykey) {
            int32_t key = -1;
            if (inb) key = ((int)floorf(sm1 * (c1 + 0.5f)) << 16) | (int)floorf(sm1 * (c2 + 0.5f));
            raykey[(size_t)b * N + n] = key;
        }
        if (inb) {
            int t[4];
            const int k = tiles_of(c1, c2, t);
            for (int i = 0; i < k; ++i) atomicAdd(&cnt[t[i]], 1);
        }
    }
    __syncthreads();
    // exclusive prefix over cnt[0..ntiles): each thread owns a chunk of consecutive tiles
    const int chunk = (ntiles + 1023) / 1024;
    const int lo = min(tid * chunk, ntiles), hi = min(lo + chunk, ntiles);
    int local = 0;
    for (int i = lo; i < hi; ++i) local += cnt[i];
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    int run = base + incl - local;
    for (int i = lo; i < hi; ++i) {
        const int c = cnt[i];
        cnt[i] = run;
        run += c;
    }
    if (tid == 1023) cnt[ntiles] = run;
    __syncthreads();
    int *ts = tile_start + (size_t)b * (ntiles + 1);
    for (int i = tid; i <= ntiles; i += 1024) ts[i] = cnt[i];
    __syncthreads();  // all starts published before the cursors move
    float4 *pts = tile_pts + (size_t)b * 4 * N;
    for (int n = tid; n < N; n += 1024) {
        const float c0 = camb[3 * n], c1 = camb[3 * n + 1], c2 = camb[3 * n + 2];  // written by this thread above
        if (!in_bounds3(c0, c1, c2)) continue;
        int t[4];
        const int k = tiles_of(c1, c2, t);
        for (int i = 0; i < k; ++i) {
            const int pos = atomicAdd(&cnt[t[i]], 1);
            pts[pos] = make_float4(c0, c1, c2, __int_as_float(n));
        }
    }
}

__global__ __launch_bounds__(256) void k_transform_fwd(const float *__restrict__ pc, const float *__restrict__ q,
                                                        float *__restrict__ cam, int N, float fov, float dist)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float nrm;
    const Quat qn = normalize_quat(q + 4 * b, &nrm);
    const Quat qs = conj_quat(qn);
    const size_t o = ((size_t)b * N + n) * 3;
    float z, yo, xo;
    transform_point(qn, qs, pc + o, fov, dist, z, yo, xo);
    cam[o] = z;  // cam:36-39 stack([z,y,x])
    cam[o + 1] = yo;
    cam[o + 2] = xo;
}

// P1 alone: PointsQuaternionsRotator.rotate_points (quaternions/points_quaternions.py:41-81), both directions:
// inverse = 0: q (x) p (x) q*   (points_quaternions.py:72-75),   inverse = 1: q* (x) p (x) q   (:67-70); same rounding
// contract as transform_point.
__global__ __launch_bounds__(256) void k_rotate_fwd(const float *__restrict__ pc, const float *__restrict__ q,
                                                     float *__restrict__ out, int N, int inverse)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float nrm;
    const Quat qn = normalize_quat(q + 4 * b, &nrm);
    const Quat qs = conj_quat(qn);
    const size_t o = ((size_t)b * N + n) * 3;
    Quat p4;
    p4.w = 0.0f;
    p4.x = pc[o];
    p4.y = pc[o + 1];
    p4.z = pc[o + 2];
    const Quat r = inverse ? hamilton(hamilton(qs, p4), qn) : hamilton(hamilton(qn, p4), qs);
    out[o] = r.x;  // points_quaternions.py:81: new_point[:, :, 1:4]
    out[o + 1] = r.y;
    out[o + 2] = r.z;
}

// r = a (x) b ; dr -> da, db (accumulating)
__device__ __forceinline__ void hamilton_bwd(const Quat a, const Quat b, const Quat dr, Quat &da, Quat &db)
{
    da.w += dr.w * b.w + dr.x * b.x + dr.y * b.y + dr.z * b.z;
    da.x += -dr.w * b.x + dr.x * b.w - dr.y * b.z + dr.z * b.y;
    da.y += -dr.w * b.y + dr.x * b.z + dr.y * b.w - dr.z * b.x;
    da.z += -dr.w * b.z - dr.x * b.y + dr.y * b.x + dr.z * b.w;
    db.w += dr.w * a.w + dr.x * a.x + dr.y * a.y + dr.z * a.z;
    db.x += -dr.w * a.x + dr.x * a.w + dr.y * a.z - dr.z * a.y;
    db.y += -dr.w * a.y - dr.x * a.z + dr.y * a.w + dr.z * a.x;
    db.z += -dr.w * a.z + dr.x * a.y - dr.y * a.x + dr.z * a.w;
}

// One workgroup per cloud: per-point dpc, block-reduced dq (through the normalisation), and the
// deterministic second-stage reduce of the renderer's per-tile dscale partials.
__global__ __launch_bounds__(256) void k_transform_bwd(const float *__restrict__ pc, const float *__restrict__ q,
                                                        const float *__restrict__ dcam, int nslots, int mask_oob,
                                                        float *__restrict__ dpc, float *__restrict__ dq,
                                                        const float *__restrict__ dscale_part, int nparts,
                                                        float *__restrict__ dscale, int N, float fov, float dist, int mode)
{
    // mode 0: camera transform (rotation + perspective); 1 / 2: rotate_points alone, forward / inverse direction
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    float nrm;
    const Quat qn = normalize_quat(q + 4 * b, &nrm);
    Quat qs;
    qs.w = qn.w;
    qs.x = -qn.x;
    qs.y = -qn.y;
    qs.z = -qn.z;
    Quat dqn = {0.f, 0.f, 0.f, 0.f};
    for (int n = tid; n < N; n += 256) {
        const size_t o = ((size_t)b * N + n) * 3;
        Quat p4 = {0.0f, pc[o], pc[o + 1], pc[o + 2]};
        if (mode != 0) {
            const float *g = dcam + o;
            const Quat dr = {0.f, g[0], g[1], g[2]};
            Quat dt = {0.f, 0.f, 0.f, 0.f}, dqs = {0.f, 0.f, 0.f, 0.f}, dp4 = {0.f, 0.f, 0.f, 0.f};
            if (mode == 1) {   // r = (qn p4) qs
                hamilton_bwd(hamilton(qn, p4), qs, dr, dt, dqs);
                hamilton_bwd(qn, p4, dt, dqn, dp4);
            } else {           // r = (qs p4) qn
                hamilton_bwd(hamilton(qs, p4), qn, dr, dt, dqn);
                hamilton_bwd(qs, p4, dt, dqs, dp4);
            }
            dqn.w += dqs.w;
            dqn.x -= dqs.x;
            dqn.y -= dqs.y;
            dqn.z -= dqs.z;
            dpc[o] = dp4.x;
            dpc[o + 1] = dp4.y;
            dpc[o + 2] = dp4.z;
            continue;
        }
        const Quat t = hamilton(qn, p4);
        const Quat r = hamilton(t, qs);
        const float z = r.x, y = r.y, x = r.z;
        const float den = z + dist;
        float dz = 0.f, dyo = 0.f, dxo = 0.f;
        bool live = true;
        if (mask_oob) live = in_bounds3(z, y * fov / den, x * fov / den);
        if (live) {
            const float *g = dcam + o * nslots;
            for (int s = 0; s < nslots; ++s) {
                dz += g[3 * s];
                dyo += g[3 * s + 1];
                dxo += g[3 * s + 2];
            }
        }
        // xo = x*fov/den, yo = y*fov/den, z passes through and feeds den
        const float inv = fov / den;
        Quat dr;
        dr.w = 0.f;
        dr.z = dxo * inv;
        dr.y = dyo * inv;
        dr.x = dz - (dxo * x + dyo * y) * inv / den;
        Quat dt = {0.f, 0.f, 0.f, 0.f}, dqs = {0.f, 0.f, 0.f, 0.f}, dp4 = {0.f, 0.f, 0.f, 0.f};
        hamilton_bwd(t, qs, dr, dt, dqs);
        hamilton_bwd(qn, p4, dt, dqn, dp4);
        dqn.w += dqs.w;
        dqn.x -= dqs.x;
        dqn.y -= dqs.y;
        dqn.z -= dqs.z;
        dpc[o] = dp4.x;
        dpc[o + 1] = dp4.y;
        dpc[o + 2] = dp4.z;
    }
    // block reduce of dqn (double for the cross-thread stage) + dscale partials
    __shared__ double red[5][256];
    double ds = 0.0;
    if (dscale_part)
        for (int i = tid; i < nparts; i += 256) ds += (double)dscale_part[(size_t)b * nparts + i];
    red[0][tid] = dqn.w;
    red[1][tid] = dqn.x;
    red[2][tid] = dqn.y;
    red[3][tid] = dqn.z;
    red[4][tid] = ds;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
            for (int k = 0; k < 5; ++k) red[k][tid] += red[k][tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        // qn = q / den, den = max(||q||, 1e-12)
        const double g[4] = {red[0][0], red[1][0], red[2][0], red[3][0]};
        const double qr[4] = {q[4 * b], q[4 * b + 1], q[4 * b + 2], q[4 * b + 3]};
        const double n = (double)nrm;
        const double den = n < 1e-12 ? 1e-12 : n;
        double dot = 0.0;
        for (int i = 0; i < 4; ++i) dot += g[i] * qr[i];
        for (int i = 0; i < 4; ++i) {
            double v = g[i] / den;
            if (n >= 1e-12) v -= dot * qr[i] / (den * den * n);
            dq[4 * b + i] = (float)v;
        }
        if (dscale) dscale[b] = (float)red[4][0];
    }
}

}  // namespace m355

extern "C" int m355_proj_transform_fwd(const float *pc, const float *q, float *cam, int B, int N, float fov, float dist,
                                       void *stream)
{
    M355_REQUIRE(B >= 0 && N >= 0, "proj_transform_fwd: negative size B=%d N=%d", B, N);
    if (B == 0 || N == 0) return M355_OK;  // empty cloud: nothing to write (pointers of empty tensors are null)
    M355_REQUIRE(pc && q && cam, "proj_transform_fwd: null pointer");
    M355_REQUIRE(B <= 65535, "proj_transform_fwd: B=%d exceeds grid.y", B);
    dim3 grid((N + 255) / 256, B);
    hipLaunchKernelGGL(m355::k_transform_fwd, grid, dim3(256), 0, (hipStream_t)stream, pc, q, cam, N, fov, dist);
    return m355::check_launch("proj_transform_fwd");
}

extern "C" int m355_proj_ntiles(int S)
{
    const int n = m355::tile_count(S);
    return n < 0 ? (int)M355_ERR_UNSUPPORTED : n;
}

extern "C" int m355_proj_bin_fwd(const float *pc, const float *q, float *cam, int32_t *raykey, int32_t *tile_start,
                                 float *tile_pts, int B, int N, int S, float fov, float dist, void *stream)
{
    M355_REQUIRE(B >= 0 && N >= 0, "proj_bin_fwd: negative size B=%d N=%d", B, N);
    m355::TileShape c;
    if (!m355::tile_shape(S, c)) {
        m355::set_error("proj_bin_fwd: S=%d not supported by the fused renderer (2..512)", S);
        return M355_ERR_UNSUPPORTED;
    }
    if (B == 0) return M355_OK;
    M355_REQUIRE(tile_start && (N == 0 || (cam && tile_pts)), "proj_bin_fwd: null pointer");
    M355_REQUIRE(q == nullptr || pc != nullptr || N == 0, "proj_bin_fwd: q given without pc");
    const int tiles_x = (S + c.tw - 1) / c.tw, ntiles = m355::tile_count(S);
    const size_t lds = sizeof(int) * (size_t)(ntiles + 1);
    hipStream_t st = (hipStream_t)stream;
    if (q) {  // transform + bin; q == NULL: cam is the input
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)m355::k_bin<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(m355::k_bin<true>, dim3(B), dim3(1024), lds, st, pc, q, cam, raykey, (int *)tile_start,
                           (float4 *)tile_pts, N, S, c.th, c.tw, tiles_x, ntiles, fov, dist);
    } else {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)m355::k_bin<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(m355::k_bin<false>, dim3(B), dim3(1024), lds, st, pc, q, cam, raykey, (int *)tile_start,
                           (float4 *)tile_pts, N, S, c.th, c.tw, tiles_x, ntiles, fov, dist);
    }
    return m355::check_launch("proj_bin_fwd");
}

extern "C" int m355_proj_transform_bwd(const float *pc, const float *q, const float *dcam, int nslots, int mask_oob,
                                       float *dpc, float *dq, const float *dscale_part, int nparts, float *dscale,
                                       int B, int N, float fov, float dist, void *stream)
{
    M355_REQUIRE(B >= 0 && N >= 0 && nslots >= 1, "proj_transform_bwd: bad size B=%d N=%d nslots=%d", B, N, nslots);
    M355_REQUIRE(q && dq && (N == 0 || (pc && dcam && dpc)), "proj_transform_bwd: null pointer");
    M355_REQUIRE((dscale_part == nullptr) == (dscale == nullptr), "proj_transform_bwd: dscale_part/dscale mismatch");
    if (B == 0) return M355_OK;
    hipLaunchKernelGGL(m355::k_transform_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, pc, q, dcam, nslots,
                       mask_oob, dpc, dq, dscale_part, nparts, dscale, N, fov, dist, 0);
    return m355::check_launch("proj_transform_bwd");
}

extern "C" int m355_quat_rotate_fwd(const float *pc, const float *q, float *out, int B, int N, int inverse, void *stream)
{
    M355_REQUIRE(B >= 0 && N >= 0, "quat_rotate_fwd: negative size B=%d N=%d", B, N);
    if (B == 0 || N == 0) return M355_OK;
    M355_REQUIRE(pc && q && out, "quat_rotate_fwd: null pointer");
    M355_REQUIRE(B <= 65535, "quat_rotate_fwd: B=%d exceeds grid.y", B);
    hipLaunchKernelGGL(m355::k_rotate_fwd, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, pc, q, out, N,
                       inverse ? 1 : 0);
    return m355::check_launch("quat_rotate_fwd");
}

extern "C" int m355_quat_rotate_bwd(const float *pc, const float *q, const float *dout, float *dpc, float *dq, int B, int N,
                                    int inverse, void *stream)
{
    M355_REQUIRE(B >= 0 && N >= 0, "quat_rotate_bwd: negative size B=%d N=%d", B, N);
    M355_REQUIRE(q && dq && (N == 0 || (pc && dout && dpc)), "quat_rotate_bwd: null pointer");
    if (B == 0) return M355_OK;
    hipLaunchKernelGGL(m355::k_transform_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, pc, q, dout, 1, 0, dpc, dq,
                       (const float *)nullptr, 0, (float *)nullptr, N, 1.0f, 1.0f, inverse ? 2 : 1);
    return m355::check_launch("quat_rotate_bwd");
}
