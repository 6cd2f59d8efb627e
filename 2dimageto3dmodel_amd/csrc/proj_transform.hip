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

__global__ __launch_bounds__(256) void k_transform_fwd(const float *__restrict__ pc, const float *__restrict__ q,
                                                        float *__restrict__ cam, int32_t *__restrict__ raykey,
                                                        int N, int S, float fov, float dist)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float nrm;
    const Quat qn = normalize_quat(q + 4 * b, &nrm);
    Quat qs;  // operations.py:131-136: q * (1,-1,-1,-1)
    qs.w = qn.w * 1.0f;
    qs.x = qn.x * -1.0f;
    qs.y = qn.y * -1.0f;
    qs.z = qn.z * -1.0f;
    const size_t o = ((size_t)b * N + n) * 3;
    Quat p4;  // points_quaternions.py:33: pad -> (0, p0, p1, p2)
    p4.w = 0.0f;
    p4.x = pc[o];
    p4.y = pc[o + 1];
    p4.z = pc[o + 2];
    const Quat t = hamilton(qn, p4);  // points_quaternions.py:72-75
    const Quat r = hamilton(t, qs);
    const float z = r.x, y = r.y, x = r.z;  // cam:25  z,y,x = unbind(dim=2)
    const float den = z + dist;
    const float xo = x * fov / den;  // cam:33
    const float yo = y * fov / den;  // cam:34
    cam[o] = z;                      // cam:36-39 stack([z,y,x])
    cam[o + 1] = yo;
    cam[o + 2] = xo;
    if (raykey) {
        int32_t key = -1;
        if (in_bounds3(z, yo, xo)) {
            const float sm1 = (float)S - 1.0f;  // tri:34
            const int f1 = (int)floorf(sm1 * (yo + 0.5f));
            const int f2 = (int)floorf(sm1 * (xo + 0.5f));
            key = (f1 << 16) | f2;
        }
        raykey[(size_t)b * N + n] = key;
    }
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
                                                        float *__restrict__ dscale, int N, float fov, float dist)
{
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

extern "C" int m355_proj_transform_fwd(const float *pc, const float *q, float *cam, int32_t *raykey, int B, int N,
                                       int S, float fov, float dist, void *stream)
{
    M355_REQUIRE(B >= 0 && N >= 0, "proj_transform_fwd: negative size B=%d N=%d", B, N);
    if (B == 0 || N == 0) return M355_OK;  // empty cloud: nothing to write (pointers of empty tensors are null)
    M355_REQUIRE(pc && q && cam, "proj_transform_fwd: null pointer");
    M355_REQUIRE(!raykey || (S >= 2 && S <= 32768), "proj_transform_fwd: raykey needs 2 <= S <= 32768 (S=%d)", S);
    if (B == 0 || N == 0) return M355_OK;
    M355_REQUIRE(B <= 65535, "proj_transform_fwd: B=%d exceeds grid.y", B);
    dim3 grid((N + 255) / 256, B);
    hipLaunchKernelGGL(m355::k_transform_fwd, grid, dim3(256), 0, (hipStream_t)stream, pc, q, cam, raykey, N, S, fov,
                       dist);
    return m355::check_launch("proj_transform_fwd");
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
                       mask_oob, dpc, dq, dscale_part, nparts, dscale, N, fov, dist);
    return m355::check_launch("proj_transform_bwd");
}
