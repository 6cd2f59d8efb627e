// mesh-template deformation + normals + flat (smoothness) loss -- "the other half of the G step"
// (code/main.py:697-699): vtx = MeshTemplate.get_vertex_positions(pred_mesh); loss_flat(mesh, compute_normals(vtx)).
// The reference runs ~45 tiny torch kernels per direction for this (grid_sample, bmm, four index_put/index_select,
// cross, normalize, three gathers, reductions) on [B, 482|962, 3] tensors: launch-bound.  Here each of the three
// reference calls is ONE kernel forward and ONE backward, fp32 throughout.
//
//   get_vertex_positions (rendering/mesh_template.py:125-149): per output vertex v with source s = src[v] (its own
//   entry in the non-negative half, or its mirror partner's):
//       local = bilinear sample of the W-padded displacement map (3 channels) at uv[s]     (grid_sample, align_corners,
//               zero padding; pad = circular by 1 column when symmetric (:166), one wrapped column otherwise (:169))
//       delta = local(1x3) @ tangent_frame[s](3x3)                                          (deform, :106-111)
//       delta.x *= xsign[v]      (-1 for mirrored vertices (:145), 0 on the symmetry plane (:146), +1 otherwise)
//       pos[v] = base[v] + delta                                                              (:148)
//   compute_normals (:113-123): n = normalize(cross(b - a, c - a)), F.normalize eps 1e-12.
//   loss_flat (utils/losses.py:5-17): sum_i mean_{b,f} (n_f . n_{ff[f][i]} - 1)^2 * F/2  =  1/(2B) * sum_{b,f,i} (...)^2.
#include "common.h"

namespace m355 {

struct Tap4 {
    int x[2], y[2];     // padded-map coordinates of the 2 x 2 taps
    float wx[2], wy[2];
};

// grid_sample(align_corners=True) tap positions / weights in the padded map of width Wp = W + (symmetric ? 2 : 1)
__device__ __forceinline__ Tap4 taps(float u, float v, int H, int Wp)
{
    const float fx = (u + 1.0f) * 0.5f * (float)(Wp - 1), fy = (v + 1.0f) * 0.5f * (float)(H - 1);
    const float x0 = floorf(fx), y0 = floorf(fy);
    Tap4 t;
    t.x[0] = (int)x0; t.x[1] = (int)x0 + 1;
    t.y[0] = (int)y0; t.y[1] = (int)y0 + 1;
    t.wx[1] = fx - x0; t.wx[0] = 1.0f - t.wx[1];
    t.wy[1] = fy - y0; t.wy[0] = 1.0f - t.wy[1];
    return t;
}

// padded column -> stored column, or -1 outside the padded map (zero padding of grid_sample)
__device__ __forceinline__ int src_col(int xp, int W, int symmetric)
{
    const int Wp = W + (symmetric ? 2 : 1);
    if (xp < 0 || xp >= Wp) return -1;
    if (symmetric) return xp == 0 ? W - 1 : (xp == W + 1 ? 0 : xp - 1);  // circpad(texture, 1)
    return xp == W ? 0 : xp;                                             // cat(texture, texture[..., :1])
}

// dmap [B,3,H,W]; uv [S,2] (already in grid_sample's [-1,1] frame of the padded map); tgm [S,3,3]; base [V,3];
// src [V] int; xsign [V]  ->  pos [B,V,3]
__global__ __launch_bounds__(256) void k_mesh_vertices_fwd(const float *__restrict__ dmap, const float *__restrict__ uv,
                                                           const float *__restrict__ tgm, const float *__restrict__ base,
                                                           const int *__restrict__ src, const float *__restrict__ xsign,
                                                           float *__restrict__ pos, int B, int V, int H, int W, int symmetric)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (v >= V) return;
    const int s = src[v];
    const Tap4 t = taps(uv[2 * s], uv[2 * s + 1], H, W + (symmetric ? 2 : 1));
    float local[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
            const int xc = src_col(t.x[ix], W, symmetric), yc = t.y[iy];
            if (xc >= 0 && yc >= 0 && yc < H) {
                const float w = t.wx[ix] * t.wy[iy];
#pragma unroll
                for (int k = 0; k < 3; ++k) local[k] += w * dmap[(((size_t)b * 3 + k) * H + yc) * W + xc];
            }
        }
    const float *m = tgm + (size_t)s * 9;
    float d[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) d[j] = local[0] * m[j] + local[1] * m[3 + j] + local[2] * m[6 + j];
    d[0] *= xsign[v];
    float *o = pos + ((size_t)b * V + v) * 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = base[3 * v + j] + d[j];
}

// dpos [B,V,3] -> ddmap [B,3,H,W] in GATHER form: one thread per (sample, texel) walks the static list of (vertex, bilinear
// weight) pairs whose taps land on its texel (CSR table built once per (template, H, W) by the host from the same fp32 tap
// arithmetic, mesh.py) and sums them in list order -- no atomics, no zero fill, the same bits on every run.  (The scatter form
// this replaces added ~4 fp32 atomics per vertex into the texel: its result depended on the order they landed in.)
__global__ __launch_bounds__(256) void k_mesh_vertices_bwd(const float *__restrict__ dpos, const float *__restrict__ tgm,
                                                           const int *__restrict__ src, const float *__restrict__ xsign,
                                                           const int *__restrict__ tptr, const int *__restrict__ tvtx,
                                                           const float *__restrict__ tw, float *__restrict__ ddmap, int V, int HW)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= HW) return;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int e = tptr[t]; e < tptr[t + 1]; ++e) {
        const int v = tvtx[e];
        const float *g = dpos + ((size_t)b * V + v) * 3;
        const float gd[3] = {g[0] * xsign[v], g[1], g[2]};
        const float *m = tgm + (size_t)src[v] * 9;
        const float w = tw[e];
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += w * (gd[0] * m[3 * k] + gd[1] * m[3 * k + 1] + gd[2] * m[3 * k + 2]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) ddmap[((size_t)b * 3 + k) * HW + t] = acc[k];
}

__device__ __forceinline__ void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// pos [B,V,3], faces [F,3] -> normals [B,F,3]
__global__ __launch_bounds__(256) void k_mesh_normals_fwd(const float *__restrict__ pos, const int *__restrict__ faces,
                                                          float *__restrict__ nrm, int B, int V, int F)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (f >= F) return;
    const float *p = pos + (size_t)b * V * 3;
    const float *A = p + 3 * faces[3 * f], *Bp = p + 3 * faces[3 * f + 1], *Cp = p + 3 * faces[3 * f + 2];
    const float v1[3] = {Bp[0] - A[0], Bp[1] - A[1], Bp[2] - A[2]}, v2[3] = {Cp[0] - A[0], Cp[1] - A[1], Cp[2] - A[2]};
    float u[3];
    cross3(v1, v2, u);
    const float inv = 1.0f / fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    float *o = nrm + ((size_t)b * F + f) * 3;
    o[0] = u[0] * inv; o[1] = u[1] * inv; o[2] = u[2] * inv;
}

// gradient of a face's normal with respect to its two edge vectors: n = normalize(v1 x v2), g = dL/dn  ->  d1 = dL/dv1, d2 = dL/dv2
__device__ __forceinline__ void face_edge_grads(const float *A, const float *Bp, const float *Cp, const float *g, float *d1, float *d2)
{
    const float v1[3] = {Bp[0] - A[0], Bp[1] - A[1], Bp[2] - A[2]}, v2[3] = {Cp[0] - A[0], Cp[1] - A[1], Cp[2] - A[2]};
    float u[3];
    cross3(v1, v2, u);
    const float len = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    float du[3];
    if (len > 1e-12f) {  // n = u / |u|:  du = (g - n (n.g)) / |u|
        const float inv = 1.0f / len;
        const float n[3] = {u[0] * inv, u[1] * inv, u[2] * inv};
        const float ng = n[0] * g[0] + n[1] * g[1] + n[2] * g[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) du[j] = (g[j] - n[j] * ng) * inv;
    } else {             // clamped denominator: n = u / eps
#pragma unroll
        for (int j = 0; j < 3; ++j) du[j] = g[j] * 1e12f;
    }
    cross3(v2, du, d1);  // d/dv1 of du . (v1 x v2)
    cross3(du, v1, d2);  // d/dv2
}

// dnrm [B,F,3] -> dpos [B,V,3] in GATHER form: one thread per (sample, vertex) walks the vertex's incident (face, corner) list
// (vptr [V+1], vfc [3F] = 4 * face + corner, sorted: built once per template by the host) and sums the corner's share of every
// face's gradient in list order: corner 0 (a) gets -d1 - d2, corner 1 (b) d1, corner 2 (c) d2.  A face is evaluated by each of
// its three corners (3 x ~60 flops) instead of scattering nine fp32 atomics: deterministic, no zero fill.
__global__ __launch_bounds__(256) void k_mesh_normals_bwd(const float *__restrict__ pos, const int *__restrict__ faces,
                                                          const float *__restrict__ dnrm, const int *__restrict__ vptr,
                                                          const int *__restrict__ vfc, float *__restrict__ dpos, int B, int V, int F)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (v >= V) return;
    const float *p = pos + (size_t)b * V * 3;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int e = vptr[v]; e < vptr[v + 1]; ++e) {
        const int f = vfc[e] >> 2, corner = vfc[e] & 3;
        float d1[3], d2[3];
        face_edge_grads(p + 3 * faces[3 * f], p + 3 * faces[3 * f + 1], p + 3 * faces[3 * f + 2], dnrm + ((size_t)b * F + f) * 3, d1, d2);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] += corner == 0 ? -d1[j] - d2[j] : (corner == 1 ? d1[j] : d2[j]);
    }
    float *o = dpos + ((size_t)b * V + v) * 3;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
}

// nrm [B,F,3], ff [F,3] -> part[b] = sum over the sample's faces of (cos - 1)^2: one workgroup per sample, fixed-order tree
__global__ __launch_bounds__(256) void k_mesh_flat_fwd(const float *__restrict__ nrm, const int *__restrict__ ff,
                                                       float *__restrict__ part, int F)
{
    __shared__ float red[256];
    const int b = blockIdx.x;
    float acc = 0.0f;
    for (int f = threadIdx.x; f < F; f += 256) {
        const float *n1 = nrm + ((size_t)b * F + f) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float *n2 = nrm + ((size_t)b * F + ff[3 * f + i]) * 3;
            const float c = n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2] - 1.0f;
            acc += c * c;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[b] = red[0];
}

// loss = scale * (part[0] + part[1] + ...) in index order (one thread: B <= 65535 additions, deterministic)
__global__ void k_mesh_flat_sum(const float *__restrict__ part, float *__restrict__ loss, int B, float scale)
{
    __shared__ float red[64];
    float s = 0.0f;
    for (int b = threadIdx.x; b < B; b += 64) s += part[b];   // lane l: samples l, l + 64, ... in order
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int l = 0; l < 64; ++l) t += red[l];
        loss[0] = t * scale;
    }
}

// dnrm [B,F,3] from the scalar upstream gradient gl[0], GATHER form: face f collects its own three terms d/dn_f of
// (n_f . n_g - 1)^2, g in ff[f], and the terms of the faces g' that list f as a neighbour (rptr [F+1], ridx: the reverse of ff,
// built once by the host; for a closed manifold it is ff itself) -- the scatter form added the latter with fp32 atomics
__global__ __launch_bounds__(256) void k_mesh_flat_bwd(const float *__restrict__ nrm, const int *__restrict__ ff,
                                                       const int *__restrict__ rptr, const int *__restrict__ ridx,
                                                       const float *__restrict__ gl, float *__restrict__ dnrm, int B, int F, float scale)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (f >= F) return;
    const float gs = gl[0] * scale * 2.0f;
    const float *n1 = nrm + ((size_t)b * F + f) * 3;
    float d[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float *n2 = nrm + ((size_t)b * F + ff[3 * f + i]) * 3;
        const float c = (n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2] - 1.0f) * gs;
#pragma unroll
        for (int j = 0; j < 3; ++j) d[j] += c * n2[j];
    }
    for (int e = rptr[f]; e < rptr[f + 1]; ++e) {
        const float *ng = nrm + ((size_t)b * F + ridx[e]) * 3;
        const float c = (ng[0] * n1[0] + ng[1] * n1[1] + ng[2] * n1[2] - 1.0f) * gs;
#pragma unroll
        for (int j = 0; j < 3; ++j) d[j] += c * ng[j];
    }
    float *o = dnrm + ((size_t)b * F + f) * 3;
    o[0] = d[0]; o[1] = d[1]; o[2] = d[2];
}

}  // namespace m355

using namespace m355;

#define MESH_GRID(n_, B_) dim3(((n_) + 255) / 256, (B_))

extern "C" int m355_mesh_vertices_fwd(const float *dmap, const float *uv, const float *tgm, const float *base, const int *src,
                                      const float *xsign, float *pos, int B, int V, int H, int W, int symmetric, void *stream)
{
    M355_REQUIRE(B >= 0 && V > 0 && H > 1 && W > 1 && B <= 65535, "mesh_vertices_fwd: bad size B=%d V=%d H=%d W=%d", B, V, H, W);
    if (B == 0) return M355_OK;
    M355_REQUIRE(dmap && uv && tgm && base && src && xsign && pos, "mesh_vertices_fwd: null pointer");
    hipLaunchKernelGGL(k_mesh_vertices_fwd, MESH_GRID(V, B), dim3(256), 0, (hipStream_t)stream, dmap, uv, tgm, base, src, xsign, pos,
                       B, V, H, W, symmetric);
    return check_launch("mesh_vertices_fwd");
}

extern "C" int m355_mesh_vertices_bwd(const float *dpos, const float *tgm, const int *src, const float *xsign, const int *tex_ptr,
                                      const int *tex_vtx, const float *tex_w, float *ddmap, int B, int V, int H, int W, void *stream)
{
    M355_REQUIRE(B >= 0 && V > 0 && H > 1 && W > 1 && B <= 65535, "mesh_vertices_bwd: bad size B=%d V=%d H=%d W=%d", B, V, H, W);
    if (B == 0) return M355_OK;
    M355_REQUIRE(dpos && tgm && src && xsign && tex_ptr && tex_vtx && tex_w && ddmap, "mesh_vertices_bwd: null pointer");
    hipLaunchKernelGGL(k_mesh_vertices_bwd, MESH_GRID(H * W, B), dim3(256), 0, (hipStream_t)stream, dpos, tgm, src, xsign, tex_ptr,
                       tex_vtx, tex_w, ddmap, V, H * W);
    return check_launch("mesh_vertices_bwd");
}

extern "C" int m355_mesh_normals_fwd(const float *pos, const int *faces, float *nrm, int B, int V, int F, void *stream)
{
    M355_REQUIRE(B >= 0 && V > 0 && F > 0 && B <= 65535, "mesh_normals_fwd: bad size B=%d V=%d F=%d", B, V, F);
    if (B == 0) return M355_OK;
    M355_REQUIRE(pos && faces && nrm, "mesh_normals_fwd: null pointer");
    hipLaunchKernelGGL(k_mesh_normals_fwd, MESH_GRID(F, B), dim3(256), 0, (hipStream_t)stream, pos, faces, nrm, B, V, F);
    return check_launch("mesh_normals_fwd");
}

extern "C" int m355_mesh_normals_bwd(const float *pos, const int *faces, const float *dnrm, const int *vtx_ptr, const int *vtx_fc,
                                     float *dpos, int B, int V, int F, void *stream)
{
    M355_REQUIRE(B >= 0 && V > 0 && F > 0 && B <= 65535, "mesh_normals_bwd: bad size B=%d V=%d F=%d", B, V, F);
    if (B == 0) return M355_OK;
    M355_REQUIRE(pos && faces && dnrm && vtx_ptr && vtx_fc && dpos, "mesh_normals_bwd: null pointer");
    hipLaunchKernelGGL(k_mesh_normals_bwd, MESH_GRID(V, B), dim3(256), 0, (hipStream_t)stream, pos, faces, dnrm, vtx_ptr, vtx_fc, dpos,
                       B, V, F);
    return check_launch("mesh_normals_bwd");
}

extern "C" int m355_mesh_flat_fwd(const float *nrm, const int *ff, float *loss, float *ws /*[B]*/, int B, int F, void *stream)
{
    M355_REQUIRE(B > 0 && F > 0 && B <= 65535, "mesh_flat_fwd: bad size B=%d F=%d", B, F);
    M355_REQUIRE(nrm && ff && loss && ws, "mesh_flat_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mesh_flat_fwd, dim3(B), dim3(256), 0, st, nrm, ff, ws, F);
    hipLaunchKernelGGL(k_mesh_flat_sum, dim3(1), dim3(64), 0, st, ws, loss, B, 0.5f / (float)B);
    return check_launch("mesh_flat_fwd");
}

extern "C" int m355_mesh_flat_bwd(const float *nrm, const int *ff, const int *rev_ptr, const int *rev_idx, const float *gloss,
                                  float *dnrm, int B, int F, void *stream)
{
    M355_REQUIRE(B > 0 && F > 0 && B <= 65535, "mesh_flat_bwd: bad size B=%d F=%d", B, F);
    M355_REQUIRE(nrm && ff && rev_ptr && rev_idx && gloss && dnrm, "mesh_flat_bwd: null pointer");
    hipLaunchKernelGGL(k_mesh_flat_bwd, MESH_GRID(F, B), dim3(256), 0, (hipStream_t)stream, nrm, ff, rev_ptr, rev_idx, gloss, dnrm, B,
                       F, 0.5f / (float)B);
    return check_launch("mesh_flat_bwd");
}
