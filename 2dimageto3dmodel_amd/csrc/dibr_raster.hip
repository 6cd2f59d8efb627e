// the DIB-R linear rasteriser behind Renderer.forward (rendering/renderer.py:39-77), i.e. the function
// `kaolin.graphics.dib_renderer.rasterizer.linear_rasterizer(height, width, points3d_bxfx9, points2d_bxfx6,
// normalz_bxfx1, vertex_attr_bxfx3d)` the reference imports (renderer.py:1,62-69) -- Kaolin is not in this image, so the
// published algorithm (Chen et al., NeurIPS 2019; kaolin defaults expand 0.02, knum 30, delta 7000) is built here from
// scratch, MI355X-first; oracle/raster_ref.py states the same algorithm per pixel in torch (parity unpinned, see there).
//
//   k_rast_setup   per face: screen-space bounding box grown by `expand` (EMPTY for back-facing / degenerate faces)
//   k_rast_fwd     one workgroup per 16 x 16 pixel tile, one pixel per lane.  The faces are scanned 256 at a time: a lane
//                  tests ONE face's box against the tile, the hits are compacted IN FACE ORDER (ballot + popcount: the
//                  `first knum faces` rule is an order rule) together with their vertices into LDS, and every lane then
//                  walks that short list for its own pixel:
//                     hard pass  barycentric inside test, keep the largest interpolated z     -> imidx, imwei, imfeat
//                     soft pass  prod_j (1 - exp(-delta d_j^2)) over the first knum boxed faces -> improb
//                  No per-tile face lists in HBM, no [B,H,W,knum] side buffers: the backward recomputes.
//   k_rast_bwd     same tiling.  Covered pixels scatter d imfeat through their saved weights (to the attributes and,
//                  through the barycentric weights, to the face's 2-D vertices); uncovered pixels re-walk the boxed faces
//                  and scatter d improb through exp(-delta d^2) to the two vertices of the closest edge.  fp32 atomics.
// Arithmetic order of the inside test mirrors the torch expressions of the oracle (file compiled with -ffp-contract=off,
// IEEE division): the coverage decision of a pixel is the same on both sides.
// Bound: the face-box scan reads 16 B per (tile, face) from L2 (0.5 GB per batch-64 256^2 render), everything else is
// per-pixel VALU over ~20 faces; output 24 B per pixel.
#include "common.h"

namespace m355 {

constexpr int RT = 16;          // tile side (pixels)
constexpr int RCH = 256;        // faces scanned per round (= threads)

// Deterministic mode (template parameter DET of the two backward kernels): every sum of per-pixel contributions
// -- per tile in LDS, across tiles in memory -- is taken in INTEGERS: integer addition is associative, so neither the order in
// which the pixels of a tile arrive nor the order in which the tiles finish can change a bit.  A contribution v is split as
// v * 2^16 = hi + frac (hi = floor, exact in fp32) and added as the pair (hi, lo = floor(frac * 2^32)) into two 64-bit cells:
// the value of a cell pair is (sum hi * 2^32 + sum lo) / 2^48.  Resolution 2^-48 = 3.6e-15, |v| < 1e9 per contribution (gradients
// near an edge grow like 1 / distance: a single 64-bit cell cannot give both), and 2^16 contributions of that size -- every pixel
// of a 256 x 256 image on one face -- stay inside int64.  fix[0] is a flag word (a non-finite or out-of-range contribution: the
// conversion pass then writes NaN -- nothing is hidden), fix[1 + 2 i], fix[2 + 2 i] the cell pair of output element i.
struct FixPair {
    unsigned long long hi, lo;
};
__device__ __forceinline__ bool rfix_ok(float v) { return fabsf(v) < 1.0e9f; }
__device__ __forceinline__ FixPair rfix_split(float v)
{
    const float t = v * 65536.0f, h = floorf(t);
    FixPair p;
    p.hi = (unsigned long long)(long long)h;
    p.lo = (unsigned long long)((t - h) * 4294967296.0f);
    return p;
}
template <bool DET>
__device__ __forceinline__ void g_add(float *dst, long long *fix, size_t idx, float v)
{
    if constexpr (DET) {
        if (!rfix_ok(v)) {
            atomicOr(reinterpret_cast<unsigned long long *>(fix), 1ull);
            return;
        }
        const FixPair p = rfix_split(v);
        unsigned long long *c = reinterpret_cast<unsigned long long *>(fix) + 1 + 2 * idx;
        atomicAdd(c, p.hi);
        atomicAdd(c + 1, p.lo);
    } else {
        atomicAdd(dst + idx, v);
    }
}
// fix -> the fp32 outputs: out0[0..n0) then out1[0..n1)
__global__ __launch_bounds__(256) void k_rfix_to_f32(const long long *__restrict__ fix, float *__restrict__ out0, size_t n0,
                                                     float *__restrict__ out1, size_t n1)
{
    const bool bad = fix[0] != 0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n0 + n1; i += (size_t)gridDim.x * 256) {
        const double hi = (double)fix[1 + 2 * i], lo = (double)(unsigned long long)fix[2 + 2 * i];
        const float v = bad ? __builtin_nanf("") : (float)(hi * (1.0 / 65536.0) + lo * (1.0 / 281474976710656.0));
        if (i < n0) out0[i] = v;
        else out1[i - n0] = v;
    }
}

struct RastArgs {
    const float *p3;     // [B,F,9]
    const float *p2;     // [B,F,6]
    const float *nz;     // [B,F]
    const float *attr;   // [B,F,3*D]
    const float4 *bbox;  // [B,F] (xmin, ymin, xmax, ymax) grown; empty for culled faces
    float *imfeat;       // [B,H,W,D]
    float *improb;       // [B,H,W]
    int *imidx;          // [B,H,W]
    float *imwei;        // [B,H,W,3]
    // backward
    const float *dfeat;  // [B,H,W,D]
    const float *dprob;  // [B,H,W]
    float *dp2;          // [B,F,6]
    float *dattr;        // [B,F,3*D]
    long long *fix;      // deterministic backward: [flag | dp2 cells | dattr cells] (zeroed by the launcher), else null
    int B, F, H, W, D, knum;
    float delta;
};

__global__ __launch_bounds__(256) void k_rast_setup(const float *__restrict__ p2, const float *__restrict__ nz,
                                                    float4 *__restrict__ bbox, int total, float expand)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float *v = p2 + (size_t)i * 6;
    const float ax = v[0], ay = v[1], bx = v[2], by = v[3], cx = v[4], cy = v[5];
    const float e1x = bx - ax, e1y = by - ay, e2x = cx - ax, e2y = cy - ay;
    const float area = e1x * e2y - e1y * e2x;
    float4 bb;
    if (nz[i] >= 0.0f && fabsf(area) > 1e-12f) {
        bb.x = fminf(fminf(ax, bx), cx) - expand;
        bb.y = fminf(fminf(ay, by), cy) - expand;
        bb.z = fmaxf(fmaxf(ax, bx), cx) + expand;
        bb.w = fmaxf(fmaxf(ay, by), cy) + expand;
    } else {
        bb = make_float4(2e30f, 2e30f, -2e30f, -2e30f);   // never contains a pixel
    }
    bbox[i] = bb;
}

__device__ __forceinline__ float cross2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

// squared distance from p to segment a-b; *t_out = clamped parameter of the closest point
__device__ __forceinline__ float seg_d2(float px, float py, float ax, float ay, float bx, float by, float &t_out)
{
    const float ex = bx - ax, ey = by - ay;
    float t = ((px - ax) * ex + (py - ay) * ey) / fmaxf(ex * ex + ey * ey, 1e-30f);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float rx = px - (ax + t * ex), ry = py - (ay + t * ey);
    t_out = t;
    return rx * rx + ry * ry;
}

struct FaceLds {
    float v[6];      // 2-D vertices
    float z[3];      // depths
    float bb[4];     // grown box
    int id;
};

// Compacts, in face order, the faces of [base, base + 256) whose grown box overlaps the tile's pixel-centre rectangle.
// Returns the number of faces placed in `fl`.
__device__ __forceinline__ int scan_faces(const RastArgs &a, int b, int base, float tx0, float tx1, float ty0, float ty1,
                                          FaceLds *fl, int *wave_cnt)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = base + tid;
    bool hit = false;
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < a.F) {
        bb = a.bbox[(size_t)b * a.F + f];
        hit = bb.x <= tx1 && bb.z > tx0 && bb.y <= ty1 && bb.w > ty0;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    const int n = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    if (hit) {
        const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
        const float *v = a.p2 + ((size_t)b * a.F + f) * 6;
        const float *p3 = a.p3 + ((size_t)b * a.F + f) * 9;
        FaceLds &o = fl[pos];
#pragma unroll
        for (int k = 0; k < 6; ++k) o.v[k] = v[k];
        o.z[0] = p3[2]; o.z[1] = p3[5]; o.z[2] = p3[8];
        o.bb[0] = bb.x; o.bb[1] = bb.y; o.bb[2] = bb.z; o.bb[3] = bb.w;
        o.id = f;
    }
    __syncthreads();
    return n;
}

__global__ __launch_bounds__(256) void k_rast_fwd(RastArgs a)
{
    __shared__ FaceLds fl[RCH];
    __shared__ int wave_cnt[4];
    const int b = blockIdx.z, tid = threadIdx.x;
    const int w = blockIdx.x * RT + (tid & (RT - 1)), h = blockIdx.y * RT + (tid >> 4);
    const bool live = w < a.W && h < a.H;
    // pixel centre (y up), and the tile's centre rectangle
    const float px = (float)(2 * w + 1 - a.W) / (float)a.W, py = (float)(a.H - 2 * h - 1) / (float)a.H;
    const int w0i = blockIdx.x * RT, h0i = blockIdx.y * RT;
    const int w1i = min(w0i + RT - 1, a.W - 1), h1i = min(h0i + RT - 1, a.H - 1);
    const float tx0 = (float)(2 * w0i + 1 - a.W) / (float)a.W, tx1 = (float)(2 * w1i + 1 - a.W) / (float)a.W;
    const float ty1 = (float)(a.H - 2 * h0i - 1) / (float)a.H, ty0 = (float)(a.H - 2 * h1i - 1) / (float)a.H;

    float zbest = -INFINITY, wb0 = 0.f, wb1 = 0.f, wb2 = 0.f, keep = 1.0f;
    int ibest = -1, kcount = 0;
    for (int base = 0; base < a.F; base += RCH) {
        const int n = scan_faces(a, b, base, tx0, tx1, ty0, ty1, fl, wave_cnt);
        for (int j = 0; j < n; ++j) {
            const FaceLds &fc = fl[j];
            if (!(px >= fc.bb[0] && px < fc.bb[2] && py >= fc.bb[1] && py < fc.bb[3])) continue;
            const float ax = fc.v[0], ay = fc.v[1], bx = fc.v[2], by = fc.v[3], cx = fc.v[4], cy = fc.v[5];
            const float area = cross2(bx - ax, by - ay, cx - ax, cy - ay);
            const float w0 = cross2(bx - px, by - py, cx - px, cy - py) / area;
            const float w1 = cross2(cx - px, cy - py, ax - px, ay - py) / area;
            const float w2 = 1.0f - w0 - w1;
            const bool inside = w0 >= 0.0f && w1 >= 0.0f && w2 >= 0.0f;
            if (inside) {
                const float z = w0 * fc.z[0] + w1 * fc.z[1] + w2 * fc.z[2];
                if (z > zbest) {
                    zbest = z; ibest = fc.id; wb0 = w0; wb1 = w1; wb2 = w2;
                }
            }
            if (kcount < a.knum) {   // the first knum boxed faces, in face order
                ++kcount;
                float t;
                const float d2 = fminf(fminf(seg_d2(px, py, ax, ay, bx, by, t), seg_d2(px, py, bx, by, cx, cy, t)),
                                       seg_d2(px, py, cx, cy, ax, ay, t));
                keep *= 1.0f - __expf(-a.delta * d2);
            }
        }
        __syncthreads();   // the list is rebuilt by the next round
    }
    if (!live) return;
    const size_t pix = ((size_t)b * a.H + h) * a.W + w;
    a.imidx[pix] = ibest;
    a.imwei[pix * 3] = wb0; a.imwei[pix * 3 + 1] = wb1; a.imwei[pix * 3 + 2] = wb2;
    a.improb[pix] = ibest >= 0 ? 1.0f : 1.0f - keep;
    for (int d = 0; d < a.D; ++d) {
        float v = 0.0f;
        if (ibest >= 0) {
            const float *at = a.attr + ((size_t)b * a.F + ibest) * 3 * a.D;
            v = wb0 * at[d] + wb1 * at[a.D + d] + wb2 * at[2 * a.D + d];
        }
        a.imfeat[pix * a.D + d] = v;
    }
}

// Gradient accumulators of the faces in the tile's current scan round: pixels of one face are neighbours, so their
// contributions are summed in LDS (ds_add_f32) and leave as ONE global atomic per (tile, face, component) -- with global
// atomics per pixel the backward was 6x the forward (15 same-address atomics for each of ~34 pixels of a face).
template <bool DET>
__global__ __launch_bounds__(256) void k_rast_bwd(RastArgs a)
{
    __shared__ FaceLds fl[RCH];
    __shared__ int wave_cnt[4];
    __shared__ float acc[DET ? 1 : RCH][6 + 3 * 3 + 1];   // D <= 3 on the LDS path (the renderer's uv + mask); wider attributes: global atomics
    __shared__ unsigned long long acch[DET ? RCH : 1][6 + 3 * 3 + 1], accl[DET ? RCH : 1][6 + 3 * 3 + 1];   // DET: the cells as (hi, lo) pairs
    const size_t nP = (size_t)a.B * a.F * 6;   // DET: cell pair of dp2[i] = i, of dattr[i] = nP + i
    // one LDS contribution (the two forms of "add v to cell [j][k] of this round")
    auto lds_add = [&](int j, int k, float v) {
        if constexpr (DET) {
            if (!rfix_ok(v)) {
                atomicOr(reinterpret_cast<unsigned long long *>(a.fix), 1ull);
            } else {
                const FixPair p = rfix_split(v);
                atomicAdd(&acch[j][k], p.hi);
                atomicAdd(&accl[j][k], p.lo);
            }
        } else {
            atomicAdd(&acc[j][k], v);
        }
    };
    const int b = blockIdx.z, tid = threadIdx.x;
    const int w = blockIdx.x * RT + (tid & (RT - 1)), h = blockIdx.y * RT + (tid >> 4);
    const bool live = w < a.W && h < a.H;
    const float px = (float)(2 * w + 1 - a.W) / (float)a.W, py = (float)(a.H - 2 * h - 1) / (float)a.H;
    const int w0i = blockIdx.x * RT, h0i = blockIdx.y * RT;
    const int w1i = min(w0i + RT - 1, a.W - 1), h1i = min(h0i + RT - 1, a.H - 1);
    const float tx0 = (float)(2 * w0i + 1 - a.W) / (float)a.W, tx1 = (float)(2 * w1i + 1 - a.W) / (float)a.W;
    const float ty1 = (float)(a.H - 2 * h0i - 1) / (float)a.H, ty0 = (float)(a.H - 2 * h1i - 1) / (float)a.H;
    const size_t pix = live ? ((size_t)b * a.H + h) * a.W + w : 0;
    const int idx = live ? a.imidx[pix] : -1;
    const bool lds_path = a.D <= 3;
    const int NA = 6 + 3 * a.D;

    // ---- this pixel's contribution to its covering face (through the saved weights)
    float gv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ga[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool has = false;
    if (live && idx >= 0) {
        const float *v = a.p2 + ((size_t)b * a.F + idx) * 6;
        const float ax = v[0], ay = v[1], bx = v[2], by = v[3], cx = v[4], cy = v[5];
        const float w0 = a.imwei[pix * 3], w1 = a.imwei[pix * 3 + 1], w2 = a.imwei[pix * 3 + 2];
        const float *at = a.attr + ((size_t)b * a.F + idx) * 3 * a.D;
        float dw0 = 0.f, dw1 = 0.f, dw2 = 0.f;
        for (int d = 0; d < a.D; ++d) {
            const float g = a.dfeat[pix * a.D + d];
            dw0 += g * at[d]; dw1 += g * at[a.D + d]; dw2 += g * at[2 * a.D + d];
            if (g != 0.0f) {
                has = true;
                if (lds_path) {
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (q == d) { ga[q] = w0 * g; ga[3 + q] = w1 * g; ga[6 + q] = w2 * g; }
                } else {
                    const size_t da = ((size_t)b * a.F + idx) * 3 * a.D;
                    g_add<DET>(a.dattr, a.fix, (DET ? nP : 0) + da + d, w0 * g);
                    g_add<DET>(a.dattr, a.fix, (DET ? nP : 0) + da + a.D + d, w1 * g);
                    g_add<DET>(a.dattr, a.fix, (DET ? nP : 0) + da + 2 * a.D + d, w2 * g);
                }
            }
        }
        // w2 = 1 - w0 - w1;  w0 = N0 / A,  w1 = N1 / A;  N0 = cross(b - p, c - p), N1 = cross(c - p, a - p), A = cross(b - a, c - a)
        const float g0 = dw0 - dw2, g1 = dw1 - dw2;
        if (g0 != 0.0f || g1 != 0.0f) {
            has = true;
            const float A = cross2(bx - ax, by - ay, cx - ax, cy - ay);
            const float dN0 = g0 / A, dN1 = g1 / A, dA = -(g0 * w0 + g1 * w1) / A;
            gv[0] = dN1 * (-(cy - py)) + dA * (by - cy);
            gv[1] = dN1 * (cx - px) + dA * (cx - bx);
            gv[2] = dN0 * (cy - py) + dA * (cy - ay);
            gv[3] = dN0 * (-(cx - px)) + dA * (-(cx - ax));
            gv[4] = dN0 * (-(by - py)) + dN1 * (ay - py) + dA * (-(by - ay));
            gv[5] = dN0 * (bx - px) + dN1 * (-(ax - px)) + dA * (bx - ax);
            if (!lds_path) {
                for (int k = 0; k < 6; ++k) g_add<DET>(a.dp2, a.fix, ((size_t)b * a.F + idx) * 6 + k, gv[k]);
            }
        }
    }
    const float gp = (live && idx < 0) ? a.dprob[pix] : 0.0f;
    // whole tiles with nothing to propagate skip the face scan
    if (!__syncthreads_or((has && lds_path) || gp != 0.0f)) return;
    const float keep = live ? 1.0f - a.improb[pix] : 1.0f;
    int kcount = 0;
    for (int base = 0; base < a.F; base += RCH) {
        const int n = scan_faces(a, b, base, tx0, tx1, ty0, ty1, fl, wave_cnt);
        for (int t = tid; t < n * 16; t += 256) {
            if constexpr (DET) acch[t >> 4][t & 15] = accl[t >> 4][t & 15] = 0;
            else acc[t >> 4][t & 15] = 0.0f;
        }
        __syncthreads();
        // covered pixel: its face is in exactly one round's list (a face covering a pixel of the tile overlaps the tile)
        if (has && lds_path && idx >= base && idx < base + RCH) {
            int jf = -1;
            for (int j = 0; j < n; ++j)
                if (fl[j].id == idx) { jf = j; break; }
            if (jf >= 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (gv[k] != 0.0f) lds_add(jf, k, gv[k]);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    if (d < a.D) {
                        if (ga[d] != 0.0f) lds_add(jf, 6 + d, ga[d]);
                        if (ga[3 + d] != 0.0f) lds_add(jf, 6 + a.D + d, ga[3 + d]);
                        if (ga[6 + d] != 0.0f) lds_add(jf, 6 + 2 * a.D + d, ga[6 + d]);
                    }
                }
            }
        }
        // uncovered pixel: improb = 1 - prod (1 - a_j), a_j = exp(-delta d_j^2) over the first knum boxed faces
        for (int j = 0; j < n; ++j) {
            const FaceLds &fc = fl[j];
            if (!(px >= fc.bb[0] && px < fc.bb[2] && py >= fc.bb[1] && py < fc.bb[3])) continue;
            if (kcount >= a.knum) break;
            ++kcount;
            if (gp == 0.0f) continue;
            const float ax = fc.v[0], ay = fc.v[1], bx = fc.v[2], by = fc.v[3], cx = fc.v[4], cy = fc.v[5];
            float t0, t1, t2;
            const float d0 = seg_d2(px, py, ax, ay, bx, by, t0), d1 = seg_d2(px, py, bx, by, cx, cy, t1),
                        d2c = seg_d2(px, py, cx, cy, ax, ay, t2);
            int e = 0;   // the closest edge (ties: the first)
            float dm = d0, t = t0;
            if (d1 < dm) { dm = d1; t = t1; e = 1; }
            if (d2c < dm) { dm = d2c; t = t2; e = 2; }
            const float aj = __expf(-a.delta * dm);
            const float om = 1.0f - aj;
            if (om <= 0.0f) continue;
            // d improb / d a_j = keep / (1 - a_j),  d a_j / d d^2 = -delta a_j
            const float gd2 = gp * (keep / om) * (-a.delta * aj);
            const float sx = e == 0 ? ax : (e == 1 ? bx : cx), sy = e == 0 ? ay : (e == 1 ? by : cy);
            const float ex_ = e == 0 ? bx : (e == 1 ? cx : ax), ey_ = e == 0 ? by : (e == 1 ? cy : ay);
            const float rx = px - (sx + t * (ex_ - sx)), ry = py - (sy + t * (ey_ - sy));
            // d d^2 / d start = -2 (1 - t) r,  d d^2 / d end = -2 t r
            const int is = 2 * e, ie = 2 * ((e + 1) % 3);
            lds_add(j, is, gd2 * -2.0f * (1.0f - t) * rx);
            lds_add(j, is + 1, gd2 * -2.0f * (1.0f - t) * ry);
            lds_add(j, ie, gd2 * -2.0f * t * rx);
            lds_add(j, ie + 1, gd2 * -2.0f * t * ry);
        }
        __syncthreads();
        // flush: one global atomic per (face of the round, component) that received something
        for (int t = tid; t < n * 16; t += 256) {
            const int jj = t >> 4, k = t & 15;
            if (k >= NA) continue;
            const int f = fl[jj].id;
            const size_t cell = k < 6 ? ((size_t)b * a.F + f) * 6 + k : ((size_t)b * a.F + f) * 3 * a.D + (k - 6);
            if constexpr (DET) {
                const unsigned long long vh = acch[jj][k], vl = accl[jj][k];
                unsigned long long *c = reinterpret_cast<unsigned long long *>(a.fix) + 1 + 2 * ((k < 6 ? 0 : nP) + cell);
                if (vh != 0) atomicAdd(c, vh);
                if (vl != 0) atomicAdd(c + 1, vl);
            } else {
                const float v = acc[jj][k];
                if (v != 0.0f) atomicAdd((k < 6 ? a.dp2 : a.dattr) + cell, v);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------- fragment shader
// fragmentshader (rendering/fragment_shader.py:22-37) with texinterpolation (:6-20) for filtering = 'bilinear':
//   grid = (uv * 2 - 1) * (1, -1);  tex = grid_sample(texture, grid, bilinear, align_corners=True, zeros padding);
//   color = tex * hard            (background: lerp(background, tex, hard))
// uvm [B,H,W,3] = (u, v, hard) as the rasteriser writes it; texture [B,3,TH,TW]; color [B,H,W,3].  One lane per pixel.
struct ShadeArgs {
    const float *uvm, *tex, *bg;
    float *color;
    const float *dcolor;
    float *duvm, *dtex, *dbg;
    long long *fix;   // deterministic backward: [flag | dtex cells] (zeroed by the launcher), else null
    int B, H, W, TH, TW;
};

__device__ __forceinline__ void shade_coords(const ShadeArgs &a, float u, float v, float &fx, float &fy, int &x0, int &y0)
{
    const float gx = u * 2.0f - 1.0f, gy = (v * 2.0f - 1.0f) * -1.0f;
    fx = (gx + 1.0f) * 0.5f * (float)(a.TW - 1);   // align_corners=True
    fy = (gy + 1.0f) * 0.5f * (float)(a.TH - 1);
    x0 = (int)floorf(fx);
    y0 = (int)floorf(fy);
}

template <bool BWD, bool DET = false>
__global__ __launch_bounds__(256) void k_shade(ShadeArgs a)
{
    const size_t HW = (size_t)a.H * a.W, total = (size_t)a.B * HW, THW = (size_t)a.TH * a.TW;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / HW;
        const float u = a.uvm[i * 3], v = a.uvm[i * 3 + 1], hard = a.uvm[i * 3 + 2];
        float fx, fy;
        int x0, y0;
        shade_coords(a, u, v, fx, fy, x0, y0);
        const float tx = fx - (float)x0, ty = fy - (float)y0;
        const float wgt[4] = {(1.0f - tx) * (1.0f - ty), tx * (1.0f - ty), (1.0f - tx) * ty, tx * ty};
        const int xs[4] = {x0, x0 + 1, x0, x0 + 1}, ys[4] = {y0, y0, y0 + 1, y0 + 1};
        const float *tb = a.tex + b * 3 * THW;
        float t[3] = {0.f, 0.f, 0.f}, c4[3][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in = (unsigned)xs[k] < (unsigned)a.TW && (unsigned)ys[k] < (unsigned)a.TH;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                c4[c][k] = in ? tb[c * THW + (size_t)ys[k] * a.TW + xs[k]] : 0.0f;
                t[c] += c4[c][k] * wgt[k];
            }
        }
        if (!BWD) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float bgv = a.bg ? a.bg[i * 3 + c] : 0.0f;
                a.color[i * 3 + c] = a.bg ? bgv + hard * (t[c] - bgv) : t[c] * hard;   // torch.lerp: start + w (end - start)
            }
        } else {
            float dt[3], dhard = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float g = a.dcolor[i * 3 + c];
                const float bgv = a.bg ? a.bg[i * 3 + c] : 0.0f;
                dt[c] = g * hard;
                dhard += g * (t[c] - bgv);
                if (a.dbg) a.dbg[i * 3 + c] = g * (1.0f - hard);
            }
            float dfx = 0.0f, dfy = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dfx += dt[c] * ((c4[c][1] - c4[c][0]) * (1.0f - ty) + (c4[c][3] - c4[c][2]) * ty);
                dfy += dt[c] * ((c4[c][2] - c4[c][0]) * (1.0f - tx) + (c4[c][3] - c4[c][1]) * tx);
            }
            // fx = u (TW - 1),  fy = (1 - v)(TH - 1)
            a.duvm[i * 3] = dfx * (float)(a.TW - 1);
            a.duvm[i * 3 + 1] = -dfy * (float)(a.TH - 1);
            a.duvm[i * 3 + 2] = dhard;
            if (a.dtex) {
                const size_t db = b * 3 * THW;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool in = (unsigned)xs[k] < (unsigned)a.TW && (unsigned)ys[k] < (unsigned)a.TH;
                    if (in && wgt[k] != 0.0f) {
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            if (dt[c] != 0.0f) g_add<DET>(a.dtex, a.fix, db + c * THW + (size_t)ys[k] * a.TW + xs[k], dt[c] * wgt[k]);
                    }
                }
            }
        }
    }
}

}  // namespace m355

using namespace m355;

extern "C" int m355_dibr_shade_fwd(const float *uvm_bxhxwx3, const float *texture_bx3xthxtw, const float *background_bxhxwx3,
                                   float *color_bxhxwx3, int B, int H, int W, int TH, int TW, void *stream)
{
    M355_REQUIRE(uvm_bxhxwx3 && texture_bx3xthxtw && color_bxhxwx3 && B > 0 && H > 0 && W > 0 && TH > 0 && TW > 0,
                 "dibr_shade_fwd: bad argument");
    ShadeArgs a = {};
    a.uvm = uvm_bxhxwx3; a.tex = texture_bx3xthxtw; a.bg = background_bxhxwx3; a.color = color_bxhxwx3;
    a.B = B; a.H = H; a.W = W; a.TH = TH; a.TW = TW;
    const size_t g = ((size_t)B * H * W + 255) / 256;
    hipLaunchKernelGGL(k_shade<false>, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("dibr_shade_fwd");
}

static int shade_bwd_impl(const float *uvm, const float *tex, const float *bg, const float *dcolor, float *duvm, float *dtex,
                          float *dbg, int B, int H, int W, int TH, int TW, void *fix_ws, hipStream_t st)
{
    M355_REQUIRE(uvm && tex && dcolor && duvm && B > 0 && H > 0 && W > 0 && TH > 0 && TW > 0, "dibr_shade_bwd: bad argument");
    const size_t nT = (size_t)B * 3 * TH * TW;
    const bool det = fix_ws != nullptr && dtex != nullptr;
    if (dtex && hipMemsetAsync(det ? fix_ws : (void *)dtex, 0, det ? sizeof(long long) * (1 + 2 * nT) : sizeof(float) * nT, st) != hipSuccess) {
        set_error("dibr_shade_bwd: memset failed");
        return M355_ERR_LAUNCH;
    }
    ShadeArgs a = {};
    a.uvm = uvm; a.tex = tex; a.bg = bg; a.dcolor = dcolor;
    a.duvm = duvm; a.dtex = dtex; a.dbg = dbg; a.fix = det ? (long long *)fix_ws : nullptr;
    a.B = B; a.H = H; a.W = W; a.TH = TH; a.TW = TW;
    const size_t g = ((size_t)B * H * W + 255) / 256;
    const dim3 grid((unsigned)(g > 16384 ? 16384 : g));
    if (det) {
        hipLaunchKernelGGL((k_shade<true, true>), grid, dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_rfix_to_f32, dim3((unsigned)((nT + 255) / 256 > 2048 ? 2048 : (nT + 255) / 256)), dim3(256), 0, st,
                           (const long long *)fix_ws, dtex, nT, (float *)nullptr, (size_t)0);
    } else {
        hipLaunchKernelGGL((k_shade<true, false>), grid, dim3(256), 0, st, a);
    }
    return check_launch("dibr_shade_bwd");
}

extern "C" int m355_dibr_shade_bwd(const float *uvm_bxhxwx3, const float *texture_bx3xthxtw, const float *background_bxhxwx3,
                                   const float *dcolor_bxhxwx3, float *duvm_bxhxwx3, float *dtexture_bx3xthxtw,
                                   float *dbackground_bxhxwx3, int B, int H, int W, int TH, int TW, void *stream)
{
    return shade_bwd_impl(uvm_bxhxwx3, texture_bx3xthxtw, background_bxhxwx3, dcolor_bxhxwx3, duvm_bxhxwx3, dtexture_bx3xthxtw,
                          dbackground_bxhxwx3, B, H, W, TH, TW, nullptr, (hipStream_t)stream);
}

extern "C" size_t m355_dibr_shade_bwd_det_ws_bytes(int B, int TH, int TW)
{
    return (B > 0 && TH > 0 && TW > 0) ? sizeof(long long) * (1 + 2 * (size_t)B * 3 * TH * TW) : 0;
}

extern "C" int m355_dibr_shade_bwd_det(const float *uvm_bxhxwx3, const float *texture_bx3xthxtw, const float *background_bxhxwx3,
                                       const float *dcolor_bxhxwx3, void *fix_ws, float *duvm_bxhxwx3, float *dtexture_bx3xthxtw,
                                       float *dbackground_bxhxwx3, int B, int H, int W, int TH, int TW, void *stream)
{
    M355_REQUIRE(fix_ws || !dtexture_bx3xthxtw, "dibr_shade_bwd_det: no workspace");
    return shade_bwd_impl(uvm_bxhxwx3, texture_bx3xthxtw, background_bxhxwx3, dcolor_bxhxwx3, duvm_bxhxwx3, dtexture_bx3xthxtw,
                          dbackground_bxhxwx3, B, H, W, TH, TW, fix_ws, (hipStream_t)stream);
}

extern "C" size_t m355_dibr_ws_bytes(int B, int F) { return (size_t)(B > 0 ? B : 0) * (size_t)(F > 0 ? F : 0) * sizeof(float4); }

static int rast_check(const char *who, int B, int F, int H, int W, int D, int knum)
{
    M355_REQUIRE(B > 0 && B <= 65535 && F > 0 && H > 0 && W > 0 && D >= 1 && D <= 8 && knum >= 0,
                 "%s: bad size B=%d F=%d H=%d W=%d D=%d knum=%d", who, B, F, H, W, D, knum);
    return 0;
}

extern "C" int m355_dibr_rasterize_fwd(int height, int width, const float *points3d_bxfx9, const float *points2d_bxfx6,
                                       const float *normalz_bxfx1, const float *attr_bxfx3d, int B, int F, int D, float expand,
                                       int knum, float delta, void *ws, float *imfeat, float *improb, int32_t *imidx,
                                       float *imwei, void *stream)
{
    if (int rc = rast_check("dibr_rasterize_fwd", B, F, height, width, D, knum)) return rc;
    M355_REQUIRE(points3d_bxfx9 && points2d_bxfx6 && normalz_bxfx1 && attr_bxfx3d && ws && imfeat && improb && imidx && imwei,
                 "dibr_rasterize_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_rast_setup, dim3((B * F + 255) / 256), dim3(256), 0, st, points2d_bxfx6, normalz_bxfx1, (float4 *)ws, B * F,
                       expand);
    RastArgs a = {};
    a.p3 = points3d_bxfx9; a.p2 = points2d_bxfx6; a.nz = normalz_bxfx1; a.attr = attr_bxfx3d; a.bbox = (const float4 *)ws;
    a.imfeat = imfeat; a.improb = improb; a.imidx = imidx; a.imwei = imwei;
    a.B = B; a.F = F; a.H = height; a.W = width; a.D = D; a.knum = knum; a.delta = delta;
    hipLaunchKernelGGL(k_rast_fwd, dim3((width + RT - 1) / RT, (height + RT - 1) / RT, B), dim3(256), 0, st, a);
    return check_launch("dibr_rasterize_fwd");
}

static int rasterize_bwd_impl(int height, int width, const float *points3d_bxfx9, const float *points2d_bxfx6,
                              const float *attr_bxfx3d, int B, int F, int D, int knum, float delta, const void *ws,
                              const float *improb, const int32_t *imidx, const float *imwei, const float *dimfeat,
                              const float *dimprob, float *dpoints2d_bxfx6, float *dattr_bxfx3d, void *fix_ws, hipStream_t st)
{
    if (int rc = rast_check("dibr_rasterize_bwd", B, F, height, width, D, knum)) return rc;
    M355_REQUIRE(points3d_bxfx9 && points2d_bxfx6 && attr_bxfx3d && ws && improb && imidx && imwei && dimfeat && dimprob &&
                     dpoints2d_bxfx6 && dattr_bxfx3d,
                 "dibr_rasterize_bwd: null pointer");
    const size_t nP = (size_t)B * F * 6, nA = (size_t)B * F * 3 * D;
    bool fail;
    if (fix_ws) fail = hipMemsetAsync(fix_ws, 0, sizeof(long long) * (1 + 2 * (nP + nA)), st) != hipSuccess;
    else fail = hipMemsetAsync(dpoints2d_bxfx6, 0, sizeof(float) * nP, st) != hipSuccess ||
                hipMemsetAsync(dattr_bxfx3d, 0, sizeof(float) * nA, st) != hipSuccess;
    if (fail) {
        set_error("dibr_rasterize_bwd: memset failed");
        return M355_ERR_LAUNCH;
    }
    RastArgs a = {};
    a.p3 = points3d_bxfx9; a.p2 = points2d_bxfx6; a.attr = attr_bxfx3d; a.bbox = (const float4 *)ws;
    a.improb = (float *)improb; a.imidx = (int *)imidx; a.imwei = (float *)imwei;
    a.dfeat = dimfeat; a.dprob = dimprob; a.dp2 = dpoints2d_bxfx6; a.dattr = dattr_bxfx3d; a.fix = (long long *)fix_ws;
    a.B = B; a.F = F; a.H = height; a.W = width; a.D = D; a.knum = knum; a.delta = delta;
    const dim3 grid((width + RT - 1) / RT, (height + RT - 1) / RT, B);
    if (fix_ws) {
        hipLaunchKernelGGL(k_rast_bwd<true>, grid, dim3(256), 0, st, a);
        const size_t blocks = (nP + nA + 255) / 256;
        hipLaunchKernelGGL(k_rfix_to_f32, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st, (const long long *)fix_ws,
                           dpoints2d_bxfx6, nP, dattr_bxfx3d, nA);
    } else {
        hipLaunchKernelGGL(k_rast_bwd<false>, grid, dim3(256), 0, st, a);
    }
    return check_launch("dibr_rasterize_bwd");
}

extern "C" int m355_dibr_rasterize_bwd(int height, int width, const float *points3d_bxfx9, const float *points2d_bxfx6,
                                       const float *attr_bxfx3d, int B, int F, int D, int knum, float delta, const void *ws,
                                       const float *improb, const int32_t *imidx, const float *imwei, const float *dimfeat,
                                       const float *dimprob, float *dpoints2d_bxfx6, float *dattr_bxfx3d, void *stream)
{
    return rasterize_bwd_impl(height, width, points3d_bxfx9, points2d_bxfx6, attr_bxfx3d, B, F, D, knum, delta, ws, improb, imidx,
                              imwei, dimfeat, dimprob, dpoints2d_bxfx6, dattr_bxfx3d, nullptr, (hipStream_t)stream);
}

extern "C" size_t m355_dibr_rasterize_bwd_det_ws_bytes(int B, int F, int D)
{
    return (B > 0 && F > 0 && D > 0) ? sizeof(long long) * (1 + 2 * (size_t)B * F * (6 + 3 * (size_t)D)) : 0;
}

extern "C" int m355_dibr_rasterize_bwd_det(int height, int width, const float *points3d_bxfx9, const float *points2d_bxfx6,
                                           const float *attr_bxfx3d, int B, int F, int D, int knum, float delta, const void *ws,
                                           const float *improb, const int32_t *imidx, const float *imwei, const float *dimfeat,
                                           const float *dimprob, void *fix_ws, float *dpoints2d_bxfx6, float *dattr_bxfx3d,
                                           void *stream)
{
    M355_REQUIRE(fix_ws, "dibr_rasterize_bwd_det: no workspace");
    return rasterize_bwd_impl(height, width, points3d_bxfx9, points2d_bxfx6, attr_bxfx3d, B, F, D, knum, delta, ws, improb, imidx,
                              imwei, dimfeat, dimprob, dpoints2d_bxfx6, dattr_bxfx3d, fix_ws, (hipStream_t)stream);
}
