// P3..P6 fused, tuned variant for odd tap counts known at compile time (the reference's kernel_size=21).
//
// Same dataflow and numerics as proj_render.hip (read its header first); what changes is how a ray is
// mapped to lanes.  v1 gave a whole wave to one ray and scanned over depth with ds_bpermute shuffles: a
// dependent chain of ~40 LDS-pipe round trips per ray, latency-bound (rocprof r01: 303 us fwd / 682 us bwd
// at B=64,N=2048,S=128).  Here a ray gets LPR = 16 lanes (one DPP row), each lane owns D = S/LPR consecutive
// depths, so that
//   * the depth convolution is a register window: the lane reads D+24 contiguous floats of its LDS row
//     (zero halos of 12 floats either side make the zero padding of smooth_voxels.py:67-68 implicit) and does
//     21*D independent FMAs;
//   * the prefix product / prefix sum over depth is D sequential steps in-lane plus a 4-step DPP row scan
//     (row_shr 1,2,4,8 -- VALU, no LDS traffic); row totals by the quad_perm / row_mirror butterfly;
//   * a wave works on 4 touched rays at once; untouched rays are compacted away first and just receive
//     the constant "empty" silhouette value.
// S in (128,256] uses 32 lanes per ray, (256,512] a full wave (row_bcast15/31 extend the scan), D stays 8.
// The silhouette value of an untouched ray depends on S only; the host evaluates it once with the kernel's
// exact fp64 recurrence (render_empty_value) and passes it in.
#include "common.h"
#include "proj_render21.h"
#include "tiles.h"

namespace m355 {

constexpr int kThreads21 = 256;
constexpr int kWaves21 = kThreads21 / 64;
constexpr int kHalo = 12;  // floats; >= ntaps/2 and a multiple of 4 (16-byte aligned window loads)

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double old, double x)
{
    const int rl = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int rh = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(rh, rl);
}

// inclusive scans over the LPR lanes of a ray group (LLVM's canonical gfx9 DPP scan)
template <int LPR>
__device__ __forceinline__ double group_scan_prod(double x)
{
    x *= dpp_d<0x111, 0xf>(1.0, x);  // row_shr:1
    x *= dpp_d<0x112, 0xf>(1.0, x);  // row_shr:2
    x *= dpp_d<0x114, 0xf>(1.0, x);  // row_shr:4
    x *= dpp_d<0x118, 0xf>(1.0, x);  // row_shr:8
    if constexpr (LPR >= 32) x *= dpp_d<0x142, 0xa>(1.0, x);  // row_bcast:15 into rows 1,3
    if constexpr (LPR == 64) x *= dpp_d<0x143, 0xc>(1.0, x);  // row_bcast:31 into rows 2,3
    return x;
}

template <int LPR>
__device__ __forceinline__ double group_scan_sum(double x)
{
    x += dpp_d<0x111, 0xf>(0.0, x);
    x += dpp_d<0x112, 0xf>(0.0, x);
    x += dpp_d<0x114, 0xf>(0.0, x);
    x += dpp_d<0x118, 0xf>(0.0, x);
    if constexpr (LPR >= 32) x += dpp_d<0x142, 0xa>(0.0, x);
    if constexpr (LPR == 64) x += dpp_d<0x143, 0xc>(0.0, x);
    return x;
}

// value of the previous lane of the group (identity for the first lane)
template <int LPR>
__device__ __forceinline__ double group_shift1(double x, double identity, int sl)
{
    double y = dpp_d<0x138, 0xf>(identity, x);  // wave_shr:1
    return sl == 0 ? identity : y;
}

// sum over the LPR lanes of a group, result in every lane
template <int LPR>
__device__ __forceinline__ double group_total(double x)
{
    x += dpp_d<0xB1, 0xf>(0.0, x);   // quad_perm [1,0,3,2]
    x += dpp_d<0x4E, 0xf>(0.0, x);   // quad_perm [2,3,0,1]
    x += dpp_d<0x141, 0xf>(0.0, x);  // row_half_mirror
    x += dpp_d<0x140, 0xf>(0.0, x);  // row_mirror
    if constexpr (LPR >= 32) x += __shfl_xor(x, 16, 64);
    if constexpr (LPR == 64) x += __shfl_xor(x, 32, 64);
    return x;
}

// Visit the records (c0,c1,c2,n) the binning kernel filed under this tile (proj_transform.hip k_bin).
template <typename F>
__device__ __forceinline__ void for_points_in_tile(const float4 *__restrict__ pts, int beg, int end, float sm1, int tid,
                                                   F f)
{
    for (int i = beg + tid; i < end; i += kThreads21) {
        const float4 r = pts[i];
        const int f1 = (int)floorf(sm1 * (r.y + 0.5f)), f2 = (int)floorf(sm1 * (r.z + 0.5f));
        f(__float_as_int(r.w), f1, f2, r.x, r.y, r.z);
    }
}

struct Corner {
    int f0;
    float w0[2], w1[2], w2[2];
};

__device__ __forceinline__ Corner corner_weights(float c0, float c1, float c2, float sm1, int fixed_weights)
{
    Corner k;
    const float g0 = sm1 * (c0 + 0.5f), g1 = sm1 * (c1 + 0.5f), g2 = sm1 * (c2 + 0.5f);  // tri:34
    const float fl0 = floorf(g0), fl1 = floorf(g1), fl2 = floorf(g2);
    k.f0 = (int)fl0;
    k.w0[1] = g0 - fl0;  // tri:66  [1.0 - grid - floor, grid - floor]
    k.w1[1] = g1 - fl1;
    k.w2[1] = g2 - fl2;
    if (fixed_weights) {
        k.w0[0] = 1.0f - k.w0[1];
        k.w1[0] = 1.0f - k.w1[1];
        k.w2[0] = 1.0f - k.w2[1];
    } else {
        k.w0[0] = (1.0f - g0) - fl0;
        k.w1[0] = (1.0f - g1) - fl1;
        k.w2[0] = (1.0f - g2) - fl2;
    }
    return k;
}

// out[k] = sum_t tap[t] * w[k + t + (kHalo - HALF)]   (FLIP: w[k - t + HALF + kHalo], the transposed convolution) for the D
// depths of a lane from its register window, two depths per v_pk_fma_f32.  Per output the taps are accumulated in ascending
// order with one fused multiply-add each -- the bits of the scalar loop this replaces.  The taps come from LDS (`taps`: a
// per-lane pointer the compiler cannot prove uniform, see k_render21): one broadcast ds_read_b32 per tap.
typedef float f2r __attribute__((ext_vector_type(2)));
template <int NT, int D, int WIN, bool FLIP>
__device__ __forceinline__ void depth_conv(const float (&w)[WIN], const float *taps, float (&out)[D])
{
    constexpr int HALF = NT / 2;
    static_assert(D % 2 == 0, "two depths per packed operation");
    f2r acc[D / 2];
#pragma unroll
    for (int m = 0; m < D / 2; ++m) acc[m] = f2r{0.0f, 0.0f};
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float tap = taps[t];
        const f2r tp2 = {tap, tap};
#pragma unroll
        for (int m = 0; m < D / 2; ++m) {
            const int j = FLIP ? 2 * m - t + HALF + kHalo : 2 * m + t + (kHalo - HALF);
            acc[m] = __builtin_elementwise_fma(tp2, f2r{w[j], w[j + 1]}, acc[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < D / 2; ++m) {
        out[2 * m] = acc[m].x;
        out[2 * m + 1] = acc[m].y;
    }
}

template <int NT, int LPR, int D, int TH, int TW, bool BWD, bool DETS = false>
__global__ __launch_bounds__(kThreads21) void k_render21(Render21Args a)
{
    static_assert(D % 4 == 0 && NT % 2 == 1 && NT / 2 <= kHalo - 2, "window layout");
    constexpr int RAYS = TH * TW;
    constexpr int SP = LPR * D;
    constexpr int STRIDE = SP + 2 * kHalo;
    constexpr int GPW = 64 / LPR;  // ray groups per wave
    constexpr int HALF = NT / 2;
    constexpr int WIN = D + 2 * kHalo;  // window floats per lane
    // one extra all-zero row (index RAYS) yields the constant silhouette value of untouched rays
    __shared__ __attribute__((aligned(16))) float tile[(RAYS + 1) * STRIDE];
    // DETS (deterministic mode): the splat's cells as 64-bit fixed point -- integer adds commute, so a voxel hit by many points
    // has the same sum whatever order the LDS atomics land in; +64 KB of LDS (one workgroup per CU instead of four: the price)
    __shared__ long long fix[DETS ? RAYS * SP : 1];
    __shared__ int rayflag[RAYS];
    __shared__ int tlist[RAYS];
    __shared__ int tcount;
    __shared__ float wave_ds[kWaves21];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / LPR, sl = lane % LPR;
    const int b = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x % a.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int S = a.S, N = a.N;
    const float sm1 = (float)S - 1.0f;  // tri:34
    const bool has_scale = a.scale != nullptr;
    const float scale = has_scale ? a.scale[b] : 1.0f;
    const float eps = 1e-5f, hi = (float)(1.0 - 1e-5);  // elf:18,32
    const double E = 1.0000100000500002;                 // exp(1e-5f): elf:40-41,48

    // this tile's slice of the binned records; a tile nobody touches never looks at LDS
    const int ntiles = a.tiles_x * a.tiles_y;
    const int *ts = a.tile_start + (size_t)b * (ntiles + 1) + blockIdx.x;
    const int beg = ts[0], end = ts[1];
    const float4 *pts = reinterpret_cast<const float4 *>(a.tile_pts) + (size_t)b * 4 * N;
    if (beg == end) {
        if (BWD) {
            if (tid == 0 && a.dscale_part) a.dscale_part[(size_t)b * ntiles + blockIdx.x] = 0.0f;
        } else {
            for (int r = tid; r < RAYS; r += kThreads21) {
                const int yy = y0 + r / TW, xx = x0 + r % TW;
                if (yy < S && xx < S) a.proj[((size_t)b * S + (S - 1 - yy)) * S + xx] = a.empty_val;
            }
        }
        return;
    }

    // The NT taps are uniform: left to the compiler they live in SGPRs, and the packed FMAs of the backward (two convolutions)
    // want each as an SGPR PAIR -- 42 + the rest overflowed the scalar file: 40-81 SGPR spills into VGPR lanes and 160
    // v_readlane_b32 (+ hazard nops) inside the per-ray loop (round 4, from the ISA).  They sit in LDS instead and each tap is
    // read (one broadcast ds_read_b32) right where its D FMAs are issued.
    // DETS: the fixed-point scale of THIS tile = the largest power of two that keeps (its number of records) maximal weights inside
    // 62 bits, never finer than 2^-40.  Literal weights reach (2S)^3 in magnitude with either sign (tri:66), the "fixed" ones 1.
    // Per tile, not per launch: with the launch-wide bound (all N points in one voxel) a 512^3 grid kept only 18 fractional bits
    // and missed the 2e-5 silhouette contract (2.4e-4); a tile holds a handful of records.  A function of (records, S) only: the
    // same on every run.
    float det_scale = 0.0f;
    if (DETS) {
        auto clog2 = [](int x) { return x > 1 ? 32 - __clz(x - 1) : 0; };
        const int lg = clog2(end - beg) + (a.fixed_weights ? 0 : 3 + 3 * clog2(S));
        const int kfix = min(40, 62 - lg);
        det_scale = __int_as_float((127 + kfix) << 23);
    }
    __shared__ float taps_s[NT + 3];
    if (tid < NT) taps_s[tid] = a.taps[tid];
    int lz = 0;                          // a zero the compiler cannot see through: keeps the tap reads VECTOR loads (a uniform
    asm volatile("" : "+v"(lz));         // address would be hoisted out of the ray loop and moved back into SGPRs)

    // ---- phase 1: zero the tile (incl. halos and the spare zero row)
    {
        float4 *t4 = reinterpret_cast<float4 *>(tile);
        for (int i = tid; i < (RAYS + 1) * STRIDE / 4; i += kThreads21) t4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid; i < RAYS; i += kThreads21) rayflag[i] = 0;
        if (tid == 0) tcount = 0;
        if (DETS)
            for (int i = tid; i < RAYS * SP; i += kThreads21) fix[i] = 0;
    }
    __syncthreads();

    // ---- phase 2: splat (tri:37-60).  ds_add_f32 of (w_i*w_j)*w_k, evaluated left to right as tri:40-41
    for_points_in_tile(pts, beg, end, sm1, tid, [&](int n, int f1, int f2, float c0, float c1, float c2) {
        const Corner k = corner_weights(c0, c1, c2, sm1, a.fixed_weights);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ry = f1 + j - y0;
            if (ry < 0 || ry >= TH) continue;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int rx = f2 + kk - x0;
                if (rx < 0 || rx >= TW) continue;
                const int ray = ry * TW + rx;
                rayflag[ray] = 1;
                float *col = tile + ray * STRIDE + kHalo + k.f0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float wv = k.w0[i] * k.w1[j] * k.w2[kk];
                    if (DETS) {
                        const int z = k.f0 + i;
                        if (z >= 0 && z < SP)
                            atomicAdd(reinterpret_cast<unsigned long long *>(&fix[ray * SP + z]), (unsigned long long)__float2ll_rn(wv * det_scale));
                    } else {
                        atomicAdd(col + i, wv);
                    }
                }
            }
        }
    });
    __syncthreads();
    if (DETS) {
        const double inv = 1.0 / (double)det_scale;
        for (int i = tid; i < RAYS * SP; i += kThreads21) {
            const int ray = i / SP, z = i - ray * SP;
            tile[ray * STRIDE + kHalo + z] = (float)((double)fix[i] * inv);
        }
        __syncthreads();
    }

    // ---- phase 2.5: compact the touched rays (wave 0)
    if (wave == 0) {
        int count = 0;
        for (int base = 0; base < RAYS; base += 64) {
            const int r = base + lane;
            const bool t = r < RAYS && rayflag[r] != 0;
            const unsigned long long m = __ballot(t);
            if (t) tlist[count + __popcll(m & ((1ull << lane) - 1ull))] = r;
            count += __popcll(m);
        }
        if (lane == 0) tcount = count;
    }
    __syncthreads();
    const int ntouched = tcount;

    // ---- phase 3: GPW rays per wave at a time, LPR lanes per ray, D depths per lane
    float ds_lane = 0.0f;
    const int iters = (ntouched + kWaves21 * GPW - 1) / (kWaves21 * GPW);
    for (int it = 0; it < iters; ++it) {
        const int li = it * (kWaves21 * GPW) + wave * GPW + grp;
        const bool valid = li < ntouched;
        const int ray = valid ? tlist[li] : RAYS;  // idle groups chew on the spare all-zero row
        float *rowp = tile + ray * STRIDE + kHalo;  // logical depth 0
        // -- load own depths, clamp (tri:74), publish the clamped values for the neighbours' windows
        float raw[D];
#pragma unroll
        for (int v = 0; v < D / 4; ++v) {
            const float4 x = *reinterpret_cast<const float4 *>(rowp + sl * D + 4 * v);
            raw[4 * v] = x.x;
            raw[4 * v + 1] = x.y;
            raw[4 * v + 2] = x.z;
            raw[4 * v + 3] = x.w;
        }
#pragma unroll
        for (int v = 0; v < D / 4; ++v) {
            float4 x;
            x.x = fminf(fmaxf(raw[4 * v], 0.0f), 1.0f);
            x.y = fminf(fmaxf(raw[4 * v + 1], 0.0f), 1.0f);
            x.z = fminf(fmaxf(raw[4 * v + 2], 0.0f), 1.0f);
            x.w = fminf(fmaxf(raw[4 * v + 3], 0.0f), 1.0f);
            if (valid) *reinterpret_cast<float4 *>(rowp + sl * D + 4 * v) = x;
        }
        // -- depth convolution from a register window (smooth_voxels.py:72, depth kernel, zero padded)
        float sm[D];
        {
            float w[WIN];
#pragma unroll
            for (int v = 0; v < WIN / 4; ++v) {
                const float4 x = *reinterpret_cast<const float4 *>(rowp + sl * D - kHalo + 4 * v);
                w[4 * v] = x.x;
                w[4 * v + 1] = x.y;
                w[4 * v + 2] = x.z;
                w[4 * v + 3] = x.w;
            }
            depth_conv<NT, D, WIN, false>(w, taps_s + lz, sm);
        }
        // -- scale/clamp (sm:80-82), occupancy clamp (elf:32), q = 1-o in fp32 (elf:34), prefix products
        float o[D], qf[D];
        double lp[D], run = 1.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float c = sm[k];
            if (has_scale) c = fminf(fmaxf(c * scale, 0.0f), 1.0f);
            float ov = fminf(fmaxf(c, eps), hi);
            float q = 1.0f - ov;
            if (sl * D + k >= S) {
                ov = 0.0f;
                q = 1.0f;
            }
            o[k] = ov;
            qf[k] = q;
            run *= (double)q;
            lp[k] = run;
        }
        const double incl = group_scan_prod<LPR>(run);
        const double ex = group_shift1<LPR>(incl, 1.0, sl);
        double pex[D], tsum = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double p = (k == 0) ? ex : ex * lp[k - 1];
            if (sl == 0 && k == 0) p = E;
            pex[k] = p;
            tsum += p * (double)o[k];
        }
        const double Ttot = group_total<LPR>(tsum);
        if (!BWD) {
            if (valid && sl == 0) {
                const int yy = y0 + ray / TW, xx = x0 + ray % TW;
                a.proj[((size_t)b * S + (S - 1 - yy)) * S + xx] = (float)Ttot;  // elf:81 flip(1)
            }
            continue;
        }
        // ================= backward =================
        double g = 0.0;
        if (valid) {
            const int yy = y0 + ray / TW, xx = x0 + ray % TW;
            g = (double)(a.dproj[((size_t)b * S + (S - 1 - yy)) * S + xx] * a.gmul);
        }
        // suffix sums of T over depth (reverse cumsum of elf:37's autograd)
        double Tl[D], ls = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            ls += pex[k] * (double)o[k];
            Tl[k] = ls;
        }
        const double exT = group_scan_sum<LPR>(ls) - ls;
        float dsm[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const double suf = Ttot - (exT + Tl[k]);
            // suf / q without the fp64 division (a dozen dependent fp64 instructions per depth: together they cost as much as
            // one of the two 21-tap convolutions of this pass): fp32 reciprocal of q = float(1 - o) in [1e-5, 1] (1 ulp), one
            // Newton step in fp64 -> relative error ~1e-14, far below the fp32 rounding of d_o
            const double qd = (double)qf[k];
            double rq = (double)__builtin_amdgcn_rcpf(qf[k]);
            rq = fma(fma(-qd, rq, 1.0), rq, rq);
            const float d_o = (float)(g * (pex[k] - suf * rq));
            float c = sm[k];
            bool pass = true;
            if (has_scale) {
                const float cs = sm[k] * scale;
                pass = (cs >= 0.0f && cs <= 1.0f);  // clamp(0,1) of sm:82
                c = fminf(fmaxf(cs, 0.0f), 1.0f);
            }
            pass = pass && (c >= eps && c <= hi) && (sl * D + k < S);  // clamp(eps,1-eps) of elf:32
            const float dsp = pass ? d_o : 0.0f;
            ds_lane += dsp * sm[k];
            dsm[k] = has_scale ? dsp * scale : dsp;
        }
        // transposed depth convolution: dV[z] = sum_t tap[t] * dsm[z - t + HALF], through the LDS row
        if (valid) {
#pragma unroll
            for (int v = 0; v < D / 4; ++v)
                *reinterpret_cast<float4 *>(rowp + sl * D + 4 * v) =
                    make_float4(dsm[4 * v], dsm[4 * v + 1], dsm[4 * v + 2], dsm[4 * v + 3]);
        }
        float dv[D];
        {
            float w[WIN];
#pragma unroll
            for (int v = 0; v < WIN / 4; ++v) {
                const float4 x = *reinterpret_cast<const float4 *>(rowp + sl * D - kHalo + 4 * v);
                w[4 * v] = x.x;
                w[4 * v + 1] = x.y;
                w[4 * v + 2] = x.z;
                w[4 * v + 3] = x.w;
            }
            depth_conv<NT, D, WIN, true>(w, taps_s + lz, dv);
        }
        // clamp(0,1) mask of tri:74 on the raw splat sum; park dV in the tile for phase 4
        if (valid) {
#pragma unroll
            for (int v = 0; v < D / 4; ++v) {
                float4 x;
                x.x = (raw[4 * v] >= 0.0f && raw[4 * v] <= 1.0f) ? dv[4 * v] : 0.0f;
                x.y = (raw[4 * v + 1] >= 0.0f && raw[4 * v + 1] <= 1.0f) ? dv[4 * v + 1] : 0.0f;
                x.z = (raw[4 * v + 2] >= 0.0f && raw[4 * v + 2] <= 1.0f) ? dv[4 * v + 2] : 0.0f;
                x.w = (raw[4 * v + 3] >= 0.0f && raw[4 * v + 3] <= 1.0f) ? dv[4 * v + 3] : 0.0f;
                *reinterpret_cast<float4 *>(rowp + sl * D + 4 * v) = x;
            }
        }
    }

    if (!BWD) {
        const float ev = a.empty_val;
        for (int r = tid; r < RAYS; r += kThreads21) {
            const int yy = y0 + r / TW, xx = x0 + r % TW;
            if (yy < S && xx < S && rayflag[r] == 0) a.proj[((size_t)b * S + (S - 1 - yy)) * S + xx] = ev;
        }
        return;
    }

    {
        float s = ds_lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) wave_ds[wave] = s;
    }
    __syncthreads();
    if (tid == 0 && a.dscale_part) {
        float s = 0.0f;
        for (int w = 0; w < kWaves21; ++w) s += wave_ds[w];
        a.dscale_part[(size_t)b * (a.tiles_x * a.tiles_y) + blockIdx.x] = s;
    }

    // ---- phase 4: every point touching the tile gathers dV at its corners; one gradient slot per ray (j,k)
    for_points_in_tile(pts, beg, end, sm1, tid, [&](int n, int f1, int f2, float c0, float c1, float c2) {
        const Corner k = corner_weights(c0, c1, c2, sm1, a.fixed_weights);
        const float dw[2] = {-1.0f, 1.0f};  // d w[0]/dg = -1, d w[1]/dg = +1 (floor has zero gradient)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ryy = f1 + j - y0;
            if (ryy < 0 || ryy >= TH) continue;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int rx = f2 + kk - x0;
                if (rx < 0 || rx >= TW) continue;
                const float *col = tile + (ryy * TW + rx) * STRIDE + kHalo + k.f0;
                float dg0 = 0.f, dg1 = 0.f, dg2 = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float gv = col[i];
                    dg0 += gv * (dw[i] * k.w1[j] * k.w2[kk]);
                    dg1 += gv * (k.w0[i] * dw[j] * k.w2[kk]);
                    dg2 += gv * (k.w0[i] * k.w1[j] * dw[kk]);
                }
                float *out = a.dcam_slots + (((size_t)b * N + n) * 4 + (j * 2 + kk)) * 3;
                out[0] = dg0 * sm1;  // g = (S-1)(c+0.5)
                out[1] = dg1 * sm1;
                out[2] = dg2 * sm1;
            }
        }
    });
}

template <bool BWD>
int launch_render21(Render21Args a, int B, hipStream_t st)
{
    TileShape c;
    if (!tile_shape(a.S, c)) {
        set_error("proj_render: S=%d not supported by the fused kernel (max 512)", a.S);
        return M355_ERR_UNSUPPORTED;
    }
    a.tiles_x = (a.S + c.tw - 1) / c.tw;
    a.tiles_y = (a.S + c.th - 1) / c.th;
    dim3 grid(a.tiles_x * a.tiles_y, B), block(kThreads21);
    if (a.det_scale != 0.0f) {
        // (the kernel derives the fixed-point scale per tile from the tile's record count; det_scale is only the switch here)
        if (a.S <= 64) hipLaunchKernelGGL((k_render21<21, 16, 4, 8, 8, BWD, true>), grid, block, 0, st, a);
        else if (a.S <= 128) hipLaunchKernelGGL((k_render21<21, 16, 8, 8, 8, BWD, true>), grid, block, 0, st, a);
        else if (a.S <= 256) hipLaunchKernelGGL((k_render21<21, 32, 8, 4, 8, BWD, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_render21<21, 64, 8, 4, 4, BWD, true>), grid, block, 0, st, a);
        return check_launch(BWD ? "proj_render_bwd(21, deterministic splat)" : "proj_render_fwd(21, deterministic splat)");
    }
    if (a.S <= 64) hipLaunchKernelGGL((k_render21<21, 16, 4, 8, 8, BWD>), grid, block, 0, st, a);
    else if (a.S <= 128) hipLaunchKernelGGL((k_render21<21, 16, 8, 8, 8, BWD>), grid, block, 0, st, a);
    else if (a.S <= 256) hipLaunchKernelGGL((k_render21<21, 32, 8, 4, 8, BWD>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_render21<21, 64, 8, 4, 4, BWD>), grid, block, 0, st, a);
    return check_launch(BWD ? "proj_render_bwd(21)" : "proj_render_fwd(21)");
}

template int launch_render21<false>(Render21Args, int, hipStream_t);
template int launch_render21<true>(Render21Args, int, hipStream_t);

// proj of a ray whose occupancy is eps everywhere: E*eps + sum_{d=1}^{S-1} q^d * eps with q = float(1 - eps),
// accumulated exactly as the kernel does (prefix product, then the sum).  Lane partitioning only changes the
// association of the fp64 sum (~1e-16 relative).
float render_empty_value(int S)
{
    const float eps = 1e-5f;
    const float q = 1.0f - eps;
    const double E = 1.0000100000500002;
    double P = 1.0, sum = 0.0;
    for (int d = 0; d < S; ++d) {
        sum += (d == 0 ? E : P) * (double)eps;
        P *= (double)q;
    }
    return (float)sum;
}

}  // namespace m355
