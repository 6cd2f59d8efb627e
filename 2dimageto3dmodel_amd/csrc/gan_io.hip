// Input / output glue of the GAN stacks: everything between the NCHW fp32 tensors the reference's callers hand over and the
// NHWC bf16 tensors the MFMA convolutions consume, one launch per reference line group instead of ~15 ATen kernels each.
//
//   k_mask_cat        ModelWrapper.forward (main.py:493,503-507): cat(pred_tex * alpha, alpha) [and, in the D step, the
//                     batch concatenation with cat(X_tex, alpha)]
//   k_pool_pack       TextureDiscriminator / MeshDiscriminator.forward up to conv1 (gan.py:79-99, 192-211):
//                     avg_pool2d(x, f), cat with the mesh map and the positional encoding, the /8 or /16 mask
//                     avg_pool2d(x[:, 3:4], g), NCHW fp32 -> NHWC bf16 with the channel count padded to 8 / 16
//   k_pool_unpack_bwd their adjoint for up to three discriminators at once
//   k_head_tail       Generator.forward after conv_final / conv_mesh (gan.py:407-419): tanh, adjust_poles
//                     (rendering/utils.py:21-26), symmetrize_texture (rendering/utils.py:15-18); the backward emits the
//                     conv's incoming gradient directly in its NHWC bf16 (8-channel) layout plus the bias gradient
//   k_hinge           GANLoss(hinge) over the list of discriminator outputs (utils/losses.py:49-120) with the
//                     [fake; real] split of main.py:414-422 done by index instead of slicing
// All memory-bound elementwise / small-reduction kernels; fp32 arithmetic in the reference's order where it is visible.
#include <stdlib.h>

#include "common.h"

namespace m355 {

// activation element type: bf16 in the product build, fp32 in the EXACT build (see csrc/gan_elem.hip / conv_exact.hip)
#ifdef M355_EXACT
typedef float act_t;
struct __attribute__((aligned(16))) bf16x8i {
    float v[8];
};
__device__ __forceinline__ float bf2f_i(float h) { return h; }
__device__ __forceinline__ bf16x8i pack8_i(const float *z)
{
    bf16x8i r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = z[j];
    return r;
}
// the first four channels of a packed pixel as floats
__device__ __forceinline__ void load4_i(const float *s, float (&f)[4])
{
    const float4 v = *reinterpret_cast<const float4 *>(s);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
#else
typedef short act_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8i;

__device__ __forceinline__ float bf2f_i(short h) { return __uint_as_float(((unsigned int)(unsigned short)h) << 16); }
__device__ __forceinline__ unsigned pack2_i(float lo, float hi)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ bf16x8i pack8_i(const float *z)
{
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 w = {pack2_i(z[0], z[1]), pack2_i(z[2], z[3]), pack2_i(z[4], z[5]), pack2_i(z[6], z[7])};
    return __builtin_bit_cast(bf16x8i, w);
}
__device__ __forceinline__ void load4_i(const short *s, float (&f)[4])
{
    const uint2 v = *reinterpret_cast<const uint2 *>(s);
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}
#endif

// ---------------------------------------------------------------------------------------------------- mask + cat
// X[m,c,:]: m < N: c < 3 ? fake[m,c,:] * alpha[m,:] : alpha[m,:];   m >= N (only with `real`): real / alpha of sample m - N
__global__ __launch_bounds__(256) void k_mask_cat(const float4 *__restrict__ fake, const float4 *__restrict__ real,
                                                  const float4 *__restrict__ alpha, float4 *__restrict__ X, int N, size_t HW4,
                                                  size_t total)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t plane = i / HW4, p = i - plane * HW4;
        const int m = (int)(plane >> 2), c = (int)(plane & 3);
        const int n = m < N ? m : m - N;
        const float4 a = alpha[(size_t)n * HW4 + p];
        float4 v = a;
        if (c < 3) {
            if (m < N) {
                const float4 t = fake[((size_t)n * 3 + c) * HW4 + p];
                v = make_float4(t.x * a.x, t.y * a.y, t.z * a.z, t.w * a.w);
            } else {
                v = real[((size_t)n * 3 + c) * HW4 + p];
            }
        }
        X[i] = v;
    }
}

// dfake[n,c,:] = dX[n,c,:] * alpha[n,:]   (n < N, c < 3)
__global__ __launch_bounds__(256) void k_mask_cat_bwd(const float4 *__restrict__ dX, const float4 *__restrict__ alpha,
                                                      float4 *__restrict__ dfake, size_t HW4, size_t total)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t plane = i / HW4, p = i - plane * HW4;
        const size_t n = plane / 3, c = plane - n * 3;
        const float4 g = dX[(n * 4 + c) * HW4 + p], a = alpha[n * HW4 + p];
        dfake[i] = make_float4(g.x * a.x, g.y * a.y, g.z * a.z, g.w * a.w);
    }
}

// ---------------------------------------------------------------------------------------------------- pool + pack
struct PoolPackArgs {
    const float *x;      // [M,C,H,W] fp32
    const float *extra;  // [M,E,Ho,Wo] fp32 or null
    const float *pos;    // [P,Ho,Wo] fp32 or null
    act_t *out;          // [M,Ho,Wo,CP] bf16
    float *mask;         // [M,Ho/g,Wo/g] fp32 or null
    int M, C, H, W, E, P, CP, Ho, Wo, g, mask_chan;
    // PARTS form (x == null): the 4-channel input is never materialised -- sample m < Nf is cat(fake[m] * alpha[m], alpha[m]),
    // sample m >= Nf is cat(real[m - Nf], alpha[m - Nf])  (k_mask_cat's definition, evaluated in the loader)
    const float *fake, *real, *alpha;
    int Nf;
};

// four consecutive texels of channel c of sample m, plane offset `off` (a multiple of 4).  PARTS: the product is rounded to
// fp32 before anything is added to it (__fmul_rn: no contraction into the pooling sum), i.e. the bits k_mask_cat would have stored.
template <bool PARTS>
__device__ __forceinline__ float4 src4(const PoolPackArgs &a, int m, int c, size_t HW, size_t off)
{
    if (!PARTS) return *reinterpret_cast<const float4 *>(a.x + ((size_t)m * a.C + c) * HW + off);
    const int n = m < a.Nf ? m : m - a.Nf;
    const float4 al = *reinterpret_cast<const float4 *>(a.alpha + (size_t)n * HW + off);
    if (c == 3) return al;
    if (m >= a.Nf) return *reinterpret_cast<const float4 *>(a.real + ((size_t)n * 3 + c) * HW + off);
    const float4 t = *reinterpret_cast<const float4 *>(a.fake + ((size_t)n * 3 + c) * HW + off);
    return make_float4(__fmul_rn(t.x, al.x), __fmul_rn(t.y, al.y), __fmul_rn(t.z, al.z), __fmul_rn(t.w, al.w));
}
template <bool PARTS>
__device__ __forceinline__ float src1(const PoolPackArgs &a, int m, int c, size_t HW, size_t off)
{
    if (!PARTS) return a.x[((size_t)m * a.C + c) * HW + off];
    const int n = m < a.Nf ? m : m - a.Nf;
    const float al = a.alpha[(size_t)n * HW + off];
    if (c == 3) return al;
    if (m >= a.Nf) return a.real[((size_t)n * 3 + c) * HW + off];
    return __fmul_rn(a.fake[((size_t)n * 3 + c) * HW + off], al);
}

// One workgroup = 16 output rows x TW output columns (TW*4 threads, 4 rows each).  F = pooling factor.
// Pooling sums the F x F window row-major and divides once, as ATen's avg_pool2d does.
template <int F, int TW, bool PARTS>
__global__ __launch_bounds__(TW * 4) void k_pool_pack(PoolPackArgs a)
{
    __shared__ float av[16][TW];
    __shared__ float rowsum[16][TW / 4];
    const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
    const int m = blockIdx.z, ox = blockIdx.x * TW + tx;
    const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oy = blockIdx.y * 16 + ty * 4 + r;
        const size_t po = (size_t)oy * a.Wo + ox;
        float z[16], mval = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {   // (c is a compile-time constant after unrolling: no dynamic register indexing)
            float v = 0.0f;
            if (c < 4 && c < a.C) {
                const size_t src = (size_t)(oy * F) * a.W + (size_t)ox * F;
                float s = 0.0f;
                for (int i = 0; i < F; ++i) {
                    if (F % 4 == 0) {
#pragma unroll
                        for (int j = 0; j < F; j += 4) {
                            const float4 q = src4<PARTS>(a, m, c, HW, src + (size_t)i * a.W + j);
                            s += q.x; s += q.y; s += q.z; s += q.w;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < F; ++j) s += src1<PARTS>(a, m, c, HW, src + (size_t)i * a.W + j);
                    }
                }
                v = F == 1 ? s : s / (float)(F * F);
                if (c == a.mask_chan) mval = v;
            } else if (c >= a.C && c - a.C < a.E) {
                v = a.extra[((size_t)m * a.E + (c - a.C)) * HWo + po];
            } else if (c >= a.C + a.E && c - a.C - a.E < a.P) {
                v = a.pos[(size_t)(c - a.C - a.E) * HWo + po];
            }
            z[c] = v;
        }
        act_t *o = a.out + ((size_t)m * HWo + po) * a.CP;
        *reinterpret_cast<bf16x8i *>(o) = pack8_i(z);
        if (a.CP == 16) *reinterpret_cast<bf16x8i *>(o + 8) = pack8_i(z + 8);
        av[ty * 4 + r][tx] = mval;
    }
    if (!a.mask) return;
    __syncthreads();
    // deterministic two-stage g x g mean: row segments, then columns of row sums
    const int g = a.g, cw = TW / g;   // mask cells per tile row
    for (int t = threadIdx.x; t < 16 * cw; t += TW * 4) {
        const int y = t / cw, cx = t - y * cw;
        float s = 0.0f;
        for (int j = 0; j < g; ++j) s += av[y][cx * g + j];
        rowsum[y][cx] = s;
    }
    __syncthreads();
    const int ch = 16 / g;
    for (int t = threadIdx.x; t < ch * cw; t += TW * 4) {
        const int cy = t / cw, cx = t - cy * cw;
        float s = 0.0f;
        for (int i = 0; i < g; ++i) s += rowsum[cy * g + i][cx];
        const int my = blockIdx.y * ch + cy, mx = blockIdx.x * cw + cx;
        a.mask[((size_t)m * (a.Ho / g) + my) * (a.Wo / g) + mx] = s / (float)(g * g);
    }
}

// F = 1, Wo % 64 == 0 (the full-resolution texture discriminator, the largest of the packs): one thread = one row x FOUR
// consecutive pixels: every plane is read with 16-byte loads (8 load instructions per thread instead of 32) and the thread's
// four packed pixels leave as 64 contiguous bytes.  Tile 16 x 64 pixels per 256 threads, as the generic kernel.
template <bool PARTS>
__global__ __launch_bounds__(256) void k_pack1x4(PoolPackArgs a)
{
    __shared__ float av[16][64];
    __shared__ float rowsum[16][16];
    const int cg = threadIdx.x & 15, row = threadIdx.x >> 4;          // column group (4 pixels), tile row
    const int m = blockIdx.z, oy = blockIdx.y * 16 + row, ox = blockIdx.x * 64 + 4 * cg;
    const size_t HW = (size_t)a.H * a.W, po = (size_t)oy * a.W + ox;
    float4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < a.C) t = src4<PARTS>(a, m, c, HW, po);
        else if (c - a.C < a.E) t = *reinterpret_cast<const float4 *>(a.extra + ((size_t)m * a.E + (c - a.C)) * HW + po);
        else if (c - a.C - a.E < a.P) t = *reinterpret_cast<const float4 *>(a.pos + (size_t)(c - a.C - a.E) * HW + po);
        v[c] = t;
    }
    act_t *o = a.out + ((size_t)m * HW + po) * 8;
    {
        const float z0[8] = {v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x};
        const float z1[8] = {v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y};
        const float z2[8] = {v[0].z, v[1].z, v[2].z, v[3].z, v[4].z, v[5].z, v[6].z, v[7].z};
        const float z3[8] = {v[0].w, v[1].w, v[2].w, v[3].w, v[4].w, v[5].w, v[6].w, v[7].w};
        *reinterpret_cast<bf16x8i *>(o) = pack8_i(z0);
        *reinterpret_cast<bf16x8i *>(o + 8) = pack8_i(z1);
        *reinterpret_cast<bf16x8i *>(o + 16) = pack8_i(z2);
        *reinterpret_cast<bf16x8i *>(o + 24) = pack8_i(z3);
    }
    if (!a.mask) return;
    float4 mv = v[0];
#pragma unroll
    for (int c = 1; c < 4; ++c)
        if (c == a.mask_chan) mv = v[c];
    av[row][4 * cg] = mv.x; av[row][4 * cg + 1] = mv.y; av[row][4 * cg + 2] = mv.z; av[row][4 * cg + 3] = mv.w;
    __syncthreads();
    const int g = a.g, cw = 64 / g;
    for (int t = threadIdx.x; t < 16 * cw; t += 256) {
        const int y = t / cw, cx = t - y * cw;
        float s = 0.0f;
        for (int j = 0; j < g; ++j) s += av[y][cx * g + j];
        rowsum[y][cx] = s;
    }
    __syncthreads();
    const int ch = 16 / g;
    for (int t = threadIdx.x; t < ch * cw; t += 256) {
        const int cy = t / cw, cx = t - cy * cw;
        float s = 0.0f;
        for (int i = 0; i < g; ++i) s += rowsum[cy * g + i][cx];
        const int my = blockIdx.y * ch + cy, mx = blockIdx.x * cw + cx;
        a.mask[((size_t)m * (a.Ho / g) + my) * (a.Wo / g) + mx] = s / (float)(g * g);
    }
}

struct PoolUnpackArgs {
    const act_t *dh[3];  // [M,H/f,W/f,CP] bf16 gradients of the packed tensors
    int f[3], CP[3];
    int K;
    float *dx;           // [M,C,H,W] fp32
    int M, C, H, W;
    const float *alpha;  // non-null: the PARTS adjoint -- dx is dfake [M,3,H,W] = (the gradient of channels 0..2) * alpha[m]
};

// dx[m,c,y,x] = sum_k dh_k[m, y/f_k, x/f_k, c] / f_k^2   (C <= 4: one 8-byte read per tensor)
__global__ __launch_bounds__(256) void k_pool_unpack_bwd(PoolUnpackArgs a)
{
    const size_t HW = (size_t)a.H * a.W, total = (size_t)a.M * HW;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / HW, p = i - m * HW;
        const int y = (int)(p / a.W), x = (int)(p - (size_t)y * a.W);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < a.K) {
                const int f = a.f[k], Wk = a.W / f, Hk = a.H / f;
                const act_t *s = a.dh[k] + ((m * Hk + (size_t)(y / f)) * Wk + (size_t)(x / f)) * a.CP[k];
                float v4[4];
                load4_i(s, v4);
                const float inv = 1.0f / (float)(f * f);
                acc[0] += v4[0] * inv;
                acc[1] += v4[1] * inv;
                acc[2] += v4[2] * inv;
                acc[3] += v4[3] * inv;
            }
        }
        if (a.alpha) {               // k_mask_cat_bwd applied in place: dfake = dX[:, :3] * alpha
            const float al = a.alpha[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) a.dx[(m * 3 + c) * HW + p] = acc[c] * al;
            continue;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < a.C) a.dx[(m * a.C + c) * HW + p] = acc[c];
    }
}

// g [M,HW,CP] bf16, channels c0 .. c0+E-1 -> out [M,E,HW] fp32
__global__ __launch_bounds__(256) void k_unpack_range(const act_t *__restrict__ g, float *__restrict__ out, int CP, int c0, int E,
                                                      size_t HW, size_t total)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / HW, p = i - m * HW;
        for (int e = 0; e < E; ++e) out[(m * E + e) * HW + p] = bf2f_i(g[i * CP + c0 + e]);
    }
}

// ---------------------------------------------------------------------------------------------------- generator tail
constexpr int HT_TANH = 1, HT_POLES = 2, HT_SYMM = 4;

__device__ __forceinline__ int mirror_col(int x, int W)  // second destination of half-width column x in the 2W-wide image
{
    return x < W / 2 ? W / 2 - 1 - x : 5 * W / 2 - 1 - x;
}

// y [N,C,H,W] (conv output) -> out [N,C,H,W or 2W]
__global__ __launch_bounds__(256) void k_head_tail_fwd(const float *__restrict__ y, float *__restrict__ out, int C, int H, int W,
                                                       int flags, size_t total)
{
    const int Wo = (flags & HT_SYMM) ? 2 * W : W;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / W;                  // (n*C + c)*H + h
        const int x = (int)(i - row * W), h = (int)(row % H);
        float v = y[i];
        if ((flags & HT_POLES) && (h == 0 || h == H - 1)) {   // rendering/utils.py:23,25: mean over the row
            float s = 0.0f;
            for (int j = 0; j < W; ++j) s += y[row * W + j];
            v = s / (float)W;
        }
        if (flags & HT_TANH) v = tanhf(v);
        if (flags & HT_SYMM) {
            out[row * Wo + W / 2 + x] = v;
            out[row * Wo + mirror_col(x, W)] = v;
        } else {
            out[i] = v;
        }
    }
}

// dout [N,C,H,Wo], out (the forward's result, for tanh') -> g [N,H,W,8] bf16 (conv's dy layout), part[block][4] = the block's
// bias-gradient sums (reduced in block order by k_head_tail_db: deterministic -- this used to be one fp32 atomic per block)
__global__ __launch_bounds__(256) void k_head_tail_bwd(const float *__restrict__ dout, const float *__restrict__ out,
                                                       act_t *__restrict__ g, float *__restrict__ part, int C, int H, int W,
                                                       int flags, size_t total)
{
    __shared__ float red[4][8];
    const int Wo = (flags & HT_SYMM) ? 2 * W : W;
    const size_t HW = (size_t)H * W;
    float bsum[3] = {0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t n = i / HW, p = i - n * HW;
        const int h = (int)(p / W), x = (int)(p - (size_t)h * W);
        float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (c >= C) continue;
            const size_t row = (n * C + c) * H + h;
            auto dv_at = [&](int xx) -> float {   // gradient wrt the (activated) half-width value at column xx
                if (flags & HT_SYMM) return dout[row * Wo + W / 2 + xx] + dout[row * Wo + mirror_col(xx, W)];
                return dout[row * Wo + xx];
            };
            float d;
            const float o = (flags & HT_SYMM) ? out[row * Wo + W / 2 + x] : out[row * Wo + x];
            if ((flags & HT_POLES) && (h == 0 || h == H - 1)) {
                float s = 0.0f;  // every column of a pole row holds the same mean: d mean = sum of the row's gradients
                for (int j = 0; j < W; ++j) s += dv_at(j);
                d = s;
                if (flags & HT_TANH) d *= 1.0f - o * o;
                d /= (float)W;
            } else {
                d = dv_at(x);
                if (flags & HT_TANH) d *= 1.0f - o * o;
            }
            z[c] = d;
            bsum[c] += d;
        }
        *reinterpret_cast<bf16x8i *>(g + i * 8) = pack8_i(z);
    }
    // bias gradient: wave reduce, one partial per workgroup and channel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = bsum[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) red[wave][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        part[(size_t)blockIdx.x * 4 + threadIdx.x] =
            threadIdx.x < 3 ? (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]) : 0.0f;
}

// dbias[c] = part[0][c] + part[1][c] + ... : 64 lanes take every 64th block in order, lane 0 adds the 64 lane sums in order
__global__ __launch_bounds__(256) void k_head_tail_db(const float *__restrict__ part, float *__restrict__ dbias, int nblk, int C)
{
    __shared__ float red[4][64];
    const int c = threadIdx.x >> 6, l = threadIdx.x & 63;
    float s = 0.0f;
    if (c < C)
        for (int b = l; b < nblk; b += 64) s += part[(size_t)b * 4 + c];
    red[c][l] = s;
    __syncthreads();
    if (l == 0 && c < C) {
        float t = 0.0f;
        for (int k = 0; k < 64; ++k) t += red[c][k];
        dbias[c] = t;
    }
}

// ---------------------------------------------------------------------------------------------------- hinge loss
struct HingeArgs {
    const float *p[3];   // logits [B,hw_k]
    const float *m[3];   // masks  [B,hw_k] or null
    float *dp[3];        // gradients (backward)
    int hw[3];
    float w[3];          // per-discriminator weight (1 when unweighted)
    int K, B, split;     // samples [0, split) -> slot 0, [split, B) -> slot 1
    int mode;            // 0: generator (v = p);  1: discriminator (slot 0 = fake target, slot 1 = real target)
    float norm;          // K (unweighted) or sum of the weights
    float *loss;         // [2]
    float *msum;         // [2,K,B]: saved mask sums (hw_k when unmasked) | the per-(discriminator, sample) loss terms (scratch)
    const float *gl;     // [2] incoming gradients of the two losses (backward)
};

__device__ __forceinline__ float hinge_v(float p, int mode, bool real)
{
    if (mode == 0) return p;
    const float t = (real ? p : -p) - 1.0f;   // utils/losses.py:84-93: clamp_max(+-x - 1, 0), negated mean
    return fminf(t, 0.0f);
}

// grid (B, K), one workgroup per (sample, discriminator):  term[k][b] = -w_k / norm / B_slot * sum(v * mask) / sum(mask);
// k_hinge_sum adds the terms of each slot in (k, b) order (was one fp32 atomic per workgroup: order-dependent bits)
__global__ __launch_bounds__(256) void k_hinge_fwd(HingeArgs a)
{
    __shared__ float red[2][4];
    const int b = blockIdx.x, k = blockIdx.y, hw = a.hw[k];
    const bool real = b >= a.split;
    const float *p = a.p[k] + (size_t)b * hw;
    const float *m = a.m[k] ? a.m[k] + (size_t)b * hw : nullptr;
    float s = 0.0f, sm = 0.0f;
    for (int i = threadIdx.x; i < hw; i += 256) {
        const float mv = m ? m[i] : 1.0f;
        s += hinge_v(p[i], a.mode, real) * mv;
        sm += mv;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        sm += __shfl_down(sm, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s;
        red[1][threadIdx.x >> 6] = sm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        sm = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        a.msum[(size_t)k * a.B + b] = sm;
        const int nslot = real ? a.B - a.split : a.split;
        // unmasked: torch.mean over all elements = mean over samples of per-sample means (equal sizes)
        a.msum[(size_t)(a.K + k) * a.B + b] = -a.w[k] / a.norm / (float)nslot * (s / sm);
    }
}

__global__ __launch_bounds__(256) void k_hinge_sum(HingeArgs a)
{
    __shared__ float red[2][256];
    const float *term = a.msum + (size_t)a.K * a.B;
    float s0 = 0.0f, s1 = 0.0f;
    for (int i = threadIdx.x; i < a.K * a.B; i += 256) {   // thread t: entries t, t + 256, ... in order
        const int b = i % a.B;
        if (b >= a.split) s1 += term[i];
        else s0 += term[i];
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (threadIdx.x < 2) {
        float t = 0.0f;
        for (int k = 0; k < 256; ++k) t += red[threadIdx.x][k];
        a.loss[threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void k_hinge_bwd(HingeArgs a)
{
    const int b = blockIdx.x, k = blockIdx.y, hw = a.hw[k];
    const bool real = b >= a.split;
    const float *p = a.p[k] + (size_t)b * hw;
    const float *m = a.m[k] ? a.m[k] + (size_t)b * hw : nullptr;
    float *dp = a.dp[k] + (size_t)b * hw;
    const int nslot = real ? a.B - a.split : a.split;
    const float coef = a.gl[real ? 1 : 0] * (-a.w[k] / a.norm / (float)nslot) / a.msum[(size_t)k * a.B + b];
    for (int i = threadIdx.x; i < hw; i += 256) {
        const float mv = m ? m[i] : 1.0f;
        float dv = 1.0f;
        if (a.mode == 1) {
            const float t = (real ? p[i] : -p[i]) - 1.0f;
            dv = t <= 0.0f ? (real ? 1.0f : -1.0f) : 0.0f;   // ATen's clamp_max backward: grad * (input <= max)
        }
        dp[i] = coef * mv * dv;
    }
}

}  // namespace m355

using namespace m355;

static unsigned grid_for(size_t total)
{
    const size_t g = (total + 255) / 256;
    return (unsigned)(g > 16384 ? 16384 : (g ? g : 1));
}

extern "C" int m355_mask_cat_fwd(const float *fake, const float *real, const float *alpha, float *X, int N, int H, int W,
                                 void *stream)
{
    M355_REQUIRE(fake && alpha && X && N > 0 && H > 0 && W > 0 && (H * W) % 4 == 0, "mask_cat_fwd: bad argument");
    const size_t HW4 = (size_t)H * W / 4, total = (size_t)(real ? 2 : 1) * N * 4 * HW4;
    hipLaunchKernelGGL(k_mask_cat, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4 *)fake,
                       (const float4 *)real, (const float4 *)alpha, (float4 *)X, N, HW4, total);
    return check_launch("mask_cat_fwd");
}

extern "C" int m355_mask_cat_bwd(const float *dX, const float *alpha, float *dfake, int N, int H, int W, void *stream)
{
    M355_REQUIRE(dX && alpha && dfake && N > 0 && H > 0 && W > 0 && (H * W) % 4 == 0, "mask_cat_bwd: bad argument");
    const size_t HW4 = (size_t)H * W / 4, total = (size_t)N * 3 * HW4;
    hipLaunchKernelGGL(k_mask_cat_bwd, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4 *)dX,
                       (const float4 *)alpha, (float4 *)dfake, HW4, total);
    return check_launch("mask_cat_bwd");
}

extern "C" int m355_pool_pack_ok(int C, int H, int W, int f, int E, int P, int g)
{
    if (C < 1 || C > 4 || E < 0 || P < 0 || C + E + P > 16) return 0;
    if (!(f == 1 || f == 2 || f == 4 || f == 8 || f == 16)) return 0;
    if (H % f || W % f) return 0;
    const int Ho = H / f, Wo = W / f;
    if (Ho % 16 || Wo % 16) return 0;
    if (!(g == 0 || g == 4 || g == 8 || g == 16)) return 0;
    if (f > 1 && W % 4) return 0;
    return 1;
}

template <int F>
static void launch_pool_pack(const PoolPackArgs &a, hipStream_t st)
{
    const int TW = a.Wo % 64 == 0 ? 64 : (a.Wo % 32 == 0 ? 32 : 16);
    const dim3 grid(a.Wo / TW, a.Ho / 16, a.M);
    if (a.x) {
        if (TW == 64) hipLaunchKernelGGL((k_pool_pack<F, 64, false>), grid, dim3(256), 0, st, a);
        else if (TW == 32) hipLaunchKernelGGL((k_pool_pack<F, 32, false>), grid, dim3(128), 0, st, a);
        else hipLaunchKernelGGL((k_pool_pack<F, 16, false>), grid, dim3(64), 0, st, a);
    } else {
        if (TW == 64) hipLaunchKernelGGL((k_pool_pack<F, 64, true>), grid, dim3(256), 0, st, a);
        else if (TW == 32) hipLaunchKernelGGL((k_pool_pack<F, 32, true>), grid, dim3(128), 0, st, a);
        else hipLaunchKernelGGL((k_pool_pack<F, 16, true>), grid, dim3(64), 0, st, a);
    }
}

static int pool_pack_launch(const PoolPackArgs &a, int f, hipStream_t st)
{
    if (f == 1 && a.CP == 8 && a.W % 64 == 0 && !getenv("M355_NO_PACK1X4")) {
        if (a.x) hipLaunchKernelGGL(k_pack1x4<false>, dim3(a.W / 64, a.H / 16, a.M), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_pack1x4<true>, dim3(a.W / 64, a.H / 16, a.M), dim3(256), 0, st, a);
        return check_launch("pool_pack_fwd");
    }
    switch (f) {
    case 1: launch_pool_pack<1>(a, st); break;
    case 2: launch_pool_pack<2>(a, st); break;
    case 4: launch_pool_pack<4>(a, st); break;
    case 8: launch_pool_pack<8>(a, st); break;
    default: launch_pool_pack<16>(a, st); break;
    }
    return check_launch("pool_pack_fwd");
}

extern "C" int m355_pool_pack_fwd(const float *x, int M, int C, int H, int W, int f, const float *extra, int E, const float *pos,
                                  int P, void *out, int CP, float *mask, int mask_chan, int g, void *stream)
{
    M355_REQUIRE(x && out && M > 0 && M <= 65535, "pool_pack_fwd: bad argument");
    M355_REQUIRE(m355_pool_pack_ok(C, H, W, f, E, P, mask ? g : 0), "pool_pack_fwd: unsupported shape C=%d H=%d W=%d f=%d E=%d P=%d g=%d", C, H,
                 W, f, E, P, g);
    M355_REQUIRE((CP == 8 || CP == 16) && C + E + P <= CP && (E == 0 || extra) && (P == 0 || pos) && mask_chan >= 0 && mask_chan < C,
                 "pool_pack_fwd: bad channel layout");
    PoolPackArgs a = {x, extra, pos, (act_t *)out, mask, M, C, H, W, E, P, CP, H / f, W / f, mask ? g : 16, mask_chan,
                      nullptr, nullptr, nullptr, 0};
    return pool_pack_launch(a, f, (hipStream_t)stream);
}

// The same assembly straight from the pieces ModelWrapper.forward concatenates (main.py:493, 503-507): x = cat(fake * alpha, alpha)
// for samples [0, Nf), cat(real, alpha) for samples [Nf, M) (M == Nf: no real half), never written to memory.  Bit-identical to
// m355_mask_cat_fwd followed by m355_pool_pack_fwd(C = 4, mask_chan = 3).
extern "C" int m355_pool_pack_parts_fwd(const float *fake, const float *real, const float *alpha, int Nf, int M, int H, int W, int f,
                                        const float *extra, int E, const float *pos, int P, void *out, int CP, float *mask, int g,
                                        void *stream)
{
    M355_REQUIRE(fake && alpha && out && Nf > 0 && (M == Nf || (M == 2 * Nf && real)) && M <= 65535, "pool_pack_parts_fwd: bad argument");
    M355_REQUIRE(m355_pool_pack_ok(4, H, W, f, E, P, mask ? g : 0) && W % 4 == 0,
                 "pool_pack_parts_fwd: unsupported shape H=%d W=%d f=%d E=%d P=%d g=%d", H, W, f, E, P, g);
    M355_REQUIRE((CP == 8 || CP == 16) && 4 + E + P <= CP && (E == 0 || extra) && (P == 0 || pos), "pool_pack_parts_fwd: bad channel layout");
    PoolPackArgs a = {nullptr, extra, pos, (act_t *)out, mask, M, 4, H, W, E, P, CP, H / f, W / f, mask ? g : 16, 3,
                      fake, real, alpha, Nf};
    return pool_pack_launch(a, f, (hipStream_t)stream);
}

extern "C" int m355_pool_unpack_bwd(const void *dh0, int f0, int cp0, const void *dh1, int f1, int cp1, const void *dh2, int f2,
                                    int cp2, float *dx, int M, int C, int H, int W, void *stream)
{
    M355_REQUIRE(dh0 && dx && M > 0 && C >= 1 && C <= 4 && H > 0 && W > 0, "pool_unpack_bwd: bad argument");
    PoolUnpackArgs a = {};
    const void *dh[3] = {dh0, dh1, dh2};
    const int f[3] = {f0, f1, f2}, cp[3] = {cp0, cp1, cp2};
    for (int k = 0; k < 3; ++k) {
        if (!dh[k]) break;
        M355_REQUIRE(f[k] >= 1 && H % f[k] == 0 && W % f[k] == 0 && cp[k] % 4 == 0, "pool_unpack_bwd: bad factor / channel stride");
        a.dh[k] = (const act_t *)dh[k];
        a.f[k] = f[k];
        a.CP[k] = cp[k];
        a.K = k + 1;
    }
    a.dx = dx; a.M = M; a.C = C; a.H = H; a.W = W; a.alpha = nullptr;
    hipLaunchKernelGGL(k_pool_unpack_bwd, dim3(grid_for((size_t)M * H * W)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("pool_unpack_bwd");
}

// adjoint of m355_pool_pack_parts_fwd with respect to `fake` (M == Nf == N; the real half never needs one): dfake [N,3,H,W] =
// (sum_k dh_k[.., c] / f_k^2) * alpha  -- m355_pool_unpack_bwd followed by m355_mask_cat_bwd, same bits
extern "C" int m355_pool_unpack_parts_bwd(const void *dh0, int f0, int cp0, const void *dh1, int f1, int cp1, const void *dh2, int f2,
                                          int cp2, const float *alpha, float *dfake, int N, int H, int W, void *stream)
{
    M355_REQUIRE(dh0 && alpha && dfake && N > 0 && H > 0 && W > 0, "pool_unpack_parts_bwd: bad argument");
    PoolUnpackArgs a = {};
    const void *dh[3] = {dh0, dh1, dh2};
    const int f[3] = {f0, f1, f2}, cp[3] = {cp0, cp1, cp2};
    for (int k = 0; k < 3; ++k) {
        if (!dh[k]) break;
        M355_REQUIRE(f[k] >= 1 && H % f[k] == 0 && W % f[k] == 0 && cp[k] % 4 == 0, "pool_unpack_parts_bwd: bad factor / channel stride");
        a.dh[k] = (const act_t *)dh[k];
        a.f[k] = f[k];
        a.CP[k] = cp[k];
        a.K = k + 1;
    }
    a.dx = dfake; a.M = N; a.C = 4; a.H = H; a.W = W; a.alpha = alpha;
    hipLaunchKernelGGL(k_pool_unpack_bwd, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("pool_unpack_parts_bwd");
}

extern "C" int m355_unpack_range(const void *g, float *out, int M, int HW, int CP, int c0, int E, void *stream)
{
    M355_REQUIRE(g && out && M > 0 && HW > 0 && c0 >= 0 && E >= 1 && c0 + E <= CP, "unpack_range: bad argument");
    const size_t total = (size_t)M * HW;
    hipLaunchKernelGGL(k_unpack_range, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const act_t *)g, out, CP, c0, E,
                       (size_t)HW, total);
    return check_launch("unpack_range");
}

extern "C" int m355_head_tail_fwd(const float *y, float *out, int N, int C, int H, int W, int flags, void *stream)
{
    M355_REQUIRE(y && out && N > 0 && C >= 1 && C <= 3 && H > 0 && W > 0, "head_tail_fwd: bad argument");
    M355_REQUIRE(!(flags & HT_SYMM) || W % 2 == 0, "head_tail_fwd: symmetrize needs an even width");
    const size_t total = (size_t)N * C * H * W;
    hipLaunchKernelGGL(k_head_tail_fwd, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y, out, C, H, W, flags, total);
    return check_launch("head_tail_fwd");
}

extern "C" int m355_head_tail_bwd(const float *dout, const float *out, void *g_nhwc8, float *dbias, float *ws, int N, int C, int H,
                                  int W, int flags, void *stream)
{
    M355_REQUIRE(dout && out && g_nhwc8 && dbias && ws && N > 0 && C >= 1 && C <= 3 && H > 0 && W > 0, "head_tail_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)N * H * W;
    const size_t gb = (total + 255) / 256;
    const unsigned nblk = (unsigned)(gb > M355_HEAD_TAIL_WS_FLOATS / 4 ? M355_HEAD_TAIL_WS_FLOATS / 4 : gb);
    hipLaunchKernelGGL(k_head_tail_bwd, dim3(nblk), dim3(256), 0, st, dout, out, (act_t *)g_nhwc8, ws, C, H, W, flags, total);
    hipLaunchKernelGGL(k_head_tail_db, dim3(1), dim3(256), 0, st, (const float *)ws, dbias, (int)nblk, C);
    return check_launch("head_tail_bwd");
}

static int fill_hinge(HingeArgs &a, int K, const float *const *p, const float *const *m, const int *hw, const float *w, int B,
                      int split, int mode, float *loss, float *msum)
{
    M355_REQUIRE(K >= 1 && K <= 3 && B > 0 && B <= 65535 && split >= 0 && split <= B && loss && msum, "hinge: bad argument");
    M355_REQUIRE(mode == 1 || split == B, "hinge: the generator loss has a single slot");
    float norm = 0.0f;
    for (int k = 0; k < K; ++k) {
        M355_REQUIRE(p[k] && hw[k] > 0, "hinge: null logits");
        a.p[k] = p[k];
        a.m[k] = m ? m[k] : nullptr;
        a.hw[k] = hw[k];
        a.w[k] = w ? w[k] : 1.0f;
        norm += w ? w[k] : 1.0f;
    }
    a.K = K; a.B = B; a.split = split; a.mode = mode; a.norm = norm; a.loss = loss; a.msum = msum;
    return 0;
}

extern "C" int m355_hinge_fwd(int K, const float *const *p, const float *const *m, const int *hw, const float *w, int B, int split,
                              int mode, float *loss2, float *msum, void *stream)
{
    HingeArgs a = {};
    if (int rc = fill_hinge(a, K, p, m, hw, w, B, split, mode, loss2, msum)) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_hinge_fwd, dim3(B, K), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_hinge_sum, dim3(1), dim3(256), 0, st, a);
    return check_launch("hinge_fwd");
}

extern "C" int m355_hinge_bwd(int K, const float *const *p, const float *const *m, const int *hw, const float *w, int B, int split,
                              int mode, const float *gl2, const float *msum, float *const *dp, void *stream)
{
    HingeArgs a = {};
    if (int rc = fill_hinge(a, K, p, m, hw, w, B, split, mode, (float *)gl2, (float *)msum)) return rc;
    M355_REQUIRE(gl2 && dp, "hinge_bwd: null pointer");
    for (int k = 0; k < K; ++k) {
        M355_REQUIRE(dp[k], "hinge_bwd: null gradient");
        a.dp[k] = dp[k];
    }
    a.gl = gl2;
    hipLaunchKernelGGL(k_hinge_bwd, dim3(B, K), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("hinge_bwd");
}
