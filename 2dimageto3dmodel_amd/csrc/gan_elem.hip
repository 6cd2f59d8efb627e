// G3 / G-bwd glue of the GAN stacks on NHWC bf16 activations (memory-bound, 16-byte vector accesses):
//   * per-channel batch statistics                     (BatchNorm2d / SynchronizedBatchNorm2d of gan.py:268-271)
//   * y = LeakyReLU(x * a[n,c] + b[n,c])                (BN normalisation + conditional affine gan.py:282-286 +
//                                                        the LeakyReLU that follows it in ResBlockUp gan.py:309-310)
//   * its backward: per-(n,c) reductions of dz and dz*x, then dx = dz*A[n,c] + x*B[c] + C[c]
//   * LeakyReLU backward + bias gradient for convs with a fused activation epilogue (discriminators)
// Reductions are two-stage and deterministic (fp32 partials per workgroup, then one finalising pass).
// HBM roofline: stats 2 B/elem, apply 4 B/elem, bwd reduce 4 B/elem, bwd apply 6 B/elem.
#include "common.h"

namespace m355 {

// The ACTIVATION element type of this translation unit.  Product build: bf16 (`act_t` = short, eight of them = one 16-byte vector).
// EXACT build (-DM355_EXACT, csrc/conv_exact.hip): fp32 -- the same kernels with `act_t` = float, the eight-element vector a
// 32-byte aggregate, and every "round to the storage type" helper the identity, so the formulas (statistics, affine, activation
// backward, projection) are the ones the product runs, minus the bf16 roundings.
// ... and `acc_t`, the type the reductions over pixels accumulate in: fp32 in the product build (the bf16 inputs are the noise), fp64
// in the EXACT build -- ATen's CPU batch-norm / sum kernels accumulate fp32 data in double (at::acc_type<float, false>), and with
// fp32 accumulators the gradients of a batch of 2 sat at 1e-3 of the reference's where fp64 gives 1e-4 (tests/test_exact_mode_gpu.py)
#ifdef M355_EXACT
typedef float act_t;
typedef double acc_t;
struct __attribute__((aligned(16))) bf16x8e {
    float v[8];
    __device__ __forceinline__ float operator[](int j) const { return v[j]; }
    __device__ __forceinline__ float &operator[](int j) { return v[j]; }
};
__device__ __forceinline__ float bf2f_e(float h) { return h; }
__device__ __forceinline__ bf16x8e pack8_e(const float (&z)[8])
{
    bf16x8e r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = z[j];
    return r;
}
__device__ __forceinline__ float f2bf_e(float f) { return f; }
#else
typedef short act_t;
typedef float acc_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8e;

__device__ __forceinline__ float bf2f_e(short h) { return __uint_as_float(((unsigned int)(unsigned short)h) << 16); }
__device__ __forceinline__ unsigned pack2_e(float lo, float hi)  // v_cvt_pk_bf16_f32: RNE, one instruction
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ bf16x8e pack8_e(const float (&z)[8])
{
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 w = {pack2_e(z[0], z[1]), pack2_e(z[2], z[3]), pack2_e(z[4], z[5]), pack2_e(z[6], z[7])};
    return __builtin_bit_cast(bf16x8e, w);
}
__device__ __forceinline__ short f2bf_e(float f)
{
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (short)(u >> 16);
}
#endif

constexpr int kMinElemsPerBlock = 65536;  // elements a workgroup reduces (more when that keeps the partial count <= 1024)

static inline int pix_per_block(size_t P, int C)
{
    size_t ppb = (P + 1023) / 1024;
    const size_t lo = (size_t)kMinElemsPerBlock / (size_t)(C > 0 ? C : 1);  // (C = 512: 128 pixels; C = 64: 1024)
    if (ppb < lo) ppb = lo;
    if (ppb < 64) ppb = 64;
    return (int)((ppb + 63) / 64 * 64);
}

// ---- generic "sum over pixels of f(row)" skeleton: rows are [C] bf16, a thread owns one 8-channel vector and
//      every (256 / (C/8))-th pixel of the workgroup's range; partials land in part[blk][NV][C].
template <int NV, typename F>
__device__ __forceinline__ void pixel_reduce(int C, size_t pix0, int npix, float *part_blk, F f)
{
    __shared__ acc_t red[256 * 8];
    const int tid = threadIdx.x;
    const int vecs = C >> 3;                 // 8-channel vectors per pixel
    const int lanes = 256 / vecs;            // pixel lanes
    const int v = tid % vecs, pl = tid / vecs;
    acc_t acc[NV][8];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0;
    if (pl < lanes) {
        // 4 independent 16-byte loads in flight per thread (the un-unrolled loop was latency bound at ~1.5 TB/s)
#pragma unroll 4
        for (int p = pl; p < npix; p += lanes) f(pix0 + p, v * 8, acc);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[k][j];
        __syncthreads();
        // thread t < C sums channel t over the pixel lanes
        for (int c = tid; c < C; c += 256) {
            const int vv = c >> 3, jj = c & 7;
            acc_t s = 0;
            for (int l = 0; l < lanes; ++l) s += red[(l * vecs + vv) * 8 + jj];
            part_blk[(size_t)k * C + c] = (float)s;
        }
    }
}

// x[P][C] -> part[blk][2][C] = (sum, sum of squares)
__global__ __launch_bounds__(256) void k_chan_stats(const act_t *__restrict__ x, float *__restrict__ part, size_t P, int C,
                                                    int ppb)
{
    const size_t pix0 = (size_t)blockIdx.x * ppb;
    const int npix = (int)min((size_t)ppb, P - pix0);
    pixel_reduce<2>(C, pix0, npix, part + (size_t)blockIdx.x * 2 * C, [&](size_t p, int c0, acc_t (&acc)[2][8]) {
        const bf16x8e v = *reinterpret_cast<const bf16x8e *>(x + p * C + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f_e(v[j]);
            acc[0][j] += f;
            acc[1][j] += f * f;
        }
    });
}

// x[P][C] -> part[blk][1][C] = sum over pixels (bias gradient of a conv whose incoming gradient is already masked)
__global__ __launch_bounds__(256) void k_chan_sum(const act_t *__restrict__ x, float *__restrict__ part, size_t P, int C, int ppb)
{
    const size_t pix0 = (size_t)blockIdx.x * ppb;
    const int npix = (int)min((size_t)ppb, P - pix0);
    pixel_reduce<1>(C, pix0, npix, part + (size_t)blockIdx.x * C, [&](size_t p, int c0, acc_t (&acc)[1][8]) {
        const bf16x8e v = *reinterpret_cast<const bf16x8e *>(x + p * C + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[0][j] += bf2f_e(v[j]);
    });
}

// part[G][nblk][W] -> out[G][W]: deterministic second stage.  A workgroup owns 32 outputs; 8 threads per output
// each sum every 8th partial, then a fixed-order LDS combine.
__global__ __launch_bounds__(256) void k_sum_partials(const float *__restrict__ part, float *__restrict__ out, int nblk,
                                                      int W)
{
    __shared__ acc_t red[8][32];
    const int g = blockIdx.y;
    const int i = blockIdx.x * 32 + (threadIdx.x & 31), l = threadIdx.x >> 5;
    acc_t s = 0;
    if (i < W)
        for (int b = l; b < nblk; b += 8) s += part[((size_t)g * nblk + b) * W + i];
    red[l][threadIdx.x & 31] = s;
    __syncthreads();
    if (l == 0 && i < W) {
        acc_t t = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
        out[(size_t)g * W + i] = (float)t;
    }
}

// y = lrelu(x * a[n,c] + b[n,c]) [+ res];  x,y [N][HW][C].  A thread keeps ONE 8-channel group for its whole loop
// (the grid stride is a multiple of C/8), so its 16 coefficients live in registers: the kernel is pure 16-byte
// streaming (the first version re-read a/b per element and ran at a third of the HBM rate).
__global__ __launch_bounds__(256) void k_affine_act(const act_t *__restrict__ x, const float *__restrict__ a,
                                                    const float *__restrict__ b, const act_t *__restrict__ res,
                                                    act_t *__restrict__ y, int HW, int C, float slope, int res_w,
                                                    float out_slope)
{
    // res_w > 0: the residual is stored at HALF resolution ([N][H/2][res_w/2][C]) and read through the nearest x2
    // upsample (a 1x1 shortcut conv commutes with it, so the shortcut runs on 4x fewer pixels -- gan.py:306-312,319)
    const int n = blockIdx.y;
    const int vecs = C >> 3;
    const size_t total = (size_t)HW * vecs;
    const act_t *xn = x + (size_t)n * HW * C;
    act_t *yn = y + (size_t)n * HW * C;
    const act_t *rn = res ? res + (size_t)n * (res_w > 0 ? HW / 4 : HW) * C : nullptr;
    const int c0 = (int)(threadIdx.x % vecs) * 8;   // (blockIdx.x * 256 + k * gridDim.x * 256) % vecs == 0
    const int lvecs = __builtin_ctz(vecs), lrw = res_w > 0 ? __builtin_ctz(res_w) : 0;
    const bool rw_pow2 = res_w > 0 && (res_w & (res_w - 1)) == 0;
    float av[8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        av[j] = a[(size_t)n * C + c0 + j];
        bv[j] = b[(size_t)n * C + c0 + j];
    }
    // FOUR independent 16-byte loads (eight with the residual) in flight per thread: the `#pragma unroll 4` this loop carried was
    // refused by the compiler (-Wpass-failed), i.e. one load per thread and trip -- 5.1 TB/s where the same access mix reaches
    // 6.2 (scripts/hbm_ceilings.py).  The stride is a multiple of C/8, so all four vectors of a thread share its coefficients.
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += 4 * stride) {
        bf16x8e v4[4], r4[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + u * stride;
            ok[u] = i < total;
            // x is read ONCE, right after the conv wrote it: a non-temporal load (streamed past the caches) runs this kernel 6.5 %
            // faster (1.22 -> 1.14 ms per cycle; non-temporal STORES of y cost 2 %, and the backward kernels' loads are indifferent:
            // profiles/r06_nt_loads_ab.txt).  The residual is read four times through the upsample: ordinary loads.
#ifndef M355_EXACT
            v4[u] = ok[u] ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8e *>(xn + i * 8)) : bf16x8e{0, 0, 0, 0, 0, 0, 0, 0};
#else
            v4[u] = ok[u] ? *reinterpret_cast<const bf16x8e *>(xn + i * 8) : bf16x8e{0, 0, 0, 0, 0, 0, 0, 0};
#endif
        }
        if (rn) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t i = i0 + u * stride;
                size_t ri = i;
                if (res_w > 0) {
                    // 32-bit shifts / one 32-bit division: the 64-bit divisions this used to be cost more than the 48 bytes moved
                    // (vecs = C / 8 divides 256: a power of two; i < HW * vecs < 2^31)
                    const unsigned iu = (unsigned)i, p = iu >> lvecs, vv = iu & (unsigned)(vecs - 1);
                    const unsigned h = rw_pow2 ? p >> lrw : p / (unsigned)res_w, w = p - h * (unsigned)res_w;
                    ri = (size_t)(((h >> 1) * ((unsigned)res_w >> 1) + (w >> 1)) * (unsigned)vecs + vv);
                }
                r4[u] = ok[u] ? *reinterpret_cast<const bf16x8e *>(rn + ri * 8) : bf16x8e{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            float zz[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float z = bf2f_e(v4[u][j]) * av[j] + bv[j];
                zz[j] = z >= 0.0f ? z : z * slope;
            }
            if (rn) {
                // residual branch of ResBlockUp (gan.py:312): the activation is rounded to bf16 first, as the
                // separate bf16 add it replaces did
                const bf16x8e q = pack8_e(zz);
#pragma unroll
                for (int j = 0; j < 8; ++j) zz[j] = bf2f_e(q[j]) + bf2f_e(r4[u][j]);
            }
            if (out_slope != 1.0f) {
                // the LeakyReLU the generator applies to a block's output in front of a head (gan.py:406,410), folded in:
                // the sum is rounded to bf16 first, as the separate activation pass it replaces saw it
                const bf16x8e q = pack8_e(zz);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = bf2f_e(q[j]);
                    zz[j] = t >= 0.0f ? t : t * out_slope;
                }
            }
            *reinterpret_cast<bf16x8e *>(yn + (i0 + u * stride) * 8) = pack8_e(zz);
        }
    }
}

// dz = dy * lrelu'(x*a+b);  part[n][blk][2][C] = (sum dz, sum dz*x) over the workgroup's pixels of sample n
__global__ __launch_bounds__(256) void k_act_bwd_reduce(const act_t *__restrict__ dy, const act_t *__restrict__ x,
                                                        const float *__restrict__ a, const float *__restrict__ b,
                                                        float *__restrict__ part, int HW, int C, float slope, int ppb)
{
    const int n = blockIdx.y, nblk = gridDim.x;
    const size_t pix0 = (size_t)blockIdx.x * ppb;
    const int npix = (int)min((size_t)ppb, (size_t)HW - pix0);
    const size_t base = (size_t)n * HW;
    const int cg = (int)(threadIdx.x % (C >> 3)) * 8;  // the 8-channel group pixel_reduce gives this thread
    float av[8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        av[j] = a[(size_t)n * C + cg + j];
        bv[j] = b[(size_t)n * C + cg + j];
    }
    pixel_reduce<2>(C, pix0, npix, part + ((size_t)n * nblk + blockIdx.x) * 2 * C,
                    [&](size_t p, int c0, acc_t (&acc)[2][8]) {
                        const bf16x8e vx = *reinterpret_cast<const bf16x8e *>(x + (base + p) * C + c0);
                        const bf16x8e vd = *reinterpret_cast<const bf16x8e *>(dy + (base + p) * C + c0);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float xf = bf2f_e(vx[j]);
                            const float z = xf * av[j] + bv[j];
                            const float dz = bf2f_e(vd[j]) * (z >= 0.0f ? 1.0f : slope);
                            acc[0][j] += dz;
                            acc[1][j] += dz * xf;
                        }
                    });
}

// dx = dz * A[n,c] + x * B[c] + Cc[c]   (coefficients of the thread's 8-channel group in registers, as above)
__global__ __launch_bounds__(256) void k_act_bwd_apply(const act_t *__restrict__ dy, const act_t *__restrict__ x,
                                                       const float *__restrict__ a, const float *__restrict__ b,
                                                       const float *__restrict__ A, const float *__restrict__ Bc,
                                                       const float *__restrict__ Cc, act_t *__restrict__ dx, int HW,
                                                       int C, float slope)
{
    const int n = blockIdx.y;
    const int vecs = C >> 3;
    const size_t total = (size_t)HW * vecs;
    const size_t off = (size_t)n * HW * C;
    const int c0 = (int)(threadIdx.x % vecs) * 8;
    float av[8], bv[8], Av[8], Bv[8], Cv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        av[j] = a[(size_t)n * C + c0 + j];
        bv[j] = b[(size_t)n * C + c0 + j];
        Av[j] = A[(size_t)n * C + c0 + j];
        Bv[j] = Bc[c0 + j];
        Cv[j] = Cc[c0 + j];
    }
    const size_t stride = (size_t)gridDim.x * 256;   // (four vectors = eight independent loads per thread and trip, as k_affine_act)
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += 4 * stride) {
        bf16x8e vx4[4], vd4[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + u * stride;
            ok[u] = i < total;
            vx4[u] = ok[u] ? *reinterpret_cast<const bf16x8e *>(x + off + i * 8) : bf16x8e{0, 0, 0, 0, 0, 0, 0, 0};
            vd4[u] = ok[u] ? *reinterpret_cast<const bf16x8e *>(dy + off + i * 8) : bf16x8e{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            float oo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xf = bf2f_e(vx4[u][j]);
                const float z = xf * av[j] + bv[j];
                const float dz = bf2f_e(vd4[u][j]) * (z >= 0.0f ? 1.0f : slope);
                oo[j] = dz * Av[j] + xf * Bv[j] + Cv[j];
            }
            *reinterpret_cast<bf16x8e *>(dx + off + (i0 + u * stride) * 8) = pack8_e(oo);
        }
    }
}

// g[P][Cp] = dy[P][C] * (y > 0 ? 1 : slope), channels C..Cp-1 zero;  part[blk][1][C] = sum over pixels of g
__global__ __launch_bounds__(256) void k_lrelu_bwd(const act_t *__restrict__ dy, const act_t *__restrict__ y,
                                                   act_t *__restrict__ g, float *__restrict__ part, size_t P, int C,
                                                   float slope, int ppb)
{
    const size_t pix0 = (size_t)blockIdx.x * ppb;
    const int npix = (int)min((size_t)ppb, P - pix0);
    pixel_reduce<1>(C, pix0, npix, part + (size_t)blockIdx.x * C, [&](size_t p, int c0, acc_t (&acc)[1][8]) {
        const bf16x8e vy = *reinterpret_cast<const bf16x8e *>(y + p * C + c0);
        const bf16x8e vd = *reinterpret_cast<const bf16x8e *>(dy + p * C + c0);
        bf16x8e o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gz = bf2f_e(vd[j]) * (bf2f_e(vy[j]) > 0.0f ? 1.0f : slope);
            o[j] = f2bf_e(gz);
            acc[0][j] += bf2f_e(o[j]);
        }
        *reinterpret_cast<bf16x8e *>(g + p * C + c0) = o;
    });
}

// NCHW fp32 image (C <= 8 planes) [+ P constant planes, e.g. the positional encoding of gan.py:9-20] -> NHWC bf16 with
// exactly 8 channels (zero filled beyond C + P): what TextureDiscriminator.forward builds with cat + permute + cast
// (gan.py:204-209) in one pass.  One thread per pixel: plane reads are coalesced along W, the store is 16 bytes.
__global__ __launch_bounds__(256) void k_pack_nhwc8(const float *__restrict__ x, const float *__restrict__ pos, act_t *__restrict__ out,
                                                    int C, int P, size_t HW, size_t total)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t n = i / HW, p = i - n * HW;
        float z[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = 0.0f;
            if (c < C) v = x[(n * C + c) * HW + p];
            else if (c - C < P) v = pos[(size_t)(c - C) * HW + p];
            z[c] = v;
        }
        *reinterpret_cast<bf16x8e *>(out + i * 8) = pack8_e(z);
    }
}

// backward of the above w.r.t. the image: d out [.., 8] bf16 -> dx NCHW fp32 (first C channels)
__global__ __launch_bounds__(256) void k_unpack_nhwc8(const act_t *__restrict__ g, float *__restrict__ dx, int C, size_t HW, size_t total)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t n = i / HW, p = i - n * HW;
        const bf16x8e v = *reinterpret_cast<const bf16x8e *>(g + i * 8);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) dx[(n * C + c) * HW + p] = bf2f_e(v[c]);
    }
}

static int check_c(int C, const char *who)
{
    if (C < 8 || C % 8 != 0 || C > 2048 || 256 % (C / 8) != 0) {
        set_error("%s: C=%d must be a multiple of 8 with C/8 dividing 256", who, C);
        return M355_ERR_BAD_ARG;
    }
    return 0;
}

// ---- projection discriminator (gan.py:104-116, 216-228): o[n,p] = sum_c feat[n,p,c] * emb[n,c] on the bf16 NHWC feature
// map (the torch path converted it to fp32 -- 67 M elements at batch 128 -- for an einsum).  One wave per pixel.
__global__ __launch_bounds__(256) void k_cproj_fwd(const act_t *__restrict__ feat, const float *__restrict__ emb, float *__restrict__ o,
                                                   int HW, int C)
{
    const int n = blockIdx.y, p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= HW) return;
    const act_t *f = feat + ((size_t)n * HW + p) * C;
    const float *e = emb + (size_t)n * C;
    float acc = 0.0f;
    for (int v = lane; v < (C >> 3); v += 64) {
        const bf16x8e x = *reinterpret_cast<const bf16x8e *>(f + v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += bf2f_e(x[j]) * e[v * 8 + j];
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    if (lane == 0) o[(size_t)n * HW + p] = acc;
}

// backward: dfeat[n,p,c] = g[n,p] * emb[n,c] (bf16), part[n][blk][c] = the workgroup's share of demb[n,c] = sum_p g[n,p] * feat[n,p,c]
// (one workgroup per sample: part IS demb; otherwise k_sum_partials adds the shares in workgroup order -- deterministic, where the
// fp32 atomics this replaces depended on the order the workgroups finished in).
// mask_slope != 1: feat is the output of a fused conv + LeakyReLU(mask_slope) whose backward is applied HERE (dfeat *= feat > 0
// ? 1 : mask_slope) -- masking is linear, so when every consumer of that activation masks its own branch of the gradient the
// producer needs no activation-backward pass over the summed gradient (TextureDiscriminator.conv4 -> conv5 + projection)
__global__ __launch_bounds__(256) void k_cproj_bwd(const act_t *__restrict__ feat, const float *__restrict__ emb,
                                                   const float *__restrict__ g, act_t *__restrict__ dfeat, float *__restrict__ part,
                                                   int HW, int C, int ppb, float mask_slope)
{
    __shared__ float red[256 * 8];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int vecs = C >> 3, lanes = 256 / vecs, v = tid % vecs, pl = tid / vecs;
    const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
    float ev[8], acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ev[j] = emb[(size_t)n * C + v * 8 + j];
        acc[j] = 0.0f;
    }
    if (pl < lanes)
        for (int p = p0 + pl; p < p1; p += lanes) {
            const float gp = g[(size_t)n * HW + p];
            const size_t off = ((size_t)n * HW + p) * C + v * 8;
            const bf16x8e x = *reinterpret_cast<const bf16x8e *>(feat + off);
            float d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = bf2f_e(x[j]);
                d[j] = gp * ev[j] * (xv > 0.0f ? 1.0f : mask_slope);
                acc[j] += gp * xv;
            }
            *reinterpret_cast<bf16x8e *>(dfeat + off) = pack8_e(d);
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[j];
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int vv = c >> 3, jj = c & 7;
        float sum = 0.0f;
        for (int l = 0; l < lanes; ++l) sum += red[(l * vecs + vv) * 8 + jj];
        part[((size_t)n * gridDim.x + blockIdx.x) * C + c] = sum;
    }
}

// ---- the discriminators' tail in ONE backward pass (round 5).  feat = LeakyReLU(conv4(.)) has two consumers, the 5x5 one-channel
// logit conv and the projection term (gan.py:110-116, 221-228); their input gradients were three passes over the 134 MB tensor (at
// batch 128): the projection's dfeat (this file), the logit conv's masked dgrad (k_conv_c8 at 48 "TF": one output channel), and
// autograd's bf16 add of the two.  Here:   dfeat[n,p,c] = mask(feat) * ( g[n,p] * emb[n,c] + sum_{a,b} dy[n, p + (a-2, b-2)] * w[c][a][b] )
// with w = the logit conv's dgrad weights (taps already flipped, W / sigma): per element 1 + 25 multiply-adds on the VECTOR unit --
// 3.4 GFLOP per launch, under the 268 MB the pass has to move anyway.  A thread owns one 8-channel vector and FOUR consecutive
// pixels of a row: the 5 x 8 weights of a tap row sit in registers for the four pixels (LDS reads per pixel / 4), dy comes from the
// sample's logit-gradient image in LDS (broadcast reads).  demb as in k_cproj_bwd.  W pad: zero or circular (wmode 0 / 2), H zero.
// the logit conv's dgrad weights as [25 taps][C] in the activation type (bf16 / EXACT: fp32): gathered ONCE per launch from the dgrad view
// (a workgroup of the main kernel copies the 25 C values coalesced; gathering them itself -- 2-byte reads 512 bytes apart -- cost
// more than its arithmetic)
__global__ __launch_bounds__(256) void k_conv5_wt(const void *__restrict__ wd, int Kp, int tap_stride, int C, act_t *__restrict__ wt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 25 * C) return;
    const int tap = i / C, c = i - tap * C;
#ifdef M355_EXACT
    wt[i] = reinterpret_cast<const float *>(wd)[(size_t)(24 - tap) * C + c];   // fp32 conv weight [1][5][5][C] (W / sigma), unflipped
#else
    wt[i] = reinterpret_cast<const short *>(wd)[(size_t)c * Kp + tap * tap_stride];   // bf16 dgrad view [row c][Kp], K = (a, b, co8), taps flipped
#endif
}

__global__ __launch_bounds__(256) void k_cproj_bwd_conv5(const act_t *__restrict__ feat, const float *__restrict__ emb,
                                                         const float *__restrict__ g, const float *__restrict__ dy5,
                                                         const act_t *__restrict__ wt, act_t *__restrict__ dfeat,
                                                         float *__restrict__ part, int H, int W, int C, int gpb, float mask_slope,
                                                         int wmode)
{
    extern __shared__ float smem[];
    const int HW = H * W;
    float *dyl = smem;                                            // [H][W]
    act_t *wl = reinterpret_cast<act_t *>(smem + HW);             // [25][C]
    __shared__ float red[256 * 8];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int vecs = C >> 3, lanes = 256 / vecs, v = tid % vecs, pl = tid / vecs;
    for (int i = tid; i < (25 * C) >> 3; i += 256)
        *reinterpret_cast<bf16x8e *>(wl + i * 8) = *reinterpret_cast<const bf16x8e *>(wt + i * 8);
    for (int i = tid; i < HW; i += 256) dyl[i] = dy5[(size_t)n * HW + i];
    float ev[8];
    acc_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ev[j] = emb[(size_t)n * C + v * 8 + j];
        acc[j] = 0;
    }
    __syncthreads();
    const int gpr = W >> 2, ngroups = H * gpr;                 // groups of four pixels along a row
    const int g0 = blockIdx.x * gpb, g1 = min(ngroups, g0 + gpb);
    if (pl < lanes)
        for (int gi = g0 + pl; gi < g1; gi += lanes) {
            const int y = gi / gpr, x0 = (gi - y * gpr) << 2;
            float o[4][8];
            bf16x8e xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = y * W + x0 + q;
                const float gp = g[(size_t)n * HW + p];
                xv[q] = *reinterpret_cast<const bf16x8e *>(feat + ((size_t)n * HW + p) * C + v * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    o[q][j] = gp * ev[j];
                    acc[j] += gp * bf2f_e(xv[q][j]);
                }
            }
            int xi[8];       // columns x0 - 2 .. x0 + 5 after the W pad rule (-1: zero padding), once per group
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                int xx = x0 - 2 + t;
                if (wmode == 2) xx = xx < 0 ? xx + W : (xx >= W ? xx - W : xx);
                xi[t] = (unsigned)xx < (unsigned)W ? xx : -1;
            }
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                const int yy = y + a - 2;
                if ((unsigned)yy >= (unsigned)H) continue;
                float dv[8];   // dy[yy][x0 - 2 .. x0 + 5]
#pragma unroll
                for (int t = 0; t < 8; ++t) dv[t] = xi[t] >= 0 ? dyl[yy * W + xi[t]] : 0.0f;
#pragma unroll
                for (int b = 0; b < 5; ++b) {
                    const bf16x8e w8 = *reinterpret_cast<const bf16x8e *>(wl + (a * 5 + b) * C + v * 8);
                    float wv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) wv[j] = bf2f_e(w8[j]);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[q][j] = fmaf(dv[q + b], wv[j], o[q][j]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[q][j] *= bf2f_e(xv[q][j]) > 0.0f ? 1.0f : mask_slope;
                *reinterpret_cast<bf16x8e *>(dfeat + ((size_t)n * HW + y * W + x0 + q) * C + v * 8) = pack8_e(o[q]);
            }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = (float)acc[j];
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int vv = c >> 3, jj = c & 7;
        acc_t sum = 0;
        for (int l = 0; l < lanes; ++l) sum += red[(l * vecs + vv) * 8 + jj];
        part[((size_t)n * gridDim.x + blockIdx.x) * C + c] = (float)sum;
    }
}

}  // namespace m355

using namespace m355;

extern "C" size_t m355_chan_reduce_ws_bytes(size_t pixels_per_group, int groups, int nvals, int C)
{
    const size_t ppb = pix_per_block(pixels_per_group, C);
    const size_t nblk = (pixels_per_group + ppb - 1) / ppb;
    return sizeof(float) * nblk * (size_t)groups * nvals * C;
}

extern "C" int m355_chan_reduce_nblk(size_t pixels_per_group, int C)
{
    const size_t ppb = pix_per_block(pixels_per_group, C);
    return (int)((pixels_per_group + ppb - 1) / ppb);
}

/* first stage only: part[nblk][2][C] (finalised by m355_bn_finalize, gan_glue.hip) */
extern "C" int m355_bn_stats_partial(const void *x, float *part, size_t P, int C, void *stream)
{
    M355_REQUIRE(x && part && P > 0, "bn_stats_partial: null pointer / empty");
    if (int rc = check_c(C, "bn_stats_partial")) return rc;
    const int ppb = pix_per_block(P, C);
    const int nblk = (int)((P + ppb - 1) / ppb);
    hipLaunchKernelGGL(k_chan_stats, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const act_t *)x, part, P, C, ppb);
    return check_launch("bn_stats_partial");
}

/* first stage only: part[N][nblk][2][C] (finalised by m355_bn_bwd_finalize) */
extern "C" int m355_affine_act_bwd_partial(const void *dy, const void *x, const float *a, const float *b, float *part, int N,
                                           int HW, int C, float slope, void *stream)
{
    M355_REQUIRE(dy && x && a && b && part && N > 0 && HW > 0, "affine_act_bwd_partial: null pointer / empty");
    if (int rc = check_c(C, "affine_act_bwd_partial")) return rc;
    const int ppb = pix_per_block((size_t)HW, C);
    const int nblk = (HW + ppb - 1) / ppb;
    hipLaunchKernelGGL(k_act_bwd_reduce, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, (const act_t *)dy, (const act_t *)x,
                       a, b, part, HW, C, slope, ppb);
    return check_launch("affine_act_bwd_partial");
}

extern "C" int m355_chan_sum(const void *x, float *sums /*[C]*/, void *ws, size_t P, int C, void *stream)
{
    M355_REQUIRE(x && sums && ws && P > 0, "chan_sum: null pointer / empty");
    if (int rc = check_c(C, "chan_sum")) return rc;
    const int ppb = pix_per_block(P, C);
    const int nblk = (int)((P + ppb - 1) / ppb);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_chan_sum, dim3(nblk), dim3(256), 0, st, (const act_t *)x, (float *)ws, P, C, ppb);
    hipLaunchKernelGGL(k_sum_partials, dim3((C + 31) / 32, 1), dim3(256), 0, st, (const float *)ws, sums, nblk, C);
    return check_launch("chan_sum");
}

extern "C" int m355_bn_stats(const void *x, float *sums /*[2][C]*/, void *ws, size_t P, int C, void *stream)
{
    M355_REQUIRE(x && sums && ws && P > 0, "bn_stats: null pointer / empty");
    if (int rc = check_c(C, "bn_stats")) return rc;
    const int ppb = pix_per_block(P, C);
    const int nblk = (int)((P + ppb - 1) / ppb);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_chan_stats, dim3(nblk), dim3(256), 0, st, (const act_t *)x, (float *)ws, P, C, ppb);
    hipLaunchKernelGGL(k_sum_partials, dim3((2 * C + 31) / 32, 1), dim3(256), 0, st, (const float *)ws, sums, nblk, 2 * C);
    return check_launch("bn_stats");
}

extern "C" int m355_affine_act_fwd(const void *x, const float *a, const float *b, const void *res, int res_w, void *y, int N,
                                   int HW, int C, float slope, float out_slope, void *stream)
{
    M355_REQUIRE(x && a && b && y && N > 0 && HW > 0, "affine_act_fwd: null pointer / empty");
    M355_REQUIRE(res_w >= 0 && (res_w == 0 || (res_w % 2 == 0 && HW % res_w == 0 && (HW / res_w) % 2 == 0)),
                 "affine_act_fwd: res_w=%d does not describe an even H x W = %d image", res_w, HW);
    if (int rc = check_c(C, "affine_act_fwd")) return rc;
    const size_t total = (size_t)HW * (C / 8);
    // ~8 vectors per thread (its coefficients are loaded once), still thousands of workgroups with N in grid.y
    const unsigned gx = (unsigned)min((size_t)4096, (total + 2047) / 2048);
    hipLaunchKernelGGL(k_affine_act, dim3(gx, N), dim3(256), 0, (hipStream_t)stream, (const act_t *)x, a, b,
                       (const act_t *)res, (act_t *)y, HW, C, slope, res ? res_w : 0, out_slope);
    return check_launch("affine_act_fwd");
}

extern "C" int m355_affine_act_bwd_reduce(const void *dy, const void *x, const float *a, const float *b,
                                          float *sums /*[N][2][C]*/, void *ws, int N, int HW, int C, float slope,
                                          void *stream)
{
    M355_REQUIRE(dy && x && a && b && sums && ws && N > 0 && HW > 0, "affine_act_bwd_reduce: null pointer / empty");
    if (int rc = check_c(C, "affine_act_bwd_reduce")) return rc;
    const int ppb = pix_per_block((size_t)HW, C);
    const int nblk = (HW + ppb - 1) / ppb;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_act_bwd_reduce, dim3(nblk, N), dim3(256), 0, st, (const act_t *)dy, (const act_t *)x, a, b,
                       (float *)ws, HW, C, slope, ppb);
    hipLaunchKernelGGL(k_sum_partials, dim3((2 * C + 31) / 32, N), dim3(256), 0, st, (const float *)ws, sums, nblk, 2 * C);
    return check_launch("affine_act_bwd_reduce");
}

extern "C" int m355_affine_act_bwd_apply(const void *dy, const void *x, const float *a, const float *b, const float *A,
                                         const float *Bc, const float *Cc, void *dx, int N, int HW, int C, float slope,
                                         void *stream)
{
    M355_REQUIRE(dy && x && a && b && A && Bc && Cc && dx && N > 0 && HW > 0, "affine_act_bwd_apply: null pointer / empty");
    if (int rc = check_c(C, "affine_act_bwd_apply")) return rc;
    const size_t total = (size_t)HW * (C / 8);
    // ~8 vectors per thread (its coefficients are loaded once), still thousands of workgroups with N in grid.y
    const unsigned gx = (unsigned)min((size_t)4096, (total + 2047) / 2048);
    hipLaunchKernelGGL(k_act_bwd_apply, dim3(gx, N), dim3(256), 0, (hipStream_t)stream, (const act_t *)dy, (const act_t *)x,
                       a, b, A, Bc, Cc, (act_t *)dx, HW, C, slope);
    return check_launch("affine_act_bwd_apply");
}

extern "C" int m355_lrelu_bwd(const void *dy, const void *y, void *g, float *dbias /*[C]*/, void *ws, size_t P, int C,
                              float slope, void *stream)
{
    M355_REQUIRE(dy && y && g && dbias && ws && P > 0, "lrelu_bwd: null pointer / empty");
    if (int rc = check_c(C, "lrelu_bwd")) return rc;
    const int ppb = pix_per_block(P, C);
    const int nblk = (int)((P + ppb - 1) / ppb);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_lrelu_bwd, dim3(nblk), dim3(256), 0, st, (const act_t *)dy, (const act_t *)y, (act_t *)g,
                       (float *)ws, P, C, slope, ppb);
    hipLaunchKernelGGL(k_sum_partials, dim3((C + 31) / 32, 1), dim3(256), 0, st, (const float *)ws, dbias, nblk, C);
    return check_launch("lrelu_bwd");
}

extern "C" int m355_pack_nhwc8(const float *x_nchw, const float *pos, void *out_nhwc8, int N, int C, int P, int H, int W,
                               void *stream)
{
    M355_REQUIRE(x_nchw && out_nhwc8 && N > 0 && C >= 1 && C <= 8 && P >= 0 && C + P <= 8 && (P == 0 || pos) && H > 0 && W > 0,
                 "pack_nhwc8: bad argument");
    const size_t HW = (size_t)H * W, total = (size_t)N * HW;
    const unsigned g = (unsigned)min((size_t)65535, (total + 255) / 256);
    hipLaunchKernelGGL(k_pack_nhwc8, dim3(g), dim3(256), 0, (hipStream_t)stream, x_nchw, pos, (act_t *)out_nhwc8, C, P, HW, total);
    return check_launch("pack_nhwc8");
}

extern "C" int m355_unpack_nhwc8(const void *g_nhwc8, float *dx_nchw, int N, int C, int H, int W, void *stream)
{
    M355_REQUIRE(g_nhwc8 && dx_nchw && N > 0 && C >= 1 && C <= 8 && H > 0 && W > 0, "unpack_nhwc8: bad argument");
    const size_t HW = (size_t)H * W, total = (size_t)N * HW;
    const unsigned g = (unsigned)min((size_t)65535, (total + 255) / 256);
    hipLaunchKernelGGL(k_unpack_nhwc8, dim3(g), dim3(256), 0, (hipStream_t)stream, (const act_t *)g_nhwc8, dx_nchw, C, HW, total);
    return check_launch("unpack_nhwc8");
}

extern "C" int m355_cproj_fwd(const void *feat, const float *emb, float *out /*[N,HW]*/, int N, int HW, int C, void *stream)
{
    M355_REQUIRE(feat && emb && out && N > 0 && HW > 0 && N <= 65535, "cproj_fwd: bad argument");
    if (int rc = check_c(C, "cproj_fwd")) return rc;
    hipLaunchKernelGGL(k_cproj_fwd, dim3((HW + 3) / 4, N), dim3(256), 0, (hipStream_t)stream, (const act_t *)feat, emb, out, HW, C);
    return check_launch("cproj_fwd");
}

/* floats of scratch m355_cproj_bwd needs (0: one workgroup per sample, demb is written directly) */
extern "C" size_t m355_cproj_bwd_ws_floats(int N, int HW, int C)
{
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    const int ppb = pix_per_block((size_t)HW, C), nblk = (HW + ppb - 1) / ppb;
    return nblk > 1 ? (size_t)N * nblk * C : 0;
}

extern "C" int m355_cproj_bwd(const void *feat, const float *emb, const float *g /*[N,HW]*/, void *dfeat, float *demb /*[N,C]*/,
                              float *ws, int N, int HW, int C, float mask_slope, void *stream)
{
    M355_REQUIRE(feat && emb && g && dfeat && demb && N > 0 && HW > 0 && N <= 65535, "cproj_bwd: bad argument");
    if (int rc = check_c(C, "cproj_bwd")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int ppb = pix_per_block((size_t)HW, C), nblk = (HW + ppb - 1) / ppb;
    M355_REQUIRE(nblk == 1 || ws, "cproj_bwd: workspace required (m355_cproj_bwd_ws_floats)");
    hipLaunchKernelGGL(k_cproj_bwd, dim3(nblk, N), dim3(256), 0, st, (const act_t *)feat, emb, g, (act_t *)dfeat, nblk == 1 ? demb : ws,
                       HW, C, ppb, mask_slope);
    if (nblk > 1) hipLaunchKernelGGL(k_sum_partials, dim3((C + 31) / 32, N), dim3(256), 0, st, (const float *)ws, demb, nblk, C);
    return check_launch("cproj_bwd");
}

/* The tail of a discriminator in one backward pass (models/gan.py:110-116, 221-228: feat -> 5x5 one-channel logit conv + projection term):
 * dfeat = mask(feat) * ( g_proj[n,p] * emb[n,c] + (5x5 "same" conv of the logit gradient dy5 [N,H,W] with the logit conv's dgrad weights) ),
 * demb as m355_cproj_bwd.  w_dgrad = that conv's dgrad view (m355_conv2d_weight_prep; Kp = its row length) -- in the EXACT build its fp32
 * weight array.  pad_w_mode 0 (zero) or 2 (circular); H zero padded.  m355_cproj_bwd_conv5_ok: C % 8 == 0, C <= 512 with C / 8 dividing
 * 256, W % 4 == 0, H W floats + 25 C weights of LDS within the budget.  ws: m355_cproj_bwd_conv5_ws_floats(N, H, W, C) floats (0: none needed). */
static int conv5_gpb(int N, int H, int W)   // groups of four pixels per workgroup: ~2048 workgroups per launch, at least 16 groups each
{
    const int ng = H * (W / 4);
    int nblk = N > 0 ? 2048 / N : 1;   // (38 KB of LDS per workgroup: four resident per CU)
    if (nblk < 1) nblk = 1;
    int gpb = (ng + nblk - 1) / nblk;
    if (gpb < 16) gpb = 16;
    return gpb;
}
static size_t conv5_lds_bytes(int H, int W, int C) { return sizeof(float) * (size_t)H * W + sizeof(act_t) * (size_t)25 * C; }
extern "C" int m355_cproj_bwd_conv5_ok(int H, int W, int C)
{
    // (LDS: 25 x C weights + the H x W logit-gradient image + the 8 KB reduction buffer within 64 KB)
    return (C >= 8 && C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0 && W >= 4 && W % 4 == 0 && H >= 1 && (H * W) % 2 == 0 &&
            conv5_lds_bytes(H, W, C) + 8192 <= 65536) ? 1 : 0;
}
extern "C" size_t m355_cproj_bwd_conv5_ws_floats(int N, int H, int W, int C)
{
    if (!m355_cproj_bwd_conv5_ok(H, W, C) || N <= 0) return 0;
    const int ng = H * (W / 4), gpb = conv5_gpb(N, H, W), nblk = (ng + gpb - 1) / gpb;
    // the [25][C] weight table (in activation elements, rounded up to whole floats) + the per-workgroup shares of demb
    return (size_t)(25 * C * sizeof(act_t) + 3) / 4 + (nblk > 1 ? (size_t)N * nblk * C : 0);
}
extern "C" int m355_cproj_bwd_conv5(const void *feat, const float *emb, const float *g /*[N,HW]*/, const float *dy5 /*[N,H,W]*/,
                                    const void *w_dgrad, int Kp, void *dfeat, float *demb /*[N,C]*/, float *ws, int N, int H, int W,
                                    int C, float mask_slope, int pad_w_mode, void *stream)
{
    M355_REQUIRE(feat && emb && g && dy5 && w_dgrad && dfeat && demb && N > 0 && N <= 65535, "cproj_bwd_conv5: bad argument");
    M355_REQUIRE(m355_cproj_bwd_conv5_ok(H, W, C) && (pad_w_mode == 0 || pad_w_mode == 2) && Kp >= 200,
                 "cproj_bwd_conv5: shape not eligible (m355_cproj_bwd_conv5_ok)");
    // the layout k_conv5_wt reads -- row length Kp, tap stride dy_channels(1) -- is the conv library's private rule: ask it, do not assume it
    m355_conv_desc d5 = {N, H, W, C, 1, 5, 5, 1, 2, 2, pad_w_mode, 0};
    m355_conv_plan p5;
    M355_REQUIRE(m355_conv2d_plan(&d5, &p5) == M355_OK && p5.w_dgrad_row_elems == Kp,
                 "cproj_bwd_conv5: Kp is not the row length of this logit conv's dgrad view (m355_conv_plan.w_dgrad_row_elems)");
    hipStream_t st = (hipStream_t)stream;
    const int ng = H * (W / 4), gpb = conv5_gpb(N, H, W), nblk = (ng + gpb - 1) / gpb;
    M355_REQUIRE(ws, "cproj_bwd_conv5: workspace required (m355_cproj_bwd_conv5_ws_floats)");
    act_t *wt = reinterpret_cast<act_t *>(ws);
    float *part = ws + (25 * C * sizeof(act_t) + 3) / 4;
    hipLaunchKernelGGL(k_conv5_wt, dim3((25 * C + 255) / 256), dim3(256), 0, st, w_dgrad, Kp, p5.dy_channels, C, wt);
    hipLaunchKernelGGL(k_cproj_bwd_conv5, dim3(nblk, N), dim3(256), conv5_lds_bytes(H, W, C), st, (const act_t *)feat, emb, g, dy5,
                       (const act_t *)wt, (act_t *)dfeat, nblk == 1 ? demb : part, H, W, C, gpb, mask_slope, pad_w_mode);
    if (nblk > 1) hipLaunchKernelGGL(k_sum_partials, dim3((C + 31) / 32, N), dim3(256), 0, st, (const float *)part, demb, nblk, C);
    return check_launch("cproj_bwd_conv5");
}
