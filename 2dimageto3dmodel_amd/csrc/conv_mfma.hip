// G: conv2d of the GAN stacks as a bf16 MFMA implicit GEMM (fp32 accumulate) for gfx950.
//
// Replaces F.conv2d behind the nn.Conv2d layers of models/gan.py (ResBlockUp convs gan.py:294-302, heads
// gan.py:359,364, TextureDiscriminator convs gan.py:163-177, MeshDiscriminator convs gan.py:57-65) together with
// the pads the reference materialises before them (F.pad replicate gan.py:329, circpad rendering/utils.py:60-64)
// and the nearest x2 upsample in front of a block (gan.py:319): all three are index arithmetic in the A-tile
// loader here, no padded / upsampled tensor is ever written.
//
// GEMM view: M = N*Ho*Wo output pixels, Ncols = Cout, K = KH*KW*Cin ordered (kh,kw,ci).
//   activations NHWC bf16 (K runs along the contiguous channel axis), weights [Cout_p][KH][KW][Cin] bf16.
// Workgroup tile 256(M) x 64(N), K step 32; 4 waves, each 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 accumulators.
// A (gathered pixels) and B (weights) are staged global -> VGPR -> LDS (row pitch 80 B: ds_read_b128 fragment
// reads are bank-conflict free), double buffered; loads of step s+1 are issued before the MFMAs of step s.
//
// The same kernel computes the data gradient: dgrad of a stride-1 conv is a forward conv of dy with flipped,
// transposed weights; dgrad of a stride-2 conv splits into 4 output-parity classes, each a 2x2 forward conv
// written with output stride 2 (k_weight_prep builds those weight views).  Gradients are produced in the padded
// (and upsampled) input frame and folded back by k_fold_pad (W pad mode, H crop, 2x2 upsample sum).
#include <hip/hip_bf16.h>

#include <stdlib.h>
#include <string.h>

#include <cstring>
#include "conv_dma.h"

namespace m355 {

constexpr int BM = 256, BN = 64, BK = 32;
constexpr int LDP = BK + 8;  // LDS row pitch in bf16 (80 bytes)


__global__ __launch_bounds__(256) void k_conv_mfma(ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned short As[2][BM * LDP];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][BN * LDP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int M = a.N * a.Ho * a.Wo;
    const int kg = tid & 3;  // which 8-channel granule of the 32-wide K step this thread stages

    // ---- the 4 A rows (pixels) this thread stages: rows (tid>>2) + 64*i
    int hi0[4], wi0[4];
    const unsigned short *xb[4];
    bool rok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (tid >> 2) + 64 * i;
        rok[i] = m < M;
        const int mm = rok[i] ? m : 0;
        const int n = mm / (a.Ho * a.Wo), r = mm - n * (a.Ho * a.Wo);
        const int ho = r / a.Wo, wo = r - ho * a.Wo;
        hi0[i] = ho * a.stride - a.pad_h;
        wi0[i] = wo * a.stride - a.pad_w;
        xb[i] = a.x + (size_t)n * a.H * a.W * a.Cin;
    }
    // B row (output channel) this thread stages
    const unsigned short *wb = a.w + (size_t)(n0 + (tid >> 2)) * a.Kp + kg * 8;

    // K is the flattened (kh, kw, ci) axis; one granule = 8 consecutive channels of one tap, so Cin only has to be a
    // multiple of 8 (D conv1: 8 channels -> a 32-wide K step spans 4 taps).  The weight rows are zero beyond K.
    const int Ktot = a.KH * a.KW * a.Cin;
    const int nsteps = a.Kp / BK;

    bf16x8 ra[4], rb;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_step = [&](int s) {
        const int k = s * BK + kg * 8;
        const int tap = k / a.Cin, c0 = k - tap * a.Cin;
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const bool kok = k < Ktot;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hi = hi0[i] + kh;
            int wi = wi0[i] + kw;
            bool ok = kok && rok[i] && hi >= 0 && hi < a.Hl;
            if (a.pad_w_mode == 1) wi = min(max(wi, 0), a.Wl - 1);               // replicate (gan.py:329)
            else if (a.pad_w_mode == 2) wi = wi < 0 ? wi + a.Wl : (wi >= a.Wl ? wi - a.Wl : wi);  // circpad
            else ok = ok && wi >= 0 && wi < a.Wl;
            const unsigned short *p = xb[i] + ((size_t)(hi >> a.ups) * a.W + (wi >> a.ups)) * a.Cin + c0;
            ra[i] = ok ? *reinterpret_cast<const bf16x8 *>(p) : zero8;
        }
        rb = *reinterpret_cast<const bf16x8 *>(wb + (size_t)s * BK);
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<bf16x8 *>(&As[buf][((tid >> 2) + 64 * i) * LDP + kg * 8]) = ra[i];
        *reinterpret_cast<bf16x8 *>(&Bs[buf][(tid >> 2) * LDP + kg * 8]) = rb;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    load_step(0);
    store_step(0);
    __syncthreads();
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load_step(s + 1);  // global loads in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(&As[buf][(wave * 64 + frow) * LDP + kk + fk]);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(&As[buf][(wave * 64 + 32 + frow) * LDP + kk + fk]);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(&Bs[buf][frow * LDP + kk + fk]);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(&Bs[buf][(32 + frow) * LDP + kk + fk]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (s + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= M) continue;
            const int n = m / (a.Ho * a.Wo), rr = m - n * (a.Ho * a.Wo);
            const int ho = rr / a.Wo, wo = rr - ho * a.Wo;
            const int oh = ho * a.oy_mul + a.oy_off, ow = wo * a.ox_mul + a.ox_off;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = n0 + 32 * j + (lane & 31);
                if (co >= a.Cout) continue;
                float v = acc[i][j][r];
                if (a.bias) v += a.bias[co];
                v = v >= 0.0f ? v : v * a.slope;
                if (a.y_f32_nchw)
                    reinterpret_cast<float *>(a.y)[(((size_t)n * a.Cout + co) * a.OH + oh) * a.OW + ow] = v;
                else
                    reinterpret_cast<unsigned short *>(a.y)[(((size_t)n * a.OH + oh) * a.OW + ow) * a.Cs + co] = f2bf(v);
            }
        }
    }
}

// =====================================================================================================
// k_conv_glds: the main-line implicit GEMM.  Same GEMM view as k_conv_mfma, restructured around the LDS-DMA
// (`buffer_load_dwordx4 ... lds`): no staging VGPRs, no ds_write pass, K step 64.
//   * A (gathered pixels) and B (weight rows) tiles are LDS images of 128-byte rows (64 bf16 of K).  One
//     wave-instruction fills 8 rows x 128 B; lane l lands at row l>>3, 16-byte slot l&7 (the DMA destination is
//     lane-linear), and FETCHES source chunk (l&7) ^ ((row>>1)&7): the XOR swizzle lives on the source address
//     and on the fragment read, never on the destination.  Fragment reads (ds_read_b128, rows = lane&31) are
//     then bank-conflict free.
//   * Padding (zero H pad, zero W pad, rows beyond M, K beyond taps*Cin) is an out-of-range buffer offset:
//     the DMA writes zeros for those lanes (verified on gfx950: scripts/probes/blds_oob.hip).
//   * Operands are swapped in the MFMA (A operand = weight rows, B operand = pixels), so a lane's 16 accumulator
//     registers are 4 x 4 consecutive output CHANNELS of one pixel: the epilogue packs them to bf16, exchanges
//     halves with v_permlane32_swap and issues 16-byte NHWC stores.
//   * Workgroup tile BM x BN = 128x128 (2x2 waves) or 256x64 (4x1 waves), each wave 64x64 = 2x2 MFMA 32x32x16;
//     2 LDS stages (64 / 80 KiB) -> 2 workgroups per CU; the DMA of step t+1 is issued before the MFMAs of step t.
//   * blockIdx is remapped so that each XCD (private L2) owns a contiguous run of pixel tiles: neighbouring tiles
//     share input halo rows and all tiles share the weights.
// FAST (Cin % 64 == 0): a K step is one tap x 64 channels -> tap arithmetic is scalar.  Otherwise (Cin % 8 == 0,
// D conv1's 8 channels, dgrad of the 1/3-channel heads) every 16-byte chunk derives its own tap.
// NST: LDS stages.  2 = the DMA of step t+1 is issued before the MFMAs of step t and a __syncthreads() per step drains it (two
// workgroups per CU cover each other's waits).  The 64 x 64 tiles exist for problems too small to give every CU two
// workgroups (the generator's 8x4 .. 32x16 stages, the mesh discriminator: one workgroup per CU or fewer), where each step then
// exposes a full DMA round trip (~1500 cycles for 128 cycles of MFMA): they run NST = 4 -- three steps in flight, a counted
// s_waitcnt vmcnt and one raw barrier per step.
template <int N>
__device__ __forceinline__ void glds_wait()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
#ifndef M355_GLDS_SMALL_NST
#define M355_GLDS_SMALL_NST 4
#endif
// SK (round 4): split-K for problems with fewer 128 x 128 tiles than CUs (the generator's 8x4 / 16x8 stages: M = 2048 .. 8192
// pixels against K = 2304 .. 4608).  They used to run 64 x 64 tiles -- one workgroup per CU, each pulling its own copy of the
// operands through L2 -> LDS: 302 MB per launch for M 2048 x N 512 x K 4608, which at the ~8.4 TB/s that path gives IS the 36 us
// the launch took.  128 x 128 tiles halve the bytes; blockIdx.y = K slice keeps every CU busy; the slices' fp32 accumulators
// are stored raw and a finishing pass adds them in slice order (deterministic), converts to bf16 and emits the batch-norm
// partial sums -- i.e. it REPLACES the bn_stats_partial pass these layers needed anyway.  Measured per launch incl. that pass
// (profiles/r04_splitk.txt): blk1's convs 36 + 9 -> 33 us, blk2.conv1 43 + 9 -> 46, blk2.conv2 27 + 9 -> 34.  Where no pass is
// replaced the extra launch costs more than the tiles save (mesh discriminator 128 -> 256 at 16x16: 27 -> 37 us; the same on the
// padded-frame dgrads, 41 -> 43 us), so only the forward convs that feed a batch norm take this path.
template <int BM, int BN, int NW, int WGN, bool FAST, int MODE, int NST = 2, bool SK = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 2) void k_conv_glds(ConvArgs a, unsigned xbytes, unsigned wbytes)
{
    static_assert(!SK || (FAST && NST == 2), "split-K: 64-channel K steps, two stages");
    // NW waves arranged (NW / WGN) x WGN over the BM x BN tile; each wave owns a WTM x WTN sub-tile
    constexpr int WGM = NW / WGN, WTM = BM / WGM, WTN = BN / WGN, PI = WTM / 32, CJ = WTN / 32;
    constexpr int RA = BM / (8 * NW), RB = BN / (8 * NW);  // DMA instructions per wave per stage
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && RA >= 1 && RB >= 1 && NW % 2 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a.N * a.Ho * a.Wo, HW = a.Ho * a.Wo;

    // ---- XCD-aware tile id (bijective for any grid size): XCD x gets the tiles [start_x, start_x + count_x)
    const int nwg = gridDim.x, nN = a.CoutP / BN;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int sid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    const int tn = sid % nN, tm = sid / nN;
    const int m0 = tm * BM, n0 = tn * BN;

    int pad_h = a.pad_h, pad_w = a.pad_w, oy_off = a.oy_off, ox_off = a.ox_off;
    const unsigned short *wv = a.w;
    if (!SK && a.ncls > 1) {
        const int cls = blockIdx.y;
        pad_h = a.cpad_h[cls]; pad_w = a.cpad_w[cls]; oy_off = a.coy[cls]; ox_off = a.cox[cls];
        wv += (size_t)cls * a.cls_w_elems;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wv, 0, wbytes, 0x00020000);

    // ---- per-lane staging roles.  DMA instruction i of wave w covers tile rows 8*(NW*i+w) .. +7, so that the
    // swizzle term (row>>1)&7 = 4*(w&1) + (lane>>4) is the same for all of a lane's rows (NW is even).
    const int csrc = (lane & 7) ^ (((wave & 1) << 2) | ((lane >> 4) & 3));  // source chunk of this lane's LDS slot
    int hi0[RA], wi0[RA];
    unsigned nb[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + 8 * (NW * i + wave) + (lane >> 3);
        if (m < M) {
            int n, ho, wo;
            if (a.lgWo >= 0) {
                wo = m & (a.Wo - 1); ho = (m >> a.lgWo) & (a.Ho - 1); n = m >> (a.lgWo + a.lgHo);
            } else {
                n = m / HW;
                const int rr = m - n * HW;
                ho = rr / a.Wo; wo = rr - ho * a.Wo;
            }
            hi0[i] = ho * a.stride - pad_h;
            wi0[i] = wo * a.stride - pad_w;
            nb[i] = (unsigned)n * (unsigned)(a.H * a.W * a.Cin * 2) + (FAST ? csrc * 16 : 0);
        } else {
            hi0[i] = -(1 << 20);  // never inside [0, Hl): the row stays zero
            wi0[i] = 0;
            nb[i] = 0;
        }
    }
    unsigned wrow[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) wrow[j] = (unsigned)(n0 + 8 * (NW * j + wave) + (lane >> 3)) * (unsigned)(a.Kp * 2) + csrc * 16;

    const int Cin2 = a.Cin * 2, Ktot = a.KH * a.KW * a.Cin;
    const int cpt = a.Cin >> 6;       // FAST: 64-channel chunks per tap
    int s_kh = 0, s_kw = 0, s_cc = 0;  // FAST: scalar tap cursor of the NEXT stage() call
    int t_beg = 0, nsteps = a.Kp / 64; // K steps [t_beg, nsteps) of this workgroup
    if constexpr (SK) {
        const int per = (nsteps + a.sk - 1) / a.sk;
        t_beg = (int)blockIdx.y * per;
        nsteps = min(nsteps, t_beg + per);
        const int tap = t_beg / cpt;
        s_cc = t_beg - tap * cpt;
        s_kh = tap / a.KW;
        s_kw = tap - s_kh * a.KW;
    }

    auto stage = [&](int t, int buf) {
        unsigned char *dstA = lds + buf * STAGE + wave * 1024, *dstB = dstA + BM * 128;
        int kh, kw;
        unsigned cb;
        bool kok = true;
        if (FAST) {
            kh = s_kh; kw = s_kw; cb = (unsigned)s_cc * 128u;
            kok = kh < a.KH;  // zero K steps beyond the last tap (Kp rounding) never occur when Cin % 64 == 0
            if (++s_cc == cpt) { s_cc = 0; if (++s_kw == a.KW) { s_kw = 0; ++s_kh; } }
        } else {
            const int k = (t * 8 + csrc) * 8;
            const int tap = k / a.Cin;
            kh = tap / a.KW; kw = tap - kh * a.KW;
            cb = (unsigned)(k - tap * a.Cin) * 2u;
            kok = k < Ktot;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int hi = hi0[i] + kh;
            int wi = wi0[i] + kw;
            bool ok = kok && (unsigned)hi < (unsigned)a.Hl;
            if (MODE == 1) wi = min(max(wi, 0), a.Wl - 1);
            else if (MODE == 2) wi = wi < 0 ? wi + a.Wl : (wi >= a.Wl ? wi - a.Wl : wi);
            else ok = ok && (unsigned)wi < (unsigned)a.Wl;
            unsigned off = nb[i] + (unsigned)(((hi >> a.ups) * a.W + (wi >> a.ups)) * Cin2);
            if (!FAST) off += cb;
            dma16(rx, dstA + i * (NW * 1024), ok ? off : OOB, FAST ? cb : 0u);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) dma16(rw, dstB + j * (NW * 1024), wrow[j], (unsigned)t * 128u);
    };

    f32x16 acc[CJ][PI];  // [co block j][pixel block i]
#pragma unroll
    for (int j = 0; j < CJ; ++j)
#pragma unroll
        for (int i = 0; i < PI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

    const int wm = wave / WGN, wn = wave % WGN;
    const int swz = (lane >> 1) & 7, half = lane >> 5;
    const unsigned char *fa = lds + (wm * WTM + (lane & 31)) * 128;             // pixel rows of this wave
    const unsigned char *fb = lds + BM * 128 + (wn * WTN + (lane & 31)) * 128;  // weight rows of this wave

    auto step_mma = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int co16 = ((kk * 2 + half) ^ swz) << 4;
            bf16x8 pf[PI], wf[CJ];
#pragma unroll
            for (int i = 0; i < PI; ++i) pf[i] = *reinterpret_cast<const bf16x8 *>(fa + buf * STAGE + i * 32 * 128 + co16);
#pragma unroll
            for (int j = 0; j < CJ; ++j) wf[j] = *reinterpret_cast<const bf16x8 *>(fb + buf * STAGE + j * 32 * 128 + co16);
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int i = 0; i < PI; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], pf[i], acc[j][i], 0, 0, 0);
        }
    };
    if constexpr (NST > 2) {
        static_assert(NST == 4, "counted waits are written out for four stages");
        constexpr int PER = RA + RB;   // DMAs one wave issues per step
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
            if (s < nsteps) stage(s, s);
        int buf = 0, nbuf = NST - 1;   // buffer of step t / the one step t + NST - 1 goes to (= that of step t - 1)
        for (int t = 0; t < nsteps; ++t) {   // (never with SK: t_beg = 0)
            // step t's DMAs were issued NST-1 steps ago; those of the (up to) NST-2 following steps may stay in flight.
            // The barrier: every wave's part of step t has landed, and every wave is done reading the buffer refilled next
            const int younger = nsteps - 1 - t;
            if (younger >= 2) glds_wait<2 * PER>();
            else if (younger == 1) glds_wait<PER>();
            else glds_wait<0>();
            __builtin_amdgcn_s_barrier();
            if (t + NST - 1 < nsteps) stage(t + NST - 1, nbuf);
            step_mma(buf);
            buf = buf == NST - 1 ? 0 : buf + 1;
            nbuf = nbuf == NST - 1 ? 0 : nbuf + 1;
        }
    } else {
        if (t_beg < nsteps) stage(t_beg, 0);
        __syncthreads();  // (the barrier's fence drains the DMA: vmcnt(0))
        for (int t = t_beg; t < nsteps; ++t) {
            const int buf = (t - t_beg) & 1;
            if (t + 1 < nsteps) stage(t + 1, buf ^ 1);
            step_mma(buf);
            __syncthreads();
        }
    }
    if constexpr (SK) {
        // raw fp32 accumulators of this K slice: skws[slice][m][CoutP]; lane: pixel m, 4 x 4 consecutive channels per (j, i)
        float *wsb = a.skws + (size_t)blockIdx.y * M * a.CoutP;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int m = m0 + wm * WTM + 32 * i + (lane & 31);
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(wsb + (size_t)m * a.CoutP + n0 + wn * WTN + 32 * j + 8 * g + 4 * half) =
                        make_float4(acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
        }
        return;
    }

    // ---- epilogue.  acc[j][i][r]: channel n0 + wn*WTN + 32j + 8*(r>>2) + 4*half + (r&3), pixel m0 + wm*WTM + 32i + (lane&31)
    unsigned short *yb = reinterpret_cast<unsigned short *>(a.y);
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int m = m0 + wm * WTM + 32 * i + (lane & 31);
        const bool mok = m < M;
        const int mm = mok ? m : 0;
        int n, ho, wo;
        if (a.lgWo >= 0) {
            wo = mm & (a.Wo - 1); ho = (mm >> a.lgWo) & (a.Ho - 1); n = mm >> (a.lgWo + a.lgHo);
        } else {
            n = mm / HW;
            const int rr = mm - n * HW;
            ho = rr / a.Wo; wo = rr - ho * a.Wo;
        }
        const size_t pix = ((size_t)n * a.OH + (ho * a.oy_mul + oy_off)) * a.OW + (wo * a.ox_mul + ox_off);
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int cbase = n0 + wn * WTN + 32 * j;
            float4 b4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = cbase + 8 * g + 4 * half;
                b4[g] = (a.bias && co < a.Cout) ? *reinterpret_cast<const float4 *>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // LeakyReLU backward of the layer below (its output = this conv's input, same shape as this output):
            // this lane's 4 x 4 channels of its pixel, fetched before the conversions so the latency overlaps them
            uint2 mk[4];
            if (a.mask_x) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cbase + 8 * g + 4 * half;
                    mk[g] = (mok && co < a.Cout) ? *reinterpret_cast<const uint2 *>(a.mask_x + pix * a.Cs + co) : make_uint2(0u, 0u);
                }
            }
            uint2 pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4] = {acc[j][i][4 * g] + b4[g].x, acc[j][i][4 * g + 1] + b4[g].y, acc[j][i][4 * g + 2] + b4[g].z,
                              acc[j][i][4 * g + 3] + b4[g].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.0f ? v[e] : v[e] * a.slope;
                if (a.mask_x) {
                    const float x0 = __uint_as_float(mk[g].x << 16), x1 = __uint_as_float(mk[g].x & 0xffff0000u);
                    const float x2 = __uint_as_float(mk[g].y << 16), x3 = __uint_as_float(mk[g].y & 0xffff0000u);
                    v[0] = x0 > 0.0f ? v[0] : v[0] * a.mask_slope;
                    v[1] = x1 > 0.0f ? v[1] : v[1] * a.mask_slope;
                    v[2] = x2 > 0.0f ? v[2] : v[2] * a.mask_slope;
                    v[3] = x3 > 0.0f ? v[3] : v[3] * a.mask_slope;
                }
                pk[g].x = pack_bf16(v[0], v[1]);
                pk[g].y = pack_bf16(v[2], v[3]);
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                // lanes < 32 end up with channels 8g..8g+7, lanes >= 32 with 8(g+1)..8(g+1)+7 of their pixel
                auto sx = __builtin_amdgcn_permlane32_swap(pk[g].x, pk[g + 1].x, false, false);
                auto sy = __builtin_amdgcn_permlane32_swap(pk[g].y, pk[g + 1].y, false, false);
                const int co = cbase + 8 * (g + half);
                if (mok && co < a.Cout) {
                    uint4 o;
                    o.x = sx[0]; o.y = sy[0]; o.z = sx[1]; o.w = sy[1];
                    *reinterpret_cast<uint4 *>(yb + pix * a.Cs + co) = o;
                }
            }
        }
    }
}

// ---- weights: fp32 torch layout [O][I][KH][KW]  ->  bf16 [Op][A][B][Ip] views used by the GEMMs
//   transpose == 0:  out[o][a][b][i] = in[o][i][th0+ths*a][tw0+tws*b]        (forward)
//   transpose == 1:  out[i][a][b][o] = in[o][i][th0+ths*a][tw0+tws*b]        (dgrad: flipped taps, swapped roles)
// rows/channels beyond the real extents are zero filled.
//   eff == 1: (th, tw) index the EFFECTIVE 4x4 kernel of "nearest x2 upsample -> 3x3 conv" on the low-res tensor,
//             W4[eh][ew] = sum of in[..][kh][kw] over kh in {eh-1, eh}, kw in {ew-1, ew} (inside the 3x3), summed in fp32 (after the
//             1/sigma scaling) and rounded to bf16 once: the sub-pixel form's four 2x2 class kernels are W4[2a+p][2b+q], the
//             adjoint stride-2 conv's kernel is W4[3-kh][3-kw] (conv_fwd_impl / conv_dgrad_impl, `subpixel`)
struct WeightPrepPart {
    unsigned short *out;
    int transpose, A, B, th0, ths, tw0, tws, Rp, Cp, Kp, eff;
};
struct WeightPrepArgs {
    const float *in;
    const float *sigma;
    int O, I, KH, KW, nparts;
    WeightPrepPart part[8];  // forward view + the dgrad view (stride 1) or its four parity classes (stride 2); upsample + 3x3 layers:
                             // + the four sub-pixel class views and the 4x4 adjoint view

    int tiled, OP, IP;       // k_weight_prep_tiled takes this layer: 32 x 32 (o, i) tiles over [0, OP) x [0, IP)
};
// one launch for all views of a layer: blockIdx.y = view
__device__ __forceinline__ void weight_prep_body(const WeightPrepArgs &w);
__global__ void k_weight_prep(WeightPrepArgs w) { weight_prep_body(w); }
// ... and for all layers of a network: blockIdx.z = layer of a device-resident table (m355_weight_prep_batched)
__global__ void k_weight_prep_batched(const WeightPrepArgs *__restrict__ tab, int skip_tiled)
{
    const WeightPrepArgs &w = tab[blockIdx.z];
    if (skip_tiled && w.tiled) return;
    if ((int)blockIdx.y < w.nparts) weight_prep_body(w);
}

// The same views through an LDS transpose.  k_weight_prep_batched gathers: a lane's eight channels of one tap sit KH*KW floats apart
// in the [O][I][KH][KW] parameter, so a load instruction touches 64 different lines and the texture addresser, not memory, sets
// the pace (74 us for the generator's 36 MB of weights, 1 TB/s).  Here a workgroup owns a 32 x 32 (o, i) tile with all its taps:
// it reads 32 contiguous runs of 32*T floats (scaled by 1/sigma on the way) into LDS and writes every view's 64-byte runs from
// there -- the forward view's rows are o with the tile's 32 i as a run per tap, the dgrad views' rows are i with its 32 o.
// "Regular" layers only (fill_weight_prep: every view's rows and channels multiples of 32, no K tail, at most 16 taps).
template <int TMAX>   // layers with TMAX/2 < KH*KW <= TMAX taps (two instantiations: the 37 KB tile of a 3x3 layer fits four times per CU)
__global__ __launch_bounds__(256) void k_weight_prep_tiled(const WeightPrepArgs *__restrict__ tab)
{
    const WeightPrepArgs &w = tab[blockIdx.z];
    const int T = w.KH * w.KW;
    if (!w.tiled || T > TMAX || (TMAX == 16 && T <= 9)) return;
    const int nti = w.IP >> 5, nto = w.OP >> 5;
    if ((int)blockIdx.x >= nti * nto) return;
    const int to = blockIdx.x / nti, ti = blockIdx.x - to * nti, o0 = to * 32, i0 = ti * 32;
    __shared__ float tile[32 * (32 * TMAX + 1)];
    const int RS = 32 * T + 1, tid = threadIdx.x, RT = 32 * T;
    const float wscale = w.sigma ? 1.0f / w.sigma[0] : 1.0f;
    const unsigned rcpT = 65536u / (unsigned)T + 1u, rcpRT = (1u << 24) / (unsigned)RT + 1u;   // f / T (f < 512), e / RT (e < 16384)
    // 32 rows x RT contiguous floats each; eight independent loads in flight per thread before the first lands in LDS
    for (int e0 = tid; e0 < 32 * RT; e0 += 8 * 256) {
        float v[8];
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u;
            v[u] = 0.0f;
            dst[u] = -1;
            if (e < 32 * RT) {
                const int ol = (int)(((unsigned)e * (unsigned long long)rcpRT) >> 24), f = e - ol * RT;
                const int o = o0 + ol, i = i0 + (int)(((unsigned)f * rcpT) >> 16);
                dst[u] = ol * RS + f;
                if (o < w.O && i < w.I) v[u] = w.in[((size_t)o * w.I + i0) * T + f];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (dst[u] >= 0) tile[dst[u]] = v[u] * wscale;
    }
    __syncthreads();
    for (int pi = 0; pi < w.nparts; ++pi) {
        const WeightPrepPart &p = w.part[pi];
        const int AB = p.A * p.B;
        // rows / channels of this view covered by the tile (whole tiles in or out: multiples of 32)
        const int r0 = p.transpose ? i0 : o0, c0 = p.transpose ? o0 : i0;
        if (r0 >= p.Rp || c0 >= p.Cp) continue;
        for (int u = tid; u < 32 * AB * 4; u += 256) {
            const int g8 = u & 3, q = u >> 2, rl = q / AB, ab = q - rl * AB;
            const int aa = ab / p.B, b = ab - aa * p.B;
            const int th = p.th0 + p.ths * aa, tw = p.tw0 + p.tws * b;
            bf16x8 o8 = {0, 0, 0, 0, 0, 0, 0, 0};
            // forward: row = o (rl), channels i = 8 g8 ..; dgrad: row = i (rl), channels o = 8 g8 ..
            const float *tb = p.transpose ? tile + (8 * g8) * RS + rl * T : tile + rl * RS + (8 * g8) * T;
            const int st = p.transpose ? RS : T;
            if (p.eff) {   // effective 4x4 kernel of upsample + 3x3: up to four source taps per entry, summed in fp32
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int kh = max(th - 1, 0); kh <= min(th, w.KH - 1); ++kh)
                    for (int kw = max(tw - 1, 0); kw <= min(tw, w.KW - 1); ++kw) {
                        const float *t0 = tb + kh * w.KW + kw;
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += t0[j * st];
                    }
#pragma unroll
                for (int j = 0; j < 8; ++j) o8[j] = (short)f2bf(v[j]);
            } else if (th >= 0 && tw >= 0 && th < w.KH && tw < w.KW) {
                const float *t0 = tb + th * w.KW + tw;
#pragma unroll
                for (int j = 0; j < 8; ++j) o8[j] = (short)f2bf(t0[j * st]);
            }
            *reinterpret_cast<bf16x8 *>(p.out + (size_t)(r0 + rl) * p.Kp + (size_t)ab * p.Cp + c0 + 8 * g8) = o8;
        }
    }
}
__device__ __forceinline__ void weight_prep_body(const WeightPrepArgs &w)
{
    const WeightPrepPart &p = w.part[blockIdx.y];
    const float *__restrict__ in = w.in;
    unsigned short *__restrict__ out = p.out;
    const int O = w.O, I = w.I, KH = w.KH, KW = w.KW, transpose = p.transpose, A = p.A, B = p.B, Cp = p.Cp, Kp = p.Kp;
    const float wscale = w.sigma ? 1.0f / w.sigma[0] : 1.0f;  // spectral norm: W_sn = W_orig / sigma (gan_glue.hip)
    // out is [Rp][Kp], a row = (A x B taps) x Cp channels then zero fill; R = transpose ? I : O, C = transpose ? O : I.
    // A thread writes EIGHT consecutive k (one 16-byte store): Cp and Kp are multiples of 8, so the eight share a tap and the
    // row / tap / channel split (run-time integer divisions: there is no divide instruction) is paid once per 16 bytes -- it
    // used to be a 64-bit and three 32-bit divisions per 2-byte element, 70 us for the generator's 36 MB of weights.
    const unsigned Kp8 = (unsigned)Kp >> 3, total8 = (unsigned)p.Rp * Kp8, AB = (unsigned)(A * B);
    const size_t sI = (size_t)KH * KW, sO = (size_t)I * KH * KW;   // element strides of the i / o index of `in`
    for (unsigned g = blockIdx.x * blockDim.x + threadIdx.x; g < total8; g += gridDim.x * blockDim.x) {
        const unsigned r = g / Kp8, k = (g - r * Kp8) << 3;
        const unsigned t = k / (unsigned)Cp, c = k - t * (unsigned)Cp;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (t < AB) {
            const unsigned aa = t / (unsigned)B, b = t - aa * (unsigned)B;
            const int th = p.th0 + p.ths * (int)aa, tw = p.tw0 + p.tws * (int)b;
            // element (o, i): forward rows are o and the eight channels are i = c ..; dgrad rows are i and the channels o = c ..
            const int nr = transpose ? I : O, nc = transpose ? O : I;
            const size_t sc = transpose ? sO : sI;
            if (p.eff) {   // effective 4x4 kernel of upsample + 3x3 (see WeightPrepPart): same fp32 sum order as the tiled kernel
                if ((int)r < nr)
                    for (int kh = max(th - 1, 0); kh <= min(th, KH - 1); ++kh)
                        for (int kw = max(tw - 1, 0); kw <= min(tw, KW - 1); ++kw) {
                            const float *src = in + (transpose ? (size_t)r * sI : (size_t)r * sO) + (size_t)kh * KW + kw;
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if ((int)c + j < nc) v[j] += src[(size_t)(c + j) * sc] * wscale;
                        }
            } else if (th >= 0 && tw >= 0 && th < KH && tw < KW) {
                if ((int)r < nr) {
                    const float *src = in + (transpose ? (size_t)r * sI : (size_t)r * sO) + (size_t)th * KW + tw;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if ((int)c + j < nc) v[j] = src[(size_t)(c + j) * sc] * wscale;
                }
            }
        }
#ifdef M355_EXACT
        float *outf = reinterpret_cast<float *>(out) + (size_t)g * 8;   // (no rounding: W / sigma in fp32)
#pragma unroll
        for (int j = 0; j < 8; ++j) outf[j] = v[j];
#else
        bf16x8 o8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = (short)f2bf(v[j]);
        *reinterpret_cast<bf16x8 *>(out + (size_t)g * 8) = o8;
#endif
    }
}

// ---- fold a gradient given in the padded/upsampled input frame back onto the stored input:
//   g[N, Hl(+0), Wl+2*pw, C] (W padded per mode; H already cropped to the logical rows)  ->  dx[N,H,W,C]
//   sums the pad columns into their source columns and the 2x2 replicas of the nearest upsample.
__global__ void k_fold_pad(const unsigned short *__restrict__ g, unsigned short *__restrict__ dx, int N, int H, int W,
                           int C, int ups, int pw, int mode, int Hg, int hoff)
{
    // one thread per 8-channel granule of dx
    const int Wl = W << ups, Wp = Wl + 2 * pw, C8 = C >> 3;
    const size_t total = (size_t)N * H * W * C8;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C8) * 8;
        size_t t = idx / C8;
        const int w = t % W;
        t /= W;
        const int h = t % H;
        const int n = t / H;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto add = [&](const unsigned short *p) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(p);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += bf2f((unsigned short)v[j]);
        };
        for (int dy = 0; dy <= ups; ++dy)
            for (int dxx = 0; dxx <= ups; ++dxx) {
                const int hl = (h << ups) + dy, wl = (w << ups) + dxx;
                const unsigned short *row = g + (((size_t)n * Hg + hl + hoff) * Wp) * C + c;
                add(row + (size_t)(wl + pw) * C);
                if (mode == 1) {  // replicate: all left pads read column 0, all right pads column Wl-1
                    if (wl == 0)
                        for (int p = 0; p < pw; ++p) add(row + (size_t)p * C);
                    if (wl == Wl - 1)
                        for (int p = 0; p < pw; ++p) add(row + (size_t)(Wl + pw + p) * C);
                } else if (mode == 2) {  // circular: pad column p <-> source Wl-pw+p (left), Wl+pw+p <-> p (right)
                    if (wl >= Wl - pw) add(row + (size_t)(wl - (Wl - pw)) * C);
                    if (wl < pw) add(row + (size_t)(Wl + pw + wl) * C);
                }
            }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(s[j]);
        *reinterpret_cast<bf16x8 *>(dx + idx * 8) = o;
    }
}

// ---- split-K finishing pass of a FORWARD conv: y[m][c] = act(bias[c] + sum_s ws[s][m][c]) as bf16 NHWC (the GEMM pixel grid IS
// the output: no parity classes / offsets here), and -- for the convs that feed a batch norm -- the workgroup's (sum, sum of
// squares) of the fp32 results, part[blockIdx.x][2][C]: the row layout m355_bn_finalize reduces (this pass replaces the
// bn_stats_partial launch those layers needed anyway).  Thread = one pixel lane x 8 channels; ppb pixels per workgroup.
__global__ __launch_bounds__(256) void k_splitk_finish(const float *__restrict__ ws, int S, size_t sstride, int M, int C, int CP,
                                                       const float *__restrict__ bias, float slope, unsigned short *__restrict__ y,
                                                       float *__restrict__ part, int ppb)
{
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x, vecs = C >> 3, lanes = 256 / vecs, v = tid % vecs, pl = tid / vecs;
    const int p0 = blockIdx.x * ppb, p1 = min(M, p0 + ppb);
    float bv[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bv[j] = bias ? bias[v * 8 + j] : 0.0f;
        s1[j] = s2[j] = 0.0f;
    }
    if (pl < lanes)
        for (int p = p0 + pl; p < p1; p += lanes) {
            float z[8] = {bv[0], bv[1], bv[2], bv[3], bv[4], bv[5], bv[6], bv[7]};
            const float *src = ws + (size_t)p * CP + v * 8;
            for (int s = 0; s < S; ++s) {   // slice order: deterministic
                const float4 a0 = *reinterpret_cast<const float4 *>(src + s * sstride), a1 = *reinterpret_cast<const float4 *>(src + s * sstride + 4);
                z[0] += a0.x; z[1] += a0.y; z[2] += a0.z; z[3] += a0.w;
                z[4] += a1.x; z[5] += a1.y; z[6] += a1.z; z[7] += a1.w;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1[j] += z[j];
                s2[j] += z[j] * z[j];
                z[j] = z[j] >= 0.0f ? z[j] : z[j] * slope;
            }
            uint4 o;
            o.x = pack_bf16(z[0], z[1]); o.y = pack_bf16(z[2], z[3]); o.z = pack_bf16(z[4], z[5]); o.w = pack_bf16(z[6], z[7]);
            *reinterpret_cast<uint4 *>(y + (size_t)p * C + v * 8) = o;
        }
    if (!part) return;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 8 + j] = k ? s2[j] : s1[j];
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            const int vv = c >> 3, jj = c & 7;
            float t = 0.0f;
            for (int l = 0; l < lanes; ++l) t += red[(l * vecs + vv) * 8 + jj];
            part[((size_t)blockIdx.x * 2 + k) * C + c] = t;
        }
    }
}

// rows of a weight view: a multiple of the N tile the launcher will pick for that many output channels
static inline int ilog2_exact(int v)
{
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

static inline int rows_padded(int cout) { return cout <= 64 ? 64 : (cout + 127) / 128 * 128; }
static inline int k_padded(int k) { return (k + 63) / 64 * 64; }
// channel stride of the gradient tensor dy the backward entry points take: 8 for the 1..8-channel heads (so that the
// dgrad GEMM's K is taps*8, not taps*32), otherwise Cout rounded up to 32
static inline int dy_channels(int cout) { return cout <= 8 ? 8 : (cout + 31) / 32 * 32; }

bool conv_halo_eligible(const ConvArgs &a);  // csrc/conv_halo.hip
int conv_halo_launch(const ConvArgs &a, unsigned xb, unsigned wb, hipStream_t st);
int conv_halo_stats_rows(const ConvArgs &a);

bool dgrad_direct_replicate_eligible(const m355_conv_desc *d, int Cy);  // csrc/conv_halo.hip
int dgrad_direct_replicate_launch(const m355_conv_desc *d, const void *dy, int Cy, const void *w_dgrad, int Kp, int rows_p,
                                  void *dx, hipStream_t st);
bool wgrad_halo_eligible(const WgradArgs &a);  // csrc/conv_halo.hip
int wgrad_halo_launch(const WgradArgs &a, unsigned xb, unsigned yb, hipStream_t st);
int wgrad_halo_up_launch(const WgradArgs &a, unsigned xb, unsigned yb, hipStream_t st);   // sub-pixel classes of upsample + 3x3
int wgrad_halo_part_rows(const WgradArgs &a);   // partial rows of the no-atomics form of a halo weight gradient (0: none)
int wgrad_halo_up_part_rows(const WgradArgs &a);   // ... of the sub-pixel classes (rows of [Cout*16*Cin | 4 x Cout] cells)
int dgrad_edge_up4_launch(const m355_conv_desc *d, const void *dy, int Cy, const void *w4, int Kp, void *dx, hipStream_t st);

static bool dma_eligible(const ConvArgs &a)
{
    const size_t xbytes = (size_t)a.N * a.H * a.W * a.Cin * 2, wbytes = (size_t)rows_padded(a.Cout) * a.Kp * 2;
    return !a.y_f32_nchw && a.Cout % 8 == 0 && a.Cs % 8 == 0 && a.Cin % 8 == 0 && a.Kp % 64 == 0 &&
           xbytes < (1ull << 31) && wbytes < (1ull << 31);
}

// can this problem read / write bit masks?  (the kernel that will run it must be k_conv_halo with the unguarded epilogue)
static bool halo_takes_bits(ConvArgs a)
{
    a.CoutP = rows_padded(a.Cout);
    if (a.ncls < 1) a.ncls = 1;
    const char *h = getenv("M355_CONV_HALO");
    return dma_eligible(a) && !(h && h[0] == '0') && conv_halo_eligible(a) && a.Cout == a.CoutP && !a.fold2 && a.Cs % 64 == 0 &&
           !a.y_f32_nchw && !getenv("M355_NO_MASKBITS");
}

// K slices for a small problem (0: no split): plain output frame, 64-channel K steps, whole 128-channel output tiles, fewer
// 128 x 128 tiles than CUs.  The caller (m355_conv2d_fwd_ws) owns the finishing pass.
static int splitk_plan(const ConvArgs &a_in)
{
    ConvArgs a = a_in;
    a.CoutP = rows_padded(a.Cout);
    if (a.ncls < 1) a.ncls = 1;
    if (getenv("M355_NO_SPLITK")) return 0;
    if (!dma_eligible(a) || a.Cin % 64 || a.ncls != 1 || a.y_f32_nchw || a.mask_x || a.bits_in || a.bits_out || a.fold2 || a.stats)
        return 0;
    if (a.Cout != a.CoutP || a.CoutP % 128 || a.oy_mul != 1 || a.ox_mul != 1 || a.oy_off || a.ox_off || a.OH != a.Ho || a.OW != a.Wo)
        return 0;
    {
        const char *h = getenv("M355_CONV_HALO");
        if (!(h && h[0] == '0') && conv_halo_eligible(a)) return 0;
    }
    const long M = (long)a.N * a.Ho * a.Wo, tiles = ((M + 127) / 128) * (a.CoutP / 128);
    const int nsteps = a.Kp / 64;
    if (tiles >= 192 || nsteps < 8) return 0;
    int S = (int)((256 + tiles - 1) / tiles);
    if (S > nsteps / 4) S = nsteps / 4;   // at least four K steps per slice
    if (S > 16) S = 16;
    return S >= 2 ? S : 0;
}

static int launch_conv(ConvArgs a, hipStream_t st)
{
    const int M = a.N * a.Ho * a.Wo;
    a.CoutP = rows_padded(a.Cout);
    if (a.ncls < 1) a.ncls = 1;
    a.lgWo = ilog2_exact(a.Wo);
    a.lgHo = ilog2_exact(a.Ho);
    if (a.lgWo < 0 || a.lgHo < 0) a.lgWo = a.lgHo = -1;
    const size_t xbytes = (size_t)a.N * a.H * a.W * a.Cin * 2, wbytes = (size_t)a.CoutP * a.Kp * 2;
    const bool dma_ok = dma_eligible(a);
    if (a.bits_out || a.bits_in) {
        const bool ok = halo_takes_bits(a);
        if (!ok) {
            set_error("conv2d: bit masks are only handled by the unguarded halo epilogue (ask m355_conv2d_maskbits_ok first)");
            return M355_ERR_BAD_ARG;
        }
    }
    if (dma_ok) {
        const bool fast = a.Cin % 64 == 0;
        const unsigned xb = (unsigned)xbytes, wb = (unsigned)wbytes;
        if (a.sk >= 2) {   // split-K (the caller planned it and owns the finishing pass)
            const dim3 grid((unsigned)((M + 127) / 128) * (a.CoutP / 128), a.sk);
            if (a.pad_w_mode == 0) hipLaunchKernelGGL((k_conv_glds<128, 128, 4, 2, true, 0, 2, true>), grid, dim3(256), 0, st, a, xb, wb);
            else if (a.pad_w_mode == 1) hipLaunchKernelGGL((k_conv_glds<128, 128, 4, 2, true, 1, 2, true>), grid, dim3(256), 0, st, a, xb, wb);
            else hipLaunchKernelGGL((k_conv_glds<128, 128, 4, 2, true, 2, 2, true>), grid, dim3(256), 0, st, a, xb, wb);
            note_kernel("k_conv_glds");
            return check_launch("conv2d (dma, split K)");
        }
        {
            const char *h = getenv("M355_CONV_HALO");  // "0": keep everything on k_conv_glds (A/B runs)
            if (!(h && h[0] == '0') && conv_halo_eligible(a)) return conv_halo_launch(a, xb, wb, st);
        }
#define M355_GO(BM_, BN_, NW_, WGN_, F_, MD_) \
    hipLaunchKernelGGL((k_conv_glds<BM_, BN_, NW_, WGN_, F_, MD_, (BM_ == 64 ? M355_GLDS_SMALL_NST : 2)>), grid, dim3(NW_ * 64), 0, st, a, xb, wb)
#define M355_MODES(BM_, BN_, NW_, WGN_, F_)                                   \
    do {                                                                      \
        if (a.pad_w_mode == 0) M355_GO(BM_, BN_, NW_, WGN_, F_, 0);           \
        else if (a.pad_w_mode == 1) M355_GO(BM_, BN_, NW_, WGN_, F_, 1);      \
        else M355_GO(BM_, BN_, NW_, WGN_, F_, 2);                             \
    } while (0)
#define M355_TILE(BM_, BN_, NW_, WGN_)                                        \
    do {                                                                      \
        grid = dim3((unsigned)((M + BM_ - 1) / BM_) * (a.CoutP / BN_), a.ncls); \
        if (fast) M355_MODES(BM_, BN_, NW_, WGN_, true);                      \
        else M355_MODES(BM_, BN_, NW_, WGN_, false);                          \
    } while (0)
        // Tile choice (measured at batch 64, profiles/r01_conv_tiles.txt): 256x256 (8 waves, 128x64 per wave: half the
        // LDS-DMA instructions and L2->LDS bytes per flop) wins by 15-20 % wherever Cout is a multiple of 256 and
        // there are >= 4 tiles per CU; 256x128 (8 waves, one workgroup per CU) never beats two co-resident 128x128
        // workgroups, so it is only reachable through M355_CONV_TILE.
        dim3 grid;
        const long tiles128 = (long)((M + 127) / 128) * (a.CoutP / 128) * a.ncls;
        const char *force = getenv("M355_CONV_TILE");  // tests / tuning: "128x128", "256x128", "256x256"
        int pick = a.CoutP == 64 ? 0 : (a.CoutP % 256 == 0 && tiles128 >= 4 * 1024) ? 3 : 1;
        // small problems (the generator's 8x4 .. 32x16 stages, the mesh discriminator): fewer 128-wide tiles than CUs
        // leave most of the chip idle for the whole K loop -- 64x64 tiles give 4x the workgroups (32 KB of LDS each)
        const long tiles_now = pick == 0 ? (long)((M + 255) / 256) * a.ncls : tiles128;
        if (tiles_now < 192 && !getenv("M355_NO_SMALL_TILE")) pick = 4;
        if (force && a.CoutP != 64) {
            if (!strcmp(force, "128x128")) pick = 1;
            else if (!strcmp(force, "256x128")) pick = 2;
            else if (!strcmp(force, "256x256") && a.CoutP % 256 == 0) pick = 3;
        }
        if (force && !strcmp(force, "64x64")) pick = 4;
        if (pick == 4) M355_TILE(64, 64, 4, 2);
        else if (pick == 0) M355_TILE(256, 64, 4, 1);
        else if (pick == 3) M355_TILE(256, 256, 8, 4);
        else if (pick == 2) M355_TILE(256, 128, 8, 2);
        else M355_TILE(128, 128, 4, 2);
#undef M355_TILE
#undef M355_MODES
#undef M355_GO
        note_kernel("k_conv_glds");
        return check_launch("conv2d (dma)");
    }
    if (a.ncls != 1) {
        set_error("conv2d: merged parity classes need the DMA kernel");
        return M355_ERR_BAD_ARG;
    }
    dim3 grid((M + BM - 1) / BM, a.CoutP / BN);
    hipLaunchKernelGGL(k_conv_mfma, grid, dim3(256), 0, st, a);
    note_kernel("k_conv_mfma");
    return check_launch("conv2d");
}

}  // namespace m355

using m355::ConvArgs;

namespace m355 {  // csrc/conv_small.hip
bool conv_small_eligible(const m355_conv_desc *d, int y_f32_nchw);
int conv_small_launch(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, float slope,
                      int Kp, size_t wbytes, hipStream_t st);
bool conv_c8_eligible(const m355_conv_desc *d, int y_f32_nchw);
int conv_c8_launch(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, float slope, int Kp,
                   size_t wbytes, unsigned *bits, hipStream_t st, const void *mask = nullptr, float mask_slope = 1.0f);
bool dgrad_small_eligible(const m355_conv_desc *d, int Cy);
int dgrad_small_launch(const m355_conv_desc *d, const void *dy, int Cy, const void *w_dgrad, int Kp, size_t wbytes, void *dx,
                       hipStream_t st, int lead = 0);
bool dgrad_c8_replicate_eligible(const m355_conv_desc *d, int Cy);  // csrc/conv_small.hip
int dgrad_c8_replicate_launch(const m355_conv_desc *d, const void *dy, const void *w_dgrad, int Kp, size_t wbytes, void *dx,
                              hipStream_t st, const void *mask_x, float mask_slope);
bool wgrad_c8_eligible(const m355_conv_desc *d, int Cy);  // csrc/conv_small.hip
int wgrad_c8_launch(const m355_conv_desc *d, const void *x, const void *dy, int Cy, float *dw, float *db, hipStream_t st,
                    long long *fix = nullptr, float *part = nullptr);
size_t wgrad_c8_ws_floats(const m355_conv_desc *d, int Cy);
bool wgrad_small_eligible(const m355_conv_desc *d, int Cy);
int wgrad_small_launch(const m355_conv_desc *d, const void *x, const void *dy, int Cy, float *dw, hipStream_t st,
                       long long *fix = nullptr, float *part = nullptr);
size_t wgrad_small_ws_floats(const m355_conv_desc *d);
}  // namespace m355

#ifdef M355_EXACT
namespace m355 { namespace exact {   // csrc/conv_exact.hip: the fp32 kernels every conv entry point dispatches to in the EXACT build
int conv_fwd(const m355_conv_desc *d, const void *x, const void *w, const float *bias, void *y, int nchw, float slope, hipStream_t st);
int conv_dgrad(const m355_conv_desc *d, const void *dy, int Cy, const void *w, void *dx, const void *mask_x, float mask_slope, hipStream_t st);
int conv_wgrad(const m355_conv_desc *d, const void *x, const void *dy, int Cy, float *dw, float *dbias, int accumulate, hipStream_t st);
int fold2x2(const void *gr, void *dx, int N, int H, int W, int C, hipStream_t st);
} }
#endif

static int conv_out_hw(const m355_conv_desc *d, int *Ho, int *Wo)
{
    const int Hl = d->H << d->upsample, Wl = d->W << d->upsample;
    *Ho = (Hl + 2 * d->pad_h - d->kh) / d->stride + 1;
    *Wo = (Wl + 2 * d->pad_w - d->kw) / d->stride + 1;
    return (*Ho > 0 && *Wo > 0) ? 0 : -1;
}

static int check_desc(const m355_conv_desc *d, const char *who)
{
    M355_REQUIRE(d, "%s: null descriptor", who);
    M355_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "%s: non-positive size", who);
    M355_REQUIRE(d->Cin % 8 == 0, "%s: Cin=%d must be a multiple of 8 (pad the channels)", who, d->Cin);
    M355_REQUIRE(d->stride == 1 || d->stride == 2, "%s: stride %d", who, d->stride);
    M355_REQUIRE(d->upsample == 0 || d->upsample == 1, "%s: upsample %d", who, d->upsample);
    M355_REQUIRE(d->pad_w_mode >= 0 && d->pad_w_mode <= 2, "%s: pad_w_mode %d", who, d->pad_w_mode);
    M355_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->kh <= 7 && d->kw <= 7, "%s: kernel %dx%d", who, d->kh, d->kw);
    int Ho, Wo;
    M355_REQUIRE(conv_out_hw(d, &Ho, &Wo) == 0, "%s: empty output", who);
    return 0;
}

extern "C" int m355_conv2d_out_hw(const m355_conv_desc *d, int *Ho, int *Wo)
{
    M355_REQUIRE(d && Ho && Wo, "conv2d_out_hw: null pointer");
    return conv_out_hw(d, Ho, Wo) == 0 ? M355_OK : M355_ERR_BAD_ARG;
}

/* adjoint of the nearest x2 upsample on NHWC bf16: dx[n,h,w,:] = sum of the 2x2 block of g[n,2h..2h+1,2w..2w+1,:] */
extern "C" int m355_fold2x2(const void *g, void *dx, int N, int H, int W, int C, void *stream)
{
    M355_REQUIRE(g && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "fold2x2: bad argument");
#ifdef M355_EXACT
    return m355::exact::fold2x2(g, dx, N, H, W, C, (hipStream_t)stream);
#endif
    const size_t total = (size_t)N * H * W * (C / 8);
    hipLaunchKernelGGL(m355::k_fold_pad, dim3((unsigned)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned short *)g, (unsigned short *)dx, N, H, W, C, 1, 0, 0, 2 * H, 0);
    return m355::check_launch("fold2x2");
}

extern "C" int m355_conv2d_dy_channels(int cout) { return m355::dy_channels(cout); }

// ---- "nearest x2 upsample -> 3x3 stride-1 conv" (models/gan.py:319,386-404 `self.up` in front of ResBlockUp.conv1 :294,309) in its
// SUB-PIXEL form.  Output pixel (2i+p, 2j+q) reads the upsampled rows 2i+p-1 .. 2i+p+1 = the stored rows {i-1, i, i} (p = 0) or
// {i, i, i+1} (p = 1): per output parity class the conv is a 2x2 stride-1 conv of the STORED tensor with pre-summed weights
// (w0, w1+w2) / (w0+w1, w2) per axis -- 4 taps instead of 9, exactly (the zero H pad and the replicate / circular W pad of the
// upsampled tensor are the same pads of the stored one).  Forward = four class convs (the launch shape of a stride-2 dgrad),
// dgrad = the adjoint 4x4 stride-2 conv of dy (+ the replicate pad columns' edge term), wgrad = per class the 2x2 all-taps problem
// into a 16-entry effective gradient, folded onto the 3x3 taps by the adjoint of the pre-sum.  `up3`: the layers whose weight
// buffers carry the extra views (a property of the layer, not of the image size: the views are prepared once per network);
// `subpixel`: ... and whose shape runs on the halo kernels in that form.
static bool up3(const m355_conv_desc *d)
{
    return d->upsample == 1 && d->stride == 1 && d->kh == 3 && d->kw == 3 && d->pad_h == 1 && d->pad_w == 1;
}
// subpixel_halo: the shapes whose class convs run on the halo kernels -- forward, dgrad AND the 16-entry weight gradient
static bool subpixel_halo(const m355_conv_desc *d)
{
#ifdef M355_EXACT
    return false;
#endif
    if (!up3(d) || d->Cin % 64 || d->Cout % 64 || d->W % 32 || d->H % 8) return false;
    const size_t xb = (size_t)d->N * d->H * d->W * d->Cin * 2, yb = (size_t)d->N * d->H * d->W * 4 * d->Cout * 2;
    static const bool off = getenv("M355_NO_SUBPIXEL") != nullptr;   // A/B runs: the 9-tap kernels with the upsample in the halo load
    return xb < (1ull << 31) && yb < (1ull << 31) && !off;
}
// subpixel: ... plus (round 6) the SMALL stages whose 9-tap form is not a halo shape either (upsampled width not a multiple of 32 or
// height not of 8: the generator's 16x8 -> 32x16 layers, blk3a.conv1 / blk3_mesh.conv1 at 256^2).  Their FORWARD and DGRAD take the
// form on the generic implicit-GEMM kernel -- four class launches are what it already runs for a stride-2 dgrad, and the adjoint is a
// plain 4x4 stride-2 conv of dy -- with 4/9 of the taps' pixel gathers and MACs; their weight gradient stays the 9-tap one (the
// 16-entry form exists on the halo tiles only), which is the same operator's gradient.  Only from 4096 stored pixels per launch:
// below that (the 8x4 stage at batch 64, everything at batch 16) the class GEMMs are a few dozen workgroups with a long K loop, and
// the split-K form of the 9-tap conv -- whose finishing pass also emits the batch-norm sums -- is the faster one (measured same box,
// profiles/r06_small_subpixel_ab.txt: blk2.conv1's dgrad 52 -> 96 us in the sub-pixel form; batch 16 +0.07 ms per cycle).
static bool subpixel(const m355_conv_desc *d)
{
#ifdef M355_EXACT
    return false;
#endif
    if (subpixel_halo(d)) return true;
    if (!up3(d) || d->Cin % 64 || d->Cout % 64) return false;
    static const bool off = getenv("M355_NO_SUBPIXEL") != nullptr || getenv("M355_NO_SUBPIXEL_SMALL") != nullptr;
    const bool halo9 = (2 * d->W) % 32 == 0 && (2 * d->H) % 8 == 0;   // (the 9-tap form runs on k_conv_halo with fused statistics: keep it)
    return !off && !halo9 && (long)d->N * d->H * d->W >= 4096;
}
// element offsets of the sub-pixel views inside the forward / dgrad weight buffers of an up3 layer (behind the 3x3 views)
static size_t up3_fwd_off(const m355_conv_desc *d) { return (size_t)m355::rows_padded(d->Cout) * m355::k_padded(9 * d->Cin); }
static size_t up3_dgrad_off(const m355_conv_desc *d)
{
    return (size_t)m355::rows_padded(d->Cin) * m355::k_padded(9 * m355::dy_channels(d->Cout));
}

/* executed / algorithmic multiply-accumulates of this layer's forward, dgrad and workspace / deterministic wgrad: 4/9 where the
 * layer runs in the sub-pixel form (an upsample + 3x3 conv: 4 taps on the stored tensor instead of 9 on the upsampled one), else 1.
 * The benchmark accounts every conv at the OPERATOR's 2*M*N*K and reports the executed figure beside it. */
extern "C" double m355_conv2d_exec_ratio(const m355_conv_desc *d) { return (d && subpixel(d)) ? 4.0 / 9.0 : 1.0; }

extern "C" size_t m355_conv2d_weight_elems(const m355_conv_desc *d, int which)
{
    // which 0: forward view [rows_padded(Cout)][ceil64(kh*kw*Cin)]; 1: dgrad views (stride 1: one
    // [rows_padded(Cin)][ceil64(kh*kw*Cout_p32)]; stride 2: four [rows_padded(Cin)][ceil64(kh/2*kw/2*Cout_p32)])
    if (!d) return 0;
#ifdef M355_EXACT
    // fp32 [Cout][kh][kw][Cin] for both views (counted in 2-byte elements: the caller allocates a bf16 buffer of that many)
    return 2 * (size_t)d->Cout * d->kh * d->kw * d->Cin;
#endif
    const size_t rows_f = m355::rows_padded(d->Cout), rows_d = m355::rows_padded(d->Cin);
    const size_t cout32 = (size_t)m355::dy_channels(d->Cout);
    if (up3(d)) {   // + the four sub-pixel class views [rows_f][ceil64(4 Cin)] / the adjoint 4x4 view [rows_d][ceil64(16 Cout_p32)]
        if (which == 0) return up3_fwd_off(d) + 4 * rows_f * (size_t)m355::k_padded(4 * d->Cin);
        return up3_dgrad_off(d) + rows_d * (size_t)m355::k_padded(16 * (int)cout32);
    }
    if (which == 0) return rows_f * (size_t)m355::k_padded(d->kh * d->kw * d->Cin);
    if (d->stride == 1) return rows_d * (size_t)m355::k_padded(d->kh * d->kw * (int)cout32);
    return 4 * rows_d * (size_t)m355::k_padded(((d->kh + 1) / 2) * ((d->kw + 1) / 2) * (int)cout32);  // four parity-class views
}

// the views of one layer: forward [rows_padded(Cout)][Kp] + the dgrad view (stride 1) or its four parity classes (stride 2)
static int fill_weight_prep(const m355_conv_desc *d, const float *w_oihw, int cin_w, const float *sigma, void *w_fwd, void *w_dgrad,
                            m355::WeightPrepArgs &w, size_t &most)
{
    const int cout64 = m355::rows_padded(d->Cout), cin64 = m355::rows_padded(d->Cin), cout32 = m355::dy_channels(d->Cout);
    w = m355::WeightPrepArgs{};
    w.in = w_oihw; w.sigma = sigma; w.O = d->Cout; w.I = cin_w; w.KH = d->kh; w.KW = d->kw;
    most = 0;
    bool ok = true;
    auto add = [&](unsigned short *out, int transpose, int A, int B, int th0, int ths, int tw0, int tws, int Rp, int Cp, int Kp,
                   int eff = 0) {
        ok = ok && Cp % 8 == 0 && Kp % 8 == 0 && (size_t)Rp * Kp < (1ull << 31);   // (the kernel writes 8 channels of one tap at a time)
        w.part[w.nparts++] = m355::WeightPrepPart{out, transpose, A, B, th0, ths, tw0, tws, Rp, Cp, Kp, eff};
        if ((size_t)Rp * Kp > most) most = (size_t)Rp * Kp;
    };
#ifdef M355_EXACT
    // EXACT build: both "views" are the fp32 array [Cout][kh][kw][Cin] (conv_exact.hip), written by the gather kernel
    if (w_fwd) add((unsigned short *)w_fwd, 0, d->kh, d->kw, 0, 1, 0, 1, d->Cout, d->Cin, d->kh * d->kw * d->Cin);
    if (w_dgrad) add((unsigned short *)w_dgrad, 0, d->kh, d->kw, 0, 1, 0, 1, d->Cout, d->Cin, d->kh * d->kw * d->Cin);
    if (!ok) {
        m355::set_error("conv2d_weight_prep: channel counts must be multiples of 8 (Cin=%d)", d->Cin);
        return M355_ERR_BAD_ARG;
    }
    w.tiled = 0; w.OP = 0; w.IP = 0;
    return M355_OK;
#endif
    if (w_fwd) add((unsigned short *)w_fwd, 0, d->kh, d->kw, 0, 1, 0, 1, cout64, d->Cin, m355::k_padded(d->kh * d->kw * d->Cin));
    if (w_dgrad) {
        if (d->stride == 1) {
            add((unsigned short *)w_dgrad, 1, d->kh, d->kw, d->kh - 1, -1, d->kw - 1, -1, cin64, cout32,
                m355::k_padded(d->kh * d->kw * cout32));
        } else {
            // odd kernels (3x3 / 5x5 stride 2 of models/reconstruction.py:53-63) are handled as the next even size with a
            // zero last row / column: same output size for an even padded frame, taps beyond the kernel read as zero
            const int A = (d->kh + 1) / 2, B = (d->kw + 1) / 2;
            const int Kp = m355::k_padded(A * B * cout32);
            const size_t each = (size_t)cin64 * Kp;
            for (int py = 0; py < 2; ++py)
                for (int px = 0; px < 2; ++px)
                    add((unsigned short *)w_dgrad + (size_t)(py * 2 + px) * each, 1, A, B, py + 2 * (A - 1), -2, px + 2 * (B - 1), -2,
                        cin64, cout32, Kp);
        }
    }
    if (up3(d)) {   // sub-pixel views (see `subpixel`): class (p, q) tap (a, b) = W4[2a+p][2b+q]; adjoint tap (kh, kw) = W4[3-kh][3-kw]
        if (w_fwd) {
            const int Kp = m355::k_padded(4 * d->Cin);
            for (int c = 0; c < 4; ++c)
                add((unsigned short *)w_fwd + up3_fwd_off(d) + (size_t)c * cout64 * Kp, 0, 2, 2, c >> 1, 2, c & 1, 2, cout64, d->Cin, Kp, 1);
        }
        if (w_dgrad)
            add((unsigned short *)w_dgrad + up3_dgrad_off(d), 1, 4, 4, 3, -1, 3, -1, cin64, cout32, m355::k_padded(16 * cout32), 1);
    }
    if (!ok) {
        m355::set_error("conv2d_weight_prep: channel counts must be multiples of 8 (Cin=%d, dy channels=%d)", d->Cin, cout32);
        return M355_ERR_BAD_ARG;
    }
    // k_weight_prep_tiled's layers: whole 32 x 32 tiles, no K tail, the taps fit its LDS tile
    bool reg = d->kh * d->kw <= 16 && w.nparts > 0 && !getenv("M355_NO_WPREP_TILED");
    int OP = 0, IP = 0;
    for (int k = 0; k < w.nparts; ++k) {
        const m355::WeightPrepPart &p = w.part[k];
        reg = reg && p.Cp % 32 == 0 && p.Rp % 32 == 0 && p.Kp == p.A * p.B * p.Cp;
        const int po = p.transpose ? p.Cp : p.Rp, pin = p.transpose ? p.Rp : p.Cp;
        OP = po > OP ? po : OP;
        IP = pin > IP ? pin : IP;
    }
    w.tiled = reg ? 1 : 0; w.OP = OP; w.IP = IP;
    return M355_OK;
}

extern "C" int m355_conv2d_weight_prep(const m355_conv_desc *d, const float *w_oihw, int cin_w, const float *sigma,
                                       void *w_fwd, void *w_dgrad, void *stream)
{
    if (int rc = check_desc(d, "conv2d_weight_prep")) return rc;
    M355_REQUIRE(w_oihw && (w_fwd || w_dgrad), "conv2d_weight_prep: null pointer");
    M355_REQUIRE(cin_w >= 1 && cin_w <= d->Cin, "conv2d_weight_prep: cin_w=%d outside 1..Cin=%d", cin_w, d->Cin);
    m355::WeightPrepArgs w;
    size_t most = 0;
    if (int rc = fill_weight_prep(d, w_oihw, cin_w, sigma, w_fwd, w_dgrad, w, most)) return rc;
    const unsigned bx = (unsigned)((most + 255) / 256 > 2048 ? 2048 : (most + 255) / 256);
    hipLaunchKernelGGL(m355::k_weight_prep, dim3(bx, w.nparts), dim3(256), 0, (hipStream_t)stream, w);
    return m355::check_launch("conv2d_weight_prep");
}

/* All weight views of a network in ONE launch.  m355_weight_prep_entry_bytes() bytes per layer, filled on the host by
 * m355_weight_prep_fill_entry (arguments of m355_conv2d_weight_prep, device pointers; returns the layer's largest view
 * in elements, -1 on error) into a buffer the caller copies to the device once; m355_weight_prep_batched runs the
 * table (max_elems = the largest value the fill calls returned). */
extern "C" size_t m355_weight_prep_entry_bytes(void) { return sizeof(m355::WeightPrepArgs); }

extern "C" long long m355_weight_prep_fill_entry(const m355_conv_desc *d, const float *w_oihw, int cin_w, const float *sigma,
                                                 void *w_fwd, void *w_dgrad, void *entry_host)
{
    if (check_desc(d, "weight_prep_fill_entry") || !w_oihw || !(w_fwd || w_dgrad) || !entry_host || cin_w < 1 || cin_w > d->Cin) {
        m355::set_error("weight_prep_fill_entry: bad argument");
        return -1;
    }
    m355::WeightPrepArgs w;
    size_t most = 0;
    if (fill_weight_prep(d, w_oihw, cin_w, sigma, w_fwd, w_dgrad, w, most)) return -1;
    memcpy(entry_host, &w, sizeof(w));
    return (long long)most;
}

extern "C" int m355_weight_prep_batched(const void *table_dev, int L, long long max_elems, void *stream)
{
    M355_REQUIRE(table_dev && L > 0 && L <= 65535 && max_elems > 0, "weight_prep_batched: bad argument");
    const unsigned bx = (unsigned)((max_elems + 255) / 256 > 256 ? 256 : (max_elems + 255) / 256);
    hipLaunchKernelGGL(m355::k_weight_prep_batched, dim3(bx, 8, L), dim3(256), 0, (hipStream_t)stream,
                       (const m355::WeightPrepArgs *)table_dev, 0);
    return m355::check_launch("weight_prep_batched");
}

/* tiles of the LDS-transpose kernel this layer's entry needs (0: the layer stays on the gather kernel) */
extern "C" int m355_weight_prep_entry_tiles(const void *entry_host)
{
    if (!entry_host) return 0;
    m355::WeightPrepArgs w;
    memcpy(&w, entry_host, sizeof(w));
    return w.tiled ? (w.OP >> 5) * (w.IP >> 5) : 0;
}

/* as m355_weight_prep_batched, with the regular layers (m355_weight_prep_entry_tiles > 0) on the LDS-transpose kernel --
 * max_tiles9 / max_tiles16 = the largest tile count among those with kh*kw <= 9 / 10..16 taps -- and the rest (max_elems = their
 * largest view, 0 if there is none) on the gather kernel */
extern "C" int m355_weight_prep_batched_tiled(const void *table_dev, int L, long long max_elems, int max_tiles9, int max_tiles16,
                                              void *stream)
{
    M355_REQUIRE(table_dev && L > 0 && L <= 65535 && max_elems >= 0 && max_tiles9 >= 0 && max_tiles16 >= 0 &&
                     (max_elems > 0 || max_tiles9 > 0 || max_tiles16 > 0),
                 "weight_prep_batched_tiled: bad argument");
    if (max_tiles9 > 0)
        hipLaunchKernelGGL(m355::k_weight_prep_tiled<9>, dim3((unsigned)max_tiles9, 1, L), dim3(256), 0, (hipStream_t)stream,
                           (const m355::WeightPrepArgs *)table_dev);
    if (max_tiles16 > 0)
        hipLaunchKernelGGL(m355::k_weight_prep_tiled<16>, dim3((unsigned)max_tiles16, 1, L), dim3(256), 0, (hipStream_t)stream,
                           (const m355::WeightPrepArgs *)table_dev);
    if (max_elems > 0) {
        const unsigned bx = (unsigned)((max_elems + 255) / 256 > 256 ? 256 : (max_elems + 255) / 256);
        hipLaunchKernelGGL(m355::k_weight_prep_batched, dim3(bx, 8, L), dim3(256), 0, (hipStream_t)stream,
                           (const m355::WeightPrepArgs *)table_dev, 1);
    }
    return m355::check_launch("weight_prep_batched");
}

// probe != 0: do not launch, return 1 / 0 = this forward can / cannot write the activation bit masks
// probe 1: can this forward emit / read bit masks?  probe 2: rows of fused batch-norm partial sums (0 = none)
static int conv_fwd_impl(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, int y_f32_nchw,
                         float lrelu_slope, unsigned *bits_out, void *stream, int probe, float *stats = nullptr)
{
    if (int rc = check_desc(d, "conv2d_fwd")) return probe ? 0 : rc;
    if (!probe) M355_REQUIRE(x && w_fwd && y, "conv2d_fwd: null pointer");
#ifdef M355_EXACT
    if (probe) return 0;   // no bit masks, no fused statistics: the callers take their generic paths
    M355_REQUIRE(!bits_out && !stats, "conv2d_fwd: the EXACT build has no bit-mask / fused-statistics forward");
    return m355::exact::conv_fwd(d, x, w_fwd, bias, y, y_f32_nchw, lrelu_slope, (hipStream_t)stream);
#endif
    if (probe == 2 || stats) {
        if (m355::conv_small_eligible(d, y_f32_nchw) || (m355::conv_c8_eligible(d, y_f32_nchw) && !getenv("M355_NO_C8"))) {
            if (probe) return 0;
            m355::set_error("conv2d_fwd_stats: this shape has no fused statistics (ask m355_conv2d_fwd_stats_rows first)");
            return M355_ERR_BAD_ARG;
        }
    }
    if (m355::conv_small_eligible(d, y_f32_nchw)) {  // heads: 1..4 output channels, HBM-bound halo kernel
        if (probe) return 0;
        M355_REQUIRE(!bits_out, "conv2d_fwd: no bit masks on the small-Cout kernel");
        const int Kp = m355::k_padded(d->kh * d->kw * d->Cin);
        return m355::conv_small_launch(d, x, w_fwd, bias, y, lrelu_slope, Kp, (size_t)m355::rows_padded(d->Cout) * Kp * 2,
                                       (hipStream_t)stream);
    }
    if (m355::conv_c8_eligible(d, y_f32_nchw) && !getenv("M355_NO_C8")) {  // TextureDiscriminator.conv1: 8 input channels
        if (probe) return getenv("M355_NO_MASKBITS") ? 0 : 1;
        const int Kp = m355::k_padded(d->kh * d->kw * d->Cin);
        return m355::conv_c8_launch(d, x, w_fwd, bias, y, lrelu_slope, Kp, (size_t)m355::rows_padded(d->Cout) * Kp * 2, bits_out,
                                    (hipStream_t)stream);
    }
    ConvArgs a = {};
    a.x = (const unsigned short *)x;
    a.w = (const unsigned short *)w_fwd;
    a.bias = bias;
    a.y = y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin;
    a.ups = d->upsample;
    a.Hl = d->H << d->upsample; a.Wl = d->W << d->upsample;
    conv_out_hw(d, &a.Ho, &a.Wo);
    a.Cout = d->Cout;
    a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
    a.pad_w_mode = d->pad_w_mode;
    a.OH = a.Ho; a.OW = a.Wo; a.oy_mul = 1; a.ox_mul = 1; a.oy_off = 0; a.ox_off = 0;
    a.y_f32_nchw = y_f32_nchw; a.Cs = d->Cout;
    a.Kp = m355::k_padded(d->kh * d->kw * d->Cin);
    a.slope = lrelu_slope;
    a.bits_out = bits_out;
    if (!y_f32_nchw && !bits_out && probe != 1 && subpixel(d)) {
        // sub-pixel form: four 2x2 class convs of the STORED tensor, class (p, q) writing the output pixels (2i+p, 2j+q)
        a.w = w_fwd ? (const unsigned short *)w_fwd + up3_fwd_off(d) : nullptr;
        a.ups = 0; a.Hl = d->H; a.Wl = d->W;
        a.Ho = d->H; a.Wo = d->W; a.OH = 2 * d->H; a.OW = 2 * d->W; a.oy_mul = a.ox_mul = 2;
        a.KH = a.KW = 2; a.pad_h = a.pad_w = 1;
        a.Kp = m355::k_padded(4 * d->Cin);
        a.ncls = 4;
        a.cls_w_elems = (unsigned)((size_t)m355::rows_padded(d->Cout) * a.Kp);
        for (int c = 0; c < 4; ++c) {
            a.cpad_h[c] = 1 - (c >> 1); a.cpad_w[c] = 1 - (c & 1);
            a.coy[c] = c >> 1; a.cox[c] = c & 1;
        }
    }
    if (probe == 2 || stats) {
        ConvArgs b = a;
        b.CoutP = m355::rows_padded(b.Cout);
        if (b.ncls < 1) b.ncls = 1;
        const char *h = getenv("M355_CONV_HALO");
        const int rows = (m355::dma_eligible(b) && !(h && h[0] == '0')) ? m355::conv_halo_stats_rows(b) : 0;
        if (probe) return rows;
        if (!rows) {
            m355::set_error("conv2d_fwd_stats: this shape has no fused statistics (ask m355_conv2d_fwd_stats_rows first)");
            return M355_ERR_BAD_ARG;
        }
        a.stats = stats;
    }
    if (probe) return m355::halo_takes_bits(a) ? 1 : 0;
    return m355::launch_conv(a, (hipStream_t)stream);
}

/* Forward of a conv whose output feeds batch-norm statistics: besides y, every persistent workgroup writes the sums of its fp32
 * results and of their squares, part[rows][2][Cout] fp32 with rows = m355_conv2d_fwd_stats_rows(d) -- the layout
 * m355_bn_finalize reduces -- so the statistics cost no pass over y.  rows == 0: this shape has no fused statistics (run
 * m355_bn_stats_partial on y).  No activation epilogue (the statistics are those of the conv's own output). */
extern "C" int m355_conv2d_fwd_stats_rows(const m355_conv_desc *d)
{
    return conv_fwd_impl(d, nullptr, nullptr, nullptr, nullptr, 0, 1.0f, nullptr, nullptr, 2);
}

extern "C" int m355_conv2d_fwd_stats(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y,
                                     float *part, void *stream)
{
    M355_REQUIRE(part, "conv2d_fwd_stats: null pointer");
    return conv_fwd_impl(d, x, w_fwd, bias, y, 0, 1.0f, nullptr, stream, 0, part);
}

extern "C" int m355_conv2d_fwd(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y,
                               int y_f32_nchw, float lrelu_slope, void *stream)
{
    return conv_fwd_impl(d, x, w_fwd, bias, y, y_f32_nchw, lrelu_slope, nullptr, stream, 0);
}

// ConvArgs of a plain forward (what conv_fwd_impl builds for the generic path)
static ConvArgs fwd_args(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, float slope)
{
    ConvArgs a = {};
    a.x = (const unsigned short *)x;
    a.w = (const unsigned short *)w_fwd;
    a.bias = bias;
    a.y = y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin;
    a.ups = d->upsample;
    a.Hl = d->H << d->upsample; a.Wl = d->W << d->upsample;
    conv_out_hw(d, &a.Ho, &a.Wo);
    a.Cout = d->Cout;
    a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
    a.pad_w_mode = d->pad_w_mode;
    a.OH = a.Ho; a.OW = a.Wo; a.oy_mul = 1; a.ox_mul = 1;
    a.Cs = d->Cout;
    a.Kp = m355::k_padded(d->kh * d->kw * d->Cin);
    a.slope = slope;
    return a;
}

static int fwd_splitk(const m355_conv_desc *d)
{
#ifdef M355_EXACT
    return 0;
#endif
    if (!d || check_desc(d, "conv2d_fwd_ws")) return 0;
    if (m355::conv_small_eligible(d, 0) || m355::conv_c8_eligible(d, 0)) return 0;
    if (d->Cout % 8 || d->Cout > 2048 || 256 % (d->Cout / 8)) return 0;   // (the finishing pass: 8-channel vectors, 256 / vecs pixel lanes)
    if (subpixel(d)) return 0;   // (four class launches with 4/9 of the taps instead of K slices of the 9-tap form)
    return m355::splitk_plan(fwd_args(d, nullptr, nullptr, nullptr, nullptr, 1.0f));
}
static int fwd_splitk_ppb(const m355_conv_desc *d, int M)   // pixels per workgroup of the finishing pass: ~512 workgroups, whole lanes
{
    const int lanes = 256 / (d->Cout / 8);
    int ppb = (M + 511) / 512;
    ppb = (ppb + lanes - 1) / lanes * lanes;
    return ppb < lanes ? lanes : ppb;
}

/* Forward of a SMALL layer as split-K (m355_conv2d_fwd_ws_bytes(d) > 0: fewer 128 x 128 output tiles than CUs -- the generator's
 * 8x4 / 16x8 stages, gan.py:294-302 at :386-391): the K slices' fp32 accumulators go to `ws` and one finishing pass adds them in
 * slice order, applies bias / LeakyReLU, stores bf16 NHWC and -- part != NULL -- writes the batch-norm partial sums
 * part[m355_conv2d_fwd_ws_stats_rows(d)][2][Cout] of the fp32 results (what m355_bn_finalize reduces).  Deterministic. */
extern "C" size_t m355_conv2d_fwd_ws_bytes(const m355_conv_desc *d)
{
    const int S = fwd_splitk(d);
    if (!S) return 0;
    int Ho, Wo;
    conv_out_hw(d, &Ho, &Wo);
    return sizeof(float) * (size_t)S * d->N * Ho * Wo * d->Cout;
}

extern "C" int m355_conv2d_fwd_ws_stats_rows(const m355_conv_desc *d)
{
    if (!fwd_splitk(d) || d->Cout > 2048 || 256 % (d->Cout / 8)) return 0;
    int Ho, Wo;
    conv_out_hw(d, &Ho, &Wo);
    const int M = d->N * Ho * Wo, ppb = fwd_splitk_ppb(d, M);
    return (M + ppb - 1) / ppb;
}

extern "C" int m355_conv2d_fwd_ws(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y,
                                  float lrelu_slope, void *ws, float *stats_part, void *stream)
{
    M355_REQUIRE(x && w_fwd && y && ws, "conv2d_fwd_ws: null pointer");
    const int S = fwd_splitk(d);
    M355_REQUIRE(S >= 2 && 256 % (d->Cout / 8) == 0, "conv2d_fwd_ws: this layer has no split-K form (m355_conv2d_fwd_ws_bytes)");
    ConvArgs a = fwd_args(d, x, w_fwd, nullptr, y, 1.0f);
    a.sk = S;
    a.skws = (float *)ws;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = m355::launch_conv(a, st)) return rc;
    const int M = d->N * a.Ho * a.Wo, ppb = fwd_splitk_ppb(d, M);
    hipLaunchKernelGGL(m355::k_splitk_finish, dim3((M + ppb - 1) / ppb), dim3(256), 0, st, (const float *)ws, S, (size_t)M * d->Cout, M,
                       d->Cout, d->Cout, bias, lrelu_slope, (unsigned short *)y, stats_part, ppb);
    return m355::check_launch("conv2d_fwd_ws (finish)");
}

/* forward with a LeakyReLU epilogue that also writes the activation's sign bits (1 bit per element,
 * [N,Ho,Wo,Cout/64,2] uint32) for the consumer's m355_conv2d_dgrad_bits; only where m355_conv2d_maskbits_ok(d, 0) */
extern "C" int m355_conv2d_fwd_bits(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y,
                                    float lrelu_slope, void *mask_bits_out, void *stream)
{
    M355_REQUIRE(mask_bits_out, "conv2d_fwd_bits: null pointer");
    return conv_fwd_impl(d, x, w_fwd, bias, y, 0, lrelu_slope, (unsigned *)mask_bits_out, stream, 0);
}

// dy[N,Ho,Wo,Cout_p32] bf16 (channel stride = ceil32(Cout), padding channels zero) -> dx[N,H,W,Cin] bf16.
// ws >= m355_conv2d_dgrad_ws_bytes(d): the gradient in the padded / upsampled frame before folding.
// does the dgrad of this layer run in the direct form (writes dx itself: no padded frame, no fold pass)?
static bool dgrad_direct(const m355_conv_desc *d)
{
#ifdef M355_EXACT
    return true;   // (the gather kernel writes dx itself)
#endif
    int Ho, Wo;
    if (conv_out_hw(d, &Ho, &Wo) != 0) return false;
    if (subpixel(d)) return true;
    const int cy = m355::dy_channels(d->Cout);
    ConvArgs a = {};
    a.N = d->N; a.H = Ho; a.W = Wo; a.Cin = cy; a.Cout = d->Cin; a.Cs = d->Cin;
    a.Kp = m355::k_padded((d->stride == 1 ? d->kh * d->kw : ((d->kh + 1) / 2) * ((d->kw + 1) / 2)) * cy);
    if (m355::dgrad_direct_replicate_eligible(d, cy) && m355::dma_eligible(a)) return true;  // 3x3 replicate (+upsample)
    return !d->upsample && d->pad_w_mode != 1 && m355::dma_eligible(a) &&
           (d->stride == 1 || (d->H % 2 == 0 && d->W % 2 == 0 && d->kh % 2 == 0 && d->kw % 2 == 0));
}

extern "C" size_t m355_conv2d_dgrad_ws_bytes(const m355_conv_desc *d)
{
    if (!d) return 0;
    if (dgrad_direct(d)) return 0;
    const size_t Hl = (size_t)d->H << d->upsample, Wl = (size_t)d->W << d->upsample;
    return (size_t)d->N * (Hl + 2 * d->pad_h) * (Wl + 2 * d->pad_w) * d->Cin * 2;
}

static int conv_dgrad_impl(const m355_conv_desc *d, const void *dy, const void *w_dgrad, void *dx, void *ws, const void *mask_x,
                           const unsigned *mask_bits, float mask_slope, void *stream, int probe, int lead = 0)
{
    if (int rc = check_desc(d, "conv2d_dgrad")) return probe ? 0 : rc;
    if (!probe) M355_REQUIRE(dy && w_dgrad && dx, "conv2d_dgrad: null pointer");
    hipStream_t st = (hipStream_t)stream;
#ifdef M355_EXACT
    if (probe) return 0;
    M355_REQUIRE(!mask_bits, "conv2d_dgrad: the EXACT build reads no bit masks");
    return m355::exact::conv_dgrad(d, dy, m355::dy_channels(d->Cout), w_dgrad, dx, mask_x, mask_slope, st);
#endif
    int Ho, Wo;
    conv_out_hw(d, &Ho, &Wo);
    const int Hl = d->H << d->upsample, Wl = d->W << d->upsample;
    const int cout32 = m355::dy_channels(d->Cout), cin64 = m355::rows_padded(d->Cin);
    // gradient frame: rows = logical rows only for stride 1 (H pad cropped via pad_h'), all padded rows for stride 2
    ConvArgs a = {};
    a.x = (const unsigned short *)dy;
    a.N = d->N; a.H = Ho; a.W = Wo; a.Cin = cout32; a.Hl = Ho; a.Wl = Wo; a.ups = 0;
    a.Cout = d->Cin; a.pad_w_mode = 0; a.stride = 1;
    a.y_f32_nchw = 0; a.Cs = d->Cin; a.slope = 1.0f;
    // DIRECT form (no padded frame, no fold): without upsample and with a zero or circular W pad the adjoint of
    // the padding is an index map on dy -- a circularly padded conv's dgrad is a circular conv of dy.
    a.Kp = m355::k_padded((d->stride == 1 ? d->kh * d->kw : ((d->kh + 1) / 2) * ((d->kw + 1) / 2)) * cout32);
    const bool direct = dgrad_direct(d);
    if (subpixel(d)) {
        // upsample + 3x3 in the sub-pixel form: dx = the adjoint 4x4 stride-2 conv of dy (kernel W4[3-kh][3-kw], pad 1) on the
        // stride-2 forward variant of k_conv_halo; a circular W pad of x is a circular pad of dy, a replicate pad adds the
        // gradient of the two pad columns to the first / last column of dx (k_dgrad_edge)
        if (probe) return 0;
        M355_REQUIRE(!mask_x && !mask_bits, "conv2d_dgrad: no fused activation backward on the sub-pixel form");
        a.x = (const unsigned short *)dy;
        a.w = (const unsigned short *)w_dgrad + up3_dgrad_off(d);
        a.y = dx;
        a.H = Ho; a.W = Wo; a.Hl = Ho; a.Wl = Wo;
        a.Ho = d->H; a.Wo = d->W; a.OH = d->H; a.OW = d->W; a.oy_mul = a.ox_mul = 1;
        a.KH = a.KW = 4; a.stride = 2; a.pad_h = a.pad_w = 1;
        a.pad_w_mode = d->pad_w_mode == 2 ? 2 : 0;
        a.Kp = m355::k_padded(16 * cout32);
        if (int rc = m355::launch_conv(a, st)) return rc;
        if (d->pad_w_mode == 1) return m355::dgrad_edge_up4_launch(d, dy, cout32, a.w, a.Kp, dx, st);
        return M355_OK;
    }
    if (probe && (!direct || d->pad_w_mode == 1)) return 0;
    const bool c8rep = !mask_bits && m355::dgrad_c8_replicate_eligible(d, cout32);   // 5x5 heads of the symmetric generator
    M355_REQUIRE((!mask_x && !mask_bits) || direct || c8rep,
                 "conv2d_dgrad: the fused activation backward needs the direct form (no upsample, no replicate pad, tensors < 2 GiB)");
    a.mask_x = (const unsigned short *)mask_x;
    a.bits_in = mask_bits;
    a.mask_slope = mask_slope;
    int rc = 0;
    if (direct && d->pad_w_mode == 1) {  // generator 3x3: zero-pad conv of dy on the halo kernel + replicate edge terms
        M355_REQUIRE(!mask_x && !mask_bits, "conv2d_dgrad: no fused activation backward on the replicate form");
        return m355::dgrad_direct_replicate_launch(d, dy, cout32, w_dgrad, a.Kp, cin64, dx, st);
    }
    if (!probe && direct && !mask_x && !mask_bits && m355::dgrad_small_eligible(d, cout32) && !getenv("M355_NO_C8"))
        return m355::dgrad_small_launch(d, dy, cout32, w_dgrad, a.Kp, (size_t)cin64 * a.Kp * 2, dx, st, lead);
    // 5x5 "same" convs with <= 8 output channels and a zero / circular W pad (TextureDiscriminator.conv5): the gradient is a
    // conv of the 8-channel dy onto Cin channels with the same pad mode -- k_conv_c8, with the activation backward of the
    // producer of x (mask_x) in its epilogue (k_conv_glds pads the 1..8 channels of dy to a 64-wide K step)
    if (!probe && direct && d->stride == 1 && !mask_bits && d->pad_w_mode != 1 && cout32 == 8 && d->kh == 5 && d->kw == 5 &&
        d->pad_h == 2 && d->pad_w == 2 && !d->upsample && !getenv("M355_NO_C8_DGRAD")) {
        m355_conv_desc t = *d;
        t.Cin = 8;
        t.Cout = d->Cin;
        if (m355::conv_c8_eligible(&t, 0))
            return m355::conv_c8_launch(&t, dy, w_dgrad, nullptr, dx, 1.0f, a.Kp, (size_t)cin64 * a.Kp * 2, nullptr, st, mask_x,
                                        mask_slope);
    }
    if (direct && d->stride == 1) {
        a.w = (const unsigned short *)w_dgrad;
        a.KH = d->kh; a.KW = d->kw;
        a.pad_h = d->kh - 1 - d->pad_h; a.pad_w = d->kw - 1 - d->pad_w;
        a.pad_w_mode = d->pad_w_mode;
        a.Ho = d->H; a.Wo = d->W; a.OH = a.Ho; a.OW = a.Wo; a.oy_mul = a.ox_mul = 1;
        a.y = dx;
        if (probe) return m355::halo_takes_bits(a) ? 1 : 0;
        return m355::launch_conv(a, st);
    }
    if (direct) {
        const int A = d->kh / 2, B = d->kw / 2;
        a.w = (const unsigned short *)w_dgrad;
        a.KH = A; a.KW = B;
        a.pad_w_mode = d->pad_w_mode;
        a.Ho = d->H / 2; a.Wo = d->W / 2; a.OH = d->H; a.OW = d->W; a.oy_mul = a.ox_mul = 2;
        a.ncls = 4;
        a.cls_w_elems = (unsigned)((size_t)cin64 * a.Kp);
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                const int c = py * 2 + px, ioff = (d->pad_h - py + 1) >> 1, joff = (d->pad_w - px + 1) >> 1;
                a.cpad_h[c] = A - 1 - ioff; a.cpad_w[c] = B - 1 - joff;
                a.coy[c] = 2 * ioff + py - d->pad_h; a.cox[c] = 2 * joff + px - d->pad_w;
            }
        a.y = dx;
        if (probe) return m355::halo_takes_bits(a) ? 1 : 0;
        return m355::launch_conv(a, st);
    }
    if (probe) return 0;
    if (c8rep)
        return m355::dgrad_c8_replicate_launch(d, dy, w_dgrad, m355::k_padded(d->kh * d->kw * cout32),
                                               (size_t)cin64 * m355::k_padded(d->kh * d->kw * cout32) * 2, dx, st, mask_x, mask_slope);
    const bool need_fold = d->upsample || (d->pad_w_mode != 0 && d->pad_w > 0) || d->stride == 2;
    M355_REQUIRE(!need_fold || ws, "conv2d_dgrad: workspace required");
    if (d->stride == 1) {
        const int pw_keep = need_fold ? d->pad_w : 0;  // keep W pad columns in the frame only if they must be folded
        a.w = (const unsigned short *)w_dgrad;
        a.KH = d->kh; a.KW = d->kw;
        a.Kp = m355::k_padded(d->kh * d->kw * cout32);
        a.pad_h = d->kh - 1 - d->pad_h;
        a.pad_w = d->kw - 1 - (d->pad_w - pw_keep);
        a.Ho = Hl; a.Wo = Wl + 2 * pw_keep;
        a.OH = a.Ho; a.OW = a.Wo; a.oy_mul = a.ox_mul = 1; a.oy_off = a.ox_off = 0;
        a.y = need_fold ? ws : dx;
        rc = m355::launch_conv(a, st);
        if (rc) return rc;
        if (need_fold) {
            const size_t total = (size_t)d->N * d->H * d->W * (d->Cin / 8);
            hipLaunchKernelGGL(m355::k_fold_pad, dim3((unsigned)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256)),
                               dim3(256), 0, st, (const unsigned short *)ws, (unsigned short *)dx, d->N, d->H, d->W, d->Cin,
                               d->upsample, d->pad_w, d->pad_w_mode, Hl, 0);
            rc = m355::check_launch("conv2d_dgrad fold");
        }
        return rc;
    }
    // stride 2: four parity classes of the padded frame, each a (kh/2 x kw/2) conv of dy written with stride 2
    const int Hp = Hl + 2 * d->pad_h, Wp = Wl + 2 * d->pad_w;
    M355_REQUIRE(Hp % 2 == 0 && Wp % 2 == 0, "conv2d_dgrad: stride-2 frame %dx%d must be even", Hp, Wp);
    const int A = (d->kh + 1) / 2, B = (d->kw + 1) / 2;  // (odd kernels: the next even size, see fill_weight_prep)
    const size_t each = (size_t)cin64 * m355::k_padded(A * B * cout32);
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            a.w = (const unsigned short *)w_dgrad + (size_t)(py * 2 + px) * each;
            a.KH = A; a.KW = B;
            a.Kp = m355::k_padded(A * B * cout32);
            a.pad_h = A - 1; a.pad_w = B - 1;
            a.Ho = Hp / 2; a.Wo = Wp / 2;
            a.OH = Hp; a.OW = Wp; a.oy_mul = 2; a.ox_mul = 2; a.oy_off = py; a.ox_off = px;
            a.y = ws;
            rc = m355::launch_conv(a, st);
            if (rc) return rc;
        }
    const size_t total = (size_t)d->N * d->H * d->W * (d->Cin / 8);
    hipLaunchKernelGGL(m355::k_fold_pad, dim3((unsigned)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256)), dim3(256),
                       0, st, (const unsigned short *)ws, (unsigned short *)dx, d->N, d->H, d->W, d->Cin, d->upsample,
                       d->pad_w, d->pad_w_mode, Hp, d->pad_h);
    return m355::check_launch("conv2d_dgrad fold");
}

extern "C" int m355_conv2d_dgrad(const m355_conv_desc *d, const void *dy, const void *w_dgrad, void *dx, void *ws,
                                 const void *mask_x, float mask_slope, void *stream)
{
    return conv_dgrad_impl(d, dy, w_dgrad, dx, ws, mask_x, nullptr, mask_slope, stream, 0);
}

/* dgrad of a layer whose input channels beyond the first `lead` are CONSTANTS of the model (TextureDiscriminator.conv1,
 * /root/reference/code/models/gan.py:204-213: image channels + the batch-constant positional encodings of gan.py:9-20): nothing reads their
 * gradient, so only dx[..., 0 .. lead-1] is specified -- the other channels of dx hold zeros or the true gradient, whichever the
 * dispatched kernel produces.  Same arguments as m355_conv2d_dgrad (no activation mask). */
extern "C" int m355_conv2d_dgrad_lead(const m355_conv_desc *d, const void *dy, const void *w_dgrad, void *dx, void *ws, int lead,
                                      void *stream)
{
    M355_REQUIRE(d && lead >= 1 && lead <= d->Cin, "conv2d_dgrad_lead: lead=%d outside 1..Cin", lead);
    return conv_dgrad_impl(d, dy, w_dgrad, dx, ws, nullptr, nullptr, 1.0f, stream, 0, lead);
}

/* dgrad whose epilogue applies the producer's LeakyReLU backward from the bit masks m355_conv2d_fwd_bits wrote
 * ([N,H,W,Cin/64,2] uint32, this conv's input frame); only where m355_conv2d_maskbits_ok(d, 1) */
extern "C" int m355_conv2d_dgrad_bits(const m355_conv_desc *d, const void *dy, const void *w_dgrad, void *dx, void *ws,
                                      const void *mask_bits, float mask_slope, void *stream)
{
    M355_REQUIRE(mask_bits, "conv2d_dgrad_bits: null pointer");
    return conv_dgrad_impl(d, dy, w_dgrad, dx, ws, nullptr, (const unsigned *)mask_bits, mask_slope, stream, 0);
}

/* can m355_conv2d_dgrad apply a fused activation backward (mask_x) on this layer? */
extern "C" int m355_conv2d_dgrad_mask_ok(const m355_conv_desc *d)
{
    if (!d || check_desc(d, "conv2d_dgrad_mask_ok")) return 0;
#ifdef M355_EXACT
    return 1;
#endif
    if (m355::dgrad_c8_replicate_eligible(d, m355::dy_channels(d->Cout))) return 1;
    return dgrad_direct(d) && d->pad_w_mode != 1 && !subpixel(d);
}

/* role 0: can the forward of this layer (bf16 NHWC output, activation epilogue) write bit masks?
 * role 1: can the dgrad of this layer read the bit masks of its input's producer? */
extern "C" int m355_conv2d_maskbits_ok(const m355_conv_desc *d, int role)
{
    if (!d) return 0;
    if (role == 0) return conv_fwd_impl(d, nullptr, nullptr, nullptr, nullptr, 0, 0.2f, nullptr, nullptr, 1);
    return conv_dgrad_impl(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.2f, nullptr, 1);
}

// =====================================================================================================
// wgrad: dw[co][kh][kw][ci] = sum over pixels p of dy[p][co] * xpatch[p][(kh,kw,ci)]       (fp32 accumulate)
// GEMM with the PIXEL axis as K.  Both operands are stored pixel-major (channels contiguous) while an MFMA fragment
// needs 8 consecutive K (= pixels) of one channel, so every operand tile is transposed on its way into LDS:
// a thread loads the same 8-channel granule of 4 consecutive pixels (4 x 16 B), transposes the 8x4 block in
// registers with v_perm_b32 and writes 8 ds_write_b64 (one per channel, 4 pixels each) into a channel-major tile
// [channel][64 pixels].  Lanes of a 16-lane group write 128 contiguous bytes of one row (conflict free); the 144-byte
// row pitch keeps the ds_read_b128 fragment reads conflict free.
// Tile: TM (64|128) output channels x 128 weight columns x 64 pixels per step; 4 waves as 2x2; the global loads of
// step s+1 are in flight under the MFMAs of step s; split-K over the pixel axis across gridDim.z with fp32
// atomicAdd of the partial tiles.
namespace m355 {

constexpr int WN = 128, WK = 64;
constexpr int WLD = WK + 8;  // LDS row pitch in bf16 (144 bytes)


// 4 granules (pixels q..q+3, channels c..c+7) -> 8 x 8 bytes: for channel j the 4 pixel values
__device__ __forceinline__ void transpose_store(const bf16x8 (&g)[4], unsigned short *row0 /* &T[c][4*pq] */)
{
    const unsigned int *w0 = reinterpret_cast<const unsigned int *>(&g[0]);
    const unsigned int *w1 = reinterpret_cast<const unsigned int *>(&g[1]);
    const unsigned int *w2 = reinterpret_cast<const unsigned int *>(&g[2]);
    const unsigned int *w3 = reinterpret_cast<const unsigned int *>(&g[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // word k holds channels 2k (low half) and 2k+1 (high half)
        uint2 lo, hi;
        lo.x = __builtin_amdgcn_perm(w1[k], w0[k], 0x05040100u);  // (p0, p1) of channel 2k
        lo.y = __builtin_amdgcn_perm(w3[k], w2[k], 0x05040100u);  // (p2, p3)
        hi.x = __builtin_amdgcn_perm(w1[k], w0[k], 0x07060302u);  // channel 2k+1
        hi.y = __builtin_amdgcn_perm(w3[k], w2[k], 0x07060302u);
        *reinterpret_cast<uint2 *>(row0 + (2 * k) * WLD) = lo;
        *reinterpret_cast<uint2 *>(row0 + (2 * k + 1) * WLD) = hi;
    }
}

template <int TM, bool DET = false>
__global__ __launch_bounds__(256) void k_wgrad_mfma(WgradArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned short At[TM * WLD];  // [co][pixel]
    __shared__ __attribute__((aligned(16))) unsigned short Bt[WN * WLD];  // [col][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = blockIdx.x * TM, col0 = blockIdx.y * WN;
    const int P = a.N * a.Ho * a.Wo, K = a.KH * a.KW * a.Cin, HW = a.Ho * a.Wo;
    const int pbeg = blockIdx.z * a.chunk, pend = min(P, pbeg + a.chunk);
    if (pbeg >= pend) return;

    // staging role: pixel quad pq (4 consecutive pixels) x 8-channel granule gi
    const int pq = lane & 15, gi = wave * 4 + (lane >> 4);
    const int xcol = col0 + gi * 8;
    const bool xcol_ok = xcol < K;
    const int tap = xcol_ok ? xcol / a.Cin : 0, ci = xcol_ok ? xcol - tap * a.Cin : 0;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int yco = co0 + gi * 8;
    const bool y_ok = gi * 8 < TM && yco < a.Cy;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    bf16x8 gx[4], gy[4];
    auto load_step = [&](int p0) {
        int p = p0 + 4 * pq;
        int n = p / HW, r = p - n * HW;
        int ho = r / a.Wo, wo = r - ho * a.Wo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool pin = p < pend;
            bool ok = xcol_ok && pin;
            const int hi = ho * a.stride - a.pad_h + kh;
            int wi = wo * a.stride - a.pad_w + kw;
            ok = ok && hi >= 0 && hi < a.Hl;
            if (a.pad_w_mode == 1) wi = min(max(wi, 0), a.Wl - 1);
            else if (a.pad_w_mode == 2) wi = wi < 0 ? wi + a.Wl : (wi >= a.Wl ? wi - a.Wl : wi);
            else ok = ok && wi >= 0 && wi < a.Wl;
            const unsigned short *src = a.x + (((size_t)n * a.H + (hi >> a.ups)) * a.W + (wi >> a.ups)) * a.Cin + ci;
            gx[i] = ok ? *reinterpret_cast<const bf16x8 *>(src) : zero8;
            gy[i] = (y_ok && pin) ? *reinterpret_cast<const bf16x8 *>(a.dy + (size_t)p * a.Cy + yco) : zero8;
            ++p;
            if (++wo == a.Wo) {
                wo = 0;
                if (++ho == a.Ho) {
                    ho = 0;
                    ++n;
                }
            }
        }
    };
    auto store_step = [&]() {
        transpose_store(gx, &Bt[(gi * 8) * WLD + 4 * pq]);
        if (gi * 8 < TM) transpose_store(gy, &At[(gi * 8) * WLD + 4 * pq]);
    };

    constexpr int MI = TM / 64;  // 32-row fragments per wave along co
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 31, fk = (lane >> 5) * 8;

    load_step(pbeg);
    for (int p0 = pbeg; p0 < pend; p0 += WK) {
        __syncthreads();  // the previous step's fragments are consumed
        store_step();
        __syncthreads();
        if (p0 + WK < pend) load_step(p0 + WK);  // in flight under the MFMAs
#pragma unroll
        for (int kk = 0; kk < WK; kk += 16) {
            bf16x8 af[MI], bfr[2];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = *reinterpret_cast<const bf16x8 *>(&At[(wr * (TM / 2) + 32 * i + frow) * WLD + kk + fk]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8 *>(&Bt[(wc * 64 + 32 * j + frow) * WLD + kk + fk]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    // ---- epilogue: row = co, col = weight column
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wr * (TM / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = col0 + wc * 64 + 32 * j + (lane & 31);
                if (co < a.Cout && col < K) wg_accum<DET>(a.dw, a.fix, (size_t)co * K + col, acc[i][j][r], (size_t)a.Cout * K + a.Cout);
            }
}


// -----------------------------------------------------------------------------------------------------
// k_wgrad_dma: the same GEMM with both operand tiles filled by the LDS-DMA in their NATURAL pixel-major form
// (a tile row = one pixel, channels contiguous) and the pixel-axis fragments produced by the gfx950 transpose
// read ds_read_b64_tr_b16: a 16-lane group reads a [4 pixels][16 channels] block (lane q supplies the address of
// pixel q>>2, channels 4(q&3)..+3) and lane i receives channel i of the 4 pixels (scripts/probes/tr16.hip) -- no
// register transposes, no ds_write pass.  Two such reads (4 + 4 pixels) make one 8-deep MFMA operand; the
// pixel <-> k-slot assignment is the same for both operands, which is all the contraction needs.
//   * tile rows are 128 / 256 / 512 bytes; the 16-byte chunk index is XORed with 4*(pixel&3) (256/512-byte rows)
//     or 4*((pixel>>1)&1) (128-byte rows) on the DMA SOURCE side and on the read side, which spreads the 32
//     eight-byte pieces of a half-wave over all 64 banks.
//   * a lane's x chunk is a fixed (tap, ci) for the whole kernel; only the pixel advances (64 per K step), and with
//     Ho, Wo powers of two its (n, ho, wo) are shifts and masks.
//   * tiles: Cout > 64: 128 (co) x 128 (weight columns), waves 2x2;  Cout <= 64: 64 x 256, waves 1x4; each wave
//     64x64 = 2x2 MFMA 32x32x16; 2 LDS stages (64 / 80 KiB) -> 2 workgroups per CU; split-K over pixels (gridDim.z).
typedef short s4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4v lds_s4v;

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char *p, int off0, int off1)
{
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v *)(p + off0));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v *)(p + off1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct WgradDmaArgs {
    WgradArgs w;
    int lgWo, lgHo;
    unsigned xbytes, ybytes;
};

template <int TM, int TN, int MODE, bool DET = false>
__global__ __launch_bounds__(256, 2) void k_wgrad_dma(WgradDmaArgs A)
{
    const WgradArgs &a = A.w;
    constexpr int RBY = TM * 2, RBX = TN * 2;              // tile row bytes
    constexpr int YB = 64 * RBY, XB = 64 * RBX, STAGE = YB + XB;
    constexpr int NIY = YB / 4096, NIX = XB / 4096;        // DMA instructions per wave per stage
    constexpr int CPY = RBY / 16, RPIY = 64 / CPY;         // chunks per row, rows per DMA instruction
    constexpr int CPX = RBX / 16, RPIX = 64 / CPX;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.x * TM, col0 = blockIdx.y * TN;
    const int P = a.N * a.Ho * a.Wo, K = a.KH * a.KW * a.Cin;
    const int pbeg = blockIdx.z * a.chunk, pend = min(P, pbeg + a.chunk);
    if (pbeg >= pend) return;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, A.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, A.ybytes, 0x00020000);

    // ---- DMA roles: instruction q = 4i + wave of a tile fills LDS bytes [1024q, 1024q + 1024)
    const int ryl = lane / CPY, rxl = lane / CPX;  // row within the instruction's RPI rows
    const int swzy = RBY == 128 ? 4 * ((ryl >> 1) & 1) : 4 * (ryl & 3);
    const int swzx = RBX == 512 ? 4 * (2 * (wave & 1) + rxl) : 4 * (rxl & 3);
    const int yco = co0 + ((lane % CPY) ^ swzy) * 8;
    const bool yok = yco < a.Cy;
    const int xcol = col0 + ((lane % CPX) ^ swzx) * 8;
    const bool xok = xcol < K;
    const int tap = xok ? xcol / a.Cin : 0;
    const int ci = xok ? xcol - tap * a.Cin : 0;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int hoff = kh - a.pad_h, woff = kw - a.pad_w;
    const unsigned Cy2 = a.Cy * 2, Cin2 = a.Cin * 2;

    auto stage = [&](int pb, int buf) {
        unsigned char *dY = lds + buf * STAGE + wave * 1024, *dX = dY + YB;
#pragma unroll
        for (int i = 0; i < NIY; ++i) {
            const int p = pb + RPIY * (4 * i + wave) + ryl;
            const unsigned off = (unsigned)p * Cy2 + (unsigned)yco * 2u;
            dma16(ry, dY + i * 4096, (yok && p < pend) ? off : OOB, 0u);
        }
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int p = pb + RPIX * (4 * i + wave) + rxl;
            const int wo = p & (a.Wo - 1), ho = (p >> A.lgWo) & (a.Ho - 1), n = p >> (A.lgWo + A.lgHo);
            const int hi = ho * a.stride + hoff;
            int wi = wo * a.stride + woff;
            bool ok = xok && p < pend && (unsigned)hi < (unsigned)a.Hl;
            if (MODE == 1) wi = min(max(wi, 0), a.Wl - 1);
            else if (MODE == 2) wi = wi < 0 ? wi + a.Wl : (wi >= a.Wl ? wi - a.Wl : wi);
            else ok = ok && (unsigned)wi < (unsigned)a.Wl;
            const unsigned off = (unsigned)((n * a.H + (hi >> a.ups)) * a.W + (wi >> a.ups)) * Cin2 + (unsigned)ci * 2u;
            dma16(rx, dX + i * 4096, ok ? off : OOB, 0u);
        }
    };

    f32x16 acc[2][2];  // [co block i][column block j]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- fragment addressing.  Lane l: q = l&15 -> pixel (q>>2) of the 4-pixel group, channel quad q&3 of the
    // 16-channel strip g16 = (l>>4)&1; half hh = l>>5 takes pixels 8hh..8hh+7 of each 16-pixel k group.
    const int wr = TM == 128 ? wave >> 1 : 0, wc = TM == 128 ? wave & 1 : wave;
    const int q = lane & 15, g16 = (lane >> 4) & 1, hh = lane >> 5;
    const int rdswzy = RBY == 128 ? 4 * ((q >> 3) & 1) : 4 * (q >> 2);
    const int rdswzx = 4 * (q >> 2);
    const int rowl = hh * 8 + (q >> 2);
    int ya[2], xb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ya[i] = rowl * RBY + (((wr * 8 + 4 * i + g16 * 2 + ((q & 3) >> 1)) ^ rdswzy) << 4) + (q & 1) * 8;
        xb[i] = YB + rowl * RBX + (((wc * 8 + 4 * i + g16 * 2 + ((q & 3) >> 1)) ^ rdswzx) << 4) + (q & 1) * 8;
    }

    // bias gradient for free: the workgroups of the first column tile also sum the dy tile they stage anyway
    // (thread -> channel tid % TM, pixel group tid / TM); rows beyond pend were zero-filled by the DMA
    const bool do_db = a.db != nullptr && blockIdx.y == 0;
    constexpr int DBG = 256 / TM, DBP = 64 / DBG;
    const int dbc = tid % TM, dbq = tid / TM;
    float dbacc = 0.0f;

    stage(pbeg, 0);
    __syncthreads();
    int buf = 0;
    for (int pb = pbeg; pb < pend; pb += 64, buf ^= 1) {
        if (pb + 64 < pend) stage(pb + 64, buf ^ 1);
        const unsigned char *base = lds + buf * STAGE;
        if (do_db) {
#pragma unroll 8
            for (int pp = 0; pp < DBP; ++pp) {
                const int prow = dbq * DBP + pp;
                const int sw = RBY == 128 ? 4 * ((prow >> 1) & 1) : 4 * (prow & 3);
                dbacc += bf2f(*reinterpret_cast<const unsigned short *>(base + prow * RBY + (((dbc >> 3) ^ sw) << 4) + (dbc & 7) * 2));
            }
        }
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const bf16x8 a0 = tr_pair(base + ya[0], kg * 16 * RBY, (kg * 16 + 4) * RBY);
            const bf16x8 a1 = tr_pair(base + ya[1], kg * 16 * RBY, (kg * 16 + 4) * RBY);
            const bf16x8 b0 = tr_pair(base + xb[0], kg * 16 * RBX, (kg * 16 + 4) * RBX);
            const bf16x8 b1 = tr_pair(base + xb[1], kg * 16 * RBX, (kg * 16 + 4) * RBX);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wr * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = col0 + wc * 64 + 32 * j + (lane & 31);
                if (co < a.Cout && col < K) wg_accum<DET>(a.dw, a.fix, (size_t)co * K + col, acc[i][j][r], (size_t)a.Cout * K + a.Cout);
            }
    if (do_db && co0 + dbc < a.Cout) wg_accum<DET>(a.db, a.fix, (DET ? (size_t)a.Cout * K : 0) + co0 + dbc, dbacc, (size_t)a.Cout * K + a.Cout);
}

}  // namespace m355

// x[N,H,W,Cin] bf16, dy[N,Ho,Wo,dy_channels(Cout)] bf16 -> dw fp32 [Cout][kh][kw][Cin] (overwritten).
static bool wgrad_dma_ok(const m355_conv_desc *d)
{
    int Ho, Wo;
    if (conv_out_hw(d, &Ho, &Wo) != 0) return false;
    const size_t xbytes = (size_t)d->N * d->H * d->W * d->Cin * 2;
    const size_t ybytes = (size_t)d->N * Ho * Wo * m355::dy_channels(d->Cout) * 2;
    return d->Cout >= 64 && m355::ilog2_exact(Wo) >= 0 && m355::ilog2_exact(Ho) >= 0 && xbytes < (1ull << 31) &&
           ybytes < (1ull << 31);
}

/* 1 when m355_conv2d_wgrad can also produce the bias gradient (column sums of dy) for this layer */
static bool wgrad_has_dbias(const m355_conv_desc *d)
{
#ifdef M355_EXACT
    return true;
#endif
    return wgrad_dma_ok(d) || m355::wgrad_c8_eligible(d, m355::dy_channels(d->Cout));
}
extern "C" int m355_conv2d_wgrad_fuses_dbias(const m355_conv_desc *d) { return d && wgrad_has_dbias(d) ? 1 : 0; }

static int conv_wgrad_impl(const m355_conv_desc *d, const void *x, const void *dy, float *dw, float *dbias, void *stream,
                           bool zero, long long *fix = nullptr, float *part = nullptr);

namespace m355 {
// fix = [flag | n fixed-point cells | nb fixed-point bias cells] (kFixCell integers each, conv_dma.h wg_accum) -> dw[n], db[nb] fp32 (one
// rounding per element); a raised flag
// (some partial tile was not finite) poisons the whole result with NaN, as an fp32 accumulation would have
__global__ __launch_bounds__(256) void k_fix_to_f32(const long long *__restrict__ fix, float *__restrict__ dw, size_t n,
                                                    float *__restrict__ db, int nb)
{
    const bool bad = fix[0] != 0;
    const size_t total = n + (db ? (size_t)nb : 0), ncell = n + (size_t)nb;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float v = bad ? __builtin_nanf("") : (float)fix_value(fix, i, ncell);
        if (i < n) dw[i] = v;
        else db[i - n] = v;
    }
}
}  // namespace m355

namespace m355 {
// sub-pixel weight gradient (see `subpixel`): src = the 16-entry effective gradient dE[co][eh][ew][ci] (+ Cout bias sums behind it),
// fp32 or -- DET -- the fixed-point cells behind the flag word; dw[co][kh][kw][ci] = sum of dE over eh in {kh, kh+1}, ew in
// {kw, kw+1}: the adjoint of the weights' pre-sum W4[e] = w[e-1] + w[e].  DET adds the integers and rounds once.
template <bool DET>
__global__ __launch_bounds__(256) void k_up16_to_9(const void *__restrict__ src, float *__restrict__ dw, float *__restrict__ db,
                                                   int Cout, int Cin)
{
    const float *sf = reinterpret_cast<const float *>(src);
    const long long *sx = reinterpret_cast<const long long *>(src);
    const bool bad = DET && sx[0] != 0;
    const size_t n9 = (size_t)Cout * 9 * Cin, n16 = (size_t)Cout * 16 * Cin, total = n9 + (db ? (size_t)Cout : 0), ncell = n16 + Cout;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        if (i >= n9) {
            db[i - n9] = DET ? (bad ? __builtin_nanf("") : (float)fix_value(sx, n16 + (i - n9), ncell)) : sf[n16 + (i - n9)];
            continue;
        }
        const int ci = (int)(i % Cin);
        const size_t t = i / Cin;
        const int tap = (int)(t % 9), co = (int)(t / 9), kh = tap / 3, kw = tap - 3 * kh;
        const size_t b = (((size_t)co * 4 + kh) * 4 + kw) * Cin + ci;   // entry (kh, kw); (kh+1, .) is 4 Cin further, (., kw+1) Cin
        if (DET) {
            const size_t e[4] = {b, b + Cin, b + 4 * (size_t)Cin, b + 5 * (size_t)Cin};
            long long big = 0, mid = 0, small = 0;   // (the four entries' integers are added BEFORE the one rounding)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mid += sx[1 + e[k]];
                big += sx[1 + ncell + e[k]];
                small += sx[1 + 2 * ncell + e[k]];
            }
            dw[i] = bad ? __builtin_nanf("") : (float)fix_value(big, mid, small);
        } else {
            dw[i] = (sf[b] + sf[b + Cin]) + (sf[b + 4 * (size_t)Cin] + sf[b + 5 * (size_t)Cin]);
        }
    }
}
}  // namespace m355

namespace m355 {
// the same fold over per-workgroup PARTIAL ROWS (round 6: no atomics): part[rows][Cout*16*Cin + 4*Cout]; every output is the sum over the rows,
// in row order, of its four effective-gradient entries (bias: of the four classes' cells) -- the same bits on every run
__global__ __launch_bounds__(256) void k_up16_rows_to_9(const float *__restrict__ part, int rows, size_t stride, float *__restrict__ dw,
                                                        float *__restrict__ db, int Cout, int Cin)
{
    const size_t n9 = (size_t)Cout * 9 * Cin, n16 = (size_t)Cout * 16 * Cin, total = n9 + (db ? (size_t)Cout : 0);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float s2[2] = {0.0f, 0.0f};
        if (i >= n9) {
            const size_t b = n16 + (i - n9);
            for (int r = 0; r < rows; ++r) {
                const float *p = part + (size_t)r * stride + b;
                s2[r & 1] += (p[0] + p[Cout]) + (p[2 * (size_t)Cout] + p[3 * (size_t)Cout]);
            }
            db[i - n9] = s2[0] + s2[1];
            continue;
        }
        const int ci = (int)(i % Cin);
        const size_t t = i / Cin;
        const int tap = (int)(t % 9), co = (int)(t / 9), kh = tap / 3, kw = tap - 3 * kh;
        const size_t b = (((size_t)co * 4 + kh) * 4 + kw) * Cin + ci;   // entry (kh, kw); (kh+1, .) is 4 Cin further, (., kw+1) Cin
#pragma unroll 4
        for (int r = 0; r < rows; ++r) {
            const float *p = part + (size_t)r * stride + b;
            s2[r & 1] += (p[0] + p[Cin]) + (p[4 * (size_t)Cin] + p[5 * (size_t)Cin]);
        }
        dw[i] = s2[0] + s2[1];
    }
}
}  // namespace m355

// ws of the sub-pixel weight gradient: fp32 (m355_conv2d_wgrad_ws) or fixed-point (m355_conv2d_wgrad_det) cells
static size_t subpixel_ws_cells(const m355_conv_desc *d) { return (size_t)d->Cout * 16 * d->Cin + (size_t)d->Cout; }
// geometry of the sub-pixel class launch, and the partial rows of its no-atomics form (0: none -> zeroed cells + fp32 atomics)
static m355::WgradArgs subpixel_geometry(const m355_conv_desc *d)
{
    m355::WgradArgs a = {};
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Hl = d->H; a.Wl = d->W; a.ups = 0;
    a.Ho = d->H; a.Wo = d->W;   // the class grid = the stored extent; dy is [N, 2 Ho, 2 Wo, Cy]
    a.Cout = d->Cout; a.Cy = m355::dy_channels(d->Cout);
    a.KH = a.KW = 4; a.stride = 1; a.pad_h = a.pad_w = 1; a.pad_w_mode = d->pad_w_mode;
    return a;
}
static int subpixel_rows(const m355_conv_desc *d) { return m355::wgrad_halo_up_part_rows(subpixel_geometry(d)); }
static size_t subpixel_row_cells(const m355_conv_desc *d) { return (size_t)d->Cout * 16 * d->Cin + 4 * (size_t)d->Cout; }

static int subpixel_wgrad(const m355_conv_desc *d, const void *x, const void *dy, float *dw, float *dbias, void *ws, bool det,
                          hipStream_t st)
{
    const size_t cells = subpixel_ws_cells(d), n16 = (size_t)d->Cout * 16 * d->Cin;
    const int rows = det ? 0 : subpixel_rows(d);   // (m355_conv2d_wgrad_ws: partial rows + an ordered sum where the launch has >= 2 replicas)
    const size_t bytes = det ? sizeof(long long) * (1 + m355::kFixCell * cells) : sizeof(float) * cells;
    if (!rows && hipMemsetAsync(ws, 0, bytes, st) != hipSuccess) {
        m355::set_error("conv2d_wgrad (sub-pixel): memset failed");
        return M355_ERR_LAUNCH;
    }
    m355::WgradArgs a = subpixel_geometry(d);
    a.x = (const unsigned short *)x;
    a.dy = (const unsigned short *)dy;
    a.dw = det ? nullptr : (float *)ws;
    a.db = dbias ? (det ? (float *)ws /* non-null marker */ : (float *)ws + n16) : nullptr;
    a.fix = det ? (long long *)ws : nullptr;
    a.part = rows ? (float *)ws : nullptr;
    const size_t xbytes = (size_t)d->N * d->H * d->W * d->Cin * 2, ybytes = (size_t)d->N * d->H * d->W * 4 * a.Cy * 2;
    if (int rc = m355::wgrad_halo_up_launch(a, (unsigned)xbytes, (unsigned)ybytes, st)) return rc;
    const size_t blocks = ((size_t)d->Cout * 9 * d->Cin + d->Cout + 255) / 256;
    const dim3 grid((unsigned)(blocks > 2048 ? 2048 : blocks));
    if (rows) {
        hipLaunchKernelGGL(m355::k_up16_rows_to_9, grid, dim3(256), 0, st, (const float *)ws, rows, subpixel_row_cells(d), dw, dbias, d->Cout,
                           d->Cin);
        return m355::check_launch("conv2d_wgrad (sub-pixel fold of the partial rows)");
    }
    if (det) hipLaunchKernelGGL(m355::k_up16_to_9<true>, grid, dim3(256), 0, st, (const void *)ws, dw, dbias, d->Cout, d->Cin);
    else hipLaunchKernelGGL(m355::k_up16_to_9<false>, grid, dim3(256), 0, st, (const void *)ws, dw, dbias, d->Cout, d->Cin);
    return m355::check_launch("conv2d_wgrad (sub-pixel fold)");
}

/* Deterministic weight gradient: same kernels, but the split-K partial tiles are accumulated as 64-bit fixed-point integers in
 * `ws` (m355_conv2d_wgrad_det_ws_bytes(d) bytes, zeroed here) and converted to fp32 once -- bit-identical from run to run
 * whatever order the workgroups finish in.  dw / dbias are OVERWRITTEN (no pre-zeroing needed). */
extern "C" size_t m355_conv2d_wgrad_det_ws_bytes(const m355_conv_desc *d)
{
    if (!d || d->Cout <= 0 || d->Cin <= 0 || d->kh <= 0 || d->kw <= 0) return 0;
    // the flag word + kFixCell 64-bit integers per cell (conv_dma.h wg_accum)
    if (subpixel_halo(d)) return sizeof(long long) * (1 + m355::kFixCell * subpixel_ws_cells(d));   // the 16-entry effective gradient
    return sizeof(long long) * (1 + m355::kFixCell * ((size_t)d->Cout * d->kh * d->kw * d->Cin + (size_t)d->Cout));
}

extern "C" int m355_conv2d_wgrad_det(const m355_conv_desc *d, const void *x, const void *dy, void *ws, float *dw, float *dbias,
                                     void *stream)
{
    M355_REQUIRE(ws, "conv2d_wgrad_det: null workspace");
    if (int rc = check_desc(d, "conv2d_wgrad_det")) return rc;
    hipStream_t st = (hipStream_t)stream;
#ifdef M355_EXACT
    return conv_wgrad_impl(d, x, dy, dw, dbias, stream, true);   // (already a sequential fp64 sum per element)
#endif
    if (subpixel_halo(d)) {
        M355_REQUIRE(x && dy && dw, "conv2d_wgrad_det: null pointer");
        return subpixel_wgrad(d, x, dy, dw, dbias, ws, true, st);
    }
    if (hipMemsetAsync(ws, 0, m355_conv2d_wgrad_det_ws_bytes(d), st) != hipSuccess) {
        m355::set_error("conv2d_wgrad_det: memset failed");
        return M355_ERR_LAUNCH;
    }
    if (int rc = conv_wgrad_impl(d, x, dy, dw, dbias, stream, false, (long long *)ws)) return rc;
    const size_t n = (size_t)d->Cout * d->kh * d->kw * d->Cin;
    const size_t blocks = (n + d->Cout + 255) / 256;
    hipLaunchKernelGGL(m355::k_fix_to_f32, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st, (const long long *)ws, dw,
                       n, dbias, d->Cout);
    return m355::check_launch("conv2d_wgrad_det (convert)");
}

extern "C" int m355_conv2d_wgrad(const m355_conv_desc *d, const void *x, const void *dy, float *dw, float *dbias,
                                 void *stream)
{
    return conv_wgrad_impl(d, x, dy, dw, dbias, stream, true);
}

/* dw (and dbias) += the weight gradient: every wgrad kernel accumulates its split-K partial tiles with fp32 atomics, so the
 * only difference to m355_conv2d_wgrad is that the caller has zeroed (or wants to accumulate into) the buffers -- e.g. ONE
 * memset for the gradients of all layers of a backward pass instead of one or two per layer */
extern "C" int m355_conv2d_wgrad_acc(const m355_conv_desc *d, const void *x, const void *dy, float *dw, float *dbias,
                                     void *stream)
{
    return conv_wgrad_impl(d, x, dy, dw, dbias, stream, false);
}

/* The weight gradient of the 8-input-channel layer (D.conv1) ended in one fp32 atomic per element and WORKGROUP on 12.8 k
 * addresses.  With a workspace every workgroup instead stores its partial tile and a second small launch adds the rows in
 * workgroup order: no atomics, deterministic in every mode, dw / dbias OVERWRITTEN (no pre-zeroing); 340 -> 335 us at batch 128.
 * m355_conv2d_wgrad_ws_bytes(d) == 0: this layer has no such form (use m355_conv2d_wgrad / _acc / _det). */
// geometry of a weight-gradient launch (what conv_wgrad_impl fills in besides the pointers)
static m355::WgradArgs wgrad_geometry(const m355_conv_desc *d)
{
    m355::WgradArgs a = {};
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ups = d->upsample;
    a.Hl = d->H << d->upsample; a.Wl = d->W << d->upsample;
    conv_out_hw(d, &a.Ho, &a.Wo);
    a.Cout = d->Cout; a.Cy = m355::dy_channels(d->Cout);
    a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
    a.pad_w_mode = d->pad_w_mode;
    return a;
}
// partial rows of the no-atomics form of a halo weight gradient (round 6; 0: this layer has none)
static int wgrad_halo_rows(const m355_conv_desc *d)
{
    if (!wgrad_dma_ok(d) || m355::wgrad_c8_eligible(d, m355::dy_channels(d->Cout)) || m355::wgrad_small_eligible(d, m355::dy_channels(d->Cout)))
        return 0;
    return m355::wgrad_halo_part_rows(wgrad_geometry(d));
}

extern "C" size_t m355_conv2d_wgrad_ws_bytes(const m355_conv_desc *d)
{
    if (!d || check_desc(d, "conv2d_wgrad_ws_bytes")) return 0;
#ifdef M355_EXACT
    return 0;
#endif
    const int cy = m355::dy_channels(d->Cout);
    if (m355::wgrad_c8_eligible(d, cy)) return sizeof(float) * m355::wgrad_c8_ws_floats(d, cy);
    // upsample + 3x3 in the sub-pixel form: the 16-entry effective gradient (zeroed here, accumulated with fp32 atomics -- NOT the
    // ordered sum of the thin layers: m355_conv2d_wgrad_det is the run-to-run reproducible form of these layers)
    if (subpixel_halo(d)) {
        const int rows = subpixel_rows(d);   // (round 6: per-workgroup partial rows, folded in row order; else zeroed cells + atomics)
        return rows ? sizeof(float) * (size_t)rows * subpixel_row_cells(d) : sizeof(float) * subpixel_ws_cells(d);
    }
    // (round 6) the stride-2 class weight gradients on k_wgrad_halo (D.conv2-4): 64 / 16 / 4 split-K replicas per (co, ci, class) block
    // used to meet in same-address fp32 atomics -- a per-launch constant of ~40 us (and ~250 us with the deterministic mode's integer
    // cells); their partial tiles go to rows of this workspace and one small launch adds the rows in order
    if (const int rows = wgrad_halo_rows(d)) return sizeof(float) * (size_t)rows * ((size_t)d->Cout * d->kh * d->kw * d->Cin + d->Cout);
    // (the <= 8-output-channel heads keep their atomics: with up to 512 partial rows of a few thousand elements the ordered sum
    // is the longer tail -- conv_final's wgrad 117.7 -> 155.9 us, D.conv5's 60.0 -> 65.6 us, profiles/r04_thin_rate_b.txt)
    return 0;
}

extern "C" int m355_conv2d_wgrad_ws(const m355_conv_desc *d, const void *x, const void *dy, float *dw, float *dbias, void *ws,
                                    void *stream)
{
    M355_REQUIRE(ws && m355_conv2d_wgrad_ws_bytes(d) > 0, "conv2d_wgrad_ws: this layer has no partial-sum form (m355_conv2d_wgrad_ws_bytes)");
    if (subpixel_halo(d)) {
        M355_REQUIRE(x && dy && dw, "conv2d_wgrad_ws: null pointer");
        return subpixel_wgrad(d, x, dy, dw, dbias, ws, false, (hipStream_t)stream);
    }
    return conv_wgrad_impl(d, x, dy, dw, dbias, stream, false, nullptr, (float *)ws);
}

static int conv_wgrad_impl(const m355_conv_desc *d, const void *x, const void *dy, float *dw, float *dbias, void *stream,
                           bool zero, long long *fix, float *part)
{
    if (int rc = check_desc(d, "conv2d_wgrad")) return rc;
    M355_REQUIRE(x && dy && dw, "conv2d_wgrad: null pointer");
#ifdef M355_EXACT
    // (zero = overwrite; the accumulating entry point adds to what the caller zeroed; the deterministic one overwrites -- every
    // exact kernel is a sequential fp64 sum, so it is its own deterministic form)
    return m355::exact::conv_wgrad(d, x, dy, m355::dy_channels(d->Cout), dw, dbias, (zero || fix) ? 0 : 1, (hipStream_t)stream);
#endif
    M355_REQUIRE(!dbias || wgrad_has_dbias(d), "conv2d_wgrad: dbias is only fused on the DMA paths (m355_conv2d_wgrad_fuses_dbias)");
    hipStream_t st = (hipStream_t)stream;
    m355::WgradArgs a = {};
    a.x = (const unsigned short *)x;
    a.dy = (const unsigned short *)dy;
    a.dw = dw;
    a.db = dbias;
    a.fix = fix;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ups = d->upsample;
    a.Hl = d->H << d->upsample; a.Wl = d->W << d->upsample;
    conv_out_hw(d, &a.Ho, &a.Wo);
    a.Cout = d->Cout; a.Cy = m355::dy_channels(d->Cout);
    a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
    a.pad_w_mode = d->pad_w_mode;
    const int K = d->kh * d->kw * d->Cin, P = d->N * a.Ho * a.Wo;
    if (zero && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->Cout * K, st) != hipSuccess) {
        m355::set_error("conv2d_wgrad: memset failed");
        return M355_ERR_LAUNCH;
    }
    if (zero && dbias && hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)d->Cout, st) != hipSuccess) {
        m355::set_error("conv2d_wgrad: memset failed");
        return M355_ERR_LAUNCH;
    }
    if (m355::wgrad_c8_eligible(d, a.Cy)) return m355::wgrad_c8_launch(d, x, dy, a.Cy, dw, dbias, st, fix, part);
    if (m355::wgrad_small_eligible(d, a.Cy)) return m355::wgrad_small_launch(d, x, dy, a.Cy, dw, st, fix, part);
    const size_t xbytes = (size_t)d->N * d->H * d->W * d->Cin * 2, ybytes = (size_t)P * a.Cy * 2;
    const int lgWo = m355::ilog2_exact(a.Wo), lgHo = m355::ilog2_exact(a.Ho);
    if (wgrad_dma_ok(d) && m355::wgrad_halo_eligible(a)) {
        if (part && wgrad_halo_rows(d)) a.part = part;   // (m355_conv2d_wgrad_ws: partial rows + ordered sum, dw / dbias overwritten)
        return m355::wgrad_halo_launch(a, (unsigned)xbytes, (unsigned)ybytes, st);
    }
    if (wgrad_dma_ok(d)) {
        const int TM = d->Cout > 64 ? 128 : 64, TN = d->Cout > 64 ? 128 : 256;
        const int gx = (d->Cout + TM - 1) / TM, gy = (K + TN - 1) / TN;
        // split the pixel axis: ~256 workgroups (one per CU), at least 4 K steps each.  Every workgroup ends with TM x TN fp32
        // atomics, so on the small layers this kernel is left with (mesh discriminator, the generator's 8x4 .. 32x16 stages,
        // 1x1 shortcuts) the split count is the cost: all its launches of a GAN cycle together 1.17 / 0.87 / 0.77 ms at
        // 1024 / 512 / 256 workgroups
        static const int wg_target = getenv("M355_WGRAD_SPLIT_WGS") ? atoi(getenv("M355_WGRAD_SPLIT_WGS")) : 256;
        static const int min_steps = getenv("M355_WGRAD_SPLIT_STEPS") ? atoi(getenv("M355_WGRAD_SPLIT_STEPS")) : 4;
        int splits = (wg_target + gx * gy - 1) / (gx * gy);
        const int max_splits = (P + min_steps * 64 - 1) / (min_steps * 64);
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        a.chunk = ((P + splits - 1) / splits + 63) / 64 * 64;
        splits = (P + a.chunk - 1) / a.chunk;
        m355::WgradDmaArgs A = {a, lgWo, lgHo, (unsigned)xbytes, (unsigned)ybytes};
        const dim3 grid(gx, gy, splits);
#define M355_WG(TM_, TN_, DET_)                                                                                              \
    do {                                                                                                                     \
        if (d->pad_w_mode == 0) hipLaunchKernelGGL((m355::k_wgrad_dma<TM_, TN_, 0, DET_>), grid, dim3(256), 0, st, A);       \
        else if (d->pad_w_mode == 1) hipLaunchKernelGGL((m355::k_wgrad_dma<TM_, TN_, 1, DET_>), grid, dim3(256), 0, st, A);  \
        else hipLaunchKernelGGL((m355::k_wgrad_dma<TM_, TN_, 2, DET_>), grid, dim3(256), 0, st, A);                          \
    } while (0)
        if (fix) {
            if (TM == 128) M355_WG(128, 128, true);
            else M355_WG(64, 256, true);
        } else if (TM == 128) M355_WG(128, 128, false);
        else M355_WG(64, 256, false);
#undef M355_WG
        m355::note_kernel("k_wgrad_dma");
        return m355::check_launch("conv2d_wgrad (dma)");
    }
    const int TM = d->Cout > 64 ? 128 : 64;
    const int gx = (d->Cout + TM - 1) / TM, gy = (K + m355::WN - 1) / m355::WN;
    // split the pixel axis so that ~1024 workgroups are in flight, at least 2 K-steps each
    int splits = (1024 + gx * gy - 1) / (gx * gy);
    const int max_splits = (P + 2 * m355::WK - 1) / (2 * m355::WK);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    a.chunk = ((P + splits - 1) / splits + m355::WK - 1) / m355::WK * m355::WK;
    splits = (P + a.chunk - 1) / a.chunk;
    if (fix) {
        if (TM == 128) hipLaunchKernelGGL((m355::k_wgrad_mfma<128, true>), dim3(gx, gy, splits), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((m355::k_wgrad_mfma<64, true>), dim3(gx, gy, splits), dim3(256), 0, st, a);
    } else if (TM == 128) hipLaunchKernelGGL(m355::k_wgrad_mfma<128>, dim3(gx, gy, splits), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(m355::k_wgrad_mfma<64>, dim3(gx, gy, splits), dim3(256), 0, st, a);
    m355::note_kernel("k_wgrad_mfma");
    return m355::check_launch("conv2d_wgrad");
}

/* the plan of a layer: every pre-launch query in one struct (include/m355.h m355_conv_plan) */
extern "C" int m355_conv2d_plan(const m355_conv_desc *d, m355_conv_plan *p)
{
    M355_REQUIRE(p, "conv2d_plan: null pointer");
    if (int rc = check_desc(d, "conv2d_plan")) return rc;
    memset(p, 0, sizeof(*p));
    conv_out_hw(d, &p->Ho, &p->Wo);
    p->dy_channels = m355::dy_channels(d->Cout);
    p->act_bytes = m355_act_bytes();
    p->w_fwd_elems = m355_conv2d_weight_elems(d, 0);
    p->w_dgrad_elems = m355_conv2d_weight_elems(d, 1);
    p->fwd_bits_ok = m355_conv2d_maskbits_ok(d, 0);
    p->dgrad_bits_ok = m355_conv2d_maskbits_ok(d, 1);
    p->dgrad_mask_ok = m355_conv2d_dgrad_mask_ok(d);
    p->fwd_stats_rows = m355_conv2d_fwd_stats_rows(d);
    p->fwd_ws_bytes = m355_conv2d_fwd_ws_bytes(d);
    p->fwd_ws_stats_rows = m355_conv2d_fwd_ws_stats_rows(d);
    p->wgrad_fuses_dbias = m355_conv2d_wgrad_fuses_dbias(d);
    p->dgrad_ws_bytes = m355_conv2d_dgrad_ws_bytes(d);
    p->wgrad_ws_bytes = m355_conv2d_wgrad_ws_bytes(d);
    p->wgrad_det_ws_bytes = m355_conv2d_wgrad_det_ws_bytes(d);
    p->exec_ratio = m355_conv2d_exec_ratio(d);
    p->w_dgrad_row_elems = d->stride == 1 ? m355::k_padded(d->kh * d->kw * (int)m355::dy_channels(d->Cout)) : 0;
    // is the workspace form an ORDERED sum (run-to-run identical in every mode)?  Everything but a sub-pixel layer whose launch has a
    // single replica per block (zeroed cells + fp32 atomics: the deterministic form of that one is m355_conv2d_wgrad_det)
    p->wgrad_ws_ordered = p->wgrad_ws_bytes > 0 && (!subpixel_halo(d) || subpixel_rows(d) > 0);
    return M355_OK;
}
