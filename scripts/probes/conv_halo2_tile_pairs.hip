// k_conv_tb: the 8-wave 2x2 class convolutions of k_conv_halo (csrc/conv_halo.hip) on PAIRS of pixel tiles that share
// every weight tile -- twice the MFMA work per byte of weights DMA'd into LDS.
//
// Why (profiles/r03_stamp_halo_base.txt, in-kernel s_memtime stamps of k_conv_halo<128,8,2,...>): a (tap, 64-channel chunk)
// step of 16 MFMAs per wave takes 2100-2300 shader cycles against 1024 of MFMA issue for the two waves of a SIMD, and the time
// follows the number of LDS-DMA instructions the step carries: steps with 2 DMAs per wave run their 12 MFMAs + 16 fragment
// reads in ~620 cycles (= the pipe), steps with 4 / 5 DMAs per wave in ~1250 / ~1650-2150 -- every KiB DMA'd per wave costs
// ~400 cycles of the phase that also reads fragments.  A 256-pixel x 128-channel tile moves 102 KB into LDS per (class, chunk)
// segment, 64 KB of it weights -- and the weights are the same for every pixel tile.  Here a workgroup walks its tiles two at a
// time ("top" / "bottom": any two tiles of its list): per chunk the four taps' weights (4 x 16 KB ring, one slot per tap) are
// loaded ONCE and used by both tiles' segments, 140 KB per 2 x the MFMAs = 18 DMA instructions per wave per 8 steps instead
// of 26.  Costs: two accumulator sets (128 registers), so the fragments of a step are read in the step itself (no register
// double buffer; the 3x3 8-wave variant of k_conv_halo runs the same way), 144 KB of LDS.
//
// Step schedule of one (class, chunk) sequence, s = 4*sub + tap (sub 0 = top tile, halo buffer 0; sub 1 = bottom, buffer 1):
//   s0 .. s3: top taps 0..3      s4 .. s7: bottom taps 0..3        ring slot = tap
//   weights : slot 3 of THIS sequence at s0 (free since the previous s7), slots 0 / 1 / 2 of the NEXT sequence at s5 / s6 / s7
//             (each right behind the barrier that follows its last read at s4 / s5 / s6)
//   halo    : the bottom tile's halo of this chunk in slices at s0 (3 DMAs per wave) and s1 (2) into buffer 1; the top tile's halo
//             of the next sequence at s4 (3) and s5 (2) into buffer 0
//   DMAs per wave and step: 5 2 0 0 3 4 2 2 (weights before the slice inside a step) -> counted waits before the barriers:
//     s0: halo 0 complete + slot 0 (last piece: the slice of the previous s5)  -> younger: s6, s7           -> vmcnt(4)
//     s1: slot 1 (previous s6) -> younger: s7, s0 -> vmcnt(7)     s2: slot 2 (previous s7) -> s0, s1 -> vmcnt(7)
//     s3: slot 3 (head of s0)  -> younger: the slice of s0, s1, s2 -> vmcnt(5)      s4: halo 1 (slice of s1) -> vmcnt(0)
//     s5 .. s7: nothing new (only the LDS read / refill ordering of the barrier itself)
//   behind a tile pair's epilogue the stores are younger than what s0 .. s2 await: their count is added (vmcnt retires in order).
#include <stdlib.h>

#include <type_traits>

#include <cstring>
#include "conv_dma.h"

namespace m355 {
namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

template <int N>
__device__ __forceinline__ void wait_vm_lgkm()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

}  // namespace

template <int MODE, int SUB, int PAIR>
__global__ __launch_bounds__(512, 2) void k_conv_tb(ConvArgs a, unsigned xbytes, unsigned wbytes)
{
    static_assert(!PAIR || SUB == 1, "class pairs: dgrad classes only");
    constexpr int BN = 128, NW = 8, KS = 2, T = 4, TH = 8, TW = 32;
    constexpr int NC = SUB == 2 ? 4 : 1;                  // classes accumulated into one output tile (stride-2 forward)
    constexpr int HH = TH + KS - 1, HWD = TW + KS - 1 + PAIR, HR = HH * HWD;
    constexpr int NA = (HR + 7) / 8, NAW = (NA + NW - 1) / NW, NAS = 3;
    static_assert(NAW <= 2 * NAS, "two slices per halo");
    constexpr int NBW = BN / (8 * NW);
    constexpr int ABUF = NW * NAW * 1024, BBUF = BN * 128, RB = T;
    constexpr int WGN = 2, PI = 2, CJ = 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * ABUF + RB * BBUF];
    unsigned char *const ldsB = lds + 2 * ABUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nN = PAIR ? 1 : a.CoutP / BN;
    const int tn = blockIdx.x % nN, bp = blockIdx.x / nN, PS = gridDim.x / nN;
    const int tpx = a.Wo / TW, tpy = a.Ho / TH, tiles_p = a.N * tpx * tpy;
    const int n0 = tn * BN;
    if (bp >= tiles_p) return;

    int pad_h = a.pad_h, pad_w = a.pad_w, oy_off = a.oy_off, ox_off = a.ox_off;
    int xsh = 0;
    const unsigned short *wv = a.w;
    if (PAIR) {
        const int c0 = 2 * blockIdx.y, cls = c0 + (wave % WGN);
        const int pw = max(a.cpad_w[c0], a.cpad_w[c0 + 1]);
        pad_h = a.cpad_h[c0]; oy_off = a.coy[c0];
        pad_w = pw; xsh = pw - a.cpad_w[cls]; ox_off = a.cox[cls];
        wv += (size_t)c0 * a.cls_w_elems;
    } else if (a.ncls > 1) {
        const int cls = blockIdx.y;
        pad_h = a.cpad_h[cls]; pad_w = a.cpad_w[cls]; oy_off = a.coy[cls]; ox_off = a.cox[cls];
        wv += (size_t)cls * a.cls_w_elems;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wv, 0, wbytes, 0x00020000);

    // halo DMA roles (as k_conv_halo): slot k of wave w = DMA instruction q = NW*k + w = halo rows 8q .. 8q+7; lane l -> row
    // 8q + (l>>3), LDS chunk slot l&7, source chunk (l&7) ^ ((row>>1)&7)
    const int csrc = (lane & 7) ^ (((wave & 1) << 2) | ((lane >> 4) & 3));
    unsigned aoff[NAW], hyx[NAW];
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
        const int rho = 8 * (NW * k + wave) + (lane >> 3);
        const int hy = rho / HWD, hx = rho - hy * HWD;
        hyx[k] = rho < HR ? (unsigned)((SUB * hy) << 16 | (SUB * hx)) : 0x7fff0000u;
    }
    const unsigned c16 = (unsigned)csrc * 16u, cin2 = (unsigned)a.Cin * 2u;
    auto tile_origin = [&](int tp, int &n, int &oy0, int &ox0) {
        n = tp / (tpx * tpy);
        const int trem = tp - n * (tpx * tpy);
        oy0 = (trem / tpx) * TH;
        ox0 = (trem % tpx) * TW;
    };
    struct Tgt {
        int Yb, Xb;
        unsigned nbase;
    };
    auto target = [&](int tp, int cls) {
        int n, oy0, ox0;
        tile_origin(tp, n, oy0, ox0);
        Tgt t;
        t.Yb = SUB == 2 ? 2 * oy0 + (cls >> 1) - pad_h : oy0 - pad_h;
        t.Xb = SUB == 2 ? 2 * ox0 + (cls & 1) - pad_w : ox0 - pad_w;
        t.nbase = (unsigned)(n * a.H * a.W) * cin2;
        return t;
    };
    auto compute_aoff = [&](const Tgt &t, auto k0c, auto k1c) {
#pragma unroll
        for (int k = decltype(k0c)::value; k < decltype(k1c)::value; ++k) {
            const int iy = t.Yb + (int)(hyx[k] >> 16);
            int ix = t.Xb + (int)(hyx[k] & 0xffffu);
            if (MODE == 1) ix = min(max(ix, 0), a.W - 1);
            else if (MODE == 2) ix = ix < 0 ? ix + a.W : (ix >= a.W ? ix - a.W : ix);
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned off = __umul24(__umul24(iy, a.W) + ix, cin2) + (t.nbase + c16);
            aoff[k] = ok ? off : OOB;
        }
    };
    using K0 = std::integral_constant<int, 0>;
    using KH = std::integral_constant<int, NAW / 2>;
    using KN = std::integral_constant<int, NAW>;
    unsigned wrow[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const unsigned row = (unsigned)(n0 + 8 * (NW * j + wave) + (lane >> 3));
        wrow[j] = PAIR ? (row & 63u) * (unsigned)(a.Kp * 2) + (row >> 6) * (a.cls_w_elems * 2u) + csrc * 16
                       : row * (unsigned)(a.Kp * 2) + csrc * 16;
    }
    const int ncc = a.Cin >> 6, NSQ = NC * ncc;

    auto issue_B = [&](int cls, int cc, auto tapc) {   // weights of (class, chunk, tap) -> ring slot `tap`
        constexpr int tap = decltype(tapc)::value;
        const int ktap = SUB == 2 ? (2 * (tap >> 1) + (cls >> 1)) * 4 + 2 * (tap & 1) + (cls & 1) : tap;
        const unsigned so = (unsigned)(ktap * a.Cin + cc * 64) * 2u;
#pragma unroll
        for (int j = 0; j < NBW; ++j) dma16(rw, ldsB + tap * BBUF + (NW * j + wave) * 1024, wrow[j], so);
    };
    auto issue_A = [&](int hbuf, int chunk, auto tapc) {   // halo slice `tap` (taps 0, 1) of the target aoff[] describes
        constexpr int tap = decltype(tapc)::value;
#pragma unroll
        for (int k = 0; k < NAW; ++k)
            if (k >= tap * NAS && k < (tap + 1) * NAS && tap <= 1)
                dma16(rx, lds + hbuf * ABUF + (NW * k + wave) * 1024, aoff[k], (unsigned)chunk * 128u);
    };

    const int wm = wave / WGN, wn = wave % WGN;
    const int tx = lane & 31, half = lane >> 5;
    f32x16 acc[2][CJ][PI];
    auto init_acc = [&]() {
        if (a.bias) {
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = n0 + wn * 64 + 32 * j + 8 * g + 4 * half;
                    const float4 b = co < a.Cout ? *reinterpret_cast<const float4 *>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int i = 0; i < PI; ++i) {
                            acc[u][j][i][4 * g] = b.x; acc[u][j][i][4 * g + 1] = b.y; acc[u][j][i][4 * g + 2] = b.z; acc[u][j][i][4 * g + 3] = b.w;
                        }
                }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < PI; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[u][j][i][r] = 0.0f;
        }
    };
    init_acc();
    const unsigned char *fb = ldsB + (wn * 64 + (lane & 31)) * 128;
    const int swzb = (lane >> 1) & 7;
    unsigned short *yb = reinterpret_cast<unsigned short *>(a.y);

    struct Frags {
        bf16x8 p[4][PI], w[4][CJ];
    };
    const int rho_lane = 2 * wm * HWD + tx + (PAIR ? xsh : 0);   // halo row of the lane's pixel (tile row 2wm, tap (0, 0))
    // k-groups [K0_, K1_) of a step's fragments (the step reads them in two halves: 32 registers each instead of 64 live at once)
    auto read_frags = [&](Frags &f, const unsigned char *ha, auto tapc, auto k0c, auto k1c) {
        constexpr int tap = decltype(tapc)::value, K0_ = decltype(k0c)::value, K1_ = decltype(k1c)::value;
        constexpr int kh = tap / KS, kw = tap - kh * KS;
        const unsigned char *bs = fb + tap * BBUF;
        // the 32 pixel-fragment addresses of a sequence are loop invariant: hoisted, they cost 32 registers this kernel does not
        // have (the fragments would then be read three at a time with a drained lgkmcnt between them) -- an opaque copy of the
        // lane's halo row makes them a dozen VALU ops per step in the MFMA shadow instead
        int rho_l = rho_lane;
        asm volatile("" : "+v"(rho_l));
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int rho = rho_l + (i + kh) * HWD + kw;
            const int rowa = rho * 128, swa = (rho >> 1) & 7;
#pragma unroll
            for (int kk = K0_; kk < K1_; ++kk)
                f.p[kk][i] = *reinterpret_cast<const bf16x8 *>(ha + rowa + (((kk * 2 + half) ^ swa) << 4));
        }
#pragma unroll
        for (int kk = K0_; kk < K1_; ++kk)
#pragma unroll
            for (int j = 0; j < CJ; ++j)
                f.w[kk][j] = *reinterpret_cast<const bf16x8 *>(bs + j * 32 * 128 + (((kk * 2 + half) ^ swzb) << 4));
    };
    auto advance = [&](int &cls, int &cc) {
        if (++cc == ncc) {
            cc = 0;
            if (NC > 1) cls = cls + 1 == NC ? 0 : cls + 1;
        }
    };

    // ---- prologue: the first top halo, ring slots 0 .. 2 of the first (class, chunk), the offsets of the first bottom halo
    int tA = bp;
    {
        const Tgt t0 = target(tA, 0);
        compute_aoff(t0, K0{}, KN{});
#pragma unroll
        for (int k = 0; k < NAW; ++k) dma16(rx, lds + (NW * k + wave) * 1024, aoff[k], 0u);
        issue_B(0, 0, std::integral_constant<int, 0>{});
        issue_B(0, 0, std::integral_constant<int, 1>{});
        issue_B(0, 0, std::integral_constant<int, 2>{});
        const Tgt t1 = target(tA + PS < tiles_p ? tA + PS : tA, 0);
        compute_aoff(t1, K0{}, KN{});
    }

    // stores one lane issues in a tile pair's epilogue: younger than the weights / halo the first three steps of the next pair
    // await (see the header).  Only counted when both tiles are stored and the epilogue is one of the two plain forms.
    const int epi_st = (a.bits_out && a.slope != 1.0f) ? 20 : 16;
    int fresh = 0;
    // Software pipeline by HALF a step: the MFMAs of k-groups 2, 3 of step s run at the head of step s+1, behind that step's
    // first eight fragment reads (whose latency they hide); k-groups 2, 3 of the fragments are carried across the barrier.
    // Zero fragments = no-op MFMAs: before the first step and after a pair's epilogue has flushed the tail.
#ifdef M355_DBG_STAMP
    // debug build only (scripts/stamp_halo.py): shader-clock stamps of the step phases, per wave, of workgroup (0,0)
    constexpr int DBG_STEPS = 96;
    __shared__ unsigned dbg_lds[8 * DBG_STEPS * 4];
    unsigned dbg_i = 0, dbg_t0 = 0, dbg_t1 = 0, dbg_t2 = 0, dbg_t3 = 0, dbg_n0 = 0;
    const bool dbg_on = a.stats != nullptr && blockIdx.x == 0 && blockIdx.y == 0;
    const unsigned long long dbg_c0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    Frags f;
    auto zero_tail = [&]() {
        const bf16x8 z = {};
#pragma unroll
        for (int kk = 2; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < PI; ++i) f.p[kk][i] = z;
#pragma unroll
            for (int j = 0; j < CJ; ++j) f.w[kk][j] = z;
        }
    };
    zero_tail();
    for (;;) {
        const bool has_B = tA + PS < tiles_p;
        const int tB = has_B ? tA + PS : tA;
        const int tA_next = tA + 2 * PS;
        const bool has_next = tA_next < tiles_p;
        const int tAn = has_next ? tA_next : tA, tBn = has_next ? (tA_next + PS < tiles_p ? tA_next + PS : tA_next) : tB;
        int nA, oyA, oxA, nB, oyB, oxB;
        tile_origin(tA, nA, oyA, oxA);
        tile_origin(tB, nB, oyB, oxB);
        unsigned rbits_pf[2][PI] = {};
        if (a.bits_in) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < PI; ++i) {
                    const int n = u ? nB : nA, oy0 = u ? oyB : oyA, ox0 = u ? oxB : oxA;
                    const size_t pix = ((size_t)n * a.OH + ((oy0 + 2 * wm + i) * a.oy_mul + oy_off)) * a.OW + ((ox0 + tx) * a.ox_mul + ox_off);
                    rbits_pf[u][i] = a.bits_in[(pix * (size_t)(a.Cs >> 6) + (size_t)(PAIR ? 0 : (n0 >> 6) + wn)) * 2 + half];
                }
        }
        int cls_cur = 0, cc_cur = 0;
        for (int sq = 0; sq < NSQ; ++sq) {
            int cls_n = cls_cur, cc_n = cc_cur;
            advance(cls_n, cc_n);
            const bool last = sq + 1 == NSQ;
            // targets whose offsets are prepared during this sequence: (top taps 2, 3) the top halo of the next sequence, fetched by
            // this sequence's bottom steps; (bottom taps 2, 3) the bottom halo of the next sequence, fetched by its top steps
            const Tgt tg_top = target(last ? tAn : tA, cls_n), tg_bot = target(last ? tBn : tB, cls_n);
            sfor<0, 2>([&](auto subc) {
                constexpr int sub = decltype(subc)::value;
                sfor<0, T>([&](auto tapc) {
                    constexpr int tap = decltype(tapc)::value;
                    constexpr int s = 4 * sub + tap;
                    constexpr int cnt = s == 0 ? 4 : (s == 1 || s == 2) ? 7 : s == 3 ? 5 : s == 4 ? 0 : -1;
#ifdef M355_DBG_STAMP
                    if (dbg_on) dbg_n0 = (unsigned)__builtin_amdgcn_s_memtime();
#endif
                    if constexpr (cnt < 0) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    } else if constexpr (s < 3) {
                        if (fresh > 0) {
                            --fresh;
                            if (epi_st == 16) wait_vm_lgkm<cnt + 16>();
                            else wait_vm_lgkm<cnt + 20>();
                        } else {
                            wait_vm_lgkm<cnt>();
                        }
                    } else {
                        wait_vm_lgkm<cnt>();
                    }
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
#ifdef M355_DBG_STAMP
                    if (dbg_on) {
                        if (dbg_i > 0 && dbg_i <= DBG_STEPS) {
                            unsigned *q = dbg_lds + (wave * DBG_STEPS + (dbg_i - 1)) * 4;
                            q[0] = dbg_t0; q[1] = dbg_t1; q[2] = dbg_t2; q[3] = dbg_t3;
                        }
                        ++dbg_i;
                        dbg_t0 = dbg_n0;
                        dbg_t1 = (unsigned)__builtin_amdgcn_s_memtime();
                    }
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    using I0 = std::integral_constant<int, 0>;
                    using I2 = std::integral_constant<int, 2>;
                    using I4 = std::integral_constant<int, 4>;
                    constexpr int psub = s == 0 ? 1 : (s - 1) / 4;   // the tile of the previous step (s0 follows the previous s7)
                    read_frags(f, lds + sub * ABUF, tapc, I0{}, I2{});
                    auto mma = [&](auto uc, int kk) {
                        constexpr int u = decltype(uc)::value;
#pragma unroll
                        for (int j = 0; j < CJ; ++j)
#pragma unroll
                            for (int i = 0; i < PI; ++i)
                                acc[u][j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[kk][j], f.p[kk][i], acc[u][j][i], 0, 0, 0);
                    };
                    mma(std::integral_constant<int, psub>{}, 2);   // the previous step's tail, behind this step's first reads
                    mma(std::integral_constant<int, psub>{}, 3);
                    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // all eight reads in flight first
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // k-groups 2, 3 (their registers are free now) ride in the issue gaps of mma(0): they have the DMA issue and
                    // mma(1) to land before the step's closing lgkmcnt(0)
                    read_frags(f, lds + sub * ABUF, tapc, I2{}, I4{});
                    mma(subc, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (s == 0) issue_B(cls_cur, cc_cur, std::integral_constant<int, 3>{});
                    if constexpr (s >= 5) issue_B(cls_n, cc_n, std::integral_constant<int, s - 5>{});
                    if constexpr (tap <= 1) issue_A(sub ^ 1, sub == 0 ? cc_cur : cc_n, tapc);
                    __builtin_amdgcn_sched_barrier(0);
#ifdef M355_DBG_STAMP
                    if (dbg_on) dbg_t2 = (unsigned)__builtin_amdgcn_s_memtime();
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    if constexpr (tap == 2) compute_aoff(sub == 0 ? tg_top : tg_bot, K0{}, KH{});
                    if constexpr (tap == 3) compute_aoff(sub == 0 ? tg_top : tg_bot, KH{}, KN{});
                    mma(subc, 1);
                    __builtin_amdgcn_sched_barrier(0);
#ifdef M355_DBG_STAMP
                    if (dbg_on) dbg_t3 = (unsigned)__builtin_amdgcn_s_memtime();
                    __builtin_amdgcn_sched_barrier(0);
#endif
                });
            });
            cls_cur = cls_n;
            cc_cur = cc_n;
        }

        // the tail of the pair's last step (bottom tile, k-groups 2, 3)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 2; kk < 4; ++kk)
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int i = 0; i < PI; ++i)
                    acc[1][j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[kk][j], f.p[kk][i], acc[1][j][i], 0, 0, 0);
        zero_tail();
        // ---- epilogue of the pair (the next pair's top halo and ring slots 0 .. 2 are in flight):
        // acc[u][j][i][r] = channel n0 + 64wn + 32j + 8(r>>2) + 4half + (r&3), pixel (2wm+i, tx) of tile u
        auto store_tile = [&](auto uc, auto plainc, auto maskc) {
            constexpr int u = decltype(uc)::value;
            constexpr bool PLAIN = decltype(plainc)::value, MASK = decltype(maskc)::value;
            const int n = u ? nB : nA, oy0 = u ? oyB : oyA, ox0 = u ? oxB : oxA;
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                const int ho = oy0 + 2 * wm + i, wo = ox0 + tx;
                const size_t pix = ((size_t)n * a.OH + (ho * a.oy_mul + oy_off)) * a.OW + (wo * a.ox_mul + ox_off);
                const size_t bword = (pix * (size_t)(a.Cs >> 6) + (size_t)(PAIR ? 0 : (n0 >> 6) + wn)) * 2 + half;
                unsigned wbits = 0;
                const unsigned rbits = MASK ? rbits_pf[u][i] : 0u;
                const bool emit_bits = !PLAIN && a.bits_out != nullptr;
#pragma unroll
                for (int j = 0; j < CJ; ++j) {
                    const int cbase = (PAIR ? 0 : n0 + wn * 64) + 32 * j;
                    uint2 pk[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4] = {acc[u][j][i][4 * g], acc[u][j][i][4 * g + 1], acc[u][j][i][4 * g + 2], acc[u][j][i][4 * g + 3]};
                        if (!PLAIN) {
                            if (emit_bits) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) wbits |= (v[e] > 0.0f ? 1u : 0u) << (16 * j + 4 * g + e);
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.0f ? v[e] : v[e] * a.slope;
                        }
                        if (MASK) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = ((rbits >> (16 * j + 4 * g + e)) & 1u) ? v[e] : v[e] * a.mask_slope;
                        }
                        pk[g].x = pack_bf16(v[0], v[1]);
                        pk[g].y = pack_bf16(v[2], v[3]);
                    }
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        auto sx = __builtin_amdgcn_permlane32_swap(pk[g].x, pk[g + 1].x, false, false);
                        auto sy = __builtin_amdgcn_permlane32_swap(pk[g].y, pk[g + 1].y, false, false);
                        const int co = cbase + 8 * (g + half);
                        uint4 o;
                        o.x = sx[0]; o.y = sy[0]; o.z = sx[1]; o.w = sy[1];
                        *reinterpret_cast<uint4 *>(yb + pix * a.Cs + co) = o;
                    }
                }
                if (emit_bits) a.bits_out[bword] = wbits;
            }
        };
        auto store_pair = [&](auto plainc, auto maskc) {
            store_tile(std::integral_constant<int, 0>{}, plainc, maskc);
            if (has_B) store_tile(std::integral_constant<int, 1>{}, plainc, maskc);
        };
        {
            using std::false_type;
            using std::true_type;
            const bool plain = a.slope == 1.0f;
            if (plain && !a.bits_in) store_pair(true_type{}, false_type{});
            else if (plain) store_pair(true_type{}, true_type{});
            else if (!a.bits_in) store_pair(false_type{}, false_type{});
            else store_pair(false_type{}, true_type{});
        }
        if (!has_next) break;
        init_acc();
        tA = tA_next;
        fresh = has_B ? 3 : 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (unused) prefetches
#ifdef M355_DBG_STAMP
    if (dbg_on) {
        __syncthreads();
        unsigned *out = reinterpret_cast<unsigned *>(a.stats);
        for (int k = tid; k < 8 * DBG_STEPS * 4; k += NW * 64) out[k] = dbg_lds[k];
        if (tid == 0) {   // whole-kernel shader cycles and 100 MHz reference ticks of this workgroup -> the effective clock
            out[8 * DBG_STEPS * 4] = (unsigned)(__builtin_amdgcn_s_memtime() - dbg_c0);
            out[8 * DBG_STEPS * 4 + 1] = (unsigned)(__builtin_amdgcn_s_memrealtime() - dbg_r0);
        }
    }
#endif
}

// which k_conv_halo problems run on tile pairs: the 8-wave 2x2 class kernels with the unguarded epilogue
bool conv_tb_eligible(const ConvArgs &a)
{
    // OPT-IN (M355_TB=1).  Measured (profiles/r03_power_tb_vs_halo.txt, D.conv3 at batch 128 back to back for 4 s): the socket sits
    // at its 1400 W cap under either kernel; this one needs 31 % fewer DMA instructions and runs at a higher clock (1829 vs
    // 1707 MHz) but spends more cycles per step (1831 vs 1648: no full-step register double buffer next to 128 accumulator
    // registers) -- 1073 vs 1112 TF forward, 1044 vs 1092 TF dgrad.  Under the power cap the step's ENERGY decides, and the
    // weight DMAs it saves are a small part of it.  Kept for the A/B and as the starting point of a 128 x 64 wave tile.
    const char *on = getenv("M355_TB");
    if (!on || on[0] == '0' || getenv("M355_NO_TB")) return false;
#ifdef M355_DBG_STAMP
    const bool stats = false;   // (a.stats carries the stamp buffer)
#else
    const bool stats = a.stats != nullptr;
#endif
    if (a.y_f32_nchw || a.fold2 || a.mask_x || stats || a.ups || a.Cout != a.CoutP || a.Cin % 64 || a.Wo % 32 || a.Ho % 8 || a.Cs % 8)
        return false;
    // at least one full pair per workgroup of the launch (M355_HALO_WGS: tests force few workgroups)
    const int tiles = a.N * (a.Ho / 8) * (a.Wo / 32);
    const char *wgs = getenv("M355_HALO_WGS");
    const int resident = wgs ? atoi(wgs) : 256;
    auto enough = [&](int lists) { return tiles >= 2 * (resident / lists > 0 ? resident / lists : 1); };
    if (a.stride == 2) {   // forward of a 4x4 stride-2 conv (four accumulated classes), >= 128 output channels
        if (!(a.KH == 4 && a.KW == 4 && a.pad_h == 1 && a.pad_w == 1 && a.ncls <= 1 && a.H == 2 * a.Ho && a.W == 2 * a.Wo)) return false;
        return a.CoutP % 128 == 0 && enough(a.CoutP / 128);
    }
    if (a.stride != 1 || a.KH != 2 || a.KW != 2 || a.ncls != 4) return false;
    const bool pair = a.CoutP == 64 && a.Cout == 64 && !a.bias && a.cpad_h[0] == a.cpad_h[1] && a.cpad_h[2] == a.cpad_h[3] &&
                      a.coy[0] == a.coy[1] && a.coy[2] == a.coy[3] && !getenv("M355_NO_HALO_PAIR");
    if (pair) return enough(2);
    return a.CoutP % 128 == 0 && enough(4 * (a.CoutP / 128));
}

int conv_tb_launch(const ConvArgs &a, unsigned xb, unsigned wb, hipStream_t st)
{
    const int tiles = a.N * (a.Ho / 8) * (a.Wo / 32);
    const bool pair = a.stride == 1 && a.CoutP == 64;
    const int nN = pair ? 1 : a.CoutP / 128, ncls = a.stride == 2 ? 1 : (pair ? 2 : a.ncls);
    const char *wgs = getenv("M355_HALO_WGS");   // tests: few workgroups, several tile pairs each
    int per = (wgs ? atoi(wgs) : 256) / (nN * ncls);
    if (per < 1) per = 1;
    if (2 * per > tiles) per = (tiles + 1) / 2;
    // every workgroup the same number of PAIRS (the last pair of a list may be half empty)
    const int pairs = (tiles + 2 * per - 1) / (2 * per);
    per = (tiles + 2 * pairs - 1) / (2 * pairs);
    const dim3 grid((unsigned)per * nN, ncls);
#define M355_TB(SUB_, PAIR_, WB_)                                                                                              \
    do {                                                                                                                       \
        if (a.pad_w_mode == 0) hipLaunchKernelGGL((k_conv_tb<0, SUB_, PAIR_>), grid, dim3(512), 0, st, a, xb, WB_);             \
        else if (a.pad_w_mode == 1) hipLaunchKernelGGL((k_conv_tb<1, SUB_, PAIR_>), grid, dim3(512), 0, st, a, xb, WB_);        \
        else hipLaunchKernelGGL((k_conv_tb<2, SUB_, PAIR_>), grid, dim3(512), 0, st, a, xb, WB_);                               \
    } while (0)
    if (a.stride == 2) M355_TB(2, 0, wb);
    else if (pair) M355_TB(1, 1, 2u * a.cls_w_elems * 2u);   // the resource spans the two classes of a pair
    else M355_TB(1, 0, wb);
#undef M355_TB
    note_kernel("k_conv_tb");
    return check_launch("conv2d (halo, tile pairs)");
}

}  // namespace m355
