// k_conv_wt: the 2x2 class convolutions of the discriminators (forward of the 4x4 stride-2 convs as four accumulated parity
// classes, and the four classes of their dgrad) with a 128-pixel x 64-channel WAVE tile.
//
// Why (profiles/r03_power_roof.txt): a 64x64 wave tile reads 1 KB of LDS per MFMA -- at full MFMA rate exactly the 128 B/clk
// a CU's LDS delivers, so k_conv_halo's main loop can keep the pipe at most 0.78 busy and, with a barrier per tap and its
// counted waits, holds 0.61.  Here a wave owns 4 rows x 32 pixels x 64 output channels (eight 32x32 accumulators, 128
// registers): a k-group of 16 channels is 4 pixel + 2 weight fragments for 8 MFMAs, 0.75 KB per MFMA.  A workgroup (8 waves =
// 4 pixel groups x 2 channel halves, one per CU) owns 16 x 32 pixels x 128 channels and walks (class, 32-CHANNEL chunk)
// segments: 17 x 33 halo pixels x 64 B = 36 KB and the four taps' weights 4 x 8 KB, BOTH double buffered (136 KB of LDS),
// so a segment is   s_waitcnt vmcnt(0); s_barrier; issue the next segment's 9 DMAs per wave; 64 MFMAs per wave with no
// barrier inside   -- one barrier per 64 MFMAs instead of one per 16, 0.13 KB of L2 -> LDS traffic per MFMA instead of 0.2.
// The schedule alone (scripts/probes/power_roof.hip k_roof_wt: same LDS layout, DMA volume and MFMA / read sequence, N(0,1)
// operands, accumulators restarted per tile, no epilogue, DMA sources contiguous) sustains 1.33 PF at 0.73 busy on the box
// where k_conv_halo's D.conv3 runs 1.08-1.11 PF; the real kernel lands at k_conv_halo's rate (see conv_wt_eligible).
//
// LDS layouts (64-byte rows: a pixel's / an output channel's 32-channel chunk = four 16-byte slots):
//   slot s of row r holds source chunk s ^ ((r >> 2) & 3): the 16 rows x 1 slot a quarter-wave reads cover all 64 banks once.
//   The DMA writes lane-linear (lane l -> row 16 q + (l >> 2), slot l & 3), so it FETCHES chunk (l & 3) ^ ((l >> 4) & 3).
// Accumulation order differs from k_conv_halo (32-channel chunks): results agree to fp32 rounding, not bit for bit.
#include <stdlib.h>

#include <type_traits>

#include <cstring>
#include "conv_dma.h"

namespace m355 {

// debug build only (-DM355_DBG_ABLATE, scripts/wt_ablate.py): switch parts of the kernel off (WRONG results) to see what each costs
#ifdef M355_DBG_ABLATE
#define M355_ABL(bit) ((abl & (bit)) != 0)
#else
#define M355_ABL(bit) false
#endif

template <int MODE, int SUB>
__global__ __launch_bounds__(512, 2) void k_conv_wt(ConvArgs a, unsigned xbytes, unsigned wbytes, int abl)
{
    (void)abl;
    constexpr int BN = 128, NW = 8, KS = 2, T = 4, TH = 16, TW = 32;
    constexpr int NC = SUB == 2 ? 4 : 1;                  // classes accumulated into one output tile (stride-2 forward)
    constexpr int HH = TH + KS - 1, HWD = TW + KS - 1, HR = HH * HWD;   // 17 x 33 = 561 halo pixels
    constexpr int NA = (HR + 15) / 16, NAW = (NA + NW - 1) / NW;        // 36 DMA instructions, <= 5 per wave
    constexpr int HB = NA * 1024, WB = BN * 64;                         // 36 KB halo, 8 KB per tap
    constexpr int PI = 4, CJ = 2;
    constexpr int STG = 2048;                                           // epilogue stage per wave: 16 pixels x 128 B
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * HB + 2 * T * WB + NW * STG];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nN = a.CoutP / BN;
    const int tn = blockIdx.x % nN, bp = blockIdx.x / nN, PS = gridDim.x / nN;
    const int tpx = a.Wo / TW, tpy = a.Ho / TH, tiles_p = a.N * tpx * tpy;
    const int n0 = tn * BN;
    if (bp >= tiles_p) return;

    int pad_h = a.pad_h, pad_w = a.pad_w, oy_off = a.oy_off, ox_off = a.ox_off;
    const unsigned short *wv = a.w;
    if (a.ncls > 1) {
        const int cls = blockIdx.y;
        pad_h = a.cpad_h[cls]; pad_w = a.cpad_w[cls]; oy_off = a.coy[cls]; ox_off = a.cox[cls];
        wv += (size_t)cls * a.cls_w_elems;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wv, 0, wbytes, 0x00020000);

    // DMA roles: instruction q = NW * k + wave covers halo pixels 16 q .. 16 q + 15 (weights: rows 16 wave .. + 15 of a tap)
    const unsigned csrc16 = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    unsigned aoff[NAW], hyx[NAW];
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
        const int rho = 16 * (NW * k + wave) + (lane >> 2);
        const int hy = rho / HWD, hx = rho - hy * HWD;
        hyx[k] = rho < HR ? (unsigned)((SUB * hy) << 16 | (SUB * hx)) : 0x7fff0000u;
    }
    const unsigned cin2 = (unsigned)a.Cin * 2u;
    auto tile_origin = [&](int tp, int &n, int &oy0, int &ox0) {
        n = tp / (tpx * tpy);
        const int trem = tp - n * (tpx * tpy);
        oy0 = (trem / tpx) * TH;
        ox0 = (trem % tpx) * TW;
    };
    auto compute_aoff = [&](int tp, int cls) {
        int n, oy0, ox0;
        tile_origin(tp, n, oy0, ox0);
        const int Yb = SUB == 2 ? 2 * oy0 + (cls >> 1) - pad_h : oy0 - pad_h;
        const int Xb = SUB == 2 ? 2 * ox0 + (cls & 1) - pad_w : ox0 - pad_w;
        const unsigned nbase = (unsigned)(n * a.H * a.W) * cin2;
#pragma unroll
        for (int k = 0; k < NAW; ++k) {
            const int iy = Yb + (int)(hyx[k] >> 16);
            int ix = Xb + (int)(hyx[k] & 0xffffu);
            if (MODE == 1) ix = min(max(ix, 0), a.W - 1);
            else if (MODE == 2) ix = ix < 0 ? ix + a.W : (ix >= a.W ? ix - a.W : ix);
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned off = __umul24(__umul24(iy, a.W) + ix, cin2) + (nbase + csrc16);
            aoff[k] = ok ? off : OOB;
            if (M355_ABL(4)) aoff[k] = nbase + (unsigned)(oy0 * a.W + ox0) * cin2 + (unsigned)((NW * k + wave) * 1024 + lane * 16);   // contiguous KBs
        }
    };
    const unsigned wrow = (unsigned)(n0 + 16 * wave + (lane >> 2)) * (unsigned)(a.Kp * 2) + csrc16;
    const int ncc = a.Cin >> 5, NSQ = NC * ncc;   // 32-channel chunks; NSQ is even (Cin % 64 == 0)

    // the DMAs of segment (class, chunk) of the tile aoff[] describes -> buffer half `buf`
    auto issue = [&](int buf, int cls, int cc) {
        unsigned char *const hn = lds + buf * HB;
#pragma unroll
        for (int k = 0; k < NAW; ++k)
            if (NW * k + wave < NA) dma16(rx, hn + (NW * k + wave) * 1024, aoff[k], (unsigned)cc * 64u);
        unsigned char *const wn_ = lds + 2 * HB + buf * T * WB + wave * 1024;
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            const int ktap = SUB == 2 ? (2 * (tap >> 1) + (cls >> 1)) * 4 + 2 * (tap & 1) + (cls & 1) : tap;
            if (M355_ABL(8)) dma16(rw, wn_ + tap * WB, (unsigned)(n0 * a.Kp * 2 + (ktap * ncc + cc) * 8192 + wave * 1024 + lane * 16) % wbytes, 0u);
            else dma16(rw, wn_ + tap * WB, wrow, (unsigned)(ktap * a.Cin + cc * 32) * 2u);
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int tx = lane & 31, half = lane >> 5;
    f32x16 acc[CJ][PI];
    auto init_acc = [&]() {
        if (a.bias && !M355_ABL(2)) {
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = n0 + wn * 64 + 32 * j + 8 * g + 4 * half;
                    const float4 b = co < a.Cout ? *reinterpret_cast<const float4 *>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < PI; ++i) {
                        acc[j][i][4 * g] = b.x; acc[j][i][4 * g + 1] = b.y; acc[j][i][4 * g + 2] = b.z; acc[j][i][4 * g + 3] = b.w;
                    }
                }
        } else {
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int i = 0; i < PI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;
        }
    };
    init_acc();

    // fragment addresses (k-group 0; k-group 1 = ^ 0x20): pixel rows 4 wm + r (r = 0 .. 4) x column shift kw, weight rows 32 j
    unsigned pa[PI + 1][KS], wa[CJ];
#pragma unroll
    for (int r = 0; r < PI + 1; ++r)
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
            const int rho = (4 * wm + r) * HWD + tx + kw;
            pa[r][kw] = (unsigned)(rho * 64 + ((half ^ ((rho >> 2) & 3)) << 4));
        }
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
        const int row = wn * 64 + 32 * j + tx;
        wa[j] = (unsigned)(2 * HB + row * 64 + ((half ^ ((row >> 2) & 3)) << 4));
    }
    unsigned short *yb = reinterpret_cast<unsigned short *>(a.y);

    struct Frags {
        bf16x8 p[PI], w[CJ];
    };
    // group g = 2 * tap + kk of buffer half BUF_
    auto rd = [&](Frags &f, auto bufc, auto gc) {
        constexpr int BUF_ = decltype(bufc)::value, g = decltype(gc)::value, tap = g >> 1, kk = g & 1, kh = tap / KS, kw = tap % KS;
#pragma unroll
        for (int i = 0; i < PI; ++i)
            f.p[i] = *reinterpret_cast<const bf16x8 *>(lds + BUF_ * HB + (pa[i + kh][kw] ^ (kk ? 0x20u : 0u)));
#pragma unroll
        for (int j = 0; j < CJ; ++j)
            f.w[j] = *reinterpret_cast<const bf16x8 *>(lds + BUF_ * T * WB + tap * WB + (wa[j] ^ (kk ? 0x20u : 0u)));
    };
    auto mm = [&](const Frags &f) {
#pragma unroll
        for (int j = 0; j < CJ; ++j)
#pragma unroll
            for (int i = 0; i < PI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[j], f.p[i], acc[j][i], 0, 0, 0);
    };

    // ---- prologue: segment 0 of the first tile
    int tp = bp;
    compute_aoff(tp, 0);
    issue(0, 0, 0);

    Frags f0, f1;
    for (;;) {
        const int tp_next = tp + PS;
        const bool has_next = tp_next < tiles_p;
        int n, oy0, ox0;
        tile_origin(tp, n, oy0, ox0);
        unsigned rbits_pf[PI] = {};
        if (a.bits_in && !M355_ABL(32)) {
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                const size_t pix = ((size_t)n * a.OH + ((oy0 + 4 * wm + i) * a.oy_mul + oy_off)) * a.OW + ((ox0 + tx) * a.ox_mul + ox_off);
                rbits_pf[i] = a.bits_in[(pix * (size_t)(a.Cs >> 6) + (size_t)((n0 >> 6) + wn)) * 2 + half];
            }
        }
        int cls_cur = 0, cc_cur = 0;
        for (int sq = 0; sq < NSQ; sq += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {   // (buffer half = segment parity: NSQ is even, every tile starts in half 0)
                // the next segment: this tile's next (class, chunk), or -- behind the last one, where (class, chunk) wrap to
                // (0, 0) -- segment 0 of the next tile (a harmless re-load of this tile's when there is none)
                int cls_n = cls_cur, cc_n = cc_cur + 1;
                if (cc_n == ncc) {
                    cc_n = 0;
                    cls_n = cls_cur + 1 == NC ? 0 : cls_cur + 1;
                }
                const bool last = sq + h + 1 == NSQ;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this segment's halo and weights (issued one segment ago)
                if (!M355_ABL(16)) __builtin_amdgcn_s_barrier();   // ... everybody's; and nobody still reads the other half
                __builtin_amdgcn_sched_barrier(0);
                if (cc_n == 0) compute_aoff(last && has_next ? tp_next : tp, cls_n);   // new (tile, class): new halo origin
                if (h == 0) issue(1, cls_n, cc_n);
                else issue(0, cls_n, cc_n);
                __builtin_amdgcn_sched_barrier(0);
                auto run = [&](auto bufc) {
                    rd(f0, bufc, std::integral_constant<int, 0>{});
                    __builtin_amdgcn_sched_barrier(0);
#define M355_WT_PAIR(G0_)                                                                                          \
    rd(f1, bufc, std::integral_constant<int, G0_ + 1>{});                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    mm(f0);                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    if constexpr (G0_ + 2 < 8) rd(f0, bufc, std::integral_constant<int, (G0_ + 2 < 8 ? G0_ + 2 : 0)>{});            \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    mm(f1);                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);
                    M355_WT_PAIR(0)
                    M355_WT_PAIR(2)
                    M355_WT_PAIR(4)
                    M355_WT_PAIR(6)
#undef M355_WT_PAIR
                };
                if (h == 0) run(std::integral_constant<int, 0>{});
                else run(std::integral_constant<int, 1>{});
                cls_cur = cls_n;
                cc_cur = cc_n;
            }
        }

        // ---- epilogue (the next tile's first segment is in flight):
        // acc[j][i][r] = channel n0 + 64 wn + 32 j + 8 (r >> 2) + 4 half + (r & 3), pixel (4 wm + i, tx) of the tile
        // Stores: straight from the MFMA layout a lane owns 16 bytes of a pixel and its half-wave partner the next 16, so a
        // store instruction would touch 32 pixels with 32 bytes each -- four instructions, four partial writes per 128-byte line
        // (8.4 M write requests per D.conv3 launch; with the stores switched off the kernel runs 11 % faster).  Instead each
        // half row (16 pixels x 64 channels = 2 KB) goes through a wave-private LDS stage and leaves as two store instructions
        // of 8 pixels x one whole line: position of (pixel p, 16-byte chunk c) = 128 p + 16 (c ^ (p & 7)), conflict free both ways.
        unsigned char *const stg = lds + 2 * HB + 2 * T * WB + wave * STG;
        auto store_tile = [&](auto plainc, auto maskc) {
            constexpr bool PLAIN = decltype(plainc)::value, MASK = decltype(maskc)::value;
            const int sp = lane >> 3, sc = lane & 7;   // read side: pixel sp (+ 8) of the half row, chunk sc
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                const int ho = oy0 + 4 * wm + i, wo = ox0 + tx;
                const size_t prow = ((size_t)n * a.OH + (ho * a.oy_mul + oy_off)) * a.OW + ox_off;   // + wo * ox_mul = the pixel
                const size_t pix = prow + (size_t)wo * a.ox_mul;
                const size_t bword = (pix * (size_t)(a.Cs >> 6) + (size_t)((n0 >> 6) + wn)) * 2 + half;
                unsigned wbits = 0;
                const unsigned rbits = MASK ? rbits_pf[i] : 0u;
                const bool emit_bits = !PLAIN && a.bits_out != nullptr;
                uint4 o[CJ][2];   // this lane's four 16-byte chunks: chunk 4 j + g + half (g = 0, 2)
#pragma unroll
                for (int j = 0; j < CJ; ++j) {
                    uint2 pk[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4] = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
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
                        o[j][g >> 1].x = sx[0]; o[j][g >> 1].y = sy[0]; o[j][g >> 1].z = sx[1]; o[j][g >> 1].w = sy[1];
                    }
                }
                if (emit_bits) a.bits_out[bword] = wbits;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the previous pass's stage reads have returned)
                    if ((tx >> 4) == pass) {
                        const int p = tx & 15;
#pragma unroll
                        for (int j = 0; j < CJ; ++j)
#pragma unroll
                            for (int q = 0; q < 2; ++q)
                                *reinterpret_cast<uint4 *>(stg + p * 128 + (((4 * j + 2 * q + half) ^ (p & 7)) << 4)) = o[j][q];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int p = sp + 8 * m;
                        const uint4 v = *reinterpret_cast<const uint4 *>(stg + p * 128 + ((sc ^ (p & 7)) << 4));
                        const size_t px = prow + (size_t)(ox0 + 16 * pass + p) * a.ox_mul;
                        *reinterpret_cast<uint4 *>(yb + px * a.Cs + (n0 + wn * 64 + 8 * sc)) = v;
                    }
                }
            }
        };
        if (M355_ABL(1)) {   // no epilogue (one conditional store keeps the accumulators alive)
            float sacc = 0.0f;
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int i = 0; i < PI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += acc[j][i][r];
            if (sacc == 12345.678f) yb[tid] = 1;
        } else {
            using std::false_type;
            using std::true_type;
            const bool plain = a.slope == 1.0f;
            if (plain && !a.bits_in) store_tile(true_type{}, false_type{});
            else if (plain) store_tile(true_type{}, true_type{});
            else if (!a.bits_in) store_tile(false_type{}, false_type{});
            else store_tile(false_type{}, true_type{});
        }
        if (!has_next) break;
        init_acc();
        tp = tp_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (unused) prefetch
}

// which k_conv_halo problems run on the wide wave tile: the 8-wave 2x2 class kernels with the unguarded epilogue
bool conv_wt_eligible(const ConvArgs &a)
{
    // OPT-IN (M355_WT=1).  Measured same-box against k_conv_halo at batch 128 (profiles/r03_wt_vs_halo.txt): D.conv3 forward
    // 503 vs 507 us, its dgrad 520 vs 522, D.conv4 458 / 459 vs 463 / 455 -- equal within a percent, the socket at its 1400 W
    // limit under either kernel (this one at a higher clock and a lower MFMA-busy fraction).  What the wide tile saves (25 % of
    // the fragment reads, 3 of 4 barriers, a third of the L2 -> LDS bytes) it pays back in the 64-byte halo fetches its 32-channel
    // chunks force on it: with the same bytes fetched from contiguous addresses it runs 11-15 % faster (scripts/wt_ablate.py).
    const char *on = getenv("M355_WT");
    if (!on || on[0] == '0' || getenv("M355_NO_WT")) return false;
    if (a.y_f32_nchw || a.fold2 || a.mask_x || a.stats || a.ups || a.Cout != a.CoutP || a.Cin % 64 || a.Wo % 32 || a.Ho % 16 || a.Cs % 8 ||
        a.CoutP % 128)
        return false;
    if (a.stride == 2)   // forward of a 4x4 stride-2 conv (four accumulated classes)
        return a.KH == 4 && a.KW == 4 && a.pad_h == 1 && a.pad_w == 1 && a.ncls <= 1 && a.H == 2 * a.Ho && a.W == 2 * a.Wo;
    return a.stride == 1 && a.KH == 2 && a.KW == 2 && a.ncls == 4;   // the four classes of a stride-2 dgrad
}

int conv_wt_launch(const ConvArgs &a, unsigned xb, unsigned wb, hipStream_t st)
{
    const int tiles = a.N * (a.Ho / 16) * (a.Wo / 32);
    const int nN = a.CoutP / 128, ncls = a.stride == 2 ? 1 : a.ncls;
    const char *wgs = getenv("M355_HALO_WGS");   // tests: few workgroups, several tiles each
    int per = (wgs ? atoi(wgs) : 256) / (nN * ncls);
    if (per < 1) per = 1;
    if (per > tiles) per = tiles;
    // every workgroup the same number of tiles (+-1)
    const int rounds = (tiles + per - 1) / per;
    per = (tiles + rounds - 1) / rounds;
    const dim3 grid((unsigned)per * nN, ncls);
    int abl = 0;
#ifdef M355_DBG_ABLATE
    if (const char *e = getenv("M355_WT_ABLATE")) abl = atoi(e);
#endif
#define M355_WT(SUB_)                                                                                            \
    do {                                                                                                         \
        if (a.pad_w_mode == 0) hipLaunchKernelGGL((k_conv_wt<0, SUB_>), grid, dim3(512), 0, st, a, xb, wb, abl);  \
        else if (a.pad_w_mode == 1) hipLaunchKernelGGL((k_conv_wt<1, SUB_>), grid, dim3(512), 0, st, a, xb, wb, abl); \
        else hipLaunchKernelGGL((k_conv_wt<2, SUB_>), grid, dim3(512), 0, st, a, xb, wb, abl);                    \
    } while (0)
    if (a.stride == 2) M355_WT(2);
    else M355_WT(1);
#undef M355_WT
    note_kernel("k_conv_wt");
    return check_launch("conv2d (halo, 128x64 wave tiles)");
}

}  // namespace m355
