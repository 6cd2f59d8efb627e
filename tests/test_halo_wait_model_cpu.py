"""CPU: a model of k_conv_halo's counted s_waitcnt schedule (csrc/conv_halo.hip) -- the invariant that round 4's race violated.

A wave of the kernel issues, per (tap) step s of a tile, NBW weight DMAs for step s + RB-1 + L and the halo slices of its tap, and
synchronises with ONE `s_waitcnt vmcnt(N)` + barrier per step: vector-memory operations retire in issue order, so "at most N still
in flight" means "all but the youngest N have landed".  N is a compile-time function of the tap (cnt_a / cnt_b in the source), plus
-- in the first `fresh` steps of a tile -- the 8-10 epilogue stores of the previous tile, which are younger than the weights
awaited there.  The model replays the issue order over three tiles and checks, at every wait, that what the step's barrier
publishes has landed: the weights of step s + L and, when s + L opens a channel chunk, that chunk's whole halo.

Rounds 2-3 set fresh = RB-1 + L; the model shows the hole for every look-ahead (L = 1) variant with an epilogue credit, and the
source is checked to carry the rule the model proves (fresh = RB - 1)."""
import itertools
import os
import re

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2dimageto3dmodel_amd", "csrc", "conv_halo.hip")


def halo_dmas_at(t, T, NAW, NAS):           # conv_halo.hip: halo_dmas_at
    t %= T
    if t > T - 3:
        return 0
    lo = t * NAS
    hi = min(lo + NAS, NAW)
    return max(hi - lo, 0)


def wait_count(tap, T, NAW, NAS, RB, L, NBW):   # conv_halo.hip: cnt_a / cnt_b / cnt
    cnt_b = (RB - 2) * NBW + sum(halo_dmas_at(tap - j, T, NAW, NAS) for j in range(1, RB))
    TL = (NAW + NAS - 1) // NAS - 1
    assert TL <= T - 3 and RB >= 2
    cnt_a = (T - 1 - L - TL) * NBW
    return cnt_a if (tap == (T - L) % T and cnt_a < cnt_b) else cnt_b


def simulate(T, NAW, RB, L, NBW, ncc, epi, fresh_rule, tiles=3):
    """-> list of violations (tile, step, what).  Operations are numbered in issue order; `landed(n)` = every operation but the
    youngest n."""
    NAS = (NAW + T - 3) // (T - 2)
    steps = ncc * T                       # steps per tile (one class)
    issued = []                           # (kind, key): ("B", global step it serves) | ("A", global chunk it fills) | ("E", tile)
    bad = []
    g0 = 0                                # global index of the tile's step 0
    # prologue: the first chunk's halo and the weights of steps 0 .. RB-2+L are issued and DRAINED (the kernel's `warm` steps)
    for tile in range(tiles):
        fresh = 0 if tile == 0 else fresh_rule(RB, L)
        for s in range(steps):
            g, tap = g0 + s, s % T
            allowed = wait_count(tap, T, NAW, NAS, RB, L, NBW)
            if fresh > 0:
                fresh -= 1
                allowed = min(63, allowed + epi)
            landed = issued[:max(0, len(issued) - allowed)] if tile > 0 or s >= RB else issued   # (first RB steps: drained)
            landed = set(landed)
            need = [("B", g + L, i) for i in range(NBW)]
            if (s + L) % T == 0:          # step g + L opens a chunk: its halo must be complete
                need += [("A", (g + L) // T, i) for i in range(NAW)]
            for op in need:
                if op in issued and op not in landed:
                    bad.append((tile, s, op))
            # this step's issues: weights of step g + RB-1 + L, then the tap's halo slices of the NEXT chunk
            for i in range(NBW):
                issued.append(("B", g + RB - 1 + L, i))
            lo = tap * NAS
            for i in range(lo, lo + halo_dmas_at(tap, T, NAW, NAS)):
                issued.append(("A", g // T + 1, i))
        for i in range(epi):              # the tile's epilogue: stores (and the next tile's mask-word loads)
            issued.append(("E", tile, i))
        g0 += steps
    return bad


VARIANTS = [  # (T, NAW, RB, L, NBW): the shipped instantiations' shapes -- 8-wave 3x3 (no look-ahead), 8-wave 2x2 classes, 4-wave 3x3 / 2x2
    (9, 5, 3, 0, 2), (9, 3, 3, 0, 2), (4, 4, 3, 1, 2), (4, 5, 3, 1, 2), (9, 10, 3, 1, 2), (9, 6, 3, 1, 2), (4, 8, 3, 1, 2), (4, 9, 3, 1, 2),
    (9, 5, 4, 0, 2), (4, 4, 4, 1, 2), (9, 10, 4, 1, 1), (4, 2, 3, 1, 4),
]


def test_counted_waits_cover_what_each_barrier_publishes():
    rule = lambda RB, L: RB - 1
    for (T, NAW, RB, L, NBW), ncc, epi in itertools.product(VARIANTS, (1, 2, 4), (0, 8, 10)):
        assert simulate(T, NAW, RB, L, NBW, ncc, epi, rule) == [], (T, NAW, RB, L, NBW, ncc, epi)


def test_the_model_sees_the_round_2_3_hole():
    old = lambda RB, L: RB - 1 + L
    holes = {v: simulate(*v, 2, 10, old) for v in VARIANTS}
    for v, bad in holes.items():
        if v[3] == 1:                      # look-ahead variants: step RB-1 of every tile after the first awaits too little (its
                                           # weights; where that step also opens a chunk, the chunk's halo as well)
            assert bad and all(t >= 1 and s == v[2] - 1 for t, s, op in bad) and any(op[0] == "B" for _, _, op in bad), (v, bad[:4])
        else:
            assert bad == [], v
    # ... and without an epilogue credit the old rule was harmless (which is why it survived the kernels without one)
    assert all(simulate(*v, 2, 0, old) == [] for v in VARIANTS)


def test_the_source_carries_the_rule_the_model_proves():
    src = open(SRC).read()
    m = re.search(r"fresh = RES \? 0 : ([^;]+);", src)
    assert m and m.group(1).replace(" ", "") == "RB-1", m and m.group(1)
    # the formulas restated above, as they stand in the source
    assert "constexpr int cnt_b = (RB - 2) * NBW + halo_dmas_behind<T, NAW, NAS, RB - 1>(tap);" in src
    assert "constexpr int cnt_a = (T - 1 - L - TL) * NBW;" in src
    assert "constexpr int cnt = (tap == (T - L) % T && cnt_a < cnt_b) ? cnt_a : cnt_b;" in src


# ---- the two other counted-wait pipelines, same in-order model -------------------------------------------------------------------
def test_glds_four_stage_ring_waits_for_its_own_step():
    """k_conv_glds, NST = 4 (conv_mfma.hip): steps 0 .. NST-2 are staged up front; step t waits with vmcnt(min(2, steps after t) *
    PER), meets the other waves, stages step t + NST-1 into the buffer read in step t-1, then multiplies buffer t % NST"""
    NST = 4
    for PER, nsteps in itertools.product((2, 3, 6), range(1, 12)):
        issued = [(s, i) for s in range(min(NST - 1, nsteps)) for i in range(PER)]
        for t in range(nsteps):
            allowed = min(2, nsteps - 1 - t) * PER
            landed = set(issued[:len(issued) - allowed])
            assert all((t, i) in landed for i in range(PER)), (PER, nsteps, t)
            if t + NST - 1 < nsteps:
                # the buffer it overwrites, (t + NST-1) % NST == (t-1) % NST, was last read in step t-1: behind this step's barrier
                assert (t + NST - 1) % NST == (t - 1) % NST
                issued += [(t + NST - 1, i) for i in range(PER)]


def test_wgrad_c8p_waits_cover_x_and_dy():
    """k_wgrad_c8p (conv_small.hip), waves that also fetch x: prologue = x of tile 0 (2 loads), dy of tiles 0 .. NST-2 (4 DMAs each),
    vmcnt(4 (NST-1)) -> x0.  Per tile: vmcnt(4 (NST-2)) -> this tile's dy; then x of the next tile (2 loads) and the dy of tile
    it + NST-1 (4 DMAs) are issued; at the end vmcnt(4) -> the next tile's x.  Waves without x loads: the same minus the x parts."""
    NST = 4
    for has_x, tiles in itertools.product((True, False), range(1, 9)):
        issued = []
        if has_x:
            issued += [("x", 0, i) for i in range(2)]
        for k in range(NST - 1):
            issued += [("dy", k, i) for i in range(4)]
        done = lambda allowed: set(issued[:len(issued) - allowed])
        if has_x:
            assert all(("x", 0, i) in done(4 * (NST - 1)) for i in range(2))
        for it in range(tiles):
            assert all(("dy", it, i) in done(4 * (NST - 2)) for i in range(4)), (has_x, tiles, it)
            if has_x:
                issued += [("x", it + 1, i) for i in range(2)]
            issued += [("dy", it + NST - 1, i) for i in range(4)]       # (past the last tile the kernel re-fetches the last one)
            if has_x:
                assert all(("x", it + 1, i) in done(4) for i in range(2)), (tiles, it)
