"""Cycle anatomy of k_conv_halo's main loop (8-wave 2x2 variants) from in-kernel s_memtime stamps.

Needs the debug build:  M355_BUILD_LIB=libstamp.so M355_BUILD_DEFS=-DM355_DBG_STAMP python 2dimageto3dmodel_amd/build.py
Run on the GPU box:      M355_LIB=libstamp.so python scripts/stamp_halo.py
Per step and wave of workgroup (0,0): t0 = before the counted wait, t1 = behind the barrier, t2 = behind the DMA issue
(after the step's first four MFMAs), t3 = behind the last MFMA.  Printed: wait = t1-t0, p1 = t2-t1, p2 = t3-t2,
tail = next t0 - t3 (shader clock cycles, 100 MHz-independent: s_memtime counts at the shader clock)."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS, TILES = 96, 8
buf = torch.zeros(8 * STEPS * 4 + 2 + 8 * TILES * 4, dtype=torch.int32, device="cuda")


def report(name):
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().astype(np.uint32).astype(np.int64)
    a, (kc, kr) = raw[:8 * STEPS * 4].reshape(8, STEPS, 4), raw[8 * STEPS * 4:8 * STEPS * 4 + 2]
    epi = raw[8 * STEPS * 4 + 2:].reshape(8, TILES, 4)
    buf.zero_()
    n = int((a[0, :, 1] != 0).sum())
    if n < 8:
        print(name, "no stamps (is M355_LIB the -DM355_DBG_STAMP build?)"); return
    a = a[:, :n]
    d = lambda x, y: (x - y) & 0xffffffff
    wait, p1, p2 = d(a[:, :, 1], a[:, :, 0]), d(a[:, :, 2], a[:, :, 1]), d(a[:, :, 3], a[:, :, 2])
    tail = d(a[:, 1:, 0], a[:, :-1, 3])
    step = d(a[:, 1:, 1], a[:, :-1, 1])
    print(f"== {name}: {n} steps stamped; cycles per step (median over steps 8..) per wave")
    if kr > 0:
        print(f"  whole kernel, workgroup (0,0): {kc} shader cycles in {kr} ticks of the 100 MHz reference = {kr / 100.0:.1f} us -> "
              f"effective shader clock {kc / kr * 0.1:.3f} GHz")
    for w in range(8):
        s = slice(8, n - 1)
        print(f"  wave {w}: step {np.median(step[w, s]):7.0f}  wait {np.median(wait[w, s]):6.0f}  p1(4 mfma + dma issue) {np.median(p1[w, s]):6.0f}"
              f"  p2(12 mfma + 16 reads) {np.median(p2[w, s]):6.0f}  tail {np.median(tail[w, s]):5.0f}")
    print("  wave 0, steps 8..23 (step wait p1 p2 tail):")
    for k in range(8, min(24, n - 1)):
        print(f"    {k:3d} {step[0, k]:6d} {wait[0, k]:6d} {p1[0, k]:6d} {p2[0, k]:6d} {tail[0, k]:6d}")
    print("  barrier release skew across waves (t1 - min t1), step 10:", (a[:, 10, 1] - a[:, 10, 1].min()).tolist())
    # tile epilogues (round 6): e0 = behind the last MFMA step, e1 = results converted and all stores issued, e2 = accumulators
    # re-initialised, e3 = next tile's prologue (tile origin, mask-word loads) done; then the next tile's first barrier
    nt = int((epi[0, :, 3] != 0).sum())
    if nt >= 2:
        print("  tile epilogues, per wave (median over tiles 1..): convert + stores | init_acc | next prologue | e0 -> e3 total;"
              "  skew of e3 across waves")
        for w in range(8):
            t = slice(1, nt)
            print(f"  wave {w}: {np.median(d(epi[w, t, 1], epi[w, t, 0])):6.0f} | {np.median(d(epi[w, t, 2], epi[w, t, 1])):5.0f} | "
                  f"{np.median(d(epi[w, t, 3], epi[w, t, 2])):5.0f} | {np.median(d(epi[w, t, 3], epi[w, t, 0])):6.0f}")
        print("  e0 skew across waves, tile 2:", (epi[:, 2, 0] - epi[:, 2, 0].min()).tolist(), " e1 skew:", (epi[:, 2, 1] - epi[:, 2, 1].min()).tolist(),
              " e3 skew:", (epi[:, 2, 3] - epi[:, 2, 3].min()).tolist())


def timeit(f, n=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


os.environ["M355_STAMP_PTR"] = hex(buf.data_ptr())
# round 6: D.conv2's dgrad (class PAIRS, K = 512 per tile) with and without the bit-mask epilogue
d2 = conv.make_desc(B, 256, 256, 64, 128, 4, 4, 2, 1, 1, 2, 0)
w2 = torch.randn(128, 64, 4, 4, device="cuda") * 0.02
_, wd2 = conv.weight_prep(d2, w2)
dy2 = torch.randn(B, 128, 128, 128, device="cuda").bfloat16()
bits2 = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 256, 256, 1, 2), dtype=torch.int32, device="cuda")
for tag, kw in (("plain epilogue", {}), ("bit-mask epilogue", dict(mask_bits=bits2, mask_slope=0.2))):
    for _ in range(2):
        conv.conv_dgrad(d2, dy2, wd2, **kw)
    print("  HIP-event time dgrad %.1f us" % timeit(lambda: conv.conv_dgrad(d2, dy2, wd2, **kw)))
    report("D.conv2 64->128 4x4 s2 dgrad (class PAIRS), " + tag + " " + conv.lib().m355_last_kernel().decode())
for name, H, Cin, Cout in (("D.conv3 128->256 4x4 s2", 128, 128, 256), ("D.conv4 256->512 4x4 s2", 64, 256, 512)):
    d = conv.make_desc(B, H, H, Cin, Cout, 4, 4, 2, 1, 1, 2, 0)
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.02
    wf, wd = conv.weight_prep(d, w)
    dy = torch.randn(B, H // 2, H // 2, Cout, device="cuda").bfloat16()
    for _ in range(2):
        conv.conv_fwd(d, x, wf, slope=0.2)
    print("  HIP-event time fwd %.1f us" % timeit(lambda: conv.conv_fwd(d, x, wf, slope=0.2)))
    report(name + " fwd (SUB=2) " + conv.lib().m355_last_kernel().decode())
    for _ in range(2):
        conv.conv_dgrad(d, dy, wd)
    print("  HIP-event time dgrad %.1f us" % timeit(lambda: conv.conv_dgrad(d, dy, wd)))
    report(name + " dgrad (4 classes) " + conv.lib().m355_last_kernel().decode())
