"""What costs k_conv_wt its MFMA-busy fraction?  The -DM355_DBG_ABLATE build switches parts of the kernel off (results are then
WRONG; only the time means something): M355_WT_ABLATE bits 1 no epilogue, 2 no bias loads, 4 contiguous halo DMA addresses,
8 contiguous weight DMA addresses,
16 no barrier, 32 no bit-mask prefetch, 128 no L2 touch of the halo's other half-lines.
Build:  M355_BUILD_LIB=libablate.so M355_BUILD_DEFS=-DM355_DBG_ABLATE python 2dimageto3dmodel_amd/build.py
Run:    M355_LIB=libablate.so python scripts/wt_ablate.py"""
import importlib, os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
def probe(name, f, flops):
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
            m, p = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out), re.search(r"Power \(W\): ([\d.]+)", out)
            if m and p: samples.append((int(m.group(1)), float(p.group(1))))
            time.sleep(0.15)
    for _ in range(5): f()
    torch.cuda.synchronize(); th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(40): f()
        n += 40; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop[0] = True; th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    s = samples[2:] if len(samples) > 4 else samples
    sclk = sum(a for a, _ in s) / max(len(s), 1)
    print(f"{name:58s} {conv.lib().m355_last_kernel().decode():12s} {us:8.1f} us {flops / us / 1e6:6.0f} TF  sclk {sclk:5.0f} MHz  power "
          f"{sum(b for _, b in s) / max(len(s), 1):5.0f} W  busy {flops / us / 1e6 / (2500.0 * sclk / 2400.0):.2f}", flush=True)
os.environ["M355_WT"] = "1"
B, H, Cin, Cout = 128, 128, 128, 256
d = conv.make_desc(B, H, H, Cin, Cout, 4, 4, 2, 1, 1, 2, 0)
ho, wo = conv.out_hw(d)
x = torch.randn(B, H, H, Cin, device="cuda").bfloat16(); w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.05
b = torch.randn(Cout, device="cuda")
wf, wd = conv.weight_prep(d, w)
fl = 2.0 * B * ho * wo * Cout * Cin * 16
for abl, what in ((0, "as shipped"), (128, "without the L2 touch of the other half-lines"), (1, "no epilogue"), (4, "contiguous halo DMA"),
                  (8, "contiguous weight DMA"), (12, "contiguous halo + weight DMA"), (13, "no epilogue, contiguous DMAs"), (16, "no barrier"),
                  (29, "all of the above")):
    os.environ["M355_WT_ABLATE"] = str(abl)
    probe(f"D.conv3 fwd (bias, lrelu, bits)  ablate {abl:2d}: {what}", lambda: conv.conv_fwd(d, x, wf, b, slope=0.2, emit_bits=True), fl)
os.environ["M355_WT_ABLATE"] = "0"; os.environ["M355_NO_WT"] = "1"
probe("D.conv3 fwd (bias, lrelu, bits)  k_conv_halo, same box", lambda: conv.conv_fwd(d, x, wf, b, slope=0.2, emit_bits=True), fl)
