"""Sustained socket power / shader clock per kernel family (each run back to back for a few seconds, rocm-smi sampled from a
thread): which kernels sit at the 1400 W cap (there the clock, not the schedule, sets the rate)?  -> profiles/r03_power_table.txt"""
import importlib, os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv"); ops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5


def probe(name, f, flops=0.0, nbytes=0.0):
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
            m, p = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out), re.search(r"Power \(W\): ([\d.]+)", out)
            if m and p: samples.append((int(m.group(1)), float(p.group(1))))
            time.sleep(0.15)
    for _ in range(5): f()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(40): f()
        n += 40; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop[0] = True; th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    s = samples[2:] if len(samples) > 4 else samples
    kern = conv.lib().m355_last_kernel().decode() if flops else "-"
    rate = f"{flops / us / 1e6:7.0f} TF" if flops else f"{nbytes / us / 1e6:7.2f} TB/s"
    print(f"{name:44s} {kern:14s} {us:8.1f} us {rate}  sclk {sum(a for a, _ in s) / max(len(s), 1):5.0f} MHz  power "
          f"{sum(b for _, b in s) / max(len(s), 1):5.0f} W  ({len(s)} samples)", flush=True)


def conv_case(name, B, H, W, Cin, Cout, k, s, ph, pw, mode, ups, which):
    d = conv.make_desc(B, H, W, Cin, Cout, k, k, s, ph, pw, mode, ups)
    ho, wo = conv.out_hw(d)
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16(); w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    wf, wd = conv.weight_prep(d, w); dy = torch.randn(B, ho, wo, conv.dy_channels(Cout), device="cuda").bfloat16()
    fl = 2.0 * B * ho * wo * Cout * Cin * k * k
    f = {"fwd": lambda: conv.conv_fwd(d, x, wf), "dgrad": lambda: conv.conv_dgrad(d, dy, wd), "wgrad": lambda: conv.conv_wgrad(d, x, dy)}[which]
    probe(f"{name} {which}", f, flops=fl)


print(f"# each kernel back to back for {secs} s; rocm-smi sclk / socket power averaged over the run (idle: ~110 MHz, ~235 W)")
conv_case("D.conv3 128->256 4x4 s2 B128", 128, 128, 128, 128, 256, 4, 2, 1, 1, 2, 0, "fwd")
conv_case("D.conv3 128->256 4x4 s2 B128", 128, 128, 128, 128, 256, 4, 2, 1, 1, 2, 0, "wgrad")
conv_case("D.conv2 64->128 4x4 s2 B128", 128, 256, 256, 64, 128, 4, 2, 1, 1, 2, 0, "dgrad")
conv_case("G.blk5.conv2 128->128 3x3 B64", 64, 128, 64, 128, 128, 3, 1, 1, 1, 1, 0, "fwd")
conv_case("G.blk6.conv2 64->64 3x3 B64", 64, 256, 128, 64, 64, 3, 1, 1, 1, 1, 0, "fwd")
conv_case("D.conv1 8->64 5x5 B128", 128, 256, 256, 8, 64, 5, 1, 2, 2, 2, 0, "fwd")
conv_case("D.conv1 8->64 5x5 B128", 128, 256, 256, 8, 64, 5, 1, 2, 2, 2, 0, "wgrad")
conv_case("G.blk1.conv1 512->512 3x3 8x4 B64", 64, 8, 4, 512, 512, 3, 1, 1, 1, 1, 0, "fwd")
# an HBM-bound elementwise pass (the fused CBN apply): 64 x 256 x 128 x 64 bf16 in, same out
x = torch.randn(64, 256, 128, 64, device="cuda").bfloat16(); a = torch.rand(64, 64, device="cuda"); b = torch.rand(64, 64, device="cuda"); y = torch.empty_like(x)
from importlib import import_module
L = import_module("2dimageto3dmodel_amd._lib")
probe("affine_act_fwd 64x256x128x64 (CBN apply)", lambda: L.launch("affine_act_fwd", L.ptr(x), L.ptr(a), L.ptr(b), None, 0, L.ptr(y), 64, 256 * 128, 64, 0.2, 1.0, L.stream()),
      nbytes=2.0 * x.numel() * 2)
