"""k_conv_wt (128x64 wave tiles) against k_conv_halo on the discriminators' stride-2 layers at batch 128: sustained rate, clock,
power (each kernel back to back for a few seconds).  -> profiles/r03_wt_vs_halo.txt"""
import importlib, os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
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
    print(f"{name:40s} {conv.lib().m355_last_kernel().decode():12s} {us:8.1f} us {flops / us / 1e6:6.0f} TF  sclk {sclk:5.0f} MHz  power "
          f"{sum(b for _, b in s) / max(len(s), 1):5.0f} W  busy {flops / us / 1e6 / (2500.0 * sclk / 2400.0):.2f}", flush=True)
for name, (B, H, Cin, Cout) in (("D.conv2 64->128", (128, 256, 64, 128)), ("D.conv3 128->256", (128, 128, 128, 256)), ("D.conv4 256->512", (128, 64, 256, 512))):
    d = conv.make_desc(B, H, H, Cin, Cout, 4, 4, 2, 1, 1, 2, 0)
    ho, wo = conv.out_hw(d)
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16(); w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    wf, wd = conv.weight_prep(d, w); dy = torch.randn(B, ho, wo, Cout, device="cuda").bfloat16()
    bits_in = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, H, H, Cin // 64, 2), dtype=torch.int32, device="cuda")
    fl = 2.0 * B * ho * wo * Cout * Cin * 16
    for env in ("", "1"):
        if env: os.environ.pop("M355_WT", None)
        else: os.environ["M355_WT"] = "1"
        probe(f"{name} fwd (bias, lrelu, bits)", lambda: conv.conv_fwd(d, x, wf, b, slope=0.2, emit_bits=True), fl)
        probe(f"{name} dgrad (bit mask)", lambda: conv.conv_dgrad(d, dy, wd, mask_bits=bits_in, mask_slope=0.2), fl)
