"""The thin 8-channel layers (k_conv_c8 forward / head dgrad, k_wgrad_c8) back to back: time, bytes / time, clock, power."""
import importlib, os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
def probe(name, f, nbytes):
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
    print(f"{name:44s} {conv.lib().m355_last_kernel().decode():14s} {us:8.1f} us  {nbytes / us / 1e6:5.2f} TB/s  sclk {sum(a for a, _ in s) / max(len(s), 1):5.0f} MHz  power "
          f"{sum(b for _, b in s) / max(len(s), 1):5.0f} W", flush=True)
B = 128
d = conv.make_desc(B, 256, 256, 8, 64, 5, 5, 1, 2, 2, 2, 0)
x = torch.randn(B, 256, 256, 8, device="cuda").bfloat16(); w = torch.randn(64, 8, 5, 5, device="cuda") * 0.05; b = torch.randn(64, device="cuda")
wf, wd = conv.weight_prep(d, w); dy = torch.randn(B, 256, 256, 64, device="cuda").bfloat16()
nb_y, nb_x = B * 256 * 256 * 64 * 2, B * 256 * 256 * 8 * 2
probe("D.conv1 8->64 5x5 B128 fwd (bias, lrelu)", lambda: conv.conv_fwd(d, x, wf, b, slope=0.2), nb_y + nb_x)
probe("D.conv1 fwd + sign bits", lambda: conv.conv_fwd(d, x, wf, b, slope=0.2, emit_bits=True), nb_y + nb_x + nb_y // 16)
probe("D.conv1 wgrad", lambda: conv.conv_wgrad(d, x, dy), nb_y + nb_x)
# the generator head's dgrad (64 -> 3, 5x5, replicate pad): dy 8-channel -> dx 64 channels, masked by the producer's activation
dh = conv.make_desc(64, 256, 128, 64, 3, 5, 5, 1, 2, 2, 1, 0)
xh = torch.randn(64, 256, 128, 64, device="cuda").bfloat16(); wh = torch.randn(3, 64, 5, 5, device="cuda") * 0.05
_, wdh = conv.weight_prep(dh, wh); dyh = torch.randn(64, 256, 128, 8, device="cuda").bfloat16()
nb = 64 * 256 * 128 * 64 * 2
probe("G head 64->3 dgrad, mask_x (B64 256x128)", lambda: conv.conv_dgrad(dh, dyh, wdh, mask_x=xh, mask_slope=0.2), 2 * nb + nb // 8)
probe("G head 64->3 dgrad, no mask", lambda: conv.conv_dgrad(dh, dyh, wdh), nb + nb // 8)
