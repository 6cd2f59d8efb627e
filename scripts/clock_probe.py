"""Sustained shader clock / socket power while ONE conv kernel runs back to back (rocm-smi sampled from a thread):
is a kernel that spends fewer cycles per MFMA paid back in clock?  usage: clock_probe.py [fwd|dgrad] [seconds]"""
import importlib, os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
what, secs = (sys.argv[1] if len(sys.argv) > 1 else "fwd"), float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
B, H, Cin, Cout = 128, 128, 128, 256
d = conv.make_desc(B, H, H, Cin, Cout, 4, 4, 2, 1, 1, 2, 0)
x = torch.randn(B, H, H, Cin, device="cuda").bfloat16(); w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.02
wf, wd = conv.weight_prep(d, w); dy = torch.randn(B, H // 2, H // 2, Cout, device="cuda").bfloat16()
f = (lambda: conv.conv_fwd(d, x, wf, slope=0.2)) if what == "fwd" else (lambda: conv.conv_dgrad(d, dy, wd))
samples, stop = [], False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        m, p = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out), re.search(r"Power \(W\): ([\d.]+)", out)
        if m and p: samples.append((int(m.group(1)), float(p.group(1))))
        time.sleep(0.2)
for _ in range(5): f()
torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
while time.perf_counter() - t0 < secs:
    for _ in range(50): f()
    n += 50; torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize(); stop = True; th.join()
us = e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * B * (H // 2) ** 2 * Cout * Cin * 16
s = samples[2:] if len(samples) > 4 else samples
print(f"{conv.lib().m355_last_kernel().decode()} {what}: {us:.1f} us = {fl / us / 1e6:.0f} TF sustained over {secs:.0f} s; sclk "
      f"{sum(a for a, _ in s) / max(len(s), 1):.0f} MHz, power {sum(b for _, b in s) / max(len(s), 1):.0f} W ({len(s)} samples)")
