"""The thin (HBM-side) conv kernels back to back: time, algorithmic bytes -> TB/s, shader clock, socket power.
    python scripts/thin_rate.py [seconds per kernel] [filter substring]
D.conv1 (8 -> 64, 5x5): forward k_conv_c8, dgrad k_conv_smallco, wgrad k_wgrad_c8 (at three batch sizes: tile-bound or not);
generator heads (64 -> 3, 5x5): forward k_head5, dgrad k_conv_c8 + edge, wgrad k_wgrad_smallco; D.conv5 (512 -> 1): forward
k_conv_smallco, dgrad k_conv_c8, wgrad k_wgrad_smallco."""
import importlib, os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
flt = sys.argv[2] if len(sys.argv) > 2 else ""
def probe(name, f, nbytes):
    if flt and flt not in name: return
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
        for _ in range(20): f()
        n += 20; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop[0] = True; th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    s = samples[2:] if len(samples) > 4 else samples
    print(f"{name:46s} {conv.lib().m355_last_kernel().decode():16s} {us:8.1f} us {nbytes / us / 1e6:6.2f} TB/s  sclk {sum(a for a, _ in s) / max(len(s), 1):5.0f} MHz  "
          f"power {sum(b for _, b in s) / max(len(s), 1):5.0f} W", flush=True)
def layer(tag, B, H, W, Cin, Cout, k, mode, f32=False, legs="fdw"):
    d = conv.make_desc(B, H, W, Cin, Cout, k, k, 1, k // 2, k // 2, mode, 0)
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    wf, wd = conv.weight_prep(d, w)
    cy = conv.dy_channels(Cout)
    dy = torch.randn(B, H, W, cy, device="cuda").bfloat16()
    P = B * H * W
    if "f" in legs:
        probe(f"{tag} fwd B{B}", lambda: conv.conv_fwd(d, x, wf, b, out_f32_nchw=f32, slope=1.0 if f32 else 0.2), P * (Cin * 2 + Cout * (4 if f32 else 2)))
    if "d" in legs:
        probe(f"{tag} dgrad B{B}", lambda: conv.conv_dgrad(d, dy, wd), P * (cy * 2 + Cin * 2))
    if "w" in legs:
        probe(f"{tag} wgrad B{B}", lambda: conv.conv_wgrad(d, x, dy), P * (cy * 2 + Cin * 2))
for B in (32, 64, 128):
    layer("D.conv1 8->64 5x5 256^2", B, 256, 256, 8, 64, 5, 2, legs="w")
layer("D.conv1 8->64 5x5 256^2", 128, 256, 256, 8, 64, 5, 2, legs="f")
layer("D.conv1 8->64 5x5 256^2", 64, 256, 256, 8, 64, 5, 2, legs="d")
layer("G.conv_final 64->3 5x5 256x128", 64, 256, 128, 64, 3, 5, 1, f32=True)
layer("D.conv5 512->1 5x5 32^2", 128, 32, 32, 512, 1, 5, 2, f32=True)
layer("MeshD.conv4 256->1 5x5 8^2", 128, 8, 8, 256, 1, 5, 2, f32=True)
