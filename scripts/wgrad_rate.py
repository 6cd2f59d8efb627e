"""k_wgrad_halo on the layers that carry its time (batch 128 discriminator stride-2 convs, batch 64 generator 3x3): back to back,
time / rate / clock / power."""
import importlib, os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.2
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
    print(f"{name:44s} {conv.lib().m355_last_kernel().decode():14s} {us:8.1f} us {flops / us / 1e6:6.0f} TF  sclk {sum(a for a, _ in s) / max(len(s), 1):5.0f} MHz  power "
          f"{sum(b for _, b in s) / max(len(s), 1):5.0f} W", flush=True)
for name, (B, H, W, Cin, Cout, k, s_, p, mode, ups) in (("D.conv2 64->128 4x4 s2 B128", (128, 256, 256, 64, 128, 4, 2, 1, 2, 0)),
                                                     ("D.conv3 128->256 4x4 s2 B128", (128, 128, 128, 128, 256, 4, 2, 1, 2, 0)),
                                                     ("D.conv4 256->512 4x4 s2 B128", (128, 64, 64, 256, 512, 4, 2, 1, 2, 0)),
                                                     ("G.blk6.conv1 128->64 3x3 up B64", (64, 128, 64, 128, 64, 3, 1, 1, 1, 1)),
                                                     ("G.blk6.conv2 64->64 3x3 B64", (64, 256, 128, 64, 64, 3, 1, 1, 1, 0)),
                                                     ("G.blk5.conv2 128->128 3x3 B64", (64, 128, 64, 128, 128, 3, 1, 1, 1, 0))):
    d = conv.make_desc(B, H, W, Cin, Cout, k, k, s_, p, p, mode, ups)
    ho, wo = conv.out_hw(d)
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16(); dy = torch.randn(B, ho, wo, conv.dy_channels(Cout), device="cuda").bfloat16()
    probe(name + " wgrad", lambda: conv.conv_wgrad(d, x, dy), 2.0 * B * ho * wo * Cout * Cin * k * k)
