"""Round 6 probe: per-launch constants of the main conv kernels -- time against the batch, intercept of the linear fit (what a launch costs
before its first image: ramp, tails, epilogue passes).  D layers at N = 16 .. 128, G layers at N = 8 .. 64."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")


def timeit(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


LAYERS = [  # name, H, W, Cin, Cout, k, s, ph, pw, mode, ups, batches
    ("D.conv1", 256, 256, 8, 64, 5, 1, 2, 2, 2, 0, (16, 32, 64, 128)),
    ("D.conv2", 256, 256, 64, 128, 4, 2, 1, 1, 2, 0, (16, 32, 64, 128)),
    ("D.conv3", 128, 128, 128, 256, 4, 2, 1, 1, 2, 0, (16, 32, 64, 128)),
    ("D.conv4", 64, 64, 256, 512, 4, 2, 1, 1, 2, 0, (16, 32, 64, 128)),
    ("G.blk6.conv2", 256, 128, 64, 64, 3, 1, 1, 1, 1, 0, (8, 16, 32, 64)),
    ("G.blk6.conv1", 128, 64, 128, 64, 3, 1, 1, 1, 1, 1, (8, 16, 32, 64)),
    ("G.blk5.conv2", 128, 64, 128, 128, 3, 1, 1, 1, 1, 0, (8, 16, 32, 64)),
    ("G.blk5.conv1", 64, 32, 128, 128, 3, 1, 1, 1, 1, 1, (8, 16, 32, 64)),
    ("G.blk4.conv2", 64, 32, 128, 128, 3, 1, 1, 1, 1, 0, (8, 16, 32, 64)),
]
for name, H, W, Cin, Cout, k, s_, ph, pw, mode, ups, Ns in LAYERS:
    rows = {"fwd": [], "dgrad": [], "wgrad": []}
    for N in Ns:
        d = conv.make_desc(N, H, W, Cin, Cout, k, k, s_, ph, pw, mode, ups)
        ho, wo = conv.out_hw(d)
        x = torch.randn(N, H, W, Cin, device="cuda").bfloat16()
        w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.02
        wf, wd = conv.weight_prep(d, w)
        dy = torch.randn(N, ho, wo, conv.dy_channels(Cout), device="cuda").bfloat16()
        if conv.conv_stats_rows(d):
            rows["fwd"].append(timeit(lambda: conv.conv_fwd_stats(d, x, wf)))
        else:
            rows["fwd"].append(timeit(lambda: conv.conv_fwd(d, x, wf, slope=0.2)))
        kf = conv.lib().m355_last_kernel().decode()
        rows["dgrad"].append(timeit(lambda: conv.conv_dgrad(d, dy, wd)))
        kd = conv.lib().m355_last_kernel().decode()
        rows["wgrad"].append(timeit(lambda: conv.conv_wgrad(d, x, dy)))
        kw_ = conv.lib().m355_last_kernel().decode()
    for what, kn in (("fwd", kf), ("dgrad", kd), ("wgrad", kw_)):
        b, a = np.polyfit(np.array(Ns, float), np.array(rows[what]), 1)
        print(f"{name:14s} {what:5s} {kn:16s} " + " ".join(f"N{n}:{t:.0f}" for n, t in zip(Ns, rows[what])) + f" us   fit: {a:6.1f} us + {b:.2f} us/image  (constant = {100 * a / rows[what][-1]:.0f} % of the largest launch)")
