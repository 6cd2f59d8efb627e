"""Round 6 probe: the per-launch constant of the stride-2 class weight gradients (k_wgrad_halo): time against the batch for D.conv2 / 3 / 4;
the intercept of the linear fit is what a launch costs before its first pixel (ramp, split-K atomics tail)."""
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


for name, H, Cin, Cout in (("conv2", 256, 64, 128), ("conv3", 128, 128, 256), ("conv4", 64, 256, 512)):
    ts, Ns = [], (4, 8, 16, 32, 64, 128)
    for N in Ns:
        d = conv.make_desc(N, H, H, Cin, Cout, 4, 4, 2, 1, 1, 2, 0)
        x = torch.randn(N, H, H, Cin, device="cuda").bfloat16()
        dy = torch.randn(N, H // 2, H // 2, Cout, device="cuda").bfloat16()
        ts.append(timeit(lambda: conv.conv_wgrad(d, x, dy)))
    b, a = np.polyfit(np.array(Ns[2:], float), np.array(ts[2:]), 1)
    print(name, conv.lib().m355_last_kernel().decode(), " ".join(f"N{n}:{t:.0f}us" for n, t in zip(Ns, ts)), f" fit over N>=16: {a:.0f} us + {b:.2f} us/image")
