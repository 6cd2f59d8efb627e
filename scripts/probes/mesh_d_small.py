"""Round 6 probe: the mesh discriminator's layers (32x32 inputs) against the batch -- the batch-16 per-layer table (profiles/r06_layers_b16.txt)
shows their stride-2 dgrads and the 8x8 logit conv's weight gradient 3-5x SLOWER at N = 32 than at N = 128."""
import importlib, os, sys
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


LAYERS = [  # name, H, W, Cin, Cout, k, s, ph, pw, mode
    ("meshD.conv1", 32, 32, 16, 64, 5, 1, 2, 2, 2),
    ("meshD.conv2", 32, 32, 64, 128, 4, 2, 1, 1, 2),
    ("meshD.conv3", 16, 16, 128, 256, 4, 2, 1, 1, 2),
    ("meshD.conv4", 8, 8, 256, 1, 5, 1, 2, 2, 2),
    ("D.conv5", 32, 32, 512, 1, 5, 1, 2, 2, 2),
]
for name, H, W, Cin, Cout, k, s_, ph, pw, mode in LAYERS:
    out = {"fwd": [], "dgrad": [], "wgrad": []}
    kn = {}
    Ns = (16, 32, 64, 128)
    for N in Ns:
        d = conv.make_desc(N, H, W, Cin, Cout, k, k, s_, ph, pw, mode, 0)
        ho, wo = conv.out_hw(d)
        x = torch.randn(N, H, W, Cin, device="cuda").bfloat16()
        w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.02
        wf, wd = conv.weight_prep(d, w)
        dy = torch.randn(N, ho, wo, conv.dy_channels(Cout), device="cuda").bfloat16()
        out["fwd"].append(timeit(lambda: conv.conv_fwd(d, x, wf, slope=0.2))); kn["fwd"] = conv.lib().m355_last_kernel().decode()
        out["dgrad"].append(timeit(lambda: conv.conv_dgrad(d, dy, wd))); kn["dgrad"] = conv.lib().m355_last_kernel().decode()
        out["wgrad"].append(timeit(lambda: conv.conv_wgrad(d, x, dy))); kn["wgrad"] = conv.lib().m355_last_kernel().decode()
    for what in ("fwd", "dgrad", "wgrad"):
        print(f"{name:12s} {what:5s} {kn[what]:16s} " + " ".join(f"N{n}:{t:.0f}" for n, t in zip(Ns, out[what])) + " us")
