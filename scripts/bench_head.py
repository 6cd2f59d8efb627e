"""A/B of the 64 -> 3 5x5 head forward (conv_final at batch 64, 256 x 128): k_head5 vs k_conv_smallco (M355_NO_HEAD5=1)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
N, H, W = 64, 256, 128
d = conv.make_desc(N, H, W, 64, 3, 5, 5, 1, 2, 2, 1, 0)
x = torch.randn(N, H, W, 64, device="cuda").bfloat16()
w = torch.randn(3, 64, 5, 5, device="cuda") * 0.02
b = torch.randn(3, device="cuda")
wf, _ = conv.weight_prep(d, w)
def run(reps=20):
    for _ in range(3): conv.conv_fwd(d, x, wf, b, out_f32_nchw=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): y = conv.conv_fwd(d, x, wf, b, out_f32_nchw=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, y
for rnd in range(2):
    os.environ.pop("M355_NO_HEAD5", None)
    t1, y1 = run()
    os.environ["M355_NO_HEAD5"] = "1"
    t0, y0 = run()
    gb = (x.numel() * 2 + y1.numel() * 4) / 1e9
    print("k_head5 %.1f us (%.2f TB/s)   k_conv_smallco %.1f us (%.2f TB/s)   max |diff| %.2e" %
          (t1, gb / t1 * 1e3 / 1e3 * 1e3, t0, gb / t0 * 1e3 / 1e3 * 1e3, (y1 - y0).abs().max().item()))
