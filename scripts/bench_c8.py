"""D.conv1 (8 -> 64, 5x5, circular) forward with bias + LeakyReLU + sign bits, and the 64 -> 3 head dgrad with the fused
activation backward -- the two users of k_conv_c8 -- at batch B (argv[1], default 128 / 64)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = conv.make_desc(B, 256, 256, 8, 64, 5, 5, 1, 2, 2, 2, 0)
x = torch.randn(B, 256, 256, 8, device="cuda").bfloat16()
w = torch.randn(64, 8, 5, 5, device="cuda") * 0.05
b = torch.randn(64, device="cuda")
wf, wd = conv.weight_prep(d, w)
dy = torch.randn(B, 256, 256, 64, device="cuda").bfloat16()
t1 = timeit(lambda: conv.conv_fwd(d, x, wf, b, slope=0.2, emit_bits=True))
t2 = timeit(lambda: conv.conv_wgrad(d, x, dy, dbias=torch.empty(64, device="cuda")))
h = conv.make_desc(64, 256, 128, 64, 3, 5, 5, 1, 2, 2, 1, 0)
xh = torch.randn(64, 256, 128, 64, device="cuda").bfloat16()
wh = torch.randn(3, 64, 5, 5, device="cuda") * 0.05
_, whd = conv.weight_prep(h, wh)
dyh = torch.randn(64, 256, 128, conv.dy_channels(3), device="cuda").bfloat16()
t3 = timeit(lambda: conv.conv_dgrad(h, dyh, whd, mask_x=xh, mask_slope=0.2)) if conv.dgrad_mask_ok(h) else float("nan")
print("%s  D.conv1 fwd+bits %.1f us   wgrad+dbias %.1f us   head dgrad+mask %.1f us" % (os.environ.get("M355_LIB", "libm355.so"), t1, t2, t3))
