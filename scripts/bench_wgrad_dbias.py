"""Cost of the fused bias gradient inside the weight-gradient kernels (discriminator layers, batch 128)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 128
for name, H, Cin, Cout, k, s, p in (("conv1", 256, 8, 64, 5, 1, 2), ("conv2", 256, 64, 128, 4, 2, 1), ("conv3", 128, 128, 256, 4, 2, 1), ("conv4", 64, 256, 512, 4, 2, 1)):
    d = conv.make_desc(B, H, H, Cin, Cout, k, k, s, p, p, 2, 0)
    ho, wo = conv.out_hw(d)
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    dy = torch.randn(B, ho, wo, Cout, device="cuda").bfloat16()
    db = torch.empty(Cout, device="cuda")
    for r in range(2):
        print("%s wgrad with dbias %.1f us   without %.1f us" % (name, timeit(lambda: conv.conv_wgrad(d, x, dy, dbias=db)), timeit(lambda: conv.conv_wgrad(d, x, dy))))
