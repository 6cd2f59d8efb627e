"""The discriminator's stride-2 convs (with bias, LeakyReLU and sign bits, as TextureDiscriminator runs them) at batch 128."""
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
out = []
for name, H, Cin, Cout in (("conv2", 256, 64, 128), ("conv3", 128, 128, 256), ("conv4", 64, 256, 512)):
    d = conv.make_desc(B, H, H, Cin, Cout, 4, 4, 2, 1, 1, 2, 0)
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.02
    b = torch.randn(Cout, device="cuda")
    wf, wd = conv.weight_prep(d, w)
    bits_ok = conv.maskbits_ok(d, 0)
    t = timeit(lambda: conv.conv_fwd(d, x, wf, b, slope=0.2, emit_bits=bits_ok))
    t0 = timeit(lambda: conv.conv_fwd(d, x, wf, None, slope=0.2, emit_bits=bits_ok))
    fl = 2.0 * B * (H // 2) ** 2 * Cout * Cin * 16
    out.append("%s fwd+bias %.1f us (%.0f TF)  no bias %.1f us" % (name, t, fl / t / 1e6, t0))
print(os.environ.get("M355_LIB", "libm355.so"), " | ".join(out))
