"""Run the 8-input-channel layer (TextureDiscriminator.conv1, batch 64) a few times: k_conv_c8 forward, k_wgrad_c8 -- for
rocprofv3 --pmc passes (scripts/pmc_sq_summary.py merges them)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
B, H, W, Cin, Cout, k = 64, 256, 256, 8, 64, 5
d = conv.make_desc(B, H, W, Cin, Cout, k, k, 1, 2, 2, 2, 0)
x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
b = torch.randn(Cout, device="cuda")
wf, wd = conv.weight_prep(d, w)
dy = torch.randn(B, H, W, Cout, device="cuda").bfloat16()
for _ in range(3):
    conv.conv_fwd(d, x, wf, b, slope=0.2, emit_bits=True)
    conv.conv_wgrad(d, x, dy)
torch.cuda.synchronize()
