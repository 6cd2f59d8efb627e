"""Run a few conv kernels a few times (for rocprofv3 --pmc diagnosis)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
B = 64
LAYERS = [("D.conv2", 256, 256, 64, 128, 4, 2, 1, 1, 2, 0), ("D.conv3", 128, 128, 128, 256, 4, 2, 1, 1, 2, 0),
          ("G.blk5.conv2", 128, 64, 128, 128, 3, 1, 1, 1, 1, 0), ("G.blk6.conv2", 256, 128, 64, 64, 3, 1, 1, 1, 1, 0)]
for name, H, W, Cin, Cout, k, s, ph, pw, mode, ups in LAYERS:
    d = conv.make_desc(B, H, W, Cin, Cout, k, k, s, ph, pw, mode, ups)
    ho, wo = conv.out_hw(d)
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    wf, wd = conv.weight_prep(d, w)
    dy = torch.randn(B, ho, wo, conv.dy_channels(Cout), device="cuda").bfloat16()
    for _ in range(3):
        conv.conv_fwd(d, x, wf); conv.conv_dgrad(d, dy, wd); conv.conv_wgrad(d, x, dy)
torch.cuda.synchronize()
