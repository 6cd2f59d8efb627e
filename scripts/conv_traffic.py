"""A few launches of the discriminators' stride-2 convs (forward / dgrad / wgrad at batch 128) for a rocprofv3 --pmc FETCH_SIZE pass:
  cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o pmc -- python scripts/conv_traffic.py
  python scripts/rocpd_pmc.py /tmp/pmc_f/pmc_results.db out.csv"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
for name, (B, H, Cin, Cout) in (("D.conv2", (128, 256, 64, 128)), ("D.conv3", (128, 128, 128, 256)), ("D.conv4", (128, 64, 256, 512))):
    d = conv.make_desc(B, H, H, Cin, Cout, 4, 4, 2, 1, 1, 2, 0)
    ho, wo = conv.out_hw(d)
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16(); w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda"); wf, wd = conv.weight_prep(d, w); dy = torch.randn(B, ho, wo, Cout, device="cuda").bfloat16()
    for _ in range(3):
        conv.conv_fwd(d, x, wf, b, slope=0.2, emit_bits=True); conv.conv_dgrad(d, dy, wd); conv.conv_wgrad(d, x, dy)
    torch.cuda.synchronize()
    print(name, "x MB", x.numel() * 2 / 1e6, "y MB", dy.numel() * 2 / 1e6, flush=True)
