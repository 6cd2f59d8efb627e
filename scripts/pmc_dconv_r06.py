"""Round 6: the discriminator's three big stride-2 layers (forward with bit masks, dgrad from bit masks, weight gradient) at the batch the
D steps run them (128) -- a few launches each for rocprofv3 --pmc passes (scripts/r06_runs/r06_run7.sh; summary: scripts/pmc_sq_mfma.py)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for name, H, Cin, Cout in (("D.conv2", 256, 64, 128), ("D.conv3", 128, 128, 256), ("D.conv4", 64, 256, 512)):
    d = conv.make_desc(N, H, H, Cin, Cout, 4, 4, 2, 1, 1, 2, 0)
    ho, wo = conv.out_hw(d)
    x = torch.randn(N, H, H, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.02
    b = torch.randn(Cout, device="cuda")
    wf, wd = conv.weight_prep(d, w)
    dy = torch.randn(N, ho, wo, conv.dy_channels(Cout), device="cuda").bfloat16()
    bits_in = torch.randint(-2 ** 31, 2 ** 31 - 1, (N, H, H, Cin // 64, 2), dtype=torch.int32, device="cuda") if conv.maskbits_ok(d, 1) else None
    for _ in range(3):
        if conv.maskbits_ok(d, 0):
            conv.conv_fwd(d, x, wf, b, slope=0.2, emit_bits=True)
        else:
            conv.conv_fwd(d, x, wf, b, slope=0.2)
        k_f = conv.lib().m355_last_kernel().decode()
        if bits_in is not None:
            conv.conv_dgrad(d, dy, wd, mask_bits=bits_in, mask_slope=0.2)
        else:
            conv.conv_dgrad(d, dy, wd)
        k_d = conv.lib().m355_last_kernel().decode()
        conv.conv_wgrad(d, x, dy)
        k_w = conv.lib().m355_last_kernel().decode()
    print(name, k_f, k_d, k_w, flush=True)
torch.cuda.synchronize()
