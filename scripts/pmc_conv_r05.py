"""Round 5: a few launches of every conv kernel family of the FINAL tree at the benchmarked shapes (for rocprofv3 --pmc; the
summary goes to profiles/r05_pmc_sq_conv.txt via scripts/pmc_sq_mfma.py).  Shapes = the rows of profiles/r05_layers.txt."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
LAYERS = [  # name, N, H, W, Cin, Cout, k, s, ph, pw, mode, ups, what
    ("D.conv2", 128, 256, 256, 64, 128, 4, 2, 1, 1, 2, 0, "fdw"), ("D.conv3", 128, 128, 128, 128, 256, 4, 2, 1, 1, 2, 0, "fdw"),
    ("D.conv4", 128, 64, 64, 256, 512, 4, 2, 1, 1, 2, 0, "fdw"),
    ("G.blk6.conv1 (sub-pixel)", 64, 128, 64, 128, 64, 3, 1, 1, 1, 1, 1, "sdw"), ("G.blk5.conv1 (sub-pixel)", 64, 64, 32, 128, 128, 3, 1, 1, 1, 1, 1, "sdw"),
    ("G.blk6.conv2", 64, 256, 128, 64, 64, 3, 1, 1, 1, 1, 0, "sdw"), ("G.blk5.conv2", 64, 128, 64, 128, 128, 3, 1, 1, 1, 1, 0, "sdw"),
    ("G.blk4.conv1", 64, 32, 16, 256, 128, 3, 1, 1, 1, 1, 1, "sdw"),
    ("D.conv1", 128, 256, 256, 8, 64, 5, 1, 2, 2, 2, 0, "fw"),
    ("G.blk6.shortcut (glds)", 64, 128, 64, 128, 64, 1, 1, 0, 0, 0, 0, "fdw"), ("G.blk3.conv2 (glds)", 64, 32, 16, 256, 256, 3, 1, 1, 1, 1, 0, "fdw"),
    ("G.blk3.conv1 (glds)", 64, 16, 8, 256, 256, 3, 1, 1, 1, 1, 1, "fdw"), ("G.blk1.conv (glds split-K)", 64, 8, 4, 512, 512, 3, 1, 1, 1, 1, 0, "fdw"),
]
for name, N, H, W, Cin, Cout, k, s, ph, pw, mode, ups, what in LAYERS:
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, s, ph, pw, mode, ups)
    ho, wo = conv.out_hw(d)
    x = torch.randn(N, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    wf, wd = conv.weight_prep(d, w)
    dy = torch.randn(N, ho, wo, conv.dy_channels(Cout), device="cuda").bfloat16()
    for _ in range(3):
        if "s" in what and conv.conv_stats_rows(d):
            conv.conv_fwd_stats(d, x, wf)
        elif "s" in what and conv._fwd_ws(d)[0]:
            conv._fwd_splitk(d, x, wf, None, 1.0, True)
        else:
            conv.conv_fwd(d, x, wf, slope=0.2 if "f" in what else 1.0)
        if "d" in what:
            conv.conv_dgrad(d, dy, wd)
        conv.conv_wgrad(d, x, dy)
    print(name, conv.lib().m355_last_kernel().decode(), flush=True)
torch.cuda.synchronize()
