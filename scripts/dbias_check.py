import importlib, sys, torch
sys.path.insert(0, "/root/repo")
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
for (N, H, W) in ((2, 16, 64), (3, 24, 32), (128, 256, 256)):
    d = conv.make_desc(N, H, W, 8, 64, 5, 5, 1, 2, 2, 2, 0)
    x = torch.randn(N, H, W, 8, device="cuda").bfloat16(); dy = torch.randn(N, H, W, 64, device="cuda").bfloat16()
    assert conv.wgrad_fuses_dbias(d)
    db = torch.zeros(64, device="cuda")
    dw = conv.conv_wgrad(d, x, dy, raw=True, dbias=db)
    print(conv.lib().m355_last_kernel().decode(), (N, H, W), "dbias max rel err", ((db - dy.float().sum((0, 1, 2))).abs().max() / dy.float().sum((0, 1, 2)).abs().max()).item())
