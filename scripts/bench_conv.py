"""Per-layer timing of the MFMA conv kernels at the GAN's shapes, B=16."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
LAYERS = [  # name, H, W (stored), Cin, Cout, k, stride, ph, pw, mode, ups
    ("G.blk4.conv1 256->128 3x3 up", 32, 16, 256, 128, 3, 1, 1, 1, 1, 1),
    ("G.blk5.conv1 128->128 3x3 up", 64, 32, 128, 128, 3, 1, 1, 1, 1, 1),
    ("G.blk5.conv2 128->128 3x3", 128, 64, 128, 128, 3, 1, 1, 1, 1, 0),
    ("G.blk6.conv1 128->64 3x3 up", 128, 64, 128, 64, 3, 1, 1, 1, 1, 1),
    ("G.blk6.conv2 64->64 3x3", 256, 128, 64, 64, 3, 1, 1, 1, 1, 0),
    ("G.conv_final 64->3 5x5", 256, 128, 64, 3, 5, 1, 2, 2, 1, 0),
    ("D.conv1 8->64 5x5", 256, 256, 8, 64, 5, 1, 2, 2, 2, 0),
    ("D.conv2 64->128 4x4 s2", 256, 256, 64, 128, 4, 2, 1, 1, 2, 0),
    ("D.conv3 128->256 4x4 s2", 128, 128, 128, 256, 4, 2, 1, 1, 2, 0),
    ("D.conv4 256->512 4x4 s2", 64, 64, 256, 512, 4, 2, 1, 1, 2, 0),
    ("D.conv5 512->1 5x5", 32, 32, 512, 1, 5, 1, 2, 2, 2, 0),
]
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
FILT = os.environ.get("M355_LAYERS")
for name, H, W, Cin, Cout, k, s, ph, pw, mode, ups in LAYERS:
    if FILT and not any(f in name for f in FILT.split(",")): continue
    d = conv.make_desc(B, H, W, Cin, Cout, k, k, s, ph, pw, mode, ups)
    ho, wo = conv.out_hw(d)
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    wf, wd = conv.weight_prep(d, w)
    dy = torch.randn(B, ho, wo, conv.dy_channels(Cout), device="cuda").bfloat16()
    fl = 2.0 * B * ho * wo * Cout * Cin * k * k
    tf = timeit(lambda: conv.conv_fwd(d, x, wf, out_f32_nchw=Cout <= 4))
    td = timeit(lambda: conv.conv_dgrad(d, dy, wd))
    tw = timeit(lambda: conv.conv_wgrad(d, x, dy))
    extra = ""
    if os.environ.get("M355_BENCH_MASK") and conv.maskbits_ok(d, 1):
        bits = torch.randint(-2**31, 2**31 - 1, (B, H, W, Cin // 64, 2), device="cuda", dtype=torch.int32)
        tdm = timeit(lambda: conv.conv_dgrad(d, dy, wd, mask_x=x, mask_slope=0.2))
        tdb = timeit(lambda: conv.conv_dgrad(d, dy, wd, mask_bits=bits, mask_slope=0.2))
        extra = f" | dgrad mask_x {tdm*1e6:8.1f} us  mask_bits {tdb*1e6:8.1f} us" 
    print(f"{name:32s} {fl/1e9:7.1f} GF  fwd {tf*1e6:8.1f} us {fl/tf/1e12:6.1f} TF | dgrad {td*1e6:8.1f} us {fl/td/1e12:6.1f} TF | wgrad {tw*1e6:8.1f} us {fl/tw/1e12:6.1f} TF" + extra, flush=True)
