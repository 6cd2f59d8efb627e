"""Where a wave of k_conv_c8 spends a tile: s_memtime stamps of workgroup 0 (debug build).
Build:  M355_BUILD_LIB=libstampc8.so M355_BUILD_DEFS=-DM355_DBG_STAMP_C8 python 2dimageto3dmodel_amd/build.py
Run:    M355_LIB=libstampc8.so python scripts/stamp_c8.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
buf = torch.zeros(4 * 64 * 4, dtype=torch.int32, device="cuda")
os.environ["M355_STAMP_PTR"] = hex(buf.data_ptr())
B = 128
d = conv.make_desc(B, 256, 256, 8, 64, 5, 5, 1, 2, 2, 2, 0)
x = torch.randn(B, 256, 256, 8, device="cuda").bfloat16(); w = torch.randn(64, 8, 5, 5, device="cuda") * 0.05; b = torch.randn(64, device="cuda")
wf, wd = conv.weight_prep(d, w)
for name, f in (("fwd (bias, lrelu)", lambda: conv.conv_fwd(d, x, wf, b, slope=0.2)), ("fwd + sign bits", lambda: conv.conv_fwd(d, x, wf, b, slope=0.2, emit_bits=True))):
    for _ in range(3): f()
    torch.cuda.synchronize(); buf.zero_(); f(); torch.cuda.synchronize()
    a = buf.cpu().numpy().astype(np.uint32).astype(np.int64).reshape(4, 64, 4)
    n = int((a[0, :, 3] != 0).sum())
    dd = lambda p, q: (p - q) & 0xffffffff
    wait, mfma, epi = dd(a[:, :n, 1], a[:, :n, 0]), dd(a[:, :n, 2], a[:, :n, 1]), dd(a[:, :n, 3], a[:, :n, 2])
    tile = dd(a[:, 1:n, 0], a[:, :n - 1, 0])
    print(f"== k_conv_c8 {name}: {n} tiles stamped (of ~43 per workgroup); shader cycles per tile, median over tiles 4..")
    for wv in range(4):
        s = slice(4, n - 1)
        print(f"  wave {wv}: tile {np.median(tile[wv, s]):7.0f} = wait+barrier {np.median(wait[wv, s]):6.0f} + halo issue / 52 MFMA {np.median(mfma[wv, s]):6.0f} + epilogue {np.median(epi[wv, s]):6.0f}")
    print("  wave 0, tiles 4..11 (tile wait mfma epilogue):")
    for k in range(4, min(12, n - 1)): print(f"    {k:3d} {tile[0, k]:6d} {wait[0, k]:6d} {mfma[0, k]:6d} {epi[0, k]:6d}")
