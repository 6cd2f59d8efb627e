import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("2dimageto3dmodel_amd")
from test_proj_gpu import synth, t
for (B, N, S) in ((1, 16384, 512), (1, 4096, 256), (2, 2048, 128)):
    pc, q, sc, mask = synth(5120, B, N, S)
    outs = []
    for det in (False, True):
        prev = pkg.set_deterministic(det)
        try:
            tpc, tq, tsc = t(pc).requires_grad_(), t(q).requires_grad_(), t(sc).requires_grad_()
            proj = pkg.EffectiveLossFunction(voxel_size=S).to("cuda:0")(tpc, tq, tsc)
            pkg.SupervisedLoss()(proj, t(mask))["full_loss"].backward()
            outs.append((proj.detach().double(), tpc.grad.double(), tq.grad.double(), tsc.grad.double()))
        finally:
            pkg.set_deterministic(prev)
    p0, p1 = outs[0][0], outs[1][0]
    r = ((p1 - p0).abs() / p0.abs().clamp_min(1e-30))
    i = int(r.argmax())
    print(f"S {S} N {N}: silhouette det vs default: max rel {r.max().item():.3e} at flat index {i} (default {p0.flatten()[i].item():.6g}, det {p1.flatten()[i].item():.6g}); "
          f"pixels off by > 2e-5: {int((r > 2e-5).sum())} of {r.numel()}; empty value {p0.min().item():.6g}; non-empty pixels default {int((p0 > p0.min()*1.0001).sum())} det {int((p1 > p1.min()*1.0001).sum())}")
    for name, a, b in zip(("dpc", "dq", "dscale"), outs[0][1:], outs[1][1:]):
        print(f"     {name}: max abs diff {(a-b).abs().max().item():.3e} of max {a.abs().max().item():.3e}")
