"""BASELINE configs[4] (stress): B clouds of 16384 points -> 512x512 projection fwd+bwd, and the Chamfer NN reduction
between two [B,16384,3] clouds; prints times, the SURVEY 8d algorithmic rates and size-independent checks."""
import importlib, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("2dimageto3dmodel_amd"); ops = importlib.import_module("2dimageto3dmodel_amd.ops")
dev = "cuda"
out = {}
for B in (1, 4, 8):
    N, S = 16384, 512
    g = torch.Generator().manual_seed(1234 + 5)
    pc = ((torch.rand(B, N, 3, generator=g) - 0.5) * 0.7).to(dev).requires_grad_()
    q = torch.randn(B, 4, generator=g).to(dev).requires_grad_()
    sc = torch.sigmoid(torch.randn(B, 1, generator=g)).to(dev).requires_grad_()
    mask = (torch.rand(B, 2 * S, 2 * S, generator=g) > 0.5).float().to(dev)
    elf = pkg.EffectiveLossFunction(voxel_size=S).to(dev); crit = pkg.SupervisedLoss()
    def step():
        pc.grad = q.grad = sc.grad = None
        loss = crit(elf(pc, q, sc), mask)["full_loss"]; loss.backward(); return loss
    for _ in range(2): l = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    proj = elf(pc, q, sc)
    assert torch.isfinite(proj).all() and proj.min() > 0 and proj.max() < 1 + 1e-5 and torch.isfinite(pc.grad).all()
    bytes_step = B * (20 * S ** 3 + 36 * N + 8 * S * S)
    out[f"proj_B{B}"] = {"ms": dt * 1e3, "clouds_per_s": B / dt, "algorithmic_GBps": bytes_step / dt / 1e9,
                         "frac_of_8TBps": bytes_step / dt / 8e12, "compulsory_io_MB": B * (36 * N + 8 * S * S + 40) / 1e6}
    a = ((torch.rand(B, N, 3, generator=g) - 0.5)).to(dev); b = ((torch.rand(B, N, 3, generator=g) - 0.5)).to(dev)
    f = getattr(ops, "chamfer_nn", None) or getattr(pkg, "chamfer_nn", None)
    if f is not None:
        for _ in range(2): d, i = f(a, b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): d, i = f(a, b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        ref = torch.cdist(a[:1, :512], b[:1]) ** 2
        assert torch.allclose(d[0, :512], ref.min(-1).values[0], rtol=1e-4, atol=1e-6)
        out[f"chamfer_B{B}"] = {"ms": dt * 1e3, "TFLOPs": 8.0 * B * N * N / dt / 1e12, "frac_of_157TF_fp32": 8.0 * B * N * N / dt / 157.3e12}
print(json.dumps(out, indent=1))
