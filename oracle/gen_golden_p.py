"""Generates tests/golden/p_*.npz by EXECUTING THE REFERENCE'S OWN CODE on CPU.

Run in the build container only (needs /root/reference):

    python -O oracle/gen_golden_p.py            # -O = shim S0 (reference asserts batch == 3)

What is recorded per case (all from reference functions, see oracle/ref_harness.py):
inputs, camera coords (P2), bins + in-bounds mask (tri:24,45-46), the clamped occupancy volume in
sparse form (P3), per-ray depth-sums of the smoothed volume and the smoothed volume at sampled rays (P4),
termination probabilities at sampled rays (P5), the silhouette (P6), SupervisedLoss (P7) and the
autograd gradients w.r.t. point_cloud / rotation / scale.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name, seed, B, N, S, cloud spread, has_scale, special
CASES = [
    ("p_cfg1", 1235, 1, 1024, 64, 0.7, True, None),          # BASELINE.json configs[0]
    ("p_b3_noscale", 1301, 3, 300, 32, 0.7, False, None),
    ("p_oob", 1302, 2, 500, 64, 1.6, True, None),            # many points outside the +-0.5 cube
    ("p_s128", 1236, 2, 2048, 128, 0.7, True, None),         # configs[1] shape, small batch
    ("p_n1", 1303, 2, 1, 32, 0.5, True, None),               # single point
    ("p_dense", 1304, 1, 4096, 32, 0.9, True, None),         # heavy voxel collisions
    ("p_sigma", 1305, 2, 256, 64, 0.7, True, "sigma1.5"),    # annealed sigma (training_test_shape_net.py:29)
]


def make_inputs(seed, B, N, S, spread, has_scale):
    g = torch.Generator().manual_seed(seed)
    pc = (torch.rand(B, N, 3, generator=g) - 0.5) * spread
    q = torch.randn(B, 4, generator=g)
    sc = torch.sigmoid(torch.randn(B, 1, generator=g)) if has_scale else None
    mask = (torch.rand(B, 2 * S, 2 * S, generator=g) > 0.5).float()
    return pc, q, sc, mask


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    for name, seed, B, N, S, spread, has_scale, special in CASES:
        sigma = 1.5 if special == "sigma1.5" else 3.0
        pc, q, sc, mask = make_inputs(seed, B, N, S, spread, has_scale)
        pc.requires_grad_()
        q.requires_grad_()
        if sc is not None:
            sc.requires_grad_()
        st = rh.ref_forward_stages(pc, q, sc, S=S, sigma=sigma)
        proj = st["proj"]
        # for B == 1 the reference's masks.squeeze() also drops the batch dim (defect D13); mse_loss then
        # broadcasts [S,S] against [1,S,S], the value is unaffected
        loss = rh.ref_supervised_loss(proj, mask)
        dproj = torch.autograd.grad(loss, proj, retain_graph=True)[0]
        loss.backward()
        cam = st["cam"].detach()
        m = rh.load()
        interp = rh.quiet(m["tri"].TrilinearInterpolation, size=S)
        inb = interp.get_point_cloud_object_borders(cam).view(B, N)
        fl = interp.get_grid(cam, cam.new(3).fill_(S)).floor().long()
        vox = st["voxels"].detach()
        nz = vox.reshape(-1).nonzero().squeeze(1)
        sm = st["smoothed"].detach()
        probs = st["probs"].detach()
        rg = np.random.RandomState(seed)
        nr = min(48, S * S)
        rays = rg.choice(S * S, nr, replace=False)
        ry, rx = rays // S, rays % S
        rec = dict(
            B=B, N=N, S=S, sigma=np.float32(sigma),
            pc=pc.detach().numpy(), q=q.detach().numpy(), mask=mask.numpy().astype(np.uint8),
            cam=cam.numpy(), inb=inb.numpy(), floor=fl.numpy().astype(np.int32),
            vox_idx=nz.numpy().astype(np.int64), vox_val=vox.reshape(-1)[nz].numpy(),
            taps=st["kernels"][2].reshape(-1).numpy(),
            sm_raysum=sm.sum(1).numpy(), ray_y=ry.astype(np.int32), ray_x=rx.astype(np.int32),
            sm_rays=sm[:, :, ry, rx].numpy(), probs_rays=probs[:, :, ry, rx].numpy(),
            proj=proj.detach().numpy(), loss=np.float64(loss.item()), dproj=dproj.numpy(),
            dpc=pc.grad.numpy(), dq=q.grad.numpy(),
        )
        if sc is not None:
            rec["scale"] = sc.detach().numpy()
            rec["dscale"] = sc.grad.numpy()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **rec)
        print(f"{name}: B={B} N={N} S={S} nnz={nz.numel()} loss={loss.item():.6f} -> {os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    if __debug__:
        os.execv(sys.executable, [sys.executable, "-O"] + sys.argv)
    main()
