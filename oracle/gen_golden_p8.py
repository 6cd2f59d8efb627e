"""Generates tests/golden/p8_unsup.npz by EXECUTING THE REFERENCE'S UnsupervisedLoss (models/unsupervised_part.py:90-143).

    python -O oracle/gen_golden_p8.py          (build container only; needs /root/reference; -O = shim S0)

The eval branch (unsup:108-109) runs unmodified.  The training branch reads `self.num_candidates`, which the class never
sets (defect D8): shim S2 = that ATTRIBUTE is set on the instance to `number_of_pose_predictor_candidates` before the
call -- the reference's own code then runs every line of unsup:111-143.  Recorded: inputs, the three losses, the argmin
indices, and the autograd gradients with respect to the projections and the student poses.
"""
import contextlib
import importlib
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    rh.load()
    with contextlib.redirect_stdout(io.StringIO()):
        up = importlib.import_module("refpkg.models.unsupervised_part")
    rec = {}
    for tag, B, K, S, seed in (("a", 3, 4, 32, 801), ("b", 2, 2, 64, 802)):
        g = torch.Generator().manual_seed(seed)
        proj = torch.rand(B * K, S, S, generator=g).requires_grad_()
        masks = (torch.rand(B, 2 * S, 2 * S, generator=g) > 0.5).float()
        ens = torch.randn(B * K, 4, generator=g)
        stu = torch.randn(B, 4, generator=g).requires_grad_()
        with contextlib.redirect_stdout(io.StringIO()):
            L = up.UnsupervisedLoss(number_of_pose_predictor_candidates=K)
            ev = L.forward((proj[:B].detach(), ens[:B]), masks, False)
            L.num_candidates = K                                               # shim S2 (defect D8)
            tr = L.forward((proj, ens, stu), masks, True)
        tr["total_loss"].backward()
        rec.update({f"{tag}:B": B, f"{tag}:K": K, f"{tag}:S": S, f"{tag}:proj": proj.detach().numpy(),
                    f"{tag}:masks": masks.numpy().astype(np.uint8), f"{tag}:ens": ens.numpy(), f"{tag}:stu": stu.detach().numpy(),
                    f"{tag}:eval_loss": ev["projection_loss"].numpy(),
                    f"{tag}:projection_loss": tr["projection_loss"].detach().numpy(),
                    f"{tag}:student_loss": tr["student_loss"].detach().numpy(), f"{tag}:total_loss": tr["total_loss"].detach().numpy(),
                    f"{tag}:min_idx": L.minimum_indexes.numpy(), f"{tag}:dproj": proj.grad.numpy(), f"{tag}:dstu": stu.grad.numpy()})
        print(tag, float(ev["projection_loss"]), float(tr["total_loss"]), L.minimum_indexes.tolist())
    np.savez_compressed(os.path.join(OUT, "p8_unsup.npz"), **rec)
    # ---- PointsQuaternionsRotator.rotate_points (quaternions/points_quaternions.py:41-81), both directions, with autograd
    # gradients; QuaternionOperations addition / subtraction (operations.py:15-66); rendering.utils qrot / qmul (:36-64)
    m = rh.load()
    g = torch.Generator().manual_seed(811)
    xyz = ((torch.rand(3, 57, 3, generator=g) - 0.5) * 1.3).requires_grad_()
    q = torch.randn(3, 4, generator=g).requires_grad_()
    wts = torch.randn(3, 57, 3, generator=g)
    rot = {"xyz": xyz.detach().numpy(), "q": q.detach().numpy(), "w": wts.numpy()}
    with contextlib.redirect_stdout(io.StringIO()):
        for inv in (False, True):
            out = m["pq"].PointsQuaternionsRotator.rotate_points(xyz, q, inv)
            gx, gq = torch.autograd.grad((out * wts).sum(), (xyz, q))
            rot[f"out{int(inv)}"], rot[f"dxyz{int(inv)}"], rot[f"dq{int(inv)}"] = out.detach().numpy(), gx.numpy(), gq.numpy()
        qo = m["ops"].QuaternionOperations()
        a, b = torch.randn(5, 4, generator=g), torch.randn(5, 4, generator=g)
        rot.update(a=a.numpy(), b=b.numpy(), add=qo.quaternion_addition(a, b).numpy(), sub=qo.quaternion_subtraction(a, b).numpy(),
                   mul=qo.quaternion_multiplication(a, b).numpy(), conj=qo.quaternion_conjugate(a).numpy())
        sys.path.insert(0, rh.REF_ROOT)
        ru = importlib.import_module("refpkg.rendering.utils")
        v = torch.randn(5, 11, 3, generator=g)
        rot.update(v=v.numpy(), qrot=ru.qrot(a, v).numpy(), qmul=ru.qmul(a, b).numpy())
        grid = torch.rand(2, 6, 7, 2, generator=g) * 2 - 1
        img = torch.randn(2, 3, 9, 8, generator=g)
        rot.update(grid=grid.numpy(), img=img.numpy(), gsb=ru.grid_sample_bilinear(img, grid).numpy())
    np.savez_compressed(os.path.join(OUT, "p1_rotate.npz"), **rot)
    print("p1_rotate:", rot["out0"].shape)


if __name__ == "__main__":
    if __debug__:
        os.execv(sys.executable, [sys.executable, "-O"] + sys.argv)
    main()
