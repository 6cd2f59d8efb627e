"""Runs the projection + silhouette loss forward/backward a few times at the bench shape (for rocprofv3 --pmc passes on
k_render21<..., false/true>):  rocprofv3 --kernel-trace --pmc <counters> -d DIR -o pmc -- python scripts/pmc_proj.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("2dimageto3dmodel_amd")
B, N, S = (int(v) for v in (sys.argv[1:4] + ["64", "2048", "128"][len(sys.argv) - 1:]))
g = torch.Generator().manual_seed(1)
pc = ((torch.rand(B, N, 3, generator=g) - 0.5) * 0.7).cuda().requires_grad_()
q = torch.randn(B, 4, generator=g).cuda().requires_grad_()
sc = torch.sigmoid(torch.randn(B, 1, generator=g)).cuda().requires_grad_()
mask = (torch.rand(B, 2 * S, 2 * S, generator=g) > 0.5).float().cuda()
elf = pkg.EffectiveLossFunction(voxel_size=S).cuda()
crit = pkg.SupervisedLoss()
for _ in range(4):
    pc.grad = q.grad = sc.grad = None
    crit(elf(pc, q, sc), mask)["full_loss"].backward()
torch.cuda.synchronize()
