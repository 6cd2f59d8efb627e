"""EXACT build: every generator parameter gradient of one G step (golden g_class128: seed, inputs) against the reference run in
fp64 on CPU (scripts/probes/_g64_class128.pt, produced here from /root/reference by the snippet in DESIGN.md) -- per tensor."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gan_modules as tm
from conftest import load_golden
lib = importlib.import_module("2dimageto3dmodel_amd._lib")
lib.set_exact(os.environ.get("PROBE_EXACT", "1") == "1")
g = load_golden("g_class128")
gan, args, G, D = tm.build(g)
G.to("cuda:0").train(); D.to("cuda:0").train()
crit = gan.GANLoss("hinge")
B, R = int(g["B"]), int(g["R"])
z, c, x_tex, x_alpha, x_mesh = [t.to("cuda:0") for t in tm.make_inputs(int(g["seed"]), B, R, 200)]
pred_tex, pred_mesh = G(z, c, None)
x_fake = torch.cat((pred_tex * x_alpha, x_alpha), dim=1)
disc, mask = D(x_fake, pred_mesh, c, None)
loss = crit(disc, True, for_discriminator=False, mask=mask, weight=None)
loss.mean().backward()
ref = torch.load(os.path.join(ROOT, "scripts/probes/_g64_class128.pt"))
rows = []
for k, p in G.named_parameters():
    if p.grad is None or k not in ref or ref[k].norm() == 0:
        continue
    a, b = p.grad.detach().cpu().flatten().double(), ref[k].flatten().double()
    rows.append((float((a - b).norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm())), float(a.norm() / b.norm() - 1), k))
for r in rows:
    print("%-40s relL2 %.3e  cos-1 %.2e  norm ratio-1 %+.2e" % (r[3], r[0], r[1] - 1, r[2]))
