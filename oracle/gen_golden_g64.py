"""Goldens of the EXACT build's module-level tests: the reference's models/gan.py + utils/losses.py executed in FLOAT64 on CPU.

Why fp64: the fp32 goldens of gen_golden_g.py carry the reference's OWN fp32 rounding noise, and at batch 2 the backward pass is
ill-conditioned enough for that noise to reach 1e-4 .. 2e-3 relative L2 on whole gradient tensors (measured here: reference fp32
against reference fp64, same weights and inputs -- printed below per case).  The EXACT build of the library (fp32 activations,
fp64 accumulation in the convolutions and in every reduction; include/m355.h m355_act_bytes) agrees with the fp64 run to ~1e-6,
i.e. it can be held much tighter than the fp32 goldens can hold anything.  Same seeds / inputs / cases as gen_golden_g.py
(a subset: the files carry full gradient tensors in fp32).

    python oracle/gen_golden_g64.py        -> tests/golden/g64_<case>.npz

Test infrastructure only.  Reads /root/reference (this container); the .npz files travel with the repository."""
import contextlib
import copy
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_g as gg  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
CASES64 = ["g_class128", "g_class256_sync", "g_uncond_circ"]   # (also run once for g_nomask128 / g_inst_color128: noise 1.6e-3 / 4.5e-4; not kept)


def run_case64(name, seed, B, over, ref_gan, GANLoss):
    args = gg.make_args(**over)
    symmetric = name != "g_uncond_circ"
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        G = ref_gan.Generator(args, 64, symmetric=symmetric, mesh_head=True)   # fp32 initialisation: the weights of the fp32 goldens
        D = ref_gan.MultiScaleDiscriminator(args, 4)
    G32, D32 = copy.deepcopy(G), copy.deepcopy(D)
    G, D = G.double(), D.double()
    R = args.texture_resolution
    z, c, x_tex, x_alpha, x_mesh = gg.make_inputs(seed, B, R, 200)
    c2, caption = gg.make_extra_inputs(seed, B, args)
    if args.conditional_color:
        c = torch.cat((c, c2), dim=1)
    if not args.conditional_class:
        c = None
    caption = None
    w = gg.d_weight(args)

    def steps(G, D, dt):
        crit = GANLoss("hinge", tensor=torch.DoubleTensor if dt == torch.float64 else torch.FloatTensor)
        zz, xt, xa, xm = z.to(dt), x_tex.to(dt), x_alpha.to(dt), x_mesh.to(dt)
        G.train(); D.train()
        pred_tex, pred_mesh = G(zz, c, caption)
        disc, mask = D(torch.cat((pred_tex * xa, xa), dim=1), pred_mesh, c, caption)
        loss_g = crit(disc, True, for_discriminator=False, mask=mask if args.mask_output else None, weight=w)
        loss_g.mean().backward()
        gG = {k: p.grad.detach().clone() for k, p in G.named_parameters() if p.grad is not None}
        for m in (G, D):
            m.zero_grad()
        with torch.no_grad():
            ft, fm = G(zz, c, caption)
            xc = torch.cat((torch.cat((ft * xa, xa), dim=1), torch.cat((xt, xa), dim=1)), dim=0)
            cc = torch.cat((c, c), dim=0) if c is not None else None
            mc = torch.cat((fm, xm), dim=0)
        disc2, mask2 = D(xc, mc, cc, None)
        fake, real = [t[:B] for t in disc2], [t[B:] for t in disc2]
        mfake = [t[:B] for t in mask2] if args.mask_output else None
        mreal = [t[B:] for t in mask2] if args.mask_output else None
        loss_fake = crit(fake, False, for_discriminator=True, mask=mfake, weight=w)
        loss_real = crit(real, True, for_discriminator=True, mask=mreal, weight=w)
        (loss_fake + loss_real).mean().backward()
        gD = {k: p.grad.detach().clone() for k, p in D.named_parameters() if p.grad is not None}
        return dict(pred_tex=pred_tex.detach(), disc=[t.detach() for t in disc], disc2=[t.detach() for t in disc2],
                    loss_g=loss_g.detach(), loss_fake=loss_fake.detach(), loss_real=loss_real.detach(), gG=gG, gD=gD)

    r64 = steps(G, D, torch.float64)
    r32 = steps(G32, D32, torch.float32)   # the reference's own fp32 noise, for the record
    noise = {}
    for kind in ("gG", "gD"):
        v = [float((r32[kind][k].double() - r64[kind][k]).norm() / r64[kind][k].norm()) for k in r64[kind] if r64[kind][k].norm() > 0]
        noise[kind] = (float(np.median(v)), float(np.max(v)))
    rec = dict(seed=seed, B=B, R=R, ref_fp32_noise=np.array([noise["gG"], noise["gD"]]),
               loss_g=r64["loss_g"].numpy(), loss_fake=r64["loss_fake"].numpy(), loss_real=r64["loss_real"].numpy(),
               gnorm_G_keys=np.array(list(r64["gG"].keys())), gnorm_G=np.array([float(v.norm()) for v in r64["gG"].values()]),
               gnorm_D_keys=np.array(list(r64["gD"].keys())), gnorm_D=np.array([float(v.norm()) for v in r64["gD"].values()]))
    ts = 2 if R >= 512 else 1
    rec["tex_stride"], rec["pred_tex"] = ts, r64["pred_tex"].numpy()[:, :, ::ts, ::ts].astype(np.float32)
    for i, (a, b) in enumerate(zip(r64["disc"], r64["disc2"])):
        rec[f"d{i + 1}"], rec[f"dd{i + 1}"] = a.numpy(), b.numpy()
    for k in gg.GRAD_KEYS_G:
        if k in r64["gG"]:
            rec["gradG:" + k] = r64["gG"][k].numpy().astype(np.float32)
    for k in gg.GRAD_KEYS_D:
        if k in r64["gD"]:
            rec["gradD:" + k] = r64["gD"][k].numpy().astype(np.float32)
    path = os.path.join(OUT, "g64_" + name[2:] + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: reference fp32 vs fp64 gradient rel-L2 (median, max): G {noise['gG'][0]:.2e} {noise['gG'][1]:.2e}  "
          f"D {noise['gD'][0]:.2e} {noise['gD'][1]:.2e}  -> {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    ref_gan, GANLoss = gg.import_reference()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    for name, seed, B, over in gg.CASES:
        if name in CASES64:
            run_case64(name, seed, B, over, ref_gan, GANLoss)
