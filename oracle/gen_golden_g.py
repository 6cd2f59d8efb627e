"""Generates tests/golden/g_*.npz by EXECUTING THE REFERENCE'S OWN GAN CODE on CPU (fp32).

    python oracle/gen_golden_g.py        (build container only; needs /root/reference)

Recipe: sys.path = [/root/reference/code]; `models.gan`, `utils.losses` import unmodified;
norm_g='syncbatch' falls back to F.batch_norm on CPU (code/sync_batchnorm/batchnorm.py:70-73).
The 11.75 M-parameter state_dict is NOT stored: the drop-in modules create parameters in the reference's order
with the reference's initialisers, so `torch.manual_seed(seed)` reproduces the same weights on both sides; the
goldens keep the key/shape list, the outputs, the losses and per-parameter gradient norms of one G step and
one D step (code/main.py:491-520).
"""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import torch

REF = "/root/reference/code"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def make_args(**kw):
    a = argparse.Namespace(norm_g="syncbatch", norm_d="none", conditional_class=True, conditional_color=False,
                           conditional_text=False, n_classes=[200], texture_resolution=128, mask_output=True,
                           num_discriminators=2, texture_only=False, text_embedding_dim=256)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def make_inputs(seed, B, R, n_classes):
    g = torch.Generator().manual_seed(seed + 1)
    z = torch.randn(B, 64, generator=g)
    c = torch.randint(0, n_classes, (B, 1), generator=g)
    x_tex = torch.rand(B, 3, R, R, generator=g) * 2 - 1
    x_alpha = (torch.rand(B, 1, R, R, generator=g) > 0.4).float()
    x_mesh = 0.05 * torch.randn(B, 3, 32, 32, generator=g)
    return z, c, x_tex, x_alpha, x_mesh


CASES = [
    ("g_class128", 4321, 2, dict(texture_resolution=128)),
    ("g_class256_nobn", 4322, 2, dict(texture_resolution=256, norm_g="none")),
    ("g_uncond_circ", 4323, 2, dict(texture_resolution=128, conditional_class=False, norm_g="batch")),
]


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        from models.gan import Generator, MultiScaleDiscriminator
        from utils.losses import GANLoss
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    for name, seed, B, over in CASES:
        args = make_args(**over)
        symmetric = name != "g_uncond_circ"
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            G = Generator(args, 64, symmetric=symmetric, mesh_head=True)
            D = MultiScaleDiscriminator(args, 4)
        crit = GANLoss("hinge", tensor=torch.FloatTensor)
        R = 256 if args.texture_resolution >= 256 else 128
        z, c, x_tex, x_alpha, x_mesh = make_inputs(seed, B, R, 200)
        if not args.conditional_class:
            c = None
        G.train(); D.train()
        # ---- G step (main.py:491-498)
        pred_tex, pred_mesh = G(z, c)
        x_fake = torch.cat((pred_tex * x_alpha, x_alpha), dim=1)
        disc, mask = D(x_fake, pred_mesh, c)
        loss_g = crit(disc, True, for_discriminator=False, mask=mask, weight=None)
        loss_g.mean().backward()
        gnorm_G = {k: float(p.grad.norm()) for k, p in G.named_parameters() if p.grad is not None}
        for m in (G, D):
            m.zero_grad()
        # ---- D step (main.py:499-520), generator under no_grad (BN running stats update once more)
        with torch.no_grad():
            ft, fm = G(z, c)
            xf = torch.cat((ft * x_alpha, x_alpha), dim=1)
            xr = torch.cat((x_tex, x_alpha), dim=1)
            xc = torch.cat((xf, xr), dim=0)
            cc = torch.cat((c, c), dim=0) if c is not None else None
            mc = torch.cat((fm, x_mesh), dim=0)
        disc2, mask2 = D(xc, mc, cc)
        fake = [t[:B] for t in disc2]; real = [t[B:] for t in disc2]
        mfake = [t[:B] for t in mask2]; mreal = [t[B:] for t in mask2]
        loss_fake = crit(fake, False, for_discriminator=True, mask=mfake, weight=None)
        loss_real = crit(real, True, for_discriminator=True, mask=mreal, weight=None)
        (loss_fake + loss_real).mean().backward()
        gnorm_D = {k: float(p.grad.norm()) for k, p in D.named_parameters() if p.grad is not None}
        rec = dict(
            seed=seed, B=B, R=R, symmetric=symmetric,
            args=np.array(repr(vars(args))),
            g_keys=np.array(list(G.state_dict().keys())), g_shapes=np.array([str(tuple(v.shape)) for v in G.state_dict().values()]),
            d_keys=np.array(list(D.state_dict().keys())), d_shapes=np.array([str(tuple(v.shape)) for v in D.state_dict().values()]),
            pred_tex=pred_tex.detach().numpy().astype(np.float16), pred_mesh=pred_mesh.detach().numpy(),
            d1=disc[0].detach().numpy(), d2=disc[1].detach().numpy(),
            m1=mask[0].numpy(), m2=mask[1].numpy(),
            loss_g=loss_g.detach().numpy(), loss_fake=loss_fake.detach().numpy(), loss_real=loss_real.detach().numpy(),
            dd1=disc2[0].detach().numpy(), dd2=disc2[1].detach().numpy(),
            gnorm_G_keys=np.array(list(gnorm_G.keys())), gnorm_G=np.array(list(gnorm_G.values()), np.float32),
            gnorm_D_keys=np.array(list(gnorm_D.keys())), gnorm_D=np.array(list(gnorm_D.values()), np.float32),
            bn_mean_blk6=G.blk6.norm2.norm.running_mean.numpy() if hasattr(G.blk6.norm2.norm, "running_mean") else np.zeros(1),
        )
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **rec)
        print(f"{name}: loss_g={float(loss_g.mean()):.5f} loss_d={float((loss_fake+loss_real).mean()):.5f} "
              f"keys G/D={len(rec['g_keys'])}/{len(rec['d_keys'])} -> {os.path.getsize(path)/1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    main()
