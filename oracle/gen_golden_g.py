"""Generates tests/golden/g_*.npz by EXECUTING THE REFERENCE'S OWN GAN CODE on CPU (fp32).

    python oracle/gen_golden_g.py        (build container only; needs /root/reference)

Recipe: sys.path = [/root/reference/code]; `models.gan`, `utils.losses` import unmodified;
norm_g='syncbatch' falls back to F.batch_norm on CPU (code/sync_batchnorm/batchnorm.py:70-73).
The 11.75 M-parameter state_dict is NOT stored: the drop-in modules create parameters in the reference's order
with the reference's initialisers, so `torch.manual_seed(seed)` reproduces the same weights on both sides; the
goldens keep the key/shape list, the outputs, the losses and per-parameter gradient norms of one G step and
one D step (code/main.py:491-520), plus a handful of FULL gradient tensors (fp16) per case for elementwise checks.

`g_train4` is a 4-iteration run of the training loop (G, D, D, G) of code/main.py:691-723 with the reference's modules,
torch.optim.Adam(betas=(0, 0.9)) as main.py:588-589 and update_generator_running_avg (main.py:431-447, restated here
because main.py parses sys.argv and opens datasets at import time); `ganloss` holds utils/losses.py:GANLoss outputs for
the four modes on fixed logits.
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


def make_extra_inputs(seed, B, args):
    """second class column (conditional_color) and the (words_emb, words_mask) pair a text encoder would hand over
    (main.py:480-484); drawn from their own stream so the five tensors above do not move"""
    g = torch.Generator().manual_seed(seed + 2)
    c2 = torch.randint(0, 10, (B, 1), generator=g)
    L = 12
    words = torch.randn(B, args.text_embedding_dim, L, generator=g)
    wmask = torch.zeros(B, L, dtype=torch.bool)
    wmask[:, 9:] = True            # padded tail of every caption
    return c2, (words, wmask)


def d_weight(args):
    """main.py:486-489"""
    return [2, 1] if args.num_discriminators == 2 and args.texture_resolution >= 512 else None


# full gradient tensors kept per case (when the parameter exists and has a gradient)
GRAD_KEYS_G = ["blk6.conv2.weight_orig", "blk1.norm1.fc_gamma.weight", "emb_class.weight", "blk4.shortcut.weight_orig",
               "conv_final.weight", "blk5.conv1.weight_orig", "blk3b.norm1.fc_gamma.weight", "att.conv_context.weight"]
GRAD_KEYS_D = ["d1.conv2.weight_orig", "d1.conv1.weight_orig", "d2.conv2.weight_orig", "d1.projector.weight", "d1.conv5.weight_orig",
               "d3.conv2.weight_orig", "d1.bn2.weight", "d2.conv1.weight_orig", "d1.conv4.bias", "d1.conv1.bias"]


CASES = [
    ("g_class128", 4321, 2, dict(texture_resolution=128)),
    ("g_class256_nobn", 4322, 2, dict(texture_resolution=256, norm_g="none")),
    ("g_uncond_circ", 4323, 2, dict(texture_resolution=128, conditional_class=False, norm_g="batch")),
    # round 2: the benchmarked network itself, the reference's default scale (README.md:92: 512^2, 3 discriminators), the
    # [2, 1] discriminator weights of main.py:486-489, and the option branches (instance norms, class + colour, text)
    ("g_class256_sync", 4324, 2, dict(texture_resolution=256)),
    ("g_class512_nd3", 4325, 2, dict(texture_resolution=512, num_discriminators=3)),
    ("g_class512_nd2w", 4326, 2, dict(texture_resolution=512)),
    ("g_inst_color128", 4327, 2, dict(texture_resolution=128, norm_g="instance", norm_d="instance", conditional_color=True,
                                      n_classes=[200, 10])),
    ("g_text128", 4328, 2, dict(texture_resolution=128, conditional_class=False, conditional_text=True, norm_g="batch")),
    ("g_nomask128", 4329, 2, dict(texture_resolution=128, mask_output=False, norm_g="batch")),
]


def import_reference():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import models.gan as ref_gan
        from utils.losses import GANLoss
    return ref_gan, GANLoss


def _grads(module, keys):
    named = dict(module.named_parameters())
    out = {}
    for k in keys:
        p = named.get(k)
        if p is not None and p.grad is not None:
            out[k] = p.grad.detach().numpy().astype(np.float16 if p.grad.abs().max() < 6e4 else np.float32)
    return out


def run_case(name, seed, B, over, ref_gan, GANLoss):
    args = make_args(**over)
    symmetric = name != "g_uncond_circ"
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        G = ref_gan.Generator(args, 64, symmetric=symmetric, mesh_head=True)
        D = ref_gan.MultiScaleDiscriminator(args, 4)
    crit = GANLoss("hinge", tensor=torch.FloatTensor)
    R = args.texture_resolution
    z, c, x_tex, x_alpha, x_mesh = make_inputs(seed, B, R, 200)
    c2, caption = make_extra_inputs(seed, B, args)
    if args.conditional_color:
        c = torch.cat((c, c2), dim=1)
    if not args.conditional_class:
        c = None
    if not args.conditional_text:
        caption = None
    w = d_weight(args)
    # ---- inference mode first (ModelWrapper.forward('inference'), main.py:521-525, under trainer.eval()): running
    # statistics, no power iteration -- leaves every buffer untouched, so the training goldens below do not move
    eval_rec = {}
    if name in ("g_class128", "g_text128", "g_uncond_circ"):
        G.eval()
        with torch.no_grad():
            te, me, att = G(z, c, caption, return_attention=True)
        eval_rec = dict(eval_tex=te.numpy().astype(np.float16), eval_mesh=me.numpy())
        if att is not None:
            eval_rec["eval_att"] = att.numpy()
    G.train(); D.train()
    # ---- G step (main.py:491-498)
    pred_tex, pred_mesh = G(z, c, caption)
    x_fake = torch.cat((pred_tex * x_alpha, x_alpha), dim=1)
    disc, mask = D(x_fake, pred_mesh, c, caption)
    loss_g = crit(disc, True, for_discriminator=False, mask=mask if args.mask_output else None, weight=w)
    loss_g.mean().backward()
    gnorm_G = {k: float(p.grad.norm()) for k, p in G.named_parameters() if p.grad is not None}
    grads_G = _grads(G, GRAD_KEYS_G)
    for m in (G, D):
        m.zero_grad()
    # ---- D step (main.py:499-520), generator under no_grad (BN running stats update once more)
    with torch.no_grad():
        ft, fm = G(z, c, caption)
        xf = torch.cat((ft * x_alpha, x_alpha), dim=1)
        xr = torch.cat((x_tex, x_alpha), dim=1)
        xc = torch.cat((xf, xr), dim=0)
        cc = torch.cat((c, c), dim=0) if c is not None else None
        capc = [torch.cat((t, t), dim=0) for t in caption] if caption is not None else None
        mc = torch.cat((fm, x_mesh), dim=0)
    disc2, mask2 = D(xc, mc, cc, capc)
    fake = [t[:B] for t in disc2]; real = [t[B:] for t in disc2]
    if args.mask_output:
        mfake = [t[:B] for t in mask2]; mreal = [t[B:] for t in mask2]
    else:
        mfake = mreal = None
    loss_fake = crit(fake, False, for_discriminator=True, mask=mfake, weight=w)
    loss_real = crit(real, True, for_discriminator=True, mask=mreal, weight=w)
    (loss_fake + loss_real).mean().backward()
    gnorm_D = {k: float(p.grad.norm()) for k, p in D.named_parameters() if p.grad is not None}
    grads_D = _grads(D, GRAD_KEYS_D)
    ts = 2 if R >= 512 else 1
    rec = dict(
        seed=seed, B=B, R=R, symmetric=symmetric,
        args=np.array(repr(vars(args))),
        g_keys=np.array(list(G.state_dict().keys())), g_shapes=np.array([str(tuple(v.shape)) for v in G.state_dict().values()]),
        d_keys=np.array(list(D.state_dict().keys())), d_shapes=np.array([str(tuple(v.shape)) for v in D.state_dict().values()]),
        # (512^2 textures are kept at every other pixel: 3 MB -> 0.75 MB per case)
        tex_stride=ts, pred_tex=pred_tex.detach().numpy()[:, :, ::ts, ::ts].astype(np.float16), pred_mesh=pred_mesh.detach().numpy(),
        loss_g=loss_g.detach().numpy(), loss_fake=loss_fake.detach().numpy(), loss_real=loss_real.detach().numpy(),
        gnorm_G_keys=np.array(list(gnorm_G.keys())), gnorm_G=np.array(list(gnorm_G.values()), np.float32),
        gnorm_D_keys=np.array(list(gnorm_D.keys())), gnorm_D=np.array(list(gnorm_D.values()), np.float32),
        bn_mean_blk6=(G.blk6.norm2.norm.running_mean.numpy() if getattr(G.blk6.norm2.norm, "running_mean", None) is not None
                      else np.zeros(1)),
    )
    rec.update(eval_rec)
    for i, t in enumerate(disc):
        rec[f"d{i + 1}"] = t.detach().numpy()
        rec[f"dd{i + 1}"] = disc2[i].detach().numpy()
        if args.mask_output:
            rec[f"m{i + 1}"] = mask[i].numpy()
    for k, v in grads_G.items():
        rec["gradG:" + k] = v
    for k, v in grads_D.items():
        rec["gradD:" + k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: loss_g={float(loss_g.mean()):.5f} loss_d={float((loss_fake+loss_real).mean()):.5f} "
          f"keys G/D={len(rec['g_keys'])}/{len(rec['d_keys'])} grads {len(grads_G)}+{len(grads_D)} "
          f"-> {os.path.getsize(path)/1024:.0f} KiB", flush=True)


def run_train4(ref_gan, GANLoss, name="g_train4", seed=4401, B=2, iters=4, epoch=0):
    """the loop body of main.py:691-723 (without the mesh template term: Kaolin-free run), 1 G step : 2 D steps"""
    import math
    args = make_args(texture_resolution=128)
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        # construction order of main.py:541-545: the discriminator is an ARGUMENT of ModelWrapper(...) (built first), then
        # ModelWrapper.__init__ instantiates the generator and the running-average generator (main.py:453-457)
        D = ref_gan.MultiScaleDiscriminator(args, 4)
        G = ref_gan.Generator(args, 64, symmetric=True, mesh_head=True)
        G_avg = ref_gan.Generator(args, 64, symmetric=True, mesh_head=True)
        G_avg.load_state_dict(G.state_dict())
        for p in G_avg.parameters():
            p.requires_grad = False
    crit = GANLoss("hinge", tensor=torch.FloatTensor)
    opt_g = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.0, 0.9))            # main.py:588-589, defaults :109-110
    opt_d = torch.optim.Adam(D.parameters(), lr=4e-4, betas=(0.0, 0.9))
    alpha0 = 0.999                                                                 # --g_running_average_alpha default

    def update_avg(epoch):                                                         # main.py:431-447
        with torch.no_grad():
            a = math.pow(alpha0, 100) if epoch < 10 else (math.pow(alpha0, 10) if epoch < 100 else alpha0)
            sd = G.state_dict()
            for k, p in G_avg.state_dict().items():
                if torch.is_floating_point(p):
                    p.mul_(a).add_(sd[k], alpha=1 - a)
                else:
                    p.fill_(sd[k])

    track_g = ["blk6.conv2.weight_orig", "blk1.norm1.fc_beta.bias", "conv_mesh.weight"]
    track_d = ["d1.conv2.weight_orig", "d2.conv3.bias"]
    w0 = {k: dict(G.named_parameters())[k].detach().clone() for k in track_g}
    w0.update({k: dict(D.named_parameters())[k].detach().clone() for k in track_d})
    G.train(); D.train(); G_avg.train()
    losses = []
    rec = dict(seed=seed, B=B, R=128, iters=iters, epoch=epoch, args=np.array(repr(vars(args))))
    for it in range(iters):
        z, c, x_tex, x_alpha, x_mesh = make_inputs(seed + 10 * it, B, 128, 200)
        if it % 3 == 0:
            opt_g.zero_grad()
            pred_tex, pred_mesh = G(z, c)
            disc, mask = D(torch.cat((pred_tex * x_alpha, x_alpha), dim=1), pred_mesh, c)
            loss = crit(disc, True, for_discriminator=False, mask=mask, weight=None).mean()
            loss.backward()
            opt_g.step()
            update_avg(epoch)
            losses.append([float(loss), 0.0])
        else:
            opt_d.zero_grad()
            with torch.no_grad():
                ft, fm = G(z, c)
                xc = torch.cat((torch.cat((ft * x_alpha, x_alpha), 1), torch.cat((x_tex, x_alpha), 1)), 0)
                cc, mc = torch.cat((c, c), 0), torch.cat((fm, x_mesh), 0)
            disc, mask = D(xc, mc, cc)
            lf = crit([t[:B] for t in disc], False, for_discriminator=True, mask=[t[:B] for t in mask], weight=None).mean()
            lr_ = crit([t[B:] for t in disc], True, for_discriminator=True, mask=[t[B:] for t in mask], weight=None).mean()
            (lf + lr_).backward()
            opt_d.step()
            losses.append([float(lf), float(lr_)])
        if it in (0, iters - 1):
            gp, dp, ap = dict(G.named_parameters()), dict(D.named_parameters()), dict(G_avg.named_parameters())
            for k in track_g:
                rec[f"it{it}:G:{k}"] = (gp[k].detach() - w0[k]).numpy()
                rec[f"it{it}:avg:{k}"] = (ap[k].detach() - w0[k]).numpy()
            for k in track_d:
                rec[f"it{it}:D:{k}"] = (dp[k].detach() - w0[k]).numpy()
            if it == iters - 1:
                # Adam's second-moment estimates (smooth in the gradients, unlike the sign-step deltas above)
                rec[f"it{it}:G:exp_avg_sq:blk6.conv2.weight_orig"] = opt_g.state[gp["blk6.conv2.weight_orig"]]["exp_avg_sq"].numpy().copy()
                rec[f"it{it}:D:exp_avg_sq:d1.conv2.weight_orig"] = opt_d.state[dp["d1.conv2.weight_orig"]]["exp_avg_sq"].numpy().copy()
            rec[f"it{it}:avg_bn_mean"] = G_avg.blk6.norm2.norm.running_mean.numpy().copy()
            rec[f"it{it}:avg_nbt"] = int(G_avg.blk6.norm2.norm.num_batches_tracked)
    rec["losses"] = np.array(losses, np.float32)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: losses {losses} -> {os.path.getsize(path)/1024:.0f} KiB", flush=True)


def run_ganloss(GANLoss):
    """utils/losses.py:21-120 on fixed logits: every mode x (real, fake) x (D, G) x (masked, unmasked, weighted)"""
    g = torch.Generator().manual_seed(99)
    preds = [torch.randn(3, 1, 16, 8, generator=g) * 2, torch.randn(3, 1, 8, 8, generator=g) * 2]
    masks = [torch.rand(3, 1, 16, 8, generator=g), torch.rand(3, 1, 8, 8, generator=g)]
    rec = dict(p0=preds[0].numpy(), p1=preds[1].numpy(), m0=masks[0].numpy(), m1=masks[1].numpy())
    for mode in ("hinge", "ls", "original", "w"):
        crit = GANLoss(mode, tensor=torch.FloatTensor)
        for real in (True, False):
            for ford in (True, False):
                if mode == "hinge" and not ford and not real:
                    continue   # asserts in the reference
                for mk, wk in ((None, None), (masks, None), (masks, [2, 1])):
                    if mode != "hinge" and mk is not None:
                        continue   # only the hinge branch takes masks / weights
                    v = crit(preds, real, for_discriminator=ford, mask=mk, weight=wk)
                    rec[f"{mode}:{int(real)}{int(ford)}:{'m' if mk is not None else '-'}{'w' if wk else '-'}"] = v.detach().numpy()
                rec[f"{mode}:{int(real)}{int(ford)}:single"] = crit(preds[0], real, for_discriminator=ford).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "ganloss.npz"), **rec)
    print("ganloss:", len(rec) - 4, "values", flush=True)


def run_disc_standalone(ref_gan):
    """TextureDiscriminator(downsample=2 / 4) on their own (the d2 of texture_only and the d3 of nd=3): the reference's
    MultiScaleDiscriminator.forward cannot run texture_only (gan.py:252 passes 4 positional arguments to a 3-argument
    forward), the member class can"""
    for name, ds, R, over in (("d_tex_ds2", 2, 256, dict(texture_resolution=256, conditional_class=False)),
                              ("d_tex_ds4_1024", 4, 256, dict(texture_resolution=1024, conditional_class=False, mask_output=False))):
        args = make_args(**over)
        torch.manual_seed(777 + ds)
        with contextlib.redirect_stdout(io.StringIO()):
            D = ref_gan.TextureDiscriminator(args, 4, ds)
        g = torch.Generator().manual_seed(778 + ds)
        x = torch.rand(2, 4, R, R, generator=g) * 2 - 1
        x.requires_grad_()
        D.train()
        y, m = D(x)
        (y * torch.linspace(-1, 1, y.numel()).view_as(y)).sum().backward()
        rec = dict(args=np.array(repr(vars(args))), ds=ds, R=R, y=y.detach().numpy(), dx=x.grad.numpy().astype(np.float16),
                   gw=D.conv2.weight_orig.grad.numpy().astype(np.float16), gw1=D.conv1.weight_orig.grad.numpy(),
                   stride_first=bool(D.stride_first), d_keys=np.array(list(D.state_dict().keys())))
        if m is not None:
            rec["m"] = m.numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(f"{name}: y {tuple(y.shape)} stride_first={D.stride_first}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None, help="case names (default: everything)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    global OUT
    if a.out:
        OUT = a.out
    ref_gan, GANLoss = import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    want = lambda n: a.only is None or n in a.only
    for name, seed, B, over in CASES:
        if want(name):
            run_case(name, seed, B, over, ref_gan, GANLoss)
    if want("g_train4"):
        run_train4(ref_gan, GANLoss)
    if want("ganloss"):
        run_ganloss(GANLoss)
    if want("d_standalone"):
        run_disc_standalone(ref_gan)


if __name__ == "__main__":
    main()
