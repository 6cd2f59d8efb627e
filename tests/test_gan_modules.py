"""GAN drop-in modules vs goldens produced by executing the reference's models/gan.py + utils/losses.py on CPU
in fp32 (oracle/gen_golden_g.py).

CPU part: parameter / buffer names, order and shapes equal the reference's state_dict key for key.
GPU part: same seed -> same initial weights -> outputs, hinge losses and per-parameter gradient norms of one
G step and one D step agree within bf16-MFMA tolerance (activations are bf16, accumulation fp32)."""
import argparse
import ast
import importlib

import numpy as np
import pytest
import torch
from conftest import load_golden

G_CASES = ["g_class128", "g_class256_nobn", "g_uncond_circ",
           # round 2: the benchmarked network (256^2 + class + syncbatch), the reference's default scale (512^2, nd=3), the
           # [2,1]-weighted nd=2 at 512^2, instance norms + class/colour conditioning, text conditioning, mask_output=False
           "g_class256_sync", "g_class512_nd3", "g_class512_nd2w", "g_inst_color128", "g_text128", "g_nomask128"]


class Tol:
    """bounds of the module-level comparisons.  BF16: the product build (bf16 activations through ~30 layers against the reference's
    fp32; DESIGN.md 8.6 shows 2-3e-3 is the floor between any two bf16 implementations).  EXACT: the fp32 EXACT build of the same
    sources (lib/libm355_exact.so, _lib.set_exact; tests/test_exact_mode_gpu.py) -- what remains is fp32 summation order, and the
    goldens' own storage precision (generated textures and full gradient tensors are kept as fp16: 2^-11 relative)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


BF16 = Tol(name="bf16", tex_mean=6e-3, tex_q999=5e-2, tex_max=1.2e-1, eval_mean=1.5e-2, eval_frac=1.5e-2, att_mean=2e-2, logit=4e-2,
           loss=2e-2, gn_med=3e-2, gn_max=0.25, gn_max_inst=0.40, cos=0.975, l2=0.25, cos_inst=0.95, l2_inst=0.35, bn_mean=2e-2,
           tr_cos0=0.85, tr_cos=0.90, tr_cos_avg=0.87, tr_mag=5e-2, tr_sq=0.10, tr_sqsum=5e-2, tr_loss0=1e-2, tr_loss=8e-2)
EXACT = Tol(name="exact", tex_mean=2e-4, tex_q999=6e-4, tex_max=1.5e-3, eval_mean=3e-4, eval_frac=1e-3, att_mean=1e-4, logit=1e-4,
            loss=1e-4, gn_med=3e-4, gn_max=1e-3, gn_max_inst=1e-3, cos=0.99999, l2=3e-3, cos_inst=0.99999, l2_inst=3e-3, bn_mean=1e-5,
            tr_cos0=0.9995, tr_cos=0.998, tr_cos_avg=0.998, tr_mag=3e-3, tr_sq=1e-2, tr_sqsum=1e-3, tr_loss0=1e-4, tr_loss=5e-3,
            grad_as_stored=True)
REPORT = {}   # measured values of the last run, per check (the exact-mode test writes them to gpurun_out/)


def _note(key, *vals):
    REPORT.setdefault(key, []).append(tuple(float(v) for v in vals))


def build(g):
    gan = importlib.import_module("2dimageto3dmodel_amd.gan")
    args = argparse.Namespace(**ast.literal_eval(str(g["args"])))
    torch.manual_seed(int(g["seed"]))
    G = gan.Generator(args, 64, symmetric=bool(g["symmetric"]), mesh_head=True)
    D = gan.MultiScaleDiscriminator(args, 4)
    return gan, args, G, D


def make_inputs(seed, B, R, n_classes):
    gen = torch.Generator().manual_seed(seed + 1)
    z = torch.randn(B, 64, generator=gen)
    c = torch.randint(0, n_classes, (B, 1), generator=gen)
    x_tex = torch.rand(B, 3, R, R, generator=gen) * 2 - 1
    x_alpha = (torch.rand(B, 1, R, R, generator=gen) > 0.4).float()
    x_mesh = 0.05 * torch.randn(B, 3, 32, 32, generator=gen)
    return z, c, x_tex, x_alpha, x_mesh


def make_extra_inputs(seed, B, args):
    """same stream as oracle/gen_golden_g.py:make_extra_inputs"""
    gen = torch.Generator().manual_seed(seed + 2)
    c2 = torch.randint(0, 10, (B, 1), generator=gen)
    L = 12
    words = torch.randn(B, args.text_embedding_dim, L, generator=gen)
    wmask = torch.zeros(B, L, dtype=torch.bool)
    wmask[:, 9:] = True
    return c2, (words, wmask)


def d_weight(args):
    return [2, 1] if args.num_discriminators == 2 and args.texture_resolution >= 512 else None


def check_full_grads(module, g, prefix, instance_norm=False, tol=None):
    """elementwise: the full gradient tensors the golden keeps (fp16) -- a transposed / permuted / sign-flipped gradient
    passes a norm check, not this one (it gives cosine ~ 0 or -1).  Thresholds: the goldens are fp32 runs at batch 2, this
    path keeps bf16 activations through ~30 layers whose batch / instance statistics are taken over 2 samples; measured
    on MI355X (round 2): cosine 0.984-0.9995 and relative L2 0.03-0.18 with batch / no normalisation, 0.97 / 0.25 with
    instance norms.  test_headline_batch8_vs_cpu_oracle shows the same quantities tighten as the batch grows."""
    tol = tol or BF16
    cos_min, l2_max = (tol.cos_inst, tol.l2_inst) if instance_norm else (tol.cos, tol.l2)
    named = dict(module.named_parameters())
    n = 0
    for k in g:
        if not k.startswith(prefix):
            continue
        want = torch.from_numpy(g[k].astype(np.float32)).flatten().double()
        got = named[k[len(prefix):]].grad.detach().cpu().flatten()
        if getattr(tol, "grad_as_stored", False) and g[k].dtype == np.float16:
            # the goldens keep these tensors as fp16 (2^-11 relative, worse below 6e-5): at the EXACT bounds that storage error is
            # the whole difference (measured 7e-4 .. 1.2e-3), so OUR gradient goes through the same rounding first
            got = got.half().float()
        got = got.double()
        if want.norm() == 0:
            assert got.abs().max() < 1e-6, k
            continue
        cos = float(torch.dot(got, want) / (got.norm() * want.norm()))
        l2 = float((got - want).norm() / want.norm())
        _note("grad " + k, cos, l2)
        assert cos >= cos_min and l2 <= l2_max, (k, cos, l2)
        n += 1
    assert n >= 4, (prefix, n)


@pytest.mark.parametrize("name", G_CASES)
def test_state_dict_keys_and_shapes_match_reference(name):
    g = load_golden(name)
    _, _, G, D = build(g)
    for mod, kk, ss in ((G, "g_keys", "g_shapes"), (D, "d_keys", "d_shapes")):
        sd = mod.state_dict()
        assert list(sd.keys()) == list(g[kk]), f"{name}: state_dict keys/order differ"
        assert [str(tuple(v.shape)) for v in sd.values()] == list(g[ss])


@pytest.mark.gpu
@pytest.mark.parametrize("name", G_CASES)
def test_g_step_and_d_step_match_reference(name):
    run_g_step_and_d_step(name, BF16)


def run_g_step_and_d_step(name, tol):
    g = load_golden(name)
    gan, args, G, D = build(g)
    dev = "cuda:0"
    G.to(dev).train()
    D.to(dev).train()
    crit = gan.GANLoss("hinge")
    B, R = int(g["B"]), int(g["R"])
    z, c, x_tex, x_alpha, x_mesh = [t.to(dev) for t in make_inputs(int(g["seed"]), B, R, 200)]
    c2, caption = make_extra_inputs(int(g["seed"]), B, args)
    if args.conditional_color:
        c = torch.cat((c, c2.to(dev)), dim=1)
    if not args.conditional_class:
        c = None
    caption = tuple(t.to(dev) for t in caption) if args.conditional_text else None
    w = d_weight(args)
    nd = args.num_discriminators
    ts = int(g.get("tex_stride", 1))
    if "eval_tex" in g:
        # ---- inference mode (ModelWrapper.forward('inference') under trainer.eval()): running statistics, sigma from the stored
        # u / v without a power iteration; must leave the training-mode goldens below untouched
        G.eval()
        with torch.no_grad():
            te, me, att = G(z, c, caption, return_attention=True)
        G.train()
        # (a freshly initialised generator in eval mode is un-normalised -- running mean 0 / variance 1 -- so conv_final's
        # pre-activation is in the hundreds and tanh saturates: where it crosses zero a bf16 rounding flips +-1.  Robust
        # comparison: mean error and the share of such pixels; measured 0.6-0.8 % mean, < 0.5 % flipped)
        e = (te.cpu() - torch.from_numpy(g["eval_tex"].astype(np.float32))).abs()
        _note("eval_tex", e.mean().item(), (e > 0.1).float().mean().item())
        assert e.mean().item() < tol.eval_mean and (e > 0.1).float().mean().item() < tol.eval_frac, (e.mean().item(), (e > 0.1).float().mean().item())
        assert (me.cpu() - torch.from_numpy(g["eval_mesh"])).abs().max().item() < 1e-6
        if "eval_att" in g:
            assert tuple(att.shape) == g["eval_att"].shape
            # (softmax over the un-normalised eval-mode activations is near one-hot: a bf16 rounding can move the argmax of
            # single pixels; the map as a whole must agree)
            _note("eval_att", (att.cpu() - torch.from_numpy(g["eval_att"])).abs().mean().item())
            assert (att.cpu() - torch.from_numpy(g["eval_att"])).abs().mean().item() < tol.att_mean
        else:
            assert att is None
    # ---- G step
    pred_tex, pred_mesh = G(z, c, caption)
    assert pred_tex.dtype == torch.float32 and tuple(pred_tex[:, :, ::ts, ::ts].shape) == g["pred_tex"].shape
    e = (pred_tex.detach().cpu()[:, :, ::ts, ::ts] - torch.from_numpy(g["pred_tex"].astype(np.float32))).abs()
    # bf16 activations through ~30 layers against fp32, after tanh (range 2): mean error, the 99.9th percentile and the
    # single worst of the ~10^5 compared values (measured over the cases: mean 3-5e-3, worst 5-8e-2; the worst value moves
    # by a few 1e-3 with any change of a summation order, e.g. the batch statistics taken from the conv's fp32 results)
    # (round 3: the text-conditioned case measured q999 = 0.0422 on one box, 0.039 on another -- sigma of the spectral norm is
    # accumulated with fp32 atomics, so the bf16 weight views differ in the last bit between runs; bound 5e-2)
    q999 = torch.quantile(e.flatten()[:1 << 24], 0.999).item()
    _note("pred_tex", e.mean().item(), q999, e.max().item())
    assert e.mean().item() < tol.tex_mean and q999 < tol.tex_q999 and e.max().item() < tol.tex_max, (e.mean().item(), q999, e.max().item())
    assert (pred_mesh.detach().cpu() - torch.from_numpy(g["pred_mesh"])).abs().max().item() < 1e-6  # zero-init head
    x_fake = torch.cat((pred_tex * x_alpha, x_alpha), dim=1)
    disc, mask = D(x_fake, pred_mesh, c, caption)
    assert len(disc) == nd and len(mask) == nd
    if args.mask_output:
        for got, want in zip(mask, [g[f"m{i + 1}"] for i in range(nd)]):
            assert np.abs(got.cpu().numpy() - want).max() < 1e-6
    else:
        assert all(m is None for m in mask)
    for got, want in zip(disc, [g[f"d{i + 1}"] for i in range(nd)]):
        _note("logits G step", np.abs(got.detach().cpu().numpy() - want).max() / max(1.0, np.abs(want).max()))
        assert np.abs(got.detach().cpu().numpy() - want).max() < tol.logit * max(1.0, np.abs(want).max())
    loss_g = crit(disc, True, for_discriminator=False, mask=mask if args.mask_output else None, weight=w)
    _note("loss_g", np.abs(loss_g.detach().cpu().numpy() - g["loss_g"]).max() / max(1.0, np.abs(g["loss_g"]).max()))
    assert np.abs(loss_g.detach().cpu().numpy() - g["loss_g"]).max() < tol.loss * max(1.0, np.abs(g["loss_g"]).max())
    loss_g.mean().backward()
    gn = {k: float(p.grad.norm()) for k, p in G.named_parameters() if p.grad is not None}
    assert list(gn.keys()) == list(g["gnorm_G_keys"])
    got, want = np.array(list(gn.values())), g["gnorm_G"]
    big = want > 1e-3 * want.max()
    rel = np.abs(got[big] / want[big] - 1)
    inst = "instance" in (args.norm_g, args.norm_d)
    # (instance norms at batch 2: measured 0.21-0.27 on the worst parameter from run to run -- the split-K atomics of the
    # weight gradients make the last bits, and with them this maximum, vary)
    _note("gnorm G", np.median(rel), rel.max())
    assert np.median(rel) < tol.gn_med and rel.max() < (tol.gn_max_inst if inst else tol.gn_max), (np.median(rel), rel.max())
    check_full_grads(G, g, "gradG:", inst, tol)
    G.zero_grad()
    D.zero_grad()
    # ---- D step
    with torch.no_grad():
        ft, fm = G(z, c, caption)
        xc = torch.cat((torch.cat((ft * x_alpha, x_alpha), 1), torch.cat((x_tex, x_alpha), 1)), 0)
        cc = torch.cat((c, c), 0) if c is not None else None
        capc = [torch.cat((t, t), 0) for t in caption] if caption is not None else None
        mc = torch.cat((fm, x_mesh), 0)
    disc2, mask2 = D(xc, mc, cc, capc)
    for got, want in zip(disc2, [g[f"dd{i + 1}"] for i in range(nd)]):
        _note("logits D step", np.abs(got.detach().cpu().numpy() - want).max() / max(1.0, np.abs(want).max()))
        assert np.abs(got.detach().cpu().numpy() - want).max() < tol.logit * max(1.0, np.abs(want).max())
    fake, real = [t[:B] for t in disc2], [t[B:] for t in disc2]
    mfake = [t[:B] for t in mask2] if args.mask_output else None
    mreal = [t[B:] for t in mask2] if args.mask_output else None
    loss_fake = crit(fake, False, for_discriminator=True, mask=mfake, weight=w)
    loss_real = crit(real, True, for_discriminator=True, mask=mreal, weight=w)
    _note("loss_d", np.abs(loss_fake.detach().cpu().numpy() - g["loss_fake"]).max() / max(1.0, np.abs(g["loss_fake"]).max()),
          np.abs(loss_real.detach().cpu().numpy() - g["loss_real"]).max() / max(1.0, np.abs(g["loss_real"]).max()))
    assert np.abs(loss_fake.detach().cpu().numpy() - g["loss_fake"]).max() < tol.loss * max(1.0, np.abs(g["loss_fake"]).max())
    assert np.abs(loss_real.detach().cpu().numpy() - g["loss_real"]).max() < tol.loss * max(1.0, np.abs(g["loss_real"]).max())
    (loss_fake + loss_real).mean().backward()
    gd = {k: float(p.grad.norm()) for k, p in D.named_parameters() if p.grad is not None}
    assert list(gd.keys()) == list(g["gnorm_D_keys"])
    got, want = np.array(list(gd.values())), g["gnorm_D"]
    big = want > 1e-3 * want.max()
    rel = np.abs(got[big] / want[big] - 1)
    _note("gnorm D", np.median(rel), rel.max())
    assert np.median(rel) < tol.gn_med and rel.max() < (tol.gn_max_inst if inst else tol.gn_max), (np.median(rel), rel.max())
    check_full_grads(D, g, "gradD:", inst, tol)
    if "running_mean" in dict(G.blk6.norm2.norm.named_buffers()):
        _note("bn running_mean", np.abs(G.blk6.norm2.norm.running_mean.cpu().numpy() - g["bn_mean_blk6"]).max())
        assert np.abs(G.blk6.norm2.norm.running_mean.cpu().numpy() - g["bn_mean_blk6"]).max() < tol.bn_mean


# ------------------------------------------------------------------------------------------------ GANLoss, all four modes
def test_ganloss_modes_match_reference():
    """utils/losses.py:21-120 executed by the reference on fixed logits (oracle/gen_golden_g.py:run_ganloss)"""
    gan = importlib.import_module("2dimageto3dmodel_amd.gan")
    g = load_golden("ganloss")
    preds = [torch.from_numpy(g["p0"]), torch.from_numpy(g["p1"])]
    masks = [torch.from_numpy(g["m0"]), torch.from_numpy(g["m1"])]
    n = 0
    for k, want in g.items():
        if ":" not in k:
            continue
        mode, flags, variant = k.split(":")
        real, ford = bool(int(flags[0])), bool(int(flags[1]))
        crit = gan.GANLoss(mode)
        if variant == "single":
            got = crit(preds[0], real, for_discriminator=ford)
        else:
            got = crit(preds, real, for_discriminator=ford, mask=masks if variant[0] == "m" else None,
                       weight=[2, 1] if variant[1] == "w" else None)
        assert np.abs(got.numpy() - want).max() < 1e-6, k
        n += 1
    assert n == 36
    with pytest.raises(AssertionError):
        gan.GANLoss("hinge")(preds, False, for_discriminator=False)
    with pytest.raises(ValueError):
        gan.GANLoss("bogus")


# ------------------------------------------------------------------------------------------------ member discriminators
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["d_tex_ds2", "d_tex_ds4_1024"])
def test_texture_discriminator_downsample_and_stride_first(name):
    """TextureDiscriminator(downsample=2) (the d2 of texture_only) and (downsample=4) at 1024^2 (stride_first through the
    resolution rule, gan.py:158-160), stand-alone, against the reference class: logits, d/dx, d/d(conv weights)"""
    gan = importlib.import_module("2dimageto3dmodel_amd.gan")
    g = load_golden(name)
    args = argparse.Namespace(**ast.literal_eval(str(g["args"])))
    ds, R = int(g["ds"]), int(g["R"])
    torch.manual_seed(777 + ds)
    D = gan.TextureDiscriminator(args, 4, ds)
    assert list(D.state_dict().keys()) == list(g["d_keys"]) and D.stride_first == bool(g["stride_first"])
    D.to("cuda:0").train()
    gen = torch.Generator().manual_seed(778 + ds)
    x = (torch.rand(2, 4, R, R, generator=gen) * 2 - 1).to("cuda:0").requires_grad_()
    y, m = D(x)
    want = g["y"]
    assert np.abs(y.detach().cpu().numpy() - want).max() < 4e-2 * max(1.0, np.abs(want).max())
    if "m" in g:
        assert np.abs(m.cpu().numpy() - g["m"]).max() < 1e-6
    else:
        assert m is None
    (y * torch.linspace(-1, 1, y.numel(), device=y.device).view_as(y)).sum().backward()
    for got, want in ((x.grad, g["dx"]), (D.conv2.weight_orig.grad, g["gw"]), (D.conv1.weight_orig.grad, g["gw1"])):
        got, want = got.detach().cpu().flatten().double(), torch.from_numpy(want.astype(np.float32)).flatten().double()
        cos = float(torch.dot(got, want) / (got.norm() * want.norm()))
        assert cos >= 0.99 and float((got - want).norm() / want.norm()) <= 0.15, (cos,)


# ------------------------------------------------------------------------------------------------ the training loop
def _trainer_args(**kw):
    a = argparse.Namespace(norm_g="syncbatch", norm_d="none", conditional_class=True, conditional_color=False,
                           conditional_text=False, n_classes=[200], texture_resolution=128, mask_output=True,
                           num_discriminators=2, texture_only=False, text_embedding_dim=256)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_divide_pred_handles_missing_masks():
    """main.py:414-422: None list / None entries (mask_output=False)"""
    train = importlib.import_module("2dimageto3dmodel_amd.train")
    assert train.divide_pred(None) == (None, None)
    f, r = train.divide_pred([None, torch.arange(4.0)])
    assert f[0] is None and r[0] is None and f[1].tolist() == [0, 1] and r[1].tolist() == [2, 3]
    f, r = train.divide_pred(torch.arange(6.0))
    assert f.tolist() == [0, 1, 2] and r.tolist() == [3, 4, 5]


def test_ema_alpha_ramp():
    """main.py:433-438"""
    train = importlib.import_module("2dimageto3dmodel_amd.train")
    assert train.ema_alpha(0.999, 0) == pytest.approx(0.999 ** 100)
    assert train.ema_alpha(0.999, 9) == pytest.approx(0.999 ** 100)
    assert train.ema_alpha(0.999, 10) == pytest.approx(0.999 ** 10)
    assert train.ema_alpha(0.999, 99) == pytest.approx(0.999 ** 10)
    assert train.ema_alpha(0.999, 100) == 0.999


@pytest.mark.gpu
def test_trainer_four_iterations_match_reference():
    run_trainer_four_iterations(BF16)


def run_trainer_four_iterations(tol):
    """GanTrainer.iteration x4 (G, D, D, G) against the reference's modules driven by the loop of main.py:691-723 with Adam
    (betas 0 / 0.9) and the running-average generator (oracle/gen_golden_g.py:run_train4): parameter DELTAS after the first
    and the fourth iteration.  Adam's first step is lr * sign(g), so the cosine of the deltas is (agreeing - disagreeing)
    signs: 0.85 = 92.5 % of the signs agree (bf16 noise flips near-zero gradients; measured 0.90-0.99); a wrong gradient,
    a wrong beta or a wrong step size gives ~0 or a different magnitude (checked separately)."""
    train = importlib.import_module("2dimageto3dmodel_amd.train")
    g = load_golden("g_train4")
    seed, B = int(g["seed"]), int(g["B"])
    torch.manual_seed(seed)
    tr = train.GanTrainer(_trainer_args(), device="cuda:0", mesh_template=None)
    tr.train()
    gp, dp, ap = (dict(m.named_parameters()) for m in (tr.generator, tr.discriminator, tr.generator_running_avg))
    track_g = ["blk6.conv2.weight_orig", "blk1.norm1.fc_beta.bias", "conv_mesh.weight"]
    track_d = ["d1.conv2.weight_orig", "d2.conv3.bias"]
    w0 = {("G", k): gp[k].detach().clone() for k in track_g}
    w0.update({("D", k): dp[k].detach().clone() for k in track_d})

    def cos(a, b):
        a, b = a.flatten().double().cpu(), torch.from_numpy(b).flatten().double()
        return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))

    losses = []
    for it in range(int(g["iters"])):
        z, c, x_tex, x_alpha, x_mesh = [t.to("cuda:0") for t in make_inputs(seed + 10 * it, B, 128, 200)]
        out = tr.iteration(x_tex, x_alpha, x_mesh, c, noise=z, epoch=int(g["epoch"]))
        losses.append([float(out["g"]), 0.0] if "g" in out else [float(out["d_fake"]), float(out["d_real"])])
        if it in (0, int(g["iters"]) - 1):
            for k in track_g:
                dG, dA = gp[k].detach() - w0[("G", k)], ap[k].detach() - w0[("G", k)]
                want, want_a = g[f"it{it}:G:{k}"], g[f"it{it}:avg:{k}"]
                _note(f"train it{it} dG {k}", cos(dG, want), abs(float(dG.abs().max()) / np.abs(want).max() - 1))
                assert cos(dG, want) > (tol.tr_cos0 if it == 0 else tol.tr_cos), (it, k, cos(dG, want))
                assert abs(float(dG.abs().max()) / np.abs(want).max() - 1) < tol.tr_mag, (it, k)
                # the running average moved by (1 - alpha_epoch) of the generator's displacement (alpha ramp, main.py:433-438)
                # (after four iterations the average's displacement is dominated by near-zero-gradient entries whose Adam sign
                # flips with the fp32-atomic summation order of the wgrad: measured 0.895-0.99 over runs; bound 0.87)
                _note(f"train it{it} dAvg {k}", cos(dA, want_a), abs(float(dA.norm()) / np.linalg.norm(want_a) - 1))
                assert cos(dA, want_a) > (tol.tr_cos0 if it == 0 else tol.tr_cos_avg), (it, k, cos(dA, want_a))
                assert abs(float(dA.norm()) / np.linalg.norm(want_a) - 1) < tol.tr_mag, (it, k)
            if it > 0:
                for k in track_d:
                    dD = dp[k].detach() - w0[("D", k)]
                    _note(f"train it{it} dD {k}", cos(dD, g[f"it{it}:D:{k}"]))
                    assert cos(dD, g[f"it{it}:D:{k}"]) > tol.tr_cos, (it, k, cos(dD, g[f"it{it}:D:{k}"]))
            bn = tr.generator_running_avg.blk6.norm2.norm
            _note(f"train it{it} avg bn mean", np.abs(bn.running_mean.cpu().numpy() - g[f"it{it}:avg_bn_mean"]).max())
            assert np.abs(bn.running_mean.cpu().numpy() - g[f"it{it}:avg_bn_mean"]).max() < tol.bn_mean
            assert int(bn.num_batches_tracked) == int(g[f"it{it}:avg_nbt"])
    # NOT sign-dominated: Adam's second-moment estimate after the last iteration (two optimiser steps each) is smooth in the
    # gradients -- exp_avg_sq = 0.9 * 0.1 g1^2 + 0.1 g2^2 (main.py:588-589: betas (0, 0.9)) -- a wrong beta2 / a squared-twice /
    # a missing step shows here although the parameter deltas above are +-lr sign steps
    for opt, mod_params, kind, k in ((tr.optimizer_g, gp, "G", "blk6.conv2.weight_orig"), (tr.optimizer_d, dp, "D", "d1.conv2.weight_orig")):
        want = torch.from_numpy(g[f"it{int(g['iters']) - 1}:{kind}:exp_avg_sq:{k}"]).flatten().double()
        got = opt.state[mod_params[k]]["exp_avg_sq"].detach().cpu().flatten().double()
        rel = float((got - want).norm() / want.norm())
        _note(f"train exp_avg_sq {kind}", rel, abs(float(got.sum() / want.sum()) - 1))
        assert rel <= tol.tr_sq, (kind, k, rel)
        assert abs(float(got.sum() / want.sum()) - 1) < tol.tr_sqsum, (kind, k)
        assert float(opt.state[mod_params[k]]["step"]) == 2.0
    # (the later losses are computed on weights that differ by the flipped first Adam steps: relative tolerance)
    # measured over repeated runs: first iteration <= 2e-3, later ones up to 3.7e-2 (lr_d = 4e-4 sign steps on flipped entries)
    ltol = np.array([[tol.tr_loss0, tol.tr_loss0]] + [[tol.tr_loss, tol.tr_loss]] * (len(losses) - 1))
    _note("train losses", (np.abs(np.array(losses) - g["losses"]) / np.maximum(1.0, np.abs(g["losses"]))).max())
    assert (np.abs(np.array(losses) - g["losses"]) <= ltol * np.maximum(1.0, np.abs(g["losses"]))).all(), (losses, g["losses"])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_headline_batch8_vs_cpu_oracle():
    """The benchmarked network (256^2, class-conditional, syncbatch, nd=2) at batch 8 against the pinned fp32 CPU oracle
    (oracle/gan_cpu.py, itself checked against the reference's goldens by tests/test_oracle_golden.py): with 4x the batch
    of the goldens the bf16 noise in the batch statistics and weight gradients averages out -- tighter elementwise bounds."""
    from oracle import gan_cpu as gc
    gan = importlib.import_module("2dimageto3dmodel_amd.gan")
    args = _trainer_args(texture_resolution=256)
    torch.manual_seed(8642)
    Gm, Dm = gan.Generator(args, 64, symmetric=True, mesh_head=True), gan.MultiScaleDiscriminator(args, 4)
    wg, wd = gc.Weights(Gm.state_dict()), gc.Weights(Dm.state_dict())
    B, R = 8, 256
    z, c, x_tex, x_alpha, x_mesh = make_inputs(8642, B, R, 200)
    torch.set_num_threads(min(32, max(1, (torch.get_num_threads()))))
    loss_r, tex_r, mesh_r, disc_r, _ = gc.g_step(wg, wd, args, z, c, x_alpha)
    loss_r.mean().backward()
    Gm.to("cuda:0").train()
    Dm.to("cuda:0").train()
    crit = gan.GANLoss("hinge")
    zd, cd, ad = z.cuda(), c.cuda(), x_alpha.cuda()
    pred_tex, pred_mesh = Gm(zd, cd)
    e = (pred_tex.detach().cpu() - tex_r.detach()).abs()
    assert e.mean().item() < 6e-3 and e.max().item() < 1.2e-1, (e.mean().item(), e.max().item())
    G_ops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
    disc, mask = Dm(G_ops.mask_cat(pred_tex, ad), pred_mesh, cd)
    for got, want in zip(disc, disc_r):
        assert (got.detach().cpu() - want.detach()).abs().max().item() < 4e-2 * max(1.0, want.abs().max().item())
    loss = crit(disc, True, for_discriminator=False, mask=mask, weight=None)
    assert abs(loss.item() - loss_r.item()) < 1e-2
    loss.mean().backward()
    ref = wg.grads()
    named = dict(Gm.named_parameters())
    report = {}
    for k in ("blk6.conv2.weight_orig", "blk5.conv1.weight_orig", "blk3a.conv2.weight_orig", "blk4.shortcut.weight_orig",
              "blk1.norm1.fc_gamma.weight", "blk6.norm2.fc_beta.weight", "emb_class.weight", "conv_final.weight", "fc.weight"):
        a, b = named[k].grad.detach().cpu().flatten().double(), ref[k].flatten().double()
        report[k] = (float(torch.dot(a, b) / (a.norm() * b.norm())), float((a - b).norm() / b.norm()))
    worst = min(v[0] for v in report.values())
    assert worst >= 0.995 and max(v[1] for v in report.values()) <= 0.10, report


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_headline_batch8_dstep_vs_cpu_oracle():
    """The D step of the benchmarked network (ModelWrapper.forward('d'), code/main.py:499-520: generator forward without grad,
    discriminators on the [fake; real] batch of 2B = 16, the two hinge terms) at batch 8 against the pinned fp32 CPU oracle
    (oracle/gan_cpu.py:d_step): logits, both losses and FULL discriminator gradient tensors elementwise.  Covers what the G-step
    test above does not touch: mask_cat with the real half, d_losses (divide_pred by index), the bit-mask dgrad chain of
    conv1..conv3, the premasked tail (conv4 -> conv5 + projection), every discriminator wgrad and the fused bias gradients."""
    from oracle import gan_cpu as gc
    gan = importlib.import_module("2dimageto3dmodel_amd.gan")
    G_ops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
    args = _trainer_args(texture_resolution=256)
    torch.manual_seed(8643)
    Gm, Dm = gan.Generator(args, 64, symmetric=True, mesh_head=True), gan.MultiScaleDiscriminator(args, 4)
    wg, wd = gc.Weights(Gm.state_dict(), grad=False), gc.Weights(Dm.state_dict())
    B, R = 8, 256
    z, c, x_tex, x_alpha, x_mesh = make_inputs(8643, B, R, 200)
    torch.set_num_threads(min(32, max(1, (torch.get_num_threads()))))
    lf_r, lr_r, disc_r = gc.d_step(wg, wd, args, z, c, x_tex, x_alpha, x_mesh)
    (lf_r.mean() + lr_r.mean()).backward()
    Gm.to("cuda:0").train()
    Dm.to("cuda:0").train()
    crit = gan.GANLoss("hinge")
    zd, cd, td, ad, md = (t.cuda() for t in (z, c, x_tex, x_alpha, x_mesh))
    with torch.no_grad():
        pred_tex, pred_mesh = Gm(zd, cd)
        X_comb = G_ops.mask_cat(pred_tex, ad, td)
        C_comb, M_comb = torch.cat((cd, cd), dim=0), torch.cat((pred_mesh, md), dim=0)
    disc, mask = Dm(X_comb, M_comb, C_comb)
    for got, want in zip(disc, disc_r):
        assert got.shape == want.shape
        # (the fake half sees a bf16 generator's texture: same tolerance as the G-step test's logits)
        assert (got.detach().cpu() - want.detach()).abs().max().item() < 4e-2 * max(1.0, want.abs().max().item())
    loss_fake, loss_real = crit.d_losses(disc, mask, None)
    assert abs(loss_fake.item() - lf_r.item()) < 1e-2 and abs(loss_real.item() - lr_r.item()) < 1e-2, \
        (loss_fake.item(), lf_r.item(), loss_real.item(), lr_r.item())
    (loss_fake.mean() + loss_real.mean()).backward()
    ref = wd.grads()
    named = dict(Dm.named_parameters())
    report = {}
    for k in ("d1.conv1.weight_orig", "d1.conv2.weight_orig", "d1.conv3.weight_orig", "d1.conv4.weight_orig", "d1.conv5.weight_orig",
              "d2.conv2.weight_orig", "d2.conv1.weight_orig", "d1.projector.weight", "d1.conv2.bias", "d1.conv4.bias", "d1.conv1.bias"):
        a, b = named[k].grad.detach().cpu().flatten().double(), ref[k].flatten().double()
        report[k] = (float(torch.dot(a, b) / (a.norm() * b.norm())), float((a - b).norm() / b.norm()))
    worst = min(v[0] for v in report.values())
    assert worst >= 0.995 and max(v[1] for v in report.values()) <= 0.10, report
    assert all(p.grad is None for p in Gm.parameters())   # the generator ran without grad


def _cycle_batches(B, R, n=3, seed0=5150):
    out = []
    for i in range(n):
        z, c, x_tex, x_alpha, x_mesh = make_inputs(seed0 + i, B, R, 200)
        out.append(([x_tex.cuda(), x_alpha.cuda(), x_mesh.cuda(), c.cuda()], z.cuda()))
    return out


def _state_bits(tr):
    """every tensor a training cycle touches, as exact bit patterns: weights + buffers of the three networks, Adam's moments"""
    out = {}
    for name, mod in (("G", tr.generator), ("D", tr.discriminator), ("avg", tr.generator_running_avg)):
        for k, v in mod.state_dict().items():
            out[f"{name}.{k}"] = v.detach().clone()
    for name, opt, mod in (("optG", tr.optimizer_g, tr.generator), ("optD", tr.optimizer_d, tr.discriminator)):
        for k, p in mod.named_parameters():
            for sk, sv in opt.state.get(p, {}).items():
                if torch.is_tensor(sv):
                    out[f"{name}.{k}.{sk}"] = sv.detach().clone()
    return out


def _assert_bit_identical(sa, sb, what):
    assert sa.keys() == sb.keys(), (what, sa.keys() ^ sb.keys())
    bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
    assert not bad, (what, len(bad), bad[:8])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_deterministic_mode_two_eager_runs_are_bit_identical(tmp_path):
    """M355_DETERMINISTIC (pkg.set_deterministic): two runs of two training cycles (G, D, D each, Adam, running-average generator,
    the mesh regulariser in the G step) from the same seed are BIT-identical in every weight, buffer and optimiser moment -- as the
    reference's CPU path is (code/main.py:691-723).  What makes it so: the split-K weight gradients accumulate as
    64-bit fixed point (m355_conv2d_wgrad_det), every other cross-workgroup sum of the path (spectral-norm norms, hinge loss, head
    bias gradient, class-projection embedding gradient, BN statistics) is an ordered sum of per-workgroup partials, and the mesh
    backward is a gather.  The default mode (fp32 atomics in the wgrad) is measured NOT to be: the same two runs differ."""
    pkg = importlib.import_module("2dimageto3dmodel_amd")
    train = importlib.import_module("2dimageto3dmodel_amd.train")
    mesh = importlib.import_module("2dimageto3dmodel_amd.mesh")
    tpl = mesh.MeshTemplate(mesh.write_uv_sphere_obj(str(tmp_path / "uvsphere_16rings.obj")), is_symmetric=True, device="cuda:0")
    batches = _cycle_batches(4, 128)

    def run():
        torch.manual_seed(515)
        tr = train.GanTrainer(_trainer_args(), device="cuda:0", mesh_template=tpl)
        tr.train()
        losses = []
        for _ in range(2):
            for b, z in batches:
                losses += [float(v) for v in tr.iteration(*b, noise=z, epoch=0).values()]
        torch.cuda.synchronize()
        return _state_bits(tr), losses

    prev = pkg.set_deterministic(True)
    try:
        s1, l1 = run()
        s2, l2 = run()
    finally:
        pkg.set_deterministic(prev)
    assert l1 == l2, (l1, l2)
    _assert_bit_identical(s1, s2, "deterministic mode, eager vs eager")
    # and the mode changes nothing but the summation: the same cycles in the default mode agree to fp32-atomic noise
    s3, l3 = run()
    assert max(abs(a - b) for a, b in zip(l1, l3)) < 6e-2


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_captured_cycle_replays_like_eager():
    """GanTrainer.capture_cycle: one training cycle (G, D, D with their Adam steps and the running-average update) recorded
    into a hipGraph.  Two trainers from the same seed, the same loader batches and the same latent batches, deterministic
    mode: A runs three cycles eagerly; B captures (two warm-up cycles, which capture_cycle undoes) and replays three times.
    Every weight, buffer and Adam moment must be BIT-identical, and so must the losses of the last cycle: a replay that
    skipped an optimiser step, froze the noise, re-used stale inputs, or a capture that left its warm-up cycles' training
    behind, differs.  (In the default mode the split-K wgrad atomics make even two eager runs differ:
    profiles/r03_graph_noise.txt -- the round-3 version of this test could only ask for cosine >= 0.90.)"""
    pkg = importlib.import_module("2dimageto3dmodel_amd")
    train = importlib.import_module("2dimageto3dmodel_amd.train")
    batches = _cycle_batches(4, 128)

    def fresh():
        torch.manual_seed(515)
        tr = train.GanTrainer(_trainer_args(), device="cuda:0", mesh_template=None, capturable=True)
        tr.train()
        return tr

    prev = pkg.set_deterministic(True)
    try:
        A, Bt = fresh(), fresh()
        _assert_bit_identical(_state_bits(A), _state_bits(Bt), "two trainers from one seed")
        out_a = {}
        for _ in range(3):
            for b, z in batches:
                out_a.update(A.iteration(*b, noise=z, epoch=0))
        before = _state_bits(Bt)
        rng_before = torch.cuda.get_rng_state("cuda:0").clone()
        cyc = Bt.capture_cycle([b for b, _ in batches], epoch=0, warmup=2, noises=[z for _, z in batches])
        torch.cuda.synchronize()
        # capturing trained nothing: weights, buffers, total_it and the RNG are where they were (the optimiser state now exists,
        # zero-filled = Adam's initial state)
        assert Bt.total_it == 0 and torch.equal(torch.cuda.get_rng_state("cuda:0"), rng_before)
        after = _state_bits(Bt)
        _assert_bit_identical(before, {k: v for k, v in after.items() if k in before}, "capture_cycle left training behind")
        assert all(float(v.abs().max()) == 0.0 for k, v in after.items() if k not in before), "fresh optimiser state must be zero"
        for _ in range(3):
            out_b = cyc.replay()
        torch.cuda.synchronize()
        assert Bt.total_it == A.total_it == 9
        for k in ("g", "d_fake", "d_real"):
            assert float(out_a[k]) == float(out_b[k]), (k, float(out_a[k]), float(out_b[k]))
        _assert_bit_identical(_state_bits(A), _state_bits(Bt), "3 eager cycles vs capture + 3 replays")
        sb = Bt.optimizer_g.state[dict(Bt.generator.named_parameters())["blk6.conv2.weight_orig"]]["step"]
        assert float(sb) == 3.0
    finally:
        pkg.set_deterministic(prev)
    # a replay with new loader batches refills the static buffers
    out_c = cyc.replay([b for b, _ in batches[::-1]])
    torch.cuda.synchronize()
    assert all(np.isfinite(float(v)) for v in out_c.values())
    with pytest.raises(RuntimeError):
        train.GanTrainer(_trainer_args(), device="cuda:0", mesh_template=None).capture_cycle([b for b, _ in batches])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_second_stream_branches_change_no_bit(monkeypatch):
    """gan_ops.Fork (the mesh discriminator and the generator's mesh head on a second HIP stream, small batches): the forked run
    must be BIT-identical to the single-stream run in deterministic mode -- a missing wait between the streams, a zero fill that
    races a slice's first use, or memory handed back to the allocator while the other stream still reads it shows up as a
    difference (or as NaNs).  Two cycles each, incl. the deferred weight-gradient finish that joins the side stream."""
    pkg = importlib.import_module("2dimageto3dmodel_amd")
    train = importlib.import_module("2dimageto3dmodel_amd.train")
    gops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
    batches = _cycle_batches(4, 128, seed0=6100)

    def run(streams):
        monkeypatch.setattr(gops, "STREAMS_ON", streams)
        torch.manual_seed(616)
        tr = train.GanTrainer(_trainer_args(), device="cuda:0", mesh_template=None)
        tr.train()
        losses = []
        for _ in range(2):
            for b, z in batches:
                losses += [float(v) for v in tr.iteration(*b, noise=z, epoch=0).values()]
        torch.cuda.synchronize()
        return _state_bits(tr), losses

    prev = pkg.set_deterministic(True)
    try:
        assert gops.FORK_MAX_BATCH >= 4
        s_on, l_on = run(True)
        assert gops.side_streams(), "the fork did not run"
        s_off, l_off = run(False)
    finally:
        pkg.set_deterministic(prev)
    assert l_on == l_off, (l_on, l_off)
    _assert_bit_identical(s_on, s_off, "second stream on vs off")


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_captured_cycle_at_batch16_256_follows_the_cpu_oracle_loop():
    """BASELINE configs[2] as bench.py now runs it by default (batch 16, 256^2, nd = 2, class-conditional, syncbatch: the captured
    cycle replayed from one hipGraph): ONE replayed cycle -- G step, D step, D step, each with its Adam(0, 0.9) update -- against the
    loop of main.py:691-723 restated on the pinned CPU oracle (oracle/gan_cpu.py: g_step / d_step / adam_step) from the same weights,
    batches and noise.  The G step's loss is a pure forward comparison; behind it every update is Adam's FIRST step (lr * sign(g)), so
    parameter displacements are compared as tests/test_gan_modules.py::run_trainer_four_iterations compares g_train4: cosine = the
    share of agreeing signs, and magnitude."""
    import os
    from oracle import gan_cpu as gc
    train = importlib.import_module("2dimageto3dmodel_amd.train")
    args = _trainer_args(texture_resolution=256)
    B, R, seed = 16, 256, 9160
    torch.manual_seed(seed)
    tr = train.GanTrainer(args, device="cuda:0", mesh_template=None, capturable=True)
    tr.train()
    sd_g = {k: v.detach().cpu().clone() for k, v in tr.generator.state_dict().items()}
    sd_d = {k: v.detach().cpu().clone() for k, v in tr.discriminator.state_dict().items()}
    ins = [make_inputs(seed + 10 * it, B, R, 200) for it in range(3)]       # (z, c, x_tex, x_alpha, x_mesh)
    batches = [tuple(t.to("cuda:0") for t in (x_tex, x_alpha, x_mesh, c)) for (_z, c, x_tex, x_alpha, x_mesh) in ins]
    cyc = tr.capture_cycle(batches, epoch=0, noises=[i[0].to("cuda:0") for i in ins])
    # (capturing trains nothing: the weights are still the initial ones)
    assert torch.equal(tr.generator.state_dict()["blk6.conv2.weight_orig"].cpu(), sd_g["blk6.conv2.weight_orig"])
    out = {k: float(v) for k, v in cyc.replay().items()}
    torch.cuda.synchronize()
    # ---- the reference loop on the CPU oracle
    nthr = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 1) // 8)))
    try:
        wg, wd = gc.Weights(sd_g), gc.Weights(sd_d)
        pg = {k: v for k, v in wg.store.items() if v.requires_grad}
        pd = {k: v for k, v in wd.store.items() if v.requires_grad}
        sg, sdd = {}, {}
        z, c, x_tex, x_alpha, x_mesh = ins[0]
        loss_g = gc.g_step(wg, wd, args, z, c, x_alpha)[0].mean()
        loss_g.backward()
        gc.adam_step(pg, wg.grads(), sg, 1e-4, 1)
        ref_d = []
        for step, (z, c, x_tex, x_alpha, x_mesh) in enumerate(ins[1:], start=1):
            wd.zero_grad()
            lf, lr, _ = gc.d_step(wg, wd, args, z, c, x_tex, x_alpha, x_mesh)
            (lf + lr).mean().backward()
            gc.adam_step(pd, wd.grads(), sdd, 4e-4, step)
            ref_d.append((float(lf.mean()), float(lr.mean())))
    finally:
        torch.set_num_threads(nthr)
    _note("b16 graph g loss", abs(out["g"] - float(loss_g)), out["g"])
    assert abs(out["g"] - float(loss_g)) < 2e-2 * max(1.0, abs(float(loss_g))), (out["g"], float(loss_g))
    # (the replay returns the LAST D step's losses; that step ran on weights one sign-step of lr_d = 4e-4 away from the first one's)
    for got, want in ((out["d_fake"], ref_d[-1][0]), (out["d_real"], ref_d[-1][1])):
        assert abs(got - want) < 6e-2 * max(1.0, abs(want)), (out, ref_d)
    gp, dp = dict(tr.generator.named_parameters()), dict(tr.discriminator.named_parameters())

    def cos_mag(now, before, want_now):
        dg = (now.detach().cpu() - before).flatten().double()
        dw = (want_now.detach() - before).flatten().double()
        return float(torch.dot(dg, dw) / (dg.norm() * dw.norm() + 1e-300)), float(dg.abs().max() / dw.abs().max())

    for k in ("blk6.conv2.weight_orig", "blk4.conv1.weight_orig", "blk1.norm1.fc_beta.bias"):
        cs, mg = cos_mag(gp[k], sd_g[k], wg.store[k])
        _note(f"b16 graph dG {k}", cs, mg)
        assert cs > 0.85 and abs(mg - 1) < 0.05, (k, cs, mg)          # one Adam step: +-lr on every entry
    for k in ("d1.conv2.weight_orig", "d1.conv4.weight_orig", "d2.conv3.bias"):
        cs, mg = cos_mag(dp[k], sd_d[k], wd.store[k])
        _note(f"b16 graph dD {k}", cs, mg)
        assert cs > 0.80 and abs(mg - 1) < 0.25, (k, cs, mg)          # two steps; the second on weights the first one's flips moved
    assert tr.total_it == 3


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_spectral_norm_prefetch_changes_no_bit(monkeypatch):
    """gan_ops.SpectralNormGroup.prefetch (round 6): the power iteration + bf16 weight views of a network's NEXT forward issued on a
    second stream as soon as its weights are final (train.GanTrainer._prefetch_sn).  An execution detail: three cycles with it must
    be BIT-identical to three cycles without, eagerly and as hipGraph replays; a stale prefetch (weights written behind its back)
    must be undone, not consumed; state_dict() must show the u / v of the last forward."""
    pkg = importlib.import_module("2dimageto3dmodel_amd")
    train = importlib.import_module("2dimageto3dmodel_amd.train")
    gops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
    batches = _cycle_batches(4, 128, seed0=6600)

    def run(prefetch, graph=False, meddle=False):
        monkeypatch.setattr(gops, "SN_PREFETCH_ON", prefetch)
        torch.manual_seed(661)
        tr = train.GanTrainer(_trainer_args(), device="cuda:0", mesh_template=None, capturable=graph)
        tr.train()
        losses = []
        if graph:
            cyc = tr.capture_cycle([b for b, _ in batches], epoch=0, noises=[z for _, z in batches])
            for _ in range(3):
                losses += [float(v) for v in cyc.replay().values()]
        else:
            for k in range(3):
                for i, (b, z) in enumerate(batches):
                    losses += [float(v) for v in tr.iteration(*b, noise=z, epoch=0).values()]
                    if meddle and k == 1 and i == 1:
                        # behind a pending prefetch's back: an in-place write (same values) bumps the version counters -> stale
                        with torch.no_grad():
                            for p in list(tr.generator.parameters()) + list(tr.discriminator.parameters()):
                                p.mul_(1.0)
        torch.cuda.synchronize()
        pend = [net._sn_group()._pending is not None for net in (tr.generator, tr.discriminator)]
        return _state_bits(tr), losses, pend        # (_state_bits -> Module.state_dict() undoes a pending step)

    prev = pkg.set_deterministic(True)
    try:
        s_off, l_off, _ = run(False)
        s_on, l_on, pend = run(True)
        assert gops._SN_SIDE, "no prefetch ran"
        assert all(pend), pend                      # (eager: the last D step prefetched for the next cycle's G step)
        s_med, l_med, _ = run(True, meddle=True)
        s_g_on, l_g_on, _ = run(True, graph=True)
        s_g_off, l_g_off, _ = run(False, graph=True)
    finally:
        pkg.set_deterministic(prev)
    assert l_on == l_off, (l_on, l_off)
    _assert_bit_identical(s_on, s_off, "spectral-norm prefetch on vs off")
    assert l_med == l_off
    _assert_bit_identical(s_med, s_off, "stale prefetch undone")
    assert l_g_on == l_g_off, (l_g_on, l_g_off)
    _assert_bit_identical(s_g_on, s_g_off, "spectral-norm prefetch inside a captured cycle")
    _assert_bit_identical(s_g_on, s_off, "captured cycle with prefetch vs eager without")


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_deterministic_cycles_at_256_in_every_execution_mode():
    """The benchmarked resolution (256^2; batch 16 keeps it short): eight training cycles run eagerly on one stream, eagerly with the
    small branches on the second stream, and as hipGraph replays with either -- four trainers from one seed on the same batches,
    deterministic mode: every weight, buffer and Adam moment bit-identical, losses equal.  The 128^2 checks above do not reach
    the kernels of the 256^2 networks (blk6, the 256 x 256 discriminator layers): this is the configuration in which round 4's
    soak (scripts/soak_determinism.py) exposed the stale-weight race of k_conv_halo's look-ahead variants."""
    pkg = importlib.import_module("2dimageto3dmodel_amd")
    train = importlib.import_module("2dimageto3dmodel_amd.train")
    gops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
    batches = _cycle_batches(16, 256, seed0=7300)
    ncyc = 8

    def fresh():
        torch.manual_seed(733)
        tr = train.GanTrainer(_trainer_args(texture_resolution=256), device="cuda:0", mesh_template=None, capturable=True)
        tr.train()
        return tr

    def eager():
        tr, last = fresh(), {}
        for _ in range(ncyc):
            for b, z in batches:
                last.update(tr.iteration(*b, noise=z, epoch=0))
        torch.cuda.synchronize()
        return _state_bits(tr), {k: float(v) for k, v in last.items()}

    def graph():
        tr = fresh()
        cyc = tr.capture_cycle([b for b, _ in batches], epoch=0, warmup=2, noises=[z for _, z in batches])
        for _ in range(ncyc):
            out = cyc.replay()
        torch.cuda.synchronize()
        return _state_bits(tr), {k: float(v) for k, v in out.items()}

    prev, prev_streams = pkg.set_deterministic(True), gops.STREAMS_ON
    try:
        runs = {}
        for streams in (False, True):
            gops.STREAMS_ON = streams
            runs[f"eager, streams {streams}"] = eager()
            runs[f"graph, streams {streams}"] = graph()
    finally:
        pkg.set_deterministic(prev)
        gops.STREAMS_ON = prev_streams
    ref_state, ref_losses = runs["eager, streams False"]
    assert all(np.isfinite(v) for v in ref_losses.values())
    for name, (state, losses) in runs.items():
        assert losses == ref_losses, (name, losses, ref_losses)
        _assert_bit_identical(ref_state, state, f"eager on one stream vs {name}")
