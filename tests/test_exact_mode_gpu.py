"""GPU: the EXACT build of the library (lib/libm355_exact.so = the same sources with -DM355_EXACT, include/m355.h m355_act_bytes;
SURVEY.md 8c "an fp32-accumulate exact mode for 1e-4 checks").

Same C-ABI, same Python orchestration (gan.py / gan_ops.py / train.py), but the activation tensors are fp32 and the convolutions
are plain fp32 kernels with fp64 accumulation (csrc/conv_exact.hip); the elementwise / reduction kernels are the product's own
(csrc/gan_elem.hip, gan_io.hip, gan_glue.hip) with the storage type switched.  What this pins -- at 1e-4 instead of through bf16
noise -- is everything ABOVE a conv layer: conditional batch-norm statistics and eps (/root/reference/code/models/gan.py:264-286,
sync_batchnorm/batchnorm.py:70-73), spectral-norm iteration order, hinge masking (utils/losses.py:62-120), Adam(0, 0.9)
(main.py:588-589) and the running-average ramp (main.py:431-447), against the goldens the reference itself produced on CPU."""
import importlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import test_gan_modules as tm
from conftest import load_golden
from test_conv_gpu import ref_conv

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def exact(pkg):
    lib = importlib.import_module("2dimageto3dmodel_amd._lib")
    prev = lib.set_exact(True)
    try:
        yield lib
    finally:
        lib.set_exact(prev)


def _dump_report(tag):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "exact_report.jsonl"), "a") as f:
        f.write(json.dumps({"case": tag, "measured": {k: v for k, v in tm.REPORT.items()}}) + "\n")
    tm.REPORT.clear()


def test_exact_build_identity(exact):
    L = exact.lib()
    assert L.m355_act_bytes() == 4 and exact.act_dtype() == torch.float32 and exact.is_exact()
    assert L.m355_abi_version() == 4
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    d = conv.make_desc(2, 16, 32, 128, 64, 3, 3, 1, 1, 1, 1, 1)
    # no bit masks, no fused statistics, no workspaces: the callers take their generic paths
    assert not conv.maskbits_ok(d, 0) and conv.conv_stats_rows(d) == 0 and conv.dgrad_mask_ok(d) and conv.wgrad_fuses_dbias(d)


def test_product_build_is_back_after_the_fixture(pkg):
    lib = importlib.import_module("2dimageto3dmodel_amd._lib")
    assert lib.lib().m355_act_bytes() == 2 and lib.act_dtype() == torch.bfloat16


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups
    (2, 8, 4, 64, 128, 3, 1, 1, 1, 1, 1),     # ResBlockUp.conv1: upsample + replicate
    (1, 16, 8, 128, 64, 1, 1, 0, 0, 0, 0),    # 1x1 shortcut
    (2, 16, 16, 8, 64, 5, 1, 2, 2, 2, 0),     # D.conv1: 5x5 circular
    (2, 16, 16, 64, 128, 4, 2, 1, 1, 2, 0),   # D.conv2: 4x4 stride 2 circular
    (1, 8, 8, 64, 3, 5, 1, 2, 2, 1, 0),       # conv_final: replicate 5x5, 3 channels
    (2, 16, 32, 128, 64, 3, 1, 1, 1, 2, 1),   # upsample + circular
    (2, 12, 6, 96, 96, 3, 1, 1, 1, 0, 0),     # zero W pad
    (2, 16, 16, 64, 128, 3, 2, 1, 1, 0, 0),   # 3x3 stride 2 (reconstruction encoder)
    (2, 8, 8, 512, 1, 5, 1, 2, 2, 2, 0),      # D.conv5: one output channel
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_exact_conv_matches_fp32_reference(exact, case):
    """forward (+ bias, LeakyReLU, both output layouts), dgrad (+ fused activation backward), wgrad (+ bias gradient) of the fp32
    kernels against torch-CPU F.conv2d on UNROUNDED fp32 operands: fp32 summation-order noise only"""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    sigma = torch.tensor([1.7])
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y_ref = ref_conv(xr, wr / 1.7, br, stride, ph, pw, mode, ups)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    wf, wd = conv.weight_prep(d, w.to(DEV), sigma=sigma.to(DEV))
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    scale = y_ref.abs().max().item()
    y = conv.conv_fwd(d, x_nhwc, wf, b.to(DEV))
    assert y.dtype == torch.float32 and (y.cpu().permute(0, 3, 1, 2) - y_ref.detach()).abs().max().item() < 5e-6 * scale
    yl = conv.conv_fwd(d, x_nhwc, wf, b.to(DEV), out_f32_nchw=True, slope=0.2).cpu()
    assert (yl - F.leaky_relu(y_ref.detach(), 0.2)).abs().max().item() < 5e-6 * scale
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    c32 = conv.dy_channels(Cout)
    dy_nhwc = torch.zeros(N, y_ref.shape[2], y_ref.shape[3], c32)
    dy_nhwc[..., :Cout] = dy.permute(0, 2, 3, 1)
    dyd = dy_nhwc.to(DEV)
    dx = conv.conv_dgrad(d, dyd, wd).cpu().permute(0, 3, 1, 2)
    assert (dx - xr.grad).abs().max().item() < 5e-6 * xr.grad.abs().max().item()
    dxm = conv.conv_dgrad(d, dyd, wd, mask_x=x_nhwc, mask_slope=0.2).cpu().permute(0, 3, 1, 2)
    want = xr.grad * torch.where(x > 0, 1.0, 0.2)
    assert (dxm - want).abs().max().item() < 5e-6 * want.abs().max().item()
    db = torch.empty(Cout, device=DEV)
    dw = conv.conv_wgrad(d, x_nhwc, dyd, dbias=db).cpu()            # gradient with respect to the conv's (divided) weight
    want_w = wr.grad * 1.7
    assert (dw - want_w).abs().max().item() < 1e-5 * want_w.abs().max().item()
    assert (db.cpu() - br.grad).abs().max().item() < 1e-5 * max(1.0, br.grad.abs().max().item())


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("name", tm.G_CASES)
def test_exact_g_step_and_d_step_match_reference(exact, name):
    """the nine reference-executed fp32 goldens (one G step + one D step each; 128^2 / 256^2 / 512^2, nd 2 / 3, sync / batch /
    instance / no norm, class / colour / text conditioning, no-mask) at the EXACT tolerances: logits and losses 1e-4, gradient
    norms 1e-3; the stored gradient tensors at relative L2 <= 3e-3 -- that bound is the GOLDENS' (fp16 storage 2e-4, and the
    reference's own fp32 noise at batch 2: up to 2.4e-3, see the fp64 test below, which holds the same tensors at 2e-5)"""
    tm.REPORT.clear()
    try:
        tm.run_g_step_and_d_step(name, tm.EXACT)
    finally:
        _dump_report(name)


@pytest.mark.parametrize("name", ["g_class128", "g_class256_sync", "g_uncond_circ"])
def test_exact_steps_match_the_reference_run_in_fp64(exact, name):
    """The fp32 goldens above carry the reference's OWN fp32 noise -- at batch 2 up to 2.4e-3 relative L2 on whole gradient tensors
    (oracle/gen_golden_g64.py prints reference-fp32 against reference-fp64 per case; stored as `ref_fp32_noise`) -- so they cannot
    hold anything tighter than that.  Against the reference executed in FLOAT64 (tests/golden/g64_*.npz) the EXACT build is held
    at: logits and losses 5e-6, every parameter's gradient norm 2e-4, the stored full gradient tensors relative L2 6e-4 and at
    most a fifth (generator) of the reference's own fp32-against-fp64 deviation."""
    g = load_golden(name)
    g64 = load_golden("g64_" + name[2:])
    gan, args, G, D = tm.build(g)
    G.to(DEV).train()
    D.to(DEV).train()
    crit = gan.GANLoss("hinge")
    B, R = int(g["B"]), int(g["R"])
    z, c, x_tex, x_alpha, x_mesh = [t.to(DEV) for t in tm.make_inputs(int(g["seed"]), B, R, 200)]
    if not args.conditional_class:
        c = None
    w, nd = tm.d_weight(args), args.num_discriminators
    rel = lambda got, want: float(np.abs(got.detach().cpu().numpy().astype(np.float64) - want).max() / max(1.0, np.abs(want).max()))

    def grads(module, prefix, keys_k, norms_k):
        named = dict(module.named_parameters())
        worst_n, worst_l2 = 0.0, 0.0
        for k, want in zip(g64[keys_k], g64[norms_k]):
            if want > 1e-4 * g64[norms_k].max():
                worst_n = max(worst_n, abs(float(named[str(k)].grad.norm()) / float(want) - 1))
        for k in g64:
            if k.startswith(prefix):
                a = named[k[len(prefix):]].grad.detach().cpu().flatten().double()
                b = torch.from_numpy(g64[k]).flatten().double()
                if b.norm() > 0:
                    worst_l2 = max(worst_l2, float((a - b).norm() / b.norm()))
        return worst_n, worst_l2

    pred_tex, pred_mesh = G(z, c, None)
    ts = int(g64["tex_stride"])
    assert rel(pred_tex[:, :, ::ts, ::ts], g64["pred_tex"].astype(np.float64)) < 5e-6
    disc, mask = D(torch.cat((pred_tex * x_alpha, x_alpha), dim=1), pred_mesh, c, None)
    e_logit = max(rel(a, g64[f"d{i + 1}"]) for i, a in enumerate(disc))
    loss_g = crit(disc, True, for_discriminator=False, mask=mask if args.mask_output else None, weight=w)
    e_loss = rel(loss_g, g64["loss_g"])
    loss_g.mean().backward()
    n_g, l2_g = grads(G, "gradG:", "gnorm_G_keys", "gnorm_G")
    G.zero_grad()
    D.zero_grad()
    with torch.no_grad():
        ft, fm = G(z, c, None)
        xc = torch.cat((torch.cat((ft * x_alpha, x_alpha), 1), torch.cat((x_tex, x_alpha), 1)), 0)
        cc = torch.cat((c, c), 0) if c is not None else None
        mc = torch.cat((fm, x_mesh), 0)
    disc2, mask2 = D(xc, mc, cc, None)
    e_logit2 = max(rel(a, g64[f"dd{i + 1}"]) for i, a in enumerate(disc2))
    fake, real = [t[:B] for t in disc2], [t[B:] for t in disc2]
    mfake = [t[:B] for t in mask2] if args.mask_output else None
    mreal = [t[B:] for t in mask2] if args.mask_output else None
    loss_fake = crit(fake, False, for_discriminator=True, mask=mfake, weight=w)
    loss_real = crit(real, True, for_discriminator=True, mask=mreal, weight=w)
    e_loss2 = max(rel(loss_fake, g64["loss_fake"]), rel(loss_real, g64["loss_real"]))
    (loss_fake + loss_real).mean().backward()
    n_d, l2_d = grads(D, "gradD:", "gnorm_D_keys", "gnorm_D")
    tm.REPORT.clear()
    tm._note("vs fp64 reference: logits G / D step, losses G / D step", e_logit, e_logit2, e_loss, e_loss2)
    tm._note("vs fp64 reference: worst gradient-norm ratio - 1 (G, D), worst full-tensor rel L2 (G, D)", n_g, n_d, l2_g, l2_d)
    tm._note("reference fp32 vs fp64 gradient rel L2 (median, max) G | D", *g64["ref_fp32_noise"].flatten())
    _dump_report("g64_" + name[2:])
    assert max(e_logit, e_logit2, e_loss, e_loss2) < 5e-6, (e_logit, e_logit2, e_loss, e_loss2)
    # measured (MI355X, round 5): 128^2 cases 5e-7 / 6e-6 on the full tensors, 6e-5 on the worst norm; the 256^2 case -- whose
    # backward is conditioned badly enough for the reference's own fp32 run to sit at 2.4e-3 -- 1e-4 (G) / 3e-4 (D): the
    # activations between the layers are still fp32 here, only the sums are fp64
    assert max(n_g, n_d) < 2e-4 and max(l2_g, l2_d) < 6e-4, (n_g, n_d, l2_g, l2_d)
    # ... and never worse than the reference's own fp32 run is against its fp64 run (generator: at least 5x better)
    noise = g64["ref_fp32_noise"]
    assert l2_g < 0.2 * noise[0][1] + 2e-5 and l2_d < 1.0 * min(noise[1][1], 1.0) + 2e-5, (l2_g, l2_d, noise)


@pytest.mark.timeout(1200)
def test_exact_trainer_four_iterations_match_reference(exact):
    """GanTrainer.iteration x4 (G, D, D, G) incl. Adam(0, 0.9) and the running-average generator (main.py:431-447,588-589,691-723):
    parameter deltas cosine >= 0.9999, Adam's second moments <= 1e-3"""
    tm.REPORT.clear()
    try:
        tm.run_trainer_four_iterations(tm.EXACT)
    finally:
        _dump_report("g_train4")
