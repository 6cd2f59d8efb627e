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
    assert L.m355_abi_version() == 3
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
    """the nine reference-executed goldens (one G step + one D step each; 128^2 / 256^2 / 512^2, nd 2 / 3, sync / batch /
    instance / no norm, class / colour / text conditioning, no-mask) at the EXACT tolerances: logits and losses 1e-4, gradient
    norms 1e-3, every stored gradient tensor relative L2 <= 1e-3 (they are kept as fp16: 3e-4 of that is storage)"""
    tm.REPORT.clear()
    try:
        tm.run_g_step_and_d_step(name, tm.EXACT)
    finally:
        _dump_report(name)


@pytest.mark.timeout(1200)
def test_exact_trainer_four_iterations_match_reference(exact):
    """GanTrainer.iteration x4 (G, D, D, G) incl. Adam(0, 0.9) and the running-average generator (main.py:431-447,588-589,691-723):
    parameter deltas cosine >= 0.9999, Adam's second moments <= 1e-3"""
    tm.REPORT.clear()
    try:
        tm.run_trainer_four_iterations(tm.EXACT)
    finally:
        _dump_report("g_train4")
