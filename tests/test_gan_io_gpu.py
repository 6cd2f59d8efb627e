"""GPU: the input / output glue kernels of the GAN stacks (csrc/gan_io.hip) against the torch op sequences of the reference
they replace (main.py:493,503-507; gan.py:79-99,192-211,406-419; rendering/utils.py:15-26; utils/losses.py:49-120)."""
import argparse
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mods():
    return importlib.import_module("2dimageto3dmodel_amd.gan_ops"), importlib.import_module("2dimageto3dmodel_amd.gan")


@pytest.mark.parametrize("with_real", [False, True])
def test_mask_cat(with_real):
    G, _ = _mods()
    g = torch.Generator().manual_seed(1)
    N, R = 3, 32
    fake = torch.randn(N, 3, R, R, generator=g)
    real = torch.randn(N, 3, R, R, generator=g)
    alpha = (torch.rand(N, 1, R, R, generator=g) > 0.4).float() * torch.rand(N, 1, R, R, generator=g)
    fr = fake.clone().requires_grad_()
    want = torch.cat((fr * alpha, alpha), 1)
    if with_real:
        want = torch.cat((want, torch.cat((real, alpha), 1)), 0)
    wt = torch.randn(want.shape, generator=g)
    (want * wt).sum().backward()
    fd = fake.to(DEV).requires_grad_()
    X = G.mask_cat(fd, alpha.to(DEV), real.to(DEV) if with_real else None)
    assert X.grad_fn.__class__.__name__ == "MaskCatFnBackward"
    assert torch.equal(X.cpu(), want.detach())
    (X * wt.to(DEV)).sum().backward()
    assert torch.equal(fd.grad.cpu(), fr.grad)


def _ref_disc_inputs(x, mesh, specs):
    """the reference's op sequence per member (gan.py:79-99, 192-211) in fp32 torch"""
    hs, masks = [], []
    for f, has_extra, pos, cp, g in specs:
        t = F.avg_pool2d(x, f) if f > 1 else x
        parts = [t] + ([mesh] if has_extra else []) + ([pos.unsqueeze(0).expand(x.shape[0], -1, -1, -1)] if pos is not None else [])
        t = torch.cat(parts, 1)
        masks.append(F.avg_pool2d(t[:, 3:4], g) if g else None)
        t = t.permute(0, 2, 3, 1)
        hs.append(F.pad(t, (0, cp - t.shape[3])))
    return hs, masks


@pytest.mark.parametrize("R,nd,stride_first", [(256, 2, False), (128, 3, False), (512, 3, True), (64, 2, False)])
def test_disc_inputs_all_members(R, nd, stride_first):
    G, gan = _mods()
    g = torch.Generator().manual_seed(R)
    M = 3
    x = torch.randn(M, 4, R, R, generator=g)
    x[:, 3] = (torch.rand(M, R, R, generator=g) > 0.4).float() * (0.5 + 0.5 * torch.rand(M, R, R, generator=g))   # fractional alpha
    mesh = 0.1 * torch.randn(M, 3, 32, 32, generator=g)
    pos = lambda n: torch.tensor(gan.positional_encoding(n, n), dtype=torch.float32)
    specs = [(1, False, pos(R), 8, 16 if stride_first else 8), (R // 32, True, pos(32), 16, 4)]
    if nd == 3:
        specs.append((4, False, pos(R // 4), 8, 8))
    xr, mr = x.clone().requires_grad_(), mesh.clone().requires_grad_()
    hs_r, masks_r = _ref_disc_inputs(xr, mr, specs)
    wts = [torch.randn(h.shape, generator=g).bfloat16().float() for h in hs_r]
    sum((h * w).sum() for h, w in zip(hs_r, wts)).backward()

    xd, md = x.to(DEV).requires_grad_(), mesh.to(DEV).requires_grad_()
    dspecs = [(f, e, p.to(DEV), cp, gg) for f, e, p, cp, gg in specs]
    assert G.disc_inputs_ok(xd, md, dspecs)
    hs, masks = G.disc_inputs(xd, md, dspecs)
    for h, hr, mk, mkr in zip(hs, hs_r, masks, masks_r):
        assert h.dtype == torch.bfloat16 and tuple(h.shape) == tuple(hr.shape)
        assert torch.equal(h.cpu(), hr.detach().bfloat16())            # same fp32 value, one bf16 rounding
        assert (mk.cpu() - mkr.detach()).abs().max().item() < 1e-6
    sum((h.float() * w.to(DEV)).sum() for h, w in zip(hs, wts)).backward()
    assert (xd.grad.cpu() - xr.grad).abs().max().item() < 1e-5 * max(1.0, xr.grad.abs().max().item())
    assert torch.equal(md.grad.cpu(), mr.grad)
    # unsupported shapes are refused (callers fall back to torch ops)
    assert not G.disc_inputs_ok(torch.zeros(1, 4, 40, 40, device=DEV), None, [(1, False, None, 8, 8)])
    assert not G.disc_inputs_ok(torch.zeros(1, 4, 64, 64), None, [(1, False, None, 8, 8)])


@pytest.mark.parametrize("R,nd,stride_first,with_real", [(256, 2, False, True), (256, 2, False, False), (128, 3, False, False),
                                                         (512, 3, True, True), (64, 2, False, False)])
def test_disc_inputs_from_parts_is_bit_identical_to_the_two_stage_form(R, nd, stride_first, with_real):
    """MaskedInput (ModelWrapper.forward's cat(fake * alpha, alpha) [; cat(real, alpha)] assembled inside the discriminators'
    loaders, never written) against mask_cat -> disc_inputs: every packed tensor, every mask and -- G step form -- dfake and the
    mesh-map gradient carry the same bits"""
    G, gan = _mods()
    g = torch.Generator().manual_seed(7 * R + nd)
    N = 3
    fake = torch.randn(N, 3, R, R, generator=g).to(DEV)
    real = torch.randn(N, 3, R, R, generator=g).to(DEV) if with_real else None
    alpha = ((torch.rand(N, 1, R, R, generator=g) > 0.4).float() * (0.5 + 0.5 * torch.rand(N, 1, R, R, generator=g))).to(DEV)
    M = 2 * N if with_real else N
    mesh = (0.1 * torch.randn(M, 3, 32, 32, generator=g)).to(DEV)
    pos = lambda n: torch.tensor(gan.positional_encoding(n, n), dtype=torch.float32, device=DEV)
    specs = [(1, False, pos(R), 8, 16 if stride_first else 8), (R // 32, True, pos(32), 16, 4)]
    if nd == 3:
        specs.append((4, False, pos(R // 4), 8, 8))
    grad = not with_real           # (the D step's generator output carries no graph: main.py:497-503 runs it under no_grad)
    f1, m1 = fake.clone().requires_grad_(grad), mesh.clone().requires_grad_(grad)
    f2, m2 = fake.clone().requires_grad_(grad), mesh.clone().requires_grad_(grad)
    X = G.mask_cat(f1, alpha, real)
    assert G.disc_inputs_ok(X, m1, specs)
    hs1, mk1 = G.disc_inputs(X, m1, specs)
    lazy = G.MaskedInput(f2, alpha, real)
    assert tuple(lazy.shape) == tuple(X.shape) and G.disc_inputs_ok(lazy, m2, specs)
    hs2, mk2 = G.disc_inputs(lazy, m2, specs)
    assert hs2[0].grad_fn is None or hs2[0].grad_fn.__class__.__name__ == "DiscPartsFnBackward"
    for a, b in zip(hs1 + mk1, hs2 + mk2):
        assert torch.equal(a, b)
    if grad:
        wts = [torch.randn(h.shape, generator=g).to(DEV) for h in hs1]
        sum((h.float() * w).sum() for h, w in zip(hs1, wts)).backward()
        sum((h.float() * w).sum() for h, w in zip(hs2, wts)).backward()
        assert torch.equal(f1.grad, f2.grad) and f1.grad.abs().max().item() > 0
        assert torch.equal(m1.grad, m2.grad)
    else:
        # a gradient into the fake half of a [fake; real] batch is not a case the loaders take: refused, the caller materialises
        assert not G.disc_inputs_ok(G.MaskedInput(fake.clone().requires_grad_(), alpha, real), mesh, specs)
    # shapes the loaders do not take fall back through materialize()
    odd = G.MaskedInput(torch.zeros(1, 3, 40, 40, device=DEV), torch.zeros(1, 1, 40, 40, device=DEV))
    assert not G.disc_inputs_ok(odd, None, [(1, False, None, 8, 8)]) and tuple(odd.materialize().shape) == (1, 4, 40, 40)


@pytest.mark.parametrize("flags,C", [(1 | 4, 3), (2 | 4, 3), (1, 3), (2, 3), (0, 2)])
def test_head_tail_matches_reference_ops(flags, C):
    """tanh_ / adjust_poles / symmetrize_texture (gan.py:407-419) and their adjoint in the conv's dy layout"""
    G, gan = _mods()
    from ctypes import c_void_p  # noqa: F401
    lib = importlib.import_module("2dimageto3dmodel_amd._lib")
    g = torch.Generator().manual_seed(flags * 7 + C)
    N, H, W = 3, 32, 16
    y = torch.randn(N, C, H, W, generator=g)
    yr = y.clone().requires_grad_()
    t = yr
    if flags & 1:
        t = torch.tanh(t)
    if flags & 2:
        t = gan.adjust_poles(t)
    if flags & 4:
        t = gan.symmetrize_texture(t)
    wt = torch.randn(t.shape, generator=g)
    (t * wt).sum().backward()
    yd = y.to(DEV)
    out = torch.empty(t.shape, device=DEV)
    lib.launch("head_tail_fwd", lib.ptr(yd), lib.ptr(out), N, C, H, W, flags, lib.stream())
    assert (out.cpu() - t.detach()).abs().max().item() < 2e-6
    gbuf = torch.empty((N, H, W, 8), dtype=torch.bfloat16, device=DEV)
    db = torch.empty(C, device=DEV)
    ws = torch.empty(lib.HEAD_TAIL_WS_FLOATS, device=DEV)
    lib.launch("head_tail_bwd", lib.ptr(wt.to(DEV)), lib.ptr(out), lib.ptr(gbuf), lib.ptr(db), lib.ptr(ws), N, C, H, W, flags,
               lib.stream())
    want = yr.grad.permute(0, 2, 3, 1)
    got = gbuf.float().cpu()
    assert (got[..., :C] - want).abs().max().item() < 1e-2 * want.abs().max().item()      # bf16 storage
    assert got[..., C:].abs().max().item() == 0.0
    assert (db.cpu() - yr.grad.sum((0, 2, 3))).abs().max().item() < 1e-4 * max(1.0, yr.grad.sum((0, 2, 3)).abs().max().item())


@pytest.mark.parametrize("symmetric", [True, False])
def test_head_conv_with_fused_block_activation(symmetric):
    """blk6 -> LeakyReLU -> conv_final -> tanh -> symmetrize of Generator.forward (gan.py:405-417): the fused path
    (CbnActFn out_slope + HeadConvFn) against torch fp32 autograd on the same bf16-rounded operands"""
    G, gan = _mods()
    C = importlib.import_module("2dimageto3dmodel_amd.conv")
    g = torch.Generator().manual_seed(5)
    N, H, W, CH = 2, 32, 16, 64
    mode = C.PAD_REPLICATE if symmetric else C.PAD_CIRCULAR
    conv = gan.Conv2d(CH, 3, 5, pad_h=2, pad_w=2, pad_w_mode=mode).to(DEV)
    with torch.no_grad():
        conv.weight.mul_(3.0)
    x = (torch.randn(N, H, W, CH, generator=g) * 1.2).bfloat16()
    gamma, beta = 0.2 * torch.randn(N, CH, generator=g), 0.2 * torch.randn(N, CH, generator=g)
    res = torch.randn(N, H, W, CH, generator=g).bfloat16()
    # ---- torch fp32 reference
    xr, gr, br = x.float().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    wr = conv.weight.detach().cpu().bfloat16().float().requires_grad_()
    b_r = conv.bias.detach().cpu().clone().requires_grad_()
    mean, var = xr.mean((0, 1, 2)), xr.var((0, 1, 2), unbiased=False)
    hcb = F.leaky_relu((xr - mean) * torch.rsqrt(var + 1e-5) * (1 + gr[:, None, None, :]) + br[:, None, None, :], 0.2) + res.float()
    act = F.leaky_relu(hcb, 0.2)
    t = act.detach().bfloat16().float() + (act - act.detach())   # value: the bf16 activation; gradient: straight through
    tn = t.permute(0, 3, 1, 2)
    tp = F.pad(tn, (2, 2, 0, 0), mode="replicate") if symmetric else torch.cat((tn[..., -2:], tn, tn[..., :2]), 3)
    o = torch.tanh(F.conv2d(tp, wr, b_r, padding=(2, 0)))
    if symmetric:
        o = gan.symmetrize_texture(o)
    wt = torch.randn(o.shape, generator=g)
    (o * wt).sum().backward()
    # ---- fused path
    bn = G.BatchNorm2d(CH).to(DEV)
    xd, gd, bd = x.to(DEV).requires_grad_(), gamma.to(DEV).requires_grad_(), beta.to(DEV).requires_grad_()
    h = bn(xd, gd, bd, 0.2, res.to(DEV), out_slope=0.2)
    out = G.head_conv(h, conv, G.HT_TANH | (G.HT_SYMM if symmetric else 0), in_slope=0.2)
    assert out.dtype == torch.float32 and tuple(out.shape) == tuple(o.shape)
    assert (out.cpu() - o.detach()).abs().max().item() < 2e-2
    (out * wt.to(DEV)).sum().backward()

    def rel(a, b):
        return (a.float().cpu() - b).abs().max().item() / b.abs().max().item()

    assert rel(conv.weight.grad, wr.grad) < 3e-2
    assert rel(conv.bias.grad, b_r.grad) < 3e-2      # (sums tanh' over every pixel: the bf16 conv input moves its argument)
    # d/dx: a LeakyReLU whose argument rounds across zero in bf16 flips its derivative (1 <-> 0.2) on isolated elements,
    # so the comparison is robust instead of max-norm: direction, and the share of elements that are off
    a, b = xd.grad.float().cpu().flatten().double(), xr.grad.flatten().double()
    assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.999
    assert float(((a - b).abs() > 3e-2 * b.abs().max()).double().mean()) < 2e-3
    assert rel(gd.grad, gr.grad) < 8e-2 and rel(bd.grad, br.grad) < 8e-2


def _hinge_ref(gan, preds, masks, weights, B):
    crit = gan.GANLoss("hinge")
    fake, real = [p[:B] for p in preds], [p[B:] for p in preds]
    mf = None if masks is None else [None if m is None else m[:B] for m in masks]
    mr = None if masks is None else [None if m is None else m[B:] for m in masks]
    return crit(fake, False, True, mf, weights), crit(real, True, True, mr, weights), crit(preds, True, False, masks, weights)


@pytest.mark.parametrize("K,masked,weighted", [(2, True, False), (3, True, True), (2, False, False), (1, True, True)])
def test_fused_hinge_matches_ganloss(K, masked, weighted):
    """GANLoss('hinge') (utils/losses.py) on CPU tensors = the torch path pinned by test_ganloss_modes_match_reference,
    against the fused kernels on the same values: discriminator losses on a [fake; real] batch, generator loss, gradients"""
    G, gan = _mods()
    g = torch.Generator().manual_seed(K * 10 + masked)
    B = 3
    shapes = [(2 * B, 1, 16, 16), (2 * B, 1, 8, 8), (2 * B, 1, 4, 4)][:K]
    preds = [torch.randn(s, generator=g) * 1.5 for s in shapes]
    masks = [torch.rand(s, generator=g) for s in shapes] if masked else None
    weights = [2, 1, 1][:K] if weighted else None
    pr = [p.clone().requires_grad_() for p in preds]
    lf_r, lr_r, lg_r = _hinge_ref(gan, pr, masks, weights, B)
    (1.3 * lf_r + 0.7 * lr_r).sum().backward()
    gd_ref = [p.grad.clone() for p in pr]
    for p in pr:
        p.grad = None
    lg_r.sum().backward()
    crit = gan.GANLoss("hinge")
    pd = [p.to(DEV).requires_grad_() for p in preds]
    md = None if masks is None else [m.to(DEV) for m in masks]
    lf, lr = crit.d_losses(pd, md, weights)
    assert lf.grad_fn.__class__.__name__ == "HingeLossFnBackward" and tuple(lf.shape) == tuple(lf_r.shape)
    assert abs(lf.item() - lf_r.item()) < 1e-5 and abs(lr.item() - lr_r.item()) < 1e-5
    (1.3 * lf + 0.7 * lr).sum().backward()
    for a, b in zip(pd, gd_ref):
        assert (a.grad.cpu() - b).abs().max().item() < 1e-6
        a.grad = None
    lg = crit(pd, True, for_discriminator=False, mask=md, weight=weights)
    assert abs(lg.item() - lg_r.item()) < 1e-5
    lg.sum().backward()
    for a, b in zip(pd, pr):
        assert (a.grad.cpu() - b.grad).abs().max().item() < 1e-6
    # whole-batch calls with one target (the two-call form of main.py:518-519 on pre-split tensors)
    half = [p.detach()[:B].contiguous() for p in pd]
    mh = None if md is None else [m[:B].contiguous() for m in md]
    assert abs(crit(half, False, True, mh, weights).item() - lf_r.item()) < 1e-5
