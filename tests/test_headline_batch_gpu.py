"""GPU: parity AT THE BENCHMARKED BATCH (BASELINE.json headline: batch 64 per GPU, 2048-point clouds -> 128^2, GAN at 256^2).

The other GPU tests compare at batch 2..8, where the persistent grids run a handful of workgroups; here the kernels run
256-CU-wide exactly as bench.py launches them -- full persistent grids, the XCD-contiguous workgroup ids
(csrc/conv_dma.h xcd_contiguous_id), tensors beyond 1 GiB (D.conv1's output at batch 128) -- and a SAMPLE of the batch is
checked against the oracle: the projection against oracle/p_oracle.c on four clouds of the 64, every heavy conv layer of
the discriminators (batch 128 = fake + real halves of a D step) and of the generator's high-resolution blocks (batch 64)
against fp32 torch-CPU on the first and the LAST image of the batch (the last image sits at the largest offsets).
Weight gradients sum over the batch: dy is non-zero on the two sampled images only, so the CPU reference needs those two
images while the kernel still walks every tile of the full batch."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_conv_gpu import ref_conv

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LRELU = 0.2


@pytest.mark.timeout(900)
@pytest.mark.parametrize("B,N,S,pick", [(64, 2048, 128, [0, 21, 42, 63]),     # the headline batch
                                        (16, 4096, 128, [0, 5, 10, 15])])      # BASELINE configs[3] per-GPU shape: 16 clouds x 4096 points
def test_projection_batch64_sampled_vs_oracle(pkg, B, N, S, pick):
    from oracle import p_oracle as po
    rs = np.random.RandomState(B)
    pc = ((rs.rand(B, N, 3) - 0.5) * 0.8).astype(np.float32)
    q = rs.randn(B, 4).astype(np.float32)
    sc = (1 / (1 + np.exp(-rs.randn(B, 1)))).astype(np.float32)
    mask = (rs.rand(B, 2 * S, 2 * S) > 0.5).astype(np.float32)
    tpc = torch.from_numpy(pc).to(DEV).requires_grad_()
    tq = torch.from_numpy(q).to(DEV).requires_grad_()
    tsc = torch.from_numpy(sc).to(DEV).requires_grad_()
    proj = pkg.EffectiveLossFunction(voxel_size=S).to(DEV)(tpc, tq, tsc)
    loss = pkg.SupervisedLoss()(proj, torch.from_numpy(mask).to(DEV))["full_loss"]
    loss.backward()
    torch.cuda.synchronize()
    taps = po.taps(3.0, 21, True)
    got = proj.detach().cpu().numpy()
    for i in pick:
        p_o = po.forward(pc[i:i + 1], q[i:i + 1], sc[i:i + 1], S, taps)
        assert np.abs(got[i:i + 1] / p_o - 1).max() < 2e-5, i                                    # silhouette, per pixel
        # per-cloud gradient: full_loss = sum_i SSE_i / (2B) (models/supervised_part.py:68-72), so cloud i's gradient in the batch
        # of B is its single-cloud gradient (B = 1) times 1 / B
        dproj_o = po.sup_loss_bwd(p_o, mask[i:i + 1])
        dp_o, dq_o, ds_o, _ = po.backward(pc[i:i + 1], q[i:i + 1], sc[i:i + 1], dproj_o, S, taps)
        scale = 1.0 / B
        assert np.abs(tpc.grad[i].cpu().numpy() - scale * dp_o[0]).max() < 1e-3 * np.abs(scale * dp_o[0]).max(), i
        assert np.abs(tq.grad[i].cpu().numpy() - scale * dq_o[0]).max() < 1e-3 * np.abs(scale * dq_o[0]).max(), i
        assert np.abs(tsc.grad[i].cpu().numpy() - scale * ds_o[0]).max() < 1e-3 * np.abs(scale * ds_o[0]).max(), i
    # the four clouds' own loss terms through the product's loss on exactly those clouds
    sub = pkg.SupervisedLoss()(proj[pick].detach(), torch.from_numpy(mask[pick]).to(DEV))["full_loss"].item()
    ref = po.sup_loss(np.concatenate([po.forward(pc[i:i + 1], q[i:i + 1], sc[i:i + 1], S, taps) for i in pick]), mask[pick])
    assert abs(sub / ref - 1) < 1e-5, (sub, ref)


def _layer(conv, N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups, seed, slope, bits_in=None, want_dgrad=True, x=None):
    """one conv layer at the full batch: forward (+ LeakyReLU epilogue and sign bits where the product path emits them),
    dgrad (with the producer's bit masks where the product path uses them), wgrad; first and last image against torch-CPU"""
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    g = torch.Generator(device=DEV).manual_seed(seed)
    if x is None:
        x = torch.randn((N, H, W, Cin), generator=g, device=DEV).bfloat16()
    w = (torch.randn((Cout, Cin, k, k), generator=g, device=DEV) / (Cin * k * k) ** 0.5).bfloat16().float()
    b = torch.randn((Cout,), generator=g, device=DEV)
    wf, wd = conv.weight_prep(d, w)
    pick = [0, N - 1]
    xs = x[pick].float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_()
    wr = w.cpu().clone().requires_grad_()
    y_ref = ref_conv(xs, wr, b.cpu(), stride, ph, pw, mode, ups)
    bits = None
    if slope != 1.0 and conv.maskbits_ok(d, 0):
        y, bits = conv.conv_fwd(d, x, wf, b, slope=slope, emit_bits=True)
    else:
        y = conv.conv_fwd(d, x, wf, b, slope=slope)
    act_ref = F.leaky_relu(y_ref.detach(), slope) if slope != 1.0 else y_ref.detach()
    got = y[pick].float().cpu().permute(0, 3, 1, 2)
    err = (got - act_ref).abs().max().item() / act_ref.abs().max().item()
    assert err < 6e-3, ("fwd", conv.tag(d), err)
    Ho, Wo = y.shape[1], y.shape[2]
    # gradients: dy random on the two sampled images, ZERO elsewhere (see the module docstring)
    dy = torch.zeros((N, Ho, Wo, Cout), dtype=torch.bfloat16, device=DEV)
    dy[pick] = torch.randn((2, Ho, Wo, Cout), generator=g, device=DEV).bfloat16()
    y_ref.backward(dy[pick].float().cpu().permute(0, 3, 1, 2))
    if want_dgrad:
        if bits_in is not None and conv.maskbits_ok(d, 1):
            dx = conv.conv_dgrad(d, dy, wd, mask_bits=bits_in, mask_slope=LRELU)
            want = xs.grad * torch.where(xs.detach() > 0, 1.0, LRELU)
        else:
            dx = conv.conv_dgrad(d, dy, wd)
            want = xs.grad
        gotx = dx[pick].float().cpu().permute(0, 3, 1, 2)
        errg = (gotx - want).abs().max().item() / want.abs().max().item()
        assert errg < 1.2e-2, ("dgrad", conv.tag(d), errg)
        mid = dx[N // 2].float().abs().max().item()      # an image whose dy is zero: its gradient must be exactly zero
        assert mid == 0.0, ("dgrad of an untouched image", conv.tag(d), mid)
    db = torch.empty(Cout, device=DEV) if conv.wgrad_fuses_dbias(d) else None
    dw = conv.conv_wgrad(d, x, dy, dbias=db).cpu()
    errw = (dw - wr.grad).abs().max().item() / wr.grad.abs().max().item()
    assert errw < 2e-4, ("wgrad", conv.tag(d), errw)
    if db is not None:
        wantb = dy[pick].float().sum((0, 1, 2)).cpu()
        assert (db.cpu() - wantb).abs().max().item() < 1e-3 * max(1.0, wantb.abs().max().item()), ("dbias", conv.tag(d))
    return y, bits


@pytest.mark.timeout(900)
def test_discriminator_convs_batch128_sampled_vs_torch_cpu(pkg):
    """TextureDiscriminator.conv1 .. conv4 (models/gan.py:163-177) at 256^2 on the 128-image batch of a D step, chained as the
    product runs them: conv + LeakyReLU epilogue + sign bits forward, the consumer's dgrad applying the producer's bits"""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N = 128
    y1, b1 = _layer(conv, N, 256, 256, 8, 64, 5, 1, 2, 2, 2, 0, 1, LRELU, want_dgrad=False)     # conv1 (dgrad: G step, batch 64 below)
    assert y1.numel() * 2 >= (1 << 30)                                                           # 1 GiB: the large-offset paths
    y2, b2 = _layer(conv, N, 256, 256, 64, 128, 4, 2, 1, 1, 2, 0, 2, LRELU, bits_in=b1, x=y1)
    del y1, b1
    y3, b3 = _layer(conv, N, 128, 128, 128, 256, 4, 2, 1, 1, 2, 0, 3, LRELU, bits_in=b2, x=y2)
    del y2, b2
    _layer(conv, N, 64, 64, 256, 512, 4, 2, 1, 1, 2, 0, 4, LRELU, bits_in=b3, x=y3)
    del y3, b3
    torch.cuda.empty_cache()
    _layer(conv, 64, 256, 256, 8, 64, 5, 1, 2, 2, 2, 0, 5, LRELU)                                 # conv1 in the G step: with its dgrad


@pytest.mark.timeout(900)
@pytest.mark.parametrize("case", [
    (128, 256, 256, 64, 128, 4, 2, 1, 1, 2, 0),    # D.conv2, the D step's 128 images: 64 partial rows per (co, ci, class) block
    (128, 64, 64, 256, 512, 4, 2, 1, 1, 2, 0),     # D.conv4: 4 rows
    (64, 128, 64, 128, 64, 3, 1, 1, 1, 1, 1),      # G.blk6.conv1 (sub-pixel form): 64 rows of the 16-entry effective gradient
    (64, 64, 32, 128, 128, 3, 1, 1, 1, 1, 1),      # G.blk5.conv1: 16 rows
])
def test_partial_row_weight_gradients_at_the_timed_batch_against_the_exact_integer_sums(pkg, case):
    """(round 6) at the benchmark's own sizes, on DENSE random gradients: the ordered sum of per-workgroup partial rows
    (m355_conv2d_wgrad_ws, what conv_wgrad takes for these layers in every mode) against the same kernel's deterministic form
    (m355_conv2d_wgrad_det: the partial tiles added as exact integers, rounded once) -- only the fp32 additions of the rows lie
    between the two, so they agree to fp32 rounding of the sum; the fused bias gradient likewise; two launches give the same bits"""
    import ctypes
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    pl = conv.plan(d)
    if not pl.wgrad_ws_ordered:
        pytest.skip("partial rows switched off (M355_WGRAD_HALO_PART / M355_WGRAD_UP_PART)")
    g = torch.Generator(device=DEV).manual_seed(41)
    ho, wo = conv.out_hw(d)
    x = torch.randn((N, H, W, Cin), generator=g, device=DEV).bfloat16()
    dy = torch.randn((N, ho, wo, Cout), generator=g, device=DEV).bfloat16()
    db1, db2 = torch.empty(Cout, device=DEV), torch.empty(Cout, device=DEV)
    dw1 = conv.conv_wgrad(d, x, dy, raw=True, dbias=db1).clone()
    assert conv.lib().m355_last_kernel().decode() == "k_wgrad_halo"
    dw1b = conv.conv_wgrad(d, x, dy, raw=True, dbias=db2)
    assert torch.equal(dw1, dw1b) and torch.equal(db1, db2)
    ws = torch.empty((pl.wgrad_det_ws_bytes,), dtype=torch.uint8, device=DEV)
    dw2 = torch.empty_like(dw1)
    conv.launch("conv2d_wgrad_det", ctypes.byref(d), conv.ptr(x), conv.ptr(dy), conv.ptr(ws), conv.ptr(dw2), conv.ptr(db2), conv.stream())
    torch.cuda.synchronize()
    assert torch.isfinite(dw2).all()
    assert ((dw1 - dw2).abs().max() / dw2.abs().max()).item() < 2e-6
    assert ((db1 - db2).abs().max() / db2.abs().max().clamp_min(1.0)).item() < 2e-6
    del x, dy, ws
    torch.cuda.empty_cache()


@pytest.mark.timeout(900)
def test_generator_blocks_batch64_sampled_vs_torch_cpu(pkg):
    """ResBlockUp conv1 (nearest x2 upsample folded in) / conv2 of blk4, blk5, blk6 (models/gan.py:294-312,386-404; symmetric
    generator: replicate W pad) at batch 64, the 1x1 shortcut of blk6, and the 64 -> 3 head at 256 x 128"""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N = 64
    for seed, (H, W, Cin, Cout, ups) in enumerate(((32, 16, 256, 128, 1), (64, 32, 128, 128, 0),      # blk4
                                                    (64, 32, 128, 128, 1), (128, 64, 128, 128, 0),     # blk5
                                                    (128, 64, 128, 64, 1), (256, 128, 64, 64, 0))):    # blk6
        _layer(conv, N, H, W, Cin, Cout, 3, 1, 1, 1, 1, ups, 10 + seed, 1.0)
    _layer(conv, N, 128, 64, 128, 64, 1, 1, 0, 0, 0, 0, 20, 1.0)                                       # blk6.shortcut
    # conv_final (gan.py:359): fp32 NCHW output, 3 channels, replicate pad
    d = conv.make_desc(N, 256, 128, 64, 3, 5, 5, 1, 2, 2, 1, 0)
    g = torch.Generator(device=DEV).manual_seed(21)
    x = torch.randn((N, 256, 128, 64), generator=g, device=DEV).bfloat16()
    w = (torch.randn((3, 64, 5, 5), generator=g, device=DEV) / 40.0).bfloat16().float()
    b = torch.randn((3,), generator=g, device=DEV)
    wf, wd = conv.weight_prep(d, w)
    y = conv.conv_fwd(d, x, wf, b, out_f32_nchw=True)
    pick = [0, N - 1]
    xs = x[pick].float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_()
    wr = w.cpu().clone().requires_grad_()
    y_ref = ref_conv(xs, wr, b.cpu(), 1, 2, 2, 1, 0)
    assert (y[pick].cpu() - y_ref.detach()).abs().max().item() / y_ref.abs().max().item() < 2e-4
    dy = torch.zeros((N, 256, 128, 8), dtype=torch.bfloat16, device=DEV)
    dy[pick, :, :, :3] = torch.randn((2, 256, 128, 3), generator=g, device=DEV).bfloat16()
    y_ref.backward(dy[pick][..., :3].float().cpu().permute(0, 3, 1, 2))
    dx = conv.conv_dgrad(d, dy, wd)
    assert (dx[pick].float().cpu().permute(0, 3, 1, 2) - xs.grad).abs().max().item() / xs.grad.abs().max().item() < 1.2e-2
    dw = conv.conv_wgrad(d, x, dy).cpu()
    assert (dw - wr.grad).abs().max().item() / wr.grad.abs().max().item() < 2e-4


@pytest.mark.timeout(900)
def test_two_gib_layers_run_as_half_batches_on_the_fast_kernels(pkg):
    """A D step at batch 128 per GPU (N = 256): D.conv1's output / D.conv2's input reach 2 GiB, past the 32-bit byte offsets of the
    specialised kernels.  conv.py then runs the layer on each half of the batch (outputs = slices of one tensor, bit masks and
    batch-norm partial rows likewise, the weight-gradient halves added): same parity bar as at batch 64 -- first image (first
    half), last image (second half), an untouched image's gradient exactly zero -- and on the FAST kernel families."""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N = 256
    y1, b1 = _layer(conv, N, 256, 256, 8, 64, 5, 1, 2, 2, 2, 0, 31, LRELU, want_dgrad=False)
    assert y1.numel() * 2 == (1 << 31) and b1 is not None
    assert conv.lib().m355_last_kernel().decode() in ("k_wgrad_c8", "k_wgrad_part_sum")
    d2 = conv.make_desc(N, 256, 256, 64, 128, 4, 4, 2, 1, 1, 2, 0)
    assert conv._halves(d2) is not None and conv.maskbits_ok(d2, 0) and conv.maskbits_ok(d2, 1) and conv.dgrad_mask_ok(d2)
    y2, b2 = _layer(conv, N, 256, 256, 64, 128, 4, 2, 1, 1, 2, 0, 32, LRELU, bits_in=b1, x=y1)
    assert conv.lib().m355_last_kernel().decode() == "k_wgrad_halo" and b2 is not None
    del y1, b1, y2, b2
    torch.cuda.empty_cache()
    # fused batch-norm statistics across the halves: G.blk6.conv2's shape at N = 512 (2 GiB in, 2 GiB out)
    d = conv.make_desc(512, 256, 128, 64, 64, 3, 3, 1, 1, 1, 1, 0)
    dh = conv._halves(d)[0]
    assert conv.conv_stats_rows(d) == 2 * conv.conv_stats_rows(dh) > 0
    g = torch.Generator(device=DEV).manual_seed(33)
    x = torch.randn((512, 256, 128, 64), generator=g, device=DEV).bfloat16()
    w = (torch.randn((64, 64, 3, 3), generator=g, device=DEV) / 24.0).bfloat16().float()
    wf, _ = conv.weight_prep(d, w, want_dgrad=False)
    y, part = conv.conv_fwd_stats(d, x, wf, None)
    assert conv.lib().m355_last_kernel().decode() == "k_conv_halo" and tuple(part.shape) == (conv.conv_stats_rows(d), 2, 64)
    yh, parth = conv.conv_fwd_stats(dh, x[256:], wf, None)
    assert torch.equal(y[256:], yh) and torch.equal(part[part.shape[0] // 2:], parth)          # the second half IS the half-batch launch
    s = part.double().sum(0).cpu()
    yf = y[::64].double()                                                                      # every 64th image: 8 of 512
    assert abs(float(s[0].sum()) / 512 - float(yf.sum()) / 8) < 0.05 * float(yf.abs().sum()) / 8 + 1.0
    assert abs(float(s[1].sum()) / 512 / (float((yf * yf).sum()) / 8) - 1) < 0.02
