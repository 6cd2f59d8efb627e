"""GPU: the bf16 MFMA conv2d (fwd / dgrad) against torch-CPU fp32 F.conv2d on the same bf16-rounded operands,
with the reference's pads (F.pad replicate gan.py:329, circpad rendering/utils.py:60-64) and nearest x2
upsample (gan.py:319) materialised on the CPU side.  Only the accumulation order differs -> tight tolerance."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref_conv(x_nchw, w, bias, stride, ph, pw, mode, ups):
    if ups:
        x_nchw = F.interpolate(x_nchw, scale_factor=2, mode="nearest")
    if pw:
        if mode == 1:
            x_nchw = F.pad(x_nchw, (pw, pw, 0, 0), mode="replicate")
        elif mode == 2:
            x_nchw = torch.cat((x_nchw[..., -pw:], x_nchw, x_nchw[..., :pw]), dim=3)
        else:
            x_nchw = F.pad(x_nchw, (pw, pw, 0, 0))
    return F.conv2d(x_nchw, w, bias, stride=stride, padding=(ph, 0))


CASES = [
    # N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups
    (2, 8, 4, 64, 64, 3, 1, 1, 1, 1, 0),      # ResBlockUp conv (replicate W pad), tiny
    (2, 8, 4, 64, 128, 3, 1, 1, 1, 1, 1),     # with the x2 upsample folded in
    (1, 16, 8, 128, 64, 1, 1, 0, 0, 0, 0),    # 1x1 shortcut
    (2, 16, 16, 32, 64, 5, 1, 2, 2, 2, 0),    # D conv1-like: 5x5, circular
    (2, 16, 16, 64, 128, 4, 2, 1, 1, 2, 0),   # D conv2-like: 4x4 stride 2, circular
    (1, 8, 8, 64, 3, 5, 1, 2, 2, 1, 0),       # head: Cout=3
    (3, 12, 6, 96, 96, 3, 1, 1, 1, 0, 0),     # zero W pad, Cout not a multiple of 64, M not a multiple of 256
    (2, 16, 16, 8, 64, 5, 1, 2, 2, 2, 0),     # D conv1: 8 input channels (K step spans 4 taps, K=200 padded to 224)
    (2, 16, 16, 16, 64, 4, 2, 1, 1, 2, 0),    # 16 input channels, stride 2
    (2, 8, 8, 40, 32, 3, 1, 1, 1, 1, 1),      # Cin = 40 (multiple of 8 only), upsample
    (3, 20, 12, 128, 256, 3, 1, 1, 1, 1, 1),  # two 128-wide N tiles, M = 2880 (ragged last pixel tile), 2-chunk taps
    (2, 24, 20, 64, 64, 3, 1, 1, 1, 0, 0),    # 256x64 tile, ragged M, zero W pad
    (2, 32, 32, 128, 256, 4, 2, 1, 1, 2, 0),  # D conv3-like, stride-2 dgrad parity classes on the DMA path
    (5, 16, 16, 8, 64, 5, 1, 2, 2, 2, 0),     # D conv1 on the per-chunk-tap path with several pixel tiles
    (2, 16, 8, 256, 128, 3, 1, 1, 1, 1, 1),   # blk4 conv1 shape (upsample + replicate), 4-chunk taps
    (2, 32, 32, 512, 1, 5, 1, 2, 2, 2, 0),    # TextureDiscriminator.conv5: halo kernel, 8 channel chunks, circular
    (2, 40, 24, 64, 3, 5, 1, 2, 2, 1, 0),     # conv_final-like, image not a multiple of the 16x16 tile, replicate
    (3, 8, 8, 256, 1, 5, 1, 2, 2, 2, 0),      # MeshDiscriminator.conv4: image narrower than the tile, circular
    (2, 20, 20, 64, 2, 3, 1, 1, 1, 0, 0),     # 3x3, zero W pad, 2 output channels
    # ---- halo kernel (conv_halo.hip): Wo % 32 == 0, Ho % 8 == 0, Cin % 64 == 0
    (2, 16, 32, 64, 64, 3, 1, 1, 1, 1, 0),    # 3x3 replicate, Cout 64 (4-wave variant), one channel chunk
    (2, 16, 32, 128, 128, 3, 1, 1, 1, 1, 0),  # 3x3 replicate, Cout 128 (8-wave variant), two chunks (halo double buffer)
    (2, 8, 16, 256, 128, 3, 1, 1, 1, 1, 1),   # upsample folded into the halo (6 x 18 stored pixels), 4 chunks
    (1, 16, 16, 128, 64, 3, 1, 1, 1, 1, 1),   # upsample, Cout 64
    (2, 24, 64, 192, 256, 3, 1, 1, 1, 0, 0),  # zero W pad, 3 chunks, two N tiles, two pixel tiles across W
    (2, 32, 64, 128, 256, 4, 2, 1, 1, 2, 0),  # stride-2 dgrad: four 2x2 classes of dy on the halo kernel, circular
    (2, 32, 64, 64, 128, 4, 2, 1, 1, 0, 0),   # stride-2 dgrad with zero W pad, dx has 64 channels (4-wave variant)
    (1, 16, 32, 64, 64, 3, 1, 1, 1, 2, 0),    # 3x3 circular
    # ---- upsample + 3x3 in the SUB-PIXEL form (csrc/conv_mfma.hip `subpixel`: stored W % 32 == 0, H % 8 == 0, channels % 64 == 0):
    # forward = four 2x2 class convs with pre-summed weights, dgrad = the adjoint 4x4 stride-2 conv (+ replicate edge term),
    # wgrad = class kernels into the 16-entry effective gradient + fold.  The reference is built from the ORIGINAL 3x3 weights.
    (2, 16, 32, 128, 64, 3, 1, 1, 1, 1, 1),   # G.blk6.conv1 shape class: 64 output channels = class PAIRS, replicate
    (2, 8, 32, 128, 128, 3, 1, 1, 1, 1, 1),   # G.blk5.conv1 shape class: 8-wave class kernels, replicate, two chunks
    (1, 16, 64, 64, 128, 3, 1, 1, 1, 2, 1),   # circular W pad, two pixel tiles across W, one chunk
    (2, 8, 32, 64, 64, 3, 1, 1, 1, 0, 1),     # zero W pad, class pairs
    (1, 24, 32, 192, 256, 3, 1, 1, 1, 1, 1),  # three chunks, two 128-channel output tiles
    # ---- (round 6) the SMALL upsample + 3x3 stages (upsampled width not a multiple of 32): forward and dgrad in the sub-pixel form on the
    # generic implicit-GEMM kernel (four class launches / the adjoint 4x4 stride-2 conv), weight gradient 9-tap.  The replicate cases
    # are above ((2, 8, 4, 64, 128), (3, 20, 12, 128, 256), (2, 16, 8, 256, 128)); here the other two W pads and 64 output channels
    (2, 8, 4, 64, 64, 3, 1, 1, 1, 2, 1),      # circular, G.blk1 -> blk2 size
    (2, 16, 8, 128, 128, 3, 1, 1, 1, 0, 1),   # zero W pad, two chunks
    (3, 16, 8, 256, 64, 3, 1, 1, 1, 1, 1),    # G.blk3_mesh.conv1 shape class: 64 output channels, replicate
    # ---- odd kernels with stride 2 (the encoder of models/reconstruction.py:53-63): dgrad through the padded even kernel
    (2, 32, 32, 8, 64, 5, 2, 2, 2, 0, 0),     # conv1e: 5x5 s2 p2 on (4 -> 8) channels
    (2, 16, 16, 64, 128, 3, 2, 1, 1, 0, 0),   # conv2e: 3x3 s2 p1
    (2, 8, 8, 512, 64, 3, 2, 1, 1, 0, 0),     # conv5e: 512 -> 64
    # ---- 8-input-channel kernel (conv_small.hip k_conv_c8): D.conv1
    (3, 24, 64, 8, 64, 5, 1, 2, 2, 2, 0),     # circular, several tiles per image
    (2, 16, 32, 8, 128, 5, 1, 2, 2, 0, 0),    # zero W pad, two 64-channel output tiles
    # ---- replicate-padded 5x5 heads whose dgrad runs on k_conv_c8 + the pad-column edge term
    (2, 16, 32, 64, 3, 5, 1, 2, 2, 1, 0),     # conv_final / conv_mesh shape class
    (3, 24, 64, 128, 2, 5, 1, 2, 2, 1, 0),    # two 64-channel tiles of dx, several pixel tiles, 2 output channels
    # ---- 64 -> 1..4 channel 5x5 heads in the scatter form (conv_small.hip k_head5): strips of 28 columns, row segments
    (2, 33, 70, 64, 3, 5, 1, 2, 2, 0, 0),     # zero W pad, three strips (ragged), ragged row segments
    (1, 64, 30, 64, 4, 5, 1, 2, 2, 2, 0),     # circular, 4 output channels, two strips
    (3, 12, 28, 64, 1, 5, 1, 2, 2, 1, 0),     # replicate, one output channel, exactly one strip
]


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_and_dgrad(pkg, case):
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    b = torch.randn(Cout, generator=g)
    xr = x.clone().requires_grad_()
    y_ref = ref_conv(xr, w, b, stride, ph, pw, mode, ups)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    wf, wd = conv.weight_prep(d, w.to(DEV))
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    y = conv.conv_fwd(d, x_nhwc, wf, b.to(DEV))
    assert tuple(y.shape) == (N, y_ref.shape[2], y_ref.shape[3], Cout)
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - y_ref.detach()).abs().max().item() / y_ref.abs().max().item()
    assert err < 6e-3, err  # bf16 rounding of the output (2^-9 relative) dominates
    # bf16 NHWC epilogue with LeakyReLU (the discriminators' conv -> LeakyReLU fusion)
    ylb = conv.conv_fwd(d, x_nhwc, wf, b.to(DEV), slope=0.2).float().cpu().permute(0, 3, 1, 2)
    assert (ylb - F.leaky_relu(y_ref.detach(), 0.2)).abs().max().item() / y_ref.abs().max().item() < 6e-3
    # fp32 NCHW epilogue (used by the heads) has no output rounding
    y32 = conv.conv_fwd(d, x_nhwc, wf, b.to(DEV), out_f32_nchw=True).cpu()
    assert (y32 - y_ref.detach()).abs().max().item() / y_ref.abs().max().item() < 2e-4
    # LeakyReLU epilogue
    yl = conv.conv_fwd(d, x_nhwc, wf, b.to(DEV), out_f32_nchw=True, slope=0.2).cpu()
    assert (yl - F.leaky_relu(y_ref.detach(), 0.2)).abs().max().item() / y_ref.abs().max().item() < 2e-4
    # dgrad
    dy = torch.randn(y_ref.shape, generator=g).bfloat16().float()
    y_ref.backward(dy)
    c32 = conv.dy_channels(Cout)
    dy_nhwc = torch.zeros(N, y_ref.shape[2], y_ref.shape[3], c32)
    dy_nhwc[..., :Cout] = dy.permute(0, 2, 3, 1)
    dx = conv.conv_dgrad(d, dy_nhwc.bfloat16().to(DEV), wd).float().cpu().permute(0, 3, 1, 2)
    errg = (dx - xr.grad).abs().max().item() / xr.grad.abs().max().item()
    assert errg < 1.2e-2, errg  # the padded-frame gradient is rounded to bf16 once before the fold
    # wgrad (fp32 accumulate, fp32 out): autograd of the same graph w.r.t. the weight
    wr = w.clone().requires_grad_()
    ref_conv(x, wr, b, stride, ph, pw, mode, ups).backward(dy)
    dw = conv.conv_wgrad(d, x_nhwc, dy_nhwc.bfloat16().to(DEV)).cpu()
    errw = (dw - wr.grad).abs().max().item() / wr.grad.abs().max().item()
    assert errw < 2e-4, errw
    # bias gradient fused into the wgrad kernel (column sums of dy)
    if conv.wgrad_fuses_dbias(d):
        db = torch.empty(Cout, device=DEV)
        conv.conv_wgrad(d, x_nhwc, dy_nhwc.bfloat16().to(DEV), dbias=db)
        want = dy.sum((0, 2, 3))
        assert (db.cpu() - want).abs().max().item() < 1e-3 * max(1.0, want.abs().max().item())
    # LeakyReLU backward of the producer of x folded into the dgrad epilogue (direct-form layers only)
    import ctypes
    direct = conv.lib().m355_conv2d_dgrad_ws_bytes(ctypes.byref(d)) == 0   # (odd-kernel stride-2 layers go through the fold)
    if mode != 1 and not ups and direct:
        dxm = conv.conv_dgrad(d, dy_nhwc.bfloat16().to(DEV), wd, mask_x=x_nhwc, mask_slope=0.2).float().cpu()
        want = xr.grad * torch.where(x > 0, 1.0, 0.2)
        assert (dxm.permute(0, 3, 1, 2) - want).abs().max().item() / want.abs().max().item() < 1.2e-2


@pytest.mark.parametrize("case", [(3, 24, 64, 8, 64, 5, 1, 2, 2, 2, 0, 4),     # D.conv1: circular, several strips of 28 columns
                                  (2, 40, 40, 8, 64, 5, 1, 2, 2, 0, 0, 3),     # zero W pad, ragged strips and row segments, 3 channels
                                  (2, 16, 32, 8, 64, 5, 1, 2, 2, 2, 0, 1),     # one leading channel
                                  (2, 16, 16, 8, 128, 5, 1, 2, 2, 2, 0, 4)])   # 128 output channels: not the scatter form's shape
def test_dgrad_of_the_leading_input_channels(pkg, case):
    """m355_conv2d_dgrad_lead (round 5): the input gradient of a layer whose trailing input channels are model constants
    (TextureDiscriminator.conv1: image channels + positional planes, models/gan.py:204-213) -- the first `lead` channels of dx equal
    the full dgrad's, whichever kernel runs (the scatter-form head kernel for 64 dy channels, else the general path)"""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups, lead = case
    g = torch.Generator().manual_seed(77)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    xr = torch.zeros(N, Cin, H, W, requires_grad=True)
    y_ref = ref_conv(xr, w, None, stride, ph, pw, mode, ups)
    dy = torch.randn(y_ref.shape, generator=g).bfloat16().float()
    y_ref.backward(dy)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    _, wd = conv.weight_prep(d, w.to(DEV))
    dyd = dy.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    for rep in range(2):
        dx = conv.conv_dgrad(d, dyd, wd, lead=lead)
        assert tuple(dx.shape) == (N, H, W, Cin)
        got = dx.float().cpu().permute(0, 3, 1, 2)[:, :lead]
        want = xr.grad[:, :lead]
        assert (got - want).abs().max().item() / want.abs().max().item() < 1.2e-2
    if Cout == 64:
        assert conv.lib().m355_last_kernel().decode() == "k_head5"
    full = conv.conv_dgrad(d, dyd, wd).float().cpu().permute(0, 3, 1, 2)
    assert (full - xr.grad).abs().max().item() / xr.grad.abs().max().item() < 1.2e-2


@pytest.mark.parametrize("tile", ["64x64", "128x128", "256x128", "256x256"])
@pytest.mark.parametrize("case", [(3, 20, 12, 128, 256, 3, 1, 1, 1, 1, 1), (2, 32, 32, 128, 256, 4, 2, 1, 1, 2, 0),
                                  (2, 16, 16, 64, 128, 4, 2, 1, 1, 2, 0), (3, 12, 6, 96, 96, 3, 1, 1, 1, 0, 0)])
def test_conv_tile_variants(pkg, case, tile, monkeypatch):
    """every workgroup tile of the DMA kernel (the launcher picks by problem size; forced here) on fwd + dgrad"""
    monkeypatch.setenv("M355_CONV_TILE", tile)
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    b = torch.randn(Cout, generator=g)
    xr = x.clone().requires_grad_()
    y_ref = ref_conv(xr, w, b, stride, ph, pw, mode, ups)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    wf, wd = conv.weight_prep(d, w.to(DEV))
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    y = conv.conv_fwd(d, x_nhwc, wf, b.to(DEV), slope=0.2).float().cpu().permute(0, 3, 1, 2)
    assert (y - F.leaky_relu(y_ref.detach(), 0.2)).abs().max().item() / y_ref.abs().max().item() < 6e-3
    dy = torch.randn(y_ref.shape, generator=g).bfloat16().float()
    y_ref.backward(dy)
    c32 = conv.dy_channels(Cout)
    dy_nhwc = torch.zeros(N, y_ref.shape[2], y_ref.shape[3], c32)
    dy_nhwc[..., :Cout] = dy.permute(0, 2, 3, 1)
    dx = conv.conv_dgrad(d, dy_nhwc.bfloat16().to(DEV), wd).float().cpu().permute(0, 3, 1, 2)
    assert (dx - xr.grad).abs().max().item() / xr.grad.abs().max().item() < 1.2e-2


@pytest.mark.parametrize("wgs", ["3", "8"])
@pytest.mark.parametrize("case", [(3, 16, 64, 128, 128, 3, 1, 1, 1, 1, 0), (2, 16, 16, 128, 64, 3, 1, 1, 1, 1, 1),
                                  (2, 32, 64, 128, 256, 4, 2, 1, 1, 2, 0), (3, 24, 32, 64, 64, 3, 1, 1, 1, 2, 0),
                                  (3, 16, 32, 128, 64, 3, 1, 1, 1, 1, 1),     # sub-pixel upsample conv: class pairs
                                  (2, 16, 64, 64, 128, 3, 1, 1, 1, 1, 1)])    # ... 8-wave classes; dgrad = stride-2 forward variant
def test_conv_halo_persistent_tiles(pkg, case, wgs, monkeypatch):
    """k_conv_halo with few persistent workgroups: each walks several pixel tiles (cross-tile halo prefetch, weight
    ring wrap-around, uneven tile counts per workgroup)"""
    monkeypatch.setenv("M355_HALO_WGS", wgs)
    monkeypatch.setenv("M355_HALO_ALL", "1")
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    b = torch.randn(Cout, generator=g)
    xr = x.clone().requires_grad_()
    y_ref = ref_conv(xr, w, b, stride, ph, pw, mode, ups)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    wf, wd = conv.weight_prep(d, w.to(DEV))
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    for _ in range(3):
        y = conv.conv_fwd(d, x_nhwc, wf, b.to(DEV)).float().cpu().permute(0, 3, 1, 2)
        assert (y - y_ref.detach()).abs().max().item() / y_ref.abs().max().item() < 6e-3
    dy = torch.randn(y_ref.shape, generator=g).bfloat16().float()
    y_ref.backward(dy)
    dy_nhwc = torch.zeros(N, y_ref.shape[2], y_ref.shape[3], conv.dy_channels(Cout))
    dy_nhwc[..., :Cout] = dy.permute(0, 2, 3, 1)
    for _ in range(3):
        dx = conv.conv_dgrad(d, dy_nhwc.bfloat16().to(DEV), wd).float().cpu().permute(0, 3, 1, 2)
        assert (dx - xr.grad).abs().max().item() / xr.grad.abs().max().item() < 1.2e-2


@pytest.mark.parametrize("chain", [
    # producer (N,H,W,Cin,Cout,k,s,ph,pw,mode) -> consumer Cout', both of the discriminators' shapes
    ((2, 16, 64, 8, 64, 5, 1, 2, 2, 2), 128),     # D.conv1 (k_conv_c8 writes the bits) -> D.conv2 dgrad (resident-weight halo)
    ((2, 32, 128, 64, 128, 4, 2, 1, 1, 2), 256),  # D.conv2 forward (stride-2 halo, bias + LeakyReLU) -> D.conv3 dgrad
    ((2, 32, 128, 128, 256, 4, 2, 1, 1, 0), 128), # zero pad, 256 channels: two 128-channel tiles of mask words
])
def test_activation_bit_masks(pkg, chain):
    """conv+LeakyReLU forward writing 1 bit per activation, the consumer's dgrad reading them: identical to the dgrad
    that re-reads the bf16 activation as its mask"""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    (N, H, W, Cin, Cout, k, stride, ph, pw, mode), Cout2 = chain
    g = torch.Generator().manual_seed(Cin + Cout)
    d1 = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, 0)
    assert conv.maskbits_ok(d1, 0)
    x = torch.randn(N, H, W, Cin, generator=g).bfloat16().to(DEV)
    w1 = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(DEV)
    b1 = torch.randn(Cout, generator=g).to(DEV)
    wf1, _ = conv.weight_prep(d1, w1, want_dgrad=False)
    y_plain = conv.conv_fwd(d1, x, wf1, b1, slope=0.2)
    y, bits = conv.conv_fwd(d1, x, wf1, b1, slope=0.2, emit_bits=True)
    assert torch.equal(y, y_plain)
    ho, wo = conv.out_hw(d1)
    # the bits are the sign of the pre-activation: as many set bits as positive activations
    popc = sum(((bits >> b) & 1).sum().item() for b in range(32))
    assert popc == (y.float() > 0).sum().item()
    # consumer: 4x4 stride-2 conv on y (the discriminators' next layer)
    d2 = conv.make_desc(N, ho, wo, Cout, Cout2, 4, 4, 2, 1, 1, mode, 0)
    assert conv.maskbits_ok(d2, 1)
    w2 = (torch.randn(Cout2, Cout, 4, 4, generator=g) / (Cout * 16) ** 0.5).to(DEV)
    _, wd2 = conv.weight_prep(d2, w2)
    ho2, wo2 = conv.out_hw(d2)
    dy2 = torch.randn(N, ho2, wo2, conv.dy_channels(Cout2), generator=g).bfloat16().to(DEV)
    want = conv.conv_dgrad(d2, dy2, wd2, mask_x=y, mask_slope=0.2)
    got = conv.conv_dgrad(d2, dy2, wd2, mask_bits=bits, mask_slope=0.2)
    assert torch.equal(got, want)


@pytest.mark.parametrize("variant", ["narrow", "wide", "twin"])
@pytest.mark.parametrize("case", [(2, 32, 64, 128, 256, 4, 2, 1, 1, 2, 0), (3, 16, 64, 64, 128, 4, 2, 1, 1, 0, 0)])
def test_wgrad_halo_class_variants(pkg, case, variant, monkeypatch):
    """the three tilings of the stride-2 class wgrad (8x32 tiles; 4x32 with two dy blocks on one x halo; 4x32 with two
    workgroups per CU = the default) give the same weight and bias gradients"""
    monkeypatch.setenv("M355_WGRAD_HALO_VARIANT", variant)
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    g = torch.Generator().manual_seed(23)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float().requires_grad_()
    y_ref = ref_conv(x, w, None, stride, ph, pw, mode, ups)
    dy = torch.randn(y_ref.shape, generator=g).bfloat16().float()
    y_ref.backward(dy)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    db = torch.empty(Cout, device=DEV)
    dw = conv.conv_wgrad(d, x_nhwc, dy_nhwc, dbias=db).cpu()
    assert (dw - w.grad).abs().max().item() < 2e-4 * w.grad.abs().max().item()
    want = dy.sum((0, 2, 3))
    assert (db.cpu() - want).abs().max().item() < 1e-3 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("wgs", [None, "4"])
@pytest.mark.parametrize("case", [(3, 32, 128, 64, 128, 4, 2, 1, 1, 2, 0), (2, 16, 64, 64, 256, 4, 2, 1, 1, 0, 0),
                                  (1, 64, 64, 64, 128, 4, 2, 1, 1, 2, 0)])
def test_stride2_dgrad_class_pairs(pkg, case, wgs, monkeypatch):
    """stride-2 dgrad with 64 input channels (D.conv2): the class-pair variant of k_conv_halo (two column-parity classes per
    workgroup on one shared dy halo) against torch autograd, against the one-class-per-workgroup launch (same MFMA
    order -> identical bits), and with the fused LeakyReLU backward from the bf16 activation"""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    if wgs:
        monkeypatch.setenv("M355_HALO_WGS", wgs)
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    xr = x.clone().requires_grad_()
    y_ref = ref_conv(xr, w, None, stride, ph, pw, mode, ups)
    dy = torch.randn(y_ref.shape, generator=g).bfloat16().float()
    y_ref.backward(dy)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    _, wd = conv.weight_prep(d, w.to(DEV))
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    for _ in range(2):
        dx = conv.conv_dgrad(d, dy_nhwc, wd)
        assert conv.lib().m355_last_kernel().decode() == "k_conv_halo"
        assert (dx.float().cpu().permute(0, 3, 1, 2) - xr.grad).abs().max().item() / xr.grad.abs().max().item() < 1.2e-2
    dxm = conv.conv_dgrad(d, dy_nhwc, wd, mask_x=x_nhwc, mask_slope=0.2)
    monkeypatch.setenv("M355_NO_HALO_PAIR", "1")
    assert torch.equal(conv.conv_dgrad(d, dy_nhwc, wd), dx)
    assert torch.equal(conv.conv_dgrad(d, dy_nhwc, wd, mask_x=x_nhwc, mask_slope=0.2), dxm)


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 3, 5, 1, 2, 2, 1, 0), (3, 24, 64, 128, 2, 5, 1, 2, 2, 1, 0)])
def test_head_dgrad_with_fused_activation_backward(pkg, case):
    """dgrad of a replicate-padded 5x5 head on k_conv_c8 + edge term, with the LeakyReLU backward of the head's input applied in
    the epilogue (HeadConvFn in_slope): equals the unmasked dgrad times the activation derivative"""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    g = torch.Generator().manual_seed(91)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    xr = x.clone().requires_grad_()
    y_ref = ref_conv(xr, w, None, stride, ph, pw, mode, ups)
    dy = torch.randn(y_ref.shape, generator=g).bfloat16().float()
    y_ref.backward(dy)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    assert conv.dgrad_mask_ok(d)
    _, wd = conv.weight_prep(d, w.to(DEV))
    dy_nhwc = torch.zeros(N, H, W, conv.dy_channels(Cout))
    dy_nhwc[..., :Cout] = dy.permute(0, 2, 3, 1)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    dxm = conv.conv_dgrad(d, dy_nhwc.bfloat16().to(DEV), wd, mask_x=x_nhwc, mask_slope=0.2).float().cpu().permute(0, 3, 1, 2)
    assert conv.lib().m355_last_kernel().decode() == "k_conv_c8"
    want = xr.grad * torch.where(x > 0, 1.0, 0.2)
    assert (dxm - want).abs().max().item() / want.abs().max().item() < 1.2e-2


@pytest.mark.parametrize("case", [(2, 32, 64, 128, 256, 4, 2, 1, 1, 2, 0), (2, 64, 64, 64, 128, 4, 2, 1, 1, 2, 0),
                                  (3, 16, 64, 256, 512, 4, 2, 1, 1, 0, 0)])
def test_class_weight_gradients_are_ordered_sums_of_partial_rows(pkg, case, monkeypatch):
    """(round 6) the stride-2 class weight gradients on k_wgrad_halo (D.conv2-4) no longer meet in same-address fp32 atomics: every
    workgroup stores its partial tile into a row of the workspace m355_conv2d_wgrad_ws_bytes sizes, one launch adds the rows in order.
    The layer therefore has the workspace form, conv_wgrad takes it in EVERY mode, repeated launches give the same bits without the
    deterministic mode, and the result (weights and the fused bias gradient) is the atomics path's up to summation order."""
    if os.environ.get("M355_WGRAD_HALO_PART") == "0" or os.environ.get("M355_WGRAD_UP_PART") == "0":
        pytest.skip("the A/B switch puts these layers back on atomics")
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    monkeypatch.setattr(conv, "_DETERMINISTIC", False)
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    g = torch.Generator().manual_seed(31)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    assert conv.plan(d).wgrad_ws_bytes > 0
    x = torch.randn(N, H, W, Cin, generator=g).bfloat16().to(DEV)
    dy = torch.randn(N, H // 2, W // 2, Cout, generator=g).bfloat16().to(DEV)
    runs = []
    for _ in range(3):
        db = torch.full((Cout,), 7.0, device=DEV)            # (overwritten, not accumulated)
        dw = conv.conv_wgrad(d, x, dy, raw=True, dbias=db)
        assert conv.lib().m355_last_kernel().decode() == "k_wgrad_halo"
        runs.append((dw.clone(), db.clone()))
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
    # against the atomics form of the same kernel (m355_conv2d_wgrad: zero fill + fp32 atomics)
    import ctypes
    ref, refb = torch.empty_like(runs[0][0]), torch.empty(Cout, device=DEV)
    conv.launch("conv2d_wgrad", ctypes.byref(d), conv.ptr(x), conv.ptr(dy), conv.ptr(ref), conv.ptr(refb), conv.stream())
    torch.cuda.synchronize()
    assert (runs[0][0] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert (runs[0][1] - refb).abs().max().item() <= 2e-5 * max(1.0, refb.abs().max().item())


@pytest.mark.parametrize("case", [(8, 32, 32, 128, 64), (4, 16, 64, 128, 128), (8, 128, 64, 128, 64)])
def test_subpixel_weight_gradients_are_ordered_sums_of_partial_rows(pkg, case, monkeypatch):
    """(round 6) the same for the sub-pixel form of upsample + 3x3 (G.blk5.conv1, G.blk6.conv1): the four classes' workgroups store
    partial rows of the 16-entry effective gradient (+ four blocks of bias cells, one per class), and the 16 -> 9 fold adds the rows in
    order.  The plan says so (wgrad_ws_ordered), the deterministic mode takes the same path (no integer cells: the bits are already
    run-to-run identical), and the result is the gradient of the stored-extent formulation (fp64 torch) at fp32 accuracy."""
    if os.environ.get("M355_WGRAD_HALO_PART") == "0" or os.environ.get("M355_WGRAD_UP_PART") == "0":
        pytest.skip("the A/B switch puts these layers back on atomics")
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(37)
    d = conv.make_desc(N, H, W, Cin, Cout, 3, 3, 1, 1, 1, 1, 1)
    pl = conv.plan(d)
    assert pl.wgrad_ws_bytes > 4 * (Cout * 16 * Cin + Cout) and pl.wgrad_ws_ordered == 1
    x = torch.randn(N, H, W, Cin, generator=g).bfloat16()
    dy = torch.randn(N, 2 * H, 2 * W, Cout, generator=g).bfloat16()
    xd, dyd = x.to(DEV), dy.to(DEV)
    runs = []
    for det in (False, False, True, True):
        monkeypatch.setattr(conv, "_DETERMINISTIC", det)
        db = torch.full((Cout,), 7.0, device=DEV)
        dw = conv.conv_wgrad(d, xd, dyd, raw=True, dbias=db)
        assert conv.lib().m355_last_kernel().decode() == "k_wgrad_halo"
        runs.append((dw.clone(), db.clone()))
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
    w64 = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    ref_conv(x.double().permute(0, 3, 1, 2), w64, None, 1, 1, 1, 1, 1).backward(dy.double().permute(0, 3, 1, 2))
    want = w64.grad.permute(0, 2, 3, 1)
    assert ((runs[0][0].double().cpu() - want).abs().max() / want.abs().max()).item() < 2e-6
    wantb = dy.double().sum((0, 1, 2))
    assert ((runs[0][1].double().cpu() - wantb).abs().max() / wantb.abs().max()).item() < 2e-6


# (16-column outputs: the 32-column 3x3 layers take ordered partial rows at small batches since round 6 -- no arena slice)
@pytest.mark.parametrize("case", [(2, 16, 16, 128, 128, 3, 1, 1, 1, 1, 0), (2, 8, 8, 128, 64, 3, 1, 1, 1, 1, 1),
                                  (2, 32, 32, 8, 64, 3, 1, 1, 1, 1, 0)])
def test_wgrad_arena_accumulates_inside_backward(pkg, case, monkeypatch):
    """m355_conv2d_wgrad_acc (no zero fill) into the per-backward-pass arena: inside an autograd backward conv_wgrad(arena=True)
    gives the gradient of m355_conv2d_wgrad; the first pass (arena not sized yet) falls back, the second uses the arena, and a
    slice is zero again in the third pass although the second one wrote into it"""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    monkeypatch.setattr(conv, "_DETERMINISTIC", False)   # (the arena is the default mode's path; M355_DETERMINISTIC=1 runs bypass it)
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    g = torch.Generator().manual_seed(29)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    x = torch.randn(N, H, W, Cin, generator=g).bfloat16().to(DEV)
    Ho, Wo = (H * (2 if ups else 1) + 2 * ph - k) // stride + 1, (W * (2 if ups else 1) + 2 * pw - k) // stride + 1
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).bfloat16().to(DEV)
    want_db = torch.empty(Cout, device=DEV)
    want = conv.conv_wgrad(d, x, dy, raw=True, dbias=want_db).clone()
    got, used = [], []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, gr):
            db = torch.empty(Cout, device=DEV)
            a = conv.conv_wgrad(d, x, dy, raw=True, dbias=db, arena=True)
            b = conv.conv_wgrad(d, x, dy, raw=True, arena=True)   # a second layer of the same pass: its own slice
            st = conv.WgradArena._state[x.device]
            used.append(st["buf"] is not None and a.data_ptr() >= st["buf"].data_ptr() and
                        a.data_ptr() < st["buf"].data_ptr() + 4 * st["buf"].numel())
            assert a.data_ptr() != b.data_ptr()
            got.append((a.clone(), b.clone(), db))
            return gr

    conv.WgradArena._state.pop(x.device, None)
    for _ in range(3):
        t = torch.zeros(1, device=DEV, requires_grad=True)
        Probe.apply(t).sum().backward()
    assert used == [False, True, True], used
    scale = want.abs().max().item()
    for a, b, db in got:
        # (split-K atomics: the summation order differs from launch to launch)
        assert (a - want).abs().max().item() < 1e-4 * scale and (b - want).abs().max().item() < 1e-4 * scale
        assert (db - want_db).abs().max().item() < 1e-3 * max(1.0, want_db.abs().max().item())


@pytest.mark.parametrize("wgs", [None, "5"])
@pytest.mark.parametrize("case", [(3, 16, 64, 128, 128, 3, 1, 1, 1, 1, 0),    # 8 waves, 3x3
                                  (2, 16, 32, 256, 256, 3, 1, 1, 1, 2, 1),    # 8 waves, two 128-channel tiles, upsample, circular
                                  (2, 16, 16, 128, 64, 3, 1, 1, 1, 1, 1),     # 4 waves, upsample (two workgroups per CU)
                                  (3, 24, 32, 64, 64, 3, 1, 1, 1, 1, 0),      # 4 waves, resident weights
                                  (2, 16, 64, 128, 64, 3, 1, 1, 1, 0, 0),     # 4 waves, streamed weights, zero pad, bias
                                  (3, 16, 32, 128, 64, 3, 1, 1, 1, 1, 1),     # sub-pixel upsample conv: class pairs (rows per class)
                                  (2, 16, 32, 64, 128, 3, 1, 1, 1, 0, 1)])    # sub-pixel, 8-wave classes, zero pad, bias
def test_conv_fwd_fused_bn_statistics(pkg, case, wgs, monkeypatch):
    """m355_conv2d_fwd_stats: y is bit-identical to m355_conv2d_fwd and the per-workgroup partial rows add up to the per-channel
    sum / sum of squares of the conv's fp32 results (fp32 reference conv on the same bf16 operands: only the summation order
    differs) -- channel by channel, so a permuted lane -> channel map cannot pass; bn_finalize on the rows gives mean / rstd"""
    if wgs:
        monkeypatch.setenv("M355_HALO_WGS", wgs)
        monkeypatch.setenv("M355_STATS_UPS_WGS", wgs)
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    lib = importlib.import_module("2dimageto3dmodel_amd._lib")
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    g = torch.Generator().manual_seed(31)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    b = (torch.randn(Cout, generator=g) * 0.5) if mode == 0 else None
    # channel-dependent scale and offset: every channel has its own statistics
    w = w * (0.5 + torch.arange(Cout).float().view(-1, 1, 1, 1) / Cout)
    w = w.bfloat16().float()
    if ups and W % 32 == 0 and H % 8 == 0:
        # the sub-pixel form rounds PRE-SUMMED weights (w0+w1, ...) to bf16: small-integer weights (x a power of two, x 1..3 per
        # channel) keep every sum of up to four taps exact, so the fp32 reference below still describes the same operator
        w = torch.randint(-15, 16, (Cout, Cin, k, k), generator=g).float() * (1 + torch.arange(Cout) % 3).float().view(-1, 1, 1, 1) / 512
    y_ref = ref_conv(x, w, b, stride, ph, pw, mode, ups)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    rows = conv.conv_stats_rows(d)
    assert rows > 0
    wf, _ = conv.weight_prep(d, w.to(DEV))
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    bd = None if b is None else b.to(DEV)
    y0 = conv.conv_fwd(d, x_nhwc, wf, bd)
    want_s = y_ref.double().sum((0, 2, 3))
    want_q = (y_ref.double() ** 2).sum((0, 2, 3))
    P = N * y_ref.shape[2] * y_ref.shape[3]
    for _ in range(2):
        y, part = conv.conv_fwd_stats(d, x_nhwc, wf, bd)
        assert tuple(part.shape) == (rows, 2, Cout)
        assert torch.equal(y.view(torch.int16), y0.view(torch.int16))
        tot = part.double().sum(0).cpu()
        scale_s = y_ref.abs().double().sum((0, 2, 3))
        assert ((tot[0] - want_s).abs() / scale_s).max().item() < 2e-5, ((tot[0] - want_s).abs() / scale_s).max().item()
        assert ((tot[1] - want_q).abs() / want_q).max().item() < 2e-5, ((tot[1] - want_q).abs() / want_q).max().item()
    # ... and through bn_finalize (the consumer of the rows)
    n = N
    gamma = torch.zeros(n, Cout, device=DEV)
    beta = torch.zeros(n, Cout, device=DEV)
    coef = torch.empty(2 * n + 2, Cout, device=DEV)
    lib.launch("bn_finalize", lib.ptr(part), rows, float(P), None, lib.ptr(gamma), lib.ptr(beta), Cout, n, Cout, 1e-5, 0.1, None, None,
               lib.ptr(coef[2 * n]), lib.ptr(coef[2 * n + 1]), lib.ptr(coef[:n]), lib.ptr(coef[n:2 * n]), lib.stream())
    mean = (want_s / P).float()
    var = (want_q / P - (want_s / P) ** 2).float()
    assert (coef[2 * n].cpu() - mean).abs().max().item() < 1e-4 * max(1.0, mean.abs().max().item())
    assert (coef[2 * n + 1].cpu() * torch.sqrt(var + 1e-5) - 1).abs().max().item() < 1e-3


def test_conv_fwd_stats_refused_where_not_fused(pkg):
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    lib = importlib.import_module("2dimageto3dmodel_amd._lib")
    for case in [(2, 8, 4, 64, 64, 3, 1, 1, 1, 1, 0), (2, 16, 16, 8, 64, 5, 1, 2, 2, 2, 0), (2, 32, 64, 64, 128, 4, 2, 1, 1, 0, 0),
                 (2, 16, 32, 64, 3, 5, 1, 2, 2, 1, 0)]:
        N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
        d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
        assert conv.conv_stats_rows(d) == 0
        x = torch.zeros(N, H, W, Cin, dtype=torch.bfloat16, device=DEV)
        wf, _ = conv.weight_prep(d, torch.zeros(Cout, Cin, k, k, device=DEV))
        with pytest.raises(lib.M355Error):
            conv.conv_fwd_stats(d, x, wf, rows=4)




@pytest.mark.timeout(600)
@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups, what, kernel family
    (16, 256, 256, 64, 128, 4, 2, 1, 1, 2, 0, "fwd_bits", "k_conv_halo"),   # D.conv2 forward: 8-wave 2x2 class variant (fragment look-ahead), bit masks
    (16, 128, 64, 128, 64, 3, 1, 1, 1, 1, 1, "fwd_stats", "k_conv_halo"),   # G.blk6.conv1 forward: 4-wave variant, x2 upsample, fused statistics
    (16, 128, 128, 128, 256, 4, 2, 1, 1, 2, 0, "fwd_bits", "k_conv_halo"),  # D.conv3 forward
    (16, 256, 256, 64, 128, 4, 2, 1, 1, 2, 0, "dgrad", "k_conv_halo"),      # D.conv2 dgrad: class pairs
    (16, 256, 128, 64, 64, 3, 1, 1, 1, 1, 0, "fwd_stats", "k_conv_halo"),   # G.blk6.conv2 forward (resident weights where eligible)
    (16, 64, 64, 256, 512, 4, 2, 1, 1, 2, 0, "dgrad", "k_conv_halo"),       # D.conv4 dgrad
    (16, 64, 32, 128, 128, 3, 1, 1, 1, 1, 1, "dgrad", "k_conv_halo"),       # G.blk5.conv1 dgrad through the upsample
    # the other pipelined families (counted waits / multi-stage LDS rings of their own)
    (16, 256, 256, 8, 64, 5, 1, 2, 2, 2, 0, "fwd_bits", "k_conv_c8"),       # D.conv1 forward
    (16, 256, 128, 64, 3, 5, 1, 2, 2, 1, 0, "dgrad", "k_conv_c8"),          # conv_final dgrad (+ the pad-column term)
    (16, 256, 128, 64, 3, 5, 1, 2, 2, 1, 0, "fwd", None),                   # conv_final forward (NHWC output form)
    (16, 128, 64, 128, 64, 1, 1, 0, 0, 0, 0, "fwd", "k_conv_glds"),         # blk6 shortcut
    (16, 16, 8, 256, 256, 3, 1, 1, 1, 1, 1, "fwd_stats", "k_conv_glds"),    # blk3 stage: 128 x 128 tiles + split-K + finishing pass
    (32, 32, 32, 16, 64, 5, 1, 2, 2, 2, 0, "fwd", "k_conv_glds"),           # MeshDiscriminator.conv1 (four-stage small-tile form)
    (32, 32, 32, 512, 1, 5, 1, 2, 2, 2, 0, "fwd", None),                    # D.conv5 forward
    (16, 256, 256, 8, 64, 5, 1, 2, 2, 2, 0, "wgrad", "k_wgrad_c8"),         # D.conv1 weight gradient (planar kernel, ordered partials)
    (16, 128, 128, 128, 256, 4, 2, 1, 1, 2, 0, "wgrad", "k_wgrad_halo"),    # D.conv3 weight gradient (deterministic mode: fixed point)
    (16, 256, 128, 64, 64, 3, 1, 1, 1, 1, 0, "wgrad", "k_wgrad_halo"),      # G.blk6.conv2 weight gradient
    (32, 32, 32, 16, 64, 5, 1, 2, 2, 2, 0, "wgrad", "k_wgrad_dma"),         # MeshDiscriminator.conv1 weight gradient
    (16, 256, 128, 64, 3, 5, 1, 2, 2, 1, 0, "wgrad", "k_wgrad_smallco"),    # conv_final weight gradient
    # the sub-pixel form of upsample + 3x3 (round 5): forward on the class kernels with fused statistics, weight gradient on the
    # class variant of k_wgrad_halo (dy addressed with stride 2) + the 16 -> 9 fold (round 6: per-workgroup partial rows, the fold adds
    # them in row order -- the same path in both modes)
    (16, 64, 32, 128, 128, 3, 1, 1, 1, 1, 1, "fwd_stats", "k_conv_halo"),   # G.blk5.conv1 forward: 8-wave 2x2 classes
    (16, 128, 64, 128, 64, 3, 1, 1, 1, 1, 1, "dgrad", "k_conv_halo"),       # G.blk6.conv1 dgrad: stride-2 forward variant on dy
    (16, 128, 64, 128, 64, 3, 1, 1, 1, 1, 1, "wgrad", "k_wgrad_halo"),      # G.blk6.conv1 weight gradient
    (16, 64, 32, 128, 128, 3, 1, 1, 1, 1, 1, "wgrad", "k_wgrad_halo"),      # G.blk5.conv1 weight gradient
])
def test_persistent_tile_kernels_repeat_bit_identically_under_memory_pressure(pkg, case):
    """Every launch of a conv kernel must produce the same bits (the weight gradients: in deterministic mode).  Round 4 found (run-to-
    run determinism of a training cycle at 256^2) that the persistent-tile variants of k_conv_halo with fragment look-ahead credited
    the previous tile's epilogue stores in ONE counted s_waitcnt too many: the weights of the third step of every tile but a
    workgroup's first were not awaited, and about one launch in a hundred read a stale weight slot when HBM was busy.  Here: 40
    launches of every pipelined kernel family, each behind a 512 MB device copy that is still draining when the kernel starts,
    against the first (launched on an idle device); the first is also compared with torch-CPU on one sample."""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups, what, family = case
    g = torch.Generator().manual_seed(4242)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    b = (0.1 * torch.randn(Cout, generator=g))
    wf, wd = conv.weight_prep(d, w.to(DEV))
    ho, wo = conv.out_hw(d)
    x = torch.randn(N, H, W, Cin, generator=g).bfloat16().to(DEV)
    cy = conv.dy_channels(Cout)
    dy = torch.zeros(N, ho, wo, cy, dtype=torch.bfloat16)
    dy[..., :Cout] = torch.randn(N, ho, wo, Cout, generator=g).bfloat16()
    dy = dy.to(DEV)
    junk_a = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    junk_b = torch.empty_like(junk_a)

    def once():
        if what == "fwd_bits":
            return list(conv.conv_fwd(d, x, wf, b.to(DEV), slope=0.2, emit_bits=True))
        if what == "fwd_stats":
            return list(conv.conv_fwd_stats(d, x, wf, b.to(DEV)))
        if what == "fwd":
            return [conv.conv_fwd(d, x, wf, b.to(DEV))]
        if what == "wgrad":
            db = torch.zeros(Cout, dtype=torch.float32, device=DEV) if conv.wgrad_fuses_dbias(d) else None
            return [conv.conv_wgrad(d, x, dy, raw=True, dbias=db).clone()] + ([db] if db is not None else [])
        return [conv.conv_dgrad(d, dy, wd)]

    prev = conv.set_deterministic(True)
    try:
        torch.cuda.synchronize()
        first = once()
        assert family is None or conv.lib().m355_last_kernel().decode() == family
        torch.cuda.synchronize()
        for rep in range(40):
            junk_b.copy_(junk_a)
            out = once()
            for t0, t in zip(first, out):
                assert torch.equal(t0, t), f"launch {rep}: {int((t0 != t).sum())} of {t.numel()} elements differ from the first launch"
    finally:
        conv.set_deterministic(prev)
    # ... and the first launch is right (sample 0 against torch-CPU on the same bf16 operands; the weight gradient: all samples)
    if what == "wgrad":
        wr = w.clone().requires_grad_()
        br = b.clone().requires_grad_()
        ref_conv(x.float().cpu().permute(0, 3, 1, 2), wr, br, stride, ph, pw, mode, ups).backward(dy[..., :Cout].float().cpu().permute(0, 3, 1, 2))
        got, want = first[0].cpu().permute(0, 3, 1, 2), wr.grad          # raw layout [Cout,kh,kw,Cin]
        if len(first) > 1:
            assert (first[1].cpu() - br.grad).abs().max().item() / br.grad.abs().max().item() < 1e-3
    elif what == "dgrad":
        xr = torch.zeros(1, Cin, H, W, requires_grad=True)
        ref_conv(xr, w, None, stride, ph, pw, mode, ups).backward(dy[:1, ..., :Cout].float().cpu().permute(0, 3, 1, 2))
        got, want = first[0][:1].float().cpu().permute(0, 3, 1, 2), xr.grad
    else:
        want = ref_conv(x[:1].float().cpu().permute(0, 3, 1, 2), w, b, stride, ph, pw, mode, ups)
        if what == "fwd_bits":
            want = F.leaky_relu(want, 0.2)
        got = first[0][:1].float().cpu().permute(0, 3, 1, 2)
    assert (got - want).abs().max().item() / want.abs().max().item() < 1.2e-2


@pytest.mark.parametrize("case", [
    (2, 32, 64, 128, 256, 4, 2, 1, 1, 2, 0),   # 4x4 stride 2: the 2x2 class variant of k_wgrad_halo
    (2, 16, 32, 128, 128, 3, 1, 1, 1, 1, 0),   # 3x3: k_wgrad_halo
    (2, 8, 32, 128, 128, 3, 1, 1, 1, 1, 1),    # upsample + 3x3 in the sub-pixel form: 16-entry cells + the integer 16 -> 9 fold
    (2, 16, 16, 32, 64, 5, 1, 2, 2, 2, 0),     # k_wgrad_dma
    (3, 12, 6, 96, 96, 3, 1, 1, 1, 0, 0),      # the generic split-K kernel
    (2, 16, 32, 64, 3, 5, 1, 2, 2, 1, 0),      # a 3-channel head: k_wgrad_smallco
    (3, 24, 64, 8, 64, 5, 1, 2, 2, 2, 0),      # D.conv1 (ordered partial rows in every mode)
])
def test_deterministic_weight_gradient_is_scale_free(pkg, case):
    """The deterministic weight gradient (integer cells, csrc/conv_dma.h wg_accum) must not have a preferred gradient magnitude: Adam
    (betas 0, 0.9 -- main.py:588-589) is invariant to the gradient's scale, so a layer whose gradients are 1e-9 matters as much as one
    at 1e-2.  Here dy is scaled by 2^-30 and by 2^12 -- exact in bf16, and exact through the MFMA partial tiles -- and the result must be
    the scaled result of the unscaled run, to the bit (the cell triple holds the exact sum of the partials; a single 2^-36 grid
    gave 1e-3 relative error at 2^-30).  The unscaled run is also held to the fp64 sum of the same bf16 operands at fp32 accuracy."""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
    g = torch.Generator().manual_seed(99)
    d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
    ho, wo = conv.out_hw(d)
    x = torch.randn(N, H, W, Cin, generator=g).bfloat16()
    cy = conv.dy_channels(Cout)
    dy = torch.zeros(N, ho, wo, cy, dtype=torch.bfloat16)
    dy[..., :Cout] = torch.randn(N, ho, wo, Cout, generator=g).bfloat16()

    def run(scale):
        db = torch.zeros(Cout, dtype=torch.float32, device=DEV) if conv.wgrad_fuses_dbias(d) else None
        dys = (dy.float() * scale).bfloat16()
        assert torch.equal(dys.float(), dy.float() * scale)      # (a power of two: no rounding)
        dw = conv.conv_wgrad(d, x.to(DEV), dys.to(DEV), raw=True, dbias=db).clone()
        return dw.cpu(), (None if db is None else db.cpu())

    prev = conv.set_deterministic(True)
    try:
        dw1, db1 = run(1.0)
        for scale in (2.0 ** -30, 2.0 ** 12):
            dws, dbs = run(scale)
            assert torch.isfinite(dws).all()
            # equal to the bit, up to the one hazard the conversion has: the cell triple goes through a double on its way to fp32, and a
            # sum that needs more than 53 bits can land within 2^-54 of an fp32 rounding midpoint (about 1e-9 per element): allow one
            # element in 10^4 to differ by one unit in the last place
            for got, want in ((dws, dw1 * scale),) + (((dbs, db1 * scale),) if db1 is not None else ()):
                diff = got != want
                assert int(diff.sum()) <= got.numel() // 10000, (scale, int(diff.sum()))
                assert float(((got - want).abs() / want.abs().clamp_min(1e-38))[diff].max()) <= 2.0 ** -22 if diff.any() else True
    finally:
        conv.set_deterministic(prev)
    wr = w64 = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    ref_conv(x.double().permute(0, 3, 1, 2), wr, None, stride, ph, pw, mode, ups).backward(dy[..., :Cout].double().permute(0, 3, 1, 2))
    want = w64.grad.permute(0, 2, 3, 1)          # raw layout [Cout, kh, kw, Cin]
    assert ((dw1.double() - want).abs().max() / want.abs().max()).item() < 2e-6


@pytest.mark.timeout(900)
def test_every_shape_class_repeats_bit_identically_under_memory_pressure(pkg):
    """the shape classes of CASES (every dispatch path: generic MFMA, LDS-DMA, halo, 8-channel, small-Cout, heads, odd-kernel
    stride-2 folds): forward, dgrad and the deterministic weight gradient eight times each behind a draining 256 MB copy -- the
    same bits every time (correctness of the first launch is test_conv_fwd_and_dgrad's business)"""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    junk_a = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    junk_b = torch.empty_like(junk_a)
    prev = conv.set_deterministic(True)
    try:
        for case in CASES:
            N, H, W, Cin, Cout, k, stride, ph, pw, mode, ups = case
            g = torch.Generator().manual_seed(hash(case) % 1000)
            d = conv.make_desc(N, H, W, Cin, Cout, k, k, stride, ph, pw, mode, ups)
            w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
            wf, wd = conv.weight_prep(d, w.to(DEV))
            b = torch.randn(Cout, generator=g).to(DEV)
            x = torch.randn(N, H, W, Cin, generator=g).bfloat16().to(DEV)
            ho, wo = conv.out_hw(d)
            dy = torch.zeros(N, ho, wo, conv.dy_channels(Cout), dtype=torch.bfloat16)
            dy[..., :Cout] = torch.randn(N, ho, wo, Cout, generator=g).bfloat16()
            dy = dy.to(DEV)

            def once():
                return [conv.conv_fwd(d, x, wf, b, slope=0.2), conv.conv_dgrad(d, dy, wd), conv.conv_wgrad(d, x, dy, raw=True).clone()]

            torch.cuda.synchronize()
            first = once()
            torch.cuda.synchronize()
            for rep in range(8):
                junk_b.copy_(junk_a)
                for name, t0, t in zip(("fwd", "dgrad", "wgrad"), first, once()):
                    assert torch.equal(t0, t), (case, name, rep, int((t0 != t).sum()))
    finally:
        conv.set_deterministic(prev)
