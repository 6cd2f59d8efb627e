"""GPU: fused normalisation / activation kernels (csrc/gan_elem.hip) against plain torch fp32 on the same bf16 inputs."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape", [(3, 8, 4, 64), (2, 33, 17, 128), (4, 64, 32, 512), (2, 256, 256, 64)])
@pytest.mark.parametrize("mode", ["batch", "none", "eval"])
def test_norm_affine_act_fwd_bwd(pkg, shape, mode):
    G = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
    n, h, w, c = shape
    g = torch.Generator().manual_seed(c + h)
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).bfloat16()
    scale = 1 + 0.3 * torch.randn(n, c, generator=g)
    shift = 0.2 * torch.randn(n, c, generator=g)
    dy = torch.randn(shape, generator=g).bfloat16()
    # ---- torch reference in fp32 (autograd through the batch statistics)
    xr, sr, tr = x.float().requires_grad_(), scale.clone().requires_grad_(), shift.clone().requires_grad_()
    rm, rv = torch.linspace(-0.5, 0.5, c), torch.linspace(0.5, 2.0, c)
    if mode == "batch":
        mean, var = xr.mean((0, 1, 2)), xr.var((0, 1, 2), unbiased=False)
    elif mode == "eval":
        mean, var = rm, rv
    else:
        mean, var = torch.zeros(c), torch.ones(c) - 1e-5
    xhat = (xr - mean) * torch.rsqrt(var + 1e-5)
    yr = F.leaky_relu(xhat * sr[:, None, None, :] + tr[:, None, None, :], 0.2)
    yr.backward(dy.float())
    # ---- HIP
    mod = (G.NoNorm() if mode == "none" else G.BatchNorm2d(c)).to(DEV)
    if mode == "eval":
        mod.running_mean.copy_(rm)
        mod.running_var.copy_(rv)
        mod.eval()
    xd, sd, td = x.to(DEV).requires_grad_(), scale.to(DEV).requires_grad_(), shift.to(DEV).requires_grad_()
    y = mod(xd, sd, td, 0.2)
    assert y.dtype == torch.bfloat16
    assert (y.float().cpu() - yr.detach()).abs().max().item() < 0.03 * yr.abs().max().item()
    y.backward(dy.to(DEV))

    def rel(a, b):
        return (a.float().cpu() - b).abs().max().item() / b.abs().max().item()

    assert rel(xd.grad, xr.grad) < 2e-2
    assert rel(sd.grad, sr.grad) < 5e-3
    assert rel(td.grad, tr.grad) < 5e-3
    if mode == "batch":
        assert (mod.running_mean.cpu() - 0.1 * mean.detach()).abs().max().item() < 1e-3
        cnt = n * h * w
        assert (mod.running_var.cpu() - (0.9 + 0.1 * var.detach() * cnt / (cnt - 1))).abs().max().item() < 2e-3


def test_lrelu_bwd(pkg):
    G = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
    g = torch.Generator().manual_seed(1)
    y = torch.randn(3, 40, 24, 128, generator=g).bfloat16().to(DEV)
    dy = torch.randn(3, 40, 24, 128, generator=g).bfloat16().to(DEV)
    gg, db = G.lrelu_bwd(dy, y, 0.2)
    want = dy.float() * torch.where(y.float() > 0, 1.0, 0.2)
    assert (gg.float() - want).abs().max().item() < 1e-2
    assert (db - want.bfloat16().float().sum((0, 1, 2))).abs().max().item() < 1e-2 * want.abs().sum((0, 1, 2)).max().item()
