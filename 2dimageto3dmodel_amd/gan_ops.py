"""Autograd glue of the GAN path: the MFMA conv2d Function and the channels-last (NHWC bf16) layer helpers.

conv2d is always the HIP implicit GEMM (csrc/conv_mfma.hip: fwd, dgrad, wgrad) -- there is no torch/MIOpen
fallback.  The normalisation layers compute their statistics in fp32 over the NHWC tensor; SynchronizedBatchNorm2d
all-reduces [sum, sum of squares, count] over torch.distributed (RCCL on ROCm) instead of the reference's
master/slave pipes (code/sync_batchnorm/batchnorm.py:110-131).
"""
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import conv as C
from ._lib import launch, lib, ptr, stream


def _ceil(a, b):
    return (a + b - 1) // b * b


def to_nhwc_bf16(x_nchw, pad_to=1):
    """NCHW (any float dtype) -> NHWC bf16, channels zero-padded up to a multiple of `pad_to`"""
    x = x_nchw.permute(0, 2, 3, 1)
    c = x.shape[3]
    cp = _ceil(c, pad_to)
    if cp != c:
        x = F.pad(x, (0, cp - c))
    return x.contiguous().to(torch.bfloat16)


def to_nchw_f32(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).float()


def upsample2x(x):
    """nearest x2 of an NHWC tensor (F.interpolate(scale_factor=2, mode='nearest'), gan.py:319)"""
    n, h, w, c = x.shape
    return x[:, :, None, :, None, :].expand(n, h, 2, w, 2, c).reshape(n, 2 * h, 2 * w, c)


def leaky_relu(x, slope):
    return F.leaky_relu(x, slope)


class Conv2dFn(torch.autograd.Function):
    """F.conv2d(pad(up(x)), w, b) [+ LeakyReLU] on NHWC bf16 through libm355."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad_h, pad_w, mode, ups, slope, out_f32_nchw):
        n, h, w, cx = x.shape
        cout, cw, kh, kw = weight.shape
        if cx % 8 or cx < cw:
            raise ValueError(f"conv2d: input has {cx} channels (must be a multiple of 8 and >= {cw})")
        d = C.make_desc(n, h, w, cx, cout, kh, kw, stride, pad_h, pad_w, mode, ups)
        need_dx = ctx.needs_input_grad[0]
        wf, wd = C.weight_prep(d, weight, want_dgrad=need_dx)
        y = C.conv_fwd(d, x.detach(), wf, None if bias is None else bias.detach(), out_f32_nchw, slope, cin_real=cw)
        ctx.d, ctx.cw, ctx.slope, ctx.f32 = d, cw, slope, out_f32_nchw
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x.detach(), wd, y if slope != 1.0 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wd, y = ctx.saved_tensors
        d = ctx.d
        c32 = C.dy_channels(d.Cout)
        if ctx.slope != 1.0 and not ctx.f32 and c32 == d.Cout and 256 % (d.Cout // 8) == 0:
            # LeakyReLU epilogue backward + bias gradient in one pass (csrc/gan_elem.hip)
            g, db = lrelu_bwd(dy.contiguous(), y, ctx.slope)
            if not (ctx.has_bias and ctx.needs_input_grad[2]):
                db = None
        else:
            g = dy.permute(0, 2, 3, 1) if ctx.f32 else dy       # -> NHWC view
            if ctx.slope != 1.0:
                yy = y.permute(0, 2, 3, 1) if ctx.f32 else y
                g = g * torch.where(yy > 0, 1.0, ctx.slope).to(g.dtype)
            db = g.float().sum((0, 1, 2)) if ctx.has_bias and ctx.needs_input_grad[2] else None
            if c32 != d.Cout:
                g = F.pad(g, (0, c32 - d.Cout))
            g = g.contiguous().to(torch.bfloat16)
        dx = C.conv_dgrad(d, g, wd, cin_real=ctx.cw) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = C.conv_wgrad(d, x, g, cin_real=ctx.cw)[:, :ctx.cw].contiguous()
        return dx, dw, db, None, None, None, None, None, None, None


def conv2d(x, weight, bias, stride, pad_h, pad_w, mode, upsample=0, slope=1.0, out_f32_nchw=False):
    return Conv2dFn.apply(x, weight, bias, stride, pad_h, pad_w, mode, int(upsample), float(slope), bool(out_f32_nchw))


# ------------------------------------------------------------------------------------------------ fused elementwise
def _ws(pixels_per_group, groups, nvals, C, device):
    n = lib().m355_chan_reduce_ws_bytes(pixels_per_group, groups, nvals, C)
    return torch.empty((max(n, 8),), dtype=torch.uint8, device=device)


def lrelu_bwd(dy, y, slope):
    """g = dy * (y > 0 ? 1 : slope) (bf16) and its per-channel sum (fp32) in one pass"""
    C_ = y.shape[-1]
    P = y.numel() // C_
    g = torch.empty_like(y)
    db = torch.empty((C_,), dtype=torch.float32, device=y.device)
    launch("lrelu_bwd", ptr(dy), ptr(y), ptr(g), ptr(db), ptr(_ws(P, 1, 1, C_, y.device)), P, C_, float(slope), stream())
    return g, db


def bn_sums(x):
    """x [N,H,W,C] bf16 -> [2,C] fp32 (sum, sum of squares) over N*H*W"""
    C_ = x.shape[-1]
    P = x.numel() // C_
    sums = torch.empty((2, C_), dtype=torch.float32, device=x.device)
    launch("bn_stats", ptr(x), ptr(sums), ptr(_ws(P, 1, 2, C_, x.device)), P, C_, stream())
    return sums


class AffineActFn(torch.autograd.Function):
    """y = LeakyReLU(((x - mean) * rstd) * scale[n,c] + shift[n,c]) on NHWC bf16, with the batch-norm backward
    (mean / rstd are functions of x when `batch_stats`) folded into three per-channel coefficients."""

    @staticmethod
    def forward(ctx, x, scale, shift, mean, rstd, slope, batch_stats, count, sync):
        n, h, w, c = x.shape
        x = x.contiguous()
        a = (rstd * scale.float()).contiguous()            # [N,C]
        b = (shift.float() - mean * a).contiguous()
        y = torch.empty_like(x)
        launch("affine_act_fwd", ptr(x), ptr(a), ptr(b), ptr(y), n, h * w, c, float(slope), stream())
        ctx.save_for_backward(x, a, b, scale, mean, rstd)
        ctx.cfg = (slope, batch_stats, count, sync)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a, b, scale, mean, rstd = ctx.saved_tensors
        slope, batch_stats, count, sync = ctx.cfg
        n, h, w, c = x.shape
        dy = dy.contiguous()
        sums = torch.empty((n, 2, c), dtype=torch.float32, device=x.device)
        launch("affine_act_bwd_reduce", ptr(dy), ptr(x), ptr(a), ptr(b), ptr(sums), ptr(_ws(h * w, n, 2, c, x.device)), n,
               h * w, c, float(slope), stream())
        s1, s2 = sums[:, 0], sums[:, 1]                    # sum dz, sum dz*x   per (n,c)
        dxhat_xhat = rstd * (s2 - mean * s1)               # sum_hw dz * xhat
        dshift = s1
        dscale = dxhat_xhat
        sc = scale.float()
        A = (rstd * sc).contiguous()
        if batch_stats:
            m = torch.stack(((sc * s1).sum(0), (sc * dxhat_xhat).sum(0)))   # [2,C] sums over the local batch
            if sync:
                dist.all_reduce(m, op=dist.ReduceOp.SUM)
            m1, m2 = m[0] / count, m[1] / count
            Bc = (-rstd * rstd * m2).contiguous()
            Cc = (-rstd * m1 + rstd * rstd * mean * m2).contiguous()
        else:
            Bc = torch.zeros(c, dtype=torch.float32, device=x.device)
            Cc = Bc
        dx = torch.empty_like(x)
        launch("affine_act_bwd_apply", ptr(dy), ptr(x), ptr(a), ptr(b), ptr(A), ptr(Bc), ptr(Cc), ptr(dx), n, h * w, c,
               float(slope), stream())
        return dx, dscale.to(scale.dtype), dshift.to(scale.dtype), None, None, None, None, None, None


def _fused_ok(x):
    c = x.shape[-1]
    return x.is_cuda and x.dtype == torch.bfloat16 and c % 8 == 0 and c <= 2048 and 256 % (c // 8) == 0


# ------------------------------------------------------------------------------------------------ normalisation
def _affine_act(xhat, scale, shift, slope):
    """(torch path, CPU tests / odd channel counts) xhat fp32 [N,H,W,C]; scale/shift [N,C]"""
    y = xhat * scale[:, None, None, :].float() + shift[:, None, None, :].float()
    if slope != 1.0:
        y = F.leaky_relu(y, slope)
    return y.to(torch.bfloat16)


class _SyncMoments(torch.autograd.Function):
    """(torch path) all-reduce (sum) of the per-channel [sum | sum of squares | count] vector; the backward
    all-reduces the incoming gradient, which is the exact adjoint of a sum over ranks."""

    @staticmethod
    def forward(ctx, v):
        v = v.clone()
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        return v

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d(ch, affine=False) on NHWC bf16 (same buffers: running_mean, running_var,
    num_batches_tracked; biased variance normalises, unbiased updates the running estimate, eps added inside
    the sqrt -- the single-device formula of code/sync_batchnorm/batchnorm.py:71-73 = F.batch_norm)."""
    sync = False

    def __init__(self, ch, affine=False, eps=1e-5, momentum=0.1):
        super().__init__()
        assert not affine
        self.num_features, self.eps, self.momentum = ch, eps, momentum
        self.register_buffer("running_mean", torch.zeros(ch))
        self.register_buffer("running_var", torch.ones(ch))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def _is_sync(self):
        return self.sync and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _update_running(self, mean, var, cnt):
        with torch.no_grad():
            unbiased = var * (cnt / (cnt - 1)) if float(cnt) > 1 else var
            self.running_mean.mul_(1 - self.momentum).add_(mean.detach(), alpha=self.momentum)
            self.running_var.mul_(1 - self.momentum).add_(unbiased.detach(), alpha=self.momentum)
            self.num_batches_tracked += 1

    def forward(self, x, scale, shift, slope=1.0):
        if _fused_ok(x):
            sync = self._is_sync()
            if self.training:
                cnt = float(x.shape[0] * x.shape[1] * x.shape[2])
                with torch.no_grad():
                    sums = bn_sums(x)
                    if sync:  # one fused [sum | sumsq | count] message per layer
                        v = torch.cat((sums.reshape(-1), sums.new_tensor([cnt])))
                        dist.all_reduce(v, op=dist.ReduceOp.SUM)
                        sums, cnt = v[:-1].view(2, -1), float(v[-1])
                    mean = sums[0] / cnt
                    var = (sums[1] / cnt - mean * mean).clamp_min(0)
                    rstd = torch.rsqrt(var + self.eps)
                self._update_running(mean, var, cnt)
                return AffineActFn.apply(x, scale, shift, mean, rstd, slope, True, cnt, sync)
            rstd = torch.rsqrt(self.running_var + self.eps)
            return AffineActFn.apply(x, scale, shift, self.running_mean, rstd, slope, False, 1.0, False)
        xf = x.float()
        if self.training:
            cnt = x.shape[0] * x.shape[1] * x.shape[2]
            s, ss = xf.sum((0, 1, 2)), (xf * xf).sum((0, 1, 2))
            if self._is_sync():
                v = _SyncMoments.apply(torch.cat((s, ss, s.new_tensor([float(cnt)]))))
                c = s.numel()
                s, ss, cnt = v[:c], v[c:2 * c], v[2 * c]
            mean = s / cnt
            var = (ss / cnt - mean * mean).clamp_min(0)
            self._update_running(mean, var, cnt)
        else:
            mean, var = self.running_mean, self.running_var
        return _affine_act((xf - mean) * torch.rsqrt(var + self.eps), scale, shift, slope)


class SynchronizedBatchNorm2d(BatchNorm2d):
    """sync_batchnorm.SynchronizedBatchNorm2d(ch, affine=False): statistics over the global batch."""
    sync = True


class InstanceNorm2d(nn.Module):
    def __init__(self, ch, eps=1e-5):
        super().__init__()
        self.eps = eps

    def forward(self, x, scale, shift, slope=1.0):
        xf = x.float()
        mean = xf.mean((1, 2), keepdim=True)
        var = xf.var((1, 2), unbiased=False, keepdim=True)
        return _affine_act((xf - mean) * torch.rsqrt(var + self.eps), scale, shift, slope)


class NoNorm(nn.Module):
    def forward(self, x, scale, shift, slope=1.0):
        if _fused_ok(x):
            c = x.shape[-1]
            zero = torch.zeros(c, dtype=torch.float32, device=x.device)
            return AffineActFn.apply(x, scale, shift, zero, torch.ones_like(zero), slope, False, 1.0, False)
        return _affine_act(x.float(), scale, shift, slope)
