"""Autograd glue of the GAN path: the MFMA conv2d Function and the channels-last (NHWC bf16) layer helpers.

conv2d is always the HIP implicit GEMM (csrc/conv_mfma.hip: fwd, dgrad, wgrad) -- there is no torch/MIOpen
fallback.  The normalisation layers compute their statistics in fp32 over the NHWC tensor; SynchronizedBatchNorm2d
all-reduces [sum, sum of squares, count] over torch.distributed (RCCL on ROCm) instead of the reference's
master/slave pipes (code/sync_batchnorm/batchnorm.py:110-131).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import conv as C
from . import _lib
from ._lib import check, launch, lib, ptr, stream


def _act():
    """dtype of the activation tensors: bfloat16, or float32 when the EXACT build of the library is loaded (_lib.set_exact)"""
    from . import _lib
    return _lib.act_dtype()


def _ceil(a, b):
    return (a + b - 1) // b * b


# ------------------------------------------------------------------------------------------------ second HIP stream
# The networks have branches made of SMALL layers that leave most of the chip idle when they run alone -- the mesh discriminator
# (32x32 .. 8x8 maps), the generator's mesh head (blk3_mesh + conv_mesh at 32x16) -- next to branches of big, power-capped
# layers they do not depend on (the texture discriminator, blk3a .. conv_final).  `Fork` runs such a branch on a second stream:
# its kernels fill the gaps of the main branch instead of extending the step.  Measured (profiles/r04_streams.txt): at batch 16
# under the hipGraph 10.74 -> 10.59 ms per cycle (-1.5 %); at batch 64 26.5 -> 26.7 ms (+0.5 %, WORSE): every big kernel already
# holds the socket at its 1400 W limit, so concurrent work only takes clock from it -- the time of a power-capped step is its
# energy, and overlap saves none.  Hence the rule: fork only up to FORK_MAX_BATCH samples PER LAUNCH (M355_STREAMS=0 never, =1 always;
# M355_FORK_MAX_BATCH overrides).  Round 6 re-measured where the line is (profiles/r06_fork_threshold.txt): the all-or-nothing A/B at
# batch 64 hid that the generator's mesh head (G step: 64 samples per launch) gains what the mesh discriminator beside the 128-sample
# D step loses -- threshold 32 / 64 / 96 / 128: 24.52 / 24.14 / 24.12 / 24.56 ms per cycle at batch 64; batch 48: 19.73 -> 19.33,
# batch 40: 17.36 -> 17.01.  96 it is.
# Autograd replays each node on the stream its forward ran on and synchronises gradients crossing streams itself; what it does
# not know about is handled here and in conv.py: tensors shared across the streams are record_stream()ed (the caching allocator
# must not hand their memory to the other stream's next allocation while a kernel still reads it), the per-pass zero fills of
# conv.WgradArena / DbiasBlock carry an event the other stream waits for, and conv.flush_wgrad_finish joins the side streams.
_SIDE = {}          # device index -> side stream
STREAMS_ON = os.environ.get("M355_STREAMS", "auto") != "0"
FORK_MAX_BATCH = 1 << 30 if os.environ.get("M355_STREAMS") == "1" else int(os.environ.get("M355_FORK_MAX_BATCH", "96"))


_SN_SIDE = {}       # device index -> the stream spectral-norm steps are prefetched on (SpectralNormGroup.prefetch)
SN_PREFETCH_ON = os.environ.get("M355_SN_PREFETCH", "1") != "0"
_FORKED = set()     # device indices whose side stream has had work enqueued since conv.flush_wgrad_finish last joined it
_SIDE_RAW = {}      # raw HIP stream handle of a side stream -> device index (the check in the launch path is one dict lookup)


def side_streams():
    return list(_SIDE.values())


def note_side_work():
    """called where a deferred weight-gradient epilogue is QUEUED (conv.wgrad_finish): if the current stream is a side stream, the
    raw gradient it will read is being produced there, and the next flush must join that stream.  Marking at the enqueue -- not
    only in Fork.__enter__, i.e. the forward -- covers every backward of a forked graph: a second backward under retain_graph, or two
    forwards followed by two backwards, find the mark cleared by the first flush."""
    idx = _SIDE_RAW.get(stream())
    if idx is not None:
        _FORKED.add(idx)


def side_streams_to_join():
    """the side streams a backward pass may have put weight-gradient kernels on since the last join (Fork.__enter__ and
    note_side_work mark them), minus -- while the current stream is being captured into a hipGraph -- those that are not part of that capture: waiting on
    their (uncaptured) work from a capturing stream is a capture-isolation error, and nothing of this pass ran there"""
    out = []
    for idx in sorted(_FORKED):
        sd = _SIDE[idx]
        if torch.cuda.is_current_stream_capturing():
            with torch.cuda.stream(sd):
                if not torch.cuda.is_current_stream_capturing():
                    continue
        out.append(sd)
    _FORKED.clear()
    return out


class Fork:
    """with Fork(tensors_read_by_the_branch) as f: branch ... ; f.join(outputs) on the main stream before they are used"""

    def __init__(self, shared):
        self.shared = [t for t in shared if torch.is_tensor(t) and t.is_cuda]
        dev = self.shared[0].device
        self.main = torch.cuda.current_stream(dev)
        self.side = _SIDE.get(dev.index)
        if self.side is None:
            self.side = _SIDE[dev.index] = torch.cuda.Stream(dev)
            _SIDE_RAW[self.side.cuda_stream] = dev.index
        self._ctx = None

    def __enter__(self):
        self.side.wait_stream(self.main)          # everything the branch reads was produced on the main stream
        _FORKED.add(self.side.device.index)       # (its backward runs there too: conv.flush_wgrad_finish joins it)
        for t in self.shared:
            t.record_stream(self.side)
        self._ctx = torch.cuda.stream(self.side)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self._ctx.__exit__(*exc)
        return False

    def join(self, outputs):
        self.main.wait_stream(self.side)
        for t in outputs:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self.main)


class GradBarrier(torch.autograd.Function):
    """identity; its backward calls `fn()` before passing the gradient on.  Placed on a tensor every gradient of a sub-network flows
    through (the generator's trunk), it marks the point of the backward pass where that sub-network's parameter gradients are
    complete -- train.GanTrainer issues their all-reduce there, under the rest of the pass (parallel.BucketedGradReducer)"""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.fn()
        return g, None


def fork_ok(*ts):
    """run a branch that reads `ts` on the second stream?  (CUDA tensors, and a batch small enough to leave power headroom)"""
    return STREAMS_ON and all(t is None or (torch.is_tensor(t) and t.is_cuda) for t in ts) and ts[0] is not None and \
        ts[0].shape[0] <= FORK_MAX_BATCH


def to_nhwc_bf16(x_nchw, pad_to=1):
    """NCHW (any float dtype) -> NHWC bf16, channels zero-padded up to a multiple of `pad_to`"""
    x = x_nchw.permute(0, 2, 3, 1)
    c = x.shape[3]
    cp = _ceil(c, pad_to)
    if cp != c:
        x = F.pad(x, (0, cp - c))
    return x.contiguous().to(_act())


class PackNHWC8(torch.autograd.Function):
    """cat((x, pos)) -> NHWC bf16 with 8 channels in one pass (csrc/gan_elem.hip k_pack_nhwc8); x [N,C,H,W] fp32,
    pos [P,H,W] fp32 or None, C + P <= 8"""

    @staticmethod
    def forward(ctx, x, pos):
        x = x.contiguous()
        n, c, h, w = x.shape
        p = 0 if pos is None else pos.shape[0]
        out = torch.empty((n, h, w, 8), dtype=_act(), device=x.device)
        launch("pack_nhwc8", ptr(x), ptr(pos), ptr(out), n, c, p, h, w, stream())
        ctx.shape = (n, c, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=g.device)
        launch("unpack_nhwc8", ptr(g.contiguous()), ptr(dx), n, c, h, w, stream())
        return dx, None


def pack_nhwc8(x_nchw, pos=None):
    """the discriminators' input assembly; falls back to torch ops off the GPU / for other dtypes"""
    if x_nchw.is_cuda and x_nchw.dtype == torch.float32 and x_nchw.shape[1] + (0 if pos is None else pos.shape[0]) <= 8:
        return PackNHWC8.apply(x_nchw, None if pos is None else pos.contiguous())
    if pos is not None:
        x_nchw = torch.cat((x_nchw, pos.unsqueeze(0).expand(x_nchw.shape[0], -1, -1, -1)), dim=1)
    return to_nhwc_bf16(x_nchw, pad_to=8)


def to_nchw_f32(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).float()


# ------------------------------------------------------------------------------------------------ input / output glue
def _plain_f32(*ts):
    return all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in ts)


class MaskCatFn(torch.autograd.Function):
    """X = cat(fake * alpha, alpha) [and cat((that, cat(real, alpha)), dim=0)] in one pass (main.py:493,503-507)"""

    @staticmethod
    def forward(ctx, fake, alpha, real):
        fake, alpha = fake.contiguous(), alpha.contiguous()
        n, _, h, w = fake.shape
        real = None if real is None else real.contiguous()
        X = torch.empty(((1 if real is None else 2) * n, 4, h, w), dtype=torch.float32, device=fake.device)
        launch("mask_cat_fwd", ptr(fake), ptr(real), ptr(alpha), ptr(X), n, h, w, stream())
        ctx.save_for_backward(alpha)
        return X

    @staticmethod
    def backward(ctx, dX):
        (alpha,) = ctx.saved_tensors
        n, _, h, w = alpha.shape
        dfake = torch.empty((n, 3, h, w), dtype=torch.float32, device=dX.device)
        launch("mask_cat_bwd", ptr(dX.contiguous()), ptr(alpha), ptr(dfake), n, h, w, stream())
        return dfake, None, None


def mask_cat(fake, alpha, real=None):
    """discriminator input of main.py:493 (real=None) / :503-507 (fake and real halves stacked on the batch axis)"""
    ok = _plain_f32(fake, alpha, real) and fake.dim() == 4 and fake.shape[1] == 3 and \
        tuple(alpha.shape) == (fake.shape[0], 1, fake.shape[2], fake.shape[3]) and \
        (fake.shape[2] * fake.shape[3]) % 4 == 0 and (real is None or real.shape == fake.shape)   # (a broadcast alpha: torch path)
    if ok:
        return MaskCatFn.apply(fake, alpha, real)
    x = torch.cat((fake * alpha, alpha), dim=1)
    return x if real is None else torch.cat((x, torch.cat((real, alpha), dim=1)), dim=0)


class MaskedInput:
    """The discriminator input of main.py:493 / 503-507 -- cat(fake * alpha, alpha) [stacked on cat(real, alpha)] -- NOT yet
    assembled: MultiScaleDiscriminator.forward builds every member's packed conv1 input straight from the three pieces
    (DiscPartsFn: the [N|2N,4,H,W] tensor is never written or re-read); anything else calls materialize()."""

    def __init__(self, fake, alpha, real=None):
        self.fake, self.alpha, self.real = fake, alpha, real

    @property
    def shape(self):
        n, _, h, w = self.fake.shape
        return torch.Size(((1 if self.real is None else 2) * n, 4, h, w))

    @property
    def device(self):
        return self.fake.device

    def fusable(self):
        f, a, r = self.fake, self.alpha, self.real
        return _plain_f32(f, a, r) and f.dim() == 4 and f.shape[1] == 3 and tuple(a.shape) == (f.shape[0], 1, f.shape[2], f.shape[3]) \
            and f.shape[3] % 4 == 0 and (r is None or r.shape == f.shape)

    def materialize(self):
        return mask_cat(self.fake, self.alpha, self.real)


class DiscPartsFn(torch.autograd.Function):
    """DiscInputsFn on a MaskedInput: MaskCatFn folded into the loaders (forward) and into the unpack (backward: only `fake`
    takes a gradient).  Same bits as the two-stage form."""

    @staticmethod
    def forward(ctx, fake, alpha, real, extra, specs):
        fake, alpha = fake.contiguous(), alpha.contiguous()
        real = None if real is None else real.contiguous()
        n, _, h, w = fake.shape
        m = n if real is None else 2 * n
        outs, masks = [], []
        for f, has_extra, pos, cp, g in specs:
            ho, wo = h // f, w // f
            e = extra.contiguous() if has_extra else None
            out = torch.empty((m, ho, wo, cp), dtype=_act(), device=fake.device)
            mask = torch.empty((m, 1, ho // g, wo // g), dtype=torch.float32, device=fake.device) if g else None
            launch("pool_pack_parts_fwd", ptr(fake), ptr(real), ptr(alpha), n, m, h, w, f, ptr(e), 0 if e is None else e.shape[1],
                   ptr(pos), 0 if pos is None else pos.shape[0], ptr(out), cp, ptr(mask), g, stream())
            outs.append(out)
            masks.append(mask)
        ctx.save_for_backward(alpha)
        ctx.shape, ctx.specs, ctx.has_real = (n, h, w), specs, real is not None
        ctx.eshape = None if extra is None else tuple(extra.shape)
        res = tuple(outs) + tuple(mk for mk in masks if mk is not None)
        ctx.mark_non_differentiable(*[mk for mk in masks if mk is not None])
        ctx.nout = len(outs)
        return res

    @staticmethod
    def backward(ctx, *grads):
        (alpha,) = ctx.saved_tensors
        n, h, w = ctx.shape
        dhs = grads[:ctx.nout]
        live = [(g.contiguous(), sp[0], sp[3]) for g, sp in zip(dhs, ctx.specs) if g is not None]
        dfake = dextra = None
        if ctx.needs_input_grad[0] and live:
            if ctx.has_real:   # (a gradient into the fake half of a [fake; real] batch: not a case the trainer has)
                raise RuntimeError("DiscPartsFn: backward to `fake` with a real half stacked behind it -- materialize() the input")
            dfake = torch.empty((n, 3, h, w), dtype=torch.float32, device=live[0][0].device)
            a = []
            for k in range(3):
                a += [ptr(live[k][0]), live[k][1], live[k][2]] if k < len(live) else [None, 1, 8]
            launch("pool_unpack_parts_bwd", *a, ptr(alpha), ptr(dfake), n, h, w, stream())
        if ctx.eshape is not None and ctx.needs_input_grad[3]:
            m = ctx.eshape[0]
            for g, sp in zip(dhs, ctx.specs):
                if sp[1] and g is not None:
                    _, e, eh, ew = ctx.eshape
                    part = torch.empty(ctx.eshape, dtype=torch.float32, device=g.device)
                    launch("unpack_range", ptr(g.contiguous()), ptr(part), m, eh * ew, sp[3], 4, e, stream())
                    dextra = part if dextra is None else dextra + part
        return dfake, None, None, dextra, None


class DiscInputsFn(torch.autograd.Function):
    """What the member discriminators do to their input before conv1 (gan.py:79-99, 192-211), for all members at once:
    per member k: h_k = NHWC bf16 of cat(avg_pool2d(x, f_k), extra_k, pos_k) (channels padded to 8 / 16) and, with
    g_k > 0, mask_k = avg_pool2d(avg_pool2d(x, f_k)[:, 3:4], g_k).  One launch per member forward, ONE backward launch
    (+ one for the mesh map).  specs: tuples (f, has_extra, pos [P,h,w] | None, CP, g)."""

    @staticmethod
    def forward(ctx, x, extra, specs):
        x = x.contiguous()
        m, c, h, w = x.shape
        outs, masks = [], []
        for f, has_extra, pos, cp, g in specs:
            ho, wo = h // f, w // f
            e = extra.contiguous() if has_extra else None
            out = torch.empty((m, ho, wo, cp), dtype=_act(), device=x.device)
            mask = torch.empty((m, 1, ho // g, wo // g), dtype=torch.float32, device=x.device) if g else None
            launch("pool_pack_fwd", ptr(x), m, c, h, w, f, ptr(e), 0 if e is None else e.shape[1], ptr(pos),
                   0 if pos is None else pos.shape[0], ptr(out), cp, ptr(mask), 3, g, stream())
            outs.append(out)
            masks.append(mask)
        ctx.shape, ctx.specs = (m, c, h, w), specs
        ctx.eshape = None if extra is None else tuple(extra.shape)
        res = tuple(outs) + tuple(mk for mk in masks if mk is not None)
        ctx.mark_non_differentiable(*[mk for mk in masks if mk is not None])
        ctx.nout = len(outs)
        return res

    @staticmethod
    def backward(ctx, *grads):
        m, c, h, w = ctx.shape
        dhs = grads[:ctx.nout]
        live = [(g.contiguous(), sp[0], sp[3]) for g, sp in zip(dhs, ctx.specs) if g is not None]
        dx = dextra = None
        if ctx.needs_input_grad[0] and live:
            dx = torch.empty((m, c, h, w), dtype=torch.float32, device=live[0][0].device)
            a = []
            for k in range(3):
                a += [ptr(live[k][0]), live[k][1], live[k][2]] if k < len(live) else [None, 1, 8]
            launch("pool_unpack_bwd", *a, ptr(dx), m, c, h, w, stream())
        if ctx.eshape is not None and ctx.needs_input_grad[1]:
            for g, sp in zip(dhs, ctx.specs):
                if sp[1] and g is not None:
                    _, e, eh, ew = ctx.eshape
                    part = torch.empty(ctx.eshape, dtype=torch.float32, device=g.device)
                    launch("unpack_range", ptr(g.contiguous()), ptr(part), m, eh * ew, sp[3], c, e, stream())
                    dextra = part if dextra is None else dextra + part   # (several members may take the mesh map)
        return dx, dextra, None


def disc_inputs_ok(x, extra, specs):
    if isinstance(x, MaskedInput):
        if not x.fusable() or not _plain_f32(extra) or (x.real is not None and x.fake.requires_grad and torch.is_grad_enabled()):
            return False
    elif not _plain_f32(x, extra) or x.dim() != 4 or x.shape[1] != 4:
        return False
    _, c, h, w = x.shape
    for f, has_extra, pos, cp, g in specs:
        e = extra.shape[1] if has_extra else 0
        p = 0 if pos is None else pos.shape[0]
        if not lib().m355_pool_pack_ok(c, h, w, f, e, p, g) or c + e + p > cp:
            return False
        if pos is not None and tuple(pos.shape[1:]) != (h // f, w // f):   # (the members cache their planes at the first call's size)
            return False
        if has_extra and tuple(extra.shape[2:]) != (h // f, w // f):
            return False
    return True


def disc_inputs(x, extra, specs):
    """-> ([h_k NHWC bf16], [mask_k or None]) for the member discriminators described by `specs`"""
    if isinstance(x, MaskedInput):
        res = DiscPartsFn.apply(x.fake, x.alpha, x.real, extra, tuple(specs))
    else:
        res = DiscInputsFn.apply(x, extra, tuple(specs))
    n = len(specs)
    hs, rest, masks = list(res[:n]), list(res[n:]), []
    for sp in specs:
        masks.append(rest.pop(0) if sp[4] else None)
    return hs, masks


HT_TANH, HT_POLES, HT_SYMM = 1, 2, 4


class HeadConvFn(torch.autograd.Function):
    """A generator head (gan.py:407-419): conv_final / conv_mesh (5x5, 64 -> 3, plain conv) followed by tanh_ /
    adjust_poles / symmetrize_texture, on NHWC bf16 input, NCHW fp32 output.  The tail is one elementwise kernel each
    way; its backward writes the conv's incoming gradient straight in the 8-channel NHWC bf16 layout the dgrad / wgrad
    kernels read, together with the bias gradient.  in_slope != 1: x is the output of a LeakyReLU(in_slope) fused into
    its producer (CbnActFn out_slope) whose backward is applied HERE to the returned grad_x."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad_h, pad_w, mode, flags, in_slope):
        n, h, w, cx = x.shape
        cout, cw, kh, kw = weight.shape
        d = C.make_desc(n, h, w, cx, cout, kh, kw, 1, pad_h, pad_w, mode, 0)
        need_dx = ctx.needs_input_grad[0]
        wf, wd = C.weight_prep(d, weight, want_dgrad=need_dx)
        y = C.conv_fwd(d, x.detach(), wf, bias.detach(), True, 1.0, cin_real=cw)
        out = torch.empty((n, cout, h, 2 * w if flags & HT_SYMM else w), dtype=torch.float32, device=x.device)
        launch("head_tail_fwd", ptr(y), ptr(out), n, cout, h, w, flags, stream())
        ctx.d, ctx.cw, ctx.flags, ctx.in_slope = d, cw, flags, in_slope
        ctx.wparam = weight if isinstance(weight, torch.nn.Parameter) else None   # (conv.deferred_wgrad_finish)
        ctx.save_for_backward(x.detach(), wd, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wd, out = ctx.saved_tensors
        d = ctx.d
        g = torch.empty((d.N, d.H, d.W, 8), dtype=_act(), device=x.device)
        db = torch.empty((d.Cout,), dtype=torch.float32, device=x.device)
        ws = torch.empty((_lib.HEAD_TAIL_WS_FLOATS,), dtype=torch.float32, device=x.device)
        launch("head_tail_bwd", ptr(dout.contiguous()), ptr(out), ptr(g), ptr(db), ptr(ws), d.N, d.Cout, d.H, d.W, ctx.flags, stream())
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if ctx.in_slope != 1.0 and C.dgrad_mask_ok(d):
                dx = C.conv_dgrad(d, g, wd, cin_real=ctx.cw, mask_x=x, mask_slope=ctx.in_slope)
            else:
                dx = C.conv_dgrad(d, g, wd, cin_real=ctx.cw)
                if ctx.in_slope != 1.0:   # (shapes whose dgrad has no fused activation backward)
                    dx = lrelu_bwd(dx, x, ctx.in_slope)[0]
        if ctx.needs_input_grad[1]:
            dw = C.wgrad_finish(d, C.conv_wgrad(d, x, g, cin_real=ctx.cw, raw=True, arena=True), ctx.cw, param=ctx.wparam)
        return dx, dw, (db if ctx.needs_input_grad[2] else None), None, None, None, None, None


def head_conv(x, conv, flags, in_slope=1.0):
    """x NHWC bf16 -> head output NCHW fp32 (see HeadConvFn); `conv` is a gan.Conv2d without spectral norm"""
    stride, pad_h, pad_w, mode = conv.m355
    assert stride == 1 and conv.bias is not None and "weight_orig" not in conv._parameters
    return HeadConvFn.apply(x, conv.weight, conv.bias, pad_h, pad_w, mode, int(flags), float(in_slope))


class HingeLossFn(torch.autograd.Function):
    """GANLoss('hinge') over a list of discriminator outputs (utils/losses.py:49-120) in one launch each way.
    mode 1: returns (loss on samples [0, split) against target False, loss on [split, B) against target True);
    mode 0 (generator): returns (-masked mean, unused)."""

    @staticmethod
    def forward(ctx, mode, split, weights, K, *ts):
        import ctypes
        preds = [t.contiguous() for t in ts[:K]]
        masks = [None if t is None else t.contiguous() for t in ts[K:]]
        B = preds[0].shape[0]
        dev = preds[0].device
        hw = [p[0].numel() for p in preds]
        PA = ctypes.c_void_p * 3
        pa = PA(*[ptr(p) for p in preds] + [None] * (3 - K))
        ma = PA(*[ptr(m) for m in masks] + [None] * (3 - K))
        ha = (ctypes.c_int * 3)(*hw + [0] * (3 - K))
        wa = None if weights is None else (ctypes.c_float * 3)(*[float(v) for v in weights] + [0.0] * (3 - K))
        loss2 = torch.empty(2, dtype=torch.float32, device=dev)
        msum = torch.empty((2 * K, B), dtype=torch.float32, device=dev)   # [mask sums | loss terms (scratch of the ordered sum)]
        launch("hinge_fwd", K, pa, ma, ha, wa, B, split, mode, ptr(loss2), ptr(msum), stream())
        ctx.cfg = (mode, split, weights, K, B, hw)
        ctx.save_for_backward(msum, *preds, *[m for m in masks if m is not None])
        ctx.mask_present = [m is not None for m in masks]
        return loss2[0:1], loss2[1:2]

    @staticmethod
    def backward(ctx, g0, g1):
        import ctypes
        mode, split, weights, K, B, hw = ctx.cfg
        msum, *rest = ctx.saved_tensors
        preds, mlist = rest[:K], list(rest[K:])
        masks = [mlist.pop(0) if present else None for present in ctx.mask_present]
        dev = preds[0].device
        gl = torch.cat((g0 if g0 is not None else torch.zeros(1, device=dev), g1 if g1 is not None else torch.zeros(1, device=dev)))
        dps = [torch.empty_like(p) for p in preds]
        PA = ctypes.c_void_p * 3
        pa = PA(*[ptr(p) for p in preds] + [None] * (3 - K))
        ma = PA(*[ptr(m) for m in masks] + [None] * (3 - K))
        da = PA(*[ptr(t) for t in dps] + [None] * (3 - K))
        ha = (ctypes.c_int * 3)(*hw + [0] * (3 - K))
        wa = None if weights is None else (ctypes.c_float * 3)(*[float(v) for v in weights] + [0.0] * (3 - K))
        launch("hinge_bwd", K, pa, ma, ha, wa, B, split, mode, ptr(gl.contiguous()), ptr(msum), da, stream())
        return (None, None, None, None) + tuple(dps) + (None,) * K


def hinge_ok(preds, masks):
    if not isinstance(preds, (list, tuple)) or not 1 <= len(preds) <= 3:
        return False
    if not all(torch.is_tensor(p) and p.is_cuda and p.dtype == torch.float32 and p.dim() == 4 and p.shape[1] == 1 for p in preds):
        return False
    if len({p.shape[0] for p in preds}) != 1:
        return False
    if masks is not None:
        for p, m in zip(preds, masks):
            if m is not None and not (m.is_cuda and m.dtype == torch.float32 and m.shape == p.shape):
                return False
    return True


def hinge_losses(preds, masks, weights, mode, split):
    ms = [None] * len(preds) if masks is None else list(masks)
    return HingeLossFn.apply(int(mode), int(split), None if weights is None else tuple(weights), len(preds), *preds, *ms)


def upsample2x(x):
    """nearest x2 of an NHWC tensor (F.interpolate(scale_factor=2, mode='nearest'), gan.py:319)"""
    n, h, w, c = x.shape
    return x[:, :, None, :, None, :].expand(n, h, 2, w, 2, c).reshape(n, 2 * h, 2 * w, c)


def leaky_relu(x, slope):
    return F.leaky_relu(x, slope)


class Conv2dFn(torch.autograd.Function):
    """F.conv2d(pad(up(x)), w / sigma, b) [+ LeakyReLU] on NHWC bf16 through libm355.  `sn` is None or the
    spectral-norm state of this forward (SpectralNormGroup.step): the division by sigma happens inside the bf16
    weight-view kernel and the gradient with respect to weight_orig (through sigma) inside the kernel that lays
    out the weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad_h, pad_w, mode, ups, slope, out_f32_nchw, sn, in_slope=1.0,
                premasked=False, in_bits=None, want_stats=False, dx_lead=0, dx_pair=None):
        n, h, w, cx = x.shape
        cout, cw, kh, kw = weight.shape
        if cx % 8 or cx < cw:
            raise ValueError(f"conv2d: input has {cx} channels (must be a multiple of 8 and >= {cw})")
        d = C.make_desc(n, h, w, cx, cout, kh, kw, stride, pad_h, pad_w, mode, ups)
        need_dx = ctx.needs_input_grad[0]
        sigma = None if sn is None else sn.sigma
        if sn is not None and sn.wkey == (cx, cout, kh, kw, stride, int(ups)) and (sn.wd is not None or not need_dx):
            wf, wd = sn.wf, (sn.wd if need_dx else None)   # prepared with the whole network's views (SpectralNormGroup.step)
        else:
            wf, wd = C.weight_prep(d, weight, want_dgrad=need_dx, sigma=sigma)
        # premasked = the only consumer's dgrad applies this activation's backward: hand it 1 bit per element instead of
        # making it re-read the bf16 activation (csrc/conv_dma.h: ConvArgs::bits_out)
        bits = part = None
        rows = C.conv_stats_rows(d) if (want_stats and slope == 1.0 and not out_f32_nchw) else 0
        if rows:
            # the consumer is a batch norm in training mode: its (sum, sum of squares) partials come out of this launch
            y, part = C.conv_fwd_stats(d, x.detach(), wf, None if bias is None else bias.detach(), cin_real=cw, rows=rows)
            ctx.mark_non_differentiable(part)
        elif premasked and slope != 1.0 and not out_f32_nchw and any(ctx.needs_input_grad[:3]) and C.maskbits_ok(d, 0):
            y, bits = C.conv_fwd(d, x.detach(), wf, None if bias is None else bias.detach(), False, slope, cin_real=cw,
                                 emit_bits=True)
            ctx.mark_non_differentiable(bits)
        else:
            y = C.conv_fwd(d, x.detach(), wf, None if bias is None else bias.detach(), out_f32_nchw, slope, cin_real=cw)
        ctx.d, ctx.cw, ctx.slope, ctx.f32, ctx.sn = d, cw, slope, out_f32_nchw, sn
        ctx.in_slope, ctx.premasked, ctx.dx_lead, ctx.dx_pair = in_slope, premasked, int(dx_lead), dx_pair
        # `bits` / `part` never get a gradient: without this autograd hands backward a zero-FILLED tensor of their shape for each
        # (17 fills per GAN cycle, the bit masks of D.conv2 at batch 128 alone 67 MB)
        ctx.set_materialize_grads(False)
        ctx.wparam = weight if isinstance(weight, torch.nn.Parameter) else None   # (conv.deferred_wgrad_finish)
        ctx.has_bias = bias is not None
        use_bits = in_bits is not None and in_slope != 1.0 and C.maskbits_ok(d, 1)
        ctx.save_for_backward(x.detach(), wd, y if (slope != 1.0 and not premasked) else None,
                              weight.detach() if sn is not None else None, in_bits if use_bits else None)
        return y, bits, part

    @staticmethod
    def backward(ctx, dy, _dbits=None, _dpart=None):
        if dy is None:   # (set_materialize_grads(False): nobody used y)
            return (None,) * 17
        x, wd, y, w_orig, in_bits = ctx.saved_tensors
        d = ctx.d
        if ctx.sn is not None:
            # wd / sigma / u / v are slots of the group, overwritten in place by later forwards (invisible to autograd's
            # version counters): both the dgrad and the wgrad branch must see this forward's snapshot
            ctx.sn.check()
        c32 = C.dy_channels(d.Cout)
        db_zeroed = False
        fused_tail = False
        if ctx.premasked:
            # the consumer's dgrad already applied this layer's LeakyReLU derivative (mask_x below): dy IS g
            assert not ctx.f32 and c32 == d.Cout
            g = dy.contiguous()
            db = None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                if ctx.needs_input_grad[1] and C.wgrad_fuses_dbias(d):
                    # accumulated by the wgrad kernel: a zeroed slice of the pass's bias-gradient block, or our own buffer
                    db = C.DbiasBlock.take(d.Cout, g.device)
                    db_zeroed = db is not None
                    if db is None:
                        db = torch.empty((d.Cout,), dtype=torch.float32, device=g.device)
                else:
                    db = chan_sum(g)
        elif ctx.slope != 1.0 and not ctx.f32 and c32 == d.Cout and 256 % (d.Cout // 8) == 0:
            # LeakyReLU epilogue backward + bias gradient in one pass (csrc/gan_elem.hip)
            g, db = lrelu_bwd(dy.contiguous(), y, ctx.slope)
            if not (ctx.has_bias and ctx.needs_input_grad[2]):
                db = None
        elif ctx.f32 and ctx.slope == 1.0 and d.Cout <= 8 and dy.is_cuda and dy.dtype == torch.float32:
            # a logit head (D.conv5: one channel, fp32 NCHW): NCHW -> zero-padded 8-channel NHWC bf16 in ONE pass (k_pack_nhwc8)
            # instead of permute / pad (fill + copy) / cast
            dyc = dy.contiguous()
            db = dyc.sum((0, 2, 3)) if ctx.has_bias and ctx.needs_input_grad[2] else None
            g = torch.empty((d.N, dyc.shape[2], dyc.shape[3], 8), dtype=_act(), device=dy.device)
            launch("pack_nhwc8", ptr(dyc), ptr(None), ptr(g), d.N, d.Cout, 0, dyc.shape[2], dyc.shape[3], stream())
            pair = ctx.dx_pair
            if (pair is not None and not pair.proj_done and ctx.needs_input_grad[0] and ctx.in_slope != 1.0 and in_bits is None
                    and d.Cout == 1):
                # the tail of a discriminator (TailPair): the projection term's backward, which runs next, produces the whole
                # gradient of x in one pass from this conv's logit gradient and dgrad weights -- no dgrad launch here
                kp = int(C.plan(d).w_dgrad_row_elems)        # row length of the stride-1 dgrad view: the library's rule, asked once
                pair.request = (dyc.view(d.N, dyc.shape[2], dyc.shape[3]), wd, kp, d.pad_w_mode, ctx.in_slope)
                fused_tail = True
        else:
            g = dy.permute(0, 2, 3, 1) if ctx.f32 else dy       # -> NHWC view
            if ctx.slope != 1.0:
                yy = y.permute(0, 2, 3, 1) if ctx.f32 else y
                g = g * torch.where(yy > 0, 1.0, ctx.slope).to(g.dtype)
            db = g.float().sum((0, 1, 2)) if ctx.has_bias and ctx.needs_input_grad[2] else None
            if c32 != d.Cout:
                g = F.pad(g, (0, c32 - d.Cout))
            g = g.contiguous().to(_act())
        dx = None
        if ctx.needs_input_grad[0] and not fused_tail:
            if ctx.in_slope != 1.0 and in_bits is not None:
                dx = C.conv_dgrad(d, g, wd, cin_real=ctx.cw, mask_bits=in_bits, mask_slope=ctx.in_slope)
            elif ctx.in_slope != 1.0:   # fold the LeakyReLU backward of the layer that produced x into the epilogue
                dx = C.conv_dgrad(d, g, wd, cin_real=ctx.cw, mask_x=x, mask_slope=ctx.in_slope)
            else:
                dx = C.conv_dgrad(d, g, wd, cin_real=ctx.cw, lead=ctx.dx_lead)
        dw = None
        if ctx.needs_input_grad[1]:
            fused_db = db if (ctx.premasked and db is not None and C.wgrad_fuses_dbias(d)) else None
            graw = C.conv_wgrad(d, x, g, cin_real=ctx.cw, raw=True, dbias=fused_db, arena=True,
                                dbias_zeroed=fused_db is not None and db_zeroed)
            sn = ctx.sn
            if sn is None:
                dw = C.wgrad_finish(d, graw, ctx.cw, param=ctx.wparam)
            else:
                dw = C.wgrad_finish(d, graw, ctx.cw, w_orig, sn.u, sn.v, sn.sigma, param=ctx.wparam)
        return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None, None, None, None


def conv2d(x, weight, bias, stride, pad_h, pad_w, mode, upsample=0, slope=1.0, out_f32_nchw=False, sn=None, in_slope=1.0,
           premasked=False, want_stats=False, dx_lead=0, dx_pair=None):
    """in_slope != 1: x is the output of a fused conv+LeakyReLU(in_slope) whose ONLY consumer is this conv: the
    returned grad_x is pre-multiplied by that activation's derivative, and that producer must be called with
    premasked=True (it then skips its own activation backward).  Both flags are set by the discriminators."""
    """want_stats: the output's only use is a batch norm in training mode -- returns (y, part): where the kernel can, part is the
    norm's [rows,2,C] partial sums produced by the conv launch (else None); the caller passes it to BatchNorm2d.forward(part=)"""
    y, bits, part = Conv2dFn.apply(x, weight, bias, stride, pad_h, pad_w, mode, int(upsample), float(slope), bool(out_f32_nchw),
                                   sn, float(in_slope), bool(premasked),
                                   getattr(x, "_m355_bits", None) if in_slope != 1.0 else None, bool(want_stats), int(dx_lead),
                                   dx_pair)
    if bits is not None:
        y._m355_bits = bits  # picked up by the consumer conv (same Python tensor object, see the discriminators' _act)
    if want_stats:
        return y, part   # (part None: this shape has no fused statistics) -- handed to the norm EXPLICITLY, not on the tensor
    return y


# ------------------------------------------------------------------------------------------------ spectral norm
class _SnState:
    """what one conv needs from one SpectralNormGroup.step(): views of that step's sigma / u / v"""
    __slots__ = ("sigma", "u", "v", "_slot", "_version", "wf", "wd", "wkey")

    def __init__(self, sigma, u, v, slot, version, wf=None, wd=None, wkey=None):
        self.sigma, self.u, self.v, self._slot, self._version = sigma, u, v, slot, version
        # bf16 GEMM views of weight_orig / sigma prepared by the group's batched launch, valid for descs with `wkey`
        self.wf, self.wd, self.wkey = wf, wd, wkey

    def check(self):
        if self._slot["version"] != self._version:
            raise RuntimeError("spectral norm: the power-iteration snapshot of this forward was overwritten by later "
                               "forwards before its backward ran (more than %d forwards in flight)" % SpectralNormGroup.SLOTS)


class SpectralNormGroup:
    """torch.nn.utils.spectral_norm for all convs of a network at once (csrc/gan_glue.hip): one power iteration per
    training forward, three launches for the whole group.  The convs keep torch's parametrisation
    (weight_orig / weight_u / weight_v, same state_dict keys and initialisation); only the per-module forward
    pre-hook is replaced by `step()`."""
    SLOTS = 4
    _ENTRY = 64  # sizeof(m355_sn_layer)

    def __init__(self, convs, eps=1e-12):
        self.convs = [c for c in convs if hasattr(c, "weight_orig")]
        self.eps = eps
        self._key = None
        self._slots = None
        self._next = 0
        self._pending = None   # a step computed ahead of its forward on the prefetch stream (prefetch())
        self._uv_backup = None

    def _build(self, dev):
        import struct
        rows = [c.weight_orig.shape[0] for c in self.convs]
        cols = [c.weight_orig[0].numel() for c in self.convs]
        self._max = (max(rows), max(cols))
        nr, nc = sum(rows), sum(cols)
        self._scratch = torch.empty(nr + nc, dtype=torch.float32, device=dev)
        # per-workgroup partial sums + tickets of the power iteration (zero once: the kernels leave the tickets zero)
        self._norms = torch.zeros(lib().m355_sn_scratch_words(len(self.convs), self._max[0], self._max[1]), dtype=torch.float32,
                                  device=dev)
        self._slots = []
        for _ in range(self.SLOTS):
            snap = torch.empty(nr + nc, dtype=torch.float32, device=dev)
            sigma = torch.empty(len(self.convs), dtype=torch.float32, device=dev)
            raw = bytearray()
            views = []
            ro, co = 0, nr
            for i, c in enumerate(self.convs):
                u_s, v_s = snap[ro:ro + rows[i]], snap[co:co + cols[i]]
                raw += struct.pack("<7Q2i", c.weight_orig.data_ptr(), c.weight_u.data_ptr(), c.weight_v.data_ptr(),
                                   self._scratch[co:].data_ptr(), self._scratch[ro:].data_ptr(), u_s.data_ptr(),
                                   v_s.data_ptr(), rows[i], cols[i])
                views.append((sigma[i:i + 1], u_s, v_s))
                ro += rows[i]
                co += cols[i]
            table = torch.frombuffer(raw, dtype=torch.uint8).to(dev)
            slot = {"table": table, "snap": snap, "sigma": sigma, "views": views, "version": 0}
            self._build_weight_views(slot, dev)
            self._slots.append(slot)

    def _build_weight_views(self, slot, dev):
        """per slot: the bf16 forward / dgrad views of every conv (m355_conv2d_weight_prep's outputs) and the device
        table that produces all of them in one launch (m355_weight_prep_batched)"""
        import ctypes
        L = lib()
        esz = L.m355_weight_prep_entry_bytes()
        raw = (ctypes.c_char * (esz * len(self.convs)))()
        weights, most, tiles = [], 0, [0, 0]
        for i, c in enumerate(self.convs):
            cout, cw, kh, kw = c.weight_orig.shape
            stride, pad_h, pad_w, mode = c.m355
            cx = (cw + 7) // 8 * 8
            ups = int(getattr(c, "m355_ups", 0))   # (layers called with the upsample folded in carry the sub-pixel views too)
            d = C.make_desc(1, 64, 64, cx, cout, kh, kw, stride, pad_h, pad_w, mode, ups)
            wf = torch.empty((L.m355_conv2d_weight_elems(ctypes.byref(d), 0),), dtype=torch.bfloat16, device=dev)
            wd = torch.empty((L.m355_conv2d_weight_elems(ctypes.byref(d), 1),), dtype=torch.bfloat16, device=dev)
            n = L.m355_weight_prep_fill_entry(ctypes.byref(d), ptr(c.weight_orig), int(cw), ptr(slot["sigma"][i:i + 1]), ptr(wf),
                                              ptr(wd), ctypes.byref(raw, esz * i))
            if n < 0:
                check(1, "weight_prep_fill_entry")
            nt = L.m355_weight_prep_entry_tiles(ctypes.byref(raw, esz * i))   # > 0: the LDS-transpose kernel takes this layer
            if nt > 0:
                tiles[0 if kh * kw <= 9 else 1] = max(tiles[0 if kh * kw <= 9 else 1], nt)
            else:
                most = max(most, n)
            weights.append((wf, wd, (cx, cout, kh, kw, stride, ups)))
        slot["wtable"] = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).to(dev)
        slot["weights"], slot["wmost"], slot["wtiles"] = weights, most, tiles

    def _ensure_built(self):
        key = tuple(t.data_ptr() for c in self.convs for t in (c.weight_orig, c.weight_u, c.weight_v)) + \
            tuple(int(getattr(c, "m355_ups", 0)) for c in self.convs) + (_act(),)   # (views belong to one build of the library)
        if key != self._key:
            self._pending = None   # (computed from tensors that no longer are the network's)
            self._build(self.convs[0].weight_orig.device)
            self._key = key

    def _take_slot(self):
        slot = self._slots[self._next]
        self._next = (self._next + 1) % self.SLOTS
        slot["version"] += 1
        return slot

    def _launch(self, slot, training):
        launch("sn_power_iter", ptr(slot["table"]), len(self.convs), self._max[0], self._max[1], ptr(self._norms),
               ptr(slot["sigma"]), int(bool(training)), float(self.eps), stream())
        launch("weight_prep_batched_tiled", ptr(slot["wtable"]), len(self.convs), int(slot["wmost"]), int(slot["wtiles"][0]), int(slot["wtiles"][1]), stream())

    def _hand_out(self, slot):
        for c, (sg, u, v), (wf, wd, wkey) in zip(self.convs, slot["views"], slot["weights"]):
            c._sn_state = _SnState(sg, u, v, slot, slot["version"], wf, wd, wkey)

    def _versions(self):
        """autograd version counters of everything a step reads: in-place updates from outside (optimiser step, load_state_dict,
        copy_) bump them; the library's own writes to u / v go through raw pointers and do not"""
        return tuple(t._version for c in self.convs for t in (c.weight_orig, c.weight_u, c.weight_v))

    def prefetch(self, training=True):
        """Run the NEXT forward's step now, on the prefetch stream, under whatever the main stream does in the meantime -- its 0.12 ms
        (generator) / 0.07 ms (discriminators) of small dependent launches are batch-independent weight work that otherwise sits
        serially in front of the network's first conv.  Legal exactly when nothing the step reads changes until
        that forward: the caller (GanTrainer) issues it where the weights are final -- after the optimiser step of THIS network, or
        right after a forward when no optimiser step of this network lies before the next one.  step() consumes the result if the
        version counters of weight_orig / u / v still are what they were here and the mode matches; otherwise u / v are restored
        from the copy taken below and the step runs in line -- the prefetch is an execution detail, never a change of arithmetic:
        same kernels, same inputs, same bits (tests/test_gan_modules.py::test_spectral_norm_prefetch_changes_no_bit)."""
        if not SN_PREFETCH_ON or not self.convs or self._pending is not None:
            return False
        dev = self.convs[0].weight_orig.device
        if dev.type != "cuda":
            return False
        self._ensure_built()
        main = torch.cuda.current_stream(dev)
        side = _SN_SIDE.get(dev.index)
        if side is None:
            side = _SN_SIDE[dev.index] = torch.cuda.Stream(dev)
        slot = self._take_slot()
        uv = [t for c in self.convs for t in (c.weight_u, c.weight_v)]
        if self._uv_backup is None or len(self._uv_backup) != len(uv) or self._uv_backup[0].device != dev:
            self._uv_backup = [torch.empty_like(t) for t in uv]
        side.wait_stream(main)            # (the weights' last writer -- the optimiser step -- and the previous step ran on `main`)
        with torch.cuda.stream(side), torch.no_grad():
            if training:
                torch._foreach_copy_(self._uv_backup, uv)   # what step() puts back if this result turns out stale
            self._launch(slot, training)
            ev = torch.cuda.Event()
            ev.record(side)
        self._pending = (slot, bool(training), self._versions(), ev, uv)
        return True

    def cancel_prefetch(self):
        """undo a pending prefetch: u / v back to their values before it (checkpoints, hipGraph capture, snapshots)"""
        p, self._pending = self._pending, None
        if p is None:
            return
        slot, ptrain, _vers, ev, uv = p
        torch.cuda.current_stream(uv[0].device).wait_event(ev)
        if ptrain:
            with torch.no_grad():
                torch._foreach_copy_(uv, self._uv_backup)

    def step(self, training):
        """advance (training) / evaluate sigma for every conv of the group and hand each conv its state"""
        if not self.convs:
            return
        self._ensure_built()
        p = self._pending
        if p is not None:
            if p[1] == bool(training) and p[2] == self._versions():
                self._pending = None
                torch.cuda.current_stream(p[4][0].device).wait_event(p[3])
                self._hand_out(p[0])
                return
            self.cancel_prefetch()   # stale (weights / u / v written since, or the other mode): as if it had never run
        slot = self._take_slot()
        self._launch(slot, training)
        self._hand_out(slot)


def strip_sn_hook(conv):
    """nn.utils.spectral_norm registers a forward pre-hook that recomputes `weight` with ~13 tiny launches; the
    group kernel replaces it (state_dict hooks, parameter and buffer names stay)."""
    for k, h in list(conv._forward_pre_hooks.items()):
        if type(h).__name__ == "SpectralNorm":
            del conv._forward_pre_hooks[k]
    return conv


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


def chan_sum(x):
    """x [..., C] bf16 -> [C] fp32 sum over all leading dims"""
    C_ = x.shape[-1]
    P = x.numel() // C_
    out = torch.empty((C_,), dtype=torch.float32, device=x.device)
    launch("chan_sum", ptr(x), ptr(out), ptr(_ws(P, 1, 1, C_, x.device)), P, C_, stream())
    return out


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
        launch("affine_act_fwd", ptr(x), ptr(a), ptr(b), None, 0, ptr(y), n, h * w, c, float(slope), 1.0, stream())
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


class CbnActFn(torch.autograd.Function):
    """y = LeakyReLU(BN_batch(x) * (1 + gamma[n,c]) + beta[n,c]) [+ res] on NHWC bf16 with every piece of coefficient
    algebra inside libm355 (csrc/gan_glue.hip): 3 launches forward (partial sums, finalise, apply), 3 backward.
    gamma / beta are [N,C] views (unit channel stride) of the batched fc_gamma / fc_beta output.
    sync: statistics over all ranks (SynchronizedBatchNorm2d) -- one all-reduce of [sum | sumsq | count] forward and
    one of the two moment sums backward (RCCL), replacing code/sync_batchnorm/batchnorm.py:110-131's pipes."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, slope, res=None, sync=False, out_slope=1.0,
                part=None):
        """out_slope != 1: y = LeakyReLU_out_slope(...) on top (the activation in front of a generator head); its backward is
        NOT applied here -- the consuming head applies it to the gradient it returns (HeadConvFn in_slope).
        part: [rows,2,C] partial (sum, sum of squares) of x already produced by the conv that wrote x (conv_fwd_stats)"""
        n, h, w, c = x.shape
        x = x.contiguous()
        res_w = 0
        if res is not None:
            res = res.contiguous()
            assert res.dtype == x.dtype
            if res.shape != x.shape:   # half-resolution residual, read through the nearest x2 upsample
                assert tuple(res.shape) == (n, h // 2, w // 2, c) and h % 2 == 0 and w % 2 == 0, (res.shape, x.shape)
                res_w = w
        assert gamma.stride(1) == 1 and beta.stride(1) == 1 and gamma.stride(0) == beta.stride(0)
        dev = x.device
        P = n * h * w
        count = float(P)
        if part is not None:
            assert part.dtype == torch.float32 and part.dim() == 3 and tuple(part.shape[1:]) == (2, c) and part.is_contiguous()
            nblk = part.shape[0]
        else:
            nblk = lib().m355_chan_reduce_nblk(P, c)
            part = torch.empty((nblk, 2, c), dtype=torch.float32, device=dev)
            launch("bn_stats_partial", ptr(x), ptr(part), P, c, stream())
        cnt_dev = None
        if sync:
            # one all-reduce of [sum | sumsq | count]: the global pixel count comes back with the sums and stays on the
            # device (ragged shards are handled like the reference's _data_parallel_master, no host round trip)
            vec = torch.empty(2 * c + 1, dtype=torch.float32, device=dev)
            launch("bn_sync_pack", ptr(part), nblk, c, count, ptr(vec), stream())
            _sync_sum(vec, (running_mean.data_ptr(), "fwd"))
            _count_syncbn()
            part, nblk, cnt_dev = vec, 1, vec[2 * c:]
        coef = torch.empty((2 * n + 2, c), dtype=torch.float32, device=dev)   # a[N,C] | b[N,C] | mean | rstd
        a, b, mean, rstd = coef[:n], coef[n:2 * n], coef[2 * n], coef[2 * n + 1]
        launch("bn_finalize", ptr(part), nblk, count, ptr(cnt_dev), ptr(gamma), ptr(beta), int(gamma.stride(0)), n, c, float(eps),
               float(momentum), ptr(running_mean), ptr(running_var), ptr(mean), ptr(rstd), ptr(a), ptr(b), stream())
        y = torch.empty_like(x)
        launch("affine_act_fwd", ptr(x), ptr(a), ptr(b), ptr(res), res_w, ptr(y), n, h * w, c, float(slope), float(out_slope),
               stream())
        ctx.save_for_backward(x, coef, gamma, cnt_dev)
        ctx.cfg = (slope, count, (0 if res is None else (2 if res_w else 1)), sync)
        ctx.site = running_mean.data_ptr()   # (the layer: keys the exchange channel of its SyncBN messages, parallel.syncbn_all_reduce)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, coef, gamma, cnt_dev = ctx.saved_tensors
        slope, count, has_res, sync = ctx.cfg
        n, h, w, c = x.shape
        a, b, mean, rstd = coef[:n], coef[n:2 * n], coef[2 * n], coef[2 * n + 1]
        dy = dy.contiguous()
        dev = x.device
        nblk = lib().m355_chan_reduce_nblk(h * w, c)
        part = torch.empty((n, nblk, 2, c), dtype=torch.float32, device=dev)
        launch("affine_act_bwd_partial", ptr(dy), ptr(x), ptr(a), ptr(b), ptr(part), n, h * w, c, float(slope), stream())
        out = torch.empty((3 * n + 4, c), dtype=torch.float32, device=dev)   # dgamma | dbeta | A | Bc | Cc | m[2]
        dgamma, dbeta, A, Bc, Cc = out[:n], out[n:2 * n], out[2 * n:3 * n], out[3 * n], out[3 * n + 1]
        m = out[3 * n + 2:] if sync else None
        launch("bn_bwd_finalize", ptr(part), nblk, count, ptr(gamma), int(gamma.stride(0)), n, c, ptr(mean), ptr(rstd), 1,
               ptr(dgamma), ptr(dbeta), ptr(A), ptr(Bc), ptr(Cc), ptr(m), stream())
        if sync:
            _sync_sum(m, (ctx.site, "bwd"))
            _count_syncbn()
            launch("bn_bwd_coeffs", ptr(m), count, ptr(cnt_dev), ptr(mean), ptr(rstd), c, ptr(Bc), ptr(Cc), stream())
        dx = torch.empty_like(x)
        launch("affine_act_bwd_apply", ptr(dy), ptr(x), ptr(a), ptr(b), ptr(A), ptr(Bc), ptr(Cc), ptr(dx), n, h * w, c,
               float(slope), stream())
        dres = None
        if has_res == 1:
            dres = dy
        elif has_res == 2:   # adjoint of the nearest x2 upsample: sum of each 2x2 block
            dres = torch.empty((n, h // 2, w // 2, c), dtype=dy.dtype, device=dy.device)
            launch("fold2x2", ptr(dy), ptr(dres), n, h // 2, w // 2, c, stream())
        return (dx, dgamma.to(gamma.dtype), dbeta.to(gamma.dtype), None, None, None, None, None, dres, None, None, None)


def _sync_sum(vec, site):
    """a SyncBN message: RCCL all-reduce, or -- M355_SYNCBN_IPC=1 -- the one-launch exchange over peer-mapped memory (`site`: layer and
    direction, the exchange's channel)"""
    from . import parallel
    parallel.syncbn_all_reduce(vec, site)


def _count_syncbn():
    from . import parallel
    parallel.stats["syncbn_collectives"] += 1


def _match_res(res, x):
    """a half-resolution residual is the nearest x2 upsample of itself (torch fallback paths)"""
    return res if res.shape == x.shape else upsample2x(res)


def _fused_ok(x):
    c = x.shape[-1]
    return x.is_cuda and x.dtype == _act() and c % 8 == 0 and c <= 2048 and 256 % (c // 8) == 0


class TailPair:
    """Links the two consumers of a discriminator's last feature map -- the one-channel 5x5 logit conv and the projection term --
    so that ONE kernel produces the feature map's whole gradient (csrc/gan_elem.hip k_cproj_bwd_conv5) instead of the conv's dgrad,
    the projection's dfeat and autograd's addition of the two.  Protocol: the projection term is created FIRST in the forward, the conv
    second; autograd runs the younger node first, so the conv's backward finds the pair empty, leaves its dgrad request here (logit
    gradient, dgrad weights, descriptor) and returns no input gradient; the projection's backward then runs the fused kernel.  Should
    the engine ever run them the other way round, the projection marks `proj_done`, and the conv's backward sees it and computes its own
    dgrad: both orders are correct, only the first is fused."""
    __slots__ = ("request", "proj_done")

    def __init__(self):
        self.request, self.proj_done = None, False


def tail_pair_ok(feat, conv_desc_kw, mode, cout):
    """can the fused tail backward take this feature map / logit conv?  (5x5, one output channel, zero or circular W pad)"""
    n, h, w, c = feat.shape
    return (feat.is_cuda and _fused_ok(feat) and conv_desc_kw == (5, 5, 1, 2, 2) and cout == 1 and mode in (C.PAD_ZERO, C.PAD_CIRCULAR)
            and not os.environ.get("M355_NO_TAIL_FUSION") and bool(lib().m355_cproj_bwd_conv5_ok(h, w, c)))


class ClassProjection(torch.autograd.Function):
    """projection discriminator term (gan.py:104-116, 216-228): feat [N,H,W,C] bf16, emb [N,C] fp32 -> [N,H,W] fp32
    = sum_c feat * emb, on the bf16 feature map (csrc/gan_elem.hip k_cproj_*)"""

    @staticmethod
    def forward(ctx, feat, emb, in_slope=1.0, pair=None):
        feat, emb = feat.detach().contiguous(), emb.detach().float().contiguous()
        n, h, w, c = feat.shape
        out = torch.empty((n, h, w), dtype=torch.float32, device=feat.device)
        launch("cproj_fwd", ptr(feat), ptr(emb), ptr(out), n, h * w, c, stream())
        ctx.save_for_backward(feat, emb)
        ctx.in_slope, ctx.pair = float(in_slope), pair
        return out

    @staticmethod
    def backward(ctx, g):
        feat, emb = ctx.saved_tensors
        n, h, w, c = feat.shape
        dfeat = torch.empty_like(feat)
        demb = torch.empty((n, c), dtype=torch.float32, device=feat.device)
        g = g.contiguous().float()
        pair = ctx.pair
        if pair is not None and pair.request is not None:
            # the logit conv's backward ran first and left its dgrad here: the whole gradient of feat in one pass
            dy5, wd, kp, mode, slope = pair.request
            pair.request = None
            assert slope == ctx.in_slope and tuple(dy5.shape) == (n, h, w)
            nws = lib().m355_cproj_bwd_conv5_ws_floats(n, h, w, c)
            ws = torch.empty((nws,), dtype=torch.float32, device=feat.device) if nws else None
            launch("cproj_bwd_conv5", ptr(feat), ptr(emb), ptr(g), ptr(dy5), ptr(wd), int(kp), ptr(dfeat), ptr(demb), ptr(ws), n, h, w, c,
                   ctx.in_slope, int(mode), stream())
            return dfeat, demb, None, None
        if pair is not None:
            pair.proj_done = True   # (the conv's backward has not run yet: it will compute its own dgrad)
        nws = lib().m355_cproj_bwd_ws_floats(n, h * w, c)
        ws = torch.empty((nws,), dtype=torch.float32, device=feat.device) if nws else None
        launch("cproj_bwd", ptr(feat), ptr(emb), ptr(g), ptr(dfeat), ptr(demb), ptr(ws), n, h * w, c,
               ctx.in_slope, stream())
        return dfeat, demb, None, None


def class_projection_fused(feat, emb):
    return _fused_ok(feat) and emb.dtype == torch.float32


def class_projection(feat, emb, in_slope=1.0, pair=None):
    """sum_c feat[n,h,w,c] * emb[n,c] -> [N,H,W] fp32.  in_slope != 1 (only where class_projection_fused): feat is a fused
    conv + LeakyReLU(in_slope) output whose producer was called with premasked=True -- the returned grad_feat is multiplied by
    that activation's derivative (every consumer of feat must do the same)"""
    if class_projection_fused(feat, emb):
        return ClassProjection.apply(feat, emb, float(in_slope), pair)
    assert in_slope == 1.0 and pair is None   # (a TailPair needs the fused kernels: gan._tail checks tail_pair_ok first)
    return torch.einsum("nhwc,nc->nhw", feat.float(), emb)


# ------------------------------------------------------------------------------------------------ normalisation
def _affine_act(xhat, scale, shift, slope):
    """(torch path, CPU tests / odd channel counts) xhat fp32 [N,H,W,C]; scale/shift [N,C]"""
    y = xhat * scale[:, None, None, :].float() + shift[:, None, None, :].float()
    if slope != 1.0:
        y = F.leaky_relu(y, slope)
    return y.to(_act())


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
        from . import parallel
        return self.sync and parallel.collectives_on()

    def _counts(self):
        """does a training forward advance num_batches_tracked?  nn.BatchNorm2d does, the reference's SynchronizedBatchNorm2d never
        does; a plain batch norm switched to global-batch statistics for data parallelism (recon_train.ReconTrainer) still does"""
        return getattr(self, "_count_batches", not self.sync)

    def _update_running(self, mean, var, cnt):
        with torch.no_grad():
            unbiased = var * (cnt / (cnt - 1)) if float(cnt) > 1 else var
            self.running_mean.mul_(1 - self.momentum).add_(mean.detach(), alpha=self.momentum)
            self.running_var.mul_(1 - self.momentum).add_(unbiased.detach(), alpha=self.momentum)
            if self._counts():   # (the reference's SynchronizedBatchNorm2d.forward never counts, batchnorm.py:66-98)
                self.num_batches_tracked += 1

    def forward(self, x, gamma, beta, slope=1.0, res=None, out_slope=1.0, part=None):
        """LeakyReLU(BN(x) * (1 + gamma) + beta) [+ res]; gamma / beta [N,C]; res: residual branch, same shape as x.
        out_slope != 1: a second LeakyReLU on the result whose BACKWARD IS LEFT TO THE CONSUMER (a generator head,
        gan_ops.head_conv(in_slope=...)): the returned tensor must have no other consumer.
        part: partial (sum, sum of squares) of THIS x from the conv launch that produced it (conv2d(want_stats=True)); the
        statistics are then those of the conv's fp32 results, not of the bf16-rounded x (relative difference <= 2^-9 of the
        spread, tests/test_gan_elem_gpu.py::test_fused_conv_statistics_match_a_pass_over_the_tensor)"""
        if _fused_ok(x):
            sync = self._is_sync()
            if self.training and gamma.dtype == torch.float32:
                # single-launch coefficient algebra (csrc/gan_glue.hip); num_batches_tracked is bumped by the owner
                # (Generator.forward batches it over all layers) or here when used stand-alone
                y = CbnActFn.apply(x, gamma, beta, self.running_mean, self.running_var, self.momentum, self.eps, slope, res,
                                   sync, out_slope, part)
                if not getattr(self, "_defer_count", False) and self._counts():
                    self.num_batches_tracked += 1
                return y
        if out_slope != 1.0:
            return _OutAct.apply(BatchNorm2d.forward(self, x, gamma, beta, slope, res), out_slope)
        if res is not None:
            return BatchNorm2d.forward(self, x, gamma, beta, slope) + _match_res(res, x)  # (subclasses change the signature)
        if _fused_ok(x):
            sync = self._is_sync()
            scale = 1 + gamma
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
                return AffineActFn.apply(x, scale, beta, mean, rstd, slope, True, cnt, sync)
            rstd = torch.rsqrt(self.running_var + self.eps)
            return AffineActFn.apply(x, scale, beta, self.running_mean, rstd, slope, False, 1.0, False)
        scale = 1 + gamma
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
        return _affine_act((xf - mean) * torch.rsqrt(var + self.eps), scale, beta, slope)


class SynchronizedBatchNorm2d(BatchNorm2d):
    """sync_batchnorm.SynchronizedBatchNorm2d(ch, affine=False): statistics over the global batch.  As in the reference
    (sync_batchnorm/batchnorm.py:66-98: its forward bypasses nn.BatchNorm2d.forward), num_batches_tracked stays 0."""
    sync = True


class _OutAct(torch.autograd.Function):
    """(torch paths) LeakyReLU forward with an IDENTITY backward: the consumer of the result applies the activation's
    derivative itself (same contract as CbnActFn's out_slope)."""

    @staticmethod
    def forward(ctx, x, slope):
        return F.leaky_relu(x, slope)

    @staticmethod
    def backward(ctx, g):
        return g, None


class InstanceNorm2d(nn.Module):
    def __init__(self, ch, eps=1e-5):
        super().__init__()
        self.eps = eps

    def forward(self, x, gamma, beta, slope=1.0, res=None, out_slope=1.0, part=None):
        if out_slope != 1.0:
            return _OutAct.apply(self.forward(x, gamma, beta, slope, res), out_slope)
        if res is not None:
            return self.forward(x, gamma, beta, slope) + _match_res(res, x)
        xf = x.float()
        mean = xf.mean((1, 2), keepdim=True)
        var = xf.var((1, 2), unbiased=False, keepdim=True)
        return _affine_act((xf - mean) * torch.rsqrt(var + self.eps), 1 + gamma, beta, slope)


class NoNorm(nn.Module):
    def forward(self, x, gamma, beta, slope=1.0, res=None, out_slope=1.0, part=None):
        if out_slope != 1.0:
            return _OutAct.apply(self.forward(x, gamma, beta, slope, res), out_slope)
        if res is not None:
            return self.forward(x, gamma, beta, slope) + _match_res(res, x)
        if _fused_ok(x):
            c = x.shape[-1]
            zero = torch.zeros(c, dtype=torch.float32, device=x.device)
            return AffineActFn.apply(x, 1 + gamma, beta, zero, torch.ones_like(zero), slope, False, 1.0, False)
        return _affine_act(x.float(), 1 + gamma, beta, slope)
