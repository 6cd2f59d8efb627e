"""Low-level bindings of the bf16 MFMA conv2d entry points (include/m355.h, csrc/conv_mfma.hip).

Activations are NHWC bf16 tensors ([N,H,W,C] contiguous); weights are the fp32 [Cout,Cin,kh,kw] parameter,
turned into bf16 GEMM views by `weight_prep`.  No fallback: CPU tensors raise."""
import ctypes
import os

import torch

from . import _lib
from ._lib import ConvDesc, check, launch, lib, ptr, stream

PAD_ZERO, PAD_REPLICATE, PAD_CIRCULAR = 0, 1, 2

# Deterministic mode (M355_DETERMINISTIC=1 or set_deterministic(True)): the weight-gradient kernels accumulate their split-K
# partial tiles as 64-bit fixed-point integers instead of fp32 atomics (m355_conv2d_wgrad_det) -- with the other reductions of
# the GAN path deterministic by construction (ordered partial sums), two runs of a training cycle are then bit-identical, as the
# reference's CPU path is.  Cost: one 8-byte memset + one conversion pass per layer.
_DETERMINISTIC = os.environ.get("M355_DETERMINISTIC", "") == "1"


def set_deterministic(on):
    """-> the previous setting"""
    global _DETERMINISTIC
    prev, _DETERMINISTIC = _DETERMINISTIC, bool(on)
    return prev


def is_deterministic():
    return _DETERMINISTIC


def _ceil(a, b):
    return (a + b - 1) // b * b


def make_desc(N, H, W, Cin, Cout, kh, kw, stride=1, pad_h=0, pad_w=0, pad_w_mode=PAD_ZERO, upsample=0):
    return ConvDesc(N, H, W, Cin, Cout, kh, kw, stride, pad_h, pad_w, pad_w_mode, upsample)


_OUT_HW, _DY_CH = {}, {}   # pure geometry (no environment switches behind them): memoised, the hot loop calls them per launch
_PLAN = {}


def plan(d):
    """m355_conv2d_plan: everything the binding has to know about a layer before it launches (output extent, buffer sizes, which
    optional forms this build of the library runs the shape in), ONE library call per descriptor, memoised -- the selection rules
    live in the library, the hot loop reads a struct.  (Environment switches the library reads are process-wide settings: change
    them before the first launch, or call _reset_caches().)"""
    key = bytes(d)
    p = _PLAN.get(key)
    if p is None:
        p = _lib.ConvPlan()
        check(lib().m355_conv2d_plan(ctypes.byref(d), ctypes.byref(p)), "conv2d_plan")
        _PLAN[key] = p
    return p


def _reset_caches():
    """the memoised answers below belong to ONE build of the library (_lib.set_exact switches it)"""
    for c in (_OUT_HW, _DY_CH, _HALVES, _FWD_WS, _WS_BYTES, _EXEC_RATIO, _PLAN):
        c.clear()


_lib._RESET_HOOKS.append(_reset_caches)


def out_hw(d):
    key = bytes(d)
    r = _OUT_HW.get(key)
    if r is None:
        p = plan(d)
        r = _OUT_HW[key] = (p.Ho, p.Wo)
    return r


def dy_channels(cout):
    """channel stride the backward entry points expect of dy (8 for the 1..8-channel heads, else ceil32)"""
    r = _DY_CH.get(cout)
    if r is None:
        r = _DY_CH[cout] = lib().m355_conv2d_dy_channels(int(cout))
    return r


def flops(d, cin_real=None):
    """algorithmic FLOPs of one pass (fwd, dgrad or wgrad) over the layer: 2*M*N*K with the REAL channel counts
    (zero-padded input channels do not count)"""
    ho, wo = out_hw(d)
    return 2.0 * d.N * ho * wo * d.Cout * (cin_real or d.Cin) * d.kh * d.kw


_EXEC_RATIO = {}


def exec_ratio(d):
    """executed / algorithmic MACs of the layer's forward, dgrad and workspace / deterministic wgrad (4/9 in the sub-pixel form)"""
    key = bytes(d)
    r = _EXEC_RATIO.get(key)
    if r is None:
        r = _EXEC_RATIO[key] = float(plan(d).exec_ratio)
    return r


def tag(d):
    return (f"N{d.N} {d.H}x{d.W} {d.Cin}->{d.Cout} k{d.kh} s{d.stride} up{d.upsample}")


def _req(t, dtype, name):
    if not t.is_cuda:
        raise _lib.M355Error(f"{name} must be a CUDA(HIP) tensor; the conv path has no CPU implementation")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def weight_prep(d, w_oihw, want_dgrad=True, sigma=None):
    """bf16 GEMM views of the fp32 parameter; `sigma` (1-element device tensor) divides it (spectral norm)"""
    w = _req(w_oihw.detach(), torch.float32, "weight")
    L = lib()
    wf = torch.empty((plan(d).w_fwd_elems,), dtype=torch.bfloat16, device=w.device)   # (2-byte elements)
    wd = None
    if want_dgrad:
        wd = torch.empty((plan(d).w_dgrad_elems,), dtype=torch.bfloat16, device=w.device)
    launch("conv2d_weight_prep", ctypes.byref(d), ptr(w), int(w.shape[1]), ptr(sigma), ptr(wf), ptr(wd), stream())
    return wf, wd


# Tensors of 2 GiB and more.  The specialised kernels address bytes with 32 bits (buffer descriptors, LDS-DMA offsets), so their
# eligibility checks hand such layers to the generic 64-bit-addressed kernels -- correct, and 3-4x slower (D.conv2 forward at N =
# 256: 300 TF instead of 1000).  No conv kernel couples the samples of a batch -- batch-norm partial rows and weight gradients
# are SUMS over samples -- so the same layer is run on each half of the batch instead (recursively), every half on the fast path;
# outputs are slices of one tensor, the weight-gradient halves are added.
_HALVES = {}


def _halves(d, out_bytes=2):
    """None, or (descriptor of half the batch, N/2) when a tensor of this layer reaches 2 GiB (out_bytes = 4: the fp32 NCHW
    output form of conv_fwd)"""
    key = bytes(d) + bytes([out_bytes])
    r = _HALVES.get(key, 0)
    if r == 0:
        r = None
        if d.N >= 2 and d.N % 2 == 0:
            ho, wo = out_hw(d)
            # (activation bytes from the loaded build: 2, or 4 under the EXACT build -- one element size for every pass of the layer;
            # only the fp32 NCHW output form of conv_fwd is wider than that)
            ab = int(plan(d).act_bytes)
            big = max(d.N * d.H * d.W * d.Cin * ab, d.N * ho * wo * max(d.Cout, dy_channels(d.Cout)) * max(out_bytes, ab))
            if big >= (1 << 31):
                r = (make_desc(d.N // 2, d.H, d.W, d.Cin, d.Cout, d.kh, d.kw, d.stride, d.pad_h, d.pad_w, d.pad_w_mode, d.upsample), d.N // 2)
        _HALVES[key] = r
    return r


def maskbits_ok(d, role):
    """role 0: the forward of this layer can write bit-packed activation masks; role 1: its dgrad can read them"""
    hv = _halves(d)
    return maskbits_ok(hv[0], role) if hv else bool(plan(d).dgrad_bits_ok if role else plan(d).fwd_bits_ok)


def dgrad_mask_ok(d):
    """can conv_dgrad(mask_x=...) apply the producer's LeakyReLU backward in its epilogue on this layer?"""
    hv = _halves(d)
    return dgrad_mask_ok(hv[0]) if hv else bool(plan(d).dgrad_mask_ok)


def conv_fwd(d, x, w_fwd, bias=None, out_f32_nchw=False, slope=1.0, cin_real=None, emit_bits=False, _out=None):
    """emit_bits: also return the sign bits of the pre-activation ([N,Ho,Wo,Cout/64,2] int32, opaque layout) for
    the consumer's conv_dgrad(mask_bits=...); requires maskbits_ok(d, 0).  (_out: the output views of a half-batch launch)"""
    x = _req(x, _lib.act_dtype(), "x")
    assert tuple(x.shape) == (d.N, d.H, d.W, d.Cin), (tuple(x.shape), (d.N, d.H, d.W, d.Cin))
    ho, wo = out_hw(d)
    if _out is not None:
        y, bits = _out
    else:
        bits = torch.empty((d.N, ho, wo, d.Cout // 64, 2), dtype=torch.int32, device=x.device) if emit_bits else None
        if out_f32_nchw:
            y = torch.empty((d.N, d.Cout, ho, wo), dtype=torch.float32, device=x.device)
        else:
            y = torch.empty((d.N, ho, wo, d.Cout), dtype=_lib.act_dtype(), device=x.device)
    hv = _halves(d, 4 if out_f32_nchw else 2)
    if hv:
        dh, h = hv
        for i in (0, 1):
            conv_fwd(dh, x[i * h:(i + 1) * h], w_fwd, bias, out_f32_nchw, slope, cin_real, emit_bits,
                     _out=(y[i * h:(i + 1) * h], None if bits is None else bits[i * h:(i + 1) * h]))
        return (y, bits) if emit_bits else y
    b = None if bias is None else _req(bias.detach(), torch.float32, "bias")
    if emit_bits:
        assert not out_f32_nchw
        launch("conv2d_fwd_bits", ctypes.byref(d), ptr(x), ptr(w_fwd), ptr(b), ptr(y), float(slope), ptr(bits), stream(),
               work=lambda: flops(d, cin_real), tag=lambda: tag(d))
        return y, bits
    launch("conv2d_fwd", ctypes.byref(d), ptr(x), ptr(w_fwd), ptr(b), ptr(y), int(out_f32_nchw), float(slope), stream(),
           work=lambda: flops(d, cin_real), tag=lambda: tag(d), exec_ratio=lambda: 1.0 if out_f32_nchw else exec_ratio(d))
    return y


def conv_stats_rows(d):
    """rows of batch-norm partial sums conv_fwd_stats writes for this shape (0: no fused statistics)"""
    hv = _halves(d)
    if hv:
        return 2 * conv_stats_rows(hv[0])   # (the halves' partial rows, one after the other)
    nws, rows = _fwd_ws(d)
    if nws:
        return rows   # (split-K layers: the finishing pass emits them)
    # (asked live, not from the memoised plan: the answer sizes the `part` buffer the kernel writes, and it follows the
    # M355_HALO_WGS / M355_STATS_UPS_WGS test switches)
    return int(lib().m355_conv2d_fwd_stats_rows(ctypes.byref(d)))


def conv_fwd_stats(d, x, w_fwd, bias=None, cin_real=None, rows=None, _out=None):
    """-> y, part: the forward (no activation) and part [rows,2,Cout] fp32 = per-workgroup (sum, sum of squares) of the fp32
    results over the workgroup's pixels -- what bn_finalize reduces; requires conv_stats_rows(d) > 0"""
    x = _req(x, _lib.act_dtype(), "x")
    assert tuple(x.shape) == (d.N, d.H, d.W, d.Cin), (tuple(x.shape), (d.N, d.H, d.W, d.Cin))
    ho, wo = out_hw(d)
    rows = conv_stats_rows(d) if rows is None else rows
    hv = _halves(d)
    if hv:
        dh, h = hv
        y, part = _out if _out is not None else (torch.empty((d.N, ho, wo, d.Cout), dtype=_lib.act_dtype(), device=x.device),
                                                 torch.empty((rows, 2, d.Cout), dtype=torch.float32, device=x.device))
        r = rows // 2
        for i in (0, 1):
            conv_fwd_stats(dh, x[i * h:(i + 1) * h], w_fwd, bias, cin_real, r, _out=(y[i * h:(i + 1) * h], part[i * r:(i + 1) * r]))
        return y, part
    b = None if bias is None else _req(bias.detach(), torch.float32, "bias")
    if _fwd_ws(d)[0]:
        assert _out is None
        return _fwd_splitk(d, x, w_fwd, b, 1.0, True)
    y, part = _out if _out is not None else (torch.empty((d.N, ho, wo, d.Cout), dtype=_lib.act_dtype(), device=x.device),
                                             torch.empty((rows, 2, d.Cout), dtype=torch.float32, device=x.device))
    launch("conv2d_fwd_stats", ctypes.byref(d), ptr(x), ptr(w_fwd), ptr(b), ptr(y), ptr(part), stream(),
           work=lambda: flops(d, cin_real), tag=lambda: tag(d), exec_ratio=lambda: exec_ratio(d))
    return y, part


def conv_dgrad(d, dy, w_dgrad, cin_real=None, mask_x=None, mask_slope=1.0, mask_bits=None, _out=None, lead=0):
    """mask_x: this conv's input x when it is the output of a fused conv+LeakyReLU(mask_slope): the returned gradient
    is then already multiplied by that activation's derivative; mask_bits: the same from the producer's bit masks
    (conv_fwd(emit_bits=True)), 1/16 of the bytes.  lead > 0: the input channels beyond the first `lead` are constants of the model
    (positional planes): only dx[..., :lead] is specified (m355_conv2d_dgrad_lead)"""
    dy = _req(dy, _lib.act_dtype(), "dy")
    ho, wo = out_hw(d)
    assert tuple(dy.shape) == (d.N, ho, wo, dy_channels(d.Cout)), tuple(dy.shape)
    dx = _out if _out is not None else torch.empty((d.N, d.H, d.W, d.Cin), dtype=_lib.act_dtype(), device=dy.device)
    hv = _halves(d)
    if hv:
        dh, h = hv
        for i in (0, 1):
            sl = slice(i * h, (i + 1) * h)
            conv_dgrad(dh, dy[sl], w_dgrad, cin_real, None if mask_x is None else mask_x[sl], mask_slope,
                       None if mask_bits is None else mask_bits[sl], _out=dx[sl], lead=lead)
        return dx
    nws = plan(d).dgrad_ws_bytes
    ws = torch.empty((nws,), dtype=torch.uint8, device=dy.device) if nws else None
    if mask_bits is not None:
        assert tuple(mask_bits.shape) == (d.N, d.H, d.W, d.Cin // 64, 2), (tuple(mask_bits.shape), d.Cin)
        launch("conv2d_dgrad_bits", ctypes.byref(d), ptr(dy), ptr(w_dgrad), ptr(dx), ptr(ws), ptr(mask_bits), float(mask_slope),
               stream(), work=lambda: flops(d, cin_real), tag=lambda: tag(d))
        return dx
    if lead and mask_x is None:
        # (the work that is asked for: `lead` of the layer's input channels)
        launch("conv2d_dgrad_lead", ctypes.byref(d), ptr(dy), ptr(w_dgrad), ptr(dx), ptr(ws), int(lead), stream(),
               work=lambda: flops(d, min(int(lead), cin_real or d.Cin)), tag=lambda: tag(d) + f" lead{int(lead)}")
        return dx
    launch("conv2d_dgrad", ctypes.byref(d), ptr(dy), ptr(w_dgrad), ptr(dx), ptr(ws), ptr(mask_x), float(mask_slope), stream(),
           work=lambda: flops(d, cin_real), tag=lambda: tag(d), exec_ratio=lambda: exec_ratio(d))
    return dx


_FWD_WS = {}


def _fwd_ws(d):
    """(workspace bytes, batch-norm partial rows) of the split-K forward of a small layer, (0, 0) if it has none"""
    key = bytes(d)
    r = _FWD_WS.get(key)
    if r is None:
        L = lib()
        r = _FWD_WS[key] = (int(plan(d).fwd_ws_bytes), int(plan(d).fwd_ws_stats_rows))
    return r


def _fwd_splitk(d, x, w_fwd, b, slope, want_part):
    nws, rows = _fwd_ws(d)
    ho, wo = out_hw(d)
    ws = torch.empty((nws,), dtype=torch.uint8, device=x.device)
    y = torch.empty((d.N, ho, wo, d.Cout), dtype=_lib.act_dtype(), device=x.device)
    part = torch.empty((rows, 2, d.Cout), dtype=torch.float32, device=x.device) if want_part else None
    launch("conv2d_fwd_ws", ctypes.byref(d), ptr(x), ptr(w_fwd), ptr(b), ptr(y), float(slope), ptr(ws), ptr(part), stream(),
           work=lambda: flops(d), tag=lambda: tag(d))
    return y, part


_WS_BYTES = {}


def _wgrad_ws_bytes(d):
    key = bytes(d)
    r = _WS_BYTES.get(key)
    if r is None:
        r = _WS_BYTES[key] = int(plan(d).wgrad_ws_bytes)
    return r


def wgrad_fuses_dbias(d):
    hv = _halves(d)
    return wgrad_fuses_dbias(hv[0]) if hv else bool(plan(d).wgrad_fuses_dbias)


def _after_fill(st, device):
    """the zero fill of a per-pass block ran on the stream that asked first; a kernel on ANOTHER stream (gan_ops.Fork) must not
    touch its slice before that fill has finished: the fill carries an event, other streams wait for it once per pass"""
    if torch.device(device).type != "cuda":
        return
    cur = torch.cuda.current_stream(device)
    if st["stream"] is not None and cur != st["stream"] and cur not in st["waited"]:
        cur.wait_event(st["event"])
        st["waited"].add(cur)


def _mark_fill(st, device):
    if torch.device(device).type != "cuda":
        return
    st["stream"] = torch.cuda.current_stream(device)
    st["event"] = torch.cuda.Event()
    st["event"].record(st["stream"])
    st["waited"] = set()


class WgradArena:
    """Zero-filled fp32 scratch for the split-K weight-gradient kernels of ONE backward pass: every wgrad kernel accumulates
    with atomics into a zeroed buffer, which used to cost one or two memsets per layer (65 per GAN cycle, 5 us each).  The
    arena is zeroed once, when the first wgrad of a backward pass (identified by autograd's graph-task id) asks for a slice;
    its size is learnt from the previous pass (a pass that outgrows it falls back to per-layer zero fills and the arena grows
    for the next one).  Slices are scratch: they are consumed by wgrad_finish inside the same backward call."""
    _state = {}   # device -> {buf, used, needed, task, stream, event, waited}

    @classmethod
    def take(cls, numel, device):
        """-> a zeroed fp32 slice of `numel` elements, or None (caller zero-fills its own buffer)"""
        task = torch._C._current_graph_task_id() if hasattr(torch._C, "_current_graph_task_id") else -1
        if task < 0:
            return None
        st = cls._state.setdefault(device, dict(buf=None, used=0, needed=0, task=None, stream=None, event=None, waited=set()))
        if st["task"] != task:   # first wgrad of a new backward pass
            if st["needed"] > (0 if st["buf"] is None else st["buf"].numel()):
                st["buf"] = torch.empty(st["needed"], dtype=torch.float32, device=device)
            st["used"], st["needed"], st["task"] = 0, 0, task
            if st["buf"] is not None:
                # (every kernel of the previous pass that used a slice has been joined: flush_wgrad_finish waits for the side streams)
                st["buf"].zero_()
                _mark_fill(st, device)
        numel = (numel + 63) // 64 * 64
        off = st["used"]
        st["used"] += numel
        st["needed"] = max(st["needed"], st["used"])
        if st["buf"] is None or off + numel > st["buf"].numel():
            return None
        _after_fill(st, device)
        return st["buf"][off:off + numel]


class DbiasBlock:
    """Bias gradients the wgrad kernels accumulate with atomics need a zeroed buffer each: here they are slices of ONE block per
    backward pass, zero-filled in one launch.  Unlike the WgradArena the block is a FRESH tensor every pass -- its slices are
    returned to autograd as the parameters' .grad and must outlive the pass.  Its size is the LARGEST need any pass has shown
    (the G pass of the 1 G : 2 D cycle takes nothing, the D passes do: sizing from the previous pass alone served one pass in
    three), and it is allocated by the first take() of a pass, so a pass that takes nothing costs nothing."""
    _state = {}   # device -> {buf, used, needed, task, stream, event, waited}

    @classmethod
    def take(cls, numel, device):
        task = torch._C._current_graph_task_id() if hasattr(torch._C, "_current_graph_task_id") else -1
        if task < 0:
            return None
        st = cls._state.setdefault(device, dict(buf=None, used=0, needed=0, task=None, stream=None, event=None, waited=set()))
        if st["task"] != task:   # first take of a new backward pass
            st["buf"] = torch.zeros(st["needed"], dtype=torch.float32, device=device) if st["needed"] else None
            st["used"], st["task"] = 0, task
            if st["buf"] is not None:
                _mark_fill(st, device)
        numel_p = (numel + 63) // 64 * 64
        off = st["used"]
        st["used"] += numel_p
        st["needed"] = max(st["needed"], st["used"])
        if st["buf"] is None or off + numel_p > st["buf"].numel():
            return None
        _after_fill(st, device)
        if st["stream"] is not None and torch.cuda.current_stream(device) != st["stream"]:
            st["buf"].record_stream(torch.cuda.current_stream(device))   # (slices become .grad tensors, freed on the main stream)
        return st["buf"][off:off + numel]


def conv_wgrad(d, x, dy, cin_real=None, raw=False, dbias=None, arena=False, dbias_zeroed=False):
    """-> dw fp32 in the parameter's layout [Cout,Cin,kh,kw] (a permuted view), or with raw=True the kernel's own
    [Cout,kh,kw,Cin] buffer.  arena=True (with raw=True, from inside a backward pass): the buffer is a slice of the
    per-pass WgradArena, valid until the next backward pass"""
    x, dy = _req(x, _lib.act_dtype(), "x"), _req(dy, _lib.act_dtype(), "dy")
    n = d.Cout * d.kh * d.kw * d.Cin
    hv = _halves(d)
    if hv:
        # a sum over samples: the first half as usual (arena slice, zeroing rules and all), the second half into fresh buffers, added
        # in this fixed order (deterministic when the halves are)
        dh, h = hv
        dw = conv_wgrad(dh, x[:h], dy[:h], cin_real, True, dbias, arena, dbias_zeroed)
        db2 = None if dbias is None else torch.empty_like(dbias)
        dw.add_(conv_wgrad(dh, x[h:], dy[h:], cin_real, True, db2))
        if dbias is not None:
            dbias.add_(db2)
        return dw if raw else dw.permute(0, 3, 1, 2)
    nws = _wgrad_ws_bytes(d)
    if nws and (plan(d).wgrad_ws_ordered or not (_DETERMINISTIC and d.upsample)):
        # thin layers: per-workgroup partial tiles + an ordered sum instead of atomics (deterministic in every mode; dw / dbias
        # are overwritten, so neither the arena's zero fill nor a zeroed bias block is needed).  Also the sub-pixel form of the
        # upsample + 3x3 layers: its 16-entry effective gradient lives in ws (zeroed by the library, fp32 atomics -- in
        # deterministic mode those layers take the fixed-point entry point below instead)
        ws = torch.empty((nws,), dtype=torch.uint8, device=x.device)
        dw = torch.empty((d.Cout, d.kh, d.kw, d.Cin), dtype=torch.float32, device=x.device)
        launch("conv2d_wgrad_ws", ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(dbias), ptr(ws), stream(),
               work=lambda: flops(d, cin_real), tag=lambda: tag(d), exec_ratio=lambda: exec_ratio(d))
        return dw if raw else dw.permute(0, 3, 1, 2)
    if _DETERMINISTIC:
        ws = torch.empty((plan(d).wgrad_det_ws_bytes,), dtype=torch.uint8, device=x.device)
        dw = torch.empty((d.Cout, d.kh, d.kw, d.Cin), dtype=torch.float32, device=x.device)
        # (issued-MAC ratio: the weight gradient takes the sub-pixel form only where the layer has the 16-entry workspace form -- the
        # small stages' forward / dgrad do, their weight gradient is the 9-tap one)
        launch("conv2d_wgrad_det", ctypes.byref(d), ptr(x), ptr(dy), ptr(ws), ptr(dw), ptr(dbias), stream(),
               work=lambda: flops(d, cin_real), tag=lambda: tag(d), exec_ratio=lambda: exec_ratio(d) if _wgrad_ws_bytes(d) else 1.0)
        return dw if raw else dw.permute(0, 3, 1, 2)
    sl = WgradArena.take(n, x.device) if (arena and raw) else None
    if sl is not None:
        dw = sl[:n].view(d.Cout, d.kh, d.kw, d.Cin)
        if dbias is not None and not dbias_zeroed:   # returned to autograd as the bias gradient: never an arena slice
            dbias.zero_()
        launch("conv2d_wgrad_acc", ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(dbias), stream(), work=lambda: flops(d, cin_real), tag=lambda: tag(d))
        return dw
    dw = torch.empty((d.Cout, d.kh, d.kw, d.Cin), dtype=torch.float32, device=x.device)
    launch("conv2d_wgrad", ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(dbias), stream(), work=lambda: flops(d, cin_real), tag=lambda: tag(d))
    return dw if raw else dw.permute(0, 3, 1, 2)


class _DeferredFinish:
    """weight-gradient epilogues queued inside `deferred_wgrad_finish()`: (entry fields, tensors kept alive, parameter)"""
    active = False
    items = []
    part = {}      # device -> fp32 scratch (one float per output channel of the queued layers)


class deferred_wgrad_finish:
    """Context manager around a backward pass whose parameter gradients nobody reads before the pass is over (GanTrainer:
    zero_grad(set_to_none=True) -> backward -> reducer -> optimiser): wgrad_finish(..., param=p) of the pass only queues its
    work and returns None to autograd; at exit ALL of them run in two launches (m355_sn_wgrad_finish_batched) instead of two
    per layer, and every result is accumulated into its parameter's .grad here (p.grad = dw, or += if one exists) -- not
    through autograd's AccumulateGrad, which would have to read the tensor before it is written (and clones any gradient
    somebody else still references).  Gradient hooks of those parameters therefore do not fire; the trainer reduces and steps
    after the pass.  The raw gradients are slices of the per-pass WgradArena, valid until the next backward pass."""

    def __enter__(self):
        import os
        self.prev = _DeferredFinish.active
        _DeferredFinish.active = not os.environ.get("M355_NO_DEFER_FINISH")   # (A/B switch: the per-layer launches)
        return self

    def __exit__(self, *exc):
        _DeferredFinish.active = self.prev
        if not self.prev:
            flush_wgrad_finish(discard=exc[0] is not None)
        return False


def flush_wgrad_finish(discard=False):
    items, _DeferredFinish.items = _DeferredFinish.items, []
    if discard or not items:
        return
    # raw gradients of branches that ran on the second stream (gan_ops.Fork): join it before the batched epilogue reads them
    from . import gan_ops
    for sd in gan_ops.side_streams_to_join():
        torch.cuda.current_stream(sd.device).wait_stream(sd)
    for i0 in range(0, len(items), _lib.SNFIN_MAX):
        chunk = items[i0:i0 + _lib.SNFIN_MAX]
        arr = (_lib.SnFinEntry * len(chunk))()
        dev = chunk[0][1][0].device
        rows = sum(f[6] for f, _k, _p in chunk)    # one partial of <g, w_orig> per output channel
        part = _DeferredFinish.part.get(dev)
        if part is None or part.numel() < rows:
            part = _DeferredFinish.part[dev] = torch.empty(max(rows, 8192), dtype=torch.float32, device=dev)
        row = 0
        for k, (f, _keep, _param) in enumerate(chunk):
            e = arr[k]
            e.g_khwc, e.w_orig, e.u, e.v, e.sigma, e.dw = f[0], f[1], f[2], f[3], f[4], f[5]
            e.part = part.data_ptr() + 4 * row if f[4] else None
            e.Cout, e.Cin, e.CinP, e.kh, e.kw = f[6:11]
            row += f[6]
        launch("sn_wgrad_finish_batched", arr, len(chunk), stream())
    with torch.no_grad():
        for f, keep, param in items:
            dw = keep[-1]
            if param.grad is None:
                param.grad = dw
            else:
                param.grad += dw


def wgrad_finish(d, g_khwc, cin_real, w_orig=None, u=None, v=None, sigma=None, param=None):
    """[Cout][kh][kw][CinP] wgrad output -> the parameter's gradient [Cout][cin_real][kh][kw]; with spectral-norm
    state, the gradient with respect to weight_orig (through sigma).  param (a leaf Parameter) inside
    `deferred_wgrad_finish()`: the work is queued, the result is accumulated into param.grad when the context exits, and
    None is returned (autograd gets no gradient for it from this call)."""
    dw = torch.empty((d.Cout, cin_real, d.kh, d.kw), dtype=torch.float32, device=g_khwc.device)
    # (the batched kernel stages one output channel's [kh*kw][CinP + 1] block in LDS, gan_glue.hip kSnFinLds: wider layers --
    # e.g. 1024 channels x 4x4 -- take the per-layer launch below instead of failing the whole flush after the pass)
    fits = (d.Cin + 1) * d.kh * d.kw <= _lib.SNFIN_LDS_FLOATS
    if fits and _DeferredFinish.active and param is not None and param.is_leaf and param.requires_grad:
        from . import gan_ops
        gan_ops.note_side_work()   # (the raw gradient may be in flight on the second stream: flush_wgrad_finish joins it)
        _DeferredFinish.items.append(((ptr(g_khwc), ptr(w_orig), ptr(u), ptr(v), ptr(sigma), ptr(dw), d.Cout, cin_real, d.Cin,
                                       d.kh, d.kw), (g_khwc, w_orig, u, v, sigma, dw), param))
        return None
    part = torch.empty((256,), dtype=torch.float32, device=g_khwc.device) if sigma is not None else None
    launch("sn_wgrad_finish", ptr(g_khwc), ptr(w_orig), ptr(u), ptr(v), ptr(sigma), ptr(part), ptr(dw), d.Cout, cin_real,
           d.Cin, d.kh, d.kw, stream())
    return dw
