"""ctypes binding of libm355.so (include/m355.h).  There is NO fallback: if the HIP extension is
missing or a call fails, the op raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# M355_LIB: A/B builds of the same ABI (build.py); M355_EXACT=1: the fp32 EXACT build (lib/libm355_exact.so, csrc/conv_exact.hip)
EXACT_LIB = "libm355_exact.so"
LIB_PATH = os.path.join(_HERE, "lib", os.environ.get("M355_LIB", EXACT_LIB if os.environ.get("M355_EXACT") == "1" else "libm355.so"))

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


class M355Error(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); mirrors include/m355.h one to one (tests/test_abi.py checks both ways)
_P = c_void_p
SIGNATURES = {
    "m355_last_error": (ctypes.c_char_p, []),
    "m355_last_kernel": (ctypes.c_char_p, []),
    "m355_abi_version": (c_int, []),
    "m355_act_bytes": (c_int, []),
    "m355_proj_transform_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "m355_quat_rotate_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "m355_quat_rotate_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "m355_proj_ntiles": (c_int, [c_int]),
    "m355_proj_bin_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P]),
    "m355_proj_transform_bwd": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, c_int, _P, c_int, c_int, c_float,
                                        c_float, _P]),
    "m355_smooth_taps": (c_int, [_P, c_int, c_int, _P, _P]),
    "m355_proj_render_fwd": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "m355_proj_render_bwd": (c_int, [_P, _P, _P, _P, c_int, _P, c_float, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "m355_trilinear_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "m355_trilinear_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "m355_smooth_axis": (c_int, [_P, _P, _P, c_int, c_int, _P, c_int, c_int, c_int, _P]),
    "m355_scale_clamp_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_size_t, _P]),
    "m355_termination_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "m355_termination_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "m355_sil_loss_ws_bytes": (c_size_t, [c_int, c_int]),
    "m355_sil_loss_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P]),
    "m355_chamfer_nn_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "m355_chamfer_nn_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "m355_chamfer_nn_fwd_ws": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "m355_conv2d_out_hw": (c_int, [_P, _P, _P]),
    "m355_conv2d_plan": (c_int, [_P, _P]),
    "m355_conv2d_dy_channels": (c_int, [c_int]),
    "m355_conv2d_exec_ratio": (ctypes.c_double, [_P]),
    "m355_fold2x2": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "m355_conv2d_weight_elems": (c_size_t, [_P, c_int]),
    "m355_conv2d_weight_prep": (c_int, [_P, _P, c_int, _P, _P, _P, _P]),
    "m355_conv2d_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_float, _P]),
    "m355_conv2d_dgrad_ws_bytes": (c_size_t, [_P]),
    "m355_conv2d_dgrad": (c_int, [_P, _P, _P, _P, _P, _P, c_float, _P]),
    "m355_conv2d_dgrad_lead": (c_int, [_P, _P, _P, _P, _P, c_int, _P]),
    "m355_mesh_vertices_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "m355_mesh_vertices_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "m355_mesh_normals_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "m355_mesh_normals_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "m355_mesh_flat_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "m355_mesh_flat_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "m355_dibr_ws_bytes": (c_size_t, [c_int, c_int]),
    "m355_dibr_rasterize_fwd": (c_int, [c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, c_float, _P, _P, _P, _P,
                                        _P, _P]),
    "m355_dibr_rasterize_bwd": (c_int, [c_int, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P,
                                        _P, _P, _P]),
    "m355_dibr_rasterize_bwd_det_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "m355_dibr_rasterize_bwd_det": (c_int, [c_int, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P,
                                            _P, _P, _P, _P]),
    "m355_dibr_shade_bwd_det_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "m355_dibr_shade_bwd_det": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "m355_dibr_shade_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "m355_dibr_shade_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "m355_weight_prep_entry_bytes": (c_size_t, []),
    "m355_weight_prep_fill_entry": (ctypes.c_longlong, [_P, _P, c_int, _P, _P, _P, _P]),
    "m355_weight_prep_batched": (c_int, [_P, c_int, ctypes.c_longlong, _P]),
    "m355_weight_prep_entry_tiles": (c_int, [_P]),
    "m355_weight_prep_batched_tiled": (c_int, [_P, c_int, ctypes.c_longlong, c_int, c_int, _P]),
    "m355_cproj_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "m355_cproj_bwd_ws_floats": (c_size_t, [c_int, c_int, c_int]),
    "m355_cproj_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "m355_cproj_bwd_conv5_ok": (c_int, [c_int, c_int, c_int]),
    "m355_cproj_bwd_conv5_ws_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "m355_cproj_bwd_conv5": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "m355_conv2d_maskbits_ok": (c_int, [_P, c_int]),
    "m355_conv2d_dgrad_mask_ok": (c_int, [_P]),
    "m355_conv2d_fwd_bits": (c_int, [_P, _P, _P, _P, _P, c_float, _P, _P]),
    "m355_conv2d_dgrad_bits": (c_int, [_P, _P, _P, _P, _P, _P, c_float, _P]),
    "m355_conv2d_wgrad_fuses_dbias": (c_int, [_P]),
    "m355_conv2d_fwd_stats_rows": (c_int, [_P]),
    "m355_conv2d_fwd_stats": (c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "m355_conv2d_fwd_ws_bytes": (c_size_t, [_P]),
    "m355_conv2d_fwd_ws_stats_rows": (c_int, [_P]),
    "m355_conv2d_fwd_ws": (c_int, [_P, _P, _P, _P, _P, c_float, _P, _P, _P]),
    "m355_conv2d_wgrad": (c_int, [_P, _P, _P, _P, _P, _P]),
    "m355_conv2d_wgrad_acc": (c_int, [_P, _P, _P, _P, _P, _P]),
    "m355_conv2d_wgrad_ws_bytes": (c_size_t, [_P]),
    "m355_conv2d_wgrad_ws": (c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "m355_conv2d_wgrad_det_ws_bytes": (c_size_t, [_P]),
    "m355_conv2d_wgrad_det": (c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "m355_chan_reduce_ws_bytes": (c_size_t, [c_size_t, c_int, c_int, c_int]),
    "m355_bn_stats": (c_int, [_P, _P, _P, c_size_t, c_int, _P]),
    "m355_chan_sum": (c_int, [_P, _P, _P, c_size_t, c_int, _P]),
    "m355_affine_act_fwd": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_float, c_float, _P]),
    "m355_mask_cat_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "m355_mask_cat_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "m355_pool_pack_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "m355_pool_pack_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, _P]),
    "m355_pool_unpack_bwd": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "m355_pool_pack_parts_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P]),
    "m355_pool_unpack_parts_bwd": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, _P, _P, c_int, c_int, c_int, _P]),
    "m355_unpack_range": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "m355_head_tail_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "m355_head_tail_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "m355_hinge_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "m355_hinge_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "m355_affine_act_bwd_reduce": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "m355_affine_act_bwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "m355_lrelu_bwd": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_int, c_float, _P]),
    "m355_pack_nhwc8": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "m355_unpack_nhwc8": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "m355_chan_reduce_nblk": (c_int, [c_size_t, c_int]),
    "m355_bn_stats_partial": (c_int, [_P, _P, c_size_t, c_int, _P]),
    "m355_bn_sync_pack": (c_int, [_P, c_int, c_int, c_float, _P, _P]),
    "m355_affine_act_bwd_partial": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "m355_sn_scratch_words": (c_size_t, [c_int, c_int, c_int]),
    "m355_sn_power_iter": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_float, _P]),
    "m355_sn_wgrad_finish": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "m355_sn_wgrad_finish_batched": (c_int, [_P, c_int, _P]),
    "m355_bn_finalize": (c_int, [_P, c_int, c_float, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P,
                                 _P, _P]),
    "m355_bn_bwd_finalize": (c_int, [_P, c_int, c_float, _P, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P,
                                     _P]),
    "m355_bn_bwd_coeffs": (c_int, [_P, c_float, _P, _P, _P, c_int, _P, _P, _P]),
    "m355_ipc_region_bytes": (c_size_t, []),
    "m355_ipc_max_floats": (c_int, []),
    "m355_ipc_alloc": (c_int, [_P, _P]),
    "m355_ipc_open": (c_int, [_P, _P]),
    "m355_ipc_close": (c_int, [_P]),
    "m355_ipc_free": (c_int, [_P]),
    "m355_ipc_channels": (c_int, []),
    "m355_ipc_allreduce": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P]),
}


class ConvDesc(ctypes.Structure):
    """m355_conv_desc (include/m355.h)"""
    _fields_ = [(n, c_int) for n in ("N", "H", "W", "Cin", "Cout", "kh", "kw", "stride", "pad_h", "pad_w",
                                      "pad_w_mode", "upsample")]


class ConvPlan(ctypes.Structure):
    """m355_conv_plan (include/m355.h): every pre-launch answer about a layer, one call per descriptor"""
    _fields_ = [("Ho", c_int), ("Wo", c_int), ("dy_channels", c_int), ("act_bytes", c_int),
                ("w_fwd_elems", c_size_t), ("w_dgrad_elems", c_size_t),
                ("fwd_bits_ok", c_int), ("dgrad_bits_ok", c_int), ("dgrad_mask_ok", c_int), ("fwd_stats_rows", c_int),
                ("fwd_ws_stats_rows", c_int), ("wgrad_fuses_dbias", c_int),
                ("fwd_ws_bytes", c_size_t), ("dgrad_ws_bytes", c_size_t), ("wgrad_ws_bytes", c_size_t),
                ("wgrad_det_ws_bytes", c_size_t), ("exec_ratio", ctypes.c_double), ("w_dgrad_row_elems", c_int), ("wgrad_ws_ordered", c_int)]


class SnFinEntry(ctypes.Structure):
    """m355_snfin_entry (include/m355.h)"""
    _fields_ = [(n, c_void_p) for n in ("g_khwc", "w_orig", "u", "v", "sigma", "part", "dw")] + \
               [(n, c_int) for n in ("Cout", "Cin", "CinP", "kh", "kw", "pad_")]


SNFIN_MAX = 24
HEAD_TAIL_WS_FLOATS = 8192   # M355_HEAD_TAIL_WS_FLOATS
SNFIN_LDS_FLOATS = 12832   # kSnFinLds (csrc/gan_glue.hip): floats of one output channel the batched finish stages in LDS


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise M355Error(
                f"HIP extension {LIB_PATH} is missing -- build it with `python 2dimageto3dmodel_amd/build.py` "
                "(there is no CPU/PyTorch fallback for the hot path)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_ACT_DTYPE = None
_RESET_HOOKS = []   # caches of library answers (conv.py) cleared when the loaded library changes


def act_dtype():
    """torch dtype of the GAN path's activation tensors in the loaded build: bfloat16 (product) or float32 (EXACT build)"""
    global _ACT_DTYPE
    if _ACT_DTYPE is None:
        import torch
        _ACT_DTYPE = torch.float32 if lib().m355_act_bytes() == 4 else torch.bfloat16
    return _ACT_DTYPE


def is_exact():
    return lib().m355_act_bytes() == 4


def set_exact(on):
    """Switch this process between the product library and the EXACT build (fp32 activations, fp32 convs with fp64 accumulation:
    SURVEY.md 8c's exact mode) -> the previous setting.  Tensors and modules created under one setting must not be used under the
    other: networks keep no activation state between forwards, but build them (and their SpectralNormGroup views) AFTER switching.
    The returned token is the PATH of the library that was current (loaded, or the one M355_LIB / M355_EXACT select when nothing is
    loaded yet); passing it back restores exactly that library, an M355_LIB A/B build included.  set_exact(False) = the product
    library this process was configured with (M355_LIB, default libm355.so), never a hard-coded name."""
    global _lib, LIB_PATH, _ACT_DTYPE
    prev = LIB_PATH
    if isinstance(on, str):
        want = on
    elif on:
        want = os.path.join(_HERE, "lib", EXACT_LIB)
    else:
        want = os.path.join(_HERE, "lib", os.environ.get("M355_LIB", "libm355.so"))
    if want != LIB_PATH or _lib is None:
        LIB_PATH, _lib, _ACT_DTYPE = want, None, None
        for h in _RESET_HOOKS:
            h()
        lib()
    return prev


def check(rc, what):
    if rc != 0:
        msg = lib().m355_last_error().decode("utf-8", "replace")
        raise M355Error(f"{what} failed (status {rc}): {msg}")


# ---- per-launch HIP-event timers (bench.py): events are recorded on torch's current stream, which is the
#      stream every libm355 kernel is launched on
_TIMERS_ON = False
_TIMER_EVENTS = []  # (name, start_event, end_event, algorithmic work of the launch)


def enable_kernel_timers(on):
    global _TIMERS_ON
    _TIMERS_ON = bool(on)
    if on:
        _TIMER_EVENTS.clear()


def collect_kernel_timers():
    """-> {entry point: (launches, total_ms, total algorithmic work, total EXECUTED work)}; call after torch.cuda.synchronize().
    (executed < algorithmic only where an algebraically cheaper form of the operator runs: the sub-pixel upsample convs)"""
    out = {}
    for name, e0, e1, work, work_x in _TIMER_EVENTS:
        c, t, w, wx = out.get(name, (0, 0.0, 0.0, 0.0))
        out[name] = (c + 1, t + e0.elapsed_time(e1), w + work, wx + work_x)
    _TIMER_EVENTS.clear()
    return out


TIMER_TAGS = bool(os.environ.get("M355_TIMER_TAGS"))  # per-shape timer keys (scripts/layer_times.py)


def launch(name, *args, work=0.0, tag=None, exec_ratio=1.0):
    """call m355_<name>(*args), raise on a non-zero status; optionally bracket it with HIP events"""
    fn = getattr(lib(), "m355_" + name)
    if _TIMERS_ON:
        if callable(work):   # (callers pass thunks: a launch costs no ctypes round trip / string formatting for a timer that is off)
            work = work()
        if callable(tag):
            tag = tag()
        if TIMER_TAGS and tag:
            name_t = name + " " + tag
        else:
            name_t = name
        import torch

        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        if name.startswith("conv2d_") and name != "conv2d_weight_prep":
            # label by the kernel family the C side dispatched to (what rocprofv3 lists), keeping the entry point
            fam = lib().m355_last_kernel().decode()
            name_t = fam + " [" + name_t + "]" if (TIMER_TAGS and tag) else fam
        if callable(exec_ratio):
            exec_ratio = exec_ratio()
        _TIMER_EVENTS.append((name_t, e0, e1, float(work), float(work) * float(exec_ratio)))
    else:
        rc = fn(*args)
    check(rc, name)


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


_RAW_STREAM = None


def stream():
    """the current HIP stream as an integer handle.  torch.cuda.current_stream() builds a Stream object through several Python
    layers (9 us per call, 2 ms per GAN cycle -- as much as the host can spare at small batch); the raw accessor is a C call"""
    global _RAW_STREAM
    if _RAW_STREAM is None:
        import torch

        raw, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if raw is not None and dev is not None:
            _RAW_STREAM = lambda: raw(dev())
        else:
            _RAW_STREAM = lambda: torch.cuda.current_stream().cuda_stream
    return _RAW_STREAM()
