"""torch.autograd bindings of the libm355 projection entry points (include/m355.h).

Each Function's forward/backward launches HIP kernels on torch's current stream through ctypes; torch is
used for device memory and stream plumbing only.  No op here has a CPU or PyTorch fallback.
"""
import torch

from . import _lib
from ._lib import check, lib, ptr, stream

FIXED_WEIGHTS = 1
TAPS_FROM_SIGMA = 2
DET_SPLAT = 8        # M355_DET_SPLAT: order-independent occupancy splat (set by EffectiveLossFunction in deterministic mode)
TRUE_GAUSSIAN = 4

FOV = 1.875       # utils/effective_loss_function.py:69
CAM_DIST = 2.0    # utils/effective_loss_function.py:70


enable_kernel_timers = _lib.enable_kernel_timers
collect_kernel_timers = _lib.collect_kernel_timers
_launch = _lib.launch


def _f32c(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.M355Error(f"{name} must be a CUDA(HIP) tensor; the hot path has no CPU implementation")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


class CameraTransform(torch.autograd.Function):
    """CameraUtilities.transformation_3d_coord_to_camera_coord (camera/coordinate_system_transformation.py:20-39)."""

    @staticmethod
    def forward(ctx, pc, q, fov, dist):
        pc, q = _f32c(pc.detach(), "point_cloud"), _f32c(q.detach(), "rotation")
        B, N, _ = pc.shape
        cam = torch.empty_like(pc)
        _launch("proj_transform_fwd", ptr(pc), ptr(q), ptr(cam), B, N, fov, dist, stream())
        ctx.save_for_backward(pc, q)
        ctx.fd = (fov, dist)
        return cam

    @staticmethod
    def backward(ctx, dcam):
        pc, q = ctx.saved_tensors
        B, N, _ = pc.shape
        dcam = _f32c(dcam, "grad")
        dpc = torch.empty_like(pc)
        dq = torch.empty_like(q)
        _launch("proj_transform_bwd", ptr(pc), ptr(q), ptr(dcam), 1, 0, ptr(dpc), ptr(dq), None, 0, None, B, N,
                                            ctx.fd[0], ctx.fd[1], stream())
        return dpc, dq, None, None


class QuatRotate(torch.autograd.Function):
    """PointsQuaternionsRotator.rotate_points (quaternions/points_quaternions.py:41-81), either direction."""

    @staticmethod
    def forward(ctx, pc, q, inverse):
        pc, q = _f32c(pc.detach(), "xyz_triplet"), _f32c(q.detach(), "q")
        B, N, _ = pc.shape
        out = torch.empty_like(pc)
        _launch("quat_rotate_fwd", ptr(pc), ptr(q), ptr(out), B, N, int(inverse), stream())
        ctx.save_for_backward(pc, q)
        ctx.inverse = int(inverse)
        return out

    @staticmethod
    def backward(ctx, dout):
        pc, q = ctx.saved_tensors
        B, N, _ = pc.shape
        dpc, dq = torch.empty_like(pc), torch.empty_like(q)
        _launch("quat_rotate_bwd", ptr(pc), ptr(q), ptr(_f32c(dout, "grad")), ptr(dpc), ptr(dq), B, N, ctx.inverse, stream())
        return dpc, dq, None


class ProjectSilhouette(torch.autograd.Function):
    """EffectiveLossFunction.forward (utils/effective_loss_function.py:58-81), fused:
    camera transform -> splat -> depth smoothing -> scale/clamp -> termination -> depth sum -> flip."""

    @staticmethod
    def forward(ctx, pc, q, scale, taps_or_sigma, ntaps, S, flags):
        pc, q = _f32c(pc.detach(), "point_cloud"), _f32c(q.detach(), "rotation")
        scale_shape = None if scale is None else tuple(scale.shape)
        scale = None if scale is None else _f32c(scale.detach(), "scale").reshape(-1)
        tp = _f32c(taps_or_sigma.detach(), "taps/sigma").reshape(-1)
        B, N, _ = pc.shape
        if q.shape[0] != B or (scale is not None and scale.numel() != B):
            raise ValueError(f"batch mismatch: point_cloud {tuple(pc.shape)}, rotation {tuple(q.shape)}")
        L, st = lib(), stream()
        ntiles = L.m355_proj_ntiles(S)
        if ntiles < 0:
            raise _lib.M355Error(f"voxel_size={S} is not supported by the fused renderer (2..512)")
        cam = torch.empty_like(pc)
        tstart = torch.empty((B, ntiles + 1), dtype=torch.int32, device=pc.device)
        tpts = torch.empty((B, 4 * N, 4), dtype=torch.float32, device=pc.device)
        proj = torch.empty((B, S, S), dtype=torch.float32, device=pc.device)
        if flags & TAPS_FROM_SIGMA:
            # VoxelsSmooth.separate_kernels once per call (3 us) instead of once per workgroup inside the renderer
            taps = torch.empty((ntaps,), dtype=torch.float32, device=pc.device)
            _launch("smooth_taps", ptr(tp), ntaps, flags & TRUE_GAUSSIAN, ptr(taps), st)
            tp, flags = taps, flags & ~(TAPS_FROM_SIGMA | TRUE_GAUSSIAN)
        _launch("proj_bin_fwd", ptr(pc), ptr(q), ptr(cam), None, ptr(tstart), ptr(tpts), B, N, S, FOV, CAM_DIST, st)
        _launch("proj_render_fwd", ptr(tstart), ptr(tpts), ptr(scale), ptr(tp), ntaps, ptr(proj), B, N, S, flags, st,
                work=B * (12 * N + 20 + 8 * S ** 3 + 4 * S ** 2))
        ctx.save_for_backward(pc, q, tstart, tpts, tp, *(() if scale is None else (scale,)))
        ctx.cfg = (ntaps, S, flags, scale_shape)
        return proj

    @staticmethod
    def backward(ctx, dproj):
        return _project_backward(ctx, _f32c(dproj, "grad"), 1.0)


def _project_backward(ctx, dproj, gmul):
    ntaps, S, flags, scale_shape = ctx.cfg
    has_scale = scale_shape is not None
    if has_scale:
        pc, q, tstart, tpts, tp, scale = ctx.saved_tensors
    else:
        pc, q, tstart, tpts, tp = ctx.saved_tensors
        scale = None
    B, N, _ = pc.shape
    L, st = lib(), stream()
    nparts = L.m355_proj_ntiles(S)
    slots = torch.empty((B, N, 4, 3), dtype=torch.float32, device=pc.device)
    dsp = torch.empty((B, nparts), dtype=torch.float32, device=pc.device) if has_scale else None
    _launch("proj_render_bwd", ptr(tstart), ptr(tpts), ptr(scale), ptr(tp), ntaps, ptr(dproj), gmul, ptr(slots), ptr(dsp),
            B, N, S, flags, st, work=B * (4 * S ** 2 + 12 * S ** 3 + 24 * N + 20))
    dpc = torch.empty_like(pc)
    dq = torch.empty_like(q)
    dscale = torch.empty((B,), dtype=torch.float32, device=pc.device) if has_scale else None
    _launch("proj_transform_bwd", ptr(pc), ptr(q), ptr(slots), 4, 1, ptr(dpc), ptr(dq), ptr(dsp), nparts, ptr(dscale),
                                    B, N, FOV, CAM_DIST, st)
    return dpc, dq, (dscale.reshape(scale_shape) if has_scale else None), None, None, None, None


class SilhouetteSSE(torch.autograd.Function):
    """mask[B,2S,2S] -> bilinear 1/2 (align_corners) -> per-cloud and total squared error against proj[B,S,S]
    (models/supervised_part.py:70-72, models/unsupervised_part.py:108-116).  Returns (total[1], sse[B])."""

    @staticmethod
    def forward(ctx, proj, mask, mask_repeat):
        proj, mask = _f32c(proj.detach(), "projection"), _f32c(mask.detach(), "masks")
        B, S, _ = proj.shape
        if mask.dim() != 3 or mask.shape[0] * mask_repeat != B:
            raise ValueError(f"masks {tuple(mask.shape)} x{mask_repeat} do not match projection {tuple(proj.shape)}")
        Hin, Win = mask.shape[-2:]
        L, st = lib(), stream()
        diff = torch.empty_like(proj)
        sse = torch.empty((B,), dtype=torch.float32, device=proj.device)
        total = torch.empty((1,), dtype=torch.float32, device=proj.device)
        ws = torch.empty((max(L.m355_sil_loss_ws_bytes(B, S), 8),), dtype=torch.uint8, device=proj.device)
        _launch("sil_loss_fwd", ptr(proj), ptr(mask), Hin, Win, mask_repeat, ptr(diff), ptr(sse), ptr(total), ptr(ws), B, S, st)
        ctx.save_for_backward(diff)
        ctx.set_materialize_grads(False)
        return total, sse

    @staticmethod
    def backward(ctx, gtotal, gsse):
        (diff,) = ctx.saved_tensors
        # d total/d proj = 2 diff ; d sse[b]/d proj[b] = 2 diff[b]
        g = 2.0 * diff
        out = None
        if gtotal is not None:
            out = g * gtotal.reshape(1, 1, 1)
        if gsse is not None:
            t = g * gsse.reshape(-1, 1, 1)
            out = t if out is None else out + t
        return out, None, None


def camera_transform(pc, q, fov=FOV, dist=CAM_DIST):
    return CameraTransform.apply(pc, q, float(fov), float(dist))


def rotate_points(pc, q, inverse=False):
    """[B,N,3] points rotated by F.normalize(q [B,4]); bit-exact with the reference's Hamilton products"""
    if pc.dim() != 3 or pc.shape[-1] != 3 or q.dim() != 2 or q.shape != (pc.shape[0], 4):
        raise ValueError(f"rotate_points: expected xyz [B,N,3] and q [B,4], got {tuple(pc.shape)} / {tuple(q.shape)}")
    return QuatRotate.apply(pc, q, bool(inverse))


def project_silhouette(pc, q, scale, taps_or_sigma, ntaps, S, flags):
    return ProjectSilhouette.apply(pc, q, scale, taps_or_sigma, int(ntaps), int(S), int(flags))


def silhouette_sse(proj, mask, mask_repeat=1):
    return SilhouetteSSE.apply(proj, mask, int(mask_repeat))


def smooth_taps(sigma, ntaps=21, true_gaussian=False):
    """VoxelsSmooth.separate_kernels (utils/smooth_voxels.py:14-42) on device: sigma scalar tensor -> taps[ntaps]."""
    sigma = _f32c(sigma.detach().reshape(-1), "sigma")
    taps = torch.empty((ntaps,), dtype=torch.float32, device=sigma.device)
    _launch("smooth_taps", ptr(sigma), ntaps, TRUE_GAUSSIAN if true_gaussian else 0, ptr(taps), stream())
    return taps


def chamfer_nn(a, b):
    """a[B,N,3], b[B,M,3] -> (dist[B,N], idx[B,N] int32): squared distance to / index of the nearest point of b
    (new capability, BASELINE configs[4]; no reference implementation exists)."""
    a, b = _f32c(a.detach(), "a"), _f32c(b.detach(), "b")
    B, N, _ = a.shape
    M = b.shape[1]
    dist = torch.empty((B, N), dtype=torch.float32, device=a.device)
    idx = torch.empty((B, N), dtype=torch.int32, device=a.device)
    nws = lib().m355_chamfer_nn_ws_bytes(B, N, M)   # small batches: the target sweep is split over workgroups too
    ws = torch.empty((nws,), dtype=torch.uint8, device=a.device) if nws else None
    _launch("chamfer_nn_fwd_ws", ptr(a), ptr(b), ptr(dist), ptr(idx), B, N, M, ptr(ws), stream(), work=8.0 * B * N * M)
    return dist, idx


def chamfer_distance(a, b):
    """symmetric Chamfer distance mean_i min_j |a_i-b_j|^2 + mean_j min_i |b_j-a_i|^2, differentiable in a and b:
    the nearest-neighbour search runs in HIP, the gradient flows through a gather of the matched pairs."""
    _, ia = chamfer_nn(a, b)
    _, ib = chamfer_nn(b, a)
    nb = torch.gather(b, 1, ia.long().unsqueeze(-1).expand(-1, -1, 3))
    na = torch.gather(a, 1, ib.long().unsqueeze(-1).expand(-1, -1, 3))
    return ((a - nb) ** 2).sum(-1).mean(1) + ((b - na) ** 2).sum(-1).mean(1)


# ------------------------------------------------------------------------------------------------ dense stage API
class Trilinear(torch.autograd.Function):
    """TrilinearInterpolation.trilinear_interpolation (utils/trilinear_interpolation.py:62-74): cam -> clamped volume"""

    @staticmethod
    def forward(ctx, cam, S, flags):
        cam = _f32c(cam.detach(), "point_cloud")
        B, N, _ = cam.shape
        L, st = lib(), stream()
        ntiles = L.m355_proj_ntiles(S)
        if ntiles < 0:
            raise _lib.M355Error(f"size={S} is not supported (2..512)")
        tstart = torch.empty((B, ntiles + 1), dtype=torch.int32, device=cam.device)
        tpts = torch.empty((B, 4 * N, 4), dtype=torch.float32, device=cam.device)
        vol = torch.empty((B, S, S, S), dtype=torch.float32, device=cam.device)
        _launch("proj_bin_fwd", None, None, ptr(cam), None, ptr(tstart), ptr(tpts), B, N, S, FOV, CAM_DIST, st)
        _launch("trilinear_fwd", ptr(tstart), ptr(tpts), ptr(vol), None, B, N, S, flags, st, work=B * (12 * N + 4 * S ** 3))
        ctx.save_for_backward(tstart, tpts)
        ctx.cfg = (B, N, S, flags)
        return vol

    @staticmethod
    def backward(ctx, dvol):
        tstart, tpts = ctx.saved_tensors
        B, N, S, flags = ctx.cfg
        dvol = _f32c(dvol, "grad")
        dcam = torch.empty((B, N, 3), dtype=torch.float32, device=dvol.device)
        _launch("trilinear_bwd", ptr(tstart), ptr(tpts), ptr(dvol), ptr(dcam), B, N, S, flags, stream())
        return dcam, None, None


class Smooth(torch.autograd.Function):
    """VoxelsSmooth.smooth (utils/smooth_voxels.py:44-84) as a chain of 1-D convolutions + the scale/clamp epilogue"""

    @staticmethod
    def forward(ctx, vox, scale, taps, axes):
        vox = _f32c(vox.detach(), "voxels")
        B, S = vox.shape[0], vox.shape[1]
        sc = None if scale is None else _f32c(scale.detach(), "scale").reshape(-1)
        st = stream()
        cur = vox
        for i, (tp, ax) in enumerate(zip(taps, axes)):
            last = i == len(axes) - 1
            out = torch.empty_like(vox)
            _launch("smooth_axis", ptr(cur), ptr(out), ptr(tp), tp.numel(), ax, ptr(sc) if last else None, 0, B, S, st,
                    work=8.0 * vox.numel())
            if last and sc is not None:
                ctx.pre_in = cur  # the epilogue's input is recomputed in backward from this
            cur = out
        ctx.save_for_backward(*taps, *(() if sc is None else (sc,)))
        ctx.cfg = (axes, sc is not None, None if scale is None else tuple(scale.shape))
        return cur

    @staticmethod
    def backward(ctx, dout):
        axes, has_scale, scale_shape = ctx.cfg
        saved = ctx.saved_tensors
        taps = saved[:len(axes)]
        g = _f32c(dout, "grad")
        B, S = g.shape[0], g.shape[1]
        st = stream()
        dscale = None
        if has_scale:
            sc = saved[-1]
            pre = torch.empty_like(g)
            _launch("smooth_axis", ptr(ctx.pre_in), ptr(pre), ptr(taps[-1]), taps[-1].numel(), axes[-1], None, 0, B, S, st)
            dpre = torch.empty_like(g)
            dscale = torch.empty((B,), dtype=torch.float32, device=g.device)
            _launch("scale_clamp_bwd", ptr(pre), ptr(sc), ptr(g), ptr(dpre), ptr(dscale), B, S ** 3, st)
            g, dscale = dpre, dscale.reshape(scale_shape)
        for tp, ax in zip(reversed(taps), reversed(axes)):
            out = torch.empty_like(g)
            _launch("smooth_axis", ptr(g), ptr(out), ptr(tp), tp.numel(), ax, None, 1, B, S, st)
            g = out
        return g, dscale, None, None


class Termination(torch.autograd.Function):
    """EffectiveLossFunction.termination_probs (utils/effective_loss_function.py:18-56)"""

    @staticmethod
    def forward(ctx, vox, eps):
        vox = _f32c(vox.detach(), "voxels")
        B, D, H, W = vox.shape
        T = torch.empty((B, D + 1, H, W), dtype=torch.float32, device=vox.device)
        _launch("termination_fwd", ptr(vox), ptr(T), B, D, H, W, eps, stream(), work=8.0 * vox.numel())
        ctx.save_for_backward(vox)
        ctx.eps = eps
        return T

    @staticmethod
    def backward(ctx, dT):
        (vox,) = ctx.saved_tensors
        B, D, H, W = vox.shape
        dT = _f32c(dT, "grad")
        dvol = torch.empty_like(vox)
        _launch("termination_bwd", ptr(vox), ptr(dT), ptr(dvol), B, D, H, W, ctx.eps, stream())
        return dvol, None


def trilinear(cam, S, fixed_weights=False):
    return Trilinear.apply(cam, int(S), FIXED_WEIGHTS if fixed_weights else 0)


def smooth(vox, taps, axes, scale=None):
    return Smooth.apply(vox, scale, tuple(taps), tuple(int(a) for a in axes))


def termination_probs(vox, eps=1e-5):
    return Termination.apply(vox, float(eps))
