"""Drop-in for the reference's mesh renderer: code/rendering/renderer.py (`Renderer`,
`ortho_projection`) and code/rendering/fragment_shader.py (`fragmentshader`, `texinterpolation`), with the Kaolin DIB-R
rasteriser they import replaced by csrc/dibr_raster.hip (`linear_rasterizer` below keeps kaolin's call signature).

Kaolin is not part of this image and its sources are not under /root/reference: the rasteriser follows the published DIB-R
algorithm with kaolin's default constants (see include/m355.h, oracle/raster_ref.py); parity with the real Kaolin build is
UNPINNED.  No CPU fallback: the rasteriser and the bilinear shader raise on CPU tensors.
"""
import torch
import torch.nn as nn

from . import _lib
from ._lib import launch, lib, ptr, stream
from .conv import is_deterministic

EXPAND, KNUM, MULTIPLIER, DELTA = 0.02, 30, 1000, 7000.0   # kaolin.graphics.dib_renderer.rasterizer defaults


def _f32c(t, name):
    if not t.is_cuda:
        raise _lib.M355Error(f"{name} must be a CUDA(HIP) tensor; the renderer has no CPU implementation")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


class LinearRasterizerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, height, width, points3d, points2d, normalz, attr, expand, knum, delta):
        points3d, points2d = _f32c(points3d.detach(), "points3d_bxfx9"), _f32c(points2d.detach(), "points2d_bxfx6")
        normalz, attr = _f32c(normalz.detach(), "normalz_bxfx1"), _f32c(attr.detach(), "vertex_attr_bxfx3d")
        B, F, _ = points2d.shape
        if points3d.shape != (B, F, 9) or points2d.shape[2] != 6 or normalz.numel() != B * F or attr.shape[2] % 3:
            raise ValueError("linear_rasterizer: expected points3d [B,F,9], points2d [B,F,6], normalz [B,F,1], attr [B,F,3D]")
        D = attr.shape[2] // 3
        dev = points2d.device
        ws = torch.empty((lib().m355_dibr_ws_bytes(B, F),), dtype=torch.uint8, device=dev)
        imfeat = torch.empty((B, height, width, D), dtype=torch.float32, device=dev)
        improb = torch.empty((B, height, width, 1), dtype=torch.float32, device=dev)
        imidx = torch.empty((B, height, width), dtype=torch.int32, device=dev)
        imwei = torch.empty((B, height, width, 3), dtype=torch.float32, device=dev)
        launch("dibr_rasterize_fwd", height, width, ptr(points3d), ptr(points2d), ptr(normalz), ptr(attr), B, F, D, float(expand),
               int(knum), float(delta), ptr(ws), ptr(imfeat), ptr(improb), ptr(imidx), ptr(imwei), stream())
        ctx.save_for_backward(points3d, points2d, attr, ws, improb, imidx, imwei)
        ctx.cfg = (height, width, B, F, D, int(knum), float(delta))
        ctx.mark_non_differentiable(imidx)
        return imfeat, improb, imidx

    @staticmethod
    def backward(ctx, dfeat, dprob, _didx=None):
        points3d, points2d, attr, ws, improb, imidx, imwei = ctx.saved_tensors
        height, width, B, F, D, knum, delta = ctx.cfg
        dev = points2d.device
        dfeat = torch.zeros((B, height, width, D), device=dev) if dfeat is None else _f32c(dfeat, "grad")
        dprob = torch.zeros((B, height, width, 1), device=dev) if dprob is None else _f32c(dprob, "grad")
        dp2 = torch.empty_like(points2d)
        dattr = torch.empty_like(attr)
        if is_deterministic():   # sums over pixels in 64-bit fixed point: the same bits on every run
            fix = torch.empty((lib().m355_dibr_rasterize_bwd_det_ws_bytes(B, F, D),), dtype=torch.uint8, device=dev)
            launch("dibr_rasterize_bwd_det", height, width, ptr(points3d), ptr(points2d), ptr(attr), B, F, D, knum, delta, ptr(ws),
                   ptr(improb), ptr(imidx), ptr(imwei), ptr(dfeat), ptr(dprob), ptr(fix), ptr(dp2), ptr(dattr), stream())
            return None, None, None, dp2, None, dattr, None, None, None
        launch("dibr_rasterize_bwd", height, width, ptr(points3d), ptr(points2d), ptr(attr), B, F, D, knum, delta, ptr(ws),
               ptr(improb), ptr(imidx), ptr(imwei), ptr(dfeat), ptr(dprob), ptr(dp2), ptr(dattr), stream())
        return None, None, None, dp2, None, dattr, None, None, None


def linear_rasterizer(height, width, points3d_bxfx9, points2d_bxfx6, normalz_bxfx1, vertex_attr_bxfx3d, expand=None, knum=None,
                      multiplier=None, delta=None, debug=False):
    """kaolin.graphics.dib_renderer.rasterizer.linear_rasterizer as renderer.py:62-69 calls it -> (imfeat [B,H,W,D],
    improb [B,H,W,1]).  `multiplier` only rescales kaolin's internal coordinates and is accepted for compatibility."""
    imfeat, improb, _ = LinearRasterizerFn.apply(int(height), int(width), points3d_bxfx9, points2d_bxfx6, normalz_bxfx1,
                                                 vertex_attr_bxfx3d, EXPAND if expand is None else expand,
                                                 KNUM if knum is None else knum, DELTA if delta is None else delta)
    return imfeat, improb


def datanormalize(data, axis):
    """kaolin.graphics.dib_renderer.utils.datanormalize (renderer.py:2,52): L2 normalisation with a 1e-8 guard"""
    return data / (torch.sqrt(torch.sum(data ** 2, dim=axis, keepdim=True)) + 1e-8)


def ortho_projection(points_bxpx3, faces_fx3):
    """renderer.py:9-30: per-face vertex triples (3-D and x,y), un-normalised face normals"""
    pf0, pf1, pf2 = (points_bxpx3[:, faces_fx3[:, k], :] for k in range(3))
    points3d_bxfx9 = torch.cat((pf0, pf1, pf2), dim=2)
    points2d_bxfx6 = torch.cat((pf0[:, :, :2], pf1[:, :, :2], pf2[:, :, :2]), dim=2)
    normal_bxfx3 = torch.cross(pf1 - pf0, pf2 - pf0, dim=2)
    return points3d_bxfx9, points2d_bxfx6, normal_bxfx3


class ShadeFn(torch.autograd.Function):
    """fragmentshader for filtering='bilinear' in one pass each way (csrc/dibr_raster.hip k_shade)"""

    @staticmethod
    def forward(ctx, uvm, texture, background):
        uvm, texture = _f32c(uvm, "imfeat"), _f32c(texture, "texture")
        bg = None if background is None else _f32c(background, "background_image")
        B, H, W, _ = uvm.shape
        color = torch.empty((B, H, W, 3), dtype=torch.float32, device=uvm.device)
        launch("dibr_shade_fwd", ptr(uvm), ptr(texture), ptr(bg), ptr(color), B, H, W, texture.shape[2], texture.shape[3], stream())
        ctx.save_for_backward(uvm, texture, *(() if bg is None else (bg,)))
        return color

    @staticmethod
    def backward(ctx, dcolor):
        uvm, texture, *rest = ctx.saved_tensors
        bg = rest[0] if rest else None
        B, H, W, _ = uvm.shape
        duvm = torch.empty_like(uvm)
        dtex = torch.empty_like(texture) if ctx.needs_input_grad[1] else None
        dbg = torch.empty_like(bg) if (bg is not None and ctx.needs_input_grad[2]) else None
        if is_deterministic() and dtex is not None:
            fix = torch.empty((lib().m355_dibr_shade_bwd_det_ws_bytes(B, texture.shape[2], texture.shape[3]),), dtype=torch.uint8,
                              device=uvm.device)
            launch("dibr_shade_bwd_det", ptr(uvm), ptr(texture), ptr(bg), ptr(_f32c(dcolor, "grad")), ptr(fix), ptr(duvm), ptr(dtex),
                   ptr(dbg), B, H, W, texture.shape[2], texture.shape[3], stream())
            return duvm, dtex, dbg
        launch("dibr_shade_bwd", ptr(uvm), ptr(texture), ptr(bg), ptr(_f32c(dcolor, "grad")), ptr(duvm), ptr(dtex), ptr(dbg), B, H, W,
               texture.shape[2], texture.shape[3], stream())
        return duvm, dtex, dbg


def texinterpolation(imtexcoord_bxhxwx2, texture_bx3xthxtw, filtering='bilinear'):
    """fragment_shader.py:6-20 (torch ops; the fused path is `fragmentshader`)"""
    g = (imtexcoord_bxhxwx2 * 2 - 1) * imtexcoord_bxhxwx2.new_tensor([1.0, -1.0])
    if filtering == 'bilinear':
        t = torch.nn.functional.grid_sample(texture_bx3xthxtw, g, mode='bilinear', align_corners=True)
    else:
        t = torch.nn.functional.grid_sample(texture_bx3xthxtw, g, mode=filtering)
    return t.permute(0, 2, 3, 1)


def fragmentshader(imtexcoord_bxhxwx2, texture_bx3xthxtw, improb_bxhxwx1, filtering='bilinear', background_image=None):
    """fragment_shader.py:22-37"""
    if filtering == 'bilinear' and imtexcoord_bxhxwx2.is_cuda and texture_bx3xthxtw.shape[1] == 3:
        return ShadeFn.apply(torch.cat((imtexcoord_bxhxwx2, improb_bxhxwx1), dim=3), texture_bx3xthxtw, background_image)
    tex = texinterpolation(imtexcoord_bxhxwx2, texture_bx3xthxtw, filtering)
    if background_image is None:
        return tex * improb_bxhxwx1
    return torch.lerp(background_image, tex, improb_bxhxwx1)


class Renderer(nn.Module):
    """rendering/renderer.py:32-77"""

    def __init__(self, height, width, filtering='bilinear'):
        super().__init__()
        self.height, self.width, self.filtering = height, width, filtering

    def forward(self, points, uv_bxpx2, texture_bx3xthxtw, ft_fx3=None, background_image=None, return_hardmask=False):
        points_bxpx3, faces_fx3 = points
        if ft_fx3 is None:
            ft_fx3 = faces_fx3
        points3d_bxfx9, points2d_bxfx6, normal_bxfx3 = ortho_projection(points_bxpx3, faces_fx3)
        normalz_bxfx1 = normal_bxfx3[:, :, 2:3]                       # front / back faces
        normal1_bxfx3 = datanormalize(normal_bxfx3, axis=2)
        c0, c1, c2 = (uv_bxpx2[:, ft_fx3[:, k], :] for k in range(3))
        mask = torch.ones_like(c0[:, :, :1])
        uv_bxfx9 = torch.cat((c0, mask, c1, mask, c2, mask), dim=2)
        imfeat, improb_bxhxwx1 = linear_rasterizer(self.height, self.width, points3d_bxfx9, points2d_bxfx6, normalz_bxfx1, uv_bxfx9)
        hardmask = imfeat[:, :, :, 2:3]
        if self.filtering == 'bilinear' and imfeat.is_cuda and texture_bx3xthxtw.shape[1] == 3:
            imrender = ShadeFn.apply(imfeat, texture_bx3xthxtw, background_image)   # (u, v, hard) as rasterised: no cat
        else:
            imrender = fragmentshader(imfeat[:, :, :, :2], texture_bx3xthxtw, hardmask, filtering=self.filtering,
                                      background_image=background_image)
        if return_hardmask:
            improb_bxhxwx1 = hardmask
        return imrender, improb_bxhxwx1, normal1_bxfx3
