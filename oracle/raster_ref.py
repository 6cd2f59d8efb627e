"""TEST INFRASTRUCTURE ONLY -- per-pixel brute-force restatement (differentiable torch ops) of the DIB-R linear
rasteriser that `Renderer.forward` (/root/reference/code/rendering/renderer.py:39-77) calls.

PARITY UNPINNED.  The algorithm lives in a third-party dependency that is absent from /root/reference and from this
image: `kaolin.graphics.dib_renderer.rasterizer.linear_rasterizer` of NVIDIAGameWorks/kaolin at commit e7e513173bd4
(the commit code/rendering/monkey_patches.py:4 names), i.e. the DIB-R renderer of Chen et al., "Learning to Predict 3D
Objects with an Interpolation-based Differentiable Renderer" (NeurIPS 2019).  What is restated here is the published
algorithm with kaolin's default constants (expand 0.02, knum 30, multiplier 1000 -- a pure rescaling that cancels --,
delta 7000); there is no golden vector of the real kaolin to check it against, so the HIP rasteriser is compared with
THIS file only, and both are anchored on the reference's call site (argument layout, output shapes, the derived
quantities of renderer.py:46-77) and on properties (coverage of a known triangle, barycentric partition of unity, ...).

  pixel (h, w) centre in normalised device coordinates:  x = (2 w + 1 - W) / W,   y = (H - 2 h - 1) / H      (y up)
  hard pass   : among the front faces (normal z >= 0) whose triangle contains the centre (all three barycentric weights
                >= 0) the one with the largest interpolated depth z wins;  im = sum_k w_k attr_k  (zero where no face)
  soft pass   : improb = 1 where a face covers the pixel; else 1 - prod_j (1 - exp(-delta d_j^2)) over the front faces j
                whose bounding box, grown by `expand`, contains the centre -- the first `knum` of them in face order --
                with d_j the Euclidean distance from the centre to triangle j.
"""
import torch


def _cross2(a, b):
    return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]


def _seg_dist2(p, a, b):
    """squared distance from points p [...,2] to segments a-b [...,2] (broadcast)"""
    e = b - a
    t = ((p - a) * e).sum(-1) / (e * e).sum(-1).clamp_min(1e-30)
    t = t.clamp(0.0, 1.0)
    r = p - (a + t.unsqueeze(-1) * e)
    return (r * r).sum(-1)


def linear_rasterizer_ref(height, width, points3d_bxfx9, points2d_bxfx6, normalz_bxfx1, attr_bxfxd3, expand=0.02, knum=30,
                          delta=7000.0, rows=None):
    """-> imfeat [B,H,W,D], improb [B,H,W,1], imidx [B,H,W] (covering face or -1), imwei [B,H,W,3].
    rows = (r0, r1): only the pixel rows r0 .. r1-1 of the height x width image (outputs [B,r1-r0,W,...]) -- the per-pixel
    brute force holds B x rows x W x F intermediates, so large images are rendered (and differentiated) in row bands"""
    B, F, _ = points2d_bxfx6.shape
    D = attr_bxfxd3.shape[2] // 3
    dt, dev = points2d_bxfx6.dtype, points2d_bxfx6.device
    xs = (2 * torch.arange(width, dtype=dt, device=dev) + 1 - width) / width
    ys = (height - 2 * torch.arange(height, dtype=dt, device=dev) - 1) / height
    if rows is not None:
        ys = ys[rows[0]:rows[1]]
    height = ys.shape[0]   # (from here on: the number of rows rendered)
    P = torch.stack((xs[None, :].expand(height, width), ys[:, None].expand(height, width)), dim=-1)   # [H,W,2]
    p = P[None, :, :, None, :]                                                                        # [1,H,W,1,2]
    v = points2d_bxfx6.view(B, 1, 1, F, 3, 2)
    v0, v1, v2 = v[..., 0, :], v[..., 1, :], v[..., 2, :]
    area = _cross2(v1 - v0, v2 - v0)                                   # [B,1,1,F]
    ok_area = area.abs() > 1e-12
    safe = torch.where(ok_area, area, torch.ones_like(area))
    w0 = _cross2(v1 - p, v2 - p) / safe
    w1 = _cross2(v2 - p, v0 - p) / safe
    w2 = 1.0 - w0 - w1
    front = (normalz_bxfx1.view(B, 1, 1, F) >= 0) & ok_area
    inside = front & (w0 >= 0) & (w1 >= 0) & (w2 >= 0)
    z = points3d_bxfx9.view(B, 1, 1, F, 3, 3)[..., 2]                  # [B,1,1,F,3]
    zi = w0 * z[..., 0] + w1 * z[..., 1] + w2 * z[..., 2]
    zi = torch.where(inside, zi, torch.full_like(zi, -float("inf")))
    zbest, idx = zi.max(dim=-1)                                        # [B,H,W]
    covered = torch.isfinite(zbest)
    imidx = torch.where(covered, idx, torch.full_like(idx, -1))
    g = idx.unsqueeze(-1)
    wsel = torch.stack((w0.gather(-1, g), w1.gather(-1, g), w2.gather(-1, g)), dim=-1).squeeze(-2)    # [B,H,W,3]
    wsel = torch.where(covered.unsqueeze(-1), wsel, torch.zeros_like(wsel))
    attr = attr_bxfxd3.view(B, F, 3, D)
    asel = attr[torch.arange(B, device=dev)[:, None, None], idx]       # [B,H,W,3,D]
    imfeat = (wsel.unsqueeze(-1) * asel).sum(-2)
    # ---- soft silhouette
    xmin = torch.minimum(torch.minimum(v0[..., 0], v1[..., 0]), v2[..., 0]) - expand
    xmax = torch.maximum(torch.maximum(v0[..., 0], v1[..., 0]), v2[..., 0]) + expand
    ymin = torch.minimum(torch.minimum(v0[..., 1], v1[..., 1]), v2[..., 1]) - expand
    ymax = torch.maximum(torch.maximum(v0[..., 1], v1[..., 1]), v2[..., 1]) + expand
    near = front & (p[..., 0] >= xmin) & (p[..., 0] < xmax) & (p[..., 1] >= ymin) & (p[..., 1] < ymax)
    near = near & (near.to(torch.int32).cumsum(-1) <= knum)            # the first knum candidates in face order
    d2 = torch.minimum(torch.minimum(_seg_dist2(p, v0, v1), _seg_dist2(p, v1, v2)), _seg_dist2(p, v2, v0))
    a = torch.exp(-delta * d2)
    keep = torch.where(near, 1.0 - a, torch.ones_like(a))
    improb = 1.0 - keep.prod(dim=-1)
    improb = torch.where(covered, torch.ones_like(improb), improb)
    return imfeat, improb.unsqueeze(-1), imidx, wsel


def ortho_projection_ref(points_bxpx3, faces_fx3):
    """renderer.py:9-30"""
    pf = [points_bxpx3[:, faces_fx3[:, k], :] for k in range(3)]
    points3d = torch.cat(pf, dim=2)
    points2d = torch.cat([t[:, :, :2] for t in pf], dim=2)
    normal = torch.cross(pf[1] - pf[0], pf[2] - pf[0], dim=2)
    return points3d, points2d, normal


def renderer_forward_ref(points, uv_bxpx2, texture_bx3xthxtw, height, width, ft_fx3=None, background_image=None,
                         return_hardmask=False, rows=None):
    """Renderer.forward (renderer.py:39-77) + fragmentshader (fragment_shader.py:6-37) on the rasteriser above"""
    import torch.nn.functional as F
    points_bxpx3, faces_fx3 = points
    if ft_fx3 is None:
        ft_fx3 = faces_fx3
    points3d, points2d, normal = ortho_projection_ref(points_bxpx3, faces_fx3)
    normalz = normal[:, :, 2:3]
    normal1 = normal / (torch.sqrt((normal ** 2).sum(dim=2, keepdim=True)) + 1e-8)   # kaolin datanormalize (L2, 1e-8 guard)
    c = [uv_bxpx2[:, ft_fx3[:, k], :] for k in range(3)]
    one = torch.ones_like(c[0][:, :, :1])
    uv9 = torch.cat((c[0], one, c[1], one, c[2], one), dim=2)
    imfeat, improb, _, _ = linear_rasterizer_ref(height, width, points3d, points2d, normalz, uv9, rows=rows)
    tc, hard = imfeat[..., :2], imfeat[..., 2:3]
    grid = (tc * 2 - 1) * tc.new_tensor([1.0, -1.0])
    tex = F.grid_sample(texture_bx3xthxtw, grid, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    color = tex * hard if background_image is None else torch.lerp(background_image, tex, hard)
    return color, (hard if return_hardmask else improb), normal1
