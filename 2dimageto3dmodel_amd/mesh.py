"""Mesh-template deformation, face normals and the flat (smoothness) loss -- SURVEY.md 8f row 1, "the other half of
the G step" (code/main.py:697-699):

    vtx = mesh_template.get_vertex_positions(pred_mesh)
    flat_loss = loss_flat(mesh_template.mesh, mesh_template.compute_normals(vtx))

`MeshTemplate` mirrors code/rendering/mesh_template.py:12-149 (same constructor arguments, attribute names, method
names and tensor shapes) without Kaolin: the OBJ reader below follows what kaolin v0.1's `TriangleMesh.from_obj`
keeps (v -> vertices, vt -> uvs, f a/b[/c] -> faces / face_textures, 0-based), and `face_adjacency` produces the
`mesh.ff` table of kaolin's `compute_adjacency_info` (code/rendering/monkey_patches.py:8-155) for a closed triangle
mesh (one neighbour per edge).  The per-step work (bilinear sampling of the displacement map, tangent-frame matmul,
symmetry scatter, cross products, normalisation, neighbour gathers, reduction) runs in libm355 kernels
(csrc/mesh_deform.hip): one launch per reference call and direction instead of ~45 tiny torch kernels.
The one-off template analysis in __init__ is host code (numpy), as it is in the reference.
"""
import math
import os
import types

import numpy as np
import torch

from . import _lib
from ._lib import lib, ptr, stream

_launch = _lib.launch


# ------------------------------------------------------------------------------------------------ OBJ in / out
def load_obj(path):
    """-> vertices [V,3] f32, faces [F,3] i64, uvs [T,2] f32, face_textures [F,3] i64 (triangles only)"""
    vs, vts, fs, fts = [], [], [], []
    with open(path) as f:
        for line in f:
            d = line.split()
            if not d:
                continue
            if d[0] == "v":
                vs.append([float(t) for t in d[1:4]])
            elif d[0] == "vt":
                vts.append([float(t) for t in d[1:3]])
            elif d[0] == "f":
                if len(d) != 4:
                    raise ValueError(f"{path}: only triangle faces are supported (got a face with {len(d) - 1} vertices)")
                c = [t.split("/") for t in d[1:]]
                fs.append([int(t[0]) for t in c])
                if len(c[0]) > 1 and c[0][1] != "":
                    fts.append([int(t[1]) for t in c])
    if len(fts) != len(fs):
        raise ValueError(f"{path}: every face needs texture coordinates (f v/vt ...)")
    return (np.asarray(vs, np.float32), np.asarray(fs, np.int64) - 1, np.asarray(vts, np.float32),
            np.asarray(fts, np.int64) - 1)


def write_uv_sphere_obj(path, segments=32, rings=16):
    """A UV sphere with the topology of the reference's templates (code/mesh_templates/uvsphere_{16,31}rings.obj:
    `segments` meridians, `rings` latitude bands, triangulated quads, triangle fans at the poles, a UV seam with
    duplicated texture coordinates) generated procedurally -- used by the tests, the bench and anyone without the asset.
    Vertices: north pole, (rings-1) x segments ring vertices, south pole = 482 for 32 x 16."""
    v = [(0.0, 1.0, 0.0)]
    for r in range(1, rings):
        th = math.pi * r / rings
        for s in range(segments):
            ph = 2.0 * math.pi * s / segments
            v.append((math.sin(th) * math.sin(ph), math.cos(th), -math.sin(th) * math.cos(ph)))
    v.append((0.0, -1.0, 0.0))
    south = len(v) - 1

    def ring(r, s):
        return 1 + (r - 1) * segments + (s % segments)

    vt, faces = [], []

    def uv(s, r):  # texture vertex on the (segments+1) x (rings+1) lattice (the seam column segments is distinct from 0)
        vt.append((s / segments, 1.0 - r / rings))
        return len(vt)

    for s in range(segments):
        # north fan: pole texture vertex sits in the middle of the segment, as Blender exports it
        faces.append(((0 + 1, uv(s + 0.5, 0)), (ring(1, s + 1) + 1, uv(s + 1, 1)), (ring(1, s) + 1, uv(s, 1))))
        for r in range(1, rings - 1):
            a, b, c, d = ring(r, s), ring(r, s + 1), ring(r + 1, s + 1), ring(r + 1, s)
            ta, tb, tc, td = uv(s, r), uv(s + 1, r), uv(s + 1, r + 1), uv(s, r + 1)
            faces.append(((a + 1, ta), (b + 1, tb), (c + 1, tc)))
            faces.append(((a + 1, ta), (c + 1, tc), (d + 1, td)))
        faces.append(((south + 1, uv(s + 0.5, rings)), (ring(rings - 1, s) + 1, uv(s, rings - 1)),
                      (ring(rings - 1, s + 1) + 1, uv(s + 1, rings - 1))))
    with open(path, "w") as f:
        f.write("# procedural UV sphere (2dimageto3dmodel_amd.mesh.write_uv_sphere_obj)\no Sphere\n")
        for p in v:
            f.write("v %.6f %.6f %.6f\n" % p)
        for t in vt:
            f.write("vt %.6f %.6f\n" % t)
        for fc in faces:
            f.write("f " + " ".join("%d/%d" % p for p in fc) + "\n")
    return path


def face_adjacency(faces):
    """[F,3] int64: for every face the faces across its three edges (kaolin `mesh.ff`, monkey_patches.py:107-121;
    the order within a row differs from kaolin's descending sort -- loss_flat sums over the row)"""
    faces = np.asarray(faces)
    F = faces.shape[0]
    edge_faces = {}
    for f in range(F):
        for i in range(3):
            a, b = int(faces[f, i]), int(faces[f, (i + 1) % 3])
            edge_faces.setdefault((min(a, b), max(a, b)), []).append(f)
    ff = np.full((F, 3), -1, np.int64)
    for f in range(F):
        for i in range(3):
            a, b = int(faces[f, i]), int(faces[f, (i + 1) % 3])
            other = [g for g in edge_faces[(min(a, b), max(a, b))] if g != f]
            if len(other) != 1:
                raise ValueError("face_adjacency: the mesh must be a closed manifold (every edge shared by two faces)")
            ff[f, i] = other[0]
    return ff


# ------------------------------------------------------------------------------------------------ autograd bindings
def _f32c(t, name):
    if not t.is_cuda:
        raise _lib.M355Error(f"{name} must be a CUDA(HIP) tensor; the mesh path has no CPU implementation")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


# ---- static gather tables of the backward kernels (csrc/mesh_deform.hip: every gradient element is written by one thread that
#      sums its contributions in table order -- deterministic, no atomics).  int32 CSR, built once per template on the host.
def _csr(keys, n, *cols):
    """rows sorted by key (stable: ties keep the given order) -> ptr [n+1] int32 and the permuted columns"""
    keys = np.asarray(keys, np.int64)
    order = np.argsort(keys, kind="stable")
    ptr_ = np.zeros(n + 1, np.int32)
    ptr_[1:] = np.cumsum(np.bincount(keys, minlength=n))
    return (ptr_,) + tuple(np.ascontiguousarray(np.asarray(c)[order]) for c in cols)


def texel_table(uv, src, H, W, symmetric):
    """per texel of the [H,W] displacement map: the (vertex, weight) pairs of the bilinear taps landing on it.  Restates
    taps() / src_col() of csrc/mesh_deform.hip in numpy fp32, operation for operation (that file is compiled with
    -ffp-contract=off), so the weights are the forward's bits.  uv [S,2] fp32, src [V] -> ptr [H*W+1], vtx [E], w [E]"""
    f32 = np.float32
    uv = np.asarray(uv, f32)
    src = np.asarray(src, np.int64)
    V = src.shape[0]
    Wp = W + (2 if symmetric else 1)
    u, v = uv[src, 0], uv[src, 1]
    fx = (u + f32(1.0)) * f32(0.5) * f32(Wp - 1)
    fy = (v + f32(1.0)) * f32(0.5) * f32(H - 1)
    x0, y0 = np.floor(fx), np.floor(fy)
    wx1, wy1 = (fx - x0).astype(f32), (fy - y0).astype(f32)
    wx = (f32(1.0) - wx1, wx1)
    wy = (f32(1.0) - wy1, wy1)
    x0i, y0i = x0.astype(np.int64), y0.astype(np.int64)
    tex, vtx, wgt = [], [], []
    vid = np.arange(V)
    for iy in range(2):
        for ix in range(2):
            xp, yc = x0i + ix, y0i + iy
            if symmetric:   # circpad(texture, 1): padded column 0 = stored W-1, W+1 = stored 0
                xc = np.where(xp == 0, W - 1, np.where(xp == W + 1, 0, xp - 1))
            else:           # cat(texture, texture[..., :1])
                xc = np.where(xp == W, 0, xp)
            ok = (xp >= 0) & (xp < Wp) & (yc >= 0) & (yc < H)
            tex.append((yc * W + xc)[ok])
            vtx.append(vid[ok])
            wgt.append((wx[ix] * wy[iy]).astype(f32)[ok])
    tex, vtx, wgt = np.concatenate(tex), np.concatenate(vtx), np.concatenate(wgt)
    order = np.lexsort((np.arange(tex.shape[0]), vtx))   # vertices ascending inside a texel, taps in (iy, ix) order
    return _csr(tex[order], H * W, vtx[order].astype(np.int32), wgt[order].astype(f32))


def vertex_corner_table(faces, V):
    """per vertex: 4 * face + corner of its incident face corners, ascending -> ptr [V+1], fc [3F]"""
    faces = np.asarray(faces, np.int64)
    F = faces.shape[0]
    fc = (4 * np.arange(F)[:, None] + np.arange(3)[None, :]).reshape(-1)
    return _csr(faces.reshape(-1), V, fc.astype(np.int32))


def reverse_adjacency(ff):
    """per face f: the faces g that list f in ff[g] (one entry per occurrence), ascending -> ptr [F+1], idx [3F]"""
    ff = np.asarray(ff, np.int64)
    F = ff.shape[0]
    g = np.repeat(np.arange(F), 3)
    return _csr(ff.reshape(-1), F, g.astype(np.int32))


def _dev_tables(arrs, device):
    return tuple(torch.from_numpy(a).to(device) for a in arrs)


class _Vertices(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dmap, uv, tgm, base, src, xsign, symmetric, tex_table):
        dmap = _f32c(dmap.detach(), "displacement_map")
        B, C, H, W = dmap.shape
        assert C == 3, dmap.shape
        V = base.shape[0]
        pos = torch.empty((B, V, 3), dtype=torch.float32, device=dmap.device)
        _launch("mesh_vertices_fwd", ptr(dmap), ptr(uv), ptr(tgm), ptr(base), ptr(src), ptr(xsign), ptr(pos), B, V, H, W,
                int(symmetric), stream())
        ctx.save_for_backward(tgm, src, xsign, *tex_table)
        ctx.cfg = (B, V, H, W)
        return pos

    @staticmethod
    def backward(ctx, dpos):
        tgm, src, xsign, tptr, tvtx, tw = ctx.saved_tensors
        B, V, H, W = ctx.cfg
        dpos = _f32c(dpos, "grad")
        ddmap = torch.empty((B, 3, H, W), dtype=torch.float32, device=dpos.device)
        _launch("mesh_vertices_bwd", ptr(dpos), ptr(tgm), ptr(src), ptr(xsign), ptr(tptr), ptr(tvtx), ptr(tw), ptr(ddmap), B, V, H, W,
                stream())
        return ddmap, None, None, None, None, None, None, None


class _Normals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, faces, vtx_table):
        pos = _f32c(pos.detach(), "vertex_positions")
        B, V, _ = pos.shape
        F = faces.shape[0]
        nrm = torch.empty((B, F, 3), dtype=torch.float32, device=pos.device)
        _launch("mesh_normals_fwd", ptr(pos), ptr(faces), ptr(nrm), B, V, F, stream())
        ctx.save_for_backward(pos, faces, *vtx_table)
        return nrm

    @staticmethod
    def backward(ctx, dnrm):
        pos, faces, vptr, vfc = ctx.saved_tensors
        B, V, _ = pos.shape
        dpos = torch.empty_like(pos)
        _launch("mesh_normals_bwd", ptr(pos), ptr(faces), ptr(_f32c(dnrm, "grad")), ptr(vptr), ptr(vfc), ptr(dpos), B, V,
                faces.shape[0], stream())
        return dpos, None, None


class _Flat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nrm, ff, rev_table):
        nrm = _f32c(nrm.detach(), "norms")
        B, F, _ = nrm.shape
        loss = torch.empty((1,), dtype=torch.float32, device=nrm.device)
        ws = torch.empty((B,), dtype=torch.float32, device=nrm.device)
        _launch("mesh_flat_fwd", ptr(nrm), ptr(ff), ptr(loss), ptr(ws), B, F, stream())
        ctx.save_for_backward(nrm, ff, *rev_table)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        nrm, ff, rptr, ridx = ctx.saved_tensors
        B, F, _ = nrm.shape
        dn = torch.empty_like(nrm)
        _launch("mesh_flat_bwd", ptr(nrm), ptr(ff), ptr(rptr), ptr(ridx), ptr(_f32c(gl.reshape(1), "grad")), ptr(dn), B, F, stream())
        return dn, None, None


def grid_sample_bilinear(input, grid):
    """rendering/utils.py:6-13: F.grid_sample(mode='bilinear', align_corners=True) (the torch >= 1.3 branch)"""
    return torch.nn.functional.grid_sample(input, grid, mode='bilinear', align_corners=True)


def qrot(q, v):
    """rendering/utils.py:36-47: rotate vectors v [B,V,3] by (un-normalised-as-given) quaternions q [B,4]:
    v + 2 (w (u x v) + u x (u x v)), u = q[:, 1:].  B x <=962 vertices per call (main.py:285,880): plain tensor ops."""
    if q.shape[-1] != 4 or v.shape[-1] != 3:
        raise ValueError(f"qrot: expected q [...,4] and v [...,3], got {tuple(q.shape)} / {tuple(v.shape)}")
    u = q[:, 1:].unsqueeze(1).expand(-1, v.shape[1], -1)
    uv = torch.cross(u, v, dim=2)
    uuv = torch.cross(u, uv, dim=2)
    return v + 2 * (q[:, :1].unsqueeze(1) * uv + uuv)


def qmul(q, r):
    """rendering/utils.py:49-64: Hamilton product q * r of [..., 4] quaternions (run_reconstruction.py:196-202)"""
    if q.shape[-1] != 4 or r.shape[-1] != 4:
        raise ValueError("qmul: the last dimension must be 4")
    shape = q.shape
    t = torch.bmm(r.reshape(-1, 4, 1), q.reshape(-1, 1, 4))      # t[i, j] = r_i q_j
    w = t[:, 0, 0] - t[:, 1, 1] - t[:, 2, 2] - t[:, 3, 3]
    x = t[:, 0, 1] + t[:, 1, 0] - t[:, 2, 3] + t[:, 3, 2]
    y = t[:, 0, 2] + t[:, 1, 3] + t[:, 2, 0] - t[:, 3, 1]
    z = t[:, 0, 3] - t[:, 1, 2] + t[:, 2, 1] + t[:, 3, 0]
    return torch.stack((w, x, y, z), dim=1).view(shape)


def loss_flat(mesh, norms):
    """utils/losses.py:5-17 -- smoothness regulariser: neighbouring faces should have similar normals.
    `mesh` is `MeshTemplate.mesh` (needs `.ff` [F,3]); norms [B,F,3]."""
    ff32 = getattr(mesh, "_ff32", None)
    if ff32 is None or ff32.device != norms.device:
        ff32 = mesh._ff32 = mesh.ff.to(device=norms.device, dtype=torch.int32).contiguous()
        mesh._ff_rev = _dev_tables(reverse_adjacency(mesh.ff.cpu().numpy()), norms.device)   # (one-off: the backward's gather table)
    if norms.shape[1] != ff32.shape[0]:
        raise ValueError(f"loss_flat: {norms.shape[1]} face normals for a mesh of {ff32.shape[0]} faces")
    return _Flat.apply(norms, ff32, mesh._ff_rev)


# ------------------------------------------------------------------------------------------------ the template
class MeshTemplate:
    """rendering/mesh_template.py:12-149.  `mesh` carries vertices [V,3], faces [F,3], uvs [T,2], face_textures [F,3]
    and ff [F,3] as device tensors (the fields the reference reads from kaolin's TriangleMesh)."""

    def __init__(self, mesh_path, is_symmetric=True, device="cuda"):
        v, f, uvs, ft = load_obj(mesh_path)
        V = v.shape[0]
        poles = [int(v[:, 1].argmax()), int(v[:, 1].argmin())]            # north, south (:21)
        # reflection information (:24-49): pair every x<0 vertex with its mirror image
        axis = 0
        neg = np.nonzero(v[:, axis] < -1e-4)[0]
        zero = np.nonzero(np.abs(v[:, axis]) < 1e-4)[0]
        pos = []
        for idx in neg:
            opp = v[idx].copy()
            opp[axis] *= -1
            dist = np.linalg.norm(v - opp, axis=-1)
            j = int(dist.argmin())
            if dist[j] >= 1e-4:
                raise ValueError(f"{mesh_path}: vertex {idx} has no mirror image (the template must be x-symmetric)")
            pos.append(j)
        pos = np.asarray(pos, np.int64)
        if len(set(pos.tolist())) != len(pos) or len(pos) + len(neg) + len(zero) != V:
            raise ValueError(f"{mesh_path}: inconsistent symmetry pairing")
        nonneg = np.concatenate([pos, zero])
        # topology map: average UV of every vertex over its incident face corners, seam wrapped (:51-75)
        segments, rings = 32, (31 if "31rings" in mesh_path else 16)
        acc = [[] for _ in range(V)]
        for faces_t, faces_v in zip(ft, f):
            for t, vert in zip(faces_t, faces_v):
                res = uvs[t].astype(np.float64) * [segments, rings]
                if math.isclose(res[0], segments, abs_tol=1e-4):
                    res[0] = 0  # wrap around
                acc[int(vert)].append(res)
        topo = np.zeros((V, 2), np.float32)
        for i, data in enumerate(acc):
            if data:
                topo[i] = (np.mean(np.array(data, dtype=np.float32), axis=0) / [segments, rings]).astype(np.float32)
        topo = (topo * 2 - 1) * np.array([1, -1], np.float32)
        sym_mask = np.ones((1, V, 3), np.float32)
        sym_mask[:, zero, 0] = 0                                              # (:78-79)
        # tangent map (:82-93): rows normal, tangent, bitangent; poles have no (bi)tangent
        vt = torch.from_numpy(v)
        normals = torch.nn.functional.normalize(vt, dim=1)
        up = torch.tensor([[0.0, 1.0, 0.0]]).expand_as(normals)
        tang = torch.nn.functional.normalize(torch.cross(normals, up, dim=1), dim=1)
        bitang = torch.cross(normals, tang, dim=1)
        for p in poles:
            tang[p] = 0
            bitang[p] = 0
        tangent_map = torch.stack((normals, tang, bitang), dim=1)

        dev = torch.device(device)
        self.mesh = types.SimpleNamespace(
            vertices=vt.to(dev), faces=torch.from_numpy(f).to(dev), uvs=torch.from_numpy(uvs).to(dev),
            face_textures=torch.from_numpy(ft).to(dev), ff=torch.from_numpy(face_adjacency(f)).to(dev))
        self.topo_map = torch.from_numpy(topo).to(dev)
        self.nonneg_indices = torch.from_numpy(nonneg).to(dev)
        self.neg_indices = torch.from_numpy(neg.astype(np.int64)).to(dev)
        self.pos_indices = torch.from_numpy(pos).to(dev)
        self.nonneg_topo_map = self.topo_map[self.nonneg_indices]
        self.symmetry_mask = torch.from_numpy(sym_mask).to(dev)
        self.tangent_map = tangent_map.to(dev)
        self.nonneg_tangent_map = self.tangent_map[self.nonneg_indices]
        self.is_symmetric = is_symmetric

        # per-vertex gather tables of the kernels: source row in the (nonneg_)topo / tangent maps and the x factor
        if is_symmetric:
            where = {int(g): k for k, g in enumerate(nonneg)}
            src = np.empty(V, np.int32)
            xs = np.ones(V, np.float32)
            for k, g in enumerate(nonneg):
                src[g] = k
            for n_, p_ in zip(neg, pos):
                src[n_] = where[int(p_)]
                xs[n_] = -1.0                                                  # vtx_n[:, pos] * [-1, 1, 1] (:145)
            xs[zero] = 0.0                                                     # symmetry_mask (:146)
        else:
            src, xs = np.arange(V, dtype=np.int32), np.ones(V, np.float32)
        self._src = torch.from_numpy(src).to(dev)
        self._xsign = torch.from_numpy(xs).to(dev)
        self._faces32 = self.mesh.faces.to(torch.int32).contiguous()
        self._tgm = (self.nonneg_tangent_map if is_symmetric else self.tangent_map).contiguous()
        self._uv_cache = {}
        self._tex_cache = {}
        self._vtx_table = _dev_tables(vertex_corner_table(f, V), dev)

    # -- the three per-step methods (HIP) -------------------------------------------------------------------------
    def _uv(self, W):
        """grid_sample coordinates in the padded map (:130-137)"""
        uv = self._uv_cache.get(W)
        if uv is None:
            topo = (self.nonneg_topo_map if self.is_symmetric else self.topo_map).clone()
            if self.is_symmetric:  # compensate for the even symmetry of the UV map: x axis only
                delta = 1 / (2 * W)
                expansion = (W + 1) / W
                topo[:, 0] = (topo[:, 0] + 1 + 2 * delta - expansion) / expansion
            uv = self._uv_cache[W] = topo.contiguous()
        return uv

    def _tex_table(self, H, W):
        """the backward's gather table for an [H,W] displacement map (one-off per map size)"""
        t = self._tex_cache.get((H, W))
        if t is None:
            t = self._tex_cache[(H, W)] = _dev_tables(
                texel_table(self._uv(W).cpu().numpy(), self._src.cpu().numpy(), H, W, self.is_symmetric), self._src.device)
        return t

    def get_vertex_positions(self, displacement_map):
        """UV displacement map [B,3,H,W] -> vertex positions in object space [B,V,3] (:125-149)"""
        H, W = displacement_map.shape[2], displacement_map.shape[3]
        return _Vertices.apply(displacement_map, self._uv(W), self._tgm, self.mesh.vertices,
                               self._src, self._xsign, self.is_symmetric, self._tex_table(H, W))

    def compute_normals(self, vertex_positions):
        """face normals of the FINAL vertex positions [B,V,3] -> [B,F,3] (:113-123)"""
        return _Normals.apply(vertex_positions, self._faces32, self._vtx_table)

    def deform(self, deltas):
        """template deformation along the tangent map (:106-111); torch (the fused path is get_vertex_positions)"""
        tgm = self.nonneg_tangent_map if self.is_symmetric else self.tangent_map
        return (deltas.unsqueeze(-2) @ tgm.expand(deltas.shape[0], -1, -1, -1)).squeeze(-2)

    def adjust_uv_and_texture(self, texture, return_texture=True):
        """UVs of the template and the texture prepared for its boundary conditions (:151-172)"""
        if self.is_symmetric:
            delta = 1 / (2 * texture.shape[3])
            expansion = (texture.shape[3] + 1) / texture.shape[3]
            uvs = self.mesh.uvs.clone()
            uvs[:, 0] = (uvs[:, 0] + delta) / expansion
            uvs = uvs.expand(texture.shape[0], -1, -1)
            texture = torch.cat((texture[:, :, :, -1:], texture, texture[:, :, :, :1]), dim=3)  # circpad(texture, 1)
        else:
            uvs = self.mesh.uvs.expand(texture.shape[0], -1, -1)
            texture = torch.cat((texture, texture[:, :, :, :1]), dim=3)
        return uvs, texture

    def forward_renderer(self, renderer, vertex_positions, texture, num_gpus=1, **kwargs):
        """rendering/mesh_template.py:172-186: render the deformed template with its texture -> (image, alpha).
        num_gpus > 1 was nn.DataParallel's replica bookkeeping (the face tables were tiled per replica); one process
        per GPU here, so it is accepted and ignored."""
        input_uvs, input_texture = self.adjust_uv_and_texture(texture)
        image, alpha, _ = renderer(points=[vertex_positions, self.mesh.faces], uv_bxpx2=input_uvs,
                                   texture_bx3xthxtw=input_texture, ft_fx3=self.mesh.face_textures, **kwargs)
        return image, alpha

    def export_obj(self, path_prefix, vertex_positions, texture=None):
        """OBJ (+MTL) export of one deformed mesh (:188-219); the texture image is written only if imageio is there"""
        assert len(vertex_positions.shape) == 2
        name = os.path.basename(path_prefix)
        with open(path_prefix + ".obj", "w") as file:
            print("mtllib " + name + ".mtl", file=file)
            for v in vertex_positions.tolist():
                print("v {:.5f} {:.5f} {:.5f}".format(*v), file=file)
            for uv in self.mesh.uvs.tolist():
                print("vt {:.5f} {:.5f}".format(*uv), file=file)
            print("usemtl " + name, file=file)
            for f, ft in zip(self.mesh.faces.tolist(), self.mesh.face_textures.tolist()):
                print("f {}/{} {}/{} {}/{}".format(f[0] + 1, ft[0] + 1, f[1] + 1, ft[1] + 1, f[2] + 1, ft[2] + 1), file=file)
        with open(path_prefix + ".mtl", "w") as file:
            for line in ("newmtl " + name, "Ka 1.000 1.000 1.000", "Kd 1.000 1.000 1.000", "Ks 0.000 0.000 0.000", "d 1.0",
                         "illum 1", "map_Ka " + name + ".png", "map_Kd " + name + ".png"):
                print(line, file=file)
        if texture is not None:
            try:
                import imageio
            except ImportError:
                return
            img = (texture.permute(1, 2, 0) * 255).clamp(0, 255).cpu().byte().numpy()
            imageio.imwrite(path_prefix + ".png", img)
