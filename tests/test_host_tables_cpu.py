"""CPU: the host-side tables and planning queries added in round 4 (no GPU, no kernel launches).

* the static gather tables of the mesh backward (mesh.texel_table / vertex_corner_table / reverse_adjacency) against brute force
  on the procedural UV sphere -- the backward kernels sum in table order, so a wrong table is a wrong gradient;
* the workspace / plan queries of the C-ABI: deterministic wgrad, partial-sum wgrad, split-K forward."""
import ctypes
import importlib

import numpy as np


def _sphere(tmp_path):
    m = importlib.import_module("2dimageto3dmodel_amd.mesh")
    v, f, uvs, ft = m.load_obj(m.write_uv_sphere_obj(str(tmp_path / "uvsphere_16rings.obj")))
    return m, v, f


def test_vertex_corner_and_reverse_adjacency_tables(tmp_path):
    m, v, f = _sphere(tmp_path)
    V, F = v.shape[0], f.shape[0]
    ptr, fc = m.vertex_corner_table(f, V)
    assert ptr[0] == 0 and ptr[-1] == 3 * F and fc.dtype == np.int32 and ptr.dtype == np.int32
    for vv in (0, 1, 17, 240, V - 1):
        want = sorted(4 * i + c for i in range(F) for c in range(3) if f[i, c] == vv)
        assert fc[ptr[vv]:ptr[vv + 1]].tolist() == want          # ascending: the summation order of the kernel
    ff = m.face_adjacency(f)
    rp, ri = m.reverse_adjacency(ff)
    assert rp[-1] == 3 * F
    for q in (0, 31, 500, F - 1):
        assert ri[rp[q]:rp[q + 1]].tolist() == sorted(g for g in range(F) for i in range(3) if ff[g, i] == q)
        assert sorted(ri[rp[q]:rp[q + 1]].tolist()) == sorted(ff[q].tolist())   # closed manifold: the adjacency is symmetric
    # an asymmetric table (not a manifold adjacency) is reversed literally
    odd = np.array([[1, 1, 2], [0, 2, 2], [0, 0, 0]])
    p2, i2 = m.reverse_adjacency(odd)
    assert [i2[p2[k]:p2[k + 1]].tolist() for k in range(3)] == [[1, 2, 2, 2], [0, 0], [0, 1, 1]]


def test_texel_table_restates_the_forward_taps():
    """every (vertex, tap) of the bilinear sampling appears exactly once under its texel, with the forward's fp32 weight; pad columns
    fold onto their source column (circular by one when symmetric, one wrapped column otherwise)"""
    m = importlib.import_module("2dimageto3dmodel_amd.mesh")
    rs = np.random.RandomState(3)
    for symmetric, H, W in ((True, 32, 16), (False, 16, 16)):
        S, V = 300, 482
        uv = rs.uniform(-1.02, 1.02, (S, 2)).astype(np.float32)      # a few taps fall outside the padded map: dropped
        src = rs.randint(0, S, V)
        ptr, vtx, w = m.texel_table(uv, src, H, W, symmetric)
        assert ptr.shape == (H * W + 1,) and ptr[-1] == vtx.shape[0] == w.shape[0] and w.dtype == np.float32
        Wp = W + (2 if symmetric else 1)
        got = {}
        for t in range(H * W):
            seg = vtx[ptr[t]:ptr[t + 1]]
            assert (np.diff(seg) >= 0).all()                        # vertices ascending inside a texel
            for e in range(ptr[t], ptr[t + 1]):
                got.setdefault((t, int(vtx[e])), []).append(float(w[e]))
        want = {}
        f32 = np.float32
        for v in range(V):
            u_, v_ = uv[src[v]]
            fx = (u_ + f32(1.0)) * f32(0.5) * f32(Wp - 1)
            fy = (v_ + f32(1.0)) * f32(0.5) * f32(H - 1)
            x0, y0 = np.floor(fx), np.floor(fy)
            wx1, wy1 = f32(fx - x0), f32(fy - y0)
            for iy, wy in ((0, f32(1.0) - wy1), (1, wy1)):
                for ix, wx in ((0, f32(1.0) - wx1), (1, wx1)):
                    xp, yc = int(x0) + ix, int(y0) + iy
                    if not (0 <= xp < Wp and 0 <= yc < H):
                        continue
                    xc = (W - 1 if xp == 0 else (0 if xp == W + 1 else xp - 1)) if symmetric else (0 if xp == W else xp)
                    want.setdefault((yc * W + xc, v), []).append(float(f32(wx) * f32(wy)))
        assert got.keys() == want.keys()
        for k in want:
            assert sorted(got[k]) == sorted(want[k]), k


def test_workspace_queries(pkg):
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    L = pkg._lib.lib()

    def desc(N, H, W, Cin, Cout, k, s=1, mode=1, ups=0):
        return conv.make_desc(N, H, W, Cin, Cout, k, k, s, k // 2 if s == 1 else 1, k // 2 if s == 1 else 1, mode, ups)

    d = desc(64, 8, 4, 512, 512, 3)
    # deterministic wgrad: [flag | Cout*K fixed-point sums | Cout bias sums], 8 bytes each
    assert L.m355_conv2d_wgrad_det_ws_bytes(ctypes.byref(d)) == 8 * (1 + 3 * (512 * 9 * 512 + 512))   # flag + three integers per cell
    # partial-sum wgrad: the 8-input-channel 5x5 layer only (one row of Cout * 201 floats per pixel-axis workgroup)
    c8 = desc(128, 256, 256, 8, 64, 5, mode=2)
    assert L.m355_conv2d_wgrad_ws_bytes(ctypes.byref(c8)) == 4 * 256 * 64 * 201
    assert L.m355_conv2d_wgrad_ws_bytes(ctypes.byref(d)) == 0
    assert L.m355_conv2d_wgrad_ws_bytes(ctypes.byref(desc(64, 256, 128, 64, 3, 5))) == 0        # heads keep their atomics
    # split-K forward: fewer 128 x 128 tiles than CUs, 64-channel K steps, whole 128-channel output tiles
    M = 64 * 8 * 4
    assert L.m355_conv2d_fwd_ws_bytes(ctypes.byref(d)) == 4 * 4 * M * 512                         # 16 x 4 tiles -> 4 K slices of 18 steps
    assert L.m355_conv2d_fwd_ws_stats_rows(ctypes.byref(d)) == 512
    assert L.m355_conv2d_fwd_ws_bytes(ctypes.byref(desc(64, 32, 16, 256, 256, 3))) == 0           # 512 tiles: no split
    assert L.m355_conv2d_fwd_ws_bytes(ctypes.byref(desc(64, 8, 4, 512, 64, 3))) == 0              # 64 output channels: 256 x 64 tiles
    assert L.m355_conv2d_fwd_ws_bytes(ctypes.byref(desc(64, 16, 32, 128, 128, 3))) == 0           # the halo kernel's layer
    # upsample + 3x3 layers (round 5): the weight buffers carry the sub-pixel views behind the 3x3 ones whatever the image size ...
    up = desc(64, 128, 64, 128, 64, 3, ups=1)      # G.blk6.conv1: runs in the sub-pixel form
    small = desc(64, 32, 16, 256, 128, 3, ups=1)   # G.blk4.conv1: stored width 16 -> the 9-tap kernels
    for dd, cin, cout in ((up, 128, 64), (small, 256, 128)):
        rows_f, rows_d = (64 if cout <= 64 else 128), (64 if cin <= 64 else (cin + 127) // 128 * 128)
        assert L.m355_conv2d_weight_elems(ctypes.byref(dd), 0) == rows_f * 9 * cin + 4 * rows_f * 4 * cin
        assert L.m355_conv2d_weight_elems(ctypes.byref(dd), 1) == rows_d * 9 * cout + rows_d * 16 * cout
    assert L.m355_conv2d_weight_elems(ctypes.byref(desc(64, 128, 64, 128, 64, 3)), 0) == 64 * 9 * 128
    # ... the 16-entry effective weight gradient (+ bias sums) only where the shape takes that form; its dgrad needs no frame
    # (round 6: as per-workgroup partial rows -- 64 replicas of the class launch here, each [Cout*16*Cin | 4 x Cout] cells -- added
    # in row order by the 16 -> 9 fold; the plan says the sum is ordered, so a binding takes it in deterministic mode too)
    assert L.m355_conv2d_wgrad_ws_bytes(ctypes.byref(up)) == 4 * 64 * (64 * 16 * 128 + 4 * 64)
    pl = pkg._lib.ConvPlan()
    assert L.m355_conv2d_plan(ctypes.byref(up), ctypes.byref(pl)) == 0 and pl.wgrad_ws_ordered == 1
    assert L.m355_conv2d_plan(ctypes.byref(small), ctypes.byref(pl)) == 0 and pl.wgrad_ws_ordered == 0
    assert L.m355_conv2d_wgrad_det_ws_bytes(ctypes.byref(up)) == 8 * (1 + 3 * (64 * 16 * 128 + 64))
    assert L.m355_conv2d_wgrad_ws_bytes(ctypes.byref(small)) == 0
    assert L.m355_conv2d_wgrad_det_ws_bytes(ctypes.byref(small)) == 8 * (1 + 3 * (128 * 9 * 256 + 128))
    assert L.m355_conv2d_dgrad_ws_bytes(ctypes.byref(up)) == 0
    assert L.m355_conv2d_fwd_stats_rows(ctypes.byref(up)) == 4 * 128     # class pairs: 128 workgroups x 2 row parities, a row block per class
    assert L.m355_conv2d_fwd_stats_rows(ctypes.byref(desc(64, 64, 32, 128, 128, 3, ups=1))) == 4 * 64
    assert L.m355_cproj_bwd_ws_floats(128, 256, 512) == 128 * 2 * 512 and L.m355_cproj_bwd_ws_floats(128, 64, 256) == 0
    # the fused discriminator tail (round 5): eligible shapes, workspace = the [25][C] bf16 weight table + the shares of demb
    assert L.m355_cproj_bwd_conv5_ok(32, 32, 512) == 1 and L.m355_cproj_bwd_conv5_ok(8, 8, 256) == 1
    assert L.m355_cproj_bwd_conv5_ok(32, 30, 512) == 0 and L.m355_cproj_bwd_conv5_ok(128, 128, 512) == 0   # W % 4, LDS budget
    assert L.m355_cproj_bwd_conv5_ws_floats(128, 32, 32, 512) == 25 * 512 // 2 + 128 * 16 * 512            # 2048 / 128 = 16 workgroups per sample
    assert L.m355_cproj_bwd_conv5_ws_floats(128, 8, 8, 256) == 25 * 256 // 2                                # one workgroup per sample: demb direct
    assert L.m355_sn_scratch_words(18, 512, 4608) == 18 * (72 + 128)


def test_masked_input_off_the_gpu():
    """gan_ops.MaskedInput (the not-yet-concatenated discriminator input the trainer hands over): on CPU tensors -- or any layout
    the loaders do not take -- it is not fusable, reports the shape of the tensor it stands for, and materialize() is the
    reference's cat(fake * alpha, alpha) [; cat(real, alpha)] (main.py:493, 503-507) with autograd intact"""
    import torch
    G = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
    g = torch.Generator().manual_seed(3)
    fake = torch.randn(2, 3, 16, 16, generator=g, requires_grad=True)
    real = torch.randn(2, 3, 16, 16, generator=g)
    alpha = (torch.rand(2, 1, 16, 16, generator=g) > 0.5).float()
    for r in (None, real):
        mi = G.MaskedInput(fake, alpha, r)
        assert not mi.fusable() and not G.disc_inputs_ok(mi, None, [(1, False, None, 8, 8)])
        want = torch.cat((fake * alpha, alpha), dim=1)
        if r is not None:
            want = torch.cat((want, torch.cat((r, alpha), dim=1)), dim=0)
        x = mi.materialize()
        assert tuple(mi.shape) == tuple(want.shape) and mi.device == fake.device and torch.equal(x, want)
        fake.grad = None
        x.sum().backward()
        assert torch.equal(fake.grad, alpha.expand(-1, 3, -1, -1))
    # a broadcast alpha is not the loaders' layout either
    assert not G.MaskedInput(fake, torch.ones(2, 1, 1, 1), None).fusable()


def test_two_gib_layers_are_halved_on_the_host():
    """conv._halves: a layer whose input or output reaches 2 GiB runs as two half-batch launches (the specialised kernels address
    bytes with 32 bits); pure geometry, no GPU"""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    mk = lambda n, h=256, w=256, ci=8, co=64, k=5, s=1, p=2: conv.make_desc(n, h, w, ci, co, k, k, s, p, p, 2, 0)
    assert conv._halves(mk(128)) is None                      # D.conv1 at the benchmarked D step: 1 GiB out
    d, h = conv._halves(mk(256))                              # 2 GiB out
    assert (d.N, h) == (128, 128) and (d.H, d.W, d.Cin, d.Cout, d.kh, d.stride, d.pad_w_mode) == (256, 256, 8, 64, 5, 1, 2)
    assert conv._halves(d) is None
    d4, _ = conv._halves(mk(512))                             # 4 GiB: halves of halves
    assert d4.N == 256 and conv._halves(d4)[0].N == 128
    assert conv._halves(mk(257)) is None                      # an odd batch cannot be halved: the generic kernels take it
    assert conv._halves(mk(256, ci=64, co=128, k=4, s=2, p=1)) is not None     # D.conv2: 2 GiB IN
    assert conv._halves(mk(254, ci=64, co=128, k=4, s=2, p=1)) is None
    # the eligibility questions of a halved layer are its halves' (no GPU needed: they are table lookups in the library)
    big, half = mk(256, ci=64, co=128, k=4, s=2, p=1), mk(128, ci=64, co=128, k=4, s=2, p=1)
    assert conv.dgrad_mask_ok(big) == conv.dgrad_mask_ok(half) and conv.wgrad_fuses_dbias(big) == conv.wgrad_fuses_dbias(half)
    assert conv.conv_stats_rows(mk(512, 256, 128, 64, 64, 3, 1, 1)) == 2 * conv.conv_stats_rows(mk(256, 256, 128, 64, 64, 3, 1, 1))


def test_conv_plan_equals_the_individual_queries(pkg):
    """m355_conv2d_plan (round 5): one call per layer answers what the individual pre-launch queries answer -- for every shape class
    of the benchmarked networks, in the build that is loaded"""
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    L = pkg._lib.lib()
    shapes = [(64, 128, 64, 128, 64, 3, 1, 1, 1, 1, 1), (64, 32, 16, 256, 128, 3, 1, 1, 1, 1, 1), (64, 8, 4, 512, 512, 3, 1, 1, 1, 1, 0),
              (128, 256, 256, 8, 64, 5, 1, 2, 2, 2, 0), (128, 256, 256, 64, 128, 4, 2, 1, 1, 2, 0), (64, 256, 128, 64, 3, 5, 1, 2, 2, 1, 0),
              (64, 128, 64, 128, 64, 1, 1, 0, 0, 0, 0), (128, 32, 32, 512, 1, 5, 1, 2, 2, 2, 0), (64, 256, 128, 64, 64, 3, 1, 1, 1, 1, 0)]
    for N, H, W, Cin, Cout, k, s, ph, pw, mode, ups in shapes:
        d = conv.make_desc(N, H, W, Cin, Cout, k, k, s, ph, pw, mode, ups)
        p = conv.plan(d)
        ho, wo = ctypes.c_int(), ctypes.c_int()
        assert L.m355_conv2d_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == 0 and (p.Ho, p.Wo) == (ho.value, wo.value)
        assert p.dy_channels == L.m355_conv2d_dy_channels(Cout) and p.act_bytes == L.m355_act_bytes()
        assert (p.w_fwd_elems, p.w_dgrad_elems) == (L.m355_conv2d_weight_elems(ctypes.byref(d), 0), L.m355_conv2d_weight_elems(ctypes.byref(d), 1))
        assert (p.fwd_bits_ok, p.dgrad_bits_ok) == (L.m355_conv2d_maskbits_ok(ctypes.byref(d), 0), L.m355_conv2d_maskbits_ok(ctypes.byref(d), 1))
        assert p.dgrad_mask_ok == L.m355_conv2d_dgrad_mask_ok(ctypes.byref(d)) and p.fwd_stats_rows == L.m355_conv2d_fwd_stats_rows(ctypes.byref(d))
        assert (p.fwd_ws_bytes, p.fwd_ws_stats_rows) == (L.m355_conv2d_fwd_ws_bytes(ctypes.byref(d)), L.m355_conv2d_fwd_ws_stats_rows(ctypes.byref(d)))
        assert p.wgrad_fuses_dbias == L.m355_conv2d_wgrad_fuses_dbias(ctypes.byref(d))
        assert (p.dgrad_ws_bytes, p.wgrad_ws_bytes, p.wgrad_det_ws_bytes) == (L.m355_conv2d_dgrad_ws_bytes(ctypes.byref(d)),
                                                                              L.m355_conv2d_wgrad_ws_bytes(ctypes.byref(d)),
                                                                              L.m355_conv2d_wgrad_det_ws_bytes(ctypes.byref(d)))
        assert p.exec_ratio == L.m355_conv2d_exec_ratio(ctypes.byref(d)) and p.exec_ratio == (4.0 / 9.0 if (ups and W % 32 == 0) else 1.0)
    assert conv.plan(d) is p   # memoised per descriptor
