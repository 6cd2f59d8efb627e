"""The composed mesh-estimation training step (2dimageto3dmodel_amd/recon_train.py = the loop body of
code/run_reconstruction.py:409-445) against tests/golden/recon_step.npz (oracle/gen_golden_recon_step.py: the reference's
network / template / pose / loss code executed on CPU in fp32; the rasteriser stage is oracle/raster_ref.py -- UNPINNED, Kaolin
is not available)."""
import argparse
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "recon_step.npz")
sys.path.insert(0, ROOT)
from oracle.gen_golden_recon_step import init_side_params, make_inputs  # noqa: E402  (seeded inputs: same streams as the golden)


def _template(tmp_path, device):
    mesh = importlib.import_module("2dimageto3dmodel_amd.mesh")
    return mesh.MeshTemplate(mesh.write_uv_sphere_obj(str(tmp_path / "uvsphere_16rings.obj")), is_symmetric=True, device=device)


def test_transform_vertices_and_iou_match_golden(pkg):
    """transform_vertices (run_reconstruction.py:237-252: scale + delta, qrot, translate + delta, flip y / z, z0 perspective) with
    DatasetParams (mirrored indices included) on the golden's object-space vertices -> the golden's camera-space vertices;
    mean_iou (:225-231) of the golden's rendered silhouette against the input mask"""
    rt = importlib.import_module("2dimageto3dmodel_amd.recon_train")
    g = np.load(GOLDEN)
    X, gt_scale, gt_translation, gt_rot, gt_idx = make_inputs(int(g["seed"]), int(g["B"]))
    dp = rt.DatasetParams(argparse.Namespace(optimize_deltas=True, optimize_z0=True), int(g["n_data"]))

    class _Net:   # (init_side_params also re-draws the mesh head: same stream position as in the generator script)
        conv_mesh = torch.nn.Conv2d(64, 3, 5)
    init_side_params(_Net, dp, int(g["seed"]))
    vtx = rt.transform_vertices(torch.from_numpy(g["raw_vtx"]), gt_scale, gt_translation, gt_rot, gt_idx, dp, True, True)
    assert (vtx - torch.from_numpy(g["vtx"])).abs().max().item() < 2e-6
    # without the learnable offsets / perspective: the plain pose transform
    plain = rt.transform_vertices(torch.from_numpy(g["raw_vtx"]), gt_scale, gt_translation, gt_rot, None, None, False, False)
    mesh = importlib.import_module("2dimageto3dmodel_amd.mesh")
    want = (mesh.qrot(gt_rot, gt_scale.unsqueeze(-1) * torch.from_numpy(g["raw_vtx"])) + gt_translation.unsqueeze(1)) * torch.tensor([1.0, -1.0, -1.0])
    assert torch.equal(plain, want)
    alpha = torch.from_numpy(g["x_fake_s2"].astype(np.float32))[:, 3]
    iou = rt.mean_iou(alpha, X[:, 3, ::2, ::2])
    assert abs(float(iou) - float(g["miou"])) < 2e-2     # (stride-2 sampling of both silhouettes)


def step_metrics(tmp_path):
    """run one iteration's forward + backward on the GPU and compare every stage with the golden -> dict of measured agreements"""
    rt = importlib.import_module("2dimageto3dmodel_amd.recon_train")
    g = np.load(GOLDEN)
    seed, B = int(g["seed"]), int(g["B"])
    tpl = _template(tmp_path, "cuda")
    torch.manual_seed(seed)
    tr = rt.ReconTrainer(tpl, dataset_size=int(g["n_data"]), texture_resolution=int(g["texture_res"]), image_resolution=int(g["res"]),
                         optimize_deltas=True, optimize_z0=True, device="cuda")
    init_side_params(tr.generator, tr.dataset_params, seed)
    tr.train()
    X, gt_scale, gt_translation, gt_rot, gt_idx = (t.cuda() for t in make_inputs(seed, B))
    m = {}
    X_fake, raw_vtx, pred_tex, mesh_map = tr.render(X, gt_scale, gt_translation, gt_rot, gt_idx)
    rel = lambda a, b: float((a - b).abs().mean() / b.abs().max())
    m["pred_tex_mean_err"] = rel(pred_tex.detach().cpu(), torch.from_numpy(g["pred_tex"].astype(np.float32)))
    m["mesh_map_mean_err"] = rel(mesh_map.detach().cpu(), torch.from_numpy(g["mesh_map"]))
    m["raw_vtx_max_err"] = float((raw_vtx.detach().cpu() - torch.from_numpy(g["raw_vtx"])).abs().max())
    want = torch.from_numpy(g["x_fake_s2"].astype(np.float32))
    got = X_fake.detach().cpu()[:, :, ::2, ::2]
    m["image_mean_err"] = float((got[:, :3] - want[:, :3]).abs().mean())
    m["alpha_mean_err"] = float((got[:, 3] - want[:, 3]).abs().mean())
    a, b = got[:, 3] > 0.5, want[:, 3] > 0.5
    m["silhouette_iou_vs_golden"] = float(((a & b).float().sum() / (a | b).float().sum()))
    tr.optimizer.zero_grad(); tr.optimizer_dataset.zero_grad()
    total, recon, flat, miou, _ = tr.losses(X, gt_scale, gt_translation, gt_rot, gt_idx)
    m["recon_loss_rel"], m["flat_rel"] = abs(float(recon.detach()) / float(g["recon_loss"]) - 1), abs(float(flat.detach()) / float(g["flat_loss"]) - 1)
    m["total_rel"], m["miou_abs"] = abs(float(total.detach()) / float(g["loss"]) - 1), abs(float(miou) - float(g["miou"]))
    cos = lambda x, y: float(torch.dot(x.flatten().double(), y.flatten().double()) / (x.norm().double() * y.norm().double() + 1e-300))
    named = dict(tr.generator.named_parameters())
    wn = dict(zip([str(k) for k in g["grad_keys"]], g["grad_norms"]))
    # (a) END TO END: the silhouette term's gradient lives on edge pixels and moves with sub-pixel vertex shifts (the bf16 network
    # places a few vertices up to 0.03 = 4 pixels away), so the mesh branch / encoder gradients are compared by norm here and
    # elementwise in (c); texture decoder and dataset parameters are smooth in those shifts
    total.backward()
    for k in ("blk5_tex.conv2.weight", "conv_tex.weight"):
        m["e2e cos " + k] = cos(named[k].grad.detach().cpu(), torch.from_numpy(g["grad:" + k].astype(np.float32)))
    for k in ("ds_translation", "ds_scale", "ds_z0"):
        m["e2e cos " + k] = cos(getattr(tr.dataset_params, k).grad.detach().cpu(), torch.from_numpy(g["g_" + k]))
    r = np.array([abs(float(p.grad.norm()) / wn[k] - 1) for k, p in named.items() if wn[k] > 1e-8])
    m["e2e grad_norm_rel_median"], m["e2e grad_norm_rel_max"] = float(np.median(r)), float(r.max())
    # (b) the RENDERER stage alone, forward and backward, on the golden's own vertices and texture (no network noise):
    # MeshTemplate.forward_renderer + MSE against the oracle renderer's image and its d loss / d (vertices, texture)
    v_g = torch.from_numpy(g["vtx"]).cuda().requires_grad_()
    t_g = torch.from_numpy(g["pred_tex"].astype(np.float32)).cuda().requires_grad_()
    img, alp = tpl.forward_renderer(tr.renderer, v_g, t_g)
    xf = torch.cat((img, alp), dim=3).permute(0, 3, 1, 2)
    got2 = xf.detach().cpu()[:, :, ::2, ::2]
    m["render image_mean_err"], m["render alpha_mean_err"] = float((got2[:, :3] - want[:, :3]).abs().mean()), float((got2[:, 3] - want[:, 3]).abs().mean())
    a2 = got2[:, 3] > 0.5
    m["render silhouette_iou"] = float(((a2 & b).float().sum() / (a2 | b).float().sum()))
    tr.criterion(xf, X).backward()
    m["render recon_loss_rel"] = abs(float(tr.criterion(xf, X).detach()) / float(g["recon_loss"]) - 1)
    m["render cos d_vtx"] = cos(v_g.grad.cpu(), torch.from_numpy(g["d_vtx"]))
    m["render cos d_tex"] = cos(t_g.grad.cpu(), torch.from_numpy(g["d_tex"]))
    m["render d_vtx_norm_rel"] = abs(float(v_g.grad.norm()) / float(np.linalg.norm(g["d_vtx"])) - 1)
    # (c) UPSTREAM of the renderer, in two well-conditioned stages.  The displacement-map gradient of this step is 99.99 % the
    # flat-loss term, and that term is chaotic in the map on a rough random-weight mesh: in the REFERENCE's own fp32 code a 1 %
    # perturbation of the map (what the bf16 network produces) leaves its gradient at cosine 0.53 (0.3 %: 0.77; 0.01 %: 0.9995).
    # So an end-to-end elementwise comparison of the mesh branch / encoder gradients measures that conditioning (0.51-0.61), not
    # the kernels.  (c1) pose transform + template deformation + normals + flat loss, forward and backward, AT THE GOLDEN'S MAP:
    mesh = importlib.import_module("2dimageto3dmodel_amd.mesh")
    tr.optimizer_dataset.zero_grad()
    mm_g = torch.from_numpy(g["mesh_map"]).cuda().requires_grad_()
    raw_g = tpl.get_vertex_positions(mm_g)
    vtx_g = rt.transform_vertices(raw_g, gt_scale, gt_translation, gt_rot, gt_idx, tr.dataset_params, True, True)
    flat_g = mesh.loss_flat(tpl.mesh, tpl.compute_normals(raw_g))
    m["c1 vtx_max_err"] = float((vtx_g.detach().cpu() - torch.from_numpy(g["vtx"])).abs().max())
    m["c1 flat_rel"] = abs(float(flat_g.detach()) / float(g["flat_loss"]) - 1)
    torch.autograd.backward([vtx_g, float(g["flat_coeff"]) * flat_g], [torch.from_numpy(g["d_vtx"]).cuda(), torch.ones((), device="cuda")])
    m["c1 cos d_mesh_map"] = cos(mm_g.grad.cpu(), torch.from_numpy(g["d_mesh_map"]))
    m["c1 d_mesh_map_rel_l2"] = float((mm_g.grad.cpu() - torch.from_numpy(g["d_mesh_map"])).norm() / np.linalg.norm(g["d_mesh_map"]))
    for k in ("ds_translation", "ds_scale", "ds_z0"):
        m["c1 cos " + k] = cos(getattr(tr.dataset_params, k).grad.detach().cpu(), torch.from_numpy(g["g_" + k]))
    # (c2) the network's backward driven by the golden's gradients of its two outputs: elementwise against the golden's parameter
    # gradients, plus two gradients inside the mesh branch (NHWC here, NCHW in the golden; the block input sits below the folded
    # x2 upsample here: its gradient is the 2x2 block sum of the golden's)
    tr.optimizer.zero_grad()
    cap = {}

    def _pre(mod, inp):   # (as the golden's hooks: a clone of the block input isolates the mesh branch's share of its gradient)
        x = inp[0].clone()
        x.register_hook(lambda gr: cap.__setitem__("d_in", gr.detach().float()))
        return (x,) + tuple(inp[1:])
    h1 = tr.generator.blk4_mesh.register_forward_pre_hook(_pre)
    h2 = tr.generator.blk4_mesh.register_forward_hook(
        lambda mod, inp, out: out.register_hook(lambda gr: cap.__setitem__("d_out", gr.detach().float())) and None)
    pred_tex, mesh_map = tr.generator(X)
    h1.remove(); h2.remove()
    torch.autograd.backward([mesh_map, pred_tex], [torch.from_numpy(g["d_mesh_map"]).cuda(), torch.from_numpy(g["d_tex"]).cuda()])
    for k in [k for k in g.files if k.startswith("grad:")]:
        m["c2 cos " + k[5:]] = cos(named[k[5:]].grad.detach().cpu(), torch.from_numpy(g[k].astype(np.float32)))
    m["c2 cos d_blk4_mesh_out"] = cos(cap["d_out"].permute(0, 3, 1, 2).cpu(), torch.from_numpy(g["d_blk4_mesh_out"]))
    gin = torch.from_numpy(g["d_blk4_mesh_in"].astype(np.float32))
    gin = gin.view(gin.shape[0], gin.shape[1], gin.shape[2] // 2, 2, gin.shape[3] // 2, 2).sum((3, 5))
    m["c2 cos d_blk4_mesh_in"] = cos(cap["d_in"].permute(0, 3, 1, 2).cpu(), gin)
    r = np.array([abs(float(p.grad.norm()) / wn[k] - 1) for k, p in named.items() if wn[k] > 1e-8])
    m["c2 grad_norm_rel_median"], m["c2 grad_norm_rel_max"] = float(np.median(r)), float(r.max())
    # the optimiser steps and the warm-up of :439-440
    out = tr.iteration(X, gt_scale, gt_translation, gt_rot, gt_idx)
    m["warmup_after"] = tr.flat_warmup
    m["finite"] = all(bool(torch.isfinite(v).all()) for v in out.values())
    return m


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_recon_step_matches_golden(pkg, tmp_path):
    """bf16 MFMA network + HIP template / rasteriser kernels vs the fp32 CPU composition.  The rasteriser stage is compared with
    oracle/raster_ref.py only (UNPINNED).  Thresholds = measured values on MI355X (round 3, printed by scripts/recon_step_check.py)
    with margin; the silhouette moves by whole pixels where a vertex crosses a pixel centre, hence the IoU-style bounds."""
    m = step_metrics(tmp_path)
    assert m["finite"] and abs(m["warmup_after"] - 9.9) < 1e-9, m
    # forward, end to end (measured: texture 2.2 %, displacement map 0.9 % of max; vertices <= 0.03; silhouette IoU 0.988)
    assert m["pred_tex_mean_err"] < 0.05 and m["mesh_map_mean_err"] < 0.03 and m["raw_vtx_max_err"] < 0.06, m
    assert m["silhouette_iou_vs_golden"] > 0.97 and m["alpha_mean_err"] < 0.006 and m["image_mean_err"] < 0.02, m
    assert m["recon_loss_rel"] < 0.05 and m["flat_rel"] < 0.03 and m["total_rel"] < 0.03 and m["miou_abs"] < 0.01, m
    assert min(v for k, v in m.items() if k.startswith("e2e cos ")) > 0.95 and m["e2e grad_norm_rel_median"] < 0.25, m
    # the renderer stage on identical inputs (vs oracle/raster_ref.py -- unpinned)
    assert m["render silhouette_iou"] > 0.995 and m["render alpha_mean_err"] < 2e-3 and m["render recon_loss_rel"] < 5e-3, m
    assert m["render cos d_vtx"] > 0.98 and m["render cos d_tex"] > 0.99 and m["render d_vtx_norm_rel"] < 0.05, m
    # upstream of the renderer: (c1) pose + template + flat loss at the golden's map, (c2) the network's backward from the golden's
    # output gradients (thresholds set from the first measurement, see profiles/r03_recon_step_agreement.txt)
    assert m["c1 vtx_max_err"] < 1e-5 and m["c1 flat_rel"] < 1e-4 and m["c1 cos d_mesh_map"] > 0.9999 and m["c1 d_mesh_map_rel_l2"] < 1e-2, m
    assert min(v for k, v in m.items() if k.startswith("c1 cos ds_")) > 0.9999, m
    assert min(v for k, v in m.items() if k.startswith("c2 cos ")) > 0.75, m


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_recon_iterations_are_bit_reproducible_in_deterministic_mode(pkg, tmp_path):
    """the composed mesh-estimation step -- encoder / decoder convs, mesh template, rasteriser + shader, both Adam optimisers -- three
    iterations from one seed, twice, deterministic mode: every parameter and every reported scalar bit-identical (round 4: the
    rasteriser's and the shader's backward in integer cells; the weight gradients in fixed point; the mesh backward as gathers)"""
    rt = importlib.import_module("2dimageto3dmodel_amd.recon_train")
    g = np.load(GOLDEN)
    seed, B = int(g["seed"]), int(g["B"])
    inputs = [t.cuda() for t in make_inputs(seed, B)]

    def run():
        tpl = _template(tmp_path, "cuda")
        torch.manual_seed(seed)
        tr = rt.ReconTrainer(tpl, dataset_size=int(g["n_data"]), texture_resolution=int(g["texture_res"]), image_resolution=int(g["res"]),
                             optimize_deltas=True, optimize_z0=True, device="cuda")
        init_side_params(tr.generator, tr.dataset_params, seed)
        tr.train()
        scalars = []
        for _ in range(3):
            out = tr.iteration(*inputs)
            scalars += [float(v) for v in out.values() if torch.is_tensor(v) or isinstance(v, float)]
        torch.cuda.synchronize()
        state = {k: v.detach().clone() for k, v in list(tr.generator.state_dict().items()) + [("ds." + k, v) for k, v in tr.dataset_params.state_dict().items()]}
        return state, scalars

    prev = pkg.set_deterministic(True)
    try:
        (sa, la), (sb, lb) = run(), run()
    finally:
        pkg.set_deterministic(prev)
    assert la == lb and all(np.isfinite(v) for v in la), (la, lb)
    bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
    assert not bad, (len(bad), bad[:8])
