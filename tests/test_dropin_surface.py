"""The Python surface of SURVEY.md 8b, imported the way the reference's callers import it (CWD = code/, here
`2dimageto3dmodel_amd/dropin` first on sys.path): every listed name resolves, with the reference's argument names.
Runs in a subprocess because the shim packages are called `utils`, `models`, ... (they would shadow test helpers)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "2dimageto3dmodel_amd", "dropin")

SURFACE = {   # module -> {name: [leading argument names of the callable (methods: after self)] or None for plain objects}
    "utils.effective_loss_function": {"EffectiveLossFunction": ["voxel_size", "kernel_size", "smooth_sigma"]},
    "camera.coordinate_system_transformation": {"CameraUtilities": []},
    "utils.trilinear_interpolation": {"TrilinearInterpolation": ["epsilon", "size"]},
    "utils.smooth_voxels": {"VoxelsSmooth": []},
    "quaternions.points_quaternions": {"PointsQuaternionsRotator": [], "PointsQuaternionsConverter": []},
    "quaternions.operations": {"QuaternionOperations": []},
    "models.supervised_part": {"SupervisedLoss": []},
    "models.unsupervised_part": {"UnsupervisedLoss": ["number_of_pose_predictor_candidates", "student_weight"]},
    "models.gan": {"Generator": ["args", "emb_dim", "symmetric", "mesh_head"], "MultiScaleDiscriminator": ["args", "nc"],
                   "TextureDiscriminator": ["args", "nc", "downsample", "circular", "positional_embeddings"],
                   "MeshDiscriminator": ["args", "nc", "circular", "positional_embeddings"],
                   "ResBlockUp": ["args", "ch_in", "ch_out", "emb_dim", "pad_fn"],
                   "ConditionalBatchNorm2d": ["args", "ch", "emb_dim"], "SpatialAttention": ["input_dim", "context_dim"],
                   "positional_encoding": ["Ny", "Nx"]},
    "models.reconstruction": {"ReconstructionNetwork": None},
    "utils.losses": {"GANLoss": ["gan_mode", "target_real_label", "target_fake_label", "tensor", "opt"],
                     "loss_flat": ["mesh", "norms"]},
    "rendering.utils": {"grid_sample_bilinear": ["input", "grid"], "symmetrize_texture": ["x"], "adjust_poles": ["tex"],
                        "circpad": ["x", "amount"], "qrot": ["q", "v"], "qmul": ["q", "r"]},
    "rendering.mesh_template": {"MeshTemplate": ["mesh_path", "is_symmetric"]},
    "rendering.renderer": {"Renderer": ["height", "width", "filtering"], "ortho_projection": ["points_bxpx3", "faces_fx3"]},
    "rendering.fragment_shader": {"fragmentshader": ["imtexcoord_bxhxwx2", "texture_bx3xthxtw", "improb_bxhxwx1", "filtering",
                                                     "background_image"],
                                  "texinterpolation": ["imtexcoord_bxhxwx2", "texture_bx3xthxtw", "filtering"]},
    "sync_batchnorm": {"SynchronizedBatchNorm2d": None, "DataParallelWithCallback": None},
}
METHODS = {   # class -> {method: argument names after self}
    ("utils.effective_loss_function", "EffectiveLossFunction"): {"forward": ["point_cloud", "rotation", "scale"],
                                                                 "termination_probs": ["voxels", "epsilon"]},
    ("camera.coordinate_system_transformation", "CameraUtilities"): {
        "transformation_3d_coord_to_camera_coord": ["point_cloud", "rotation", "field_of_view", "camera_view_distance"]},
    ("utils.trilinear_interpolation", "TrilinearInterpolation"): {"trilinear_interpolation": ["point_cloud"]},
    ("utils.smooth_voxels", "VoxelsSmooth"): {"separate_kernels": ["std_dev", "kernel_size"], "smooth": ["voxels", "kernels", "scale"]},
    ("quaternions.points_quaternions", "PointsQuaternionsRotator"): {"rotate_points": ["xyz_triplet", "q", "inverse_rotation_direction"]},
    ("quaternions.operations", "QuaternionOperations"): {"quaternion_addition": ["q1", "q2"], "quaternion_subtraction": ["q1", "q2"],
                                                         "quaternion_multiplication": ["q1", "q2"], "quaternion_square": ["q"],
                                                         "quaternion_conjugate": ["q"]},
    ("models.supervised_part", "SupervisedLoss"): {"forward": ["projection", "masks"]},
    ("models.unsupervised_part", "UnsupervisedLoss"): {"forward": ["predictions", "masks", "training"]},
    ("models.gan", "Generator"): {"forward": ["z", "c", "caption", "return_attention"]},
    ("rendering.renderer", "Renderer"): {"forward": ["points", "uv_bxpx2", "texture_bx3xthxtw", "ft_fx3", "background_image",
                                                     "return_hardmask"]},
    ("rendering.mesh_template", "MeshTemplate"): {"forward_renderer": ["renderer", "vertex_positions", "texture", "num_gpus"]},
    ("models.gan", "MultiScaleDiscriminator"): {"forward": ["x", "mesh_map", "c", "caption"]},
}

_PROBE = r'''
import inspect, importlib, json, sys
sys.path.insert(0, sys.argv[1])
surface, methods = json.loads(sys.argv[2]), json.loads(sys.argv[3])
bad = []
for mod, names in surface.items():
    try:
        m = importlib.import_module(mod)
    except Exception as e:
        bad.append(f"import {mod}: {e!r}")
        continue
    for name, want in names.items():
        obj = getattr(m, name, None)
        if obj is None:
            bad.append(f"{mod}.{name} missing")
            continue
        if want is None:
            continue
        params = [p for p in inspect.signature(obj).parameters]
        if params[:len(want)] != want:
            bad.append(f"{mod}.{name}{params} != {want}")
for key, ms in methods.items():
    mod, cls = key.split("|")
    c = getattr(importlib.import_module(mod), cls)
    for meth, want in ms.items():
        f = getattr(c, meth, None)
        if f is None:
            bad.append(f"{mod}.{cls}.{meth} missing")
            continue
        params = [p for p in inspect.signature(f).parameters if p != "self"]
        if params[:len(want)] != want:
            bad.append(f"{mod}.{cls}.{meth}{params} != {want}")
print(json.dumps(bad))
'''


def test_every_8b_name_imports_from_dropin():
    methods = {f"{m}|{c}": v for (m, c), v in METHODS.items()}
    r = subprocess.run([sys.executable, "-c", _PROBE, DROPIN, json.dumps(SURFACE), json.dumps(methods)], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    bad = json.loads(r.stdout.strip().splitlines()[-1])
    assert bad == [], "\n".join(bad)


def test_quaternion_helpers_match_reference():
    """quaternions/operations.py and rendering/utils.py:{qrot,qmul,grid_sample_bilinear} executed by the reference
    (oracle/gen_golden_p8.py) -- tensor helpers, CPU"""
    import importlib
    P = importlib.import_module("2dimageto3dmodel_amd.projection")
    M = importlib.import_module("2dimageto3dmodel_amd.mesh")
    g = load_golden("p1_rotate")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    qo = P.QuaternionOperations()
    assert np.array_equal(qo.quaternion_addition(a, b).numpy(), g["add"])
    assert np.array_equal(qo.quaternion_subtraction(a, b).numpy(), g["sub"])
    assert np.array_equal(qo.quaternion_multiplication(a, b).numpy(), g["mul"])
    assert np.array_equal(qo.quaternion_conjugate(a).numpy(), g["conj"])
    # q^2 = q (x) q: the reference's quaternion_square raises (math.pow on tensors); its formula must equal the product
    assert np.allclose(qo.quaternion_square(a).numpy(), qo.quaternion_multiplication(a, a).numpy(), atol=1e-6)
    assert np.allclose(M.qrot(a, torch.from_numpy(g["v"])).numpy(), g["qrot"], atol=1e-6)
    assert np.allclose(M.qmul(a, b).numpy(), g["qmul"], atol=1e-6)
    assert np.allclose(M.grid_sample_bilinear(torch.from_numpy(g["img"]), torch.from_numpy(g["grid"])).numpy(), g["gsb"], atol=1e-6)
    with pytest.raises(ValueError):
        M.qrot(a[:, :3], torch.from_numpy(g["v"]))


@pytest.mark.gpu
@pytest.mark.parametrize("inverse", [False, True])
def test_rotate_points_bit_exact_and_gradients(inverse):
    """PointsQuaternionsRotator.rotate_points on the HIP kernel (m355_quat_rotate_fwd/_bwd) vs the reference's own output
    and autograd gradients: forward bit-exact (same Hamilton-product order, no FMA contraction)"""
    import importlib
    P = importlib.import_module("2dimageto3dmodel_amd.projection")
    g = load_golden("p1_rotate")
    xyz = torch.from_numpy(g["xyz"]).cuda().requires_grad_()
    q = torch.from_numpy(g["q"]).cuda().requires_grad_()
    out = P.PointsQuaternionsRotator.rotate_points(xyz, q, inverse)
    k = int(inverse)
    assert np.array_equal(out.detach().cpu().numpy().view(np.uint32), g[f"out{k}"].view(np.uint32))
    (out * torch.from_numpy(g["w"]).cuda()).sum().backward()
    assert np.abs(xyz.grad.cpu().numpy() - g[f"dxyz{k}"]).max() < 1e-5 * np.abs(g[f"dxyz{k}"]).max()
    assert np.abs(q.grad.cpu().numpy() - g[f"dq{k}"]).max() < 1e-4 * np.abs(g[f"dq{k}"]).max()
    # empty cloud
    e = P.PointsQuaternionsRotator.rotate_points(torch.zeros(2, 0, 3, device="cuda"), torch.ones(2, 4, device="cuda"), inverse)
    assert tuple(e.shape) == (2, 0, 3)


def test_cross_module_attribute_references_resolve(pkg):
    """every `alias.name` in the package where `alias` is one of the package's own modules names something that exists --
    the CPU suite cannot execute the GPU code paths, so a helper deleted from one module while another still calls it
    would otherwise only surface on the GPU box"""
    import ast
    import glob
    import importlib
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2dimageto3dmodel_amd")
    missing = []
    for path in sorted(glob.glob(os.path.join(root, "*.py"))):
        tree = ast.parse(open(path).read())
        alias = {}
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.level == 1 and node.module is None:
                for a in node.names:   # from . import conv as C
                    alias[a.asname or a.name] = a.name
        mods = {}
        for k, v in alias.items():
            try:
                mods[k] = importlib.import_module("2dimageto3dmodel_amd." + v)
            except ImportError:
                pass
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in mods:
                if not hasattr(mods[node.value.id], node.attr):
                    missing.append("%s:%d %s.%s" % (os.path.basename(path), node.lineno, node.value.id, node.attr))
    assert not missing, missing
