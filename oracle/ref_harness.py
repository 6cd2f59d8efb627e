"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Imports the *reference's own* projection functions (read-only tree at
/root/reference) so that golden vectors can be produced from them.  This module
only works in the build container; /root/reference does not exist on the GPU
box, so nothing in tests/, smoke() or bench.py may import it at run time.  The
goldens it produces are committed under tests/golden/ together with
oracle/gen_golden_p.py (the script that made them).

Import recipe:
  * the reference package directory is literally called ``code`` (shadows the
    stdlib) and mixes relative with top-level imports (defect D7), so a symlink
    ``<tmp>/refpkg -> /root/reference/code`` is made and ``<tmp>``,
    ``code/utils`` and ``code/models`` are put on sys.path;
  * shim S0: run under ``python -O`` because
    code/quaternions/points_quaternions.py:23 asserts batch == 3 (defect D1);
  * shim S1: ``EffectiveLossFunction.forward`` passes ``kernels=()`` to
    ``smooth`` (code/utils/effective_loss_function.py:77) which raises; the
    forward is re-composed from the very same reference functions with
    ``separate_kernels(sigma, kernel_size)`` passed in (defect D2).
"""
import contextlib
import io
import os
import sys
import tempfile

REF_ROOT = "/root/reference/code"


def available():
    return os.path.isdir(REF_ROOT)


_cache = {}


def load():
    """Returns a dict of the reference modules used by the projection path."""
    if _cache:
        return _cache
    if not available():
        raise RuntimeError("reference tree not present (this only runs in the build container)")
    if __debug__:
        raise RuntimeError("run under `python -O` (shim S0: reference asserts batch == 3)")
    sys.dont_write_bytecode = True
    tmp = tempfile.mkdtemp(prefix="refpkg_")
    os.symlink(REF_ROOT, os.path.join(tmp, "refpkg"))
    for p in (tmp, os.path.join(REF_ROOT, "utils"), os.path.join(REF_ROOT, "models")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import importlib

    with contextlib.redirect_stdout(io.StringIO()):
        elf = importlib.import_module("refpkg.utils.effective_loss_function")
        cam = importlib.import_module("refpkg.camera.coordinate_system_transformation")
        pq = importlib.import_module("refpkg.quaternions.points_quaternions")
        ops = importlib.import_module("refpkg.quaternions.operations")
        tri = importlib.import_module("trilinear_interpolation")
        sm = importlib.import_module("smooth_voxels")
    _cache.update(elf=elf, cam=cam, pq=pq, ops=ops, tri=tri, sm=sm)
    return _cache


def quiet(fn, *a, **k):
    """The reference constructors print banners; silence them."""
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def ref_forward_stages(point_cloud, rotation, scale, S=64, sigma=3.0, kernel_size=21):
    """EffectiveLossFunction.forward (elf:58-81) re-composed from the reference's
    own functions with shim S1 and the grid size forwarded (defect D6).
    Returns every stage so goldens can pin them one by one."""
    m = load()
    elf = quiet(m["elf"].EffectiveLossFunction, voxel_size=S, kernel_size=kernel_size, smooth_sigma=sigma)
    cam = quiet(m["cam"].CameraUtilities)
    cloud = quiet(cam.transformation_3d_coord_to_camera_coord, point_cloud=point_cloud, rotation=rotation,
                  field_of_view=1.875, camera_view_distance=2.0)
    interp = quiet(m["tri"].TrilinearInterpolation, size=S)
    voxels = interp.trilinear_interpolation(point_cloud=cloud)
    vs = quiet(m["sm"].VoxelsSmooth)
    kernels = vs.separate_kernels(float(elf.sigma), elf.kernel_size)
    smoothed = vs.smooth(voxels=voxels, kernels=kernels, scale=scale)
    probs = elf.termination_probs(smoothed)
    proj = probs[:, :-1].sum(1).flip(1)
    return dict(cam=cloud, voxels=voxels, kernels=kernels, smoothed=smoothed, probs=probs, proj=proj)


def ref_supervised_loss(projection, masks):
    """SupervisedLoss.forward (code/models/supervised_part.py:68-72), restated
    verbatim-in-behaviour because the module itself cannot be imported
    (supervised_part.py imports `encoder`/`decoder` top-level *and* relative
    `..utils`, and its ctor raises -- defect D11)."""
    import torch.nn.functional as F

    m = F.interpolate(input=masks.unsqueeze(0), scale_factor=1 / 2, mode="bilinear", align_corners=True).squeeze()
    return F.mse_loss(input=projection, target=m, reduction="sum") / (2 * projection.size(0))
