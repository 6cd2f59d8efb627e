"""CPU: the C-ABI library loads, exports every symbol include/m355.h declares, and the ctypes table in
2dimageto3dmodel_amd/_lib.py mirrors the header (no compute calls here -- there is no GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "m355.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|double|long long|const char \*)\s*\*?\s*(m355_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        out[m.group(1)] = len(args)
    return out


def test_library_exports_every_declared_symbol(pkg):
    path = pkg._lib.LIB_PATH
    assert os.path.exists(path), "run python 2dimageto3dmodel_amd/build.py"
    L = ctypes.CDLL(path)
    decl = header_functions()
    assert len(decl) >= 8
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/m355.h but not exported"


def test_ctypes_table_mirrors_header(pkg):
    decl = header_functions()
    sig = pkg._lib.SIGNATURES
    assert set(sig) == set(decl), set(sig) ^ set(decl)
    for name, nargs in decl.items():
        assert len(sig[name][1]) == nargs, name


def test_bad_arguments_report_errors_not_crashes(pkg):
    L = pkg._lib.lib()
    assert L.m355_abi_version() == 4 and L.m355_act_bytes() == 2
    rc = L.m355_proj_transform_fwd(None, None, None, 1, 1, 1.875, 2.0, None)
    assert rc == -1 and b"null" in L.m355_last_error()
    assert L.m355_proj_ntiles(128) == 256 and L.m355_proj_ntiles(64) == 64 and L.m355_proj_ntiles(512) == 16384
    assert L.m355_proj_ntiles(100000) == -2
    rc = L.m355_proj_render_fwd(1, 1, None, 1, 20, 1, 1, 1, 64, 0, None)  # even tap count
    assert rc == -1 and b"odd" in L.m355_last_error()


def test_hot_path_refuses_cpu_tensors(pkg):
    import pytest
    import torch

    elf = pkg.EffectiveLossFunction()
    with pytest.raises(pkg._lib.M355Error):
        elf(torch.zeros(1, 4, 3), torch.ones(1, 4))


def test_dropin_paths_resolve_to_the_hip_classes(pkg):
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, r'%s/2dimageto3dmodel_amd/dropin');"
            "from utils.effective_loss_function import EffectiveLossFunction as E;"
            "from models.supervised_part import SupervisedLoss;"
            "from models.unsupervised_part import UnsupervisedLoss;"
            "from camera.coordinate_system_transformation import CameraUtilities;"
            "print(E.__module__)") % ROOT
    out = subprocess.check_output([sys.executable, "-c", code]).decode()
    assert "2dimageto3dmodel_amd.projection" in out


def test_host_side_launch_rules(pkg):
    """the launch-shaping decisions that run on the host, without a GPU: the Chamfer slicing rule (workspace > 0 <=> the sweep is
    split: fewer than 2048 query blocks and at least two 1024-target tiles) and which layers the LDS-transpose weight-view kernel
    takes (rows and channels of every view multiples of 32, no K padding, at most 16 taps)"""
    import importlib
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    L = pkg._lib.lib()
    for B, N, M in ((1, 16384, 16384), (8, 16384, 16384), (32, 16384, 16384), (64, 2048, 2048), (1, 1000, 1500), (3, 300, 3000)):
        want = B * ((N + 255) // 256) < 2048 and M >= 2048
        nb = L.m355_chamfer_nn_ws_bytes(B, N, M)   # one row of B * N 8-byte keys per target slice (2 .. 16 slices)
        assert (nb > 0) == want and nb % (B * N * 8) == 0 and nb // (B * N * 8) in ((0,) if not want else (2, 4, 8, 16))
    esz = L.m355_weight_prep_entry_bytes()
    fake = ctypes.c_void_p(0x1000)   # (device pointers are only recorded here, nothing is launched)

    def tiles(cin, cout, k, stride):
        d = conv.make_desc(1, 64, 64, cin, cout, k, k, stride, k // 2, k // 2, 0, 0)
        entry = (ctypes.c_char * esz)()
        n = L.m355_weight_prep_fill_entry(ctypes.byref(d), fake, cin, None, fake, fake, entry)
        assert n > 0
        return L.m355_weight_prep_entry_tiles(entry)
    assert tiles(64, 128, 3, 1) == (128 // 32) * (64 // 32)        # G 3x3: forward rows 128, dgrad rows 64
    assert tiles(512, 512, 3, 1) == 16 * 16
    assert tiles(128, 256, 4, 2) == (256 // 32) * (128 // 32)      # D 4x4 stride 2: forward + four dgrad classes, 16 taps
    assert tiles(128, 64, 1, 1) == (64 // 32) * (128 // 32)        # 1x1 shortcut
    assert tiles(8, 64, 5, 1) == 0                                 # D.conv1: 25 taps, 8 channels -> the gather kernel
    assert tiles(64, 3, 5, 1) == 0                                 # a head
