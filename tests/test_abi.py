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
    for m in re.finditer(r"\b(?:int|size_t|long long|const char \*)\s*\*?\s*(m355_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
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
    assert L.m355_abi_version() == 1
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
