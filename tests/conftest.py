import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("2dimageto3dmodel_amd")


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: g[k] for k in g.files}


P_CASES = ["p_cfg1", "p_b3_noscale", "p_oob", "p_s128", "p_n1", "p_dense", "p_sigma"]


@pytest.fixture(autouse=True)
def _fresh_conv_plans():
    """conv.py memoises the library's per-layer answers (m355_conv2d_plan); several tests flip the library's environment
    switches (monkeypatch.setenv: tile / workgroup / variant overrides), which are process-wide settings the memo cannot see --
    every test starts, and leaves, with empty caches"""
    conv = sys.modules.get("2dimageto3dmodel_amd.conv")
    if conv is not None:
        conv._reset_caches()
    yield
    conv = sys.modules.get("2dimageto3dmodel_amd.conv")
    if conv is not None:
        conv._reset_caches()
