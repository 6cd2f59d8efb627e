"""CPU: the oracle (oracle/p_oracle.c) against golden vectors produced by executing the reference's own
code (oracle/gen_golden_p.py).  This is what pins the oracle; the GPU tests then compare HIP against both."""
import numpy as np
import pytest
from conftest import P_CASES, load_golden

from oracle import p_oracle as po


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("name", P_CASES)
def test_transform_and_bins_exact(name):
    g = load_golden(name)
    cam = po.transform(g["pc"], g["q"])
    # bit-exact camera coordinates (P1+P2)
    assert np.array_equal(cam.view(np.uint32), g["cam"].view(np.uint32))
    bn = po.bins(cam, int(g["S"]))
    assert np.array_equal(bn[..., 0].astype(bool), g["inb"])
    assert np.array_equal(bn[..., 1:][g["inb"]], g["floor"][g["inb"]])


@pytest.mark.parametrize("name", P_CASES)
def test_volume_bit_exact(name):
    g = load_golden(name)
    V = po.splat(g["cam"], int(g["S"])).reshape(-1)
    assert np.array_equal(np.flatnonzero(V), g["vox_idx"])
    assert np.array_equal(V[g["vox_idx"]].view(np.uint32), g["vox_val"].view(np.uint32))


@pytest.mark.parametrize("name", P_CASES)
def test_smooth_termination_projection_loss(name):
    g = load_golden(name)
    S = int(g["S"])
    sc = g.get("scale")
    taps = po.taps(float(g["sigma"]), 21, True)
    assert rel(taps, g["taps"]) < 5e-7
    V = po.splat(g["cam"], S)
    sm = po.smooth(V, g["taps"], 1, sc)
    ry, rx = g["ray_y"], g["ray_x"]
    assert np.abs(sm[:, :, ry, rx] - g["sm_rays"]).max() < 1e-6
    assert np.abs(sm.sum(1) - g["sm_raysum"]).max() < 1e-5
    T = po.termination(sm)
    assert np.abs(T[:, :, ry, rx] / g["probs_rays"] - 1).max() < 5e-6
    proj = po.project(T)
    assert np.abs(proj / g["proj"] - 1).max() < 2e-6
    loss = po.sup_loss(proj, g["mask"].astype(np.float32))
    assert abs(loss / float(g["loss"]) - 1) < 1e-6
    assert np.abs(po.sup_loss_bwd(proj, g["mask"].astype(np.float32)) - g["dproj"]).max() < 1e-6


@pytest.mark.parametrize("name", P_CASES)
def test_whole_forward_matches(name):
    g = load_golden(name)
    proj = po.forward(g["pc"], g["q"], g.get("scale"), int(g["S"]), g["taps"])
    assert np.abs(proj / g["proj"] - 1).max() < 2e-6


@pytest.mark.parametrize("name", P_CASES)
def test_gradients_match_reference_autograd(name):
    g = load_golden(name)
    dp, dq, dsc, _ = po.backward(g["pc"], g["q"], g.get("scale"), g["dproj"], int(g["S"]), g["taps"])
    assert rel(dp, g["dpc"]) < 5e-6
    assert rel(dq, g["dq"]) < 5e-6
    if "scale" in g:
        assert rel(dsc, g["dscale"]) < 5e-6


def test_axis_mask_chaining_is_separable():
    # chained x,y,depth smoothing (the "fixed" D5 option) equals applying the three axes in any order
    rs = np.random.RandomState(0)
    V = (rs.rand(1, 16, 16, 16) > 0.97).astype(np.float32)
    t = po.taps(3.0, 7, False)
    a = po.smooth(V, t, 7)
    b = po.smooth(po.smooth(po.smooth(V, t, 4), t, 2), t, 1)
    assert np.abs(a - b).max() < 1e-6


def test_chamfer_oracle_vs_bruteforce_numpy():
    rs = np.random.RandomState(1)
    a, b = rs.rand(2, 50, 3).astype(np.float32), rs.rand(2, 70, 3).astype(np.float32)
    d, i = po.chamfer_nn(a, b)
    ref = ((a[:, :, None, :].astype(np.float64) - b[:, None, :, :]) ** 2).sum(-1)
    assert np.array_equal(i, ref.argmin(-1))
    assert np.abs(d - ref.min(-1)).max() < 1e-6
