"""GPU: Frenet frame construction (fp_frames_build) and Cartesian->Frenet projection (fp_from_state) against the
reference goldens (G2 spline coefficients, G7 from_state) and the oracle on ragged batches."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["flens", "sinus"])
def test_frames_build_matches_reference(engine, name):
    g = load_golden("g2_spline.npz")
    pts = g[f"{name}_pts"]
    knots, coef = engine.build_frames(pts[None])
    np.testing.assert_allclose(knots[0], g[f"{name}_knots"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(coef[0], g[f"{name}_coef"], rtol=1e-9, atol=1e-11)


def test_frames_build_ragged_batch_vs_oracle(oracle, engine):
    rng = np.random.default_rng(5)
    F, NX = 9, 96
    n = rng.integers(3, NX + 1, F).astype(np.int32)
    n[0], n[1] = 2, NX
    pts = np.zeros((F, NX, 2))
    for f in range(F):
        x = np.cumsum(rng.uniform(0.5, 9.0, n[f]))
        pts[f, : n[f], 0] = x
        pts[f, : n[f], 1] = rng.uniform(0, 6) * np.sin(x / rng.uniform(20, 70)) + rng.uniform(-50, 50)
    knots, coef = engine.build_frames(pts, n)
    for f in range(F):
        k, cx, cy = oracle.spline2d_build(pts[f, : n[f], 0], pts[f, : n[f], 1])
        np.testing.assert_allclose(knots[f, : n[f]], k, rtol=0, atol=1e-10)
        assert np.isinf(knots[f, n[f]:]).all()
        np.testing.assert_allclose(coef[f, 0:4, : n[f]], cx, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(coef[f, 4:8, : n[f]], cy, rtol=1e-8, atol=1e-10)
        assert (coef[f, :, n[f]:] == 0).all()


def test_from_state_matches_reference(engine):
    g5 = load_golden("g5_closed_loop.npz")
    g7 = load_golden("g7_from_state.npz")
    knots, coef = engine.build_frames(g5["centerline"][None])
    n = np.array([len(g5["centerline"])], dtype=np.int32)
    B = len(g7["poses"])
    ego = engine.from_state(knots, coef, n, np.zeros(B, dtype=np.int32), g7["poses"])
    np.testing.assert_allclose(ego, g7["frenet"], rtol=0, atol=1e-8)
    # the planning problem's initial state (SURVEY 8d config 1 anchor)
    np.testing.assert_allclose(ego[0, [0, 3, 1, 4]], [51.5936, 0.3687, 14.6660, -0.1639], atol=5e-4)
