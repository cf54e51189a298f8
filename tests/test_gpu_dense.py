"""GPU parity: the HIP dense-lattice pass (through the C ABI) against the CPU oracle.

Bars (BASELINE.json north_star): cost within 1e-6, selected index exact, flag words exact.
"""
import numpy as np
import pytest

from fiss_plus_planner_amd import synth

pytestmark = pytest.mark.gpu

COST_TOL = 1e-6


def oracle_dense(O, batch, egos=None):
    probs = O.problems_from_batch(batch, egos)
    res = [p.fop_plan() for p in probs]
    return (np.array([r.best_idx for r in res]), np.array([r.best_cost for r in res]), np.stack([r.cost for r in res]),
            np.stack([r.flags for r in res]), np.stack([r.stats for r in res]))


def compare_dense(O, engine, batch):
    out = engine.plan_dense(batch)
    bi, bc, cost, flags, stats = oracle_dense(O, batch)
    np.testing.assert_allclose(out.cost, cost, rtol=0, atol=COST_TOL)
    assert np.abs(out.cost - cost).max() < 1e-9  # in practice ~1e-13
    np.testing.assert_array_equal(out.flags, flags)
    np.testing.assert_array_equal(out.best_idx, bi)
    np.testing.assert_allclose(out.best_cost[bi >= 0], bc[bi >= 0], rtol=0, atol=COST_TOL)
    assert np.isnan(out.best_cost[bi < 0]).all()
    np.testing.assert_array_equal(out.stats, stats)
    return out, flags


@pytest.mark.parametrize("cfg", [
    dict(B=6, nd=5, nv=5, nt=5, n_obs=10, T_obs=100, moving=False, seed=11),
    dict(B=4, nd=9, nv=9, nt=7, n_obs=50, T_obs=50, moving=True, seed=12),
    dict(B=3, nd=5, nv=5, nt=5, n_obs=0, T_obs=0, moving=False, seed=13),
    dict(B=3, nd=3, nv=4, nt=2, n_obs=7, T_obs=33, moving=True, seed=14),
    dict(B=2, nd=1, nv=1, nt=1, n_obs=3, T_obs=20, moving=True, seed=15),
])
def test_dense_vs_oracle(oracle, engine, cfg):
    batch = synth.make_batch(cfg["B"], cfg["nd"], cfg["nv"], cfg["nt"], cfg["n_obs"], cfg["T_obs"], cfg["moving"], cfg["seed"])
    out, flags = compare_dense(oracle, engine, batch)
    if cfg["n_obs"] >= 10:
        coll = (flags & 4) != 0
        assert 0.02 < coll.mean() < 0.98, "scene should be neither empty nor fully blocked"
