"""CPU: the host-side table walks (fiss_plus_planner_amd/search.py) against the reference goldens.

The dense tables the walks read are produced here by the oracle (no GPU needed); on the GPU box
tests/test_gpu_planners.py repeats the comparison with tables produced by the HIP kernels.
"""
import numpy as np
import pytest

from conftest import batch_from_golden, load_golden
from fiss_plus_planner_amd import search


def _tables(oracle, b, e):
    p = oracle.problems_from_batch(b, [e])[0]
    cost, flags = p.dense_tables()
    return p, cost, flags


def test_cost_est_table_matches_reference():
    g = load_golden("g8_cost_est.npz")
    b = batch_from_golden(g, "in_")
    for e in range(b.B):
        for k, prev in enumerate(g["prev"]):
            est = search.cost_est_table(b.d_samples, b.v_samples[e], b.t_samples, b.samp_min[e], b.samp_max[e],
                                        None if prev[0] < 0 else prev)
            np.testing.assert_allclose(est, g["est"][e, k], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("kind", ["FISS", "FISS+"])
def test_walks_match_reference_g6(oracle, kind):
    g = load_golden("g6_fiss_search.npz")
    b = batch_from_golden(g, "in_")
    for e in range(b.B):
        p, cost, flags = _tables(oracle, b, e)
        J, F = search.tables_to_dvt(cost, flags, b.nd, b.nv, b.nt)
        prev = g["prev_in"][e]
        E = search.cost_est_table(b.d_samples, b.v_samples[e], b.t_samples, b.samp_min[e], b.samp_max[e], None if prev[0] < 0 else prev)
        if kind == "FISS":
            idx, st = search.fiss_search(J, F, E)
            np.testing.assert_array_equal(st, g["FISS_stats"][e])
            if g["FISS_found"][e]:
                np.testing.assert_array_equal(idx, g["FISS_idx"][e])
            else:
                assert idx is None
        else:
            idx, st = search.fissplus_search(J, F, E)
            ref = p.fissplus_plan(None if prev[0] < 0 else prev, max_refine_iters=0)  # coarse stage only
            np.testing.assert_array_equal(st, ref.stats)
            if g["FISS+_found"][e]:
                np.testing.assert_array_equal(idx, g["FISS+_prev_out"][e])  # prev_best_idx = coarse winner
            else:
                assert idx is None


def test_fopplus_heap_order_on_exact_ties(oracle):
    """Mirror-symmetric start: +/-d candidates tie exactly; FOP+ pops them in CPython heapq order."""
    g = load_golden("g3_fop_tables.npz")
    b = batch_from_golden(g, "mirror_in_")
    for e in range(b.B):
        p, cost, flags = _tables(oracle, b, e)
        assert len(np.unique(cost)) < len(cost), "fixture must contain exact ties"
        idx, st = search.fopplus_search(cost, flags)
        ref = p.fopplus_plan()
        assert idx == ref.best_idx
        np.testing.assert_array_equal(st, ref.stats)


def test_refine_step_zero_gradient_is_flagged():
    x = np.array([0.1, 5.0, 9.0])
    xl = [x - [0.2, 0, 0], x - [0, 1, 0], x - [0, 0, 0.5]]
    xr = [x + [0.2, 0, 0], x + [0, 1, 0], x + [0, 0, 0.5]]
    x_new, res = search.refine_step([1, 1, 1], [1, 1, 1], xl, xr, x, np.array([0.2, 1, 0.5]), 0.5, x - 5, x + 5)
    assert x_new is None and np.allclose(res, [0.1, 0.5, 0.25])
