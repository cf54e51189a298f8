"""GPU: reference lines too long for a residency's share of the LDS (round 6: the coefficient WINDOW of the three- / four-per-CU
lattice instances, csrc/frenet_lattice_fused.hip `WIN`).

A workgroup keeps every knot but only a window of the spline's coefficient columns (one segment behind the ego, as many ahead as the
layout holds); a trajectory point whose segment lies outside the window reads the batch's table in global memory.  The results must
not depend on any of it: flag words exact, costs within 1e-6 and winners (index + series) equal to the oracle's on lines of 200
knots (the window covers every point), 400 and 1000 knots (the fast profiles run out of the window: the global path), egos at both
ends of the line (the window clamps), and equal to the two-per-CU instance, which keeps the whole table.
"""
import numpy as np
import pytest

from conftest import assert_series_close
from fiss_plus_planner_amd import synth

pytestmark = pytest.mark.gpu


def _check(oracle, batch, out, egos, what):
    for e, pr in zip(egos, oracle.problems_from_batch(batch, egos)):
        r = pr.fop_plan()
        np.testing.assert_array_equal(out.flags[e], r.flags, err_msg=f"{what} ego {e}")
        np.testing.assert_allclose(out.cost[e], r.cost, rtol=0, atol=1e-6, err_msg=f"{what} ego {e}")
        assert out.best_idx[e] == r.best_idx, (what, e)
        if r.best_idx >= 0:
            i_v, q = r.best_idx % batch.nv, r.best_idx // batch.nv
            i_t, i_d = q % batch.nt, q // batch.nt
            t = pr.eval_traj(float(batch.d_samples[i_d]), float(batch.v_samples[e, i_v]), float(batch.t_samples[i_t]), dump=True)
            assert_series_close(out.best_traj[e], t.arrays, batch.tick_t, f"{what} ego {e} series")


@pytest.mark.parametrize("n_knots,B", [(200, 1100), (400, 900), (1000, 800)])
def test_long_lines_take_the_windowed_instances_and_match_the_oracle(oracle, engine, n_knots, B):
    batch = synth.make_batch(B, 9, 9, 7, 50, 50, True, 8800 + n_knots, layout="survey8d", n_knots=n_knots)
    # some egos at the ends of their lines: the window clamps to the first / last segments, trajectories truncate
    batch.ego[5::97, 0] = 0.5
    batch.ego[7::89, 0] = 396.0
    batch.ego[11::101, 0] = 399.9
    before = [engine.get_option(k) for k in ("lattice_launches_2", "lattice_launches_3", "lattice_launches_4")]
    out = engine.plan_dense(batch, tables=True, winner=True)
    after = [engine.get_option(k) for k in ("lattice_launches_2", "lattice_launches_3", "lattice_launches_4")]
    if n_knots <= 400:  # (1000 knots: the knots and the bucket table alone - 12 bytes per knot, kept whole - outgrow a third of the LDS: two per CU)
        assert after[1] + after[2] > before[1] + before[2], f"{n_knots} knots: the launch fell back to two workgroups per CU"
    egos = np.unique(np.concatenate([np.arange(0, B, max(1, B // 24)), np.arange(5, B, 97)[:3], np.arange(7, B, 89)[:3], np.arange(11, B, 101)[:3]]))
    _check(oracle, batch, out, egos, f"{n_knots}-knot lines")
    # the whole batch against the instance that keeps the whole table (two per CU)
    engine.set_option("lattice_occupancy", 2)
    try:
        ref = engine.plan_dense(batch, tables=True, winner=True)
    finally:
        engine.set_option("lattice_occupancy", 0)
    np.testing.assert_array_equal(out.flags, ref.flags)
    np.testing.assert_array_equal(out.best_idx, ref.best_idx)
    np.testing.assert_array_equal(out.cost, ref.cost)
    assert np.array_equal(out.best_traj, ref.best_traj, equal_nan=True)


def test_windowed_run_time_shape_and_three_per_cu(oracle, engine):
    """Another lattice shape (the run-time-shape WIN instances) and the three-per-CU cap."""
    batch = synth.make_batch(900, 7, 6, 5, 23, 64, True, 8901, layout="lanes", n_knots=300)
    for occ in (0, 3):
        engine.set_option("lattice_occupancy", occ)
        try:
            out = engine.plan_dense(batch, tables=True, winner=True)
        finally:
            engine.set_option("lattice_occupancy", 0)
        _check(oracle, batch, out, np.arange(0, 900, 60), f"7x6x5 on 300-knot lines, occupancy cap {occ}")


def test_fissplus_on_long_lines(oracle, engine):
    """A FISS+ call on long lines: the lattice pass takes a windowed instance (no appended search workgroups there: the search follows in
    its own launch) - index, Stats, refinement and cost as the oracle has them."""
    batch = synth.make_batch(900, 9, 9, 7, 50, 50, True, 8902, kind="FISS+", layout="survey8d", n_knots=220)
    out = engine.plan_fiss(batch, "FISS+")
    for e, pr in zip(range(0, 900, 45), oracle.problems_from_batch(batch, range(0, 900, 45))):
        r = pr.fissplus_plan()
        np.testing.assert_array_equal(out.best_ijk[e], r.best_ijk, err_msg=f"ego {e}")
        np.testing.assert_array_equal(out.stats[e], r.stats, err_msg=f"ego {e}")
        assert bool(out.refined[e]) == r.refined
        assert (np.isnan(out.best_cost[e]) and np.isnan(r.best_cost)) or abs(out.best_cost[e] - r.best_cost) <= 1e-6


def test_polygon_scenes_on_long_lines(oracle, engine):
    """Convex-polygon obstacle columns on 240-knot lines: the windowed POLY instances (three per CU), against the oracle and against two per CU."""
    base = synth.make_batch(900, 9, 9, 7, 50, 50, True, 8903, layout="survey8d", n_knots=240)
    batch = synth.with_random_shapes(base, 77, frac=0.5)
    before = engine.get_option("lattice_launches_3")
    out = engine.plan_dense(batch, tables=True, winner=True)
    assert engine.get_option("lattice_launches_3") > before
    _check(oracle, batch, out, np.arange(0, 900, 50), "polygons on 240-knot lines")
    engine.set_option("lattice_occupancy", 2)
    try:
        ref = engine.plan_dense(batch, tables=True, winner=True)
    finally:
        engine.set_option("lattice_occupancy", 0)
    np.testing.assert_array_equal(out.flags, ref.flags)
    np.testing.assert_array_equal(out.best_idx, ref.best_idx)
    assert np.array_equal(out.best_traj, ref.best_traj, equal_nan=True)
