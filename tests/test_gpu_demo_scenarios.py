"""GPU: the reference's five demo scenarios (cfgs/demo_config.yaml: data/demo/*.xml, 5x5x5 samples) end to end.

The scenario files themselves stay in the build container; tests/golden/g11_demo_scenarios.npz holds what the reference's
harness extracts from them (planning.py:36-100: route centerline, obstacle tables, initial state, goal centre, speed limit) and,
for FOP+, FISS and FISS+, the closed loop of the REFERENCE planners on them (planning.py:101-162): per cycle the start state,
cost, N, M, index, Stats, end state, and the Cartesian state after the cycle.  Here the drop-in planners drive the same loop
through the C ABI; every cycle must match (index / Stats / N / M exact, states and costs to 1e-6).  (FOP's exhaustive loop is pinned
on Flensburg-1 by G5; tests/test_commonroad_xml_cpu.py checks the product's XML reader against the same fixture arrays.)
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _names():
    return [str(n) for n in load_golden("g11_demo_scenarios.npz")["names"]]


@pytest.mark.parametrize("kind", ["FOP+", "FISS", "FISS+"])
@pytest.mark.parametrize("name", _names())
def test_demo_scenario_closed_loop(engine, name, kind):
    from fiss_plus_planner_amd import planners as P
    from fiss_plus_planner_amd.closed_loop import run_closed_loop
    from fiss_plus_planner_amd.obstacles import ObstacleTable
    from fiss_plus_planner_amd.vehicle import Vehicle

    g = load_golden("g11_demo_scenarios.npz")
    want, want_states = g[f"{name}_{kind}_rows"], g[f"{name}_{kind}_states"]
    cls, st = {"FOP+": (P.FopPlusPlanner, P.FrenetOptimalPlannerSettings), "FISS": (P.FissPlanner, P.FissPlannerSettings),
               "FISS+": (P.FissPlusPlanner, P.FissPlusPlannerSettings)}[kind]
    pl = cls(st(5, 5, 5), Vehicle(), None, engine=engine)
    fts = int(g[f"{name}_final_time_step"])
    table = ObstacleTable(g[f"{name}_obs_pose"], g[f"{name}_obs_dims"], fts)
    res = run_closed_loop(pl, g[f"{name}_centerline"], g[f"{name}_init_state"], table, g[f"{name}_goal_center"],
                          max_speed=float(g[f"{name}_max_speed"]))
    # the reference's rows include a last row for a cycle in which plan() returned None (cost NaN)
    n_ok = int(np.sum(~np.isnan(want[:, 6])))
    assert len(res.cycles) == n_ok, (len(res.cycles), n_ok, len(want))
    for i, (r, w) in enumerate(zip(res.cycles, want)):
        np.testing.assert_allclose(r.start, w[0:6], rtol=0, atol=1e-7, err_msg=f"{name} {kind} cycle {i} start state")
        assert abs(r.cost - w[6]) < TOL, (name, kind, i)
        assert (r.N, r.M) == (int(w[7]), int(w[8])), (name, kind, i)
        if kind != "FOP+":
            np.testing.assert_array_equal(r.idx, w[9:12].astype(int), err_msg=f"{name} {kind} cycle {i}")
            np.testing.assert_allclose(r.end, w[16:19], rtol=0, atol=1e-7)
        assert r.stats == tuple(int(v) for v in w[12:16]), (name, kind, i)
    np.testing.assert_allclose(np.array(res.states), want_states, rtol=0, atol=1e-6)
    assert len(res.cycles) >= 40
