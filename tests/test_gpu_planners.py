"""GPU: the four drop-in planner classes against the golden plan() results of the reference
(tests/golden/g4_plan.npz, g6_fiss_search.npz) and the Flensburg closed loop (g5_closed_loop.npz).

Bars: selected index / Stats exact, cost_final within 1e-6 (observed ~1e-13), winner series within 1e-6.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import GOLDEN, batch_from_golden, load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-6


@pytest.fixture(params=["device", "host"])
def search_on(request):
    """FISS / FISS+ drop-ins run their index walk either on the GPU (fp_plan_fiss) or in Python over GPU tables."""
    return request.param


def _planner(kind, b, engine, refine_iters=3, search_on="device"):
    from fiss_plus_planner_amd import planners as P
    from fiss_plus_planner_amd.vehicle import Vehicle, vw_vanagon_params

    vp = vw_vanagon_params()
    vp.l, vp.w = b.veh_l, b.veh_w
    vp.longitudinal.v_max, vp.longitudinal.a_max = b.max_speed, b.max_accel
    veh = Vehicle(vp)
    cls, st = {"FOP": (P.FrenetOptimalPlanner, P.FrenetOptimalPlannerSettings), "FOP+": (P.FopPlusPlanner, P.FrenetOptimalPlannerSettings),
               "FISS": (P.FissPlanner, P.FissPlannerSettings), "FISS+": (P.FissPlusPlanner, P.FissPlusPlannerSettings)}[kind]
    kw = {"search_on": search_on} if kind in ("FISS", "FISS+") else {}
    return cls(st(b.nd, b.nv, b.nt), veh, None, engine=engine, **kw)


def _inputs(b, e):
    from fiss_plus_planner_amd.frenet import FrenetState
    from fiss_plus_planner_amd.obstacles import ObstacleTable

    f = int(b.frame_of[e]); nx = int(b.nx[f])
    pts = np.column_stack([b.coef[f, 0, :nx], b.coef[f, 4, :nx]])
    s, s_d, s_dd, d, d_d, d_dd = b.ego[e]
    fs = FrenetState(t=0.0, s=s, s_d=s_d, s_dd=s_dd, d=d, d_d=d_d, d_dd=d_dd)
    sc = int(b.scene_of[e])
    obs = [] if sc < 0 or b.n_obs == 0 else ObstacleTable(b.obs_pose[sc], b.obs_dims[sc], int(b.final_time_step[sc]))
    return pts, fs, obs


def _check_winner(best, want_dump, NM):
    from fiss_plus_planner_amd.frenet import ARRAY_NAMES

    assert len(best.t) == NM[0] and len(best.x) == NM[1]
    for k, name in enumerate(ARRAY_NAMES):
        got = np.asarray(getattr(best, name))
        want = want_dump[k][~np.isnan(want_dump[k])]
        assert len(got) == len(want), name
        if name in ("c", "c_d", "c_dd"):
            # curvature = diff(yaw)/ds amplifies rounding at crawl speed: relative comparison
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6, err_msg=name)
        else:
            np.testing.assert_allclose(got, want, rtol=0, atol=TOL, err_msg=name)


def _g4_keys():
    p = os.path.join(GOLDEN, "g4_plan.npz")
    return [str(n) for n in np.load(p)["names"]] if os.path.exists(p) else []


@pytest.mark.parametrize("key", _g4_keys())
def test_plan_matches_reference(engine, key, search_on):
    g = load_golden("g4_plan.npz")
    b = batch_from_golden(g, f"{key}_in_")
    kind = key.rsplit("_", 1)[1]
    if kind in ("FOP", "FOP+") and search_on == "host":
        pytest.skip("FOP/FOP+ have a single code path")
    for e in range(b.B):
        pl = _planner(kind, b, engine, search_on=search_on)
        pts, fs, obs = _inputs(b, e)
        pl.generate_frenet_frame(pts)
        best = pl.plan(fs, float(b.target_speed[e]), obs, int(b.t_now[e]))
        found = bool(g[f"{key}_found"][e])
        assert (best is not None) == found, (key, e)
        assert pl.stats.as_tuple() == tuple(g[f"{key}_stats"][e]), (key, e)
        if not found:
            continue
        assert abs(best.cost_final - g[f"{key}_cost"][e]) < TOL
        if kind in ("FOP", "FOP+"):
            assert best.lattice_index == g[f"{key}_flat"][e]
        else:
            np.testing.assert_array_equal(best.idx, g[f"{key}_idx"][e])
            np.testing.assert_allclose([best.end_state.d, best.end_state.s_d, best.end_state.t], g[f"{key}_end"][e], atol=1e-9)
            np.testing.assert_array_equal(pl.prev_best_idx, g[f"{key}_prev_out"][e])
        _check_winner(best, g[f"{key}_win"][e], g[f"{key}_NM"][e])


@pytest.mark.parametrize("kind", ["FISS", "FISS+"])
def test_history_heuristic_and_refinement(engine, kind, search_on):
    g = load_golden("g6_fiss_search.npz")
    b = batch_from_golden(g, "in_")
    for e in range(b.B):
        pl = _planner(kind, b, engine, search_on=search_on)
        pts, fs, obs = _inputs(b, e)
        pl.generate_frenet_frame(pts)
        prev = g["prev_in"][e]
        pl.prev_best_idx = None if prev[0] < 0 else prev.copy()
        best = pl.plan(fs, float(b.target_speed[e]), obs, int(b.t_now[e]))
        assert pl.stats.as_tuple() == tuple(g[f"{kind}_stats"][e]), e
        assert (best is not None) == bool(g[f"{kind}_found"][e])
        if best is not None:
            assert abs(best.cost_final - g[f"{kind}_cost"][e]) < TOL
            np.testing.assert_allclose([best.end_state.d, best.end_state.s_d, best.end_state.t], g[f"{kind}_end"][e], atol=1e-9)
            np.testing.assert_array_equal(pl.prev_best_idx, g[f"{kind}_prev_out"][e])


def _closed_loop(kind, g, engine, search_on="device", frame_on="device"):
    """planners/benchmark/planning.py:101-162 on the Flensburg fixture inputs."""
    from fiss_plus_planner_amd import planners as P
    from fiss_plus_planner_amd.closed_loop import run_closed_loop
    from fiss_plus_planner_amd.obstacles import ObstacleTable
    from fiss_plus_planner_amd.vehicle import Vehicle

    cls, st = {"FOP": (P.FrenetOptimalPlanner, P.FrenetOptimalPlannerSettings), "FOP+": (P.FopPlusPlanner, P.FrenetOptimalPlannerSettings),
               "FISS": (P.FissPlanner, P.FissPlannerSettings), "FISS+": (P.FissPlusPlanner, P.FissPlusPlannerSettings)}[kind]
    kw = {"search_on": search_on} if kind in ("FISS", "FISS+") else {}
    pl = cls(st(5, 5, 5), Vehicle(), None, engine=engine, frame_on=frame_on, **kw)
    fts = int(g["final_time_step"])
    res = run_closed_loop(pl, g["centerline"], g["init_state"], ObstacleTable(g["obs_pose"][:fts], g["obs_dims"], fts), g["goal_center"])
    return res.cycles, np.array(res.states)


@pytest.mark.parametrize("kind", ["FOP", "FOP+", "FISS", "FISS+"])
def test_flensburg_closed_loop(engine, kind, search_on):
    g = load_golden("g5_closed_loop.npz")
    if f"{kind}_rows" not in g.files:
        pytest.skip(f"no closed-loop golden for {kind}")
    if kind in ("FOP", "FOP+") and search_on == "host":
        pytest.skip("FOP/FOP+ have a single code path")
    want = g[f"{kind}_rows"]
    # device search runs with the device-built frame + device projection, the host walk with the numpy ones
    rows, states = _closed_loop(kind, g, engine, search_on, frame_on=search_on)
    assert len(rows) == len(want)
    for i, (r, w) in enumerate(zip(rows, want)):
        np.testing.assert_allclose(r.start, w[0:6], rtol=0, atol=1e-7, err_msg=f"cycle {i} start state")
        assert abs(r.cost - w[6]) < TOL, i
        assert (r.N, r.M) == (int(w[7]), int(w[8])), i
        if kind in ("FISS", "FISS+"):
            np.testing.assert_array_equal(r.idx, w[9:12].astype(int), err_msg=f"cycle {i}")
        assert r.stats == tuple(int(v) for v in w[12:16]), i
        if kind in ("FISS", "FISS+"):  # FOP/FOP+ trajectories carry no end_state in the reference
            np.testing.assert_allclose(r.end, w[16:19], rtol=0, atol=1e-7)
    np.testing.assert_allclose(states, g[f"{kind}_states"], rtol=0, atol=1e-6)


def test_fresh_obstacle_lists_are_never_served_from_a_stale_table(engine):
    """plan() is called cycle after cycle with a NEW list of NEW obstacle objects of the same length (a perception stack that
    rebuilds its tracks): the flattened table must follow the objects, never the recycled address or length of the list.
    A blocked lane must be reported blocked even when the previous cycle's (equal-length) list left it free, and vice versa."""
    import gc
    import sys

    sys.path.insert(0, GOLDEN)
    from refshim import StubObstacle

    from fiss_plus_planner_amd import planners as P
    from fiss_plus_planner_amd.frenet import FrenetState
    from fiss_plus_planner_amd.vehicle import Vehicle

    pl = P.FrenetOptimalPlanner(P.FrenetOptimalPlannerSettings(5, 5, 5), Vehicle(), None, engine=engine)
    pl.generate_frenet_frame(np.column_stack([np.linspace(0, 300, 61), np.zeros(61)]))
    fs = FrenetState(t=0.0, s=10.0, s_d=8.0, s_dd=0.0, d=0.1, d_d=0.0, d_dd=0.0)

    def wall(x):  # a 40 m wide wall across the road at arclength x, for 60 time steps
        return [StubObstacle(2.0, 40.0, np.tile([x, 0.0, 0.0], (60, 1)), 59) for _ in range(3)]

    verdicts = []
    for cycle in range(12):
        blocked = cycle % 2 == 0
        obstacles = wall(25.0 if blocked else 5000.0)
        pl.best_traj = None
        best = pl.plan(fs, 10.0, obstacles, 0)
        verdicts.append(best is None)
        coll = (pl.last_tables[1] & 4) != 0
        assert coll.all() if blocked else not coll.any(), cycle
        del obstacles, best
        gc.collect()  # lets CPython hand the freed list's address to the next one
    assert verdicts == [c % 2 == 0 for c in range(12)]
    # the same objects in a new list: one flattening serves both
    obstacles = wall(25.0)
    t1 = pl._obstacle_table(obstacles)
    assert pl._obstacle_table(list(obstacles)) is t1
    assert pl._obstacle_table(wall(25.0)) is not t1


@pytest.mark.parametrize("kind", ["FISS", "FISS+"])
def test_fiss_all_trajs_is_the_generated_set_in_generation_order(engine, oracle, kind):
    """all_trajs of the FISS planners (visualisation payload, planning.py:339-355): the reference appends the trajectories it
    GENERATED during plan(), in generation order (trajs_per_timestep, fiss_planner.py:131,262-265).  G10 holds that order; the
    series of every listed trajectory are the oracle's."""
    from conftest import assert_series_close

    g = load_golden("g10_generated_order.npz")
    for name in [str(n) for n in g["names"]]:
        b = batch_from_golden(g, f"{name}_in_")
        for e in range(b.B):
            pl = _planner(kind, b, engine)
            pl.materialize_all = True
            pts, fs, obs = _inputs(b, e)
            pl.generate_frenet_frame(pts)
            pl.plan(fs, float(b.target_speed[e]), obs, int(b.t_now[e]))
            n = int(g[f"{name}_{kind}_count"][e])
            gen = pl.all_trajs[-1]
            assert len(gen) == n, (name, e)
            if kind == "FISS":
                assert n == pl.stats.num_trajs_generated
            np.testing.assert_array_equal(np.array([t.idx for t in gen]), g[f"{name}_{kind}_order"][e, :n], err_msg=f"{name} ego {e}")
            np.testing.assert_allclose([t.cost_final for t in gen], g[f"{name}_{kind}_cost"][e, :n], rtol=0, atol=TOL)
            p = oracle.problems_from_batch(b, [e])[0]
            for t in gen[:: max(1, n // 6)]:
                i, j, k = (int(v) for v in t.idx)
                w = p.eval_traj(b.d_samples[i], b.v_samples[e, j], b.t_samples[k], dump=True)
                assert (len(t.t), len(t.x)) == (w.N, w.M)
                got = np.full((16, 128), np.nan)
                from fiss_plus_planner_amd.frenet import ARRAY_NAMES
                for r, nm in enumerate(ARRAY_NAMES):
                    a = np.asarray(getattr(t, nm)); got[r, :len(a)] = a
                assert_series_close(got, w.arrays, b.tick_t, f"{name} ego {e} idx {t.idx}")


def test_fopplus_batch_on_the_device(oracle, engine):
    """FopPlusPlanner.plan for a whole batch without a Python heap per ego (fp_plan_dense with result.fopplus): selected index, cost
    and Stats = the oracle's lazy validation in cost order; mirror-symmetric egos (exact cost ties at the decision point) are
    flagged by the kernel and replayed with the reference's heap order - with the winner's series when asked for."""
    import numpy as np

    from fiss_plus_planner_amd import search, synth

    b = synth.make_batch(96, 4, 5, 5, 10, 100, False, 123)
    # d = d_d = d_dd = 0 on an EVEN number of lateral samples: the two inner samples mirror each other, tie exactly and are the
    # cheapest - the decision hangs on the heap order.  No obstacles for those egos, so the tied pair is feasible.
    b.ego[:6, 3:] = 0.0
    b.scene_of[:6] = -1
    out = engine.plan_fopplus(b, winner=True)
    tab = engine.plan_dense(b, tables=True)
    assert set(out.replayed.tolist()) <= set(range(6)) and len(out.replayed) >= 3  # (an ego whose tied pair is infeasible needs no replay)
    n_found = 0
    for e, pr in enumerate(oracle.problems_from_batch(b)):
        r = pr.fopplus_plan()
        if e in out.replayed:  # tie order = CPython's heapq over the tables, like the reference; the oracle's own heap is not CPython's
            idx, st = search.fopplus_search(tab.cost[e], tab.flags[e])
            assert out.best_idx[e] == (-1 if idx is None else idx) and tuple(out.stats[e]) == tuple(st)
            continue
        assert out.best_idx[e] == r.best_idx, e
        np.testing.assert_array_equal(out.stats[e], r.stats, err_msg=f"ego {e}")
        if r.best_idx >= 0:
            n_found += 1
            assert abs(out.best_cost[e] - r.best_cost) < 1e-6
            assert out.best_idx[e] == tab.best_idx[e]  # without ties FOP+ and FOP agree (fop_plus_planner.py:29-39 vs :263-268)
    assert n_found > 20
    w = engine.winner_trajs(b, out.best_idx)
    np.testing.assert_array_equal(out.best_flags, w.best_flags)
    assert np.array_equal(out.best_traj, w.best_traj, equal_nan=True)
    # a bigger lattice, dynamic obstacles, nothing symmetric: no replay at all
    b3 = synth.make_config(3, B=128)
    o3 = engine.plan_fopplus(b3)
    assert len(o3.replayed) == 0
    for e, pr in enumerate(oracle.problems_from_batch(b3, range(0, 128, 4))):
        r = pr.fopplus_plan()
        assert o3.best_idx[4 * e] == r.best_idx
        np.testing.assert_array_equal(o3.stats[4 * e], r.stats)


# ---- G13: the reference run with settings.tick_t = 0.05 (160 .. 200 points per trajectory; round 5: FP_MAX_POINTS 256)
@pytest.mark.parametrize("name", ["tick005", "tick005_short"])
def test_dense_tables_match_the_reference_at_tick_005(engine, name):
    """Cost, N, M, masks and collision verdicts of every candidate on both lattice kernels, and the three full series per ego the
    fixture holds (through fp_eval_trajs, fp_materialize_all and - for the argmin - the dense call's own series) against the
    reference's, incl. series truncated beyond point 128."""
    from conftest import assert_series_close

    g = load_golden("g13_tick005.npz")
    b = batch_from_golden(g, f"{name}_in_")
    for kernel in (2, 1):
        engine.set_option("lattice_kernel", kernel)
        try:
            out = engine.plan_dense(b)
        finally:
            engine.set_option("lattice_kernel", 0)
        np.testing.assert_allclose(out.cost, g[f"{name}_cost"], rtol=0, atol=TOL)
        np.testing.assert_array_equal((out.flags >> 8) & 0xFFF, g[f"{name}_N"])
        np.testing.assert_array_equal(out.flags >> 20, g[f"{name}_M"])
        np.testing.assert_array_equal((out.flags & 1) != 0, g[f"{name}_speed"])
        np.testing.assert_array_equal((out.flags & 2) != 0, g[f"{name}_accel"])
        np.testing.assert_array_equal((out.flags & 4) != 0, g[f"{name}_coll"])
    m = engine.materialize_all(b, traj_stride=208)
    for e in range(b.B):
        for k, idx in enumerate(g[f"{name}_dump_idx"][e]):
            want = g[f"{name}_dumps"][e, k]
            assert_series_close(m.traj[e, idx], want, b.tick_t, f"{name} materialise ego {e} candidate {idx}")
            iv, it, i_d = idx % b.nv, (idx // b.nv) % b.nt, idx // (b.nv * b.nt)
            es = np.array([[[b.d_samples[i_d], b.v_samples[e, iv], b.t_samples[it]]]])
            d = engine.eval_trajs(b.take(np.array([e])), es, dump=True, traj_stride=208)
            assert_series_close(d.traj[0, 0], want, b.tick_t, f"{name} eval_trajs ego {e} candidate {idx}")


@pytest.mark.parametrize("name", ["tick005", "tick005_short"])
@pytest.mark.parametrize("kind", ["FOP", "FOP+", "FISS", "FISS+"])
def test_plan_matches_reference_at_tick_005(engine, name, kind):
    """The drop-in classes with settings.tick_t = 0.05 return the reference's plan: cost, Stats, end state and the winner's series
    (chunked writer).  Both FISS planners walk on the device - since round 6 FissPlusPlanner's refinement too (fiss_refine_kernel<4>: up to
    256 points per trajectory; rounds 4-5 refined on the host here) - and, a second time, on the host: same plan either way."""
    g = load_golden("g13_tick005.npz")
    key = f"{name}_{kind}"
    b = batch_from_golden(g, f"{key}_in_" if kind in ("FISS", "FISS+") else f"{name}_in_")
    for e, where in [(e, w) for e in range(b.B) for w in (("device", "host") if kind in ("FISS", "FISS+") else ("device",))]:
        pl = _planner(kind, b, engine)
        pl.settings.tick_t = 0.05
        if kind in ("FISS", "FISS+"):
            pl.search_on = where
            assert pl._device_walk() == (where == "device")
        pts, fs, obs = _inputs(b, e)
        pl.generate_frenet_frame(pts)
        best = pl.plan(fs, float(b.target_speed[e]), obs, int(b.t_now[e]))
        found = bool(g[f"{key}_found"][e])
        assert (best is not None) == found, (key, e)
        assert pl.stats.as_tuple() == tuple(g[f"{key}_stats"][e]), (key, e)
        if not found:
            continue
        assert abs(best.cost_final - g[f"{key}_cost"][e]) < TOL
        if kind in ("FISS", "FISS+"):
            np.testing.assert_array_equal(best.idx, g[f"{key}_idx"][e])
            np.testing.assert_allclose([best.end_state.d, best.end_state.s_d, best.end_state.t], g[f"{key}_end"][e], atol=1e-9)
        _check_winner(best, g[f"{key}_win"][e], g[f"{key}_NM"][e])


def _g14_names():
    return [str(n) for n in load_golden("g14_time_limit.npz")["names"]]


@pytest.mark.parametrize("where", ["device", "host"])
@pytest.mark.parametrize("key", _g14_names())
def test_fissplus_time_limit_already_spent(engine, key, where):
    """FissPlusPlannerSettings.time_limit is a wall-clock budget in the reference (fiss_plus_planner.py:152-158, :293-299).  Fixture G14
    holds the reference's plans for the two cases the clock cannot change - the budget is spent when the coarse search returns
    (time_limit = -1): has_time_limit False -> ONE refinement round, True -> none - and the drop-in class returns them, whichever side
    walks (device: one fp_plan_fiss call with 1 / 0 rounds; host: the loop's own clock check)."""
    g = load_golden("g14_time_limit.npz")
    b = batch_from_golden(g, f"{key}_in_")
    for e in range(b.B):
        pl = _planner("FISS+", b, engine, search_on=where)
        pl.settings.time_limit = -1.0
        pl.settings.has_time_limit = key.endswith("_over_limited")
        pts, fs, obs = _inputs(b, e)
        pl.generate_frenet_frame(pts)
        best = pl.plan(fs, float(b.target_speed[e]), obs, int(b.t_now[e]))
        found = bool(g[f"{key}_found"][e])
        assert (best is not None) == found, (key, e)
        assert pl.stats.as_tuple() == tuple(g[f"{key}_stats"][e]), (key, e, where)
        if found:
            assert abs(best.cost_final - g[f"{key}_cost"][e]) < TOL
            np.testing.assert_array_equal(best.idx, g[f"{key}_idx"][e])
            np.testing.assert_allclose([best.end_state.d, best.end_state.s_d, best.end_state.t], g[f"{key}_end"][e], atol=1e-9)
    # and a budget that is NOT spent changes nothing: the default 0.5 s gives the same plan as no limit at all
    pl = _planner("FISS+", b, engine, search_on=where)
    assert pl._refine_rounds() == pl.settings.max_refine_iters
