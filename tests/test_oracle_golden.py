"""CPU: the C oracle against the golden vectors generated from the imported Python reference.

This is what pins the oracle (tests/golden/gen_golden.py made the vectors by running the reference
itself).  Tolerances: the reference solves its small linear systems with LAPACK, the oracle with its
own LU, so coefficients agree to ~1e-13 relative; everything downstream is compared at 1e-9 or tighter,
integer outputs (N, M, masks, indices, Stats) exactly.
"""
import numpy as np
import pytest

from conftest import batch_from_golden, load_golden


# ------------------------------------------------------------------ G1 polynomials
def test_g1_polynomial_coefficients(oracle):
    g = load_golden("g1_poly.npz")
    for k in range(len(g["quintic_in"])):
        a = oracle.quintic_coefs(*g["quintic_in"][k])
        np.testing.assert_allclose(a, g["quintic_coef"][k], rtol=1e-11, atol=1e-13)
        b = oracle.quartic_coefs(*g["quartic_in"][k])
        np.testing.assert_allclose(b, g["quartic_coef"][k], rtol=1e-11, atol=1e-13)
        for m, t in enumerate(g["eval_t"][k]):
            np.testing.assert_allclose(oracle.poly_eval(g["quintic_coef"][k], t), g["quintic_eval"][k, m], rtol=1e-14, atol=1e-14)
            np.testing.assert_allclose(oracle.poly_eval(g["quartic_coef"][k], t), g["quartic_eval"][k, m], rtol=1e-14, atol=1e-14)


def test_g1_singular_time_raises(oracle):
    with pytest.raises(np.linalg.LinAlgError):  # reference: numpy.linalg.LinAlgError at T = 0
        oracle.quintic_coefs(0, 0, 0, 1, 0, 0, 0.0)


# ------------------------------------------------------------------ G2 spline
@pytest.mark.parametrize("name", ["flens", "sinus"])
def test_g2_spline(oracle, name):
    g = load_golden("g2_spline.npz")
    pts = g[f"{name}_pts"]
    knots, cx, cy = oracle.spline2d_build(pts[:, 0], pts[:, 1])
    np.testing.assert_allclose(knots, g[f"{name}_knots"], rtol=0, atol=1e-12)
    coef = np.concatenate([cx, cy])
    np.testing.assert_allclose(coef, g[f"{name}_coef"], rtol=1e-10, atol=1e-12)
    for s, want in zip(g[f"{name}_s_eval"], g[f"{name}_eval"]):
        got = oracle.spline2d_eval(knots, cx, cy, s)
        if np.isnan(want[0]):
            assert got is None, s
        else:
            assert got is not None, s
            np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)
    # the 0.1 m resampled reference line of generate_frenet_frame (frenet_optimal_planner.py:272-278)
    ref = g[f"{name}_refline"]
    s_line = np.arange(0, knots[-1], 0.1)
    assert len(s_line) == len(ref)
    for k in (0, 1, len(ref) // 2, len(ref) - 1):
        np.testing.assert_allclose(oracle.spline2d_eval(knots, cx, cy, s_line[k]), ref[k], rtol=1e-10, atol=1e-10)


# ------------------------------------------------------------------ G7 Cartesian -> Frenet
def test_g7_from_state(oracle):
    g = load_golden("g7_from_state.npz")
    for pose, want in zip(g["poses"], g["frenet"]):
        got = oracle.from_state(*pose, g["refline"])
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------ G8 FISS cost estimate
def test_g8_cost_est(oracle):
    g = load_golden("g8_cost_est.npz")
    b = batch_from_golden(g, "in_")
    probs = oracle.problems_from_batch(b)
    for e, p in enumerate(probs):
        for k, prev in enumerate(g["prev"]):
            est = p.fiss_cost_est(None if prev[0] < 0 else prev)
            np.testing.assert_allclose(est, g["est"][e, k], rtol=1e-13, atol=1e-13)


# ------------------------------------------------------------------ G3 dense FOP tables
def _g3_names():
    return [str(n) for n in load_golden("g3_fop_tables.npz")["names"]]


@pytest.mark.parametrize("name", _g3_names())
def test_g3_dense_tables(oracle, name):
    g = load_golden("g3_fop_tables.npz")
    b = batch_from_golden(g, f"{name}_in_")
    probs = oracle.problems_from_batch(b)
    for e, p in enumerate(probs):
        cost, flags = p.dense_tables()
        np.testing.assert_allclose(cost, g[f"{name}_cost"][e], rtol=0, atol=1e-9)
        np.testing.assert_array_equal((flags >> 8) & 0xFFF, g[f"{name}_N"][e])
        np.testing.assert_array_equal(flags >> 20, g[f"{name}_M"][e])
        np.testing.assert_array_equal((flags & 1) != 0, g[f"{name}_speed"][e])
        np.testing.assert_array_equal((flags & 2) != 0, g[f"{name}_accel"][e])
        np.testing.assert_array_equal((flags & 4) != 0, g[f"{name}_coll"][e])
        np.testing.assert_array_equal((flags & 8) != 0, g[f"{name}_M"][e] < g[f"{name}_N"][e])
        # full 16-series dumps of three candidates (one of them truncated when the case has any)
        for k, idx in enumerate(g[f"{name}_dump_idx"][e]):
            iv, it, i_d = idx % b.nv, (idx // b.nv) % b.nt, idx // (b.nv * b.nt)
            t = p.eval_traj(b.d_samples[i_d], b.v_samples[e, iv], b.t_samples[it], dump=True)
            want = g[f"{name}_dumps"][e, k]
            assert np.array_equal(np.isnan(t.arrays), np.isnan(want))
            m = ~np.isnan(want)
            # curvature series divide by ds: near-stationary points amplify 1e-16 -> compare relatively
            np.testing.assert_allclose(t.arrays[m], want[m], rtol=1e-7, atol=1e-9)
            np.testing.assert_allclose(t.arrays[:11][m[:11]], want[:11][m[:11]], rtol=0, atol=1e-10)


def test_g3_cases_cover_the_edge_cases():
    g = load_golden("g3_fop_tables.npz")
    assert (g["trunc_M"] == 0).any() and (g["trunc_M"] == 1).any() and (g["trunc_M"] < g["trunc_N"]).any()
    assert g["limits_speed"].any() and g["limits_accel"].any()
    assert g["c3_moving50_coll"].any() and not g["c3_moving50_coll"].all()
    assert not g["c0_noobs_coll"].any()


# ------------------------------------------------------------------ G4 plan() of the four planners
def _g4_names():
    import os
    from conftest import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, "g4_plan.npz")):
        return []
    return [str(n) for n in load_golden("g4_plan.npz")["names"]]


@pytest.mark.parametrize("key", _g4_names())
def test_g4_plan(oracle, key):
    g = load_golden("g4_plan.npz")
    b = batch_from_golden(g, f"{key}_in_")
    kind = key.rsplit("_", 1)[1]
    probs = oracle.problems_from_batch(b)
    for e, p in enumerate(probs):
        found = bool(g[f"{key}_found"][e])
        if kind == "FOP":
            r = p.fop_plan()
            # the reference returns a stale best_traj when nothing survives; -1 here, found=False there means "stale/None"
            assert (r.best_idx >= 0) == found
            if found:
                assert r.best_idx == g[f"{key}_flat"][e]
                assert abs(r.best_cost - g[f"{key}_cost"][e]) < 1e-9
        elif kind == "FOP+":
            r = p.fopplus_plan()
            assert (r.best_idx >= 0) == found
            if found:
                assert r.best_idx == g[f"{key}_flat"][e]
                assert abs(r.best_cost - g[f"{key}_cost"][e]) < 1e-9
        elif kind == "FISS":
            r = p.fiss_plan()
            assert (r.best_ijk[0] >= 0) == found
            if found:
                np.testing.assert_array_equal(r.best_ijk, g[f"{key}_idx"][e])
                assert abs(r.best_cost - g[f"{key}_cost"][e]) < 1e-9
            np.testing.assert_array_equal(r.prev_best_idx, g[f"{key}_prev_out"][e])
        else:
            r = p.fissplus_plan()
            assert (not np.isnan(r.best_cost)) == found
            if found:
                assert abs(r.best_cost - g[f"{key}_cost"][e]) < 1e-9
                np.testing.assert_allclose(r.end_state, g[f"{key}_end"][e], rtol=0, atol=1e-9)
                # a refined winner carries idx [-1,-1,-1] in the reference (fiss_plus_planner.py:172-205)
                want_idx = g[f"{key}_idx"][e]
                assert r.refined == bool(want_idx[0] < 0)
                if not r.refined:
                    np.testing.assert_array_equal(r.best_ijk, want_idx)
                tr = g[f"{key}_trace"][e]
                n = int((~np.isnan(tr[:, 0])).sum())
                np.testing.assert_allclose(r.trace.reshape(-1, 4)[:n], tr[:n], rtol=0, atol=1e-8)
            np.testing.assert_array_equal(r.prev_best_idx, g[f"{key}_prev_out"][e])
        np.testing.assert_array_equal(r.stats, g[f"{key}_stats"][e])


# ------------------------------------------------------------------ G6 FISS / FISS+ search with history heuristic
@pytest.mark.parametrize("kind", ["FISS", "FISS+"])
def test_g6_search(oracle, kind):
    g = load_golden("g6_fiss_search.npz")
    b = batch_from_golden(g, "in_")
    probs = oracle.problems_from_batch(b)
    for e, p in enumerate(probs):
        prev = g["prev_in"][e]
        prev = None if prev[0] < 0 else prev
        r = p.fiss_plan(prev) if kind == "FISS" else p.fissplus_plan(prev)
        np.testing.assert_array_equal(r.stats, g[f"{kind}_stats"][e])
        found = bool(g[f"{kind}_found"][e])
        assert (not np.isnan(r.best_cost)) == found
        if found:
            assert abs(r.best_cost - g[f"{kind}_cost"][e]) < 1e-9
            if kind == "FISS":
                np.testing.assert_array_equal(r.best_ijk, g[f"{kind}_idx"][e])
            else:
                np.testing.assert_allclose(r.end_state, g[f"{kind}_end"][e], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(r.prev_best_idx, g[f"{kind}_prev_out"][e])


# ------------------------------------------------------------------ G9 optional curvature checks
def _g9_names():
    return [str(n) for n in load_golden("g9_curvature.npz")["names"]]


def g9_batch(g, name, kind="FOP"):
    """Problem batch of a G9 case with the curvature checks on (FISS kinds use their own lateral lattice)."""
    from fiss_plus_planner_amd.batch import ProblemBatch

    b = batch_from_golden(g, f"{name}_in_")
    b.curvature_limits = tuple(g[f"{name}_limits"])
    if kind in ("FISS", "FISS+"):
        smin, smax, sres = g[f"{name}_fiss_samp"]
        kw = {k: getattr(b, k) for k in ("t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
                                         "obs_pose", "obs_dims", "final_time_step", "veh_l", "veh_w", "max_speed", "max_accel", "tick_t", "check_stride")}
        b = ProblemBatch(d_samples=g[f"{name}_fiss_d_samples"], samp_min=smin, samp_max=smax, samp_res=sres, curvature_limits=b.curvature_limits, **kw)
    return b


@pytest.mark.parametrize("name", _g9_names())
def test_g9_curvature_masks(oracle, name):
    """The three checks the reference carries commented out (frenet_optimal_planner.py:145-150), switched on in the generator
    by un-commenting them in the imported class: per candidate, which check trips - exactly."""
    g = load_golden("g9_curvature.npz")
    b = g9_batch(g, name)
    for e, p in enumerate(oracle.problems_from_batch(b)):
        _, flags = p.dense_tables()
        for k, bit in enumerate((oracle.FLAG_CURVATURE, oracle.FLAG_KAPPA_D, oracle.FLAG_KAPPA_DD)):
            np.testing.assert_array_equal((flags & bit) != 0, g[f"{name}_curv"][e, :, k], err_msg=f"{name} ego {e} check {k}")
        np.testing.assert_array_equal((flags & 1) != 0, g[f"{name}_speed"][e])
        np.testing.assert_array_equal((flags & 2) != 0, g[f"{name}_accel"][e])
    # off by default: no curvature bit is ever set
    b.curvature_limits = None
    for p in oracle.problems_from_batch(b):
        assert not (p.dense_tables()[1] & 0x70).any()


@pytest.mark.parametrize("kind", ["FOP", "FOP+", "FISS", "FISS+"])
@pytest.mark.parametrize("name", _g9_names())
def test_g9_plans_with_curvature_checks(oracle, name, kind):
    g = load_golden("g9_curvature.npz")
    b = g9_batch(g, name, kind)
    for e, p in enumerate(oracle.problems_from_batch(b)):
        r = {"FOP": p.fop_plan, "FOP+": p.fopplus_plan, "FISS": p.fiss_plan, "FISS+": p.fissplus_plan}[kind]()
        key = f"{name}_{kind}"
        np.testing.assert_array_equal(r.stats, g[f"{key}_stats"][e], err_msg=f"{key} ego {e}")
        found = bool(g[f"{key}_found"][e])
        assert (not np.isnan(r.best_cost)) == found, (key, e)
        if not found:
            continue
        assert abs(r.best_cost - g[f"{key}_cost"][e]) < 1e-9
        if kind in ("FOP", "FOP+"):
            assert r.best_idx == g[f"{key}_flat"][e]
        elif kind == "FISS":
            np.testing.assert_array_equal(r.best_ijk, g[f"{key}_idx"][e])
        else:
            np.testing.assert_allclose(r.end_state, g[f"{key}_end"][e], rtol=0, atol=1e-9)


# ------------------------------------------------------------------ G13 tick_t = 0.05: 160 .. 200 points per trajectory
G13_STRIDE = 208


@pytest.mark.parametrize("name", ["tick005", "tick005_short"])
def test_g13_tables_and_series_at_tick_005(oracle, name):
    """The reference run with settings.tick_t = 0.05 (round 5: FP_MAX_POINTS 256): cost, N, M, masks, collision verdicts of every
    candidate and three full series per ego (one of them truncated beyond point 110 in the short-line scene)."""
    g = load_golden("g13_tick005.npz")
    b = batch_from_golden(g, f"{name}_in_")
    assert b.tick_t == 0.05
    for e, p in enumerate(oracle.problems_from_batch(b)):
        cost, flags = p.dense_tables()
        np.testing.assert_allclose(cost, g[f"{name}_cost"][e], rtol=0, atol=1e-9)
        np.testing.assert_array_equal((flags >> 8) & 0xFFF, g[f"{name}_N"][e])
        np.testing.assert_array_equal(flags >> 20, g[f"{name}_M"][e])
        np.testing.assert_array_equal((flags & 1) != 0, g[f"{name}_speed"][e])
        np.testing.assert_array_equal((flags & 2) != 0, g[f"{name}_accel"][e])
        np.testing.assert_array_equal((flags & 4) != 0, g[f"{name}_coll"][e])
        for k, idx in enumerate(g[f"{name}_dump_idx"][e]):
            iv, it, i_d = idx % b.nv, (idx // b.nv) % b.nt, idx // (b.nv * b.nt)
            t = p.eval_traj(b.d_samples[i_d], b.v_samples[e, iv], b.t_samples[it], dump=True, stride=G13_STRIDE)
            want = g[f"{name}_dumps"][e, k]
            assert np.array_equal(np.isnan(t.arrays), np.isnan(want))
            m = ~np.isnan(want)
            np.testing.assert_allclose(t.arrays[:11][m[:11]], want[:11][m[:11]], rtol=0, atol=1e-10)
            # rows 11-15 are difference chains of x / y divided by ds and by tick_t (0.05 here: four times the amplification of 0.1)
            from conftest import assert_series_close
            assert_series_close(t.arrays, want, b.tick_t, f"{name} ego {e} dump {k}")
    assert g[f"{name}_N"].min() == 160 and g[f"{name}_N"].max() == 200 and g[f"{name}_coll"].any()
    if name == "tick005_short":
        assert ((g[f"{name}_M"] < g[f"{name}_N"]) & (g[f"{name}_M"] > 128)).any()


@pytest.mark.parametrize("name", ["tick005", "tick005_short"])
@pytest.mark.parametrize("kind", ["FOP", "FOP+", "FISS", "FISS+"])
def test_g13_plans_at_tick_005(oracle, name, kind):
    g = load_golden("g13_tick005.npz")
    key = f"{name}_{kind}"
    b = batch_from_golden(g, f"{key}_in_" if kind in ("FISS", "FISS+") else f"{name}_in_")
    for e, p in enumerate(oracle.problems_from_batch(b)):
        r = {"FOP": p.fop_plan, "FOP+": p.fopplus_plan, "FISS": p.fiss_plan, "FISS+": p.fissplus_plan}[kind]()
        found = bool(g[f"{key}_found"][e])
        assert (not np.isnan(r.best_cost)) == found
        np.testing.assert_array_equal(r.stats, g[f"{key}_stats"][e])
        if found:
            assert abs(r.best_cost - g[f"{key}_cost"][e]) < 1e-9
            if kind in ("FISS", "FISS+"):
                end = r.end_state if kind == "FISS+" else np.array([b.d_samples[r.best_ijk[0]], b.v_samples[e, r.best_ijk[1]], b.t_samples[r.best_ijk[2]]])
                np.testing.assert_allclose(end, g[f"{key}_end"][e], rtol=0, atol=1e-9)


# ------------------------------------------------------------------ G14 FissPlusPlanner's wall-clock budget, where the clock cannot matter
def _g14_names():
    return [str(n) for n in load_golden("g14_time_limit.npz")["names"]]


@pytest.mark.parametrize("key", _g14_names())
def test_g14_spent_time_budget(oracle, key):
    """fiss_plus_planner.py:152-158, :293-299 with time_limit = -1 (generated by running the reference): has_time_limit False -> the
    refinement loop breaks after its first gradient step = the oracle with max_refine_iters = 1; True -> no refinement = 0 rounds."""
    g = load_golden("g14_time_limit.npz")
    b = batch_from_golden(g, f"{key}_in_")
    rounds = 0 if key.endswith("_over_limited") else 1
    for e, p in enumerate(oracle.problems_from_batch(b)):
        r = p.fissplus_plan(max_refine_iters=rounds)
        found = bool(g[f"{key}_found"][e])
        assert (not np.isnan(r.best_cost)) == found
        np.testing.assert_array_equal(r.stats, g[f"{key}_stats"][e])
        assert g[f"{key}_n_refined"][e] in (0, 7 * rounds)
        if found:
            assert abs(r.best_cost - g[f"{key}_cost"][e]) < 1e-9
            np.testing.assert_allclose(r.end_state, g[f"{key}_end"][e], rtol=0, atol=1e-9)
            want_idx = g[f"{key}_idx"][e]
            assert r.refined == bool(want_idx[0] < 0)
            if not r.refined:
                np.testing.assert_array_equal(r.best_ijk, want_idx)
