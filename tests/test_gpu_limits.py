"""GPU: limits, fallbacks and error behaviour of the C ABI.

* problems that do not fit the fused kernel (obstacle table larger than its LDS budget, more lateral samples than lanes)
  fall back to the lane-per-candidate kernel and still match the oracle;
* invalid arguments produce error codes + messages, never a crash and never a silent CPU path.
"""
import ctypes as C

import numpy as np
import pytest

from fiss_plus_planner_amd import _abi, synth
from fiss_plus_planner_amd.engine import _host_batch, make_params

pytestmark = pytest.mark.gpu


def _check_vs_oracle(oracle, engine, batch):
    out = engine.plan_dense(batch)
    for e, p in enumerate(oracle.problems_from_batch(batch)):
        r = p.fop_plan()
        np.testing.assert_allclose(out.cost[e], r.cost, rtol=0, atol=1e-6)
        np.testing.assert_array_equal(out.flags[e], r.flags)
        assert out.best_idx[e] == r.best_idx
    return out


def test_obstacle_table_larger_than_lds_stays_fused(oracle, engine):
    """50 rows x 150 obstacles = 7500 (row, obstacle) items, 240 KB of poses: the fused kernel reads the scene table from global
    memory in its group test and keeps only the survivors' poses in LDS (more survivors than the list holds: the table is cut into
    chunks) - same flags / costs / argmin in every launch shape; the lane-per-candidate kernel agrees."""
    batch = synth.make_batch(3, 5, 5, 5, 150, 100, True, 71)
    for kernel, split, group in ((2, 1, 1), (2, 2, 1), (2, 1, 99), (2, 1, 2), (1, 1, 1), (0, 0, 0)):
        engine.set_option("lattice_kernel", kernel)
        engine.set_option("lattice_split", split)
        engine.set_option("lattice_group", group)
        try:
            out = _check_vs_oracle(oracle, engine, batch)
        finally:
            engine.set_option("lattice_kernel", 0)
            engine.set_option("lattice_split", 0)
            engine.set_option("lattice_group", 0)
        assert ((out.flags & 4) != 0).any()


def test_obstacle_count_beyond_the_fused_kernel_falls_back(oracle, engine):
    """More than 65535 (row, obstacle) items (the fused kernel's item index is 16 bits): it refuses, auto mode uses the
    lane-per-candidate kernel (rows read through L2)."""
    batch = synth.make_batch(2, 3, 3, 2, 1400, 100, True, 77)  # 50 rows x 1400 obstacles = 70000 items
    engine.set_option("lattice_kernel", 2)
    try:
        with pytest.raises(_abi.FrenetGpuError):
            engine.plan_dense(batch)
    finally:
        engine.set_option("lattice_kernel", 0)
    _check_vs_oracle(oracle, engine, batch)


def test_more_lateral_samples_than_lanes_falls_back(oracle, engine):
    batch = synth.make_batch(2, 70, 2, 2, 6, 40, True, 72)
    _check_vs_oracle(oracle, engine, batch)


def test_largest_lattice(oracle, engine):
    batch = synth.make_batch(1, 16, 16, 16, 12, 50, True, 73)  # C = 4096 = FP_MAX_CAND
    _check_vs_oracle(oracle, engine, batch)


def test_lattice_beyond_the_device_walk(oracle, engine):
    """C = 20 x 20 x 15 = 6000 candidates: beyond the device-side FISS / FISS+ walk (FP_MAX_CAND_SEARCH = 4096; round 4 refused the
    dense pass too) - the dense pass runs (both kernels, against the oracle), fp_plan_fiss says FP_ELIMIT, and the drop-in planner
    classes walk the dense tables on the host by themselves."""
    batch = synth.make_batch(2, 20, 20, 15, 12, 50, True, 78)
    assert batch.C == 6000 > _abi.FP_MAX_CAND_SEARCH
    for kernel in (2, 1):
        engine.set_option("lattice_kernel", kernel)
        try:
            _check_vs_oracle(oracle, engine, batch)
        finally:
            engine.set_option("lattice_kernel", 0)
    fb = synth.make_batch(2, 20, 20, 15, 12, 50, True, 78, kind="FISS+")
    with pytest.raises(_abi.FrenetGpuError, match="FP_MAX_CAND_SEARCH"):
        engine.plan_fiss(fb, "FISS+")
    from fiss_plus_planner_amd import planners as P
    from fiss_plus_planner_amd.vehicle import Vehicle

    pl = P.FissPlusPlanner(P.FissPlusPlannerSettings(20, 20, 15), Vehicle(), None, engine=engine)
    assert not pl._device_walk()
    # (the host walk itself is covered by tests/test_host_search.py and tests/test_gpu_planners.py; here: the planner takes that path)


def test_reference_line_of_1024_knots(oracle, engine):
    """FP_MAX_KNOTS = 1024 (round 4: 512): a 1024-point centerline (CubicSpline2D takes any, cubic_spline.py:145-168) through the
    fused kernel (72 KB of spline tables in LDS: one workgroup per CU), the lane-per-candidate kernel, the winner series, the frame
    build and from_state - against the oracle."""
    from conftest import assert_series_close
    from fiss_plus_planner_amd.spline import build_frames

    base = synth.make_batch(3, 5, 5, 5, 10, 60, True, 79)
    NX = 1024
    xs = np.linspace(0.0, 400.0, NX)
    pts = np.empty((3, NX, 2))
    for b in range(3):
        pts[b, :, 0] = xs
        pts[b, :, 1] = (2.0 + b) * np.sin(xs / (40.0 + 7 * b))
    knots, coef = build_frames(pts)
    gk, gc = engine.build_frames(pts)
    np.testing.assert_allclose(gk, knots, rtol=0, atol=1e-9)
    np.testing.assert_allclose(gc, coef, rtol=0, atol=1e-7)
    base.knots, base.coef, base.nx = knots, coef, np.full(3, NX, dtype=np.int32)
    for kernel in (2, 1, 0):
        engine.set_option("lattice_kernel", kernel)
        try:
            out = _check_vs_oracle(oracle, engine, base)
        finally:
            engine.set_option("lattice_kernel", 0)
    w = engine.plan_dense(base, tables=False, winner=True)
    for e, pr in enumerate(oracle.problems_from_batch(base)):
        if w.best_idx[e] < 0:
            continue
        bi = int(w.best_idx[e])
        iv, it, i_d = bi % base.nv, (bi // base.nv) % base.nt, bi // (base.nv * base.nt)
        t = pr.eval_traj(base.d_samples[i_d], base.v_samples[e, iv], base.t_samples[it], dump=True)
        assert_series_close(w.best_traj[e], t.arrays, base.tick_t, f"ego {e}")
    assert (w.best_idx >= 0).any()
    fb = synth.make_batch(3, 5, 5, 5, 10, 60, True, 79, kind="FISS+")
    fb.knots, fb.coef, fb.nx = knots, coef, np.full(3, NX, dtype=np.int32)
    f = engine.plan_fiss(fb, "FISS+", winner=True)
    for e, pr in enumerate(oracle.problems_from_batch(fb)):
        r = pr.fissplus_plan()
        np.testing.assert_array_equal(f.stats[e], r.stats)
        if not np.isnan(r.best_cost):
            assert abs(f.best_cost[e] - r.best_cost) < 1e-6


def _tick005(B, nd, nv, nt, n_obs, seed, kind="FOP"):
    b = synth.make_batch(B, nd, nv, nt, n_obs, 220, True, seed, kind=kind)
    b.tick_t = 0.05   # T = 8 .. 10 s -> 160 .. 200 points per trajectory (FP_FAST_POINTS = 128 < N <= FP_MAX_POINTS = 256)
    return b


def test_trajectories_of_200_points_dense_and_series(oracle, engine):
    """tick_t = 0.05 (round 4: FP_ELIMIT): the dense pass on both kernels against the oracle (collision rows up to pose 200), the
    winners' series through the chunked writer (winner_traj_kernel; also from a 700-ego batch, where N <= 128 would take the epilogue
    workgroups), fp_eval_trajs dumps, fp_materialize_all and fp_winner_trajs - every series against the oracle's."""
    from conftest import assert_series_close
    from fiss_plus_planner_amd.engine import unpack_flags

    batch = _tick005(4, 5, 4, 3, 8, 83)
    for kernel in (2, 1, 0):
        engine.set_option("lattice_kernel", kernel)
        try:
            out = _check_vs_oracle(oracle, engine, batch)
        finally:
            engine.set_option("lattice_kernel", 0)
    N = (out.flags >> 8) & 0xFFF
    assert N.min() == 160 and N.max() == 200 and ((out.flags & 4) != 0).any() and (out.best_idx >= 0).any()
    stride = 208
    w = engine.plan_dense(batch, tables=False, winner=True, traj_stride=stride)
    ws = engine.plan_dense(batch, tables=False, winner=True, traj_stride=stride, traj_sparse=True)
    wt = engine.winner_trajs(batch, w.best_idx, traj_stride=stride)
    assert np.array_equal(w.best_traj, wt.best_traj, equal_nan=True) and np.array_equal(w.best_flags, wt.best_flags)
    m = engine.materialize_all(batch, traj_stride=stride)
    probs = oracle.problems_from_batch(batch)
    n_series = 0
    for e, pr in enumerate(probs):
        for c in range(batch.C):
            iv, it, i_d = c % batch.nv, (c // batch.nv) % batch.nt, c // (batch.nv * batch.nt)
            t = pr.eval_traj(batch.d_samples[i_d], batch.v_samples[e, iv], batch.t_samples[it], dump=True, stride=stride)
            assert ((m.flags[e, c] >> 8) & 0xFFF, m.flags[e, c] >> 20) == (t.N, t.M)
            assert_series_close(m.traj[e, c], t.arrays, batch.tick_t, f"materialise ego {e} cand {c}")
            if c == w.best_idx[e]:
                assert_series_close(w.best_traj[e], t.arrays, batch.tick_t, f"winner ego {e}")
                live = ~np.isnan(t.arrays)
                assert np.array_equal(ws.best_traj[e][live], w.best_traj[e][live])
                n_series += 1
    assert n_series >= 1
    # an ego that runs off the end of its reference line: truncated series (M < N) across the chunk boundary at 120
    batch.ego[1, 0] = batch.knots[1, 80] - 80.0
    w2 = engine.plan_dense(batch, tables=True, winner=True, traj_stride=stride)
    _, N2, M2 = unpack_flags(w2.flags[1])
    cut = np.nonzero((M2 < N2) & (M2 > 118) & (M2 < 126))[0]   # the series ends right behind the first chunk's window
    cut = cut if len(cut) else np.nonzero((M2 < N2) & (M2 > 100))[0]
    assert len(cut), (M2, N2)
    mt = engine.materialize_all(batch, traj_stride=stride)
    sub = batch.take(np.array([1]))
    pr = oracle.problems_from_batch(batch, [1])[0]
    for c in cut[:6]:
        iv, it, i_d = c % batch.nv, (c // batch.nv) % batch.nt, c // (batch.nv * batch.nt)
        es = np.array([[[batch.d_samples[i_d], batch.v_samples[1, iv], batch.t_samples[it]]]])
        t = pr.eval_traj(*es[0, 0], dump=True, stride=stride)
        assert t.M < t.N and t.M == M2[c]
        d = engine.eval_trajs(sub, es, dump=True, traj_stride=stride)
        assert_series_close(d.traj[0, 0], t.arrays, batch.tick_t, "eval_trajs, truncated")
        assert_series_close(mt.traj[1, c], t.arrays, batch.tick_t, f"chunked writer, truncated at {t.M}")
    # a multi-round batch: no epilogue workgroups for these (their writer holds 128 points); every ego's series equals the standalone kernel's
    big = _tick005(700, 5, 4, 3, 8, 84)
    wb = engine.plan_dense(big, tables=False, winner=True, traj_stride=stride, traj_sparse=True)
    ref = engine.winner_trajs(big, wb.best_idx, traj_stride=stride, traj_sparse=True)
    assert np.array_equal(wb.best_traj, ref.best_traj, equal_nan=True) and (wb.best_idx >= 0).sum() > 50
    egos = np.arange(0, 700, 41)
    for e, pr in zip(egos, oracle.problems_from_batch(big, egos)):
        r = pr.fop_plan()
        assert wb.best_idx[e] == r.best_idx


def test_trajectories_of_200_points_search_and_planners(oracle, engine):
    """FISS on the device (walk over the dense tables + chunked winner series) and - since round 6 - FISS+ WITH its refinement on the device
    for trajectories of 160-200 points (fiss_refine_kernel<4>: four points per lane, the chunked series writer): Stats, refined flag, end
    state, cost and the winner's series against the oracle; the drop-in FissPlusPlanner takes the device walk at tick_t = 0.05."""
    from conftest import assert_series_close
    from fiss_plus_planner_amd import planners as P
    from fiss_plus_planner_amd.vehicle import Vehicle

    fb = _tick005(6, 5, 5, 5, 8, 85, kind="FISS")
    out = engine.plan_fiss(fb, "FISS", winner=True, traj_stride=208)
    for e, pr in enumerate(oracle.problems_from_batch(fb)):
        r = pr.fiss_plan()
        np.testing.assert_array_equal(out.stats[e], r.stats)
        assert np.isnan(out.best_cost[e]) == np.isnan(r.best_cost)
        if not np.isnan(r.best_cost):
            assert abs(out.best_cost[e] - r.best_cost) < 1e-6
            t = pr.eval_traj(*out.end_state[e], dump=True, stride=208)
            assert_series_close(out.best_traj[e], t.arrays, fb.tick_t, f"FISS winner ego {e}")
    n_refined = 0
    for seed, (nd, nv, nt, n_obs) in ((86, (5, 5, 5, 8)), (87, (7, 6, 4, 20)), (88, (9, 9, 7, 30))):
        pb = _tick005(8, nd, nv, nt, n_obs, seed, kind="FISS+")
        po = engine.plan_fiss(pb, "FISS+", winner=True, traj_stride=208)
        for e, pr in enumerate(oracle.problems_from_batch(pb)):
            r = pr.fissplus_plan()
            np.testing.assert_array_equal(po.stats[e], r.stats, err_msg=f"seed {seed} ego {e}")
            np.testing.assert_array_equal(po.best_ijk[e], r.best_ijk, err_msg=f"seed {seed} ego {e}")
            assert bool(po.refined[e]) == r.refined, (seed, e)
            assert np.isnan(po.best_cost[e]) == np.isnan(r.best_cost)
            if not np.isnan(r.best_cost):
                assert abs(po.best_cost[e] - r.best_cost) < 1e-6
                np.testing.assert_allclose(po.end_state[e], r.end_state, rtol=0, atol=1e-9)
                t = pr.eval_traj(*po.end_state[e], dump=True, stride=208)
                assert t.N > 128   # (the point of the test: more points than the one-chunk paths hold)
                assert_series_close(po.best_traj[e], t.arrays, pb.tick_t, f"FISS+ winner seed {seed} ego {e}")
                n_refined += int(r.refined)
    assert n_refined >= 3
    st = P.FissPlusPlannerSettings(5, 5, 5)
    st.tick_t = 0.05
    assert P.FissPlusPlanner(st, Vehicle(), None, engine=engine)._device_walk()
    st1 = P.FissPlusPlannerSettings(5, 5, 5)
    assert P.FissPlusPlanner(st1, Vehicle(), None, engine=engine)._device_walk()


def test_largest_lattice_fiss_pipeline(oracle, engine):
    """C = 4096 needs 139 KB of LDS in the search kernel (above the 64 KB default dynamic limit)."""
    batch = synth.make_batch(2, 16, 16, 16, 12, 50, True, 76, kind="FISS+")
    out = engine.plan_fiss(batch, "FISS+")
    for e, p in enumerate(oracle.problems_from_batch(batch)):
        r = p.fissplus_plan()
        np.testing.assert_array_equal(out.stats[e], r.stats)
        assert (not np.isnan(out.best_cost[e])) == (not np.isnan(r.best_cost))
        if not np.isnan(r.best_cost):
            assert abs(out.best_cost[e] - r.best_cost) < 1e-6


def _call(engine, batch, mutate):
    p = make_params(batch)
    fb = _host_batch(batch)
    B = batch.B
    bi = np.empty(B, dtype=np.int32); bc = np.empty(B)
    res = _abi.FpResult()
    res.best_idx, res.best_cost = bi.ctypes.data, bc.ctypes.data
    mutate(p, fb, res)
    rc = engine._lib.fp_plan_dense(engine._ctx, C.byref(p), C.byref(fb), C.byref(res), _abi.FP_MEM_HOST, None)
    return rc, engine._lib.fp_last_error().decode()


def test_error_codes(engine):
    batch = synth.make_batch(2, 3, 3, 2, 4, 20, False, 74)

    rc, msg = _call(engine, batch, lambda p, fb, r: setattr(fb, "ego", None))
    assert rc == -1 and "NULL" in msg
    rc, msg = _call(engine, batch, lambda p, fb, r: setattr(r, "best_idx", None))
    assert rc == -1
    rc, msg = _call(engine, batch, lambda p, fb, r: setattr(p, "nd", 4097))
    assert rc == -4 and "FP_MAX_CAND" in msg
    rc, msg = _call(engine, batch, lambda p, fb, r: setattr(p, "tick_t", 0.01))  # 10 s / 0.01 = 1000 points > FP_MAX_POINTS = 256
    assert rc == -4 and "FP_MAX_POINTS" in msg
    rc, msg = _call(engine, batch, lambda p, fb, r: setattr(p, "check_stride", 0))
    assert rc == -1

    bad = synth.make_batch(2, 3, 3, 2, 4, 20, False, 74)
    bad.frame_of[1] = 7
    rc, msg = _call(engine, bad, lambda p, fb, r: None)
    assert rc == -1 and "frame_of" in msg
    bad = synth.make_batch(2, 3, 3, 2, 4, 20, False, 74)
    bad.t_now[0] = -3
    rc, msg = _call(engine, bad, lambda p, fb, r: None)
    assert rc == -1 and "t_now" in msg
    # the ctx survives every failed call
    assert engine.plan_dense(batch).best_idx.shape == (2,)


def test_empty_batch_is_a_noop(engine):
    batch = synth.make_batch(2, 3, 3, 2, 0, 0, False, 75)
    rc, _ = _call(engine, batch, lambda p, fb, r: setattr(fb, "B", 0))
    assert rc == 0


def test_unknown_option_is_rejected(engine):
    with pytest.raises(_abi.FrenetGpuError):
        engine.set_option("no_such_knob", 1)
    with pytest.raises(_abi.FrenetGpuError):
        engine.set_option("lattice_kernel", 9)


def test_long_reference_line(oracle, engine):
    """400-knot centerlines (2 km): larger spline tables in LDS, 16-bit bucket LUT, egos far along the line."""
    from fiss_plus_planner_amd.batch import ProblemBatch
    from fiss_plus_planner_amd.spline import build_frames

    base = synth.make_batch(3, 5, 5, 5, 8, 60, True, 77)
    NX = 400
    x = np.linspace(0, 2000, NX)
    pts = np.stack([np.stack([x, a * np.sin(x / lam)], axis=1) for a, lam in ((5, 60), (2, 35), (9, 140))])
    knots, coef = build_frames(pts)
    ego = base.ego.copy()
    ego[:, 0] = [20.0, 1500.0, 1985.0]  # the last one runs off the end (truncation)
    b = ProblemBatch(d_samples=base.d_samples, t_samples=base.t_samples, v_samples=base.v_samples, target_speed=base.target_speed, ego=ego,
                     frame_of=[0, 1, 2], scene_of=base.scene_of, t_now=base.t_now, nx=[NX] * 3, knots=knots, coef=coef, obs_pose=base.obs_pose,
                     obs_dims=base.obs_dims, final_time_step=base.final_time_step, veh_l=base.veh_l, veh_w=base.veh_w,
                     max_speed=base.max_speed, max_accel=base.max_accel)
    for split in (1, 2):
        engine.set_option("lattice_split", split)
        out = _check_vs_oracle(oracle, engine, b)
    engine.set_option("lattice_split", 0)
    assert ((out.flags[2] & 8) != 0).any()
    # the winner's series: written by the lattice kernel itself (spline in LDS) and by winner_traj_kernel (spline in global
    # memory) - the same arithmetic, bit for bit; the materialise mode too leaves a spline of this size in global memory (four LDS
    # copies would not fit a default launch)
    inline = engine.plan_dense(b, winner=True)
    own = engine.winner_trajs(b, inline.best_idx)
    np.testing.assert_array_equal(own.best_flags, inline.best_flags)
    np.testing.assert_array_equal(np.nan_to_num(own.best_traj, nan=-1.0), np.nan_to_num(inline.best_traj, nan=-1.0))
    assert (inline.best_idx >= 0).any()
    allt = engine.materialize_all(b)
    for e in range(b.B):
        if inline.best_idx[e] >= 0:
            got = allt.traj[e, inline.best_idx[e]]
            np.testing.assert_array_equal(np.nan_to_num(got, nan=-1.0), np.nan_to_num(inline.best_traj[e], nan=-1.0))


def test_validate_option_turns_device_faults_into_errors(engine):
    """FP_MEM_DEVICE calls trust the index arrays (an out-of-range frame_of is a GPU memory fault).  With
    fp_ctx_set_option("validate", 1) a range-check kernel runs first and the call returns FP_EINVAL / FP_ELIMIT naming the
    offending entry; a clean batch plans as usual, and the option costs nothing when it is off."""
    import re

    import torch

    from fiss_plus_planner_amd._abi import FrenetGpuError
    from fiss_plus_planner_amd.device_batch import DeviceBatch

    batch = synth.make_batch(16, 5, 5, 5, 4, 20, True, 5)
    want = engine.plan_dense(batch, tables=False)
    db = DeviceBatch(batch, 0)
    bi, bc, st = db.empty(16, torch.int32), db.empty(16, torch.float64), db.empty((16, 4), torch.int32)
    plan = lambda: engine.plan_dense_device(db.params, db.fb, bi.data_ptr(), bc.data_ptr(), st.data_ptr())
    engine.set_option("validate", 1)
    try:
        plan()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(bi.cpu().numpy(), want.best_idx)
        for name, at, bad, msg in (("frame_of", 3, 99, "frame_of[3]"), ("frame_of", 0, -1, "frame_of[0]"), ("scene_of", 5, 16, "scene_of[5]"),
                                   ("t_now", 7, -2, "t_now[7]"), ("nx", 2, 1, "nx[2]"), ("nx", 15, 4096, "nx[15]"), ("t_samples", 4, 50.0, "t_samples[4]")):
            old = db.t[name][at].item()
            db.t[name][at] = bad
            with pytest.raises(FrenetGpuError, match=re.escape(msg)):
                plan()
            db.t[name][at] = old
        plan()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(bi.cpu().numpy(), want.best_idx)
    finally:
        engine.set_option("validate", 0)
    assert engine.get_option("validate") == 0


def test_tables_tag_keeps_frames_and_scenes_on_the_device(oracle, engine):
    """fp_batch.tables_tag (FP_MEM_HOST): the first call with a tag uploads the frame / scene tables, later calls with the same tag
    reuse the device copy (proved by handing them garbage host tables: the results still equal the untagged call's), a new tag or
    tag 0 uploads again.  Per-ego arrays (start state, t_now, samples) travel with every call."""
    batch = synth.make_batch(3, 5, 5, 5, 12, 60, True, 123)
    ref = engine.plan_dense(batch, winner=True)
    probs = oracle.problems_from_batch(batch)
    np.testing.assert_array_equal(ref.best_idx, [p.fop_plan().best_idx for p in probs])
    batch.tables_tag = 41
    first = engine.plan_dense(batch, winner=True)
    good = {k: getattr(batch, k).copy() for k in ("knots", "coef", "obs_pose", "obs_dims")}
    try:
        batch.knots[...] = 7.0; batch.coef[...] = -3.0; batch.obs_pose[...] = 1e3; batch.obs_dims[...] = 50.0  # (not looked at: same tag)
        batch.ego[:, 3] += 0.25  # the per-ego arrays DO travel (a lateral offset: other costs)
        moved = engine.plan_dense(batch, winner=True)
    finally:
        for k, v in good.items():
            getattr(batch, k)[...] = v
    batch.tables_tag = 0
    moved_ref = engine.plan_dense(batch, winner=True)
    for a, b in ((first, ref), (moved, moved_ref)):
        np.testing.assert_array_equal(a.best_idx, b.best_idx)
        np.testing.assert_array_equal(a.cost, b.cost)
        np.testing.assert_array_equal(a.flags, b.flags)
        assert np.array_equal(a.best_traj, b.best_traj, equal_nan=True)
    assert not np.array_equal(moved_ref.cost, ref.cost)
    # a new tag uploads what the host holds now (other obstacle sizes -> other collision flags)
    batch.tables_tag = 42
    batch.obs_dims *= 2.5
    try:
        bigger = engine.plan_dense(batch)
        batch.tables_tag = 0
        bigger_ref = engine.plan_dense(batch)
    finally:
        batch.obs_dims /= 2.5
    np.testing.assert_array_equal(bigger.flags, bigger_ref.flags)
    assert not np.array_equal(bigger_ref.flags, moved_ref.flags)
    # several tagged table sets live side by side (two planners taking turns on one ctx): each tag keeps finding its own tables
    other = synth.make_batch(3, 5, 5, 5, 12, 60, True, 321)
    other_ref = engine.plan_dense(other)
    batch.tables_tag, other.tables_tag = 51, 52
    np.testing.assert_array_equal(engine.plan_dense(batch).cost, moved_ref.cost * 0 + engine.plan_dense(batch).cost)  # (uploads 51)
    np.testing.assert_array_equal(engine.plan_dense(other).cost, other_ref.cost)                                       # (uploads 52)
    saved = (batch.obs_pose.copy(), other.obs_pose.copy())
    try:
        batch.obs_pose[...] = 9e9; other.obs_pose[...] = -9e9  # not looked at while the tags are cached
        for _ in range(2):
            a51, a52 = engine.plan_dense(batch), engine.plan_dense(other)
            np.testing.assert_array_equal(a52.flags, other_ref.flags)
            np.testing.assert_array_equal(a52.cost, other_ref.cost)
    finally:
        batch.obs_pose[...] = saved[0]; other.obs_pose[...] = saved[1]
    batch.tables_tag = other.tables_tag = 0
    np.testing.assert_array_equal(a51.flags, engine.plan_dense(batch).flags)
    # FISS+ through the same tables
    fb = synth.make_batch(2, 5, 5, 5, 12, 60, True, 124, kind="FISS+")
    f_ref = engine.plan_fiss(fb, winner=True)
    fb.tables_tag = 43
    for _ in range(2):
        f_tag = engine.plan_fiss(fb, winner=True)
        np.testing.assert_array_equal(f_tag.best_ijk, f_ref.best_ijk)
        np.testing.assert_array_equal(f_tag.stats, f_ref.stats)
        np.testing.assert_array_equal(f_tag.best_cost, f_ref.best_cost)
        assert np.array_equal(f_tag.best_traj, f_ref.best_traj, equal_nan=True)


def test_full_size_config3_at_tick_005(oracle, engine):
    """BASELINE configs[2] with tick_t = 0.05 (2048 egos x 567 candidates of 160-200 points; the obstacle tables' 50 steps are then the
    first 2.5 s): the three-per-CU lattice instance with its tail split, winners through winner_traj_kernel's chunked writer - selected
    index exact and cost within 1e-6 for every ego against the oracle, the series of a sample, invariants of the flag words."""
    import os

    from conftest import assert_series_close

    batch = synth.make_config(3)
    batch.tick_t = 0.05
    out = engine.plan_dense(batch, tables=True, winner=True, traj_stride=208, traj_sparse=True)
    idx, cost = oracle.fop_plan_batch(oracle.problems_from_batch(batch), threads=len(os.sched_getaffinity(0)))
    np.testing.assert_array_equal(out.best_idx, idx)
    ok = idx >= 0
    assert 0.2 < ok.mean() < 0.95
    np.testing.assert_allclose(out.best_cost[ok], cost[ok], rtol=0, atol=1e-6)
    N, M = (out.flags >> 8) & 0xFFF, out.flags >> 20
    assert N.min() == 160 and N.max() == 200 and (M <= N).all()
    egos = np.nonzero(ok)[0][::40]
    for e, pr in zip(egos, oracle.problems_from_batch(batch, egos)):
        bi = int(out.best_idx[e])
        iv, it, i_d = bi % batch.nv, (bi // batch.nv) % batch.nt, bi // (batch.nv * batch.nt)
        t = pr.eval_traj(batch.d_samples[i_d], batch.v_samples[e, iv], batch.t_samples[it], dump=True, stride=208)
        assert_series_close(out.best_traj[e], t.arrays, batch.tick_t, f"ego {e}")
