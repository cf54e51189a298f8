"""GPU: randomised settings sweep - both lattice kernels and the trajectory kernels against the CPU oracle on
configurations the fixed cases do not reach: odd lattice sizes, other tick / horizon ranges, check strides 1 and 3,
pose tables shorter or longer than the horizon, t_now > 0, ragged spline sizes, obstacles without a state at some steps,
egos sharing frames and scenes."""
import numpy as np
import pytest

from conftest import assert_series_close
from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.batch import ProblemBatch
from fiss_plus_planner_amd.spline import build_frames

pytestmark = pytest.mark.gpu


def random_batch(seed):
    rng = np.random.default_rng(seed)
    nd, nv, nt = int(rng.integers(1, 8)), int(rng.integers(1, 8)), int(rng.integers(1, 6))
    B = int(rng.integers(2, 7))
    n_obs = int(rng.choice([0, 1, 7, 23]))
    T_obs = int(rng.choice([5, 31, 64, 90]))
    base = synth.make_batch(B, max(nd, 2), max(nv, 2), max(nt, 2), n_obs, T_obs, bool(rng.integers(0, 2)), int(rng.integers(1, 10**6)))
    tick = float(rng.choice([0.1, 0.125, 0.2]))
    t_lo = float(rng.uniform(2.0, 6.0))
    t_hi = t_lo + float(rng.uniform(0.5, 5.0))
    if t_hi / tick > 127:
        t_hi = 127 * tick
    t_samples = np.linspace(t_lo, t_hi, nt)
    sw = float(rng.uniform(0.5, 3.0))
    d_samples = np.linspace(-sw / 2, sw / 2, nd)
    vmax = rng.uniform(6, 16, B)
    v_samples = np.stack([np.linspace(0.0, vm, nv) for vm in vmax])
    # ragged frames: every frame keeps a random number of knots; egos share frames / scenes
    F = int(rng.integers(1, B + 1))
    pts = np.stack([base.coef[:F, 0, :], base.coef[:F, 4, :]], axis=-1)
    n_knots = rng.integers(12, 82, F)
    knots, coef = build_frames(pts, n_knots)
    frame_of = rng.integers(0, F, B)
    pose, dims, fts = base.obs_pose, base.obs_dims, base.final_time_step
    if n_obs:
        S = int(rng.integers(1, B + 1))
        pose, dims = pose[:S].copy(), dims[:S]
        drop = rng.uniform(size=pose.shape[:3]) < 0.1   # obstacle without a state at some steps
        pose[drop] = 0.0
        fts = rng.integers(max(T_obs - 10, 1), T_obs + 15, S).astype(np.int32)
        scene_of = rng.integers(-1, S, B)
        t_now = rng.integers(0, T_obs // 2 + 1, B)
    else:
        scene_of = np.full(B, -1)
        t_now = np.zeros(B)
    ego = base.ego.copy()
    ego[:, 0] = rng.uniform(2, 50, B)
    return ProblemBatch(d_samples=d_samples, t_samples=t_samples, v_samples=v_samples, target_speed=vmax, ego=ego, frame_of=frame_of,
                        scene_of=scene_of, t_now=t_now, nx=n_knots, knots=knots, coef=coef, obs_pose=pose, obs_dims=dims,
                        final_time_step=fts, veh_l=float(rng.uniform(3, 6)), veh_w=float(rng.uniform(1.5, 2.2)),
                        max_speed=float(rng.uniform(9, 40)), max_accel=float(rng.uniform(1.0, 12.0)), tick_t=tick,
                        check_stride=int(rng.choice([1, 2, 3])))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FP_PROPERTY_SEEDS", "400"))))
def test_random_settings_vs_oracle(oracle, engine, seed):
    b = random_batch(1000 + seed)
    probs = oracle.problems_from_batch(b)
    ref = [p.fop_plan() for p in probs]
    # (kernel, split, group, resident_groups, occupancy cap): the last two model a one-CU device, so these few egos take the multi-round
    # instances - three and four workgroups per CU (run-time shapes: 80 / 64 VGPRs, the slim fp16-bound layout where it fits)
    try:
        for kernel, split, group, resident, occ in ((2, 1, 1, 0, 0), (2, 2, 1, 0, 0), (2, 1, 3, 0, 0), (2, 1, 99, 0, 0), (1, 1, 1, 0, 0), (2, 1, 1, 2, 3), (2, 1, 1, 2, 0)):
            engine.set_option("lattice_kernel", kernel)
            engine.set_option("lattice_split", split)
            engine.set_option("lattice_group", group)
            engine.set_option("resident_groups", resident)
            engine.set_option("lattice_occupancy", occ)
            out = engine.plan_dense(b, winner=True)
            for e, r in enumerate(ref):
                np.testing.assert_allclose(out.cost[e], r.cost, rtol=0, atol=1e-6, err_msg=f"seed {seed} kernel {kernel} ego {e}")
                np.testing.assert_array_equal(out.flags[e], r.flags, err_msg=f"seed {seed} kernel {kernel} resident {resident} ego {e}")
                assert out.best_idx[e] == r.best_idx, (seed, kernel, resident, e)
    finally:
        engine.set_option("lattice_kernel", 0)
        engine.set_option("lattice_group", 0)
        engine.set_option("resident_groups", 0)
        engine.set_option("lattice_occupancy", 0)
    # the tail split (the last half of the dispatch slots cut in two workgroups each) changes nothing
    engine.set_option("lattice_split", 1)
    engine.set_option("lattice_tail", max(2, b.B // 2))
    try:
        cut = engine.plan_dense(b, winner=True)
    finally:
        engine.set_option("lattice_split", 0)
        engine.set_option("lattice_tail", 0)
    np.testing.assert_array_equal(cut.best_idx, out.best_idx)
    np.testing.assert_array_equal(cut.flags, out.flags)
    assert np.array_equal(cut.best_traj, out.best_traj, equal_nan=True)
    # every candidate's series by the profile-sharing materialise kernel == the winner epilogue's series of the same candidate
    # (fp_winner_trajs: one wavefront per trajectory) and the dense table's N / M words
    m = engine.materialize_all(b, traj_stride=128, traj_sparse=bool(seed & 1))
    dense = engine.plan_dense(b)
    assert np.array_equal(m.flags >> 8, dense.flags >> 8) and np.array_equal(m.flags & 8, dense.flags & 8)
    rngc = np.random.default_rng(7000 + seed)
    for c in rngc.choice(b.C, size=min(b.C, 6), replace=False):
        w = engine.winner_trajs(b, np.full(b.B, c, dtype=np.int32))
        assert np.array_equal(m.traj[:, c], w.best_traj, equal_nan=True), (seed, c)  # (the sparse layout's holes keep the host's NaN prefill)
        np.testing.assert_array_equal(m.flags[:, c] >> 8, w.best_flags >> 8)
    # explicit end states (continuous): cost, flags and the full series
    rng = np.random.default_rng(seed)
    K = 3
    es = np.stack([rng.uniform(b.d_samples[0], b.d_samples[-1] + 1e-9, (b.B, K)), rng.uniform(0, 12, (b.B, K)),
                   rng.uniform(b.t_samples[0], b.t_samples[-1] + 1e-9, (b.B, K))], axis=-1)
    got = engine.eval_trajs(b, es, dump=True)
    for e, p in enumerate(probs):
        for k in range(K):
            t = p.eval_traj(*es[e, k], dump=True)
            assert abs(got.cost[e, k] - t.cost) < 1e-6
            assert got.flags[e, k] == (t.flags | (t.N << 8) | (t.M << 20)), (seed, e, k)
            a, w = got.traj[e, k], t.arrays
            assert np.array_equal(np.isnan(a), np.isnan(w))
            m = ~np.isnan(w)
            np.testing.assert_allclose(a[:11][m[:11]], w[:11][m[:11]], rtol=0, atol=1e-8)
            assert_series_close(a, w, b.tick_t, f"seed {seed} ego {e} traj {k}")  # all 16 rows, rows 11-15 with derived bounds


@pytest.mark.parametrize("kind", ["FISS", "FISS+"])
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FP_PROPERTY_SEEDS_SEARCH", "200"))))
def test_random_settings_search_vs_oracle(oracle, engine, seed, kind):
    """The device-side search walk + refinement on the same randomised settings (lattice axes of at least two samples):
    Stats, selected index / refined end state and history index exactly as the oracle's restatement of the planners."""
    rng = np.random.default_rng(5000 + seed)
    b = None
    for attempt in range(50):  # draw until every lattice axis has >= 2 samples
        b = random_batch(int(rng.integers(1, 10**6)))
        if min(b.nd, b.nv, b.nt) >= 2:
            break
    assert min(b.nd, b.nv, b.nt) >= 2
    b.samp_min = np.column_stack([np.full(b.B, b.d_samples[0]), b.v_samples[:, 0], np.full(b.B, b.t_samples[0])])
    b.samp_max = np.column_stack([np.full(b.B, b.d_samples[-1]), b.v_samples[:, -1], np.full(b.B, b.t_samples[-1])])
    b.samp_res = np.column_stack([np.full(b.B, b.d_samples[1] - b.d_samples[0]), b.v_samples[:, 1] - b.v_samples[:, 0],
                                  np.full(b.B, b.t_samples[1] - b.t_samples[0])])
    prev = np.where(rng.uniform(size=(b.B, 1)) < 0.5, -1, np.column_stack([rng.integers(0, b.nd, b.B), rng.integers(0, b.nv, b.B),
                                                                          rng.integers(0, b.nt, b.B)])).astype(np.int32)
    for table_kb in ((24, 0) if kind == "FISS+" else (24,)):
        engine.set_option("refine_table_kb", table_kb)
        try:
            out = engine.plan_fiss(b, kind, prev_best_idx=prev, trace=True)
        finally:
            engine.set_option("refine_table_kb", 96)
        for e, p in enumerate(oracle.problems_from_batch(b)):
            pv = None if prev[e, 0] < 0 else prev[e]
            r = p.fiss_plan(pv) if kind == "FISS" else p.fissplus_plan(pv)
            np.testing.assert_array_equal(out.stats[e], r.stats, err_msg=f"seed {seed} ego {e}")
            found = not np.isnan(r.best_cost)
            assert (not np.isnan(out.best_cost[e])) == found
            np.testing.assert_array_equal(out.prev_best_idx[e], r.prev_best_idx)
            if not found:
                continue
            assert abs(out.best_cost[e] - r.best_cost) < 1e-6
            if kind == "FISS":
                np.testing.assert_array_equal(out.best_ijk[e], r.best_ijk)
            else:
                assert bool(out.refined[e]) == r.refined
                np.testing.assert_allclose(out.end_state[e], r.end_state, rtol=0, atol=1e-9)
