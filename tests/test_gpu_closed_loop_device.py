"""GPU: closed-loop stepping on the device (fp_plan_dense / fp_plan_fiss + fp_advance, no host round trips)
against the reference's closed loop on DEU_Flensburg-1_1_T-1 (tests/golden/g5_closed_loop.npz)."""
import numpy as np
import pytest

from conftest import load_golden
from fiss_plus_planner_amd import synth

pytestmark = pytest.mark.gpu


def _flensburg_batch(g, kind, copies=1):
    from fiss_plus_planner_amd.batch import ProblemBatch, speed_samples
    from fiss_plus_planner_amd.frenet import FrenetState, State
    from fiss_plus_planner_amd.spline import CubicSpline2D
    from fiss_plus_planner_amd.vehicle import Vehicle

    veh = Vehicle()
    sp = CubicSpline2D(g["centerline"][:, 0], g["centerline"][:, 1])
    ref = g["refline"]
    fs = FrenetState()
    init = g["init_state"]
    fs.from_state(State(t=0.0, x=init[0], y=init[1], yaw=init[2], v=init[3]), ref)
    sw = 3.5 - veh.w + (0.3 if kind in ("FISS", "FISS+") else 0.0)
    d, rd = np.linspace(-sw / 2, sw / 2, 5, retstep=True)
    t, rt = np.linspace(8.0, 10.0, 5, retstep=True)
    v, rv = speed_samples(0.0, np.full(copies, 13.5), 5)
    fts = int(g["final_time_step"])
    B = copies
    return ProblemBatch(
        d_samples=d, t_samples=t, v_samples=v, target_speed=np.full(B, 13.5), ego=np.tile(fs.as_start_vector(), (B, 1)),
        frame_of=np.zeros(B), scene_of=np.zeros(B), t_now=np.zeros(B), nx=[len(sp.knots)], knots=sp.knots[None], coef=sp.coef[None],
        obs_pose=g["obs_pose"][:fts][None], obs_dims=g["obs_dims"][None], final_time_step=[fts], veh_l=veh.l, veh_w=veh.w,
        max_speed=veh.max_speed, max_accel=veh.max_accel,
        samp_min=np.tile([-sw / 2, 0.0, 8.0], (B, 1)), samp_max=np.tile([sw / 2, 13.5, 10.0], (B, 1)), samp_res=np.tile([rd, rv[0], rt], (B, 1)))


@pytest.mark.parametrize("kind", ["FOP", "FISS", "FISS+"])
def test_device_closed_loop_matches_reference(engine, kind):
    from fiss_plus_planner_amd import _abi
    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch

    g = load_golden("g5_closed_loop.npz")
    want = g[f"{kind}_rows"]
    batch = _flensburg_batch(g, kind, copies=3)
    db = DeviceBatch(batch, 0)
    run = ClosedLoopRunner(engine, db, np.tile(g["goal_center"], (3, 1)), planner=kind)
    out = run.run(100, trace=True)
    assert len(out.trace) == len(want)
    for i, (row, w) in enumerate(zip(out.trace, want)):
        for e in range(3):  # three identical copies must march in lock step
            np.testing.assert_allclose(row.start[e], w[0:6], rtol=0, atol=1e-7, err_msg=f"cycle {i}")
            assert abs(row.cost[e] - w[6]) < 1e-6, i
            if kind != "FISS+" or True:
                np.testing.assert_array_equal(row.stats[e], w[12:16].astype(int), err_msg=f"cycle {i}")
        np.testing.assert_allclose(row.cart[0], g[f"{kind}_states"][i], rtol=0, atol=1e-6)
    # the reference run ends "within l/2 of the goal centre" after 44 cycles
    assert (out.done == _abi.DONE_GOAL).all() and (out.cycles == len(want)).all()
    # once done, further steps are no-ops
    before = out.ego.copy()
    run.step()
    assert np.array_equal(run.db.t["ego"].cpu().numpy(), before)


def test_device_closed_loop_batch_no_host_sync(engine):
    """A batch of different egos for a fixed number of cycles, enqueued back to back; the final states equal those of
    the host-driven loop (one plan_dense + advance per cycle through the host-buffer ABI)."""
    import ctypes as C

    from fiss_plus_planner_amd import _abi, synth
    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch
    from fiss_plus_planner_amd.engine import _host_batch, make_params

    batch = synth.make_batch(32, 5, 5, 5, 10, 100, False, 61)
    goal = np.stack([batch.coef[:, 0, 40], batch.coef[:, 4, 40]], axis=1)  # knot 40 of every centerline (200 m ahead)
    cycles = 12
    run = ClosedLoopRunner(engine, DeviceBatch(batch, 0), goal, "FOP")
    dev = run.run(cycles)
    # host-driven replay
    hb = synth.make_batch(32, 5, 5, 5, 10, 100, False, 61)
    done = np.zeros(32, dtype=np.int32); cyc = np.zeros(32, dtype=np.int32); cart = np.full((32, 3), np.nan)
    io = _abi.FpLoopIo()
    io.ego, io.t_now, io.done, io.cycles = hb.ego.ctypes.data, hb.t_now.ctypes.data, done.ctypes.data, cyc.ctypes.data
    io.goal_xy, io.cart_state = goal.ctypes.data, cart.ctypes.data
    for _ in range(cycles):
        res = engine.plan_dense(hb, tables=False)
        res.best_idx[done != 0] = -1
        fb = _host_batch(hb)
        p = make_params(hb)
        _abi.check(engine._lib.fp_advance(engine._ctx, C.byref(p), C.byref(fb), res.best_idx.ctypes.data, None, C.byref(io), _abi.FP_MEM_HOST, None))
    np.testing.assert_array_equal(dev.done, done)
    np.testing.assert_array_equal(dev.cycles, cyc)
    np.testing.assert_array_equal(dev.t_now, hb.t_now)
    np.testing.assert_allclose(dev.ego, hb.ego, rtol=0, atol=1e-12)
    assert (dev.cycles > 0).any()


@pytest.mark.parametrize("kind,big", [("FOP", False), ("FISS+", False), ("FOP", True), ("FISS+", True)])
def test_hip_graph_replay_equals_eager_loop(engine, kind, big):
    """One captured [plan -> advance] cycle replayed from a HIP graph gives the same final states as the eager loop.  big: more egos
    than stay resident - the three-workgroups-per-CU lattice instance with its tail split (ticket counters), the winners' index copy
    and the feedback launch order (index order inside a capture) all run inside the captured cycle."""
    from fiss_plus_planner_amd import synth
    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch

    cycles = 5 if big else 9
    goal = None
    outs = []
    for use_graph in (False, True):
        batch = synth.make_config(3, B=900, kind=kind) if big else synth.make_batch(48, 5, 5, 5, 10, 100, False, 62, kind=kind)
        goal = np.stack([batch.coef[:, 0, 30], batch.coef[:, 4, 30]], axis=1)
        run = ClosedLoopRunner(engine, DeviceBatch(batch, 0), goal, kind)
        outs.append(run.run_graph(cycles) if use_graph else run.run(cycles))
    a, b = outs
    np.testing.assert_array_equal(a.done, b.done)
    np.testing.assert_array_equal(a.cycles, b.cycles)
    np.testing.assert_array_equal(a.t_now, b.t_now)
    np.testing.assert_array_equal(a.ego, b.ego)
    assert (a.cycles > 0).any()


# ---- fp_plan_step (ABI 11): plan + hand-over in one launch, and the goal region rule
@pytest.mark.parametrize("B,cfg", [(7, 2), (300, 2), (900, 3)])
def test_plan_step_equals_plan_dense_plus_advance(engine, B, cfg):
    """One launch per cycle (the workgroup that finds an ego's argmin advances the ego) == fp_plan_dense + fp_advance, bit for bit, in
    the latency instances (B = 7), the two-per-CU ones (300) and the three-per-CU ones with their tail split (900)."""
    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch

    goal = np.full((B, 2), 1e9)
    runs = []
    n4 = engine.get_option("lattice_launches_4")
    for fused in (False, True):
        batch = synth.make_config(cfg, B=B)
        runs.append(ClosedLoopRunner(engine, DeviceBatch(batch, 0), goal, "FOP", fused=fused).run(9, trace=True))
    assert engine.get_option("lattice_launches_4") == n4  # (a closed loop's skip mask keeps the batch off the four-per-CU instances)
    a, b = runs
    for k in ("done", "cycles", "t_now"):
        np.testing.assert_array_equal(getattr(a, k), getattr(b, k), err_msg=k)
    assert np.array_equal(a.ego, b.ego) and np.array_equal(a.cart, b.cart, equal_nan=True)
    assert len(a.trace) == len(b.trace) and a.cycles.sum() > B
    for ra, rb in zip(a.trace, b.trace):
        assert np.array_equal(ra.cost, rb.cost, equal_nan=True) and np.array_equal(ra.done, rb.done) and np.array_equal(ra.stats, rb.stats)


@pytest.mark.parametrize("planner,B,cfg", [("FISS+", 7, 2), ("FISS+", 300, 4), ("FISS+", 900, 4), ("FISS", 300, 4)])
def test_plan_fiss_step_equals_plan_fiss_plus_advance(engine, planner, B, cfg):
    """fp_plan_fiss_step (FISS+: the refinement workgroup that settles an ego's trajectory hands the ego over itself; FISS: the advance
    kernel behind the pipeline) == fp_plan_fiss + fp_advance, cycle for cycle and bit for bit, incl. egos that run out of solutions."""
    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch

    goal = np.full((B, 2), 1e9)
    runs = []
    for fused in (False, True):
        batch = synth.make_config(cfg, B=B, kind=planner)
        runs.append(ClosedLoopRunner(engine, DeviceBatch(batch, 0), goal, planner, fused=fused).run(8, trace=True))
    a, b = runs
    for k in ("done", "cycles", "t_now"):
        np.testing.assert_array_equal(getattr(a, k), getattr(b, k), err_msg=k)
    assert np.array_equal(a.ego, b.ego) and np.array_equal(a.cart, b.cart, equal_nan=True)
    assert len(a.trace) == len(b.trace) and a.cycles.sum() > B and (a.done != 0).any() == (b.done != 0).any()
    for ra, rb in zip(a.trace, b.trace):
        assert np.array_equal(ra.cost, rb.cost, equal_nan=True) and np.array_equal(ra.done, rb.done) and np.array_equal(ra.stats, rb.stats)


def test_goal_region_rule_against_the_oracle(oracle, engine):
    """goal_region.is_reached() on the device (fp_loop_io.goal_poly / goal_intervals) against the oracle's exact predicate: every ego
    gets a polygon placed around, beside or exactly ON the position it reaches after one cycle, with and without intervals."""
    from fiss_plus_planner_amd import _abi
    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch

    B = 240
    goal = np.full((B, 2), 1e9)
    free = ClosedLoopRunner(engine, DeviceBatch(synth.make_config(2, B=B), 0), goal, "FOP").run(1)   # where every ego lands
    moved = free.cycles == 1
    assert moved.sum() > B // 3
    rng = np.random.default_rng(8)
    V = 6
    poly = np.zeros((B, V, 2)); nv = np.zeros(B, dtype=np.int32); iv = np.full((B, 6), np.nan)
    for b in range(B):
        x, y = (free.cart[b, 0], free.cart[b, 1]) if moved[b] else (0.0, 0.0)
        kind = b % 6
        w = 2.0 ** rng.integers(-2, 3)          # dyadic sizes: "on the boundary" is exact in fp64
        if kind == 0:    ring = [[x - w, y - w], [x + w, y - w], [x + w, y + w], [x - w, y + w]]                      # around
        elif kind == 1:  ring = [[x + w, y - w], [x + 3 * w, y - w], [x + 3 * w, y + w], [x + w, y + w]]              # beside
        elif kind == 2:  ring = [[x, y - w], [x + w, y - w], [x + w, y + w], [x, y + w]]                              # position ON the left edge
        elif kind == 3:  ring = [[x, y], [x + w, y], [x + w, y + w], [x, y + w]][::-1]                                # ON a vertex, clockwise
        elif kind == 4:  ring = [[x - w, y - w], [x + w, y - w], [x + w, y + w], [x, y + w], [x, y + w / 2], [x - w, y + w / 2]]  # non-convex, inside
        else:            ring = [[x - w, y - w], [x + w, y - w], [x + w, y + 2 * w]]                                  # triangle, half a width inside its long edge
        nv[b] = len(ring)
        poly[b, :len(ring)] = ring
        if b % 4 == 1: iv[b, 0:2] = (0, 0)          # the cycle index of the first cycle is 0
        if b % 4 == 2: iv[b, 0:2] = (1, 5)          # too early
        if b % 4 == 3: iv[b, 2:4] = (free.ego[b, 1] - 1e-3, free.ego[b, 1] + 1e-3) if b % 8 == 3 else (free.ego[b, 1] + 1.0, free.ego[b, 1] + 2.0)
    for fused in (True, False):
        run = ClosedLoopRunner(engine, DeviceBatch(synth.make_config(2, B=B), 0), goal, "FOP", fused=fused, goal_poly=poly, goal_nv=nv, goal_intervals=iv).run(1)
        np.testing.assert_array_equal(run.cycles, free.cycles)
        want = np.array([moved[b] and oracle.goal_reached(poly[b, :nv[b]], free.cart[b, 0], free.cart[b, 1], 0, free.ego[b, 1], free.cart[b, 2], iv[b]) for b in range(B)])
        got = run.done == _abi.DONE_GOAL_REGION
        np.testing.assert_array_equal(got, want)
        assert want.sum() > 20 and (moved & ~want).sum() > 20
        np.testing.assert_array_equal(run.done[~want], free.done[~want])   # the other rules are untouched
