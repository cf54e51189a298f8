"""GPU: closed-loop stepping on the device (fp_plan_dense / fp_plan_fiss + fp_advance, no host round trips)
against the reference's closed loop on DEU_Flensburg-1_1_T-1 (tests/golden/g5_closed_loop.npz)."""
import numpy as np
import pytest

from conftest import load_golden

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
