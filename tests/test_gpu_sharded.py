"""GPU: ShardedEngine with several logical shards on the one device of the test box == the single-engine result, for the dense
FOP pass, the FISS+ pipeline and the device-resident closed loop (the 8-GPU node runs the same code with one shard per device)."""
import numpy as np
import pytest

from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.sharded import ShardedEngine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shards", [2, 3])
def test_sharded_equals_single_engine(engine, shards):
    batch = synth.make_config(3, B=301)
    fb = synth.make_config(4, B=301)
    rng = np.random.default_rng(3)
    prev = np.where(rng.uniform(size=(fb.B, 1)) < 0.5, -1, np.column_stack([rng.integers(0, fb.nd, fb.B), rng.integers(0, fb.nv, fb.B),
                                                                           rng.integers(0, fb.nt, fb.B)])).astype(np.int32)
    ref = engine.plan_dense(batch, tables=True, winner=True, traj_stride=112, traj_sparse=True)
    ref_f = engine.plan_fiss(fb, "FISS+", prev_best_idx=prev, winner=True, trace=True)
    with ShardedEngine(devices=[0], shards_per_device=shards) as eng:
        assert eng.world == shards
        out = eng.plan_dense(batch, tables=True, winner=True, traj_stride=112, traj_sparse=True)
        for k in ("best_idx", "best_cost", "stats", "cost", "flags", "best_flags"):
            np.testing.assert_array_equal(getattr(out, k), getattr(ref, k), err_msg=k)
        assert np.array_equal(out.best_traj, ref.best_traj, equal_nan=True)
        f = eng.plan_fiss(fb, "FISS+", prev_best_idx=prev, winner=True, trace=True)
        for k in ("best_ijk", "stats", "refined", "prev_best_idx", "best_flags"):
            np.testing.assert_array_equal(getattr(f, k), getattr(ref_f, k), err_msg=k)
        for k in ("best_cost", "end_state", "best_traj"):
            assert np.array_equal(getattr(f, k), getattr(ref_f, k), equal_nan=True), k
        found = ~np.isnan(ref_f.best_cost)  # (the refinement trace of an ego without a coarse winner is not written)
        assert found.any() and np.array_equal(f.trace[found], ref_f.trace[found], equal_nan=True)


def test_sharded_closed_loop_equals_single_runner(engine):
    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch

    batch = synth.make_batch(150, 5, 5, 5, 10, 100, False, 99)
    goal = np.full((batch.B, 2), 1e9)
    ref = ClosedLoopRunner(engine, DeviceBatch(batch, 0), goal, "FOP").run(12)
    with ShardedEngine(devices=[0], shards_per_device=3) as eng:
        out = eng.closed_loop(synth.make_batch(150, 5, 5, 5, 10, 100, False, 99), goal, "FOP", max_cycles=12)
    for k in ("done", "cycles", "t_now"):
        np.testing.assert_array_equal(getattr(out, k), getattr(ref, k), err_msg=k)
    assert np.array_equal(out.ego, ref.ego) and np.array_equal(out.cart, ref.cart, equal_nan=True)
    assert out.cycles.sum() > batch.B


@pytest.mark.parametrize("kind", ["FOP", "FISS+"])
def test_two_contexts_on_two_streams_interleaved(engine, kind):
    """Independent batches pipelined on one GPU (bench.py's two_streams legs): two engines, each with its own
    fp_ctx and HIP stream, enqueue device-resident closed-loop cycles of two multi-round fleets alternately from one host thread -
    the launches overlap on the device (three-workgroup lattice instances with their tail split and ticket counters, search and
    refinement kernels side by side) and every fleet ends in exactly the state it reaches alone on a single stream."""
    import torch

    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch
    from fiss_plus_planner_amd.engine import FrenetEngine

    cfg, cycles, B = (3 if kind == "FOP" else 4), 6, 1000
    fleets = [synth.make_config(cfg, B=B, ego_offset=k * B, kind=kind) for k in range(2)]
    goal = np.full((B, 2), 1e9)
    alone = [ClosedLoopRunner(engine, DeviceBatch(synth.make_config(cfg, B=B, ego_offset=k * B, kind=kind), 0), goal, kind).run(cycles) for k in range(2)]
    dev = torch.device("cuda", 0)
    other = FrenetEngine(0)
    try:
        parts = []
        for k, eng in enumerate((engine, other)):
            st = torch.cuda.Stream(dev)
            with torch.cuda.stream(st):
                parts.append((ClosedLoopRunner(eng, DeviceBatch(fleets[k], 0), goal, kind), st))
        torch.cuda.synchronize(dev)
        for _ in range(cycles):
            for run, st in parts:
                run.step(st.cuda_stream)
        torch.cuda.synchronize(dev)
        for (run, _), ref in zip(parts, alone):
            np.testing.assert_array_equal(run.done.cpu().numpy(), ref.done)
            np.testing.assert_array_equal(run.cycles.cpu().numpy(), ref.cycles)
            np.testing.assert_array_equal(run.db.t["t_now"].cpu().numpy(), ref.t_now)
            assert np.array_equal(run.db.t["ego"].cpu().numpy(), ref.ego)
        assert alone[0].cycles.sum() > 0 and alone[1].cycles.sum() > 0
    finally:
        other.close()


def test_out_arrays_are_checked_before_they_cross_the_abi(engine):
    """plan_dense(out=...) hands bare addresses to the C side: a strided / wrong-dtype array must be refused, not written through."""
    from fiss_plus_planner_amd.engine import FrenetEngine

    batch = synth.make_config(2, B=8)
    out = FrenetEngine.dense_outputs(batch.B, batch.C, True, False)
    out.cost = np.empty((batch.B, 2 * batch.C))[:, ::2]
    with pytest.raises(ValueError, match="out.cost"):
        engine.plan_dense(batch, tables=True, out=out)
    out = FrenetEngine.dense_outputs(batch.B, batch.C, True, False)
    out.best_idx = np.empty(batch.B, dtype=np.int64)
    with pytest.raises(ValueError, match="out.best_idx"):
        engine.plan_dense(batch, tables=True, out=out)
    with ShardedEngine(devices=[0], shards_per_device=2) as eng:  # kind as the ABI constant sizes the outputs like the name does
        fb = synth.make_config(4, B=16)
        a, b = eng.plan_fiss(fb, 1, trace=True), eng.plan_fiss(fb, "FISS+", trace=True)
        assert a.trace is not None and np.array_equal(a.trace, b.trace, equal_nan=True)


# ---- resident shards (ShardedEngine.upload): BASELINE configs[4] size, 8 logical shards on the one device of the box
CONFIG5_B, CONFIG5_W = 16384, 8


@pytest.fixture(scope="module")
def sharded8():
    with ShardedEngine(devices=[0], shards_per_device=CONFIG5_W) as eng:
        yield eng


def test_resident_shards_config5_fop(engine, sharded8):
    """16 384 egos uploaded once, eight resident shards: index / cost / Stats / flag tables / winner series bit-equal to the single
    engine's host-buffer call; a second resident call (nothing uploaded) gives the same again."""
    batch = synth.make_config(5, B=CONFIG5_B)
    ref = engine.plan_dense(batch, tables=True, winner=True, traj_stride=112, traj_sparse=True)
    sdb = sharded8.upload(batch, tables=True, winner=True, traj_stride=112)
    assert len(sdb.shards) == CONFIG5_W and [(s.lo, s.hi) for s in sdb.shards] == ShardedEngine.bounds(CONFIG5_B, CONFIG5_W)
    for rep in range(2):
        out = sharded8.plan_dense(sdb, tables=True, winner=True)
        for k in ("best_idx", "best_cost", "stats", "best_flags"):
            assert np.array_equal(getattr(out, k), getattr(ref, k), equal_nan=True), (k, rep)
        cost, flags = sdb.fetch_tables()
        assert np.array_equal(cost, ref.cost, equal_nan=True) and np.array_equal(flags, ref.flags)
        assert np.array_equal(sdb.fetch_series(), ref.best_traj, equal_nan=True)
    # asynchronous form: enqueue twice, synchronise once
    sdb.host.best_idx[:] = -7
    sharded8.plan_dense(sdb, sync=False)
    sharded8.plan_dense(sdb, sync=False)
    sdb.synchronize()
    np.testing.assert_array_equal(sdb.host.best_idx, ref.best_idx)
    with pytest.raises(ValueError):
        sharded8.plan_dense(sharded8.upload(synth.make_config(2, B=16)), tables=True)


def test_resident_shards_config5_fissplus(engine, sharded8):
    fb = synth.make_config(4, B=CONFIG5_B)
    rng = np.random.default_rng(5)
    prev = np.where(rng.uniform(size=(fb.B, 1)) < 0.5, -1, np.column_stack([rng.integers(0, fb.nd, fb.B), rng.integers(0, fb.nv, fb.B),
                                                                           rng.integers(0, fb.nt, fb.B)])).astype(np.int32)
    ref = engine.plan_fiss(fb, "FISS+", prev_best_idx=prev, winner=True, traj_stride=112, traj_sparse=True)
    sdb = sharded8.upload(fb, winner=True, traj_stride=112)
    out = sharded8.plan_fiss(sdb, "FISS+", prev_best_idx=prev, winner=True)
    for k in ("best_ijk", "stats", "refined", "prev_best_idx", "best_flags", "best_cost", "end_state"):
        assert np.array_equal(getattr(out, k), getattr(ref, k), equal_nan=True), k
    assert np.array_equal(sdb.fetch_series(), ref.best_traj, equal_nan=True)
    # the history stays resident: a second call without prev_best_idx continues from the first one's winners
    ref2 = engine.plan_fiss(fb, "FISS+", prev_best_idx=ref.prev_best_idx)
    out2 = sharded8.plan_fiss(sdb, "FISS+")
    for k in ("best_ijk", "stats", "refined", "prev_best_idx", "best_cost", "end_state"):
        assert np.array_equal(getattr(out2, k), getattr(ref2, k), equal_nan=True), k


@pytest.mark.parametrize("planner", ["FOP", "FISS+"])
def test_resident_shards_config5_closed_loop(engine, sharded8, planner):
    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch

    cfg, cycles = (5 if planner == "FOP" else 4), 4
    batch = synth.make_config(cfg, B=CONFIG5_B, kind=planner)
    goal = np.full((batch.B, 2), 1e9)
    ref = ClosedLoopRunner(engine, DeviceBatch(batch, 0), goal, planner).run(cycles)
    sdb = sharded8.upload(batch)
    out = sharded8.closed_loop(sdb, goal, planner, max_cycles=cycles)
    for k in ("done", "cycles", "t_now"):
        np.testing.assert_array_equal(getattr(out, k), getattr(ref, k), err_msg=k)
    assert np.array_equal(out.ego, ref.ego) and np.array_equal(out.cart, ref.cart, equal_nan=True)
    assert out.cycles.sum() > batch.B
    # the resident states moved; rewound, a dense pass equals the single engine's on the original batch
    sdb.reset_state(batch)
    if planner == "FOP":
        np.testing.assert_array_equal(sharded8.plan_dense(sdb).best_idx, engine.plan_dense(batch, tables=False).best_idx)


def test_group_rounds_pipelined_and_empty_shards(engine):
    """fp_group (the library's per-ctx worker threads behind the resident calls): 300 rounds posted back to back without waiting
    (the mailbox is one deep: a round waits for the worker's previous call), fewer egos than shards (empty mail slots), results equal
    to the single engine's."""
    batch = synth.make_batch(5, 5, 5, 5, 10, 60, True, 17)
    ref = engine.plan_dense(batch, tables=False)
    with ShardedEngine(devices=[0], shards_per_device=8) as eng:
        sdb = eng.upload(batch)
        assert len(sdb.shards) == 5
        for _ in range(300):
            eng.plan_dense(sdb, sync=False)
        sdb.synchronize()
        np.testing.assert_array_equal(sdb.host.best_idx, ref.best_idx)
        assert np.array_equal(sdb.host.best_cost, ref.best_cost, equal_nan=True)


def test_group_reports_the_failing_shard():
    """A worker's error comes back from fp_group_wait with the shard's index; the group keeps working afterwards."""
    import ctypes as C

    from fiss_plus_planner_amd import _abi

    batch = synth.make_batch(8, 5, 5, 5, 10, 60, True, 18)
    with ShardedEngine(devices=[0], shards_per_device=2) as eng:
        sdb = eng.upload(batch)
        good = eng.plan_dense(sdb).best_idx.copy()
        calls = sdb.round_calls(("dense", False, False), None)  # (cached by the call above)
        bad = (_abi.FpShardCall * 2)()
        C.memmove(bad, calls, C.sizeof(bad))
        p = _abi.FpParams.from_buffer_copy(sdb.shards[1].db.params)
        p.nd = 0
        bad[1].params = C.pointer(p)
        eng.group().submit(bad)
        # a shard that holds an unreported error takes no new work - and the NEXT round is refused as a whole, before any of it is posted
        # (round 5 skipped that one shard's call silently and let the others run ahead)
        with pytest.raises(_abi.FrenetGpuError, match="shard 1 still holds the error"):
            eng.group().submit(calls)
        with pytest.raises(_abi.FrenetGpuError, match="shard 1: lattice sizes"):
            eng.group().wait()
        np.testing.assert_array_equal(eng.plan_dense(sdb).best_idx, good)
        with pytest.raises(_abi.FrenetGpuError, match="needs result"):
            empty = (_abi.FpShardCall * 2)()
            empty[0].params, empty[0].batch = calls[0].params, calls[0].batch
            eng.group().submit(empty)
