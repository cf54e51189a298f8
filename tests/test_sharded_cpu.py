"""ShardedEngine's partition / merge logic on the CPU, with a stand-in engine (no GPU, no HIP library): contiguous ego ranges,
frames / scenes re-indexed per shard, every shard writing into its own slice of the merged outputs, one host thread per shard."""
import threading

import numpy as np
import pytest

from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.engine import FrenetEngine
from fiss_plus_planner_amd.sharded import ShardedEngine


class FakeEngine:
    """Answers from the shard's OWN arrays, so a wrong slice / a wrong re-indexing shows: best_idx = a hash of the ego state and of
    the first obstacle pose of the ego's (re-indexed) scene and the knot count of its (re-indexed) frame."""
    calls = []

    def __init__(self, device):
        self.device = device

    def close(self):
        pass

    def set_option(self, name, value):
        self.opt = (name, value)

    @staticmethod
    def key(batch):
        sc = batch.scene_of
        first = np.where(sc >= 0, batch.obs_pose[np.maximum(sc, 0), 0, 0, 0], -1.0) if batch.S else np.full(batch.B, -1.0)
        return (np.floor(batch.ego[:, 0] * 1000) + np.floor(first * 10) + batch.nx[batch.frame_of] + batch.knots[batch.frame_of, 1] * 7).astype(np.int64)

    def plan_dense(self, batch, tables=True, winner=False, traj_stride=128, traj_sparse=False, out=None):
        FakeEngine.calls.append((self.device, batch.B, threading.current_thread().name))
        out.best_idx[:] = (self.key(batch) % 1000).astype(np.int32)
        out.best_cost[:] = batch.ego[:, 1]
        out.stats[:] = self.device
        if tables:
            out.cost[:] = batch.ego[:, 3, None]
            out.flags[:] = batch.t_now[:, None]
        return out

    def plan_fiss(self, batch, kind, prev, w_heuristic, R, decay, winner, trace, traj_stride, traj_sparse, out=None):
        out.prev_best_idx[...] = -1 if prev is None else prev
        out.best_ijk[:] = out.prev_best_idx + 1
        out.best_cost[:] = batch.samp_max[:, 1]
        out.end_state[:] = batch.samp_min
        out.refined[:] = 1
        out.stats[:] = self.device
        return out


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_partition_and_merge(world):
    B = 37
    batch = synth.make_batch(B, 5, 5, 5, 6, 20, True, seed=5, kind="FISS+")
    batch.t_now[:] = np.arange(B)
    # frames / scenes shared between neighbouring egos and referenced out of order: re-indexing must follow
    batch.frame_of[:] = (np.arange(B) * 7) % B
    batch.scene_of[:] = np.where(np.arange(B) % 5 == 0, -1, (np.arange(B) * 3) % B)
    FakeEngine.calls = []
    with ShardedEngine(devices=list(range(world)), engine_factory=FakeEngine) as eng:
        assert eng.world == world
        out = eng.plan_dense(batch)
        assert sorted(c[1] for c in FakeEngine.calls) == sorted(hi - lo for lo, hi in ShardedEngine.bounds(B, world) if hi > lo)
        assert all(c[2].startswith("frenet-shard") for c in FakeEngine.calls)
        np.testing.assert_array_equal(out.best_idx, (FakeEngine.key(batch) % 1000).astype(np.int32))
        np.testing.assert_array_equal(out.best_cost, batch.ego[:, 1])
        np.testing.assert_array_equal(out.flags[:, 0], np.arange(B))
        np.testing.assert_array_equal(out.cost[:, 3], batch.ego[:, 3])
        for r, (lo, hi) in enumerate(ShardedEngine.bounds(B, world)):
            assert (out.stats[lo:hi] == r).all()  # contiguous ranges, shard r on engine r
        prev = np.arange(3 * B, dtype=np.int32).reshape(B, 3)
        f = eng.plan_fiss(batch, "FISS+", prev_best_idx=prev)
        np.testing.assert_array_equal(f.best_ijk, prev + 1)
        np.testing.assert_array_equal(f.end_state, batch.samp_min)
        f0 = eng.plan_fiss(batch, "FISS")
        assert (f0.prev_best_idx == -1).all()
        eng.set_option("fiss_jump", 0)
        assert all(e.opt == ("fiss_jump", 0) for e in eng.engines)


def test_more_shards_than_egos_and_errors():
    batch = synth.make_batch(3, 5, 5, 5, 0, 0, False, seed=6)
    with ShardedEngine(devices=[0, 1, 2, 3, 4], engine_factory=FakeEngine) as eng:
        out = eng.plan_dense(batch, tables=False)
        np.testing.assert_array_equal(out.best_idx, (FakeEngine.key(batch) % 1000).astype(np.int32))

    class Broken(FakeEngine):
        def plan_dense(self, *a, **k):
            raise RuntimeError("shard failed")

    with ShardedEngine(devices=[0, 1], engine_factory=Broken) as eng:
        with pytest.raises(RuntimeError, match="shard failed"):
            eng.plan_dense(batch)


def test_output_shapes_match_the_single_engine():
    a = FrenetEngine.dense_outputs(5, 125, tables=True, winner=True, traj_stride=100, traj_sparse=True)
    assert a.cost.shape == (5, 125) and a.best_traj.shape == (5, 16, 100) and np.isnan(a.best_traj).all()
    f = FrenetEngine.fiss_outputs(4, 3, winner=False, trace=True)
    assert f.trace.shape == (4, 21, 4) and f.best_traj is None
