"""ShardedEngine - one batch of ego problems planned across several GPUs.

Ego problems are independent inside a plan cycle (planners/benchmark/planning.py:120-162 runs them one after the other; nothing
is exchanged), so the batch is cut into CONTIGUOUS ego ranges, one per shard.  A shard = one fp_ctx (its own HIP stream and
device arena) + one host thread; the frames / scenes an ego range references are re-indexed for the shard
(ProblemBatch.shard), every shard writes its results straight into its slice of the caller's output arrays, and there is no
collective of any kind: xGMI is not on the path.  Sequential dependence exists only across the cycles of one ego (next state =
point 1 of the winner), which stays on one device - `closed_loop` keeps every shard's state resident on its GPU.

    with ShardedEngine() as eng:              # every visible GPU
        out = eng.plan_dense(batch)           # same result object as FrenetEngine.plan_dense(batch)

`shards_per_device` > 1 cuts finer than the device count (a 1-GPU box can exercise the whole path; on real multi-GPU nodes it
lets one device's H2D staging overlap another shard's kernels).
"""
from __future__ import annotations

import os

# Several shards on one device only overlap when their streams sit on different hardware queues; the HIP runtime's default of four
# lets two created streams share one (their launches then serialise: a two-shard closed loop measured 270 instead of 147 us per
# cycle).  Read once when the runtime starts, so this only helps when it has not started yet; export it yourself otherwise.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from concurrent.futures import ThreadPoolExecutor  # noqa: E402
from types import SimpleNamespace

import numpy as np

from .batch import ProblemBatch
from .engine import TRAJ_STRIDE, FrenetEngine, device_count


def _rows(ns: SimpleNamespace, lo: int, hi: int) -> SimpleNamespace:
    """The same output object restricted to egos [lo, hi): contiguous views, written in place by the shard's call."""
    return SimpleNamespace(**{k: (v[lo:hi] if isinstance(v, np.ndarray) else v) for k, v in vars(ns).items()})


class ShardedEngine:
    def __init__(self, devices=None, shards_per_device: int = 1, engine_factory=FrenetEngine):
        if devices is None:
            devices = list(range(device_count()))
        if not devices:
            raise RuntimeError("ShardedEngine: no GPU visible (the engine has no CPU path)")
        self.devices = [int(d) for d in devices for _ in range(max(1, int(shards_per_device)))]
        self.engines = [engine_factory(d) for d in self.devices]
        self._pool = ThreadPoolExecutor(max_workers=len(self.engines), thread_name_prefix="frenet-shard")

    world = property(lambda self: len(self.engines))

    def close(self):
        pool, self._pool = getattr(self, "_pool", None), None
        if pool is not None:
            pool.shutdown(wait=True)
        for e in getattr(self, "engines", []):
            e.close()
        self.engines = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, name: str, value: int):
        for e in self.engines:
            e.set_option(name, value)

    @staticmethod
    def bounds(B: int, world: int):
        """Contiguous ego ranges, the same rule as ProblemBatch.shard: rank r owns [B r / world, B (r + 1) / world)."""
        return [((B * r) // world, (B * (r + 1)) // world) for r in range(world)]

    def _run(self, batch: ProblemBatch, out: SimpleNamespace, call):
        """call(engine, shard batch, output views, lo, hi) on every shard, one host thread each (ctypes drops the GIL inside the C
        call, so the shards' staging + kernels really run side by side)."""
        W = self.world
        futs = []
        for r, (lo, hi) in enumerate(self.bounds(batch.B, W)):
            if hi > lo:
                futs.append(self._pool.submit(call, self.engines[r], batch.shard(r, W), _rows(out, lo, hi), lo, hi))
        for f in futs:
            f.result()  # re-raises a shard's exception
        return out

    # ------------------------------------------------------------------ the FrenetEngine surface, for the whole batch
    def plan_dense(self, batch: ProblemBatch, tables: bool = True, winner: bool = False, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """FrenetOptimalPlanner.plan() for every ego (frenet_optimal_planner.py:247-270), sharded: the result object of
        FrenetEngine.plan_dense for the whole batch."""
        out = FrenetEngine.dense_outputs(batch.B, batch.C, tables, winner, traj_stride, traj_sparse)
        return self._run(batch, out, lambda eng, sb, view, lo, hi: eng.plan_dense(sb, tables, winner, traj_stride, traj_sparse, out=view))

    def plan_fiss(self, batch: ProblemBatch, kind: str = "FISS+", prev_best_idx=None, w_heuristic: float = 10.0, max_refine_iters: int = 3,
                  decaying_factor: float = 0.5, winner: bool = False, trace: bool = False, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """FissPlanner.plan / FissPlusPlanner.plan for every ego (fiss_planner.py:190-270, fiss_plus_planner.py:61-170), sharded."""
        R = max_refine_iters if kind in ("FISS+",) else 0
        out = FrenetEngine.fiss_outputs(batch.B, R, winner, trace, traj_stride, traj_sparse)
        prev = None if prev_best_idx is None else np.ascontiguousarray(prev_best_idx, dtype=np.int32)
        return self._run(batch, out, lambda eng, sb, view, lo, hi: eng.plan_fiss(
            sb, kind, None if prev is None else prev[lo:hi], w_heuristic, max_refine_iters, decaying_factor, winner, trace, traj_stride,
            traj_sparse, out=view))

    def closed_loop(self, batch: ProblemBatch, goal_xy: np.ndarray, planner: str = "FOP", max_cycles: int = 100):
        """The closed loop of planners/benchmark/planning.py:120-162 for every ego, device-resident: every shard uploads its egos
        once, steps [plan -> advance] max_cycles times on its own GPU without a host round trip (device_batch.ClosedLoopRunner) and
        only the final states come back, merged in ego order."""
        from .device_batch import ClosedLoopRunner, DeviceBatch

        goal = np.ascontiguousarray(goal_xy, dtype=np.float64).reshape(batch.B, 2)
        B = batch.B
        out = SimpleNamespace(done=np.empty(B, dtype=np.int32), cycles=np.empty(B, dtype=np.int32), ego=np.empty((B, 6)),
                              t_now=np.empty(B, dtype=np.int32), cart=np.empty((B, 3)))

        def call(eng, sb, view, lo, hi):
            import torch

            # every shard's loop runs on a HIP stream of its own (torch's current stream is per thread): several shards on ONE
            # device overlap their cycles - a cycle of a few hundred running egos is one round of workgroups, as long as its slowest
            # ego, and leaves most of the chip idle for the other shard
            with torch.cuda.stream(torch.cuda.Stream(torch.device("cuda", eng.device))):
                res = ClosedLoopRunner(eng, DeviceBatch(sb, eng.device), goal[lo:hi], planner).run(max_cycles)
            for k in ("done", "cycles", "ego", "t_now", "cart"):
                getattr(view, k)[...] = getattr(res, k)

        return self._run(batch, out, call)
