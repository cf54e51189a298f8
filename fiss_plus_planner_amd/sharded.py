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

import ctypes as C
import logging
import os
from concurrent.futures import ThreadPoolExecutor
from types import SimpleNamespace

import numpy as np

from . import _abi
from .batch import ProblemBatch
from .engine import TRAJ_STRIDE, FrenetEngine, device_count, fiss_rounds

_log = logging.getLogger(__name__)


def _rows(ns: SimpleNamespace, lo: int, hi: int) -> SimpleNamespace:
    """The same output object restricted to egos [lo, hi): contiguous views, written in place by the shard's call."""
    return SimpleNamespace(**{k: (v[lo:hi] if isinstance(v, np.ndarray) else v) for k, v in vars(ns).items()})


class _Group:
    """fp_group over the shards' contexts (include/frenet_gpu.h): one PERSISTENT worker thread per ctx inside the library, fed through
    a mailbox.  A resident plan step of W shards is ONE ctypes call that posts W prebuilt argument blocks (`submit`) - the enqueues
    then run side by side on the workers' threads, without the GIL, a thread-pool hop or per-call argument marshalling (the
    ThreadPoolExecutor path of round 4 cost a Python submit + a future per shard and step: host-bound long before 8 x 0.146 ms steps).
    `wait` returns when every worker has enqueued its call, with the first error."""

    def __init__(self, engines):
        self._lib = _abi.load()
        self.n = len(engines)
        ctxs = (C.c_void_p * self.n)(*[e._ctx for e in engines])
        self._h = C.c_void_p()
        _abi.check(self._lib.fp_group_create(ctxs, self.n, C.byref(self._h)))

    def submit(self, calls):
        _abi.check(self._lib.fp_group_submit(self._h, calls))

    def wait(self):
        _abi.check(self._lib.fp_group_wait(self._h))

    def close(self):
        if self._h:
            self._lib.fp_group_destroy(self._h)
            self._h = C.c_void_p()


class ShardedDeviceBatch:
    """A problem batch cut into contiguous ego ranges and uploaded ONCE: shard r's rows of every per-ego array, and the frame /
    scene tables its egos reference, live in the HBM of shard r's device (`device_batch.DeviceBatch`), next to the shard's output
    buffers; every shard of the ENGINE owns one HIP stream, shared by all batches uploaded to it.  Per-ego results (index, cost, Stats, flag word; the FISS arrays) land in PINNED host
    arrays covering the whole batch - rank r's slice is written by rank r only, nothing is gathered or concatenated, no collective.
    Shards on one device write those results straight through the device mapping of the pinned block (no copy command on the
    stream); with several devices every shard brings its packed results home with one asynchronous copy.  Big outputs (dense
    tables, winner series) stay in HBM per shard (`shard.cost_tbl`, `.flag_tbl`, `.best_traj` torch tensors) and are fetched on
    demand (`fetch_tables`, `fetch_series`)."""

    def __init__(self, eng: "ShardedEngine", batch: ProblemBatch, tables: bool = False, winner: bool = False, fiss_rounds_max: int = 3,
                 traj_stride: int | None = None):
        import torch

        from .device_batch import DeviceBatch

        self.torch, self.eng, self.B, self.C = torch, eng, batch.B, batch.C
        self.tables, self.winner = bool(tables), bool(winner)
        self.traj_stride = int(traj_stride) if traj_stride else (int(np.ceil(batch.t_samples.max() / batch.tick_t)) + 15) // 16 * 16
        self.R = int(fiss_rounds_max)
        B, W = batch.B, eng.world
        one_device = len(set(eng.devices)) == 1
        self.zero_copy = one_device  # results written through the pinned block's device mapping (see class docstring)
        i32, f64 = torch.int32, torch.float64
        pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()  # noqa: E731
        # whole-batch pinned results; numpy views of the same memory are what the plan calls return
        self._pinned = dict(best_idx=pin(B, i32), best_cost=pin(B, f64), stats=pin((B, 4), i32), best_flags=pin(B, i32),
                            best_ijk=pin((B, 3), i32), end_state=pin((B, 3), f64), refined=pin(B, i32), prev_best_idx=pin((B, 3), i32))
        self.host = SimpleNamespace(**{k: v.numpy() for k, v in self._pinned.items()})
        self.host.best_flags = self.host.best_flags.view(np.uint32)
        self.shards = []
        for r, (lo, hi) in enumerate(eng.bounds(B, W)):
            if hi <= lo:
                continue
            sb = batch.shard(r, W)
            dev = eng.devices[r]
            with torch.cuda.device(dev):
                db = DeviceBatch(sb, dev)
                n = hi - lo
                # the stream belongs to the SHARD (one fp_ctx = one stream at a time, include/frenet_gpu.h): every batch uploaded to
                # this engine enqueues on it, so calls on different resident batches of one shard stay in order
                sh = SimpleNamespace(rank=r, lo=lo, hi=hi, engine=eng.engines[r], db=db, stream=eng.shard_stream(r))
                d = torch.device("cuda", dev)
                sh.cost_tbl = torch.empty((n, batch.C), dtype=f64, device=d) if tables else None
                sh.flag_tbl = torch.empty((n, batch.C), dtype=i32, device=d) if tables else None
                sh.best_traj = torch.full((n, 16, self.traj_stride), float("nan"), dtype=f64, device=d) if winner else None
                sh.prev = torch.full((n, 3), -1, dtype=i32, device=d)
                sh.trace = None
                if self.zero_copy:
                    sh.res = {k: v[lo:hi] for k, v in self._pinned.items()}  # device-visible views of the pinned rows
                else:
                    sh.res = {k: torch.empty((n,) + tuple(v.shape[1:]), dtype=v.dtype, device=d) for k, v in self._pinned.items()}
                sh.res["prev_best_idx"] = sh.prev  # (in / out history: always in HBM, copied home after the call)
            self.shards.append(sh)

    # -- plumbing
    def synchronize(self):
        """Every shard's stream has run dry (the calls posted to the engine's workers have been enqueued first)."""
        self.eng.group_wait()
        for sh in self.shards:
            sh.stream.synchronize()

    def round_calls(self, key, build):
        """The fp_shard_call array of one kind of resident call on this batch, built once (`build(shard)` -> (FpShardCall fields,
        objects to keep alive)) and re-posted every step; slots of shards without egos stay empty (params = NULL)."""
        cache = self.__dict__.setdefault("_rounds", {})
        if key not in cache:
            calls = (_abi.FpShardCall * self.eng.world)()
            keep = []
            for sh in self.shards:
                fields, alive = build(sh)
                c = calls[sh.rank]
                c.params, c.batch, c.stream = C.pointer(sh.db.params), C.pointer(sh.db.fb), sh.stream.cuda_stream
                for k, v in fields.items():
                    setattr(c, k, v)
                keep.append(alive)
            cache[key] = (calls, keep)
        return cache[key][0]

    def home_copies(self, sh, names):
        """fp_copy list that brings a shard's per-ego results to the pinned rows (nothing when they are written there directly)."""
        todo = [k for k in names if not self.zero_copy or k == "prev_best_idx"]
        arr = (_abi.FpCopy * max(1, len(todo)))()
        for i, k in enumerate(todo):
            dst = self._pinned[k][sh.lo:sh.hi]
            arr[i].dst, arr[i].src, arr[i].bytes = dst.data_ptr(), sh.res[k].data_ptr(), dst.numel() * dst.element_size()
        return arr, len(todo)

    def reset_state(self, batch: ProblemBatch):
        """Upload the start states / time steps of `batch` again (a closed loop advances the resident ones in place)."""
        torch = self.torch
        for sh in self.shards:
            with torch.cuda.stream(sh.stream):
                sh.db.t["ego"].copy_(torch.from_numpy(np.ascontiguousarray(batch.ego[sh.lo:sh.hi])), non_blocking=False)
                sh.db.t["t_now"].copy_(torch.from_numpy(np.ascontiguousarray(batch.t_now[sh.lo:sh.hi])), non_blocking=False)
        self.synchronize()

    def fetch_tables(self):
        """(cost [B, C], flags [B, C]) from the shards' HBM (requires upload(..., tables=True) and a plan_dense(tables=True) call)."""
        self.synchronize()
        cost = np.concatenate([sh.cost_tbl.cpu().numpy() for sh in self.shards])
        flags = np.concatenate([sh.flag_tbl.cpu().numpy().view(np.uint32) for sh in self.shards])
        return cost, flags

    def fetch_series(self):
        """best_traj [B, 16, traj_stride] (compact layout: elements past a row's length are whatever the buffer held - NaN here)."""
        self.synchronize()
        return np.concatenate([sh.best_traj.cpu().numpy() for sh in self.shards])


class ShardedEngine:
    def __init__(self, devices=None, shards_per_device: int = 1, engine_factory=FrenetEngine):
        if devices is None:
            devices = list(range(device_count()))
        if not devices:
            raise RuntimeError("ShardedEngine: no GPU visible (the engine has no CPU path)")
        self.devices = [int(d) for d in devices for _ in range(max(1, int(shards_per_device)))]
        if int(shards_per_device) > 1 and "GPU_MAX_HW_QUEUES" not in os.environ:
            # Several shards on one device only overlap when their streams sit on different hardware queues; the HIP runtime's default
            # of four lets two created streams share one (a two-shard closed loop measured 270 instead of 147 us per cycle).  The
            # runtime reads the variable once, when it starts: this helps only if no HIP call was made yet - export it otherwise.
            os.environ["GPU_MAX_HW_QUEUES"] = "8"
            _log.info("ShardedEngine(shards_per_device=%d): set GPU_MAX_HW_QUEUES=8 (takes effect only if the HIP runtime has not started)",
                      int(shards_per_device))
        self.engines = [engine_factory(d) for d in self.devices]
        self._pool = ThreadPoolExecutor(max_workers=len(self.engines), thread_name_prefix="frenet-shard")  # host-staged calls (the *_host methods)
        self._group = None   # resident calls: the library's own worker threads (created on first use; real engines only)

    world = property(lambda self: len(self.engines))

    def shard_stream(self, r: int):
        """The HIP stream (torch handle) shard r's resident calls enqueue on; created on first use, one per shard for the engine's life."""
        streams = self.__dict__.setdefault("_streams", {})
        if r not in streams:
            import torch

            streams[r] = torch.cuda.Stream(torch.device("cuda", self.devices[r]))
        return streams[r]

    def group(self) -> _Group:
        if self._group is None:
            self._group = _Group(self.engines)
        return self._group

    def group_wait(self):
        if self._group is not None:
            self._group.wait()

    def close(self):
        pool, self._pool = getattr(self, "_pool", None), None
        if pool is not None:
            pool.shutdown(wait=True)
        grp, self._group = getattr(self, "_group", None), None
        if grp is not None:
            grp.close()
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
    def _plan_dense_host(self, batch: ProblemBatch, tables: bool = True, winner: bool = False, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """FrenetOptimalPlanner.plan() for every ego (frenet_optimal_planner.py:247-270), sharded: the result object of
        FrenetEngine.plan_dense for the whole batch."""
        out = FrenetEngine.dense_outputs(batch.B, batch.C, tables, winner, traj_stride, traj_sparse)
        return self._run(batch, out, lambda eng, sb, view, lo, hi: eng.plan_dense(sb, tables, winner, traj_stride, traj_sparse, out=view))

    def _plan_fiss_host(self, batch: ProblemBatch, kind: str = "FISS+", prev_best_idx=None, w_heuristic: float = 10.0, max_refine_iters: int = 3,
                  decaying_factor: float = 0.5, winner: bool = False, trace: bool = False, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """FissPlanner.plan / FissPlusPlanner.plan for every ego (fiss_planner.py:190-270, fiss_plus_planner.py:61-170), sharded."""
        R = fiss_rounds(kind, max_refine_iters)
        out = FrenetEngine.fiss_outputs(batch.B, R, winner, trace, traj_stride, traj_sparse)
        prev = None if prev_best_idx is None else np.ascontiguousarray(prev_best_idx, dtype=np.int32)
        return self._run(batch, out, lambda eng, sb, view, lo, hi: eng.plan_fiss(
            sb, kind, None if prev is None else prev[lo:hi], w_heuristic, max_refine_iters, decaying_factor, winner, trace, traj_stride,
            traj_sparse, out=view))

    def _closed_loop_host(self, batch: ProblemBatch, goal_xy: np.ndarray, planner: str = "FOP", max_cycles: int = 100):
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

    # ------------------------------------------------------------------ entry points: host batch (staged per call) or resident shards
    def plan_dense(self, batch, *args, **kw):
        """FrenetOptimalPlanner.plan() for every ego (frenet_optimal_planner.py:247-270).  `batch`: a ProblemBatch (host arrays,
        staged through FP_MEM_HOST on every call: `_plan_dense_host`) or a ShardedDeviceBatch from `upload` (resident:
        `_plan_dense_resident`, the call only enqueues kernels)."""
        return (self._plan_dense_resident if isinstance(batch, ShardedDeviceBatch) else self._plan_dense_host)(batch, *args, **kw)

    def plan_fiss(self, batch, *args, **kw):
        """FissPlanner.plan / FissPlusPlanner.plan for every ego; host batch (`_plan_fiss_host`) or resident shards
        (`_plan_fiss_resident`)."""
        return (self._plan_fiss_resident if isinstance(batch, ShardedDeviceBatch) else self._plan_fiss_host)(batch, *args, **kw)

    def closed_loop(self, batch, *args, **kw):
        """The closed loop of planning.py:120-162 for every ego; host batch (uploaded for the loop: `_closed_loop_host`) or resident
        shards (`_closed_loop_resident`: the resident states are advanced in place)."""
        return (self._closed_loop_resident if isinstance(batch, ShardedDeviceBatch) else self._closed_loop_host)(batch, *args, **kw)

    # ------------------------------------------------------------------ resident shards: upload once, plan many times
    # (the *_host methods above re-stage the batch through FP_MEM_HOST on every call - the PCIe-inclusive path, ~30x below the
    # resident rate on BASELINE configs[2])
    def upload(self, batch: ProblemBatch, tables: bool = False, winner: bool = False, fiss_rounds_max: int = 3, traj_stride: int | None = None) -> ShardedDeviceBatch:
        """Cut `batch` into the engine's shards and make every shard resident on its device (see ShardedDeviceBatch)."""
        return ShardedDeviceBatch(self, batch, tables, winner, fiss_rounds_max, traj_stride)

    def _round(self, sdb: ShardedDeviceBatch, calls, sync: bool):
        """One resident call on every shard: a single ctypes call posts the prebuilt argument blocks to the library's per-ctx worker
        threads (fp_group_submit), which enqueue side by side.  sync=False returns at once - the next round may be posted while the
        workers are still enqueuing this one; sdb.synchronize() waits for the workers first."""
        self.group().submit(calls)
        if sync:
            sdb.synchronize()

    def _plan_dense_resident(self, sdb: ShardedDeviceBatch, tables: bool = False, winner: bool = False, sync: bool = True):
        """fp_plan_dense on every resident shard (FP_MEM_DEVICE: the call only enqueues kernels on the shard's stream).  Returns
        sdb.host (numpy views of the pinned result block: best_idx, best_cost, stats, best_flags); with sync=False the arrays are
        valid after sdb.synchronize().  tables / winner: the dense tables / the winners' series are written to the shard's HBM
        buffers (upload(..., tables=True / winner=True))."""
        if (tables and not sdb.tables) or (winner and not sdb.winner):
            raise ValueError("upload(batch, tables=..., winner=...) did not reserve the buffers this call asks for")

        def build(sh):
            r = sh.res
            res = _abi.FpResult()
            res.best_idx, res.best_cost, res.stats = r["best_idx"].data_ptr(), r["best_cost"].data_ptr(), r["stats"].data_ptr()
            res.cost_tbl, res.flag_tbl = (sh.cost_tbl.data_ptr(), sh.flag_tbl.data_ptr()) if tables else (None, None)
            res.best_flags, res.best_traj = (r["best_flags"].data_ptr(), sh.best_traj.data_ptr()) if winner else (None, None)
            res.traj_stride, res.traj_sparse = sdb.traj_stride, 1
            copies, n = sdb.home_copies(sh, ("best_idx", "best_cost", "stats") + (("best_flags",) if winner else ()))
            return dict(result=C.pointer(res), copies=copies, n_copies=n), (res, copies)

        self._round(sdb, sdb.round_calls(("dense", bool(tables), bool(winner)), build), sync)
        return sdb.host

    def _plan_fiss_resident(self, sdb: ShardedDeviceBatch, kind="FISS+", prev_best_idx=None, w_heuristic: float = 10.0, max_refine_iters: int = 3,
                            decaying_factor: float = 0.5, winner: bool = False, sync: bool = True):
        """fp_plan_fiss on every resident shard.  prev_best_idx [B, 3] (-1 = None): uploaded when given, else the history the
        previous call left on the device is used (None on the first call = no history).  Returns sdb.host (best_ijk, best_cost,
        end_state, refined, stats, prev_best_idx (+ best_flags)); the series go to shard.best_traj."""
        if winner and not sdb.winner:
            raise ValueError("upload(batch, winner=True) did not reserve the series buffers")
        torch = sdb.torch
        R = fiss_rounds(kind, max_refine_iters)
        plus = kind in ("FISS+", _abi.FP_FISS_PLUS)
        if prev_best_idx is not None:
            prev = np.ascontiguousarray(prev_best_idx, dtype=np.int32).reshape(sdb.B, 3)
            self.group_wait()  # (the copies below go to the shards' streams from this thread: behind whatever the workers still enqueue)
            for sh in sdb.shards:
                with torch.cuda.device(sh.db.dev), torch.cuda.stream(sh.stream):
                    sh.prev.copy_(torch.from_numpy(prev[sh.lo:sh.hi]), non_blocking=False)

        def build(sh):
            r = sh.res
            opts = _abi.FpFissOpts(_abi.FP_FISS_PLUS if plus else _abi.FP_FISS, R, w_heuristic, decaying_factor)
            io = _abi.FpFissIo()
            io.samp_min, io.samp_max, io.samp_res = (sh.db.t[k].data_ptr() for k in ("samp_min", "samp_max", "samp_res"))
            io.prev_best_idx, io.best_ijk, io.best_cost = sh.prev.data_ptr(), r["best_ijk"].data_ptr(), r["best_cost"].data_ptr()
            io.end_state, io.refined, io.stats, io.trace = r["end_state"].data_ptr(), r["refined"].data_ptr(), r["stats"].data_ptr(), None
            io.best_flags = r["best_flags"].data_ptr() if winner else None
            io.best_traj = sh.best_traj.data_ptr() if winner else None
            io.traj_stride, io.traj_sparse = sdb.traj_stride, 1
            copies, n = sdb.home_copies(sh, ("best_ijk", "best_cost", "end_state", "refined", "stats", "prev_best_idx") + (("best_flags",) if winner else ()))
            return dict(fiss_opts=C.pointer(opts), fiss_io=C.pointer(io), copies=copies, n_copies=n), (opts, io, copies)

        self._round(sdb, sdb.round_calls(("fiss", plus, R, float(w_heuristic), float(decaying_factor), bool(winner)), build), sync)
        return sdb.host

    def _closed_loop_resident(self, sdb: ShardedDeviceBatch, goal_xy: np.ndarray, planner: str = "FOP", max_cycles: int = 100):
        """The device-resident closed loop (planning.py:120-162) on shards that are ALREADY resident: no upload, the resident start
        states are advanced in place (sdb.reset_state(batch) rewinds them).  One fp_group round per cycle: fp_plan_step for FOP,
        fp_plan_fiss + fp_advance for FISS / FISS+ - the host thread posts max_cycles rounds, the workers enqueue them."""
        from .device_batch import ClosedLoopRunner

        torch = sdb.torch
        goal = np.ascontiguousarray(goal_xy, dtype=np.float64).reshape(sdb.B, 2)
        B = sdb.B
        out = SimpleNamespace(done=np.empty(B, dtype=np.int32), cycles=np.empty(B, dtype=np.int32), ego=np.empty((B, 6)),
                              t_now=np.empty(B, dtype=np.int32), cart=np.empty((B, 3)))
        self.group_wait()
        runners = {}
        for sh in sdb.shards:
            with torch.cuda.device(sh.db.dev), torch.cuda.stream(sh.stream):
                runners[sh.rank] = ClosedLoopRunner(sh.engine, sh.db, goal[sh.lo:sh.hi], planner)  # (its own fp_batch view: skip = its `done` array, no hint)
        calls = (_abi.FpShardCall * self.world)()
        keep = []
        for sh in sdb.shards:
            run, c = runners[sh.rank], calls[sh.rank]
            c.params, c.batch, c.stream, c.loop = C.pointer(sh.db.params), C.pointer(run.fb), sh.stream.cuda_stream, C.pointer(run.io)
            if planner == "FOP":
                res = _abi.FpResult()
                res.best_idx, res.best_cost, res.stats = run.best_idx.data_ptr(), run.best_cost.data_ptr(), run.stats.data_ptr()
                c.result = C.pointer(res)
                keep.append(res)
            else:
                c.fiss_opts, c.fiss_io = C.pointer(run.fopts), C.pointer(run.fio)
        grp = self.group()
        for _ in range(max_cycles):
            grp.submit(calls)
        sdb.synchronize()
        for sh in sdb.shards:
            run = runners[sh.rank]
            for k, v in (("done", run.done), ("cycles", run.cycles), ("ego", sh.db.t["ego"]), ("t_now", sh.db.t["t_now"]), ("cart", run.cart)):
                getattr(out, k)[sh.lo:sh.hi] = v.cpu().numpy()
        return out
