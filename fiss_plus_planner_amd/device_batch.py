"""Problem batches resident in HBM and closed-loop stepping without host round trips.

PyTorch is plumbing here (device allocations + the stream handle); every array crosses the C ABI as a raw
device address.  `DeviceBatch` uploads a ProblemBatch once; `ClosedLoopRunner` enqueues
[plan -> advance] x cycles on one stream (reference loop: planners/benchmark/planning.py:120-162) and only
reads back at the end (or per cycle, when a trace is requested).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from . import _abi
from .batch import ProblemBatch
from .engine import FrenetEngine, device_batch, launch_order_hint, make_params

_NAMES = ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
          "obs_pose", "obs_dims", "final_time_step")


class DeviceBatch:
    def __init__(self, batch: ProblemBatch, device: int = 0, order_hint: bool = True):
        """order_hint: pass fp_batch.launch_order = the egos by descending speed (engine.launch_order_hint; results do not depend on it)."""
        import torch

        self.torch = torch
        self.host = batch
        self.dev = torch.device("cuda", device)
        self.t = {k: torch.from_numpy(np.ascontiguousarray(getattr(batch, k))).to(self.dev) for k in _NAMES}
        for k in ("samp_min", "samp_max", "samp_res", "obs_poly", "obs_nvert"):
            if getattr(batch, k, None) is not None:
                self.t[k] = torch.from_numpy(getattr(batch, k)).to(self.dev)
        if order_hint and batch.B > 0:
            self.t["launch_order"] = torch.from_numpy(launch_order_hint(batch)).to(self.dev)
        self.params = make_params(batch)
        self.fb = device_batch(batch, {k: (v.data_ptr() if v.numel() else 0) for k, v in self.t.items() if k in _NAMES + ("obs_poly", "obs_nvert", "launch_order")})

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.dev)

    @property
    def B(self):
        return self.host.B

    @property
    def C(self):
        return self.host.C


class ClosedLoopRunner:
    """[plan -> advance] for a whole batch on the device.  planner: "FOP" (fp_plan_dense) or "FISS"/"FISS+" (fp_plan_fiss)."""

    def __init__(self, engine: FrenetEngine, dbatch: DeviceBatch, goal_xy: np.ndarray, planner: str = "FOP", fused: bool = True,
                 goal_poly: np.ndarray | None = None, goal_nv: np.ndarray | None = None, goal_intervals: np.ndarray | None = None):
        """fused: an FOP cycle is ONE launch (fp_plan_step: the workgroup that finds an ego's argmin advances the ego); False = the two
        calls fp_plan_dense + fp_advance (same results; the A/B of tests and bench).
        goal_poly [B, V, 2] + goal_nv [B] (+ goal_intervals [B, 6] = time_step / velocity / orientation lo, hi; NaN = undefined): the
        goal region of goal_region.is_reached() (planning.py:150-153), see fp_loop_io in include/frenet_gpu.h."""
        torch = dbatch.torch
        self.eng, self.db, self.planner, self.fused = engine, dbatch, planner, fused
        B = dbatch.B
        i32, f64 = torch.int32, torch.float64
        self.best_idx = dbatch.empty(B, i32)
        self.best_cost = dbatch.empty(B, f64)
        self.stats = dbatch.empty((B, 4), i32)
        self.done = torch.zeros(B, dtype=i32, device=dbatch.dev)
        self.cycles = torch.zeros(B, dtype=i32, device=dbatch.dev)
        self.goal = torch.from_numpy(np.ascontiguousarray(goal_xy, dtype=np.float64).reshape(B, 2)).to(dbatch.dev)
        self.cart = torch.full((B, 3), float("nan"), dtype=f64, device=dbatch.dev)
        # the runner's OWN view of the resident batch (the shared DeviceBatch keeps its skip mask and its launch-order hint for other callers)
        self.fb = _abi.FpBatch.from_buffer_copy(dbatch.fb)
        self.fb.skip = self.done.data_ptr()  # finished egos are not planned any more
        self.fb.launch_order = None  # (the egos' states move on: the order the ctx learns from its own launches follows them, the upload's hint would not)
        self.io = _abi.FpLoopIo()
        self.io.ego, self.io.t_now = dbatch.t["ego"].data_ptr(), dbatch.t["t_now"].data_ptr()
        self.io.done, self.io.cycles = self.done.data_ptr(), self.cycles.data_ptr()
        self.io.goal_xy, self.io.cart_state = self.goal.data_ptr(), self.cart.data_ptr()
        if goal_poly is not None:
            gp = np.ascontiguousarray(goal_poly, dtype=np.float64).reshape(B, -1, 2)
            self.goal_poly = torch.from_numpy(gp).to(dbatch.dev)
            self.goal_nv = torch.from_numpy(np.ascontiguousarray(goal_nv, dtype=np.int32).reshape(B)).to(dbatch.dev)
            self.io.goal_poly, self.io.goal_nv, self.io.goal_max_vertices = self.goal_poly.data_ptr(), self.goal_nv.data_ptr(), gp.shape[1]
            if goal_intervals is not None:
                self.goal_iv = torch.from_numpy(np.ascontiguousarray(goal_intervals, dtype=np.float64).reshape(B, 6)).to(dbatch.dev)
                self.io.goal_intervals = self.goal_iv.data_ptr()
        if planner != "FOP":
            self.prev = torch.full((B, 3), -1, dtype=i32, device=dbatch.dev)
            self.ijk = dbatch.empty((B, 3), i32)
            self.end_state = dbatch.empty((B, 3), f64)
            self.refined = dbatch.empty(B, i32)
            self.fopts = _abi.FpFissOpts(_abi.FP_FISS_PLUS if planner == "FISS+" else _abi.FP_FISS, 3 if planner == "FISS+" else 0, 10.0, 0.5)
            f = _abi.FpFissIo()
            f.samp_min, f.samp_max, f.samp_res = (dbatch.t[k].data_ptr() for k in ("samp_min", "samp_max", "samp_res"))
            f.prev_best_idx, f.best_ijk, f.best_cost, f.end_state = self.prev.data_ptr(), self.ijk.data_ptr(), self.best_cost.data_ptr(), self.end_state.data_ptr()
            f.refined, f.stats, f.trace, f.best_flags, f.best_traj = self.refined.data_ptr(), self.stats.data_ptr(), None, None, None
            self.fio = f

    def step(self, stream: int = 0):
        """One plan cycle for every running ego + the state hand-over, enqueued on `stream`."""
        import ctypes as C

        lib, ctx = self.eng._lib, self.eng._ctx
        if self.planner == "FOP" and self.fused:
            self.eng.plan_step_device(self.db.params, self.fb, self.io, self.best_idx.data_ptr(), self.best_cost.data_ptr(), self.stats.data_ptr(), stream=stream)
        elif self.planner == "FOP":
            self.eng.plan_dense_device(self.db.params, self.fb, self.best_idx.data_ptr(), self.best_cost.data_ptr(), self.stats.data_ptr(), stream=stream)
            _abi.check(lib.fp_advance(ctx, C.byref(self.db.params), C.byref(self.fb), self.best_idx.data_ptr(), None, C.byref(self.io),
                                      _abi.FP_MEM_DEVICE, stream or None))
        elif self.fused:  # fp_plan_fiss_step: FISS+ hands the egos over inside its refinement launch, FISS by the advance kernel behind the pipeline
            _abi.check(lib.fp_plan_fiss_step(ctx, C.byref(self.db.params), C.byref(self.fb), C.byref(self.fopts), C.byref(self.fio), C.byref(self.io),
                                             _abi.FP_MEM_DEVICE, stream or None))
        else:
            self.eng.plan_fiss_device(self.db.params, self.fb, self.fopts, self.fio, stream=stream)
            _abi.check(lib.fp_advance(ctx, C.byref(self.db.params), C.byref(self.fb), None, self.end_state.data_ptr(), C.byref(self.io),
                                      _abi.FP_MEM_DEVICE, stream or None))

    def run_graph(self, max_cycles: int):
        """The same loop as run(), but one [plan -> advance] cycle is captured into a HIP graph once and replayed: the cycle is
        launch-bound for small batches (2-5 short kernels), and every pointer it touches is fixed (state lives in HBM and is
        updated in place), so a replay needs no host work beyond hipGraphLaunch."""
        torch = self.db.torch
        self.step(torch.cuda.current_stream(self.db.dev).cuda_stream)  # warm-up outside capture: first-use allocations, LDS attributes
        torch.cuda.synchronize(self.db.dev)
        side = torch.cuda.Stream(self.db.dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            self.step(side.cuda_stream)
        torch.cuda.synchronize(self.db.dev)
        import time
        t0 = time.perf_counter()
        for _ in range(max_cycles - 1):
            graph.replay()
        torch.cuda.synchronize(self.db.dev)
        self.replay_seconds = time.perf_counter() - t0  # capture / instantiation excluded
        return SimpleNamespace(done=self.done.cpu().numpy(), cycles=self.cycles.cpu().numpy(), ego=self.db.t["ego"].cpu().numpy(),
                               t_now=self.db.t["t_now"].cpu().numpy(), cart=self.cart.cpu().numpy(), trace=[])

    def run(self, max_cycles: int, trace: bool = False):
        torch = self.db.torch
        stream = torch.cuda.current_stream(self.db.dev).cuda_stream
        rows = []
        for _ in range(max_cycles):
            if trace:
                start = self.db.t["ego"].cpu().numpy().copy()
            self.step(stream)
            if trace:
                rows.append(SimpleNamespace(start=start, cost=self.best_cost.cpu().numpy().copy(), stats=self.stats.cpu().numpy().copy(),
                                            done=self.done.cpu().numpy().copy(), cart=self.cart.cpu().numpy().copy()))
                if (rows[-1].done != 0).all():
                    break
        torch.cuda.synchronize(self.db.dev)
        return SimpleNamespace(done=self.done.cpu().numpy(), cycles=self.cycles.cpu().numpy(), ego=self.db.t["ego"].cpu().numpy(),
                               t_now=self.db.t["t_now"].cpu().numpy(), cart=self.cart.cpu().numpy(), trace=rows)
