"""FrenetEngine - thin Python handle on one fp_ctx (one GPU).

Two ways in, matching the two memory spaces of the C ABI:

* numpy (`plan_dense`, `eval_trajs`): host arrays; the library stages them through its own
  device arena, runs the kernels and copies the results back (FP_MEM_HOST).  This is what
  the drop-in planner classes use.
* device pointers (`plan_dense_device`, `eval_trajs_device`): the caller keeps the problem
  batch resident in HBM (e.g. torch tensors; only `.data_ptr()` crosses the boundary) and the
  calls just enqueue kernels on the given HIP stream (FP_MEM_DEVICE).  This is what bench.py
  and the multi-GPU shard runner use.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np

from . import _abi
from .batch import ProblemBatch

TRAJ_STRIDE = _abi.FP_DEFAULT_STRIDE

# CostFunction("WX1") weights, reference common/cost/cost_function.py:6-12 (w_T and w_D are unused there)
COST_WX1 = dict(cost_horizon=10.0, w_speed=1.0, w_accel=0.1, w_jerk=0.1, w_offset=10.0)


def make_params(batch, nd=None, nv=None, nt=None) -> _abi.FpParams:
    """fp_params of a batch.  A planner re-plans with the same batch object every cycle: the struct is cached on it, keyed on
    every scalar it is built from."""
    lim0 = getattr(batch, "curvature_limits", None)
    key = (nd or batch.nd, nv or batch.nv, nt or batch.nt, batch.check_stride, batch.tick_t, batch.veh_l, batch.veh_w, batch.max_speed, batch.max_accel,
           None if lim0 is None else tuple(lim0), _t_max(batch))  # (the VALUE points_max is derived from: an in-place edit of the arrays is seen)
    cached = getattr(batch, "__dict__", {}).get("_fp_cache")
    if cached is not None and cached[0] == key:
        return _abi.FpParams.from_buffer_copy(cached[1])  # a copy: callers may edit their struct
    p = _make_params(batch, nd, nv, nt)
    if hasattr(batch, "__dict__"):
        batch.__dict__["_fp_cache"] = (key, bytes(p))
    return p


def _t_max(batch) -> float:
    """The longest time horizon a call on this batch can sample: max(t_samples), and for the FISS planners the sampling box's upper edge."""
    t_max = float(np.max(batch.t_samples)) if len(batch.t_samples) else 0.0
    if getattr(batch, "samp_max", None) is not None and len(batch.samp_max):
        t_max = max(t_max, float(np.nanmax(batch.samp_max[:, 2])))
    return t_max


def _make_params(batch, nd=None, nv=None, nt=None) -> _abi.FpParams:
    p = _abi.FpParams()
    p.nd, p.nv, p.nt = nd or batch.nd, nv or batch.nv, nt or batch.nt
    p.check_stride = int(batch.check_stride)
    p.tick_t = float(batch.tick_t)
    p.cost_horizon, p.w_speed, p.w_accel, p.w_jerk, p.w_offset = (COST_WX1[k] for k in ("cost_horizon", "w_speed", "w_accel", "w_jerk", "w_offset"))
    p.veh_l, p.veh_w, p.max_speed, p.max_accel = float(batch.veh_l), float(batch.veh_w), float(batch.max_speed), float(batch.max_accel)
    # fp_params.points_max: what a FP_MEM_DEVICE call cannot see for itself - the points per trajectory (> 128: tick_t below 0.08 s)
    t_max = _t_max(batch)
    n_pts = int(np.ceil(t_max / float(batch.tick_t))) if t_max > 0 else 0
    p.points_max = min(n_pts, _abi.FP_MAX_POINTS) if n_pts > _abi.FP_FAST_POINTS else 0
    lim = getattr(batch, "curvature_limits", None)
    if lim is not None:  # optional checks of check_constraints (reference :145-150, commented out there)
        p.curvature_mask = 1
        p.max_curvature, p.max_kappa_d, p.max_kappa_dd = (float(v) for v in lim)
    return p


_BATCH_PTRS = ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
               "obs_pose", "obs_dims", "final_time_step")


def _ptr(a: np.ndarray) -> int:
    """Address of a contiguous array's data (several times cheaper than a.ctypes.data, which builds a helper object)."""
    return a.__array_interface__["data"][0]


def _host_batch(batch: ProblemBatch) -> _abi.FpBatch:
    """FpBatch over the batch's own numpy arrays.  A planner re-plans with the same (in-place updated) arrays every cycle, so
    the struct is cached on the batch and rebuilt only when one of the arrays was replaced."""
    arrays = [getattr(batch, name) for name in _BATCH_PTRS]
    poly = [batch.obs_poly, batch.obs_nvert] if getattr(batch, "obs_nvert", None) is not None else []
    tag = int(getattr(batch, "tables_tag", 0) or 0)  # fp_batch.tables_tag: the frame / scene tables stay on the device between calls
    key = tuple(map(id, arrays + poly)) + (tag,)
    cached = batch.__dict__.get("_fb_cache")
    if cached is not None and cached[0] == key:
        return _abi.FpBatch.from_buffer_copy(cached[1])  # a copy: callers may edit their struct
    fb = _abi.FpBatch()
    fb.B, fb.F, fb.NX, fb.S, fb.T_obs, fb.n_obs = batch.B, batch.F, batch.NX, batch.S, batch.T_obs, batch.n_obs
    for name, a in zip(_BATCH_PTRS, arrays):
        setattr(fb, name, _ptr(a) if a.size else None)
    fb.tables_tag = tag
    if poly and batch.S > 0 and batch.n_obs > 0:  # convex-polygon obstacle columns (fp_batch.obs_poly / obs_nvert)
        fb.obs_poly, fb.obs_nvert, fb.poly_stride = _ptr(poly[0]), _ptr(poly[1]), int(poly[0].shape[2])
    batch.__dict__["_fb_cache"] = (key, _abi.FpBatch.from_buffer_copy(fb), arrays + poly)  # the arrays are kept alive with the pointers
    return fb


def host_structs(batch: ProblemBatch, freeze: bool = False):
    """(fp_params, fp_batch) of a host batch.  freeze=True pins them on the batch: for a caller that owns the batch, never replaces
    its arrays and never changes its scalars (planners.py updates the start state in place) later calls skip even the cache checks."""
    st = batch.__dict__.get("_structs")
    if st is not None:
        return st
    st = (make_params(batch), _host_batch(batch))
    if freeze:
        batch.__dict__["_structs"] = st
    return st


def device_batch(sizes, ptrs: dict) -> _abi.FpBatch:
    """FpBatch from raw device addresses (ints).  sizes: object with B, F, NX, S, T_obs, n_obs."""
    fb = _abi.FpBatch()
    fb.B, fb.F, fb.NX, fb.S, fb.T_obs, fb.n_obs = sizes.B, sizes.F, sizes.NX, sizes.S, sizes.T_obs, sizes.n_obs
    for name in _BATCH_PTRS:
        setattr(fb, name, ptrs.get(name) or None)
    if ptrs.get("obs_nvert"):
        fb.obs_poly, fb.obs_nvert, fb.poly_stride = ptrs["obs_poly"], ptrs["obs_nvert"], int(sizes.poly_stride)
    fb.launch_order = ptrs.get("launch_order") or None  # (fp_batch.launch_order: the caller's launch-order hint, see launch_order_hint)
    return fb


def launch_order_hint(batch) -> np.ndarray:
    """fp_batch.launch_order for a resident batch: the egos by descending speed (ties in index order).  A launch of more egos than the
    device holds workgroups ends on the egos that start last, and an ego's work grows with its speed (a faster ego reaches more obstacle
    rows: more group-test survivors, longer walks) - measured on BASELINE config 3: 136.8 -> 130.9 us per 2048-ego step, the same as the
    order learnt from the batch's own durations.  Input-only, computed once at upload; results do not depend on it."""
    return np.argsort(-np.asarray(batch.ego)[:, 1], kind="stable").astype(np.int32)


def fiss_rounds(kind, max_refine_iters: int) -> int:
    """Refinement rounds a plan_fiss call runs (the rule FrenetEngine.plan_fiss applies): FISS+ by name or by its ABI constant."""
    return int(max_refine_iters) if kind in ("FISS+", _abi.FP_FISS_PLUS) else 0


def _check_out(out: SimpleNamespace, B: int, spec: dict) -> None:
    """Caller-provided output arrays cross the C ABI as bare addresses: every array the call writes must be a writable,
    C-contiguous array of the dtype and shape [B, ...] the engine would have allocated itself (a strided or wrong-dtype array
    would be written out of bounds by the C side)."""
    for k, want in spec.items():
        if want is None:
            continue
        dtype, tail = np.dtype(want[0]), (B,) + tuple(want[1])
        a = getattr(out, k, None)
        if not isinstance(a, np.ndarray) or a.shape != tail or a.dtype != dtype or not a.flags.c_contiguous or not a.flags.writeable:
            raise ValueError(f"out.{k}: need a writable C-contiguous {dtype} array of shape {tail}, got "
                             f"{type(a).__name__ if not isinstance(a, np.ndarray) else (str(a.dtype), a.shape, bool(a.flags.c_contiguous))}")


def unpack_flags(flags: np.ndarray):
    """flag word -> (bits, N, M)."""
    return flags & 0xFF, (flags >> _abi.FLAG_N_SHIFT) & 0xFFF, (flags >> _abi.FLAG_M_SHIFT) & 0xFFF


class FrenetEngine:
    def __init__(self, device: int = 0):
        self._lib = _abi.load()
        self._ctx = C.c_void_p()
        _abi.check(self._lib.fp_ctx_create(int(device), C.byref(self._ctx)))
        self.device = int(device)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.fp_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def set_option(self, name: str, value: int):
        """Diagnostic knobs of the ctx, e.g. set_option("lattice_kernel", 1) pins the lane-per-candidate kernel."""
        _abi.check(self._lib.fp_ctx_set_option(self._ctx, name.encode(), int(value)))

    def join(self, stream: int = 0):
        """set_option("overlap", 1): orders `stream` after every dense call of this engine that is still in flight on its internal streams
        (fp_ctx_join).  With overlap on, plan_dense_device returns with `stream` ordered after the PREVIOUS call's results only."""
        _abi.check(self._lib.fp_ctx_join(self._ctx, stream or None))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        _abi.check(self._lib.fp_ctx_get_option(self._ctx, name.encode(), C.byref(v)))
        return v.value

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ host arrays
    @staticmethod
    def dense_outputs(B: int, Cn: int, tables: bool = True, winner: bool = False, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False,
                      audit: bool = False):
        """Output arrays of plan_dense for B egos (ShardedEngine allocates them once and hands every shard its slice)."""
        return SimpleNamespace(
            audit=np.zeros(B, dtype=np.uint32) if audit else None,
            best_idx=np.empty(B, dtype=np.int32), best_cost=np.empty(B), stats=np.empty((B, 4), dtype=np.int32),
            cost=np.empty((B, Cn)) if tables else None, flags=np.empty((B, Cn), dtype=np.uint32) if tables else None,
            best_flags=np.empty(B, dtype=np.uint32) if winner else None,
            # sparse: the kernels write only the elements that exist; everything else keeps this NaN fill
            best_traj=(np.full((B, 16, traj_stride), np.nan) if traj_sparse else np.empty((B, 16, traj_stride))) if winner else None)

    def plan_dense(self, batch: ProblemBatch, tables: bool = True, winner: bool = False, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False,
                   out: SimpleNamespace | None = None, audit: bool = False):
        """FrenetOptimalPlanner.plan() for every ego of the batch (reference frenet_optimal_planner.py:247-270).

        Returns best_idx [B] (flat (i_d*nt+i_T)*nv+i_v, -1 = none), best_cost [B], stats [B,4] and, with
        tables=True, cost [B,C] and flags [B,C]; with winner=True also best_flags [B] and best_traj [B,16,128]
        (the argmin's full series, written by the lattice kernel itself).  audit=True: also `audit` [B] (FP_AUDIT_* bits: another
        feasible candidate within 1e-9 of the winner's cost - settled by point-by-point sums -, a collision verdict within 1e-9 m of
        contact; include/frenet_gpu.h).
        """
        B, Cn = batch.B, batch.C
        if out is None:  # (out: arrays of dense_outputs' shapes, e.g. contiguous slices of a bigger batch's outputs)
            out = self.dense_outputs(B, Cn, tables, winner, traj_stride, traj_sparse, audit)
        elif "_res" not in out.__dict__:  # caller's arrays, first use: they cross the ABI as bare addresses
            _check_out(out, B, dict(best_idx=("int32", ()), best_cost=("float64", ()), stats=("int32", (4,)),
                                    cost=("float64", (Cn,)) if tables else None, flags=("uint32", (Cn,)) if tables else None,
                                    best_flags=("uint32", ()) if winner else None, best_traj=("float64", (16, traj_stride)) if winner else None))
        if B == 0:
            return out
        # a caller that re-plans into the same `out` every cycle (planners.py) finds the fp_result of its arrays cached on it
        # (building the struct costs several microseconds of a ~55 us plan cycle); the arrays of `out` must not be replaced then
        key = (tables, winner, int(traj_stride), int(traj_sparse), bool(audit))
        cached = out.__dict__.get("_res")
        if cached is not None and cached[0] == key:
            res = cached[1]
        else:
            res = _abi.FpResult()
            res.best_idx, res.best_cost, res.stats = _ptr(out.best_idx), _ptr(out.best_cost), _ptr(out.stats)
            res.cost_tbl = _ptr(out.cost) if tables else None
            res.flag_tbl = _ptr(out.flags) if tables else None
            res.best_flags = _ptr(out.best_flags) if winner else None
            res.best_traj = _ptr(out.best_traj) if winner else None
            res.audit = _ptr(out.audit) if audit else None
            res.traj_stride, res.traj_sparse = int(traj_stride), int(traj_sparse)
            out.__dict__["_res"] = (key, res)
        p, fb = host_structs(batch)
        _abi.check(self._lib.fp_plan_dense(self._ctx, C.byref(p), C.byref(fb), C.byref(res), _abi.FP_MEM_HOST, None))
        return out

    def plan_fopplus(self, batch: ProblemBatch, winner: bool = False, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """FopPlusPlanner.plan() for every ego of the batch (fop_plus_planner.py:16-41) on the device: the cheapest feasible
        candidate + Stats = how many candidates the cost-ordered lazy validation pops before it.  Egos whose outcome hangs on an
        exact cost tie (mirror-symmetric start states) are replayed with the reference's heap order over their dense tables
        (search.fopplus_search); `replayed` lists them.  Same result object as plan_dense plus `fopplus` [B,2]."""
        from . import search

        B = batch.B
        out = self.dense_outputs(B, batch.C, False, winner, traj_stride, traj_sparse)
        out.fopplus = np.zeros((B, 2), dtype=np.int32)
        out.replayed = np.zeros(0, dtype=np.int64)
        if B == 0:
            return out
        res = _abi.FpResult()
        res.best_idx, res.best_cost, res.stats, res.fopplus = _ptr(out.best_idx), _ptr(out.best_cost), _ptr(out.stats), _ptr(out.fopplus)
        res.best_flags = _ptr(out.best_flags) if winner else None
        res.best_traj = _ptr(out.best_traj) if winner else None
        res.traj_stride, res.traj_sparse = int(traj_stride), int(traj_sparse)
        p = make_params(batch)
        fb = _host_batch(batch)
        _abi.check(self._lib.fp_plan_dense(self._ctx, C.byref(p), C.byref(fb), C.byref(res), _abi.FP_MEM_HOST, None))
        ties = np.nonzero(out.fopplus[:, 1])[0]
        if len(ties):  # exact ties: CPython's heap order decides, as in the reference
            sub = batch.take(ties)
            tab = self.plan_dense(sub, tables=True)
            for k, e in enumerate(ties):
                idx, st = search.fopplus_search(tab.cost[k], tab.flags[k])
                out.best_idx[e] = -1 if idx is None else idx
                out.best_cost[e] = np.nan if idx is None else tab.cost[k, idx]
                out.stats[e] = st
            if winner:
                w = self.winner_trajs(sub, out.best_idx[ties], traj_stride, traj_sparse)
                out.best_flags[ties] = w.best_flags
                out.best_traj[ties] = w.best_traj
            out.replayed = ties
        return out

    def eval_trajs(self, batch: ProblemBatch, end_states: np.ndarray, dump: bool = False, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """Explicit end states [B,K,3] = (d_end, v_end, T_end) -> cost [B,K], flags [B,K] (+ traj [B,K,16,stride])."""
        es = np.ascontiguousarray(end_states, dtype=np.float64)
        B, K = es.shape[0], es.shape[1]
        assert B == batch.B and es.shape[2] == 3
        cost = np.empty((B, K)); flags = np.empty((B, K), dtype=np.uint32)
        traj = (np.full((B, K, 16, traj_stride), np.nan) if traj_sparse else np.empty((B, K, 16, traj_stride))) if dump else None
        p = make_params(batch)
        fb = _host_batch(batch)
        _abi.check(self._lib.fp_eval_trajs(self._ctx, C.byref(p), C.byref(fb), K, es.ctypes.data, cost.ctypes.data, flags.ctypes.data,
                                           traj.ctypes.data if dump else None, int(traj_stride), int(traj_sparse), _abi.FP_MEM_HOST, None))
        return SimpleNamespace(cost=cost, flags=flags, traj=traj)

    @staticmethod
    def fiss_outputs(B: int, R: int, winner: bool = False, trace: bool = False, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """Output arrays of plan_fiss for B egos with R refinement rounds (prev_best_idx is the in / out history array)."""
        return SimpleNamespace(prev_best_idx=np.empty((B, 3), dtype=np.int32), best_ijk=np.empty((B, 3), dtype=np.int32), best_cost=np.empty(B),
                               end_state=np.empty((B, 3)), refined=np.empty(B, dtype=np.int32), stats=np.empty((B, 4), dtype=np.int32),
                               trace=np.empty((B, max(R, 1) * 7, 4)) if trace and R > 0 else None,
                               best_flags=np.empty(B, dtype=np.uint32) if winner else None,
                               best_traj=(np.full((B, 16, traj_stride), np.nan) if traj_sparse else np.empty((B, 16, traj_stride))) if winner else None)

    def plan_fiss(self, batch: ProblemBatch, kind: str = "FISS+", prev_best_idx: np.ndarray | None = None, w_heuristic: float = 10.0,
                  max_refine_iters: int = 3, decaying_factor: float = 0.5, winner: bool = False, trace: bool = False,
                  traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False, out: SimpleNamespace | None = None):
        """FissPlanner.plan / FissPlusPlanner.plan for every ego of the batch, entirely on the device (fp_plan_fiss):
        dense tables -> per-ego search walk -> (FISS+) refinement -> optional winner series.

        batch.d_samples must be the FISS lattice (max_road_width - w + 0.3) and batch.samp_min/max/res must be set.
        prev_best_idx [B,3] (-1 = None) is not modified; the updated copy is returned.
        """
        B = batch.B
        plus = kind in ("FISS+", _abi.FP_FISS_PLUS)
        R = fiss_rounds(kind, max_refine_iters)
        if out is None:
            out = self.fiss_outputs(B, R, winner, trace, traj_stride, traj_sparse)
        elif "_io" not in out.__dict__:
            _check_out(out, B, dict(prev_best_idx=("int32", (3,)), best_ijk=("int32", (3,)), best_cost=("float64", ()), end_state=("float64", (3,)),
                                    refined=("int32", ()), stats=("int32", (4,)), trace=("float64", (max(R, 1) * 7, 4)) if trace and R > 0 else None,
                                    best_flags=("uint32", ()) if winner else None, best_traj=("float64", (16, traj_stride)) if winner else None))
        out.prev_best_idx[...] = -1 if prev_best_idx is None else np.asarray(prev_best_idx, dtype=np.int32)
        prev = out.prev_best_idx
        if B == 0:
            return out
        okey = (plus, R, w_heuristic, decaying_factor)
        ocached = out.__dict__.get("_opts")
        if ocached is not None and ocached[0] == okey:
            opts = ocached[1]
        else:
            opts = _abi.FpFissOpts(_abi.FP_FISS_PLUS if plus else _abi.FP_FISS, R, w_heuristic, decaying_factor)
            out.__dict__["_opts"] = (okey, opts)
        # (a caller that re-plans into the same `out` with the same batch arrays finds its fp_fiss_io cached, like plan_dense's fp_result)
        key = (winner, int(traj_stride), int(traj_sparse), id(batch.samp_min), id(batch.samp_max), id(batch.samp_res))
        cached = out.__dict__.get("_io")
        if cached is not None and cached[0] == key:
            io = cached[1]
        else:
            io = _abi.FpFissIo()
            io.samp_min, io.samp_max, io.samp_res = _ptr(batch.samp_min), _ptr(batch.samp_max), _ptr(batch.samp_res)
            io.prev_best_idx, io.best_ijk, io.best_cost = _ptr(prev), _ptr(out.best_ijk), _ptr(out.best_cost)
            io.end_state, io.refined, io.stats = _ptr(out.end_state), _ptr(out.refined), _ptr(out.stats)
            io.trace = _ptr(out.trace) if out.trace is not None else None
            io.best_flags = _ptr(out.best_flags) if winner else None
            io.best_traj = _ptr(out.best_traj) if winner else None
            io.traj_stride, io.traj_sparse = int(traj_stride), int(traj_sparse)
            out.__dict__["_io"] = (key, io, batch.samp_min, batch.samp_max, batch.samp_res)  # (the arrays kept alive: their ids are the key)
        p, fb = host_structs(batch)
        _abi.check(self._lib.fp_plan_fiss(self._ctx, C.byref(p), C.byref(fb), C.byref(opts), C.byref(io), _abi.FP_MEM_HOST, None))
        return out

    def materialize_all(self, batch: ProblemBatch, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """Full series of every lattice candidate (fp_materialize_all): traj [B,C,16,traj_stride], flags [B,C] (N, M, truncated)."""
        B, Cn = batch.B, batch.C
        traj = np.full((B, Cn, 16, traj_stride), np.nan) if traj_sparse else np.empty((B, Cn, 16, traj_stride))
        flags = np.empty((B, Cn), dtype=np.uint32)
        p = make_params(batch)
        fb = _host_batch(batch)
        _abi.check(self._lib.fp_materialize_all(self._ctx, C.byref(p), C.byref(fb), flags.ctypes.data, traj.ctypes.data, int(traj_stride), int(traj_sparse),
                                                _abi.FP_MEM_HOST, None))
        return SimpleNamespace(traj=traj, flags=flags)

    def materialize_all_device(self, params: _abi.FpParams, fb: _abi.FpBatch, flags: int, traj: int, stream: int = 0,
                               traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        _abi.check(self._lib.fp_materialize_all(self._ctx, C.byref(params), C.byref(fb), flags, traj, int(traj_stride), int(traj_sparse),
                                                _abi.FP_MEM_DEVICE, stream or None))

    def build_frames(self, points: np.ndarray, n: np.ndarray | None = None):
        """CubicSpline2D construction for F centerlines on the GPU (fp_frames_build): points [F,NX,2] -> knots [F,NX], coef [F,8,NX]."""
        pts = np.ascontiguousarray(points, dtype=np.float64)
        if pts.ndim == 2:
            pts = pts[None]
        F, NX, _ = pts.shape
        n = np.full(F, NX, dtype=np.int32) if n is None else np.ascontiguousarray(n, dtype=np.int32)
        knots = np.empty((F, NX)); coef = np.empty((F, 8, NX))
        _abi.check(self._lib.fp_frames_build(self._ctx, F, NX, n.ctypes.data, pts.ctypes.data, knots.ctypes.data, coef.ctypes.data, _abi.FP_MEM_HOST, None))
        return knots, coef

    def from_state(self, knots: np.ndarray, coef: np.ndarray, nx: np.ndarray, frame_of: np.ndarray, states: np.ndarray) -> np.ndarray:
        """FrenetState.from_state for B Cartesian states [B,4] = x, y, yaw, v (fp_from_state) -> ego [B,6]."""
        st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 4)
        knots = np.ascontiguousarray(knots, dtype=np.float64); coef = np.ascontiguousarray(coef, dtype=np.float64)
        nx = np.ascontiguousarray(nx, dtype=np.int32); fo = np.ascontiguousarray(frame_of, dtype=np.int32)
        fb = _abi.FpBatch()
        fb.B, fb.F, fb.NX = st.shape[0], knots.shape[0], knots.shape[1]
        fb.frame_of, fb.nx, fb.knots, fb.coef = fo.ctypes.data, nx.ctypes.data, knots.ctypes.data, coef.ctypes.data
        ego = np.empty((st.shape[0], 6))
        _abi.check(self._lib.fp_from_state(self._ctx, C.byref(fb), st.ctypes.data, ego.ctypes.data, _abi.FP_MEM_HOST, None))
        return ego

    # ------------------------------------------------------------------ resident device memory
    def plan_dense_device(self, params: _abi.FpParams, fb: _abi.FpBatch, best_idx: int, best_cost: int, stats: int = 0,
                          cost_tbl: int = 0, flag_tbl: int = 0, stream: int = 0, best_flags: int = 0, best_traj: int = 0,
                          traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """Enqueue the dense pass on `stream`; every argument is a device address (int)."""
        res = _abi.FpResult()
        res.best_idx, res.best_cost = best_idx, best_cost
        res.stats, res.cost_tbl, res.flag_tbl = stats or None, cost_tbl or None, flag_tbl or None
        res.best_flags, res.best_traj = best_flags or None, best_traj or None
        res.traj_stride, res.traj_sparse = int(traj_stride), int(traj_sparse)
        _abi.check(self._lib.fp_plan_dense(self._ctx, C.byref(params), C.byref(fb), C.byref(res), _abi.FP_MEM_DEVICE, stream or None))

    def plan_step_device(self, params: _abi.FpParams, fb: _abi.FpBatch, io: _abi.FpLoopIo, best_idx: int, best_cost: int, stats: int = 0,
                         stream: int = 0, best_flags: int = 0, best_traj: int = 0, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """One cycle of the simulation loop (planning.py:120-162) for every running ego in ONE launch (fp_plan_step): the dense FOP
        pass, and the workgroup that finds an ego's argmin hands the ego over to its next state.  Device addresses throughout."""
        res = _abi.FpResult()
        res.best_idx, res.best_cost, res.stats = best_idx, best_cost, stats or None
        res.best_flags, res.best_traj = best_flags or None, best_traj or None
        res.traj_stride, res.traj_sparse = int(traj_stride), int(traj_sparse)
        _abi.check(self._lib.fp_plan_step(self._ctx, C.byref(params), C.byref(fb), C.byref(res), C.byref(io), _abi.FP_MEM_DEVICE, stream or None))

    def winner_trajs(self, batch: ProblemBatch, best_idx: np.ndarray, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """The standalone winner epilogue (fp_winner_trajs): series of lattice candidate best_idx[b] for every ego
        -> best_flags [B], best_traj [B,16,128] (NaN rows and flag 0 where best_idx < 0)."""
        bi = np.ascontiguousarray(best_idx, dtype=np.int32)
        out = SimpleNamespace(best_flags=np.empty(batch.B, dtype=np.uint32),
                              best_traj=np.full((batch.B, 16, traj_stride), np.nan) if traj_sparse else np.empty((batch.B, 16, traj_stride)))
        p = make_params(batch)
        fb = _host_batch(batch)
        _abi.check(self._lib.fp_winner_trajs(self._ctx, C.byref(p), C.byref(fb), _ptr(bi), _ptr(out.best_flags), _ptr(out.best_traj),
                                             int(traj_stride), int(traj_sparse), _abi.FP_MEM_HOST, None))
        return out

    def winner_trajs_device(self, params: _abi.FpParams, fb: _abi.FpBatch, best_idx: int, best_flags: int, best_traj: int, stream: int = 0,
                            traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        """Enqueue the winner epilogue alone (device addresses)."""
        _abi.check(self._lib.fp_winner_trajs(self._ctx, C.byref(params), C.byref(fb), best_idx, best_flags, best_traj,
                                             int(traj_stride), int(traj_sparse), _abi.FP_MEM_DEVICE, stream or None))

    def plan_fiss_device(self, params: _abi.FpParams, fb: _abi.FpBatch, opts: _abi.FpFissOpts, io: _abi.FpFissIo, stream: int = 0):
        """Enqueue the whole FISS / FISS+ pipeline (lattice, search, refinement, winner series); device addresses in `io`."""
        _abi.check(self._lib.fp_plan_fiss(self._ctx, C.byref(params), C.byref(fb), C.byref(opts), C.byref(io), _abi.FP_MEM_DEVICE, stream or None))

    def eval_trajs_device(self, params: _abi.FpParams, fb: _abi.FpBatch, K: int, end_states: int, cost: int, flags: int,
                          traj: int = 0, stream: int = 0, traj_stride: int = TRAJ_STRIDE, traj_sparse: bool = False):
        _abi.check(self._lib.fp_eval_trajs(self._ctx, C.byref(params), C.byref(fb), K, end_states, cost or None, flags or None,
                                           traj or None, int(traj_stride), int(traj_sparse), _abi.FP_MEM_DEVICE, stream or None))


def device_count() -> int:
    n = C.c_int(0)
    lib = _abi.load()
    rc = lib.fp_device_count(C.byref(n))
    return n.value if rc == 0 else 0
