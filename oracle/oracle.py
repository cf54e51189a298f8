"""ctypes binding of the CPU oracle (oracle/libfrenet_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports this module.

The binding is deliberately independent of the product package: it takes plain
numpy arrays with the batch layout documented in DESIGN.md ("problem batch").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRENET_ORACLE_LIB: load another build of the same source instead (the sanitizer build, tests/test_oracle_sanitize.py)
_LIB_PATH = os.environ.get("FRENET_ORACLE_LIB") or os.path.join(_HERE, "libfrenet_oracle.so")

FLAG_SPEED, FLAG_ACCEL, FLAG_COLLISION, FLAG_TRUNCATED = 1, 2, 4, 8
FLAG_CURVATURE, FLAG_KAPPA_D, FLAG_KAPPA_DD = 16, 32, 64
FLAG_CONSTRAINTS = 1 | 2 | 16 | 32 | 64
FLAG_INFEASIBLE = FLAG_CONSTRAINTS | 4
ARRAY_NAMES = ["t", "s", "s_d", "s_dd", "s_ddd", "d", "d_d", "d_dd", "d_ddd", "x", "y", "yaw", "ds", "c", "c_d", "c_dd"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint32)


class OrcProblem(C.Structure):
    _fields_ = [
        ("nd", C.c_int32), ("nv", C.c_int32), ("nt", C.c_int32),
        ("d_samples", _dp), ("v_samples", _dp), ("t_samples", _dp),
        ("tick_t", C.c_double), ("target_speed", C.c_double),
        ("samp_min", C.c_double * 3), ("samp_max", C.c_double * 3), ("samp_res", C.c_double * 3),
        ("veh_l", C.c_double), ("veh_w", C.c_double), ("max_speed", C.c_double), ("max_accel", C.c_double),
        ("ego", C.c_double * 6),
        ("nx", C.c_int32), ("knots", _dp), ("coef_x", _dp), ("coef_y", _dp),
        ("n_obs", C.c_int32), ("T_obs", C.c_int32), ("obs_pose", _dp), ("obs_dims", _dp),
        ("final_time_step", C.c_int32), ("t_now", C.c_int32), ("check_stride", C.c_int32),
        ("curvature_mask", C.c_int32), ("max_curvature", C.c_double), ("max_kappa_d", C.c_double), ("max_kappa_dd", C.c_double),
        ("obs_poly", _dp), ("obs_nvert", _ip), ("poly_stride", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make -C oracle)."""
    src = os.path.join(_HERE, "frenet_oracle.c")
    if os.environ.get("FRENET_ORACLE_LIB"):
        return _LIB_PATH  # somebody else's build
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_version.restype = C.c_int
        L.orc_quintic_coefs.argtypes = [C.c_double] * 7 + [_dp]
        L.orc_quartic_coefs.argtypes = [C.c_double] * 6 + [_dp]
        L.orc_poly_eval.argtypes = [_dp, C.c_int, C.c_double, _dp]
        L.orc_poly_eval.restype = None
        L.orc_spline1d_build.argtypes = [C.c_int32, _dp, _dp, _dp]
        L.orc_spline2d_build.argtypes = [C.c_int32, _dp, _dp, _dp, _dp, _dp]
        L.orc_spline2d_eval.argtypes = [C.c_int32, _dp, _dp, _dp, C.c_double, _dp]
        PP = C.POINTER(OrcProblem)
        L.orc_eval_traj.argtypes = [PP, C.c_double, C.c_double, C.c_double, C.c_int, _dp, C.c_int32, _ip, _ip, _dp, _up]
        L.orc_dense_tables.argtypes = [PP, _dp, _up]
        L.orc_fop_plan.argtypes = [PP, _ip, _dp, _ip, _dp, _up]
        L.orc_goal_reached.argtypes = [_dp, C.c_int32, _dp, C.c_double, C.c_double, C.c_int32, C.c_double, C.c_double]
        L.orc_point_in_polygon_closed.argtypes = [_dp, C.c_int32, C.c_double, C.c_double]
        L.orc_fopplus_plan.argtypes = [PP, _ip, _dp, _ip]
        L.orc_fiss_plan.argtypes = [PP, C.c_double, _ip, _ip, _dp, _ip]
        L.orc_fissplus_plan.argtypes = [PP, C.c_double, C.c_int32, C.c_double, _ip, _ip, _dp, _ip, _ip, _dp, _dp]
        L.orc_fiss_cost_est.argtypes = [PP, C.c_double, _ip, _dp]
        L.orc_from_state.argtypes = [_dp, C.c_int32, _dp, C.c_int32, _dp]
        L.orc_fop_plan_batch.argtypes = [PP, C.c_int32, C.c_int32, _ip, _dp]
        L.orc_boxes_intersect.argtypes = [C.c_double] * 10
        L.orc_boxes_intersect.restype = C.c_int
        L.orc_boxes_intersect_exact.argtypes = [C.c_double] * 10
        L.orc_boxes_intersect_exact.restype = C.c_int
        L.orc_boxes_intersect_batch.argtypes = [C.c_int32, _dp, _dp, C.c_int32, C.c_int32, C.POINTER(C.c_int8)]
        L.orc_box_vertices.argtypes = [C.c_double] * 5 + [_dp]
        L.orc_box_ring_intersect.argtypes = [C.c_double] * 5 + [_dp, C.c_int32] + [C.c_double] * 3 + [C.c_int32, _dp]
        L.orc_box_ring_intersect.restype = C.c_int
        _lib = L
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


# ---------------------------------------------------------------------------
# small wrappers
# ---------------------------------------------------------------------------
def quintic_coefs(xs, vxs, axs, xe, vxe, axe, T):
    a = np.empty(6)
    rc = lib().orc_quintic_coefs(xs, vxs, axs, xe, vxe, axe, T, _p(a))
    if rc:
        raise np.linalg.LinAlgError("Singular matrix")
    return a


def quartic_coefs(xs, vxs, axs, vxe, axe, T):
    a = np.empty(5)
    rc = lib().orc_quartic_coefs(xs, vxs, axs, vxe, axe, T, _p(a))
    if rc:
        raise np.linalg.LinAlgError("Singular matrix")
    return a


def poly_eval(a, t):
    a = _f64(a)
    out = np.empty(4)
    lib().orc_poly_eval(_p(a), len(a) - 1, float(t), _p(out))
    return out


def spline2d_build(px, py):
    px, py = _f64(px), _f64(py)
    n = len(px)
    knots = np.empty(n)
    cx = np.empty((4, n))
    cy = np.empty((4, n))
    rc = lib().orc_spline2d_build(n, _p(px), _p(py), _p(knots), _p(cx), _p(cy))
    if rc:
        raise ValueError(f"orc_spline2d_build rc={rc}")
    return knots, cx, cy


def spline2d_eval(knots, cx, cy, s):
    """-> (x, y, yaw, kappa) or None when s is out of range."""
    out = np.empty(4)
    rc = lib().orc_spline2d_eval(len(knots), _p(knots), _p(cx), _p(cy), float(s), _p(out))
    return None if rc else out


def boxes_intersect(box_a, box_b):
    """The collision primitive: boxes are (length, width, x, y, yaw).  True / False, or None when a polygon cannot be built."""
    rc = lib().orc_boxes_intersect(*[float(v) for v in box_a], *[float(v) for v in box_b])
    return None if rc < 0 else bool(rc)


def boxes_intersect_exact(box_a, box_b):
    """The same polygons decided exactly (binary128 orientation signs on the fp64 vertex coordinates)."""
    rc = lib().orc_boxes_intersect_exact(*[float(v) for v in box_a], *[float(v) for v in box_b])
    return None if rc < 0 else bool(rc)


def boxes_intersect_batch(a, b, exact=False, threads=1):
    """a, b: [n, 5] boxes (l, w, x, y, yaw) -> int8 [n]: 1 / 0 / -1 (a polygon could not be built)."""
    a, b = _f64(a).reshape(-1, 5), _f64(b).reshape(-1, 5)
    out = np.empty(len(a), dtype=np.int8)
    lib().orc_boxes_intersect_batch(len(a), _p(a), _p(b), int(exact), int(threads), out.ctypes.data_as(C.POINTER(C.c_int8)))
    return out


def box_vertices(box):
    out = np.empty(8)
    rc = lib().orc_box_vertices(*[float(v) for v in box], _p(out))
    return None if rc else out.reshape(4, 2)


def box_ring_intersect(box, ring, pose, exact=False, world=False):
    """Ego box (l, w, x, y, yaw) against a convex counter-clockwise ring [n, 2] (relative to its rotation centre) at pose (x, y, yaw):
    construct_polygon + Polygon.intersects for a polygon obstacle.  True / False / None; world=True also returns the ring's world
    coordinates [n, 2]."""
    u = _f64(ring).reshape(-1, 2)
    w = np.empty_like(u)
    rc = lib().orc_box_ring_intersect(*[float(v) for v in box], _p(u), len(u), *[float(v) for v in pose], int(exact), _p(w))
    res = None if rc < 0 else bool(rc)
    return (res, w) if world else res


def goal_reached(poly, x, y, time_step=0, velocity=0.0, orientation=0.0, intervals=None):
    """goal_region.is_reached for one goal state: poly [n, 2], intervals None or [6] (time_step / velocity / orientation lo, hi; NaN = undefined)."""
    pl = _f64(poly).reshape(-1, 2)
    iv = None if intervals is None else _f64(intervals).reshape(6)
    return bool(lib().orc_goal_reached(_p(pl), len(pl), None if iv is None else _p(iv), float(x), float(y), int(time_step), float(velocity), float(orientation)))


def from_state(x, y, yaw, v, polyline):
    pl = _f64(polyline)
    st = _f64([x, y, yaw, v])
    out = np.empty(6)
    lib().orc_from_state(_p(st), pl.shape[0], _p(pl), pl.shape[1], _p(out))
    return out


# ---------------------------------------------------------------------------
# problems
# ---------------------------------------------------------------------------
class Problem:
    """Owns the numpy buffers an OrcProblem points into."""

    def __init__(self, *, d_samples, v_samples, t_samples, tick_t, target_speed, veh_l, veh_w, max_speed, max_accel,
                 ego, knots, coef_x, coef_y, obs_pose=None, obs_dims=None, final_time_step=0, t_now=0, check_stride=2,
                 samp_min=None, samp_max=None, samp_res=None, curvature_limits=None, obs_poly=None, obs_nvert=None):
        self._keep = k = SimpleNamespace()
        k.d = _f64(d_samples); k.v = _f64(v_samples); k.t = _f64(t_samples)
        k.knots = _f64(knots); k.cx = _f64(coef_x); k.cy = _f64(coef_y)
        nx = len(k.knots)
        assert k.cx.shape == (4, nx) and k.cy.shape == (4, nx)
        if obs_pose is None or np.size(obs_pose) == 0:
            k.pose = np.zeros((0, 0, 4)); k.dims = np.zeros((0, 2))
        else:
            k.pose = _f64(obs_pose); k.dims = _f64(obs_dims)
        P = OrcProblem()
        P.nd, P.nv, P.nt = len(k.d), len(k.v), len(k.t)
        P.d_samples, P.v_samples, P.t_samples = _p(k.d), _p(k.v), _p(k.t)
        P.tick_t = tick_t; P.target_speed = target_speed
        smin = samp_min if samp_min is not None else [k.d[0], k.v[0], k.t[0]]
        smax = samp_max if samp_max is not None else [k.d[-1], k.v[-1], k.t[-1]]
        if samp_res is None:
            samp_res = [(a[-1] - a[0]) / (len(a) - 1) if len(a) > 1 else np.nan for a in (k.d, k.v, k.t)]
        for i in range(3):
            P.samp_min[i] = smin[i]; P.samp_max[i] = smax[i]; P.samp_res[i] = samp_res[i]
        P.veh_l, P.veh_w, P.max_speed, P.max_accel = veh_l, veh_w, max_speed, max_accel
        for i, e in enumerate(np.asarray(ego, dtype=float).reshape(6)):
            P.ego[i] = e
        P.nx = nx; P.knots = _p(k.knots); P.coef_x = _p(k.cx); P.coef_y = _p(k.cy)
        P.T_obs, P.n_obs = (k.pose.shape[0], k.pose.shape[1]) if k.pose.size else (0, 0)
        P.obs_pose = _p(k.pose); P.obs_dims = _p(k.dims)
        P.final_time_step = int(final_time_step); P.t_now = int(t_now); P.check_stride = int(check_stride)
        if curvature_limits is not None:  # (max_curvature, max_kappa_d, max_kappa_dd): turns the optional checks on
            P.curvature_mask = 1
            P.max_curvature, P.max_kappa_d, P.max_kappa_dd = (float(v) for v in curvature_limits)
        if obs_nvert is not None and k.pose.size:  # convex-polygon obstacle columns: [n_obs, PV, 2] rings + [n_obs] vertex counts
            k.poly = _f64(obs_poly); k.nvert = np.ascontiguousarray(obs_nvert, dtype=np.int32)
            assert k.nvert.shape == (P.n_obs,) and k.poly.shape[0] == P.n_obs and k.poly.shape[2] == 2
            P.obs_poly = _p(k.poly); P.obs_nvert = k.nvert.ctypes.data_as(_ip); P.poly_stride = k.poly.shape[1]
        self.c = P

    @property
    def C(self):
        return self.c.nd * self.c.nv * self.c.nt

    # ---- single trajectory
    def eval_traj(self, d_end, v_end, T_end, collision=True, dump=False, stride=128):
        N = C.c_int32(); M = C.c_int32(); cost = C.c_double(); flags = C.c_uint32()
        arr = np.empty((16, stride)) if dump else None
        rc = lib().orc_eval_traj(C.byref(self.c), d_end, v_end, T_end, int(collision), _p(arr) if dump else None,
                                 stride if dump else 0, C.byref(N), C.byref(M), C.byref(cost), C.byref(flags))
        if rc:
            raise ValueError(f"orc_eval_traj rc={rc}")
        out = SimpleNamespace(N=N.value, M=M.value, cost=cost.value, flags=flags.value)
        if dump:
            out.arrays = arr
        return out

    def dense_tables(self):
        cost = np.empty(self.C); flags = np.empty(self.C, dtype=np.uint32)
        rc = lib().orc_dense_tables(C.byref(self.c), _p(cost), flags.ctypes.data_as(_up))
        if rc:
            raise ValueError(f"orc_dense_tables rc={rc}")
        return cost, flags

    def fop_plan(self):
        bi = C.c_int32(); bc = C.c_double(); st = np.zeros(4, dtype=np.int32)
        cost = np.empty(self.C); flags = np.empty(self.C, dtype=np.uint32)
        rc = lib().orc_fop_plan(C.byref(self.c), C.byref(bi), C.byref(bc), st.ctypes.data_as(_ip), _p(cost),
                                flags.ctypes.data_as(_up))
        if rc:
            raise ValueError(f"orc_fop_plan rc={rc}")
        return SimpleNamespace(best_idx=bi.value, best_cost=bc.value, stats=st, cost=cost, flags=flags)

    def fopplus_plan(self):
        bi = C.c_int32(); bc = C.c_double(); st = np.zeros(4, dtype=np.int32)
        rc = lib().orc_fopplus_plan(C.byref(self.c), C.byref(bi), C.byref(bc), st.ctypes.data_as(_ip))
        if rc:
            raise ValueError(f"orc_fopplus_plan rc={rc}")
        return SimpleNamespace(best_idx=bi.value, best_cost=bc.value, stats=st)

    def fiss_plan(self, prev_best_idx=None, w_heuristic=10.0):
        prev = np.array([-1, -1, -1] if prev_best_idx is None else prev_best_idx, dtype=np.int32)
        ijk = np.zeros(3, dtype=np.int32); bc = C.c_double(); st = np.zeros(4, dtype=np.int32)
        rc = lib().orc_fiss_plan(C.byref(self.c), w_heuristic, prev.ctypes.data_as(_ip), ijk.ctypes.data_as(_ip),
                                 C.byref(bc), st.ctypes.data_as(_ip))
        if rc:
            raise ValueError(f"orc_fiss_plan rc={rc}")
        return SimpleNamespace(best_ijk=ijk, best_cost=bc.value, stats=st, prev_best_idx=prev)

    def fissplus_plan(self, prev_best_idx=None, w_heuristic=10.0, max_refine_iters=3, decaying_factor=0.5):
        prev = np.array([-1, -1, -1] if prev_best_idx is None else prev_best_idx, dtype=np.int32)
        ijk = np.zeros(3, dtype=np.int32); bc = C.c_double(); st = np.zeros(4, dtype=np.int32)
        refined = C.c_int32(); end_state = np.empty(3); trace = np.empty((max(max_refine_iters, 1), 7, 4))
        rc = lib().orc_fissplus_plan(C.byref(self.c), w_heuristic, max_refine_iters, decaying_factor,
                                     prev.ctypes.data_as(_ip), ijk.ctypes.data_as(_ip), C.byref(bc),
                                     st.ctypes.data_as(_ip), C.byref(refined), _p(end_state), _p(trace))
        if rc:
            raise ValueError(f"orc_fissplus_plan rc={rc}")
        return SimpleNamespace(best_ijk=ijk, best_cost=bc.value, stats=st, prev_best_idx=prev, refined=bool(refined.value),
                               end_state=end_state, trace=trace)

    def fiss_cost_est(self, prev_best_idx=None, w_heuristic=10.0):
        prev = np.array([-1, -1, -1] if prev_best_idx is None else prev_best_idx, dtype=np.int32)
        est = np.empty((self.c.nd, self.c.nv, self.c.nt))
        lib().orc_fiss_cost_est(C.byref(self.c), w_heuristic, prev.ctypes.data_as(_ip), _p(est))
        return est


def problems_from_batch(batch, egos=None, d_samples=None):
    """Build oracle Problems from a "problem batch" (any object with the
    attributes documented in DESIGN.md: settings/vehicle scalars + arrays)."""
    idx = range(batch.B) if egos is None else egos
    out = []
    for b in idx:
        f = int(batch.frame_of[b]); sc = int(batch.scene_of[b])
        nx = int(batch.nx[f])
        coef = batch.coef[f]
        has_obs = sc >= 0 and batch.n_obs > 0
        out.append(Problem(
            d_samples=batch.d_samples if d_samples is None else d_samples, v_samples=batch.v_samples[b], t_samples=batch.t_samples,
            tick_t=batch.tick_t, target_speed=float(batch.target_speed[b]),
            veh_l=batch.veh_l, veh_w=batch.veh_w, max_speed=batch.max_speed, max_accel=batch.max_accel,
            ego=batch.ego[b], knots=batch.knots[f, :nx], coef_x=coef[0:4, :nx], coef_y=coef[4:8, :nx],
            obs_pose=batch.obs_pose[sc] if has_obs else None, obs_dims=batch.obs_dims[sc] if has_obs else None,
            final_time_step=int(batch.final_time_step[sc]) if has_obs else 0, t_now=int(batch.t_now[b]),
            check_stride=batch.check_stride,
            samp_min=batch.samp_min[b] if getattr(batch, "samp_min", None) is not None else None,
            samp_max=batch.samp_max[b] if getattr(batch, "samp_max", None) is not None else None,
            samp_res=batch.samp_res[b] if getattr(batch, "samp_res", None) is not None else None,
            curvature_limits=getattr(batch, "curvature_limits", None),
            obs_poly=batch.obs_poly[sc] if has_obs and getattr(batch, "obs_nvert", None) is not None else None,
            obs_nvert=batch.obs_nvert[sc] if has_obs and getattr(batch, "obs_nvert", None) is not None else None))
    return out


_fast = None


def fast_lib():
    """libfrenet_oracle_fast.so (-DORC_FAST, -O3): the labelled faster CPU baseline of bench.py; only orc_fop_plan_batch is bound."""
    global _fast
    if _fast is None:
        path = os.path.join(_HERE, "libfrenet_oracle_fast.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "-s", "fast"])
        L = C.CDLL(path)
        L.orc_fop_plan_batch.argtypes = [C.POINTER(OrcProblem), C.c_int32, C.c_int32, _ip, _dp]
        _fast = L
    return _fast


def fop_plan_batch(problems, threads=1, fast=False):
    B = len(problems)
    arr = (OrcProblem * B)(*[p.c for p in problems])
    bi = np.empty(B, dtype=np.int32); bc = np.empty(B)
    rc = (fast_lib() if fast else lib()).orc_fop_plan_batch(arr, B, threads, bi.ctypes.data_as(_ip), _p(bc))
    if rc:
        raise ValueError(f"orc_fop_plan_batch rc={rc}")
    return bi, bc
