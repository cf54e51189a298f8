"""ctypes binding of libfrenetgpu.so (include/frenet_gpu.h).

There is no CPU implementation behind this module: loading fails loudly when the
HIP library has not been built, and every call fails when no GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfrenetgpu.so")

FP_ABI_VERSION = 15
FP_FISS, FP_FISS_PLUS = 0, 1
FP_MEM_HOST, FP_MEM_DEVICE = 0, 1
FP_MAX_POINTS, FP_MAX_KNOTS, FP_MAX_CAND, FP_MAX_POLY_VERTS = 256, 1024, 16384, 128
FP_FAST_POINTS, FP_DEFAULT_STRIDE = 128, 128  # the fast paths' points per trajectory; columns of a series row when traj_stride = 0
FP_MAX_CAND_SEARCH = 4096  # device-side FISS / FISS+ walks (fp_plan_fiss)
FLAG_SPEED, FLAG_ACCEL, FLAG_COLLISION, FLAG_TRUNCATED = 1, 2, 4, 8
FLAG_CURVATURE, FLAG_KAPPA_D, FLAG_KAPPA_DD = 16, 32, 64   # optional checks (fp_params.curvature_mask)
FLAG_CONSTRAINTS = FLAG_SPEED | FLAG_ACCEL | FLAG_CURVATURE | FLAG_KAPPA_D | FLAG_KAPPA_DD
FLAG_INFEASIBLE = FLAG_CONSTRAINTS | FLAG_COLLISION
FLAG_N_SHIFT, FLAG_M_SHIFT = 8, 20

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint32)

# every symbol include/frenet_gpu.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = ("fp_abi_version", "fp_build_flags", "fp_build_compiler", "fp_last_error", "fp_device_count", "fp_device_info", "fp_ctx_create", "fp_ctx_destroy", "fp_ctx_set_option", "fp_ctx_get_option", "fp_ctx_join",
                    "fp_plan_dense", "fp_winner_trajs", "fp_eval_trajs", "fp_plan_fiss", "fp_advance", "fp_plan_step", "fp_plan_fiss_step", "fp_frames_build", "fp_from_state", "fp_materialize_all",
                    "fp_group_create", "fp_group_destroy", "fp_group_submit", "fp_group_wait")


class FpParams(C.Structure):
    _fields_ = [("nd", C.c_int32), ("nv", C.c_int32), ("nt", C.c_int32), ("check_stride", C.c_int32),
                ("tick_t", C.c_double), ("cost_horizon", C.c_double),
                ("w_speed", C.c_double), ("w_accel", C.c_double), ("w_jerk", C.c_double), ("w_offset", C.c_double),
                ("veh_l", C.c_double), ("veh_w", C.c_double), ("max_speed", C.c_double), ("max_accel", C.c_double),
                ("curvature_mask", C.c_int32), ("points_max", C.c_int32),
                ("max_curvature", C.c_double), ("max_kappa_d", C.c_double), ("max_kappa_dd", C.c_double)]


class FpBatch(C.Structure):
    _fields_ = [("B", C.c_int32), ("F", C.c_int32), ("NX", C.c_int32), ("S", C.c_int32), ("T_obs", C.c_int32), ("n_obs", C.c_int32),
                ("d_samples", C.c_void_p), ("t_samples", C.c_void_p), ("v_samples", C.c_void_p), ("target_speed", C.c_void_p),
                ("ego", C.c_void_p), ("frame_of", C.c_void_p), ("scene_of", C.c_void_p), ("t_now", C.c_void_p),
                ("nx", C.c_void_p), ("knots", C.c_void_p), ("coef", C.c_void_p),
                ("obs_pose", C.c_void_p), ("obs_dims", C.c_void_p), ("final_time_step", C.c_void_p), ("skip", C.c_void_p),
                ("tables_tag", C.c_int32), ("poly_stride", C.c_int32), ("obs_poly", C.c_void_p), ("obs_nvert", C.c_void_p),
                ("launch_order", C.c_void_p)]


class FpResult(C.Structure):
    _fields_ = [("best_idx", C.c_void_p), ("best_cost", C.c_void_p), ("cost_tbl", C.c_void_p), ("flag_tbl", C.c_void_p),
                ("stats", C.c_void_p), ("best_flags", C.c_void_p), ("best_traj", C.c_void_p), ("fopplus", C.c_void_p), ("audit", C.c_void_p),
                ("traj_stride", C.c_int32), ("traj_sparse", C.c_int32)]


AUDIT_NEAR_TIE, AUDIT_CONTACT, AUDIT_REORDERED, AUDIT_TIES_OVERFLOW = 1, 2, 4, 8


class FpFissOpts(C.Structure):
    _fields_ = [("kind", C.c_int32), ("max_refine_iters", C.c_int32), ("w_heuristic", C.c_double), ("decaying_factor", C.c_double)]


class FpFissIo(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("samp_min", "samp_max", "samp_res", "prev_best_idx", "best_ijk", "best_cost", "end_state",
                                          "refined", "stats", "trace", "best_flags", "best_traj")] + \
               [("traj_stride", C.c_int32), ("traj_sparse", C.c_int32)]


class FpLoopIo(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ego", "t_now", "done", "cycles", "goal_xy", "cart_state", "goal_poly", "goal_nv", "goal_intervals")] + \
               [("goal_max_vertices", C.c_int32), ("reserved0", C.c_int32)]


RUNNING, DONE_GOAL, DONE_END_OF_LINE, DONE_NO_SOLUTION, DONE_GOAL_REGION = 0, 1, 2, 3, 4


class FpCopy(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("bytes", C.c_size_t)]


class FpShardCall(C.Structure):
    """One shard's call of an fp_group round (include/frenet_gpu.h: which entry point runs follows from the pointers that are set)."""
    _fields_ = [("params", C.POINTER(FpParams)), ("batch", C.POINTER(FpBatch)), ("result", C.POINTER(FpResult)), ("loop", C.POINTER(FpLoopIo)),
                ("fiss_opts", C.POINTER(FpFissOpts)), ("fiss_io", C.POINTER(FpFissIo)), ("stream", C.c_void_p),
                ("copies", C.POINTER(FpCopy)), ("n_copies", C.c_int32), ("reserved0", C.c_int32)]


class FrenetGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libfrenetgpu error {code}: {msg}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load libfrenetgpu.so (built by `make -C fiss_plus_planner_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C fiss_plus_planner_amd/csrc).  There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own HIP runtime.  If torch is going to be used in this process (device tensors as
    # plumbing), its runtime must be the one both sides share: load it first, otherwise a later `torch.cuda` init finds the
    # system runtime already initialised and reports "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.fp_abi_version.restype = C.c_int
    L.fp_last_error.restype = C.c_char_p
    L.fp_build_flags.restype = C.c_char_p
    L.fp_build_compiler.restype = C.c_char_p
    flags = (L.fp_build_flags() or b"").decode()
    if flags and not os.environ.get("FP_ALLOW_DIAGNOSTIC_BUILD"):
        raise ImportError(
            f"{LIB_PATH} is a DIAGNOSTIC build ({flags}): timing ablations and stamp / counter builds share the production file name and ABI "
            "version, and some of them produce wrong results by design.  Rebuild without EXTRA (make -B -C fiss_plus_planner_amd/csrc), or set "
            "FP_ALLOW_DIAGNOSTIC_BUILD=1 if this is a profiling run.")
    L.fp_device_count.argtypes = [C.POINTER(C.c_int)]
    L.fp_device_info.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.fp_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.fp_ctx_destroy.argtypes = [C.c_void_p]
    L.fp_ctx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.fp_ctx_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
    L.fp_ctx_join.argtypes = [C.c_void_p, C.c_void_p]
    L.fp_plan_dense.argtypes = [C.c_void_p, C.POINTER(FpParams), C.POINTER(FpBatch), C.POINTER(FpResult), C.c_int, C.c_void_p]
    L.fp_winner_trajs.argtypes = [C.c_void_p, C.POINTER(FpParams), C.POINTER(FpBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                  C.c_int, C.c_void_p]
    L.fp_eval_trajs.argtypes = [C.c_void_p, C.POINTER(FpParams), C.POINTER(FpBatch), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_void_p]
    L.fp_plan_fiss.argtypes = [C.c_void_p, C.POINTER(FpParams), C.POINTER(FpBatch), C.POINTER(FpFissOpts), C.POINTER(FpFissIo), C.c_int, C.c_void_p]
    L.fp_advance.argtypes = [C.c_void_p, C.POINTER(FpParams), C.POINTER(FpBatch), C.c_void_p, C.c_void_p, C.POINTER(FpLoopIo), C.c_int, C.c_void_p]
    L.fp_plan_step.argtypes = [C.c_void_p, C.POINTER(FpParams), C.POINTER(FpBatch), C.POINTER(FpResult), C.POINTER(FpLoopIo), C.c_int, C.c_void_p]
    L.fp_plan_fiss_step.argtypes = [C.c_void_p, C.POINTER(FpParams), C.POINTER(FpBatch), C.POINTER(FpFissOpts), C.POINTER(FpFissIo), C.POINTER(FpLoopIo), C.c_int, C.c_void_p]
    L.fp_frames_build.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.fp_from_state.argtypes = [C.c_void_p, C.POINTER(FpBatch), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.fp_materialize_all.argtypes = [C.c_void_p, C.POINTER(FpParams), C.POINTER(FpBatch), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_void_p]
    L.fp_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]
    L.fp_group_destroy.argtypes = [C.c_void_p]
    L.fp_group_submit.argtypes = [C.c_void_p, C.POINTER(FpShardCall)]
    L.fp_group_wait.argtypes = [C.c_void_p]
    if L.fp_abi_version() != FP_ABI_VERSION:
        raise ImportError(f"libfrenetgpu ABI {L.fp_abi_version()} != binding {FP_ABI_VERSION}: rebuild")
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise FrenetGpuError(rc, load().fp_last_error().decode(errors="replace"))
