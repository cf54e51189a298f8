"""Host-side state types of the planner API.

Mirrors the public surface of the reference's planners/common/scenario/frenet.py
(State :6-13, FrenetState :15-99, FrenetTrajectory :101-222) so that callers of
``plan()`` keep working: ``best.state_at_time_step(1)``,
``best.frenet_state_at_time_step(1)``, ``best.x/.y/.yaw``, ``best.cost_final``,
ordering by ``cost_final``.  All per-point series are numpy float64 arrays
(the reference mixes python lists and arrays; both index and slice alike).
"""
from __future__ import annotations

import math
from enum import Enum

import numpy as np

# order of the 16 per-point series in every [16, stride] trajectory dump that
# crosses the C ABI (include/frenet_gpu.h, FP_ARR_*)
ARRAY_NAMES = ("t", "s", "s_d", "s_dd", "s_ddd", "d", "d_d", "d_dd", "d_ddd", "x", "y", "yaw", "ds", "c", "c_d", "c_dd")


_ARRAY_INDEX = {n: i for i, n in enumerate(ARRAY_NAMES)}


class LaneType(Enum):
    """planners/common/scenario/lane.py:8-12"""
    UNDEFINED = 0
    LEFT = 1
    EGO = 2
    RIGHT = 3


def unify_angle_range(angle: float) -> float:
    """planners/common/geometry/math_utils.py:28-34"""
    while angle > math.pi:
        angle -= 2 * math.pi
    while angle < -math.pi:
        angle += 2 * math.pi
    return angle


class State:
    def __init__(self, t: float = 0.0, x: float = 0.0, y: float = 0.0, yaw: float = 0.0, v: float = 0.0, a: float = 0.0):
        self.t, self.x, self.y, self.yaw, self.v, self.a = t, x, y, yaw, v, a


class FrenetState:
    def __init__(self, t: float = 0.0, s: float = 0.0, s_d: float = 0.0, s_dd: float = 0.0, s_ddd: float = 0.0,
                 d: float = 0.0, d_d: float = 0.0, d_dd: float = 0.0, d_ddd: float = 0.0):
        self.t = t
        self.s, self.s_d, self.s_dd, self.s_ddd = s, s_d, s_dd, s_ddd
        self.d, self.d_d, self.d_dd, self.d_ddd = d, d_d, d_dd, d_ddd

    def __str__(self):
        return f"FrenetState with d={self.d:.2f}, s_d={self.s_d:.2f}, t={self.t:.2f}"

    def as_start_vector(self) -> np.ndarray:
        """The six numbers the planner reads from a start state (s, s_d, s_dd, d, d_d, d_dd)."""
        return np.array([self.s, self.s_d, self.s_dd, self.d, self.d_d, self.d_dd], dtype=np.float64)

    def from_state(self, state: State, polyline: np.ndarray):
        """Cartesian -> Frenet projection on a resampled reference line [n, >=3] = x, y, yaw.

        Same rule set as reference frenet.py:32-99: nearest point, next-waypoint
        choice by heading, projection on the prev->next segment, sign from
        ``wp_yaw <= x_yaw``, s = polyline length up to the previous waypoint.
        """
        pl = np.asarray(polyline, dtype=np.float64)
        n = pl.shape[0]
        nearest = int(np.argmin(np.hypot(pl[:, 0] - state.x, pl[:, 1] - state.y)))
        heading = math.atan2(pl[nearest, 1] - state.y, pl[nearest, 0] - state.x)
        angle = abs(state.yaw - heading)
        angle = min(2 * math.pi - angle, angle)
        nxt = nearest + 1 if angle > math.pi / 2 else nearest
        if nxt < 1:
            nxt = 1
        elif nxt >= n:
            nxt = n - 1
        prv = max(nxt - 1, 0)
        n_x, n_y = pl[nxt, 0] - pl[prv, 0], pl[nxt, 1] - pl[prv, 1]
        x_x, x_y = state.x - pl[prv, 0], state.y - pl[prv, 1]
        x_yaw = math.atan2(x_y, x_x)
        proj = (x_x * n_x + x_y * n_y) / (n_x * n_x + n_y * n_y)
        d = math.hypot(x_x - proj * n_x, x_y - proj * n_y)
        wp_yaw = pl[prv, 2]
        delta_yaw = unify_angle_range(state.yaw - wp_yaw)
        if wp_yaw <= x_yaw:
            d = -d
        seg = np.hypot(np.diff(pl[: prv + 1, 0]), np.diff(pl[: prv + 1, 1]))
        s = 0.0
        for v in seg:  # left-to-right like the reference loop
            s += float(v)
        self.t = state.t
        self.s, self.s_d, self.s_dd, self.s_ddd = s, state.v * math.cos(delta_yaw), 0.0, 0.0
        self.d, self.d_d, self.d_dd, self.d_ddd = d, state.v * math.sin(delta_yaw), 0.0, 0.0
        return state


_TRAJ_DEFAULTS = dict(lane_id=-1, lane_type=None, is_generated=False, is_searched=False, constraint_passed=False,
                      collision_passed=False, end_state=None, cost_fix=0.0, cost_dyn=0.0, cost_heu=0.0, cost_est=0.0, cost_final=0.0)


class FrenetTrajectory:
    """Result object of ``plan()``; filled from a [16, stride] device dump."""

    def __init__(self):
        self.idx = np.array([-1, -1, -1])
        self.lane_id = -1
        self.lane_type = LaneType.UNDEFINED
        self.is_generated = False
        self.is_searched = False
        self.constraint_passed = False
        self.collision_passed = False
        self.end_state = None
        self.cost_fix = 0.0
        self.cost_dyn = 0.0
        self.cost_heu = 0.0
        self.cost_est = 0.0
        self.cost_final = 0.0
        for name in ARRAY_NAMES:
            setattr(self, name, np.empty(0))

    @classmethod
    def from_dump(cls, dump: np.ndarray, N: int, M: int, cost_final: float, end_state: "FrenetState | None" = None,
                  idx=None) -> "FrenetTrajectory":
        """dump: [16, stride]; N = len(t); M = len(x) (points that stayed on the spline)."""
        tr = cls.__new__(cls)
        dct = tr.__dict__
        dct.update(_TRAJ_DEFAULTS)
        tr.idx = np.array((-1, -1, -1) if idx is None else idx)
        # one copy of the block; the sixteen series are views of it, made when they are first read (__getattr__): a plan cycle
        # that only takes the next state out of its trajectory never builds the Cartesian rows
        dct["_dump"] = np.array(dump[:, :N])
        dct["_M"] = M
        tr.lane_type = LaneType.UNDEFINED
        tr.cost_final = float(cost_final)
        tr.is_generated = True
        tr.end_state = end_state
        return tr

    def __getattr__(self, name):  # (only reached for names that are not in __dict__: the series of a from_dump trajectory)
        k = _ARRAY_INDEX.get(name)
        d = self.__dict__.get("_dump")
        if k is None or d is None:
            raise AttributeError(name)
        if k < 9:
            v = d[k]
        else:
            M = self.__dict__["_M"]
            # lengths of x, y, yaw, ds, c, c_d, c_dd (frenet_optimal_planner.py:121-134)
            ln = (M, M, M, M - 1, M - 1, M - 2, M - 3)[k - 9] if M >= 2 else (M if k < 11 else 0)
            v = d[k, : max(ln, 0)]
        self.__dict__[name] = v
        return v

    # ordering by cost_final only (reference frenet.py:150-166)
    def __eq__(self, other):
        return self.cost_final == other.cost_final

    def __ne__(self, other):
        return self.cost_final != other.cost_final

    def __lt__(self, other):
        return self.cost_final < other.cost_final

    def __le__(self, other):
        return self.cost_final <= other.cost_final

    def __gt__(self, other):
        return self.cost_final > other.cost_final

    def __ge__(self, other):
        return self.cost_final >= other.cost_final

    __hash__ = object.__hash__

    def __repr__(self):
        return "%f" % self.cost_final

    def __str__(self):
        return (f"FrenetTrajectory with cost_final={self.cost_final:.2f},  d={self.end_state.d:.2f}, "
                f"s_d={self.end_state.s_d:.2f}, t={self.end_state.t:.2f}")

    def state_at_time_step(self, t: int) -> State:
        assert t < len(self.s) and t >= 0
        return State(self.t[t], self.x[t], self.y[t], self.yaw[t], self.s_d[t], self.s_dd[t])

    def frenet_state_at_time_step(self, t: int) -> FrenetState:
        assert t < len(self.s) and t >= 0
        return FrenetState(self.t[t], self.s[t], self.s_d[t], self.s_dd[t], self.s_ddd[t],
                           self.d[t], self.d_d[t], self.d_dd[t], self.d_ddd[t])

    def forward_t_steps(self, steps: int):
        if steps < 0 or steps >= len(self.t):
            return None
        import copy

        new = copy.deepcopy(self)
        for name in ARRAY_NAMES[:13] + ("c",):
            setattr(new, name, getattr(new, name)[steps:])
        return new
