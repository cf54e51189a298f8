"""Closed-loop simulation harness = the planning loop of the reference's planners/benchmark/planning.py:101-162
(frenet frame, initial Cartesian->Frenet projection, `for i in range(final_time_step): plan(...)`, next state = point 1
of the winner, the three stop rules) without its commonroad / matplotlib dependencies.

Inputs are plain arrays: centerline [n,2], initial state (x, y, yaw, v), an ObstacleTable, the goal-lanelet centre.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from .frenet import FrenetState, State
from .obstacles import ObstacleTable
from .planners import Stats


@dataclass
class CycleRecord:
    start: list           # s, s_d, s_dd, d, d_d, d_dd fed to plan()
    cost: float
    N: int
    M: int
    idx: np.ndarray
    stats: tuple
    end: list             # end state (d, v, T) of the winner
    seconds: float        # wall time of plan(), measured where the reference measures it (planning.py:124-128)


@dataclass
class ClosedLoopResult:
    cycles: list = field(default_factory=list)
    states: list = field(default_factory=list)   # x, y, yaw after every cycle
    goal_reached: bool = False
    stats: Stats = field(default_factory=Stats)

    @property
    def plan_seconds(self):
        return np.array([c.seconds for c in self.cycles])


def run_closed_loop(planner, centerline: np.ndarray, init_state, obstacles: ObstacleTable, goal_center, max_speed: float = 13.5,
                    max_cycles: int | None = None) -> ClosedLoopResult:
    sp, ref = planner.generate_frenet_frame(centerline)
    cur = FrenetState()
    if getattr(planner, "frame_on", "host") == "device":
        # Cartesian -> Frenet projection on the GPU (fp_from_state), same rules as FrenetState.from_state
        e = planner._engine.from_state(sp.knots[None], sp.coef[None], [len(sp.knots)], [0], np.asarray(init_state, dtype=np.float64)[None, :4])[0]
        cur = FrenetState(t=0.0, s=e[0], s_d=e[1], s_dd=e[2], d=e[3], d_d=e[4], d_dd=e[5])
    else:
        cur.from_state(State(t=0.0, x=init_state[0], y=init_state[1], yaw=init_state[2], v=init_state[3], a=0.0), ref)
    res = ClosedLoopResult()
    n_cycles = obstacles.final_time_step if max_cycles is None else min(max_cycles, obstacles.final_time_step)
    half_len = planner.vehicle.l / 2
    for i in range(n_cycles):
        start = [cur.s, cur.s_d, cur.s_dd, cur.d, cur.d_d, cur.d_dd]
        t0 = time.perf_counter()
        best = planner.plan(cur, max_speed, obstacles, i)
        dt = time.perf_counter() - t0
        res.stats += planner.stats
        if best is None:
            break
        cs = best.state_at_time_step(1)
        cur = best.frenet_state_at_time_step(1)
        es = best.end_state
        res.cycles.append(CycleRecord(start, best.cost_final, len(best.t), len(best.x), best.idx, planner.stats.as_tuple(),
                                      [es.d, es.s_d, es.t] if es is not None else [np.nan] * 3, dt))
        res.states.append([cs.x, cs.y, cs.yaw])
        if np.hypot(cs.x - goal_center[0], cs.y - goal_center[1]) <= half_len:     # planning.py:154-157
            res.goal_reached = True
            break
        if np.hypot(cs.x - ref[-1, 0], cs.y - ref[-1, 1]) <= 3.0:                   # planning.py:158-161
            res.goal_reached = True
            break
    return res
