#!/usr/bin/env python3
"""Where the single-ego plan() cycle of bench.py's latency leg spends its wall time (run on the GPU box):
per-call medians of the host-side stages of FrenetOptimalPlanner.plan, and a cProfile of the closed loop."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiss_plus_planner_amd import planners as P  # noqa: E402
from fiss_plus_planner_amd.closed_loop import run_closed_loop  # noqa: E402
from fiss_plus_planner_amd.engine import FrenetEngine  # noqa: E402
from fiss_plus_planner_amd.obstacles import ObstacleTable  # noqa: E402
from fiss_plus_planner_amd.vehicle import Vehicle  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, "tests", "golden", "g5_closed_loop.npz"))
fts = int(g["final_time_step"])
table = ObstacleTable(g["obs_pose"][:fts], g["obs_dims"], fts)
eng = FrenetEngine(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "FOP"
for kv in sys.argv[2:]:  # ctx options: name=value
    eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
cls, st = {"FOP": (P.FrenetOptimalPlanner, P.FrenetOptimalPlannerSettings), "FISS+": (P.FissPlusPlanner, P.FissPlusPlannerSettings)}[kind]
for _ in range(2):
    pl = cls(st(5, 5, 5), Vehicle(), None, engine=eng)
    res = run_closed_loop(pl, g["centerline"], g["init_state"], table, g["goal_center"])
ms = res.plan_seconds * 1e3
print(f"{kind}: plan() p50 {np.median(ms) * 1e3:.1f} us  p90 {np.percentile(ms, 90) * 1e3:.1f} us over {len(ms)} cycles")
# stage timers: wrap the planner's helpers
acc = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        acc.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    setattr(obj, name, w)
pl = cls(st(5, 5, 5), Vehicle(), None, engine=eng)
for n in ("_make_batch", "_dense", "_plan_on_device"):
    if hasattr(pl, n):
        wrap(pl, n)
lib = eng._lib
for n in ("fp_plan_dense", "fp_plan_fiss"):
    f = getattr(lib, n)
    def mk(f, n):
        def w(*a):
            t0 = time.perf_counter(); r = f(*a); acc.setdefault(n, []).append(time.perf_counter() - t0); return r
        return w
    try:
        setattr(lib, n, mk(f, n))
    except Exception as e:  # ctypes function objects may refuse attribute replacement
        print("cannot wrap", n, e)
res = run_closed_loop(pl, g["centerline"], g["init_state"], table, g["goal_center"])
for n, v in acc.items():
    print(f"  {n:18s} median {np.median(v) * 1e6:7.1f} us  ({len(v)} calls)")
print(f"  plan() with the timers on: p50 {np.median(res.plan_seconds) * 1e6:.1f} us")
pr = cProfile.Profile()
pl = cls(st(5, 5, 5), Vehicle(), None, engine=eng)
pr.enable()
run_closed_loop(pl, g["centerline"], g["init_state"], table, g["goal_center"])
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
