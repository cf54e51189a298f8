#!/usr/bin/env python3
"""Stress of the winner-series epilogue workgroups appended to the dense launch ("lattice_winner" = 0, the default for batches of more
egos than stay resident): the argmin travels between workgroups of ONE launch (agent-scope store + a flag per ego); a stale read or a
missed flag would show up as a series / flag word that differs from winner_traj_kernel's ("lattice_winner" = 2).

    python tools/epilogue_stress.py [repetitions per batch = 300]      (run on the GPU box; prints the number of differing calls)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiss_plus_planner_amd import synth  # noqa: E402
from fiss_plus_planner_amd.engine import FrenetEngine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng = FrenetEngine(0)
bad = total = 0
for off, B, tables in ((0, 2048, False), (1, 2048, True), (2, 1501, False), (3, 4096, False), (4, 769, True)):
    batch = synth.make_config(3, B=B, ego_offset=off * 4096)
    eng.set_option("lattice_winner", 2)
    ref = eng.plan_dense(batch, tables=tables, winner=True)
    eng.set_option("lattice_winner", 0)
    keys = ("best_idx", "stats", "best_flags") + (("flags",) if tables else ())
    for _ in range(reps):
        out = eng.plan_dense(batch, tables=tables, winner=True)
        ok = all(np.array_equal(getattr(out, k), getattr(ref, k)) for k in keys) and np.array_equal(out.best_cost, ref.best_cost, equal_nan=True) and \
             np.array_equal(out.best_traj, ref.best_traj, equal_nan=True) and (not tables or np.array_equal(out.cost, ref.cost, equal_nan=True))
        bad += not ok
        total += 1
print(f"calls that differ from the winner_traj_kernel path: {bad} of {total}")
