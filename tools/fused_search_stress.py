#!/usr/bin/env python3
"""Stress of the FISS+ search appended to the lattice launch ("fiss_fused"): the dense tables travel between workgroups of ONE launch
(agent-scope stores / loads + a flag per ego); a stale read would show up as a walk that differs from the three-launch pipeline's.

    python tools/fused_search_stress.py [repetitions per batch = 150]      (run on the GPU box; prints the number of differing calls)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiss_plus_planner_amd import synth  # noqa: E402
from fiss_plus_planner_amd.engine import FrenetEngine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
eng = FrenetEngine(0)
bad = total = 0
for off, B in ((0, 2048), (1, 2048), (2, 1500), (3, 4096)):
    batch = synth.make_config(4, B=B, ego_offset=off * 4096)
    eng.set_option("fiss_fused", 0)
    ref = eng.plan_fiss(batch, "FISS+")
    eng.set_option("fiss_fused", 1)
    for _ in range(reps):
        out = eng.plan_fiss(batch, "FISS+")
        ok = all(np.array_equal(getattr(out, k), getattr(ref, k)) for k in ("best_ijk", "stats", "refined", "prev_best_idx")) and \
             np.array_equal(out.best_cost, ref.best_cost, equal_nan=True) and np.array_equal(out.end_state, ref.end_state, equal_nan=True)
        bad += not ok
        total += 1
print(f"calls that differ from the three-launch pipeline: {bad} of {total}")
