#!/usr/bin/env python3
"""Diagnostic: which egos of a run-time-shape batch differ between the fused FISS+ search and its own launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.engine import FrenetEngine

eng = FrenetEngine(0)
for shape, B in (((7, 7, 7, 20, 50), 600), ((7, 7, 7, 20, 50), 2048), ((9, 9, 7, 50, 50), 600), ((5, 8, 9, 12, 50), 600)):
    b = synth.make_batch(B, *shape, True, 81, kind="FISS+")
    eng.set_option("fiss_fused", 0); ref = eng.plan_fiss(b, "FISS+")
    eng.set_option("fiss_fused", 1); out = eng.plan_fiss(b, "FISS+")
    bad = np.nonzero((out.best_ijk != ref.best_ijk).any(axis=1) | (out.stats != ref.stats).any(axis=1))[0]
    print(shape, B, "differing egos:", len(bad), bad[:20].tolist(), bad[-5:].tolist() if len(bad) else "")
    print("   appended_workgroups", eng.get_option("appended_workgroups"))
    if len(bad):
        e = bad[0]
        from oracle import oracle as O
        O.build()
        for e2, pr in zip(bad[:6], O.problems_from_batch(b, bad[:6])):
            r = pr.fissplus_plan()
            print("   oracle ego", e2, r.stats.tolist(), r.best_cost, "| fused", out.stats[e2].tolist(), out.best_cost[e2], "| own", ref.stats[e2].tolist(), ref.best_cost[e2])
        print("   ego", e, "fused", out.best_ijk[e], out.stats[e], out.best_cost[e], "own launch", ref.best_ijk[e], ref.stats[e], ref.best_cost[e])
