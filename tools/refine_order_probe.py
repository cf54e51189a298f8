#!/usr/bin/env python3
"""What a launch order could win for fiss_refine_kernel (build with EXTRA=-DFP_PHASE_STAMPS; run on the GPU box): per-workgroup
durations from the stamps, a list-scheduling model on 768 slots in index order, longest-first (the bound) and in the order of
predictors known before the launch (collision share of the ego's lattice, coarse search statistics)."""
import heapq
import os
os.environ.setdefault("FP_ALLOW_DIAGNOSTIC_BUILD", "1")  # runs against a library built with EXTRA=-DFP_...
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiss_plus_planner_amd import _abi, synth  # noqa: E402
from fiss_plus_planner_amd.engine import FrenetEngine  # noqa: E402

batch = synth.make_config(4)
eng = FrenetEngine(0)
eng.set_option("fiss_fused", 0)
for _ in range(3):
    out = eng.plan_fiss(batch, winner=True, traj_stride=128, traj_sparse=True)
raw = out.best_traj[:, 15, 112:128]
act = raw[:, 11] > 0
dur = np.where(act, raw[:, 11] * 0.01, 0.6)
groups = np.where(act, (raw[:, 3:10] > 0).sum(axis=1), 0)
dense = eng.plan_dense(batch, tables=True)
fl = dense.flags
passc = ((fl & _abi.FLAG_CONSTRAINTS) == 0).sum(axis=1)
coll = (((fl & _abi.FLAG_CONSTRAINTS) == 0) & ((fl & _abi.FLAG_COLLISION) != 0)).sum(axis=1)
share = coll / np.maximum(passc, 1)
st = out.stats


def makespan(order, slots=768):
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for b in order:
        t = heapq.heappop(free)
        e = t + dur[b]
        end = max(end, e)
        heapq.heappush(free, e)
    return end


B = batch.B
idx = np.arange(B)
print(f"active {act.sum()}  groups {np.bincount(groups).tolist()}  duration of the active: median {np.median(dur[act]):.1f}  max {dur.max():.1f}")
print(f"index order            {makespan(idx):6.1f} us")
print(f"longest first (bound)  {makespan(np.argsort(-dur, kind='stable')):6.1f} us")
print(f"active first           {makespan(np.argsort(~act, kind='stable')):6.1f} us")
for name, key in (("collision share", share), ("collisions", coll), ("search iterations", st[:, 0]), ("search validated", st[:, 2]), ("ego speed", np.asarray(batch.ego)[:, 1])):
    a = key[act].astype(float)
    g = dur[act]
    ra, rg = np.argsort(np.argsort(a)), np.argsort(np.argsort(g))
    rho = np.corrcoef(ra, rg)[0, 1]
    print(f"by {name:18s}  {makespan(np.argsort(-(key + 1e6 * act), kind='stable')):6.1f} us   (Spearman with the duration among the active: {rho:.2f})")
