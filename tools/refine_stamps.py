#!/usr/bin/env python3
"""Where a workgroup of fiss_refine_kernel spends its time (build with EXTRA=-DFP_PHASE_STAMPS; run on the GPU box): thread 0
stamps the phase boundaries of every ego that reaches the refinement (config 4)."""
import os
os.environ.setdefault("FP_ALLOW_DIAGNOSTIC_BUILD", "1")  # runs against a library built with EXTRA=-DFP_...
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiss_plus_planner_amd import synth  # noqa: E402
from fiss_plus_planner_amd.engine import FrenetEngine  # noqa: E402

batch = synth.make_config(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
eng = FrenetEngine(0)
for _ in range(3):
    out = eng.plan_fiss(batch, winner=True, traj_stride=128, traj_sparse=True)
raw = out.best_traj[:, 15, 112:128]
act = raw[:, 11] > 0  # egos whose workgroup wrote a series after refinement rounds
t = raw[act] * 0.01   # us
start = t[:, 0] - t[:, 0].min()
print(f"{act.sum()} of {batch.B} egos refine; workgroup starts (us after the first): median {np.median(start):.1f}  p90 {np.percentile(start, 90):.1f}  max {start.max():.1f}")
groups = (t[:, 3:10] > 0).sum(axis=1)
print("validation groups per ego:", np.bincount(groups).tolist())
print(f"rounds (wavefront 0)   median {np.median(t[:, 1]):6.2f}  p90 {np.percentile(t[:, 1], 90):6.2f}")
print(f"... + staging barrier  median {np.median(t[:, 2]):6.2f}  p90 {np.percentile(t[:, 2], 90):6.2f}")
g1 = t[:, 3] - t[:, 2]
print(f"first group            median {np.median(g1):6.2f}  p90 {np.percentile(g1, 90):6.2f}")
print(f"  inside (wavefront 0): points done at +{np.median(t[:, 12] - t[:, 2]):.2f}, broad phase done at +{np.median((t[:, 13] - t[:, 2])[t[:, 13] > 0]):.2f} ({(t[:, 13] > 0).sum()} egos), verdicts at +{np.median(g1):.2f}")
more = groups > 1
if more.any():
    g2 = t[more, 4] - t[more, 3]
    print(f"second group           median {np.median(g2):6.2f}  p90 {np.percentile(g2, 90):6.2f}  ({more.sum()} egos)")
ser = t[:, 11] - t[:, 10]
print(f"results -> series      median {np.median(ser):6.2f}  p90 {np.percentile(ser, 90):6.2f}")
print(f"workgroup total        median {np.median(t[:, 11]):6.2f}  p90 {np.percentile(t[:, 11], 90):6.2f}  max {t[:, 11].max():6.2f}")
end = start + t[:, 11]
print(f"last workgroup ends {end.max():.1f} us after the first one started")

# the search kernel's stamps (columns 96..111 of the same row): egos whose walk ran past its first iteration
sr = out.best_traj[:, 15, 96:112] * 0.01
walked = sr[:, 10] > 0
w = sr[walked]
print(f"\nfissplus_search_kernel: {walked.sum()} egos walk past the first iteration ({(sr[:, 7] > 0).sum()} rank their lattice)")
names = [(1, "T1 tables (global reads) + reductions"), (3, "T2 histogram + T3 scan"), (4, "T4 scatter"), (5, "T5 rank in bucket"), (6, "T6 cost_est + records"),
         (7, "first iteration"), (8, "jump: level relaxation"), (9, "jump: beta + state"), (10, "remaining iterations + results")]
prev = np.zeros(len(w))
for k, n in names:
    cur = w[:, k]
    print(f"  {n:40s} median {np.median(cur - prev):6.2f}  p90 {np.percentile(cur - prev, 90):6.2f}")
    prev = cur
print(f"  {'workgroup total':40s} median {np.median(w[:, 10]):6.2f}  p90 {np.percentile(w[:, 10], 90):6.2f}  max {w[:, 10].max():6.2f};  iterations walked one by one after the jump: median {np.median(w[:, 11] * 100):.0f}  p90 {np.percentile(w[:, 11] * 100, 90):.0f}  max {w[:, 11].max() * 100:.0f}")
t0 = w[:, 12] * 100 * 0.01 - (w[:, 12] * 100 * 0.01).min()
print(f"  walking workgroups start (us after the first of them): median {np.median(t0):.1f}  p90 {np.percentile(t0, 90):.1f}  max {t0.max():.1f};  the last one ends at {(t0 + w[:, 10]).max():.1f}")
