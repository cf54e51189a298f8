#!/usr/bin/env python3
"""Where a workgroup of fiss_refine_kernel spends its time (build with EXTRA=-DFP_PHASE_STAMPS; run on the GPU box): thread 0
stamps the phase boundaries of every ego that reaches the refinement (config 4)."""
import os
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
