#!/usr/bin/env python3
"""Work statistics of lattice_fused_kernel's collision stages per ego (build with EXTRA=-DFP_COUNTERS; run on the GPU box):
items that survive the group test, (item, profile) pairs tested / passed by stage B, narrow-phase items, how many of them
belonged to candidates that had collided already, and how many were new collisions."""
import os
os.environ.setdefault("FP_ALLOW_DIAGNOSTIC_BUILD", "1")  # runs against a library built with EXTRA=-DFP_...
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fiss_plus_planner_amd import synth  # noqa: E402
from fiss_plus_planner_amd.engine import FrenetEngine, device_batch, make_params  # noqa: E402

NAMES = ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
         "obs_pose", "obs_dims", "final_time_step")
# usage: work_counters.py [layout | config number] [lattice_group]
arg = sys.argv[1] if len(sys.argv) > 1 else "survey8d"
config, layout = (int(arg), "survey8d") if arg.isdigit() else (3, arg)
batch = synth.make_config(config, layout=layout)
dev = torch.device("cuda", 0)
eng = FrenetEngine(0)
eng.set_option("lattice_winner", 1)
eng.set_option("lattice_split", 1)  # (the counters of one workgroup per ego)
if len(sys.argv) > 2:
    eng.set_option("lattice_group", int(sys.argv[2]))
dten = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in NAMES}
fb = device_batch(batch, {k: (v.data_ptr() if v.numel() else 0) for k, v in dten.items()})
params = make_params(batch)
B = batch.B
bi = torch.empty(B, dtype=torch.int32, device=dev); bc = torch.empty(B, dtype=torch.float64, device=dev)
bf = torch.zeros(B, dtype=torch.int32, device=dev); bt = torch.zeros((B, 16, 128), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev)
eng.plan_dense_device(params, fb, bi.data_ptr(), bc.data_ptr(), stream=st.cuda_stream, best_flags=bf.data_ptr(), best_traj=bt.data_ptr(),
                      traj_stride=128, traj_sparse=True)
torch.cuda.synchronize()
c = bt[:, 14, 112:128].cpu().numpy()
names = ["G survivors (items)", "B wave rounds (64 items of one profile each)", "B pairs passed (hits)", "N lane tests (live candidates only)", "-",
         "N lane tests that found a collision", "N wave rounds (floor(64 / nd) hits each)", "profiles whose walk ended early (all nd collided)"]
if os.environ.get("FP_SLICE_LOOP"):  # the counters of the older slice-loop build
    names = ["G survivors (items)", "B pairs tested", "B pairs passed (hits)", "N items", "N items, candidate already collided", "N new collisions"]
print(f"config {config}, layout {layout}: per ego, {B} egos")
for k, n in enumerate(names):
    print(f"  {n:40s} mean {c[:, k].mean():9.1f}  median {np.median(c[:, k]):9.1f}  p90 {np.percentile(c[:, k], 90):9.1f}  max {c[:, k].max():9.0f}")
print("  blocked profiles by the pose index at which the last lateral sample collided (bins of 8 steps = 0.8 s): " +
      "  ".join(f"k<{8 * (i + 1)}: {c[:, 10 + i].mean():.1f}" if i < 5 else f"k>=40: {c[:, 15].mean():.1f}" for i in range(6)) + "   (mean per ego)")
print("  slices whose lon profiles needed the point-by-point mask scan: mean", round(float(c[:, 9].mean()), 2), "of", batch.nt)
