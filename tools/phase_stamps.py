#!/usr/bin/env python3
"""Where a workgroup of lattice_fused_kernel spends its time: phase boundaries stamped by thread 0 (build with
EXTRA=-DFP_PHASE_STAMPS; run on the GPU box).  Prints the median / p90 duration of every phase over the egos of config 3."""
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
PHASES = ["scalars + spline tables + obstacle sizes", "LUT + speed bound + lateral extremes", "A0 boundary-value solves + proofs + lateral bound",
          "power sums + point scans + arclength ranges", "cost sums + row circles", "group test G (scene table read) + sincos",
          "slices: frames / lat / prep / B / N", "assembly + argmin", "results"]
STAMPS = [0, 1, 2, 3, 4, 7, 8, 9, 10]

# usage: phase_stamps.py [config | "b1"] [lattice_split] [lattice_group];  b1 = one ego, 5 x 5 x 5, 27 obstacles (the plan-cycle case)
CONFIG = sys.argv[1] if len(sys.argv) > 1 else "3"
batch = synth.make_batch(1, 5, 5, 5, 27, 100, True, 7) if CONFIG == "b1" else synth.make_config(int(CONFIG))
dev = torch.device("cuda", 0)
eng = FrenetEngine(0)
eng.set_option("lattice_winner", 1)  # the stamps travel in the winner block the lattice kernel itself writes
if len(sys.argv) > 2:
    eng.set_option("lattice_split", int(sys.argv[2]))
if len(sys.argv) > 3:
    eng.set_option("lattice_group", int(sys.argv[3]))
dten = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in NAMES}
fb = device_batch(batch, {k: (v.data_ptr() if v.numel() else 0) for k, v in dten.items()})
params = make_params(batch)
B = batch.B
bi = torch.empty(B, dtype=torch.int32, device=dev); bc = torch.empty(B, dtype=torch.float64, device=dev)
bf = torch.zeros(B, dtype=torch.int32, device=dev); bt = torch.zeros((B, 16, 128), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev)
for _ in range(12):
    eng.plan_dense_device(params, fb, bi.data_ptr(), bc.data_ptr(), stream=st.cuda_stream, best_flags=bf.data_ptr(), best_traj=bt.data_ptr(),
                          traj_stride=128, traj_sparse=True)
torch.cuda.synchronize()
raw = bt[:, 15, 112:128].cpu().numpy() * 0.01  # us
stamps = raw[:, STAMPS]
d = np.diff(np.concatenate([np.zeros((B, 1)), stamps], axis=1), axis=1)
print(f"{'phase':48s} {'median us':>10s} {'p90 us':>10s} {'mean us':>10s}")
for k, name in enumerate(PHASES):
    print(f"{name:48s} {np.median(d[:, k]):10.2f} {np.percentile(d[:, k], 90):10.2f} {d[:, k].mean():10.2f}")
print(f"{'workgroup total':48s} {np.median(stamps[:, -1]):10.2f} {np.percentile(stamps[:, -1], 90):10.2f} {stamps[:, -1].mean():10.2f}")
print(f"sum of workgroup durations / 512 slots = {stamps[:, -1].sum() / 512:.1f} us, / 768 slots = {stamps[:, -1].sum() / 768:.1f} us")
# the walk's imbalance: when the first / the last wavefront of the workgroup ran out of profiles (columns 11, 12)
print("walk: first wavefront done at {:.2f} us, last at {:.2f} us after the workgroup's start (medians); G ended at {:.2f}: the walk takes {:.2f}, of which {:.2f} is waiting for the slowest wavefront".format(
    np.median(raw[:, 11]), np.median(raw[:, 12]), np.median(raw[:, 7]), np.median(raw[:, 12] - raw[:, 7]), np.median(raw[:, 12] - raw[:, 11])))
# (slice-loop builds, -DFP_SLICE_LOOP) inside the 4th slice of the workgroup: stamps 11..14 after the frames / prep / B / N barriers
sl = raw[:, 11:15]
print("one slice (the 4th):  prep {:.2f}  B {:.2f}  N {:.2f} us (medians);  frames + lat = slice total - these".format(
    np.median(sl[:, 1] - sl[:, 0]), np.median(sl[:, 2] - sl[:, 1]), np.median(sl[:, 3] - sl[:, 2])))
print("absolute stamps of that slice / group (us since the workgroup started): after frames+lat {:.2f}, prep {:.2f}, B {:.2f}, N {:.2f}; G ended at {:.2f}".format(
    *(np.median(sl[:, i]) for i in range(4)), np.median(raw[:, 7])))

# occupancy of the launch over time: workgroups running at every instant (absolute starts in column 15)
t0 = raw[:, 15] - raw[:, 15].min()
t1 = t0 + stamps[:, -1]
span = t1.max()
grid = np.linspace(0, span, 400)
running = ((t0[None, :] <= grid[:, None]) & (t1[None, :] > grid[:, None])).sum(axis=1)
print(f"launch: first start to last end {span:.1f} us; workgroups running: max {running.max()}, mean {running.mean():.0f}")
for lo, hi in ((0, 0.1), (0.1, 0.3), (0.3, 0.5), (0.5, 0.7), (0.7, 0.8), (0.8, 0.9), (0.9, 1.0)):
    m = (grid >= lo * span) & (grid < hi * span)
    print(f"  {lo * span:6.1f} .. {hi * span:6.1f} us: mean {running[m].mean():6.0f} running")
order = np.argsort(t0)
print("start time of the n-th workgroup (us):", {n: round(float(t0[order[n]]), 1) for n in (0, 255, 511, 767, 768, 1023, 1535, 2047) if n < B})
if raw[:, 5].max() > 0:  # latency mode: the last workgroup of an ego to arrive (its own clock)
    print("last workgroup of an ego (us since IT started): ticket taken {:.2f}, results written {:.2f}, series written {:.2f}".format(
        np.median(raw[:, 5]), np.median(raw[:, 10]), np.median(raw[:, 6])))
