#!/usr/bin/env python3
"""Probe: config-3 work on a 27-knot reference line (a workgroup's LDS then fits four per CU); step time of the dense call."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.device_batch import DeviceBatch
from fiss_plus_planner_amd.engine import FrenetEngine
from fiss_plus_planner_amd.spline import build_frames
NXp = int(sys.argv[1]) if len(sys.argv) > 1 else 27
eng = FrenetEngine(0)
dev = torch.device("cuda", 0)
res = []
for off in (0, 1):
    b = synth.make_config(3, ego_offset=off * 4096)
    if NXp != 81:
        xs = np.linspace(0.0, 400.0, NXp)
        pts = np.empty((b.B, NXp, 2))
        for e in range(b.B):
            pts[e, :, 0] = xs
            pts[e, :, 1] = np.interp(xs, b.coef[e, 0, :81], b.coef[e, 4, :81])
        b.knots, b.coef = build_frames(pts)
        b.nx = np.full(b.B, NXp, dtype=np.int32)
    db = DeviceBatch(b, 0)
    B = b.B
    bi = torch.empty(B, dtype=torch.int32, device=dev); bc = torch.empty(B, dtype=torch.float64, device=dev)
    bf = torch.zeros(B, dtype=torch.int32, device=dev); bt = torch.empty((B, 16, 112), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev)
    res.append((db, bi, bc, bf, bt, st))
def run(k):
    db, bi, bc, bf, bt, st = res[k % 2]
    eng.plan_dense_device(db.params, db.fb, bi.data_ptr(), bc.data_ptr(), stream=st.cuda_stream, best_flags=bf.data_ptr(), best_traj=bt.data_ptr(), traj_stride=112, traj_sparse=True)
for k in range(400): run(k)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(200): run(k)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 5)
print(f"NX={NXp}: {np.median(ts):.1f} us per step (min {min(ts):.1f}); winners {float((res[0][1] >= 0).float().mean()):.3f} idx checksum {int(res[0][1].sum())}")
