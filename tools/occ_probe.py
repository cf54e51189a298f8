#!/usr/bin/env python3
"""Probe: dense step time of a resident batch of any lattice shape at the lattice kernel's occupancy choices
(fp_ctx_set_option("lattice_occupancy"): 0 auto - four per CU when the 40 KB layout fits -, 3, 2); results must be identical.
    python tools/occ_probe.py nd nv nt n_obs T_obs [B]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.device_batch import DeviceBatch
from fiss_plus_planner_amd.engine import FrenetEngine

nd, nv, nt, n_obs, T_obs = (int(x) for x in sys.argv[1:6])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 2048
eng = FrenetEngine(0)
dev = torch.device("cuda", 0)
res = []
for seed in (11, 12):
    b = synth.make_batch(B, nd, nv, nt, n_obs, T_obs, True, seed)
    db = DeviceBatch(b, 0)
    bi = torch.empty(B, dtype=torch.int32, device=dev); bc = torch.empty(B, dtype=torch.float64, device=dev)
    res.append((db, bi, bc))
st = torch.cuda.current_stream(dev)
ref = None
for occ in (0, 3, 2, 0, 3):
    eng.set_option("lattice_occupancy", occ)
    def run(k):
        db, bi, bc = res[k % 2]
        eng.plan_dense_device(db.params, db.fb, bi.data_ptr(), bc.data_ptr(), stream=st.cuda_stream)
    for k in range(300): run(k)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(100): run(k)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 10)
    out = (res[0][1].cpu().numpy().copy(), res[0][2].cpu().numpy().copy())
    if ref is None: ref = out
    same = np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1], equal_nan=True)
    print(f"{nd}x{nv}x{nt} n_obs={n_obs} T_obs={T_obs} B={B} NX={res[0][0].host.NX} occupancy {occ}: {np.median(ts):7.1f} us per step (min {min(ts):.1f})  winners {float((out[0] >= 0).mean()):.3f}  same={same}", flush=True)
eng.set_option("lattice_occupancy", 0)
