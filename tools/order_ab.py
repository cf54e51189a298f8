#!/usr/bin/env python3
"""Same-process A/B of the launch-order sources on the headline workload (4 resident config-3 batches cycled, series written):
own history per batch (+ hint until it exists) / own history, no hint / the caller's hint alone / index order.  Interleaved, 3 passes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.device_batch import DeviceBatch
from fiss_plus_planner_amd.engine import FrenetEngine

eng = FrenetEngine(0)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
res = []
for off in range(4):
    b = synth.make_config(3, ego_offset=off * 2048)
    db = DeviceBatch(b, 0, order_hint=True)
    B = b.B
    res.append((db, db.fb.launch_order, torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.float64, device=dev),
                torch.zeros(B, dtype=torch.int32, device=dev), torch.empty((B, 16, 112), dtype=torch.float64, device=dev)))


def run(k):
    db, _h, bi, bc, bf, bt = res[k % 4]
    eng.plan_dense_device(db.params, db.fb, bi.data_ptr(), bc.data_ptr(), stream=st.cuda_stream, best_flags=bf.data_ptr(), best_traj=bt.data_ptr(), traj_stride=112, traj_sparse=True)


def measure(order, hint, n=400):
    eng.set_option("lattice_order", order)
    for db, h, *_ in res:
        db.fb.launch_order = h if hint else None
    for k in range(200): run(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n): run(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for k in range(1000): run(k)
for rep in range(3):
    print(f"pass {rep}: own history + hint {measure(1, True):6.1f} | own history, no hint {measure(1, False):6.1f} | hint alone {measure(0, True):6.1f} | index order {measure(0, False):6.1f}  us per 2048-ego call", flush=True)
eng.set_option("lattice_order", 1)
