#!/usr/bin/env python3
"""Probe: what an input-only launch order is worth.  Two config-3 batches alternate (so a learnt order never sees its own batch);
the egos of each batch are permuted before the upload by: nothing, descending ego speed; 1 / 2 / 4 batches cycled (a feedback order is
learnt on the launch two steps earlier: the same batch with 1 or 2 batches, another one with 4)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.device_batch import DeviceBatch
from fiss_plus_planner_amd.engine import FrenetEngine

eng = FrenetEngine(0)
dev = torch.device("cuda", 0)
_orig = DeviceBatch


def measure(tag, keyfn, order_opt, n_batches=2, hint=False):
    eng.set_option("lattice_order", order_opt)
    res = []
    for off in range(n_batches):
        b = synth.make_config(3, ego_offset=off * 4096)
        if keyfn is not None:
            b = b.take(np.argsort(keyfn(b), kind="stable"))
        db = _orig(b, 0, order_hint=hint)
        B = b.B
        bi = torch.empty(B, dtype=torch.int32, device=dev); bc = torch.empty(B, dtype=torch.float64, device=dev)
        bf = torch.zeros(B, dtype=torch.int32, device=dev); bt = torch.empty((B, 16, 112), dtype=torch.float64, device=dev)
        res.append((db, bi, bc, bf, bt, torch.cuda.current_stream(dev)))

    def run(k):
        db, bi, bc, bf, bt, st = res[k % len(res)]
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
    print(f"{tag:34s} {np.median(ts):7.1f} us per step (min {min(ts):.1f})", flush=True)


for nb in (1, 2, 4):  # 85 MB of tables per batch: one or two stay in the 256 MB Infinity Cache, four do not
    measure(f"{nb} batch(es): index order", None, 0, nb)
    measure(f"{nb} batch(es): speed descending", lambda b: -b.ego[:, 1], 0, nb)
    measure(f"{nb} batch(es): feedback order", None, 1, nb)
    measure(f"{nb} batch(es): speed hint (launch_order)", None, 0, nb, hint=True)
eng.set_option("lattice_order", 1)
