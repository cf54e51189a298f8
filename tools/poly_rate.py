#!/usr/bin/env python3
"""Step time of the dense call on polygon scenes (config-3 sizes, half of the obstacle columns random convex polygons) next to the
same scenes with plain rectangles: python tools/poly_rate.py [max_vertices]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fiss_plus_planner_amd import synth  # noqa: E402
from fiss_plus_planner_amd.device_batch import DeviceBatch  # noqa: E402
from fiss_plus_planner_amd.engine import FrenetEngine  # noqa: E402

mv = int(sys.argv[1]) if len(sys.argv) > 1 else 12
base = synth.make_config(3)
eng = FrenetEngine(0)
dev = torch.device("cuda", 0)
def rect_rings(b, frac=0.5):
    """half of the columns as 4-vertex rings that ARE their rectangles: the same collisions, the polygon code path (behind the constructor's
    back: ProblemBatch itself turns such rings into rectangle columns)"""
    return synth.with_rectangle_rings(b, 1, frac=frac, canonical=False)


norect = rect_rings(base)
norect.obs_nvert = np.zeros_like(norect.obs_nvert)  # (set behind the constructor's back: the POLY instance with no polygon column)
only = os.environ.get("POLY_ONLY")  # POLY_ONLY=3: the random-polygon scenes alone (PMC passes: one scene type per instance)
for sel, (name, b) in enumerate((("rectangles", base), ("POLY instance, no polygon", norect), ("rectangles as 4-vertex rings", rect_rings(base)), (f"polygons <= {mv} vertices", synth.with_random_shapes(base, 4242, frac=0.5, max_vertices=mv)))):
    if only is not None and int(only) != sel:
        continue
    db = DeviceBatch(b, 0, order_hint=False)  # (one batch replayed: the ctx orders the launch by the durations the batch itself left behind)
    B = b.B
    bi = torch.empty(B, dtype=torch.int32, device=dev); bc = torch.empty(B, dtype=torch.float64, device=dev)
    bf = torch.zeros(B, dtype=torch.int32, device=dev); bt = torch.empty((B, 16, 112), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev)
    run = lambda: eng.plan_dense_device(db.params, db.fb, bi.data_ptr(), bc.data_ptr(), stream=st.cuda_stream, best_flags=bf.data_ptr(),
                                        best_traj=bt.data_ptr(), traj_stride=112, traj_sparse=True)
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) * 10:7.1f} us per step   winners {float((bi >= 0).float().mean()):.2f}")
