#!/usr/bin/env python3
"""Timing ablations of the dense lattice kernel on the GPU box (profiling aid, not a test).

    python tools/ablate.py [--egos 2048]

Variants of BASELINE config 3: full scene / obstacles moved 1 km away (broad phase only, no hits) /
no obstacles at all (profiles + cost + argmin only) / per-candidate kernel.
"""
import argparse
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


def time_batch(eng, batch, steps=20, warmup=3, tables=False):
    dev = torch.device("cuda", 0)
    dten = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in NAMES}
    fb = device_batch(batch, {k: (v.data_ptr() if v.numel() else 0) for k, v in dten.items()})
    params = make_params(batch)
    bi = torch.empty(batch.B, dtype=torch.int32, device=dev)
    bc = torch.empty(batch.B, dtype=torch.float64, device=dev)
    ct = torch.empty((batch.B, batch.C), dtype=torch.float64, device=dev) if tables else None
    ft = torch.empty((batch.B, batch.C), dtype=torch.int32, device=dev) if tables else None
    st = torch.cuda.current_stream(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i in range(warmup + steps):
        if i >= warmup:
            ev[i - warmup][0].record(st)
        eng.plan_dense_device(params, fb, bi.data_ptr(), bc.data_ptr(), 0, ct.data_ptr() if tables else 0, ft.data_ptr() if tables else 0,
                              stream=st.cuda_stream)
        if i >= warmup:
            ev[i - warmup][1].record(st)
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    return float(np.median(ms)), float(ms.min())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--egos", type=int, default=2048)
    ap.add_argument("--config", type=int, default=3)
    args = ap.parse_args()
    eng = FrenetEngine(0)
    full = synth.make_config(args.config, B=args.egos)
    far = synth.make_config(args.config, B=args.egos)
    far.obs_pose[..., 0] += 1000.0
    none = synth.make_batch(args.egos, full.nd, full.nv, full.nt, 0, 0, False, synth.CONFIG_SEEDS.get(args.config, 1))
    rows = []
    for name, b, opt in (("full", full, 2), ("far-obstacles", far, 2), ("no-obstacles", none, 2), ("full+tables", full, 2), ("percand", full, 1)):
        eng.set_option("lattice_kernel", opt)
        med, mn = time_batch(eng, b, tables=name.endswith("tables"))
        rows.append((name, med, mn, b.B * b.C / med / 1e6))
        print(f"{name:16s} median {med:8.3f} ms   min {mn:8.3f} ms   {b.B * b.C / med / 1e6:9.3f} Gcand/s", flush=True)
    eng.set_option("lattice_kernel", 0)
    # host-buffer entry point (FP_MEM_HOST): H2D staging of the whole batch + kernel + D2H, i.e. the PCIe-inclusive rate
    import time
    eng.plan_dense(full, tables=False)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.plan_dense(full, tables=False)
    dt = (time.perf_counter() - t0) / 3
    print(f"host-buffers     {dt * 1e3:8.3f} ms per call (PCIe-inclusive)   {full.B * full.C / dt / 1e6:9.1f} Mcand/s", flush=True)


if __name__ == "__main__":
    main()
