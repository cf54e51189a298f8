#!/usr/bin/env python3
"""Timeline of the fused lattice + FISS+ search launch (build with EXTRA=-DFP_TL; run on the GPU box; the outputs carry clock ticks,
not results): when each ego's lattice workgroup starts and ends, when its search part becomes resident, sees the flag and ends."""
import os
os.environ.setdefault("FP_ALLOW_DIAGNOSTIC_BUILD", "1")  # runs against a library built with EXTRA=-DFP_...
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiss_plus_planner_amd import synth  # noqa: E402
from fiss_plus_planner_amd.engine import FrenetEngine  # noqa: E402

batch = synth.make_config(4)
eng = FrenetEngine(0)
for _ in range(4):
    out = eng.plan_fiss(batch, max_refine_iters=0)
tick = 0.01  # us
es = out.end_state * tick
lat_end = out.best_cost * tick
t_res, t_flag, t_end = es[:, 0], es[:, 1], es[:, 2]
lat_start = (out.refined.astype(np.int64) & 0x7FFFFFFF) * tick
base = (np.asarray(es[:, 0] / tick, dtype=np.int64) >> 31 << 31) * tick  # the lattice start keeps 31 bits
lat_start = lat_start + base
t0 = lat_start.min()
print(f"config 4, {batch.B} egos, fused lattice + FISS+ search launch (no refinement); times in us after the first lattice workgroup's start")
for name, v in (("lattice start", lat_start), ("lattice end", lat_end), ("search resident", t_res), ("search sees flag", t_flag), ("search end", t_end)):
    w = v - t0
    print(f"{name:18s} min {w.min():7.1f}  p10 {np.percentile(w, 10):7.1f}  median {np.median(w):7.1f}  p90 {np.percentile(w, 90):7.1f}  p99 {np.percentile(w, 99):7.1f}  max {w.max():7.1f}")
wait = t_flag - t_res
work = t_end - t_flag
lag = t_flag - lat_end
print(f"search waits for its flag: median {np.median(wait):.1f}  p90 {np.percentile(wait, 90):.1f}  max {wait.max():.1f} us;  egos whose search was resident before the lattice ended: {(t_res < lat_end).sum()}")
print(f"flag seen after the lattice end: median {np.median(lag):.1f}  p90 {np.percentile(lag, 90):.1f}  max {lag.max():.1f} us")
print(f"search work after the flag: median {np.median(work):.1f}  p90 {np.percentile(work, 90):.1f}  max {work.max():.1f} us;  > 10 us: {(work > 10).sum()} egos")
end = t_end.max() - t0
print(f"launch: lattice ends at {lat_end.max() - t0:.1f}, last search at {end:.1f} us")
edges = np.arange(0, end + 10, 10.0)
for a, bb in zip(edges[:-1], edges[1:]):
    m = 0.5 * (a + bb) + t0
    print(f"  {a:5.0f}..{bb:5.0f} us: lattice workgroups running {((lat_start <= m) & (lat_end > m)).sum():5d}   search workgroups resident {((t_res <= m) & (t_end > m)).sum():5d}  of which working {((t_flag <= m) & (t_end > m)).sum():5d}")
late = np.argsort(-t_end)[:8]
print("last searches: " + "; ".join(f"ego {b}: lattice end {lat_end[b] - t0:.0f}, resident {t_res[b] - t0:.0f}, flag {t_flag[b] - t0:.0f}, end {t_end[b] - t0:.0f}" for b in late))
if len(sys.argv) > 1:  # (durations of the lattice workgroups, for launch-order studies: tools/order_probe.py)
    np.save(sys.argv[1], np.stack([lat_start - t0, lat_end - t0]))
