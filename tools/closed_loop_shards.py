#!/usr/bin/env python3
"""Device-resident closed loop of one fleet cut into K half fleets, each with its own fp_ctx + HIP stream, cycles enqueued
alternately from one host thread (bench.py's closed_loop.two_streams is K = 2):  python tools/closed_loop_shards.py [FOP|FISS+] [B]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the runtime's default of four lets two streams share a hardware queue)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiss_plus_planner_amd import _abi, synth
from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch
from fiss_plus_planner_amd.engine import FrenetEngine

planner = sys.argv[1] if len(sys.argv) > 1 else "FOP"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
cycles = 50
dev = torch.device("cuda", 0)
batch = synth.make_config(3 if planner == "FOP" else 4, B=B)
goal = np.full((B, 2), 1e9)
ref = None
for K in (1, 2, 3, 4, 6, 8):
    parts = []
    for r in range(K):
        eng, st = FrenetEngine(0), torch.cuda.Stream(dev)
        sb = batch.shard(r, K)
        lo, hi = (B * r) // K, (B * (r + 1)) // K
        with torch.cuda.stream(st):
            rn = ClosedLoopRunner(eng, DeviceBatch(sb, 0), goal[lo:hi], planner)
        parts.append((rn, st, sb, eng))
    torch.cuda.synchronize(dev)
    for rn, st, sb, _ in parts:
        rn.step(st.cuda_stream); rn.step(st.cuda_stream)
    torch.cuda.synchronize(dev)
    for rn, st, sb, _ in parts:
        with torch.cuda.stream(st):
            rn.db.t["ego"].copy_(torch.from_numpy(sb.ego)); rn.db.t["t_now"].zero_(); rn.done.zero_(); rn.cycles.zero_()
            if planner != "FOP":
                rn.prev.fill_(-1)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(cycles - 1):
        for rn, st, sb, _ in parts:
            rn.step(st.cuda_stream)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    done = np.concatenate([rn.done.cpu().numpy() for rn, *_ in parts]); cyc = np.concatenate([rn.cycles.cpu().numpy() for rn, *_ in parts])
    ego = np.concatenate([rn.db.t["ego"].cpu().numpy() for rn, *_ in parts])
    plans = int(cyc.sum() + (done == _abi.DONE_NO_SOLUTION).sum())
    if ref is None:
        ref = (ego, done, cyc)
    same = all(np.array_equal(a, b) for a, b in zip(ref, (ego, done, cyc)))
    print(f"{planner} B={B} K={K}: {plans / dt:.4g} ego-plans/s  {dt / (cycles - 1) * 1e6:6.1f} us per cycle  (host enqueue {t_enq / (cycles - 1) * 1e6:5.1f} us)  same final states: {same}")
    for *_, eng in parts:
        eng.close()
