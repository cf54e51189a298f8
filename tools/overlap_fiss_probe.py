#!/usr/bin/env python3
"""FISS+ pipeline, two resident batches alternating on ONE engine and ONE caller stream: ordered against fp_ctx_set_option("overlap", 1);
several passes (is the overlapped rate stable?) and the number of calls that really started beside their predecessor."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.engine import FrenetEngine

dev = torch.device("cuda", 0)
eng = FrenetEngine(0)
stream = torch.cuda.current_stream(dev)
ws = [bench.Workload(torch, eng, synth.make_config(4, ego_offset=k * 2048), dev, stream, fiss=True, hint=False) for k in range(2)]


def rate(n=200):
    for k in range(40): ws[k % 2].step()
    eng.join(stream.cuda_stream); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n): ws[k % 2].step()
    eng.join(stream.cuda_stream); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for k in range(600): ws[k % 2].step()
torch.cuda.synchronize()
for rep in range(4):
    eng.set_option("overlap", 0)
    a = rate()
    eng.set_option("overlap", 1)
    n0 = eng.get_option("overlapped_calls")
    b = rate()
    print(f"pass {rep}: ordered {a:6.1f} us   overlap {b:6.1f} us per FISS+ step   ({eng.get_option('overlapped_calls') - n0} of 240 calls started beside their predecessor)", flush=True)
eng.set_option("overlap", 0)
