#!/usr/bin/env python3
"""Materialise mode alone (bench.py's materialize leg): python tools/mat_rate.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.engine import FrenetEngine
dev = torch.device("cuda", 0)
eng = FrenetEngine(0)
wl = bench.Workload(torch, eng, synth.make_config(3, B=256), dev, torch.cuda.current_stream(dev))
m = bench.materialize_leg(torch, eng, wl, dev, torch.cuda.current_stream(dev))
for k in [k for k in m if isinstance(m[k], dict)]:
    v = m[k]; print(k, f"median {v['kernel_ms']:.3f} ms  min {v['kernel_ms_min']:.3f}  max {v['kernel_ms_max']:.3f}  spread {v['spread']:.2f}  written {v['achieved_GBps']:.0f} GB/s  algorithmic {v['algorithmic_GBps']:.0f} GB/s")
