#!/usr/bin/env python3
"""Plan cycles per second of the device-resident closed loop (eager launches vs one captured HIP graph replayed).

    python tools/closed_loop_rate.py [--egos 256] [--cycles 40] [--planner FOP]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fiss_plus_planner_amd import synth  # noqa: E402
from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch  # noqa: E402
from fiss_plus_planner_amd.engine import FrenetEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--egos", type=int, default=256)
    ap.add_argument("--cycles", type=int, default=40)
    ap.add_argument("--planner", default="FOP")
    args = ap.parse_args()
    eng = FrenetEngine(0)
    for mode in ("eager", "graph"):
        batch = synth.make_batch(args.egos, 5, 5, 5, 10, 100, False, 99, kind=args.planner)
        goal = np.full((args.egos, 2), 1e9)  # never reached: every ego runs all cycles
        run = ClosedLoopRunner(eng, DeviceBatch(batch, 0), goal, args.planner)
        run.run(2)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run.run_graph(args.cycles) if mode == "graph" else run.run(args.cycles)
        dt = time.perf_counter() - t0
        if mode == "graph":
            dt = run.replay_seconds * args.cycles / max(args.cycles - 1, 1)  # replays only (capture + instantiate excluded)
        n = int(out.cycles.sum())
        print(f"{mode:6s} {args.planner} B={args.egos}: {args.cycles} cycles in {dt * 1e3:.2f} ms -> {dt / args.cycles * 1e6:.1f} us/cycle, "
              f"{args.egos * args.cycles / dt / 1e6:.2f} M ego-plans/s (completed cycles {n})")


if __name__ == "__main__":
    main()
