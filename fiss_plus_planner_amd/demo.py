"""Command-line counterpart of the reference's scripts/demo_cr.py + cfgs/demo_config.yaml (no GIF rendering):

    python -m fiss_plus_planner_amd.demo --cfg_file cfgs/demo_config.yaml
    python -m fiss_plus_planner_amd.demo --input_dir /path/to/scenarios --planner FISS+ -w 5 -s 5

Config keys as in the reference (INPUT_DIR, FILES, PLANNER, N_W_SAMPLE, N_S_SAMPLE, N_T_SAMPLE); like the reference
(planners/benchmark/planning.py:294) the time-sample count is taken from N_W_SAMPLE.
"""
from __future__ import annotations

import argparse
import glob
import os

import numpy as np


def run_file(path: str, planner_name: str, n_w: int, n_s: int, device: int = 0):
    from . import planners as P
    from .closed_loop import run_closed_loop
    from .commonroad_xml import load_scenario
    from .vehicle import Vehicle

    sc = load_scenario(path)
    num = (n_w, n_s, n_w)  # planning.py:294
    cls, st = {"FOP": (P.FrenetOptimalPlanner, P.FrenetOptimalPlannerSettings), "FOP+": (P.FopPlusPlanner, P.FrenetOptimalPlannerSettings),
               "FISS": (P.FissPlanner, P.FissPlannerSettings), "FISS+": (P.FissPlusPlanner, P.FissPlusPlannerSettings)}[planner_name]
    planner = cls(st(*num), Vehicle(), None, device=device)
    res = run_closed_loop(planner, sc.centerline, sc.init_state, sc.obstacles, sc.goal_center, sc.max_speed)
    ms = res.plan_seconds * 1e3
    n = max(len(res.cycles), 1)
    stats = res.stats.average(n)
    print(f"{sc.benchmark_id}: {planner_name} {len(res.cycles)} cycles, goal_reached={res.goal_reached}, "
          f"plan() p50 {np.median(ms) if len(ms) else float('nan'):.3f} ms, avg generated {stats.num_trajs_generated:.1f}, "
          f"validated {stats.num_trajs_validated:.1f}, collision checks {stats.num_collison_checks:.1f}")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg_file", default=None)
    ap.add_argument("--input_dir", default=None)
    ap.add_argument("--planner", default=None, choices=["FOP", "FOP+", "FISS", "FISS+"])
    ap.add_argument("-w", type=int, default=None, help="N_W_SAMPLE")
    ap.add_argument("-s", type=int, default=None, help="N_S_SAMPLE")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    cfg = {"INPUT_DIR": "data/demo/", "FILES": [], "PLANNER": "FISS+", "N_W_SAMPLE": 5, "N_S_SAMPLE": 5}
    if args.cfg_file:
        import yaml

        cfg.update(yaml.safe_load(open(args.cfg_file)))
    if args.input_dir:
        cfg["INPUT_DIR"] = args.input_dir
    if args.planner:
        cfg["PLANNER"] = args.planner
    if args.w:
        cfg["N_W_SAMPLE"] = args.w
    if args.s:
        cfg["N_S_SAMPLE"] = args.s
    files = [os.path.join(cfg["INPUT_DIR"], f) for f in cfg["FILES"]] or sorted(glob.glob(os.path.join(cfg["INPUT_DIR"], "*.xml")))
    for f in files:
        run_file(f, cfg["PLANNER"], cfg["N_W_SAMPLE"], cfg["N_S_SAMPLE"], args.device)


if __name__ == "__main__":
    main()
