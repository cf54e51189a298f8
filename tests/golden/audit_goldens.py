#!/usr/bin/env python3
"""Collision audit of the golden fixtures (build container only: needs /root/reference).

    python tests/golden/audit_goldens.py [--out /tmp/golden_audit] [--skip-generate]

Regenerates every fixture whose generation calls Polygon.intersects (g3 g4 g5 g6 g9 g10 g11 g12) into --out, with
refshim.Polygon.intersects = the EXACT predicate behind a float filter, compares every array of every regenerated .npz with the
committed file bit for bit, and writes tests/golden/collision_audit.json: per generator the number of intersects calls, how many of
them were within 1e-9 (relative) of contact and went to rational arithmetic, and how many of THOSE the plain fp64 separating-axis
test would have answered differently.
"""
import argparse
import glob
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GENS = ["g3", "g4", "g5", "g6", "g9", "g10", "g11", "g12"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="/tmp/golden_audit")
    ap.add_argument("--skip-generate", action="store_true")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    if not args.skip_generate:
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "gen_golden.py"), g], env=dict(os.environ, GOLDEN_OUT=args.out),
                                  stdout=open(os.path.join(args.out, g + ".log"), "w"), stderr=subprocess.STDOUT) for g in GENS]
        for p in procs:
            assert p.wait() == 0
    identical = True
    files = {}
    for path in sorted(glob.glob(os.path.join(args.out, "*.npz"))):
        name = os.path.basename(path)
        new, old = np.load(path, allow_pickle=False), np.load(os.path.join(HERE, name), allow_pickle=False)
        same = sorted(new.files) == sorted(old.files) and all(
            new[k].dtype == old[k].dtype and new[k].shape == old[k].shape and new[k].tobytes() == old[k].tobytes() for k in new.files)
        files[name] = {"arrays": len(new.files), "identical": bool(same)}
        identical &= same
    gens, total = {}, {"calls": 0, "near_contact": 0, "float_differs_from_exact": 0}
    for g in GENS:
        a = json.load(open(os.path.join(args.out, f"collision_audit_{g}.json")))
        gens[g] = {k: a[k] for k in total}
        for k in total:
            total[k] += a[k]
    out = {"what": "Polygon.intersects calls while regenerating the fixtures with the exact predicate (refshim.Polygon.intersects): near_contact = "
                   "calls within 1e-9 (relative) of touching, decided in rational arithmetic; float_differs_from_exact = of those, how many the "
                   "plain fp64 separating-axis test answers differently",
           "generators": gens, "total": total, "float_differs_from_exact_expected": total["float_differs_from_exact"], "files": files,
           "fixtures_identical_to_committed": bool(identical and len(files) == len(GENS))}
    json.dump(out, open(os.path.join(HERE, "collision_audit.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
