import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.engine import FrenetEngine
eng = FrenetEngine(0)
def run(B, opts, shape=(7, 7, 7, 20, 50)):
    b = synth.make_batch(B, *shape, True, 81, kind="FISS+")
    for k, v in opts.items(): eng.set_option(k, v)
    eng.set_option("fiss_fused", 0); ref = eng.plan_fiss(b, "FISS+")
    eng.set_option("fiss_fused", 1); out = eng.plan_fiss(b, "FISS+")
    out2 = eng.plan_fiss(b, "FISS+")
    bad = np.nonzero((out.stats != ref.stats).any(axis=1))[0]
    bad2 = np.nonzero((out2.stats != ref.stats).any(axis=1))[0]
    print(B, shape, opts, "differing:", len(bad), "second call:", len(bad2), bad[:8].tolist(), "tail share", float((bad >= B - 192).mean()) if len(bad) else None)
    for k in opts: eng.set_option(k, 0)
for B in (600, 700, 769, 1000, 2048):
    run(B, {})
run(2048, {"lattice_tail": 1})
run(2048, {"lattice_order": 0})
eng.set_option("lattice_order", 1)
run(2048, {"lattice_tail": 1, "lattice_order": 0})
eng.set_option("lattice_order", 1)
run(2048, {}, (6, 6, 6, 50, 50))
run(2048, {}, (9, 9, 7, 50, 40))
