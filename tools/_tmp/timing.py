import numpy as np, sys
sys.path.insert(0, ".")
from fiss_plus_planner_amd import synth
from fiss_plus_planner_amd.engine import FrenetEngine
eng = FrenetEngine(0)
b = synth.make_config(4)
for _ in range(3):
    out = eng.plan_fiss(b, "FISS+", trace=True)
t = out.trace.reshape(b.B, -1)[:, :12]
ok = ~np.isnan(t[:, 0])
t = t[ok]
us = 0.01  # wall_clock64: 100 MHz
print("egos", ok.sum())
for i, n in enumerate(["prologue", "rounds", "validation(total)", "wave0 traj_flags"]):
    print(f"{n:20s} mean {t[:, i].mean()*us:8.2f} us  p50 {np.median(t[:, i])*us:8.2f}  max {t[:, i].max()*us:8.2f}")
for i, n in enumerate(["bvp", "phase1 points", "pre-pair", "pair loop"]):
    print(f"  {n:18s} per call {t[:, 8+i].sum()/t[:, 4].sum()*us:8.2f} us")
print("groups mean", t[:, 4].mean(), "max", t[:, 4].max(), " validated mean", t[:, 5].mean(), "max", t[:, 5].max())
print("per traj_flags call (wave 0) mean us", (t[:, 3].sum() / t[:, 4].sum()) * us)
span = (t[:, 7].max() - t[:, 6].min()) * us
print("kernel span us", span, " block duration mean", ((t[:, 7] - t[:, 6]).mean()) * us)
st = np.sort(t[:, 6] - t[:, 6].min()) * us
print("block start times percentiles", np.percentile(st, [10, 37, 50, 75, 90, 100]))
