#!/usr/bin/env python3
"""Install a PMC summary (profiles/collect_pmc.sh) as the round's committed one and refresh profiles/traffic.json from it.

    python tools/finalize_profiles.py gpurun_out/pmc_<tag>/summary.json [round prefix, default r05]

Run on the GPU box between the PMC passes and bench.py (bench reads both files), and again here on the merged copy.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r05"
summary = json.load(open(src))
name, k = next((n, v) for n, v in summary.items() if "lattice_fused" in n)
dst = os.path.join(ROOT, "profiles", f"{rnd}_config3_survey8d_pmc_summary.json")
json.dump(summary, open(dst, "w"), indent=1, sort_keys=True)
tpath = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tpath))
total = 2.0 * k["FETCH_SIZE"] * 1024.0 + k["WRITE_SIZE"] * 1024.0  # (FETCH_SIZE doubled: MI355X_MICROARCH.md's gfx950 correction)
t["config3_B2048_survey8d"] = total
t["config3_B2048_survey8d_parts"] = {"lattice_fused_kernel (lattice + appended epilogue workgroups: one launch)": total}
t[f"_note_{rnd}"] = (f"config3_B2048_survey8d: 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 of the ONE launch of a dense call, profiles/{os.path.basename(dst)} "
                     f"(kernel {name.replace('void fp::', '')}; collected in the same gpurun call as profiles/{rnd}_config3_kernel_stats.csv and "
                     f"profiles/{rnd}_bench.json; tools/finalize_profiles.py)")
json.dump(t, open(tpath, "w"), indent=1)
print(f"{dst}: {name}\n  traffic {total / 1e6:.1f} MB per launch, VALU {k['SQ_INSTS_VALU'] / 1e6:.2f} M")
