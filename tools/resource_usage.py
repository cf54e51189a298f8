#!/usr/bin/env python3
"""Register / scratch / occupancy table of every gfx950 kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fiss_plus_planner_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -disable-machine-licm".split()  # as in csrc/Makefile
rows = []
for f in sorted(os.listdir(CSRC)):
    if not f.endswith(".hip"):
        continue
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[1:], "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, f), "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"kernel": re.sub(r"\(.*", "", name).replace("fp::", "")}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[\w/]+\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
cols = ["VGPRs", "AGPRs", "TotalSGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize", "Occupancy", "LDS Size"]
print(f"{'kernel':44s} " + " ".join(f"{c:>11s}" for c in cols))
for r in rows:
    print(f"{r['kernel'][:44]:44s} " + " ".join(f"{r.get(c, -1):11d}" for c in cols))
