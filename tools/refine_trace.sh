#!/bin/bash
# Per-dispatch durations of the FISS+ pipeline's kernels under rocprofv3 --kernel-trace (config 4): is fiss_refine_kernel's average a
# property of the kernel or of its first (unordered, cold) launches?   gpurun -- 'bash tools/refine_trace.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o c4 -- python $R/bench.py --config 4 --steps ${1:-100} --warmup 5 --cpu-seconds 0 --no-latency --no-extras > /tmp/tr.log 2>&1
python - <<'PY'
import csv, glob, numpy as np
f = glob.glob('/tmp/tr/**/c4_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for key in ("fiss_refine", "lattice_fused", "fissplus_search"):
    d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]])
    if len(d) == 0: continue
    print(f"{key:16s} n {len(d):4d}  mean {d.mean():7.1f}  median {np.median(d):7.1f}  min {d.min():7.1f}  p90 {np.percentile(d, 90):7.1f}  max {d.max():7.1f}  | first 12: {np.round(d[:12], 1).tolist()}  | last 6: {np.round(d[-6:], 1).tolist()}")
PY
