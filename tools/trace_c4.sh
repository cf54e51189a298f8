#!/bin/bash
export FP_ALLOW_DIAGNOSTIC_BUILD=1  # these builds carry EXTRA=-DFP_...: the binding refuses them otherwise (fp_build_flags)
# kernel-trace stats of the FISS+ pipeline (bench.py --config 4) for build variants:  bash tools/trace_c4.sh "<EXTRA 1>" "<EXTRA 2>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for EXTRA in "$@"; do
  make -C fiss_plus_planner_amd/csrc -B -s EXTRA="$EXTRA" > /dev/null 2>&1 || { echo "[$EXTRA] build failed"; continue; }
  rm -rf /tmp/tr; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o c4 -- python $R/bench.py --config 4 --steps 30 --cpu-seconds 0 --no-latency --no-extras > /dev/null 2>&1)
  echo "variant [$EXTRA]"; python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/tr/c4_kernel_stats.csv')):
    if 'fp::' in r['Name']:
        print(f"   {r['Name'].split('(')[0][:40]:40s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
PY
done
make -C fiss_plus_planner_amd/csrc -B -s > /dev/null 2>&1
