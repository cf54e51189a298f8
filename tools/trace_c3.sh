#!/bin/bash
# kernel-trace stats of the headline workload under ctx options:  bash tools/trace_c3.sh "opt=v,..." ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for OPT in "$@"; do
  rm -rf /tmp/tr; (cd /tmp && TMPDIR=/tmp BENCH_CTX_OPTIONS="$OPT" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o c3 -- python $R/bench.py --steps 40 --warmup 5 --cpu-seconds 0 --no-latency --no-extras > /dev/null 2>&1)
  echo "options [$OPT]"; python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/tr/c3_kernel_stats.csv')):
    if 'fp::' in r['Name']:
        print(f"   {r['Name'].split('(')[0][:44]:44s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
PY
done
