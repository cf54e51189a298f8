#!/bin/bash
# Time bench.py's main workload under different ctx options (no rebuild):  bash tools/options.sh <tag> "name=v,name=v" ...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
i=0
for OPT in "$@"; do
  i=$((i+1))
  BENCH_CTX_OPTIONS="$OPT" python bench.py --steps ${STEPS:-60} --warmup 8 --cpu-seconds 0 --no-latency --no-extras ${BENCH_ARGS:-} > $OUT/opt_$i.json 2> $OUT/opt_$i.err
  python - "$OUT/opt_$i.json" "$OPT" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d["kernel_ms_stats"]
    print(f"options [{sys.argv[2]:36s}] kernel mean {k['mean']*1e3:7.1f} us  median {k['median']*1e3:7.1f}  min {k['min']*1e3:7.1f}   step {d['ms_per_step']*1e3:7.1f} us  value {d['value']:.4g}")
except Exception as e:
    print(f"options [{sys.argv[2]}] FAILED: {e}"); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
done
