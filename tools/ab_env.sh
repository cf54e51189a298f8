#!/bin/bash
# Same build, different environments (A/B inside one GPU-box visit):   bash tools/ab_env.sh <tag> "<env 1>" "<env 2>" ...   ("" = plain)
# Every variant runs REPS times, interleaved; prints kernel mean / median and step time of bench.py's config-3 headline (extras off).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for E in "$@"; do
    i=$((i+1))
    env $E python bench.py --steps ${STEPS:-200} --warmup 8 --cpu-seconds 0 --no-latency --no-extras ${BENCH_ARGS:-} > $OUT/bench_${i}_$rep.json 2> $OUT/bench_${i}_$rep.err
    python - "$OUT/bench_${i}_$rep.json" "$E" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d["kernel_ms_stats"]
    print(f"[{sys.argv[2]:36s}] kernel mean {k['mean']*1e3:7.1f} us  median {k['median']*1e3:7.1f}  min {k['min']*1e3:7.1f}   step {d['ms_per_step']*1e3:7.2f} us  cold {d.get('ms_per_step_cold') and d['ms_per_step_cold']*1e3 or 0:7.2f}")
except Exception as e:
    print(f"[{sys.argv[2]}] FAILED: {e}"); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
  done
done
