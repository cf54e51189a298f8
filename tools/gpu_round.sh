#!/bin/bash
# One GPU-box visit: tests, the bench line, kernel-trace stats and (optionally) the PMC passes.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh <tag> [tests] [bench] [trace] [pmc]'
# Everything lands under gpurun_out/<tag>/ (scratch); copy what should be judged into profiles/.
set -u
TAG=${1:-run}; shift || true
WHAT="${*:-tests bench trace}"
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
fi
if has bench; then
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; cp $R/bench_extras.json $OUT/bench_extras.json 2>/dev/null
fi
if has trace; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c3 -o c3 -- python $R/bench.py --steps 50 --warmup 5 --cpu-seconds 0 --no-latency --no-extras > $OUT/trace_c3.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c4 -o c4 -- python $R/bench.py --config 4 --steps 50 --warmup 5 --cpu-seconds 0 --no-latency --no-extras > $OUT/trace_c4.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -o c2 -- python $R/bench.py --config 2 --egos 256 --steps 50 --warmup 5 --cpu-seconds 0 --no-latency --no-extras > $OUT/trace_c2.log 2>&1
  cd $R
  for c in c3 c4 c2; do f=$(ls $OUT/trace_$c/*kernel_stats.csv $OUT/trace_$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/${c}_kernel_stats.csv && head -6 $f; done
fi
if has pmc; then
  bash profiles/collect_pmc.sh $TAG > $OUT/pmc.log 2>&1; tail -40 $OUT/pmc.log
  cp $R/gpurun_out/pmc_$TAG/summary.json $OUT/pmc_summary.json 2>/dev/null
fi
echo done
