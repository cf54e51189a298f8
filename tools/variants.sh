#!/bin/bash
export FP_ALLOW_DIAGNOSTIC_BUILD=1  # these builds carry EXTRA=-DFP_...: the binding refuses them otherwise (fp_build_flags)
# Build + time variants of libfrenetgpu.so on the GPU box:   bash tools/variants.sh <tag> "<EXTRA flags 1>" "<EXTRA flags 2>" ...
# ("" = the plain build).  Prints the lattice kernel's time per variant (bench.py's HIP events, config 3, extras off) and, with
# PMC=1 in the environment, the VALU / SALU / LDS instruction counts of one counter pass.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
export TMPDIR=/tmp
i=0
for EXTRA in "$@"; do
  i=$((i+1))
  SRCDIR=fiss_plus_planner_amd/csrc; MKX="$EXTRA"
  # "@base [flags]": build the snapshot of an earlier commit under tools/_tmp/base instead (A/B inside one call)
  if [ "${EXTRA#@base}" != "$EXTRA" ]; then SRCDIR=tools/_tmp/base/fiss_plus_planner_amd/csrc; MKX="${EXTRA#@base}"; fi
  make -C $SRCDIR -B -s EXTRA="$MKX" OUT=$R/fiss_plus_planner_amd/libfrenetgpu.so > $OUT/build_$i.log 2>&1 || { echo "variant $i [$EXTRA]: BUILD FAILED"; tail -5 $OUT/build_$i.log; continue; }
  env ${BENCH_ENV:-} python bench.py --steps ${STEPS:-60} --warmup 8 --cpu-seconds ${CPU:-0} --no-latency --no-extras ${BENCH_ARGS:-} > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python - "$OUT/bench_$i.json" "$EXTRA" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d["kernel_ms_stats"]
    print(f"variant [{sys.argv[2]:40s}] kernel mean {k['mean']*1e3:7.1f} us  median {k['median']*1e3:7.1f}  min {k['min']*1e3:7.1f}   step {d['ms_per_step']*1e3:7.1f} us  value {d['value']:.4g}")
except Exception as e:
    print(f"variant [{sys.argv[2]}] FAILED: {e}"); print(open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
  if [ "${PMC:-0}" = "1" ]; then
    (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_INT32 SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_$i -o p -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-latency --no-extras ${BENCH_ARGS:-} > $OUT/pmc_$i.log 2>&1)
    python - $OUT/pmc_$i <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "lattice_fused" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("      " + "  ".join(f"{k.replace('SQ_', '')}={sum(v)/len(v)/1e6:.2f}M" for k, v in sorted(agg.items())))
PY
  fi
done
make -C fiss_plus_planner_amd/csrc -B -s > /dev/null 2>&1
