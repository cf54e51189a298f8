#!/bin/bash
# Collect PMC counters for the lattice kernel (run ON the GPU box via gpurun).  One rocprofv3 pass per
# counter group (SQ: <= 8 per pass; FETCH_SIZE and WRITE_SIZE cannot share a pass).  No trace domains are
# combined with --pmc.   usage: profiles/collect_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters
  rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o $1 -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-latency --no-extras "${@:3}" > $OUT/$1.log 2>&1
}
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "$@"
run sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@"
run sq3 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU" "$@"
run fetch "FETCH_SIZE" "$@"
run write "WRITE_SIZE" "$@"
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT" "$@"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(summary, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
for k, d in summary.items():
    print(k)
    for c in sorted(d):
        print(f"   {c:28s} {d[c]:.4g}")
PY
