#!/bin/bash
export FP_ALLOW_DIAGNOSTIC_BUILD=1  # these builds carry EXTRA=-DFP_...: the binding refuses them otherwise (fp_build_flags)
# Materialise mode under prebuilt variants of libfrenetgpu.so:  build them here (no GPU needed), time them on the box.
#   build:  bash tools/mat_variants.sh build name1="-DFLAG ..." name2="..."      -> tools/_tmp/var/<name>.so
#   run  :  gpurun -- 'bash tools/mat_variants.sh run [script.py]'               (default script: tools/mat_rate.py)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
MODE=$1; shift
if [ "$MODE" = build ]; then
  mkdir -p tools/_tmp/var
  for kv in "$@"; do
    name=${kv%%=*}; flags=${kv#*=}
    make -C fiss_plus_planner_amd/csrc -B -s EXTRA="$flags" OUT=$R/tools/_tmp/var/$name.so 2>&1 | grep -v "^$" | head -5 &
  done
  wait
  ls -la tools/_tmp/var
else
  SCRIPT=${1:-tools/mat_rate.py}   # (CMD="python bench.py ..." in the environment: any command instead, its last TAIL lines are shown)
  cp fiss_plus_planner_amd/libfrenetgpu.so /tmp/plain.so
  for so in /tmp/plain.so $(ls tools/_tmp/var/*.so); do
    cp $so fiss_plus_planner_amd/libfrenetgpu.so
    echo "== $(basename $so .so)"
    if [ -n "${CMD:-}" ]; then timeout 600 bash -c "$CMD" 2>&1 | tail -${TAIL:-4}; else timeout 300 python $SCRIPT 2>&1 | tail -${TAIL:-4}; fi
  done
  cp /tmp/plain.so fiss_plus_planner_amd/libfrenetgpu.so
fi
