#!/bin/bash
export FP_ALLOW_DIAGNOSTIC_BUILD=1  # these builds carry EXTRA=-DFP_...: the binding refuses them otherwise (fp_build_flags)
# run one pytest selection against several builds:  bash tools/bisect_test.sh "<pytest args>" "<EXTRA or @base [EXTRA]>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
SEL=$1; shift
for EXTRA in "$@"; do
  SRCDIR=fiss_plus_planner_amd/csrc; MKX="$EXTRA"
  if [ "${EXTRA#@base}" != "$EXTRA" ]; then SRCDIR=tools/_tmp/base/fiss_plus_planner_amd/csrc; MKX="${EXTRA#@base}"; fi
  make -C $SRCDIR -B -s EXTRA="$MKX" OUT=$R/fiss_plus_planner_amd/libfrenetgpu.so > /tmp/b.log 2>&1 || { echo "[$EXTRA] BUILD FAILED"; tail -3 /tmp/b.log; continue; }
  echo "== [$EXTRA]"; timeout 300 python -m pytest $SEL -x -q 2>&1 | grep -v amdgpu | tail -4
done
make -C fiss_plus_planner_amd/csrc -B -s > /dev/null 2>&1
