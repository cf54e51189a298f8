#!/bin/bash
export FP_ALLOW_DIAGNOSTIC_BUILD=1  # these builds carry EXTRA=-DFP_...: the binding refuses them otherwise (fp_build_flags)
# config 2 (256 egos x 5x5x5, 10 static obstacles): step time under launch-shape options + phase stamps
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for OPT in "lattice_group=0" "lattice_group=1,lattice_split=1" "lattice_group=1,lattice_split=0" "lattice_group=1,lattice_split=2"; do
  BENCH_CTX_OPTIONS="$OPT" timeout 120 python bench.py --config 2 --egos 256 --steps 200 --warmup 10 --cpu-seconds 0 --no-latency --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$OPT] step', round(d['ms_per_step']*1e3,2), 'us  kernel', round(d['kernel_ms_stats']['median']*1e3,2))"
done
make -C fiss_plus_planner_amd/csrc -B -s EXTRA=-DFP_PHASE_STAMPS > /dev/null 2>&1
for A in "2 0 0" "2 1 1"; do echo "== stamps $A"; timeout 100 python tools/phase_stamps.py $A 2>&1 | grep -v amdgpu | head -14; done
make -C fiss_plus_planner_amd/csrc -B -s EXTRA=-DFP_COUNTERS > /dev/null 2>&1; timeout 100 python tools/work_counters.py 2 2>&1 | grep -v amdgpu | tail -9
make -C fiss_plus_planner_amd/csrc -B -s > /dev/null 2>&1
