#!/bin/bash
export FP_ALLOW_DIAGNOSTIC_BUILD=1  # these builds carry EXTRA=-DFP_...: the binding refuses them otherwise (fp_build_flags)
# stamps + counters + option A/Bs of the current lattice kernel, all under short timeouts:  bash tools/gpu_probe.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=gpurun_out/$1; mkdir -p $OUT
make -C fiss_plus_planner_amd/csrc -B -s EXTRA=-DFP_PHASE_STAMPS > /dev/null 2>&1 && timeout 120 python tools/phase_stamps.py 3 > $OUT/stamps.txt 2>&1; tail -22 $OUT/stamps.txt | head -16
make -C fiss_plus_planner_amd/csrc -B -s EXTRA=-DFP_COUNTERS > /dev/null 2>&1 && timeout 120 python tools/work_counters.py 3 > $OUT/counters.txt 2>&1; tail -10 $OUT/counters.txt
make -C fiss_plus_planner_amd/csrc -B -s > /dev/null 2>&1
STEPS=90 timeout 300 bash tools/options.sh $1 "lattice_tail=0" "lattice_tail=1" "lattice_tail=96" "lattice_tail=384" 2>&1 | tail -5
