#!/bin/bash
# The round's judged artefacts in ONE GPU-box visit, in the order their consumers need them:
#   tests -> PMC passes of config 3 (installed as profiles/<rnd>_config3_survey8d_pmc_summary.json + traffic.json: bench.py reads both)
#   -> the bench line -> kernel-trace stats of configs 3 / 4 / 2 -> PMC passes of config 4.
#   gpurun --timeout 2700 -- 'bash tools/final_round.sh <tag> [rnd]'      then, here:  python tools/finalize_profiles.py gpurun_out/pmc_<tag>/summary.json
set -u
TAG=${1:-final}; RND=${2:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/gpu_round.sh $TAG tests
bash profiles/collect_pmc.sh $TAG > gpurun_out/$TAG/pmc.log 2>&1; tail -32 gpurun_out/$TAG/pmc.log | head -30
python tools/finalize_profiles.py gpurun_out/pmc_$TAG/summary.json $RND
bash tools/gpu_round.sh $TAG bench trace
bash profiles/collect_pmc.sh ${TAG}_c4 --config 4 > gpurun_out/$TAG/pmc_c4.log 2>&1; grep -A12 "lattice_fused\|fiss_refine" gpurun_out/$TAG/pmc_c4.log | head -40
