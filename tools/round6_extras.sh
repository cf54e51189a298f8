#!/bin/bash
# The same-box experiments DESIGN.md section 3.2 / EXPERIMENTS.md "Round 6" quote, in one GPU-box visit:  gpurun -- 'bash tools/round6_extras.sh <tag>'
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r6x}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
python tools/order_ab.py 2>&1 | grep -v amdgpu > $OUT/order_ab.txt; cat $OUT/order_ab.txt
bash tools/refine_trace.sh 100 2>&1 | tail -3 > $OUT/refine_trace.txt; cat $OUT/refine_trace.txt
python tools/poly_rate.py 2>&1 | grep -v amdgpu | tail -5 > $OUT/poly_rate.txt; cat $OUT/poly_rate.txt
(cd tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o any_order any_order.hip 2>/dev/null && ./any_order) > $OUT/any_order.txt 2>&1; cat $OUT/any_order.txt
STEPS=200 bash tools/ab_options.sh "" "lattice_tail=1" "lattice_tail=256" "lattice_tail=512" > $OUT/tail_ab.txt 2>&1; cat $OUT/tail_ab.txt
python tools/sharded_host_rate.py 150 2>/dev/null | tail -1 > $OUT/sharded_host_rate_150.json
python tools/sharded_host_rate.py 3000 2>/dev/null | tail -1 > $OUT/sharded_host_rate_3000.json
