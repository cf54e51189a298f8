#!/bin/bash
# Same-box A/B of ctx options on the headline workload: bash tools/ab_options.sh "" "lattice_tail=1" "lattice_occupancy=3" ...
# (each variant twice, interleaved; prints ms per step and the kernel's median from HIP events)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for OPT in "$@"; do
  BENCH_CTX_OPTIONS="$OPT" python bench.py --steps ${STEPS:-200} --warmup 20 --cpu-seconds 0 --no-latency --no-extras ${ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$OPT] step %.2f us  kernel %.2f us  cold %.2f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, (d['ms_per_step_cold'] or 0)*1e3))"
done; done
