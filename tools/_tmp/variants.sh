#!/bin/bash
cd $GRAFT_REPO_ROOT/fiss_plus_planner_amd/csrc
export TMPDIR=/tmp
for W in 2 4; do for MW in 2 3 4; do
  touch frenet_fiss.hip
  make EXTRA="-DREFINE_WAVES=$W -DREFINE_MIN_WAVES=$MW" > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; continue; }
  (cd /tmp; rm -rf /tmp/pv; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o v -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 20 --no-latency > /dev/null 2>&1)
  echo "W=$W MINWAVES=$MW $(grep fiss_refine /tmp/pv/v_kernel_stats.csv | cut -d, -f5)"
done; done
