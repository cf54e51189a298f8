cd $GRAFT_REPO_ROOT/fiss_plus_planner_amd/csrc
touch frenet_fiss.hip; make EXTRA="-DREFINE_TIMING" > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
cd $GRAFT_REPO_ROOT; python tools/_tmp/timing.py
