#!/usr/bin/env python3
"""Host-side rate of the in-process multi-GPU path, measured on ONE device with logical shards: how many resident plan
calls per second the host can ENQUEUE at world = 1 / 2 / 4 / 8 when the GPU is not the limit (tiny batches: a few egos per shard, a
5 x 5 x 5 lattice without obstacles - every launch is a ~10-20 us kernel, eight streams on eight hardware queues).  What an 8-GPU node
needs: 8 x (1 / 0.146 ms) = 55 000 dense step-launches/s and 8 x (1 / 68 us) = 120 000 closed-loop cycles/s.

    GPU_MAX_HW_QUEUES=8 python tools/sharded_host_rate.py [steps = 3000]       (run on the GPU box; prints one JSON object)

Two paths are timed: `group` = ShardedEngine's resident calls (ONE ctypes call per round posts prebuilt argument blocks to the
library's persistent per-ctx worker threads, fp_group_submit) and `pool` = what round 4 shipped (a ThreadPoolExecutor task + future
per shard and step, each marshalling its own arguments).  HOST-ONLY numbers: the 1 / 2 / 4 / 8-GPU scaling curve itself is unmeasured
(no multi-GPU node was available)."""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from fiss_plus_planner_amd import synth  # noqa: E402
from fiss_plus_planner_amd.sharded import ShardedEngine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
per_shard = 4
out = {"steps": steps, "egos_per_shard": per_shard, "lattice": "5x5x5, no obstacles (the GPU is not the limit)", "worlds": {}}
for world in (1, 2, 4, 8):
    batch = synth.make_batch(per_shard * world, 5, 5, 5, 0, 0, False, 7)
    with ShardedEngine(devices=[0], shards_per_device=world) as eng:
        sdb = eng.upload(batch)
        ref = eng.plan_dense(sdb).best_idx.copy()
        # (a) dense steps through the group
        for _ in range(50):
            eng.plan_dense(sdb, sync=False)
        sdb.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.plan_dense(sdb, sync=False)
        t_post = time.perf_counter() - t0
        sdb.synchronize()
        t_all = time.perf_counter() - t0
        assert np.array_equal(sdb.host.best_idx, ref)
        row = {"dense_group": {"launches_per_s": steps * world / t_all, "rounds_per_s": steps / t_all, "host_post_us_per_round": t_post / steps * 1e6}}
        # (b) the round-4 path: one thread-pool task per shard and step
        def call(sh):
            r = sh.res
            sh.engine.plan_dense_device(sh.db.params, sh.db.fb, r["best_idx"].data_ptr(), r["best_cost"].data_ptr(), r["stats"].data_ptr(), stream=sh.stream.cuda_stream)
        n_pool = max(200, steps // 5)
        t0 = time.perf_counter()
        for _ in range(n_pool):
            if world == 1:
                call(sdb.shards[0])
            else:
                for f in [eng._pool.submit(call, sh) for sh in sdb.shards]:
                    f.result()
        sdb.synchronize()
        row["dense_pool"] = {"launches_per_s": n_pool * world / (time.perf_counter() - t0)}
        # (c) closed-loop cycles (fp_plan_step per shard and cycle) through the group
        goal = np.full((batch.B, 2), 1e9)
        eng.closed_loop(sdb, goal, "FOP", max_cycles=50)
        sdb.reset_state(batch)
        t0 = time.perf_counter()
        res = eng.closed_loop(sdb, goal, "FOP", max_cycles=steps)
        dt = time.perf_counter() - t0
        row["plan_step_group"] = {"launches_per_s": steps * world / dt, "cycles_per_s": steps / dt, "egos_done": int((res.done != 0).sum())}
        out["worlds"][str(world)] = row
need = {"dense_launches_per_s_at_world_8": 55000, "plan_step_launches_per_s_at_world_8": 120000}
w8 = out["worlds"]["8"]
out["verdict_targets"] = {**need, "dense_met": w8["dense_group"]["launches_per_s"] >= need["dense_launches_per_s_at_world_8"],
                          "plan_step_met": w8["plan_step_group"]["launches_per_s"] >= need["plan_step_launches_per_s_at_world_8"]}
print(json.dumps(out))
