"""GPU: the N-rank bench path and BASELINE configs[4] (16384 egos, sharded).

* `python bench.py --gpus 2` spawns two ranks by itself (torch.distributed.run); on the 1-GPU test box both ranks share
  device 0 and rendezvous over gloo (BENCH_ALL_ON_DEVICE0 / BENCH_DIST_BACKEND hooks) - everything else is the code the
  8-GPU run executes: per-rank shard generation, barrier, max-over-ranks, one JSON line with n_gpus = 2.
* the full 16384-ego config-5 batch on ONE GPU: every shard a rank would plan equals the same rows of the full-batch
  result (that is the whole multi-GPU contract: no collective, concatenation), invariants of the tables hold for every
  ego, and an oracle sample spread over all eight shards agrees exactly.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from fiss_plus_planner_amd import synth

pytestmark = pytest.mark.gpu
COST_TOL = 1e-6


def _run_bench(extra, env_extra=None, timeout=900, tmp_path=None):
    """Run bench.py; returns (the contract line = ALL of stdout, the full record from the side file)."""
    import bench as bench_mod

    extras_file = os.path.join(str(tmp_path) if tmp_path else ROOT, "bench_extras_test.json")
    env = dict(os.environ, BENCH_EXTRAS_FILE=extras_file, **(env_extra or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    # (the gloo test hook chats on stdout - "[Gloo] Rank 1 is connected to 7 peer ranks ...", interleaved between ranks; RCCL prints nothing)
    lines = [l for l in out.stdout.splitlines() if l.strip() and "Gloo" not in l and "peer ranks" not in l]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]   # nothing before, nothing after the line
    assert out.stdout.rstrip("\n").splitlines()[-1] == lines[0]               # ... and it is the LAST line
    assert len(lines[0].encode()) < bench_mod.LINE_LIMIT
    line = json.loads(lines[0], parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))
    assert tuple(line) == bench_mod.CONTRACT_KEYS
    full = json.load(open(extras_file))
    os.remove(extras_file)
    assert line["extras_file"] == "bench_extras_test.json"
    return line, full


def test_bench_gpus_2_spawns_two_ranks(tmp_path):
    line, _ = _run_bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--egos", "512"],
                         {"BENCH_DIST_BACKEND": "gloo", "BENCH_ALL_ON_DEVICE0": "1"}, tmp_path=tmp_path)
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 2
    assert line["scaling"] == "weak" and line["unit"] == "candidates/s"
    assert "configs[4]" in line["config"]["workload"]
    # whole-job aggregate: both ranks' candidates over the max-over-ranks time
    assert abs(line["value"] - 2 * 512 * 567 * 6 / (line["ms_per_step"] * 6e-3)) / line["value"] < 1e-5
    assert line["roofline"]["frac"] > 0 and line["roofline"]["kernel_ms"] > 0
    assert line["cpu_baseline"] is None  # rank 0 at N = 1 only


def test_bench_rejects_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)


def test_bench_gpus_8_on_one_device(tmp_path):
    """The N = 8 line of the driver's scaling run (BASELINE configs[4]: 8 ranks x 2048 egos), with all eight ranks on device 0 over
    gloo and small shards: rank generation, barrier, max-over-ranks and ONE compact line with n_gpus = 8."""
    line, full = _run_bench(["--gpus", "8", "--steps", "4", "--warmup", "1", "--egos", "128"],
                            {"BENCH_DIST_BACKEND": "gloo", "BENCH_ALL_ON_DEVICE0": "1", "BENCH_PREWARM_S": "0.05"}, tmp_path=tmp_path, timeout=1500)
    assert line["n_gpus"] == 8 and line["steps"] == 4 and line["scaling"] == "weak"
    assert "configs[4]" in line["config"]["workload"] and line["config"]["egos_per_gpu"] == 128
    assert "x8" in line["config"]["parallelism"]
    assert abs(line["value"] - 8 * 128 * 567 * 4 / (line["ms_per_step"] * 4e-3)) / line["value"] < 1e-5
    assert line["cpu_baseline"] is None and line["legs"] is None and line["roofline"]["frac"] > 0
    assert full["n_gpus"] == 8


def test_bench_single_gpu_line_has_every_configuration(tmp_path):
    cline, line = _run_bench(["--steps", "8", "--warmup", "2", "--cpu-seconds", "2", "--no-latency"], tmp_path=tmp_path)
    # the compact line: the contract's numbers, and one ms + verdict per leg
    assert cline["n_gpus"] == 1 and "configs[2]" in cline["config"]["workload"] and cline["parity"]["index_exact"] is True
    assert cline["roofline"]["bound"] == "hbm" and 0 < cline["roofline"]["frac"] < 1 and cline["roofline"]["kernel_ms"] > 0
    assert cline["cpu_baseline"]["kind"] == "port" and cline["cpu_baseline"]["value"] > 0 and cline["cpu_baseline"]["cores"] >= 1
    assert cline["value_cold"] > 0 and cline["parity"]["max_abs_cost_err"] <= 1e-6 and cline["parity"]["checked_egos"] >= 64
    for key in ("config2", "config4", "polygon_scenes", "launch_order_hint_off", "two_streams", "closed_loop_FOP", "closed_loop_FISS+"):
        assert cline["legs"][key]["parity_ok"] is True, key
    assert abs(cline["legs"]["config4"]["ms_per_step"] - line["config4"]["ms_per_step"]) <= 1e-5 * line["config4"]["ms_per_step"]
    assert abs(cline["value"] - line["value"]) <= 1e-8 * line["value"]
    # the full record (the side file)
    assert line["n_gpus"] == 1 and "configs[2]" in line["config"]["workload"]
    assert line["config"]["obstacle_layout"] == "survey8d" and line["config"]["rotating_batches"] == 4
    assert len(set(line["config"]["input_digests"])) == 4
    # every leg that prints a number was compared with the oracle in the same run
    assert len(line["parity"]["batches"]) == 4 and all(p["index_exact"] for p in line["parity"]["batches"])
    for key in ("config2", "config4", "lattice_order_off", "single_batch_replayed", "lanes_layout"):
        assert line[key]["value"] > 0 and 0 < line[key]["roofline"]["frac"] < 1, key
        assert line[key]["parity"]["index_exact"] and line[key]["parity"]["max_abs_cost_err"] <= 1e-6, key
    assert line["config2"]["parity"]["checked_egos"] == 256
    # SURVEY 8(d)'s full metric: per-candidate tables written + Stats fetched; the resident sharded entry point; configs[4] on one GPU
    tw = line["tables_written"]
    assert tw["tables_written"] and tw["value"] > 0 and tw["parity"]["flag_words_exact"] and tw["parity"]["stats_exact"]
    assert tw["parity"]["max_abs_cost_err"] <= 1e-6 and tw["parity"]["checked_candidates"] >= 40 * 567
    sr = line["sharded_resident"]
    assert sr["value"] > 0 and sr["parity"]["index_exact"] and 0.5 < sr["vs_headline"] < 1.3
    c5 = line["config5_single_gpu"]
    assert c5["value"] > 0 and c5["parity"]["index_exact"] and "16384" in c5["workload"]
    assert line["config4"]["parity"]["stats_exact"] and line["config4"]["parity"]["checked_egos"] >= 64
    st = line["config4"]["stage_ms"]
    assert set(st) == {"lattice_fused_kernel (dense tables)", "fissplus_search_kernel (own launch)", "fiss_refine_kernel (3 rounds + validation + winner series)",
                       "three launches", "search appended to the lattice launch (default): two launches"}
    assert all(v > 0 for v in st.values())
    assert line["cold_start"]["ms_per_step"] > 0 and line["polygon_scenes"]["parity"]["index_exact"] and line["two_streams"]["config2"]["parity"]["index_exact"] and line["prewarm"]["steps"] > 0
    for planner in ("FOP", "FISS+"):
        cl = line["closed_loop"][planner]
        assert cl["value"] > 0 and cl["ego_plans"] >= 2048 and cl["parity"]["max_abs_cost_err"] <= 1e-6
    assert line["host_buffers"]["value"] > 0 and line["host_buffers"]["parity"]["index_exact"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline_1thread"]["cores"] == 1
    assert line["cpu_baseline"]["cores"] >= 1 and "thread_sweep" in line["cpu_baseline"] and "host" in line["cpu_baseline"]


def test_config5_full_batch_equals_its_eight_shards(oracle, engine):
    """BASELINE configs[4] at full size on one GPU."""
    B, W = 16384, 8
    full = synth.make_config(5, B=B)
    out = engine.plan_dense(full, tables=True)
    # (1) the multi-GPU contract: rank r generates + plans only its shard; concatenation == full batch
    for r in (0, 3, 7):
        part = synth.make_config(5, B=B // W, ego_offset=r * (B // W))
        po = engine.plan_dense(part, tables=False)
        sl = slice(r * (B // W), (r + 1) * (B // W))
        np.testing.assert_array_equal(po.best_idx, out.best_idx[sl])
        np.testing.assert_array_equal(po.best_cost, out.best_cost[sl])
    # (2) invariants for every ego: the winner is feasible and no feasible candidate is cheaper; N/M words are consistent
    ok = out.best_idx >= 0
    assert 0.3 < ok.mean() < 1.0
    feas = (out.flags & 7) == 0
    masked = np.where(feas, out.cost, np.inf)
    assert np.array_equal(masked.min(axis=1)[ok], out.best_cost[ok])
    assert np.array_equal(feas.any(axis=1), ok)
    # FOP keeps the LAST minimal index (frenet_optimal_planner.py:266)
    last_min = out.cost.shape[1] - 1 - np.argmin(masked[:, ::-1], axis=1)
    np.testing.assert_array_equal(out.best_idx[ok], last_min[ok])
    N, M = (out.flags >> 8) & 0xFFF, out.flags >> 20
    assert (M <= N).all() and (N >= 80).all() and (N <= 100).all()
    assert np.array_equal(((out.flags & 8) != 0), M < N)
    np.testing.assert_array_equal(out.stats, np.tile([0, full.C, full.C, full.C], (B, 1)))
    # (3) oracle sample: 24 egos of every shard
    egos = np.concatenate([np.arange(24) * 85 + r * (B // W) for r in range(W)])
    threads = len(os.sched_getaffinity(0))
    idx, cost = oracle.fop_plan_batch(oracle.problems_from_batch(full, egos), threads=threads)
    np.testing.assert_array_equal(out.best_idx[egos], idx)
    good = idx >= 0
    np.testing.assert_allclose(out.best_cost[egos][good], cost[good], rtol=0, atol=COST_TOL)
