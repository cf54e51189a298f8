#!/usr/bin/env python3
"""bench.py - throughput of the Frenet candidate hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): per GPU a batch of
2048 synthetic ego problems, 9x9x7 (d, v, T) lattice = 567 candidates per ego, 50 dynamic rectangle
obstacles with a 5 s / 50-step pose table, one 81-knot reference spline per ego.  One "step" = one
full pass of the hot path over the batch: lattice generation + cost + Frenet->Cartesian +
speed/acceleration masks + OBB collision + per-ego argmin (+ winner epilogue), inputs resident in HBM,
results (best index / cost per ego) copied to pinned host memory.  N > 1: every rank owns its own
2048-ego shard (weak scaling, no collectives on the data path; torch.distributed only for the
barrier and the max-over-ranks of the elapsed time).

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TF = 78.6   # MI355X FP64 vector peak (datasheet; the kernel's real bound)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--egos", type=int, default=2048, help="egos per GPU")
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[N-1]; 4 = the FISS+ pipeline (dense tables + search walk + 3 refinement rounds)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--tables", action="store_true", help="also write the dense cost/flag tables (materialised mode)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-ego plan-cycle latency leg (configs[0])")
    return ap.parse_args()


def algorithmic_bytes_per_ego(batch, tables: bool) -> float:
    """Bytes one ego problem must move through HBM once (DESIGN.md 'algorithmic bytes')."""
    nx = int(batch.nx.max())
    reads = 6 * 8 + 8 + 3 * 4 + batch.nv * 8 + 9 * 8 * nx            # ego, target speed, ids, v samples, spline
    if batch.n_obs:
        # only the pose rows has_collision() can query: steps t_now, t_now+stride, ... below min(final_time_step, T_obs)
        horizon = min(int(batch.final_time_step.max()) - int(batch.t_now.min()), batch.T_obs - int(batch.t_now.min()), 128)
        rows = max(0, (horizon + batch.check_stride - 1) // batch.check_stride)
        reads += 32 * rows * batch.n_obs + 16 * batch.n_obs + 4         # pose rows, dims, final_time_step
    # best idx, best cost, stats, winner flags + the winner's 16 series over its N points (the kernel pads them to 128 columns
    # with NaN: 16 KiB per ego reach HBM, the padding is not counted as algorithmic)
    writes = 4 + 8 + 16 + 4 + 16 * 8 * float(np.mean(batch.points_per_candidate()))
    if tables:
        writes += 12 * batch.C
    return float(reads + writes)


def flops_per_ego(batch) -> float:
    """Algorithmic FP64 flop count (SURVEY.md 8d accounting, upper bound: no early exit / broad phase)."""
    N = batch.points_per_candidate()
    pts = batch.nd * batch.nv * int(N.sum())
    horizon = min(int(batch.final_time_step.max()) if batch.n_obs else 0, int(N.max()))
    poses = batch.C * ((horizon + batch.check_stride - 1) // batch.check_stride)
    return 96.0 * pts + 60.0 * poses * batch.n_obs + 40.0 * batch.C


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # test hooks (a 1-GPU box can exercise the N>1 code path): BENCH_DIST_BACKEND=gloo, BENCH_ALL_ON_DEVICE0=1
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_ALL_ON_DEVICE0"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from fiss_plus_planner_amd import synth
    from fiss_plus_planner_amd.engine import FrenetEngine, device_batch, make_params

    # ---- this rank's shard, generated directly (every ego has its own RNG stream)
    batch = synth.make_config(args.config, B=args.egos, ego_offset=rank * args.egos)
    dev = torch.device("cuda", local_rank)
    names = ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
             "obs_pose", "obs_dims", "final_time_step")
    dten = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in names}
    fb = device_batch(batch, {k: (v.data_ptr() if v.numel() else 0) for k, v in dten.items()})
    params = make_params(batch)
    B, C = batch.B, batch.C
    # per-ego results, packed in one device buffer [cost f64 x B | index i32 x B] so that one async copy brings both to the host
    packed = torch.empty(12 * B, dtype=torch.uint8, device=dev)
    best_cost = packed[:8 * B].view(torch.float64)
    best_idx = packed[8 * B:].view(torch.int32)
    stats = torch.empty((B, 4), dtype=torch.int32, device=dev)
    cost_tbl = torch.empty((B, C), dtype=torch.float64, device=dev) if args.tables else None
    flag_tbl = torch.empty((B, C), dtype=torch.int32, device=dev) if args.tables else None
    best_flags = torch.empty(B, dtype=torch.int32, device=dev)
    best_traj = torch.empty((B, 16, 128), dtype=torch.float64, device=dev)   # winner epilogue output, stays in HBM
    h_packed = torch.empty(12 * B, dtype=torch.uint8).pin_memory()
    h_cost = h_packed[:8 * B].view(torch.float64)
    h_idx = h_packed[8 * B:].view(torch.int32)
    eng = FrenetEngine(local_rank)
    stream = torch.cuda.current_stream(dev)

    fiss = args.config == 4
    if fiss:
        from fiss_plus_planner_amd import _abi

        f_t = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in ("samp_min", "samp_max", "samp_res")}
        prev = torch.full((B, 3), -1, dtype=torch.int32, device=dev)
        ijk = torch.empty((B, 3), dtype=torch.int32, device=dev)
        end_state = torch.empty((B, 3), dtype=torch.float64, device=dev)
        refined = best_idx  # the per-ego int result of this mode (refined yes/no) rides in the packed buffer
        opts = _abi.FpFissOpts(_abi.FP_FISS_PLUS, 3, 10.0, 0.5)
        io = _abi.FpFissIo()
        io.samp_min, io.samp_max, io.samp_res = (f_t[k].data_ptr() for k in ("samp_min", "samp_max", "samp_res"))
        io.prev_best_idx, io.best_ijk, io.best_cost, io.end_state = prev.data_ptr(), ijk.data_ptr(), best_cost.data_ptr(), end_state.data_ptr()
        io.refined, io.stats, io.trace = refined.data_ptr(), stats.data_ptr(), None
        io.best_flags, io.best_traj = best_flags.data_ptr(), best_traj.data_ptr()

    def step_fiss():
        prev.fill_(-1)  # every step plans the same cycle: no history carried over
        eng.plan_fiss_device(params, fb, opts, io, stream=stream.cuda_stream)

    def step_dense():
        # one launch: lattice + argmin + the winner's series (what plan() returns) written by the workgroup that found it
        eng.plan_dense_device(params, fb, best_idx.data_ptr(), best_cost.data_ptr(), stats.data_ptr(),
                              cost_tbl.data_ptr() if args.tables else 0, flag_tbl.data_ptr() if args.tables else 0,
                              stream=stream.cuda_stream, best_flags=best_flags.data_ptr(), best_traj=best_traj.data_ptr())

    step = step_fiss if fiss else step_dense

    def epilogue():
        return  # both entry points produce the winner series themselves

    def fetch():
        h_packed.copy_(packed, non_blocking=True)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
        epilogue()
        fetch()
    barrier()

    # ---- timed region
    # HIP events bracket every 4th launch of the timed region (each recorded pair costs ~3 us of stream time)
    ev_every = 4 if args.steps >= 16 else 1
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range((args.steps + ev_every - 1) // ev_every)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        timed = k % ev_every == 0
        if timed:
            ev[k // ev_every][0].record(stream)
        step()                    # the dominant kernel: the events bracket exactly this launch
        if timed:
            ev[k // ev_every][1].record(stream)
        fetch()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # ---- parity gate + CPU baseline (rank 0, N=1 only): the oracle on a bounded sample of the same egos.
    # Runs AFTER the timed region (libgomp workers spin after a parallel region and would steal the launch thread's core);
    # a parity failure aborts before anything is printed.
    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and not fiss:
        from oracle import oracle as O

        O.build()
        cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        probe = O.problems_from_batch(batch, range(min(16, B)))
        t0 = time.perf_counter()
        O.fop_plan_batch(probe, threads=cores)
        per_ego = (time.perf_counter() - t0) / len(probe)
        n_s = int(max(16, min(B, args.cpu_seconds / max(per_ego, 1e-6))))
        probs = O.problems_from_batch(batch, range(n_s))
        t0 = time.perf_counter()
        o_idx, o_cost = O.fop_plan_batch(probs, threads=cores)
        dt = time.perf_counter() - t0
        g_idx, g_cost = h_idx.numpy()[:n_s], h_cost.numpy()[:n_s]
        if not np.array_equal(g_idx, o_idx):
            raise SystemExit(f"PARITY FAILURE: selected index differs on egos {np.nonzero(g_idx != o_idx)[0][:8].tolist()}")
        ok = o_idx >= 0
        if ok.any() and np.abs(g_cost[ok] - o_cost[ok]).max() > 1e-6:
            raise SystemExit("PARITY FAILURE: best cost differs by more than 1e-6")
        cpu_baseline = {"value": n_s * C / dt, "unit": "candidates/s", "cores": cores, "kind": "port",
                        "sample": f"first {n_s} egos of the same batch ({n_s * C} candidates), oracle/libfrenet_oracle.so with "
                                  f"OpenMP over egos, {dt:.1f} s; GPU index/cost parity checked on this sample before timing"}


    # ---- plan-cycle latency (rank 0, N=1): BASELINE configs[0] - single ego, FOP 5x5x5, DEU_Flensburg-1_1_T-1 closed loop,
    # timed around plan() exactly where the reference times it (planners/benchmark/planning.py:124-128).  Inputs are the
    # committed fixture arrays (centerline + the XML's 27 rectangle obstacles), see tests/golden/gen_golden.py:flensburg().
    plan_cycle = None
    fixture = os.path.join(ROOT, "tests", "golden", "g5_closed_loop.npz")
    if rank == 0 and world == 1 and not args.no_latency and os.path.exists(fixture):
        from fiss_plus_planner_amd import planners as P
        from fiss_plus_planner_amd.closed_loop import run_closed_loop
        from fiss_plus_planner_amd.obstacles import ObstacleTable
        from fiss_plus_planner_amd.vehicle import Vehicle

        g = np.load(fixture)
        fts = int(g["final_time_step"])
        table = ObstacleTable(g["obs_pose"][:fts], g["obs_dims"], fts)
        plan_cycle = {"config": "BASELINE.json configs[0]: single ego, DEU_Flensburg-1_1_T-1, 5x5x5 lattice, 27 dynamic obstacles, closed loop",
                      "unit": "ms", "timed": "wall time of plan() per cycle, B=1 through the host-buffer C ABI (H2D + kernels + D2H)"}
        for kind, cls, st in (("FOP", P.FrenetOptimalPlanner, P.FrenetOptimalPlannerSettings), ("FISS+", P.FissPlusPlanner, P.FissPlusPlannerSettings)):
            pl = cls(st(5, 5, 5), Vehicle(), None, engine=eng)
            run_closed_loop(pl, g["centerline"], g["init_state"], table, g["goal_center"], max_cycles=3)  # warm-up
            pl = cls(st(5, 5, 5), Vehicle(), None, engine=eng)
            res = run_closed_loop(pl, g["centerline"], g["init_state"], table, g["goal_center"])
            ms = res.plan_seconds * 1e3
            plan_cycle[kind] = {"p50": float(np.median(ms)), "p90": float(np.percentile(ms, 90)), "cycles": len(ms)}

    # ---- materialise mode (rank 0, N=1): the one HBM-bound mode of the path - every candidate's full series written out
    # (fp_materialize_all = the reference's all_trajs payload).  256 egos of the same batch: 2.4 GB per launch.
    materialize = None
    if rank == 0 and world == 1 and not args.no_latency and not fiss:
        Bm = min(B, 256)
        fbm = device_batch(batch, {k: (v.data_ptr() if v.numel() else 0) for k, v in dten.items()})
        fbm.B = Bm
        m_traj = torch.empty((Bm * C, 16, 128), dtype=torch.float64, device=dev)
        m_flags = torch.empty(Bm * C, dtype=torch.int32, device=dev)
        for _ in range(2):
            eng.materialize_all_device(params, fbm, m_flags.data_ptr(), m_traj.data_ptr(), stream=stream.cuda_stream)
        mev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b_ in mev:
            a.record(stream)
            eng.materialize_all_device(params, fbm, m_flags.data_ptr(), m_traj.data_ptr(), stream=stream.cuda_stream)
            b_.record(stream)
        torch.cuda.synchronize(dev)
        m_ms = float(np.median([a.elapsed_time(b_) for a, b_ in mev]))
        n_mean = float(batch.points_per_candidate().mean())
        written = Bm * C * (16 * 128 * 8 + 4)
        materialize = {"kernel": "winner_traj_kernel (all candidates)", "egos": Bm, "kernel_ms": m_ms, "candidates_per_s": Bm * C / (m_ms * 1e-3),
                       "bound": "hbm", "bytes_written_per_launch": written, "achieved_GBps": written / (m_ms * 1e-3) / 1e9,
                       "algorithmic_GBps": Bm * C * 16 * 8 * n_mean / (m_ms * 1e-3) / 1e9, "peak_GBps": HBM_PEAK_GBS,
                       "frac": written / (m_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del m_traj, m_flags

    if rank == 0:
        total_cand = world * B * (C + (21 if fiss else 0)) * args.steps  # config 4 counts the 21 refinement trajectories too
        value = total_cand / elapsed
        bytes_launch = algorithmic_bytes_per_ego(batch, args.tables) * B
        flops_launch = flops_per_ego(batch) * B
        ach_gbs = bytes_launch / (kern_ms * 1e-3) / 1e9
        ach_tf = flops_launch / (kern_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"config{args.config}_B{B}")
            except Exception:
                traffic = None
        # VALU issue: the resource that actually binds the lattice kernel.  Wave-level VALU instructions per launch come from the
        # committed PMC pass (profiles/, SQ_INSTS_VALU - a property of kernel + input, not of the run); the rate uses this run's
        # kernel time; the peak is one VALU instruction per SIMD every 4 cycles (wave64 on a 16-lane SIMD) at the boost clock.
        valu_issue = None
        ppath = os.path.join(ROOT, "profiles", "r01_config3_pmc_summary.json")
        if not fiss and args.config == 3 and B == 2048 and os.path.exists(ppath):
            try:
                pmc = json.load(open(ppath))
                insts = next(v["SQ_INSTS_VALU"] for k, v in pmc.items() if "lattice_fused" in k)
                peak = 256 * 4 * 2.4e9 / 4.0
                valu_issue = {"wave_instructions_per_launch": insts, "rate": insts / (kern_ms * 1e-3), "peak": peak,
                              "unit": "wave-instructions/s", "frac": insts / (kern_ms * 1e-3) / peak,
                              "source": "SQ_INSTS_VALU from profiles/r01_config3_pmc_summary.json (rocprofv3 --pmc), this run's kernel time; "
                                        "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction"}
            except Exception:
                valu_issue = None
        line = {
            "metric": "candidate trajectories/sec (gen+cost+collision) per GPU; plan-cycle p50 latency", "value": value, "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{args.config - 1}]: {B} egos/GPU x {batch.nd}x{batch.nv}x{batch.nt} lattice "
                                   f"({C} cand/ego), {batch.n_obs} {'dynamic' if batch.meta.get('moving') else 'static'} obstacles, "
                                   f"T_obs={batch.T_obs}, stride-2 OBB checks, " + ("FISS+ search + 3 refinement rounds" if fiss else "FOP argmin"),
                       "egos_per_gpu": B, "candidates_per_ego": C, "tables_written": bool(args.tables),
                       "parallelism": f"ego-shard x{world} (no collectives)", "input_digest": batch.digest()[:16]},
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "lattice_fused_kernel (lattice + argmin + winner series)" if not fiss else "lattice_fused + fiss_search + fiss_refine + winner_traj (whole pipeline)",
                         "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "note": "the kernel is VALU-issue bound, not HBM bound: see valu_issue (executed instructions) and valu_fp64"},
            "valu_fp64": {"reference_algorithm_rate": ach_tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                          "reference_algorithm_flops_per_launch": flops_launch,
                          "note": "flops the reference's per-candidate algorithm would need (SURVEY 8d accounting) / kernel time; "
                                  "the kernel executes far fewer (profile sharing, broad phase) - executed-instruction "
                                  "counts from rocprofv3 PMC are in DESIGN.md"},
            "valu_issue": valu_issue,
            "cpu_baseline": cpu_baseline,
            "plan_cycle_latency": plan_cycle,
            "materialize_mode": materialize,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
