#!/usr/bin/env python3
"""bench.py - throughput of the Frenet candidate hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload at N = 1 (BASELINE.json configs[2], the configuration the metric is quoted on): a batch of 2048 synthetic ego
problems, 9x9x7 (d, v, T) lattice = 567 candidates per ego, 50 dynamic rectangle obstacles with a 5 s / 50-step pose
table, one 81-knot reference spline per ego.  At N > 1 (BASELINE.json configs[4]) every rank owns egos
[rank*2048, (rank+1)*2048) of the 16384-ego batch of the same generator: weak scaling, no collective on the data path
(torch.distributed only carries the barrier and the max-over-ranks of the elapsed time).  `--gpus N` without a launcher
spawns the N ranks itself (python -m torch.distributed.run); under a launcher WORLD_SIZE must equal N.

One "step" = one full pass of the hot path over the batch: lattice generation + cost + Frenet->Cartesian + speed /
acceleration masks + OBB collision + per-ego argmin + the winner's series, inputs resident in HBM, results (best index /
cost per ego) copied to pinned host memory.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`; at N = 1 the line
also carries the other single-GPU configurations (`config2`, `config4`), the index-order and rotating-batch variants of
the headline workload, and the SURVEY 8d obstacle layout.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# The CPU-baseline leg runs the oracle with one OpenMP thread per host core; by default the runtime leaves those threads spinning
# for a while after the parallel region, next to the thread that launches the workloads measured after it.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TF = 78.6   # MI355X FP64 vector peak (datasheet)
BATCH_ARRAYS = ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
                "obs_pose", "obs_dims", "final_time_step")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--egos", type=int, default=2048, help="egos per GPU")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="BASELINE.json configs[N-1]; default: 3 on one GPU, 5 (the sharded 16384-ego batch) on several; "
                         "4 = the FISS+ pipeline (dense tables + search walk + 3 refinement rounds)")
    ap.add_argument("--layout", default="lanes", choices=["lanes", "survey8d"], help="obstacle layout of the synthetic scenes (synth.py)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--tables", action="store_true", help="also write the dense cost/flag tables (materialised mode)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-ego plan-cycle latency leg (configs[0]) and the materialise leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra single-GPU workloads (config2, config4, index order, rotating batches, survey layout)")
    return ap.parse_args()


def spawn_ranks(args) -> None:
    """`python bench.py --gpus N` without a launcher: re-run this file as N ranks of one node, one rank per GPU (RCCL)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


def read_bytes_per_ego(batch) -> float:
    """Bytes one ego problem must READ from HBM once (DESIGN.md 'algorithmic bytes')."""
    nx = int(batch.nx.max())
    reads = 6 * 8 + 8 + 3 * 4 + batch.nv * 8 + 9 * 8 * nx            # ego, target speed, ids, v samples, spline
    if batch.n_obs:
        # only the pose rows has_collision() can query: steps t_now, t_now+stride, ... below min(final_time_step, T_obs)
        horizon = min(int(batch.final_time_step.max()) - int(batch.t_now.min()), batch.T_obs - int(batch.t_now.min()), 128)
        rows = max(0, (horizon + batch.check_stride - 1) // batch.check_stride)
        reads += 32 * rows * batch.n_obs + 16 * batch.n_obs + 4         # pose rows, dims, final_time_step
    return float(reads)


def series_bytes(best_flags: np.ndarray, lines: bool = False) -> float:
    """Bytes of the winners' series: 9 Frenet rows of N points, x / y / yaw of M, ds / c of M-1, c_d of M-2, c_dd of M-3 - only
    for egos that HAVE a winner (flag word != 0).  lines=True: every row rounded up to whole 128-byte lines (what the sparse
    layout of the ABI stores)."""
    fl = best_flags[best_flags != 0].astype(np.int64)
    N, M = (fl >> 8) & 0xFFF, fl >> 20
    ln = (lambda a: (np.maximum(a, 0) + 15) // 16 * 16) if lines else (lambda a: np.maximum(a, 0))
    M = np.where(M >= 2, M, 0) if lines else M
    return float(8 * (9 * ln(N) + 3 * ln(M) + 2 * ln(M - 1) + ln(M - 2) + ln(M - 3)).sum())


class Workload:
    """One problem batch resident in HBM + its output buffers + the step that plans it."""

    def __init__(self, torch, eng, batch, dev, stream, fiss=False, tables=False):
        from fiss_plus_planner_amd import _abi
        from fiss_plus_planner_amd.engine import device_batch, make_params

        self.torch, self.eng, self.batch, self.dev, self.stream, self.fiss, self.tables = torch, eng, batch, dev, stream, fiss, tables
        self.dten = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in BATCH_ARRAYS}
        self.fb = device_batch(batch, {k: (v.data_ptr() if v.numel() else 0) for k, v in self.dten.items()})
        self.params = make_params(batch)
        B, C = batch.B, batch.C
        # per-ego results, packed in one device buffer [cost f64 x B | index i32 x B] so that one async copy brings both to the host
        self.packed = torch.empty(12 * B, dtype=torch.uint8, device=dev)
        self.best_cost = self.packed[:8 * B].view(torch.float64)
        self.best_idx = self.packed[8 * B:].view(torch.int32)
        self.stats = torch.empty((B, 4), dtype=torch.int32, device=dev)
        self.cost_tbl = torch.empty((B, C), dtype=torch.float64, device=dev) if tables else None
        self.flag_tbl = torch.empty((B, C), dtype=torch.int32, device=dev) if tables else None
        self.best_flags = torch.zeros(B, dtype=torch.int32, device=dev)
        # winner epilogue output, stays in HBM.  Compact layout of the ABI (fp_result.traj_stride / traj_sparse): rows of
        # ceil(max T / tick) columns, only the elements that exist are written - the bytes written ARE the algorithmic bytes
        self.traj_stride = (int(np.ceil(batch.t_samples.max() / batch.tick_t)) + 15) // 16 * 16   # whole 128-byte lines per row
        self.best_traj = torch.empty((B, 16, self.traj_stride), dtype=torch.float64, device=dev)
        self.h_packed = torch.empty(12 * B, dtype=torch.uint8).pin_memory()
        self.h_cost = self.h_packed[:8 * B].view(torch.float64)
        self.h_idx = self.h_packed[8 * B:].view(torch.int32)
        if fiss:
            self.f_t = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in ("samp_min", "samp_max", "samp_res")}
            self.prev = torch.full((B, 3), -1, dtype=torch.int32, device=dev)
            self.ijk = torch.empty((B, 3), dtype=torch.int32, device=dev)
            self.end_state = torch.empty((B, 3), dtype=torch.float64, device=dev)
            self.opts = _abi.FpFissOpts(_abi.FP_FISS_PLUS, 3, 10.0, 0.5)
            io = self.io = _abi.FpFissIo()
            io.samp_min, io.samp_max, io.samp_res = (self.f_t[k].data_ptr() for k in ("samp_min", "samp_max", "samp_res"))
            io.prev_best_idx, io.best_ijk, io.best_cost, io.end_state = self.prev.data_ptr(), self.ijk.data_ptr(), self.best_cost.data_ptr(), self.end_state.data_ptr()
            # the per-ego int result of this mode (refined yes/no) rides in the packed buffer
            io.refined, io.stats, io.trace = self.best_idx.data_ptr(), self.stats.data_ptr(), None
            io.best_flags, io.best_traj = self.best_flags.data_ptr(), self.best_traj.data_ptr()
            io.traj_stride, io.traj_sparse = self.traj_stride, 1

    @property
    def candidates(self) -> int:
        return self.batch.B * (self.batch.C + (21 if self.fiss else 0))  # FISS+ counts its 21 refinement trajectories too

    def step(self):
        if self.fiss:
            self.prev.fill_(-1)  # every step plans the same cycle: no history carried over
            self.eng.plan_fiss_device(self.params, self.fb, self.opts, self.io, stream=self.stream.cuda_stream)
        else:
            # one launch: lattice + argmin + the winner's series (what plan() returns) written by the workgroup that found it
            self.eng.plan_dense_device(self.params, self.fb, self.best_idx.data_ptr(), self.best_cost.data_ptr(), self.stats.data_ptr(),
                                       self.cost_tbl.data_ptr() if self.tables else 0, self.flag_tbl.data_ptr() if self.tables else 0,
                                       stream=self.stream.cuda_stream, best_flags=self.best_flags.data_ptr(), best_traj=self.best_traj.data_ptr(),
                                       traj_stride=self.traj_stride, traj_sparse=True)

    def fetch(self):
        self.h_packed.copy_(self.packed, non_blocking=True)

    def algorithmic_bytes(self) -> float:
        """Per launch, exact for this batch's results: reads + per-ego results + the series of the egos that have a winner."""
        B, C = self.batch.B, self.batch.C
        writes = B * (4 + 8 + 16 + 4) + series_bytes(self.best_flags.cpu().numpy().view(np.uint32))
        if self.fiss:
            writes += B * (12 + 24 + 12)  # best_ijk, end_state, prev_best_idx
        if self.tables:
            writes += 12 * B * C
        return read_bytes_per_ego(self.batch) * B + writes


def timed_run(torch, workloads, steps, warmup, stream, barrier, ev_every=None):
    """W untimed + K timed steps cycling through `workloads`; returns (elapsed s, per-launch kernel ms list from HIP events on the
    launch stream).  Events bracket every ev_every-th launch (a recorded pair costs ~3 us of stream time)."""
    n = len(workloads)
    for k in range(warmup):
        workloads[k % n].step()
        workloads[k % n].fetch()
    barrier()
    if ev_every is None:
        ev_every = 4 if steps >= 16 else 1
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range((steps + ev_every - 1) // ev_every)]
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        w = workloads[(warmup + k) % n]
        timed = k % ev_every == 0
        if timed:
            ev[k // ev_every][0].record(stream)
        w.step()                    # the events bracket exactly this entry point's launches
        if timed:
            ev[k // ev_every][1].record(stream)
        w.fetch()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, [a.elapsed_time(b) for a, b in ev]


def roofline_obj(bytes_launch, kern_ms, kernel, traffic=None, note=None):
    ach = bytes_launch / (kern_ms * 1e-3) / 1e9
    o = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
         "kernel": kernel, "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_launch}
    if note:
        o["note"] = note
    return o


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path))
    except Exception:
        return None


def cpu_baseline_leg(batch, h_idx, h_cost, seconds, threads, gate):
    """The oracle (plain-C restatement, OpenMP over egos) on a bounded sample of the same egos.  gate: its indices / costs must
    equal the GPU's (index exact, cost <= 1e-6) or the bench aborts before printing anything."""
    from oracle import oracle as O

    O.build()
    B, C = batch.B, batch.C
    probe = O.problems_from_batch(batch, range(min(2 * threads, 16, B)))
    t0 = time.perf_counter()
    O.fop_plan_batch(probe, threads=threads)
    per_ego = (time.perf_counter() - t0) / len(probe)
    n_s = int(max(min(16, B), min(B, seconds / max(per_ego, 1e-6))))
    probs = O.problems_from_batch(batch, range(n_s))
    t0 = time.perf_counter()
    o_idx, o_cost = O.fop_plan_batch(probs, threads=threads)
    dt = time.perf_counter() - t0
    if gate:
        g_idx, g_cost = h_idx[:n_s], h_cost[:n_s]
        if not np.array_equal(g_idx, o_idx):
            raise SystemExit(f"PARITY FAILURE: selected index differs on egos {np.nonzero(g_idx != o_idx)[0][:8].tolist()}")
        ok = o_idx >= 0
        if ok.any() and np.abs(g_cost[ok] - o_cost[ok]).max() > 1e-6:
            raise SystemExit("PARITY FAILURE: best cost differs by more than 1e-6")
    return {"value": n_s * C / dt, "unit": "candidates/s", "cores": threads, "kind": "port",
            "sample": f"first {n_s} egos of the same batch ({n_s * C} candidates), oracle/libfrenet_oracle.so, {threads} OpenMP thread(s) over egos, "
                      f"{dt:.1f} s" + ("; GPU index/cost parity checked on this sample" if gate else "")}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    config = args.config or (3 if world == 1 else 5)
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
        assert dist.get_world_size() == args.gpus

    from fiss_plus_planner_amd import synth
    from fiss_plus_planner_amd.engine import FrenetEngine

    # ---- this rank's shard, generated directly (every ego has its own RNG stream)
    fiss = config == 4
    batch = synth.make_config(config, B=args.egos, ego_offset=rank * args.egos, layout=args.layout)
    dev = torch.device("cuda", local_rank)
    B, C = batch.B, batch.C
    eng = FrenetEngine(local_rank)
    for kv in filter(None, os.environ.get("BENCH_CTX_OPTIONS", "").split(",")):  # diagnostic: "name=value,..." -> fp_ctx_set_option
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    stream = torch.cuda.current_stream(dev)
    main_wl = Workload(torch, eng, batch, dev, stream, fiss=fiss, tables=args.tables)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()

    # ---- timed region (the contract: W warm-up steps, exactly K timed steps between barrier + synchronize)
    elapsed, kern_list = timed_run(torch, [main_wl], args.steps, args.warmup, stream, barrier)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean(kern_list))
    solo = rank == 0 and world == 1

    # ---- parity gate + CPU baseline (rank 0, N=1 only): the oracle on a bounded sample of the same egos.
    # Runs AFTER the timed region (libgomp workers spin after a parallel region and would steal the launch thread's core);
    # a parity failure aborts before anything is printed.
    cpu_baseline = cpu_1t = None
    if solo and args.cpu_seconds > 0 and not fiss:
        cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        cpu_baseline = cpu_baseline_leg(batch, main_wl.h_idx.numpy(), main_wl.h_cost.numpy(), args.cpu_seconds, cores, gate=True)
        cpu_1t = cpu_baseline_leg(batch, main_wl.h_idx.numpy(), main_wl.h_cost.numpy(), min(6.0, args.cpu_seconds), 1, gate=True)

    # ---- plan-cycle latency (rank 0, N=1): BASELINE configs[0] - single ego, FOP 5x5x5, DEU_Flensburg-1_1_T-1 closed loop,
    # timed around plan() exactly where the reference times it (planners/benchmark/planning.py:124-128).  Inputs are the
    # committed fixture arrays (centerline + the XML's 27 rectangle obstacles), see tests/golden/gen_golden.py:flensburg().
    plan_cycle = None
    fixture = os.path.join(ROOT, "tests", "golden", "g5_closed_loop.npz")
    if solo and not args.no_latency and os.path.exists(fixture):
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
    if solo and not args.no_latency and not fiss:
        from fiss_plus_planner_amd.engine import device_batch

        Bm = min(B, 256)
        fbm = device_batch(batch, {k: (v.data_ptr() if v.numel() else 0) for k, v in main_wl.dten.items()})
        fbm.B = Bm
        m_flags = torch.empty(Bm * C, dtype=torch.int32, device=dev)
        materialize = {"kernel": "winner_traj_kernel (all candidates)", "egos": Bm, "bound": "hbm", "peak_GBps": HBM_PEAK_GBS}
        # two layouts of the same payload: the compact one (rows of ceil(max T / tick) columns, only existing elements written) and
        # the round-1 layout (16 x 128 NaN-padded block per candidate)
        for label, m_stride, m_sparse in (("compact", main_wl.traj_stride, True), ("padded128", 128, False)):
            m_traj = torch.empty((Bm * C, 16, m_stride), dtype=torch.float64, device=dev)
            kw = dict(stream=stream.cuda_stream, traj_stride=m_stride, traj_sparse=m_sparse)
            for _ in range(2):
                eng.materialize_all_device(main_wl.params, fbm, m_flags.data_ptr(), m_traj.data_ptr(), **kw)
            mev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(11)]
            for a, b_ in mev:
                a.record(stream)
                eng.materialize_all_device(main_wl.params, fbm, m_flags.data_ptr(), m_traj.data_ptr(), **kw)
                b_.record(stream)
            torch.cuda.synchronize(dev)
            m_all = [a.elapsed_time(b_) for a, b_ in mev]
            m_ms = float(np.median(m_all))
            fl_m = m_flags.cpu().numpy().view(np.uint32)
            alg = series_bytes(fl_m) + 4 * Bm * C
            written = (series_bytes(fl_m, lines=True) + 4 * Bm * C) if m_sparse else Bm * C * (16 * m_stride * 8 + 4)
            materialize[label] = {"kernel_ms": m_ms, "kernel_ms_min": float(np.min(m_all)), "launches_timed": len(m_all), "candidates_per_s": Bm * C / (m_ms * 1e-3), "bytes_written_per_launch": written,
                                  "algorithmic_bytes_per_launch": alg, "achieved_GBps": written / (m_ms * 1e-3) / 1e9,
                                  "algorithmic_GBps": alg / (m_ms * 1e-3) / 1e9, "frac": written / (m_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "algorithmic_over_written": alg / written}
            del m_traj
        del m_flags

    # ---- the other single-GPU configurations and variants of the headline workload (rank 0, N=1), each timed like the main run
    extras = {}
    if solo and not args.no_extras and not fiss and config == 3:
        steps_x, warm_x = min(args.steps, 50), min(args.warmup, 5)

        def measure(wls, label_kernel, ev_every=None):
            el, kl = timed_run(torch, wls, steps_x, warm_x, stream, barrier, ev_every)
            k_ms = float(np.mean(kl))
            cand = sum(w.candidates for w in wls) / len(wls)
            bytes_l = float(np.mean([w.algorithmic_bytes() for w in wls]))
            return {"value": cand * steps_x / el, "unit": "candidates/s", "ms_per_step": el / steps_x * 1e3, "steps": steps_x, "warmup": warm_x,
                    "roofline": roofline_obj(bytes_l, k_ms, label_kernel)}

        # (a) index order: the same launches without the feedback-directed launch order
        ordered = eng.get_option("lattice_ordered_launches")
        launches = eng.get_option("lattice_launches")
        eng.set_option("lattice_order", 0)
        extras["lattice_order_off"] = measure([main_wl], "lattice_fused_kernel, workgroups dispatched in ego index order")
        eng.set_option("lattice_order", 1)
        extras["lattice_order_on"] = {"launches_of_the_main_run": launches, "of_them_in_feedback_order": ordered,
                                      "note": "the main run replays one batch, the best case for the order predictor; lattice_order_off and "
                                              "rotating_batches are the other two points"}
        # (b) rotating batches: 4 distinct 2048-ego batches cycled, feedback order on (it is keyed on the batch size only, so every
        # launch is dispatched in the order of an EARLIER, different batch)
        rot = [main_wl] + [Workload(torch, eng, synth.make_config(3, B=B, ego_offset=(i + 1) * B, layout=args.layout), dev, stream) for i in range(3)]
        extras["rotating_batches"] = dict(measure(rot, "lattice_fused_kernel, 4 distinct batches cycled (stale feedback order)"),
                                          batches=len(rot), egos_per_batch=B)
        del rot
        # (c) BASELINE configs[1]: 256 egos x 5x5x5 x 10 static obstacles
        b2 = synth.make_config(2)
        w2 = Workload(torch, eng, b2, dev, stream)
        extras["config2"] = dict(measure([w2], "lattice_fused_kernel (latency mode: slices of an ego spread over workgroups)"),
                                 workload=f"BASELINE.json configs[1]: {b2.B} egos x 5x5x5 lattice ({b2.C} cand/ego), {b2.n_obs} static obstacles, T_obs={b2.T_obs}")
        del w2
        # (d) BASELINE configs[3]: the FISS+ pipeline on 2048 egos; per-stage times from runs that stop after stage 1 / 2
        b4 = synth.make_config(4, B=B)
        w4 = Workload(torch, eng, b4, dev, stream, fiss=True)
        o4 = measure([w4], "lattice_fused + fiss_search + fiss_refine (whole FISS+ pipeline, 3 kernels)")
        stage_ms = {}
        for st_n in (1, 2):
            eng.set_option("fiss_stages", st_n)
            _, kl = timed_run(torch, [w4], 20, 3, stream, barrier, ev_every=1)
            stage_ms[st_n] = float(np.mean(kl))
        eng.set_option("fiss_stages", 3)
        w4.step(); torch.cuda.synchronize(dev)  # leave complete outputs behind
        k4 = o4["roofline"]["kernel_ms"]
        o4["stage_ms"] = {"lattice_fused_kernel (dense tables)": stage_ms[1], "fiss_search_kernel": stage_ms[2] - stage_ms[1],
                          "fiss_refine_kernel (3 rounds + validation + winner series)": k4 - stage_ms[2]}
        o4["workload"] = f"BASELINE.json configs[3]: FISS+ (search walk + 3 refinement rounds) over {b4.B} egos x 9x9x7, 50 dynamic obstacles; counts C + 21 trajectories per ego"
        extras["config4"] = o4
        del w4
        # (e) SURVEY 8d obstacle layout verbatim (d_o ~ U(-4, 4), s_o = s + U(8, 120), speed U(0, 12))
        b8 = synth.make_config(3, B=B, layout="survey8d")
        w8 = Workload(torch, eng, b8, dev, stream)
        o8 = measure([w8], "lattice_fused_kernel")
        idx8 = w8.h_idx.numpy()
        o8["workload"] = "configs[2] sizes with the SURVEY 8d obstacle layout verbatim (every obstacle at d_o ~ U(-4, 4) around the ego lane)"
        o8["egos_with_a_feasible_candidate"] = float((idx8 >= 0).mean())
        extras["survey8d_layout"] = o8
        del w8

    if rank == 0:
        value = world * main_wl.candidates * args.steps / elapsed
        bytes_launch = main_wl.algorithmic_bytes()
        tr = load_profile_json("traffic.json") or {}
        traffic = tr.get(f"config{config}_B{B}")
        # executed VALU work of the dominant kernel from the committed PMC pass (a property of kernel + input, not of the run);
        # the rates use this run's kernel time
        valu_issue = fp64_exec = None
        pmc = load_profile_json("r02_config3_pmc_summary.json") or load_profile_json("r01_config3_pmc_summary.json")
        if pmc and not fiss and config == 3 and B == 2048 and args.layout == "lanes":
            try:
                k = next(v for kk, v in pmc.items() if "lattice_fused" in kk)
                insts = k["SQ_INSTS_VALU"]
                peak = 256 * 4 * 2.4e9 / 4.0
                valu_issue = {"wave_instructions_per_launch": insts, "rate": insts / (kern_ms * 1e-3), "peak": peak,
                              "unit": "wave-instructions/s", "frac": insts / (kern_ms * 1e-3) / peak,
                              "source": "SQ_INSTS_VALU from profiles/*_config3_pmc_summary.json (rocprofv3 --pmc), this run's kernel time; "
                                        "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction"}
                flops = 64.0 * (2 * k["SQ_INSTS_VALU_FMA_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + k["SQ_INSTS_VALU_ADD_F64"] + k["SQ_INSTS_VALU_TRANS_F64"])
                fp64_exec = {"executed_flops_per_launch": flops, "rate": flops / (kern_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                             "frac": flops / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                             "source": "64 lanes x (2 FMA + MUL + ADD + TRANS) wave-level FP64 instructions from the same PMC pass (inactive lanes counted: upper bound)"}
            except Exception:
                valu_issue = fp64_exec = None
        kname = ("lattice_fused_kernel (lattice + argmin; 94 % of the time) + winner_traj_kernel (the winners' series): the two launches of one "
                 "fp_plan_dense call") if not fiss else "lattice_fused + fiss_search + fiss_refine (whole pipeline)"
        line = {
            "metric": "candidate trajectories/sec (gen+cost+collision) per GPU; plan-cycle p50 latency", "value": value, "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{config - 1}]: {B} egos/GPU x {batch.nd}x{batch.nv}x{batch.nt} lattice "
                                   f"({C} cand/ego), {batch.n_obs} {'dynamic' if batch.meta.get('moving') else 'static'} obstacles, "
                                   f"T_obs={batch.T_obs}, stride-2 OBB checks, " + ("FISS+ search + 3 refinement rounds" if fiss else "FOP argmin")
                                   + (f"; rank r plans egos [r*{B}, (r+1)*{B}) of the {world * B}-ego batch" if world > 1 else ""),
                       "egos_per_gpu": B, "candidates_per_ego": C, "tables_written": bool(args.tables), "obstacle_layout": args.layout,
                       "parallelism": f"ego-shard x{world} (no collectives)", "input_digest": batch.digest()[:16]},
            "roofline": roofline_obj(bytes_launch, kern_ms, kname, traffic,
                                     "the kernel is VALU-issue bound, not HBM bound: see valu_issue / valu_fp64_executed (PMC instruction counts)"),
            "kernel_ms_stats": {"mean": kern_ms, "min": float(np.min(kern_list)), "median": float(np.median(kern_list)), "max": float(np.max(kern_list)),
                                "launches_timed": len(kern_list)},
            "valu_issue": valu_issue,
            "valu_fp64_executed": fp64_exec,
            "cpu_baseline": cpu_baseline,
            "cpu_baseline_1thread": cpu_1t,
            "plan_cycle_latency": plan_cycle,
            "materialize_mode": materialize,
        }
        line.update(extras)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
