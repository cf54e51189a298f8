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
cost per ego) written by the kernels into pinned host memory, the winners' series into HBM.

The synthetic scenes come from SURVEY.md section 8(d)'s generator verbatim (`--layout survey8d`, synth.py), and the timed
region cycles through `--rotate` (default 4) DISTINCT batches of that generator, so no launch ever sees the batch the
feedback-directed launch order was learnt on.

Prints ONE compact JSON line (rank 0; < 4 KB, the last and only thing on stdout) with the driver's contract fields plus
`roofline`, `cpu_baseline`, a short `parity` and `legs` (ms per step + parity verdict of the other single-GPU legs).  The
full record of the run - the other configurations (`config2`, `config4`), the index-order and single-batch variants of
the headline workload, the builder's `lanes` obstacle layout, the device-resident closed loop (`closed_loop`), the
PCIe-inclusive host-buffer entry (`host_buffers`), every leg's complete `parity` object - goes to `bench_extras.json`
next to this file ($BENCH_EXTRAS_FILE overrides the path).  EVERY leg that reports a number was compared with the CPU
oracle in this run (index / Stats exact, cost <= 1e-6) and a mismatch aborts before anything is printed.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TF = 78.6   # MI355X FP64 vector peak (datasheet)
BATCH_ARRAYS = ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
                "obs_pose", "obs_dims", "final_time_step")


# ---- the ONE stdout line.  The driver keeps the tail of stdout and parses its last line; a line that outgrows its reader leaves the
# round unmeasured (round 5's 24.7 KB line did).  So the line is the contract's keys and nothing else, scalars only below the first
# level, strings short, < LINE_LIMIT bytes, strict JSON (no NaN / Infinity); every other leg of the run goes, in full, to the side
# file EXTRAS_FILE (bench_extras.json next to this file, or $BENCH_EXTRAS_FILE).
LINE_LIMIT = 4096
EXTRAS_FILE = "bench_extras.json"
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline", "parity", "value_cold", "ms_per_step_cold", "plan_cycle_p50_ms", "legs", "extras_file")
LEG_KEYS = ("config2", "config4", "config5_single_gpu", "launch_order_hint_off", "launch_order_hint_only", "lattice_order_off", "single_batch_replayed", "tables_written",
            "lanes_layout", "survey8d_layout", "polygon_scenes", "two_streams", "overlap", "overlap_config4", "overlap_config2", "sharded_resident", "long_reference_lines", "rectangles_as_rings")


def _sig(x, digits=6):
    """A float at `digits` significant digits (None / non-finite -> None: strict JSON has no NaN)."""
    if x is None:
        return None
    x = float(x)
    return float(f"{x:.{digits}g}") if math.isfinite(x) else None


def _parity_ok(p):
    """One boolean out of a leg's parity object(s): every exactness flag true and every cost error within its tolerance."""
    if p is None:
        return None
    if isinstance(p, (list, tuple)):
        oks = [_parity_ok(q) for q in p]
        return None if any(o is None for o in oks) else all(oks)
    if not isinstance(p, dict):
        return None
    if "batches" in p:
        return _parity_ok(p["batches"])
    flags = [v for k, v in p.items() if k.endswith("_exact")]
    err = p.get("max_abs_cost_err")
    return bool(all(flags) and (err is None or err <= p.get("cost_tolerance", COST_TOL)))


def contract_line(full: dict) -> dict:
    """The compact line the driver reads, built from the run's full record (what EXTRAS_FILE holds)."""
    cfg, roof, cpu = full.get("config") or {}, full.get("roofline") or {}, full.get("cpu_baseline")
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _sig(line["value"], 9), _sig(line["ms_per_step"], 7)
    line["config"] = {"workload": str(cfg.get("workload_short") or cfg.get("workload", ""))[:200], "egos_per_gpu": cfg.get("egos_per_gpu"),
                      "candidates_per_ego": cfg.get("candidates_per_ego"), "parallelism": cfg.get("parallelism")}
    fp64 = roof.get("valu_fp64") or {}
    line["roofline"] = {"bound": roof.get("bound"), "achieved": _sig(roof.get("achieved")), "peak": roof.get("peak"), "unit": roof.get("unit"),
                        "frac": _sig(roof.get("frac")), "traffic": _sig(roof.get("traffic"), 9),
                        "algorithmic_bytes_per_launch": _sig(roof.get("algorithmic_bytes_per_launch"), 9),
                        "kernel": str(roof.get("kernel_short") or roof.get("kernel", ""))[:100], "kernel_ms": _sig(roof.get("kernel_ms")),
                        "binding": roof.get("binding"), "valu_fp64_tflops": _sig(fp64.get("achieved")), "valu_fp64_peak": fp64.get("peak"),
                        "valu_fp64_frac": _sig(fp64.get("frac"))}
    line["cpu_baseline"] = None if not cpu else {"value": _sig(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                                                 "sample": str(cpu.get("sample_short") or cpu.get("sample", ""))[:110]}
    par = full.get("parity")
    if par:
        rows = par["batches"] if "batches" in par else [par]
        errs = [r.get("max_abs_cost_err") for r in rows if r.get("max_abs_cost_err") is not None]
        line["parity"] = {"checked_egos": int(sum(r.get("checked_egos", 0) for r in rows)), "index_exact": _parity_ok(par),
                          "max_abs_cost_err": _sig(max(errs)) if errs else None, "cost_tolerance": COST_TOL,
                          "series_checked": int(sum(r.get("series_checked", 0) for r in rows))}
    else:
        line["parity"] = None
    line["value_cold"], line["ms_per_step_cold"] = _sig(full.get("value_cold"), 9), _sig(full.get("ms_per_step_cold"), 7)
    pc = full.get("plan_cycle_latency") or {}
    line["plan_cycle_p50_ms"] = {k: _sig(pc[k]["p50"]) for k in ("FOP", "FISS+") if isinstance(pc.get(k), dict)} or None
    legs = {}
    for k in LEG_KEYS:
        leg = full.get(k)
        if isinstance(leg, dict) and "ms_per_step" in leg:
            legs[k] = {"ms_per_step": _sig(leg["ms_per_step"]), "parity_ok": _parity_ok(leg.get("parity"))}
    if isinstance(full.get("config4"), dict) and "stage_ms" in full["config4"]:
        st = full["config4"]["stage_ms"]
        legs["config4"]["refine_ms"] = _sig(next((v for kk, v in st.items() if kk.startswith("fiss_refine")), None))
    ts = full.get("two_streams") or {}
    for k in ("config2", "config4"):
        if isinstance(ts.get(k), dict):
            legs[f"two_streams_{k}"] = {"ms_per_step": _sig(ts[k]["ms_per_step"]), "parity_ok": _parity_ok(ts[k].get("parity"))}
    cl = full.get("closed_loop") or {}
    for k in ("FOP", "FISS+"):
        if isinstance(cl.get(k), dict):
            legs[f"closed_loop_{k}"] = {"us_per_cycle": _sig(cl[k].get("us_per_cycle")), "parity_ok": _parity_ok(cl[k].get("parity"))}
    for k in ("host_buffers", "host_buffers_tagged"):
        if isinstance(full.get(k), dict):
            legs[k] = {"ms_per_call": _sig(full[k].get("ms_per_call")), "parity_ok": _parity_ok(full[k].get("parity"))}
    line["legs"] = legs or None
    line["extras_file"] = full.get("extras_file")
    assert tuple(line) == CONTRACT_KEYS
    return line


def dump_line(line: dict) -> str:
    """Strict, compact JSON of the contract line; refuses a line the driver's reader could not take."""
    text = json.dumps(line, allow_nan=False, separators=(", ", ": "))
    json.loads(text)
    if len(text.encode()) >= LINE_LIMIT:
        raise SystemExit(f"bench.py: the contract line is {len(text.encode())} bytes (limit {LINE_LIMIT}); move fields to {EXTRAS_FILE}")
    return text


def _finite(o):
    """The full record with non-finite floats replaced by None (the side file is strict JSON too)."""
    if isinstance(o, dict):
        return {str(k): _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, (float, np.floating)):
        return float(o) if math.isfinite(float(o)) else None
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, np.bool_):
        return bool(o)
    return o


def emit(full: dict) -> None:
    """Write the full record to the side file, then print the contract line - the LAST and only thing this process writes to stdout."""
    path = os.environ.get("BENCH_EXTRAS_FILE") or os.path.join(ROOT, EXTRAS_FILE)
    full = _finite(full)
    full["extras_file"] = os.path.basename(path)
    text = dump_line(contract_line(full))
    try:
        with open(path + ".tmp", "w") as f:
            json.dump(full, f, allow_nan=False, indent=1)
        os.replace(path + ".tmp", path)
    except OSError as e:  # a read-only checkout must not cost the round its number
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    sys.stdout.flush()
    print(text, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--egos", type=int, default=2048, help="egos per GPU")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="BASELINE.json configs[N-1]; default: 3 on one GPU, 5 (the sharded 16384-ego batch) on several; "
                         "4 = the FISS+ pipeline (dense tables + search walk + 3 refinement rounds)")
    ap.add_argument("--layout", default="survey8d", choices=["lanes", "survey8d"], help="obstacle layout of the synthetic scenes (synth.py): SURVEY 8d verbatim, or the builder's lanes layout")
    ap.add_argument("--rotate", type=int, default=4, help="distinct batches cycled through in the timed region")
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
    """Bytes one ego problem must READ from HBM once."""
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
    kPrevRing = 64

    def order_hint(self, on: bool):
        self.fb.launch_order = self.dten["launch_order"].data_ptr() if on else None
        self.hinted = bool(on)

    def __init__(self, torch, eng, batch, dev, stream, fiss=False, tables=False, hint=True):
        """hint: pass fp_batch.launch_order (legs that cycle DISTINCT batches; a leg that replays one batch leaves it off - the ctx then
        orders the launch by the durations the same batch left behind, as in every earlier round)."""
        from fiss_plus_planner_amd import _abi
        from fiss_plus_planner_amd.engine import device_batch, launch_order_hint, make_params

        self.torch, self.eng, self.batch, self.dev, self.stream, self.fiss, self.tables = torch, eng, batch, dev, stream, fiss, tables
        self.dten = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in BATCH_ARRAYS}
        if getattr(batch, "obs_nvert", None) is not None:  # convex-polygon obstacle columns (ABI 12)
            self.dten.update({k: torch.from_numpy(getattr(batch, k)).to(dev) for k in ("obs_poly", "obs_nvert")})
        # fp_batch.launch_order (ABI 14), as device_batch.DeviceBatch passes it: the egos by descending speed - input-only, one argsort on the
        # host at upload; results do not depend on it (BENCH_NO_ORDER_HINT=1: without, the ctx's feedback order as before)
        self.dten["launch_order"] = torch.from_numpy(launch_order_hint(batch)).to(dev)
        self.fb = device_batch(batch, {k: (v.data_ptr() if v.numel() else 0) for k, v in self.dten.items()})
        self.order_hint(hint and not os.environ.get("BENCH_NO_ORDER_HINT"))
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
        # The kernels write best index / cost straight into the pinned (device-mapped) host buffer: the results are in host memory when
        # the step's kernels are done, with no copy command on the stream (a D2H copy costs ~14 us of stream time per step: a 12 us
        # dependency bubble + a 2 us blit kernel).  BENCH_COPY_RESULTS=1: results into HBM + an explicit async copy instead.
        # (FISS+ too since round 6: best_cost / refined are only ever WRITTEN by the search and refinement kernels - until then a FISS+ step
        # carried a D2H copy command the dense step did not: 0.219 ms against 0.198 for the same kernels)
        self.zero_copy = not os.environ.get("BENCH_COPY_RESULTS")
        if self.zero_copy:
            self.best_cost, self.best_idx = self.h_cost, self.h_idx
        if fiss:
            self.f_t = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in ("samp_min", "samp_max", "samp_res")}
            # prev_best_idx is in/out (the search's history heuristic); every step must plan the same cycle, so every step gets a fresh
            # all -1 array out of a ring that is refilled by ONE fill per kPrevRing steps (a fill per step was a 4.6 us kernel + a
            # launch gap inside every timed step - an artefact of replaying one cycle, not a part of the pipeline)
            self.prev_ring = torch.full((self.kPrevRing, B, 3), -1, dtype=torch.int32, device=dev)
            self.prev_k = 0
            self.prev = self.prev_ring[0]
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
            if self.prev_k == self.kPrevRing:
                self.eng.join(self.stream.cuda_stream)  # ("overlap": the fill below must come after the call still in flight; a no-op otherwise)
                with self.torch.cuda.stream(self.stream):
                    self.prev_ring.fill_(-1)  # (all of the ring's arrays have been used: the kernels before this fill are done with them in stream order)
                self.prev_k = 0
            self.prev = self.prev_ring[self.prev_k]
            self.io.prev_best_idx = self.prev.data_ptr()
            self.prev_k += 1
            self.eng.plan_fiss_device(self.params, self.fb, self.opts, self.io, stream=self.stream.cuda_stream)
        else:
            # one launch: lattice + argmin + the winner's series (what plan() returns) written by the workgroup that found it
            self.eng.plan_dense_device(self.params, self.fb, self.best_idx.data_ptr(), self.best_cost.data_ptr(), self.stats.data_ptr(),
                                       self.cost_tbl.data_ptr() if self.tables else 0, self.flag_tbl.data_ptr() if self.tables else 0,
                                       stream=self.stream.cuda_stream, best_flags=self.best_flags.data_ptr(), best_traj=self.best_traj.data_ptr(),
                                       traj_stride=self.traj_stride, traj_sparse=True)

    def fetch(self):
        if not self.zero_copy:
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
    if ev_every is None:  # every few launches, and not in step with the rotation (the events must sample every batch of it)
        ev_every = next(e for e in (3, 4, 5, 7) if math.gcd(e, n) == 1) if steps >= 16 else 1
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
    timed_run.enqueue_s = time.perf_counter() - t0  # host time to enqueue the K steps (the GPU may still be running them)
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, [a.elapsed_time(b) for a, b in ev]


def wake(torch, dev, fn, seconds):
    """Device wake-up before a timed leg (see `prewarm` in main): the CPU-side parity checks between the legs leave the GPU idle for
    seconds, and a leg's own few warm-up steps are too short for its clocks to come back.  Runs fn() untimed for `seconds`."""
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(16):
            fn()
            n += 1
        torch.cuda.synchronize(dev)
    return n


def roofline_obj(bytes_launch, kern_ms, kernel, traffic=None, note=None, fp64=None):
    """The contract's roofline object for the HBM bound (north_star asks for HBM GB/s) and, when the PMC pass of this build is at hand,
    the bound that actually binds: executed FP64 VALU work against the vector FP64 peak.  `binding` names which of the two is closer
    to its roof; the top-level keys stay the HBM figures the contract defines."""
    ach = bytes_launch / (kern_ms * 1e-3) / 1e9
    hbm = {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
           "algorithmic_bytes_per_launch": bytes_launch}
    o = {"bound": "hbm", **hbm, "kernel": kernel, "kernel_ms": kern_ms}
    if fp64:
        o["binding"] = "valu_fp64" if fp64["frac"] > hbm["frac"] else "hbm"
        o["hbm"] = dict(hbm)
        o["valu_fp64"] = fp64
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


def materialize_leg(torch, eng, main_wl, dev, stream):
    from fiss_plus_planner_amd.engine import device_batch

    batch = main_wl.batch
    B, C = batch.B, batch.C
    Bm = min(B, 256)
    fbm = device_batch(batch, {k: (v.data_ptr() if v.numel() else 0) for k, v in main_wl.dten.items()})
    fbm.B = Bm
    m_flags = torch.empty(Bm * C, dtype=torch.int32, device=dev)
    materialize = {"kernel": "materialize_profiles_kernel (all candidates; one wavefront per ego and longitudinal profile writes its nd candidates)", "egos": Bm, "bound": "hbm", "peak_GBps": HBM_PEAK_GBS}
    # two layouts of the same payload: the compact one (rows of ceil(max T / tick) columns, only existing elements written) and
    # the round-1 layout (16 x 128 NaN-padded block per candidate).  Every launch writes into a FRESH allocation (the spread over
    # buffer placements is part of the measurement); median and maximum are both reported.
    n_launch = 30
    for label, m_stride, m_sparse in (("compact", main_wl.traj_stride, True), ("padded128", 128, False)):
        kw = dict(stream=stream.cuda_stream, traj_stride=m_stride, traj_sparse=m_sparse)
        m_all = []
        for it in range(n_launch + 2):
            m_traj = torch.empty((Bm * C, 16, m_stride), dtype=torch.float64, device=dev)
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            eng.materialize_all_device(main_wl.params, fbm, m_flags.data_ptr(), m_traj.data_ptr(), **kw)
            b_.record(stream)
            torch.cuda.synchronize(dev)
            if it >= 2:
                m_all.append(a.elapsed_time(b_))
            if it % 3 == 2:
                keep = m_traj  # noqa: F841  (holding every third buffer moves the following allocations)
            else:
                del m_traj
        m_ms = float(np.median(m_all))
        fl_m = m_flags.cpu().numpy().view(np.uint32)
        alg = series_bytes(fl_m) + 4 * Bm * C
        written = (series_bytes(fl_m, lines=True) + 4 * Bm * C) if m_sparse else Bm * C * (16 * m_stride * 8 + 4)
        materialize[label] = {"kernel_ms": m_ms, "kernel_ms_min": float(np.min(m_all)), "kernel_ms_max": float(np.max(m_all)),
                              "spread": float((np.max(m_all) - np.min(m_all)) / m_ms), "launches_timed": len(m_all),
                              "candidates_per_s": Bm * C / (m_ms * 1e-3), "bytes_written_per_launch": written,
                              "algorithmic_bytes_per_launch": alg, "achieved_GBps": written / (m_ms * 1e-3) / 1e9,
                              "algorithmic_GBps": alg / (m_ms * 1e-3) / 1e9, "frac": written / (m_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "algorithmic_over_written": alg / written}
        keep = None
    del m_flags
    return materialize


COST_TOL = 1e-6  # BASELINE.json north_star: costs within 1e-6, selected index exact


def parity_fail(leg, what):
    raise SystemExit(f"PARITY FAILURE ({leg}): {what}")


def series_parity(leg, wl, egos, max_egos=64):
    """The winners' series the step left in HBM (best_flags / best_traj of `wl`, compact layout) against the oracle's restatement of
    calc_frenet_paths + calc_global_paths (frenet_optimal_planner.py:69-138) for up to max_egos of `egos` that have a winner: the flag
    word's N / M exact, rows 0-10 (t, s .. s_ddd, d .. d_ddd, x, y) within 1e-8, rows 11-15 (yaw, ds, c, c_d, c_dd: difference chains
    of x / y) within the bound derived from a few-ulp position error.  -> {"series_checked": n, ...} merged into the leg's `parity`."""
    from oracle import oracle as O

    batch = wl.batch
    idx = wl.h_idx.numpy() if wl.zero_copy else wl.best_idx.cpu().numpy()
    egos = [int(e) for e in np.asarray(egos) if idx[e] >= 0][:max_egos]
    if not egos:
        return {"series_checked": 0}
    sel = wl.torch.as_tensor(egos, device=wl.dev)
    fl = wl.best_flags[sel].cpu().numpy().view(np.uint32)
    tr = wl.best_traj[sel].cpu().numpy()
    err_pos = err_chain = 0.0
    for k, (e, pr) in enumerate(zip(egos, O.problems_from_batch(batch, egos))):
        bi = int(idx[e])
        iv, it, i_d = bi % batch.nv, (bi // batch.nv) % batch.nt, bi // (batch.nv * batch.nt)
        t = pr.eval_traj(batch.d_samples[i_d], batch.v_samples[e, iv], batch.t_samples[it], dump=True, stride=wl.traj_stride)
        N, M = int((fl[k] >> 8) & 0xFFF), int(fl[k] >> 20)
        if (N, M) != (t.N, t.M):
            parity_fail(leg, f"series of ego {e}: N / M = {(N, M)}, oracle {(t.N, t.M)}")
        want, got = t.arrays, tr[k]
        m = ~np.isnan(want)  # (compact layout: only the elements that exist are written; the rest of the block is the caller's bytes)
        if np.isnan(got[m]).any():
            parity_fail(leg, f"series of ego {e}: an element the oracle has is NaN")
        d = np.abs(np.where(m, got - want, 0.0))
        err_pos = max(err_pos, float(d[:11].max()))
        if not d[:11].max() <= 1e-8:
            parity_fail(leg, f"series of ego {e}: rows 0-10 differ by {d[:11].max():.3e}")
        # rows 11-15: 2 * (position error) / ds through the chain, as tests/conftest.py:series_tol derives it
        ds = want[12]
        with np.errstate(divide="ignore", invalid="ignore"):
            t_yaw = np.where(ds > 0, 2 * 4e-13 / ds, np.inf)
            if M >= 2:
                t_yaw[M - 1] = t_yaw[M - 2]
            nxt = lambda a: np.append(a[1:], np.inf)  # noqa: E731
            t_c = (t_yaw + nxt(t_yaw)) / ds + np.abs(want[13]) * 2 * 4e-13 / ds
            t_cd = (t_c + nxt(t_c)) / batch.tick_t
            t_cdd = (t_cd + nxt(t_cd)) / batch.tick_t
        for row, tl in ((11, t_yaw), (12, np.full(ds.shape, 8e-13)), (13, t_c), (14, t_cd), (15, t_cdd)):
            tol = np.maximum(1e-9, np.nan_to_num(tl, nan=np.inf, posinf=np.inf)) * 4 + 1e-9
            if (m[row] & ~(d[row] <= tol)).any():
                parity_fail(leg, f"series of ego {e}: row {row} differs by {d[row][m[row]].max():.3e}")
            fin = m[row] & np.isfinite(tol)
            err_chain = max(err_chain, float(d[row][fin].max()) if fin.any() else 0.0)
    return {"series_checked": len(egos), "series_NM_exact": True, "series_rows_0_10_max_abs_err": err_pos, "series_rows_0_10_tolerance": 1e-8,
            "series_rows_11_15_max_abs_err": err_chain, "series_oracle": "oracle/libfrenet_oracle.so orc_eval_traj (dump)"}


def fop_parity(leg, batch, egos, g_idx, g_cost, threads, wl=None):
    """FOP outputs of `egos` against the oracle: selected index exact, cost <= 1e-6; with `wl` (the workload whose last step wrote the
    outputs) also the winners' flag words and series (series_parity).  -> the leg's `parity` object."""
    from oracle import oracle as O

    egos = np.asarray(egos)
    o_idx, o_cost = O.fop_plan_batch(O.problems_from_batch(batch, egos), threads=threads)
    if not np.array_equal(g_idx[egos], o_idx):
        parity_fail(leg, f"selected index differs on egos {egos[np.nonzero(g_idx[egos] != o_idx)[0][:8]].tolist()}")
    ok = o_idx >= 0
    err = float(np.abs(g_cost[egos][ok] - o_cost[ok]).max()) if ok.any() else 0.0
    if not err <= COST_TOL:
        parity_fail(leg, f"best cost differs by {err:.3e}")
    par = {"checked_egos": int(len(egos)), "index_exact": True, "max_abs_cost_err": err, "cost_tolerance": COST_TOL,
           "egos_with_a_winner": int(ok.sum()), "oracle": "oracle/libfrenet_oracle.so orc_fop_plan"}
    if wl is not None:
        par.update(series_parity(leg, wl, egos))
    return par


def tables_parity(leg, batch, egos, g_cost_tbl, g_flag_tbl, g_stats):
    """The per-candidate outputs SURVEY 8(d)'s metric counts - cost_tbl / flag_tbl rows and Stats - of `egos` against the oracle:
    every flag word (feasibility bits, N, M) exact, every cost within the tolerance."""
    from oracle import oracle as O

    egos = np.asarray(egos)
    ref = [pr.fop_plan() for pr in O.problems_from_batch(batch, egos)]
    r_flags, r_cost = np.stack([r.flags for r in ref]), np.stack([r.cost for r in ref])
    r_stats = np.stack([r.stats for r in ref])
    if not np.array_equal(g_flag_tbl[egos], r_flags):
        bad = np.argwhere(g_flag_tbl[egos] != r_flags)[:4]
        parity_fail(leg, f"flag words differ at (ego, candidate) {[(int(egos[a]), int(c)) for a, c in bad]}")
    both = np.isfinite(r_cost)
    err = float(np.abs(g_cost_tbl[egos][both] - r_cost[both]).max())
    if not err <= COST_TOL or not np.array_equal(np.isnan(g_cost_tbl[egos]), np.isnan(r_cost)):
        parity_fail(leg, f"cost table differs by {err:.3e}")
    if not np.array_equal(g_stats[egos], r_stats):
        parity_fail(leg, "Stats differ")
    return {"checked_egos": int(len(egos)), "checked_candidates": int(r_flags.size), "flag_words_exact": True, "stats_exact": True,
            "max_abs_cost_err": err, "cost_tolerance": COST_TOL, "oracle": "oracle/libfrenet_oracle.so orc_fop_plan (per-candidate tables)"}


def fiss_parity(leg, wl, egos):
    """FISS+ pipeline outputs (coarse index, refined yes/no, end state, cost, all four Stats) of `egos` against the oracle."""
    from oracle import oracle as O

    egos = np.asarray(egos)
    ijk, refined = wl.ijk.cpu().numpy(), wl.best_idx.cpu().numpy()
    cost, stats, end = wl.best_cost.cpu().numpy(), wl.stats.cpu().numpy(), wl.end_state.cpu().numpy()
    err = 0.0
    n_found = n_refined = 0
    for e, pr in zip(egos, O.problems_from_batch(wl.batch, egos)):
        r = pr.fissplus_plan(None)
        found = not np.isnan(r.best_cost)
        if not np.array_equal(stats[e], r.stats):
            parity_fail(leg, f"Stats differ on ego {e}: {stats[e].tolist()} vs {r.stats.tolist()}")
        if (not np.isnan(cost[e])) != found:
            parity_fail(leg, f"found / not found differs on ego {e}")
        if not found:
            continue
        n_found += 1
        if bool(refined[e]) != r.refined:
            parity_fail(leg, f"refined flag differs on ego {e}")
        n_refined += int(r.refined)
        if not r.refined and not np.array_equal(ijk[e], r.best_ijk):
            parity_fail(leg, f"coarse index differs on ego {e}")
        err = max(err, abs(cost[e] - r.best_cost), float(np.abs(end[e] - r.end_state).max()))
    if not err <= COST_TOL:
        parity_fail(leg, f"cost / end state differ by {err:.3e}")
    return {"checked_egos": int(len(egos)), "index_exact": True, "stats_exact": True, "refined_exact": True, "max_abs_cost_err": err,
            "cost_tolerance": COST_TOL, "egos_with_a_winner": n_found, "of_them_refined": n_refined,
            "oracle": "oracle/libfrenet_oracle.so orc_fissplus_plan"}


def host_info():
    """What the CPU baseline ran on: the affinity mask says how many CPUs the process MAY use, the cgroup quota how much CPU
    time it actually gets."""
    info = {"affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "os_cpu_count": os.cpu_count()}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            info["cgroup_" + os.path.basename(path)] = open(path).read().strip()
        except OSError:
            pass
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        info["cpu_model"] = model[0] if model else None
    except OSError:
        pass
    return info


def cpu_rate(batch, egos, threads, fast=False):
    from oracle import oracle as O

    probs = O.problems_from_batch(batch, egos)
    t0 = time.perf_counter()
    o_idx, o_cost = O.fop_plan_batch(probs, threads=threads, fast=fast)
    dt = time.perf_counter() - t0
    return len(probs) * batch.C / dt, dt, o_idx, o_cost


def cpu_baseline_leg(batch, h_idx, h_cost, seconds, max_threads):
    """The oracle (plain-C restatement, OpenMP over egos, per-thread scratch) on a bounded sample of the same egos.
    A short sweep over thread counts first: the reported all-core figure uses the count that was FASTEST, and the sweep is in
    the line (an affinity mask of 256 CPUs does not mean 256 CPUs' worth of quota).  Its indices / costs gate the GPU's."""
    from oracle import oracle as O

    O.build()
    B, C = batch.B, batch.C
    cpu_rate(batch, range(min(8, B)), 1)  # page the library in
    rate1, dt1, _, _ = cpu_rate(batch, range(min(8, B)), 1)
    sweep = {1: rate1}
    t = 2
    cand = []
    while t < max_threads:
        cand.append(t)
        t *= 2
    cand.append(max_threads)
    for t in cand:
        n = min(B, max(2 * t, 16))
        sweep[t] = cpu_rate(batch, range(n), t)[0]
    best_t = max(sweep, key=lambda k: sweep[k])
    out = {}
    for label, threads, secs in (("cpu_baseline", best_t, seconds), ("cpu_baseline_1thread", 1, min(6.0, seconds))):
        n_s = int(max(min(16, B), min(B, secs * sweep[threads] / C)))
        rate, dt, o_idx, o_cost = cpu_rate(batch, range(n_s), threads)
        if not np.array_equal(h_idx[:n_s], o_idx):
            parity_fail(label, f"selected index differs on egos {np.nonzero(h_idx[:n_s] != o_idx)[0][:8].tolist()}")
        ok = o_idx >= 0
        err = float(np.abs(h_cost[:n_s][ok] - o_cost[ok]).max()) if ok.any() else 0.0
        if not err <= COST_TOL:
            parity_fail(label, f"best cost differs by {err:.3e}")
        out[label] = {"value": rate, "unit": "candidates/s", "cores": threads, "kind": "port",
                      "sample": f"first {n_s} egos of the first timed batch ({n_s * C} candidates), oracle/libfrenet_oracle.so (literal "
                                f"restatement: pow() per term, point-by-point sums, polygon SAT), {threads} OpenMP thread(s) over egos, {dt:.1f} s; "
                                f"GPU index / cost parity checked on this sample",
                      "parity": {"checked_egos": n_s, "index_exact": True, "max_abs_cost_err": err, "cost_tolerance": COST_TOL}}
    # a second, LABELLED baseline: the same restatement built -O3 with `t ** k` as multiplications (oracle/Makefile `fast`)
    n_f = int(max(min(16, B), min(B, min(6.0, seconds) * 1.5 * sweep[best_t] / C)))
    cpu_rate(batch, range(min(8, B)), 1, fast=True)
    rate, dt, o_idx, o_cost = cpu_rate(batch, range(n_f), best_t, fast=True)
    ok = o_idx >= 0
    if not np.array_equal(h_idx[:n_f], o_idx) or (ok.any() and not np.abs(h_cost[:n_f][ok] - o_cost[ok]).max() <= COST_TOL):
        parity_fail("cpu_baseline_optimised", "index / cost differ")
    out["cpu_baseline_optimised"] = {"value": rate, "unit": "candidates/s", "cores": best_t, "kind": "port",
                                     "sample": f"first {n_f} egos of the first timed batch, oracle/libfrenet_oracle_fast.so (the same per-candidate restatement, "
                                               f"-O3, integer powers as multiplications instead of pow()), {best_t} OpenMP threads, {dt:.1f} s; GPU parity checked"}
    out["cpu_baseline"]["thread_sweep"] = {str(k): v for k, v in sorted(sweep.items())}
    out["cpu_baseline"]["host"] = host_info()
    return out


def closed_loop_leg(torch, eng, dev, B, layout, cycles, threads, eng2=None, stream2=None, wake_s=0.0):
    """north_star's workload is "many scenarios x many cycles": B egos stepped `cycles` plan cycles entirely on the device
    ([plan -> advance] per cycle, planners/benchmark/planning.py:120-162; no host round trip).  Parity: the state the loop left
    behind is planned once more and compared with the oracle planning the same states."""
    from fiss_plus_planner_amd import _abi, synth
    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch
    from oracle import oracle as O

    out = {"egos": B, "cycles": cycles, "unit": "ego-plans/s",
           "workload": f"{B} egos x {cycles} device-resident plan cycles (fp_plan_* + fp_advance per cycle, results stay in HBM), configs[2] scenes"}
    for planner, cfg in (("FOP", 3), ("FISS+", 4)):
        batch = synth.make_config(cfg, B=B, layout=layout)
        goal = np.full((B, 2), 1e9)  # never reached: an ego runs until the map ends or no candidate survives
        run = ClosedLoopRunner(eng, DeviceBatch(batch, dev.index), goal, planner)
        ego0 = torch.from_numpy(batch.ego).to(dev)

        def restart():  # back to the initial states
            run.db.t["ego"].copy_(ego0); run.db.t["t_now"].zero_(); run.done.zero_(); run.cycles.zero_()
            if planner != "FOP":
                run.prev.fill_(-1)

        # device wake-up (see `prewarm`): the whole loop, untimed, until the clocks are up; then the timed loop from the initial states
        t_w = time.perf_counter()
        while True:
            run.run(cycles - 1 if wake_s > 0 else 2)
            restart()
            if time.perf_counter() - t_w >= wake_s:
                break
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        res = run.run(cycles - 1)
        dt = time.perf_counter() - t0
        plans = int(res.cycles.sum() + (res.done == _abi.DONE_NO_SOLUTION).sum())  # an ego's last plan may have found nothing
        # parity of the state the loop left behind: one more plan (no advance) of the running egos vs the oracle on those states
        running = np.nonzero(res.done == 0)[0]
        egos = running[:: max(1, len(running) // 48)][:48]
        snap = synth.make_config(cfg, B=B, layout=layout)
        snap.ego[:] = res.ego
        snap.t_now[:] = res.t_now
        stream = torch.cuda.current_stream(dev).cuda_stream
        if planner == "FOP":
            eng.plan_dense_device(run.db.params, run.fb, run.best_idx.data_ptr(), run.best_cost.data_ptr(), run.stats.data_ptr(), stream=stream)
            torch.cuda.synchronize(dev)
            par = fop_parity(f"closed_loop {planner}", snap, egos, run.best_idx.cpu().numpy(), run.best_cost.cpu().numpy(), threads) if len(egos) else None
        else:
            prev = run.prev.cpu().numpy().copy()
            eng.plan_fiss_device(run.db.params, run.fb, run.fopts, run.fio, stream=stream)
            torch.cuda.synchronize(dev)
            cost, stats = run.best_cost.cpu().numpy(), run.stats.cpu().numpy()
            err = 0.0
            for e, pr in zip(egos, O.problems_from_batch(snap, egos)):
                r = pr.fissplus_plan(None if prev[e, 0] < 0 else prev[e])
                if not np.array_equal(stats[e], r.stats):
                    parity_fail("closed_loop FISS+", f"Stats differ on ego {e} after {cycles - 1} cycles")
                if np.isnan(r.best_cost) != np.isnan(cost[e]):
                    parity_fail("closed_loop FISS+", f"found / not found differs on ego {e}")
                if not np.isnan(r.best_cost):
                    err = max(err, abs(cost[e] - r.best_cost))
            if not err <= COST_TOL:
                parity_fail("closed_loop FISS+", f"cost differs by {err:.3e}")
            par = {"checked_egos": int(len(egos)), "stats_exact": True, "max_abs_cost_err": err, "cost_tolerance": COST_TOL,
                   "oracle": "orc_fissplus_plan on the device loop's states and prev_best_idx"} if len(egos) else None
        out[planner] = {"value": plans / dt, "ego_plans": plans, "seconds": dt, "us_per_cycle": dt / (cycles - 1) * 1e6,
                        "egos_still_running": int(len(running)), "candidates_per_s": plans * batch.C / dt, "parity": par}
        del run
        # the same fleet as two half fleets, each with its own fp_ctx and HIP stream, their cycles enqueued alternately (what
        # ShardedEngine(shards_per_device=2).closed_loop does with a host thread per shard): a cycle of a few hundred running egos
        # is one round of workgroups - as long as its slowest ego - so two of them overlap.  Parity: the final states must equal
        # the single-stream loop's, bit for bit.
        if eng2 is not None:
            halves = []
            for r, (e_, st_) in enumerate(((eng, torch.cuda.current_stream(dev)), (eng2, stream2))):
                sb = batch.shard(r, 2)
                lo, hi = (B * r) // 2, (B * (r + 1)) // 2
                with torch.cuda.stream(st_):
                    rn = ClosedLoopRunner(e_, DeviceBatch(sb, dev.index), goal[lo:hi], planner)
                halves.append((rn, st_, sb))
            torch.cuda.synchronize(dev)
            def restart_halves():
                for rn, st_, sb in halves:
                    with torch.cuda.stream(st_):
                        rn.db.t["ego"].copy_(torch.from_numpy(sb.ego)); rn.db.t["t_now"].zero_(); rn.done.zero_(); rn.cycles.zero_()
                        if planner != "FOP":
                            rn.prev.fill_(-1)

            t_w = time.perf_counter()
            while True:  # warm-up (and device wake-up: the parity check above left the GPU idle), then back to the initial states
                for _ in range(24):
                    for rn, st_, sb in halves:
                        rn.step(st_.cuda_stream)
                torch.cuda.synchronize(dev)
                restart_halves()
                if time.perf_counter() - t_w >= wake_s:
                    break
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(cycles - 1):
                for rn, st_, sb in halves:
                    rn.step(st_.cuda_stream)
            torch.cuda.synchronize(dev)
            dt2 = time.perf_counter() - t0
            ego2 = np.concatenate([rn.db.t["ego"].cpu().numpy() for rn, _, _ in halves])
            done2 = np.concatenate([rn.done.cpu().numpy() for rn, _, _ in halves])
            cyc2 = np.concatenate([rn.cycles.cpu().numpy() for rn, _, _ in halves])
            if not (np.array_equal(ego2, res.ego) and np.array_equal(done2, res.done) and np.array_equal(cyc2, res.cycles)):
                parity_fail(f"closed_loop {planner} two_streams", "final states differ from the single-stream loop")
            out[planner]["two_streams"] = {"value": plans / dt2, "seconds": dt2, "us_per_cycle": dt2 / (cycles - 1) * 1e6,
                                           "parity": "final ego states / done / cycles equal the single-stream loop bit for bit"}
            del halves
    return out


def process_env():
    """Environment of a bench PROCESS (set in main(), not at import: the tests import this module for the line builder)."""
    # The CPU-baseline leg runs the oracle with one OpenMP thread per host core; by default the runtime leaves those threads spinning
    # for a while after the parallel region, next to the thread that launches the workloads measured after it.
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    # The two-stream legs need their streams on different hardware queues: with the runtime's default of four, two created streams can
    # share one, and their launches then serialise (a two-half-fleet closed loop measured 270 instead of 147 us per cycle).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    process_env()
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

    # ---- this rank's shards, generated directly (every ego has its own RNG stream): batch k of the rotation holds egos
    # [(k * world + rank) * B, +B) of the generator's stream - at k = 0 that is rank r's slice of the 16384-ego config-5 batch
    fiss = config == 4
    n_rot = max(1, args.rotate)
    dev = torch.device("cuda", local_rank)
    eng = FrenetEngine(local_rank)
    for kv in filter(None, os.environ.get("BENCH_CTX_OPTIONS", "").split(",")):  # diagnostic: "name=value,..." -> fp_ctx_set_option
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    stream = torch.cuda.current_stream(dev)
    wls = [Workload(torch, eng, synth.make_config(config, B=args.egos, ego_offset=(k * world + rank) * args.egos, layout=args.layout), dev, stream,
                    fiss=fiss, tables=args.tables) for k in range(n_rot)]
    main_wl = wls[0]
    batch = main_wl.batch
    B, C = batch.B, batch.C

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()

    # ---- device wake-up, outside the contract's W + K steps and reported in the line (`prewarm`): a process that has just started
    # finds the GPU at idle clocks, and W = 5 steps are 0.8 ms of work - the K steps would be timed on the clock ramp (measured on
    # MI355X with `--steps 20`: 0.160-0.164 ms per step with --warmup 5 against 0.150-0.155 with --warmup 300, same box, same
    # process otherwise).  A quarter of a second of the same steps, untimed, brings the clocks up; then the W warm-up steps, then
    # exactly K timed ones.  BENCH_PREWARM_S=0 turns it off.
    prewarm_s = float(os.environ.get("BENCH_PREWARM_S", "0.25"))
    cold = None
    if prewarm_s > 0:  # the same W + K steps WITHOUT the wake-up first, reported beside the headline (`cold_start`)
        el_c, _ = timed_run(torch, wls, args.steps, args.warmup, stream, barrier)
        cold = {"ms_per_step": el_c / args.steps * 1e3, "value": float(np.mean([w.candidates for w in wls])) * args.steps / el_c * world,
                "what": "the W warm-up + K timed steps of this invocation run first, in the fresh process, before any wake-up: the GPU is on its clock ramp (rank 0's time)"}
    prewarm_steps, t_pre = 0, time.perf_counter()
    while time.perf_counter() - t_pre < prewarm_s:
        for _ in range(32):
            wls[prewarm_steps % n_rot].step(); wls[prewarm_steps % n_rot].fetch()
            prewarm_steps += 1
        torch.cuda.synchronize(dev)
    # ---- timed region (the contract: W warm-up steps, exactly K timed steps between barrier + synchronize)
    elapsed, kern_list = timed_run(torch, wls, args.steps, args.warmup, stream, barrier)
    enqueue_ms = timed_run.enqueue_s / args.steps * 1e3
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean(kern_list))
    solo = rank == 0 and world == 1
    threads_all = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    gate_threads = min(threads_all, 64)

    # ---- parity gates + CPU baseline (rank 0, N=1 only), AFTER the timed region (libgomp workers spin after a parallel region
    # and would steal the launch thread's core); any mismatch aborts before anything is printed.
    for w in wls:  # every batch of the rotation leaves the results of its last step behind
        w.step(); w.fetch()
    torch.cuda.synchronize(dev)
    cpu = {}
    parity_main = None
    if solo and args.cpu_seconds > 0:
        if fiss:
            parity_main = {"batches": [fiss_parity(f"main batch {k}", w, np.arange(0, B, max(1, B // (64 if k == 0 else 16)))) for k, w in enumerate(wls)]}
        else:
            cpu = cpu_baseline_leg(batch, main_wl.h_idx.numpy(), main_wl.h_cost.numpy(), args.cpu_seconds, threads_all)
            parity_main = {"batches": [dict(cpu["cpu_baseline"]["parity"], note="the cpu_baseline sample")] +
                                      [fop_parity(f"main batch {k}", w.batch, np.arange(0, B, max(1, B // 64)), w.h_idx.numpy(), w.h_cost.numpy(), gate_threads, wl=w)
                                       for k, w in enumerate(wls) if k > 0]}
            parity_main["batches"][0].update(series_parity("main batch 0", main_wl, np.arange(0, B, max(1, B // 160)), 96))

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
            # parity: the per-cycle costs of the reference's own closed loop on this scenario (fixture G5, generated by running the reference)
            rows = g[f"{kind}_rows"]
            costs = np.array([c.cost for c in res.cycles])
            n = min(len(rows), len(costs))
            err = float(np.abs(costs[:n] - rows[:n, 6]).max()) if n else 0.0
            if len(costs) != len(rows) or not err <= COST_TOL:
                parity_fail(f"plan_cycle_latency {kind}", f"{len(costs)} cycles vs {len(rows)}, cost err {err:.3e}")
            par = {"checked_cycles": n, "max_abs_cost_err": err, "cost_tolerance": COST_TOL,
                   "reference": "tests/golden/g5_closed_loop.npz: the imported reference's own closed loop on this scenario"}
            plan_cycle[kind] = {"p50": float(np.median(ms)), "p90": float(np.percentile(ms, 90)), "cycles": len(ms), "parity": par}

    # ---- materialise mode (rank 0, N=1): the one HBM-bound mode of the path - every candidate's full series written out
    # (fp_materialize_all = the reference's all_trajs payload).  256 egos of the same batch: 2.4 GB per launch.
    materialize = None
    if solo and not args.no_latency and not fiss:
        materialize = materialize_leg(torch, eng, main_wl, dev, stream)

    # ---- the other single-GPU configurations and variants of the headline workload (rank 0, N=1), each timed like the main run
    extras = {}
    if solo and not args.no_extras and not fiss and config == 3:
        steps_x, warm_x = min(args.steps, 50), min(args.warmup, 5)

        def measure(ws, label_kernel, ev_every=None):
            if prewarm_s > 0:
                wake(torch, dev, lambda: (ws[0].step(), ws[0].fetch()), prewarm_s)
            el, kl = timed_run(torch, ws, steps_x, warm_x, stream, barrier, ev_every)
            k_ms = float(np.mean(kl))
            cand = sum(w.candidates for w in ws) / len(ws)
            bytes_l = float(np.mean([w.algorithmic_bytes() for w in ws]))
            return {"value": cand * steps_x / el, "unit": "candidates/s", "ms_per_step": el / steps_x * 1e3, "steps": steps_x, "warmup": warm_x,
                    "roofline": roofline_obj(bytes_l, k_ms, label_kernel)}

        def gate(leg, w, n):
            w.step(); w.fetch(); torch.cuda.synchronize(dev)
            return fop_parity(leg, w.batch, np.arange(0, w.batch.B, max(1, w.batch.B // n)), w.h_idx.numpy(), w.h_cost.numpy(), gate_threads, wl=w)

        if args.cpu_seconds <= 0:
            gate = lambda leg, w, n: None  # noqa: E731  (--cpu-seconds 0: profiling runs, no oracle in the process)
        # (a) one batch replayed (the best case for the feedback-directed launch order) and the same in index order
        ordered, launches = eng.get_option("lattice_ordered_launches"), eng.get_option("lattice_launches")
        hinted = main_wl.hinted
        for w in wls:
            w.order_hint(False)  # (the three legs below: the ctx's own orders)
        extras["single_batch_replayed"] = dict(measure([main_wl], "lattice_fused_kernel, one batch replayed, no hint: launch order learnt on the batch itself"),
                                               parity=gate("single_batch_replayed", main_wl, 64))
        extras["launch_order_hint_off"] = dict(measure(wls, "lattice_fused_kernel, no hint: every batch of the rotation dispatched in the order learnt from ITS OWN earlier launches (one order per resident batch)"),
                                               parity=gate("launch_order_hint_off", wls[-1], 64))
        eng.set_option("lattice_order", 0)
        extras["lattice_order_off"] = dict(measure(wls, "lattice_fused_kernel, workgroups dispatched in ego index order"),
                                           parity=gate("lattice_order_off", wls[-1], 64))
        for w in wls:
            w.order_hint(True)
        extras["launch_order_hint_only"] = dict(measure(wls, "lattice_fused_kernel, fp_batch.launch_order alone (egos by descending speed, input-only; no learnt order): round 5's headline path"),
                                                parity=gate("launch_order_hint_only", wls[-1], 64))
        eng.set_option("lattice_order", 1)
        for w in wls:
            w.order_hint(hinted)
        extras["launch_order"] = {"main_run": "the order the ctx learnt on each resident batch (durations its own earlier launches left behind; fetched asynchronously every 8 launches, "
                                              "sorted on the host INSIDE the timed region); until a batch has one: " +
                                              ("fp_batch.launch_order = the egos by descending speed (engine.launch_order_hint: input-only, one host argsort at upload)" if hinted else "index order"),
                                  "launches_of_the_main_run": launches, "of_them_in_a_given_or_learnt_order": ordered,
                                  "note": f"the main run cycles {n_rot} distinct batches; since round 6 the ctx keeps one learnt order PER resident batch (keyed by the batch's ego array), "
                                          "so an order is never applied to another batch than the one it was learnt on. launch_order_hint_only = the input-only order alone; "
                                          "lattice_order_off = index order"}
        # (a2) what SURVEY 8(d)'s metric definition counts in full: the per-candidate cost / flag tables written (12 B per candidate) and
        # the Stats brought to the host every step, next to index / cost / series as in the headline
        wt = [Workload(torch, eng, w.batch, dev, stream, tables=True) for w in wls]
        h_stats = [torch.empty((B, 4), dtype=torch.int32).pin_memory() for _ in wt]
        if os.environ.get("BENCH_COPY_RESULTS"):
            for w, hs in zip(wt, h_stats):
                w.fetch = (lambda w=w, hs=hs: hs.copy_(w.stats, non_blocking=True))  # stats D2H on the launch stream, every step
        else:  # like index / cost: the kernels write the Stats straight into pinned (device-mapped) host memory - no copy command (~14 us of stream time)
            for w, hs in zip(wt, h_stats):
                w.stats = hs
        ot = measure(wt, "lattice_fused_kernel (tables written), one launch")
        ot["what"] = "the headline workload with cost_tbl + flag_tbl written to HBM (tables_written: true) and the Stats of every ego landing in pinned host memory every step"
        ot["tables_written"] = True
        if args.cpu_seconds > 0:
            wt[0].step(); torch.cuda.synchronize(dev)
            ot["parity"] = tables_parity("tables_written", wt[0].batch, np.arange(0, B, max(1, B // 48)), wt[0].cost_tbl.cpu().numpy(),
                                         wt[0].flag_tbl.cpu().numpy().view(np.uint32), wt[0].stats.cpu().numpy())
        extras["tables_written"] = ot
        del wt, h_stats
        # (a3) the resident sharded entry point (ShardedEngine.upload + plan_dense on resident shards): the same four batches, one shard
        # on this GPU, enqueue-only calls; must sit within a few percent of the headline (it is the same launches behind one more
        # Python layer)
        from fiss_plus_planner_amd.sharded import ShardedEngine

        def sharded_leg(shards):
            with ShardedEngine(devices=[local_rank], shards_per_device=shards) as seng:
                sdbs = [seng.upload(w.batch, winner=True) for w in wls]
                if prewarm_s > 0:  # device wake-up (the engine has its own ctx: its launch order is learnt here too, like the headline's in `prewarm`)
                    t_w, kw = time.perf_counter(), 0
                    while time.perf_counter() - t_w < prewarm_s:
                        for _ in range(32):
                            seng.plan_dense(sdbs[kw % len(sdbs)], winner=True, sync=False)
                            kw += 1
                        for sd in sdbs:
                            sd.synchronize()
                for k in range(warm_x):
                    seng.plan_dense(sdbs[k % len(sdbs)], winner=True, sync=False)
                for sd in sdbs:
                    sd.synchronize()
                barrier()
                t0 = time.perf_counter()
                for k in range(steps_x):
                    seng.plan_dense(sdbs[(warm_x + k) % len(sdbs)], winner=True, sync=False)
                for sd in sdbs:
                    sd.synchronize()
                el = time.perf_counter() - t0
                osr = {"value": float(np.mean([w.candidates for w in wls])) * steps_x / el, "unit": "candidates/s", "ms_per_step": el / steps_x * 1e3,
                       "steps": steps_x, "warmup": warm_x, "shards": seng.world,
                       "what": f"ShardedEngine(devices=[0], shards_per_device={shards}).upload(batch) once per batch, then plan_dense(resident shards, winner=True, "
                               "sync=False) per step: one fp_group_submit per step posts every shard's call to the library's per-ctx worker threads"}
                osr["vs_headline"] = osr["ms_per_step"] / (elapsed / args.steps * 1e3)
                if args.cpu_seconds > 0:
                    seng.plan_dense(sdbs[-1], winner=True)
                    osr["parity"] = fop_parity("sharded_resident", wls[-1].batch, np.arange(0, B, max(1, B // 64)), sdbs[-1].host.best_idx, sdbs[-1].host.best_cost, gate_threads)
                del sdbs
            return osr

        extras["sharded_resident"] = sharded_leg(1)
        # two logical shards on the one device (each step = two 1024-ego launches on two contexts / streams, posted by one FFI call): what
        # `two_streams` does by hand, through the product's multi-GPU path
        extras["sharded_resident_two_shards"] = sharded_leg(2)
        # (a4) BASELINE configs[4] on ONE GPU: all 16 384 egos in one launch (21 rounds of workgroups: the tail of the last round
        # amortised over eight times the work of the headline launch)
        if B == 2048:
            b5 = synth.make_config(5, layout=args.layout)
            w5 = Workload(torch, eng, b5, dev, stream, hint=False)
            steps_keep, steps_x = steps_x, min(steps_x, 12)
            o5 = measure([w5], "lattice_fused_kernel + winner_traj_kernel, 16384 egos in one launch", ev_every=1)
            steps_x = steps_keep
            o5["workload"] = f"BASELINE.json configs[4] on one GPU: {b5.B} egos x 9x9x7, 50 dynamic obstacles (the 8-GPU configuration's whole batch)"
            o5["parity"] = gate("config5_single_gpu", w5, 96)
            extras["config5_single_gpu"] = o5
            del w5, b5
        # (b) BASELINE configs[1]: 256 egos x 5x5x5 x 10 static obstacles; parity on ALL egos
        b2 = synth.make_config(2, layout=args.layout)
        w2 = Workload(torch, eng, b2, dev, stream)
        extras["config2"] = dict(measure([w2], "lattice_fused_kernel (latency mode: slices of an ego spread over workgroups)"),
                                 workload=f"BASELINE.json configs[1]: {b2.B} egos x 5x5x5 lattice ({b2.C} cand/ego), {b2.n_obs} static obstacles, T_obs={b2.T_obs}",
                                 parity=gate("config2", w2, b2.B))
        del w2
        # (c) BASELINE configs[3]: the FISS+ pipeline on 2048 egos; per-stage times from runs that stop after stage 1 / 2
        b4 = synth.make_config(4, B=B, layout=args.layout)
        w4 = Workload(torch, eng, b4, dev, stream, fiss=True, hint=False)
        o4 = measure([w4], "lattice_fused (+ the FISS+ search in workgroups appended to its grid) + fiss_refine: the whole FISS+ pipeline in 2 launches")
        # stage times: the pipeline with the search in its OWN launch (fiss_fused = 0), stopped after stage 1 / 2 / 3; the leg's number
        # above is the default pipeline, whose search runs in workgroups appended to the lattice launch
        stage_ms = {}
        eng.set_option("fiss_fused", 0)
        for st_n in (1, 2, 3):
            eng.set_option("fiss_stages", st_n)
            _, kl = timed_run(torch, [w4], 20, 3, stream, barrier, ev_every=1)
            stage_ms[st_n] = float(np.mean(kl))
        eng.set_option("fiss_fused", 1)
        w4.step(); torch.cuda.synchronize(dev)  # leave complete outputs behind
        k4 = o4["roofline"]["kernel_ms"]
        o4["stage_ms"] = {"lattice_fused_kernel (dense tables)": stage_ms[1], "fissplus_search_kernel (own launch)": stage_ms[2] - stage_ms[1],
                          "fiss_refine_kernel (3 rounds + validation + winner series)": stage_ms[3] - stage_ms[2],
                          "three launches": stage_ms[3], "search appended to the lattice launch (default): two launches": k4}
        o4["workload"] = f"BASELINE.json configs[3]: FISS+ (search walk + 3 refinement rounds) over {b4.B} egos x 9x9x7, 50 dynamic obstacles; counts C + 21 trajectories per ego"
        if args.cpu_seconds > 0:
            o4["parity"] = fiss_parity("config4", w4, np.arange(0, B, max(1, B // 128)))
        extras["config4"] = o4
        del w4
        # (d) the builder's lanes obstacle layout (round 1 / 2 headline): ~15 % of the obstacles in the ego lane, the rest beside it
        other = "lanes" if args.layout == "survey8d" else "survey8d"
        b8 = synth.make_config(3, B=B, layout=other)
        w8 = Workload(torch, eng, b8, dev, stream, hint=False)
        o8 = measure([w8], "lattice_fused_kernel")
        o8["workload"] = f"configs[2] sizes with the '{other}' obstacle layout (synth.py), one batch replayed"
        o8["parity"] = gate(f"{other}_layout", w8, 64)
        o8["egos_with_a_feasible_candidate"] = float((w8.h_idx.numpy() >= 0).mean())
        extras[f"{other}_layout"] = o8
        del w8
        # (d1) obstacle shapes that are not rectangles (ABI 12): the headline's first batch with half of its obstacle columns turned
        # into random convex polygons (3-12 vertices) - the run-time-shape instances with the polygon narrow phase
        bp = synth.with_random_shapes(batch, 4242, frac=0.5)
        wp = Workload(torch, eng, bp, dev, stream, hint=False)
        op = measure([wp], "lattice_fused_kernel<run-time shape, POLY> (polygon narrow phase), one batch replayed")
        op["workload"] = (f"configs[2] sizes, {int((bp.obs_nvert > 0).sum())} of {bp.obs_nvert.size} obstacle columns convex polygons "
                          "(fp_batch.obs_poly / obs_nvert), the rest rectangles; one batch replayed")
        op["parity"] = gate("polygon_scenes", wp, 64)
        op["egos_with_a_feasible_candidate"] = float((wp.h_idx.numpy() >= 0).mean())
        extras["polygon_scenes"] = op
        del wp, bp
        # (d1b) rectangles handed over as 4-vertex rings (half of the columns): ProblemBatch recognises a ring that IS its column's rectangle
        # and keeps the column a rectangle - the scene takes the rectangle-only instances again.  Against the same batch as plain rectangles.
        wr0 = Workload(torch, eng, batch, dev, stream, hint=False)
        wr1 = Workload(torch, eng, synth.with_rectangle_rings(batch, 99, frac=0.5), dev, stream, hint=False)
        orr0, orr = measure([wr0], "lattice_fused_kernel, one batch replayed"), measure([wr1], "lattice_fused_kernel, one batch replayed (rectangle rings -> rectangle columns)")
        orr["workload"] = "the headline's first batch with half of its obstacle columns handed over as 4-vertex rings that are their rectangles (fp_batch.obs_poly)"
        orr["rectangles_ms_per_step"], orr["vs_rectangles"] = orr0["ms_per_step"], orr["ms_per_step"] / orr0["ms_per_step"]
        orr["parity"] = gate("rectangles_as_rings", wr1, 64)
        extras["rectangles_as_rings"] = orr
        del wr0, wr1
        # (d1c) reference lines of 200 knots (the same roads sampled every 2 m instead of 5): a workgroup's spline tables no longer fit the
        # 40 KB layout of the four-per-CU instance, the launch takes three per CU
        bl = synth.make_batch(B, 9, 9, 7, 50, 50, True, synth.CONFIG_SEEDS[3], "FOP", layout=args.layout, n_knots=200)
        wl200 = Workload(torch, eng, bl, dev, stream, hint=False)
        ol = measure([wl200], "lattice_fused_kernel, one batch replayed, 200-knot reference lines")
        ol["workload"] = "configs[2] sizes on 200-knot reference lines (the same roads, a knot every 2 m), one batch replayed"
        ol["vs_81_knots"] = ol["ms_per_step"] / orr0["ms_per_step"]
        ol["parity"] = gate("long_reference_lines", wl200, 48)
        extras["long_reference_lines"] = ol
        del wl200, bl
        # (d2) two contexts, two streams: the steps alternate between two engines (each its own fp_ctx and stream, what
        # ShardedEngine(shards_per_device=2) does on the product side), so the draining tail of one launch - and the one-round search /
        # refinement kernels of a FISS+ step - run beside the next step's lattice kernel.  Same batches, same outputs; an extra leg,
        # never the headline (whose steps run back to back on one stream).
        eng2 = FrenetEngine(local_rank)
        stream2 = torch.cuda.Stream(dev)

        def measure2(ws):
            if prewarm_s > 0:
                wake(torch, dev, lambda: [(w.step(), w.fetch()) for w in ws[:2]], prewarm_s)
            for k in range(warm_x):
                ws[k % len(ws)].step(); ws[k % len(ws)].fetch()
            barrier()
            t0 = time.perf_counter()
            for k in range(steps_x):
                ws[(warm_x + k) % len(ws)].step(); ws[(warm_x + k) % len(ws)].fetch()
            barrier()
            el = time.perf_counter() - t0
            return {"value": float(np.mean([w.candidates for w in ws])) * steps_x / el, "unit": "candidates/s", "ms_per_step": el / steps_x * 1e3,
                    "steps": steps_x, "warmup": warm_x}

        torch.cuda.synchronize(dev)
        ws2 = [wls[k] if k % 2 == 0 else Workload(torch, eng2, wls[k].batch, dev, stream2) for k in range(len(wls))] if len(wls) >= 2 else \
              [main_wl, Workload(torch, eng2, batch, dev, stream2)]
        w4a, w4b = Workload(torch, eng, b4, dev, stream, fiss=True, hint=False), Workload(torch, eng2, b4, dev, stream2, fiss=True, hint=False)
        torch.cuda.synchronize(dev)
        o2 = measure2(ws2)
        o2["what"] = "the headline workload with its steps alternating between two fp_ctx / two HIP streams (launches of consecutive steps overlap)"
        o2["parity"] = gate("two_streams", ws2[1], 64)
        o2["config4"] = measure2([w4a, w4b])
        torch.cuda.synchronize(dev)
        # config 2 (256 egos: one workgroup per CU, latency bound) on two streams: two such launches share the chip
        w2a, w2b = Workload(torch, eng, b2, dev, stream), Workload(torch, eng2, b2, dev, stream2)
        torch.cuda.synchronize(dev)
        o2["config2"] = measure2([w2a, w2b])
        torch.cuda.synchronize(dev)
        o2["config2"]["parity"] = gate("two_streams config2", w2b, b2.B)
        del w2a, w2b
        if args.cpu_seconds > 0:
            o2["config4"]["parity"] = fiss_parity("two_streams config4", w4b, np.arange(0, B, max(1, B // 64)))
        extras["two_streams"] = o2
        # (d3) the same overlap for ONE caller, through the product's own switch: fp_ctx_set_option("overlap", 1) - consecutive independent
        # FP_MEM_DEVICE calls of ONE engine on ONE caller stream alternate between the ctx's two internal streams (ABI 15; the caller's
        # stream is joined with the call before the previous one, fp_ctx_join joins it with everything).  Same batches, same outputs.
        eng.set_option("overlap", 1)

        def measure_ov(ws):
            if prewarm_s > 0:
                wake(torch, dev, lambda: [(w.step(), w.fetch()) for w in ws[:2]], prewarm_s)
            for k in range(warm_x):
                ws[k % len(ws)].step(); ws[k % len(ws)].fetch()
            eng.join(stream.cuda_stream)
            barrier()
            t0 = time.perf_counter()
            for k in range(steps_x):
                ws[(warm_x + k) % len(ws)].step(); ws[(warm_x + k) % len(ws)].fetch()
            eng.join(stream.cuda_stream)
            barrier()
            el = time.perf_counter() - t0
            return {"value": float(np.mean([w.candidates for w in ws])) * steps_x / el, "unit": "candidates/s", "ms_per_step": el / steps_x * 1e3,
                    "steps": steps_x, "warmup": warm_x}

        try:
            n_ov = eng.get_option("overlapped_calls")
            oo = measure_ov(wls if len(wls) >= 2 else [main_wl, Workload(torch, eng, batch, dev, stream)])
            oo["what"] = ("the headline workload with fp_ctx_set_option(\"overlap\", 1): ONE engine, ONE caller stream; consecutive steps (distinct batches, distinct "
                          "output arrays) run on the ctx's two internal streams")
            oo["overlapped_calls"] = eng.get_option("overlapped_calls") - n_ov
            eng.join(stream.cuda_stream); torch.cuda.synchronize(dev)
            oo["parity"] = gate("overlap", wls[-1], 64)
            w4c = Workload(torch, eng, b4, dev, stream, fiss=True, hint=False)
            torch.cuda.synchronize(dev)
            oo4 = measure_ov([w4a, w4c])
            eng.join(stream.cuda_stream); torch.cuda.synchronize(dev)
            if args.cpu_seconds > 0:
                oo4["parity"] = fiss_parity("overlap config4", w4c, np.arange(0, B, max(1, B // 64)))
            extras["overlap"], extras["overlap_config4"] = oo, oo4
            del w4c
            # config 2 (256 egos: one workgroup per CU, latency bound): two such launches share the chip
            w2c, w2d = Workload(torch, eng, b2, dev, stream), Workload(torch, eng, b2, dev, stream)
            torch.cuda.synchronize(dev)
            oo2 = measure_ov([w2c, w2d])
            eng.join(stream.cuda_stream); torch.cuda.synchronize(dev)
            oo2["parity"] = gate("overlap config2", w2d, b2.B)
            extras["overlap_config2"] = oo2
            del w2c, w2d
        finally:
            eng.set_option("overlap", 0)
        del ws2, w4a, w4b
        # (e) many scenarios x many cycles on the device, and the PCIe-inclusive host-buffer entry
        if args.cpu_seconds > 0:
            extras["closed_loop"] = closed_loop_leg(torch, eng, dev, B, args.layout, 50, gate_threads, eng2, stream2, wake_s=0.5 * prewarm_s)
        del eng2
        t_host = []
        for _ in range(4):
            t0 = time.perf_counter()
            ho = eng.plan_dense(batch)
            t_host.append(time.perf_counter() - t0)
        par_h = fop_parity("host_buffers", batch, np.arange(0, B, max(1, B // 64)), ho.best_idx, ho.best_cost, gate_threads) if args.cpu_seconds > 0 else None
        extras["host_buffers"] = {"value": B * C / float(np.median(t_host[1:])), "unit": "candidates/s", "ms_per_call": float(np.median(t_host[1:])) * 1e3,
                                  "what": "fp_plan_dense with FP_MEM_HOST: pageable numpy inputs -> pack -> H2D -> kernels -> D2H (PCIe-inclusive; never the headline value)",
                                  "parity": par_h}
        # the same entry with fp_batch.tables_tag set (what the drop-in planners do): the frame / scene tables stay on the device after the
        # first call, only the per-ego arrays travel
        batch.tables_tag = 777001
        t_tag = []
        for _ in range(6):
            t0 = time.perf_counter()
            ht = eng.plan_dense(batch, tables=False)
            t_tag.append(time.perf_counter() - t0)
        batch.tables_tag = 0
        par_t = fop_parity("host_buffers_tagged", batch, np.arange(0, B, max(1, B // 64)), ht.best_idx, ht.best_cost, gate_threads) if args.cpu_seconds > 0 else None
        extras["host_buffers_tagged"] = {"value": B * C / float(np.median(t_tag[1:])), "unit": "candidates/s", "ms_per_call": float(np.median(t_tag[1:])) * 1e3,
                                         "first_call_ms": t_tag[0] * 1e3,
                                         "what": "fp_plan_dense with FP_MEM_HOST and fp_batch.tables_tag set, no tables asked for: the frame / scene tables are uploaded by the first "
                                                 "call and stay on the device; later calls move the per-ego arrays in and index / cost / Stats out (PCIe-inclusive; never the headline value)",
                                         "parity": par_t}

    if rank == 0:
        value = world * np.mean([w.candidates for w in wls]) * args.steps / elapsed
        bytes_launch = float(np.mean([w.algorithmic_bytes() for w in wls]))
        tr = load_profile_json("traffic.json") or {}
        traffic = tr.get(f"config{config}_B{B}_{args.layout}", tr.get(f"config{config}_B{B}") if args.layout == "lanes" else None)
        # executed VALU work of the dominant kernel from the committed PMC pass (a property of kernel + input, not of the run);
        # the rates use this run's kernel time
        valu_issue = fp64_exec = None
        # (the newest committed PMC pass; profiles/README.md says which commit it was collected on - the file's kernel name is carried in the line)
        pmc_file = next((f for f in (f"r06_config3_{args.layout}_pmc_summary.json", f"r05_config3_{args.layout}_pmc_summary.json", f"r04_config3_{args.layout}_pmc_summary.json")
                         if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
        pmc = load_profile_json(pmc_file) if pmc_file else None
        pmc_kernel = None
        if pmc and not fiss and config == 3 and B == 2048:
            try:
                pmc_kernel, k = next((kk, v) for kk, v in pmc.items() if "lattice_fused" in kk)
                insts = k["SQ_INSTS_VALU"]
                peak = 256 * 4 * 2.4e9 / 4.0
                valu_issue = {"wave_instructions_per_launch": insts, "rate": insts / (kern_ms * 1e-3), "peak": peak,
                              "unit": "wave-instructions/s", "frac": insts / (kern_ms * 1e-3) / peak,
                              "source": f"SQ_INSTS_VALU from profiles/{pmc_file} (rocprofv3 --pmc; kernel instance and commit in profiles/README.md), this run's kernel time; "
                                        "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction (the measured issue cost of nearly every instruction "
                                        "this kernel executes: profiles/r05_valu_issue_rate_microbench.txt)", "pmc_kernel": pmc_kernel}
                if "SQ_ACTIVE_INST_VALU" in k and k.get("GRBM_GUI_ACTIVE"):  # the hardware's own count of the VALU pipes' busy time in that pass
                    valu_issue["valu_busy_frac_pmc"] = k["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * k["GRBM_GUI_ACTIVE"] / 8.0)
                    valu_issue["valu_busy_note"] = "SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): VALU-busy share of the launch, tail included"
                flops = 64.0 * (2 * k["SQ_INSTS_VALU_FMA_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + k["SQ_INSTS_VALU_ADD_F64"] + k["SQ_INSTS_VALU_TRANS_F64"])
                fp64_exec = {"executed_flops_per_launch": flops, "achieved": flops / (kern_ms * 1e-3) / 1e12, "rate": flops / (kern_ms * 1e-3) / 1e12,
                             "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": flops / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                             "fp64_share_of_valu_instructions": (k["SQ_INSTS_VALU_FMA_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + k["SQ_INSTS_VALU_ADD_F64"] +
                                                                 k["SQ_INSTS_VALU_TRANS_F64"]) / insts,
                             "source": "64 lanes x (2 FMA + MUL + ADD + TRANS) wave-level FP64 instructions from the same PMC pass (inactive lanes counted: upper bound)"}
            except Exception:
                valu_issue = fp64_exec = None
        kname = ("lattice_fused_kernel: lattice + argmin workgroups and, appended to the same grid, the epilogue workgroups that write the winners' "
                 "series - the ONE launch of an fp_plan_dense call") if not fiss else "lattice_fused + fissplus_search + fiss_refine (whole pipeline)"
        line = {
            "metric": "candidate trajectories/sec (gen+cost+collision) per GPU; plan-cycle p50 latency", "value": value, "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{config - 1}]: {B} egos/GPU x {batch.nd}x{batch.nv}x{batch.nt} lattice "
                                   f"({C} cand/ego), {batch.n_obs} {'dynamic' if batch.meta.get('moving') else 'static'} obstacles, "
                                   f"T_obs={batch.T_obs}, stride-2 OBB checks, " + ("FISS+ search + 3 refinement rounds" if fiss else "FOP argmin")
                                   + f"; SURVEY 8d generator ('{args.layout}' layout), {n_rot} distinct batches cycled through the timed steps"
                                   + (f"; rank r plans egos [r*{B}, (r+1)*{B}) of the {world * B}-ego batch (+ {n_rot - 1} further shards of the same stream)" if world > 1 else ""),
                       "workload_short": f"BASELINE.json configs[{config - 1}]: {B} egos/GPU x {batch.nd}x{batch.nv}x{batch.nt} ({C} cand/ego), {batch.n_obs} "
                                         f"{'dynamic' if batch.meta.get('moving') else 'static'} obstacles, " + ("FISS+" if fiss else "FOP") + f", {n_rot} batches cycled",
                       "egos_per_gpu": B, "candidates_per_ego": C, "tables_written": bool(args.tables), "obstacle_layout": args.layout,
                       "rotating_batches": n_rot, "parallelism": f"ego-shard x{world} (no collectives)",
                       "launch_order": "per-batch learnt order (ctx feedback); before it exists: " + ("fp_batch.launch_order, egos by descending speed (input-only hint, host argsort at upload; results identical)" if main_wl.hinted else "index order"),
                       "input_digest": batch.digest()[:16], "input_digests": [w.batch.digest()[:16] for w in wls]},
            "parity": parity_main,
            "roofline": roofline_obj(bytes_launch, kern_ms, kname, traffic,
                                     "the kernel is VALU-issue bound, not HBM bound: `binding` / `valu_fp64` carry the executed FP64 work of the committed PMC pass "
                                     f"(profiles/{pmc_file}), `valu_issue` the instruction issue rate", fp64_exec),
            "host_enqueue_ms_per_step": enqueue_ms,
            # the SAME W + K steps timed first, in the fresh process, before the wake-up (the literal contract; `prewarm` below says what `value` adds)
            "value_cold": cold["value"] if cold else None, "ms_per_step_cold": cold["ms_per_step"] if cold else None,
            "cold_start": cold,
            "prewarm": {"steps": prewarm_steps, "seconds": prewarm_s, "what": "untimed steps of the same workload before the W warm-up steps: the GPU leaves its "
                        "idle clocks (a fresh process, W = 5: 0.160-0.164 ms per step timed on the ramp against 0.150-0.155); BENCH_PREWARM_S=0 disables it"},
            "kernel_ms_stats": {"mean": kern_ms, "min": float(np.min(kern_list)), "median": float(np.median(kern_list)), "max": float(np.max(kern_list)),
                                "launches_timed": len(kern_list)},
            "egos_with_a_feasible_candidate": float(np.mean([(w.h_idx.numpy() >= 0).mean() for w in wls])) if not fiss else None,
            "valu_issue": valu_issue,
            "valu_fp64_executed": fp64_exec,
            "cpu_baseline": cpu.get("cpu_baseline"),
            "cpu_baseline_1thread": cpu.get("cpu_baseline_1thread"),
            "cpu_baseline_optimised": cpu.get("cpu_baseline_optimised"),
            "plan_cycle_latency": plan_cycle,
            "materialize_mode": materialize,
        }
        line["roofline"]["kernel_short"] = "lattice_fused_kernel (lattice + argmin + winners' series: the one launch of fp_plan_dense)" if not fiss else \
            "lattice_fused + fiss_refine (the FISS+ pipeline's two launches)"
        if line["cpu_baseline"]:
            line["cpu_baseline"]["sample_short"] = f"first {line['cpu_baseline'].get('parity', {}).get('checked_egos', '?')} egos of the first timed batch, oracle/libfrenet_oracle.so"
        line.update(extras)
        emit(line)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
