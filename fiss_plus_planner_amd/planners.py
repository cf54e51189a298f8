"""Drop-in planner classes: same names, constructor and plan() signature as the reference
(planners/frenet_optimal_planner.py, fop_plus_planner.py, fiss_planner.py, fiss_plus_planner.py).

What runs where
---------------
Every trajectory evaluation (polynomials, cost, Frenet->Cartesian, speed/acceleration masks, OBB
collision tests, FOP argmin) runs in the gfx950 kernels behind the C ABI.  The host keeps only what the
reference also keeps per planner object - settings, the reference-line spline, the cross-cycle state
(`prev_best_idx`, `best_traj`) - plus the data-dependent *order of visits* of FOP+/FISS/FISS+, which is a
walk over the GPU-computed dense tables (search.py).  There is no CPU evaluation path: constructing a
planner without a GPU / built library raises.
"""
from __future__ import annotations

import itertools
import time

import numpy as np

from . import _abi, search
from .batch import FISS_KINDS, ProblemBatch
from .engine import TRAJ_STRIDE, FrenetEngine, host_structs, unpack_flags
from .frenet import FrenetState, FrenetTrajectory
from .obstacles import ObstacleTable, flatten_obstacles, obstacles_fingerprint
from .spline import CubicSpline2D
from .vehicle import Vehicle


class Stats:
    """reference frenet_optimal_planner.py:15-36"""

    def __init__(self, num_iter=0, num_trajs_generated=0, num_trajs_validated=0, num_collison_checks=0):
        self.num_iter = num_iter
        self.num_trajs_generated = num_trajs_generated
        self.num_trajs_validated = num_trajs_validated
        self.num_collison_checks = num_collison_checks

    def __add__(self, other):
        self.num_iter += other.num_iter
        self.num_trajs_generated += other.num_trajs_generated
        self.num_trajs_validated += other.num_trajs_validated
        self.num_collison_checks += other.num_collison_checks
        return self

    def average(self, value: int):
        self.num_iter /= value
        self.num_trajs_generated /= value
        self.num_trajs_validated /= value
        self.num_collison_checks /= value
        return self

    def as_tuple(self):
        return (self.num_iter, self.num_trajs_generated, self.num_trajs_validated, self.num_collison_checks)


_TABLE_TAGS = itertools.count(1)  # fp_batch.tables_tag values: one per (planner, centerline, obstacle list) the process ever builds

class FrenetOptimalPlannerSettings:
    """reference frenet_optimal_planner.py:38-56"""

    def __init__(self, num_width: int = 5, num_speed: int = 5, num_t: int = 5):
        self.tick_t = 0.1
        self.max_road_width = 3.5
        self.num_width = num_width
        self.highest_speed = 13.4112
        self.lowest_speed = 0.0
        self.num_speed = num_speed
        self.min_t = 8.0
        self.max_t = 10.0
        self.num_t = num_t
        self.check_obstacle = True   # present in the reference, never read there either
        self.check_boundary = True
        # not in the reference's settings: turns on the curvature / curvature-rate checks that its check_constraints carries
        # commented out (:145-150), with the Vehicle's max_curvature / max_kappa_d / max_kappa_dd (vehicle.py:44-46)
        self.check_curvature = False


class FissPlannerSettings(FrenetOptimalPlannerSettings):
    """reference fiss_planner.py:13-18"""

    def __init__(self, num_width: int = 5, num_speed: int = 5, num_t: int = 5):
        super().__init__(num_width, num_speed, num_t)
        self.w_heuristic = 10.0
        self.vis_all_candidates = False


class FissPlusPlannerSettings(FissPlannerSettings):
    """reference fiss_plus_planner.py:15-22.  `time_limit` is a WALL-CLOCK budget (:152-158, :293-299): refinement is entered when
    `not has_time_limit or time_left > 0` and its loop breaks after the first gradient step that ends past `time_left` - even with
    has_time_limit False.  The drop-in applies it with its own clock (FissPlusPlanner._refine_rounds): a budget that is already spent
    when the coarse search returns gives no refinement (has_time_limit) or exactly one round (fixture G14: the two cases the clock
    cannot change); otherwise all rounds run - on the device they take tens of microseconds, far inside any budget the reference's
    own coarse search (tens of milliseconds on a CPU) would leave."""

    def __init__(self, num_width: int = 5, num_speed: int = 5, num_t: int = 5, refine_iters: int = 3):
        super().__init__(num_width, num_speed, num_t)
        self.refine_trajectory = True
        self.max_refine_iters = refine_iters
        self.has_time_limit = False
        self.time_limit = 0.5
        self.decaying_factor = 0.5


_shared_engines: dict[int, FrenetEngine] = {}


def _engine_for(device: int) -> FrenetEngine:
    if device not in _shared_engines:
        _shared_engines[device] = FrenetEngine(device)
    return _shared_engines[device]


class FrenetOptimalPlanner:
    """FOP: exhaustive lattice, argmin over the feasible candidates (reference frenet_optimal_planner.py:58-278).

    Table caching (``cache_tables=True``, the default): the reference-line spline and the flattened obstacle table of a plan()
    call stay on the device until the planner sees ANOTHER centerline / obstacle list / ``ObstacleTable`` (or the same table
    after ``ObstacleTable.update()``); per cycle only the start state travels.  The cached numpy arrays are frozen (write flag
    off) while the planner holds them, so editing an ``ObstacleTable`` or the spline tables in place raises ``ValueError``
    instead of silently checking collisions against stale device data.  A list of obstacle OBJECTS is re-flattened whenever the
    list holds other objects or another horizon (``obstacles_fingerprint``); mutating such an object in place is not visible -
    pass ``cache_tables=False`` to flatten nothing ahead and upload every cycle, as the planner did before round 3."""

    KIND = "FOP"

    def __init__(self, planner_settings: FrenetOptimalPlannerSettings, ego_vehicle: Vehicle, scenario=None, *,
                 device: int = 0, engine: FrenetEngine | None = None, materialize_all: bool = False, frame_on: str = "device",
                 cache_tables: bool = True):
        self.cache_tables = cache_tables
        self.settings = planner_settings
        self.vehicle = ego_vehicle
        self.cubic_spline = None
        self.best_traj = None
        self.all_trajs = []
        self.stats = Stats()
        self.materialize_all = materialize_all  # fill all_trajs with every candidate (visualisation payload)
        assert frame_on in ("device", "host")
        self.frame_on = frame_on  # where generate_frenet_frame() solves the spline system (fp_frames_build or numpy)
        self._engine = engine if engine is not None else _engine_for(device)
        self._obs_cache = (None, None, None)
        self.last_tables = None  # (cost [C], flags [C]) of the last dense pass, flat FOP order

    # ------------------------------------------------------------------ frame
    def generate_frenet_frame(self, centerline_pts: np.ndarray):
        """reference :272-278 -> (CubicSpline2D, [n', 4] = x, y, yaw, kappa every 0.1 m)."""
        pts = np.asarray(centerline_pts, dtype=np.float64)
        tables = None
        if self.frame_on == "device" and 2 <= len(pts) <= 512:
            knots, coef = self._engine.build_frames(pts[None, :, :2])  # fp_frames_build: Thomas sweep on the GPU
            tables = (knots[0], coef[0])
        self.cubic_spline = CubicSpline2D(pts[:, 0], pts[:, 1], tables)
        s = np.arange(0, self.cubic_spline.s[-1], 0.1)
        x, y, yaw, kappa = self.cubic_spline.sample(s)
        return self.cubic_spline, np.column_stack((x, y, yaw, kappa))

    # ------------------------------------------------------------------ problem marshalling
    def _obstacle_table(self, obstacles) -> ObstacleTable | None:
        if isinstance(obstacles, ObstacleTable):
            return obstacles
        if obstacles is None or len(obstacles) == 0:
            return None  # has_collision: empty list -> no collision (:170-171)
        # The flattened table is reused only while the list holds the SAME obstacle objects (element identities + horizon); the
        # cache keeps them alive, so a recycled address cannot alias a different obstacle.
        if not self.cache_tables:  # re-read every object every cycle (in-place updates of the obstacle objects are seen)
            return flatten_obstacles(obstacles)
        key = obstacles_fingerprint(obstacles)
        if self._obs_cache[0] != key:
            self._obs_cache = (key, flatten_obstacles(obstacles), list(obstacles))
        return self._obs_cache[1]

    def _sampling_width(self) -> float:
        return self.settings.max_road_width - self.vehicle.w + (0.3 if self.KIND in FISS_KINDS else 0.0)

    def _make_batch(self, frenet_state: FrenetState, obstacles, time_step_now: int) -> ProblemBatch:
        """The B = 1 problem batch of this plan() call.  Everything that does not change between cycles (lattice grids, spline
        table, obstacle table) is built once and reused; per cycle only the start state, t_now and - when max_target_speed
        changed - the speed samples are rewritten in place."""
        if self.cubic_spline is None:
            raise RuntimeError("generate_frenet_frame() must be called before plan()")
        st = self.settings
        tab = self._obstacle_table(obstacles)
        sp = self.cubic_spline
        curv = (self.vehicle.max_curvature, self.vehicle.max_kappa_d, self.vehicle.max_kappa_dd) if getattr(st, "check_curvature", False) else None
        cache_tables = getattr(self, "cache_tables", True)
        key = (id(sp), id(tab), getattr(tab, "version", 0), cache_tables, st.num_width, st.num_speed, st.num_t, st.min_t, st.max_t, st.tick_t, st.max_road_width, st.lowest_speed,
               self.vehicle.l, self.vehicle.w, self.vehicle.max_speed, self.vehicle.max_accel, curv)
        cache = getattr(self, "_batch_cache", None)
        if cache is None or cache[0] != key:
            sw = self._sampling_width()
            d, rd = np.linspace(-sw / 2, sw / 2, st.num_width, retstep=True)
            t, rt = np.linspace(st.min_t, st.max_t, st.num_t, retstep=True)
            if tab is None:
                pose, dims, fts, scene = np.zeros((0, 1, 0, 4)), np.zeros((0, 0, 2)), np.zeros(0, dtype=np.int32), -1
            else:
                pose, dims, fts, scene = tab.pose[None], tab.dims[None], np.array([tab.final_time_step], dtype=np.int32), 0
            batch = ProblemBatch(
                d_samples=d, t_samples=t, v_samples=np.zeros((1, st.num_speed)), target_speed=np.zeros(1), ego=np.zeros((1, 6)),
                frame_of=[0], scene_of=[scene], t_now=[0], nx=[len(sp.knots)], knots=sp.knots[None], coef=sp.coef[None],
                obs_pose=pose, obs_dims=dims, final_time_step=fts, veh_l=self.vehicle.l, veh_w=self.vehicle.w,
                max_speed=self.vehicle.max_speed, max_accel=self.vehicle.max_accel, tick_t=st.tick_t, check_stride=2,
                samp_min=np.array([[-sw / 2, st.lowest_speed, st.min_t]]), samp_max=np.array([[sw / 2, 0.0, st.max_t]]),
                samp_res=np.array([[rd, 0.0, rt]]), curvature_limits=curv,
                obs_poly=None if tab is None or tab.nvert is None else tab.poly[None],
                obs_nvert=None if tab is None or tab.nvert is None else tab.nvert[None])
            # fp_batch.tables_tag: the library keeps this batch's spline and obstacle tables on the device until the planner builds
            # a new batch (another centerline / another obstacle list / ObstacleTable.update()) - per cycle only the start state
            # travels.  Contract (class docstring): the cached arrays are frozen, so an in-place edit raises instead of going stale;
            # `cache_tables=False` uploads them every cycle and freezes nothing.
            batch.tables_tag = next(_TABLE_TAGS) if cache_tables else 0
            if cache_tables:
                if tab is not None:
                    tab.freeze()
                sp.knots.setflags(write=False)
                sp.coef.setflags(write=False)
            host_structs(batch, freeze=True)  # (this batch is the planner's own: its arrays are only ever updated in place)
            cache = [key, batch, None, sp, tab]  # sp / tab kept alive so their ids cannot be recycled
            self._batch_cache = cache
        batch = cache[1]
        if cache[2] != st.highest_speed:
            v, rv = np.linspace(st.lowest_speed, st.highest_speed, st.num_speed, retstep=True)
            batch.v_samples[0] = v
            batch.target_speed[0] = st.highest_speed
            batch.samp_max[0, 1] = st.highest_speed
            batch.samp_res[0, 1] = rv
            cache[2] = st.highest_speed
        fs = frenet_state
        batch.ego[0] = (fs.s, fs.s_d, fs.s_dd, fs.d, fs.d_d, fs.d_dd)
        batch.t_now[0] = time_step_now
        return batch

    def _stride(self) -> int:
        """Columns of a series row: 128 as long as max_t / tick_t fits (the ABI's default), else the next multiple of 16 (tick_t = 0.05:
        200 points -> 208)."""
        st = self.settings
        n = int(np.ceil(st.max_t / st.tick_t))
        return TRAJ_STRIDE if n <= TRAJ_STRIDE else (n + 15) // 16 * 16

    def _materialize(self, batch: ProblemBatch, end_states: np.ndarray, idxs=None):
        """end_states [K,3] -> list of FrenetTrajectory (full series from the GPU dump)."""
        es = np.asarray(end_states, dtype=np.float64).reshape(1, -1, 3)
        out = self._engine.eval_trajs(batch, es, dump=True, traj_stride=self._stride())
        _, N, M = unpack_flags(out.flags[0])
        trajs = []
        for k in range(es.shape[1]):
            end = FrenetState(t=es[0, k, 2], s=0.0, s_d=es[0, k, 1], d=es[0, k, 0])
            trajs.append(FrenetTrajectory.from_dump(out.traj[0, k], int(N[k]), int(M[k]), out.cost[0, k], end,
                                                    None if idxs is None else idxs[k]))
        return trajs, out

    def _end_state_of_flat(self, batch: ProblemBatch, flat: int):
        nv, nt = batch.nv, batch.nt
        iv, it, i_d = flat % nv, (flat // nv) % nt, flat // (nv * nt)
        return np.array([batch.d_samples[i_d], batch.v_samples[0, iv], batch.t_samples[it]]), (i_d, iv, it)

    def _dense(self, batch: ProblemBatch, winner: bool = False):
        # the output arrays (and the fp_result over them) are allocated once per planner and reused every cycle; what outlives
        # the call is copied out of them (last_tables here, the winner's series in FrenetTrajectory.from_dump)
        stride = self._stride()
        key = (batch.C, winner, stride)
        outs = self.__dict__.setdefault("_dense_outs", {})
        reuse = outs.get(key)
        if reuse is None:
            reuse = outs[key] = self._engine.dense_outputs(1, batch.C, True, winner, stride)
        out = self._engine.plan_dense(batch, tables=True, winner=winner, traj_stride=stride, out=reuse)
        self.last_tables = (out.cost[0].copy(), out.flags[0].copy())
        if self.materialize_all:  # visualisation payload (reference :102): every candidate's series in one launch
            m = self._engine.materialize_all(batch, traj_stride=stride)
            _, N, M = unpack_flags(m.flags[0])
            self.all_trajs.append([FrenetTrajectory.from_dump(m.traj[0, c], int(N[c]), int(M[c]), float(out.cost[0, c]))
                                   for c in range(batch.C)])
        else:
            self.all_trajs.append([])
        return out

    # ------------------------------------------------------------------ plan
    def plan(self, frenet_state: FrenetState, max_target_speed: float, obstacles, time_step_now: int = 0):
        """reference :247-270.  Returns the minimum-cost feasible FrenetTrajectory; when no candidate survives the
        reference returns its previous `best_traj` (stale object or None) - reproduced."""
        self.stats = Stats()
        self.settings.highest_speed = max_target_speed
        batch = self._make_batch(frenet_state, obstacles, time_step_now)
        out = self._dense(batch, winner=True)  # one call, one launch: lattice + argmin + the winner's series
        C = batch.C
        self.stats = Stats(0, C, C, C)
        best = int(out.best_idx[0])
        if best >= 0:
            fl = int(out.best_flags[0])  # N and M ride in the flag word (FP_FLAG_N_SHIFT / FP_FLAG_M_SHIFT)
            self.best_traj = FrenetTrajectory.from_dump(out.best_traj[0], (fl >> 8) & 0xFFF, fl >> 20, float(out.best_cost[0]))
            self.best_traj.lattice_index = best
        return self.best_traj


class FopPlusPlanner(FrenetOptimalPlanner):
    """FOP+: validate in cost order, stop at the first survivor (reference fop_plus_planner.py:11-41)."""

    KIND = "FOP+"

    def plan(self, frenet_state: FrenetState, max_target_speed: float, obstacles, time_step_now: int = 0):
        self.stats = Stats()
        self.settings.highest_speed = max_target_speed
        batch = self._make_batch(frenet_state, obstacles, time_step_now)
        out = self._dense(batch)
        best, st = search.fopplus_search(out.cost[0], out.flags[0])
        self.stats = Stats(*st)
        if best is None:
            return None
        es, _ = self._end_state_of_flat(batch, best)
        trajs, _ = self._materialize(batch, es[None])
        self.best_traj = trajs[0]
        self.best_traj.lattice_index = best
        return self.best_traj


class FissPlanner(FrenetOptimalPlanner):
    """FISS: heuristic initial guess + gradient walk on the index grid (reference fiss_planner.py:20-270)."""

    KIND = "FISS"
    _search = staticmethod(search.fiss_search)

    def __init__(self, planner_settings: FissPlannerSettings, ego_vehicle: Vehicle, scenario=None, *, search_on: str = "device", **kw):
        """search_on="device": the whole plan() is one fp_plan_fiss call (lattice, search walk, refinement and winner series
        all on the GPU); "host": dense tables from the GPU, the index walk replayed in Python (search.py) - same results,
        kept as an independent cross-check."""
        super().__init__(planner_settings, ego_vehicle, scenario, **kw)
        assert search_on in ("device", "host")
        self.search_on = search_on
        self.sampling_res = np.empty(3)
        self.sampling_min = np.empty(3)
        self.sampling_max = np.empty(3)
        self.sizes = None
        self.start_state = None
        self.prev_best_idx = None

    def _coarse(self, frenet_state, max_target_speed, obstacles, time_step_now):
        self.stats = Stats()
        self.settings.highest_speed = max_target_speed
        self.start_state = frenet_state
        self.best_traj = None
        batch = self._make_batch(frenet_state, obstacles, time_step_now)
        self.sampling_min, self.sampling_max, self.sampling_res = batch.samp_min[0].copy(), batch.samp_max[0].copy(), batch.samp_res[0].copy()
        self.sizes = np.array([batch.nd, batch.nv, batch.nt])
        out = self._engine.plan_dense(batch, tables=True)
        self.last_tables = (out.cost[0], out.flags[0])
        J, F = search.tables_to_dvt(out.cost[0], out.flags[0], batch.nd, batch.nv, batch.nt)
        E = search.cost_est_table(batch.d_samples, batch.v_samples[0], batch.t_samples, self.sampling_min, self.sampling_max,
                                  self.prev_best_idx, self.settings.w_heuristic)
        order = []
        idx, st = self._search(J, F, E, order)
        self.stats = Stats(*st)
        # all_trajs (visualisation payload): the reference appends the trajectories it GENERATED this cycle, in generation order
        # (trajs_per_timestep, fiss_planner.py:131,262-265); with materialize_all their full series come from one eval launch
        if self.materialize_all and order:
            es = np.array([[batch.d_samples[i], batch.v_samples[0, j], batch.t_samples[k]] for i, j, k in order])
            trajs, _ = self._materialize(batch, es, [np.array(o) for o in order])
            self.all_trajs.append(trajs)
        else:
            self.all_trajs.append([])
        return batch, idx

    def _plan_on_device(self, frenet_state, max_target_speed, obstacles, time_step_now):
        """plan() as ONE C-ABI call: fp_plan_fiss (reference fiss_planner.py:190-270 / fiss_plus_planner.py:61-170)."""
        self.stats = Stats()
        self.settings.highest_speed = max_target_speed
        self.start_state = frenet_state
        self.best_traj = None
        batch = self._make_batch(frenet_state, obstacles, time_step_now)
        self.sampling_min, self.sampling_max, self.sampling_res = batch.samp_min[0].copy(), batch.samp_max[0].copy(), batch.samp_res[0].copy()
        self.sizes = np.array([batch.nd, batch.nv, batch.nt])
        st = self.settings
        plus = self.KIND == "FISS+"
        R = self._refine_rounds() if plus else 0
        prev = None if self.prev_best_idx is None else np.asarray(self.prev_best_idx, dtype=np.int32)[None]
        outs = self.__dict__.setdefault("_fiss_outs", {})  # output arrays (+ the fp_fiss_io over them) reused every cycle
        stride = self._stride()
        reuse = outs.get((R, stride))
        if reuse is None:
            reuse = outs[(R, stride)] = self._engine.fiss_outputs(1, R, True, traj_stride=stride)
        out = self._engine.plan_fiss(batch, self.KIND, prev_best_idx=prev, w_heuristic=st.w_heuristic, max_refine_iters=R,
                                     decaying_factor=getattr(st, "decaying_factor", 0.5), winner=True, traj_stride=stride, out=reuse)
        # (plain Python scalars from here on: every numpy call on a one-element array costs about a microsecond of the plan cycle)
        self.stats = Stats(*out.stats[0].tolist())
        self.all_trajs.append([])
        best_cost = float(out.best_cost[0])
        found = best_cost == best_cost
        if plus and R > 0 and found:
            self.sampling_res = self.sampling_res * st.decaying_factor ** R  # decays in place in the reference (:282)
        if not found:
            return None
        self.prev_best_idx = out.prev_best_idx[0].copy()
        fl = int(out.best_flags[0])  # N and M ride in the flag word (FP_FLAG_N_SHIFT / FP_FLAG_M_SHIFT)
        es = out.end_state[0].tolist()
        end = FrenetState(t=es[2], s=0.0, s_d=es[1], d=es[0])
        idx = [-1, -1, -1] if out.refined[0] else out.best_ijk[0].tolist()
        self.best_traj = FrenetTrajectory.from_dump(out.best_traj[0], (fl >> 8) & 0xFFF, fl >> 20, best_cost, end, idx)
        return self.best_traj

    def _refine_rounds(self) -> int:
        """FissPlusPlanner's refinement rounds for this plan() call under the reference's wall-clock budget (fiss_plus_planner.py:152-158,
        :293-299): time_left = time_limit - (time since plan() started).  Spent already: none with has_time_limit, else ONE (the loop
        checks the clock after its first gradient step).  Otherwise max_refine_iters."""
        st = self.settings
        if not getattr(st, "refine_trajectory", False):
            return 0
        R = int(getattr(st, "max_refine_iters", 0))
        if R <= 0:
            return 0
        time_left = float(getattr(st, "time_limit", float("inf"))) - (time.perf_counter() - getattr(self, "_t_plan", time.perf_counter()))
        if time_left <= 0.0:
            return 0 if getattr(st, "has_time_limit", False) else 1
        return R

    def _device_walk(self) -> bool:
        """One fp_plan_fiss call, unless the caller wants the host walk, the generated set (`all_trajs`: only the host walk knows it) or
        a lattice beyond the device walk's FP_MAX_CAND_SEARCH samples (the dense pass takes up to FP_MAX_CAND; its tables are then
        walked on the host)."""
        st = self.settings
        # (round 6: the device refinement holds up to FP_MAX_POINTS = 256 points per trajectory - fiss_refine_kernel<4> -, as the dense pass does)
        return self.search_on == "device" and not self.materialize_all and st.num_width * st.num_speed * st.num_t <= _abi.FP_MAX_CAND_SEARCH

    def plan(self, frenet_state: FrenetState, max_target_speed: float, obstacles, time_step_now: int = 0):
        if self._device_walk():
            return self._plan_on_device(frenet_state, max_target_speed, obstacles, time_step_now)
        batch, idx = self._coarse(frenet_state, max_target_speed, obstacles, time_step_now)
        if idx is None:
            return None
        es = np.array([batch.d_samples[idx[0]], batch.v_samples[0, idx[1]], batch.t_samples[idx[2]]])
        trajs, _ = self._materialize(batch, es[None], [np.array(idx)])
        self.best_traj = trajs[0]
        self.prev_best_idx = self.best_traj.idx  # persists across cycles (:252)
        return self.best_traj


class FissPlusPlanner(FissPlanner):
    """FISS+: best-first frontier search + continuous refinement (reference fiss_plus_planner.py:24-326)."""

    KIND = "FISS+"
    _search = staticmethod(search.fissplus_search)

    def plan(self, frenet_state: FrenetState, max_target_speed: float, obstacles, time_step_now: int = 0):
        self._t_plan = time.perf_counter()  # (the reference's start_time: `time_limit` is measured from here, :152-154)
        if self._device_walk():
            return self._plan_on_device(frenet_state, max_target_speed, obstacles, time_step_now)
        batch, idx = self._coarse(frenet_state, max_target_speed, obstacles, time_step_now)
        if idx is None:
            return None
        x = np.array([batch.d_samples[idx[0]], batch.v_samples[0, idx[1]], batch.t_samples[idx[2]]])
        coarse_cost = float(self.last_tables[0][(idx[0] * batch.nt + idx[2]) * batch.nv + idx[1]])
        self.prev_best_idx = np.array(idx)  # stays the coarse index even if a refined trajectory wins (:140)
        winner = (x, np.array(idx))
        st = self.settings
        # refinement is entered when `not has_time_limit or time_left > 0` (:156); inside, the clock is checked after every gradient step
        time_left = float(getattr(st, "time_limit", float("inf"))) - (time.perf_counter() - self._t_plan)
        if st.refine_trajectory and st.max_refine_iters > 0 and (not getattr(st, "has_time_limit", False) or time_left > 0.0):
            refined = self._refine(batch, x, coarse_cost, time_left)
            if refined is not None:
                winner = (refined, np.array([-1, -1, -1]))
        trajs, _ = self._materialize(batch, winner[0][None], [winner[1]])
        self.best_traj = trajs[0]
        return self.best_traj

    def _refine(self, batch: ProblemBatch, x: np.ndarray, coarse_cost: float, time_limit: float = float("inf")):
        """refine_solution + gradient_decent (reference :207-326): per round six probe trajectories at
        clip(x -/+ res_dim e_dim), finite-difference gradient, resolution decay, one trajectory at the new x.
        All 7 trajectories of a round are evaluated on the GPU (two launches: 6 probes, then the step)."""
        st = self.settings
        res = self.sampling_res  # decays in place like the reference (aliases self.sampling_res, :282)
        cand = []  # (cost, order, end_state, flags)
        lo, hi = self.sampling_min, self.sampling_max
        # The coarse cost is re-evaluated by the same kernel as the refined trajectories: a probe clipped back onto x is
        # the same trajectory and must tie with it exactly (`cost > coarse cost` ends the validation loop, :303-304).
        coarse_cost = float(self._engine.eval_trajs(batch, x[None, None]).cost[0, 0])
        t_start = time.perf_counter()
        for _ in range(st.max_refine_iters):
            probes = np.empty((6, 3)); x_l = []; x_r = []
            for dim in range(3):
                a = x.copy(); a[dim] -= res[dim]; a = np.clip(a, lo, hi)
                b = x.copy(); b[dim] += res[dim]; b = np.clip(b, lo, hi)
                probes[2 * dim], probes[2 * dim + 1] = a, b
                x_l.append(a); x_r.append(b)
            if np.isnan(probes).any():
                break
            out = self._engine.eval_trajs(batch, probes[None])
            self.stats.num_trajs_generated += 6
            for k in range(6):
                cand.append((float(out.cost[0, k]), len(cand), probes[k].copy(), int(out.flags[0, k])))
            x_new, res_new = search.refine_step(out.cost[0, 0::2], out.cost[0, 1::2], x_l, x_r, x, res, st.decaying_factor, lo, hi)
            res[:] = res_new
            if x_new is None:
                break  # zero gradient: the reference raises inside np.arange(nan); we stop refining
            out1 = self._engine.eval_trajs(batch, x_new[None, None])
            self.stats.num_trajs_generated += 1
            cand.append((float(out1.cost[0, 0]), len(cand), x_new.copy(), int(out1.flags[0, 0])))
            x = x_new
            if time.perf_counter() - t_start >= time_limit:  # "Refinement time is up" (:296-299): checked AFTER a gradient step
                break
        # refined_trajs PriorityQueue: pop in cost order while cost <= coarse cost (:301-323)
        for cost, _, es, fl in sorted(cand, key=lambda c: (c[0], c[1])):
            if cost > coarse_cost:
                break
            self.stats.num_trajs_validated += 1
            if fl & search.FLAG_CONSTRAINTS:
                continue
            self.stats.num_collison_checks += 1
            if not (fl & search.FLAG_COLLISION):
                return es
        return None
