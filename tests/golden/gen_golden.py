#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by RUNNING THE REFERENCE.

Build-container only (needs /root/reference; see refshim.py).  Usage:

    python tests/golden/gen_golden.py [g1 g2 g3 g4 g5 g6 g7 g8 ...]   (default: all)

Every fixture is self-contained data: the inputs (problem-batch arrays, so the
GPU box can rebuild the identical problem without the reference) and the
reference's outputs.  Collision outcomes come from the reference's own
has_collision() running on refshim's convex-polygon stand-in for shapely, i.e.
they are pinned to OUR separating-axis test, not to GEOS ("collision parity
unpinned").
"""
from __future__ import annotations

import os
import sys
import time
import xml.etree.ElementTree as ET
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import refshim  # noqa: E402

R = refshim.load_reference()

from fiss_plus_planner_amd import synth  # noqa: E402
from fiss_plus_planner_amd.batch import ProblemBatch  # noqa: E402

ARRAYS = ["t", "s", "s_d", "s_dd", "s_ddd", "d", "d_d", "d_dd", "d_ddd", "x", "y", "yaw", "ds", "c", "c_d", "c_dd"]
BATCH_FIELDS = ["d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots",
                "coef", "obs_pose", "obs_dims", "final_time_step", "samp_min", "samp_max", "samp_res"]
BATCH_SCALARS = ["veh_l", "veh_w", "max_speed", "max_accel", "tick_t", "check_stride"]


OUT_DIR = os.environ.get("GOLDEN_OUT", HERE)  # GOLDEN_OUT=/tmp/x: regenerate beside the committed fixtures (tools/audit_goldens.sh)


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def batch_to_dict(b: ProblemBatch, prefix="in_"):
    out = {prefix + k: getattr(b, k) for k in BATCH_FIELDS if getattr(b, k) is not None}
    out[prefix + "scalars"] = np.array([getattr(b, k) for k in BATCH_SCALARS], dtype=np.float64)
    return out


def vehicle_for(b: ProblemBatch):
    p = refshim.vw_vanagon_params()
    p.l, p.w = b.veh_l, b.veh_w
    p.longitudinal.v_max, p.longitudinal.a_max = b.max_speed, b.max_accel
    return R.Vehicle(p)


def make_planner(kind: str, b: ProblemBatch, e: int, refine_iters=3):
    veh = vehicle_for(b)
    if kind == "FOP":
        pl = R.FrenetOptimalPlanner(R.FrenetOptimalPlannerSettings(b.nd, b.nv, b.nt), veh, None)
    elif kind == "FOP+":
        pl = R.FopPlusPlanner(R.FrenetOptimalPlannerSettings(b.nd, b.nv, b.nt), veh, None)
    elif kind == "FISS":
        pl = R.FissPlanner(R.FissPlannerSettings(b.nd, b.nv, b.nt), veh, None)
    elif kind == "FISS+":
        st = R.FissPlusPlannerSettings(b.nd, b.nv, b.nt, refine_iters)
        st.time_limit = 1e9  # the limit is enforced even with has_time_limit=False (fiss_plus_planner.py:296-298)
        pl = R.FissPlusPlanner(st, veh, None)
    else:
        raise ValueError(kind)
    if b.tick_t != 0.1:
        pl.settings.tick_t = float(b.tick_t)  # (a free attribute of the settings object, frenet_optimal_planner.py:38-56; G13)
    f = int(b.frame_of[e])
    nx = int(b.nx[f])
    pts = np.column_stack([b.coef[f, 0, :nx], b.coef[f, 4, :nx]])
    pl.generate_frenet_frame(pts)
    return pl


def ego_state(b: ProblemBatch, e: int):
    s, s_d, s_dd, d, d_d, d_dd = b.ego[e]
    return R.FrenetState(t=0.0, s=s, s_d=s_d, s_dd=s_dd, d=d, d_d=d_d, d_dd=d_dd)


def obstacles_for(b: ProblemBatch, e: int):
    sc = int(b.scene_of[e])
    if sc < 0 or b.n_obs == 0:
        return []
    return refshim.obstacles_from_table(b.obs_pose[sc], b.obs_dims[sc], int(b.final_time_step[sc]))


def dump_traj(fp, stride=128):
    out = np.full((16, stride), np.nan)
    for k, name in enumerate(ARRAYS):
        a = np.asarray(getattr(fp, name), dtype=np.float64)
        out[k, : len(a)] = a
    return out


# ---------------------------------------------------------------------------
# special inputs
# ---------------------------------------------------------------------------
def with_overrides(b: ProblemBatch, **kw) -> ProblemBatch:
    d = {k: getattr(b, k) for k in BATCH_FIELDS + BATCH_SCALARS}
    d.update(kw)
    return ProblemBatch(**d, meta=dict(b.meta))


def short_frame_batch(b: ProblemBatch, n_keep=25) -> ProblemBatch:
    """Cut every centerline to its first n_keep knots (120 m) so trajectories run off the end."""
    from fiss_plus_planner_amd.spline import build_frames

    pts = np.stack([b.coef[:, 0, :], b.coef[:, 4, :]], axis=-1)
    n = np.full(b.F, n_keep)
    knots, coef = build_frames(pts, n)
    return with_overrides(b, knots=knots, coef=coef, nx=n)


def fop_cases():
    """(name, batch) list used by g3/g4."""
    cases = []
    c2 = synth.make_batch(4, 5, 5, 5, 10, 100, False, 9102)
    cases.append(("c2_static10", c2))
    c3 = synth.make_batch(3, 9, 9, 7, 50, 50, True, 9103)
    cases.append(("c3_moving50", c3))
    c0 = synth.make_batch(3, 5, 5, 5, 0, 0, False, 9100)
    cases.append(("c0_noobs", c0))
    # truncation: short centerline; ego 2 starts beyond the end (M = 0)
    ct = short_frame_batch(synth.make_batch(4, 5, 5, 5, 10, 100, False, 9104), 25)
    ego = ct.ego.copy()
    ego[2, 0] = 130.0
    ego[3, 0] = ct.knots[3, 24] - 0.02  # M = 1: s(0) is on the spline, s(0.1) is already past its end
    ego[3, 1] = 0.5
    ego[3, 2] = 0.0
    cases.append(("trunc", with_overrides(ct, ego=ego)))
    # constraint violations: tight vehicle limits
    cv = synth.make_batch(3, 5, 5, 5, 10, 100, False, 9105)
    cases.append(("limits", with_overrides(cv, max_speed=11.0, max_accel=1.2)))
    # exact mirror ties (d = d_d = d_dd = 0) - FOP "last minimum wins"
    cm = synth.make_batch(2, 5, 5, 5, 0, 0, False, 9106)
    ego = cm.ego.copy()
    ego[:, 3:] = 0.0
    cases.append(("mirror", with_overrides(cm, ego=ego)))
    # later planning cycle: t_now > 0 shortens the collision horizon
    cn = synth.make_batch(2, 5, 5, 5, 10, 60, True, 9107)
    cases.append(("tnow", with_overrides(cn, t_now=np.array([7, 40]))))
    return cases


# ---------------------------------------------------------------------------
def g1():
    rng = np.random.default_rng(11)
    K = 64
    q5 = np.column_stack([rng.uniform(-2, 2, K), rng.uniform(-1, 1, K), rng.uniform(-1, 1, K), rng.uniform(-1, 1, K),
                          np.zeros(K), np.zeros(K), rng.uniform(2, 12, K)])
    q5[K // 2:, 4:6] = rng.uniform(-0.5, 0.5, (K - K // 2, 2))
    q4 = np.column_stack([rng.uniform(0, 80, K), rng.uniform(0, 15, K), rng.uniform(-2, 2, K), rng.uniform(0, 14, K),
                          np.zeros(K), rng.uniform(2, 12, K)])
    c5 = np.empty((K, 6)); c4 = np.empty((K, 5))
    ts = rng.uniform(0, 10, (K, 4))
    e5 = np.empty((K, 4, 4)); e4 = np.empty((K, 4, 4))
    for k in range(K):
        p = R.QuinticPolynomial(*q5[k])
        c5[k] = [p.a0, p.a1, p.a2, p.a3, p.a4, p.a5]
        q = R.QuarticPolynomial(*q4[k])
        c4[k] = [q.a0, q.a1, q.a2, q.a3, q.a4]
        for m, t in enumerate(ts[k]):
            t = np.float64(t)
            e5[k, m] = [p.calc_point(t), p.calc_first_derivative(t), p.calc_second_derivative(t), p.calc_third_derivative(t)]
            e4[k, m] = [q.calc_point(t), q.calc_first_derivative(t), q.calc_second_derivative(t), q.calc_third_derivative(t)]
    save("g1_poly.npz", quintic_in=q5, quintic_coef=c5, quartic_in=q4, quartic_coef=c4, eval_t=ts, quintic_eval=e5, quartic_eval=e4)


def flensburg():
    """Centerline [203,1397,785] + obstacle table parsed from the reference's data file."""
    root = ET.parse(os.path.join(refshim.REFERENCE_ROOT, "data/demo/DEU_Flensburg-1_1_T-1.xml")).getroot()

    def bound(ll, tag):
        return np.array([[float(p.find("x").text), float(p.find("y").text)] for p in ll.find(tag).findall("point")])

    lanelets = {int(ll.get("id")): ll for ll in root.findall("lanelet")}
    route = [203, 1397, 785]
    succ = [int(s.get("ref")) for s in lanelets[route[-1]].findall("successor")]
    if succ:  # global_planner.py:68-75 appends successor[0] of the last lanelet
        route = route + [succ[0]]
    centers = [(bound(lanelets[i], "leftBound") + bound(lanelets[i], "rightBound")) / 2 for i in route]
    cc = np.concatenate(centers)
    _, uniq = np.unique(cc, return_index=True, axis=0)
    cc = cc[np.sort(uniq)]  # global_planner.py:79-82
    goal_c = centers[2] if len(route) >= 3 else centers[-1]
    goal_center = ((bound(lanelets[785], "leftBound") + bound(lanelets[785], "rightBound")) / 2)
    goal_center = goal_center[int((goal_center.shape[0] - 1) / 2)]  # planning.py:57-58
    obs = []
    T = 0
    for ob in root.findall("dynamicObstacle"):
        rect = ob.find("shape/rectangle")
        l, w = float(rect.find("length").text), float(rect.find("width").text)
        states = {}

        def rd(st):
            t = int(st.find("time/exact").text)
            states[t] = (float(st.find("position/point/x").text), float(st.find("position/point/y").text),
                         float(st.find("orientation/exact").text))

        rd(ob.find("initialState"))
        for st in ob.find("trajectory").findall("state"):
            rd(st)
        obs.append((l, w, states))
        T = max(T, max(states) + 1)
    pose = np.zeros((T, len(obs), 4))
    dims = np.zeros((len(obs), 2))
    for j, (l, w, states) in enumerate(obs):
        dims[j] = [l, w]
        for t, (x, y, yaw) in states.items():
            pose[t, j] = [x, y, yaw, 1.0]
    fts0 = max(obs[0][2])  # obstacles[0].prediction.final_time_step
    init = root.find("planningProblem/initialState")
    init_state = np.array([float(init.find("position/point/x").text), float(init.find("position/point/y").text),
                           float(init.find("orientation/exact").text), float(init.find("velocity/exact").text)])
    return SimpleNamespace(route=np.array(route), centerline=cc, pose=pose, dims=dims, final_time_step=fts0,
                           init_state=init_state, goal_center=goal_center)


def g2():
    fl = flensburg()
    x = np.linspace(0, 400, 81)
    sets = {"flens": fl.centerline, "sinus": np.column_stack([x, 5.5 * np.sin(x / 47.0)])}
    out = {}
    rng = np.random.default_rng(22)
    for name, pts in sets.items():
        sp = R.CubicSpline2D(pts[:, 0], pts[:, 1])
        n = len(pts)
        knots = np.array(sp.s, dtype=np.float64)
        coef = np.zeros((8, n))
        for r, one in ((0, sp.sx), (4, sp.sy)):
            coef[r] = one.a; coef[r + 1, : n - 1] = one.b; coef[r + 2] = one.c; coef[r + 3, : n - 1] = one.d
        s_eval = np.concatenate([rng.uniform(-5, knots[-1] + 5, 180), knots[:10], knots[1:11] - 1e-9, [knots[-1] - 1e-12, -1e-9]])
        ev = np.full((len(s_eval), 4), np.nan)
        for k, s in enumerate(s_eval):
            s = np.float64(s)
            if s == knots[-1]:
                continue
            px, py = sp.calc_position(s)
            if px is None:
                continue
            ev[k] = [px, py, sp.calc_yaw(s), sp.calc_curvature(s)]
        veh = R.Vehicle(refshim.vw_vanagon_params())
        pl = R.FrenetOptimalPlanner(R.FrenetOptimalPlannerSettings(), veh, None)
        _, ref = pl.generate_frenet_frame(pts)
        out.update({f"{name}_pts": pts, f"{name}_knots": knots, f"{name}_coef": coef, f"{name}_s_eval": s_eval,
                    f"{name}_eval": ev, f"{name}_refline": ref})
    save("g2_spline.npz", **out)


def g3():
    out = {}
    names = []
    for name, b in fop_cases():
        t0 = time.time()
        names.append(name)
        C = b.C
        cost = np.empty((b.B, C)); N = np.empty((b.B, C), dtype=np.int32); M = np.empty((b.B, C), dtype=np.int32)
        speed = np.zeros((b.B, C), dtype=bool); accel = np.zeros((b.B, C), dtype=bool); coll = np.zeros((b.B, C), dtype=bool)
        dumps = np.full((b.B, 3, 16, 128), np.nan)
        dump_idx = np.zeros((b.B, 3), dtype=np.int32)
        for e in range(b.B):
            pl = make_planner("FOP", b, e)
            pl.settings.highest_speed = float(b.target_speed[e])
            obstacles = obstacles_for(b, e)
            fpl = pl.calc_global_paths(pl.calc_frenet_paths(ego_state(b, e)))
            for i, fp in enumerate(fpl):
                cost[e, i] = fp.cost_final; N[e, i] = len(fp.t); M[e, i] = len(fp.x)
                speed[e, i] = any(v > pl.vehicle.max_speed for v in fp.s_d)
                accel[e, i] = any(abs(a) > pl.vehicle.max_accel for a in fp.s_dd)
                coll[e, i] = pl.has_collision(fp, obstacles, int(b.t_now[e]), 2)[0]
            trunc = [i for i in range(C) if M[e, i] < N[e, i]]
            pick = [0, C // 2 + 1, trunc[len(trunc) // 2] if trunc else C - 1]
            for k, i in enumerate(pick):
                dump_idx[e, k] = i
                dumps[e, k] = dump_traj(fpl[i])
        out.update(batch_to_dict(b, f"{name}_in_"))
        out.update({f"{name}_cost": cost, f"{name}_N": N, f"{name}_M": M, f"{name}_speed": speed, f"{name}_accel": accel,
                    f"{name}_coll": coll, f"{name}_dumps": dumps, f"{name}_dump_idx": dump_idx})
        print(f"  g3 {name}: {time.time() - t0:.1f}s  coll={coll.mean():.2f} trunc={(M < N).mean():.2f} "
              f"speed={speed.mean():.2f} accel={accel.mean():.2f}")
    out["names"] = np.array(names)
    save("g3_fop_tables.npz", **out)


def run_plan(kind, b, e, prev_best_idx=None, trace=None):
    pl = make_planner(kind, b, e)
    if prev_best_idx is not None:
        pl.prev_best_idx = np.array(prev_best_idx)
    if trace is not None and kind == "FISS+":
        orig = pl.generate_trajectory_by_end_state

        def hooked(end_state):
            J = orig(end_state)
            trace.append([end_state.d, end_state.s_d, end_state.t, J])
            return J

        pl.generate_trajectory_by_end_state = hooked
    try:
        best = pl.plan(ego_state(b, e), float(b.target_speed[e]), obstacles_for(b, e), int(b.t_now[e]))
        err = ""
    except ValueError as ex:  # FISS/FISS+ exact-tie crash ("truth value of an array ...")
        best, err = None, str(ex)[:60]
    return pl, best, err


def g4():
    out = {}
    names = []
    for name, b in fop_cases():
        for kind in ("FOP", "FOP+", "FISS", "FISS+"):
            bb = b
            if kind in ("FISS", "FISS+"):
                if name == "mirror":
                    continue  # the reference raises ValueError on exact ties
                # FISS samples d over max_road_width - w + 0.3 (fiss_planner.py:40)
                sw = 3.5 - b.veh_w + 0.3
                d, rd = np.linspace(-sw / 2, sw / 2, b.nd, retstep=True)
                smin = b.samp_min.copy(); smax = b.samp_max.copy(); sres = b.samp_res.copy()
                smin[:, 0] = -sw / 2; smax[:, 0] = sw / 2; sres[:, 0] = rd
                bb = with_overrides(b, d_samples=d, samp_min=smin, samp_max=smax, samp_res=sres)
            t0 = time.time()
            key = f"{name}_{kind}"
            names.append(key)
            B = bb.B
            idx = np.full((B, 3), -1, dtype=np.int32); flat = np.full(B, -1, dtype=np.int32)
            cost = np.full(B, np.nan); stats = np.zeros((B, 4), dtype=np.int32); end = np.full((B, 3), np.nan)
            found = np.zeros(B, dtype=bool); win = np.full((B, 16, 128), np.nan); NM = np.zeros((B, 2), dtype=np.int32)
            traces = np.full((B, 21, 4), np.nan); prev_out = np.full((B, 3), -1, dtype=np.int32)
            for e in range(B):
                tr = []
                pl, best, err = run_plan(kind, bb, e, trace=tr)
                assert not err, (key, e, err)
                stats[e] = [pl.stats.num_iter, pl.stats.num_trajs_generated, pl.stats.num_trajs_validated, pl.stats.num_collison_checks]
                if tr:
                    traces[e, : len(tr)] = np.array(tr)
                if getattr(pl, "prev_best_idx", None) is not None:
                    prev_out[e] = pl.prev_best_idx
                if best is None:
                    continue
                found[e] = True
                cost[e] = best.cost_final
                NM[e] = [len(best.t), len(best.x)]
                win[e] = dump_traj(best)
                if kind in ("FOP", "FOP+"):
                    pl2 = make_planner("FOP", bb, e)
                    pl2.settings.highest_speed = float(bb.target_speed[e])
                    fpl = pl2.calc_frenet_paths(ego_state(bb, e))
                    hits = [i for i, fp in enumerate(fpl) if np.array_equal(fp.d, best.d) and np.array_equal(fp.s, best.s)]
                    if len(hits) == 1:
                        flat[e] = hits[0]
                    else:  # mirror ties have distinct d arrays, so this cannot happen
                        raise AssertionError((key, e, hits))
                else:
                    idx[e] = best.idx
                    es = best.end_state
                    end[e] = [es.d, es.s_d, es.t]
            out.update(batch_to_dict(bb, f"{key}_in_"))
            out.update({f"{key}_flat": flat, f"{key}_idx": idx, f"{key}_cost": cost, f"{key}_stats": stats, f"{key}_end": end,
                        f"{key}_found": found, f"{key}_win": win, f"{key}_NM": NM, f"{key}_trace": traces,
                        f"{key}_prev_out": prev_out})
            print(f"  g4 {key}: {time.time() - t0:.1f}s found={found.tolist()} stats={stats.tolist()}")
    out["names"] = np.array(names)
    save("g4_plan.npz", **out)


def closed_loop(kind, fl, nd=5, nv=5, nt=5, max_cycles=100):
    """planners/benchmark/planning.py:101-162 with duck-typed obstacles."""
    veh = R.Vehicle(refshim.vw_vanagon_params())
    if kind == "FOP":
        pl = R.FrenetOptimalPlanner(R.FrenetOptimalPlannerSettings(nd, nv, nt), veh, None)
    elif kind == "FOP+":
        pl = R.FopPlusPlanner(R.FrenetOptimalPlannerSettings(nd, nv, nt), veh, None)
    elif kind == "FISS":
        pl = R.FissPlanner(R.FissPlannerSettings(nd, nv, nt), veh, None)
    else:
        st = R.FissPlusPlannerSettings(nd, nv, nt)
        st.time_limit = 1e9
        pl = R.FissPlusPlanner(st, veh, None)
    _, ref = pl.generate_frenet_frame(fl.centerline)
    start = R.State(t=0.0, x=fl.init_state[0], y=fl.init_state[1], yaw=fl.init_state[2], v=fl.init_state[3], a=0.0)
    cur = R.FrenetState()
    cur.from_state(start, ref)
    obstacles = refshim.obstacles_from_table(fl.pose, fl.dims, fl.final_time_step)
    rows = []
    states = []
    for i in range(min(fl.final_time_step, max_cycles)):
        start_vec = [cur.s, cur.s_d, cur.s_dd, cur.d, cur.d_d, cur.d_dd]
        best = pl.plan(cur, getattr(fl, "max_speed", 13.5), obstacles, i)
        st = pl.stats
        if best is None:
            rows.append(start_vec + [np.nan, 0, 0, -1, -1, -1, st.num_iter, st.num_trajs_generated, st.num_trajs_validated,
                                     st.num_collison_checks, np.nan, np.nan, np.nan])
            break
        cs = best.state_at_time_step(1)
        cur = best.frenet_state_at_time_step(1)
        es = best.end_state
        rows.append(start_vec + [best.cost_final, len(best.t), len(best.x), *[int(v) for v in best.idx], st.num_iter,
                                 st.num_trajs_generated, st.num_trajs_validated, st.num_collison_checks,
                                 *( [es.d, es.s_d, es.t] if es is not None else [np.nan] * 3)])
        states.append([cs.x, cs.y, cs.yaw])
        if np.hypot(cs.x - fl.goal_center[0], cs.y - fl.goal_center[1]) <= veh.l / 2:
            break
        if np.hypot(cs.x - ref[-1, 0], cs.y - ref[-1, 1]) <= 3.0:
            break
    return np.array(rows, dtype=np.float64), np.array(states), ref


def g5(kinds=("FOP+", "FISS", "FISS+", "FOP")):
    fl = flensburg()
    path = os.path.join(HERE, "g5_closed_loop.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    out.update(centerline=fl.centerline, obs_pose=fl.pose, obs_dims=fl.dims, final_time_step=np.array(fl.final_time_step),
               init_state=fl.init_state, goal_center=fl.goal_center, route=fl.route,
               columns=np.array(["s", "s_d", "s_dd", "d", "d_d", "d_dd", "cost", "N", "M", "i_d", "i_v", "i_t", "num_iter",
                                 "generated", "validated", "collision_checks", "end_d", "end_v", "end_T"]))
    for kind in kinds:
        t0 = time.time()
        rows, states, ref = closed_loop(kind, fl)
        out[f"{kind}_rows"] = rows
        out[f"{kind}_states"] = states
        out["refline"] = ref
        print(f"  g5 {kind}: {len(rows)} cycles in {time.time() - t0:.1f}s  cost0={rows[0, 6]:.6f}")
        save("g5_closed_loop.npz", **out)


def g7():
    fl = flensburg()
    veh = R.Vehicle(refshim.vw_vanagon_params())
    pl = R.FrenetOptimalPlanner(R.FrenetOptimalPlannerSettings(), veh, None)
    sp, ref = pl.generate_frenet_frame(fl.centerline)
    rng = np.random.default_rng(77)
    K = 24
    s = rng.uniform(1, sp.s[-1] - 1, K)
    poses = np.empty((K, 4)); outv = np.empty((K, 6))
    for k in range(K):
        px, py = sp.calc_position(s[k]); yaw = sp.calc_yaw(s[k])
        off = rng.uniform(-3, 3)
        poses[k] = [px - off * np.sin(yaw), py + off * np.cos(yaw), yaw + rng.uniform(-0.6, 0.6), rng.uniform(0, 15)]
    poses[0] = fl.init_state
    poses[1, :2] = ref[0, :2] + [-2.0, 0.3]      # behind the first waypoint
    poses[2, :2] = ref[-1, :2] + [1.0, -0.5]     # past the last waypoint
    for k in range(K):
        fs = R.FrenetState()
        fs.from_state(R.State(t=0.0, x=poses[k, 0], y=poses[k, 1], yaw=poses[k, 2], v=poses[k, 3]), ref)
        outv[k] = [fs.s, fs.s_d, fs.s_dd, fs.d, fs.d_d, fs.d_dd]
    save("g7_from_state.npz", refline=ref, poses=poses, frenet=outv)


def g8():
    b = synth.make_batch(2, 5, 6, 4, 0, 0, False, 9108, kind="FISS")
    ests = []
    prevs = [None, (0, 0, 0), (4, 5, 3), (2, 1, 3)]
    for e in range(b.B):
        for prev in prevs:
            pl = make_planner("FISS", b, e)
            pl.settings.highest_speed = float(b.target_speed[e])
            pl.prev_best_idx = None if prev is None else np.array(prev)
            t3 = pl.sample_end_frenet_states()
            ests.append([[[tr.cost_est for tr in row] for row in plane] for plane in t3])
    out = batch_to_dict(b)
    save("g8_cost_est.npz", est=np.array(ests).reshape(b.B, len(prevs), b.nd, b.nv, b.nt),
         prev=np.array([(-1, -1, -1) if p is None else p for p in prevs], dtype=np.int32), **out)


def g6():
    """FISS/FISS+ with history heuristic (prev_best_idx) and refinement traces on dynamic scenes."""
    b0 = synth.make_batch(6, 9, 9, 7, 50, 50, True, 9109, kind="FISS+")
    prevs = [None, (4, 8, 6), (0, 0, 0), (8, 4, 3), None, (2, 7, 5)]
    out = batch_to_dict(b0)
    for kind in ("FISS", "FISS+"):
        B = b0.B
        idx = np.full((B, 3), -1, dtype=np.int32); cost = np.full(B, np.nan); stats = np.zeros((B, 4), dtype=np.int32)
        end = np.full((B, 3), np.nan); traces = np.full((B, 21, 4), np.nan); found = np.zeros(B, dtype=bool)
        prev_out = np.full((B, 3), -1, dtype=np.int32)
        for e in range(B):
            tr = []
            pl, best, err = run_plan(kind, b0, e, prev_best_idx=prevs[e], trace=tr)
            assert not err, err
            stats[e] = [pl.stats.num_iter, pl.stats.num_trajs_generated, pl.stats.num_trajs_validated, pl.stats.num_collison_checks]
            if tr:
                traces[e, : len(tr)] = np.array(tr)
            if pl.prev_best_idx is not None:
                prev_out[e] = pl.prev_best_idx
            if best is not None:
                found[e] = True; idx[e] = best.idx; cost[e] = best.cost_final
                end[e] = [best.end_state.d, best.end_state.s_d, best.end_state.t]
        out.update({f"{kind}_idx": idx, f"{kind}_cost": cost, f"{kind}_stats": stats, f"{kind}_end": end, f"{kind}_trace": traces,
                    f"{kind}_found": found, f"{kind}_prev_out": prev_out})
        print(f"  g6 {kind}: stats={stats.tolist()} idx={idx.tolist()}")
    out["prev_in"] = np.array([(-1, -1, -1) if p is None else p for p in prevs], dtype=np.int32)
    save("g6_fiss_search.npz", **out)


def curvature_checked_classes():
    """The four reference planner classes with the curvature checks of check_constraints SWITCHED ON.

    The reference carries the three checks commented out (frenet_optimal_planner.py:145-150).  Nothing of its text is kept here:
    the method's source is read from the imported class at generation time, the comment markers of those six lines are removed,
    and the result is compiled in the module's own namespace and installed on subclasses."""
    import inspect
    import textwrap

    fn = R.FrenetOptimalPlanner.check_constraints
    lines = textwrap.dedent(inspect.getsource(fn)).splitlines()
    out, armed, n = [], False, 0
    for ln in lines:
        body = ln.lstrip()
        if body.startswith("# if any([abs(c"):
            ln = ln.replace("# ", "", 1); armed = True; n += 1
        elif armed and body.startswith("#     continue"):
            ln = ln.replace("# ", "", 1); armed = False; n += 1
        else:
            armed = False
        out.append(ln)
    assert n == 6, f"expected the 6 commented curvature lines, patched {n}"
    ns = dict(fn.__globals__)
    exec(compile("\n".join(out), "<check_constraints with curvature checks>", "exec"), ns)
    patched = ns["check_constraints"]
    return {k: type(c.__name__ + "Curv", (c,), {"check_constraints": patched})
            for k, c in (("FOP", R.FrenetOptimalPlanner), ("FOP+", R.FopPlusPlanner), ("FISS", R.FissPlanner), ("FISS+", R.FissPlusPlanner))}


def g9():
    """Optional curvature / curvature-rate checks (north_star "curvature ... feasibility masks"; reference :145-150 un-commented).
    Per candidate: which of the three checks the patched check_constraints trips (each isolated by lifting the other two limits),
    and that speed / accel / the three together reproduce the patched method's verdict; plan() of the four patched planners."""
    classes = curvature_checked_classes()
    cases = []
    cases.append(("c2_static10", synth.make_batch(3, 5, 5, 5, 10, 100, False, 9102)))
    cases.append(("c3_moving50", synth.make_batch(2, 9, 9, 7, 50, 50, True, 9103)))
    ct = short_frame_batch(synth.make_batch(3, 5, 5, 5, 10, 100, False, 9104), 25)
    cases.append(("trunc", ct))
    # a winding road driven fast with tight limits: the checks also bite on ordinary (non-crawling) candidates
    cw = synth.make_batch(3, 5, 5, 5, 0, 0, False, 9111)
    x = np.linspace(0.0, 400.0, 81)
    from fiss_plus_planner_amd.spline import build_frames
    pts = np.stack([np.stack([x, a * np.sin(x / lam)], axis=1) for a, lam in ((6.0, 14.0), (9.0, 18.0), (4.0, 9.0))])
    knots, coef = build_frames(pts)
    cases.append(("winding", with_overrides(cw, knots=knots, coef=coef)))
    out, names = {}, []
    for name, b in cases:
        t0 = time.time()
        names.append(name)
        C = b.C
        veh = vehicle_for(b)
        limits = (veh.max_curvature, veh.max_kappa_d, veh.max_kappa_dd) if name != "winding" else (0.12, 0.15, 2.0)
        curv = np.zeros((b.B, C, 3), dtype=bool); passed_all = np.zeros((b.B, C), dtype=bool)
        speed = np.zeros((b.B, C), dtype=bool); accel = np.zeros((b.B, C), dtype=bool)
        maxabs = np.full((b.B, C, 3), np.nan)
        for e in range(b.B):
            pl = make_planner("FOP", b, e)
            pl.__class__ = classes["FOP"]
            pl.vehicle.max_curvature, pl.vehicle.max_kappa_d, pl.vehicle.max_kappa_dd = limits
            pl.settings.highest_speed = float(b.target_speed[e])
            fpl = pl.calc_global_paths(pl.calc_frenet_paths(ego_state(b, e)))
            ok = {id(t) for t in pl.check_constraints(fpl)}
            passed_all[e] = [id(t) in ok for t in fpl]
            speed[e] = [any(v > pl.vehicle.max_speed for v in t.s_d) for t in fpl]
            accel[e] = [any(abs(a) > pl.vehicle.max_accel for a in t.s_dd) for t in fpl]
            v_max, a_max = pl.vehicle.max_speed, pl.vehicle.max_accel
            pl.vehicle.max_speed = pl.vehicle.max_accel = np.inf
            for k in range(3):  # isolate check k: the other two limits lifted
                lim = [np.inf] * 3
                lim[k] = limits[k]
                pl.vehicle.max_curvature, pl.vehicle.max_kappa_d, pl.vehicle.max_kappa_dd = lim
                ok_k = {id(t) for t in pl.check_constraints(fpl)}
                curv[e, :, k] = [id(t) not in ok_k for t in fpl]
            pl.vehicle.max_speed, pl.vehicle.max_accel = v_max, a_max
            for i, t in enumerate(fpl):
                for k, arr in enumerate((t.c, t.c_d, t.c_dd)):
                    a = np.abs(np.asarray(arr, dtype=np.float64))
                    a = a[~np.isnan(a)]
                    if a.size:
                        maxabs[e, i, k] = a.max()
            assert np.array_equal(passed_all[e], ~(speed[e] | accel[e] | curv[e].any(axis=1))), (name, e)
        bb = with_overrides(b)
        out.update(batch_to_dict(bb, f"{name}_in_"))
        out.update({f"{name}_limits": np.array(limits), f"{name}_curv": curv, f"{name}_speed": speed, f"{name}_accel": accel,
                    f"{name}_maxabs": maxabs})
        # plan() of the four patched planners
        for kind in ("FOP", "FOP+", "FISS", "FISS+"):
            bk = b
            if kind in ("FISS", "FISS+"):
                sw = 3.5 - b.veh_w + 0.3
                d, rd = np.linspace(-sw / 2, sw / 2, b.nd, retstep=True)
                smin = b.samp_min.copy(); smax = b.samp_max.copy(); sres = b.samp_res.copy()
                smin[:, 0] = -sw / 2; smax[:, 0] = sw / 2; sres[:, 0] = rd
                bk = with_overrides(b, d_samples=d, samp_min=smin, samp_max=smax, samp_res=sres)
                out[f"{name}_fiss_d_samples"] = d
                out[f"{name}_fiss_samp"] = np.stack([smin, smax, sres])
            B = bk.B
            cost = np.full(B, np.nan); stats = np.zeros((B, 4), dtype=np.int32); found = np.zeros(B, dtype=bool)
            flat = np.full(B, -1, dtype=np.int32); idx = np.full((B, 3), -1, dtype=np.int32); end = np.full((B, 3), np.nan)
            for e in range(B):
                pl = make_planner(kind, bk, e)
                pl.__class__ = classes[kind]
                pl.vehicle.max_curvature, pl.vehicle.max_kappa_d, pl.vehicle.max_kappa_dd = limits
                try:
                    best = pl.plan(ego_state(bk, e), float(bk.target_speed[e]), obstacles_for(bk, e), int(bk.t_now[e]))
                except ValueError as ex:
                    raise AssertionError((name, kind, e, str(ex)[:80]))
                stats[e] = [pl.stats.num_iter, pl.stats.num_trajs_generated, pl.stats.num_trajs_validated, pl.stats.num_collison_checks]
                if best is None:
                    continue
                found[e] = True; cost[e] = best.cost_final
                if kind in ("FOP", "FOP+"):
                    pl2 = make_planner("FOP", bk, e)
                    pl2.settings.highest_speed = float(bk.target_speed[e])
                    fpl = pl2.calc_frenet_paths(ego_state(bk, e))
                    hits = [i for i, fp in enumerate(fpl) if np.array_equal(fp.d, best.d) and np.array_equal(fp.s, best.s)]
                    assert len(hits) == 1
                    flat[e] = hits[0]
                else:
                    idx[e] = best.idx
                    end[e] = [best.end_state.d, best.end_state.s_d, best.end_state.t]
            out.update({f"{name}_{kind}_cost": cost, f"{name}_{kind}_stats": stats, f"{name}_{kind}_found": found, f"{name}_{kind}_flat": flat,
                        f"{name}_{kind}_idx": idx, f"{name}_{kind}_end": end})
            print(f"    {name} {kind}: found={found.tolist()} stats={stats.tolist()}")
        print(f"  g9 {name}: {time.time() - t0:.1f}s  curv={curv.mean(axis=(0, 1)).round(3).tolist()} passed={passed_all.mean():.2f}")
    out["names"] = np.array(names)
    save("g9_curvature.npz", **out)


def parse_demo(path):
    """One of the reference's demo scenario files (data/demo/*.xml) -> the arrays planning.py:36-100 hands to the planner.
    Written independently of fiss_plus_planner_amd/commonroad_xml.py (which tests compare against this): exhaustive enumeration of
    the successor-only lanelet paths from every lanelet under the initial position to the goal lanelet; exactly ONE exists in each of
    the five files, which is what stands in for commonroad-route-planner's choice (not installable here; route parity unpinned)."""
    root = ET.parse(path).getroot()

    def bound(ll, tag):
        return np.array([[float(p.find("x").text), float(p.find("y").text)] for p in ll.find(tag).findall("point")])

    lanelets = {int(ll.get("id")): ll for ll in root.findall("lanelet")}
    succ = {i: [int(x.get("ref")) for x in ll.findall("successor")] for i, ll in lanelets.items()}
    pp = root.find("planningProblem")
    init = pp.find("initialState")
    init_state = np.array([float(init.find("position/point/x").text), float(init.find("position/point/y").text),
                           float(init.find("orientation/exact").text), float(init.find("velocity/exact").text)])
    goal = pp.find("goalState")
    goal_id = int(goal.find("position/lanelet").get("ref"))
    gv = goal.find("velocity")
    max_speed = float(gv.find("intervalEnd").text) if gv is not None else 13.5  # planning.py:44-52

    def winding(poly, q):  # non-zero winding number = inside
        a = poly - q
        b = np.roll(a, -1, axis=0)
        ang = np.arctan2(a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0], (a * b).sum(axis=1))
        return abs(ang.sum()) > np.pi

    starts = [i for i, ll in lanelets.items() if winding(np.vstack([bound(ll, "leftBound"), bound(ll, "rightBound")[::-1]]), init_state[:2])]
    paths = []

    def walk(path):
        if path[-1] == goal_id:
            paths.append(list(path)); return
        for v in succ[path[-1]]:
            if v in lanelets and v not in path:
                walk(path + [v])

    for st in starts:
        walk([st])
    assert len(paths) == 1, (path, starts, paths)
    route = paths[0]
    lanes = route + ([succ[route[-1]][0]] if succ[route[-1]] else [])  # global_planner.py:68-75
    centers = {i: (bound(lanelets[i], "leftBound") + bound(lanelets[i], "rightBound")) / 2 for i in set(lanes) | {goal_id}}
    cc = np.concatenate([centers[i] for i in lanes])
    _, uniq = np.unique(cc, return_index=True, axis=0)
    cc = cc[np.sort(uniq)]  # global_planner.py:79-82
    gcv = centers[goal_id]
    goal_center = gcv[int((gcv.shape[0] - 1) / 2)]  # planning.py:57-58
    obs, T = [], 0
    for ob in root.findall("dynamicObstacle"):
        rect = ob.find("shape/rectangle")
        states = {}
        for st in [ob.find("initialState")] + ob.find("trajectory").findall("state"):
            states[int(st.find("time/exact").text)] = (float(st.find("position/point/x").text), float(st.find("position/point/y").text),
                                                       float(st.find("orientation/exact").text))
        obs.append((float(rect.find("length").text), float(rect.find("width").text), states))
        T = max(T, max(states) + 1)
    pose = np.zeros((T, len(obs), 4)); dims = np.zeros((len(obs), 2))
    for j, (l, w, states) in enumerate(obs):
        dims[j] = [l, w]
        for t, (x, y, yaw) in states.items():
            pose[t, j] = [x, y, yaw, 1.0]
    return SimpleNamespace(route=np.array(route), centerline=cc, pose=pose, dims=dims, final_time_step=max(obs[0][2]), init_state=init_state,
                           goal_center=goal_center, max_speed=max_speed, benchmark_id=root.get("benchmarkID"))


def g11(kinds=("FOP+", "FISS", "FISS+")):
    """The reference's five demo scenarios (cfgs/demo_config.yaml: INPUT_DIR data/demo/, 5x5x5 samples): inputs as arrays + the closed
    loop of the reference planners on them (planning.py:101-162), G5-style rows.  FOP itself is pinned on Flensburg-1 by G5 (its
    exhaustive per-cycle validation takes minutes per scenario on the polygon stand-in)."""
    import glob

    out, names = {}, []
    for path in sorted(glob.glob(os.path.join(refshim.REFERENCE_ROOT, "data/demo/*.xml"))):
        sc = parse_demo(path)
        name = sc.benchmark_id
        names.append(name)
        fts = sc.final_time_step
        out.update({f"{name}_centerline": sc.centerline, f"{name}_obs_pose": sc.pose[:max(fts, 1)], f"{name}_obs_dims": sc.dims,
                    f"{name}_final_time_step": np.array(fts), f"{name}_init_state": sc.init_state, f"{name}_goal_center": sc.goal_center,
                    f"{name}_route": sc.route, f"{name}_max_speed": np.array(sc.max_speed)})
        for kind in kinds:
            t0 = time.time()
            rows, states, ref = closed_loop(kind, sc)
            out[f"{name}_{kind}_rows"] = rows
            out[f"{name}_{kind}_states"] = states
            print(f"  g11 {name} {kind}: {len(rows)} cycles in {time.time() - t0:.1f}s, last cost {rows[-1][6]:.6f}")
    out["names"] = np.array(names)
    save("g11_demo_scenarios.npz", **out)


def g10():
    """FISS / FISS+ visualisation payload: the lattice indices the reference GENERATED during plan(), in generation order
    (trajs_per_timestep -> all_trajs, fiss_planner.py:131,262-265 / fiss_plus_planner.py:166-168)."""
    out, names = {}, []
    for name, b in fop_cases():
        if name in ("mirror", "limits", "c0_noobs"):
            continue
        sw = 3.5 - b.veh_w + 0.3
        d, rd = np.linspace(-sw / 2, sw / 2, b.nd, retstep=True)
        smin = b.samp_min.copy(); smax = b.samp_max.copy(); sres = b.samp_res.copy()
        smin[:, 0] = -sw / 2; smax[:, 0] = sw / 2; sres[:, 0] = rd
        bb = with_overrides(b, d_samples=d, samp_min=smin, samp_max=smax, samp_res=sres)
        names.append(name)
        out.update(batch_to_dict(bb, f"{name}_in_"))
        for kind in ("FISS", "FISS+"):
            order = np.full((bb.B, bb.C, 3), -1, dtype=np.int32); count = np.zeros(bb.B, dtype=np.int32)
            cost = np.full((bb.B, bb.C), np.nan)
            for e in range(bb.B):
                pl, best, err = run_plan(kind, bb, e)
                assert not err, err
                gen = pl.all_trajs[-1]
                count[e] = len(gen)
                for k, t in enumerate(gen):
                    order[e, k] = t.idx; cost[e, k] = t.cost_final
            out.update({f"{name}_{kind}_order": order, f"{name}_{kind}_count": count, f"{name}_{kind}_cost": cost})
            print(f"  g10 {name} {kind}: generated {count.tolist()}")
    out["names"] = np.array(names)
    save("g10_generated_order.npz", **out)


# ---------------------------------------------------------------------------
# G12: obstacle shapes that are not rectangles.  has_collision hands obstacle.obstacle_shape.shapely_object - ANY polygon - to
# construct_polygon / intersects (frenet_optimal_planner.py:186-193).  The scenes keep synth's obstacle MOTION and swap the shapes:
def g12_shapes():
    """name -> own-frame geometry (a ring, or a list of rings for a group).  None: the scene's own rectangle (length, width)."""
    ang = -np.arange(64) * (2.0 * np.pi / 64)  # shapely's Point.buffer(r): 64 vertices, clockwise from (r, 0)
    c, sn = np.cos(ang), np.sin(ang)
    c[np.abs(c) < 5e-16] = 0.0; sn[np.abs(sn) < 5e-16] = 0.0
    rot = np.array([[np.cos(0.5), -np.sin(0.5)], [np.sin(0.5), np.cos(0.5)]])
    rect = np.array([(-2.25, -0.95), (2.25, -0.95), (2.25, 0.95), (-2.25, 0.95)])
    return [
        ("circle", 1.6 * np.stack([c, sn], axis=1)),                                   # a commonroad Circle's shapely_object
        ("triangle", np.array([(-2.4, -1.0), (2.4, -1.0), (0.0, 1.2)])),
        ("rotated_rect", rect @ rot.T),                                                # a Rectangle with orientation 0.5
        ("L", np.array([(0, 0), (4.2, 0), (4.2, 1.2), (1.3, 1.2), (1.3, 3.0), (0, 3.0)], dtype=float)),  # non-convex, off-centre
        ("pentagon", np.array([(1.0, 0.0), (3.0, 0.5), (3.5, 2.0), (2.0, 3.0), (0.5, 1.8)])),           # convex, off-centre
        ("rectangle", None),
        ("group", [np.array([(-3.0, -0.5), (-1.0, -0.5), (-1.0, 0.5), (-3.0, 0.5)]), np.array([(1.0, -0.8), (3.0, 0.0), (1.0, 0.8)])]),
    ]


def g12_scene(b: ProblemBatch, seed: int) -> ProblemBatch:
    """synth spreads its obstacles over +-4 m and 120 m: few verdicts would hang on a shape.  Here every obstacle drives slowly in
    or beside the ego's lane, a few car lengths ahead, at a heading a little off the lane's - the ego's candidates brush past their
    corners within the horizon."""
    rng = np.random.default_rng(seed)
    S, T, n = b.obs_pose.shape[:3]
    tt = np.arange(T) * b.tick_t
    pose = np.zeros_like(b.obs_pose)
    for sc in range(S):
        e = int(np.nonzero(b.scene_of == sc)[0][0])
        s0 = b.ego[e, 0] + 14.0 + 11.0 * rng.permutation(n) + rng.uniform(-3, 3, n)
        v = rng.uniform(0.0, 3.5, n)
        side = np.where(rng.uniform(size=n) < 0.5, -1.0, 1.0)
        d = side * rng.uniform(1.9, 3.3, n)
        s_t = (s0[None, :] + v[None, :] * tt[:, None])[None]
        f = int(b.frame_of[e])
        px, py, yaw = synth.sample_frames(b.knots[f:f + 1], b.coef[f:f + 1], s_t)
        px, py, yaw = px[0], py[0], yaw[0]
        pose[sc, :, :, 0] = px - d[None, :] * np.sin(yaw)
        pose[sc, :, :, 1] = py + d[None, :] * np.cos(yaw)
        pose[sc, :, :, 2] = yaw + rng.uniform(-0.6, 0.6, n)[None, :] + 0.02 * tt[:, None] * rng.uniform(-1, 1, n)[None, :]
        pose[sc, :, :, 3] = 1.0
    return with_overrides(b, obs_pose=pose)


def g12_obstacles(b: ProblemBatch, e: int):
    sc = int(b.scene_of[e])
    pose, dims, fts = b.obs_pose[sc], b.obs_dims[sc], int(b.final_time_step[sc])
    out = []
    for j, (_, geom) in enumerate(g12_shapes()):
        p = pose[:, j, :3].copy()
        p[pose[:, j, 3] == 0.0] = np.nan
        if geom is None:
            out.append(refshim.StubObstacle(dims[j, 0], dims[j, 1], p, fts))
        elif isinstance(geom, list):
            out.append(refshim.StubShapeObstacle(refshim.MultiPolygon(geom), p, fts))
        else:
            out.append(refshim.StubShapeObstacle(refshim.Polygon(geom), p, fts))
    return out


def g12():
    """Per case: the raw inputs (synth batch = motion + the rectangle's sizes, the shape rings), the reference's per-candidate
    has_collision verdicts and the four planners' answers with those shapes, and - for the tests' sanity - the verdicts the same
    scene gives when every shape is replaced by its bounding box (what the build did before ABI 12): they must differ."""
    out = {}
    names = []
    shapes = g12_shapes()
    rings = np.full((len(shapes), 2, 64, 2), np.nan)
    ring_n = np.zeros((len(shapes), 2), dtype=np.int32)
    for j, (_, geom) in enumerate(shapes):
        for k, ring in enumerate([] if geom is None else (geom if isinstance(geom, list) else [geom])):
            rings[j, k, :len(ring)] = ring
            ring_n[j, k] = len(ring)
    out["shape_names"] = np.array([n for n, _ in shapes])
    out["shape_rings"] = rings
    out["shape_ring_n"] = ring_n
    # (scene seeds picked with the oracle so that the shapes decide: verdicts and the selected index differ from the boxed scene's)
    b5, b9 = synth.make_batch(3, 5, 5, 5, 7, 60, True, 9112), synth.make_batch(2, 9, 9, 7, 7, 50, True, 9113)
    cases = [("p555", g12_scene(b5, 3)), ("p555b", g12_scene(b5, 4)), ("p997", g12_scene(b9, 11)), ("p997b", g12_scene(b9, 4))]
    for name, b in cases:
        names.append(name)
        C = b.C
        coll = np.zeros((b.B, C), dtype=bool); coll_box = np.zeros((b.B, C), dtype=bool); cost = np.empty((b.B, C))
        for e in range(b.B):
            pl = make_planner("FOP", b, e)
            pl.settings.highest_speed = float(b.target_speed[e])
            obstacles = g12_obstacles(b, e)
            boxes = []
            for ob in obstacles:  # the bounding-box stand-in of every shape
                so = ob.obstacle_shape.shapely_object
                minx, miny, maxx, maxy = so.bounds
                bx = refshim.StubShapeObstacle(refshim.Polygon([(minx, miny), (maxx, miny), (maxx, maxy), (minx, maxy)]), ob.poses, ob.prediction.final_time_step)
                boxes.append(bx)
            fpl = pl.calc_global_paths(pl.calc_frenet_paths(ego_state(b, e)))
            for i, fp in enumerate(fpl):
                cost[e, i] = fp.cost_final
                coll[e, i] = pl.has_collision(fp, obstacles, int(b.t_now[e]), 2)[0]
                coll_box[e, i] = pl.has_collision(fp, boxes, int(b.t_now[e]), 2)[0]
        out.update(batch_to_dict(b, f"{name}_in_"))
        out.update({f"{name}_coll": coll, f"{name}_coll_box": coll_box, f"{name}_cost": cost})
        print(f"  g12 {name}: coll={coll.mean():.3f} box={coll_box.mean():.3f} differ={int((coll != coll_box).sum())} of {coll.size}")
        for kind in ("FOP", "FOP+", "FISS", "FISS+"):
            bb = b
            if kind in ("FISS", "FISS+"):
                sw = 3.5 - b.veh_w + 0.3
                d, rd = np.linspace(-sw / 2, sw / 2, b.nd, retstep=True)
                smin = b.samp_min.copy(); smax = b.samp_max.copy(); sres = b.samp_res.copy()
                smin[:, 0] = -sw / 2; smax[:, 0] = sw / 2; sres[:, 0] = rd
                bb = with_overrides(b, d_samples=d, samp_min=smin, samp_max=smax, samp_res=sres)
            key = f"{name}_{kind}"
            B = bb.B
            idx = np.full((B, 3), -1, dtype=np.int32); flat = np.full(B, -1, dtype=np.int32)
            pcost = np.full(B, np.nan); stats = np.zeros((B, 4), dtype=np.int32); end = np.full((B, 3), np.nan)
            found = np.zeros(B, dtype=bool)
            for e in range(B):
                pl = make_planner(kind, bb, e)
                best = pl.plan(ego_state(bb, e), float(bb.target_speed[e]), g12_obstacles(bb, e), int(bb.t_now[e]))
                stats[e] = [pl.stats.num_iter, pl.stats.num_trajs_generated, pl.stats.num_trajs_validated, pl.stats.num_collison_checks]
                if best is None:
                    continue
                found[e] = True
                pcost[e] = best.cost_final
                if kind in ("FOP", "FOP+"):
                    pl2 = make_planner("FOP", bb, e)
                    pl2.settings.highest_speed = float(bb.target_speed[e])
                    fpl = pl2.calc_frenet_paths(ego_state(bb, e))
                    hits = [i for i, fp in enumerate(fpl) if np.array_equal(fp.d, best.d) and np.array_equal(fp.s, best.s)]
                    assert len(hits) == 1, (key, e, hits)
                    flat[e] = hits[0]
                else:
                    idx[e] = best.idx
                    es = best.end_state
                    end[e] = [es.d, es.s_d, es.t]
            if kind in ("FISS", "FISS+"):
                out.update({f"{key}_in_d_samples": bb.d_samples, f"{key}_in_samp_min": bb.samp_min, f"{key}_in_samp_max": bb.samp_max, f"{key}_in_samp_res": bb.samp_res})
            out.update({f"{key}_flat": flat, f"{key}_idx": idx, f"{key}_cost": pcost, f"{key}_stats": stats, f"{key}_end": end, f"{key}_found": found})
            print(f"  g12 {key}: found={found.tolist()} stats={stats.tolist()}")
    out["names"] = np.array(names)
    save("g12_shapes.npz", **out)


def g13():
    """tick_t = 0.05 (T = 8 .. 10 s -> 160 .. 200 points per trajectory: beyond round 4's FP_MAX_POINTS = 128): the reference's own
    tables, series and plans.  Two scenes: 8 moving obstacles over 220 time steps (has_collision's pose k is the obstacle's time
    step k whatever the planner's tick, :173-176), and a short reference line (truncated series)."""
    STRIDE = 208
    b0 = synth.make_batch(2, 5, 4, 3, 8, 220, True, 9113)
    b0.tick_t = 0.05
    b0 = with_overrides(b0)
    bt = short_frame_batch(synth.make_batch(2, 5, 4, 3, 8, 220, True, 9114), 25)
    bt.tick_t = 0.05
    bt = with_overrides(bt)
    out, names = {}, []
    for name, b in (("tick005", b0), ("tick005_short", bt)):
        names.append(name)
        C = b.C
        cost = np.empty((b.B, C)); N = np.empty((b.B, C), dtype=np.int32); M = np.empty((b.B, C), dtype=np.int32)
        speed = np.zeros((b.B, C), dtype=bool); accel = np.zeros((b.B, C), dtype=bool); coll = np.zeros((b.B, C), dtype=bool)
        dumps = np.full((b.B, 3, 16, STRIDE), np.nan); dump_idx = np.zeros((b.B, 3), dtype=np.int32)
        for e in range(b.B):
            pl = make_planner("FOP", b, e)
            pl.settings.highest_speed = float(b.target_speed[e])
            obstacles = obstacles_for(b, e)
            fpl = pl.calc_global_paths(pl.calc_frenet_paths(ego_state(b, e)))
            for i, fp in enumerate(fpl):
                cost[e, i] = fp.cost_final; N[e, i] = len(fp.t); M[e, i] = len(fp.x)
                speed[e, i] = any(v > pl.vehicle.max_speed for v in fp.s_d)
                accel[e, i] = any(abs(a) > pl.vehicle.max_accel for a in fp.s_dd)
                coll[e, i] = pl.has_collision(fp, obstacles, int(b.t_now[e]), 2)[0]
            trunc = [i for i in range(C) if 2 <= M[e, i] < N[e, i]]
            long_cut = [i for i in trunc if M[e, i] > 110]
            pick = [0, C // 2 + 1, (long_cut or trunc or [C - 1])[len(long_cut or trunc or [0]) // 2]]
            for k, i in enumerate(pick):
                dump_idx[e, k] = i
                dumps[e, k] = dump_traj(fpl[i], STRIDE)
        out.update(batch_to_dict(b, f"{name}_in_"))
        out.update({f"{name}_cost": cost, f"{name}_N": N, f"{name}_M": M, f"{name}_speed": speed, f"{name}_accel": accel,
                    f"{name}_coll": coll, f"{name}_dumps": dumps, f"{name}_dump_idx": dump_idx})
        print(f"  g13 {name}: N {N.min()}..{N.max()} coll={coll.mean():.2f} trunc={(M < N).mean():.2f} M range {M.min()}..{M.max()}")
        # plan() of the four planners on the same problems
        for kind in ("FOP", "FOP+", "FISS", "FISS+"):
            bb = b
            if kind in ("FISS", "FISS+"):
                sw = 3.5 - b.veh_w + 0.3
                d, rd = np.linspace(-sw / 2, sw / 2, b.nd, retstep=True)
                smin = b.samp_min.copy(); smax = b.samp_max.copy(); sres = b.samp_res.copy()
                smin[:, 0] = -sw / 2; smax[:, 0] = sw / 2; sres[:, 0] = rd
                bb = with_overrides(b, d_samples=d, samp_min=smin, samp_max=smax, samp_res=sres)
            key = f"{name}_{kind}"
            B = bb.B
            idx = np.full((B, 3), -1, dtype=np.int32); pcost = np.full(B, np.nan); stats = np.zeros((B, 4), dtype=np.int32)
            end = np.full((B, 3), np.nan); found = np.zeros(B, dtype=bool); win = np.full((B, 16, STRIDE), np.nan); NM = np.zeros((B, 2), dtype=np.int32)
            for e in range(B):
                pl, best, err = run_plan(kind, bb, e)
                assert not err, (key, e, err)
                stats[e] = [pl.stats.num_iter, pl.stats.num_trajs_generated, pl.stats.num_trajs_validated, pl.stats.num_collison_checks]
                if best is None:
                    continue
                found[e] = True
                pcost[e] = best.cost_final
                NM[e] = [len(best.t), len(best.x)]
                win[e] = dump_traj(best, STRIDE)
                if kind in ("FISS", "FISS+"):
                    idx[e] = best.idx
                    end[e] = [best.end_state.d, best.end_state.s_d, best.end_state.t]
            if kind in ("FISS", "FISS+"):
                out.update(batch_to_dict(bb, f"{key}_in_"))
            out.update({f"{key}_idx": idx, f"{key}_cost": pcost, f"{key}_stats": stats, f"{key}_end": end, f"{key}_found": found,
                        f"{key}_win": win, f"{key}_NM": NM})
            print(f"  g13 {key}: found={found.tolist()} stats={stats.tolist()}")
    out["names"] = np.array(names)
    save("g13_tick005.npz", **out)


def g14():
    """FissPlusPlanner's wall-clock `time_limit` (fiss_plus_planner.py:152-158, :293-299), in the two cases the clock cannot change:
    a limit that is already over when the coarse search returns (time_limit = -1).  has_time_limit = False: refine_solution is entered
    with a negative time_left and its loop breaks after the FIRST gradient step (7 refinement trajectories instead of 21);
    has_time_limit = True: no refinement at all, plan() returns the coarse winner."""
    out, names = {}, []
    for name, b in fop_cases():
        if name not in ("c2_static10", "c3_moving50", "tnow"):
            continue
        sw = 3.5 - b.veh_w + 0.3
        d, rd = np.linspace(-sw / 2, sw / 2, b.nd, retstep=True)
        smin = b.samp_min.copy(); smax = b.samp_max.copy(); sres = b.samp_res.copy()
        smin[:, 0] = -sw / 2; smax[:, 0] = sw / 2; sres[:, 0] = rd
        bb = with_overrides(b, d_samples=d, samp_min=smin, samp_max=smax, samp_res=sres)
        for mode, has_limit in (("over_unlimited", False), ("over_limited", True)):
            key = f"{name}_{mode}"
            names.append(key)
            B = bb.B
            idx = np.full((B, 3), -1, dtype=np.int32); cost = np.full(B, np.nan); stats = np.zeros((B, 4), dtype=np.int32)
            end = np.full((B, 3), np.nan); found = np.zeros(B, dtype=bool); n_refined = np.zeros(B, dtype=np.int32)
            for e in range(B):
                pl = make_planner("FISS+", bb, e)
                pl.settings.time_limit = -1.0
                pl.settings.has_time_limit = has_limit
                tr = []
                orig = pl.generate_trajectory_by_end_state

                def hooked(end_state, orig=orig, tr=tr):
                    J = orig(end_state)
                    tr.append(J)
                    return J

                pl.generate_trajectory_by_end_state = hooked
                best = pl.plan(ego_state(bb, e), float(bb.target_speed[e]), obstacles_for(bb, e), int(bb.t_now[e]))
                stats[e] = [pl.stats.num_iter, pl.stats.num_trajs_generated, pl.stats.num_trajs_validated, pl.stats.num_collison_checks]
                n_refined[e] = len(tr)
                if best is None:
                    continue
                found[e] = True
                cost[e] = best.cost_final
                idx[e] = best.idx
                end[e] = [best.end_state.d, best.end_state.s_d, best.end_state.t]
            out.update(batch_to_dict(bb, f"{key}_in_"))
            out.update({f"{key}_idx": idx, f"{key}_cost": cost, f"{key}_stats": stats, f"{key}_end": end, f"{key}_found": found, f"{key}_n_refined": n_refined})
            print(f"  g14 {key}: found={found.tolist()} refinement trajectories={n_refined.tolist()} stats={stats.tolist()}")
    out["names"] = np.array(names)
    save("g14_time_limit.npz", **out)


if __name__ == "__main__":
    todo = sys.argv[1:] or ["g1", "g2", "g7", "g8", "g3", "g4", "g6", "g5", "g9", "g10", "g11", "g12", "g13", "g14"]
    for g in todo:
        t0 = time.time()
        print(f"== {g}")
        if g.startswith("g5:"):
            g5(tuple(g[3:].split(",")))
        else:
            globals()[g]()
        print(f"== {g} done in {time.time() - t0:.1f}s")
        # collision audit of this generator: Polygon.intersects calls, calls near contact (decided in rational arithmetic), and how
        # many of those the plain fp64 separating-axis test would have answered differently
        import json
        with open(os.path.join(OUT_DIR, f"collision_audit_{g.replace(':', '_').replace(',', '_')}.json"), "w") as fh:
            json.dump(dict(refshim.AUDIT, generator=g), fh)
        for k in refshim.AUDIT:
            refshim.AUDIT[k] = 0
