"""Synthetic CommonRoad-like problem batches (BASELINE.json configs 2-5).

Deterministic: numpy Generator(PCG64([seed, ego])).  One sinusoidal centerline and one
obstacle scene per ego.  Obstacles are rectangles that sit on (static) or drive along
(dynamic) the ego's own road.  Two obstacle layouts:
  * "survey8d" - SURVEY.md section 8(d) verbatim, the contract workload and make_config's default: every obstacle at
    s_o = s + U(8, 120), d_o ~ U(-4, 4), speed U(0, 12) when moving.  With 50 obstacles the lane is densely blocked: most
    of the fan collides and ~58 % of the egos keep no feasible candidate.
  * "lanes" - the builder's own layout (make_batch's default, used by the unit tests): ~15 % of the obstacles in the ego lane
    ahead of the ego (lead vehicles, |d| <= 1 m), the rest in the neighbouring lanes on both sides (2.9 m <= |d| <= 7.5 m)
    from 15 m behind to 120 m ahead; ~75 % of the egos keep a feasible candidate.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from .batch import FISS_KINDS, ProblemBatch, speed_samples
from .spline import build_frames
from .vehicle import Vehicle

CONFIG_SEEDS = {2: 2302, 3: 2303, 4: 2304, 5: 2305}


def default_settings(nd=5, nv=5, nt=5):
    """Reference defaults, frenet_optimal_planner.py:38-56."""
    return SimpleNamespace(tick_t=0.1, max_road_width=3.5, num_width=nd, num_speed=nv, num_t=nt, highest_speed=13.4112,
                           lowest_speed=0.0, min_t=8.0, max_t=10.0)


def sample_frames(knots, coef, s):
    """Evaluate frame f at arclength s[f, ...] -> x, y, yaw (vectorised over frames)."""
    F = knots.shape[0]
    shp = s.shape
    s2 = s.reshape(F, -1)
    idx = np.empty(s2.shape, dtype=np.int64)
    for f in range(F):
        idx[f] = np.searchsorted(knots[f], s2[f], side="right") - 1
    nxm2 = (np.isfinite(knots).sum(axis=1) - 2)[:, None]
    idx = np.clip(idx, 0, nxm2)
    fr = np.arange(F)[:, None]
    dx = s2 - knots[fr, idx]
    g = lambda r: coef[fr, r, idx]
    px = g(0) + g(1) * dx + g(2) * dx ** 2 + g(3) * dx ** 3
    py = g(4) + g(5) * dx + g(6) * dx ** 2 + g(7) * dx ** 3
    yaw = np.arctan2(g(5) + 2 * g(6) * dx + 3 * g(7) * dx ** 2, g(1) + 2 * g(2) * dx + 3 * g(3) * dx ** 2)
    return px.reshape(shp), py.reshape(shp), yaw.reshape(shp)


def make_batch(B: int, nd: int, nv: int, nt: int, n_obs: int, T_obs: int, moving: bool, seed: int,
               kind: str = "FOP", vehicle: Vehicle | None = None, max_target_speed: float = 13.5,
               ego_offset: int = 0, layout: str = "lanes", n_knots: int = 81) -> ProblemBatch:
    """B egos, each with its own 81-knot centerline and its own n_obs-rectangle scene.

    ego_offset lets a rank generate only its shard [ego_offset, ego_offset+B) of a larger batch:
    every ego draws from its own child stream SeedSequence(seed).spawn-like key (seed, index).
    layout: "lanes" (module docstring; the bench default) or "survey8d" = SURVEY.md section 8d verbatim: every obstacle
    at s_o = s + U(8, 120), d_o ~ U(-4, 4), speed U(0, 12) when moving.
    """
    assert layout in ("lanes", "survey8d")
    veh = vehicle or Vehicle()
    st = default_settings(nd, nv, nt)
    NX = int(n_knots)  # knots of every ego's reference line (the same 400 m road sampled finer or coarser; 81 = one knot per 5 m)
    pts = np.empty((B, NX, 2))
    ego = np.empty((B, 6))
    dims = np.empty((B, max(n_obs, 0), 2))
    so = np.empty((B, max(n_obs, 0)))
    do = np.empty((B, max(n_obs, 0)))
    vo = np.zeros((B, max(n_obs, 0)))
    xs = np.linspace(0.0, 400.0, NX)
    for b in range(B):
        rng = np.random.Generator(np.random.PCG64([seed, ego_offset + b]))
        A, lam = rng.uniform(0, 8), rng.uniform(30, 80)
        pts[b, :, 0] = xs
        pts[b, :, 1] = A * np.sin(xs / lam)
        d0 = rng.uniform(-0.8, 0.8)
        if abs(d0) < 0.01:
            d0 = 0.01 if d0 >= 0 else -0.01
        ego[b] = [rng.uniform(5, 60), rng.uniform(2, 13), rng.uniform(-1, 1), d0, rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2)]
        if n_obs > 0:
            dims[b, :, 0] = rng.uniform(3.5, 7.5, n_obs)
            dims[b, :, 1] = rng.uniform(1.6, 2.3, n_obs)
            if layout == "survey8d":
                so[b] = ego[b, 0] + rng.uniform(8, 120, n_obs)
                do[b] = rng.uniform(-4, 4, n_obs)
                if moving:
                    vo[b] = rng.uniform(0, 12, n_obs)
                continue
            n_lane = max(1, int(round(0.15 * n_obs)))
            side = rng.uniform(2.9, 7.5, n_obs) * np.where(rng.uniform(size=n_obs) < 0.5, -1.0, 1.0)
            lane = rng.uniform(-1.0, 1.0, n_obs)
            ahead_lane = rng.uniform(12, 90, n_obs)
            ahead_side = rng.uniform(-15, 120, n_obs)
            in_lane = np.arange(n_obs) < n_lane
            do[b] = np.where(in_lane, lane, side)
            so[b] = np.maximum(ego[b, 0] + np.where(in_lane, ahead_lane, ahead_side), 1.0)
            if moving:
                vo[b] = rng.uniform(0, 10, n_obs)
    knots, coef = build_frames(pts)
    if n_obs > 0:
        tt = np.arange(T_obs) * st.tick_t
        s_t = so[:, None, :] + vo[:, None, :] * tt[None, :, None]  # [B, T, n]
        px, py, yaw = sample_frames(knots, coef, s_t)
        pose = np.stack([px - do[:, None, :] * np.sin(yaw), py + do[:, None, :] * np.cos(yaw), yaw, np.ones_like(px)], axis=-1)
        fts = np.full(B, T_obs - 1, dtype=np.int32)
        scene_of = np.arange(B, dtype=np.int32)
    else:
        pose = np.zeros((0, max(T_obs, 1), 0, 4))
        dims = np.zeros((0, 0, 2))
        fts = np.zeros(0, dtype=np.int32)
        scene_of = np.full(B, -1, dtype=np.int32)
    sw = st.max_road_width - veh.w + (0.3 if kind in FISS_KINDS else 0.0)
    d_samples, rd = np.linspace(-sw / 2, sw / 2, nd, retstep=True)
    t_samples, rt = np.linspace(st.min_t, st.max_t, nt, retstep=True)
    vmax = np.full(B, max_target_speed)
    v_samples, rv = speed_samples(st.lowest_speed, vmax, nv)
    samp_min = np.column_stack([np.full(B, -sw / 2), np.full(B, st.lowest_speed), np.full(B, st.min_t)])
    samp_max = np.column_stack([np.full(B, sw / 2), vmax, np.full(B, st.max_t)])
    samp_res = np.column_stack([np.full(B, rd), rv, np.full(B, rt)])
    return ProblemBatch(
        d_samples=d_samples, t_samples=t_samples, v_samples=v_samples, target_speed=vmax, ego=ego,
        frame_of=np.arange(B), scene_of=scene_of, t_now=np.zeros(B), nx=np.full(B, NX), knots=knots, coef=coef,
        obs_pose=pose, obs_dims=dims, final_time_step=fts, veh_l=veh.l, veh_w=veh.w, max_speed=veh.max_speed,
        max_accel=veh.max_accel, tick_t=st.tick_t, check_stride=2, samp_min=samp_min, samp_max=samp_max, samp_res=samp_res,
        meta=dict(seed=seed, kind=kind, moving=moving, ego_offset=ego_offset, layout=layout, **({"n_knots": NX} if NX != 81 else {})))


def with_rectangle_rings(batch: ProblemBatch, seed: int, frac: float = 0.5, canonical: bool = True) -> ProblemBatch:
    """`batch` with a share of its obstacle columns spelled as 4-vertex rings that ARE their rectangles (the same polygons, handed over as
    fp_batch.obs_poly / obs_nvert).  canonical=True: through ProblemBatch's constructor, which turns such rings back into rectangle
    columns; False: behind its back (the polygon code path on rectangle geometry: what the A/B tools and tests measure)."""
    rng = np.random.default_rng(seed)
    S, n = batch.S, batch.n_obs
    hl, hw = 0.5 * batch.obs_dims[..., 0], 0.5 * batch.obs_dims[..., 1]
    poly = np.stack([np.stack([-hl, -hw], -1), np.stack([hl, -hw], -1), np.stack([hl, hw], -1), np.stack([-hl, hw], -1)], axis=2)
    nvert = np.where(rng.uniform(size=(S, n)) < frac, 4, 0).astype(np.int32)
    kw = {k: getattr(batch, k) for k in ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
                                          "obs_pose", "obs_dims", "final_time_step", "veh_l", "veh_w", "max_speed", "max_accel", "tick_t", "check_stride",
                                          "samp_min", "samp_max", "samp_res")}
    if canonical:
        return ProblemBatch(**kw, obs_poly=poly, obs_nvert=nvert, meta=dict(batch.meta, rectangle_rings=seed))
    out = ProblemBatch(**kw, obs_poly=poly, obs_nvert=np.full((S, n), 3, dtype=np.int32), meta=dict(batch.meta, rectangle_rings=seed))
    out.obs_nvert = nvert
    return out


def make_config(config: int, B: int | None = None, ego_offset: int = 0, kind: str | None = None, layout: str = "survey8d") -> ProblemBatch:
    """BASELINE.json configs[config-1] (2..5) on SURVEY.md section 8(d)'s generator (layout="lanes": the builder's own obstacle layout)."""
    if config == 2:
        return make_batch(B or 256, 5, 5, 5, 10, 100, False, CONFIG_SEEDS[2], kind or "FOP", ego_offset=ego_offset, layout=layout)
    if config == 3:
        return make_batch(B or 2048, 9, 9, 7, 50, 50, True, CONFIG_SEEDS[3], kind or "FOP", ego_offset=ego_offset, layout=layout)
    if config == 4:
        return make_batch(B or 2048, 9, 9, 7, 50, 50, True, CONFIG_SEEDS[4], kind or "FISS+", ego_offset=ego_offset, layout=layout)
    if config == 5:
        return make_batch(B or 16384, 9, 9, 7, 50, 50, True, CONFIG_SEEDS[5], kind or "FOP", ego_offset=ego_offset, layout=layout)
    raise ValueError(f"unknown config {config}")


def random_convex_ring(rng, n: int, rx: float, ry: float) -> np.ndarray:
    """n vertices on an ellipse (rx, ry) at sorted random angles: a convex counter-clockwise ring around the origin."""
    while True:
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        if np.max(np.diff(np.concatenate([ang, [ang[0] + 2 * np.pi]]))) < np.pi - 0.2:
            break
    return np.stack([rx * np.cos(ang), ry * np.sin(ang)], axis=1)


def with_random_shapes(batch: ProblemBatch, seed: int, frac: float = 0.6, max_vertices: int = 12) -> ProblemBatch:
    """The same batch with a fraction of its rectangle columns turned into random convex polygons that fit the rectangle they replace
    (fp_batch.obs_poly / obs_nvert: rings centred on their own bounding box - the pose is the rotation centre - and obs_dims = that
    box).  Collisions can only disappear against the rectangle scene; the motion, the lattice and the egos are untouched."""
    rng = np.random.default_rng(seed)
    S, n = batch.S, batch.n_obs
    poly = np.zeros((S, n, max_vertices, 2))
    nvert = np.zeros((S, n), dtype=np.int32)
    dims = batch.obs_dims.copy()
    for sc in range(S):
        for j in range(n):
            if rng.uniform() > frac:
                continue
            k = int(rng.integers(3, max_vertices + 1))
            ring = random_convex_ring(rng, k, 0.5 * dims[sc, j, 0], 0.5 * dims[sc, j, 1])
            lo, hi = ring.min(axis=0), ring.max(axis=0)
            ring = ring - 0.5 * (lo + hi)
            poly[sc, j, :k] = ring
            nvert[sc, j] = k
            dims[sc, j] = 2.0 * np.abs(ring).max(axis=0)
    kw = {k: getattr(batch, k) for k in ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
                                          "obs_pose", "final_time_step", "veh_l", "veh_w", "max_speed", "max_accel", "tick_t", "check_stride", "samp_min", "samp_max",
                                          "samp_res", "curvature_limits")}
    return ProblemBatch(**kw, obs_dims=dims, obs_poly=poly, obs_nvert=nvert, meta=dict(batch.meta, shapes=seed))
