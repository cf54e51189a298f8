"""The "problem batch": host-side (numpy) layout of B independent ego planning problems.

This is exactly what crosses the C ABI (include/frenet_gpu.h): contiguous
little-endian float64 / int32 arrays, one row per ego / frame / scene.

    lattice   d_samples [nd], t_samples [nt]           shared by the batch
              v_samples [B, nv], target_speed [B]      per ego (depend on max_target_speed)
    ego       ego [B, 6] = s, s_d, s_dd, d, d_d, d_dd  start FrenetState (reference frenet.py:15-27)
              frame_of [B], scene_of [B] (-1 = no obstacles), t_now [B]
    frames    nx [F], knots [F, NX] (+inf padded), coef [F, 8, NX] (ax bx cx dx ay by cy dy)
    scenes    obs_pose [S, T_obs, n_obs, 4] = x, y, yaw, valid;  obs_dims [S, n_obs, 2] = length, width
              final_time_step [S]  (= obstacles[0].prediction.final_time_step, frenet_optimal_planner.py:173)
              optional obs_nvert [S, n_obs] + obs_poly [S, n_obs, PV, 2]: convex-polygon columns (0 vertices = the rectangle of
              obs_dims; else a counter-clockwise ring relative to the column's rotation centre, obs_dims = the box that holds it)

The three sample vectors are produced with numpy.linspace on the host, as the
reference does (frenet_optimal_planner.py:75,78,89; fiss_planner.py:47,59,69), so the
kernels never re-derive them and stay bit-identical on the grid values.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field

import numpy as np

FOP_KINDS = ("FOP", "FOP+")
FISS_KINDS = ("FISS", "FISS+")


def lattice_samples(settings, vehicle_w: float, kind: str = "FOP"):
    """(d_samples, t_samples, samp_res_d, samp_res_t) for a planner family.

    FOP/FOP+ sample d over ``max_road_width - w`` (frenet_optimal_planner.py:72); FISS/FISS+
    over ``max_road_width - w + 0.3`` (fiss_planner.py:40).
    """
    sw = settings.max_road_width - vehicle_w + (0.3 if kind in FISS_KINDS else 0.0)
    d, rd = np.linspace(-sw / 2, sw / 2, settings.num_width, retstep=True)
    t, rt = np.linspace(settings.min_t, settings.max_t, settings.num_t, retstep=True)
    return d, t, rd, rt


def speed_samples(lowest: float, highest, num_speed: int):
    """np.linspace(lowest, highest, nv) per ego -> ([B, nv], res [B])."""
    highest = np.atleast_1d(np.asarray(highest, dtype=np.float64))
    v = np.empty((highest.size, num_speed))
    res = np.empty(highest.size)
    for b, hi in enumerate(highest):  # linspace per ego keeps numpy's exact rounding
        v[b], res[b] = np.linspace(lowest, hi, num_speed, retstep=True)
    return v, res


def rectangle_rings_to_rectangles(poly: np.ndarray, nvert: np.ndarray, dims: np.ndarray) -> np.ndarray:
    """obs_nvert with every 4-vertex ring that IS its column's rectangle set to 0 (= "the rectangle of obs_dims").

    A ring whose vertices are exactly the corners (+-l/2, +-w/2) of obs_dims, walked counter-clockwise from any corner, is the same
    polygon as the rectangle column - and the rectangle column takes the kernels' rectangle narrow phase (a scene made of such rings
    only takes the rectangle-only instances: the polygon instances carry run-time shapes and ~15 more registers, 5-19 % per launch).
    Exact comparison on the doubles: anything else stays a polygon column."""
    nvert = np.array(nvert, dtype=np.int32, copy=True)
    four = nvert == 4
    if not four.any():
        return nvert
    hl, hw = 0.5 * dims[..., 0], 0.5 * dims[..., 1]
    corners = np.stack([np.stack([hl, hw], -1), np.stack([-hl, hw], -1), np.stack([-hl, -hw], -1), np.stack([hl, -hw], -1)], axis=-2)  # CCW
    ring = poly[..., :4, :]
    is_box = np.zeros(nvert.shape, dtype=bool)
    for start in range(4):
        is_box |= np.all(ring == np.roll(corners, -start, axis=-2), axis=(-1, -2))
    nvert[four & is_box & (hl > 0) & (hw > 0)] = 0
    return nvert


@dataclass
class ProblemBatch:
    d_samples: np.ndarray
    t_samples: np.ndarray
    v_samples: np.ndarray
    target_speed: np.ndarray
    ego: np.ndarray
    frame_of: np.ndarray
    scene_of: np.ndarray
    t_now: np.ndarray
    nx: np.ndarray
    knots: np.ndarray
    coef: np.ndarray
    obs_pose: np.ndarray
    obs_dims: np.ndarray
    final_time_step: np.ndarray
    veh_l: float
    veh_w: float
    max_speed: float
    max_accel: float
    tick_t: float = 0.1
    check_stride: int = 2
    samp_min: np.ndarray | None = None  # [B, 3] (d, v, t) FISS/FISS+ sampling box
    samp_max: np.ndarray | None = None
    samp_res: np.ndarray | None = None
    # (max_curvature, max_kappa_d, max_kappa_dd): turns on the curvature checks the reference carries commented out
    # (frenet_optimal_planner.py:145-150); None = off = the reference's behaviour
    curvature_limits: tuple | None = None
    # obstacle columns that are convex polygons instead of rectangles (fp_batch.obs_poly / obs_nvert); None = rectangles only
    obs_poly: np.ndarray | None = None   # [S, n_obs, PV, 2]
    obs_nvert: np.ndarray | None = None  # [S, n_obs] int32
    meta: dict = field(default_factory=dict)

    def __post_init__(self):
        f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i4 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self.d_samples, self.t_samples = f8(self.d_samples), f8(self.t_samples)
        self.v_samples, self.target_speed, self.ego = f8(self.v_samples), f8(self.target_speed), f8(self.ego)
        self.frame_of, self.scene_of, self.t_now = i4(self.frame_of), i4(self.scene_of), i4(self.t_now)
        self.nx, self.knots, self.coef = i4(self.nx), f8(self.knots), f8(self.coef)
        self.obs_pose, self.obs_dims = f8(self.obs_pose), f8(self.obs_dims)
        self.final_time_step = i4(self.final_time_step)
        B = self.ego.shape[0]
        assert self.ego.shape == (B, 6) and self.v_samples.shape == (B, self.nv)
        assert self.knots.ndim == 2 and self.coef.shape == (self.F, 8, self.NX)
        assert self.obs_pose.ndim == 4 and self.obs_pose.shape[-1] == 4
        assert self.obs_dims.shape == (self.S, self.n_obs, 2)
        for name in ("samp_min", "samp_max", "samp_res"):
            if getattr(self, name) is not None:
                setattr(self, name, f8(getattr(self, name)))
        if self.obs_nvert is not None:
            self.obs_nvert = rectangle_rings_to_rectangles(np.asarray(self.obs_poly, dtype=np.float64), np.asarray(self.obs_nvert), self.obs_dims)
        if self.obs_nvert is not None and not np.any(np.asarray(self.obs_nvert)):
            self.obs_poly = self.obs_nvert = None  # no polygon column after all
        if self.obs_nvert is not None:
            self.obs_poly, self.obs_nvert = f8(self.obs_poly), i4(self.obs_nvert)
            assert self.obs_nvert.shape == (self.S, self.n_obs) and self.obs_poly.ndim == 4
            assert self.obs_poly.shape[:2] == (self.S, self.n_obs) and self.obs_poly.shape[3] == 2 and self.obs_poly.shape[2] >= 3

    B = property(lambda self: self.ego.shape[0])
    nd = property(lambda self: self.d_samples.shape[0])
    nv = property(lambda self: self.v_samples.shape[1])
    nt = property(lambda self: self.t_samples.shape[0])
    C = property(lambda self: self.nd * self.nv * self.nt)
    F = property(lambda self: self.knots.shape[0])
    NX = property(lambda self: self.knots.shape[1])
    S = property(lambda self: self.obs_pose.shape[0])
    T_obs = property(lambda self: self.obs_pose.shape[1])
    n_obs = property(lambda self: self.obs_pose.shape[2])
    poly_stride = property(lambda self: 0 if self.obs_nvert is None else self.obs_poly.shape[2])

    def points_per_candidate(self) -> np.ndarray:
        """N(T) = len(np.arange(0, T, tick_t)) for each T sample."""
        return np.ceil(self.t_samples / self.tick_t).astype(np.int64)

    def take(self, egos, meta: dict | None = None) -> "ProblemBatch":
        """The sub-batch of the given egos (index array or slice, in that order); the frames / scenes they reference are re-indexed."""
        sel = egos if isinstance(egos, slice) else np.asarray(egos, dtype=np.int64)
        fr, fi = np.unique(self.frame_of[sel], return_inverse=True)
        sc_all = self.scene_of[sel]
        sc, si = np.unique(sc_all[sc_all >= 0], return_inverse=True)
        scene_of = np.full(len(sc_all), -1, dtype=np.int32)
        scene_of[sc_all >= 0] = si
        keep_s = sc if sc.size else np.zeros(0, dtype=np.int64)
        return ProblemBatch(
            d_samples=self.d_samples, t_samples=self.t_samples, v_samples=self.v_samples[sel],
            target_speed=self.target_speed[sel], ego=self.ego[sel], frame_of=fi, scene_of=scene_of,
            t_now=self.t_now[sel], nx=self.nx[fr], knots=self.knots[fr], coef=self.coef[fr],
            obs_pose=self.obs_pose[keep_s], obs_dims=self.obs_dims[keep_s],
            final_time_step=self.final_time_step[keep_s], veh_l=self.veh_l, veh_w=self.veh_w,
            max_speed=self.max_speed, max_accel=self.max_accel, tick_t=self.tick_t, check_stride=self.check_stride,
            samp_min=None if self.samp_min is None else self.samp_min[sel],
            samp_max=None if self.samp_max is None else self.samp_max[sel],
            samp_res=None if self.samp_res is None else self.samp_res[sel], curvature_limits=self.curvature_limits,
            obs_poly=None if self.obs_nvert is None else self.obs_poly[keep_s],
            obs_nvert=None if self.obs_nvert is None else self.obs_nvert[keep_s],
            meta=dict(self.meta, **(meta or {})))

    def shard(self, rank: int, world: int) -> "ProblemBatch":
        """Contiguous ego range of one rank; frames/scenes referenced by it are re-indexed."""
        lo, hi = (self.B * rank) // world, (self.B * (rank + 1)) // world
        return self.take(slice(lo, hi), meta=dict(rank=rank, world=world))

    def digest(self) -> str:
        """SHA-256 over every array: lets the GPU box prove it regenerated the same inputs."""
        h = hashlib.sha256()
        for name in ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx",
                     "knots", "coef", "obs_pose", "obs_dims", "final_time_step"):
            a = getattr(self, name)
            h.update(name.encode()); h.update(str(a.shape).encode()); h.update(np.ascontiguousarray(a).tobytes())
        if self.obs_nvert is not None:  # (rectangle-only batches keep the digests pinned before ABI 12)
            for name in ("obs_poly", "obs_nvert"):
                a = getattr(self, name)
                h.update(name.encode()); h.update(str(a.shape).encode()); h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()
