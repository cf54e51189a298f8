"""Obstacle lists -> the flat pose table the kernels read.

The reference's has_collision (frenet_optimal_planner.py:168-195) touches obstacles only through
``obstacles[0].prediction.final_time_step``, ``obstacle.state_at_time(t)`` (-> None or an object with
``.position[0:2]``, ``.orientation``) and ``obstacle.obstacle_shape`` (a rectangle).  flatten_obstacles
walks exactly that duck-typed surface once per scenario and produces

    pose [T_obs, n_obs, 4] = x, y, yaw, valid(0/1)      dims [n_obs, 2] = length, width

Only time steps below ``final_time_step`` of the FIRST obstacle can ever be queried (i + t_now < that
bound, :173-176), so T_obs = final_time_step rows are enough.
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass

import numpy as np


@dataclass
class ObstacleTable:
    """Immutability contract: a planner that is handed an ObstacleTable keeps its device copy across plan() calls (the tables do
    not travel every cycle).  While a planner holds the table its arrays are FROZEN (numpy write flag off): an in-place update
    raises instead of silently planning against stale device tables.  To change the scene build a new table, or call
    ``update(pose=..., dims=..., final_time_step=...)``, which swaps the arrays and bumps ``version`` - the planners key their
    device cache on (table, version) and upload again."""
    pose: np.ndarray           # [T_obs, n_obs, 4]
    dims: np.ndarray           # [n_obs, 2]
    final_time_step: int       # obstacles[0].prediction.final_time_step
    version: int = 0           # bumped by update(); part of the planners' cache key

    def __post_init__(self):
        self.pose = np.ascontiguousarray(self.pose, dtype=np.float64)
        self.dims = np.ascontiguousarray(self.dims, dtype=np.float64)
        assert self.pose.ndim == 3 and self.pose.shape[2] == 4 and self.dims.shape == (self.pose.shape[1], 2)

    def freeze(self) -> None:
        """Called by a planner when it caches the table on the device: writes through these arrays raise from now on."""
        self.pose.setflags(write=False)
        self.dims.setflags(write=False)

    def update(self, pose=None, dims=None, final_time_step=None) -> "ObstacleTable":
        """Replace (never edit) the arrays; every planner that cached the old content uploads the new one on its next plan()."""
        if pose is not None:
            self.pose = np.array(pose, dtype=np.float64, order="C")
        if dims is not None:
            self.dims = np.array(dims, dtype=np.float64, order="C")
        if final_time_step is not None:
            self.final_time_step = int(final_time_step)
        assert self.pose.ndim == 3 and self.pose.shape[2] == 4 and self.dims.shape == (self.pose.shape[1], 2)
        self.version += 1
        return self


def _shape_vertices(poly):
    """Exterior ring of a polygon-like object as an [n, 2] array (closing point dropped), or None when it exposes none."""
    for get in (lambda p: p.exterior.coords, lambda p: p.pts, lambda p: p.vertices):
        try:
            v = np.asarray(get(poly), dtype=float).reshape(-1, 2)
        except Exception:
            continue
        if len(v) >= 2 and np.array_equal(v[0], v[-1]):
            v = v[:-1]
        return v
    return None


def _rect_dims(shape) -> tuple[float, float, float, float]:
    """(length, width, cx, cy) of an obstacle shape in its own frame.

    A commonroad-like Rectangle exposes ``.length`` / ``.width`` (centred, cx = cy = 0).  Anything else is read through its
    polygon (``.shapely_object`` or the object itself): an axis-aligned rectangle is taken exactly - including one that is not
    centred on the origin: the reference rotates the translated polygon about the centre of ITS bounding box
    (affinity.rotate(origin='center'), frenet_optimal_planner.py:162-166), i.e. it behaves like a centred rectangle displaced by
    the unrotated offset (cx, cy).  Any other shape is replaced by its bounding box with a warning: the kernels test rectangles
    only, and the box over-approximates the shape (never misses a collision the reference reports at yaw = 0, but is not the same
    test)."""
    if hasattr(shape, "length") and hasattr(shape, "width"):
        return float(shape.length), float(shape.width), 0.0, 0.0
    poly = getattr(shape, "shapely_object", shape)
    minx, miny, maxx, maxy = poly.bounds
    v = _shape_vertices(poly)
    is_rect = v is not None and len(v) == 4 and all((p[0] in (minx, maxx)) and (p[1] in (miny, maxy)) for p in v) and \
        len({(float(p[0]), float(p[1])) for p in v}) == 4
    if not is_rect:
        warnings.warn("obstacle shape is not an axis-aligned rectangle in its own frame: the collision kernels use its bounding box "
                      f"({maxx - minx:.3f} x {maxy - miny:.3f} m), which over-approximates the reference's polygon test", RuntimeWarning, stacklevel=3)
    return float(maxx - minx), float(maxy - miny), float(0.5 * (minx + maxx)), float(0.5 * (miny + maxy))


def flatten_obstacles(obstacles) -> ObstacleTable:
    fts = int(obstacles[0].prediction.final_time_step)  # AttributeError for a StaticObstacle first, as in the reference
    T = max(fts, 1)
    n = len(obstacles)
    pose = np.zeros((T, n, 4))
    dims = np.zeros((n, 2))
    for j, ob in enumerate(obstacles):
        l, w, cx, cy = _rect_dims(ob.obstacle_shape)
        dims[j] = (l, w)
        for t in range(T):
            st = ob.state_at_time(t)
            if st is None:
                continue
            pose[t, j] = (st.position[0] + cx, st.position[1] + cy, st.orientation, 1.0)
    return ObstacleTable(pose, dims, fts)


def obstacles_fingerprint(obstacles) -> tuple:
    """Cheap identity of an obstacle list for the planners' table cache: the objects themselves (ids of the elements, in order)
    plus the horizon.  A caller that rebuilds the list from NEW obstacle objects every cycle gets a fresh table; mutating an
    obstacle object in place is not detected (pass an ObstacleTable to control the table yourself)."""
    return (tuple(id(o) for o in obstacles), int(obstacles[0].prediction.final_time_step))
