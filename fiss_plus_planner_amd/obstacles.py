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

from dataclasses import dataclass

import numpy as np


@dataclass
class ObstacleTable:
    pose: np.ndarray           # [T_obs, n_obs, 4]
    dims: np.ndarray           # [n_obs, 2]
    final_time_step: int       # obstacles[0].prediction.final_time_step

    def __post_init__(self):
        self.pose = np.ascontiguousarray(self.pose, dtype=np.float64)
        self.dims = np.ascontiguousarray(self.dims, dtype=np.float64)
        assert self.pose.ndim == 3 and self.pose.shape[2] == 4 and self.dims.shape == (self.pose.shape[1], 2)


def _rect_dims(shape) -> tuple[float, float]:
    """length, width of a commonroad-like Rectangle (``.length``/``.width``) or of a polygon-like object
    exposing ``.bounds`` = (minx, miny, maxx, maxy) in the obstacle's own frame."""
    if hasattr(shape, "length") and hasattr(shape, "width"):
        return float(shape.length), float(shape.width)
    poly = getattr(shape, "shapely_object", shape)
    minx, miny, maxx, maxy = poly.bounds
    return float(maxx - minx), float(maxy - miny)


def flatten_obstacles(obstacles) -> ObstacleTable:
    fts = int(obstacles[0].prediction.final_time_step)  # AttributeError for a StaticObstacle first, as in the reference
    T = max(fts, 1)
    n = len(obstacles)
    pose = np.zeros((T, n, 4))
    dims = np.zeros((n, 2))
    for j, ob in enumerate(obstacles):
        dims[j] = _rect_dims(ob.obstacle_shape)
        for t in range(T):
            st = ob.state_at_time(t)
            if st is None:
                continue
            pose[t, j] = (st.position[0], st.position[1], st.orientation, 1.0)
    return ObstacleTable(pose, dims, fts)
