"""Shared by the shape tests (G12: obstacle shapes that are not rectangles).  The fixture holds the RAW inputs - synth's obstacle
motion, the own-frame rings of every shape - and the reference's verdicts; the problem batch is rebuilt here through the product's own
shape handling (fiss_plus_planner_amd/obstacles.py: shapely_object -> obstacle columns), so that path is under test too."""
import os
import sys

import numpy as np

from conftest import GOLDEN, batch_from_golden

sys.path.insert(0, GOLDEN)
import refshim  # noqa: E402  (Polygon / MultiPolygon / StubShapeObstacle: plain python, no reference needed)

from fiss_plus_planner_amd.batch import ProblemBatch  # noqa: E402
from fiss_plus_planner_amd.obstacles import flatten_obstacles  # noqa: E402


def g12_obstacles(g, pose, dims, fts):
    """The obstacle list of one G12 scene, as the reference saw it (duck-typed commonroad obstacles)."""
    out = []
    for j in range(pose.shape[1]):
        p = pose[:, j, :3].copy()
        p[pose[:, j, 3] == 0.0] = np.nan
        n = g["shape_ring_n"][j]
        rings = [g["shape_rings"][j, k, :n[k]] for k in range(2) if n[k] > 0]
        if not rings:
            out.append(refshim.StubObstacle(dims[j, 0], dims[j, 1], p, fts))
        elif len(rings) == 2:
            out.append(refshim.StubShapeObstacle(refshim.MultiPolygon(rings), p, fts))
        else:
            out.append(refshim.StubShapeObstacle(refshim.Polygon(rings[0]), p, fts))
    return out


def g12_batch(g, name, kind="FOP", boxes=False):
    """ProblemBatch of a G12 case with the shapes as polygon columns (boxes=True: every shape replaced by its bounding box, what the
    build did before ABI 12)."""
    b = batch_from_golden(g, f"{name}_in_")
    tabs = []
    for sc in range(b.S):
        obs = g12_obstacles(g, b.obs_pose[sc], b.obs_dims[sc], int(b.final_time_step[sc]))
        if boxes:
            for ob in obs:
                minx, miny, maxx, maxy = ob.obstacle_shape.shapely_object.bounds
                ob.obstacle_shape = type(ob.obstacle_shape)(shapely_object=refshim.Polygon([(minx, miny), (maxx, miny), (maxx, maxy), (minx, maxy)]))
        tabs.append(flatten_obstacles(obs))
    pv = max(t.poly.shape[1] for t in tabs if t.nvert is not None) if any(t.nvert is not None for t in tabs) else 0
    poly = nvert = None
    if pv:
        poly = np.zeros((b.S, tabs[0].pose.shape[1], pv, 2))
        nvert = np.zeros((b.S, tabs[0].pose.shape[1]), dtype=np.int32)
        for sc, t in enumerate(tabs):
            poly[sc, :, :t.poly.shape[1]] = t.poly
            nvert[sc] = t.nvert
    kw = {k: getattr(b, k) for k in ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
                                      "final_time_step", "veh_l", "veh_w", "max_speed", "max_accel", "tick_t", "check_stride", "samp_min", "samp_max", "samp_res")}
    if kind in ("FISS", "FISS+"):
        for k in ("d_samples", "samp_min", "samp_max", "samp_res"):
            kw[k] = g[f"{name}_{kind}_in_{k}"]
    T = b.obs_pose.shape[1]
    pose = np.stack([np.concatenate([t.pose, np.zeros((T - t.pose.shape[0],) + t.pose.shape[1:])]) for t in tabs])  # (flatten keeps final_time_step rows)
    return ProblemBatch(**kw, obs_pose=pose, obs_dims=np.stack([t.dims for t in tabs]), obs_poly=poly, obs_nvert=nvert)


from fiss_plus_planner_amd.synth import random_convex_ring, with_random_shapes  # noqa: E402,F401  (the generator lives with the other synthetic inputs)
