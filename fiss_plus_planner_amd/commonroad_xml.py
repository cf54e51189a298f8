"""CommonRoad 2020a scenario files -> the arrays the planners need (no commonroad-io dependency).

Replaces, for the reference's demo inputs, `CommonRoadFileReader(...).open()` (planners/benchmark/planning.py:303) and
`GlobalPlanner.plan_global_route` (planners/commonroad_interface/global_planner.py:22-106):

* lanelet centre line = mean of the left and right bound vertices;
* route = shortest successor-only lanelet sequence from the lanelet under the initial position to the goal lanelet
  (the reference asks commonroad-route-planner for it; all five demo scenarios have a unique such route - SURVEY.md 8f);
  like the reference (:68-75) the first successor of the last lanelet is appended when there is one;
* centerline = concatenated centre vertices with duplicates removed, first occurrence kept (:79-82);
* dynamic obstacles -> pose table [T, n, 4] = x, y, yaw, valid + dims [n, 2] (what has_collision() reads); shapes other than
  centred rectangles - rectangles with a centre / orientation, circles, polygons - become polygon columns (obstacles.shape_columns:
  `obstacle_shape.shapely_object` is any polygon in the reference, frenet_optimal_planner.py:189-191).

Route choice parity with commonroad-route-planner is unpinned (package not installable offline).
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from collections import deque
from dataclasses import dataclass

import numpy as np

from types import SimpleNamespace

from .obstacles import CIRCLE_BUFFER_FACTOR, ObstacleTable, buffer_circle_ring, shape_columns


@dataclass
class Scenario:
    benchmark_id: str
    dt: float
    route: list
    centerline: np.ndarray        # [n, 2]
    obstacles: ObstacleTable
    init_state: np.ndarray        # x, y, yaw, v
    goal_lanelet: int
    goal_center: np.ndarray       # middle centre vertex of the goal lanelet (planning.py:54-58)
    goal_speed: tuple | None      # (start, end) of the goal velocity interval, if any (planning.py:44-48)

    @property
    def max_speed(self) -> float:
        return self.goal_speed[1] if self.goal_speed else 13.5  # planning.py:49-52


def _points(node):
    return np.array([[float(p.find("x").text), float(p.find("y").text)] for p in node.findall("point")])


def _inside(poly: np.ndarray, x: float, y: float) -> bool:
    inside = False
    n = len(poly)
    for i in range(n):
        x1, y1 = poly[i]
        x2, y2 = poly[(i + 1) % n]
        if (y1 > y) != (y2 > y) and x < (x2 - x1) * (y - y1) / (y2 - y1) + x1:
            inside = not inside
    return inside


def _shape(node, circle_buffer_factor: float):
    """<shape> of a CommonRoad obstacle -> the object shape_columns() reads (what commonroad-io's Shape classes expose)."""
    kinds = [c for c in node if c.tag in ("rectangle", "circle", "polygon")]
    if not kinds:
        raise ValueError("obstacle without a rectangle / circle / polygon shape")

    def one(c):
        ctr = c.find("center")
        center = np.array([float(ctr.find("x").text), float(ctr.find("y").text)]) if ctr is not None else np.zeros(2)
        if c.tag == "rectangle":
            l, w = float(c.find("length").text), float(c.find("width").text)
            o = c.find("orientation")
            theta = float(o.find("exact").text if o.find("exact") is not None else o.text) if o is not None else 0.0
            if theta == 0.0 and not center.any():
                return SimpleNamespace(length=l, width=w)
            # commonroad Rectangle.shapely_object: the corners turned by the orientation, then moved to the centre
            corners = np.array([(-0.5 * l, -0.5 * w), (-0.5 * l, 0.5 * w), (0.5 * l, 0.5 * w), (0.5 * l, -0.5 * w)])
            rot = np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
            return SimpleNamespace(vertices=corners @ rot.T + center)
        if c.tag == "circle":
            # commonroad-io's Circle.shapely_object is `Point(center).buffer(radius / 2)` in the releases of the reference's era
            # (recalled, not checkable offline - like the vehicle constants; circle_buffer_factor=1.0 for a full-radius polygon)
            return SimpleNamespace(vertices=buffer_circle_ring(circle_buffer_factor * float(c.find("radius").text), center[0], center[1]))
        return SimpleNamespace(vertices=_points(c))

    shapes = [one(c) for c in kinds]
    return shapes[0] if len(shapes) == 1 else SimpleNamespace(shapes=shapes)  # several shapes under one <shape>: a ShapeGroup


def load_scenario(path: str, circle_buffer_factor: float = CIRCLE_BUFFER_FACTOR) -> Scenario:
    root = ET.parse(path).getroot()
    lanelets = {}
    for ll in root.findall("lanelet"):
        left, right = _points(ll.find("leftBound")), _points(ll.find("rightBound"))
        lanelets[int(ll.get("id"))] = dict(left=left, right=right, center=(left + right) / 2,
                                           succ=[int(s.get("ref")) for s in ll.findall("successor")])
    pp = root.find("planningProblem")
    ini = pp.find("initialState")
    init = np.array([float(ini.find("position/point/x").text), float(ini.find("position/point/y").text),
                     float(ini.find("orientation/exact").text), float(ini.find("velocity/exact").text)])
    goal = pp.find("goalState")
    goal_lanelet = int(goal.find("position/lanelet").get("ref"))
    gv = goal.find("velocity")
    goal_speed = (float(gv.find("intervalStart").text), float(gv.find("intervalEnd").text)) if gv is not None else None

    # start lanelets: polygons (left bound + reversed right bound) containing the initial position, best heading match first
    starts = []
    for lid, ll in lanelets.items():
        if _inside(np.vstack([ll["left"], ll["right"][::-1]]), init[0], init[1]):
            c = ll["center"]
            k = int(np.argmin(np.hypot(c[:, 0] - init[0], c[:, 1] - init[1])))
            k = min(k, len(c) - 2)
            heading = np.arctan2(c[k + 1, 1] - c[k, 1], c[k + 1, 0] - c[k, 0])
            dev = abs((heading - init[2] + np.pi) % (2 * np.pi) - np.pi)
            starts.append((dev, lid))
    if not starts:
        raise ValueError("initial position is not on any lanelet")
    route = None
    for _, s in sorted(starts):
        prev = {s: None}
        dq = deque([s])
        while dq and goal_lanelet not in prev:
            u = dq.popleft()
            for v in lanelets[u]["succ"]:
                if v not in prev and v in lanelets:
                    prev[v] = u
                    dq.append(v)
        if goal_lanelet in prev:
            route = [goal_lanelet]
            while prev[route[-1]] is not None:
                route.append(prev[route[-1]])
            route.reverse()
            break
    if route is None:
        raise ValueError("no successor-only route from the initial lanelet to the goal lanelet")
    lanes = list(route)
    if lanelets[route[-1]]["succ"]:
        lanes.append(lanelets[route[-1]]["succ"][0])
    cc = np.concatenate([lanelets[i]["center"] for i in lanes])
    _, first = np.unique(cc, return_index=True, axis=0)
    centerline = cc[np.sort(first)]
    gc = lanelets[goal_lanelet]["center"]
    goal_center = gc[int((gc.shape[0] - 1) / 2)]

    obs, T = [], 0
    for ob in root.findall("dynamicObstacle"):
        states = {}
        for st in [ob.find("initialState")] + ob.find("trajectory").findall("state"):
            states[int(st.find("time/exact").text)] = (float(st.find("position/point/x").text), float(st.find("position/point/y").text),
                                                       float(st.find("orientation/exact").text))
        for col in shape_columns(_shape(ob.find("shape"), circle_buffer_factor)):  # (one column per convex piece, same states)
            obs.append((col, states))
        T = max(T, max(states) + 1)
    if not obs:
        raise ValueError("scenario has no dynamic obstacle (the reference reads dynamic_obstacles[0], planning.py:70)")
    pose = np.zeros((T, len(obs), 4))
    dims = np.zeros((len(obs), 2))
    pv = max([len(c[4]) for c, _ in obs if c[4] is not None], default=0)
    poly = np.zeros((len(obs), max(pv, 3), 2)) if pv else None
    nvert = np.zeros(len(obs), dtype=np.int32) if pv else None
    for j, ((l, w, cx, cy, ring), states) in enumerate(obs):
        dims[j] = (l, w)
        if ring is not None:
            poly[j, :len(ring)] = ring
            nvert[j] = len(ring)
        for t, (x, y, yaw) in states.items():
            pose[t, j] = (x + cx, y + cy, yaw, 1.0)
    fts = max(obs[0][1])  # dynamic_obstacles[0].prediction.final_time_step
    return Scenario(root.get("benchmarkID"), float(root.get("timeStepSize")), route, centerline,
                    ObstacleTable(pose[:max(fts, 1)], dims, fts, poly=poly, nvert=nvert), init, goal_lanelet, np.asarray(goal_center), goal_speed)
