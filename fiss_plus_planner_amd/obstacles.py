"""Obstacle lists -> the flat pose table the kernels read.

The reference's has_collision (frenet_optimal_planner.py:168-195) touches obstacles only through
``obstacles[0].prediction.final_time_step``, ``obstacle.state_at_time(t)`` (-> None or an object with
``.position[0:2]``, ``.orientation``) and ``obstacle.obstacle_shape.shapely_object`` (any polygon).  flatten_obstacles
walks exactly that duck-typed surface once per scenario and produces

    pose [T_obs, n_obs, 4] = x, y, yaw, valid(0/1)      dims [n_obs, 2] = length, width
    (+ poly / nvert for columns that are convex polygons: circles, rotated rectangles, the convex pieces of anything else)

Only time steps below ``final_time_step`` of the FIRST obstacle can ever be queried (i + t_now < that
bound, :173-176), so T_obs = final_time_step rows are enough.
"""
from __future__ import annotations

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
    # columns that are convex polygons (fp_batch.obs_poly / obs_nvert): nvert[j] = 0 for a rectangle, else the vertices of the
    # counter-clockwise ring poly[j, :nvert[j]] relative to the column's rotation centre (dims[j] = the centred box that holds it)
    poly: np.ndarray | None = None    # [n_obs, PV, 2]
    nvert: np.ndarray | None = None   # [n_obs] int32

    def __post_init__(self):
        self.pose = np.ascontiguousarray(self.pose, dtype=np.float64)
        self.dims = np.ascontiguousarray(self.dims, dtype=np.float64)
        assert self.pose.ndim == 3 and self.pose.shape[2] == 4 and self.dims.shape == (self.pose.shape[1], 2)
        if self.nvert is not None:
            self.poly = np.ascontiguousarray(self.poly, dtype=np.float64)
            self.nvert = np.ascontiguousarray(self.nvert, dtype=np.int32)
            assert self.nvert.shape == (self.pose.shape[1],) and self.poly.ndim == 3 and self.poly.shape[0] == self.pose.shape[1] and self.poly.shape[2] == 2

    def freeze(self) -> None:
        """Called by a planner when it caches the table on the device: writes through these arrays raise from now on."""
        self.pose.setflags(write=False)
        self.dims.setflags(write=False)
        if self.nvert is not None:
            self.poly.setflags(write=False)
            self.nvert.setflags(write=False)

    def update(self, pose=None, dims=None, final_time_step=None, poly=None, nvert=None) -> "ObstacleTable":
        """Replace (never edit) the arrays; every planner that cached the old content uploads the new one on its next plan()."""
        if pose is not None:
            self.pose = np.array(pose, dtype=np.float64, order="C")
        if dims is not None:
            self.dims = np.array(dims, dtype=np.float64, order="C")
        if final_time_step is not None:
            self.final_time_step = int(final_time_step)
        if (poly is None) != (nvert is None):
            raise ValueError("ObstacleTable.update: poly and nvert come together (the rings and their vertex counts)")
        if nvert is not None:
            self.poly, self.nvert = np.array(poly, dtype=np.float64, order="C"), np.array(nvert, dtype=np.int32, order="C")
        n = self.pose.shape[1] if self.pose.ndim == 3 else -1
        if self.pose.ndim != 3 or self.pose.shape[2] != 4 or self.dims.shape != (n, 2):
            raise ValueError(f"ObstacleTable.update: pose {self.pose.shape} / dims {self.dims.shape} are not [T, n, 4] / [n, 2]")
        if self.nvert is not None and (self.nvert.shape != (n,) or self.poly.ndim != 3 or self.poly.shape[0] != n or self.poly.shape[2] != 2
                                       or (self.nvert.size and int(self.nvert.max()) > self.poly.shape[1])):
            raise ValueError(f"ObstacleTable.update: poly {self.poly.shape} / nvert {self.nvert.shape} do not match the {n} obstacle columns "
                             "(a new pose table with another column count needs new rings too)")
        self.version += 1
        return self


def _ring(poly):
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


def buffer_circle_ring(radius: float, cx: float = 0.0, cy: float = 0.0, quad_segs: int = 16) -> np.ndarray:
    """The ring shapely's ``Point(cx, cy).buffer(radius)`` returns (what a commonroad Circle's ``shapely_object`` is): GEOS's
    OffsetSegmentGenerator::createCircle - 4 * quad_segs vertices, clockwise from (cx + r, cy), cos / sin snapped to 0 below 5e-16.
    Restated from GEOS 3.11 (shapely 2.0.0 bundles it; environment.yml:184); not checkable offline beyond the vertex count and the
    first coordinates shapely's documentation prints - the vertices may differ from the real ones in the last ulp."""
    n = 4 * int(quad_segs)
    ang = -np.arange(n) * (2.0 * np.pi / n)
    c, s = np.cos(ang), np.sin(ang)
    c[np.abs(c) < 5e-16] = 0.0
    s[np.abs(s) < 5e-16] = 0.0
    return np.stack([cx + radius * c, cy + radius * s], axis=1)


def _signed_area2(v: np.ndarray) -> float:
    x, y = v[:, 0], v[:, 1]
    return float(np.sum(x * np.roll(y, -1) - y * np.roll(x, -1)))


def _is_convex(v: np.ndarray) -> bool:
    """Counter-clockwise ring: every turn is a left turn (or straight) and the ring winds once."""
    e = np.roll(v, -1, axis=0) - v
    cr = e[:, 0] * np.roll(e[:, 1], -1) - e[:, 1] * np.roll(e[:, 0], -1)
    if np.any(cr < 0.0):
        return False
    for d in (e[:, 0], e[:, 1]):  # a ring that winds once changes direction twice along either axis
        sg = np.sign(d[d != 0.0])
        if sg.size and np.count_nonzero(sg != np.roll(sg, 1)) > 2:
            return False
    return True


def _triangulate(v: np.ndarray) -> list:
    """Ear clipping of a simple counter-clockwise ring -> index triples.  The triangles' vertices are the ring's own vertices, so the
    union of the (closed) triangles IS the polygon: Polygon.intersects(ego, shape) == any(intersects(ego, triangle))."""
    idx = list(range(len(v)))
    tris = []

    def cross(a, b, c):
        return (v[b, 0] - v[a, 0]) * (v[c, 1] - v[a, 1]) - (v[b, 1] - v[a, 1]) * (v[c, 0] - v[a, 0])

    guard = 0
    while len(idx) > 3 and guard < 10 * len(v) + 10:
        guard += 1
        n = len(idx)
        for k in range(n):
            a, b, c = idx[(k - 1) % n], idx[k], idx[(k + 1) % n]
            cr = cross(a, b, c)
            if cr < 0.0:
                continue  # reflex corner
            if cr == 0.0:  # straight: drop the middle vertex, no triangle
                idx.pop(k)
                break
            inside = False
            for m in idx:
                if m in (a, b, c):
                    continue
                if cross(a, b, m) >= 0.0 and cross(b, c, m) >= 0.0 and cross(c, a, m) >= 0.0:
                    inside = True
                    break
            if not inside:
                tris.append((a, b, c))
                idx.pop(k)
                break
        else:
            raise ValueError("obstacle shape is not a simple polygon (ear clipping found no ear)")
    if len(idx) == 3 and cross(*idx) > 0.0:
        tris.append(tuple(idx))
    return tris


def _merge_across_edge(a: list, b: list):
    """Two counter-clockwise index rings that share an edge (a: e0 -> e1, b: e1 -> e0) -> the ring of their union, else None."""
    for ka in range(len(a)):
        e0, e1 = a[ka], a[(ka + 1) % len(a)]
        if e1 in b and b[(b.index(e1) + 1) % len(b)] == e0:
            kb = b.index(e1)
            around_a = a[ka + 1:] + a[:ka + 1]                  # e1 ... e0
            rest_of_b = [b[(kb + 2 + m) % len(b)] for m in range(len(b) - 2)]  # what follows e0 in b, up to (not including) e1
            return around_a + rest_of_b
    return None


def _convex_pieces(v: np.ndarray, max_vertices: int) -> list:
    """A counter-clockwise simple ring -> convex counter-clockwise rings of at most max_vertices vertices whose union is the polygon."""
    if _is_convex(v):
        pieces = [v]
    else:
        # triangles, then neighbours merged while the union stays convex (fewer columns; any convex partition gives the same verdicts)
        idx = [list(t) for t in _triangulate(v)]
        merged = True
        while merged:
            merged = False
            for i in range(len(idx)):
                for j in range(i + 1, len(idx)):
                    ring = _merge_across_edge(idx[i], idx[j])
                    if ring is not None and len(set(ring)) == len(ring) and _is_convex(v[ring]):
                        idx[i] = ring
                        idx.pop(j)
                        merged = True
                        break
                if merged:
                    break
        pieces = [v[r] for r in idx]
    out = []
    for pc in pieces:  # a fan keeps the pieces of a many-sided convex ring convex
        while len(pc) > max_vertices:
            out.append(pc[:max_vertices])
            pc = np.concatenate([pc[:1], pc[max_vertices - 1:]])
        out.append(pc)
    return out


MAX_POLY_VERTS = 128  # FP_MAX_POLY_VERTS (include/frenet_gpu.h)
# commonroad-io's Circle.shapely_object is `Point(center).buffer(radius / 2)` in the releases of the reference's era (recalled, not
# checkable offline - like the vehicle constants).  ONE factor for every path that has to make a circle's polygon itself: shape objects
# that bring a radius but no polygon (shape_columns) and the XML reader (commonroad_xml.load_scenario's default).  Objects that bring
# their own shapely_object never use it.
CIRCLE_BUFFER_FACTOR = 0.5


def _ring_is_simple(v: np.ndarray) -> bool:
    """True iff no two non-adjacent edges of the closed ring `v` touch or cross and no two adjacent edges fold back onto each other
    (O(n^2) orientation tests on the ring's own coordinates; n <= a few hundred).  Ear clipping cannot tell: the ears of a
    self-crossing ring are all positively oriented and their signed areas still sum to the ring's."""
    n = len(v)
    if n < 3:
        return False
    a, b = v, np.roll(v, -1, axis=0)
    if np.any(np.all(a == b, axis=1)):
        return False  # a repeated vertex: a zero-length edge (shapely: invalid ring)

    def orient(p, q, r):  # sign of the cross product (q - p) x (r - p), vectorised
        return np.sign((q[..., 0] - p[..., 0]) * (r[..., 1] - p[..., 1]) - (q[..., 1] - p[..., 1]) * (r[..., 0] - p[..., 0]))

    def on_seg(p, q, r):  # r collinear with pq: inside its bounding box?
        return (np.minimum(p[..., 0], q[..., 0]) <= r[..., 0]) & (r[..., 0] <= np.maximum(p[..., 0], q[..., 0])) & \
               (np.minimum(p[..., 1], q[..., 1]) <= r[..., 1]) & (r[..., 1] <= np.maximum(p[..., 1], q[..., 1]))

    i, j = np.triu_indices(n, 1)
    adj = (j == i + 1) | ((i == 0) & (j == n - 1))
    # adjacent edges share one vertex by construction; they overlap beyond it only when they are collinear and point back
    ia, ja = i[adj], j[adj]
    first_last = (ia == 0) & (ja == n - 1)  # edges (n-1 -> 0) and (0 -> 1): shared vertex v[0]
    shared = np.where(first_last[:, None], a[ia], a[ja])
    e1 = np.where(first_last[:, None], b[ia] - a[ia], a[ia] - a[ja])   # away from the shared vertex along one edge
    e2 = np.where(first_last[:, None], a[ja] - shared, b[ja] - a[ja])  # ... and along the other
    cross = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    dot = e1[:, 0] * e2[:, 0] + e1[:, 1] * e2[:, 1]
    if np.any((cross == 0.0) & (dot > 0.0)):
        return False
    i, j = i[~adj], j[~adj]
    if len(i) == 0:
        return True
    p1, q1, p2, q2 = a[i], b[i], a[j], b[j]
    o1, o2, o3, o4 = orient(p1, q1, p2), orient(p1, q1, q2), orient(p2, q2, p1), orient(p2, q2, q1)
    hit = (o1 != o2) & (o3 != o4)
    hit |= (o1 == 0) & on_seg(p1, q1, p2)
    hit |= (o2 == 0) & on_seg(p1, q1, q2)
    hit |= (o3 == 0) & on_seg(p2, q2, p1)
    hit |= (o4 == 0) & on_seg(p2, q2, q1)
    return not bool(hit.any())


def shape_columns(shape, circle_buffer_factor: float = CIRCLE_BUFFER_FACTOR) -> list:
    """An obstacle shape -> the obstacle columns that stand for it: [(length, width, cx, cy, ring or None)].

    The reference hands ``obstacle_shape.shapely_object`` - any polygon - to translate + rotate(origin='center') + intersects
    (frenet_optimal_planner.py:162-166, :189-191): the shape turns about the centre of ITS bounding box, i.e. it behaves like the
    same shape centred on the origin, displaced by the unrotated offset (cx, cy) of that centre.

    * a commonroad-like Rectangle (``.length`` / ``.width``, centred, orientation 0) or any polygon that is an axis-aligned
      rectangle in its own frame: one rectangle column, ring None;
    * a convex polygon - a rotated rectangle, a commonroad Circle (its shapely_object is shapely's 64-gon) - one polygon column:
      ring = counter-clockwise vertices relative to the bounding-box centre, (length, width) = the bounding box;
    * a non-convex simple polygon or a group of shapes (``.shapes``, a commonroad ShapeGroup: its MultiPolygon turns about the
      centre of the WHOLE group's bounding box): several columns that share the obstacle's poses, one convex piece each (a convex
      partition on the shape's own vertices: the union of the closed pieces is the shape, so `intersects` is the OR over the
      pieces), every ring relative to the whole shape's centre and (length, width) = the centred box that holds the piece.
    Holes are ignored (the exterior ring is taken)."""
    subs = getattr(shape, "shapes", None)
    if subs is None and hasattr(shape, "length") and hasattr(shape, "width") and not np.any(np.asarray(getattr(shape, "center", 0.0), dtype=float)) \
            and not float(getattr(shape, "orientation", 0.0) or 0.0):
        return [(float(shape.length), float(shape.width), 0.0, 0.0, None)]
    rings = []
    for sh in (subs if subs is not None else [shape]):
        poly = getattr(sh, "shapely_object", sh)
        v = _ring(poly)
        if v is None and hasattr(sh, "radius"):  # a circle that brings no polygon: the one shapely would make of it
            c = np.asarray(getattr(sh, "center", (0.0, 0.0)), dtype=float)
            v = buffer_circle_ring(circle_buffer_factor * float(sh.radius), float(c[0]), float(c[1]))
        if v is None or len(v) < 3:
            raise ValueError(f"obstacle shape {sh!r} exposes no polygon (shapely_object / exterior / vertices) and no radius")
        if not np.all(np.isfinite(v)):
            raise ValueError("obstacle shape has a non-finite vertex")
        rings.append(v if _signed_area2(v) > 0.0 else v[::-1])
    allv = np.concatenate(rings)
    minx, miny = allv.min(axis=0)
    maxx, maxy = allv.max(axis=0)
    cx, cy = 0.5 * (minx + maxx), 0.5 * (miny + maxy)
    if len(rings) == 1:
        v = rings[0]
        is_rect = len(v) == 4 and all((p[0] in (minx, maxx)) and (p[1] in (miny, maxy)) for p in v) and len({(float(p[0]), float(p[1])) for p in v}) == 4 \
            and all((v[k][0] == v[(k + 1) % 4][0]) != (v[k][1] == v[(k + 1) % 4][1]) for k in range(4))  # (walked along its sides: the four corners in crossing order are a bow tie)
        if is_rect:
            return [(float(maxx - minx), float(maxy - miny), float(cx), float(cy), None)]
    cols = []
    for v in rings:
        if _signed_area2(v) == 0.0:
            continue  # a degenerate ring has no interior; shapely would call the polygon invalid
        # a self-intersecting ring has no convex partition on its own vertices.  Tested directly (edge pairs); the area comparison
        # below stays as a second guard only - the ears of a crossing ring are all positive and sum to the ring's signed area, so by
        # itself it fires for the 4-vertex bow tie and little else
        if not _ring_is_simple(v):
            raise ValueError("obstacle shape is not a simple polygon (its ring crosses itself): no convex pieces stand for it")
        pieces = _convex_pieces(v, MAX_POLY_VERTS)
        a_ring, a_pieces = abs(_signed_area2(v)), sum(abs(_signed_area2(pc)) for pc in pieces)
        if not all(_is_convex(pc) and _signed_area2(pc) >= 0.0 for pc in pieces) or abs(a_pieces - a_ring) > 1e-9 * max(a_ring, 1e-300) + 1e-12:
            raise ValueError("obstacle shape is not a simple polygon (its ring crosses itself): no convex pieces stand for it")
        for pc in pieces:
            u = pc - np.array([cx, cy])
            cols.append((float(2.0 * np.abs(u[:, 0]).max()), float(2.0 * np.abs(u[:, 1]).max()), float(cx), float(cy), np.ascontiguousarray(u)))
    if not cols:
        raise ValueError("obstacle shape has no area")
    return cols


def flatten_obstacles(obstacles) -> ObstacleTable:
    fts = int(obstacles[0].prediction.final_time_step)  # AttributeError for a StaticObstacle first, as in the reference
    T = max(fts, 1)
    cols, owner = [], []
    for j, ob in enumerate(obstacles):
        for col in shape_columns(ob.obstacle_shape):
            cols.append(col)
            owner.append(j)
    n = len(cols)
    pose = np.zeros((T, n, 4))
    dims = np.zeros((n, 2))
    pv = max([len(c[4]) for c in cols if c[4] is not None], default=0)
    poly = np.zeros((n, max(pv, 3), 2)) if pv else None
    nvert = np.zeros(n, dtype=np.int32) if pv else None
    states = {}
    for k, (l, w, cx, cy, ring) in enumerate(cols):
        dims[k] = (l, w)
        if ring is not None:
            poly[k, :len(ring)] = ring
            nvert[k] = len(ring)
        ob = obstacles[owner[k]]
        if owner[k] not in states:  # (state_at_time is walked once per obstacle, whatever the number of its columns)
            states[owner[k]] = [ob.state_at_time(t) for t in range(T)]
        for t, st in enumerate(states[owner[k]]):
            if st is None:
                continue
            pose[t, k] = (st.position[0] + cx, st.position[1] + cy, st.orientation, 1.0)
    return ObstacleTable(pose, dims, fts, poly=poly, nvert=nvert)


def obstacles_fingerprint(obstacles) -> tuple:
    """Cheap identity of an obstacle list for the planners' table cache: the objects themselves (ids of the elements, in order)
    plus the horizon.  A caller that rebuilds the list from NEW obstacle objects every cycle gets a fresh table; mutating an
    obstacle object in place is not detected (pass an ObstacleTable to control the table yourself)."""
    return (tuple(id(o) for o in obstacles), int(obstacles[0].prediction.final_time_step))
