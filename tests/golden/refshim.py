"""Import shim for generating golden vectors from the Python reference.

Two halves.  `load_reference()` imports the reference from /root/reference and RUNS ONLY IN THE BUILD CONTAINER (the GPU box has no
/root/reference): it is called by gen_golden.py / audit_goldens.py alone, never by the product, by smoke() or by bench.py, and no
`-m gpu` test calls it.  The stand-ins for the third-party packages the reference imports but this image lacks (the `shapely`
Polygon / affinity restatement with its exact rational `intersects`, the commonroad stubs) are the builder's own code, import nothing
from the reference, and ARE imported by tests - tests/shapes_util.py (used by tests/test_gpu_shapes.py) and tests/test_gpu_planners.py
build obstacle OBJECTS with `shapely_object` polygons from them, tests/test_collision_exact.py uses the exact predicate as ground
truth.  The vectors the first half helps to produce are committed as .npz files next to this script.

What it does
------------
The reference planners (planners/*.py) import two third-party packages that are
not installed here and cannot be installed (no network):

* ``commonroad`` - only for type hints (``Scenario``, ``Obstacle``), see
  planners/frenet_optimal_planner.py:5, planners/fop_plus_planner.py:3-4.
  -> empty stub classes.
* ``shapely`` (pinned 2.0.0 in environment.yml:184) - ``Polygon``,
  ``affinity.translate``, ``affinity.rotate`` and ``Polygon.intersects`` at
  planners/frenet_optimal_planner.py:162-195 and planners/common/vehicle/vehicle.py:31.
  -> a small stand-in for exactly those four calls on convex polygons.  The
  transform restates shapely 2.0's affinity (rotation about the bounding-box
  centre, 2.5e-16 snap) in fp64; `intersects` is decided EXACTLY on the
  resulting fp64 vertex coordinates (rational arithmetic behind a float filter),
  which is what GEOS's robust predicates do.  This is OUR code, not shapely's;
  it is pinned by hand-computed known-answer tests (tests/test_oracle_kats.py)
  and audited against the plain float test (collision_audit.json).

Everything else (polynomials, cubic spline, cost, FrenetState/FrenetTrajectory,
the four planner classes) is the unmodified reference, imported by path.
"""
from __future__ import annotations

import math
import sys
import types
from fractions import Fraction
from types import SimpleNamespace

import numpy as np

REFERENCE_ROOT = "/root/reference"
AUDIT = {"calls": 0, "near_contact": 0, "float_differs_from_exact": 0}  # of Polygon.intersects, see there


# --------------------------------------------------------------------------
# mini "shapely": convex polygons only
# --------------------------------------------------------------------------
def _ring_is_convex(pts: np.ndarray) -> bool:
    e = np.roll(pts, -1, axis=0) - pts
    cr = e[:, 0] * np.roll(e[:, 1], -1) - e[:, 1] * np.roll(e[:, 0], -1)
    if np.any(cr > 0.0) and np.any(cr < 0.0):
        return False
    for d in (e[:, 0], e[:, 1]):  # winds once: two direction changes along either axis
        sg = np.sign(d[d != 0.0])
        if sg.size and np.count_nonzero(sg != np.roll(sg, 1)) > 2:
            return False
    return True


class Polygon:
    """Simple polygon given by its exterior ring (closing point optional).  Convex polygons - everything the reference's own demo
    inputs produce - take the separating-axis path; any other simple polygon is decided by exact segment / containment tests."""

    def __init__(self, coords):
        pts = np.asarray(coords, dtype=float).reshape(-1, 2)
        if len(pts) >= 2 and np.array_equal(pts[0], pts[-1]):
            pts = pts[:-1]
        if len(pts) < 3:
            raise ValueError("A polygon needs at least 3 distinct vertices")
        if not np.all(np.isfinite(pts)):
            raise ValueError("non-finite polygon coordinate")
        self.pts = pts
        self.convex = _ring_is_convex(pts)

    @property
    def exterior(self):
        return SimpleNamespace(coords=[tuple(p) for p in self.pts] + [tuple(self.pts[0])])

    @property
    def bounds(self):
        mn = self.pts.min(axis=0)
        mx = self.pts.max(axis=0)
        return (mn[0], mn[1], mx[0], mx[1])

    @property
    def is_empty(self):
        return False

    def _axes(self):
        e = np.roll(self.pts, -1, axis=0) - self.pts
        return np.stack([-e[:, 1], e[:, 0]], axis=1)

    def intersects_float(self, other: "Polygon") -> bool:
        """Closed-set separating axis test in fp64 (touching counts as intersecting): rounded projections."""
        for ax in np.concatenate([self._axes(), other._axes()]):
            pa = self.pts @ ax
            pb = other.pts @ ax
            if pa.max() < pb.min() or pb.max() < pa.min():
                return False
        return True

    def intersects_exact(self, other: "Polygon") -> bool:
        """The same closed-set predicate decided EXACTLY on the fp64 vertex coordinates (rational arithmetic): what a robust
        geometry kernel (GEOS: orientation predicates) answers for these polygons.  Two closed convex polygons are disjoint iff
        the line through some edge of one has the whole other polygon strictly on its outer side."""
        A = [(Fraction(float(x)), Fraction(float(y))) for x, y in self.pts]
        B = [(Fraction(float(x)), Fraction(float(y))) for x, y in other.pts]

        def orient(a, b, c):
            v = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
            return (v > 0) - (v < 0)

        for P, Q in ((A, B), (B, A)):
            n = len(P)
            for k in range(n):
                p0, p1 = P[k], P[(k + 1) % n]
                inside = 0
                for m in range(2, n):
                    inside = orient(p0, p1, P[(k + m) % n])
                    if inside:
                        break
                if inside == 0:
                    continue
                if all(orient(p0, p1, q) == -inside for q in Q):
                    return False
        return True

    def intersects_general_exact(self, other: "Polygon") -> bool:
        """Closed-set intersection of two SIMPLE polygons, exactly (rational arithmetic): some edge of one meets some edge of the
        other, or one polygon holds a vertex of the other (then, with no edges meeting, it holds all of it)."""
        A = [(Fraction(float(x)), Fraction(float(y))) for x, y in self.pts]
        B = [(Fraction(float(x)), Fraction(float(y))) for x, y in other.pts]

        def orient(a, b, c):
            v = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
            return (v > 0) - (v < 0)

        def on_seg(a, b, q):  # q collinear with a-b: inside the closed segment?
            return min(a[0], b[0]) <= q[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= q[1] <= max(a[1], b[1])

        def seg_meet(a, b, c, d):
            o1, o2, o3, o4 = orient(a, b, c), orient(a, b, d), orient(c, d, a), orient(c, d, b)
            if o1 != o2 and o3 != o4:
                return True
            return (o1 == 0 and on_seg(a, b, c)) or (o2 == 0 and on_seg(a, b, d)) or (o3 == 0 and on_seg(c, d, a)) or (o4 == 0 and on_seg(c, d, b))

        def holds(P, q):  # closed point-in-polygon: boundary counts; crossing number otherwise
            n = len(P)
            inside = False
            for i in range(n):
                a, b = P[i], P[(i + 1) % n]
                if orient(a, b, q) == 0 and on_seg(a, b, q):
                    return True
                if (a[1] > q[1]) != (b[1] > q[1]):
                    xi = a[0] + (q[1] - a[1]) * (b[0] - a[0]) / (b[1] - a[1])
                    if q[0] < xi:
                        inside = not inside
            return inside

        na, nb = len(A), len(B)
        for i in range(na):
            for j in range(nb):
                if seg_meet(A[i], A[(i + 1) % na], B[j], B[(j + 1) % nb]):
                    return True
        return holds(B, A[0]) or holds(A, B[0])

    def intersects(self, other) -> bool:
        """Polygon.intersects as the goldens see it: the EXACT predicate, reached through a float filter - when every axis
        overlaps, or some axis separates, by a margin far above any rounding (1e-9 relative), the float test's answer is the exact
        one; everything nearer to contact is decided in rational arithmetic.  AUDIT counts the calls, the near-contact calls and
        how often the plain float test would have answered differently (tests/golden/collision_audit.json)."""
        if isinstance(other, MultiPolygon):
            return other.intersects(self)
        AUDIT["calls"] += 1
        if not (self.convex and other.convex):  # (no separating-axis filter for a non-convex operand: bounding boxes, then the exact test)
            a, b = self.bounds, other.bounds
            pad = 1e-9 * (1.0 + max(float(np.abs(self.pts).max()), float(np.abs(other.pts).max())))
            if a[2] + pad < b[0] or b[2] + pad < a[0] or a[3] + pad < b[1] or b[3] + pad < a[1]:
                return False
            AUDIT["general_exact"] = AUDIT.get("general_exact", 0) + 1
            return self.intersects_general_exact(other)
        scale = 1.0 + max(float(np.abs(self.pts).max()), float(np.abs(other.pts).max()))
        worst = -math.inf  # largest separation over the axes, in units of the tolerance
        for ax in np.concatenate([self._axes(), other._axes()]):
            pa = self.pts @ ax
            pb = other.pts @ ax
            sep = max(pb.min() - pa.max(), pa.min() - pb.max())
            tol = 1e-9 * float(np.hypot(ax[0], ax[1])) * scale
            worst = max(worst, sep / tol if tol > 0 else (math.inf if sep > 0 else -math.inf))
        if worst > 1.0:
            return False
        if worst < -1.0:
            return True
        AUDIT["near_contact"] += 1
        exact = self.intersects_exact(other)
        if exact != self.intersects_float(other):
            AUDIT["float_differs_from_exact"] += 1
        return exact


class MultiPolygon:
    """Several polygons as one geometry (a commonroad ShapeGroup's shapely_object): bounds of the whole, intersects = any part."""

    def __init__(self, polygons):
        self.geoms = [g if isinstance(g, Polygon) else Polygon(g) for g in polygons]

    @property
    def bounds(self):
        b = np.array([g.bounds for g in self.geoms])
        return (b[:, 0].min(), b[:, 1].min(), b[:, 2].max(), b[:, 3].max())

    def intersects(self, other) -> bool:
        return any(g.intersects(other) for g in self.geoms)


class _Affinity:
    @staticmethod
    def translate(geom, xoff=0.0, yoff=0.0, zoff=0.0):
        if isinstance(geom, MultiPolygon):
            return MultiPolygon([_Affinity.translate(g, xoff, yoff) for g in geom.geoms])
        return Polygon(geom.pts + np.array([xoff, yoff], dtype=float))

    @staticmethod
    def rotate(geom, angle, origin="center", use_radians=False):
        if not use_radians:
            angle = angle * math.pi / 180.0
        cosp = math.cos(angle)
        sinp = math.sin(angle)
        if abs(cosp) < 2.5e-16:
            cosp = 0.0
        if abs(sinp) < 2.5e-16:
            sinp = 0.0
        if origin != "center":
            raise NotImplementedError(origin)
        minx, miny, maxx, maxy = geom.bounds
        x0 = (minx + maxx) / 2.0
        y0 = (miny + maxy) / 2.0
        xoff = x0 - x0 * cosp + y0 * sinp
        yoff = y0 - x0 * sinp - y0 * cosp
        def turn(g):
            x = g.pts[:, 0]
            y = g.pts[:, 1]
            return Polygon(np.stack([cosp * x - sinp * y + xoff, sinp * x + cosp * y + yoff], axis=1))

        # (the affine matrix comes from the bounds of the WHOLE geometry: a multi-part geometry turns about its common centre)
        return MultiPolygon([turn(g) for g in geom.geoms]) if isinstance(geom, MultiPolygon) else turn(geom)


affinity = _Affinity()


# --------------------------------------------------------------------------
# duck-typed commonroad obstacle (what has_collision() touches,
# planners/frenet_optimal_planner.py:173,187-189)
# --------------------------------------------------------------------------
class StubObstacle:
    """Rectangle obstacle with per-time-step poses.

    poses: [T,3] array of (x, y, yaw) for absolute time steps 0..T-1; a row of
    NaN means "no state at that step" (state_at_time -> None).
    final_time_step mirrors TrajectoryPrediction.final_time_step (= T-1 when the
    prediction covers steps 1..T-1 after the initial state at step 0).
    """

    def __init__(self, length: float, width: float, poses: np.ndarray, final_time_step: int | None = None):
        poses = np.asarray(poses, dtype=float).reshape(-1, 3)
        self.poses = poses
        hl, hw = length / 2.0, width / 2.0
        # commonroad Rectangle(length, width) centred at the origin, orientation 0
        self.obstacle_shape = SimpleNamespace(
            shapely_object=Polygon([(-hl, -hw), (-hl, hw), (hl, hw), (hl, -hw)]), length=length, width=width
        )
        fts = poses.shape[0] - 1 if final_time_step is None else final_time_step
        self.prediction = SimpleNamespace(final_time_step=fts)

    def state_at_time(self, t: int):
        if t < 0 or t >= self.poses.shape[0] or not np.isfinite(self.poses[t, 0]):
            return None
        x, y, yaw = self.poses[t]
        return SimpleNamespace(position=np.array([x, y]), orientation=float(yaw), time_step=t)


class StubShapeObstacle(StubObstacle):
    """An obstacle whose shape is any polygon-like `shapely_object` (what obstacle_shape.shapely_object may be in the reference,
    frenet_optimal_planner.py:189): a Polygon, or a MultiPolygon for a group of shapes."""

    def __init__(self, shapely_object, poses: np.ndarray, final_time_step: int | None = None):
        super().__init__(1.0, 1.0, poses, final_time_step)
        self.obstacle_shape = SimpleNamespace(shapely_object=shapely_object)
        if isinstance(shapely_object, MultiPolygon):  # commonroad's ShapeGroup exposes its parts
            self.obstacle_shape.shapes = [SimpleNamespace(shapely_object=g) for g in shapely_object.geoms]


def obstacles_from_table(pose: np.ndarray, dims: np.ndarray, final_time_step: int) -> list:
    """pose [T,n,4]=(x,y,yaw,valid), dims [n,2]=(l,w) -> list of StubObstacle."""
    T, n, _ = pose.shape
    out = []
    for j in range(n):
        p = pose[:, j, :3].copy()
        p[pose[:, j, 3] == 0.0] = np.nan
        out.append(StubObstacle(dims[j, 0], dims[j, 1], p, final_time_step))
    return out


# --------------------------------------------------------------------------
# vehicle parameters (VW_VANAGON = commonroad vehicle type 3; values of
# commonroad-vehicle-models parameters_vehicle3 recalled from memory - the
# package is not installed, so they are inputs of every API here, not truths)
# --------------------------------------------------------------------------
def vw_vanagon_params() -> SimpleNamespace:
    return SimpleNamespace(
        l=4.569,
        w=1.844,
        a=1.1489,
        b=1.2859,
        T_f=1.5740,
        T_r=1.5740,
        longitudinal=SimpleNamespace(v_max=41.7, a_max=11.5),
        steering=SimpleNamespace(max=1.023, min=-1.023, v_max=0.4, v_min=-0.4, kappa_dot_max=0.4, kappa_dot_dot_max=20.0),
    )


_installed = False


def install():
    """Register the stub modules and put the reference on sys.path."""
    global _installed
    if _installed:
        return
    cr = types.ModuleType("commonroad")
    cr_s = types.ModuleType("commonroad.scenario")
    cr_ss = types.ModuleType("commonroad.scenario.scenario")
    cr_so = types.ModuleType("commonroad.scenario.obstacle")
    cr_ss.Scenario = type("Scenario", (), {})
    cr_so.Obstacle = type("Obstacle", (), {})
    cr.scenario = cr_s
    cr_s.scenario = cr_ss
    cr_s.obstacle = cr_so
    sh = types.ModuleType("shapely")
    sh_g = types.ModuleType("shapely.geometry")
    sh.Polygon = Polygon
    sh.affinity = affinity
    sh.geometry = sh_g
    sh_g.Polygon = Polygon
    sh_g.MultiPolygon = MultiPolygon
    for name, mod in [
        ("commonroad", cr),
        ("commonroad.scenario", cr_s),
        ("commonroad.scenario.scenario", cr_ss),
        ("commonroad.scenario.obstacle", cr_so),
        ("shapely", sh),
        ("shapely.geometry", sh_g),
    ]:
        sys.modules.setdefault(name, mod)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def load_reference():
    """Return a namespace with the reference classes used by the generators."""
    install()
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # `cost_type is "WX1"` SyntaxWarning in cost_function.py:6
        from planners.common.cost.cost_function import CostFunction
        from planners.common.geometry.cubic_spline import CubicSpline1D, CubicSpline2D
        from planners.common.geometry.polynomial import QuarticPolynomial, QuinticPolynomial
        from planners.common.scenario.frenet import FrenetState, FrenetTrajectory, State
        from planners.common.vehicle.vehicle import Vehicle
        from planners.fiss_planner import FissPlanner, FissPlannerSettings
        from planners.fiss_plus_planner import FissPlusPlanner, FissPlusPlannerSettings
        from planners.fop_plus_planner import FopPlusPlanner
        from planners.frenet_optimal_planner import FrenetOptimalPlanner, FrenetOptimalPlannerSettings, Stats
    return SimpleNamespace(**{k: v for k, v in locals().items() if not k.startswith("_") and k != "warnings"})
