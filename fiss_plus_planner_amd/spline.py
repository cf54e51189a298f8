"""Host-side reference-line spline (natural cubic spline by arclength).

API surface of the reference's planners/common/geometry/cubic_spline.py
(CubicSpline2D :145-232: ``.s``, ``calc_position``, ``calc_yaw``,
``calc_curvature``) because ``generate_frenet_frame`` hands the spline object
back to the caller.  The construction is ours: a batched Thomas (tridiagonal)
solve vectorised over many frames at once instead of one dense
``np.linalg.solve`` per axis, producing the packed coefficient table the device
kernels stage into LDS:

    knots [F, NX]          cumulative chord length, padded with +inf
    coef  [F, 8, NX]       rows ax, bx, cx, dx, ay, by, cy, dy (segment i in column i)

Natural boundary conditions c_0 = c_{n-1} = 0 as in reference :118-142.
"""
from __future__ import annotations

import bisect
import math

import numpy as np


def natural_spline_coefs(knots: np.ndarray, y: np.ndarray, n: np.ndarray | None = None) -> np.ndarray:
    """Batched natural cubic spline.  knots, y: [F, NX]; n: [F] valid counts (default NX).

    Returns coef [F, 4, NX] (a, b, c, d); entries at and after column n-1 of b, d are 0.
    """
    knots = np.asarray(knots, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    F, NX = y.shape
    n = np.full(F, NX, dtype=np.int64) if n is None else np.asarray(n, dtype=np.int64)
    col = np.arange(NX)[None, :]
    h = np.zeros((F, NX))
    h[:, :-1] = np.diff(knots, axis=1)
    seg = col < (n[:, None] - 1)  # valid segments
    h = np.where(seg, h, 1.0)
    if np.any(h[seg] < 0):
        raise ValueError("x coordinates must be sorted in ascending order")
    a = np.where(col < n[:, None], y, 0.0)
    # interior rows i=1..n-2:  h[i-1] c[i-1] + 2 (h[i-1]+h[i]) c[i] + h[i] c[i+1] = rhs[i]
    interior = (col >= 1) & (col < (n[:, None] - 1))
    da = np.zeros((F, NX))
    da[:, :-1] = np.diff(a, axis=1)
    slope = np.where(seg, da / h, 0.0)
    rhs = np.zeros((F, NX))
    rhs[:, 1:] = 3.0 * (slope[:, 1:] - slope[:, :-1])
    rhs = np.where(interior, rhs, 0.0)
    lower = np.zeros((F, NX)); lower[:, 1:] = h[:, :-1]
    diag = np.ones((F, NX)); diag[:, 1:] = 2.0 * (h[:, :-1] + h[:, 1:])
    upper = h.copy()
    lower = np.where(interior, lower, 0.0)
    diag = np.where(interior, diag, 1.0)
    upper = np.where(interior, upper, 0.0)
    # Thomas sweep, vectorised over frames
    cp = np.zeros((F, NX)); dp = np.zeros((F, NX))
    cp[:, 0] = upper[:, 0] / diag[:, 0]
    dp[:, 0] = rhs[:, 0] / diag[:, 0]
    for i in range(1, NX):
        den = diag[:, i] - lower[:, i] * cp[:, i - 1]
        cp[:, i] = upper[:, i] / den
        dp[:, i] = (rhs[:, i] - lower[:, i] * dp[:, i - 1]) / den
    c = np.zeros((F, NX))
    c[:, NX - 1] = dp[:, NX - 1]
    for i in range(NX - 2, -1, -1):
        c[:, i] = dp[:, i] - cp[:, i] * c[:, i + 1]
    c = np.where(col < n[:, None], c, 0.0)
    cn = np.zeros((F, NX)); cn[:, :-1] = c[:, 1:]
    an = np.zeros((F, NX)); an[:, :-1] = a[:, 1:]
    d = np.where(seg, (cn - c) / (3.0 * h), 0.0)
    b = np.where(seg, 1.0 / h * (an - a) - h / 3.0 * (2.0 * c + cn), 0.0)
    return np.stack([a, b, c, d], axis=1)


def build_frames(points: np.ndarray, n: np.ndarray | None = None):
    """points [F, NX, 2] centerline vertices (rows >= n[f] ignored) -> (knots [F,NX], coef [F,8,NX]).

    knots beyond n[f] are +inf so that a binary search never lands there.
    """
    pts = np.asarray(points, dtype=np.float64)
    if pts.ndim == 2:
        pts = pts[None]
    F, NX, _ = pts.shape
    n = np.full(F, NX, dtype=np.int64) if n is None else np.asarray(n, dtype=np.int64)
    col = np.arange(NX)[None, :]
    ds = np.zeros((F, NX))
    ds[:, 1:] = np.hypot(np.diff(pts[:, :, 0], axis=1), np.diff(pts[:, :, 1], axis=1))
    ds = np.where(col < n[:, None], ds, 0.0)
    knots = np.cumsum(ds, axis=1)  # [0, cumsum(ds)] like reference :162-168
    cx = natural_spline_coefs(knots, pts[:, :, 0], n)
    cy = natural_spline_coefs(knots, pts[:, :, 1], n)
    knots = np.where(col < n[:, None], knots, np.inf)
    return knots, np.concatenate([cx, cy], axis=1)


class _Spline1D:
    """View of one axis: attributes a, b, c, d, x like the reference's CubicSpline1D."""

    def __init__(self, knots, coef):
        self.x = list(knots)
        self.nx = len(knots)
        self.a, self.b, self.c, self.d = coef[0], coef[1][:-1], coef[2], coef[3][:-1]
        self._coef = coef

    def _seg(self, x):
        if x < self.x[0] or x > self.x[-1]:
            return None
        return bisect.bisect(self.x, x) - 1

    def calc_position(self, x):
        i = self._seg(x)
        if i is None:
            return None
        dx = x - self.x[i]
        return self.a[i] + self.b[i] * dx + self.c[i] * dx ** 2.0 + self.d[i] * dx ** 3.0

    def calc_first_derivative(self, x):
        i = self._seg(x)
        if i is None:
            return None
        dx = x - self.x[i]
        return self.b[i] + 2.0 * self.c[i] * dx + 3.0 * self.d[i] * dx ** 2.0

    def calc_second_derivative(self, x):
        i = self._seg(x)
        if i is None:
            return None
        dx = x - self.x[i]
        return 2.0 * self.c[i] + 6.0 * self.d[i] * dx


class CubicSpline2D:
    """Reference line through centerline vertices, parameterised by chord length."""

    def __init__(self, x, y, tables=None):
        """tables = (knots [nx], coef [8, nx]) built elsewhere (e.g. on the GPU by fp_frames_build); default: host build."""
        if tables is None:
            pts = np.column_stack([np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)])
            knots, coef = build_frames(pts[None])
            tables = (knots[0], coef[0])
        self.knots = np.ascontiguousarray(tables[0], dtype=np.float64)
        self.coef = np.ascontiguousarray(tables[1], dtype=np.float64)  # [8, nx]
        self.s = list(self.knots)
        self.ds = np.diff(self.knots)
        self.sx = _Spline1D(self.knots, self.coef[0:4])
        self.sy = _Spline1D(self.knots, self.coef[4:8])

    def calc_position(self, s):
        return self.sx.calc_position(s), self.sy.calc_position(s)

    def calc_yaw(self, s):
        dx = self.sx.calc_first_derivative(s)
        dy = self.sy.calc_first_derivative(s)
        return math.atan2(dy, dx)

    def calc_curvature(self, s):
        dx = self.sx.calc_first_derivative(s)
        ddx = self.sx.calc_second_derivative(s)
        dy = self.sy.calc_first_derivative(s)
        ddy = self.sy.calc_second_derivative(s)
        return (ddy * dx - ddx * dy) / ((dx ** 2 + dy ** 2) ** (3 / 2))

    def sample(self, s: np.ndarray):
        """Vectorised (x, y, yaw, kappa) at in-range arclengths."""
        s = np.asarray(s, dtype=np.float64)
        i = np.clip(np.searchsorted(self.knots, s, side="right") - 1, 0, len(self.knots) - 2)
        dx = s - self.knots[i]
        c = self.coef
        px = c[0, i] + c[1, i] * dx + c[2, i] * dx ** 2 + c[3, i] * dx ** 3
        py = c[4, i] + c[5, i] * dx + c[6, i] * dx ** 2 + c[7, i] * dx ** 3
        d1x = c[1, i] + 2 * c[2, i] * dx + 3 * c[3, i] * dx ** 2
        d1y = c[5, i] + 2 * c[6, i] * dx + 3 * c[7, i] * dx ** 2
        d2x = 2 * c[2, i] + 6 * c[3, i] * dx
        d2y = 2 * c[6, i] + 6 * c[7, i] * dx
        return px, py, np.arctan2(d1y, d1x), (d2y * d1x - d2x * d1y) / (d1x ** 2 + d1y ** 2) ** 1.5
