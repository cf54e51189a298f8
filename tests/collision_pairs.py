"""Box pairs at the edge of contact, for the collision audit (tests/test_collision_exact.py).

A pair = the ego box A = (l, w, x, y, yaw) and an obstacle box B.  B is placed so that one corner of one box lies on an edge of
the other (computed in fp64: the corner lands within a rounding error of the edge line), then pushed k units in the last place of
its centre coordinates along the edge's outward normal, k in [-K, K]: the pair is within a few ulp of touching, on either side.
Which side it really is on is NOT known by construction - the exact predicate says.  Vectorised: a million pairs in a second.
"""
import numpy as np

_LOC = np.array([[0.5, 0.5], [0.5, -0.5], [-0.5, -0.5], [-0.5, 0.5]])  # corners in units of (l, w)


def _corners(l, w, x, y, yaw):
    """[n, 4, 2] fp64 corners (plain rotation about the centre; only used to aim, never as the truth)."""
    c, s = np.cos(yaw)[:, None], np.sin(yaw)[:, None]
    lx, ly = _LOC[None, :, 0] * l[:, None], _LOC[None, :, 1] * w[:, None]
    return np.stack([x[:, None] + c * lx - s * ly, y[:, None] + s * lx + c * ly], axis=2)


def near_contact(a, rng, per, K=4, scale=1):
    """a: [m, 5] ego boxes -> (b [m * per, 5] obstacle boxes, k [m * per] offsets in units of `scale` ulp, ego index [m * per])."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 5)
    m = len(a)
    n = m * per
    ego = np.repeat(np.arange(m), per)
    A = a[ego]
    VA = _corners(A[:, 0], A[:, 1], A[:, 2], A[:, 3], A[:, 4])
    cen = A[:, 2:4]
    lb, wb, thb = rng.uniform(2.0, 7.5, n), rng.uniform(1.2, 2.6, n), rng.uniform(-np.pi, np.pi, n)
    cb, sb = np.cos(thb), np.sin(thb)
    ks = rng.integers(-K, K + 1, n)
    rows = np.arange(n)
    # --- variant 1: a corner of B on an edge of A
    e = rng.integers(0, 4, n)
    u = rng.uniform(0.02, 0.98, n)[:, None]
    p0, p1 = VA[rows, e], VA[rows, (e + 1) % 4]
    p = p0 * (1 - u) + p1 * u
    ed = p1 - p0
    nrm1 = np.stack([ed[:, 1], -ed[:, 0]], axis=1)
    nrm1 *= np.where(np.einsum("ij,ij->i", nrm1, p - cen) < 0, -1.0, 1.0)[:, None]
    nrm1 /= np.hypot(nrm1[:, 0], nrm1[:, 1])[:, None]
    VB0 = _corners(lb, wb, np.zeros(n), np.zeros(n), thb)  # B's corners about its own centre
    q = VB0[rows, np.argmin(np.einsum("ikj,ij->ik", VB0, nrm1), axis=1)]  # the corner that points most against the normal
    c1 = p - q
    # --- variant 2: a corner of A on an edge of B
    e2 = rng.integers(0, 4, n)
    nl = np.array([[1.0, 0.0], [0.0, -1.0], [-1.0, 0.0], [0.0, 1.0]])[e2]
    half = np.where(e2 % 2 == 0, lb / 2, wb / 2)
    other = np.where(e2 % 2 == 0, wb / 2, lb / 2)
    nrm_b = np.stack([cb * nl[:, 0] - sb * nl[:, 1], sb * nl[:, 0] + cb * nl[:, 1]], axis=1)  # outward normal of B's edge (faces A)
    tng_b = np.stack([-nrm_b[:, 1], nrm_b[:, 0]], axis=1)
    va = VA[rows, np.argmin(np.einsum("ikj,ij->ik", VA, nrm_b), axis=1)]
    c2 = va - nrm_b * half[:, None] - tng_b * (rng.uniform(-0.96, 0.96, n) * other)[:, None]
    pick = rng.random(n) < 0.5
    c = np.where(pick[:, None], c1, c2)
    nrm = np.where(pick[:, None], nrm1, -nrm_b)
    ulp = np.spacing(np.maximum(np.maximum(np.abs(c[:, 0]), np.abs(c[:, 1])), 1.0))
    b = np.stack([lb, wb, c[:, 0] + ks * scale * ulp * nrm[:, 0], c[:, 1] + ks * scale * ulp * nrm[:, 1], thb], axis=1)
    return b, ks, ego
