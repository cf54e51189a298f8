"""CPU: obstacle lists -> pose tables (fiss_plus_planner_amd/obstacles.py), the duck-typed surface has_collision() touches
(reference frenet_optimal_planner.py:168-195)."""
import os
import sys
import warnings
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from refshim import Polygon, StubObstacle  # noqa: E402

from fiss_plus_planner_amd.obstacles import buffer_circle_ring, flatten_obstacles, obstacles_fingerprint, shape_columns  # noqa: E402


def test_flatten_walks_state_at_time_and_the_first_obstacles_horizon():
    poses = np.tile([3.0, 4.0, 0.5], (6, 1))
    poses[2] = np.nan  # no state at step 2
    a = StubObstacle(4.0, 2.0, poses, final_time_step=4)
    b = StubObstacle(1.0, 0.5, np.tile([7.0, 8.0, -1.0], (3, 1)), final_time_step=99)  # only obstacles[0] sets the horizon (:173)
    tab = flatten_obstacles([a, b])
    assert tab.final_time_step == 4 and tab.pose.shape == (4, 2, 4)
    np.testing.assert_array_equal(tab.dims, [[4.0, 2.0], [1.0, 0.5]])
    np.testing.assert_array_equal(tab.pose[:, 0, 3], [1, 1, 0, 1])
    np.testing.assert_array_equal(tab.pose[:, 1, 3], [1, 1, 1, 0])     # b has no state at step 3
    np.testing.assert_array_equal(tab.pose[0, 0, :3], [3.0, 4.0, 0.5])


def test_static_obstacle_first_raises_like_the_reference():
    static = SimpleNamespace(obstacle_shape=SimpleNamespace(length=1.0, width=1.0), state_at_time=lambda t: None)
    with pytest.raises(AttributeError):
        flatten_obstacles([static])  # obstacles[0].prediction does not exist (:173)


def _poly_obstacle(coords, pose=(10.0, 20.0, 0.3)):
    ob = StubObstacle(1.0, 1.0, np.tile(pose, (3, 1)), 2)
    ob.obstacle_shape = SimpleNamespace(shapely_object=Polygon(coords))
    return ob


def test_off_centre_rectangle_is_displaced_by_its_unrotated_offset():
    """affinity.rotate(origin='center') turns the translated polygon about ITS bounding-box centre (:162-166): a rectangle
    [1, 5] x [0, 2] at position p behaves like a centred 4 x 2 rectangle at p + (3, 1), whatever the yaw."""
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        tab = flatten_obstacles([_poly_obstacle([(1, 0), (5, 0), (5, 2), (1, 2)])])
    np.testing.assert_array_equal(tab.dims, [[4.0, 2.0]])
    np.testing.assert_array_equal(tab.pose[0, 0], [13.0, 21.0, 0.3, 1.0])


def test_convex_shapes_become_polygon_columns_about_their_bounding_box_centre():
    """obstacle_shape.shapely_object is ANY polygon in the reference (:189-191): a convex one is one column - counter-clockwise ring
    relative to the centre of its bounding box (the point affinity.rotate(origin='center') turns it about), dims = that box."""
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # (no bounding-box fallback, no warning)
        tab = flatten_obstacles([_poly_obstacle([(-2, -1), (2, -1), (0, 3)])])  # a triangle, bounding box [-2, 2] x [-1, 3]
    np.testing.assert_array_equal(tab.dims, [[4.0, 4.0]])
    np.testing.assert_array_equal(tab.pose[0, 0, :2], [10.0, 21.0])
    np.testing.assert_array_equal(tab.nvert, [3])
    np.testing.assert_array_equal(tab.poly[0, :3], [(-2, -2), (2, -2), (0, 2)])
    tab = flatten_obstacles([_poly_obstacle([(1, 0), (0, -1), (-1, 0), (0, 1)])])   # a diamond given CLOCKWISE: the ring is reversed
    np.testing.assert_array_equal(tab.nvert, [4])
    v = tab.poly[0, :4]
    assert np.sum(v[:, 0] * np.roll(v[:, 1], -1) - v[:, 1] * np.roll(v[:, 0], -1)) > 0
    np.testing.assert_array_equal(tab.dims, [[2.0, 2.0]])


def test_rectangles_and_polygons_share_a_table():
    rect = StubObstacle(4.0, 2.0, np.tile([1.0, 2.0, 0.1], (3, 1)), 2)
    tab = flatten_obstacles([rect, _poly_obstacle([(-2, -1), (2, -1), (0, 3)])])
    np.testing.assert_array_equal(tab.nvert, [0, 3])
    assert tab.poly.shape == (2, 3, 2)
    only_rects = flatten_obstacles([rect])
    assert only_rects.nvert is None and only_rects.poly is None


def test_a_circle_is_the_polygon_shapely_makes_of_it():
    ring = buffer_circle_ring(1.5, 0.25, -0.5)
    assert ring.shape == (64, 2)
    np.testing.assert_allclose(ring[0], [1.75, -0.5])
    np.testing.assert_allclose(ring[1], [0.25 + 1.5 * np.cos(np.pi / 32), -0.5 - 1.5 * np.sin(np.pi / 32)], rtol=0, atol=1e-15)  # clockwise
    assert ring[16, 0] == 0.25 and ring[32, 1] == -0.5  # snapped quarter points
    ob = _poly_obstacle([(0, 0), (1, 0), (0, 1)])
    ob.obstacle_shape = SimpleNamespace(radius=1.5, center=np.array([0.25, -0.5]))  # no shapely_object: the library builds the ring
    tab = flatten_obstacles([ob])
    np.testing.assert_array_equal(tab.nvert, [64])
    np.testing.assert_array_equal(tab.pose[0, 0, :2], [10.25, 19.5])
    # ONE radius convention for every path that builds a circle's polygon itself (obstacles.CIRCLE_BUFFER_FACTOR: commonroad-io's
    # Point(center).buffer(radius / 2), recalled): a radius-only shape object and the XML reader give the same obstacle
    from fiss_plus_planner_amd.obstacles import CIRCLE_BUFFER_FACTOR, shape_columns
    r = CIRCLE_BUFFER_FACTOR * 1.5
    np.testing.assert_allclose(tab.dims, [[2 * r, 2 * r]], rtol=0, atol=1e-15)
    np.testing.assert_allclose(np.hypot(tab.poly[0, :, 0], tab.poly[0, :, 1]), r, rtol=0, atol=1e-15)
    full = shape_columns(ob.obstacle_shape, circle_buffer_factor=1.0)
    np.testing.assert_allclose(full[0][:2], [3.0, 3.0], rtol=0, atol=1e-15)


def test_self_intersecting_rings_and_mismatched_updates_raise():
    import pytest

    from fiss_plus_planner_amd.obstacles import ObstacleTable, shape_columns

    bowtie = SimpleNamespace(vertices=np.array([(0.0, 0.0), (2.0, 2.0), (2.0, 0.0), (0.0, 2.0)]))
    with pytest.raises(ValueError, match="no area|simple polygon"):   # (a symmetric bow tie: the signed areas of its lobes cancel)
        shape_columns(bowtie)
    with pytest.raises(ValueError, match="simple polygon"):
        shape_columns(SimpleNamespace(vertices=np.array([(0.0, 0.0), (3.0, 2.0), (3.0, 0.0), (0.0, 1.0)])))   # a lopsided one
    star = SimpleNamespace(vertices=np.array([(0, 3), (1.8, -2.4), (-2.9, 0.9), (2.9, 0.9), (-1.8, -2.4)], dtype=float))  # pentagram
    with pytest.raises(ValueError, match="simple polygon"):
        shape_columns(star)
    tab = flatten_obstacles([_poly_obstacle([(0, 0), (1, 0), (0, 1)])])
    with pytest.raises(ValueError, match="come together"):
        tab.update(poly=tab.poly.copy())
    with pytest.raises(ValueError, match="do not match"):
        tab.update(pose=np.zeros((tab.pose.shape[0], 2, 4)), dims=np.ones((2, 2)))   # two columns now, the rings still describe one
    v = tab.version
    tab.update(pose=np.zeros((tab.pose.shape[0], 2, 4)), dims=np.ones((2, 2)), poly=np.zeros((2, 3, 2)), nvert=np.zeros(2, dtype=np.int32))
    assert tab.version == v + 1


def test_non_convex_shapes_and_groups_are_cut_into_convex_pieces_about_the_common_centre():
    L = [(0, 0), (4.2, 0), (4.2, 1.2), (1.3, 1.2), (1.3, 3.0), (0, 3.0)]
    tab = flatten_obstacles([_poly_obstacle(L)])
    assert tab.pose.shape[1] == 2 and list(tab.nvert) == [4, 4]          # two convex quadrilaterals, same poses
    np.testing.assert_array_equal(tab.pose[:, 0], tab.pose[:, 1])
    np.testing.assert_array_equal(tab.pose[0, 0, :2], [10.0 + 2.1, 20.0 + 1.5])
    area = 0.0
    for k in range(2):
        v = tab.poly[k, :tab.nvert[k]]
        a2 = np.sum(v[:, 0] * np.roll(v[:, 1], -1) - v[:, 1] * np.roll(v[:, 0], -1))
        assert a2 > 0
        area += 0.5 * a2
        assert tab.dims[k, 0] >= 2 * np.abs(v[:, 0]).max() and tab.dims[k, 1] >= 2 * np.abs(v[:, 1]).max()
    np.testing.assert_allclose(area, 4.2 * 1.2 + 1.3 * 1.8)
    # the pieces' vertices are the shape's own vertices, relative to the centre of the WHOLE shape's bounding box
    want = {(x - 2.1, y - 1.5) for x, y in L}
    got = {(float(x), float(y)) for k in range(2) for x, y in tab.poly[k, :tab.nvert[k]]}
    assert got <= {(float(np.float64(x)), float(np.float64(y))) for x, y in want}
    group = _poly_obstacle(L)
    parts = [Polygon([(-3.0, -0.5), (-1.0, -0.5), (-1.0, 0.5), (-3.0, 0.5)]), Polygon([(1.0, -0.8), (3.0, 0.0), (1.0, 0.8)])]
    group.obstacle_shape = SimpleNamespace(shapes=[SimpleNamespace(shapely_object=p) for p in parts])
    tab = flatten_obstacles([group])
    assert list(tab.nvert) == [4, 3]
    np.testing.assert_array_equal(tab.pose[0, 0, :2], [10.0, 20.0])       # the group's box is [-3, 3] x [-0.8, 0.8]: centred
    np.testing.assert_array_equal(tab.poly[0, :4], parts[0].pts)


def test_random_simple_polygons_partition_into_convex_pieces_of_the_same_area():
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = int(rng.integers(4, 15))
        while True:
            ang = np.sort(rng.uniform(0, 2 * np.pi, n))
            if np.max(np.diff(np.concatenate([ang, [ang[0] + 2 * np.pi]]))) < np.pi - 0.05:
                break
        r = rng.uniform(0.5, 2.0, n)
        ring = np.stack([r * np.cos(ang), r * np.sin(ang)], axis=1)
        cols = shape_columns(SimpleNamespace(shapely_object=Polygon(ring)))
        area = sum(0.5 * np.sum(c[4][:, 0] * np.roll(c[4][:, 1], -1) - c[4][:, 1] * np.roll(c[4][:, 0], -1)) for c in cols)
        want = 0.5 * np.sum(ring[:, 0] * np.roll(ring[:, 1], -1) - ring[:, 1] * np.roll(ring[:, 0], -1))
        np.testing.assert_allclose(area, want, rtol=1e-12)


def test_fingerprint_follows_the_objects_not_the_list():
    obs = [StubObstacle(4, 2, np.zeros((5, 3))) for _ in range(3)]
    fp = obstacles_fingerprint(obs)
    assert obstacles_fingerprint(list(obs)) == fp                 # same objects, new list
    assert obstacles_fingerprint(obs[::-1]) != fp                 # order matters (obstacles[0] sets the horizon)
    assert obstacles_fingerprint([StubObstacle(4, 2, np.zeros((5, 3))) for _ in range(3)]) != fp
    obs[0].prediction.final_time_step = 2
    assert obstacles_fingerprint(obs) != fp


def test_self_crossing_rings_of_more_than_four_vertices_raise():
    """The ears of a self-crossing ring are all positively oriented and sum to the ring's signed area: only a direct edge-pair test sees
    it (8 % of random crossing 5-8-gons passed the area comparison and came back as convex columns)."""
    from fiss_plus_planner_amd.obstacles import _ring_is_simple

    seven = np.array([[-2.46, 2.0], [0.11, -2.23], [0.15, 0.26], [-0.02, -1.76], [-0.39, 2.23], [-0.7, 0.02], [2.58, -1.6]])
    assert not _ring_is_simple(seven) and not _ring_is_simple(seven[::-1])
    with pytest.raises(ValueError, match="simple polygon"):
        shape_columns(SimpleNamespace(vertices=seven))
    six = np.array([(0.0, 0.0), (4.0, 0.0), (4.0, 3.0), (1.0, -1.0), (0.5, 3.0), (0.0, 3.0)])  # one edge dips through the base
    with pytest.raises(ValueError, match="simple polygon"):
        shape_columns(SimpleNamespace(vertices=six))
    # touching is crossing too (shapely: invalid ring): a vertex ON a non-adjacent edge, a spike folded back, a repeated vertex
    assert not _ring_is_simple(np.array([(0.0, 0.0), (4.0, 0.0), (4.0, 2.0), (2.0, 0.0), (0.0, 2.0)]))
    assert not _ring_is_simple(np.array([(0.0, 0.0), (4.0, 0.0), (2.0, 0.0), (2.0, 2.0)]))
    assert not _ring_is_simple(np.array([(0.0, 0.0), (4.0, 0.0), (4.0, 0.0), (2.0, 2.0)]))
    # random rings: a brute-force segment test agrees, and every ring the reader accepts is simple
    rng = np.random.default_rng(5)
    n_cross = 0
    for _ in range(400):
        n = int(rng.integers(5, 9))
        v = rng.uniform(-3, 3, (n, 2)).round(2)

        def crosses(v):
            m = len(v)
            for i in range(m):
                for j in range(i + 2, m):
                    if i == 0 and j == m - 1:
                        continue
                    p, q, r, s_ = v[i], v[(i + 1) % m], v[j], v[(j + 1) % m]
                    d = lambda a, b, c: (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])  # noqa: E731
                    if d(p, q, r) * d(p, q, s_) < 0 and d(r, s_, p) * d(r, s_, q) < 0:
                        return True
            return False

        if crosses(v):
            n_cross += 1
            assert not _ring_is_simple(v)
            with pytest.raises(ValueError):
                shape_columns(SimpleNamespace(vertices=v))
    assert n_cross > 200
    # simple non-convex rings still pass: an L and a star
    ell = np.array([(0, 0), (4, 0), (4, 1), (1, 1), (1, 3), (0, 3)], dtype=float)
    assert _ring_is_simple(ell) and len(shape_columns(SimpleNamespace(vertices=ell))) >= 2
    ang = np.linspace(0, 2 * np.pi, 10, endpoint=False)
    star = np.stack([np.where(np.arange(10) % 2, 1.0, 2.5) * np.cos(ang), np.where(np.arange(10) % 2, 1.0, 2.5) * np.sin(ang)], axis=1)
    assert _ring_is_simple(star) and len(shape_columns(SimpleNamespace(vertices=star))) >= 3
