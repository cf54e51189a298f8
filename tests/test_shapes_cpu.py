"""CPU: obstacle shapes that are not rectangles - the oracle (and the product's shape -> column decomposition feeding it) against G12,
the verdicts and plans of the imported reference on circles, triangles, rotated rectangles, a non-convex L, an off-centre pentagon and
a group of two shapes (tests/golden/gen_golden.py:g12; has_collision hands obstacle_shape.shapely_object - any polygon - to
construct_polygon / Polygon.intersects, frenet_optimal_planner.py:186-193)."""
import numpy as np
import pytest

from conftest import load_golden
from shapes_util import g12_batch, with_random_shapes

NAMES = ["p555", "p555b", "p997", "p997b"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_collision_verdicts_match_the_reference_on_g12(oracle, name):
    g = load_golden("g12_shapes.npz")
    b = g12_batch(g, name)
    assert b.obs_nvert is not None and b.n_obs == 9  # 7 obstacles: the L and the group are two convex pieces each
    for e, p in enumerate(oracle.problems_from_batch(b)):
        cost, flags = p.dense_tables()
        np.testing.assert_allclose(cost, g[f"{name}_cost"][e], rtol=0, atol=1e-9)
        np.testing.assert_array_equal((flags & 4) != 0, g[f"{name}_coll"][e])


@pytest.mark.parametrize("name", NAMES)
def test_g12_shapes_decide_verdicts_their_bounding_boxes_would_not(oracle, name):
    """The fixture carries the reference's verdicts with every shape replaced by its bounding box (the build's behaviour before ABI
    12): they differ from the true ones, and the oracle on the boxed batch reproduces THEM - so the polygons are what is tested."""
    g = load_golden("g12_shapes.npz")
    coll, box = g[f"{name}_coll"], g[f"{name}_coll_box"]
    assert (coll != box).sum() >= 20 and not (coll & ~box).any()   # a box can only add collisions
    bb = g12_batch(g, name, boxes=True)
    for e, p in enumerate(oracle.problems_from_batch(bb)):
        _, flags = p.dense_tables()
        np.testing.assert_array_equal((flags & 4) != 0, box[e])


@pytest.mark.parametrize("kind", ["FOP", "FOP+", "FISS", "FISS+"])
@pytest.mark.parametrize("name", NAMES)
def test_oracle_plans_match_the_reference_on_g12(oracle, name, kind):
    g = load_golden("g12_shapes.npz")
    key = f"{name}_{kind}"
    b = g12_batch(g, name, kind)
    for e, p in enumerate(oracle.problems_from_batch(b)):
        found = bool(g[f"{key}_found"][e])
        if kind in ("FOP", "FOP+"):
            r = p.fop_plan() if kind == "FOP" else p.fopplus_plan()
            assert (r.best_idx >= 0) == found
            if found:
                assert r.best_idx == g[f"{key}_flat"][e] and abs(r.best_cost - g[f"{key}_cost"][e]) < 1e-9
        elif kind == "FISS":
            r = p.fiss_plan()
            assert (r.best_ijk[0] >= 0) == found
            if found:
                np.testing.assert_array_equal(r.best_ijk, g[f"{key}_idx"][e])
                assert abs(r.best_cost - g[f"{key}_cost"][e]) < 1e-9
        else:
            r = p.fissplus_plan()
            assert (not np.isnan(r.best_cost)) == found
            if found:
                assert abs(r.best_cost - g[f"{key}_cost"][e]) < 1e-9
                np.testing.assert_allclose(r.end_state, g[f"{key}_end"][e], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(r.stats, g[f"{key}_stats"][e])


def test_ring_predicate_known_answers(oracle):
    """Hand-derived: a 4 x 2 ego box at the origin against the triangle (-2,-2) (2,-2) (0,2) (ring about its bounding-box centre).
    Its left edge runs from (p - 2, -2) to (p, 2): the ego's corner (2, -1) lies ON it at p = 3.5 (touching counts), clear of it
    beyond; the float and the exact predicate agree on these dyadic coordinates."""
    tri = np.array([(-2.0, -2.0), (2.0, -2.0), (0.0, 2.0)])
    ego = (4.0, 2.0, 0.0, 0.0, 0.0)
    for p, want in ((3.4, True), (3.5, True), (3.5 + 2.0 ** -40, False), (3.6, False), (-3.5, True), (-3.5 - 2.0 ** -40, False)):
        for exact in (False, True):
            assert oracle.box_ring_intersect(ego, tri, (p, 0.0, 0.0), exact=exact) is want, (p, exact)
    # containment either way is an intersection: a small triangle inside the box, a big one around it
    assert oracle.box_ring_intersect(ego, 0.1 * tri, (0.5, 0.2, 1.0))
    assert oracle.box_ring_intersect(ego, 10.0 * tri, (0.0, 0.0, 2.0))
    # rotation is about the pose position: a quarter turn of the triangle about (3.5, 0) puts its apex at x = 1.5, inside the box
    res, world = oracle.box_ring_intersect(ego, tri, (3.5, 0.0, np.pi / 2), world=True)
    assert res
    np.testing.assert_allclose(world, [(5.5, -2.0), (5.5, 2.0), (1.5, 0.0)], rtol=0, atol=1e-15)
    # a 4-vertex ring that IS a rectangle gives the rectangle's verdicts
    rect = np.array([(-1.5, -1.0), (1.5, -1.0), (1.5, 1.0), (-1.5, 1.0)])
    rng = np.random.default_rng(3)
    for _ in range(300):
        pose = (rng.uniform(-5, 5), rng.uniform(-4, 4), rng.uniform(-np.pi, np.pi))
        assert oracle.box_ring_intersect(ego, rect, pose) == oracle.boxes_intersect(ego, (3.0, 2.0, *pose))


def test_random_polygon_scenes_keep_rectangles_verdicts_where_nothing_changed(oracle):
    """with_random_shapes turns ~60 % of the columns into polygons INSIDE their old rectangles: collisions can only disappear."""
    from fiss_plus_planner_amd import synth

    base = synth.make_batch(4, 5, 5, 5, 10, 60, True, 321)
    shaped = with_random_shapes(base, 7)
    assert (shaped.obs_nvert > 0).any() and (shaped.obs_nvert == 0).any()
    for p0, p1 in zip(oracle.problems_from_batch(base), oracle.problems_from_batch(shaped)):
        c0 = (p0.dense_tables()[1] & 4) != 0
        c1 = (p1.dense_tables()[1] & 4) != 0
        assert not (c1 & ~c0).any()


def test_random_non_convex_shapes_against_the_stand_in_geometry(oracle):
    """The whole chain on random simple (star-shaped) polygons at random poses: obstacles.shape_columns (bounding-box centre, convex
    pieces) + the oracle's ring predicate per piece, OR-ed, against construct_polygon + Polygon.intersects on the UNDIVIDED ring in the
    goldens' stand-in geometry (translate, rotate about the bounding-box centre, exact general intersection)."""
    import os
    import sys
    from types import SimpleNamespace

    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    import refshim
    from fiss_plus_planner_amd.obstacles import shape_columns

    rng = np.random.default_rng(12)
    veh = refshim.Polygon([(-2.0, -1.0), (2.0, -1.0), (2.0, 1.0), (-2.0, 1.0)])
    n_hit = n_nonconvex = 0
    for case in range(250):
        n = int(rng.integers(5, 12))
        while True:
            ang = np.sort(rng.uniform(0, 2 * np.pi, n))
            if np.max(np.diff(np.concatenate([ang, [ang[0] + 2 * np.pi]]))) < np.pi - 0.05:
                break
        r = rng.uniform(0.6, 3.0, n)
        ring = np.stack([r * np.cos(ang), r * np.sin(ang)], axis=1) + rng.uniform(-2, 2, 2)   # (off-centre on purpose)
        shape = refshim.Polygon(ring)
        n_nonconvex += not shape.convex
        cols = shape_columns(SimpleNamespace(shapely_object=shape))
        for _ in range(4):
            ex, ey, eyaw = rng.uniform(-6, 6), rng.uniform(-6, 6), rng.uniform(-np.pi, np.pi)
            ox, oy, oyaw = rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-np.pi, np.pi)
            ego_poly = refshim.affinity.rotate(refshim.affinity.translate(veh, ex, ey), eyaw, use_radians=True)
            obs_poly = refshim.affinity.rotate(refshim.affinity.translate(shape, ox, oy), oyaw, use_radians=True)
            want = ego_poly.intersects_general_exact(obs_poly)
            got = any(oracle.box_ring_intersect((4.0, 2.0, ex, ey, eyaw), c[4], (ox + c[2], oy + c[3], oyaw)) for c in cols)
            if got != want:  # only a pair within rounding of contact may differ: decide it by the exact piecewise predicate
                got = any(oracle.box_ring_intersect((4.0, 2.0, ex, ey, eyaw), c[4], (ox + c[2], oy + c[3], oyaw), exact=True) for c in cols)
            assert got == want, (case, ring.tolist(), (ex, ey, eyaw), (ox, oy, oyaw))
            n_hit += want
    assert n_nonconvex > 100 and 100 < n_hit < 900


def test_rings_that_are_their_rectangles_become_rectangle_columns():
    """A 4-vertex ring equal to the corners of its column's obs_dims rectangle is the same polygon: ProblemBatch turns it into a
    rectangle column (nvert 0), and a scene made of such rings only carries no polygon table at all (it takes the rectangle-only kernel
    instances).  Exact comparison: a ring one ulp off, a clockwise ring, a rotated ring stay polygon columns."""
    from fiss_plus_planner_amd import synth
    from fiss_plus_planner_amd.batch import ProblemBatch, rectangle_rings_to_rectangles

    b = synth.make_batch(3, 5, 5, 5, 6, 20, True, seed=5)
    hl, hw = 0.5 * b.obs_dims[..., 0], 0.5 * b.obs_dims[..., 1]
    ccw = np.stack([np.stack([-hl, -hw], -1), np.stack([hl, -hw], -1), np.stack([hl, hw], -1), np.stack([-hl, hw], -1)], axis=2)
    poly = np.concatenate([ccw, np.zeros_like(ccw[:, :, :2])], axis=2)   # poly_stride 6
    kw = {k: getattr(b, k) for k in ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
                                      "obs_pose", "obs_dims", "final_time_step", "veh_l", "veh_w", "max_speed", "max_accel", "tick_t", "check_stride")}
    allrect = ProblemBatch(**kw, obs_poly=poly, obs_nvert=np.full(b.obs_dims.shape[:2], 4, dtype=np.int32))
    assert allrect.obs_nvert is None and allrect.obs_poly is None and allrect.poly_stride == 0
    nv = np.full(b.obs_dims.shape[:2], 4, dtype=np.int32)
    p2 = poly.copy()
    p2[0, 0, 1, 0] = np.nextafter(p2[0, 0, 1, 0], np.inf)       # one ulp off
    p2[0, 1, :4] = p2[0, 1, :4][::-1]                            # clockwise
    p2[0, 2, :4] = np.roll(p2[0, 2, :4], 2, axis=0)              # the same ring from another corner: still the rectangle
    p2[1, 0, :4] = p2[1, 0, :4] @ np.array([[0.0, 1.0], [-1.0, 0.0]])  # turned by 90 degrees: another polygon
    nv[2, 3] = 5                                                 # not a quadrilateral
    got = rectangle_rings_to_rectangles(p2, nv, b.obs_dims)
    want = np.zeros_like(nv)
    want[0, 0] = want[0, 1] = want[1, 0] = 4
    want[2, 3] = 5
    np.testing.assert_array_equal(got, want)
    mixed = ProblemBatch(**kw, obs_poly=p2, obs_nvert=nv)
    np.testing.assert_array_equal(mixed.obs_nvert, want)
