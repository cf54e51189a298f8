"""Known-answer tests of the COLLISION PRIMITIVE, derived by hand (no code produced the expected values).

The reference tests an ego rectangle against an obstacle rectangle with
    construct_polygon = affinity.rotate(affinity.translate(polygon, x, y), yaw, use_radians=True)   frenet_optimal_planner.py:162-166
    ego_polygon.intersects(obstacle_polygon)                                                       frenet_optimal_planner.py:191
on centred rectangles (vehicle.py:24-31; commonroad Rectangle).  shapely 2.0.0 is not installable here, so three
implementations of that primitive exist in this repo, all ours: the polygon stand-in the goldens were generated with
(tests/golden/refshim.py), the C oracle (oracle/frenet_oracle.c make_box + quads_intersect) and the HIP separating-axis test
(csrc/frenet_device.h obb_overlap).  Every case below states the geometry and the answer a reader can check on paper; all
three implementations must give it.  Facts used: `intersects` is a closed-set predicate (a shared boundary point counts);
rotate(origin='center') turns the translated polygon about the centre of its bounding box = the rectangle's own centre.

A box is (length, width, x, y, yaw): half extents length/2 along its heading, width/2 across.
"""
import math
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import refshim  # noqa: E402  (the polygon stand-in only; nothing here touches /root/reference)

R2 = math.sqrt(2.0)
PI = math.pi

# name, box A (the ego), box B (the obstacle), expected, derivation
KATS = [
    ("separated_along_x", (4, 2, 0, 0, 0), (2, 2, 3.5, 0, 0), False,
     "A spans x in [-2, 2]; B spans x in [2.5, 4.5]: a gap of 0.5 along x"),
    ("overlapping", (4, 2, 0, 0, 0), (2, 2, 2.5, 0, 0), True,
     "B spans x in [1.5, 3.5], y in [-1, 1]: the strip x in [1.5, 2] is in both"),
    ("edge_touching", (4, 2, 0, 0, 0), (2, 2, 3, 0, 0), True,
     "B spans x in [2, 4]: the boxes share the edge x = 2, y in [-1, 1]; closed sets intersect"),
    ("corner_touching", (4, 2, 0, 0, 0), (2, 2, 3, 2, 0), True,
     "B spans x in [2, 4], y in [1, 3]: the only common point is the corner (2, 1)"),
    ("corner_just_apart", (4, 2, 0, 0, 0), (2, 2, 3.25, 2.25, 0), False,
     "B spans x in [2.25, 4.25], y in [1.25, 3.25]: 0.25 beyond A's corner (2, 1) in both axes"),
    ("identical", (4, 2, 1, -1, 0.3), (4, 2, 1, -1, 0.3), True, "the same box twice"),
    ("separated_only_along_a_diagonal_axis", (4, 2, 0, 0, 0), (2, 2, 3.2, 2.2, PI / 4), False,
     "B is a diamond with half diagonal sqrt 2 about (3.2, 2.2): its x range [1.786, 4.614] and y range [0.786, 3.614] both "
     "overlap A's, but along u = (1, 1)/sqrt 2 (an edge normal of B) A reaches at most (2 + 1)/sqrt 2 = 2.1213 while B starts at "
     "(3.2 + 2.2)/sqrt 2 - 1 = 2.8184"),
    ("diamond_apart", (2, 2, 0, 0, 0), (2, 2, 1 + R2 + 0.01, 0, PI / 4), False,
     "A is the square [-1, 1]^2; the diamond's leftmost vertex is at x = (1 + sqrt 2 + 0.01) - sqrt 2 = 1.01"),
    ("diamond_pierces", (2, 2, 0, 0, 0), (2, 2, 1 + R2 - 0.01, 0, PI / 4), True,
     "the diamond's leftmost vertex (0.99, 0) lies inside the square"),
    ("quarter_turn_shortens_the_box", (4, 2, 0, 0, 0), (4, 2, 4, 0, PI / 2), False,
     "B turned by pi/2 is 2 long in x: it spans x in [3, 5] and A ends at x = 2 (unrotated it would span [2, 6] and touch)"),
    ("quarter_turn_same_centre_unrotated", (4, 2, 0, 0, 0), (4, 2, 4, 0, 0), True,
     "the control for the case above: B spans x in [2, 6] and touches A's edge x = 2"),
    ("quarter_turn_touching", (4, 2, 0, 0, 0), (4, 2, 3, 0, PI / 2), True,
     "B turned by pi/2 spans x in [2, 4], y in [-2, 2]: it shares the edge x = 2 with A (needs cos(pi/2) snapped to exactly 0)"),
    ("half_turn_touching", (4, 2, 0, 0, 0), (4, 2, 4, 0, PI), True,
     "a half turn maps the rectangle onto itself: B spans x in [2, 6] and touches x = 2 (needs sin(pi) snapped to exactly 0)"),
    ("half_turn_apart", (4, 2, 0, 0, 0), (4, 2, 4.0000001, 0, PI), False, "as above, moved 1e-7 away"),
    ("negative_quarter_turn_touching_in_y", (4, 2, 0, 0, 0), (2, 6, 0, 2, -PI / 2), True,
     "B (length 2, width 6) turned by -pi/2 is 6 wide in x and 2 tall in y: it spans y in [1, 3] and shares the edge y = 1 with A"),
    ("thin_bars_cross", (10, 0.2, 0, 0, 0), (10, 0.2, 0, 0, PI / 2), True,
     "a plus sign: the bars cross at the origin although no vertex of either lies inside the other"),
    ("thin_bars_T_touch", (10, 0.25, 0, 0, 0), (10, 0.25, 0, 5.125, PI / 2), True,
     "the upright bar spans y in [0.125, 10.125] and stands on the top edge y = 0.125 of the flat one"),
    ("thin_bars_T_apart", (10, 0.25, 0, 0, 0), (10, 0.25, 0, 5.25, PI / 2), False,
     "the upright bar spans y in [0.25, 10.25]: 0.125 above the flat one"),
    ("containment", (4, 2, 0, 0, 0), (1, 0.5, 0.5, 0.2, 0.3), True,
     "B's bounding circle (radius 0.56 about (0.5, 0.2)) lies inside A: no edges cross, the interiors still intersect"),
    ("rotation_is_about_the_box_centre", (2, 2, 10, 1.5, 0), (4, 2, 10, 0, PI / 2), True,
     "B at (10, 0) turned by pi/2 occupies x in [9, 11], y in [-2, 2] and overlaps A = [9, 11] x [0.5, 2.5]; a rotation about "
     "the ORIGIN would have carried B to (0, 10)"),
    ("rotation_is_not_about_the_origin", (2, 2, 0, 10, 0), (4, 2, 10, 0, PI / 2), False,
     "the counterpart: where a rotation about the origin would have put B there is nothing"),
    ("thirty_degrees_corner_inside", (4, 2, 0, 0, PI / 6), (2, 2, 3.1, 0, 0), True,
     "A's corner (hl, -hw) turned by 30 deg is (2 cos30 + sin30, 2 sin30 - cos30) = (2.232, 0.134): inside B = [2.1, 4.1] x [-1, 1]"),
    ("thirty_degrees_apart", (4, 2, 0, 0, PI / 6), (2, 2, 3.3, 0, 0), False,
     "A reaches x = 2 cos30 + 1 sin30 = 2.232 at most; B starts at x = 2.3"),
]
IDS = [k[0] for k in KATS]


def shim_intersects(a, b):
    """The reference's own call sequence (construct_polygon twice + intersects) on the stand-in polygon class."""
    def poly(box):
        l, w, x, y, yaw = box
        hl, hw = l / 2.0, w / 2.0
        base = refshim.Polygon([(hl, hw), (hl, -hw), (-hl, -hw), (-hl, hw), (hl, hw)])  # vehicle.py:24-31
        return refshim.affinity.rotate(refshim.affinity.translate(base, xoff=x, yoff=y), yaw, use_radians=True)
    return poly(a).intersects(poly(b))


@pytest.mark.parametrize("name,a,b,expected,why", KATS, ids=IDS)
def test_stand_in_polygon_known_answers(name, a, b, expected, why):
    assert shim_intersects(a, b) is expected, why
    assert shim_intersects(b, a) is expected, why  # the predicate is symmetric


@pytest.mark.parametrize("name,a,b,expected,why", KATS, ids=IDS)
def test_oracle_known_answers(oracle, name, a, b, expected, why):
    assert oracle.boxes_intersect(a, b) is expected, why
    assert oracle.boxes_intersect(b, a) is expected, why


def test_oracle_reports_unbuildable_polygons(oracle):
    """A non-finite coordinate makes the reference's polygon construction raise; has_collision turns that into 'collision' for the
    ego (bare except, :178-182)."""
    assert oracle.boxes_intersect((4, 2, float("nan"), 0, 0), (2, 2, 0, 0, 0)) is None
    assert oracle.boxes_intersect((4, 2, 0, 0, float("inf")), (2, 2, 0, 0, 0)) is None


# ---------------------------------------------------------------------------------------------------------------
# the HIP narrow phase, through the C ABI, on one-pose scenes
# ---------------------------------------------------------------------------------------------------------------
def one_pose_scene(a, b):
    """A problem batch whose ONLY collision test is box A (the ego at pose 0) against box B (one obstacle at time step 0).

    The reference line is the straight line through A's centre along A's heading, so that the trajectory with d = 0 and constant
    speed puts the ego at exactly (x_a, y_a) with heading yaw_a at pose 0; final_time_step = 1 limits has_collision to pose 0."""
    from fiss_plus_planner_amd.batch import ProblemBatch
    from fiss_plus_planner_amd.spline import build_frames

    la, wa, xa, ya, tha = a
    lb, wb, xb, yb, thb = b
    j = np.arange(17) * 5.0 - 20.0
    pts = np.stack([xa + j * math.cos(tha), ya + j * math.sin(tha)], axis=1)[None]
    knots, coef = build_frames(pts)
    return ProblemBatch(
        d_samples=[0.0], t_samples=[8.0], v_samples=[[5.0]], target_speed=[5.0], ego=[[20.0, 5.0, 0.0, 0.0, 0.0, 0.0]],
        frame_of=[0], scene_of=[0], t_now=[0], nx=[17], knots=knots, coef=coef,
        obs_pose=np.array([[[[xb, yb, thb, 1.0]]]]), obs_dims=np.array([[[lb, wb]]]), final_time_step=[1],
        veh_l=la, veh_w=wa, max_speed=100.0, max_accel=100.0)


def hip_answers(engine, a, b):
    """{path: collides} for the three kernels that hold a copy of the narrow phase."""
    batch = one_pose_scene(a, b)
    out = {"eval_trajs": bool(engine.eval_trajs(batch, np.array([[[0.0, 5.0, 8.0]]])).flags[0, 0] & 4)}
    for name, which in (("lattice_percand", 1), ("lattice_fused", 2)):
        engine.set_option("lattice_kernel", which)
        try:
            out[name] = bool(engine.plan_dense(batch, tables=True).flags[0, 0] & 4)
        finally:
            engine.set_option("lattice_kernel", 0)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name,a,b,expected,why", KATS, ids=IDS)
def test_hip_known_answers(engine, name, a, b, expected, why):
    got = hip_answers(engine, a, b)
    assert got == {k: expected for k in got}, why


GRAZE = []
for yaw_b, half in ((0.0, 1.5), (PI / 2, 1.0), (PI, 1.5), (-PI / 2, 1.0), (3 * PI / 2, 1.0), (2 * PI, 1.5)):
    for gap, expected in ((-1e-12, True), (0.0, True), (1e-12, False)):
        GRAZE.append((f"x_yaw{yaw_b:.3f}_gap{gap:+.0e}", (4.5, 2, 0, 0, 0), (3, 2, 2.25 + half + gap, 0.25, yaw_b), expected))
        GRAZE.append((f"y_yaw{yaw_b:.3f}_gap{gap:+.0e}", (4.5, 2, 0, 0, 0), (3, 2, 0.5, 1.0 + (2.5 - half) + gap, yaw_b), expected))
for gap, expected in ((-1e-12, True), (0.0, True), (1e-12, False)):  # corner against corner
    GRAZE.append((f"corner_gap{gap:+.0e}", (4.5, 2, 0, 0, 0), (3, 2, 3.75 + gap, 2.0 + gap, 0.0), expected))


@pytest.mark.parametrize("name,a,b,expected", GRAZE, ids=[g[0] for g in GRAZE])
def test_grazing_boxes_cpu(oracle, name, a, b, expected):
    """Axis-parallel boxes with dyadic sizes and positions: every coordinate is exact in binary floating point, so touching (gap 0)
    is decided exactly and +-1e-12 flips the answer.  yaw = k pi/2 only rotates exactly with shapely's snap of |cos|, |sin| < 2.5e-16."""
    assert oracle.boxes_intersect(a, b) is expected
    assert shim_intersects(a, b) is expected


@pytest.mark.gpu
@pytest.mark.parametrize("name,a,b,expected", GRAZE, ids=[g[0] for g in GRAZE])
def test_grazing_boxes_hip(engine, oracle, name, a, b, expected):
    got = hip_answers(engine, a, b)
    assert got == {k: expected for k in got}
    assert oracle.boxes_intersect(a, b) is expected


@pytest.mark.gpu
def test_grazing_rotated_boxes_hip_equals_oracle(engine, oracle):
    """General headings: the ego's heading comes out of the trajectory's finite differences (not exact), so the sweep stays 1e-9
    away from touching - far above the rounding of either formulation, far below anything a scene would resolve."""
    rng = np.random.default_rng(7)
    n_hit = 0
    for _ in range(40):
        tha, thb = rng.uniform(-PI, PI, 2)
        lb, wb = rng.uniform(2, 6), rng.uniform(1, 2.5)
        # push B away from A along a random direction until the oracle says "apart", then bisect to the contact distance
        phi = rng.uniform(-PI, PI)
        a = (4.5, 2.0, 3.0, -2.0, tha)
        box = lambda r: (lb, wb, a[2] + r * math.cos(phi), a[3] + r * math.sin(phi), thb)
        lo, hi = 0.0, 12.0
        assert oracle.boxes_intersect(a, box(lo)) and not oracle.boxes_intersect(a, box(hi))
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            lo, hi = (mid, hi) if oracle.boxes_intersect(a, box(mid)) else (lo, mid)
        for r, expected in ((lo - 1e-9, True), (lo + 1e-9, False)):
            assert oracle.boxes_intersect(a, box(r)) is expected
            got = hip_answers(engine, a, box(r))
            assert got == {k: expected for k in got}, (a, box(r))
            n_hit += expected
    assert n_hit == 40


# ---------------------------------------------------------------------------------------------------------------
# has_collision()-level rules around the primitive (frenet_optimal_planner.py:168-195) on the oracle
# ---------------------------------------------------------------------------------------------------------------
def test_has_collision_rules_on_the_oracle(oracle):
    from fiss_plus_planner_amd.batch import ProblemBatch

    a, b = (4, 2, 0, 0, 0), (2, 2, 3, 0, 0)  # edge_touching: collides at pose 0
    base = one_pose_scene(a, b)

    def flags(batch):
        return oracle.problems_from_batch(batch)[0].eval_traj(0.0, 5.0, 8.0).flags

    assert flags(base) & 4
    # state_at_time(t) is None -> the obstacle is skipped (:187-188): valid flag 0
    gone = one_pose_scene(a, b); gone.obs_pose[0, 0, 0, 3] = 0.0
    assert not flags(gone) & 4
    # final_time_step - time_step_now bounds the checked poses (:173-174): horizon 0 -> nothing is checked
    none = one_pose_scene(a, b); none.final_time_step[0] = 0
    assert not flags(none) & 4
    # check_res = 2 (:202): pose 1 is never tested.  The obstacle sits where the ego is at step 1 only.
    late = one_pose_scene(a, b)
    late.obs_pose = np.zeros((1, 3, 1, 4)); late.obs_pose[0, 1, 0] = (0.5, 0.0, 0.0, 1.0); late.final_time_step[0] = 3
    late = ProblemBatch(**{k: getattr(late, k) for k in ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of",
                                                         "t_now", "nx", "knots", "coef", "obs_pose", "obs_dims", "final_time_step", "veh_l", "veh_w",
                                                         "max_speed", "max_accel")})
    assert not flags(late) & 4
    late.obs_pose[0, 2, 0] = (1.0, 0.0, 0.0, 1.0)  # ... but pose 2 is
    assert flags(late) & 4
    # M == 1: traj.yaw is empty -> IndexError -> bare except -> "collision" (:178-182).  Start 0.2 m before the end of the line.
    short = one_pose_scene(a, (2, 2, 500.0, 500.0, 0))
    short.ego[0, 0] = 79.8
    r = oracle.problems_from_batch(short)[0].eval_traj(0.0, 5.0, 8.0)
    assert r.M == 1 and r.flags & 4
