"""CPU: the CommonRoad XML reader on obstacle shapes the five demo files do not contain - circles, polygons, rectangles with a centre
and an orientation, several shapes under one <shape>.  The scenario is written here (a two-lanelet straight road), so the test needs
no reference data and runs everywhere."""
import numpy as np

from fiss_plus_planner_amd.commonroad_xml import load_scenario
from fiss_plus_planner_amd.obstacles import buffer_circle_ring


def _pt(x, y):
    return f"<point><x>{x}</x><y>{y}</y></point>"


def _lanelet(i, x0, x1, succ=None):
    xs = np.linspace(x0, x1, 6)
    left = "".join(_pt(x, 1.75) for x in xs)
    right = "".join(_pt(x, -1.75) for x in xs)
    s = f'<successor ref="{succ}"/>' if succ else ""
    return f'<lanelet id="{i}"><leftBound>{left}</leftBound><rightBound>{right}</rightBound>{s}</lanelet>'


def _obstacle(i, shape, x0):
    def state(t):
        return (f"<time><exact>{t}</exact></time><position>{_pt(x0 + 0.5 * t, 0.25)}</position><orientation><exact>0.1</exact></orientation>"
                "<velocity><exact>5.0</exact></velocity>")
    traj = "".join(f"<state>{state(t)}</state>" for t in range(1, 6))
    return (f'<dynamicObstacle id="{i}"><type>car</type><shape>{shape}</shape><initialState>{state(0)}</initialState>'
            f"<trajectory>{traj}</trajectory></dynamicObstacle>")


def _scenario(tmp_path, shapes):
    body = _lanelet(1, 0, 50, succ=2) + _lanelet(2, 50, 100)
    body += "".join(_obstacle(10 + k, sh, 20.0 + 10 * k) for k, sh in enumerate(shapes))
    body += ('<planningProblem id="99"><initialState><position>' + _pt(5.0, 0.0) + "</position><orientation><exact>0.0</exact></orientation>"
             "<time><exact>0</exact></time><velocity><exact>8.0</exact></velocity></initialState>"
             '<goalState><time><intervalStart>10</intervalStart><intervalEnd>60</intervalEnd></time><position><lanelet ref="2"/></position></goalState></planningProblem>')
    p = tmp_path / "shapes.xml"
    p.write_text(f'<commonRoad benchmarkID="ZAM_Shapes-1_1_T-1" timeStepSize="0.1" commonRoadVersion="2020a">{body}</commonRoad>')
    return str(p)


RECT = "<rectangle><length>4.5</length><width>2.0</width></rectangle>"
CIRCLE = "<circle><radius>1.2</radius></circle>"
POLY = "<polygon>" + _pt(-2, -1) + _pt(2, -1) + _pt(0, 3) + "</polygon>"
ROT = "<rectangle><length>4.0</length><width>2.0</width><orientation><exact>0.5</exact></orientation><center><x>1.0</x><y>0.5</y></center></rectangle>"


def test_plain_rectangles_stay_rectangle_columns(tmp_path):
    sc = load_scenario(_scenario(tmp_path, [RECT, RECT]))
    assert sc.obstacles.nvert is None and sc.obstacles.poly is None
    np.testing.assert_array_equal(sc.obstacles.dims, [[4.5, 2.0]] * 2)
    assert sc.obstacles.final_time_step == 5 and sc.obstacles.pose.shape == (5, 2, 4)


def test_circle_polygon_and_rotated_rectangle_become_polygon_columns(tmp_path):
    sc = load_scenario(_scenario(tmp_path, [RECT, CIRCLE, POLY, ROT]))
    tab = sc.obstacles
    assert list(tab.nvert) == [0, 64, 3, 4]
    # the circle: commonroad's shapely_object is Point.buffer(radius / 2) (recalled; circle_buffer_factor picks the convention)
    np.testing.assert_allclose(np.hypot(*tab.poly[1, :64].T), 0.6, rtol=0, atol=1e-15)
    full = load_scenario(_scenario(tmp_path, [RECT, CIRCLE]), circle_buffer_factor=1.0).obstacles
    want = buffer_circle_ring(1.2)[::-1]  # shapely's ring runs clockwise; the column stores it counter-clockwise
    assert {tuple(v) for v in full.poly[1, :64].tolist()} == {tuple(v) for v in want.tolist()}
    np.testing.assert_allclose(np.hypot(*full.poly[1, :64].T), 1.2, rtol=0, atol=1e-15)
    # the triangle turns about the centre of its bounding box [-2, 2] x [-1, 3]: pose displaced by (0, 1), ring relative to it
    np.testing.assert_array_equal(tab.poly[2, :3], [(-2, -2), (2, -2), (0, 2)])
    np.testing.assert_allclose(tab.pose[0, 2, :2], [20.0 + 20.0, 0.25 + 1.0])
    np.testing.assert_array_equal(tab.dims[2], [4.0, 4.0])
    # the rotated rectangle: its four corners, counter-clockwise, about the centre of THEIR bounding box (= its own centre (1, 0.5))
    ring = tab.poly[3, :4]
    np.testing.assert_allclose(sorted(np.hypot(*ring.T)), [np.hypot(2.0, 1.0)] * 4, rtol=1e-15)
    assert np.sum(ring[:, 0] * np.roll(ring[:, 1], -1) - ring[:, 1] * np.roll(ring[:, 0], -1)) > 0
    np.testing.assert_allclose(tab.pose[0, 3, :2], [20.0 + 30.0 + 1.0, 0.25 + 0.5])
    e = np.roll(ring, -1, axis=0) - ring
    assert np.isclose(sorted(np.hypot(*e.T)), [2.0, 2.0, 4.0, 4.0]).all()
    assert np.isclose(np.abs(np.arctan2(e[:, 1], e[:, 0]) % (np.pi / 2) - 0.5).min(), 0.0)
    # every polygon column's box contains its ring (what the broad phases test)
    for j in (1, 2, 3):
        v = tab.poly[j, :tab.nvert[j]]
        assert tab.dims[j, 0] >= 2 * np.abs(v[:, 0]).max() and tab.dims[j, 1] >= 2 * np.abs(v[:, 1]).max()


def test_several_shapes_under_one_obstacle_share_its_poses(tmp_path):
    sc = load_scenario(_scenario(tmp_path, [RECT, "<rectangle><length>2.0</length><width>1.0</width><center><x>-2.0</x><y>0.0</y></center></rectangle>" + POLY]))
    tab = sc.obstacles
    assert tab.pose.shape[1] == 3 and list(tab.nvert) == [0, 4, 3]
    np.testing.assert_array_equal(tab.pose[:, 1], tab.pose[:, 2])   # one obstacle, two convex pieces
    # the group's bounding box is [-3, 2] x [-1, 3]: both rings are relative to its centre (-0.5, 1)
    np.testing.assert_allclose(sorted(tab.poly[2, :3].tolist()), sorted([[-1.5, -2.0], [2.5, -2.0], [0.5, 2.0]]))
