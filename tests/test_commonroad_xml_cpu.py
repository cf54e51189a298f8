"""CPU: the CommonRoad XML reader against (a) the committed Flensburg fixture arrays and (b) all five demo files.
Needs the reference's data files, which exist in the build container only: skipped elsewhere."""
import glob
import os

import numpy as np
import pytest

from conftest import load_golden

DEMO = "/root/reference/data/demo"
pytestmark = pytest.mark.skipif(not os.path.isdir(DEMO), reason="reference data files are not present on this machine")


def test_flensburg_matches_fixture():
    from fiss_plus_planner_amd.commonroad_xml import load_scenario

    sc = load_scenario(os.path.join(DEMO, "DEU_Flensburg-1_1_T-1.xml"))
    g = load_golden("g5_closed_loop.npz")
    assert sc.route == [203, 1397, 785]
    np.testing.assert_array_equal(sc.centerline, g["centerline"])
    fts = int(g["final_time_step"])
    assert sc.obstacles.final_time_step == fts
    np.testing.assert_array_equal(sc.obstacles.pose, g["obs_pose"][:fts])
    np.testing.assert_array_equal(sc.obstacles.dims, g["obs_dims"])
    np.testing.assert_array_equal(sc.init_state, g["init_state"])
    np.testing.assert_array_equal(sc.goal_center, g["goal_center"])
    assert sc.max_speed == 13.5 and sc.dt == 0.1


def test_all_demo_scenarios_load():
    from fiss_plus_planner_amd.commonroad_xml import load_scenario
    from fiss_plus_planner_amd.spline import CubicSpline2D

    files = sorted(glob.glob(os.path.join(DEMO, "*.xml")))
    assert len(files) == 5
    for f in files:
        sc = load_scenario(f)
        assert len(sc.route) >= 1 and sc.route[-1] == sc.goal_lanelet
        assert sc.obstacles.pose.shape[1] >= 20 and sc.obstacles.pose[..., 3].any()
        assert (np.hypot(*np.diff(sc.centerline, axis=0).T) > 0).all()  # deduplicated: a valid spline parameterisation
        sp = CubicSpline2D(sc.centerline[:, 0], sc.centerline[:, 1])
        # the initial position projects onto the reference line within a lane width
        s = np.arange(0, sp.s[-1], 0.5)
        x, y, _, _ = sp.sample(s)
        assert np.hypot(x - sc.init_state[0], y - sc.init_state[1]).min() < 3.0


def test_every_demo_scenario_matches_its_fixture():
    """The product's XML reader against the arrays of tests/golden/g11_demo_scenarios.npz, which the golden generator parsed
    with its OWN reader (tests/golden/gen_golden.py:parse_demo - exhaustive route enumeration instead of a breadth-first search,
    winding-number instead of ray-casting point location): route, centerline, obstacle tables, initial state, goal centre, speed."""
    from fiss_plus_planner_amd.commonroad_xml import load_scenario

    g = load_golden("g11_demo_scenarios.npz")
    files = sorted(glob.glob(os.path.join(DEMO, "*.xml")))
    assert [os.path.basename(f)[:-4] for f in files] == [str(n) for n in g["names"]]
    for f in files:
        sc = load_scenario(f)
        n = sc.benchmark_id
        assert sc.route == g[f"{n}_route"].tolist(), n
        np.testing.assert_array_equal(sc.centerline, g[f"{n}_centerline"])
        assert sc.obstacles.final_time_step == int(g[f"{n}_final_time_step"])
        np.testing.assert_array_equal(sc.obstacles.pose, g[f"{n}_obs_pose"])
        np.testing.assert_array_equal(sc.obstacles.dims, g[f"{n}_obs_dims"])
        np.testing.assert_array_equal(sc.init_state, g[f"{n}_init_state"])
        np.testing.assert_array_equal(sc.goal_center, g[f"{n}_goal_center"])
        assert sc.max_speed == float(g[f"{n}_max_speed"])
