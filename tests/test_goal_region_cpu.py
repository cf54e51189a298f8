"""goal_region.is_reached() (planners/benchmark/planning.py:150-153), the oracle's restatement: hand-derived known answers.
(commonroad-io is not in the image: these cases follow its published rule - position inside or ON the goal shape, every interval the
goal state defines contains the state's value - and are the only pin the rule has; the header says so.)"""
import numpy as np
import pytest

RECT = [[0.0, 0.0], [4.0, 0.0], [4.0, 2.0], [0.0, 2.0]]
ELL = [[0.0, 0.0], [4.0, 0.0], [4.0, 4.0], [2.0, 4.0], [2.0, 2.0], [0.0, 2.0]]  # non-convex, like a bent lanelet
NAN = float("nan")


@pytest.mark.parametrize("poly,pt,inside", [
    (RECT, (1.0, 1.0), True), (RECT, (4.0, 1.0), True), (RECT, (0.0, 0.0), True), (RECT, (2.0, 2.0), True),   # interior, edge, vertex, edge
    (RECT, (4.0 + 2 ** -40, 1.0), False), (RECT, (2.0, 2.5), False), (RECT, (-1.0, 1.0), False),
    (ELL, (1.0, 1.0), True), (ELL, (3.0, 3.0), True), (ELL, (1.0, 3.0), False), (ELL, (2.0, 3.0), True), (ELL, (1.0, 2.0), True),
    (ELL, (2.0 - 2 ** -40, 3.0), False),
    (RECT[::-1], (1.0, 1.0), True), (RECT[::-1], (5.0, 1.0), False),          # clockwise ring
    (RECT[:2], (1.0, 0.0), False),                                             # fewer than three vertices: no region
])
def test_point_in_polygon_closed(oracle, poly, pt, inside):
    assert oracle.goal_reached(poly, *pt) is inside


def test_rotated_rectangle_and_vertex_rays(oracle):
    # the horizontal ray through a vertex must not be counted twice
    diamond = [[0.0, -1.0], [1.0, 0.0], [0.0, 1.0], [-1.0, 0.0]]
    assert oracle.goal_reached(diamond, 0.0, 0.0) and oracle.goal_reached(diamond, -0.5, 0.0) and oracle.goal_reached(diamond, 0.5, 0.5)
    assert not oracle.goal_reached(diamond, -2.0, 0.0) and not oracle.goal_reached(diamond, 2.0, 0.0) and not oracle.goal_reached(diamond, 0.75, 0.5)


def test_goal_state_intervals(oracle):
    assert oracle.goal_reached(RECT, 1, 1, time_step=6, intervals=[6, 10, NAN, NAN, NAN, NAN])
    assert not oracle.goal_reached(RECT, 1, 1, time_step=5, intervals=[6, 10, NAN, NAN, NAN, NAN])
    assert oracle.goal_reached(RECT, 1, 1, time_step=10, velocity=3.0, orientation=0.5, intervals=[6, 10, 3.0, 8.0, -1.0, 1.0])
    assert not oracle.goal_reached(RECT, 1, 1, time_step=10, velocity=2.999, orientation=0.5, intervals=[6, 10, 3.0, 8.0, -1.0, 1.0])
    assert not oracle.goal_reached(RECT, 1, 1, time_step=10, velocity=3.0, orientation=1.5, intervals=[6, 10, 3.0, 8.0, -1.0, 1.0])
    assert not oracle.goal_reached(RECT, 9, 9, time_step=7, intervals=[6, 10, NAN, NAN, NAN, NAN])
    assert oracle.goal_reached(RECT, 1, 1, time_step=99, intervals=None)
