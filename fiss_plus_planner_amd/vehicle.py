"""Ego vehicle constants (reference planners/common/vehicle/vehicle.py:15-46).

The reference reads them from commonroad-vehicle-models (VehicleType.VW_VANAGON,
planners/benchmark/planning.py:297-298), which is not installable offline: the
defaults below are recalled values and flagged UNVERIFIED; every API takes them as
inputs.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np


def vw_vanagon_params() -> SimpleNamespace:
    """commonroad vehicle type 3 (VW Vanagon) - UNVERIFIED recalled values."""
    return SimpleNamespace(
        l=4.569, w=1.844, a=1.1489, b=1.2859, T_f=1.5740, T_r=1.5740,
        longitudinal=SimpleNamespace(v_max=41.7, a_max=11.5),
        steering=SimpleNamespace(max=1.023, min=-1.023, v_max=0.4, v_min=-0.4, kappa_dot_max=0.4,
                                 kappa_dot_dot_max=20.0),
    )


class Vehicle:
    def __init__(self, vehicle_params=None, safety_factor: float = 1.0):
        p = vehicle_params if vehicle_params is not None else vw_vanagon_params()
        self.l = p.l * safety_factor
        self.w = p.w * safety_factor
        self.h = 1.5 * safety_factor
        self.bbox_size = np.array([self.l, self.w, self.h])
        # footprint corners, clockwise, box-centre frame (closed ring)
        self.corners = [(self.l / 2, self.w / 2), (self.l / 2, -self.w / 2), (-self.l / 2, -self.w / 2),
                        (-self.l / 2, self.w / 2), (self.l / 2, self.w / 2)]
        self.polygon = np.array(self.corners)  # the reference holds a shapely Polygon here; the OBB test needs l, w only
        self.a = p.a
        self.b = p.b
        self.L = self.a + self.b
        self.T_f = p.T_f
        self.T_r = p.T_r
        self.max_speed = p.longitudinal.v_max
        self.max_accel = p.longitudinal.a_max
        self.deccel = -self.max_speed / 5.0
        self.max_steering_angle = p.steering.max
        self.max_steering_rate = p.steering.v_max
        self.max_curvature = math.sin(self.max_steering_angle) / self.L
        self.max_kappa_d = p.steering.kappa_dot_max
        self.max_kappa_dd = p.steering.kappa_dot_dot_max
