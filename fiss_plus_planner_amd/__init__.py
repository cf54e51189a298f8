"""fiss_plus_planner_amd - MI355X-native Frenet trajectory sampling-and-scoring engine.

Drop-in for the candidate-generation hot path of SS47816/fiss_plus_planner
(FrenetOptimalPlanner / FopPlusPlanner / FissPlanner / FissPlusPlanner .plan()).
The compute runs in hand-written gfx950 HIP kernels behind a flat C ABI
(include/frenet_gpu.h, csrc/); there is no CPU fallback - importing the engine
without the built library raises.
"""
from .batch import ProblemBatch, lattice_samples, speed_samples  # noqa: F401
from .frenet import FrenetState, FrenetTrajectory, State  # noqa: F401
from .spline import CubicSpline2D  # noqa: F401
from .vehicle import Vehicle, vw_vanagon_params  # noqa: F401

__version__ = "0.1.0"
