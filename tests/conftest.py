import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def batch_from_golden(g, prefix):
    """Rebuild a ProblemBatch from the `<prefix>in_*` arrays of a fixture."""
    from fiss_plus_planner_amd.batch import ProblemBatch

    sc = g[prefix + "scalars"]
    kw = {}
    for k in ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
              "obs_pose", "obs_dims", "final_time_step", "samp_min", "samp_max", "samp_res"):
        if prefix + k in g.files:
            kw[k] = g[prefix + k]
    return ProblemBatch(**kw, veh_l=float(sc[0]), veh_w=float(sc[1]), max_speed=float(sc[2]), max_accel=float(sc[3]),
                        tick_t=float(sc[4]), check_stride=int(sc[5]))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def engine():
    from fiss_plus_planner_amd.engine import FrenetEngine

    eng = FrenetEngine(0)  # raises when there is no GPU / no built library: no silent fallback
    yield eng
    eng.close()
