"""CPU: fp_batch.launch_order as the Python side builds it (engine.launch_order_hint / engine.device_batch)."""
import numpy as np

from fiss_plus_planner_amd import _abi, synth
from fiss_plus_planner_amd.engine import device_batch, launch_order_hint


def test_hint_is_the_permutation_by_descending_speed():
    b = synth.make_config(3, B=300)
    h = launch_order_hint(b)
    assert h.dtype == np.int32 and sorted(h.tolist()) == list(range(300))
    v = b.ego[h, 1]
    assert (np.diff(v) <= 0).all()
    # ties keep index order (a stable sort: the launch order is reproducible)
    b.ego[:, 1] = 3.0
    assert launch_order_hint(b).tolist() == list(range(300))


def test_device_batch_passes_the_hint_only_when_given():
    b = synth.make_config(3, B=8)
    ptrs = {k: 0x1000 for k in ("d_samples", "t_samples", "v_samples", "target_speed", "ego", "frame_of", "scene_of", "t_now", "nx", "knots", "coef",
                                "obs_pose", "obs_dims", "final_time_step")}
    assert not device_batch(b, ptrs).launch_order
    assert device_batch(b, dict(ptrs, launch_order=0x2000)).launch_order == 0x2000
    assert [n for n, _ in _abi.FpBatch._fields_][-1] == "launch_order"
