"""GPU: adversarial inputs (tests/adversarial.py) through every lattice kernel instance, flag words EXACT against the oracle.

What is being attacked: the conservative collision culls of the fused kernel (group circle, fan half-widths in fp32 / fp16 rounded
outwards, the Hermite bounds of the lateral offsets - csrc/frenet_lattice_fused.hip `float_above`, `FanType::above`, stage G / B)
and its proof-or-scan speed / acceleration / range masks (1e-9 margins, phase A0).  A cull that is not conservative loses a collision
the reference finds; a proof that is not a proof clears a candidate the reference rejects.  The reference semantics are
planners/frenet_optimal_planner.py:140-160 (constraints), :168-208 (collision), :106-138 (truncation).

The multi-round instances (three / four workgroups per compute unit: 80 / 64 VGPRs, the slim 40 KB layout with fp16 fan bounds) only
run on batches of more than 512 / 768 egos on a whole MI355X; fp_ctx_set_option("resident_groups", 2) models a one-CU device, so a
handful of egos - what the oracle can check in seconds - takes exactly those instances (asserted through the launch counters).
"""
import os

import numpy as np
import pytest

import adversarial as A
from fiss_plus_planner_amd import synth

pytestmark = pytest.mark.gpu
SEEDS = int(os.environ.get("FP_ADVERSARIAL_SEEDS", "12"))

# (label, ctx options, which per-CU launch counter must move)
MODES = [
    ("two per CU, latency split", {"lattice_kernel": 2}, "lattice_launches_2"),
    ("two per CU, one workgroup per ego", {"lattice_kernel": 2, "lattice_split": 1}, "lattice_launches_2"),
    ("three per CU", {"lattice_kernel": 2, "resident_groups": 2, "lattice_occupancy": 3}, "lattice_launches_3"),
    ("four per CU (slim layout; three when the 40 KB layout does not hold the shape)", {"lattice_kernel": 2, "resident_groups": 2}, "lattice_launches_4|lattice_launches_3"),
    ("lane per candidate", {"lattice_kernel": 1}, None),
]
RESET = {"lattice_kernel": 0, "lattice_split": 0, "resident_groups": 0, "lattice_occupancy": 0}


def run_modes(engine, batch, ref, what, modes=MODES, winner=True):
    try:
        for label, opts, counter in modes:
            for k, v in {**RESET, **opts}.items():
                engine.set_option(k, v)
            before = sum(engine.get_option(c) for c in counter.split("|")) if counter else 0
            out = engine.plan_dense(batch, winner=winner)
            if counter:
                assert sum(engine.get_option(c) for c in counter.split("|")) > before, f"{what}: '{label}' did not take its instance"
            for e, r in enumerate(ref):
                bad = np.nonzero(out.flags[e] != r.flags)[0]
                assert bad.size == 0, (f"{what} [{label}] ego {e}: {bad.size} flag words differ, first candidate {bad[0]}: "
                                       f"got {out.flags[e][bad[0]]:#x} want {r.flags[bad[0]]:#x}")
                np.testing.assert_allclose(out.cost[e], r.cost, rtol=0, atol=1e-6, err_msg=f"{what} [{label}] ego {e}")
                assert out.best_idx[e] == r.best_idx, (what, label, e)
    finally:
        for k, v in RESET.items():
            engine.set_option(k, v)


def shapes(seed):
    """BASELINE's dense shape (the shaped instances), an odd one (the run-time-shape instances) and, every sixth seed, BASELINE's shape on
    220-knot reference lines (the WIN instances: a window of the spline's coefficient columns in LDS, the rest read from global memory)."""
    if seed % 6 == 4:
        return synth.make_batch(12, 9, 9, 7, 50, 50, True, 7000 + seed, layout="survey8d", n_knots=220)
    if seed % 2 == 0:
        return synth.make_batch(12, 9, 9, 7, 50, 50, True, 7000 + seed, layout="survey8d")
    return synth.make_batch(10, 7, 5, 4, 23, 64, True, 7000 + seed, layout="lanes")


@pytest.mark.parametrize("seed", range(SEEDS))
def test_obstacles_within_a_millimetre_of_contact(oracle, engine, seed):
    base = shapes(seed)
    gaps = (1e-6, 1e-5, 1e-4, 1e-3) if seed % 3 else (1e-6, 3e-6)
    batch, placed = A.contact_scene(oracle, base, 100 + seed, gaps=gaps)
    ref = [p.fop_plan() for p in oracle.problems_from_batch(batch)]
    # the generator did what it says: most placements decide their candidate's collision bit by the sign of the gap
    decisive = sum(bool(ref[e].flags[c] & 4) == (eps < 0) for e, c, _k, _j, eps in placed)
    assert len(placed) >= batch.B and decisive >= 0.4 * len(placed), (len(placed), decisive)   # (a sanity check of the generator, not of the kernels)
    n_coll = sum(int(((r.flags & 4) != 0).sum()) for r in ref)
    assert 0 < n_coll < batch.B * batch.C
    before4 = engine.get_option("lattice_launches_4")
    run_modes(engine, batch, ref, f"contact scene seed {seed}")
    if seed % 2 == 0:  # BASELINE's dense shape (81-knot lines, or 220 knots through the WIN instance): the slim four-per-CU layout (fp16 fan bounds) was attacked
        assert engine.get_option("lattice_launches_4") > before4


@pytest.mark.parametrize("seed", range(SEEDS))
def test_polygon_obstacles_within_a_millimetre_of_contact(oracle, engine, seed):
    """The same attack with convex-polygon columns (the POLY instances: ring narrow phase behind the box test, inner-disk shortcut): the
    contact distance is the oracle's polygon predicate's own edge.  The four-per-CU instances have no polygon variant: three per CU
    (on the 220-knot seeds: the windowed POLY instances)."""
    base = shapes(seed)
    batch, placed = A.contact_scene(oracle, base, 1100 + seed, polygons=0.7, gaps=(1e-6, 1e-5, 1e-4, 1e-3) if seed % 2 else (1e-6, 2e-6))
    assert batch.obs_nvert is not None and (batch.obs_nvert > 0).sum() >= batch.B // 2
    ref = [p.fop_plan() for p in oracle.problems_from_batch(batch)]
    decisive = sum(bool(ref[e].flags[c] & 4) == (eps < 0) for e, c, _k, _j, eps in placed)
    assert decisive >= 0.35 * len(placed), (len(placed), decisive)
    poly_modes = [m for m in MODES if "four per CU" not in m[0]] + [("three per CU (no cap)", {"lattice_kernel": 2, "resident_groups": 2}, "lattice_launches_3")]
    run_modes(engine, batch, ref, f"polygon contact scene seed {seed}", modes=poly_modes)


@pytest.mark.parametrize("seed", range(max(2, SEEDS // 2)))
def test_contact_at_the_edge_of_the_fan_with_t_now_and_short_horizons(oracle, engine, seed):
    """The same attack with t_now > 0, a final_time_step inside the table and check stride 1 / 3 (rows, pose limits and the last
    checked pose move)."""
    rng = np.random.default_rng(900 + seed)
    base = shapes(seed + 1)
    base = A._copy(base, t_now=rng.integers(0, 12, base.B), final_time_step=rng.integers(base.T_obs - 14, base.T_obs + 3, len(base.final_time_step)).astype(np.int32),
                   check_stride=int(rng.choice([1, 3])))
    batch, placed = A.contact_scene(oracle, base, 300 + seed)
    ref = [p.fop_plan() for p in oracle.problems_from_batch(batch)]
    assert len(placed) >= batch.B // 2
    run_modes(engine, batch, ref, f"contact scene (t_now, stride {batch.check_stride}) seed {seed}")


@pytest.mark.parametrize("seed", range(max(2, SEEDS // 2)))
def test_limits_at_a_candidates_own_extreme(oracle, engine, seed):
    base = shapes(seed)
    for batch, what in A.limit_batch(oracle, base, 500 + seed):
        ref = [p.fop_plan() for p in oracle.problems_from_batch(batch)]
        run_modes(engine, batch, ref, what, modes=MODES[1:], winner=False)


@pytest.mark.parametrize("seed", range(max(2, SEEDS // 2)))
def test_egos_at_the_ends_of_the_reference_line(oracle, engine, seed):
    base = synth.make_batch(24, 9, 9, 7, 50, 50, True, 7100 + seed, layout="survey8d") if seed % 2 == 0 else synth.make_batch(24, 5, 6, 3, 12, 40, True, 7100 + seed)
    batch = A.spline_end_batch(base, 700 + seed)
    ref = [p.fop_plan() for p in oracle.problems_from_batch(batch)]
    M = np.concatenate([(r.flags >> 20) for r in ref])
    N = np.concatenate([(r.flags >> 8) & 0xFFF for r in ref])
    assert (M == 0).any() and (M == 1).any() and ((M > 1) & (M < N)).any() and (M == N).any()  # every truncation case is in the batch
    run_modes(engine, batch, ref, f"spline ends seed {seed}")
