"""GPU: the FISS+ search in workgroups appended to the lattice launch ("fiss_fused", default on for multi-round batches) against the
search kernel in its own launch: every output of fp_plan_fiss identical, on repeated calls (the dense tables travel between workgroups
of ONE launch - possibly on different XCDs, whose L2s are not coherent inside a launch: agent-scope stores / loads and a flag per ego;
any stale read would show up as a differing walk)."""
import numpy as np
import pytest

from fiss_plus_planner_amd import synth

pytestmark = pytest.mark.gpu

KEYS = ("best_ijk", "stats", "refined", "prev_best_idx", "best_flags")
FKEYS = ("best_cost", "end_state", "best_traj", "trace")


def _run(engine, batch, fused, prev, **kw):
    engine.set_option("fiss_fused", fused)
    try:
        return engine.plan_fiss(batch, "FISS+", prev_best_idx=prev, winner=True, trace=True, **kw)
    finally:
        engine.set_option("fiss_fused", 1)


def _same(a, b, what):
    for k in KEYS:
        np.testing.assert_array_equal(getattr(a, k), getattr(b, k), err_msg=f"{what}: {k}")
    found = ~np.isnan(b.best_cost)
    for k in FKEYS:
        x, y = getattr(a, k), getattr(b, k)
        if k == "trace":  # (the refinement trace of an ego without a coarse winner is not written)
            x, y = x[found], y[found]
        assert np.array_equal(x, y, equal_nan=True), f"{what}: {k}"


@pytest.mark.parametrize("B", [2048, 700, 513])
def test_fused_search_equals_its_own_launch(engine, B):
    batch = synth.make_config(4, B=B)
    rng = np.random.default_rng(B)
    prev = np.where(rng.uniform(size=(B, 1)) < 0.5, -1, np.column_stack([rng.integers(0, batch.nd, B), rng.integers(0, batch.nv, B), rng.integers(0, batch.nt, B)])).astype(np.int32)
    ref = _run(engine, batch, 0, prev)
    assert (~np.isnan(ref.best_cost)).any() and np.isnan(ref.best_cost).any() and ref.refined.any()
    other = synth.make_config(4, B=B, ego_offset=50000)
    for rep in range(6):
        if rep % 2 == 0:  # other egos' tables in the ctx's scratch: a search that read its rows too early would walk THOSE
            _run(engine, other, rep // 2 % 2, None)
        _same(_run(engine, batch, 1, prev), ref, f"B={B} repetition {rep}")


def test_fused_search_against_the_oracle(oracle, engine):
    batch = synth.make_config(4, B=640)
    out = _run(engine, batch, 1, None)
    egos = list(range(0, 640, 16))
    for e, p in zip(egos, oracle.problems_from_batch(batch, egos)):
        r = p.fissplus_plan()
        np.testing.assert_array_equal(out.stats[e], r.stats, err_msg=f"ego {e}")
        assert np.isnan(out.best_cost[e]) == np.isnan(r.best_cost)
        if not np.isnan(r.best_cost):
            assert abs(out.best_cost[e] - r.best_cost) < 1e-6
            np.testing.assert_allclose(out.end_state[e], r.end_state, rtol=0, atol=1e-9)


def test_fused_search_other_lattice_shape_and_skipped_egos(engine):
    """The run-time-shape instance (7 x 7 x 7 = 343 samples, 20 obstacles) and a batch with finished egos (fp_batch.skip: their
    lattice workgroups leave at once and still have to release their search workgroups)."""
    b = synth.make_batch(600, 7, 7, 7, 20, 50, True, 81, kind="FISS+")
    _same(_run(engine, b, 1, None), _run(engine, b, 0, None), "7x7x7")
    # several rounds of workgroups on the run-time-shape instances (round 5: a VGPR spill the compiler stored under an empty exec mask
    # gave the second round's workgroups another workgroup's values - only on these instances, only beyond 768 egos)
    for shape, B in (((7, 7, 7, 20, 50), 769), ((7, 7, 7, 20, 50), 2048), ((9, 9, 7, 50, 40), 1500), ((6, 9, 5, 12, 50), 1000)):
        b = synth.make_batch(B, *shape, True, 82, kind="FISS+")
        _same(_run(engine, b, 1, None), _run(engine, b, 0, None), f"{shape} B={B}")
    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch

    goal = np.full((600, 2), 1e9)
    mk = lambda: synth.make_config(4, B=600)
    engine.set_option("fiss_fused", 0)
    try:
        ref = ClosedLoopRunner(engine, DeviceBatch(mk(), 0), goal, "FISS+").run(10)
    finally:
        engine.set_option("fiss_fused", 1)
    out = ClosedLoopRunner(engine, DeviceBatch(mk(), 0), goal, "FISS+").run(10)
    for k in ("done", "cycles", "t_now"):
        np.testing.assert_array_equal(getattr(out, k), getattr(ref, k), err_msg=k)
    assert np.array_equal(out.ego, ref.ego) and (ref.done != 0).any()


def test_fused_search_beside_a_second_stream(oracle, engine):
    """The appended search workgroups while a second ctx / stream runs its own fused FISS+ launches (and their hand-overs) beside them:
    60 repeated 2048-ego calls, every output equal to the three-launch pipeline's; a sample against the oracle."""
    from conftest import SecondStream

    B = 2048
    batch = synth.make_config(4, B=B)
    batch.tables_tag = 9200
    ref = _run(engine, batch, 0, None)
    egos = list(range(0, B, 64))
    for e, p in zip(egos, oracle.problems_from_batch(batch, egos)):
        r = p.fissplus_plan()
        np.testing.assert_array_equal(ref.stats[e], r.stats, err_msg=f"ego {e}")
        assert np.isnan(ref.best_cost[e]) == np.isnan(r.best_cost)
        if not np.isnan(r.best_cost):
            assert abs(ref.best_cost[e] - r.best_cost) < 1e-6
    for rep in range(20):
        _same(_run(engine, batch, 1, None), ref, f"alone, repetition {rep}")
    with SecondStream(synth.make_config(4, B=1300, ego_offset=40000), fiss=True) as h:
        for rep in range(40):
            _same(_run(engine, batch, 1, None), ref, f"beside a second stream, repetition {rep}")
    assert h.calls >= 2
