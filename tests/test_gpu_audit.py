"""GPU: fp_result.audit (ABI 11) - the margins under an ego's answer.  "Index exact" rests on two margins: the closed-form cost sums
(~1e-12 from the reference's point-by-point sums) and fp64 box overlaps (GEOS decides exactly).  The audit pass reports when either
is gone and settles near-tied costs with the reference's own summation."""
import os

import numpy as np
import pytest

import collision_pairs as cp
from fiss_plus_planner_amd import _abi, synth
from test_collision_exact import contact_batch, ego_boxes

pytestmark = pytest.mark.gpu


def test_near_tie_bit_and_the_index_the_oracle_picks(oracle, engine):
    """Mirrored lateral samples tie exactly when the ego sits on the reference line (d = 0) and by ~1e-12 when it sits 1e-13 m beside
    it - a gap the closed-form sums cannot be trusted with.  With an even number of lateral samples the cheapest candidates ARE such a
    pair: the bit must be set and the index must be the oracle's (point-by-point sums + FOP's last-minimum rule), for every offset."""
    offs = [0.0, 1e-13, -1e-13, 3e-13, -3e-13, 1e-12, -1e-12, 0.3]
    batch = synth.make_batch(len(offs), 4, 3, 2, 0, 0, False, 77)   # nd = 4: no centre sample; no obstacles
    batch.ego[:, 3] = offs
    batch.ego[:, 4:6] = 0.0
    out = engine.plan_dense(batch, tables=True, audit=True)
    probs = oracle.problems_from_batch(batch)
    ref = [p.fop_plan() for p in probs]
    ref_idx = np.array([r.best_idx for r in ref])
    assert (ref_idx >= 0).all()
    np.testing.assert_array_equal(out.best_idx, ref_idx)
    tie = (out.audit & _abi.AUDIT_NEAR_TIE) != 0
    assert tie[:-1].all() and not tie[-1], out.audit            # 0.3 m beside the line: the mirror pair is ~1 apart in cost
    assert (out.audit & _abi.AUDIT_CONTACT == 0).all()           # no obstacles
    # the gap really is that small: runner-up within 1e-9 of the winner in the oracle's own sums
    for b in range(len(offs) - 1):
        c = np.where((ref[b].flags & 7) == 0, ref[b].cost, np.inf)
        two = np.sort(c)[:2]
        assert two[1] - two[0] < 1e-9
    # best_cost of a settled ego is the point-by-point sum of its winner: within rounding of the oracle's
    assert np.abs(out.best_cost - np.array([r.cost[r.best_idx] for r in ref])).max() < 1e-12
    # without the audit the same call leaves the closed-form argmin (possibly the other twin) and no bits
    plain = engine.plan_dense(batch, tables=True)
    assert plain.audit is None
    same_pair = [abs(int(a) // (batch.nt * batch.nv) - int(b) // (batch.nt * batch.nv)) in (0, batch.nd - 1 - 2 * (min(int(a), int(b)) // (batch.nt * batch.nv)))
                 for a, b in zip(plain.best_idx, out.best_idx)]
    assert all(same_pair)


def test_contact_bit_on_pairs_a_few_ulp_from_touching(oracle, engine):
    """One-pose scenes (tests/test_collision_exact.py): the ego's box and one obstacle within +-4 ulp (~1e-13 m) of touching - whatever
    the verdict, it hangs on the last places: FP_AUDIT_CONTACT.  The same constructions pushed ~1e-6 m apart: no bit."""
    rng = np.random.default_rng(21)
    a = ego_boxes(rng, 200)
    b, k, ego = cp.near_contact(a, rng, 50)
    out = engine.plan_dense(contact_batch(a, b, ego), tables=True, audit=True)
    thin = (out.audit & _abi.AUDIT_CONTACT) != 0
    assert thin.mean() > 0.995, thin.mean()
    assert (out.audit & _abi.AUDIT_NEAR_TIE == 0).all()          # one candidate per ego
    far, kf, egof = cp.near_contact(a, rng, 50, K=4, scale=2 ** 26)   # k x 2^26 ulp: 1e-6 .. 1e-5 m either side of contact
    sel = kf != 0
    out_far = engine.plan_dense(contact_batch(a, far[sel], egof[sel]), tables=True, audit=True)
    assert (out_far.audit == 0).all()
    # ... and the far verdicts are the exact predicate's, as ever
    exact = oracle.boxes_intersect_batch(a[egof][sel], far[sel], exact=True, threads=min(16, len(os.sched_getaffinity(0))))
    hit = (out_far.flags[:, 0] & 4) != 0
    # (box A here is the nominal box: the kernel's pose 0 equals it to ~1e-13, far inside 1e-6)
    np.testing.assert_array_equal(hit.astype(np.int8), exact)


def test_audit_is_quiet_on_the_synthetic_configurations(engine):
    """BASELINE configs[1] / configs[2] at full size: the margins measured in tests/test_gpu_near_ties.py (winner ahead by > 1e-7)
    mean no near-tie bit; contacts within 1e-9 m are as good as absent in continuous random scenes.  The answers do not move."""
    for cfg, B in ((2, 256), (3, 2048)):
        batch = synth.make_config(cfg, B=B)
        ref = engine.plan_dense(batch, tables=False, winner=True, traj_stride=112, traj_sparse=True)
        out = engine.plan_dense(batch, tables=False, winner=True, traj_stride=112, traj_sparse=True, audit=True)
        np.testing.assert_array_equal(out.best_idx, ref.best_idx)
        assert np.array_equal(out.best_cost, ref.best_cost, equal_nan=True)
        assert np.array_equal(out.best_traj, ref.best_traj, equal_nan=True)
        assert (out.audit & (_abi.AUDIT_NEAR_TIE | _abi.AUDIT_REORDERED)).sum() == 0
        assert ((out.audit & _abi.AUDIT_CONTACT) != 0).mean() < 0.01


def test_more_ties_than_slots_is_deterministic_and_flagged(oracle, engine):
    """80 end-speed samples that are all the same speed: 80 candidates per lateral sample with bit-identical costs - more than the 63
    the audit pass re-prices.  The subset it takes is the first 63 in index order whatever the thread timing (ballot compaction, not
    an atomic counter): ten calls give the same answer, FP_AUDIT_TIES_OVERFLOW says that not every tied candidate was looked at, and
    the index is still the reference's (last minimum among equal sums)."""
    batch = synth.make_batch(3, 2, 80, 1, 0, 0, False, 78)
    batch.v_samples[:] = batch.v_samples[:, 40:41]
    ref_idx = np.array([p.fop_plan().best_idx for p in oracle.problems_from_batch(batch)])
    first = engine.plan_dense(batch, tables=True, audit=True)
    assert (first.audit & _abi.AUDIT_NEAR_TIE).all() and (first.audit & _abi.AUDIT_TIES_OVERFLOW).all()
    np.testing.assert_array_equal(first.best_idx, ref_idx)
    for _ in range(10):
        out = engine.plan_dense(batch, tables=True, audit=True)
        np.testing.assert_array_equal(out.best_idx, first.best_idx)
        np.testing.assert_array_equal(out.audit, first.audit)
        assert np.array_equal(out.best_cost, first.best_cost)
    few = synth.make_batch(3, 2, 20, 1, 0, 0, False, 78)
    few.v_samples[:] = few.v_samples[:, 10:11]
    assert (engine.plan_dense(few, tables=True, audit=True).audit & _abi.AUDIT_TIES_OVERFLOW == 0).all()
