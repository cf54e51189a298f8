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


def series_tol(w: np.ndarray, dt: float = 0.1, pos_err: float = 4e-13) -> np.ndarray:
    """Per-element tolerance for comparing a [16, n] trajectory dump (NaN padded) with a reference dump `w`.

    Rows 0-10 (t, s.., d.., x, y): 1e-8 absolute.  Rows 11-15 (yaw, ds, c, c_d, c_dd) are finite differences of x / y
    (frenet_optimal_planner.py:121-134), so two correct implementations whose positions differ by `pos_err` (a few ulp of a
    500 m coordinate) differ by that error propagated through the difference chain: 2 pos_err / ds in yaw, then a division by ds
    for c and by dt for each further derivative.  Candidates that come to a stop have ds ~ 1e-4 m at the end, which makes their
    last curvature samples noise in ANY implementation (the reference's included); the bound follows the noise instead of hiding it
    behind one loose number."""
    tol = np.full(w.shape, 1e-8)
    ds = w[12]
    n = w.shape[1]
    with np.errstate(divide="ignore", invalid="ignore"):
        t_yaw = np.where(ds > 0, 2 * pos_err / ds, np.inf)
        m_last = int(np.sum(~np.isnan(w[11])))  # yaw has M entries, the last one repeats the one before
        if m_last >= 2:
            t_yaw[m_last - 1] = t_yaw[m_last - 2]
        nxt = lambda a: np.append(a[1:], np.inf)
        t_c = (t_yaw + nxt(t_yaw)) / ds + np.abs(w[13]) * 2 * pos_err / ds
        t_cd = (t_c + nxt(t_c)) / dt
        t_cdd = (t_cd + nxt(t_cd)) / dt
    for row, t in ((11, t_yaw), (12, np.full(n, 2 * pos_err)), (13, t_c), (14, t_cd), (15, t_cdd)):
        tol[row] = np.maximum(1e-9, np.nan_to_num(t, nan=np.inf, posinf=np.inf)) * 4 + 1e-9
    return tol


def assert_series_close(got: np.ndarray, want: np.ndarray, dt: float = 0.1, what: str = ""):
    """NaN pattern identical and every finite element within series_tol of the reference dump."""
    assert got.shape == want.shape, what
    assert np.array_equal(np.isnan(got), np.isnan(want)), what
    m = ~np.isnan(want)
    tol = series_tol(want, dt)
    bad = m & ~(np.abs(np.where(m, got - want, 0.0)) <= tol)
    assert not bad.any(), f"{what}: rows {sorted(set(np.nonzero(bad)[0].tolist()))} first {np.argwhere(bad)[:4].tolist()} " \
                          f"err {np.abs(got - want)[bad][:4]} tol {tol[bad][:4]}"


class SecondStream:
    """A second fp_ctx + HIP stream that keeps the device busy with multi-round dense launches of ANOTHER batch from its own host thread
    while the test's engine runs (ctypes drops the GIL inside the C calls, so the two streams' launches interleave on the device)."""

    def __init__(self, batch, fiss=False):
        import threading

        from fiss_plus_planner_amd.engine import FrenetEngine

        self.eng, self.batch, self.fiss = FrenetEngine(0), batch, fiss
        batch.tables_tag = 9900  # (its tables stay on the device: the calls are launches, not uploads)
        self.stop, self.calls, self.err = threading.Event(), 0, None
        self.ref = self._call()
        self.th = threading.Thread(target=self._loop, daemon=True)

    def _call(self):
        if self.fiss:
            return self.eng.plan_fiss(self.batch, "FISS+")
        return self.eng.plan_dense(self.batch, tables=False, winner=True, traj_stride=112, traj_sparse=True)

    def _loop(self):
        try:
            while not self.stop.is_set():
                out = self._call()
                self.calls += 1
                if self.fiss:  # (the hammer's own hand-overs are checked too: it runs the same appended workgroups)
                    assert np.array_equal(out.best_ijk, self.ref.best_ijk) and np.array_equal(out.stats, self.ref.stats)
                else:
                    assert np.array_equal(out.best_idx, self.ref.best_idx) and np.array_equal(out.best_traj, self.ref.best_traj, equal_nan=True)
        except BaseException as ex:  # noqa: BLE001  (reported by the test thread)
            self.err = ex

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.th.join(timeout=120)
        self.eng.close()
        if self.err is not None and exc[0] is None:
            raise self.err
