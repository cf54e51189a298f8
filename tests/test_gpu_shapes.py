"""GPU: obstacle shapes that are not rectangles (ABI 12: fp_batch.obs_poly / obs_nvert) - every kernel that holds a copy of the narrow
phase (fused lattice kernel in all its launch shapes, lane-per-candidate kernel, eval_trajs, the FISS+ refinement kernel, the audit
pass) against G12 (the imported reference's verdicts and plans on circles, triangles, rotated rectangles, a non-convex L, an
off-centre pentagon, a group of shapes) and against the oracle on random convex polygons."""
import numpy as np
import pytest

from conftest import load_golden
from fiss_plus_planner_amd import _abi, synth
from shapes_util import g12_batch, g12_obstacles, with_random_shapes
from test_gpu_dense import any_kernel, oracle_dense  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
NAMES = ["p555", "p555b", "p997", "p997b"]
TOL = 1e-6


@pytest.mark.parametrize("name", NAMES)
def test_dense_tables_match_the_reference_on_g12(any_kernel, name):
    g = load_golden("g12_shapes.npz")
    b = g12_batch(g, name)
    out = any_kernel.plan_dense(b)
    np.testing.assert_allclose(out.cost, g[f"{name}_cost"], rtol=0, atol=TOL)
    np.testing.assert_array_equal((out.flags & 4) != 0, g[f"{name}_coll"])
    found = g[f"{name}_FOP_found"]
    np.testing.assert_array_equal(out.best_idx[found], g[f"{name}_FOP_flat"][found])
    assert (out.best_idx[~found] == -1).all()
    # ... and the boxed scene gives the boxed verdicts: the polygons, not their boxes, decided the above
    ob = any_kernel.plan_dense(g12_batch(g, name, boxes=True))
    np.testing.assert_array_equal((ob.flags & 4) != 0, g[f"{name}_coll_box"])


@pytest.mark.parametrize("kind", ["FOP", "FOP+", "FISS", "FISS+"])
@pytest.mark.parametrize("name", NAMES)
def test_drop_in_planners_take_the_reference_obstacle_objects(engine, name, kind):
    """The four planner classes are handed what the reference's planners are handed - a list of obstacle objects whose
    obstacle_shape.shapely_object is any polygon - and must return the reference's answer."""
    from test_gpu_planners import _planner

    g = load_golden("g12_shapes.npz")
    key = f"{name}_{kind}"
    b = g12_batch(g, name, kind)
    raw = g12_batch.__globals__["batch_from_golden"](g, f"{name}_in_")
    from fiss_plus_planner_amd.frenet import FrenetState

    for e in range(b.B):
        pl = _planner(kind, b, engine)
        f = int(b.frame_of[e]); nx = int(b.nx[f])
        pl.generate_frenet_frame(np.column_stack([b.coef[f, 0, :nx], b.coef[f, 4, :nx]]))
        sc = int(raw.scene_of[e])
        obstacles = g12_obstacles(g, raw.obs_pose[sc], raw.obs_dims[sc], int(raw.final_time_step[sc]))
        s, s_d, s_dd, d, d_d, d_dd = b.ego[e]
        best = pl.plan(FrenetState(t=0.0, s=s, s_d=s_d, s_dd=s_dd, d=d, d_d=d_d, d_dd=d_dd), float(b.target_speed[e]), obstacles, int(b.t_now[e]))
        found = bool(g[f"{key}_found"][e])
        assert (best is not None) == found, (key, e)
        assert pl.stats.as_tuple() == tuple(g[f"{key}_stats"][e]), (key, e)
        if not found:
            continue
        assert abs(best.cost_final - g[f"{key}_cost"][e]) < TOL
        if kind in ("FOP", "FOP+"):
            assert best.lattice_index == g[f"{key}_flat"][e]
        else:
            np.testing.assert_allclose([best.end_state.d, best.end_state.s_d, best.end_state.t], g[f"{key}_end"][e], atol=1e-9)
            if g[f"{key}_idx"][e][0] >= 0:
                np.testing.assert_array_equal(best.idx, g[f"{key}_idx"][e])


@pytest.mark.parametrize("kind", ["FISS", "FISS+"])
@pytest.mark.parametrize("name", NAMES)
def test_batch_fiss_pipeline_on_g12(engine, name, kind):
    g = load_golden("g12_shapes.npz")
    key = f"{name}_{kind}"
    out = engine.plan_fiss(g12_batch(g, name, kind), kind)
    np.testing.assert_array_equal(out.stats, g[f"{key}_stats"])
    found = g[f"{key}_found"]
    np.testing.assert_array_equal(~np.isnan(out.best_cost), found)
    np.testing.assert_allclose(out.best_cost[found], g[f"{key}_cost"][found], rtol=0, atol=TOL)
    np.testing.assert_allclose(out.end_state[found], g[f"{key}_end"][found], rtol=0, atol=1e-9)


@pytest.mark.parametrize("cfg", [dict(B=6, nd=5, nv=5, nt=5, n_obs=10, T_obs=100, moving=False, seed=61),
                                 dict(B=8, nd=9, nv=9, nt=7, n_obs=50, T_obs=50, moving=True, seed=62),
                                 dict(B=3, nd=3, nv=4, nt=2, n_obs=7, T_obs=33, moving=True, seed=63)])
def test_random_convex_polygons_vs_oracle(oracle, any_kernel, cfg):
    base = synth.make_batch(cfg["B"], cfg["nd"], cfg["nv"], cfg["nt"], cfg["n_obs"], cfg["T_obs"], cfg["moving"], cfg["seed"])
    b = with_random_shapes(base, cfg["seed"])
    out = any_kernel.plan_dense(b)
    bi, bc, cost, flags, stats = oracle_dense(oracle, b)
    np.testing.assert_array_equal(out.flags, flags)
    np.testing.assert_array_equal(out.best_idx, bi)
    np.testing.assert_allclose(out.cost, cost, rtol=0, atol=TOL)
    # the polygons sit inside the rectangles they replaced: collisions can only disappear (and in the dense scene some do)
    flags_rect = oracle_dense(oracle, base)[3]
    assert not (((flags & 4) != 0) & ((flags_rect & 4) == 0)).any()
    if cfg["n_obs"] >= 50:
        assert ((flags_rect & 4) != 0).sum() > ((flags & 4) != 0).sum()


@pytest.mark.parametrize("n_obs", [50, 48])
def test_random_polygons_at_config3_size_both_kernels_and_the_fiss_pipeline(oracle, engine, n_obs):
    """Three-workgroups-per-CU instances (B > 512; 50 obstacles: the instance compiled for BASELINE's config-3 shape, 48: the
    run-time-shape one) + tail split + feedback order on polygon scenes: identical to the lane-per-candidate kernel on every ego, to the
    oracle on a sample; FISS+ (search + refinement kernel) against the oracle on a sample."""
    base = synth.make_config(3, B=640) if n_obs == 50 else synth.make_batch(640, 9, 9, 7, n_obs, 50, True, synth.CONFIG_SEEDS[3], "FOP")
    b = with_random_shapes(base, 99, frac=0.5)
    fused = engine.plan_dense(b, tables=True)
    engine.set_option("lattice_kernel", 1)
    try:
        pc = engine.plan_dense(b, tables=True)
    finally:
        engine.set_option("lattice_kernel", 0)
    np.testing.assert_array_equal(fused.flags, pc.flags)
    np.testing.assert_array_equal(fused.best_idx, pc.best_idx)
    sample = list(range(0, b.B, 40))
    bi, bc, cost, flags, stats = oracle_dense(oracle, b, sample)
    np.testing.assert_array_equal(fused.flags[sample], flags)
    np.testing.assert_array_equal(fused.best_idx[sample], bi)
    bf = with_random_shapes(synth.make_config(4, B=96), 98, frac=0.5)
    out = engine.plan_fiss(bf, "FISS+")
    for e, p in zip(range(0, 96, 4), oracle.problems_from_batch(bf, range(0, 96, 4))):
        r = p.fissplus_plan()
        np.testing.assert_array_equal(out.stats[e], r.stats, err_msg=f"ego {e}")
        assert np.isnan(out.best_cost[e]) == np.isnan(r.best_cost)
        if not np.isnan(r.best_cost):
            assert abs(out.best_cost[e] - r.best_cost) < TOL
            np.testing.assert_allclose(out.end_state[e], r.end_state, rtol=0, atol=1e-9)


def test_eval_trajs_and_closed_loop_see_the_polygons(oracle, engine):
    b = with_random_shapes(synth.make_batch(6, 5, 5, 5, 10, 100, True, 64), 64)
    rng = np.random.default_rng(64)
    K = 12
    es = np.stack([rng.uniform(-0.8, 0.8, (b.B, K)), rng.uniform(2, 13, (b.B, K)), rng.uniform(3, 8, (b.B, K))], axis=-1)
    out = engine.eval_trajs(b, es)
    for e, p in enumerate(oracle.problems_from_batch(b)):
        for k in range(K):
            t = p.eval_traj(*es[e, k])
            assert (out.flags[e, k] & 0xFF) == (t.flags & 0xFF), (e, k)
            assert abs(out.cost[e, k] - t.cost) < TOL


def _one_pose_polygon_scene(a, ring, pose):
    """test_collision_exact.contact_batch with the obstacle a polygon column: ego box `a` = (l, w, x, y, yaw) at pose 0 against the
    ring at `pose`."""
    from test_collision_exact import contact_batch

    ring = np.asarray(ring, dtype=float)
    box = np.array([[2 * np.abs(ring[:, 0]).max(), 2 * np.abs(ring[:, 1]).max(), *pose]])
    bt = contact_batch(np.array([a], dtype=float), box, np.zeros(1, dtype=np.int64))
    bt.obs_poly = np.ascontiguousarray(ring[None, None])
    bt.obs_nvert = np.array([[len(ring)]], dtype=np.int32)
    bt.veh_l, bt.veh_w = float(a[0]), float(a[1])  # (contact_batch sizes the ego as the module's vehicle)
    return bt


@pytest.mark.parametrize("kernel", [2, 1])
def test_polygon_known_answers_through_the_kernels(oracle, engine, kernel):
    """The hand-derived cases of tests/test_shapes_cpu.py (dyadic geometry: exact in fp64) through the C ABI on one-pose scenes."""
    tri = np.array([(-2.0, -2.0), (2.0, -2.0), (0.0, 2.0)])
    ego = (4.0, 2.0, 0.0, 0.0, 0.0)
    engine.set_option("lattice_kernel", kernel)
    try:
        for p, want in ((3.4, True), (3.5, True), (3.5 + 2.0 ** -40, False), (3.6, False), (-3.5, True), (-3.5 - 2.0 ** -40, False)):
            out = engine.plan_dense(_one_pose_polygon_scene(ego, tri, (p, 0.0, 0.0)), tables=True)
            assert bool(out.flags[0, 0] & 4) is want, p
        inside = engine.plan_dense(_one_pose_polygon_scene(ego, 0.1 * tri, (0.5, 0.2, 1.0)), tables=True)
        around = engine.plan_dense(_one_pose_polygon_scene(ego, 10.0 * tri, (0.0, 0.0, 2.0)), tables=True)
        assert inside.flags[0, 0] & 4 and around.flags[0, 0] & 4
        turned = engine.plan_dense(_one_pose_polygon_scene(ego, tri, (3.5, 0.0, np.pi / 2)), tables=True)
        assert turned.flags[0, 0] & 4
        # the box around the ring is a necessary condition only: inside the triangle's box, outside the triangle
        miss = engine.plan_dense(_one_pose_polygon_scene(ego, tri, (3.9, 0.0, 0.0)), tables=True)
        assert not miss.flags[0, 0] & 4
    finally:
        engine.set_option("lattice_kernel", 0)


def test_host_entry_validates_polygon_columns(engine):
    tri = np.array([(-2.0, -2.0), (2.0, -2.0), (0.0, 2.0)])
    ok = _one_pose_polygon_scene((4.0, 2.0, 0.0, 0.0, 0.0), tri, (9.0, 0.0, 0.0))
    engine.plan_dense(ok)
    cw = _one_pose_polygon_scene((4.0, 2.0, 0.0, 0.0, 0.0), tri[::-1], (9.0, 0.0, 0.0))
    with pytest.raises(_abi.FrenetGpuError, match="counter-clockwise"):
        engine.plan_dense(cw)
    small = _one_pose_polygon_scene((4.0, 2.0, 0.0, 0.0, 0.0), tri, (9.0, 0.0, 0.0))
    small.obs_dims = small.obs_dims * 0.5
    with pytest.raises(_abi.FrenetGpuError, match="does not contain"):
        engine.plan_dense(small)
    two = _one_pose_polygon_scene((4.0, 2.0, 0.0, 0.0, 0.0), tri, (9.0, 0.0, 0.0))
    two.obs_nvert = np.array([[2]], dtype=np.int32)
    with pytest.raises(_abi.FrenetGpuError, match="obs_nvert"):
        engine.plan_dense(two)
    # a reflex vertex: the half-plane test would shrink the arrow head to its kernel (missed collisions) - rejected, not shrunk
    arrow = np.array([(-2.0, -2.0), (0.0, -1.0), (2.0, -2.0), (0.0, 2.0)])
    with pytest.raises(_abi.FrenetGpuError, match="not convex"):
        engine.plan_dense(_one_pose_polygon_scene((4.0, 2.0, 0.0, 0.0, 0.0), arrow, (9.0, 0.0, 0.0)))
    # collinear vertices (a rectangle with a vertex in the middle of an edge) are convex
    flat = np.array([(-2.0, -1.0), (0.0, -1.0), (2.0, -1.0), (2.0, 1.0), (-2.0, 1.0)])
    engine.plan_dense(_one_pose_polygon_scene((4.0, 2.0, 0.0, 0.0, 0.0), flat, (9.0, 0.0, 0.0)))


def test_device_entry_validates_vertex_counts_on_request(engine):
    """FP_MEM_DEVICE calls cannot look at the arrays; with the "validate" option a check kernel does (a count beyond poly_stride would
    walk off the ring table)."""
    from fiss_plus_planner_amd.device_batch import DeviceBatch

    b = with_random_shapes(synth.make_batch(4, 5, 5, 5, 10, 60, True, 66), 66)
    bad = b.obs_nvert.copy()
    bad[1, 3] = b.obs_poly.shape[2] + 1
    b.obs_nvert = bad
    db = DeviceBatch(b, 0)
    import torch
    bi = torch.empty(4, dtype=torch.int32, device="cuda:0"); bc = torch.empty(4, dtype=torch.float64, device="cuda:0")
    engine.set_option("validate", 1)
    try:
        with pytest.raises(_abi.FrenetGpuError, match="obs_nvert"):
            engine.plan_dense_device(db.params, db.fb, bi.data_ptr(), bc.data_ptr())
    finally:
        engine.set_option("validate", 0)


@pytest.mark.parametrize("what,match", [("clockwise", "counter-clockwise"), ("reflex", "not convex"), ("box", "does not contain"), ("nan", "NaN")])
def test_device_entry_validates_rings_on_request(engine, what, match):
    """The "validate" option also runs check_batch_host's ring checks on device-resident columns (one lane per column): orientation,
    convexity, the box of obs_dims - a bad ring would be tested as a smaller shape than it is (include/frenet_gpu.h)."""
    import torch

    from fiss_plus_planner_amd.device_batch import DeviceBatch

    b = with_random_shapes(synth.make_batch(4, 5, 5, 5, 10, 60, True, 68), 68)
    s, j = [(s, j) for s in range(b.obs_nvert.shape[0]) for j in range(b.obs_nvert.shape[1]) if b.obs_nvert[s, j] >= 4][-1]
    n = int(b.obs_nvert[s, j])
    poly, dims = b.obs_poly.copy(), b.obs_dims.copy()
    if what == "clockwise":
        poly[s, j, :n] = poly[s, j, :n][::-1]
    elif what == "reflex":
        poly[s, j, 1] = 0.25 * (poly[s, j, 0] + poly[s, j, 2]) + 0.5 * poly[s, j, (n // 2 + 1) % n]  # pulled inside the ring
    elif what == "box":
        dims[s, j] *= 0.5
    else:
        poly[s, j, 0, 1] = np.nan
    b.obs_poly, b.obs_dims = poly, dims
    db = DeviceBatch(b, 0)
    bi = torch.empty(4, dtype=torch.int32, device="cuda:0"); bc = torch.empty(4, dtype=torch.float64, device="cuda:0")
    engine.plan_dense_device(db.params, db.fb, bi.data_ptr(), bc.data_ptr())  # unvalidated: runs (the result for that ring is undefined)
    torch.cuda.synchronize()
    engine.set_option("validate", 1)
    try:
        with pytest.raises(_abi.FrenetGpuError, match=match):
            engine.plan_dense_device(db.params, db.fb, bi.data_ptr(), bc.data_ptr())
    finally:
        engine.set_option("validate", 0)


def test_closed_loop_on_polygon_scenes_one_launch_equals_two(engine):
    """fp_plan_step (plan + hand-over in one launch, the run-time-shape POLY instances) against fp_plan_dense + fp_advance, cycle for
    cycle, on scenes whose obstacles are half polygons."""
    from fiss_plus_planner_amd.device_batch import ClosedLoopRunner, DeviceBatch

    goal = np.full((300, 2), 1e9)
    mk = lambda: with_random_shapes(synth.make_batch(300, 5, 5, 5, 10, 100, True, 67), 67)
    one = ClosedLoopRunner(engine, DeviceBatch(mk(), 0), goal, "FOP", fused=True).run(12)
    two = ClosedLoopRunner(engine, DeviceBatch(mk(), 0), goal, "FOP", fused=False).run(12)
    for k in ("done", "cycles", "t_now"):
        np.testing.assert_array_equal(getattr(one, k), getattr(two, k), err_msg=k)
    assert np.array_equal(one.ego, two.ego) and np.array_equal(one.cart, two.cart, equal_nan=True)
    assert one.cycles.sum() > 300


def test_audit_contact_bit_on_a_polygon_within_ulps_of_touching(engine):
    tri = np.array([(-2.0, -2.0), (2.0, -2.0), (0.0, 2.0)])
    ego = (4.0, 2.0, 0.0, 0.0, 0.0)
    for p, thin in ((3.5, True), (3.5 + 2.0 ** -40, True), (3.5 - 2.0 ** -40, True), (3.6, False), (3.0, False)):
        out = engine.plan_dense(_one_pose_polygon_scene(ego, tri, (p, 0.0, 0.0)), tables=True, audit=True)
        assert bool(out.audit[0] & _abi.AUDIT_CONTACT) is thin, p


def test_sharded_engine_and_resident_batches_carry_the_polygons(engine):
    from fiss_plus_planner_amd.sharded import ShardedEngine

    b = with_random_shapes(synth.make_batch(30, 5, 5, 5, 10, 100, True, 65), 65)
    ref = engine.plan_dense(b, tables=False)
    with ShardedEngine(devices=[0], shards_per_device=3) as se:
        out = se.plan_dense(b)
        np.testing.assert_array_equal(out.best_idx, ref.best_idx)
        res = se.plan_dense(se.upload(b))
        np.testing.assert_array_equal(res.best_idx, ref.best_idx)
