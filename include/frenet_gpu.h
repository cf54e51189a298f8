/*
 * frenet_gpu.h - C ABI of libfrenetgpu.so, the MI355X (gfx950) Frenet trajectory
 * sampling-and-scoring engine.
 *
 * This is the drop-in boundary for the candidate-generation hot path of
 * SS47816/fiss_plus_planner.  The reference is pure Python and has no FFI of its
 * own; each entry point below names the reference code it replaces (paths relative
 * to the reference checkout).  Bindings: ctypes (fiss_plus_planner_amd/_abi.py);
 * the stub a maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a negative FP_E* code on failure;
 *     fp_last_error() returns a thread-local description.  Nothing throws.
 *   - all arrays are contiguous, little-endian, float64 / int32 / uint32.
 *   - `mem` says where EVERY pointer of the call lives: FP_MEM_HOST (the library
 *     stages through device buffers it owns inside the ctx, runs, copies back and
 *     synchronises before returning) or FP_MEM_DEVICE (pointers are device
 *     addresses on the ctx's GPU; the call only enqueues work on `stream` and
 *     returns; no synchronisation and no allocation once the ctx's scratch buffers
 *     have reached the size the batch needs, i.e. after the first call of that size).
 *     FP_MEM_DEVICE calls cannot look at the arrays: frame_of / scene_of / nx / t_now / best_idx are NOT range-checked (an
 *     out-of-range index is a GPU memory fault, not FP_EINVAL - unless fp_ctx_set_option("validate", 1) is on), and a time sample that needs more than FP_MAX_POINTS points
 *     yields NaN cost + FP_FLAG_INFEASIBLE for its candidates instead of FP_ELIMIT.  Validate on the host, or use FP_MEM_HOST.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - a ctx is bound to one device; calls on one ctx must not overlap in time, and the work they enqueue must not either: use
 *     one stream per ctx at a time (the ctx keeps scratch buffers - partial results, launch order - that successive calls reuse
 *     in stream order).  FP_MEM_DEVICE calls may allocate or grow such a scratch buffer on first use (never inside a stream capture
 *     after a warm-up call of the same size).  A captured graph holds the addresses of those buffers: a later call on the same ctx
 *     with a LARGER batch may reallocate them - capture again after such a call.
 */
#ifndef FRENET_GPU_H
#define FRENET_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FP_ABI_VERSION 15

/* error codes */
#define FP_OK 0
#define FP_EINVAL (-1)  /* bad argument */
#define FP_EHIP (-2)    /* HIP runtime error (see fp_last_error) */
#define FP_ENOMEM (-3)
#define FP_ELIMIT (-4)  /* problem exceeds a compiled-in limit (FP_MAX_*) */
#define FP_ENODEV (-5)  /* no usable gfx950 device */

/* memory space of the pointers of a call */
#define FP_MEM_HOST 0
#define FP_MEM_DEVICE 1

/* compiled-in limits */
#define FP_MAX_POINTS 256   /* N = ceil(T / tick_t) per trajectory (T = 10 s at tick_t = 0.05 s: 200).  Up to FP_FAST_POINTS every
                               kernel runs its two-points-per-lane fast path; beyond it the series come from a chunked writer, the
                               epilogue workgroups and the per-profile materialiser give way to winner_traj_kernel, and the FISS+
                               refinement runs its four-points-per-lane instance (two workgroups per CU instead of three) */
#define FP_FAST_POINTS 128
#define FP_DEFAULT_STRIDE 128 /* columns of a series row when traj_stride = 0 */
#define FP_MAX_KNOTS 1024   /* reference-line knots per frame (72 bytes of LDS each in every kernel that walks the line) */
#define FP_MAX_CAND 16384   /* nd*nv*nt of the dense pass (fp_plan_dense and everything built on its tables) */
#define FP_MAX_CAND_SEARCH 4096 /* nd*nv*nt of the device-side FISS / FISS+ walks (fp_plan_fiss: rank bit sets of 64 x 64 bits); above
                                   it FP_ELIMIT - walk the dense tables on the host (fiss_plus_planner_amd/search.py), as the drop-in
                                   planner classes then do by themselves */
#define FP_MAX_POLY_VERTS 128 /* vertices of one convex-polygon obstacle column (shapely's buffer() circle has 64) */

/* candidate flag word: low bits = why a candidate is infeasible, then N and M */
#define FP_FLAG_SPEED 1u      /* any(s_d > max_speed)       frenet_optimal_planner.py:152 */
#define FP_FLAG_ACCEL 2u      /* any(|s_dd| > max_accel)    frenet_optimal_planner.py:155 */
#define FP_FLAG_COLLISION 4u  /* has_collision()            frenet_optimal_planner.py:168-195 */
#define FP_FLAG_TRUNCATED 8u  /* M < N: left the spline     frenet_optimal_planner.py:112-113 */
/* the three checks check_constraints carries commented out (frenet_optimal_planner.py:145-150); set only when
 * fp_params.curvature_mask != 0 */
#define FP_FLAG_CURVATURE 16u /* any(|c| > max_curvature)    :145-146 */
#define FP_FLAG_KAPPA_D 32u   /* any(|c_d| > max_kappa_d)    :147-148 */
#define FP_FLAG_KAPPA_DD 64u  /* any(|c_dd| > max_kappa_dd)  :149-150 */
#define FP_FLAG_CONSTRAINTS (FP_FLAG_SPEED | FP_FLAG_ACCEL | FP_FLAG_CURVATURE | FP_FLAG_KAPPA_D | FP_FLAG_KAPPA_DD) /* check_constraints */
#define FP_FLAG_INFEASIBLE (FP_FLAG_CONSTRAINTS | FP_FLAG_COLLISION)
#define FP_FLAG_N_SHIFT 8     /* bits 8..19  N = len(t) */
#define FP_FLAG_M_SHIFT 20    /* bits 20..31 M = len(x) */

/* rows of a [16][stride] trajectory dump = FrenetTrajectory series, frenet.py:129-146 */
enum {
    FP_ARR_T = 0, FP_ARR_S, FP_ARR_S_D, FP_ARR_S_DD, FP_ARR_S_DDD, FP_ARR_D, FP_ARR_D_D, FP_ARR_D_DD, FP_ARR_D_DDD,
    FP_ARR_X, FP_ARR_Y, FP_ARR_YAW, FP_ARR_DS, FP_ARR_C, FP_ARR_C_D, FP_ARR_C_DD, FP_ARR_COUNT
};

typedef struct fp_ctx fp_ctx;

/* Settings + vehicle + cost weights.
 * Replaces: FrenetOptimalPlannerSettings (frenet_optimal_planner.py:38-56), Vehicle
 * (common/vehicle/vehicle.py:15-46), CostFunction("WX1") weights (common/cost/cost_function.py:6-12). */
typedef struct {
    int32_t nd, nv, nt;     /* num_width, num_speed, num_t */
    int32_t check_stride;   /* check_res of has_collision (2)              :202 */
    double tick_t;          /* 0.1 */
    double cost_horizon;    /* the 10.0 of `cost_time = 10.0 - t[-1]`      cost_function.py:42 */
    double w_speed, w_accel, w_jerk, w_offset; /* w_V=1, w_A=0.1, w_J=0.1, w_LC=10 */
    double veh_l, veh_w;    /* ego footprint */
    double max_speed, max_accel;
    /* Optional curvature checks: the reference has them in check_constraints but commented out (:145-150), so 0 = off is the
     * reference's behaviour.  != 0: a candidate whose Cartesian series has any |c| > max_curvature, |c_d| > max_kappa_d or
     * |c_dd| > max_kappa_dd (c = diff(yaw)/ds, c_d = diff(c)/tick_t, c_dd = diff(c_d)/tick_t, :131-134) fails the constraints
     * like a speed / acceleration violation does.  Limits: Vehicle.max_curvature / max_kappa_d / max_kappa_dd (vehicle.py:44-46). */
    int32_t curvature_mask;
    /* FP_MEM_DEVICE calls only (FP_MEM_HOST calls look at t_samples / samp_max themselves): an upper bound of the points per trajectory,
     * ceil(T / tick_t) over every time sample the call can generate.  0 = at most FP_FAST_POINTS (128); a caller whose batch needs
     * more says so here (<= FP_MAX_POINTS) and gets the kernels that can do it.  A device batch that needs more points than announced
     * yields NaN cost + FP_FLAG_INFEASIBLE for those candidates, as beyond FP_MAX_POINTS. */
    int32_t points_max;
    double max_curvature, max_kappa_d, max_kappa_dd;
} fp_params;

/* A batch of B independent ego planning problems (layout: DESIGN.md "problem batch"). */
typedef struct {
    int32_t B, F, NX, S, T_obs, n_obs;
    const double* d_samples;     /* [nd]        np.linspace(-sw/2, sw/2, nd)      :75 */
    const double* t_samples;     /* [nt]        np.linspace(min_t, max_t, nt)     :78 */
    const double* v_samples;     /* [B][nv]     np.linspace(lowest, highest, nv)  :89 */
    const double* target_speed;  /* [B]         settings.highest_speed            :250 */
    const double* ego;           /* [B][6]      s, s_d, s_dd, d, d_d, d_dd        frenet.py:15-27 */
    const int32_t* frame_of;     /* [B]         index into the frame table */
    const int32_t* scene_of;     /* [B]         index into the scene table, -1 = no obstacles */
    const int32_t* t_now;        /* [B]         time_step_now */
    const int32_t* nx;           /* [F]         knots per frame */
    const double* knots;         /* [F][NX]     cumulative chord length, +inf padded   cubic_spline.py:162-168 */
    const double* coef;          /* [F][8][NX]  ax bx cx dx ay by cy dy                cubic_spline.py:30-43 */
    const double* obs_pose;      /* [S][T_obs][n_obs][4]  x, y, yaw, valid(0/1) */
    const double* obs_dims;      /* [S][n_obs][2]         length, width (a polygon column: the centred box that contains it, see obs_poly) */
    const int32_t* final_time_step; /* [S]      obstacles[0].prediction.final_time_step :173 */
    const int32_t* skip;         /* NULL or [B]: egos with skip[b] != 0 are not planned (best_idx = -1, no table rows written);
                                    the closed-loop driver passes its `done` array here */
    /* FP_MEM_HOST only, 0 = off.  A non-zero tag is the caller's promise that the frame and scene tables of this call (nx, knots,
     * coef, obs_pose, obs_dims, final_time_step - with F, NX, S, T_obs, n_obs) are byte for byte those of the previous
     * FP_MEM_HOST call on this ctx that carried the same tag: the library then keeps them on the device and uploads only the
     * per-ego arrays (a planner re-plans against the same centerline and obstacle predictions cycle after cycle; for one ego that
     * is ~100 KB per call that need not travel, ~12 us of a ~43 us call).  A new tag (or changed sizes) uploads again; a ctx keeps
     * the tables of its four most recently used tags.  Ignored by FP_MEM_DEVICE calls. */
    int32_t tables_tag;
    /* Obstacle shapes other than rectangles (ABI 12).  has_collision hands `obstacle.obstacle_shape.shapely_object` - ANY polygon - to
     * construct_polygon and Polygon.intersects (frenet_optimal_planner.py:162-166, :189-191).  obs_nvert == NULL: every column is the
     * rectangle of obs_dims.  Else obs_nvert[s][j] = 0 for a rectangle, or the number of vertices (3 .. poly_stride) of a CONVEX
     * polygon whose ring is obs_poly[s][j][0 .. nvert): (x, y) relative to the column's rotation centre - the centre of the shape's
     * bounding box, about which affinity.rotate(origin='center') turns it; the column's poses carry that centre - in COUNTER-CLOCKWISE
     * order, closing vertex not repeated.  At a pose (x, y, yaw) vertex i sits at (x, y) + R(yaw) (u_x, u_y).  obs_dims of such a column
     * MUST be (2 max|u_x|, 2 max|u_y|) or larger (the broad phases test that box; FP_MEM_HOST calls check it).  A circle is the polygon
     * shapely's buffer() makes of it; a non-convex shape or a group of shapes is several columns with the same poses, one convex piece
     * each, every piece relative to the WHOLE shape's centre (fiss_plus_planner_amd/obstacles.py does all of this).  Part of the
     * scene tables as far as tables_tag is concerned.
     * The narrow phase tests a ring as the intersection of its edge half-planes: a ring that is NOT convex, NOT counter-clockwise or
     * that reaches outside obs_dims is tested as a smaller shape than it is (missed collisions; a clockwise ring never collides).
     * FP_MEM_HOST calls reject such columns with FP_EINVAL; FP_MEM_DEVICE calls do so only with fp_ctx_set_option("validate", 1) -
     * rings handed over in device memory without it are the caller's responsibility, and the result for a bad ring is undefined. */
    int32_t poly_stride;          /* vertices per column of obs_poly (<= FP_MAX_POLY_VERTS); ignored when obs_nvert == NULL */
    const double* obs_poly;       /* NULL or [S][n_obs][poly_stride][2] */
    const int32_t* obs_nvert;     /* NULL or [S][n_obs] */
    /* Launch-order hint (ABI 14), FP_MEM_DEVICE calls: NULL, or a permutation of 0 .. B-1 in device memory - the egos in the order their
     * workgroups should START when the batch has more egos than the device holds workgroups at once (the launch ends on the egos that
     * start last: put the ones expected to take longest first; ego speed, descending, is a good stand-in - a faster ego reaches more
     * obstacle rows - and is what fiss_plus_planner_amd.device_batch.DeviceBatch passes, computed once at upload).  Results do not
     * depend on it; entries outside 0 .. B-1 or repeated entries are the caller's responsibility (fp_ctx_set_option("validate", 1)
     * checks them).  NULL: the library orders a launch by the durations an earlier launch of the ctx left behind ("lattice_order").
     * Ignored by FP_MEM_HOST calls. */
    const int32_t* launch_order;
} fp_batch;

/* Outputs of the dense lattice pass; any pointer except best_idx/best_cost may be NULL.
 * best_traj requires best_flags. */
typedef struct {
    int32_t* best_idx;   /* [B]     flat FOP index (i_d*nt+i_T)*nv+i_v of the argmin, -1 = no survivor  :263-268 */
    double* best_cost;   /* [B]     its cost_final (NaN when -1) */
    double* cost_tbl;    /* [B][C]  cost_final of every candidate                                       :99 */
    uint32_t* flag_tbl;  /* [B][C]  FP_FLAG_* | N << 8 | M << 20 */
    int32_t* stats;      /* [B][4]  num_iter, generated, validated, collision_checks                    :254-256 */
    uint32_t* best_flags; /* [B]    flag word (N, M) of the argmin, 0 when best_idx = -1 */
    double* best_traj;   /* [B][16][traj_stride]  winner epilogue: the argmin's full FrenetTrajectory series (NaN padded;
                            all NaN when best_idx = -1) = what plan() returns                           :264-270 */
    int32_t* fopplus;    /* NULL or [B][2]: FopPlusPlanner.plan for every ego (fop_plus_planner.py:16-41: candidates validated lazily in
                            cost order, the first feasible one wins).  [b][0] = candidates popped up to and including the winner
                            (all C when nothing is feasible) = num_iter = validated = collision checks, also written to stats;
                            [b][1] = 1 when an exact cost tie (or a NaN cost) at the decision point leaves the outcome to the
                            reference's heap order - replay that ego on the host (fiss_plus_planner_amd/search.py:fopplus_search) -
                            else 0: best_idx / best_cost ARE FopPlusPlanner's answer */
    uint32_t* audit;     /* NULL or [B]: how thin the margins under this ego's answer are (FP_AUDIT_* bits, see below).  Asking for it
                            adds a pass over the ego's tables (one more launch) and, where FP_AUDIT_NEAR_TIE is found, settles the
                            choice among the tied candidates by the reference's own point-by-point sums. */
    int32_t traj_stride; /* columns per series row; 0 = FP_DEFAULT_STRIDE.  Must be >= the largest N = ceil(T / tick_t) of the batch
                            (e.g. 100 for T <= 10 s at 0.1 s): a smaller stride is FP_EINVAL (host) / truncates the rows (device) */
    int32_t traj_sparse; /* 0: every element of the [16][traj_stride] block is written (NaN where a row has no element).
                            1: only the rows' leading elements are written - row r gets its len(r) elements (N for the Frenet rows,
                            M / M-1 / M-2 / M-3 for x y yaw / ds c / c_d / c_dd) plus NaN up to the next multiple of 16 columns (so
                            that, with a stride that is a multiple of 16, only whole 128-byte lines are stored); the rest of the
                            block and the blocks of egos without a winner (best_flags = 0) are left untouched.  Bytes written =
                            the algorithmic bytes rounded up to lines: ~6 % more at N ~ 90.  Use traj_stride = 112 for T <= 10 s */
} fp_result;

/* fp_result.audit bits.  The kernels sum the cost terms in closed form (~1e-12 from the reference's point-by-point sums) and decide box
 * overlaps in fp64 (they can differ from shapely / GEOS's exact predicates for boxes closer than ~1e-13 m to touching), so "the
 * selected index is exact" holds with a margin.  These bits say when the margin is gone:
 *   FP_AUDIT_NEAR_TIE    another FEASIBLE candidate's cost lies within FP_AUDIT_COST_TOL of the winner's.  The tied candidates (and
 *                        the winner) were then re-priced with the reference's summation (one term per trajectory point, in order)
 *                        and the argmin rule (:263-268, last minimum wins) applied to those sums; best_idx / best_cost are the outcome.
 *   FP_AUDIT_REORDERED   ... and that changed the winner.
 *   FP_AUDIT_CONTACT     the collision verdict of the winner, or of a candidate at most as expensive that was rejected ONLY for
 *                        colliding, hangs on a pair of boxes within FP_AUDIT_GAP_TOL metres of touching (their deepest overlap over
 *                        all checked poses and obstacles is shallower than that, or their closest miss nearer): a different
 *                        rounding of the same geometry - GEOS, another compiler - could decide this ego differently.
 *   FP_AUDIT_TIES_OVERFLOW  more than 63 candidates were within the tolerance: the first 63 in index order (and the winner) were
 *                        re-priced, the others were not - the outcome is deterministic but not settled among all of them.
 * best_cost of an ego whose FP_AUDIT_NEAR_TIE bit is set is the POINT-BY-POINT sum of its (possibly new) winner; every other ego keeps
 * the closed-form sum (the two differ by ~1e-12: one output array, two summation orders - compare against cost_tbl with that in mind).
 * No bit set: every comparison behind best_idx was decided by more than the tolerances. */
#define FP_AUDIT_NEAR_TIE 1u
#define FP_AUDIT_CONTACT 2u
#define FP_AUDIT_REORDERED 4u
#define FP_AUDIT_TIES_OVERFLOW 8u
#define FP_AUDIT_COST_TOL 1e-9
#define FP_AUDIT_GAP_TOL 1e-9

int fp_abi_version(void);
const char* fp_last_error(void);

/* Build identity.  fp_build_flags: the diagnostic macros the library was compiled with (the Makefile's EXTRA and every FP_ABL_* /
 * FP_NO_* / stamp / counter switch of the sources) - "" for a production build; the timing ablations produce WRONG results by design,
 * so the Python binding refuses a library whose string is not empty (FP_ALLOW_DIAGNOSTIC_BUILD=1 overrides, for the profiling tools).
 * fp_build_compiler: the compiler's version string (the kernels are validated on one toolchain, see csrc/Makefile). */
const char* fp_build_flags(void);
const char* fp_build_compiler(void);

/* Number of visible HIP devices / name+arch of one (buf may be NULL). */
int fp_device_count(int* count);
int fp_device_info(int device, char* buf, int buflen, int* compute_units, int64_t* hbm_bytes);

int fp_ctx_create(int device, fp_ctx** out);
int fp_ctx_destroy(fp_ctx* ctx);

/* Diagnostic knobs.  "lattice_kernel": 0 = auto (default), 1 = lane-per-candidate kernel, 2 = fused
 * profile-sharing kernel only (fails with FP_EHIP if the problem does not fit it).  Results are identical
 * (flags / indices exactly, costs to ~1e-13); used by the A/B parity tests and by profiling.
 * "lattice_split": 0 = auto (default: an ego's time-horizon slices are spread over as many workgroups - at most one per
 * slice - as keep ALL workgroups of the launch resident at once, 2 per compute unit: latency mode for small batches), 1 = never,
 * 2 = always one workgroup per slice.  Identical results either way.
 * "lattice_group": time-horizon slices one workgroup's collision stages take per barrier interval.  0 = auto (default: one at a
 * time, except for batches of at most one ego per compute unit that "lattice_split" could only cut in two: one 1024-thread
 * workgroup per ego then takes all nt slices at once, when its LDS holds their tables), 1 = always one at a time, n >= 2 = up to n
 * at a time (as many as fit).  Identical results.
 * "lattice_tail": 0 = auto (default: a launch with more one-workgroup egos than stay resident cuts the LAST quarter round of its
 * dispatch slots in two workgroups each - slices split, ticket + merge as in latency mode - so that it does not end on whole egos
 * that started last), 1 = never, n >= 2 = the last n slots.  Identical results.
 * "lattice_occupancy": 0 = auto (default: batches of more egos than stay resident run three lattice workgroups per compute unit - four
 * when a workgroup's tables fit a quarter of the unit's LDS: rectangle scenes, e.g. a 9 x 9 x 7 lattice against 50 obstacles on
 * reference lines of up to ~80 knots - instead of two), 2 / 3 / 4 = at most that many.  Identical results.
 * "overlap": 0 (default) = every call is ordered on the caller's stream like a kernel launch.  1 = FP_MEM_DEVICE fp_plan_dense calls
 * alternate between two INTERNAL streams of the ctx, so that two consecutive independent calls run side by side on the device: the
 * draining tail of one launch (a fifth of a 2048-ego launch runs on a chip that is emptying) overlaps the ramp of the next.  Ordering
 * with "overlap" on: (1) a call starts after everything enqueued on the caller's `stream` before it; (2) when fp_plan_dense returns,
 * `stream` is ordered after the results of the PREVIOUS dense call on the ctx (and of every older one), not yet after this call's - at
 * most two calls are in flight; (3) fp_ctx_join(ctx, stream), or any other entry point of the ctx, orders `stream` after all of them;
 * (4) a call that writes an array its predecessor writes (any fp_result member equal) is not independent: it waits for the predecessor.
 * The inputs of a call must not be outputs of the call before it (dense outputs are not dense inputs; fp_plan_step / fp_plan_fiss, which
 * do feed themselves, never overlap).  Calls inside a stream capture are never overlapped.  Identical results.
 * "resident_groups": lattice workgroups the device holds at once at two per compute unit.  0 / default = 2 x the device's compute units.
 * The latency-mode split, the tail split, the three- / four-per-unit instances and the launch order key on it: set it to 2 x the units the
 * process really has when it runs under a CU mask (HSA_CU_MASK / ROC_GLOBAL_CU_MASK - the runtime still reports every unit), or lower to
 * model a smaller device (the tests run the multi-round instances on a handful of egos that way).  Even, >= 2.  Identical results.
 * "lattice_order": 1 (default) = launches with more egos than resident workgroups dispatch the egos longest-first, from the
 * durations earlier launches left behind (fetched asynchronously, sorted on the host); 0 = index order.  Identical results.
 * "refine_table_kb": LDS budget (KiB, default 96, 0 = off) of the FISS+ refinement kernel's per-ego pose-obstacle pair
 * table; scenes whose table does not fit are checked straight from the scene table.  Identical results either way.
 * "lattice_winner": who writes fp_plan_dense's best_traj: 0 = auto (the lattice workgroups themselves while one round of them
 * holds the batch; for bigger batches epilogue workgroups appended to the lattice launch, which become resident in the slots the
 * draining launch leaves empty and wait for their egos' argmins - one launch per call), 1 = always the lattice workgroups, 2 = always
 * winner_traj_kernel behind the lattice launch.  Identical results.
 * "validate": 1 = FP_MEM_DEVICE calls first run a one-lane-per-ego range check of frame_of / scene_of / t_now / nx / t_samples on the
 * device and return FP_EINVAL / FP_ELIMIT (with the offending index in fp_last_error) instead of faulting the GPU; the call then
 * WAITS for the stream (one small kernel + an 8-byte copy).  0 (default) = no check, nothing waits.
 * "fiss_jump": 1 (default) = the FISS+ search walk runs its first iteration, then jumps to the state the reference's walk has
 * when the first feasible sample becomes reachable (minimax cost level, computed in parallel) and resumes there; 0 = every
 * iteration one after the other.  Identical results and Stats.
 * "stage_kernel" (1 default), "inline_inputs" (1 default), "zero_copy_in" (0 default): how an FP_MEM_HOST call of at most 8 egos
 * moves its inputs.  stage_kernel: the pinned staging block goes to the device by one copy KERNEL instead of copy commands (a
 * copy command plus the cross-engine dependency behind it costs ~15 us of a single-ego call).  inline_inputs: fp_plan_dense and
 * fp_plan_fiss with fp_batch.tables_tag set pass the per-ego arrays inside the lattice kernel's argument block - nothing is copied at
 * all (fp_plan_fiss: the lattice kernel leaves them in device memory for the kernels behind it).
 * zero_copy_in: the kernels read inputs from the pinned host block over the link (1: the per-ego arrays of a tagged call, 2:
 * everything) - measured slower than the copy kernel, kept for experiments.  Identical results in every combination.
 * "fiss_fused": 1 (default) = a FISS+ call of more egos than stay resident runs its search walk in workgroups APPENDED to the lattice
 * launch (one per ego; they become resident in the slots the draining launch leaves empty, wait for their ego's dense tables and walk
 * them - two launches per call instead of three); 0 = the search kernel always follows in its own launch.  Identical results.
 * "fiss_stages": timing diagnostic of fp_plan_fiss, 3 (default) = the whole pipeline, 2 = stop after the search walk (no
 * refinement), 1 = stop after the dense lattice pass; with 1 or 2 the outputs of the skipped stages are NOT produced. */
int fp_ctx_set_option(fp_ctx* ctx, const char* name, int value);
/* Reads an option back, or one of the read-only counters "lattice_launches" (dense lattice launches of this ctx so far) and
 * "lattice_ordered_launches" (those dispatched in a feedback order or in the order of fp_batch.launch_order) - bench.py reports when an
 * order took effect -, "lattice_launches_2" / "_3" / "_4" (PROCESS-wide: fused lattice launches so far by workgroups per compute unit). */
int fp_ctx_get_option(fp_ctx* ctx, const char* name, int* value);
/* "overlap" = 1: orders `stream` (a hipStream_t, NULL = the default stream) after every dense call of the ctx that is still in flight on
 * its internal streams.  A no-op otherwise.  Also read-only: "overlapped_calls" = dense calls that started beside their predecessor. */
int fp_ctx_join(fp_ctx* ctx, void* stream);

/* Dense lattice pass = FrenetOptimalPlanner.plan() for B egos at once:
 *   calc_frenet_paths (:69-104) + CostFunction.cost_total (cost_function.py:41-50)
 *   + calc_global_paths (:106-138) + check_constraints (:140-160)
 *   + check_collisions/has_collision (:168-208) + argmin with the `>=` tie rule (:263-268).
 * The tables also feed the FOP+/FISS/FISS+ drop-ins (cost-ordered validation and
 * the index walks read them instead of generating candidates one by one). */
int fp_plan_dense(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_result* result, int mem,
                  void* stream);

/* Winner epilogue on its own: full series of lattice candidate best_idx[b] for every ego -> best_flags [B],
 * best_traj [B][16][traj_stride] (traj_stride / traj_sparse as in fp_result).  fp_plan_dense produces the same output inside its own launch when result.best_traj is set
 * (the workgroup that finds the argmin writes the series); exported separately for callers that pick the trajectory
 * themselves.  Replaces the object hand-back of plan() (:264-270). */
int fp_winner_trajs(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const int32_t* best_idx, uint32_t* best_flags,
                    double* best_traj, int32_t traj_stride, int32_t traj_sparse, int mem, void* stream);

/* Materialise the whole lattice: the full series of EVERY candidate of every ego, in FOP order
 *   traj [B][C][16][traj_stride] (traj_stride / traj_sparse as in fp_result), flags [B][C] (N << 8 | M << 20 | FP_FLAG_TRUNCATED; the feasibility bits
 *   come from fp_plan_dense's flag_tbl).
 * = what calc_frenet_paths + calc_global_paths leave in `all_trajs` for the visualisation (frenet_optimal_planner.py:102,
 * planners/benchmark/planning.py:336-357).  This is the one mode of the path that is HBM bound: 128 B per trajectory point. */
int fp_materialize_all(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, uint32_t* flags, double* traj, int32_t traj_stride,
                       int32_t traj_sparse, int mem, void* stream);

/* Explicit end states: K trajectories per ego with end state (d_end, v_end, T_end)
 *   = generate_trajectory_by_end_state (fiss_plus_planner.py:172-205) / generate_trajectory
 *   (fiss_planner.py:101-138) + calc_global_paths + check_constraints + has_collision.
 * end_states [B][K][3]; cost [B][K]; flags [B][K]; traj NULL or [B][K][16][traj_stride] (traj_stride / traj_sparse as in fp_result),
 * the full FrenetTrajectory series of every requested trajectory (winner epilogue,
 * FISS+ refinement, all_trajs visualisation payload).  One difference to the other entry points: with traj_sparse = 1 this one
 * writes exactly the elements that exist and NO padding up to the 16-column boundary (one lane walks a trajectory point by point);
 * a trajectory that needs more points than traj_stride gets NaN cost + FP_FLAG_INFEASIBLE. */
int fp_eval_trajs(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, int32_t K, const double* end_states,
                  double* cost, uint32_t* flags, double* traj, int32_t traj_stride, int32_t traj_sparse, int mem, void* stream);

/* ---- FISS / FISS+ for a whole batch, entirely on the device --------------------------------------------------
 * fp_plan_fiss = FissPlanner.plan (fiss_planner.py:190-270) or FissPlusPlanner.plan (fiss_plus_planner.py:61-170)
 * for B egos: dense lattice tables (fp_plan_dense kernels) -> one wavefront per ego replays the reference's
 * visiting order over the tables (initial guess :140-150, gradient walk :152-188 / frontier search
 * fiss_plus_planner.py:30-59,106-116, cost-ordered validation :229-258) -> FISS+ only: refinement
 * (gradient_decent :207-277, refine_solution :279-326; 6 probes + 1 step per round, all rounds in one launch).
 * Tie rule: exact cost ties resolve to the lower (i_d, i_v, i_t) raster index (the reference raises ValueError).
 * The wall-clock `time_limit` of refine_solution (:152-158, :293-299) is the CALLER's business: this entry point runs max_refine_iters
 * rounds; the drop-in FissPlusPlanner turns a budget that is already spent into 0 rounds (has_time_limit) or 1 (fixture G14). */
#define FP_FISS 0
#define FP_FISS_PLUS 1

typedef struct {
    int32_t kind;              /* FP_FISS or FP_FISS_PLUS */
    int32_t max_refine_iters;  /* FissPlusPlannerSettings.max_refine_iters (3); 0 = no refinement */
    double w_heuristic;        /* FissPlannerSettings.w_heuristic (10.0)          fiss_planner.py:16 */
    double decaying_factor;    /* FissPlusPlannerSettings.decaying_factor (0.5)   fiss_plus_planner.py:22 */
} fp_fiss_opts;

typedef struct {
    /* inputs, order (d, v, t): sampling_min / sampling_max / np.linspace retstep   fiss_planner.py:45-47,57-59,67-69 */
    const double* samp_min;    /* [B][3] */
    const double* samp_max;    /* [B][3] */
    const double* samp_res;    /* [B][3] */
    int32_t* prev_best_idx;    /* [B][3] in/out, -1 = None; updated to the coarse winner           :30,:252 / :140 */
    /* outputs */
    int32_t* best_ijk;         /* [B][3] coarse winner (i_d, i_v, i_t), -1 = none */
    double* best_cost;         /* [B]    cost_final of the returned trajectory (NaN = none) */
    double* end_state;         /* [B][3] (d, v, T) of the returned trajectory (refined or lattice) */
    int32_t* refined;          /* [B]    1 when a refined trajectory replaced the coarse winner (its idx is [-1,-1,-1]) */
    int32_t* stats;            /* [B][4] num_iter, generated, validated, collision_checks (incl. refinement) */
    double* trace;             /* NULL or [B][max_refine_iters*7][4] = d, v, T, cost of every refinement trajectory */
    uint32_t* best_flags;      /* NULL or [B] flag word (N, M) of the returned trajectory */
    double* best_traj;         /* NULL or [B][16][traj_stride] its full series (requires best_flags) */
    int32_t traj_stride;       /* as in fp_result (0 = FP_DEFAULT_STRIDE) */
    int32_t traj_sparse;       /* as in fp_result */
} fp_fiss_io;

/* In FP_MEM_DEVICE mode the ctx grows an internal scratch arena (dense tables, B*C*12 bytes) on first use. */
int fp_plan_fiss(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_fiss_opts* opts, const fp_fiss_io* io,
                 int mem, void* stream);

/* ---- Frenet frame construction on the device ------------------------------------------------------------------
 * fp_frames_build = CubicSpline2D.__init__ for F centerlines at once (common/geometry/cubic_spline.py:157-168 arclength
 * knots, :19-43 natural cubic spline per axis; the reference solves the dense system with np.linalg.solve, here a
 * tridiagonal Thomas sweep - same solution to ~1e-12).
 * points [F][NX][2] (rows >= n[f] ignored), n [F]  ->  knots [F][NX] (+inf padded), coef [F][8][NX]. */
int fp_frames_build(fp_ctx* ctx, int32_t F, int32_t NX, const int32_t* n, const double* points, double* knots, double* coef, int mem,
                    void* stream);

/* fp_from_state = FrenetState.from_state (common/scenario/frenet.py:32-99) for B egos: projection of a Cartesian state
 * (x, y, yaw, v) on the 0.1 m-resampled reference line of its frame (generate_frenet_frame, frenet_optimal_planner.py:272-278):
 * nearest point (first minimum), next-waypoint rule by heading, projection on the segment, sign from `wp_yaw <= x_yaw`,
 * s = polyline length up to the previous waypoint.  states [B][4]  ->  ego [B][6] = s, s_d, 0, d, d_d, 0.
 * Uses batch->F, NX, nx, knots, coef, frame_of (the other batch fields may be NULL). */
int fp_from_state(fp_ctx* ctx, const fp_batch* batch, const double* states, double* ego, int mem, void* stream);

/* ---- closed-loop stepping on the device ---------------------------------------------------------------------
 * fp_advance = the bookkeeping between two plan() calls of the reference's simulation loop
 * (planners/benchmark/planning.py:131-162): next FrenetState = point 1 of the returned trajectory
 * (frenet.py:185-196), time_step_now += 1, stop when there is no solution (:131-133), when the new position is within
 * l/2 of the goal-lanelet centre (:154-157) or within 3 m of the end of the 0.1 m-resampled reference line (:158-161).
 * goal_region.is_reached() (:150-153) is evaluated when the caller supplies the goal geometry (goal_poly below); commonroad itself is
 * not a dependency: the rule is restated from commonroad-io's GoalRegion.is_reached / Shape.contains_point (closed point-in-polygon
 * test + the optional time-step / velocity / orientation intervals of the goal state).
 * The chosen trajectory is given either as a lattice index (best_idx, FOP order) or as explicit end states
 * (end_state [B][3] = d, v, T; NaN = no solution) - exactly one of the two pointers is non-NULL.
 * Plan + advance can be enqueued back to back on one stream for as many cycles as wanted: no host round trip; fp_plan_step does
 * both in ONE launch. */
#define FP_RUNNING 0
#define FP_DONE_GOAL 1
#define FP_DONE_END_OF_LINE 2
#define FP_DONE_NO_SOLUTION 3
#define FP_DONE_GOAL_REGION 4   /* goal_region.is_reached(state)                                      planning.py:150-153 */

typedef struct {
    double* ego;            /* [B][6] in/out  (aliases fp_batch.ego) */
    int32_t* t_now;         /* [B]    in/out  (aliases fp_batch.t_now) */
    int32_t* done;          /* [B]    in/out  FP_RUNNING / FP_DONE_*  (pass it as fp_batch.skip to the plan calls) */
    int32_t* cycles;        /* [B]    in/out  number of completed plan cycles */
    const double* goal_xy;  /* [B][2] centre vertex of the goal lanelet                      planning.py:54-58 */
    double* cart_state;     /* NULL or [B][3] out: x, y, yaw of the new state                planning.py:135 */
    /* Optional goal region (NULL = the rule is skipped, as before ABI 11).  One goal state per ego: its position is a simple polygon
     * (a lanelet's left bound + reversed right bound, a rectangle's corners, ...; either orientation, closing vertex not repeated),
     * reached when the new position lies inside or ON the boundary (commonroad's Shape.contains_point is shapely's `intersects`).
     * goal_intervals (NULL or [B][6]) = the goal state's time_step / velocity / orientation intervals as lo, hi pairs; a NaN bound
     * means the goal state does not define that attribute.  time_step is compared with the cycle index i = t_now BEFORE the
     * increment (state.time_step = i, planning.py:138), velocity with s_d, orientation with yaw[1] (closed intervals). */
    const double* goal_poly;       /* NULL or [B][goal_max_vertices][2] */
    const int32_t* goal_nv;        /* [B] vertices of the ego's polygon (< 3: this ego has no goal region) */
    const double* goal_intervals;  /* NULL or [B][6] */
    int32_t goal_max_vertices;
    int32_t reserved0;
} fp_loop_io;

int fp_advance(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const int32_t* best_idx, const double* end_state,
               const fp_loop_io* io, int mem, void* stream);

/* fp_plan_step = one cycle of the simulation loop for B egos (planning.py:120-162): fp_plan_dense (FrenetOptimalPlanner.plan) and
 * fp_advance in ONE launch - the workgroup that finds an ego's argmin hands the ego over to its next state itself (io->ego / t_now /
 * done / cycles / cart_state updated in place; egos with done != FP_RUNNING are skipped: batch->skip is ignored, io->done is used).
 * result: as for fp_plan_dense (best_idx / best_cost mandatory; the tables and the winner's series optional - the series describe the
 * trajectory the ego just left behind).  A problem that does not fit the fused lattice kernel, or a multi-round launch that also asks
 * for the series, takes the two-launch path (same results).  Identical to fp_plan_dense + fp_advance in every output. */
int fp_plan_step(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_result* result, const fp_loop_io* io, int mem,
                 void* stream);

/* fp_plan_fiss_step = one cycle of the same loop with FissPlanner / FissPlusPlanner planning it: fp_plan_fiss followed by the hand-over of
 * fp_advance (the returned trajectory's end state decides; loop->done is used as batch->skip).  With refinement rounds (FP_FISS_PLUS,
 * max_refine_iters > 0) the refinement workgroup that settles an ego's trajectory hands the ego over itself - no advance launch behind
 * the pipeline; otherwise the advance kernel follows.  Identical to fp_plan_fiss + fp_advance in every output. */
int fp_plan_fiss_step(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_fiss_opts* opts, const fp_fiss_io* io,
                      const fp_loop_io* loop, int mem, void* stream);

/* ---- several devices from one host thread ----------------------------------------------------------------------
 * The path shards over independent egos, one shard per GPU, no collective (planning.py:120-162 plans its scenarios one after the
 * other; nothing is exchanged).  An 8-GPU node running 0.15 ms plan steps - or 0.07 ms closed-loop cycles - per device needs
 * 50 000-120 000 enqueued calls per second: more than one host thread issues when every call crosses an FFI.  A group owns one
 * PERSISTENT worker thread per ctx (a mailbox per worker: it spins for a few tens of microseconds after its last call, then sleeps on
 * a condition variable); fp_group_submit posts one call per ctx and returns, fp_group_wait returns when every worker has ENQUEUED its
 * call (not when the GPUs are done: synchronise the streams for that) with the first error.  FP_MEM_DEVICE calls only; the structs a
 * call points to are copied at submit time.  One submitter at a time per group; a ctx of a group must not be used from other
 * threads while the group has work in flight. */
typedef struct fp_group fp_group;

typedef struct {
    void* dst;        /* host (pinned) or device address */
    const void* src;  /* device address */
    size_t bytes;
} fp_copy;            /* hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault) on the call's stream, behind its kernels */

/* One shard's call.  Which entry point runs follows from the pointers that are set:
 *   result                        fp_plan_dense(ctx, params, batch, result, FP_MEM_DEVICE, stream)
 *   result + loop                 fp_plan_step (..., result, loop, ...)
 *   fiss_opts + fiss_io           fp_plan_fiss (..., fiss_opts, fiss_io, ...)
 *   fiss_opts + fiss_io + loop    fp_plan_fiss_step(..., fiss_opts, fiss_io, loop, ...)
 * params == NULL: nothing for this ctx in this round. */
typedef struct {
    const fp_params* params;
    const fp_batch* batch;
    const fp_result* result;
    const fp_loop_io* loop;
    const fp_fiss_opts* fiss_opts;
    const fp_fiss_io* fiss_io;
    void* stream;
    const fp_copy* copies;   /* NULL or [n_copies] (at most 8) */
    int32_t n_copies;
    int32_t reserved0;
} fp_shard_call;

int fp_group_create(fp_ctx* const* ctxs, int32_t n, fp_group** out);
int fp_group_destroy(fp_group* group);
/* calls [n], calls[i] runs on ctxs[i].  A worker that is still busy with the previous round is waited for first. */
int fp_group_submit(fp_group* group, const fp_shard_call* calls);
/* FP_OK, or the first failing worker's code (fp_last_error() then carries its message, prefixed with the shard index). */
int fp_group_wait(fp_group* group);

#ifdef __cplusplus
}
#endif
#endif /* FRENET_GPU_H */
