// frenet_kernels.hip - gfx950 kernels of the Frenet sampling-and-scoring engine.
//
// Kernel inventory (DESIGN.md has the roofline accounting of each):
//   lattice_percand_kernel   one workgroup per ego, one lane per lattice candidate: generation,
//                            cost, Frenet->Cartesian, constraint + collision flags, block argmin.
//   eval_trajs_kernel        one lane per explicit-end-state trajectory; optional full
//                            FrenetTrajectory dump (winner epilogue / FISS+ refinement / all_trajs).
//
// LDS per workgroup: the ego's reference-line spline (knots + 8 coefficient rows) and the
// rows of its obstacle table that the collision horizon can touch, converted once to
// (x, y, cos, sin) so that the per-pose test does no trigonometry.
#include "frenet_device.h"
#include "frenet_kernels.h"
#include "frenet_winner.h"
#include "frenet_advance.h"

namespace fp {

// ---------------------------------------------------------------------------
// per-workgroup shared state
// ---------------------------------------------------------------------------
struct EgoCtx {
    // start state
    double s0, s_d0, s_dd0, d0, d_d0, d_dd0;
    double target_speed;
    // spline in LDS
    SplineLds sp;
    // obstacles
    const double* obs_lds;   // [rows][n_obs][4] = x, y, cos, sin (x = NaN: no state) or nullptr
    const double* obs_dim;   // [n_obs][4] = hl, hw, bounding radius, unused
    const double* obs_glb;   // global pose table of the scene (fallback when LDS is too small)
    int n_obs, T_obs, rows, t_now, horizon_cap;  // horizon_cap = final_time_step - t_now
    size_t col0;             // first obstacle column of the ego's scene (scene * n_obs): index base of obs_nvert / obs_poly
};

struct TrajOut {
    double cost;
    uint32_t flags;  // FP_FLAG_* | N<<8 | M<<20
};

__device__ __forceinline__ int lds_layout_doubles(int nx, int n_obs, int rows)
{
    return nx * 9 + n_obs * 4 + rows * n_obs * 4;
}

// Stage spline + obstacle rows of ego `b` into LDS.  All threads of the block take part.
__device__ void stage_ego(const KernelArgs& ka, int b, double* lds, EgoCtx& e, int lds_doubles)
{
    const fp_batch& bt = ka.b;
    const int tid = threadIdx.x, nth = blockDim.x;
    const double* eg = bt.ego + (size_t)b * 6;
    e.s0 = eg[0]; e.s_d0 = eg[1]; e.s_dd0 = eg[2]; e.d0 = eg[3]; e.d_d0 = eg[4]; e.d_dd0 = eg[5];
    e.target_speed = bt.target_speed[b];
    const int f = bt.frame_of[b];
    const int nx = bt.nx[f];
    const int NX = bt.NX;
    double* knots = lds;
    double* coef = lds + nx;
    const double* gk = bt.knots + (size_t)f * NX;
    const double* gc = bt.coef + (size_t)f * 8 * NX;
    for (int i = tid; i < nx; i += nth) knots[i] = gk[i];
    for (int i = tid; i < 8 * nx; i += nth) {
        const int r = i / nx, c = i - r * nx;
        coef[r * nx + c] = gc[(size_t)r * NX + c];
    }
    e.sp.knots = knots;
    e.sp.coef = coef;
    e.sp.nx = nx;
    e.sp.ld = nx;

    const int sc = bt.scene_of[b];
    e.t_now = bt.t_now[b];
    e.n_obs = (sc >= 0) ? bt.n_obs : 0;
    e.T_obs = bt.T_obs;
    e.obs_lds = nullptr;
    e.obs_dim = nullptr;
    e.obs_glb = nullptr;
    e.rows = 0;
    e.horizon_cap = 0;
    e.col0 = (size_t)(sc >= 0 ? sc : 0) * bt.n_obs;
    if (e.n_obs > 0) {
        const int n = e.n_obs;
        e.horizon_cap = bt.final_time_step[sc] - e.t_now;
        const int stride = ka.p.check_stride;
        int hmax = e.horizon_cap < points_cap(ka.p) ? e.horizon_cap : points_cap(ka.p);
        if (hmax < 0) hmax = 0;
        int rows = (hmax + stride - 1) / stride;  // poses 0, stride, 2*stride, ... < hmax
        // rows past the end of the table hold no state at all: do not stage them
        const int in_table = bt.T_obs - e.t_now;
        const int rows_tab = in_table > 0 ? (in_table + stride - 1) / stride : 0;
        if (rows_tab < rows) rows = rows_tab;
        double* dim = coef + 8 * nx;
        const double* gd = bt.obs_dims + (size_t)sc * n * 2;
        for (int j = tid; j < n; j += nth) {
            const double hl = 0.5 * gd[2 * j], hw = 0.5 * gd[2 * j + 1];
            dim[4 * j] = hl;
            dim[4 * j + 1] = hw;
            dim[4 * j + 2] = sqrt(fma(hl, hl, hw * hw));
            dim[4 * j + 3] = 0.0;
        }
        e.obs_dim = dim;
        e.obs_glb = bt.obs_pose + (size_t)sc * bt.T_obs * n * 4;
        if (lds_layout_doubles(nx, n, rows) <= lds_doubles) {
            double* tab = dim + 4 * n;
            for (int i = tid; i < rows * n; i += nth) {
                const int r = i / n, j = i - r * n;
                const int ts = r * stride + e.t_now;
                double x = __builtin_nan(""), y = 0.0, c = 1.0, s = 0.0;
                if (ts >= 0 && ts < bt.T_obs) {
                    const double* ps = e.obs_glb + ((size_t)ts * n + j) * 4;
                    if (ps[3] != 0.0) {
                        x = ps[0];
                        y = ps[1];
                        sincos_snapped(ps[2], s, c);
                    }
                }
                tab[4 * i] = x; tab[4 * i + 1] = y; tab[4 * i + 2] = c; tab[4 * i + 3] = s;
            }
            e.obs_lds = tab;
            e.rows = rows;
        }
    }
    __syncthreads();
}

// ego box at pose index i (absolute obstacle row i + t_now) against every obstacle
__device__ __forceinline__ bool pose_collides(const KernelArgs& ka, const EgoCtx& e, int i, double x, double y, double c, double s)
{
    if (!(x == x) || !(y == y) || !(c == c)) return true;  // polygon construction fails in the reference -> "collision" (:178-182)
    Obb ego{x, y, c, s, 0.5 * ka.p.veh_l, 0.5 * ka.p.veh_w};
    const double r_e = sqrt(fma(ego.hl, ego.hl, ego.hw * ego.hw));
    const int n = e.n_obs;
    const int stride = ka.p.check_stride;
    if (e.obs_lds) {
        if (i / stride >= e.rows) return false;  // beyond the table: state_at_time() is None for every obstacle
        const double* row = e.obs_lds + (size_t)(i / stride) * n * 4;
        for (int j = 0; j < n; ++j) {
            const double ox = row[4 * j], oy = row[4 * j + 1];
            const double R = (r_e + e.obs_dim[4 * j + 2]) * (1.0 + 1e-12);
            const double dx = ox - x, dy = oy - y;
            if (!(fma(dx, dx, dy * dy) <= R * R)) continue;  // also skips NaN (no state)
            Obb ob{ox, oy, row[4 * j + 2], row[4 * j + 3], e.obs_dim[4 * j], e.obs_dim[4 * j + 1]};
            if (shape_overlap(ego, ob, ka.b.obs_nvert, ka.b.obs_poly, ka.b.poly_stride, e.col0 + j)) return true;
        }
    } else {
        const int ts = i + e.t_now;
        if (ts < 0 || ts >= e.T_obs) return false;
        const double* row = e.obs_glb + (size_t)ts * n * 4;
        for (int j = 0; j < n; ++j) {
            if (row[4 * j + 3] == 0.0) continue;
            const double ox = row[4 * j], oy = row[4 * j + 1];
            const double R = (r_e + e.obs_dim[4 * j + 2]) * (1.0 + 1e-12);
            const double dx = ox - x, dy = oy - y;
            if (!(fma(dx, dx, dy * dy) <= R * R)) continue;
            double oc, os;
            sincos_snapped(row[4 * j + 2], os, oc);
            Obb ob{ox, oy, oc, os, e.obs_dim[4 * j], e.obs_dim[4 * j + 1]};
            if (shape_overlap(ego, ob, ka.b.obs_nvert, ka.b.obs_poly, ka.b.poly_stride, e.col0 + j)) return true;
        }
    }
    return false;
}

// One trajectory: generation + cost + conversion + flags (+ optional dump).
// `dump` = nullptr or the [16][stride] block of this trajectory.
template <bool DUMP, bool CURV>
__device__ TrajOut traj_eval(const KernelArgs& ka, const EgoCtx& e, double d_end, double v_end, double T_end, bool do_collision,
                             double* dump, int stride_d)
{
    const fp_params& p = ka.p;
    const int N = arange_len(T_end, p.tick_t);
    TrajOut out;
    if (N <= 0 || N > points_cap(p) || (DUMP && N > stride_d)) {  // (a dump row holds stride_d points)
        out.cost = __builtin_nan("");
        out.flags = FP_FLAG_SPEED | FP_FLAG_ACCEL | FP_FLAG_COLLISION;  // "no trajectory"
        return out;
    }
    const Quintic lat = quintic_bvp(e.d0, e.d_d0, e.d_dd0, d_end, 0.0, 0.0, T_end);
    const Quartic lon = quartic_bvp(e.s0, e.s_d0, e.s_dd0, v_end, 0.0, T_end);
    double sum_v = 0, sum_as = 0, sum_ad = 0, sum_js = 0, sum_jd = 0, sum_d = 0;
    uint32_t flags = 0;
    int M = -1, seg = -1;
    double xp = 0, yp = 0, hc = 1.0, hs = 0.0;
    const bool check = do_collision && e.n_obs > 0;
    const int cstride = p.check_stride;
    bool hit = false;
    for (int i = 0; i < N; ++i) {
        const double t = (double)i * p.tick_t;
        double d, d_d, d_dd, d_ddd, s, s_d, s_dd, s_ddd;
        quintic_eval(lat, t, d, d_d, d_dd, d_ddd);
        quartic_eval(lon, t, s, s_d, s_dd, s_ddd);
        const double ev = s_d - e.target_speed;
        sum_v = fma(ev, ev, sum_v);
        sum_as = fma(s_dd, s_dd, sum_as);
        sum_ad = fma(d_dd, d_dd, sum_ad);
        sum_js = fma(s_ddd, s_ddd, sum_js);
        sum_jd = fma(d_ddd, d_ddd, sum_jd);
        sum_d = fma(d, d, sum_d);
        if (s_d > p.max_speed) flags |= FP_FLAG_SPEED;
        if (fabs(s_dd) > p.max_accel) flags |= FP_FLAG_ACCEL;
        if (DUMP) {
            dump[FP_ARR_T * stride_d + i] = t;
            dump[FP_ARR_S * stride_d + i] = s; dump[FP_ARR_S_D * stride_d + i] = s_d;
            dump[FP_ARR_S_DD * stride_d + i] = s_dd; dump[FP_ARR_S_DDD * stride_d + i] = s_ddd;
            dump[FP_ARR_D * stride_d + i] = d; dump[FP_ARR_D_D * stride_d + i] = d_d;
            dump[FP_ARR_D_DD * stride_d + i] = d_dd; dump[FP_ARR_D_DDD * stride_d + i] = d_ddd;
        }
        if (M < 0) {
            seg = spline_segment(e.sp, s, seg);
            if (seg < 0) {
                M = i;  // calc_position -> None: truncate (frenet_optimal_planner.py:112-113)
            } else {
                double px, py, tx, ty, x, y;
                spline_frame(e.sp, seg, s - e.sp.knots[seg], px, py, tx, ty);
                frenet_to_cartesian(px, py, tx, ty, d, x, y);
                if (DUMP) { dump[FP_ARR_X * stride_d + i] = x; dump[FP_ARR_Y * stride_d + i] = y; }
                if (i >= 1) {
                    const double ddx = x - xp, ddy = y - yp;
                    step_heading(ddx, ddy, hc, hs);
                    if (DUMP) {
                        dump[FP_ARR_YAW * stride_d + i - 1] = atan2(ddy, ddx);
                        dump[FP_ARR_DS * stride_d + i - 1] = hypot(ddx, ddy);
                    }
                    const int k = i - 1;  // pose k now has its heading
                    if (check && !hit && (k % cstride) == 0 && k < e.horizon_cap) hit = pose_collides(ka, e, k, xp, yp, hc, hs);
                }
                xp = x;
                yp = y;
            }
        }
    }
    if (M < 0) M = N;
    if (check && !hit) {
        const int k = M - 1;  // last pose repeats the previous heading (:129)
        if (M >= 2) {
            if ((k % cstride) == 0 && k < e.horizon_cap) hit = pose_collides(ka, e, k, xp, yp, hc, hs);
        } else if (M == 1 && e.horizon_cap >= 1) {
            hit = true;  // traj.yaw is empty -> IndexError -> bare except -> collision (:178-182)
        }
    }
    if (hit) flags |= FP_FLAG_COLLISION;
    if (M < N) flags |= FP_FLAG_TRUNCATED;
    if (CURV) flags |= curvature_flags(p, e.sp, lon, lat, N);  // optional checks (:145-150), a second pass over the points
    // cost_function.py:41-50, same grouping
    const double cost_time = p.cost_horizon - (double)(N - 1) * p.tick_t;
    const double cost_speed = p.w_speed * sum_v;
    const double cost_accel = p.w_accel * sum_as + p.w_accel * sum_ad;
    const double cost_jerk = p.w_jerk * sum_js + p.w_jerk * sum_jd;
    const double cost_offset = p.w_offset * sum_d;
    out.cost = (cost_time + 0.0 + cost_speed + cost_accel + cost_jerk + cost_offset) / (double)N;
    out.flags = flags | ((uint32_t)N << FP_FLAG_N_SHIFT) | ((uint32_t)M << FP_FLAG_M_SHIFT);
    if (DUMP && M >= 2) {
        // yaw[M-1] = yaw[M-2]; c = diff(yaw)/ds; c_d = diff(c)/dt; c_dd = diff(c_d)/dt  (:127-134)
        double* yaw = dump + FP_ARR_YAW * stride_d;
        double* ds = dump + FP_ARR_DS * stride_d;
        double* c = dump + FP_ARR_C * stride_d;
        double* c_d = dump + FP_ARR_C_D * stride_d;
        double* c_dd = dump + FP_ARR_C_DD * stride_d;
        yaw[M - 1] = yaw[M - 2];
        for (int i = 0; i < M - 1; ++i) c[i] = (yaw[i + 1] - yaw[i]) / ds[i];
        for (int i = 0; i < M - 2; ++i) c_d[i] = (c[i + 1] - c[i]) / p.tick_t;
        for (int i = 0; i < M - 3; ++i) c_dd[i] = (c_d[i + 1] - c_d[i]) / p.tick_t;
    }
    return out;
}

// ---------------------------------------------------------------------------
// dense lattice, one lane per candidate
// ---------------------------------------------------------------------------
// (at most 512 threads: 2 wavefronts per SIMD, so the per-point loop may keep up to 256 VGPRs - no scratch)
template <bool CURV>
__global__ __launch_bounds__(512) void lattice_percand_kernel(KernelArgs ka, int lds_doubles)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ Best wave_best_slot[16];
    const int b = blockIdx.x;
    const fp_params& p = ka.p;
    const int C = p.nd * p.nv * p.nt;
    if (ka.b.skip && ka.b.skip[b]) {  // finished ego of a closed-loop batch
        if (threadIdx.x == 0) { ka.r.best_idx[b] = -1; ka.r.best_cost[b] = __builtin_nan(""); if (ka.idx_shadow) ka.idx_shadow[b] = -1; }
        return;
    }
    EgoCtx e;
    stage_ego(ka, b, lds, e, lds_doubles);
    const double* vs = ka.b.v_samples + (size_t)b * p.nv;
    Best mine{0.0, -1};
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int iv = c % p.nv, it = (c / p.nv) % p.nt, id = c / (p.nv * p.nt);
        const TrajOut o = traj_eval<false, CURV>(ka, e, ka.b.d_samples[id], vs[iv], ka.b.t_samples[it], true, nullptr, 0);
        if (ka.r.cost_tbl) ka.r.cost_tbl[(size_t)b * C + c] = o.cost;
        if (ka.r.flag_tbl) ka.r.flag_tbl[(size_t)b * C + c] = o.flags;
        // `min_cost >= cost` is False for NaN: a NaN cost can never win (:266)
        if (!(o.flags & FP_FLAG_INFEASIBLE) && o.cost == o.cost) mine = best_merge(mine, Best{o.cost, c});
    }
    mine = wave_best(mine);
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    if (lane == 0) wave_best_slot[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + kWave - 1) / kWave;
        Best r = wave_best_slot[0];
        for (int w = 1; w < nw; ++w) r = best_merge(r, wave_best_slot[w]);
        ka.r.best_idx[b] = r.idx;
        if (ka.idx_shadow) ka.idx_shadow[b] = r.idx;
        ka.r.best_cost[b] = r.idx >= 0 ? r.cost : __builtin_nan("");
        if (ka.r.stats) {  // frenet_optimal_planner.py:254-256
            int32_t* st = ka.r.stats + (size_t)b * 4;
            st[0] = 0; st[1] = C; st[2] = C; st[3] = C;
        }
    }
}

// ---------------------------------------------------------------------------
// audit pass (fp_result.audit): how thin are the margins under an ego's answer?  One workgroup per ego over its dense tables.
//   (1) near ties: feasible candidates whose cost lies within FP_AUDIT_COST_TOL of the winner's.  The lattice kernel prices candidates
//       by closed-form sums (~1e-12 from the reference's point-by-point sums): among near-tied candidates the closed form may order
//       them differently than the reference does.  They - and the winner - are re-priced by traj_eval (one term per trajectory point,
//       in the reference's order) and the argmin rule (:263-268, `>=`: the last minimum wins) is applied to those sums.
//   (2) thin contacts: for the winner and for every candidate at most as expensive that was rejected ONLY for colliding, the deepest
//       overlap / closest miss over all checked poses and obstacles (obb_gap); |gap| < FP_AUDIT_GAP_TOL: the verdict hangs on the
//       last places of the geometry (GEOS's exact predicates, another compiler's rounding could decide it differently).
// Opt-in and off the throughput path: lane per candidate, sequential loops, like the lane-per-candidate kernel it shares code with.
// ---------------------------------------------------------------------------
// signed gap of the ego box at pose k against every obstacle present at that step: the smallest (most negative = deepest overlap).
// +inf when no obstacle is near; -inf when the reference would raise (non-finite pose: "collision" regardless of geometry).
__device__ double pose_min_gap(const KernelArgs& ka, const EgoCtx& e, int i, double x, double y, double c, double s)
{
    if (!(x == x) || !(y == y) || !(c == c)) return -__builtin_inf();
    Obb ego{x, y, c, s, 0.5 * ka.p.veh_l, 0.5 * ka.p.veh_w};
    const double r_e = sqrt(fma(ego.hl, ego.hl, ego.hw * ego.hw));
    const int n = e.n_obs;
    const int ts = i + e.t_now;
    double g = __builtin_inf();
    if (ts < 0 || ts >= e.T_obs) return g;
    const double* row = e.obs_glb + (size_t)ts * n * 4;
    for (int j = 0; j < n; ++j) {
        if (row[4 * j + 3] == 0.0) continue;
        const double ox = row[4 * j], oy = row[4 * j + 1];
        const double R = r_e + e.obs_dim[4 * j + 2] + 2.0 * FP_AUDIT_GAP_TOL;  // beyond it the boxes miss by more than the tolerance
        const double dx = ox - x, dy = oy - y;
        if (!(fma(dx, dx, dy * dy) <= R * R)) {
            if (!(dx == dx) || !(dy == dy)) g = -__builtin_inf();  // NaN pose: polygon construction fails -> collision
            continue;
        }
        double oc, os;
        sincos_snapped(row[4 * j + 2], os, oc);
        // (a polygon column: the larger of the two gaps - its box is a necessary condition, the polygon itself decides)
        double gj = obb_gap(ego, Obb{ox, oy, oc, os, e.obs_dim[4 * j], e.obs_dim[4 * j + 1]});
        const int nvert = ka.b.obs_nvert ? ka.b.obs_nvert[e.col0 + j] : 0;
        if (nvert > 0) gj = fmax(gj, poly_gap(ego, ox, oy, oc, os, ka.b.obs_poly + (e.col0 + j) * 2 * (size_t)ka.b.poly_stride, nvert));
        g = fmin(g, gj);
    }
    return g;
}

// smallest gap of one candidate over its checked poses (has_collision's poses, :168-195); out_of_geometry: the collision flag does not
// come from geometry (M == 1: the reference's IndexError)
__device__ double candidate_min_gap(const KernelArgs& ka, const EgoCtx& e, double d_end, double v_end, double T_end)
{
    const fp_params& p = ka.p;
    const int N = arange_len(T_end, p.tick_t);
    if (N <= 0 || N > points_cap(ka.p) || e.n_obs <= 0) return __builtin_inf();
    const Quintic lat = quintic_bvp(e.d0, e.d_d0, e.d_dd0, d_end, 0.0, 0.0, T_end);
    const Quartic lon = quartic_bvp(e.s0, e.s_d0, e.s_dd0, v_end, 0.0, T_end);
    int seg = -1, M = N;
    double xp = 0, yp = 0, hc = 1.0, hs = 0.0, g = __builtin_inf();
    const int cs = p.check_stride;
    for (int i = 0; i < N; ++i) {
        const double t = (double)i * p.tick_t;
        const double sv = fma(fma(fma(fma(lon.a4, t, lon.a3), t, lon.a2), t, lon.a1), t, lon.a0);
        seg = spline_segment(e.sp, sv, seg);
        if (seg < 0) { M = i; break; }
        const double dv = fma(fma(fma(fma(fma(lat.a5, t, lat.a4), t, lat.a3), t, lat.a2), t, lat.a1), t, lat.a0);
        double px, py, tx, ty, x, y;
        spline_frame(e.sp, seg, sv - e.sp.knots[seg], px, py, tx, ty);
        frenet_to_cartesian(px, py, tx, ty, dv, x, y);
        if (i >= 1) {
            step_heading(x - xp, y - yp, hc, hs);
            const int k = i - 1;
            if ((k % cs) == 0 && k < e.horizon_cap) g = fmin(g, pose_min_gap(ka, e, k, xp, yp, hc, hs));
        }
        xp = x; yp = y;
    }
    if (M >= 2) {
        const int k = M - 1;  // the last pose repeats the previous heading (:129)
        if ((k % cs) == 0 && k < e.horizon_cap) g = fmin(g, pose_min_gap(ka, e, k, xp, yp, hc, hs));
    } else if (M == 1 && e.horizon_cap >= 1) {
        g = -__builtin_inf();  // traj.yaw is empty -> IndexError -> collision: not a matter of geometry
    }
    return g;
}

constexpr int kAuditThreads = 256;
constexpr int kAuditTies = 64;  // near-tied candidates re-priced per ego: the first 63 in index order (more: FP_AUDIT_TIES_OVERFLOW)
__global__ __launch_bounds__(kAuditThreads) void audit_kernel(KernelArgs ka, int lds_doubles, uint32_t* audit)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int s_ties[kAuditTies + 1];
    __shared__ Best s_wbest[kAuditThreads / kWave];
    __shared__ int s_thin;
    const int b = blockIdx.x, tid = threadIdx.x;
    const fp_params& p = ka.p;
    const int C = p.nd * p.nv * p.nt;
    if (ka.b.skip && ka.b.skip[b]) {
        if (tid == 0) audit[b] = 0u;
        return;
    }
    if (tid == 0) { s_ties[kAuditTies] = 0; s_thin = 0; }
    EgoCtx e;
    stage_ego(ka, b, lds, e, lds_doubles);  // (ends with a barrier)
    const double* cost = ka.r.cost_tbl + (size_t)b * C;
    const uint32_t* flag = ka.r.flag_tbl + (size_t)b * C;
    const double* vs = ka.b.v_samples + (size_t)b * p.nv;
    const int win = ka.r.best_idx[b];
    const double cw = win >= 0 ? cost[win] : __builtin_inf();
    auto end_state = [&](int c, double& d_end, double& v_end, double& T_end) {
        const int iv = c % p.nv, it = (c / p.nv) % p.nt, id = c / (p.nv * p.nt);
        d_end = ka.b.d_samples[id]; v_end = vs[iv]; T_end = ka.b.t_samples[it];
    };
    // (1) near ties: the first kAuditTies - 1 in INDEX order, whatever the thread timing - one wavefront walks the table 64 candidates at
    // a time and compacts by ballot + popcount (an atomic counter kept a timing-dependent subset when more than 63 tied, and with it a
    // run-to-run different re-priced winner; the other wavefronts go on to the contact pass meanwhile)
    if (win >= 0 && tid < kWave) {
        int n = 0;
        for (int c0 = 0; c0 < C; c0 += kWave) {
            const int c = c0 + tid;
            const bool tie = c < C && c != win && !(flag[c] & FP_FLAG_INFEASIBLE) && cost[c] == cost[c] && fabs(cost[c] - cw) <= FP_AUDIT_COST_TOL;
            const unsigned long long m = __ballot(tie);
            if (tie) {
                const int pos = n + __popcll(m & ((1ull << tid) - 1ull));
                if (pos < kAuditTies - 1) s_ties[pos] = c;
            }
            n += __popcll(m);
        }
        if (tid == 0) s_ties[kAuditTies] = n;
    }
    // (2) thin contacts: the winner + every candidate at most as expensive that only the collision check rejected (no winner: every
    // candidate only the collision check rejected - any of them flipping would give the ego a solution)
    for (int c = tid; c < C; c += kAuditThreads) {
        const uint32_t fl = flag[c] & FP_FLAG_INFEASIBLE;
        const bool in_set = c == win || (fl == FP_FLAG_COLLISION && cost[c] == cost[c] && (win < 0 || cost[c] <= cw + FP_AUDIT_COST_TOL));
        if (!in_set) continue;
        double d_end, v_end, T_end;
        end_state(c, d_end, v_end, T_end);
        const double g = candidate_min_gap(ka, e, d_end, v_end, T_end);
        if (fabs(g) < FP_AUDIT_GAP_TOL) atomicOr(&s_thin, 1);
    }
    __syncthreads();
    const int n_ties = s_ties[kAuditTies] < kAuditTies - 1 ? s_ties[kAuditTies] : kAuditTies - 1;
    uint32_t bits = s_thin ? FP_AUDIT_CONTACT : 0u;
    if (s_ties[kAuditTies] > kAuditTies - 1) bits |= FP_AUDIT_TIES_OVERFLOW;  // (more tied candidates than are re-priced)
    if (n_ties > 0) {
        bits |= FP_AUDIT_NEAR_TIE;
        // re-price the tied candidates and the winner point by point; the argmin rule on those sums
        if (tid == 0) s_ties[n_ties] = win;
        __syncthreads();
        Best mine{0.0, -1};
        if (tid <= n_ties) {
            const int c = s_ties[tid];
            double d_end, v_end, T_end;
            end_state(c, d_end, v_end, T_end);
            const TrajOut o = traj_eval<false, false>(ka, e, d_end, v_end, T_end, false, nullptr, 0);
            if (o.cost == o.cost) mine = Best{o.cost, c};
        }
        mine = wave_best(mine);
        if ((tid & (kWave - 1)) == 0) s_wbest[tid / kWave] = mine;
        __syncthreads();
        if (tid == 0) {
            Best r = s_wbest[0];
            for (int w = 1; w < kAuditThreads / kWave; ++w) r = best_merge(r, s_wbest[w]);
            if (r.idx >= 0 && r.idx != win) {
                bits |= FP_AUDIT_REORDERED;
                ka.r.best_idx[b] = r.idx;
                if (ka.idx_shadow) ka.idx_shadow[b] = r.idx;
            }
            if (r.idx >= 0) ka.r.best_cost[b] = r.cost;  // the point-by-point sum of the (possibly new) winner
        }
    }
    if (tid == 0) audit[b] = bits;
}

// ---------------------------------------------------------------------------
// optional curvature checks of the whole lattice (fp_params.curvature_mask): one workgroup per ego, one lane per candidate,
// the ego's spline in LDS.  Runs ahead of the fused lattice kernel, which ORs the bytes into its flag words (the checks need
// every Cartesian point of every candidate - exactly the per-candidate work the fused kernel is built to avoid - so they live
// in their own launch instead of in its register budget).
// ---------------------------------------------------------------------------
__global__ void curvature_flags_kernel(KernelArgs ka, uint8_t* out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    const int b = blockIdx.x;
    const int C = p.nd * p.nv * p.nt;
    if (bt.skip && bt.skip[b]) return;
    const int f = bt.frame_of[b];
    const int nx = bt.nx[f];
    const double* gk = bt.knots + (size_t)f * bt.NX;
    const double* gc = bt.coef + (size_t)f * 8 * bt.NX;
    double* knots = lds;
    double* coef = lds + nx;
    for (int i = threadIdx.x; i < nx; i += blockDim.x) knots[i] = gk[i];
    for (int i = threadIdx.x; i < 8 * nx; i += blockDim.x) {
        const int r = i / nx, c = i - r * nx;
        coef[r * nx + c] = gc[(size_t)r * bt.NX + c];
    }
    __syncthreads();
    const SplineLds sp{knots, coef, nx, nx};
    const double* eg = bt.ego + (size_t)b * 6;
    const double* vs = bt.v_samples + (size_t)b * p.nv;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int iv = c % p.nv, it = (c / p.nv) % p.nt, id = c / (p.nv * p.nt);
        const double T = bt.t_samples[it];
        const int N = arange_len(T, p.tick_t);
        uint32_t fl = 0;
        if (N > 0 && N <= points_cap(p)) {
            const Quintic lat = quintic_bvp(eg[3], eg[4], eg[5], bt.d_samples[id], 0.0, 0.0, T);
            const Quartic lon = quartic_bvp(eg[0], eg[1], eg[2], vs[iv], 0.0, T);
            fl = curvature_flags(p, sp, lon, lat, N);
        }
        out[(size_t)b * C + c] = (uint8_t)fl;
    }
}

hipError_t launch_curvature_flags(const KernelArgs& ka, uint8_t* out, hipStream_t stream)
{
    const int C = ka.p.nd * ka.p.nv * ka.p.nt;
    int threads = ((C + kWave - 1) / kWave) * kWave;
    if (threads > 512) threads = 512;
    const int bytes = 9 * ka.b.NX * (int)sizeof(double);
    hipLaunchKernelGGL(curvature_flags_kernel, dim3(ka.b.B), dim3(threads), bytes, stream, ka, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// explicit end states, one lane per trajectory
// ---------------------------------------------------------------------------
template <bool CURV>
__global__ __launch_bounds__(256) void eval_trajs_kernel(KernelArgs ka, int K, const double* end_states, double* cost, uint32_t* flags, double* traj,
                                  int stride, int sparse, int lds_doubles)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x;
    EgoCtx e;
    stage_ego(ka, b, lds, e, lds_doubles);
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const double* es = end_states + ((size_t)b * K + k) * 3;
        TrajOut o;
        if (traj) {
            double* dump = traj + ((size_t)b * K + k) * FP_ARR_COUNT * stride;
            if (!sparse)  // sparse: only the elements that exist are written (fp_result.traj_sparse)
                for (int i = 0; i < FP_ARR_COUNT * stride; ++i) dump[i] = __builtin_nan("");
            o = traj_eval<true, CURV>(ka, e, es[0], es[1], es[2], true, dump, stride);
        } else {
            o = traj_eval<false, CURV>(ka, e, es[0], es[1], es[2], true, nullptr, 0);
        }
        if (cost) cost[(size_t)b * K + k] = o.cost;
        if (flags) flags[(size_t)b * K + k] = o.flags;
    }
}

// ---------------------------------------------------------------------------
// winner epilogue: one WAVEFRONT per trajectory (two time points per lane; no LDS staging of tables: a single trajectory touches
// ~100 spline segments, read straight through L2); kWinnerWaves trajectories per workgroup.
// ---------------------------------------------------------------------------
// all_C > 0: materialise EVERY lattice candidate (slot = (ego, candidate) in FOP order): the all_trajs payload.
constexpr int kWinnerWaves = 4;
// ALL: the materialise instance (its own register budget: the winners' instance stays at 127 VGPRs, four waves per SIMD).
#ifndef FP_WINNER_OCC
#define FP_WINNER_OCC 1
#endif
template <bool ALL>
__global__ __launch_bounds__(kWave * kWinnerWaves, ALL ? FP_WINNER_OCC : 1) void winner_traj_kernel(KernelArgs ka, const double* end_states, int all_C_arg, int n_slots,
                                                                                               int spline_in_lds)
{
    const int all_C = ALL ? all_C_arg : 0;
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    const int lane = threadIdx.x & (kWave - 1);
    const int slot = blockIdx.x * kWinnerWaves + (int)threadIdx.x / kWave;  // wave-uniform
    if (slot >= n_slots) return;
    const int b = all_C > 0 ? slot / all_C : slot;
    const double nan = __builtin_nan("");
    const int best = all_C > 0 ? slot - b * all_C : (end_states ? 0 : (ka.idx_shadow ? ka.idx_shadow : ka.r.best_idx)[b]);
    double d_end = nan, v_end = nan, T = nan;
    if (end_states) {
        d_end = end_states[(size_t)b * 3]; v_end = end_states[(size_t)b * 3 + 1]; T = end_states[(size_t)b * 3 + 2];
    } else if (best >= 0) {
        const int iv = best % p.nv, it = (best / p.nv) % p.nt, id = best / (p.nv * p.nt);
        d_end = bt.d_samples[id]; v_end = bt.v_samples[(size_t)b * p.nv + iv]; T = bt.t_samples[it];
    }
    const int f = bt.frame_of[b];
    // Materialise mode: the wavefront's own LDS copy of the ego's spline (9 NX doubles, one round of independent global reads) - the
    // segment search and the coefficient reads of the series are then LDS reads; against global memory the bisection alone is ~7
    // DEPENDENT reads per point.  Wave-private: no barrier (LDS operations of one wavefront execute in order).  The winner epilogue
    // (one trajectory per ego, every ego another spline) reads the few segments it touches from global memory instead: copying
    // the whole table doubled its traffic and cost 0.9 us of 10.7.
    extern __shared__ __attribute__((aligned(16))) unsigned char wt_smem[];
    const int NX = bt.NX;
    double* my = (double*)wt_smem + (size_t)((int)threadIdx.x / kWave) * 9 * NX;
    const double* gk = bt.knots + (size_t)f * NX;
    const double* gc = bt.coef + (size_t)f * 8 * NX;
    // more points than the one-chunk writer holds (tick_t < 0.08 s at T = 10 s): the chunked writer, straight from global memory
    const int n_pts = (best >= 0 && T == T) ? arange_len(T, p.tick_t) : 0;
    if (n_pts > kSeriesChunk && n_pts <= points_cap(p) && (d_end == d_end) && (v_end == v_end)) {
        winner_series_wave_long(ka, b, slot, d_end, v_end, T, lane, SplineLds{gk, gc, bt.nx[f], NX});
        return;
    }
    if (!spline_in_lds) {  // (a reference line too long for four LDS copies: the tables stay where they are)
        winner_series_wave(ka, b, slot, best >= 0, d_end, v_end, T, lane, SplineLds{gk, gc, bt.nx[f], NX});
        return;
    }
#if defined(FP_ABL_NO_SPLINE_COPY)  // timing ablation (garbage series after the first launch's leftovers)
    if (false) {
#else
    if (best >= 0 && T == T) {  // (wave-uniform; an ego without a winner needs no spline)
#endif
        for (int i = lane; i < 9 * NX; i += kWave) my[i] = i < NX ? gk[i] : gc[i - NX];
    }
    const SplineLds sp{my, my + NX, bt.nx[f], NX};
    winner_series_wave(ka, b, slot, best >= 0, d_end, v_end, T, lane, sp);
}

// Materialise mode, the production kernel: one wavefront per (ego, longitudinal profile) writes the series of the nd candidates
// that share the profile (profile_series_wave, frenet_winner.h).  Candidate (i_d, i_T, i_v) is block (i_d * nt + i_T) * nv + i_v of
// its ego: the wavefront's nd blocks are nt * nv blocks apart.  winner_traj_kernel<true> (a wavefront per candidate) stays as the
// A/B reference (-DFP_MAT_PER_CANDIDATE).
#ifndef FP_MAT_OCC
#define FP_MAT_OCC 1
#endif
constexpr int kMatXcdRun = 16;
__global__ __launch_bounds__(kWave * kWinnerWaves, FP_MAT_OCC) void materialize_profiles_kernel(KernelArgs ka, int n_tasks, int spline_in_lds)
{
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    const int lane = threadIdx.x & (kWave - 1);
    // XCD-aware dispatch: the hardware deals workgroups round-robin to the 8 XCDs, so with the plain mapping every XCD writes every
    // eighth 53 KB piece of the output.  Here an XCD takes runs of kMatXcdRun consecutive workgroups (about one ego's profiles): its
    // stores stream over contiguous blocks.  Measured on MI355X: padded layout 0.440 -> 0.412 ms (5.4 -> 5.8 TB/s written), compact
    // layout 0.351 -> 0.348 ms and the spread over buffer placements 10 -> 7 % (runs of 4 ... 504 workgroups all within 1 %).
    const int bid = (int)blockIdx.x;
#if defined(FP_MAT_NO_XCD)  // (A/B)
    const int wg = bid;
#else
    const int wg = ((bid / (8 * kMatXcdRun)) * 8 + (bid & 7)) * kMatXcdRun + ((bid >> 3) % kMatXcdRun);  // (the grid is a multiple of 8 runs)
#endif
    const int task = wg * kWinnerWaves + (int)threadIdx.x / kWave;  // wave-uniform
    if (task >= n_tasks) return;
    const int nq = p.nt * p.nv;
    const int b = task / nq, q = task - b * nq;
    const int it = q / p.nv, iv = q - it * p.nv;
    const double T = bt.t_samples[it], v_end = bt.v_samples[(size_t)b * p.nv + iv];
    const size_t slot0 = (size_t)b * p.nd * nq + q;
    const int f = bt.frame_of[b];
    extern __shared__ __attribute__((aligned(16))) unsigned char wt_smem[];
    const int NX = bt.NX;
    const double* gk = bt.knots + (size_t)f * NX;
    const double* gc = bt.coef + (size_t)f * 8 * NX;
    if (!spline_in_lds) {  // (a reference line too long for four LDS copies: the tables stay where they are)
        profile_series_wave(ka, b, slot0, (size_t)nq, p.nd, bt.d_samples, v_end, T, lane, SplineLds{gk, gc, bt.nx[f], NX});
        return;
    }
    // the wavefront's own LDS copy of the ego's spline (wave-private: no barrier), paid once per nd candidates
    double* my = (double*)wt_smem + (size_t)((int)threadIdx.x / kWave) * 9 * NX;
    if (T == T) {
        for (int i = lane; i < 9 * NX; i += kWave) my[i] = i < NX ? gk[i] : gc[i - NX];
    }
    profile_series_wave(ka, b, slot0, (size_t)nq, p.nd, bt.d_samples, v_end, T, lane, SplineLds{my, my + NX, bt.nx[f], NX});
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
// LDS of winner_traj_kernel: one spline copy per wavefront while four of them stay under the 64 KB a launch gets without asking
static int winner_lds_bytes(const KernelArgs& ka, bool all)
{
    const int bytes = kWinnerWaves * 9 * ka.b.NX * 8;
    return all && bytes <= 48 * 1024 ? bytes : 0;
}

hipError_t launch_winner_traj(const KernelArgs& ka, const double* end_states, hipStream_t stream)
{
    const int n = ka.b.B;
    hipLaunchKernelGGL(winner_traj_kernel<false>, dim3((n + kWinnerWaves - 1) / kWinnerWaves), dim3(kWave * kWinnerWaves), winner_lds_bytes(ka, false), stream, ka, end_states, 0, n, 0);
    return hipGetLastError();
}

hipError_t launch_materialize_all(const KernelArgs& ka, hipStream_t stream)
{
    const int C = ka.p.nd * ka.p.nv * ka.p.nt;
    const unsigned n = (unsigned)ka.b.B * (unsigned)C;
#if !defined(FP_MAT_PER_CANDIDATE)
    if (points_cap(ka.p) <= FP_FAST_POINTS) {  // (trajectories of more than 128 points: the per-candidate kernel and its chunked writer)
        const unsigned n_tasks = (unsigned)ka.b.B * (unsigned)(ka.p.nt * ka.p.nv);
        const int lds = winner_lds_bytes(ka, true);
        const unsigned n_wg = (n_tasks + kWinnerWaves - 1) / kWinnerWaves, unit = 8 * kMatXcdRun;
        hipLaunchKernelGGL(materialize_profiles_kernel, dim3((n_wg + unit - 1) / unit * unit), dim3(kWave * kWinnerWaves), lds, stream, ka, (int)n_tasks, lds > 0);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(winner_traj_kernel<true>, dim3((n + kWinnerWaves - 1) / kWinnerWaves), dim3(kWave * kWinnerWaves), winner_lds_bytes(ka, true), stream, ka, nullptr, C, (int)n, winner_lds_bytes(ka, true) > 0);
    return hipGetLastError();
}

static int ego_lds_bytes(const fp_params& p, const fp_batch& b, int max_bytes, int* lds_doubles)
{
    int hmax = points_cap(p);
    const int rows = (hmax + p.check_stride - 1) / p.check_stride;
    const long base = (long)b.NX * 9 + (long)b.n_obs * 4;
    long full = base + (long)rows * b.n_obs * 4;
    // never more rows than the table has
    const long rows_tab = ((long)b.T_obs + p.check_stride - 1) / p.check_stride;
    if (rows_tab < rows) full = base + rows_tab * b.n_obs * 4;
    long use = full * 8 <= max_bytes ? full : base;  // obstacle rows stay in HBM/L2 when they do not fit
    *lds_doubles = (int)use;
    return (int)(use * 8);
}

hipError_t launch_audit(const KernelArgs& ka, uint32_t* audit, hipStream_t stream)
{
    int lds_doubles = 0;
    const int bytes = ego_lds_bytes(ka.p, ka.b, 0, &lds_doubles)  /* spline + sizes only: the poses are read from the scene table */;
    FP_LDS_SLOTS(configured);
    hipError_t err = ensure_dynamic_lds((const void*)audit_kernel, bytes, configured);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(audit_kernel, dim3(ka.b.B), dim3(kAuditThreads), bytes, stream, ka, lds_doubles, audit);
    return hipGetLastError();
}

hipError_t launch_lattice_percand(const KernelArgs& ka, hipStream_t stream)
{
    const int C = ka.p.nd * ka.p.nv * ka.p.nt;
    int threads = ((C + kWave - 1) / kWave) * kWave;
    if (threads > 512) threads = 512;
    int lds_doubles = 0;
    const int bytes = ego_lds_bytes(ka.p, ka.b, 150 * 1024, &lds_doubles);
    FP_LDS_SLOTS(configured);
    FP_LDS_SLOTS(configured_curv);
    if (ka.p.curvature_mask) {
        hipError_t err = ensure_dynamic_lds((const void*)lattice_percand_kernel<true>, bytes, configured_curv);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(lattice_percand_kernel<true>, dim3(ka.b.B), dim3(threads), bytes, stream, ka, lds_doubles);
    } else {
        hipError_t err = ensure_dynamic_lds((const void*)lattice_percand_kernel<false>, bytes, configured);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(lattice_percand_kernel<false>, dim3(ka.b.B), dim3(threads), bytes, stream, ka, lds_doubles);
    }
    return hipGetLastError();
}


// ---------------------------------------------------------------------------
// closed-loop bookkeeping (planners/benchmark/planning.py:131-162), one lane per ego
// ---------------------------------------------------------------------------
__global__ void advance_kernel(KernelArgs ka, const int32_t* best_idx, const double* end_state, fp_loop_io io)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= ka.b.B) return;
    advance_ego(ka, b, best_idx ? best_idx[b] : -1, end_state, io);
}

hipError_t launch_advance(const KernelArgs& ka, const int32_t* best_idx, const double* end_state, const fp_loop_io& io, hipStream_t stream)
{
    const int threads = 64;
    hipLaunchKernelGGL(advance_kernel, dim3((ka.b.B + threads - 1) / threads), dim3(threads), 0, stream, ka, best_idx, end_state, io);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// FopPlusPlanner.plan (fop_plus_planner.py:16-41) from the dense tables: every candidate goes on a priority queue ordered by cost_final,
// candidates are validated in pop order and the first feasible one is the answer - i.e. the cheapest feasible candidate, which
// the lattice kernel's argmin already found, after as many pops as there are cheaper candidates (+ 1).  Only an exact cost tie at
// the decision point (or a NaN cost, whose comparisons are all false) leaves the outcome to CPython's heap order: flagged, the
// caller replays that ego on the host.  One wavefront per ego.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kWave * 4) void fopplus_count_kernel(int B, int C, const double* cost_tbl, const uint32_t* flag_tbl, const int32_t* best_idx,
                                                                   const double* best_cost, int32_t* out, int32_t* stats, const int32_t* skip)
{
    const int b = blockIdx.x * 4 + (int)threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    if (b >= B) return;
    if (skip && skip[b]) {  // a finished ego of a closed loop: the lattice kernel wrote no table rows for it; nothing popped, no tie, Stats untouched
        if (lane == 0) { out[2 * b] = 0; out[2 * b + 1] = 0; }
        return;
    }
    const int win = best_idx[b];
    const double cw = best_cost[b];
    const double* cost = cost_tbl + (size_t)b * C;
    const uint32_t* fl = flag_tbl + (size_t)b * C;
    int cheaper = 0, tied = 0, odd = 0;
    for (int c = lane; c < C; c += kWave) {
        const double v = cost[c];
        odd += !(v == v);
        if (win >= 0) {
            cheaper += v < cw;
            // another candidate at exactly the winner's cost: feasible or not, the heap decides who is popped first
            tied += v == cw && c != win;
        } else {
            odd += (fl[c] & FP_FLAG_INFEASIBLE) == 0;  // (cannot happen: a feasible candidate without an argmin)
        }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        cheaper += __shfl_xor(cheaper, off, kWave);
        tied += __shfl_xor(tied, off, kWave);
        odd += __shfl_xor(odd, off, kWave);
    }
    if (lane == 0) {
        const int popped = win >= 0 ? cheaper + 1 : C;
        out[2 * b] = popped;
        out[2 * b + 1] = (tied > 0 || odd > 0) ? 1 : 0;
        if (stats) {
            int32_t* st = stats + (size_t)b * 4;
            st[0] = popped; st[1] = C; st[2] = popped; st[3] = popped;  // fop_plus_planner.py:30-35, generated = the whole lattice (:21-23)
        }
    }
}

hipError_t launch_fopplus_count(int B, int C, const double* cost_tbl, const uint32_t* flag_tbl, const int32_t* best_idx, const double* best_cost,
                                int32_t* out, int32_t* stats, const int32_t* skip, hipStream_t stream)
{
    hipLaunchKernelGGL(fopplus_count_kernel, dim3((B + 3) / 4), dim3(kWave * 4), 0, stream, B, C, cost_tbl, flag_tbl, best_idx, best_cost, out, stats, skip);
    return hipGetLastError();
}

hipError_t launch_lattice(const KernelArgs& ka, hipStream_t stream, int which, void* part_scratch, int nsplit, bool* winner_done, const int* perm,
                          int* dur, int group, const InlineIn* inl, int tail, const FissTail* ft, bool* search_done)
{
    if (winner_done) *winner_done = false;
    if (search_done) *search_done = false;
    if (which == 1) return inl && inl->on ? hipErrorInvalidValue : launch_lattice_percand(ka, stream);
    hipError_t e = launch_lattice_fused(ka, stream, part_scratch, nsplit, winner_done, perm, dur, group, inl, tail, nullptr, ft, search_done);
    if (e == hipErrorInvalidValue && which != 2 && !(inl && inl->on)) {
        (void)hipGetLastError();
        if (winner_done) *winner_done = false;
        if (search_done) *search_done = false;
        return launch_lattice_percand(ka, stream);
    }
    return e;
}

hipError_t launch_eval_trajs(const KernelArgs& ka, int K, const double* end_states, double* cost, uint32_t* flags, double* traj,
                             int stride, int sparse, hipStream_t stream)
{
    int threads = ((K + kWave - 1) / kWave) * kWave;
    if (threads > 256) threads = 256;
    int lds_doubles = 0;
    const int bytes = ego_lds_bytes(ka.p, ka.b, 150 * 1024, &lds_doubles);
    FP_LDS_SLOTS(configured);
    FP_LDS_SLOTS(configured_curv);
    if (ka.p.curvature_mask) {
        hipError_t err = ensure_dynamic_lds((const void*)eval_trajs_kernel<true>, bytes, configured_curv);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(eval_trajs_kernel<true>, dim3(ka.b.B), dim3(threads), bytes, stream, ka, K, end_states, cost, flags, traj, stride,
                           sparse, lds_doubles);
    } else {
        hipError_t err = ensure_dynamic_lds((const void*)eval_trajs_kernel<false>, bytes, configured);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(eval_trajs_kernel<false>, dim3(ka.b.B), dim3(threads), bytes, stream, ka, K, end_states, cost, flags, traj, stride,
                           sparse, lds_doubles);
    }
    return hipGetLastError();
}

}  // namespace fp
