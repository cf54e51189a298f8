// frenet_advance.h - the hand-over of ONE ego between two plan cycles (planners/benchmark/planning.py:131-162), as a device function:
// called by advance_kernel (one lane per ego) and by the lattice kernel's last thread standing (fp_plan_step: the workgroup that
// found the argmin advances its ego itself - no second launch).
#pragma once
#include "frenet_device.h"
#include "frenet_kernels.h"

namespace fp {

// Closed point-in-polygon test = commonroad-io's Shape.contains_point (shapely `polygon.intersects(Point)`): inside or on the
// boundary.  poly: nv vertices (x, y), either orientation, closing vertex not repeated; nv < 3: no region.  Boundary first (a
// point is on an edge when the cross product vanishes and it lies inside the edge's bounding box - exact for points that are exactly
// on it, which is all a float test can promise), then the crossing number.
__device__ __forceinline__ bool point_in_polygon_closed(const double* poly, int nv, double x, double y)
{
    if (nv < 3 || !(x == x) || !(y == y)) return false;
    bool inside = false;
    for (int i = 0, j = nv - 1; i < nv; j = i++) {
        const double xi = poly[2 * i], yi = poly[2 * i + 1], xj = poly[2 * j], yj = poly[2 * j + 1];
        const double cross = (xj - xi) * (y - yi) - (yj - yi) * (x - xi);
        if (cross == 0.0 && x >= fmin(xi, xj) && x <= fmax(xi, xj) && y >= fmin(yi, yj) && y <= fmax(yi, yj)) return true;
        if ((yi > y) != (yj > y)) {
            // x coordinate of the edge at height y, compared without a division: sign of (xj - xi)(y - yi) - (x - xi)(yj - yi) vs sign of (yj - yi)
            const bool left = (yj > yi) ? (cross > 0.0) : (cross < 0.0);
            if (left) inside = !inside;
        }
    }
    return inside;
}

// closed interval [lo, hi]; a NaN bound = the goal state does not define the attribute (always satisfied)
__device__ __forceinline__ bool in_goal_interval(double v, double lo, double hi)
{
    if (!(lo == lo) || !(hi == hi)) return true;
    return v >= lo && v <= hi;
}

// best >= 0: lattice candidate (flat FOP index); best < 0 with end_state == nullptr: no solution.  end_state (optional): explicit
// (d, v, T) of the chosen trajectory (NaN = none), takes precedence.
__device__ __forceinline__ void advance_ego_to(const KernelArgs& ka, int b, double d_end, double v_end, double T, const fp_loop_io& io);

__device__ __forceinline__ void advance_ego(const KernelArgs& ka, int b, int best, const double* end_state, const fp_loop_io& io)
{
    if (io.done[b] != FP_RUNNING) return;
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    const double nan = __builtin_nan("");
    double d_end = nan, v_end = nan, T = nan;
    if (end_state) {
        d_end = end_state[(size_t)b * 3]; v_end = end_state[(size_t)b * 3 + 1]; T = end_state[(size_t)b * 3 + 2];
    } else if (best >= 0) {
        const int iv = best % p.nv, it = (best / p.nv) % p.nt, id = best / (p.nv * p.nt);
        d_end = bt.d_samples[id]; v_end = bt.v_samples[(size_t)b * p.nv + iv]; T = bt.t_samples[it];
    }
    advance_ego_to(ka, b, d_end, v_end, T, io);
}

// The same with the chosen trajectory's end state (d, v, T) in registers (NaN = plan() returned None); the caller has checked nothing.
__device__ __forceinline__ void advance_ego_to(const KernelArgs& ka, int b, double d_end, double v_end, double T, const fp_loop_io& io)
{
    if (io.done[b] != FP_RUNNING) return;
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    if (!(T == T) || !(d_end == d_end) || !(v_end == v_end)) {  // plan() returned None (:131-133)
        io.done[b] = FP_DONE_NO_SOLUTION;
        return;
    }
    double* eg = io.ego + (size_t)b * 6;
    const Quintic lat = quintic_bvp(eg[3], eg[4], eg[5], d_end, 0.0, 0.0, T);
    const Quartic lon = quartic_bvp(eg[0], eg[1], eg[2], v_end, 0.0, T);
    const int f = bt.frame_of[b];
    const int nx = bt.nx[f];
    const double* knots = bt.knots + (size_t)f * bt.NX;
    SplineLds sp{knots, bt.coef + (size_t)f * 8 * bt.NX, nx, bt.NX};
    // state_at_time_step(1) needs x[1], y[1], yaw[1] (:135): point 1, then ONE neighbour - point 2 for the forward difference, or point 0
    // when point 2 is off the spline (the repeated last heading, :127-129).  Points are evaluated one at a time (few live registers:
    // this runs at the tail of the lattice kernel, whose register budget is the collision stages').
    const int N = arange_len(T, p.tick_t);
    struct Fr { double s, s_d, s_dd, d, d_d, d_dd; };  // (by value: an array reached through a pointer parameter ends up in scratch memory)
    auto point = [&](int i, double& x, double& y, Fr& fr) -> bool {
        const double t = (double)i * p.tick_t;
        double s3, d3;
        quartic_eval(lon, t, fr.s, fr.s_d, fr.s_dd, s3);
        quintic_eval(lat, t, fr.d, fr.d_d, fr.d_dd, d3);
        const int seg = (i < N) ? spline_segment(sp, fr.s, -1) : -1;
        if (seg < 0) return false;
        double px, py, tx, ty;
        spline_frame(sp, seg, fr.s - knots[seg], px, py, tx, ty);
        frenet_to_cartesian(px, py, tx, ty, fr.d, x, y);
        return true;
    };
    double x0, y0, x1, y1, xn, yn;
    Fr next, other;
    // (the first point off the spline truncates the series: point 1 exists only if point 0 does)
    if (!point(0, x0, y0, other) || !point(1, x1, y1, next)) {  // the reference indexes x[1] of a trajectory that left the spline at once: IndexError -> the run ends
        io.done[b] = FP_DONE_NO_SOLUTION;
        return;
    }
    const bool fwd = point(2, xn, yn, other);
    const double yaw = fwd ? atan2(yn - y1, xn - x1) : atan2(y1 - y0, x1 - x0);
    const double xs1 = x1, ys1 = y1, s_d1 = next.s_d;
    eg[0] = next.s; eg[1] = next.s_d; eg[2] = next.s_dd; eg[3] = next.d; eg[4] = next.d_d; eg[5] = next.d_dd;
    const int cycle = io.t_now[b];  // state.time_step = i (:138)
    io.t_now[b] = cycle + 1;
    io.cycles[b] += 1;
    if (io.cart_state) {
        io.cart_state[(size_t)b * 3] = xs1; io.cart_state[(size_t)b * 3 + 1] = ys1; io.cart_state[(size_t)b * 3 + 2] = yaw;
    }
    // stop rules, in the reference's order (:150-161)
    if (io.goal_poly && io.goal_nv) {  // goal_region.is_reached(state)
        const int gn = io.goal_nv[b];
        if (gn >= 3 && gn <= io.goal_max_vertices) {
            const double* gi = io.goal_intervals ? io.goal_intervals + (size_t)b * 6 : nullptr;
            const bool attrs = !gi || (in_goal_interval((double)cycle, gi[0], gi[1]) && in_goal_interval(s_d1, gi[2], gi[3]) && in_goal_interval(yaw, gi[4], gi[5]));
            if (attrs && point_in_polygon_closed(io.goal_poly + (size_t)b * io.goal_max_vertices * 2, gn, xs1, ys1)) {
                io.done[b] = FP_DONE_GOAL_REGION;
                return;
            }
        }
    }
    const double gx = io.goal_xy[(size_t)b * 2], gy = io.goal_xy[(size_t)b * 2 + 1];
    if (hypot(xs1 - gx, ys1 - gy) <= 0.5 * p.veh_l) {
        io.done[b] = FP_DONE_GOAL;
        return;
    }
    // the end of the map is the last point of np.arange(0, s_last, 0.1)
    const double s_last = knots[nx - 1];
    int n_ref = (int)ceil(s_last / 0.1);
    if (n_ref < 1) n_ref = 1;
    const double s_ref = (double)(n_ref - 1) * 0.1;
    const int seg = spline_segment(sp, s_ref, -1);
    if (seg >= 0) {
        double px, py, tx, ty;
        spline_frame(sp, seg, s_ref - knots[seg], px, py, tx, ty);
        if (hypot(xs1 - px, ys1 - py) <= 3.0) io.done[b] = FP_DONE_END_OF_LINE;
    }
}

}  // namespace fp
