// frenet_device.h - device-side math shared by the gfx950 kernels.
//
// FP64 throughout: costs reach ~100 and world coordinates ~500 m, so the 1e-6
// cost parity bar of the reference comparison rules out FP32.
//
// What each block restates (paths relative to the reference checkout):
//   quintic / quartic BVP         planners/common/geometry/polynomial.py:5-19,45-62
//   polynomial value/derivatives  planners/common/geometry/polynomial.py:21-41,64-84
//   spline segment + evaluation   planners/common/geometry/cubic_spline.py:45-116
//   OBB-vs-OBB overlap            planners/frenet_optimal_planner.py:162-195 (shapely intersects)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/frenet_gpu.h"

namespace fp {

// Points per trajectory this call is sized for: fp_params.points_max when the caller announced more than the fast paths' 128 (FP_MEM_HOST
// calls: worked out by the library), else FP_FAST_POINTS.  A trajectory that needs more is reported as NaN cost + infeasible by every
// kernel (include/frenet_gpu.h).
__host__ __device__ inline int points_cap(const fp_params& p)
{
    return p.points_max > FP_FAST_POINTS ? (p.points_max < FP_MAX_POINTS ? p.points_max : FP_MAX_POINTS) : FP_FAST_POINTS;
}


constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// Polynomials.  The reference solves a 3x3 / 2x2 linear system per trajectory
// with np.linalg.solve; the systems have closed-form solutions (end velocity
// error V, end acceleration error A, end position error D):
//   quintic: a3 = (20D - 8VT + AT^2) / (2T^3)
//            a4 = (-30D + 14VT - 2AT^2) / (2T^4)
//            a5 = (12D - 6VT + AT^2) / (2T^5)
//   quartic: a3 = (3V - AT) / (3T^2),  a4 = (AT - 2V) / (4T^3)
// Both are odd in the boundary data, so mirrored lateral candidates produce
// bit-identical costs (the exact-tie rule of FOP depends on that).
// ---------------------------------------------------------------------------
struct Quintic {
    double a0, a1, a2, a3, a4, a5;
};
struct Quartic {
    double a0, a1, a2, a3, a4;
};

__device__ __forceinline__ Quintic quintic_bvp(double xs, double vxs, double axs, double xe, double vxe, double axe, double T)
{
    Quintic q;
    q.a0 = xs;
    q.a1 = vxs;
    q.a2 = axs * 0.5;
    const double T2 = T * T;
    const double D = xe - q.a0 - q.a1 * T - q.a2 * T2;
    const double V = vxe - q.a1 - 2.0 * q.a2 * T;
    const double A = axe - 2.0 * q.a2;
    const double iT = 1.0 / T;
    const double iT3 = iT * iT * iT;
    q.a3 = (20.0 * D - 8.0 * V * T + A * T2) * (0.5 * iT3);
    q.a4 = (-30.0 * D + 14.0 * V * T - 2.0 * A * T2) * (0.5 * iT3 * iT);
    q.a5 = (12.0 * D - 6.0 * V * T + A * T2) * (0.5 * iT3 * iT * iT);
    return q;
}

__device__ __forceinline__ Quartic quartic_bvp(double xs, double vxs, double axs, double vxe, double axe, double T)
{
    Quartic q;
    q.a0 = xs;
    q.a1 = vxs;
    q.a2 = axs * 0.5;
    const double V = vxe - q.a1 - 2.0 * q.a2 * T;
    const double A = axe - 2.0 * q.a2;
    const double iT = 1.0 / T;
    q.a3 = (3.0 * V - A * T) * (iT * iT * (1.0 / 3.0));
    q.a4 = (A * T - 2.0 * V) * (0.25 * iT * iT * iT);
    return q;
}

// value and three derivatives, Horner form
__device__ __forceinline__ void quintic_eval(const Quintic& q, double t, double& p, double& v, double& a, double& j)
{
    p = fma(fma(fma(fma(fma(q.a5, t, q.a4), t, q.a3), t, q.a2), t, q.a1), t, q.a0);
    v = fma(fma(fma(fma(5.0 * q.a5, t, 4.0 * q.a4), t, 3.0 * q.a3), t, 2.0 * q.a2), t, q.a1);
    a = fma(fma(fma(20.0 * q.a5, t, 12.0 * q.a4), t, 6.0 * q.a3), t, 2.0 * q.a2);
    j = fma(fma(60.0 * q.a5, t, 24.0 * q.a4), t, 6.0 * q.a3);
}
__device__ __forceinline__ void quartic_eval(const Quartic& q, double t, double& p, double& v, double& a, double& j)
{
    p = fma(fma(fma(fma(q.a4, t, q.a3), t, q.a2), t, q.a1), t, q.a0);
    v = fma(fma(fma(4.0 * q.a4, t, 3.0 * q.a3), t, 2.0 * q.a2), t, q.a1);
    a = fma(fma(12.0 * q.a4, t, 6.0 * q.a3), t, 2.0 * q.a2);
    j = fma(24.0 * q.a4, t, 6.0 * q.a3);
}

// ---------------------------------------------------------------------------
// Cost sums in closed form.  Every summand of CostFunction.cost_total (cost_function.py:29-50) is the square of a
// polynomial in t, so  sum_i p(t_i)^2 = sum_k (p * p)_k S_k  with the power sums S_k = sum_{i<N} t_i^k, k = 0..10.
//   lon[3] = sum (s_d - v_target)^2, sum s_dd^2, sum s_ddd^2      lat[3] = sum d_dd^2, sum d_ddd^2, sum d^2
// Even in the lateral boundary data: mirrored candidates give bit-identical sums.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void lon_cost_sums(const Quartic& q, double target_speed, const double* S, double* lon)
{
    const double e0 = q.a1 - target_speed, e1 = 2.0 * q.a2, e2 = 3.0 * q.a3, e3 = 4.0 * q.a4;
    const double g0 = 2.0 * q.a2, g1 = 6.0 * q.a3, g2 = 12.0 * q.a4;
    const double h0 = 6.0 * q.a3, h1 = 24.0 * q.a4;
    double sv = e0 * e0 * S[0];
    sv = fma(2.0 * e0 * e1, S[1], sv);
    sv = fma(fma(2.0 * e0, e2, e1 * e1), S[2], sv);
    sv = fma(2.0 * fma(e0, e3, e1 * e2), S[3], sv);
    sv = fma(fma(2.0 * e1, e3, e2 * e2), S[4], sv);
    sv = fma(2.0 * e2 * e3, S[5], sv);
    sv = fma(e3 * e3, S[6], sv);
    double sa = g0 * g0 * S[0];
    sa = fma(2.0 * g0 * g1, S[1], sa);
    sa = fma(fma(2.0 * g0, g2, g1 * g1), S[2], sa);
    sa = fma(2.0 * g1 * g2, S[3], sa);
    sa = fma(g2 * g2, S[4], sa);
    double sj = h0 * h0 * S[0];
    sj = fma(2.0 * h0 * h1, S[1], sj);
    sj = fma(h1 * h1, S[2], sj);
    lon[0] = sv; lon[1] = sa; lon[2] = sj;
}

__device__ __forceinline__ void lat_cost_sums(const Quintic& q, const double* S, double* lat)
{
    const double c[6] = {q.a0, q.a1, q.a2, q.a3, q.a4, q.a5};
    double sd = 0.0;
#pragma unroll
    for (int kk = 0; kk <= 10; ++kk) {
        double ck = 0.0;
#pragma unroll
        for (int a2 = 0; a2 <= 5; ++a2) {
            const int b2 = kk - a2;
            if (b2 >= 0 && b2 <= 5) ck = fma(c[a2], c[b2], ck);
        }
        sd = fma(ck, S[kk], sd);
    }
    const double g[4] = {2.0 * q.a2, 6.0 * q.a3, 12.0 * q.a4, 20.0 * q.a5};  // d_dd
    double sa = 0.0;
#pragma unroll
    for (int kk = 0; kk <= 6; ++kk) {
        double ck = 0.0;
#pragma unroll
        for (int a2 = 0; a2 <= 3; ++a2) {
            const int b2 = kk - a2;
            if (b2 >= 0 && b2 <= 3) ck = fma(g[a2], g[b2], ck);
        }
        sa = fma(ck, S[kk], sa);
    }
    const double h[3] = {6.0 * q.a3, 24.0 * q.a4, 60.0 * q.a5};  // d_ddd
    double sj = 0.0;
#pragma unroll
    for (int kk = 0; kk <= 4; ++kk) {
        double ck = 0.0;
#pragma unroll
        for (int a2 = 0; a2 <= 2; ++a2) {
            const int b2 = kk - a2;
            if (b2 >= 0 && b2 <= 2) ck = fma(h[a2], h[b2], ck);
        }
        sj = fma(ck, S[kk], sj);
    }
    lat[0] = sa; lat[1] = sj; lat[2] = sd;
}

// cost_total with the reference's grouping (cost_function.py:41-50)
__device__ __forceinline__ double combine_cost(const fp_params& p, int N, const double* lon, const double* lat)
{
    const double cost_time = p.cost_horizon - (double)(N - 1) * p.tick_t;
    const double cost_speed = p.w_speed * lon[0];
    const double cost_accel = p.w_accel * lon[1] + p.w_accel * lat[0];
    const double cost_jerk = p.w_jerk * lon[2] + p.w_jerk * lat[1];
    const double cost_offset = p.w_offset * lat[2];
    return (cost_time + 0.0 + cost_speed + cost_accel + cost_jerk + cost_offset) / (double)N;
}

// len(np.arange(0, T, tick)) = ceil(T / tick) evaluated in double
__device__ __forceinline__ int arange_len(double T, double tick)
{
    const double n = ceil(T / tick);
    return n > 0.0 ? (int)n : 0;
}

// S_k = sum_{i<N} (i*tick)^k = tick^k P_k(N), k = 0..10, with Faulhaber's polynomials P_k(N) = sum_{i<N} i^k
// = 1/(k+1) sum_j C(k+1, j) B_j N^(k+1-j) (Bernoulli numbers, B_1 = -1/2; coefficients generated with exact rationals and
// checked against the sums).  One lane per slice, ~90 instructions, instead of a wavefront per slice reducing 11 sums by DPP
// trees; N <= 128 keeps the Horner evaluation at full double accuracy (leading term dominates: N^(k+1)/(k+1) vs N^k/2).
__device__ __forceinline__ void power_sums_closed(int N, double tick, double* out)
{
    const double x = (double)N;
    double tk = 1.0;
    out[0] = (1) * x * tk; tk *= tick;
    out[1] = (fma(0.5, x, -0.5)) * x * tk; tk *= tick;
    out[2] = (fma(fma(0.33333333333333331, x, -0.5), x, 0.16666666666666666)) * x * tk; tk *= tick;
    out[3] = ((fma(fma(0.25, x, -0.5), x, 0.25)) * x) * x * tk; tk *= tick;
    out[4] = (fma((fma(fma(0.20000000000000001, x, -0.5), x, 0.33333333333333331)) * x, x, -0.033333333333333333)) * x * tk; tk *= tick;
    out[5] = ((fma((fma(fma(0.16666666666666666, x, -0.5), x, 0.41666666666666669)) * x, x, -0.083333333333333329)) * x) * x * tk; tk *= tick;
    out[6] = (fma((fma((fma(fma(0.14285714285714285, x, -0.5), x, 0.5)) * x, x, -0.16666666666666666)) * x, x, 0.023809523809523808)) * x * tk; tk *= tick;
    out[7] = ((fma((fma((fma(fma(0.125, x, -0.5), x, 0.58333333333333337)) * x, x, -0.29166666666666669)) * x, x, 0.083333333333333329)) * x) * x * tk; tk *= tick;
    out[8] = (fma((fma((fma((fma(fma(0.1111111111111111, x, -0.5), x, 0.66666666666666663)) * x, x, -0.46666666666666667)) * x, x, 0.22222222222222221)) * x, x, -0.033333333333333333)) * x * tk; tk *= tick;
    out[9] = ((fma((fma((fma((fma(fma(0.10000000000000001, x, -0.5), x, 0.75)) * x, x, -0.69999999999999996)) * x, x, 0.5)) * x, x, -0.14999999999999999)) * x) * x * tk; tk *= tick;
    out[10] = (fma((fma((fma((fma((fma(fma(0.090909090909090912, x, -0.5), x, 0.83333333333333337)) * x, x, -1)) * x, x, 1)) * x, x, -0.5)) * x, x, 0.07575757575757576)) * x * tk; tk *= tick;
}


// 1/sqrt(x) to ~1 ulp: hardware seed (v_rsq_f64, ~2^-23 relative) + two Newton steps.
// Used for the unit tangent / heading vectors; replaces a full sqrt + divide.
__device__ __forceinline__ double rsqrt_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    y = y * fma(-hx * y, y, 1.5);
    return y;
}

// ---------------------------------------------------------------------------
// Reference-line spline, staged in LDS as knots[NX] + coef[8][NX].
// Segment rule = bisect.bisect(knots, s) - 1 (cubic_spline.py:112-116) with the
// reference's range test (:56-59); s == last knot indexes past the b/d lists in
// the reference (IndexError) and is treated as out of range.
// ---------------------------------------------------------------------------
struct SplineLds {
    const double* knots;  // [nx]
    const double* coef;   // [8][ld]
    int nx;
    int ld;
};

// returns segment index, or -1 when s is outside [knots[0], knots[nx-1])
// guess_scale (optional, with hint < 0): (nx - 1) / (last knot - first knot), computed once by the caller - the guess then costs a
// multiplication instead of an fp64 division per point (it is only a guess: the knot comparisons below decide).
__device__ __forceinline__ int spline_segment(const SplineLds& sp, double s, int hint, double guess_scale = 0.0)
{
    const int last = sp.nx - 1;
    const double k0 = sp.knots[0], kl = sp.knots[last];
    if (!(s >= k0) || !(s < kl)) return -1;
    // s is nearly monotone along a trajectory: try the previous segment and its successor first.  Without a hint: the segment a
    // uniform knot spacing would give (centerlines are close to it) and its two neighbours - two or three dependent reads instead of
    // the bisection's log2(nx), which matters when the table sits in global memory (winner_traj_kernel: one trajectory per wavefront)
    int i = hint;
    if (i < 0) {
        i = guess_scale > 0.0 ? (int)((s - k0) * guess_scale) : (int)((s - k0) / (kl - k0) * (double)last);
        i = i < 0 ? 0 : (i > last - 1 ? last - 1 : i);
        if (i > 0 && s < sp.knots[i]) --i;
    }
    if (i >= 0 && i < last && sp.knots[i] <= s) {
        if (s < sp.knots[i + 1]) return i;
        if (i + 1 < last && s < sp.knots[i + 2]) return i + 1;
    }
    int lo = 0, hi = sp.nx;  // bisect_right
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s < sp.knots[mid]) hi = mid; else lo = mid + 1;
    }
    return lo - 1;
}

// position (px,py) and UNIT tangent (tx,ty) of the reference line at (segment i, offset dx)
__device__ __forceinline__ void spline_frame(const SplineLds& sp, int i, double dx, double& px, double& py, double& tx, double& ty)
{
    const double* c = sp.coef + i;
    const int ld = sp.ld;
    const double ax = c[0], bx = c[ld], cx = c[2 * ld], dx3 = c[3 * ld];
    const double ay = c[4 * ld], by = c[5 * ld], cy = c[6 * ld], dy3 = c[7 * ld];
    px = fma(fma(fma(dx3, dx, cx), dx, bx), dx, ax);
    py = fma(fma(fma(dy3, dx, cy), dx, by), dx, ay);
    const double gx = fma(fma(3.0 * dx3, dx, 2.0 * cx), dx, bx);
    const double gy = fma(fma(3.0 * dy3, dx, 2.0 * cy), dx, by);
    // cos/sin(atan2(gy, gx)): normalise the tangent instead of atan2 + sincos
    const double inv = rsqrt_nr(fma(gx, gx, gy * gy));
    tx = gx * inv;
    ty = gy * inv;
}

// Frenet (s via segment/dx, d) -> Cartesian: x = px + d cos(yaw + pi/2), y = py + d sin(yaw + pi/2)
// (frenet_optimal_planner.py:116-119) with cos(yaw+pi/2) = -ty, sin(yaw+pi/2) = tx.
__device__ __forceinline__ void frenet_to_cartesian(double px, double py, double tx, double ty, double d, double& x, double& y)
{
    x = fma(-d, ty, px);
    y = fma(d, tx, py);
}

// ---------------------------------------------------------------------------
// Oriented boxes.  shapely's Polygon.intersects on two rectangles is a closed-set
// overlap test (touching counts); for boxes it is the 4-axis separating-axis test.
// Box = centre (x,y), unit heading (c,s), half extents (hl, hw).
// ---------------------------------------------------------------------------
struct Obb {
    double x, y, c, s, hl, hw;
};

__device__ __forceinline__ bool obb_overlap(const Obb& a, const Obb& b)
{
    const double dx = b.x - a.x, dy = b.y - a.y;
    const double C = fabs(fma(a.c, b.c, a.s * b.s));  // |cos(delta)|
    const double S = fabs(fma(a.s, b.c, -a.c * b.s)); // |sin(delta)|
    // axes of a
    if (fabs(fma(dx, a.c, dy * a.s)) > a.hl + fma(b.hl, C, b.hw * S)) return false;
    if (fabs(fma(dy, a.c, -dx * a.s)) > a.hw + fma(b.hl, S, b.hw * C)) return false;
    // axes of b
    if (fabs(fma(dx, b.c, dy * b.s)) > b.hl + fma(a.hl, C, a.hw * S)) return false;
    if (fabs(fma(dy, b.c, -dx * b.s)) > b.hw + fma(a.hl, S, a.hw * C)) return false;
    return true;
}

// The same four axes as a signed distance: max over the axes of (centre distance along the axis - the two boxes' reach along it).
// <= 0: the boxes overlap (obb_overlap is `obb_gap <= 0` with early exits); > 0: separated by at least that much along the best
// axis (a lower bound of the true distance).  The audit pass (audit_kernel) calls a decision "thin" when |gap| is below its tolerance.
__device__ __forceinline__ double obb_gap(const Obb& a, const Obb& b)
{
    const double dx = b.x - a.x, dy = b.y - a.y;
    const double C = fabs(fma(a.c, b.c, a.s * b.s));
    const double S = fabs(fma(a.s, b.c, -a.c * b.s));
    const double g0 = fabs(fma(dx, a.c, dy * a.s)) - (a.hl + fma(b.hl, C, b.hw * S));
    const double g1 = fabs(fma(dy, a.c, -dx * a.s)) - (a.hw + fma(b.hl, S, b.hw * C));
    const double g2 = fabs(fma(dx, b.c, dy * b.s)) - (b.hl + fma(a.hl, C, a.hw * S));
    const double g3 = fabs(fma(dy, b.c, -dx * b.s)) - (b.hw + fma(a.hl, S, a.hw * C));
    return fmax(fmax(g0, g1), fmax(g2, g3));
}

// ---------------------------------------------------------------------------
// Convex-polygon obstacles (fp_batch.obs_poly / obs_nvert): `obstacle_shape.shapely_object` is any polygon in the reference
// (frenet_optimal_planner.py:189-191); a commonroad Circle is one too (a 64-gon from shapely's buffer()).  An obstacle column carries
// n vertices u_i relative to its rotation centre, counter-clockwise; at a pose (x, y, yaw) vertex i sits at (x, y) + R(yaw) u_i - what
// affinity.translate + rotate(origin='center') give a polygon whose bounding box is centred on the origin.  Polygon.intersects of two
// convex polygons is the closed separating-axis test over the edge normals of both: the ego box's two axes + the n edge normals.
// The test runs in the EGO's frame (ego = [-hl, hl] x [-hw, hw] at the origin): few live values, one pass over the vertices.
// Signed gap like obb_gap: <= 0 overlap (touching counts), > 0 separated by at least that much along the best axis.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double poly_gap(const Obb& ego, double ox, double oy, double oc, double os, const double* __restrict__ v, int n)
{
    const double dx = ox - ego.x, dy = oy - ego.y;
    const double px = fma(dx, ego.c, dy * ego.s), py = fma(dy, ego.c, -dx * ego.s);   // obstacle centre in the ego frame
    const double C = fma(oc, ego.c, os * ego.s), S = fma(os, ego.c, -oc * ego.s);     // cos / sin of (obstacle yaw - ego yaw)
    double ux = v[2 * (n - 1)], uy = v[2 * (n - 1) + 1];
    double qpx = px + fma(C, ux, -S * uy), qpy = py + fma(S, ux, C * uy);             // previous vertex
    double minx = qpx, maxx = qpx, miny = qpy, maxy = qpy;
    double g = -__builtin_inf();
    for (int i = 0; i < n; ++i) {
        ux = v[2 * i]; uy = v[2 * i + 1];
        const double qx = px + fma(C, ux, -S * uy), qy = py + fma(S, ux, C * uy);
        minx = fmin(minx, qx); maxx = fmax(maxx, qx); miny = fmin(miny, qy); maxy = fmax(maxy, qy);
        // edge q_prev -> q, outward normal (counter-clockwise ring) n = (ey, -ex): the ego's lowest point along n against the edge
        const double ex = qx - qpx, ey = qy - qpy;
        const double len2 = fma(ex, ex, ey * ey);
        if (len2 > 0.0) {  // (a repeated vertex spans no half plane)
            const double sep = -fma(ey, qpx, -ex * qpy) - fma(ego.hl, fabs(ey), ego.hw * fabs(ex));
            g = fmax(g, sep * rsqrt_nr(len2));
        }
        qpx = qx; qpy = qy;
    }
    g = fmax(g, fmax(minx, -maxx) - ego.hl);
    g = fmax(g, fmax(miny, -maxy) - ego.hw);
    return g;
}

// the same test as a verdict: separated only when some axis separates STRICTLY (sign of the un-normalised value: no division).
// The ring lives in global memory (L2): its vertices are fetched four at a time (independent 16-byte loads, clamped indices) - one
// dependent load per vertex made a 12-gon cost twelve L2 round trips per lane.
__device__ __forceinline__ bool poly_overlap(const Obb& ego, double ox, double oy, double oc, double os, const double* __restrict__ v, int n)
{
    const double dx = ox - ego.x, dy = oy - ego.y;
    const double px = fma(dx, ego.c, dy * ego.s), py = fma(dy, ego.c, -dx * ego.s);
    const double C = fma(oc, ego.c, os * ego.s), S = fma(os, ego.c, -oc * ego.s);
    const double2* __restrict__ v2 = (const double2*)v;  // (rings start 16-byte aligned: obs_poly is an array of vertex pairs)
    const double2 last = v2[n - 1];
    double qpx = px + fma(C, last.x, -S * last.y), qpy = py + fma(S, last.x, C * last.y);
    double minx = qpx, maxx = qpx, miny = qpy, maxy = qpy;
    bool separated = false;
    for (int i0 = 0; i0 < n; i0 += 4) {
        double2 u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = v2[i0 + k < n ? i0 + k : n - 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < n) {
                const double qx = px + fma(C, u[k].x, -S * u[k].y), qy = py + fma(S, u[k].x, C * u[k].y);
                minx = fmin(minx, qx); maxx = fmax(maxx, qx); miny = fmin(miny, qy); maxy = fmax(maxy, qy);
                const double ex = qx - qpx, ey = qy - qpy;
                separated = separated || -fma(ey, qpx, -ex * qpy) - fma(ego.hl, fabs(ey), ego.hw * fabs(ex)) > 0.0;
                qpx = qx; qpy = qy;
            }
        }
    }
    return !(separated || minx > ego.hl || maxx < -ego.hl || miny > ego.hw || maxy < -ego.hw);
}

// Radius of the largest disk about the rotation centre that lies inside a convex counter-clockwise ring (0 when the centre is not
// strictly inside): min over the edges of the centre's distance to the edge line.  An ego box that reaches into that disk overlaps
// the polygon whatever its edges look like - for a circle (shapely's 64-gon: the disk is 99.9 % of it) that settles nearly every
// pair without walking the 64 edges.  Shrunk by 1e-9 so that only pairs far from any rounding doubt take the shortcut.
__device__ __forceinline__ double poly_inner_radius(const double* __restrict__ v, int n)
{
    double r = __builtin_inf();
    double px = v[2 * (n - 1)], py = v[2 * (n - 1) + 1];
    for (int i = 0; i < n; ++i) {
        const double qx = v[2 * i], qy = v[2 * i + 1];
        const double ex = qx - px, ey = qy - py;
        const double len2 = fma(ex, ex, ey * ey);
        if (len2 > 0.0) r = fmin(r, fma(ey, px, -ex * py) * rsqrt_nr(len2));  // outward normal (ey, -ex) . vertex = distance of the origin to the edge line
        px = qx; py = qy;
    }
    return r > 0.0 && r < __builtin_inf() ? r * (1.0 - 1e-9) : 0.0;
}

// The narrow phase of one (ego pose, obstacle) pair whatever the obstacle's shape: the box test on the obstacle's dims (for a polygon:
// the centred box that contains it, a necessary condition), then the polygon itself.  nvert == nullptr: every obstacle is a rectangle.
// r_in (optional, poly_inner_radius of the column): an ego box within r_in of the rotation centre overlaps without looking at the ring.
__device__ __forceinline__ bool ring_overlap(const Obb& ego, const Obb& ob, const double* ring, int n, double r_in)
{
    if (r_in > 0.0) {  // distance of the ego box to the obstacle's centre, in the ego's frame
        const double dx = ob.x - ego.x, dy = ob.y - ego.y;
        const double ax = fmax(fabs(fma(dx, ego.c, dy * ego.s)) - ego.hl, 0.0), ay = fmax(fabs(fma(dy, ego.c, -dx * ego.s)) - ego.hw, 0.0);
        if (fma(ax, ax, ay * ay) <= r_in * r_in) return true;
    }
    return poly_overlap(ego, ob.x, ob.y, ob.c, ob.s, ring, n);
}
__device__ __forceinline__ bool shape_overlap(const Obb& ego, const Obb& ob, const int32_t* nvert, const double* poly, int poly_stride, size_t col,
                                              double r_in = 0.0)
{
    if (!obb_overlap(ego, ob)) return false;
    if (nvert) {
        const int n = nvert[col];
        if (n > 0) return ring_overlap(ego, ob, poly + col * 2 * (size_t)poly_stride, n, r_in);
    }
    return true;
}

// heading unit vector of the step (dx,dy): cos/sin(atan2(dy,dx)); atan2(0,0) = 0 -> (1,0).
// An axis-parallel step gives exactly (+-1, 0) / (0, +-1), as cos / sin of atan2's exact 0, pi, +-pi/2 do after shapely's snap
// (see sincos_snapped): boxes that touch exactly are then decided by exact arithmetic, like in the reference.
__device__ __forceinline__ void step_heading(double dx, double dy, double& c, double& s)
{
    const double h2 = fma(dx, dx, dy * dy);
    if (h2 > 0.0) {
        const double inv = rsqrt_nr(h2);
        c = dy == 0.0 ? (dx > 0.0 ? 1.0 : -1.0) : dx * inv;
        s = dx == 0.0 ? (dy > 0.0 ? 1.0 : -1.0) : dy * inv;
    } else {
        c = 1.0;
        s = 0.0;
    }
}

// cos / sin of an obstacle's orientation as shapely.affinity.rotate computes them (construct_polygon,
// frenet_optimal_planner.py:162-166): |cos| or |sin| below 2.5e-16 is snapped to 0, so that yaw = k pi/2 rotates exactly.
__device__ __forceinline__ void sincos_snapped(double yaw, double& s, double& c)
{
    sincos(yaw, &s, &c);
    if (fabs(c) < 2.5e-16) c = 0.0;
    if (fabs(s) < 2.5e-16) s = 0.0;
}

// ---------------------------------------------------------------------------
// Optional curvature checks (frenet_optimal_planner.py:145-150, commented out in the reference; fp_params.curvature_mask).
// The checked series are finite-difference chains over the Cartesian points (:121-134):
//     yaw_k = atan2(y_{k+1} - y_k, x_{k+1} - x_k), k < M-1;  yaw_{M-1} = yaw_{M-2};   ds_k = hypot(...)
//     c_k = (yaw_{k+1} - yaw_k) / ds_k (M-1 values),  c_d = diff(c) / dt (M-2),  c_dd = diff(c_d) / dt (M-3)
// CurvTrack consumes the segments (yaw_k, ds_k) in order and keeps only the previous element of each chain.
// `abs(nan) > limit` is False in the reference (a 0/0 of a standing segment never violates); inf does.
// ---------------------------------------------------------------------------
struct CurvTrack {
    double yaw_prev, ds_prev, c_prev, cd_prev, inv_dt;
    double max_c, max_cd, max_cdd;
    uint32_t flags;
    int have;  // bit 0: a segment, bit 1: a c value, bit 2: a c_d value

    __device__ __forceinline__ void init(const fp_params& p)
    {
        max_c = p.max_curvature; max_cd = p.max_kappa_d; max_cdd = p.max_kappa_dd;
        yaw_prev = ds_prev = c_prev = cd_prev = 0.0;
        flags = 0; have = 0;
    }
    __device__ __forceinline__ void push_cd(double cd, double dt)
    {
        if (fabs(cd) > max_cd) flags |= FP_FLAG_KAPPA_D;
        if ((have & 4) && fabs((cd - cd_prev) / dt) > max_cdd) flags |= FP_FLAG_KAPPA_DD;
        cd_prev = cd; have |= 4;
    }
    __device__ __forceinline__ void push_c(double c, double dt)
    {
        if (fabs(c) > max_c) flags |= FP_FLAG_CURVATURE;
        if (have & 2) push_cd((c - c_prev) / dt, dt);
        c_prev = c; have |= 2;
    }
    __device__ __forceinline__ void push_segment(double yaw, double ds, double dt)
    {
        if (have & 1) push_c((yaw - yaw_prev) / ds_prev, dt);
        yaw_prev = yaw; ds_prev = ds; have |= 1;
    }
    // yaw[M-1] repeats yaw[M-2] (:129): the last curvature sample is 0 / ds_{M-2}
    __device__ __forceinline__ void finish(double dt)
    {
        if (have & 1) push_c((yaw_prev - yaw_prev) / ds_prev, dt);
    }
};

// FP_FLAG_CURVATURE / KAPPA_D / KAPPA_DD of ONE trajectory, sequentially over its points (one lane per trajectory).  Positions come
// from the same spline_frame / frenet_to_cartesian arithmetic as every series dump of this library.
__device__ __forceinline__ uint32_t curvature_flags(const fp_params& p, const SplineLds& sp, const Quartic& lon, const Quintic& lat, int N)
{
    CurvTrack ct;
    ct.init(p);
    int seg = -1;
    double xp = 0.0, yp = 0.0;
    for (int i = 0; i < N; ++i) {
        const double t = (double)i * p.tick_t;
        const double s = fma(fma(fma(fma(lon.a4, t, lon.a3), t, lon.a2), t, lon.a1), t, lon.a0);
        seg = spline_segment(sp, s, seg);
        if (seg < 0) break;  // truncation (:112-113)
        const double d = fma(fma(fma(fma(fma(lat.a5, t, lat.a4), t, lat.a3), t, lat.a2), t, lat.a1), t, lat.a0);
        double px, py, tx, ty, x, y;
        spline_frame(sp, seg, s - sp.knots[seg], px, py, tx, ty);
        frenet_to_cartesian(px, py, tx, ty, d, x, y);
        if (i >= 1) ct.push_segment(atan2(y - yp, x - xp), hypot(x - xp, y - yp), p.tick_t);
        xp = x; yp = y;
    }
    ct.finish(p.tick_t);
    return ct.flags;
}

// ---------------------------------------------------------------------------
// wave / block reductions (64-lane wavefronts)
// ---------------------------------------------------------------------------
// DPP lane exchange inside a row of 16 lanes (no LDS round trip, unlike __shfl_xor -> ds_bpermute):
//   0xB1 quad_perm[1,0,3,2] (xor 1), 0x4E quad_perm[2,3,0,1] (xor 2), 0x141 row_half_mirror, 0x140 row_mirror.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// v_min / v_max without the canonicalising v_max x, x, x the compiler puts in front of every fmin / fmax operand (IEEE mode);
// a quiet NaN operand yields the other operand, like fmin / fmax.  Halves the length of the DPP reduction chains.
__device__ __forceinline__ double vmin_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax_f32(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// after these four steps every lane holds the reduction of its row of 16 lanes
__device__ __forceinline__ double row16_sum(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return v;
}
__device__ __forceinline__ double row16_min(double v)
{
    v = vmin_f64(v, dpp_f64<0xB1>(v));
    v = vmin_f64(v, dpp_f64<0x4E>(v));
    v = vmin_f64(v, dpp_f64<0x141>(v));
    v = vmin_f64(v, dpp_f64<0x140>(v));
    return v;
}
__device__ __forceinline__ double lane_value(double v, int src_lane)  // src_lane must be wave-uniform
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane), __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}
// sum over the 64 lanes, same value (and same summation tree) in every lane
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v = row16_sum(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ double wave_min_f64(double v)
{
    v = row16_min(v);
    return vmin_f64(vmin_f64(lane_value(v, 0), lane_value(v, 16)), vmin_f64(lane_value(v, 32), lane_value(v, 48)));
}
struct Best {
    double cost;
    int idx;
};
// FOP keeps the LAST minimal candidate (`min_cost >= fp.cost_final`, frenet_optimal_planner.py:266)
__device__ __forceinline__ Best best_merge(Best a, Best b)
{
    const bool take_b = (b.idx >= 0) && (a.idx < 0 || b.cost < a.cost || (b.cost == a.cost && b.idx > a.idx));
    return take_b ? b : a;
}
__device__ __forceinline__ Best wave_best(Best v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        Best o;
        o.cost = __shfl_xor(v.cost, off, kWave);
        o.idx = __shfl_xor(v.idx, off, kWave);
        v = best_merge(v, o);
    }
    return v;
}

}  // namespace fp
