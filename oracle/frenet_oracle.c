/*
 * frenet_oracle.c - CPU ORACLE (test infrastructure, NOT product code).
 * See frenet_oracle.h for scope, citations and the parity pin statement.
 *
 * Style: deliberately literal.  Every trajectory is generated point by point
 * into full arrays, exactly like the Python reference does, with the same
 * expression grouping (pow() for `t ** k`, left-to-right sums, LU solves with
 * partial pivoting in place of np.linalg.solve).  No sharing of work between
 * candidates, no closed forms: that is the product's job, and the point of the
 * oracle is to be an independent statement of what the product must equal.
 */
#include "frenet_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>

/* -DORC_FAST (libfrenet_oracle_fast.so, `make -C oracle fast`): the same restatement with `t ** k` as multiplications instead of
   pow() calls - a LABELLED second CPU baseline for bench.py (not bit-identical to the reference any more: ~1e-13 on the costs; the
   golden-vector suites run on the literal build only). */
#ifdef ORC_FAST
static inline double orc_ipow(double x, double k)
{
    if (k == 2.0) return x * x;
    if (k == 3.0) return x * x * x;
    if (k == 4.0) { const double y = x * x; return y * y; }
    if (k == 5.0) { const double y = x * x; return y * y * x; }
    return pow(x, k);
}
#define pow orc_ipow
#endif

#endif

#define ORC_PI 3.141592653589793

int orc_version(void) { return 100; }

/* ------------------------------------------------------------------------ */
/* np.linalg.solve stand-in: LU with partial pivoting (LAPACK dgesv scheme).  */
/* A is n x n row-major and is destroyed; b is overwritten by the solution.   */
/* ------------------------------------------------------------------------ */
static int solve_dense(int n, double* A, double* b)
{
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(A[k * n + k]);
        for (int r = k + 1; r < n; ++r) {
            double v = fabs(A[r * n + k]);
            if (v > best) { best = v; piv = r; }
        }
        if (best == 0.0) return -1; /* numpy.linalg.LinAlgError("Singular matrix") */
        if (piv != k) {
            for (int c = 0; c < n; ++c) { double t = A[k * n + c]; A[k * n + c] = A[piv * n + c]; A[piv * n + c] = t; }
            double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        for (int r = k + 1; r < n; ++r) {
            double m = A[r * n + k] / A[k * n + k];
            if (m == 0.0) continue;
            A[r * n + k] = m;
            for (int c = k + 1; c < n; ++c) A[r * n + c] -= m * A[k * n + c];
            b[r] -= m * b[k];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        double s = b[k];
        for (int c = k + 1; c < n; ++c) s -= A[k * n + c] * b[c];
        b[k] = s / A[k * n + k];
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* polynomial.py                                                             */
/* ------------------------------------------------------------------------ */
int orc_quintic_coefs(double xs, double vxs, double axs, double xe, double vxe, double axe, double T, double* a)
{
    /* polynomial.py:48-62 */
    a[0] = xs;
    a[1] = vxs;
    a[2] = axs / 2.0;
    double A[9] = {pow(T, 3), pow(T, 4), pow(T, 5),
                   3 * pow(T, 2), 4 * pow(T, 3), 5 * pow(T, 4),
                   6 * T, 12 * pow(T, 2), 20 * pow(T, 3)};
    double b[3] = {xe - a[0] - a[1] * T - a[2] * pow(T, 2), vxe - a[1] - 2 * a[2] * T, axe - 2 * a[2]};
    if (solve_dense(3, A, b)) return -1;
    a[3] = b[0];
    a[4] = b[1];
    a[5] = b[2];
    return 0;
}

int orc_quartic_coefs(double xs, double vxs, double axs, double vxe, double axe, double T, double* a)
{
    /* polynomial.py:8-19 */
    a[0] = xs;
    a[1] = vxs;
    a[2] = axs / 2.0;
    double A[4] = {3 * pow(T, 2), 4 * pow(T, 3), 6 * T, 12 * pow(T, 2)};
    double b[2] = {vxe - a[1] - 2 * a[2] * T, axe - 2 * a[2]};
    if (solve_dense(2, A, b)) return -1;
    a[3] = b[0];
    a[4] = b[1];
    return 0;
}

void orc_poly_eval(const double* a, int order, double t, double* out)
{
    if (order == 5) {
        /* polynomial.py:64-84 */
        out[0] = a[0] + a[1] * t + a[2] * pow(t, 2) + a[3] * pow(t, 3) + a[4] * pow(t, 4) + a[5] * pow(t, 5);
        out[1] = a[1] + 2 * a[2] * t + 3 * a[3] * pow(t, 2) + 4 * a[4] * pow(t, 3) + 5 * a[5] * pow(t, 4);
        out[2] = 2 * a[2] + 6 * a[3] * t + 12 * a[4] * pow(t, 2) + 20 * a[5] * pow(t, 3);
        out[3] = 6 * a[3] + 24 * a[4] * t + 60 * a[5] * pow(t, 2);
    } else {
        /* polynomial.py:21-41 */
        out[0] = a[0] + a[1] * t + a[2] * pow(t, 2) + a[3] * pow(t, 3) + a[4] * pow(t, 4);
        out[1] = a[1] + 2 * a[2] * t + 3 * a[3] * pow(t, 2) + 4 * a[4] * pow(t, 3);
        out[2] = 2 * a[2] + 6 * a[3] * t + 12 * a[4] * pow(t, 2);
        out[3] = 6 * a[3] + 24 * a[4] * t;
    }
}

/* ------------------------------------------------------------------------ */
/* cubic_spline.py                                                           */
/* ------------------------------------------------------------------------ */
int orc_spline1d_build(int32_t nx, const double* x, const double* y, double* coef)
{
    /* cubic_spline.py:19-43 with __calc_A (:118-133) and __calc_B (:135-142) */
    if (nx < 2) return -2;
    double* a = coef;
    double* b = coef + nx;
    double* c = coef + 2 * nx;
    double* d = coef + 3 * nx;
    double* h = (double*)malloc(sizeof(double) * (size_t)nx);
    double* A = (double*)calloc((size_t)nx * (size_t)nx, sizeof(double));
    if (!h || !A) { free(h); free(A); return -3; }
    for (int i = 0; i < nx - 1; ++i) {
        h[i] = x[i + 1] - x[i];
        if (h[i] < 0) { free(h); free(A); return -4; } /* ValueError: x must be sorted (:22-23) */
    }
    for (int i = 0; i < nx; ++i) a[i] = y[i];
    A[0] = 1.0;
    for (int i = 0; i < nx - 1; ++i) {
        if (i != nx - 2) A[(i + 1) * nx + (i + 1)] = 2.0 * (h[i] + h[i + 1]);
        A[(i + 1) * nx + i] = h[i];
        A[i * nx + (i + 1)] = h[i];
    }
    A[0 * nx + 1] = 0.0;
    A[(nx - 1) * nx + (nx - 2)] = 0.0;
    A[(nx - 1) * nx + (nx - 1)] = 1.0;
    for (int i = 0; i < nx; ++i) c[i] = 0.0;
    for (int i = 0; i < nx - 2; ++i) c[i + 1] = 3.0 * (a[i + 2] - a[i + 1]) / h[i + 1] - 3.0 * (a[i + 1] - a[i]) / h[i];
    int rc = solve_dense(nx, A, c);
    if (rc == 0) {
        for (int i = 0; i < nx - 1; ++i) {
            d[i] = (c[i + 1] - c[i]) / (3.0 * h[i]);
            b[i] = 1.0 / h[i] * (a[i + 1] - a[i]) - h[i] / 3.0 * (2.0 * c[i] + c[i + 1]);
        }
        b[nx - 1] = 0.0; /* the reference lists b, d have nx-1 entries; pad */
        d[nx - 1] = 0.0;
    }
    free(h);
    free(A);
    return rc;
}

int orc_spline2d_build(int32_t n, const double* px, const double* py, double* knots, double* coef_x, double* coef_y)
{
    /* cubic_spline.py:157-168: s = [0] + cumsum(hypot(diff x, diff y)) */
    knots[0] = 0.0;
    double acc = 0.0;
    for (int i = 0; i < n - 1; ++i) {
        acc += hypot(px[i + 1] - px[i], py[i + 1] - py[i]);
        knots[i + 1] = acc;
    }
    int rc = orc_spline1d_build(n, knots, px, coef_x);
    if (rc) return rc;
    return orc_spline1d_build(n, knots, py, coef_y);
}

/* bisect.bisect(self.x, x) - 1   (cubic_spline.py:112-116) */
static int search_index(int nx, const double* knots, double s)
{
    int lo = 0, hi = nx;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (s < knots[mid]) hi = mid; else lo = mid + 1;
    }
    return lo - 1;
}

/* returns 1 for "None" (:56-59); s == knots[nx-1] raises IndexError in the
   reference (b[nx-1] does not exist) - treated as out of range too. */
static int spline_seg(int nx, const double* knots, double s, int* seg, double* dx)
{
    if (!(s >= knots[0]) || !(s <= knots[nx - 1])) return 1;
    int i = search_index(nx, knots, s);
    if (i >= nx - 1) return 1;
    *seg = i;
    *dx = s - knots[i];
    return 0;
}

static double sp_pos(const double* coef, int nx, int i, double dx)
{
    /* cubic_spline.py:63-64 */
    return coef[i] + coef[nx + i] * dx + coef[2 * nx + i] * pow(dx, 2.0) + coef[3 * nx + i] * pow(dx, 3.0);
}
static double sp_d1(const double* coef, int nx, int i, double dx)
{
    /* cubic_spline.py:86-87 */
    return coef[nx + i] + 2.0 * coef[2 * nx + i] * dx + 3.0 * coef[3 * nx + i] * pow(dx, 2.0);
}
static double sp_d2(const double* coef, int nx, int i, double dx)
{
    /* cubic_spline.py:108-109 */
    return 2.0 * coef[2 * nx + i] + 6.0 * coef[3 * nx + i] * dx;
}

int orc_spline2d_eval(int32_t nx, const double* knots, const double* coef_x, const double* coef_y, double s, double* out)
{
    int i;
    double dx;
    if (spline_seg(nx, knots, s, &i, &dx)) return 1;
    out[0] = sp_pos(coef_x, nx, i, dx);
    out[1] = sp_pos(coef_y, nx, i, dx);
    double ddx = sp_d1(coef_x, nx, i, dx), ddy = sp_d1(coef_y, nx, i, dx);
    out[2] = atan2(ddy, ddx); /* calc_yaw :229-231 */
    double d2x = sp_d2(coef_x, nx, i, dx), d2y = sp_d2(coef_y, nx, i, dx);
    out[3] = (d2y * ddx - d2x * ddy) / pow(ddx * ddx + ddy * ddy, 1.5); /* calc_curvature :207-211 */
    return 0;
}

/* ------------------------------------------------------------------------ */
/* one trajectory                                                            */
/* ------------------------------------------------------------------------ */
typedef struct {
    int N, M;
    double* a[ORC_NARR]; /* each of length N (x.. arrays use the first M / M-1 / .. entries) */
    double* block;
} traj_t;

/* One trajectory is alive per thread at a time (orc_eval_traj), so its 16 arrays live in a per-thread scratch block that only
   grows: a batch of plans on many threads does not go through the allocator once per candidate. */
static _Thread_local double* tl_block = NULL;
static _Thread_local size_t tl_cap = 0;

static int traj_alloc(traj_t* t, int N)
{
    const size_t n = (size_t)(N > 0 ? N : 1);
    t->N = N;
    t->M = 0;
    if (tl_cap < (size_t)ORC_NARR * n) {
        double* nb = (double*)realloc(tl_block, sizeof(double) * (size_t)ORC_NARR * n);
        if (!nb) return -3;
        tl_block = nb;
        tl_cap = (size_t)ORC_NARR * n;
    }
    t->block = tl_block;
    for (int k = 0; k < ORC_NARR; ++k) t->a[k] = t->block + (size_t)k * n;
    return 0;
}
static void traj_free(traj_t* t) { t->block = NULL; }

/* len(np.arange(0.0, T, tick)) */
static int arange_len(double T, double tick)
{
    double n = ceil(T / tick);
    if (!(n > 0)) return 0;
    return (int)n;
}

/* Frenet-frame generation + cost: frenet_optimal_planner.py:79-99,
   fiss_planner.py:113-130, cost_function.py:29-50 */
static int traj_generate(const orc_problem* p, double d_end, double v_end, double T_end, traj_t* t, double* cost)
{
    double lat[6], lon[5];
    if (orc_quintic_coefs(p->ego[3], p->ego[4], p->ego[5], d_end, 0.0, 0.0, T_end, lat)) return -1;
    if (orc_quartic_coefs(p->ego[0], p->ego[1], p->ego[2], v_end, 0.0, T_end, lon)) return -1;
    int N = t->N;
    for (int i = 0; i < N; ++i) {
        double ti = (double)i * p->tick_t; /* np.arange: start + i*step */
        double o[4];
        t->a[ORC_T][i] = ti;
        orc_poly_eval(lat, 5, ti, o);
        t->a[ORC_D][i] = o[0]; t->a[ORC_D_D][i] = o[1]; t->a[ORC_D_DD][i] = o[2]; t->a[ORC_D_DDD][i] = o[3];
        orc_poly_eval(lon, 4, ti, o);
        t->a[ORC_S][i] = o[0]; t->a[ORC_S_D][i] = o[1]; t->a[ORC_S_DD][i] = o[2]; t->a[ORC_S_DDD][i] = o[3];
    }
    /* cost_function.py:41-50; python sum() = left-to-right from 0 */
    double s_speed = 0, s_acc_s = 0, s_acc_d = 0, s_jerk_s = 0, s_jerk_d = 0, s_off = 0;
    for (int i = 0; i < N; ++i) { double e = t->a[ORC_S_D][i] - p->target_speed; s_speed += e * e; }
    for (int i = 0; i < N; ++i) s_acc_s += t->a[ORC_S_DD][i] * t->a[ORC_S_DD][i];
    for (int i = 0; i < N; ++i) s_acc_d += t->a[ORC_D_DD][i] * t->a[ORC_D_DD][i];
    for (int i = 0; i < N; ++i) s_jerk_s += t->a[ORC_S_DDD][i] * t->a[ORC_S_DDD][i];
    for (int i = 0; i < N; ++i) s_jerk_d += t->a[ORC_D_DDD][i] * t->a[ORC_D_DDD][i];
    for (int i = 0; i < N; ++i) s_off += t->a[ORC_D][i] * t->a[ORC_D][i];
    const double w_V = 1, w_A = 0.1, w_J = 0.1, w_LC = 10; /* cost_function.py:6-12 */
    double cost_time = 10.0 - (N > 0 ? t->a[ORC_T][N - 1] : 0.0);
    double cost_obstacle = 0.0;
    double cost_speed = w_V * s_speed;
    double cost_accel = w_A * s_acc_s + w_A * s_acc_d;
    double cost_jerk = w_J * s_jerk_s + w_J * s_jerk_d;
    double cost_offset = w_LC * s_off;
    *cost = (cost_time + cost_obstacle + cost_speed + cost_accel + cost_jerk + cost_offset) / (double)N;
    return 0;
}

/* calc_global_paths, frenet_optimal_planner.py:106-138 */
static void traj_to_global(const orc_problem* p, traj_t* t)
{
    int N = t->N, M = 0;
    for (int k = ORC_X; k < ORC_NARR; ++k)
        for (int i = 0; i < N; ++i) t->a[k][i] = NAN;
    for (int i = 0; i < N; ++i) {
        int seg;
        double dx;
        if (spline_seg(p->nx, p->knots, t->a[ORC_S][i], &seg, &dx)) break; /* :112-113 */
        double ix = sp_pos(p->coef_x, p->nx, seg, dx), iy = sp_pos(p->coef_y, p->nx, seg, dx);
        double i_yaw = atan2(sp_d1(p->coef_y, p->nx, seg, dx), sp_d1(p->coef_x, p->nx, seg, dx));
        double di = t->a[ORC_D][i];
        t->a[ORC_X][i] = ix + di * cos(i_yaw + ORC_PI / 2.0);
        t->a[ORC_Y][i] = iy + di * sin(i_yaw + ORC_PI / 2.0);
        ++M;
    }
    t->M = M;
    if (M >= 2) {
        double dt = p->tick_t;
        for (int i = 0; i < M - 1; ++i) {
            double xd = t->a[ORC_X][i + 1] - t->a[ORC_X][i], yd = t->a[ORC_Y][i + 1] - t->a[ORC_Y][i];
            t->a[ORC_YAW][i] = atan2(yd, xd);
            t->a[ORC_DS][i] = hypot(xd, yd);
        }
        t->a[ORC_YAW][M - 1] = t->a[ORC_YAW][M - 2];
        for (int i = 0; i < M - 1; ++i) t->a[ORC_C][i] = (t->a[ORC_YAW][i + 1] - t->a[ORC_YAW][i]) / t->a[ORC_DS][i];
        for (int i = 0; i < M - 2; ++i) t->a[ORC_C_D][i] = (t->a[ORC_C][i + 1] - t->a[ORC_C][i]) / dt;
        for (int i = 0; i < M - 3; ++i) t->a[ORC_C_DD][i] = (t->a[ORC_C_D][i + 1] - t->a[ORC_C_D][i]) / dt;
    } else {
        /* reference keeps x,y as python lists and leaves yaw.. empty */
    }
}

/* check_constraints, frenet_optimal_planner.py:140-160 */
static uint32_t traj_constraints(const orc_problem* p, const traj_t* t)
{
    uint32_t f = 0;
    for (int i = 0; i < t->N; ++i)
        if (t->a[ORC_S_D][i] > p->max_speed) { f |= ORC_FLAG_SPEED; break; }
    for (int i = 0; i < t->N; ++i)
        if (fabs(t->a[ORC_S_DD][i]) > p->max_accel) { f |= ORC_FLAG_ACCEL; break; }
    if (p->curvature_mask && t->M >= 2) { /* :145-150 (commented out in the reference); c / c_d / c_dd are [] when M < 2 */
        for (int i = 0; i < t->M - 1; ++i)
            if (fabs(t->a[ORC_C][i]) > p->max_curvature) { f |= ORC_FLAG_CURVATURE; break; }
        for (int i = 0; i < t->M - 2; ++i)
            if (fabs(t->a[ORC_C_D][i]) > p->max_kappa_d) { f |= ORC_FLAG_KAPPA_D; break; }
        for (int i = 0; i < t->M - 3; ++i)
            if (fabs(t->a[ORC_C_DD][i]) > p->max_kappa_dd) { f |= ORC_FLAG_KAPPA_DD; break; }
    }
    return f;
}

/* shapely.affinity.translate + rotate(origin='center', use_radians=True) on a
   centred l x w rectangle (construct_polygon, frenet_optimal_planner.py:162-166;
   vehicle.py:24-31).  Returns -1 for a non-finite coordinate. */
static int make_box(double l, double w, double x, double y, double yaw, double c[4][2])
{
    double hl = l / 2, hw = w / 2;
    double base[4][2] = {{hl, hw}, {hl, -hw}, {-hl, -hw}, {-hl, hw}};
    double tx[4], ty[4];
    double minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY;
    for (int k = 0; k < 4; ++k) {
        tx[k] = base[k][0] + x;
        ty[k] = base[k][1] + y;
        if (!isfinite(tx[k]) || !isfinite(ty[k])) return -1;
        if (tx[k] < minx) minx = tx[k];
        if (tx[k] > maxx) maxx = tx[k];
        if (ty[k] < miny) miny = ty[k];
        if (ty[k] > maxy) maxy = ty[k];
    }
    if (!isfinite(yaw)) return -1;
    double cosp = cos(yaw), sinp = sin(yaw);
    if (fabs(cosp) < 2.5e-16) cosp = 0.0;
    if (fabs(sinp) < 2.5e-16) sinp = 0.0;
    double x0 = (minx + maxx) / 2.0, y0 = (miny + maxy) / 2.0;
    double xoff = x0 - x0 * cosp + y0 * sinp;
    double yoff = y0 - x0 * sinp - y0 * cosp;
    for (int k = 0; k < 4; ++k) {
        c[k][0] = cosp * tx[k] - sinp * ty[k] + xoff;
        c[k][1] = sinp * tx[k] + cosp * ty[k] + yoff;
    }
    return 0;
}

/* Polygon.intersects for two convex quads: closed-set separating-axis test
   (touching counts as intersecting). */
static int quads_intersect(double A[4][2], double B[4][2])
{
    for (int pass = 0; pass < 2; ++pass) {
        double(*P)[2] = pass ? B : A;
        for (int k = 0; k < 4; ++k) {
            double ex = P[(k + 1) & 3][0] - P[k][0], ey = P[(k + 1) & 3][1] - P[k][1];
            double ax = -ey, ay = ex;
            double amin = INFINITY, amax = -INFINITY, bmin = INFINITY, bmax = -INFINITY;
            for (int v = 0; v < 4; ++v) {
                double pa = A[v][0] * ax + A[v][1] * ay;
                double pb = B[v][0] * ax + B[v][1] * ay;
                if (pa < amin) amin = pa;
                if (pa > amax) amax = pa;
                if (pb < bmin) bmin = pb;
                if (pb > bmax) bmax = pb;
            }
            if (amax < bmin || bmax < amin) return 0;
        }
    }
    return 1;
}

int orc_boxes_intersect(double l1, double w1, double x1, double y1, double yaw1, double l2, double w2, double x2, double y2,
                        double yaw2)
{
    double A[4][2], B[4][2];
    if (make_box(l1, w1, x1, y1, yaw1, A) || make_box(l2, w2, x2, y2, yaw2, B)) return -1;
    return quads_intersect(A, B);
}

/* ---- the same predicate, decided EXACTLY on the fp64 vertex coordinates (what a robust geometry kernel such as GEOS decides with
   its orientation predicates).  Test infrastructure for the collision audit (tests/test_collision_exact.py): the float test above
   rounds its projections, this one cannot be wrong about the polygons it is handed.
   Exactness: a vertex coordinate is a double; the difference of two doubles is exact in binary128 as long as their exponents are
   within 60 of each other (or one is zero) - true for every coordinate a scene produces; the product of two such differences
   has at most 108 significant bits (exact); the SIGN of the correctly rounded difference of two exact values is exact. */
static int orient_sign(const double* a, const double* b, const double* c)
{
    const __float128 abx = (__float128)b[0] - (__float128)a[0], aby = (__float128)b[1] - (__float128)a[1];
    const __float128 acx = (__float128)c[0] - (__float128)a[0], acy = (__float128)c[1] - (__float128)a[1];
    const __float128 l = abx * acy, r = aby * acx;
    return (l > r) - (l < r);
}

/* Two closed convex polygons are disjoint iff the line through some edge of one has the whole other polygon strictly on its outer
   side. */
static int quads_intersect_exact(double A[4][2], double B[4][2])
{
    for (int pass = 0; pass < 2; ++pass) {
        double(*P)[2] = pass ? B : A;
        double(*Q)[2] = pass ? A : B;
        for (int k = 0; k < 4; ++k) {
            const double* p0 = P[k];
            const double* p1 = P[(k + 1) & 3];
            int inside = orient_sign(p0, p1, P[(k + 2) & 3]);
            if (inside == 0) inside = orient_sign(p0, p1, P[(k + 3) & 3]);
            if (inside == 0) continue; /* degenerate edge: no half plane */
            int all_out = 1;
            for (int v = 0; v < 4 && all_out; ++v) {
                const int o = orient_sign(p0, p1, Q[v]);
                if (o == 0 || o == inside) all_out = 0;
            }
            if (all_out) return 0;
        }
    }
    return 1;
}

int orc_boxes_intersect_exact(double l1, double w1, double x1, double y1, double yaw1, double l2, double w2, double x2, double y2,
                              double yaw2)
{
    double A[4][2], B[4][2];
    if (make_box(l1, w1, x1, y1, yaw1, A) || make_box(l2, w2, x2, y2, yaw2, B)) return -1;
    return quads_intersect_exact(A, B);
}

/* n pairs of boxes (l, w, x, y, yaw): out[i] = 1 / 0 / -1 (unbuildable); exact = 0: the float test, 1: the exact one. */
int orc_boxes_intersect_batch(int32_t n, const double* a, const double* b, int32_t exact, int32_t threads, int8_t* out)
{
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static, 1024)
#endif
    for (int i = 0; i < n; ++i) {
        double A[4][2], B[4][2];
        const double* pa = a + (size_t)i * 5;
        const double* pb = b + (size_t)i * 5;
        if (make_box(pa[0], pa[1], pa[2], pa[3], pa[4], A) || make_box(pb[0], pb[1], pb[2], pb[3], pb[4], B)) out[i] = -1;
        else out[i] = (int8_t)(exact ? quads_intersect_exact(A, B) : quads_intersect(A, B));
    }
    (void)threads;
    return 0;
}

/* the rotated rectangle's four vertices as the reference's construct_polygon produces them (for the audit's second, independent
   exact implementation in python) */
int orc_box_vertices(double l, double w, double x, double y, double yaw, double* out8)
{
    double A[4][2];
    if (make_box(l, w, x, y, yaw, A)) return -1;
    for (int k = 0; k < 4; ++k) { out8[2 * k] = A[k][0]; out8[2 * k + 1] = A[k][1]; }
    return 0;
}

/* ---- obstacle shapes other than rectangles.  `obstacle_shape.shapely_object` is any polygon (frenet_optimal_planner.py:189-191); a
   problem's column j with obs_nvert[j] >= 3 carries a convex counter-clockwise ring relative to its rotation centre (the centre of the
   shape's bounding box, which the column's poses carry).  construct_polygon (:162-166) on it: translate every vertex by the pose,
   rotate about (x0, y0) = the pose position with shapely's affine arithmetic (cos / sin snapped below 2.5e-16). */
static int make_ring(const double* u, int n, double x, double y, double yaw, double* out)
{
    if (!isfinite(x) || !isfinite(y) || !isfinite(yaw)) return -1;
    double cosp = cos(yaw), sinp = sin(yaw);
    if (fabs(cosp) < 2.5e-16) cosp = 0.0;
    if (fabs(sinp) < 2.5e-16) sinp = 0.0;
    const double x0 = x, y0 = y;
    const double xoff = x0 - x0 * cosp + y0 * sinp;
    const double yoff = y0 - x0 * sinp - y0 * cosp;
    for (int k = 0; k < n; ++k) {
        const double tx = u[2 * k] + x, ty = u[2 * k + 1] + y;
        if (!isfinite(tx) || !isfinite(ty)) return -1;
        out[2 * k] = cosp * tx - sinp * ty + xoff;
        out[2 * k + 1] = sinp * tx + cosp * ty + yoff;
    }
    return 0;
}

/* Polygon.intersects for two convex polygons (rings of na / nb vertices): closed-set separating-axis test over the edge normals of
   both (touching counts as intersecting) - quads_intersect for any vertex count. */
static int convex_intersect(const double* A, int na, const double* B, int nb)
{
    for (int pass = 0; pass < 2; ++pass) {
        const double* P = pass ? B : A;
        const int np_ = pass ? nb : na;
        for (int k = 0; k < np_; ++k) {
            const int k1 = (k + 1) % np_;
            const double ex = P[2 * k1] - P[2 * k], ey = P[2 * k1 + 1] - P[2 * k + 1];
            const double ax = -ey, ay = ex;
            double amin = INFINITY, amax = -INFINITY, bmin = INFINITY, bmax = -INFINITY;
            for (int v = 0; v < na; ++v) {
                const double pa = A[2 * v] * ax + A[2 * v + 1] * ay;
                if (pa < amin) amin = pa;
                if (pa > amax) amax = pa;
            }
            for (int v = 0; v < nb; ++v) {
                const double pb = B[2 * v] * ax + B[2 * v + 1] * ay;
                if (pb < bmin) bmin = pb;
                if (pb > bmax) bmax = pb;
            }
            if (amax < bmin || bmax < amin) return 0;
        }
    }
    return 1;
}

/* the same predicate decided exactly (see quads_intersect_exact) */
static int convex_intersect_exact(const double* A, int na, const double* B, int nb)
{
    for (int pass = 0; pass < 2; ++pass) {
        const double* P = pass ? B : A;
        const double* Q = pass ? A : B;
        const int np_ = pass ? nb : na, nq = pass ? na : nb;
        for (int k = 0; k < np_; ++k) {
            const double* p0 = P + 2 * k;
            const double* p1 = P + 2 * ((k + 1) % np_);
            int inside = 0;
            for (int m = 2; m < np_ && inside == 0; ++m) inside = orient_sign(p0, p1, P + 2 * ((k + m) % np_));
            if (inside == 0) continue; /* degenerate edge: no half plane */
            int all_out = 1;
            for (int v = 0; v < nq && all_out; ++v) {
                const int o = orient_sign(p0, p1, Q + 2 * v);
                if (o == 0 || o == inside) all_out = 0;
            }
            if (all_out) return 0;
        }
    }
    return 1;
}

/* Ego box (l, w at x, y, yaw) against a convex ring `u` (n vertices relative to its rotation centre) at pose (ox, oy, oyaw):
   1 / 0 / -1 (a polygon could not be built).  exact = 0: the float test, 1: the exact one.  world (may be NULL): the ring's 2 n
   world coordinates, for the audit's second implementation in python. */
int orc_box_ring_intersect(double l, double w, double x, double y, double yaw, const double* u, int32_t n, double ox, double oy,
                           double oyaw, int32_t exact, double* world)
{
    double A[4][2], B[2 * ORC_MAX_POLY_VERTS];
    if (n < 3 || n > ORC_MAX_POLY_VERTS) return -1;
    if (make_box(l, w, x, y, yaw, A) || make_ring(u, n, ox, oy, oyaw, B)) return -1;
    if (world) memcpy(world, B, sizeof(double) * 2 * (size_t)n);
    return exact ? convex_intersect_exact(&A[0][0], 4, B, n) : convex_intersect(&A[0][0], 4, B, n);
}

/* has_collision, frenet_optimal_planner.py:168-195 */
static int traj_has_collision(const orc_problem* p, const traj_t* t)
{
    if (p->n_obs <= 0) return 0; /* :170-171 */
    int t_step_max = t->M < (p->final_time_step - p->t_now) ? t->M : (p->final_time_step - p->t_now);
    int stride = p->check_stride > 0 ? p->check_stride : 1;
    for (int i = 0; i < t_step_max; ++i) {
        if (i % stride != 0) continue;
        double ego[4][2];
        /* M == 1: traj.yaw is an empty list -> IndexError -> bare except -> collision (:178-182) */
        if (t->M < 2) return 1;
        if (make_box(p->veh_l, p->veh_w, t->a[ORC_X][i], t->a[ORC_Y][i], t->a[ORC_YAW][i], ego)) return 1;
        int t_step = i + p->t_now;
        for (int j = 0; j < p->n_obs; ++j) {
            if (t_step < 0 || t_step >= p->T_obs) continue; /* state_at_time -> None */
            const double* ps = p->obs_pose + ((size_t)t_step * (size_t)p->n_obs + (size_t)j) * 4;
            if (ps[3] == 0.0) continue; /* :187-188 */
            const int nvert = p->obs_nvert ? p->obs_nvert[j] : 0;
            if (nvert >= 3) { /* a polygon column: the ring itself (obs_dims only holds the box around it) */
                double ring[2 * ORC_MAX_POLY_VERTS];
                if (nvert > ORC_MAX_POLY_VERTS) return -1;
                if (make_ring(p->obs_poly + (size_t)j * 2 * (size_t)p->poly_stride, nvert, ps[0], ps[1], ps[2], ring)) continue;
                if (convex_intersect(&ego[0][0], 4, ring, nvert)) return 1;
                continue;
            }
            double ob[4][2];
            if (make_box(p->obs_dims[2 * j], p->obs_dims[2 * j + 1], ps[0], ps[1], ps[2], ob)) continue;
            if (quads_intersect(ego, ob)) return 1;
        }
    }
    return 0;
}

static void traj_dump(const traj_t* t, double* out, int stride)
{
    if (!out) return;
    for (int k = 0; k < ORC_NARR; ++k)
        for (int i = 0; i < stride; ++i) out[(size_t)k * stride + i] = NAN;
    int N = t->N < stride ? t->N : stride;
    for (int k = 0; k < ORC_X; ++k)
        for (int i = 0; i < N; ++i) out[(size_t)k * stride + i] = t->a[k][i];
    for (int k = ORC_X; k < ORC_NARR; ++k)
        for (int i = 0; i < N; ++i) out[(size_t)k * stride + i] = t->a[k][i];
}

int orc_eval_traj(const orc_problem* p, double d_end, double v_end, double T_end, int do_collision, double* traj,
                  int32_t stride, int32_t* N, int32_t* M, double* cost, uint32_t* flags)
{
    traj_t t;
    int n = arange_len(T_end, p->tick_t);
    if (n <= 0) return -5;
    int rc = traj_alloc(&t, n);
    if (rc) return rc;
    double c = NAN;
    rc = traj_generate(p, d_end, v_end, T_end, &t, &c);
    if (rc) { traj_free(&t); return rc; }
    traj_to_global(p, &t);
    uint32_t f = traj_constraints(p, &t);
    if (t.M < t.N) f |= ORC_FLAG_TRUNCATED;
    if (do_collision && traj_has_collision(p, &t)) f |= ORC_FLAG_COLLISION;
    if (N) *N = t.N;
    if (M) *M = t.M;
    if (cost) *cost = c;
    if (flags) *flags = f;
    traj_dump(&t, traj, stride);
    traj_free(&t);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* dense lattice + FOP                                                       */
/* ------------------------------------------------------------------------ */
int orc_dense_tables(const orc_problem* p, double* cost, uint32_t* flags)
{
    /* loop nest d -> T -> v, frenet_optimal_planner.py:75-100 */
    for (int id = 0; id < p->nd; ++id)
        for (int it = 0; it < p->nt; ++it)
            for (int iv = 0; iv < p->nv; ++iv) {
                int idx = (id * p->nt + it) * p->nv + iv;
                int32_t N, M;
                double c;
                uint32_t f;
                int rc = orc_eval_traj(p, p->d_samples[id], p->v_samples[iv], p->t_samples[it], 1, NULL, 0, &N, &M, &c, &f);
                if (rc) return rc;
                if (cost) cost[idx] = c;
                if (flags) flags[idx] = f | ((uint32_t)N << 8) | ((uint32_t)M << 20);
            }
    return 0;
}

int orc_fop_plan(const orc_problem* p, int32_t* best_idx, double* best_cost, int32_t* stats, double* cost_tbl,
                 uint32_t* flag_tbl)
{
    int C = p->nd * p->nv * p->nt;
    double* cost = cost_tbl ? cost_tbl : (double*)malloc(sizeof(double) * (size_t)C);
    uint32_t* flags = flag_tbl ? flag_tbl : (uint32_t*)malloc(sizeof(uint32_t) * (size_t)C);
    if (!cost || !flags) return -3;
    int rc = orc_dense_tables(p, cost, flags);
    if (rc == 0) {
        /* :263-268: `if min_cost >= fp.cost_final` -> the LAST minimal survivor wins */
        double min_cost = INFINITY;
        int best = -1;
        for (int i = 0; i < C; ++i) {
            if (flags[i] & ORC_FLAG_INFEASIBLE) continue;
            if (min_cost >= cost[i]) { min_cost = cost[i]; best = i; }
        }
        *best_idx = best;
        *best_cost = best >= 0 ? min_cost : NAN;
        if (stats) { stats[0] = 0; stats[1] = C; stats[2] = C; stats[3] = C; } /* :254-256 */
    }
    if (!cost_tbl) free(cost);
    if (!flag_tbl) free(flags);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* FOP+ : queue.PriorityQueue == heapq over objects ordered by cost_final     */
/* (frenet.py:150-166).  heapq's sift functions are restated so that exact    */
/* cost ties pop in CPython's order.                                          */
/* ------------------------------------------------------------------------ */
typedef struct { double cost; int idx; } hitem;

static void heap_siftdown(hitem* h, int startpos, int pos)
{
    hitem newitem = h[pos];
    while (pos > startpos) {
        int parentpos = (pos - 1) >> 1;
        if (newitem.cost < h[parentpos].cost) { h[pos] = h[parentpos]; pos = parentpos; continue; }
        break;
    }
    h[pos] = newitem;
}
static void heap_siftup(hitem* h, int n, int pos)
{
    int endpos = n, startpos = pos;
    hitem newitem = h[pos];
    int childpos = 2 * pos + 1;
    while (childpos < endpos) {
        int rightpos = childpos + 1;
        if (rightpos < endpos && !(h[childpos].cost < h[rightpos].cost)) childpos = rightpos;
        h[pos] = h[childpos];
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    h[pos] = newitem;
    heap_siftdown(h, startpos, pos);
}
static void heap_push(hitem* h, int* n, hitem it) { h[*n] = it; ++*n; heap_siftdown(h, 0, *n - 1); }
static hitem heap_pop(hitem* h, int* n)
{
    hitem last = h[*n - 1];
    --*n;
    if (*n > 0) { hitem ret = h[0]; h[0] = last; heap_siftup(h, *n, 0); return ret; }
    return last;
}

int orc_fopplus_plan(const orc_problem* p, int32_t* best_idx, double* best_cost, int32_t* stats)
{
    /* fop_plus_planner.py:16-41 */
    int C = p->nd * p->nv * p->nt;
    hitem* heap = (hitem*)malloc(sizeof(hitem) * (size_t)C);
    if (!heap) return -3;
    int n = 0;
    for (int id = 0; id < p->nd; ++id)
        for (int it = 0; it < p->nt; ++it)
            for (int iv = 0; iv < p->nv; ++iv) {
                int idx = (id * p->nt + it) * p->nv + iv;
                double c;
                int rc = orc_eval_traj(p, p->d_samples[id], p->v_samples[iv], p->t_samples[it], 0, NULL, 0, NULL, NULL, &c, NULL);
                if (rc) { free(heap); return rc; }
                hitem hi = {c, idx};
                heap_push(heap, &n, hi);
            }
    stats[0] = 0; stats[1] = C; stats[2] = 0; stats[3] = 0;
    *best_idx = -1;
    *best_cost = NAN;
    while (n > 0) {
        stats[0] += 1;
        hitem c = heap_pop(heap, &n);
        int iv = c.idx % p->nv, it = (c.idx / p->nv) % p->nt, id = c.idx / (p->nv * p->nt);
        uint32_t f;
        double cc;
        orc_eval_traj(p, p->d_samples[id], p->v_samples[iv], p->t_samples[it], 1, NULL, 0, NULL, NULL, &cc, &f);
        stats[2] += 1;
        stats[3] += 1;
        if (!(f & ORC_FLAG_INFEASIBLE)) { *best_idx = c.idx; *best_cost = c.cost; break; }
    }
    free(heap);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* FISS / FISS+                                                              */
/* ------------------------------------------------------------------------ */
typedef struct {
    const orc_problem* p;
    int nd, nv, nt, C;
    double* est;        /* cost_est  [nd][nv][nt] */
    double* cost;       /* cost_final, valid where generated */
    unsigned char* gen; /* is_generated */
    unsigned char* inq; /* member of candidate_trajs */
    int32_t* stats;
} fiss_t;

static inline int fidx(const fiss_t* f, int i, int j, int k) { return (i * f->nv + j) * f->nt + k; }

int orc_fiss_cost_est(const orc_problem* p, double w_heuristic, const int32_t* prev, double* est)
{
    /* fiss_planner.py:33-99 */
    int nd = p->nd, nv = p->nv, nt = p->nt;
    double max_sqr_dist = (double)(nd * nd + nv * nv + nt * nt);
    double left_bound = p->samp_min[0], right_bound = p->samp_max[0];
    double lat_norm = fmax(left_bound * left_bound, right_bound * right_bound);
    for (int i = 0; i < nd; ++i) {
        double d = p->d_samples[i];
        double cost_est_lat = (d * d) / lat_norm;
        for (int j = 0; j < nv; ++j) {
            double v = p->v_samples[j];
            double e = p->samp_max[1] - v, r = p->samp_max[1] - p->samp_min[1];
            double cost_est_speed = (e * e) / (r * r);
            for (int k = 0; k < nt; ++k) {
                double t = p->t_samples[k];
                double cost_est_time = 1.0 - (t - p->samp_min[2]) / (p->samp_max[2] - p->samp_min[2]);
                double cost_est = cost_est_lat + cost_est_time + cost_est_speed;
                double cost_heu = 0.0;
                if (prev && prev[0] >= 0) {
                    int a = i - prev[0], b = j - prev[1], c = k - prev[2];
                    double heu_sqr_dist = (double)(a * a + b * b + c * c);
                    cost_heu = w_heuristic * heu_sqr_dist / max_sqr_dist;
                }
                est[(i * nv + j) * nt + k] = cost_est + cost_heu;
            }
        }
    }
    return 0;
}

/* generate_trajectory, fiss_planner.py:101-138 */
static int fiss_generate(fiss_t* f, int i, int j, int k, int* is_new, double* cost)
{
    int q = fidx(f, i, j, k);
    if (f->gen[q]) { *is_new = 0; *cost = f->cost[q]; return 0; }
    f->stats[1] += 1;
    f->gen[q] = 1;
    double c;
    int rc = orc_eval_traj(f->p, f->p->d_samples[i], f->p->v_samples[j], f->p->t_samples[k], 0, NULL, 0, NULL, NULL, &c, NULL);
    if (rc) return rc;
    f->cost[q] = c;
    f->inq[q] = 1; /* candidate_trajs.put((cost_final, idx)) */
    *is_new = 1;
    *cost = c;
    return 0;
}

/* candidate_trajs head.  The reference stores (cost, ndarray) tuples, so an
   exact cost tie raises ValueError there; here a tie resolves to the lower
   raster index (documented divergence). */
static int fiss_queue_min(const fiss_t* f)
{
    int best = -1;
    for (int q = 0; q < f->C; ++q)
        if (f->inq[q] && (best < 0 || f->cost[q] < f->cost[best])) best = q;
    return best;
}

/* find_initial_guess, fiss_planner.py:140-150: `<=` -> last minimum wins */
static int fiss_initial_guess(const fiss_t* f)
{
    int best = -1;
    double min_cost = INFINITY;
    for (int q = 0; q < f->C; ++q)
        if (!f->gen[q] && f->est[q] <= min_cost) { min_cost = f->est[q]; best = q; }
    return best;
}

static int fiss_init(fiss_t* f, const orc_problem* p, double w_heu, const int32_t* prev, int32_t* stats)
{
    f->p = p; f->nd = p->nd; f->nv = p->nv; f->nt = p->nt; f->C = p->nd * p->nv * p->nt;
    f->est = (double*)malloc(sizeof(double) * (size_t)f->C);
    f->cost = (double*)malloc(sizeof(double) * (size_t)f->C);
    f->gen = (unsigned char*)calloc((size_t)f->C, 1);
    f->inq = (unsigned char*)calloc((size_t)f->C, 1);
    f->stats = stats;
    if (!f->est || !f->cost || !f->gen || !f->inq) return -3;
    stats[0] = stats[1] = stats[2] = stats[3] = 0;
    return orc_fiss_cost_est(p, w_heu, prev, f->est);
}
static void fiss_free(fiss_t* f) { free(f->est); free(f->cost); free(f->gen); free(f->inq); }

/* validation step shared by FISS and FISS+ (fiss_planner.py:229-258).
   returns 1 if the popped candidate is the answer, 0 to continue, <0 on error */
static int fiss_validate(fiss_t* f, int q, uint32_t* flags_out)
{
    int k = q % f->nt, j = (q / f->nt) % f->nv, i = q / (f->nt * f->nv);
    f->inq[q] = 0;
    f->stats[2] += 1;
    /* constraints first; collision only when they pass (:240-246) */
    uint32_t fl;
    double c;
    int rc = orc_eval_traj(f->p, f->p->d_samples[i], f->p->v_samples[j], f->p->t_samples[k], 0, NULL, 0, NULL, NULL, &c, &fl);
    if (rc) return rc;
    if (fl & ORC_FLAG_CONSTRAINTS) { *flags_out = fl; return 0; }
    rc = orc_eval_traj(f->p, f->p->d_samples[i], f->p->v_samples[j], f->p->t_samples[k], 1, NULL, 0, NULL, NULL, &c, &fl);
    if (rc) return rc;
    f->stats[3] += 1;
    *flags_out = fl;
    return (fl & ORC_FLAG_COLLISION) ? 0 : 1;
}

int orc_fiss_plan(const orc_problem* p, double w_heuristic, int32_t* prev_best_idx, int32_t* best_ijk, double* best_cost,
                  int32_t* stats)
{
    fiss_t f;
    int rc = fiss_init(&f, p, w_heuristic, prev_best_idx, stats);
    if (rc) { fiss_free(&f); return rc; }
    int sizes[3] = {f.nd, f.nv, f.nt};
    best_ijk[0] = best_ijk[1] = best_ijk[2] = -1;
    *best_cost = NAN;
    int found = 0;
    while (!found) {
        stats[0] += 1;
        int q = fiss_queue_min(&f);
        if (q < 0) {
            q = fiss_initial_guess(&f);
            if (q < 0) break; /* :203-206 */
        }
        int idx[3] = {q / (f.nt * f.nv), (q / f.nt) % f.nv, q % f.nt};
        /* search: explore_next_sample (:174-188) until it lands on a generated sample */
        for (;;) {
            if (f.gen[fidx(&f, idx[0], idx[1], idx[2])]) break;
            /* find_gradients (:152-172) */
            int is_new;
            double cost_center, cost;
            if ((rc = fiss_generate(&f, idx[0], idx[1], idx[2], &is_new, &cost_center))) goto done;
            double grad[3];
            for (int dim = 0; dim < 3; ++dim) {
                int nb[3] = {idx[0], idx[1], idx[2]};
                if (idx[dim] < sizes[dim] - 1) {
                    nb[dim] += 1;
                    if ((rc = fiss_generate(&f, nb[0], nb[1], nb[2], &is_new, &cost))) goto done;
                    grad[dim] = cost - cost_center;
                    if (grad[dim] >= 0 && idx[dim] == 0) grad[dim] = 0.0;
                } else {
                    nb[dim] -= 1;
                    if (nb[dim] < 0) nb[dim] = sizes[dim] - 1; /* python negative index wraps (size-1 axis) */
                    if ((rc = fiss_generate(&f, nb[0], nb[1], nb[2], &is_new, &cost))) goto done;
                    grad[dim] = cost_center - cost;
                    if (grad[dim] <= 0 && idx[dim] == sizes[dim] - 1) grad[dim] = 0.0;
                }
            }
            for (int dim = 0; dim < 3; ++dim) {
                idx[dim] += (grad[dim] > 0.0) ? -1 : +1;
                if (idx[dim] < 0) idx[dim] = 0;
                if (idx[dim] > sizes[dim] - 1) idx[dim] = sizes[dim] - 1;
            }
        }
        /* validation (:229-258) */
        q = fiss_queue_min(&f);
        if (q < 0) break;
        uint32_t fl;
        int v = fiss_validate(&f, q, &fl);
        if (v < 0) { rc = v; goto done; }
        if (v == 1) {
            found = 1;
            best_ijk[0] = q / (f.nt * f.nv); best_ijk[1] = (q / f.nt) % f.nv; best_ijk[2] = q % f.nt;
            *best_cost = f.cost[q];
            prev_best_idx[0] = best_ijk[0]; prev_best_idx[1] = best_ijk[1]; prev_best_idx[2] = best_ijk[2]; /* :252 */
        }
    }
done:
    fiss_free(&f);
    return rc;
}

/* one refinement trajectory (generate_trajectory_by_end_state, fiss_plus_planner.py:172-205) */
typedef struct { double x[3]; double cost; int order; int alive; } rtraj;

int orc_fissplus_plan(const orc_problem* p, double w_heuristic, int32_t max_refine_iters, double decaying_factor,
                      int32_t* prev_best_idx, int32_t* best_ijk, double* best_cost, int32_t* stats, int32_t* refined,
                      double* end_state, double* trace)
{
    fiss_t f;
    int rc = fiss_init(&f, p, w_heuristic, prev_best_idx, stats);
    if (rc) { fiss_free(&f); return rc; }
    int sizes[3] = {f.nd, f.nv, f.nt};
    unsigned char* frontier = (unsigned char*)calloc((size_t)f.C, 1);
    rtraj* rt = (rtraj*)calloc((size_t)(7 * (max_refine_iters > 0 ? max_refine_iters : 1)), sizeof(rtraj));
    if (!frontier || !rt) { rc = -3; goto done; }
    best_ijk[0] = best_ijk[1] = best_ijk[2] = -1;
    *best_cost = NAN;
    *refined = 0;
    end_state[0] = end_state[1] = end_state[2] = NAN;
    if (trace)
        for (int i = 0; i < max_refine_iters * 7 * 4; ++i) trace[i] = NAN;
    int found = 0, best_q = -1;
    while (!found) {
        stats[0] += 1;
        int q = fiss_queue_min(&f);
        if (q < 0) {
            q = fiss_initial_guess(&f);
            if (q < 0) break;
        }
        /* search (fiss_plus_planner.py:106-116) */
        for (;;) {
            int idx[3] = {q / (f.nt * f.nv), (q / f.nt) % f.nv, q % f.nt};
            /* explore_neighbors (:30-59) */
            int is_new;
            double cost_center, cost;
            if ((rc = fiss_generate(&f, idx[0], idx[1], idx[2], &is_new, &cost_center))) goto done;
            for (int dim = 0; dim < 3; ++dim) {
                if (idx[dim] >= 1) {
                    int nb[3] = {idx[0], idx[1], idx[2]};
                    nb[dim] -= 1;
                    if ((rc = fiss_generate(&f, nb[0], nb[1], nb[2], &is_new, &cost))) goto done;
                    if (is_new && cost <= cost_center) frontier[fidx(&f, nb[0], nb[1], nb[2])] = 1;
                }
                if (idx[dim] < sizes[dim] - 1) {
                    int nb[3] = {idx[0], idx[1], idx[2]};
                    nb[dim] += 1;
                    if ((rc = fiss_generate(&f, nb[0], nb[1], nb[2], &is_new, &cost))) goto done;
                    if (is_new && cost <= cost_center) frontier[fidx(&f, nb[0], nb[1], nb[2])] = 1;
                }
            }
            /* frontier_idxs.get(): lowest cost (tie -> lower raster index, see fiss_queue_min) */
            int nq = -1;
            for (int r = 0; r < f.C; ++r)
                if (frontier[r] && (nq < 0 || f.cost[r] < f.cost[nq])) nq = r;
            if (nq < 0) break;
            frontier[nq] = 0;
            q = nq;
        }
        q = fiss_queue_min(&f);
        if (q < 0) break;
        uint32_t fl;
        int v = fiss_validate(&f, q, &fl);
        if (v < 0) { rc = v; goto done; }
        if (v == 1) {
            found = 1;
            best_q = q;
            best_ijk[0] = q / (f.nt * f.nv); best_ijk[1] = (q / f.nt) % f.nv; best_ijk[2] = q % f.nt;
            *best_cost = f.cost[q];
            prev_best_idx[0] = best_ijk[0]; prev_best_idx[1] = best_ijk[1]; prev_best_idx[2] = best_ijk[2]; /* :140 */
        }
    }
    if (found) {
        end_state[0] = p->d_samples[best_ijk[0]];
        end_state[1] = p->v_samples[best_ijk[1]];
        end_state[2] = p->t_samples[best_ijk[2]];
    }
    /* refine_solution (:279-326) with gradient_decent (:207-277); no wall-clock limit */
    if (found && max_refine_iters > 0) {
        double res[3] = {p->samp_res[0], p->samp_res[1], p->samp_res[2]};
        double x[3] = {end_state[0], end_state[1], end_state[2]};
        double coarse_cost = f.cost[best_q];
        int nrt = 0, ok = 1;
        for (int it = 0; it < max_refine_iters && ok; ++it) {
            double dJ[3], dx[3];
            for (int dim = 0; dim < 3 && ok; ++dim) {
                double xl[3] = {x[0], x[1], x[2]}, xr[3] = {x[0], x[1], x[2]};
                xl[dim] -= res[dim];
                xr[dim] += res[dim];
                for (int m = 0; m < 3; ++m) {
                    xl[m] = fmin(fmax(xl[m], p->samp_min[m]), p->samp_max[m]);
                    xr[m] = fmin(fmax(xr[m], p->samp_min[m]), p->samp_max[m]);
                }
                double Jl, Jr;
                if (orc_eval_traj(p, xl[0], xl[1], xl[2], 0, NULL, 0, NULL, NULL, &Jl, NULL)) { ok = 0; break; }
                stats[1] += 1;
                rt[nrt] = (rtraj){{xl[0], xl[1], xl[2]}, Jl, nrt, 1}; ++nrt;
                if (orc_eval_traj(p, xr[0], xr[1], xr[2], 0, NULL, 0, NULL, NULL, &Jr, NULL)) { ok = 0; break; }
                stats[1] += 1;
                rt[nrt] = (rtraj){{xr[0], xr[1], xr[2]}, Jr, nrt, 1}; ++nrt;
                dJ[dim] = Jr - Jl;
                dx[dim] = xr[dim] - xl[dim];
            }
            if (!ok) break;
            double g[3], nrm = 0;
            for (int m = 0; m < 3; ++m) { g[m] = dJ[m] / dx[m]; nrm += g[m] * g[m]; }
            nrm = sqrt(nrm);
            double xn[3];
            for (int m = 0; m < 3; ++m) {
                res[m] *= decaying_factor;
                xn[m] = x[m] - res[m] * g[m] / nrm;
                xn[m] = fmin(fmax(xn[m], p->samp_min[m]), p->samp_max[m]);
            }
            if (!(xn[0] == xn[0]) || !(xn[1] == xn[1]) || !(xn[2] == xn[2])) break; /* reference raises in np.arange */
            double Jn;
            if (orc_eval_traj(p, xn[0], xn[1], xn[2], 0, NULL, 0, NULL, NULL, &Jn, NULL)) break;
            stats[1] += 1;
            rt[nrt] = (rtraj){{xn[0], xn[1], xn[2]}, Jn, nrt, 1}; ++nrt;
            x[0] = xn[0]; x[1] = xn[1]; x[2] = xn[2];
        }
        if (trace)
            for (int r = 0; r < nrt; ++r) {
                trace[r * 4 + 0] = rt[r].x[0]; trace[r * 4 + 1] = rt[r].x[1]; trace[r * 4 + 2] = rt[r].x[2];
                trace[r * 4 + 3] = rt[r].cost;
            }
        /* pop refined_trajs in cost order (:301-323); equal costs are equal trajectories or order-free */
        for (;;) {
            int b = -1;
            for (int r = 0; r < nrt; ++r)
                if (rt[r].alive && (b < 0 || rt[r].cost < rt[b].cost)) b = r;
            if (b < 0) break;
            rt[b].alive = 0;
            if (rt[b].cost > coarse_cost) break;
            stats[2] += 1;
            uint32_t fl;
            double c;
            orc_eval_traj(p, rt[b].x[0], rt[b].x[1], rt[b].x[2], 0, NULL, 0, NULL, NULL, &c, &fl);
            if (fl & ORC_FLAG_CONSTRAINTS) continue;
            orc_eval_traj(p, rt[b].x[0], rt[b].x[1], rt[b].x[2], 1, NULL, 0, NULL, NULL, &c, &fl);
            stats[3] += 1;
            if (!(fl & ORC_FLAG_COLLISION)) {
                *refined = 1;
                *best_cost = rt[b].cost;
                end_state[0] = rt[b].x[0]; end_state[1] = rt[b].x[1]; end_state[2] = rt[b].x[2];
                break;
            }
        }
    }
done:
    free(frontier);
    free(rt);
    fiss_free(&f);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* FrenetState.from_state, frenet.py:32-99                                   */
/* ------------------------------------------------------------------------ */
static double unify_angle(double a)
{
    /* math_utils.py:28-34 */
    while (a > ORC_PI) a -= 2 * ORC_PI;
    while (a < -ORC_PI) a += 2 * ORC_PI;
    return a;
}

/* ---- goal_region.is_reached(state) (planners/benchmark/planning.py:150-153) --------------------------------------------------
   commonroad-io is not a dependency of this repo (and absent from the image: PARITY UNPINNED against the real package); restated
   from its published GoalRegion.is_reached / Shape.contains_point: a state reaches a goal state when its position lies in the goal
   shape - shapely `polygon.intersects(Point)`, i.e. inside OR on the boundary - and every attribute the goal state defines
   (time_step, velocity, orientation: closed intervals) contains the state's value.  Decided exactly: binary128 orientation signs
   (orient_sign above) and the winding number, an algorithm independent of the kernels' crossing-number test. */
static int on_segment_exact(const double* a, const double* b, const double* q)
{
    if (orient_sign(a, b, q) != 0) return 0;
    return q[0] >= fmin(a[0], b[0]) && q[0] <= fmax(a[0], b[0]) && q[1] >= fmin(a[1], b[1]) && q[1] <= fmax(a[1], b[1]);
}

int orc_point_in_polygon_closed(const double* poly, int32_t nv, double x, double y)
{
    if (nv < 3 || !(x == x) || !(y == y)) return 0;
    const double q[2] = {x, y};
    int wn = 0;
    for (int i = 0; i < nv; ++i) {
        const double* a = poly + 2 * i;
        const double* b = poly + 2 * ((i + 1) % nv);
        if (on_segment_exact(a, b, q)) return 1;
        if (a[1] <= y) {
            if (b[1] > y && orient_sign(a, b, q) > 0) ++wn;   /* upward crossing, q strictly left */
        } else {
            if (b[1] <= y && orient_sign(a, b, q) < 0) --wn;  /* downward crossing, q strictly right */
        }
    }
    return wn != 0;
}

static int in_goal_interval(double v, double lo, double hi) { return (lo != lo || hi != hi) ? 1 : (v >= lo && v <= hi); }

/* intervals (may be NULL): time_step lo, hi, velocity lo, hi, orientation lo, hi; NaN = not defined by the goal state */
int orc_goal_reached(const double* poly, int32_t nv, const double* intervals, double x, double y, int32_t time_step, double velocity,
                     double orientation)
{
    if (intervals && !(in_goal_interval((double)time_step, intervals[0], intervals[1]) && in_goal_interval(velocity, intervals[2], intervals[3]) &&
                       in_goal_interval(orientation, intervals[4], intervals[5])))
        return 0;
    return orc_point_in_polygon_closed(poly, nv, x, y);
}

int orc_from_state(const double* state, int32_t n, const double* pl, int32_t ld, double* out)
{
    double sx = state[0], sy = state[1], syaw = state[2], sv = state[3];
    /* find_nearest_point_idx (:34-36): np.argmin -> first minimum */
    int nearest = 0;
    double best = INFINITY;
    for (int i = 0; i < n; ++i) {
        double dd = hypot(pl[(size_t)i * ld] - sx, pl[(size_t)i * ld + 1] - sy);
        if (dd < best) { best = dd; nearest = i; }
    }
    /* find_next_point_idx (:38-56) */
    double heading = atan2(pl[(size_t)nearest * ld + 1] - sy, pl[(size_t)nearest * ld] - sx);
    double angle = fabs(syaw - heading);
    angle = fmin(2 * ORC_PI - angle, angle);
    int next = angle > ORC_PI / 2 ? nearest + 1 : nearest;
    if (next < 1) next = 1;
    else if (next >= n) next = n - 1;
    int prev = next - 1 > 0 ? next - 1 : 0;
    double n_x = pl[(size_t)next * ld] - pl[(size_t)prev * ld], n_y = pl[(size_t)next * ld + 1] - pl[(size_t)prev * ld + 1];
    double x_x = sx - pl[(size_t)prev * ld], x_y = sy - pl[(size_t)prev * ld + 1];
    double x_yaw = atan2(x_y, x_x);
    double proj_norm = (x_x * n_x + x_y * n_y) / (n_x * n_x + n_y * n_y);
    double proj_x = proj_norm * n_x, proj_y = proj_norm * n_y;
    double d = hypot(x_x - proj_x, x_y - proj_y);
    double wp_yaw = pl[(size_t)prev * ld + 2];
    double delta_yaw = unify_angle(syaw - wp_yaw);
    if (wp_yaw <= x_yaw) d *= -1; /* :82-83 */
    double s = 0;
    for (int i = 0; i < prev; ++i)
        s += hypot(pl[(size_t)(i + 1) * ld] - pl[(size_t)i * ld], pl[(size_t)(i + 1) * ld + 1] - pl[(size_t)i * ld + 1]);
    out[0] = s;
    out[1] = sv * cos(delta_yaw);
    out[2] = 0.0;
    out[3] = d;
    out[4] = sv * sin(delta_yaw);
    out[5] = 0.0;
    return 0;
}

/* ------------------------------------------------------------------------ */
int orc_fop_plan_batch(const orc_problem* probs, int32_t B, int32_t threads, int32_t* best_idx, double* best_cost)
{
    int err = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < B; ++b) {
        int32_t st[4];
        int rc = orc_fop_plan(&probs[b], &best_idx[b], &best_cost[b], st, NULL, NULL);
        if (rc) err = rc;
    }
    (void)threads;
    return err;
}
