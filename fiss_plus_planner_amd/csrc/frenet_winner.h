// frenet_winner.h - the series of ONE trajectory (what plan() returns), written by one wavefront (two adjacent time points per lane).
// Shared by winner_traj_kernel (standalone epilogue / materialise mode), by lattice_fused_kernel, which appends the epilogue
// of its own argmin when the caller asked for it (no second launch, no re-staging; the stores hide behind the other workgroups'
// arithmetic), and by fiss_refine_kernel.
#pragma once
#include "frenet_device.h"
#include "frenet_kernels.h"

namespace fp {

// One WAVEFRONT writes the series of one trajectory: lane l owns points 2l and 2l + 1 (N <= FP_FAST_POINTS = 128), so a row goes out
// as 16-byte stores, 1 KB per instruction.  No LDS scratch and no workgroup barrier: of the neighbour elements the difference chains
// need (x[i+1], yaw[i+1], c[i+1], c_d[i+1]) one is the lane's own, the other the next lane's.  `valid`, d_end, v_end, T must be
// wave-uniform.  sp may point at global memory or at an LDS copy of the spline.
// Restates calc_global_paths' per-trajectory part (frenet_optimal_planner.py:106-138): yaw / ds / c / c_d / c_dd exactly as the
// np.arctan2 / hypot / diff chains (:121-134), truncation at the first point off the spline (:112-113).
// Output layout: fp_result.traj_stride / traj_sparse (include/frenet_gpu.h).
__device__ __forceinline__ void winner_series_wave(const KernelArgs& ka, int b, int slot, bool valid, double d_end, double v_end, double T, int lane,
                                                   const SplineLds& sp, const double* ego_row = nullptr)
{
    typedef double double2v __attribute__((ext_vector_type(2)));
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    const double nan = __builtin_nan("");
    const int stride = ka.r.traj_stride > 0 ? ka.r.traj_stride : FP_DEFAULT_STRIDE;
    const bool sparse = ka.r.traj_sparse != 0;
    double* out = ka.r.best_traj + (size_t)slot * FP_ARR_COUNT * stride;
    const bool pairs = (stride & 1) == 0 && ((uintptr_t)out & 15) == 0;  // (wave-uniform) every row starts on a 16-byte boundary
    const int N = (T == T) ? arange_len(T, p.tick_t) : 0;
    // elements 2l, 2l + 1 of a row that holds `len` elements: the value, NaN padding (whole block, or up to the end of the 128-byte
    // line in the sparse layout: partial-line stores cost a read-modify-write at the memory side), or nothing.  Write-once stream:
    // non-temporal stores (plain ones are 1.6x slower in the materialise mode).
#if defined(FP_ABL_MAT_NO_STORE)  // timing ablation (tools/mat_variants.sh): the arithmetic without the stream of stores
    double abl_sum = 0.0;
#endif
    auto put = [&](int r, double v0, double v1, int len) {
        const int i = 2 * lane;
#if defined(FP_ABL_MAT_NO_STORE)
        abl_sum += (i < len ? v0 : 0.0) + (i + 1 < len ? v1 : 0.0);
        if (r != FP_ARR_C_DD) return;
        v0 = v1 = abl_sum;
        len = 2 * lane == 0 ? 2 : 0;
        if (lane != 0) return;
#endif
        const int upto = sparse ? ((len + 15) & ~15) : FP_FAST_POINTS;
        const int lim = stride < upto ? stride : upto;
        const double a = i < len ? v0 : nan, c = i + 1 < len ? v1 : nan;
        double* dst = &out[r * stride + i];
        if (pairs) {  // (lim is even then)
            if (i < lim) __builtin_nontemporal_store(double2v{a, c}, (double2v*)dst);
        } else {
            if (i < lim) __builtin_nontemporal_store(a, dst);
            if (i + 1 < lim) __builtin_nontemporal_store(c, dst + 1);
        }
        // dense layout with rows wider than the 128 points a lane pair covers: the rest of the row is padding too (every element of
        // the [16][traj_stride] block is written, as the header promises)
        if (!sparse && stride > FP_FAST_POINTS)
            for (int k = FP_FAST_POINTS + lane; k < stride; k += kWave) __builtin_nontemporal_store(nan, &out[r * stride + k]);
    };
    if (!valid || N <= 0 || N > FP_FAST_POINTS || !(d_end == d_end) || !(v_end == v_end)) {  // wave-uniform
#pragma unroll
        for (int r = 0; r < FP_ARR_COUNT; ++r) put(r, nan, nan, 0);
        if (lane == 0 && ka.r.best_flags) ka.r.best_flags[slot] = 0u;
        return;
    }
#if defined(FP_ABL_MAT_STORE_ONLY)  // timing ablation: the stream of stores without the arithmetic
    {
        const int Ms = N - (slot & 3);
#pragma unroll
        for (int r = 0; r < FP_ARR_COUNT; ++r) put(r, 1.0, 2.0, r < FP_ARR_X ? N : (r <= FP_ARR_YAW ? Ms : Ms - 1));
        if (lane == 0 && ka.r.best_flags) ka.r.best_flags[slot] = ((uint32_t)N << FP_FLAG_N_SHIFT) | ((uint32_t)Ms << FP_FLAG_M_SHIFT);
        return;
    }
#endif
    const double* eg = ego_row ? ego_row : bt.ego + (size_t)b * 6;  // (ego_row: the caller resolved the ego's state itself, InlineIn)
    const Quintic lat = quintic_bvp(eg[3], eg[4], eg[5], d_end, 0.0, 0.0, T);
    const Quartic lon = quartic_bvp(eg[0], eg[1], eg[2], v_end, 0.0, T);
    double x[2] = {nan, nan}, y[2] = {nan, nan};
    double t[2], s[2], s_d[2], s_dd[2], s_ddd[2], d[2], d_d[2], d_dd[2], d_ddd[2];
    unsigned long long off_mask[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = 2 * lane + h;
        bool off = false;
        t[h] = s[h] = s_d[h] = s_dd[h] = s_ddd[h] = d[h] = d_d[h] = d_dd[h] = d_ddd[h] = nan;
        if (i < N) {
            t[h] = (double)i * p.tick_t;
            quartic_eval(lon, t[h], s[h], s_d[h], s_dd[h], s_ddd[h]);
            quintic_eval(lat, t[h], d[h], d_d[h], d_dd[h], d_ddd[h]);
            const int seg = spline_segment(sp, s[h], -1);
            off = seg < 0;  // first point off the spline truncates the Cartesian series (:112-113)
            if (!off) {
                double px, py, tx, ty;
                spline_frame(sp, seg, s[h] - sp.knots[seg], px, py, tx, ty);
                frenet_to_cartesian(px, py, tx, ty, d[h], x[h], y[h]);
            }
        }
        off_mask[h] = __ballot(off);
    }
    put(FP_ARR_T, t[0], t[1], N);
    put(FP_ARR_S, s[0], s[1], N); put(FP_ARR_S_D, s_d[0], s_d[1], N); put(FP_ARR_S_DD, s_dd[0], s_dd[1], N); put(FP_ARR_S_DDD, s_ddd[0], s_ddd[1], N);
    put(FP_ARR_D, d[0], d[1], N); put(FP_ARR_D_D, d_d[0], d_d[1], N); put(FP_ARR_D_DD, d_dd[0], d_dd[1], N); put(FP_ARR_D_DDD, d_ddd[0], d_ddd[1], N);
    // first point off the spline (points 2l of the lanes in off_mask[0], points 2l + 1 in off_mask[1])
    int M = N;
    if (off_mask[0]) M = 2 * (__ffsll((long long)off_mask[0]) - 1);
    if (off_mask[1]) { const int m1 = 2 * (__ffsll((long long)off_mask[1]) - 1) + 1; M = m1 < M ? m1 : M; }
    // element (2 lane + h) + 1 / - 1 of a chain held as two values per lane (the last lane's "next" and the first lane's "previous"
    // are never used: they would be elements 128 and -1)
    auto next = [&](const double* v, int h) { return h == 0 ? v[1] : __shfl_down(v[0], 1, kWave); };
    auto prev = [&](const double* v, int h) { return h == 1 ? v[0] : __shfl_up(v[1], 1, kWave); };
    double yaw[2], ds[2], c[2], cd[2], cdd[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double ddx = next(x, h) - x[h], ddy = next(y, h) - y[h];
#if defined(FP_ABL_MAT_NO_MATH)  // timing ablation: no atan2 / hypot
        yaw[h] = ddy + ddx;
        ds[h] = ddx * ddy;
#else
        yaw[h] = atan2(ddy, ddx);
        ds[h] = hypot(ddx, ddy);
#endif
    }
    {   // the last point repeats the previous heading (:129)
        const double p0 = prev(yaw, 0), p1 = prev(yaw, 1);
        if (2 * lane == M - 1) yaw[0] = p0;
        if (2 * lane + 1 == M - 1) yaw[1] = p1;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) c[h] = (next(yaw, h) - yaw[h]) / ds[h];
#pragma unroll
    for (int h = 0; h < 2; ++h) cd[h] = (next(c, h) - c[h]) / p.tick_t;
#pragma unroll
    for (int h = 0; h < 2; ++h) cdd[h] = (next(cd, h) - cd[h]) / p.tick_t;
    const int My = M >= 2 ? M : 0;  // x, y keep their M points; the difference chains need two
    put(FP_ARR_X, x[0], x[1], M); put(FP_ARR_Y, y[0], y[1], M);
    put(FP_ARR_YAW, yaw[0], yaw[1], My); put(FP_ARR_DS, ds[0], ds[1], My - 1); put(FP_ARR_C, c[0], c[1], My - 1);
    put(FP_ARR_C_D, cd[0], cd[1], My - 2); put(FP_ARR_C_DD, cdd[0], cdd[1], My - 3);
    if (lane == 0 && ka.r.best_flags) {
        uint32_t fl = ((uint32_t)N << FP_FLAG_N_SHIFT) | ((uint32_t)M << FP_FLAG_M_SHIFT);
        if (M < N) fl |= FP_FLAG_TRUNCATED;
        ka.r.best_flags[slot] = fl;
    }
}

// Trajectories of MORE than kSeriesChunk = 128 points (tick_t below 0.08 s at T = 10 s: FP_FAST_POINTS is 256) by the same wavefront,
// in chunks: chunk c writes elements [120 c, 120 c + 120) of every row and evaluates the points [120 c - 2, 120 c + 126) for them -
// the difference chains of an element reach four points ahead (c_dd[i] needs x[i + 4]) and the repeated last heading one point back
// (yaw[M - 1] = yaw[M - 2], :129), so six points of halo ahead and two behind make every element of the window what the one-chunk
// body computes for it (same arithmetic per element).  M - the first point off the spline - is found by the chunk that holds it; the
// chunks before it never look at elements that depend on it.  winner_traj_kernel calls this for N > kSeriesChunk only: the one-chunk
// body above stays what the fused kernel, the refinement kernel and every N <= 128 trajectory run.
constexpr int kSeriesChunk = FP_FAST_POINTS, kSeriesStep = 120;
__device__ __forceinline__ void winner_series_wave_long(const KernelArgs& ka, int b, int slot, double d_end, double v_end, double T, int lane, const SplineLds& sp)
{
    typedef double double2v __attribute__((ext_vector_type(2)));
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    const double nan = __builtin_nan("");
    const int stride = ka.r.traj_stride > 0 ? ka.r.traj_stride : FP_DEFAULT_STRIDE;
    const bool sparse = ka.r.traj_sparse != 0;
    double* out = ka.r.best_traj + (size_t)slot * FP_ARR_COUNT * stride;
    const bool pairs = (stride & 1) == 0 && ((uintptr_t)out & 15) == 0;
    const int N = arange_len(T, p.tick_t);  // (the caller checked: kSeriesChunk < N <= FP_MAX_POINTS, finite end state)
    const double* eg = bt.ego + (size_t)b * 6;
    const Quintic lat = quintic_bvp(eg[3], eg[4], eg[5], d_end, 0.0, 0.0, T);
    const Quartic lon = quartic_bvp(eg[0], eg[1], eg[2], v_end, 0.0, T);
    int M = N;
    const int cover = sparse ? ((N + 15) & ~15) : stride;  // elements of a row some chunk has to write (values, then NaN padding)
    for (int w0 = 0; w0 < cover && w0 < stride; w0 += kSeriesStep) {
        const int e0 = w0 == 0 ? 0 : w0 - 2, w1 = w0 + kSeriesStep;  // evaluated points [e0, e0 + 128), written elements [w0, w1)
        auto put = [&](int r, double v0, double v1, int len) {
            const int i = e0 + 2 * lane;
            const int upto = sparse ? ((len + 15) & ~15) : stride;
            int lim = stride < upto ? stride : upto;
            lim = lim < w1 ? lim : w1;
            const double a = i < len ? v0 : nan, c = i + 1 < len ? v1 : nan;
            double* dst = &out[r * stride + i];
            if (i < w0) return;
            if (pairs) {
                if (i < lim) __builtin_nontemporal_store(double2v{a, c}, (double2v*)dst);
            } else {
                if (i < lim) __builtin_nontemporal_store(a, dst);
                if (i + 1 < lim) __builtin_nontemporal_store(c, dst + 1);
            }
        };
        double x[2] = {nan, nan}, y[2] = {nan, nan};
        double t[2], s[2], s_d[2], s_dd[2], s_ddd[2], d[2], d_d[2], d_dd[2], d_ddd[2];
        unsigned long long off_mask[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = e0 + 2 * lane + h;
            bool off = false;
            t[h] = s[h] = s_d[h] = s_dd[h] = s_ddd[h] = d[h] = d_d[h] = d_dd[h] = d_ddd[h] = nan;
            if (i < N) {
                t[h] = (double)i * p.tick_t;
                quartic_eval(lon, t[h], s[h], s_d[h], s_dd[h], s_ddd[h]);
                quintic_eval(lat, t[h], d[h], d_d[h], d_dd[h], d_ddd[h]);
                const int seg = i < M ? spline_segment(sp, s[h], -1) : -1;  // (beyond an earlier chunk's M nothing Cartesian exists)
                off = seg < 0;
                if (!off) {
                    double px, py, tx, ty;
                    spline_frame(sp, seg, s[h] - sp.knots[seg], px, py, tx, ty);
                    frenet_to_cartesian(px, py, tx, ty, d[h], x[h], y[h]);
                }
            }
            off_mask[h] = __ballot(off);
        }
        put(FP_ARR_T, t[0], t[1], N);
        put(FP_ARR_S, s[0], s[1], N); put(FP_ARR_S_D, s_d[0], s_d[1], N); put(FP_ARR_S_DD, s_dd[0], s_dd[1], N); put(FP_ARR_S_DDD, s_ddd[0], s_ddd[1], N);
        put(FP_ARR_D, d[0], d[1], N); put(FP_ARR_D_D, d_d[0], d_d[1], N); put(FP_ARR_D_DD, d_dd[0], d_dd[1], N); put(FP_ARR_D_DDD, d_ddd[0], d_ddd[1], N);
        if (off_mask[0]) { const int m0 = e0 + 2 * (__ffsll((long long)off_mask[0]) - 1); M = m0 < M ? m0 : M; }
        if (off_mask[1]) { const int m1 = e0 + 2 * (__ffsll((long long)off_mask[1]) - 1) + 1; M = m1 < M ? m1 : M; }
        auto next = [&](const double* v, int h) { return h == 0 ? v[1] : __shfl_down(v[0], 1, kWave); };
        auto prev = [&](const double* v, int h) { return h == 1 ? v[0] : __shfl_up(v[1], 1, kWave); };
        double yaw[2], ds[2], c[2], cd[2], cdd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double ddx = next(x, h) - x[h], ddy = next(y, h) - y[h];
            yaw[h] = atan2(ddy, ddx);
            ds[h] = hypot(ddx, ddy);
        }
        {   // the last point repeats the previous heading (:129)
            const double p0 = prev(yaw, 0), p1 = prev(yaw, 1);
            if (e0 + 2 * lane == M - 1) yaw[0] = p0;
            if (e0 + 2 * lane + 1 == M - 1) yaw[1] = p1;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) c[h] = (next(yaw, h) - yaw[h]) / ds[h];
#pragma unroll
        for (int h = 0; h < 2; ++h) cd[h] = (next(c, h) - c[h]) / p.tick_t;
#pragma unroll
        for (int h = 0; h < 2; ++h) cdd[h] = (next(cd, h) - cd[h]) / p.tick_t;
        const int My = M >= 2 ? M : 0;
        put(FP_ARR_X, x[0], x[1], M); put(FP_ARR_Y, y[0], y[1], M);
        put(FP_ARR_YAW, yaw[0], yaw[1], My); put(FP_ARR_DS, ds[0], ds[1], My - 1); put(FP_ARR_C, c[0], c[1], My - 1);
        put(FP_ARR_C_D, cd[0], cd[1], My - 2); put(FP_ARR_C_DD, cdd[0], cdd[1], My - 3);
    }
    if (lane == 0 && ka.r.best_flags) {
        uint32_t fl = ((uint32_t)N << FP_FLAG_N_SHIFT) | ((uint32_t)M << FP_FLAG_M_SHIFT);
        if (M < N) fl |= FP_FLAG_TRUNCATED;
        ka.r.best_flags[slot] = fl;
    }
}

// The same series written by TWO wavefronts, one time point per lane (point i = 64 * (wavefront of the pair) + lane): half the live values
// of winner_series_wave (its two-points-per-lane body needs ~125 VGPRs; this one fits the 80 of the three-workgroups-per-CU lattice
// instances, whose appended epilogue workgroups run it - see lattice_fused_kernel).  The neighbour elements of the difference chains
// travel through `scratch` (LDS, 5 x FP_FAST_POINTS doubles of this trajectory) between workgroup barriers: EVERY thread of the
// workgroup must call this function (the same number of barriers), whatever its trajectory.  Element for element the arithmetic of
// winner_series_wave (bit-identical output).  write = false: no stores at all (the thread only keeps the barriers' count).
__device__ __forceinline__ void winner_series_pair(const KernelArgs& ka, int b, int slot, bool valid, double d_end, double v_end, double T, int i,
                                                   const SplineLds& sp, double* scratch, int* m_scratch, bool write = true)
{
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    const double nan = __builtin_nan("");
    const int stride = ka.r.traj_stride > 0 ? ka.r.traj_stride : FP_DEFAULT_STRIDE;
    const bool sparse = ka.r.traj_sparse != 0;
    double* out = ka.r.best_traj + (size_t)slot * FP_ARR_COUNT * stride;
    const int N = (T == T) ? arange_len(T, p.tick_t) : 0;
    const bool ok = valid && N > 0 && N <= FP_FAST_POINTS && (d_end == d_end) && (v_end == v_end);  // uniform over the pair
    auto put = [&](int r, double v, int len) {  // element i of a row that holds `len` elements (see winner_series_wave)
        const int upto = sparse ? ((len + 15) & ~15) : FP_FAST_POINTS;
        const int lim = stride < upto ? stride : upto;
        if (!write) return;  // (a thread without a trajectory only takes part in the barriers)
        if (i < lim) __builtin_nontemporal_store(i < len ? v : nan, &out[r * stride + i]);
        if (!sparse && stride > FP_FAST_POINTS)
            for (int k = FP_FAST_POINTS + i; k < stride; k += 2 * kWave) __builtin_nontemporal_store(nan, &out[r * stride + k]);
    };
    double* xs = scratch;
    double* ys = scratch + FP_FAST_POINTS;
    double* ws = scratch + 2 * FP_FAST_POINTS;  // yaw, then c_d
    double* cs = scratch + 3 * FP_FAST_POINTS;  // c
    double x = nan, y = nan;
    bool off = false;
    if (ok) {
        const double* eg = bt.ego + (size_t)b * 6;
        double t = nan, s = nan, s_d = nan, s_dd = nan, s_ddd = nan, d = nan, d_d = nan, d_dd = nan, d_ddd = nan;
        if (i < N) {
            const Quintic lat = quintic_bvp(eg[3], eg[4], eg[5], d_end, 0.0, 0.0, T);
            const Quartic lon = quartic_bvp(eg[0], eg[1], eg[2], v_end, 0.0, T);
            t = (double)i * p.tick_t;
            quartic_eval(lon, t, s, s_d, s_dd, s_ddd);
            quintic_eval(lat, t, d, d_d, d_dd, d_ddd);
            const int seg = spline_segment(sp, s, -1);
            off = seg < 0;  // first point off the spline truncates the Cartesian series (:112-113)
            if (!off) {
                double px, py, tx, ty;
                spline_frame(sp, seg, s - sp.knots[seg], px, py, tx, ty);
                frenet_to_cartesian(px, py, tx, ty, d, x, y);
            }
        }
        put(FP_ARR_T, t, N);
        put(FP_ARR_S, s, N); put(FP_ARR_S_D, s_d, N); put(FP_ARR_S_DD, s_dd, N); put(FP_ARR_S_DDD, s_ddd, N);
        put(FP_ARR_D, d, N); put(FP_ARR_D_D, d_d, N); put(FP_ARR_D_DD, d_dd, N); put(FP_ARR_D_DDD, d_ddd, N);
    } else {  // no trajectory: NaN rows in the dense layout, nothing in the sparse one
#pragma unroll
        for (int r = 0; r < FP_ARR_COUNT; ++r) put(r, nan, 0);
    }
    xs[i] = x; ys[i] = y;
    {   // first point off the spline over both wavefronts of the pair
        const unsigned long long m = __ballot(off);
        if ((i & (kWave - 1)) == 0) m_scratch[i >> 6] = m ? (i & ~(kWave - 1)) + __ffsll((long long)m) - 1 : FP_FAST_POINTS;
    }
    __syncthreads();
    int M = N;
    {
        const int m0 = m_scratch[0], m1 = m_scratch[1];
        const int mf = m0 < m1 ? m0 : m1;
        M = mf < M ? mf : M;
    }
    const bool last = i + 1 >= FP_FAST_POINTS;  // (element 128 does not exist: the value is never used)
    const double xn = last ? nan : xs[i + 1], yn = last ? nan : ys[i + 1];
    const double ddx = xn - x, ddy = yn - y;
    const double yaw_raw = atan2(ddy, ddx), ds = hypot(ddx, ddy);
    ws[i] = yaw_raw;
    __syncthreads();
    // the last point repeats the previous heading (:129): yaw[M - 1] = yaw[M - 2]
    const double yaw = (i == M - 1 && i >= 1) ? ws[i - 1] : yaw_raw;
    const double yaw_next = last ? nan : ((i + 1 == M - 1) ? yaw_raw : ws[i + 1]);
    const double c = (yaw_next - yaw) / ds;
    cs[i] = c;
    __syncthreads();
    const double cd = ((last ? nan : cs[i + 1]) - c) / p.tick_t;
    ws[i] = cd;  // (every thread is past its reads of the headings: they happened before the barrier above)
    __syncthreads();
    const double cdd = ((last ? nan : ws[i + 1]) - cd) / p.tick_t;
    if (ok) {
        const int My = M >= 2 ? M : 0;  // x, y keep their M points; the difference chains need two
        put(FP_ARR_X, x, M); put(FP_ARR_Y, y, M);
        put(FP_ARR_YAW, yaw, My); put(FP_ARR_DS, ds, My - 1); put(FP_ARR_C, c, My - 1);
        put(FP_ARR_C_D, cd, My - 2); put(FP_ARR_C_DD, cdd, My - 3);
    }
    if (i == 0 && ka.r.best_flags && write) {
        uint32_t fl = 0u;
        if (ok) {
            fl = ((uint32_t)N << FP_FLAG_N_SHIFT) | ((uint32_t)M << FP_FLAG_M_SHIFT);
            if (M < N) fl |= FP_FLAG_TRUNCATED;
        }
        ka.r.best_flags[slot] = fl;
    }
}

// Materialise mode (fp_materialize_all): one WAVEFRONT writes the series of ALL nd lattice candidates that share one longitudinal
// profile (i_T, i_v).  Of a candidate's series everything but the lateral polynomial belongs to the profile: t, s and its
// derivatives, the spline segment of every point, the reference-line frame (position + unit tangent) and the truncation index M -
// computed once and kept in registers (two adjacent points per lane, as in winner_series_wave), then every lateral sample adds its
// own quintic, offsets the frames and runs the yaw / ds / curvature difference chains.  Same arithmetic per element as
// winner_series_wave (the tests compare both with the CPU restatement), a ninth of the segment searches / frame evaluations / input reads
// per candidate in a 9-wide lattice.  slot0 = the block of lateral sample 0, slot_step = blocks between two lateral samples.
__device__ __forceinline__ void profile_series_wave(const KernelArgs& ka, int b, size_t slot0, size_t slot_step, int n_lat, const double* d_ends,
                                                    double v_end, double T, int lane, const SplineLds& sp)
{
    typedef double double2v __attribute__((ext_vector_type(2)));
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    const double nan = __builtin_nan("");
    const int stride = ka.r.traj_stride > 0 ? ka.r.traj_stride : FP_DEFAULT_STRIDE;
    const bool sparse = ka.r.traj_sparse != 0;
    const bool pairs = (stride & 1) == 0 && ((uintptr_t)ka.r.best_traj & 15) == 0;  // every row of every block starts on a 16-byte boundary
    const int N = (T == T) ? arange_len(T, p.tick_t) : 0;
    double* out = nullptr;
#if defined(FP_ABL_MAT_NO_STORE)
    double abl_sum = 0.0;
#endif
    auto put = [&](int r, double v0, double v1, int len) {  // (see winner_series_wave)
        const int i = 2 * lane;
#if defined(FP_ABL_MAT_NO_STORE)
        abl_sum += (i < len ? v0 : 0.0) + (i + 1 < len ? v1 : 0.0);
        if (r != FP_ARR_C_DD) return;
        v0 = v1 = abl_sum;
        len = 2 * lane == 0 ? 2 : 0;
        if (lane != 0) return;
#endif
        const int upto = sparse ? ((len + 15) & ~15) : FP_FAST_POINTS;
        const int lim = stride < upto ? stride : upto;
        const double a = i < len ? v0 : nan, c = i + 1 < len ? v1 : nan;
        double* dst = &out[r * stride + i];
        if (pairs) {
            if (i < lim) __builtin_nontemporal_store(double2v{a, c}, (double2v*)dst);
        } else {
            if (i < lim) __builtin_nontemporal_store(a, dst);
            if (i + 1 < lim) __builtin_nontemporal_store(c, dst + 1);
        }
        if (!sparse && stride > FP_FAST_POINTS)
            for (int k = FP_FAST_POINTS + lane; k < stride; k += kWave) __builtin_nontemporal_store(nan, &out[r * stride + k]);
    };
    const bool lon_ok = N > 0 && N <= FP_FAST_POINTS && (v_end == v_end);  // wave-uniform
    const double* eg = bt.ego + (size_t)b * 6;
    double t[2], s[2], s_d[2], s_dd[2], s_ddd[2], px[2], py[2], tx[2], ty[2];
    bool on[2] = {false, false};
    int M = N;
    if (lon_ok) {
        const Quartic lon = quartic_bvp(eg[0], eg[1], eg[2], v_end, 0.0, T);
        unsigned long long off_mask[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * lane + h;
            bool off = false;
            t[h] = s[h] = s_d[h] = s_dd[h] = s_ddd[h] = px[h] = py[h] = tx[h] = ty[h] = nan;
            if (i < N) {
                t[h] = (double)i * p.tick_t;
                quartic_eval(lon, t[h], s[h], s_d[h], s_dd[h], s_ddd[h]);
                const int seg = spline_segment(sp, s[h], -1);
                off = seg < 0;  // first point off the spline truncates the Cartesian series (:112-113)
                if (!off) {
                    spline_frame(sp, seg, s[h] - sp.knots[seg], px[h], py[h], tx[h], ty[h]);
                    on[h] = true;
                }
            }
            off_mask[h] = __ballot(off);
        }
        if (off_mask[0]) M = 2 * (__ffsll((long long)off_mask[0]) - 1);
        if (off_mask[1]) { const int m1 = 2 * (__ffsll((long long)off_mask[1]) - 1) + 1; M = m1 < M ? m1 : M; }
    }
    const int My = M >= 2 ? M : 0;
    uint32_t fl = ((uint32_t)N << FP_FLAG_N_SHIFT) | ((uint32_t)M << FP_FLAG_M_SHIFT);
    if (M < N) fl |= FP_FLAG_TRUNCATED;
    auto next = [&](const double* v, int h) { return h == 0 ? v[1] : __shfl_down(v[0], 1, kWave); };
    auto prev = [&](const double* v, int h) { return h == 1 ? v[0] : __shfl_up(v[1], 1, kWave); };
    for (int id = 0; id < n_lat; ++id) {
        const size_t slot = slot0 + (size_t)id * slot_step;
        out = ka.r.best_traj + slot * FP_ARR_COUNT * stride;
        const double d_end = d_ends[id];
        if (!lon_ok || !(d_end == d_end)) {  // wave-uniform
#pragma unroll
            for (int r = 0; r < FP_ARR_COUNT; ++r) put(r, nan, nan, 0);
            if (lane == 0 && ka.r.best_flags) ka.r.best_flags[slot] = 0u;
            continue;
        }
#if defined(FP_ABL_MAT_STORE_ONLY)
        {
#pragma unroll
            for (int r = 0; r < FP_ARR_COUNT; ++r) put(r, 1.0, 2.0, r < FP_ARR_X ? N : (r <= FP_ARR_YAW ? My : My - 1));
            if (lane == 0 && ka.r.best_flags) ka.r.best_flags[slot] = fl;
            continue;
        }
#endif
        const Quintic lat = quintic_bvp(eg[3], eg[4], eg[5], d_end, 0.0, 0.0, T);
        double x[2] = {nan, nan}, y[2] = {nan, nan}, d[2], d_d[2], d_dd[2], d_ddd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            d[h] = d_d[h] = d_dd[h] = d_ddd[h] = nan;
            if (2 * lane + h < N) quintic_eval(lat, t[h], d[h], d_d[h], d_dd[h], d_ddd[h]);
            if (on[h]) frenet_to_cartesian(px[h], py[h], tx[h], ty[h], d[h], x[h], y[h]);
        }
        put(FP_ARR_T, t[0], t[1], N);
        put(FP_ARR_S, s[0], s[1], N); put(FP_ARR_S_D, s_d[0], s_d[1], N); put(FP_ARR_S_DD, s_dd[0], s_dd[1], N); put(FP_ARR_S_DDD, s_ddd[0], s_ddd[1], N);
        put(FP_ARR_D, d[0], d[1], N); put(FP_ARR_D_D, d_d[0], d_d[1], N); put(FP_ARR_D_DD, d_dd[0], d_dd[1], N); put(FP_ARR_D_DDD, d_ddd[0], d_ddd[1], N);
        double yaw[2], ds[2], c[2], cd[2], cdd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double ddx = next(x, h) - x[h], ddy = next(y, h) - y[h];
            yaw[h] = atan2(ddy, ddx);
            ds[h] = hypot(ddx, ddy);
        }
        {   // the last point repeats the previous heading (:129)
            const double p0 = prev(yaw, 0), p1 = prev(yaw, 1);
            if (2 * lane == M - 1) yaw[0] = p0;
            if (2 * lane + 1 == M - 1) yaw[1] = p1;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) c[h] = (next(yaw, h) - yaw[h]) / ds[h];
#pragma unroll
        for (int h = 0; h < 2; ++h) cd[h] = (next(c, h) - c[h]) / p.tick_t;
#pragma unroll
        for (int h = 0; h < 2; ++h) cdd[h] = (next(cd, h) - cd[h]) / p.tick_t;
        put(FP_ARR_X, x[0], x[1], M); put(FP_ARR_Y, y[0], y[1], M);
        put(FP_ARR_YAW, yaw[0], yaw[1], My); put(FP_ARR_DS, ds[0], ds[1], My - 1); put(FP_ARR_C, c[0], c[1], My - 1);
        put(FP_ARR_C_D, cd[0], cd[1], My - 2); put(FP_ARR_C_DD, cdd[0], cdd[1], My - 3);
        if (lane == 0 && ka.r.best_flags) ka.r.best_flags[slot] = fl;
    }
}

}  // namespace fp
