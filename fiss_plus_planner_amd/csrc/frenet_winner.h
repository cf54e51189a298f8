// frenet_winner.h - the series of ONE trajectory (what plan() returns), written by one wavefront (two time points per lane).
// Shared by winner_traj_kernel (standalone epilogue / materialise mode), by lattice_fused_kernel, which appends the epilogue
// of its own argmin when the caller asked for it (no second launch, no re-staging; the stores hide behind the other workgroups'
// arithmetic), and by fiss_refine_kernel.
#pragma once
#include "frenet_device.h"
#include "frenet_kernels.h"

namespace fp {

// One WAVEFRONT writes the series of one trajectory: lane l owns points l and l + 64 (N <= FP_MAX_POINTS = 128).  No LDS scratch and
// no workgroup barrier: the neighbour elements the difference chains need (x[i+1], yaw[i+1], c[i+1], c_d[i+1]) come from the next
// lane.  `valid`, d_end, v_end, T must be wave-uniform.  sp may point at global memory or at an LDS copy of the spline.
// Restates calc_global_paths' per-trajectory part (frenet_optimal_planner.py:106-138): yaw / ds / c / c_d / c_dd exactly as the
// np.arctan2 / hypot / diff chains (:121-134), truncation at the first point off the spline (:112-113).
// Output layout: fp_result.traj_stride / traj_sparse (include/frenet_gpu.h).
__device__ __forceinline__ void winner_series_wave(const KernelArgs& ka, int b, int slot, bool valid, double d_end, double v_end, double T, int lane,
                                                   const SplineLds& sp)
{
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    const double nan = __builtin_nan("");
    const int stride = ka.r.traj_stride > 0 ? ka.r.traj_stride : FP_MAX_POINTS;
    const bool sparse = ka.r.traj_sparse != 0;
    double* out = ka.r.best_traj + (size_t)slot * FP_ARR_COUNT * stride;
    const int N = (T == T) ? arange_len(T, p.tick_t) : 0;
    // element i of a row that holds `len` elements: the value, NaN padding (whole block, or up to the end of the 128-byte line in
    // the sparse layout: partial-line stores cost a read-modify-write at the memory side), or nothing
    auto put = [&](int r, int h, double v, int len) {
        const int i = lane + h * kWave;
        const int upto = sparse ? ((len + 15) & ~15) : FP_MAX_POINTS;
        if (i < stride && i < upto) __builtin_nontemporal_store(i < len ? v : nan, &out[r * stride + i]);  // write-once stream
    };
    if (!valid || N <= 0 || N > FP_MAX_POINTS || !(d_end == d_end) || !(v_end == v_end)) {  // wave-uniform
#pragma unroll
        for (int r = 0; r < FP_ARR_COUNT; ++r) { put(r, 0, nan, 0); put(r, 1, nan, 0); }
        if (lane == 0 && ka.r.best_flags) ka.r.best_flags[slot] = 0u;
        return;
    }
    const double* eg = bt.ego + (size_t)b * 6;
    const Quintic lat = quintic_bvp(eg[3], eg[4], eg[5], d_end, 0.0, 0.0, T);
    const Quartic lon = quartic_bvp(eg[0], eg[1], eg[2], v_end, 0.0, T);
    double x[2] = {nan, nan}, y[2] = {nan, nan};
    unsigned long long off_mask[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = lane + h * kWave;
        bool off = false;
        double t = nan, s = nan, s_d = nan, s_dd = nan, s_ddd = nan, d = nan, d_d = nan, d_dd = nan, d_ddd = nan;
        if (i < N) {
            t = (double)i * p.tick_t;
            quartic_eval(lon, t, s, s_d, s_dd, s_ddd);
            quintic_eval(lat, t, d, d_d, d_dd, d_ddd);
            const int seg = spline_segment(sp, s, -1);
            off = seg < 0;  // first point off the spline truncates the Cartesian series (:112-113)
            if (!off) {
                double px, py, tx, ty;
                spline_frame(sp, seg, s - sp.knots[seg], px, py, tx, ty);
                frenet_to_cartesian(px, py, tx, ty, d, x[h], y[h]);
            }
        }
        off_mask[h] = __ballot(off);
        put(FP_ARR_T, h, t, N);
        put(FP_ARR_S, h, s, N); put(FP_ARR_S_D, h, s_d, N); put(FP_ARR_S_DD, h, s_dd, N); put(FP_ARR_S_DDD, h, s_ddd, N);
        put(FP_ARR_D, h, d, N); put(FP_ARR_D_D, h, d_d, N); put(FP_ARR_D_DD, h, d_dd, N); put(FP_ARR_D_DDD, h, d_ddd, N);
    }
    const int M = off_mask[0] ? __ffsll((long long)off_mask[0]) - 1 : (off_mask[1] ? kWave + __ffsll((long long)off_mask[1]) - 1 : N);
    // element (lane + 64 h) + 1 / - 1 of a chain held as two values per lane
    auto next = [&](const double* v, int h) {
        const double dn = __shfl_down(v[h], 1, kWave);
        return (h == 0 && lane == kWave - 1) ? lane_value(v[1], 0) : dn;
    };
    auto prev = [&](const double* v, int h) {
        const double up = __shfl_up(v[h], 1, kWave);
        return (h == 1 && lane == 0) ? lane_value(v[0], kWave - 1) : up;
    };
    double yaw[2], ds[2], c[2], cd[2], cdd[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double ddx = next(x, h) - x[h], ddy = next(y, h) - y[h];
        yaw[h] = atan2(ddy, ddx);
        ds[h] = hypot(ddx, ddy);
    }
    {   // the last point repeats the previous heading (:129)
        const double p0 = prev(yaw, 0), p1 = prev(yaw, 1);
        if (lane == M - 1) yaw[0] = p0;
        if (lane + kWave == M - 1) yaw[1] = p1;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) c[h] = (next(yaw, h) - yaw[h]) / ds[h];
#pragma unroll
    for (int h = 0; h < 2; ++h) cd[h] = (next(c, h) - c[h]) / p.tick_t;
#pragma unroll
    for (int h = 0; h < 2; ++h) cdd[h] = (next(cd, h) - cd[h]) / p.tick_t;
    const int My = M >= 2 ? M : 0;  // x, y keep their M points; the difference chains need two
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        put(FP_ARR_X, h, x[h], M); put(FP_ARR_Y, h, y[h], M);
        put(FP_ARR_YAW, h, yaw[h], My); put(FP_ARR_DS, h, ds[h], My - 1); put(FP_ARR_C, h, c[h], My - 1);
        put(FP_ARR_C_D, h, cd[h], My - 2); put(FP_ARR_C_DD, h, cdd[h], My - 3);
    }
    if (lane == 0 && ka.r.best_flags) {
        uint32_t fl = ((uint32_t)N << FP_FLAG_N_SHIFT) | ((uint32_t)M << FP_FLAG_M_SHIFT);
        if (M < N) fl |= FP_FLAG_TRUNCATED;
        ka.r.best_flags[slot] = fl;
    }
}

}  // namespace fp
