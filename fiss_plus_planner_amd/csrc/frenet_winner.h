// frenet_winner.h - the series of ONE trajectory (what plan() returns), one lane per time point.
// Shared by winner_traj_kernel (standalone epilogue / materialise mode) and by lattice_fused_kernel, which appends the epilogue
// of its own argmin when the caller asked for it (no second launch, no re-staging; the 16 KB of stores per ego hide behind the
// other workgroups' arithmetic).
#pragma once
#include "frenet_device.h"
#include "frenet_kernels.h"

namespace fp {

constexpr int kWinnerScratchDoubles = 6 * (FP_MAX_POINTS + 1) + 2;

// Every thread of the workgroup must call this (it contains workgroup barriers); threads with i >= FP_MAX_POINTS only take part in
// the barriers.  `valid`, d_end, v_end, T are workgroup-uniform.  sp may point at global memory or at an LDS copy of the spline.
// Restates calc_global_paths' per-trajectory part (frenet_optimal_planner.py:106-138): yaw / ds / c / c_d / c_dd exactly as the
// np.arctan2 / hypot / diff chains (:121-134), truncation at the first point off the spline (:112-113).
__device__ __forceinline__ void winner_series(const KernelArgs& ka, int b, int slot, bool valid, double d_end, double v_end, double T, int i,
                                              const SplineLds& sp, double* scratch)
{
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    double* sx = scratch;
    double* sy = sx + FP_MAX_POINTS + 1;
    double* syaw = sy + FP_MAX_POINTS + 1;
    double* sds = syaw + FP_MAX_POINTS + 1;
    double* sc = sds + FP_MAX_POINTS + 1;
    double* scd = sc + FP_MAX_POINTS + 1;
    int* sM = (int*)(scd + FP_MAX_POINTS + 1);
    const bool worker = i < FP_MAX_POINTS;
    const double nan = __builtin_nan("");
    // output layout (fp_result.traj_stride / traj_sparse): [16][stride] per slot; sparse = only existing elements are written
    const int stride = ka.r.traj_stride > 0 ? ka.r.traj_stride : FP_MAX_POINTS;
    const bool sparse = ka.r.traj_sparse != 0;
    double* out = ka.r.best_traj + (size_t)slot * FP_ARR_COUNT * stride;
    const int N = (T == T) ? arange_len(T, p.tick_t) : 0;
    double row[FP_ARR_COUNT];
#pragma unroll
    for (int r = 0; r < FP_ARR_COUNT; ++r) row[r] = nan;
    if (!valid || N <= 0 || N > FP_MAX_POINTS || !(d_end == d_end) || !(v_end == v_end)) {  // workgroup-uniform
        if (worker && i < stride && !sparse) {
#pragma unroll
            for (int r = 0; r < FP_ARR_COUNT; ++r) out[r * stride + i] = nan;
        }
        if (i == 0 && ka.r.best_flags) ka.r.best_flags[slot] = 0u;
        return;
    }
    const double* eg = bt.ego + (size_t)b * 6;
    const Quintic lat = quintic_bvp(eg[3], eg[4], eg[5], d_end, 0.0, 0.0, T);
    const Quartic lon = quartic_bvp(eg[0], eg[1], eg[2], v_end, 0.0, T);
    if (i == 0) *sM = N;
    __syncthreads();
    bool on = false;
    double x = nan, y = nan;
    if (i < N) {
        const double t = (double)i * p.tick_t;
        row[FP_ARR_T] = t;
        quartic_eval(lon, t, row[FP_ARR_S], row[FP_ARR_S_D], row[FP_ARR_S_DD], row[FP_ARR_S_DDD]);
        quintic_eval(lat, t, row[FP_ARR_D], row[FP_ARR_D_D], row[FP_ARR_D_DD], row[FP_ARR_D_DDD]);
        const int seg = spline_segment(sp, row[FP_ARR_S], -1);
        if (seg < 0) {
            atomicMin(sM, i);  // first point off the spline truncates the Cartesian series (:112-113)
        } else {
            double px, py, tx, ty;
            spline_frame(sp, seg, row[FP_ARR_S] - sp.knots[seg], px, py, tx, ty);
            frenet_to_cartesian(px, py, tx, ty, row[FP_ARR_D], x, y);
            on = true;
        }
    }
    if (worker) { sx[i] = x; sy[i] = y; }
    __syncthreads();
    const int M = *sM;
    on = on && i < M;
    if (on) { row[FP_ARR_X] = x; row[FP_ARR_Y] = y; }
    double yaw = nan, ds = nan;
    if (M >= 2 && i < M - 1) {
        const double ddx = sx[i + 1] - x, ddy = sy[i + 1] - y;
        yaw = atan2(ddy, ddx);
        ds = hypot(ddx, ddy);
    }
    if (worker) { syaw[i] = yaw; sds[i] = ds; }
    __syncthreads();
    if (M >= 2 && i == M - 1) { yaw = syaw[M - 2]; syaw[i] = yaw; }
    __syncthreads();
    double c = nan, c_d = nan, c_dd = nan;
    if (M >= 2 && i < M - 1) c = (syaw[i + 1] - syaw[i]) / sds[i];
    if (worker) sc[i] = c;
    __syncthreads();
    if (M >= 2 && i < M - 2) c_d = (sc[i + 1] - sc[i]) / p.tick_t;
    if (worker) scd[i] = c_d;
    __syncthreads();
    if (M >= 2 && i < M - 3) c_dd = (scd[i + 1] - scd[i]) / p.tick_t;
    if (M >= 2) {
        if (i < M) row[FP_ARR_YAW] = yaw;
        row[FP_ARR_DS] = ds; row[FP_ARR_C] = c; row[FP_ARR_C_D] = c_d; row[FP_ARR_C_DD] = c_dd;
    }
    if (worker && i < stride) {
        if (!sparse) {
#pragma unroll
            for (int r = 0; r < FP_ARR_COUNT; ++r) __builtin_nontemporal_store(row[r], &out[r * stride + i]);  // write-once stream
        } else {
            // row lengths: N (t, s.., d..), M (x, y, yaw), M-1 (ds, c), M-2 (c_d), M-3 (c_dd); M < 2 leaves only x / y of length M
            const int Mx = M, My = M >= 2 ? M : 0;
#pragma unroll
            for (int r = 0; r < FP_ARR_COUNT; ++r) {
                const int len = r < FP_ARR_X ? N : (r <= FP_ARR_Y ? Mx : (r == FP_ARR_YAW ? My : (r <= FP_ARR_C ? My - 1 : (r == FP_ARR_C_D ? My - 2 : My - 3))));
                // ... rounded up to the end of the 128-byte line (NaN): with a stride that is a multiple of 16 every line is written
                // whole - partial-line stores cost a read-modify-write at the memory side (measured: 3.4 vs 6 TB/s)
                if (i < ((len + 15) & ~15)) __builtin_nontemporal_store(row[r], &out[r * stride + i]);
            }
        }
    }
    if (i == 0 && ka.r.best_flags) {
        uint32_t fl = ((uint32_t)N << FP_FLAG_N_SHIFT) | ((uint32_t)M << FP_FLAG_M_SHIFT);
        if (M < N) fl |= FP_FLAG_TRUNCATED;
        ka.r.best_flags[slot] = fl;
    }
}

}  // namespace fp
