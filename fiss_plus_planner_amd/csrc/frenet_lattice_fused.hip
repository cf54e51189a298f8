// frenet_lattice_fused.hip - the production dense-lattice kernel (gfx950).
//
// One workgroup per ego problem.  The lattice is separable, and the kernel is built around that:
//
//   * the longitudinal polynomial s(t) depends on (T, v) only  -> nt*nv "lon profiles"
//   * the lateral polynomial d(t) depends on (d_end, T) only    -> nd*nt "lat profiles"
//   * the reference-line frame (position + unit tangent at s(t)) belongs to the lon profile, so the
//     spline is evaluated nt*nv*N times per ego, not nd*nv*nt*N times (9x fewer at 9x9x7)
//   * cost_final = (time + lon sums + lat sums) / N recombines per candidate with the reference's grouping
//
// Once per ego (phase A0): one lane per lon profile solves the boundary-value problem and tries to PROVE the profile clean from
// the extrema of its polynomials (speed / acceleration limits, inside the spline's range); only slices with an unproven profile
// run the reference's point-by-point scan (lane per (profile, point), LDS atomics).  The cost sums of every profile follow in
// closed form from the per-slice power sums (Faulhaber) - no per-point work.
// Once per ego, collision stage G: for every checked pose row the arclength range of the reference points of ALL lon profiles of
// ALL slices (a polynomial evaluation each) becomes a circle - centre on the line at the middle of the range, radius = half the
// range x an upper bound of the spline's parametric speed + ego reach + a closed-form bound of the lateral offsets - and one lane
// per (row, obstacle) item tests the obstacle against it: ~7 % survive, and only their orientations are turned into (cos, sin).
// Per time-horizon slice i_T (all candidates that share T):
//   phase A  one lane per (profile, time point) the collision horizon can touch: spline frames and lateral offsets -> LDS, fan
//            bounds by LDS atomic max
//   prep     one lane per (pose row, lon profile): the fan's half-width along the reference normal
//   B        one lane per (surviving item, lon profile): circle fattened by the largest lateral offset of the slice, separating
//            axes along the reference normal and tangent
//   N        one lane per (hit, lateral sample): exact ego centre + heading, exact circle test, 4-axis separating-axis test
//            (closed: touching collides).  Survivors of G and B are appended to block-wide LDS lists with ballot / popcount + one
//            atomic per wavefront.
// Finally one lane per candidate assembles cost + flag word, a wave/LDS argmin with FOP's "last minimum wins" rule picks the
// winner, and - when the caller asked for it - one wavefront of the same workgroup writes the winner's series (frenet_winner.h).
//
// Semantics restated from the reference (paths relative to its checkout):
//   lattice + cost      planners/frenet_optimal_planner.py:69-104, planners/common/cost/cost_function.py:41-50
//   Frenet->Cartesian   planners/frenet_optimal_planner.py:106-138 (truncation at the spline end :112-113)
//   constraints         planners/frenet_optimal_planner.py:140-160
//   collision           planners/frenet_optimal_planner.py:168-208 (stride 2, horizon from obstacles[0], M==1 -> collision)
//   argmin              planners/frenet_optimal_planner.py:263-268
#include <atomic>
#include <type_traits>

#include "frenet_device.h"
#include "frenet_kernels.h"
#include "frenet_winner.h"
#include "frenet_advance.h"
#include "frenet_fissplus.h"

namespace fp {

namespace {

// fp64 -> fp32 that never rounds below the argument (x >= 0): to nearest, then one part in 2^22 up.  Replaces the software
// emulation of the directed-rounding conversion (~15 instructions) where only a conservative bound is needed.
__device__ __forceinline__ float float_above(double x) { return (float)x * 1.00000024f; }

// monotone map fp32 -> uint32 (and back), so that LDS integer atomic min / max order floats of either sign
__device__ __forceinline__ uint32_t f32_ordered(float f)
{
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float f32_from_ordered(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }
constexpr uint32_t kOrdPosInf = 0xFF800000u, kOrdNegInf = 0x007FFFFFu;  // f32_ordered(+inf), f32_ordered(-inf)

// e / d for 0 <= e, e + d < 2^22 with inv_d = 1.0f / d: (e + 0.5) / d lies at least 0.5 / d away from an integer and the two fp32
// roundings move it by less than (e / d + 1) 2^-23, so the truncation is exact.  Four full-rate instructions instead of the
// ~20 of an integer division; the remainder uses the 24-bit multiplier (full rate; v_mul_lo_u32 is quarter rate).
__device__ __forceinline__ int div_small(int e, float inv_d) { return (int)(((float)e + 0.5f) * inv_d); }
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
// e / D for a divisor known at compile time (the specialised instances of the kernel): one 24-bit multiply and a shift.
// Exact for 0 <= e <= MAXE: with M = ceil(2^20 / D) the error term e (M D - 2^20) / (D 2^20) stays below 1 / D, and e M < 2^32.
// D = 0: the divisor is a run-time value, div_small with its reciprocal.
template <int D, int MAXE>
__device__ __forceinline__ int div_by(int e, float inv_d)
{
    if constexpr (D > 0) {
        constexpr unsigned long long M = ((1ull << 20) + D - 1) / D;
        static_assert(M < (1ull << 24) && (unsigned long long)MAXE < (1ull << 24), "24-bit multiplier");
        static_assert((unsigned long long)MAXE * M < (1ull << 32), "product overflows");
        static_assert((unsigned long long)MAXE * (M * D - (1ull << 20)) < (1ull << 20), "not exact over the whole range");
        return (int)(__umul24((unsigned int)e, (unsigned int)M) >> 20);
    } else {
        return div_small(e, inv_d);
    }
}

// Orders the LDS accesses of ONE wavefront for the compiler (the hardware executes a wavefront's LDS instructions in program order):
// what lanes wrote before it, other lanes of the same wavefront read after it.  No instruction is emitted.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Hand-over wait of the workgroups appended to a launch (winner-series epilogue, FISS+ search): spin on the ego's flag - set by the
// lattice workgroup that published the ego's results - for at most kHandoverTimeout ticks of the 100 MHz wall clock.  A wait that
// runs out (only possible if the workgroup distributors did NOT start this XCD's lattice workgroups before this one, see the kernel)
// does not trap - a trap aborts the queue and with it the ctx: it leaves `code` in the ctx's error word (device-mapped host memory,
// system scope: the host sees it without a synchronisation) and the workgroup returns without output; the slot falls free, the
// launch completes, and the next call on the ctx reports the failure, resets the flags and stops using appended workgroups.
constexpr long long kHandoverTimeout = 200000000ll;  // 2 s
__device__ __forceinline__ bool handover_wait(const int32_t* flag, int32_t* err_word, int code, int timeout_us = 0)
{
    const long long t0 = wall_clock64();
    const long long limit = timeout_us > 0 ? (long long)timeout_us * 100ll : kHandoverTimeout;  // (100 MHz wall clock; timeout_us: KernelArgs::handover_timeout_us, tests only)
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > limit) {
            if (err_word) __hip_atomic_store(err_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
    }
    return true;
}

#if defined(FP_ABL_NO_SLICE_SYNC)  // timing ablation: the slice loop without its barriers (results are wrong)
#define SLICE_SYNC() do { } while (0)
#else
#define SLICE_SYNC() __syncthreads()
#endif
constexpr int kThreads = 512;
#ifndef FP_GROUP_THREADS
#define FP_GROUP_THREADS 1024  // threads per workgroup of the grouped instances
#endif
constexpr int kHitCap = 1024;    // block-wide list of (lon profile, row, obstacle) hits of one B pass  // block-wide list of live narrow-phase items (pair, lateral sample) of one B pass: 24 KB
// block-wide list of (row, obstacle) items that pass the group test (+ their poses, 32 B each).  Two kernel variants: OCC = 4 waves per
// SIMD (two workgroups per CU, up to 128 VGPRs, winner epilogue inside) and OCC = 6 (THREE workgroups per CU: 80 VGPRs - a few spill
// - and at most 53 KB of LDS, so a shorter list; no winner epilogue, the batches it serves get theirs from winner_traj_kernel or from the
// appended epilogue workgroups) and, for BASELINE.json's dense shape, OCC = 8 (FOUR per CU: 64 VGPRs, none spilled, and the 40 KB "slim"
// layout of make_layout with a 256-entry list - the largest ego of the headline workload keeps 221; longer lists take the chunked redo)
__host__ __device__ constexpr int item_cap(int occ) { return occ > 6 ? 256 : occ > 4 ? 320 : 512; }
constexpr int kItemCapMax = 512;

struct __attribute__((aligned(16))) Frame {  // reference-line frame of one lon-profile point
    double px, py, tx, ty;
};
struct __attribute__((aligned(16))) ObsPose {
    double x, y, c, s;
};
struct __attribute__((aligned(16))) ObsDim {
    double hl, hw, r, pad;
};
// A table of 32-byte records {a, b, c, d} kept as two arrays of 16-byte halves.  Consecutive lanes reading consecutive records as two
// ds_read_b128 at a 32-byte stride touch only half of the LDS banks per instruction (every fourth lane lands on the same banks: a 2-way
// conflict on every read of a frame or a survivor's pose); at a 16-byte stride the same reads are conflict free.  Same bytes in total.
template <typename T>
struct SplitTab {
    double2* lo;
    double2* hi;
    __device__ __forceinline__ T get(int i) const { const double2 a = lo[i], b = hi[i]; return T{a.x, a.y, b.x, b.y}; }
    __device__ __forceinline__ void set(int i, double a, double b, double c, double d) const { lo[i] = make_double2(a, b); hi[i] = make_double2(c, d); }
    __device__ __forceinline__ SplitTab at(int off) const { return SplitTab{lo + off, hi + off}; }
};

// LDS carve-up (all offsets in bytes, 16-byte aligned)
struct Layout {
    int knots, coef, lut, dim, pose, frames, lat, dmax, ddmax, wfat, grp, iqueue, pows, samples, lon_sum, lat_sum, lon_meta, qlon, qlat, box, coll, queue, cnt, nslice, best, konst, nvert, poly, total;
};

__host__ __device__ inline int align16(int v) { return (v + 15) & ~15; }

// Two organisations of the collision stages share the kernel's prologue (FP_SLICE_LOOP selects the older one for A/B runs):
//   walk (default)  every wavefront takes one lon profile (T, v) at a time and does everything for it by itself - frames, fan
//                   half-widths, broad phase, narrow phase - with NO workgroup barrier; the per-slice tables (frames, half-widths,
//                   hit list) are per WAVEFRONT, the lateral bounds and coefficients of all slices are computed once per ego
//   slice loop      all wavefronts work on one time-horizon slice (or a group of gs slices) per barrier interval
#if defined(FP_SLICE_LOOP)
constexpr bool kWalk = false;
#else
constexpr bool kWalk = true;
#endif
// gs = time-horizon slices the collision stages work on per barrier interval (1: one at a time; the per-slice tables are gs deep)
// Polygon columns (POLY instances): the vertex counts always sit in LDS, the rings when they are small (kPolyLdsMax: a few KB keep the
// three-workgroups-per-CU instance inside its 52 KB; 9.6 KB of 12-gons pushed config-3-sized scenes to two per CU: 208 -> 234 us) - a
// narrow-phase lane reads another obstacle than its neighbour, and from global memory a ring costs it two dependent L2 round trips
// (count, then vertices) per round.  Measured on config-3 sizes, half of the columns rings (same box): rectangle-only scene on its
// shaped instance 138.7 us; the run-time-shape POLY instance with no polygon at all 162.5; 4-vertex rings that ARE their rectangles
// 177.6 (185.9 with counts and rings in global memory).
constexpr int kPolyLdsMax = 4 * 1024;
__host__ __device__ inline int poly_lds_verts(int n_obs, int poly_stride) { return poly_stride > 0 && n_obs * poly_stride * 16 <= kPolyLdsMax ? n_obs * poly_stride : 0; }

// the walk's fan bounds in LDS: fp32, or (slim layout) fp16 that is never below the fp32 value: x (1 + 2^-10) survives the rounding to
// 11 bits (relative error <= 2^-11), the offset keeps the result a normal half (no dependence on the denormal mode), NaN stays NaN
template <bool SLIM> struct FanType { using type = float; static __device__ __forceinline__ float above(float x) { return x; } };
template <> struct FanType<true> { using type = _Float16; static __device__ __forceinline__ _Float16 above(float x) { return (_Float16)(x * 1.001f + 6.2e-5f); } };

// coef_cols: columns (segments) of the spline's coefficient rows the workgroup keeps in LDS: all nx_max of them (default), or a WINDOW
// of that many segments placed at the ego's position (the kernel's `wcap`): 64 of the 76 bytes a knot costs.  Long reference lines
// (200+ knots) then still fit the three- / four-per-CU layouts; a point whose segment lies outside the window reads global memory.
__host__ __device__ inline Layout make_layout(int nx_max, int n_obs, int rows, int hp, int nd, int nv, int nt, int kItemCap, int gs, int nwaves, int poly_stride = 0, bool slim = false,
                                              int coef_cols = -1)
{
    // slim (the four-per-CU instance, kWalk, no polygon columns): the same tables in 40 KB - no inner radii, fp16 fan bounds, 16-bit
    // hit codes, the row boxes inside the power sums' / hit lists' bytes (they are read for the last time before the first hit is written)
    Layout L;
    int o = 0;
    L.dim = o;      o = align16(o + (slim ? 24 : 32) * n_obs);
    L.pose = o;     o = align16(o + 32 * kItemCap);  // poses of the group test's survivors (x, y, cos, sin), in list order
    if (kWalk) {
        L.frames = o;   o = align16(o + 32 * nwaves * hp);   // [wavefront][point]: the profile the wavefront is working on
        L.lat = o;
        L.dmax = o;     o = align16(o + (slim ? 2 : 4) * nt * hp);   // [slice][point] float (slim: half), rounded up: max |d| over the lateral samples
        L.ddmax = o;    o = align16(o + (slim ? 2 : 4) * nt * hp);   // max |d(i + 1) - d(i)|
        L.wfat = o;     o = align16(o + 4 * nwaves * (rows > 0 ? rows : 1));  // [wavefront][row] float, rounded up
    } else {
        L.frames = o;   o = align16(o + 32 * gs * nv * hp);
        L.lat = o;      o = align16(o + 8 * gs * nd * hp);
        L.dmax = o;     o = align16(o + 2 * 4 * gs * hp);   // float, rounded up; two buffers (group parity): LDS atomic max in phase A
        L.ddmax = o;    o = align16(o + 2 * 4 * gs * hp);
        L.wfat = o;     o = align16(o + 4 * gs * nv * hp);  // float, rounded up
    }
    L.grp = o;      o = align16(o + 32 * (rows > 0 ? rows : 1));       // per checked pose row: circle enclosing all lon profiles' points
    L.iqueue = o;   o = align16(o + 2 * kItemCap);                           // (row, obstacle) items that pass the group test
    L.samples = o;  o = align16(o + 8 * (nt + nv + nd));  // t / v / d sample grids (read all over the kernel: keep them out of HBM latency)
    L.lon_sum = o;  o = align16(o + 24 * nt * nv);   // sum_v, sum_as, sum_js
    L.lat_sum = o;  o = align16(o + 24 * nd * nt);   // sum_ad, sum_jd, sum_d
    L.lon_meta = o; o = align16(o + 8 * nt * nv);    // int M, uint flags
    L.qlon = o;     o = align16(o + 16 * nt * nv);   // a3, a4 of every lon profile (a0..a2 are the ego state)
    L.qlat = o;     o = align16(o + 24 * (kWalk ? nt : gs) * nd);   // a3, a4, a5 of the lat profiles (walk: of every slice; else of the CURRENT slices)
    L.box = o;      if (!slim) o = align16(o + 16 * (rows > 0 ? rows : 1));  // per checked pose row: bounding box of the slice's reference points (ordered-uint fp32)
    L.coll = o;     o = align16(o + (kWalk ? 8 * nt * nv : nd * nv * nt));  // walk: one bit per lateral sample, a 64-bit word per lon profile; else a byte per candidate
    // per-wave hit queues; before the slice loop the same bytes hold the power sums S_k(N) = sum_i (i*tick)^k, k = 0..10, per slice
    L.queue = o;    L.pows = o;
    {
        const int q = kWalk ? (slim ? 2 : 4) * 64 * nwaves : 4 * kHitCap;
        int pw = align16(88 * nt);
        if (slim) { L.box = o + pw; pw += 16 * (rows > 0 ? rows : 1); }
        o = align16(o + (q > pw ? q : pw));
    }
    L.cnt = o;      o = align16(o + 32);  // list counters (monotone) + scan mask + ticket + two fp32 bounds
    L.nslice = o;   o = align16(o + 4 * nt);  // points per slice, len(np.arange(0, T, tick))
    L.best = o;     o = align16(o + 16 * (slim ? 8 : 16));  // (up to 16 wavefronts; slim: 8)
    L.konst = o;    o = align16(o + 96);  // per-ego constants the collision stages re-read (instead of registers held through the kernel)
    L.nvert = o;    o = align16(o + (poly_stride > 0 ? 4 * n_obs : 0));                              // polygon columns: vertices per obstacle
    L.poly = o;     o = align16(o + 16 * poly_lds_verts(n_obs, poly_stride));                        // ... and the rings, when they fit
    // the spline tables last: theirs is the one size no instance of the kernel knows at compile time, so every other offset folds
    L.knots = o;    o = align16(o + 8 * nx_max);
    L.coef = o;     o = align16(o + 64 * (coef_cols >= 0 && coef_cols < nx_max ? coef_cols : nx_max));
    L.lut = o;      o = align16(o + 2 * (2 * nx_max + 1));  // uint16 segment hint per arclength bucket
    L.total = o;
    return L;
}

}  // namespace

__device__ __forceinline__ double gk_first(const fp_batch& bt, int f) { return bt.knots[(size_t)f * bt.NX]; }
__device__ __forceinline__ double gk_last(const fp_batch& bt, int f, int nx) { return bt.knots[(size_t)f * bt.NX + nx - 1]; }

// segment of s (known to be inside [knot0, knot_last)) from the bucket table + walk
__device__ __forceinline__ int lut_segment(const double* knots, const unsigned short* lut, int nx, double s, double knot0,
                                           double inv_bucket_w, int n_buckets)
{
    int bkt = (int)((s - knot0) * inv_bucket_w);
    bkt = bkt < 0 ? 0 : (bkt > n_buckets ? n_buckets : bkt);
    int seg = lut[bkt];
    while (seg > 0 && s < knots[seg]) --seg;            // rounding at a bucket edge
    while (seg < nx - 2 && s >= knots[seg + 1]) ++seg;  // knots inside the bucket
    return seg;
}

// nsplit > 1 (latency mode for small batches): the time-horizon slices of one ego are spread over nsplit workgroups, each
// writes its partial argmin to part_best[ego * nsplit + part]; the last one to arrive (ticket counter) merges them.
//
// Template parameters = the problem shape when it is known at compile time (0 = run-time value from the arguments): lattice sizes,
// collision check stride, obstacles per scene, checked pose rows.  BASELINE.json has two shapes (9 x 9 x 7 with 50 obstacles over
// 25 rows, 5 x 5 x 5 with 10 obstacles over 50 rows); in their instances every index decode is a constant multiply-shift, the LDS
// carve-up folds to immediates and the loop bounds are known.  The <0, ...> instance is the same source with everything read from
// the arguments.
//
// GS: time-horizon slices per barrier interval of the collision stages.  1 = one slice at a time (the throughput instances: their LDS
// holds one slice's frames); 0 = gs_arg slices at a time (run-time value): a GROUP of g slices is handled like one slice with g * nv
// lon profiles and g * nd lat profiles - same four barriers, g times the lanes per pass.  Small lattices with few obstacles are bound
// by the latency of those barrier intervals (a handful of wavefronts of work each), not by arithmetic: with all nt slices in one group
// an ego costs 4 intervals instead of 4 nt and needs no split over workgroups (no ticket, no merge).
// NTH: threads per workgroup, 512 or - the grouped latency instances, one workgroup per CU - 1024 (twice the wavefronts per pass).
// the kernel's parameters as one struct: where InlineIn::bytes sits in the argument segment
// LDS of an epilogue workgroup (kThreads / 128 trajectories): [4][FP_FAST_POINTS] doubles of difference-chain scratch per trajectory,
// {first point off the spline} x 2, the argmin and the "had to wait" flag per trajectory; then, for reference lines of at most
// kEpiSplineNX knots, room for one spline copy per trajectory
constexpr int kEpiPairsC = 512 / (2 * kWave);
constexpr int kEpiLdsBytes = kEpiPairsC * 4 * FP_FAST_POINTS * 8 + kEpiPairsC * 4 * 4 + 16;
constexpr int kEpiSplineNX = 96;
struct LatticeKernarg {
    KernelArgs ka; int rows_max_arg, hp_max_arg, nsplit; Best* part_best; int* part_count; const int* perm; int* dur; int gs_arg, tail_from, epi_from, wcap; FissTail ft; InlineIn inl;
};
constexpr size_t kInlineOffset = offsetof(LatticeKernarg, inl) + offsetof(InlineIn, bytes);

// POLY: the scene may hold convex-polygon obstacle columns (fp_batch.obs_nvert != NULL).  Only the run-time-shape instances exist with
// POLY = true: the polygon branch in the narrow phase costs the rectangle-only instances 26 more spilled SGPRs and 2 % of the headline
// step (same-box A/B), and scenes with polygons are not what the shaped instances were made for.
// FISS: the launch carries one appended workgroup per ego that runs the FISS+ search (frenet_fissplus.h) on the ego's dense tables as
// soon as the ego's lattice workgroup has written them - inside the launch's drain instead of in a launch of its own (a launch that is
// one round of 4-wavefront workgroups as long as its slowest ego, behind a kernel boundary).  The tables travel between workgroups of
// ONE launch, possibly on different XCDs whose L2s are not coherent with each other: they are written with agent-scope stores and read
// with agent-scope loads; the flag follows the stores of ALL threads of the writing workgroup (s_waitcnt + barrier).
// WIN: the instance can keep a WINDOW of the spline's coefficient columns in LDS (wcap < NX, see make_layout) - reference lines too long
// for a residency's LDS share.  Its own instances: the window's bookkeeping costs the others 10-35 spilled SGPRs each (v_writelane /
// v_readlane pairs in a kernel that is VALU-issue bound), so the instances without it are exactly what they were.
template <int ND, int NV, int NT, int STRIDE, int NOBS, int ROWS, int OCC, int GS, int NTH, bool POLY = false, bool FISS = false, bool WIN = false>
__global__ __launch_bounds__(NTH, OCC) void lattice_fused_kernel(KernelArgs ka, int rows_max_arg, int hp_max_arg, int nsplit_arg, Best* part_best, int* part_count, const int* perm,
                                                                   int* dur, int gs_arg, int tail_from, int epi_from, int wcap, FissTail ft, InlineIn inl)
{
    if constexpr (FISS) {
        if (epi_from >= 0 && (int)blockIdx.x >= epi_from) {  // ---- appended search workgroup: one ego, in launch order
            extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
            const int eslot = (int)blockIdx.x - epi_from;
            const int eb = perm ? perm[eslot] : eslot;
#if defined(FP_TL)  // timeline diagnostic (tools/fused_timeline.py, -DFP_TL): the FISS+ outputs carry clock ticks instead of results
            const long long tl0 = wall_clock64();
            long long tl1 = 0;
#endif
            int timed_out = 0;
            if (threadIdx.x == 0) {
                timed_out = !handover_wait(&ft.flag[eb], ka.err_word, 2, ka.handover_timeout_us);
                if (!timed_out) __hip_atomic_store(&ft.flag[eb], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
#if defined(FP_TL)
                tl1 = wall_clock64();
#endif
            }
            if (__syncthreads_or(timed_out)) return;  // (timed out: reported through the ctx's error word, see handover_wait)
            __atomic_signal_fence(__ATOMIC_SEQ_CST);  // the tables are read after the flag
            FissArgs fa;
            fa.ka = ka;
            fa.opts = ft.opts;
            fa.io = ft.io;
            fa.cost_tbl = ka.r.cost_tbl;
            fa.flag_tbl = ka.r.flag_tbl;
            fa.walk_jump = ft.walk_jump;
            fsp::fissplus_search_ego<NTH / kWave, 1, 1024, true>(fa, ft.NB, eb, fsm);
#if defined(FP_TL)
            if (threadIdx.x == 0) {  // (the lattice workgroup left its start / end in the dense call's best_idx / best_cost)
                const long long tl2 = wall_clock64();
                const double lat_end = __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)&ka.r.best_cost[eb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const int lat_start = __hip_atomic_load(&ka.r.best_idx[eb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_waitcnt(0);
                double* es = ft.io.end_state + (size_t)eb * 3;
                es[0] = (double)(tl0 & 0xFFFFFFFFFFll); es[1] = (double)(tl1 & 0xFFFFFFFFFFll); es[2] = (double)(tl2 & 0xFFFFFFFFFFll);
                ft.io.best_cost[eb] = lat_end;
                ft.io.refined[eb] = lat_start;
            }
#endif
            return;
        }
    }
    // ---------------------------------------------------------------- appended epilogue workgroups (blockIdx >= epi_from)
    // The three-workgroups-per-CU instances cannot write the winner's series themselves (the slot time of a one-wavefront dependent
    // chain, and 125 registers), and a winner_traj_kernel behind the launch costs ~9.5 us of a ~155 us step in which the last ~25 us
    // run on a draining chip.  So the series are written INSIDE the drain: the grid carries B / 4 more workgroups at its end - the
    // hardware dispatches workgroups in index order, so they become resident when every lattice workgroup has been dispatched and
    // slots fall free - each of which takes four dispatch slots (in launch order: the egos that started first), waits for their argmins
    // (flag per ego, set by the workgroup that published it; relaxed agent-scope atomics, see the ticket's ordering argument below)
    // and writes the four series, two wavefronts per trajectory (winner_series_pair: one point per lane, 80 registers).
    // No deadlock: an epilogue workgroup only ever waits for lattice workgroups, and a workgroup distributor (one per XCD, each
    // taking every eighth workgroup) starts its workgroups in index order: when an epilogue workgroup holds a slot, every lattice
    // workgroup of the same XCD has been started, and those of the other XCDs do not depend on this XCD's slots.  That order is
    // observed, not documented: the host offers appended workgroups only on the architectures it was verified on (gfx942 / gfx950,
    // fp_ctx_create), and should it ever not hold the wait gives up after 2 s WITHOUT a trap (handover_wait): the slot falls free,
    // the launch completes, and the next call on the ctx reports the failed hand-over and falls back to winner_traj_kernel.
    if constexpr (OCC > 4) {
        if (epi_from >= 0 && (int)blockIdx.x >= epi_from) {
            extern __shared__ __attribute__((aligned(16))) unsigned char esm[];
            constexpr int kPairs = NTH / (2 * kWave);  // trajectories per workgroup
            double* scratch = (double*)esm;                         // [kPairs][4][FP_FAST_POINTS]
            int* s_m = (int*)(esm + kPairs * 4 * FP_FAST_POINTS * 8);  // [kPairs][2] first point off the spline per wavefront
            int* s_idx = s_m + 2 * kPairs;                           // [kPairs] the egos' argmins
            const int pair = (int)threadIdx.x / (2 * kWave), i = (int)threadIdx.x - pair * 2 * kWave;
            const int eslot = ((int)blockIdx.x - epi_from) * kPairs + pair;
            const bool have = eslot < ka.b.B;
            const int eb = have ? (perm ? perm[eslot] : eslot) : 0;
            const fp_params& pp = ka.p;
            const fp_batch& bb = ka.b;
            const int ef = bb.frame_of[eb];
            const double* gk = bb.knots + (size_t)ef * bb.NX;
            const double* gc = bb.coef + (size_t)ef * 8 * bb.NX;
            // A pair that finds its ego's argmin not published yet is one of the launch's last: the launch ends when IT is done.  It spends
            // the wait copying the ego's spline to LDS (9 NX doubles, when four copies fit), so that the segment searches and coefficient
            // reads of the series - three to four dependent L2 round trips after the flag - become LDS reads.  Pairs whose argmin is
            // already there (the many that run inside the drain) read the few segments they touch from global memory as before.
            int* s_wait = s_idx + kPairs;
            double* s_spl = (double*)(esm + kEpiLdsBytes) + (size_t)pair * 9 * bb.NX;
            const bool can_copy = bb.NX <= kEpiSplineNX;
            if (have && i == 0) s_wait[pair] = can_copy && __hip_atomic_load(&ka.epi_flag[eb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
            __syncthreads();
            const bool in_lds = have && s_wait[pair] != 0;
            if (in_lds)
                for (int k = i; k < 9 * bb.NX; k += 2 * kWave) s_spl[k] = k < bb.NX ? gk[k] : gc[k - bb.NX];
            if (have && i == 0) {
                if (handover_wait(&ka.epi_flag[eb], ka.err_word, 1, ka.handover_timeout_us)) {
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);  // the index is read after the flag
                    s_idx[pair] = __hip_atomic_load(&ka.idx_shadow[eb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&ka.epi_flag[eb], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                } else {
                    s_idx[pair] = -2;  // timed out: nothing is written for this ego (reported through the ctx's error word)
                }
            }
            __syncthreads();
            const int win = have ? s_idx[pair] : -1;
            const bool handed = have && win != -2;
            double d_end = __builtin_nan(""), v_end = d_end, T_end = d_end;
            if (win >= 0) {
                const int iv = win % pp.nv, it = (win / pp.nv) % pp.nt, id = win / (pp.nv * pp.nt);
                d_end = bb.d_samples[id]; v_end = bb.v_samples[(size_t)eb * pp.nv + iv]; T_end = bb.t_samples[it];
            }
            // (a pair beyond the batch still runs the barriers; it writes nothing)
            const SplineLds sp = in_lds ? SplineLds{s_spl, s_spl + bb.NX, bb.nx[ef], bb.NX} : SplineLds{gk, gc, bb.nx[ef], bb.NX};
            winner_series_pair(ka, eb, eb, win >= 0, d_end, v_end, T_end, i, sp, scratch + pair * 4 * FP_FAST_POINTS, s_m + 2 * pair, handed);
            return;
        }
    }
    // Dispatch slot -> (ego, part).  Uniform split (latency mode): slot = blockIdx / nsplit.  Tail split (tail_from >= 0, nsplit_arg
    // == 1): the slots from tail_from on - the workgroups that start when the launch's last round is already draining - are cut in
    // two like a latency-mode ego, so the launch does not end on whole egos that started last (see launch_lattice_fused).
    const bool in_tail = tail_from >= 0 && (int)blockIdx.x >= tail_from;
    const int nsplit = in_tail ? 2 : nsplit_arg;
    const int slot = in_tail ? tail_from + (((int)blockIdx.x - tail_from) >> 1) : (int)blockIdx.x / nsplit_arg;
    const int part_of_slot = in_tail ? (((int)blockIdx.x - tail_from) & 1) : (int)blockIdx.x - slot * nsplit_arg;
    constexpr bool kShape = ND > 0;  // (all six are set together)
    // The run-time-shape three-per-CU instances have no register to spare (80 VGPRs, run-time sizes in place of immediates): from the
    // prologue's second barrier on they re-read the ego's start state from LDS (s_k[5..10]) where it is used instead of holding twelve
    // VGPRs through the kernel.  NOT an optimisation: this build of the compiler places VGPR spill stores in the exit block of a
    // divergent loop BEFORE the exec mask is restored (exec = 0: nothing is stored, the reload returns whatever the scratch slot held -
    // zeros in a launch's first round of workgroups, another workgroup's values later), so no instance may spill a VGPR at all
    // (tools/resource_usage.py must show 0 in "VGPRs Spill" for every kernel; tests/test_abi_cpu.py checks it).
    constexpr bool kEgoLds = (ND == 0 && OCC > 4) || OCC > 6;
    constexpr bool kEgoEarly = OCC > 6;  // four per CU (64 VGPRs): from the FIRST barrier on
    constexpr bool kGroup = GS != 1;
    constexpr int kThreads = NTH, kWaves = NTH / kWave;  // (shadow the file-scope defaults)
    static_assert(GS == 0 || GS == 1, "GS: 1 or 0 (run-time group size)");
    constexpr int kItemCap = item_cap(OCC);
    constexpr bool kSlim = OCC > 6;  // four workgroups per CU: the 40 KB layout (make_layout)
    static_assert(!kSlim || (kWalk && !POLY && NTH <= 512 && item_cap(OCC) <= 256), "slim layout: walk, boxes only, 8 wavefronts, 8-bit survivor index");
    const int gs = kGroup ? gs_arg : 1;
    const int rows_max = kShape ? ROWS : rows_max_arg;
    const int hp_max = kShape ? (ROWS > 0 ? (ROWS * STRIDE + 1 > FP_MAX_POINTS ? FP_MAX_POINTS : ROWS * STRIDE + 1) : 0) : hp_max_arg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#if defined(FP_PHASE_STAMPS)
    const long long t_begin = wall_clock64();
#elif defined(FP_TL)
    const long long t_begin = wall_clock64();
#else
    const long long t_begin = dur ? wall_clock64() : 0;
#endif
    // Timing diagnostic (tools/phase_stamps.py, -DFP_PHASE_STAMPS): thread 0 leaves the time since the workgroup started (10 ns
    // ticks) at the phase boundaries in columns 112.. of the last row of its best_traj block (free with traj_stride = 128, sparse, T <= 11 s).
#if defined(FP_PHASE_STAMPS)
#define FP_STAMP(k) do { if (threadIdx.x == 0 && ka.r.best_traj && (perm || blockIdx.x % nsplit == 0)) ka.r.best_traj[((size_t)(perm ? perm[blockIdx.x] : blockIdx.x / nsplit) * FP_ARR_COUNT + 15) * (ka.r.traj_stride > 0 ? ka.r.traj_stride : FP_DEFAULT_STRIDE) + 112 + (k)] = (double)(wall_clock64() - t_begin); } while (0)
// (the one workgroup of an ego that is left after the ticket - whichever part it is - stamps columns 5, 6 and 10)
#define FP_STAMP_LAST(k) do { if (threadIdx.x == 0 && ka.r.best_traj) ka.r.best_traj[((size_t)b * FP_ARR_COUNT + 15) * (ka.r.traj_stride > 0 ? ka.r.traj_stride : FP_DEFAULT_STRIDE) + 112 + (k)] = (double)(wall_clock64() - t_begin); } while (0)
#else
#define FP_STAMP(k) do { } while (0)
#define FP_STAMP_LAST(k) do { } while (0)
#endif
#if defined(FP_PHASE_STAMPS)
    __shared__ unsigned int s_walk_t[2];  // when the first / the last wavefront of the workgroup finished its profiles
#endif
#if defined(FP_COUNTERS)  // work statistics (tools/work_counters.py): LDS counters, left in row 14, columns 112.. of the winner block
    __shared__ int s_dbg[16];
    if (threadIdx.x < 16) s_dbg[threadIdx.x] = 0;
#define FP_COUNT(k, v) atomicAdd(&s_dbg[k], (int)(v))
#else
#define FP_COUNT(k, v) do { } while (0)
#endif
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    // launch order: longest egos first when the host has an order for this batch (perm; nsplit == 1 then)
    const int b = perm ? perm[slot] : slot, part = part_of_slot;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    const int nd = kShape ? ND : p.nd, nv = kShape ? NV : p.nv, nt = kShape ? NT : p.nt;
    const int C = nd * nv * nt;
    const int it_lo = part * nt / nsplit, it_hi = (part + 1) * nt / nsplit;  // this workgroup's slices
    const int n_it = it_hi - it_lo;
    const int stride = kShape ? STRIDE : p.check_stride;
    const int n_obs_tab = kShape ? NOBS : bt.n_obs;  // obstacles per scene of the table
    const double tick = p.tick_t;

    // wcap < NX: the coefficient rows in LDS are a WINDOW of wcap segments (make_layout's coef_cols) that starts one segment behind the ego
    const bool windowed = WIN && wcap < bt.NX;
    const int ld = windowed ? wcap : bt.NX;  // row stride of the LDS coefficient table
    const Layout L = make_layout(bt.NX, n_obs_tab, rows_max, hp_max, nd, nv, nt, kItemCap, gs, kWaves, POLY ? bt.poly_stride : 0, kSlim, wcap);
    double* s_knots = (double*)(smem + L.knots);
    double* s_coef = (double*)(smem + L.coef);
    unsigned short* s_lut = (unsigned short*)(smem + L.lut);
    // obstacle sizes: (half length, half width) pairs and the bounding radii in their own array (a lane reads another obstacle's record
    // than its neighbour: 24 useful bytes per lane instead of a 32-byte record)
    struct DimTab {
        double2* hlw;
        double* rad;
        double* rin;  // polygon columns: radius of the disk about the rotation centre inside the ring (poly_inner_radius); else 0
        __device__ __forceinline__ ObsDim get(int j) const { const double2 a = hlw[j]; return ObsDim{a.x, a.y, rad[j], 0.0}; }
    };
    const DimTab s_dim{(double2*)(smem + L.dim), (double*)(smem + L.dim) + 2 * n_obs_tab, (double*)(smem + L.dim) + 3 * n_obs_tab};
    // polygon rings of the ego's scene: the LDS copy when it exists (indexed by obstacle), else the global table (indexed by column)
    const bool rings_in_lds = POLY && poly_lds_verts(n_obs_tab, ka.b.poly_stride) > 0;
    // poses of the (row, obstacle) items that survive the group test, in the order of the item list: the table itself is read from
    // global memory exactly once (registers -> group test), only the ~7 % the slices can touch are kept
    const SplitTab<ObsPose> s_spose{(double2*)(smem + L.pose), (double2*)(smem + L.pose) + kItemCap};
    const int n_frames = kWalk ? mul24(kWaves, hp_max) : mul24(mul24(gs, kShape ? NV : p.nv), hp_max);  // records of the frame table
    const SplitTab<Frame> s_frames{(double2*)(smem + L.frames), (double2*)(smem + L.frames) + n_frames};
    double* s_lat = (double*)(smem + L.lat);
    float* s_dmax2 = (float*)(smem + L.dmax);    // [2][gs][hp_max]
    float* s_ddmax2 = (float*)(smem + L.ddmax);  // [2][gs][hp_max]
    using fan_t = typename FanType<kSlim>::type;  // the walk's view of the two: [slice][hp_max]
    fan_t* s_fan_d = (fan_t*)(smem + L.dmax);
    fan_t* s_fan_dd = (fan_t*)(smem + L.ddmax);
    float* s_wfat = (float*)(smem + L.wfat);
    ObsDim* s_grp = (ObsDim*)(smem + L.grp);  // reused as {cx, cy, radius, -}
    unsigned short* s_items = (unsigned short*)(smem + L.iqueue);  // item index mul24(r, n_obs) + j
    double* s_pows = (double*)(smem + L.pows);
    double* s_ts = (double*)(smem + L.samples);
    double* s_vs = s_ts + nt;
    double* s_ds = s_vs + nv;
    double* s_lon_sum = (double*)(smem + L.lon_sum);
    double* s_lat_sum = (double*)(smem + L.lat_sum);
    int2* s_lon_meta = (int2*)(smem + L.lon_meta);
    double* s_qlon = (double*)(smem + L.qlon);  // [nt][nv][2]
    double* s_qlat = (double*)(smem + L.qlat);  // [gs][nd][3]
    uint4* s_box = (uint4*)(smem + L.box);      // [rows] {min x, max x, min y, max y} relative to the first knot
    unsigned char* s_coll = smem + L.coll;
    uint32_t* s_collmask = (uint32_t*)(smem + L.coll);  // walk: [lon profile][2], bit id = lateral sample id of the profile collides
    uint32_t* s_hits = (uint32_t*)(smem + L.queue);
    using hit_t = typename std::conditional<kSlim, uint16_t, uint32_t>::type;  // walk: (point k, survivor index) codes
    int* s_cnt = (int*)(smem + L.cnt);
    int* s_nslice = (int*)(smem + L.nslice);
    Best* s_best = (Best*)(smem + L.best);
    // {first knot, bucket width reciprocal, ego half length, half width, half diagonal}: uniform FP64 values live in VECTOR registers
    // (the scalar unit has no FP64); kept in registers from the prologue to the last slice they cost the three-workgroup variant spills
    double* s_k = (double*)(smem + L.konst);

    // the ego's scalars: independent global reads, all issued before the first one is waited for (a read costs ~1-2 us)
    // the per-ego arrays: at the addresses in the batch, or (InlineIn, latency instances only) inside this kernel's own argument block
    const double *in_d = bt.d_samples, *in_t = bt.t_samples, *in_v = bt.v_samples, *in_ts = bt.target_speed, *in_ego = bt.ego;
    const int32_t *in_frame = bt.frame_of, *in_scene = bt.scene_of, *in_tnow = bt.t_now;
    if constexpr (OCC <= 4) {
        if (inl.on) {
            // (the blob's place in this kernel's own argument segment: taking the address of the by-value parameter itself would
            // make the compiler keep a private copy of it)
            const unsigned char* base = (const unsigned char*)__builtin_amdgcn_kernarg_segment_ptr() + kInlineOffset;
            in_d = (const double*)(base + (size_t)bt.d_samples); in_t = (const double*)(base + (size_t)bt.t_samples);
            in_v = (const double*)(base + (size_t)bt.v_samples); in_ts = (const double*)(base + (size_t)bt.target_speed);
            in_ego = (const double*)(base + (size_t)bt.ego); in_frame = (const int32_t*)(base + (size_t)bt.frame_of);
            in_scene = (const int32_t*)(base + (size_t)bt.scene_of); in_tnow = (const int32_t*)(base + (size_t)bt.t_now);
            if (inl.publish && blockIdx.x == 0)  // (InlineIn::publish: the following kernels of the call read the arrays from device memory)
                for (int i = threadIdx.x; i < inl.n8; i += kThreads) ((unsigned long long*)inl.publish)[i] = ((const unsigned long long*)base)[i];
        }
    }
    const int skip_flag = bt.skip ? bt.skip[b] : 0;
    const int f = in_frame[b];
    const int sc = in_scene[b];
    const double* ring_base = rings_in_lds ? (const double*)(smem + L.poly) : ka.b.obs_poly;
    const size_t ring_col0 = rings_in_lds ? (size_t)0 : (size_t)(sc >= 0 ? sc : 0) * (size_t)ka.b.n_obs;
    const int t_now = in_tnow[b];
    const double target_speed = in_ts[b];
    const double* eg = in_ego + (size_t)b * 6;
    const double s0 = eg[0], s_d0 = eg[1], s_dd0 = eg[2], d0 = eg[3], d_d0 = eg[4], d_dd0 = eg[5];
    if (skip_flag) {  // finished ego of a closed-loop batch (block-uniform exit)
        if (part == 0) {
            if (tid == 0 && dur) dur[b] = 0;
            if (tid == 0) { ka.r.best_idx[b] = -1; ka.r.best_cost[b] = __builtin_nan(""); }
            if constexpr (OCC > 4) {
                if (ka.epi_flag) {  // (the appended epilogue workgroups write the NaN series / the zero flag word of a skipped ego too)
                    if (tid == 0) {
                        __atomic_signal_fence(__ATOMIC_SEQ_CST);
                        __hip_atomic_store(&ka.idx_shadow[b], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __atomic_signal_fence(__ATOMIC_SEQ_CST);
                        __builtin_amdgcn_s_waitcnt(0);
                        __atomic_signal_fence(__ATOMIC_SEQ_CST);
                        __hip_atomic_store(&ka.epi_flag[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return;
                }
            }
            if (tid == 0 && ka.idx_shadow) ka.idx_shadow[b] = -1;
            if constexpr (FISS) {
                if (tid == 0 && ft.flag) __hip_atomic_store(&ft.flag[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (its search workgroup writes the "none" outputs)
            }
            if (ka.r.best_traj) {  // NaN series, flag word 0 (returns before it touches the spline or the scratch)
                const double nan = __builtin_nan("");
                if (wave == 0) winner_series_wave(ka, b, b, false, nan, nan, nan, lane, SplineLds{nullptr, nullptr, 0, 0});
            }
        }
        return;
    }
    // ---------------------------------------------------------------- stage: ego, spline, obstacle rows
    // Global reads cost 1-2 us each, so everything that depends only on the scalars above is requested at once.  The obstacle
    // table is NOT staged: the group test reads it once, straight from global memory, and keeps the poses of its survivors
    // (rows_stage = the rows the table holds from t_now on).
    const int NX = bt.NX;
    const int n_obs = sc >= 0 ? n_obs_tab : 0;
    const float inv_nobs_s = 1.0f / (float)(n_obs > 0 ? n_obs : 1);
    int rows_stage = 0;  // obstacle rows of the table from t_now on: poses k = r*stride, k + t_now < T_obs
    if (n_obs > 0) {
        const int in_table = bt.T_obs - t_now;
        rows_stage = in_table > 0 ? (in_table + stride - 1) / stride : 0;
        if (rows_stage > rows_max) rows_stage = rows_max;
    }
    const double* gp = bt.obs_pose + (size_t)(sc >= 0 ? sc : 0) * bt.T_obs * n_obs_tab * 4;
#ifndef FP_POSE_FLIGHT
#define FP_POSE_FLIGHT 2
#endif
    // pose reads in flight per lane in the group test.  The four-per-CU instances take three (512 x 3 items: the 1250 of config 3 in ONE pass;
    // same-box A/B 130.7-131.2 against 131.6-132.2 us per step); three in the three-per-CU shaped instance spill two VGPRs.
    constexpr int kPoseFlight = OCC > 6 ? FP_POSE_FLIGHT + 1 : FP_POSE_FLIGHT;
    auto fetch_poses = [&](int i0, double4* ps) {
#pragma unroll
        for (int u = 0; u < kPoseFlight; ++u) {
            const int i = i0 + u * kThreads + tid;
            ps[u] = make_double4(0.0, 0.0, 0.0, 0.0);
            if (i < rows_stage * n_obs) {
                const int r = div_by<NOBS, ROWS * NOBS>(i, inv_nobs_s), j = i - mul24(r, n_obs);
                ps[u] = *(const double4*)(gp + ((size_t)(mul24(r, stride) + t_now) * n_obs + j) * 4);
            }
        }
    };
    const int nx = bt.nx[f];
    const int fts = n_obs > 0 ? bt.final_time_step[sc] : 0;
    int w0 = 0;  // first segment of the coefficient window (windowed), else 0
    {
        const double* gk = bt.knots + (size_t)f * NX;
        const double* gc = bt.coef + (size_t)f * 8 * NX;
        // [0], [1]: list counters; [2]: slices whose lon profiles need the point-by-point scan; [3]: ticket; [5], [6]: fp32 bit patterns
        // of the spline's speed bound and of the largest lateral offset
        if (tid < 8) s_cnt[tid] = 0;
#if defined(FP_PHASE_STAMPS)
        if (tid == 0) { s_walk_t[0] = 0xFFFFFFFFu; s_walk_t[1] = 0u; }
#endif
        // all NX columns (the copy does not wait for nx; columns >= nx hold the +inf padding / are never addressed)
        for (int i = tid; i < NX; i += kThreads) s_knots[i] = gk[i];
        for (int i = tid; i < nt + nv + nd; i += kThreads)
            s_ts[i] = i < nt ? in_t[i] : (i < nt + nv ? in_v[(size_t)b * nv + (i - nt)] : in_d[i - nt - nv]);
        if (windowed) {
            // Coefficient WINDOW: the segment the ego stands on = (knots <= s0) - 1, counted by the threads that hold a knot each (one more
            // barrier and one more dependent round trip than the whole-table copy: only lines too long for the layout pay it); the window
            // starts one segment behind it.  Whatever a trajectory point needs outside the window is read from global memory (rare: the
            // window is sized by the host for the residency it buys, not for a guarantee).
            int below = 0;
            for (int i0 = 0; i0 < NX; i0 += kThreads) below += __syncthreads_count(i0 + tid < nx && gk[i0 + tid] <= s0);
            w0 = below - 2;  // (the ego's segment - 1)
            w0 = w0 > nx - 1 - wcap ? nx - 1 - wcap : w0;
            w0 = __builtin_amdgcn_readfirstlane(w0 < 0 ? 0 : w0);
            for (int i = tid; i < 8 * wcap; i += kThreads) {
                const int r = i / wcap, c = i - r * wcap;
                s_coef[i] = w0 + c < NX ? gc[(size_t)r * NX + w0 + c] : 0.0;
            }
        } else if constexpr (OCC > 6) {
            // [8][NX], same layout; the four-per-CU instances in 16-byte pieces (8 NX doubles are an even count; a global load needs no more
            // than the doubles' own alignment) - one trip instead of two; in the polygon three-per-CU instance the same change spills two VGPRs
            for (int i = tid; i < 4 * NX; i += kThreads) ((double2*)s_coef)[i] = ((const double2*)gc)[i];
        } else {
            for (int i = tid; i < 8 * NX; i += kThreads) s_coef[i] = gc[i];
        }
        if (n_obs > 0) {
            const double* gd = bt.obs_dims + (size_t)sc * n_obs * 2;
            if constexpr (POLY) {  // the scene's rings, when they fit (same layout as in global memory: [obstacle][poly_stride] vertex pairs)
                const int nv2 = poly_lds_verts(n_obs_tab, bt.poly_stride);
                const double2* gr = (const double2*)bt.obs_poly + (size_t)sc * n_obs * bt.poly_stride;
                for (int i = tid; i < nv2; i += kThreads) ((double2*)(smem + L.poly))[i] = gr[i];
            }
            for (int j = tid; j < n_obs; j += kThreads) {
                const double hl = 0.5 * gd[2 * j], hw = 0.5 * gd[2 * j + 1];
                s_dim.hlw[j] = make_double2(hl, hw);
                s_dim.rad[j] = sqrt(fma(hl, hl, hw * hw));
                if constexpr (POLY) {
                    const size_t col = (size_t)sc * n_obs + j;
                    const int nvx = bt.obs_nvert[col];
                    s_dim.rin[j] = nvx > 0 ? poly_inner_radius(bt.obs_poly + col * 2 * (size_t)bt.poly_stride, nvx) : 0.0;
                    ((int32_t*)(smem + L.nvert))[j] = nvx;
                }
            }
        }
    }
    // the LDS spline: every knot, and the coefficient columns of segments [w0, w0 + ld) addressed by their ABSOLUTE segment index
    SplineLds sp{s_knots, s_coef - w0, nx, ld};
    // frame of (segment, offset): from the LDS table, or - a segment outside the coefficient window - from the batch's table in global memory
    auto frame_at = [&](int seg, double dxs, double& px, double& py, double& tx, double& ty) {
        if (!windowed || (unsigned)(seg - w0) < (unsigned)wcap) {
            spline_frame(sp, seg, dxs, px, py, tx, ty);
        } else {
            const int fg = in_frame[b];
            spline_frame(SplineLds{bt.knots + (size_t)fg * bt.NX, bt.coef + (size_t)fg * 8 * bt.NX, nx, bt.NX}, seg, dxs, px, py, tx, ty);
        }
    };
    if (kEgoEarly && tid == 8) { s_k[5] = s0; s_k[6] = s_d0; s_k[7] = s_dd0; s_k[8] = d0; s_k[9] = d_d0; s_k[10] = d_dd0; }
    __syncthreads();
    FP_STAMP(0);
    // the ego's start state as the prologue's solves read it (kEgoEarly: from LDS already)
    auto pS0 = [&]() -> double { if constexpr (kEgoEarly) return s_k[5]; else return s0; };
    auto pSd0 = [&]() -> double { if constexpr (kEgoEarly) return s_k[6]; else return s_d0; };
    auto pSdd0 = [&]() -> double { if constexpr (kEgoEarly) return s_k[7]; else return s_dd0; };
    auto pD0 = [&]() -> double { if constexpr (kEgoEarly) return s_k[8]; else return d0; };
    auto pDd0 = [&]() -> double { if constexpr (kEgoEarly) return s_k[9]; else return d_d0; };
    auto pDdd0 = [&]() -> double { if constexpr (kEgoEarly) return s_k[10]; else return d_dd0; };
    // Arclength buckets -> segment hint: lut[b] = bisect_right(knots, k0 + b*width) - 1, 2*nx buckets.  A point then
    // needs one table read plus a short walk instead of a log2(nx) search (knots may be non-uniform: the walk fixes it).
    const int n_buckets = 2 * nx;
    const double knot0 = s_knots[0], knot_last = s_knots[nx - 1];
    const double bucket_w = (knot_last - knot0) / (double)n_buckets;
    const double inv_bucket_w = bucket_w > 0.0 ? 1.0 / bucket_w : 0.0;
    for (int bkt = tid; bkt <= n_buckets; bkt += kThreads) {
        const double sb = knot0 + (double)bkt * bucket_w;
        int lo = 0, hi = nx;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sb < s_knots[mid]) hi = mid; else lo = mid + 1;
        }
        int seg = lo - 1;
        seg = seg < 0 ? 0 : (seg > nx - 2 ? nx - 2 : seg);
        s_lut[bkt] = (unsigned short)seg;
    }
#if defined(FP_ABL_NO_VMAX)
    if (false) {
#else
    if (n_obs > 0) {
#endif
        // upper bound of the spline's parametric speed |P'(s)| (the arclength parameter is the cumulative CHORD length, so it is a
        // little above 1 in bends): per segment the exact maxima of the two quadratic derivative components (ends + vertex).  The
        // once-per-ego group test turns an arclength interval into a circle with it.  Wave maximum of fp32 bit patterns, one LDS
        // atomic per wavefront (a NaN coefficient is the largest pattern: the circle then keeps every obstacle).
        // (coefficient window: over its segments only - a row whose arclength range leaves the window gets an infinite circle below)
        const int seg_end = windowed ? (w0 + wcap < nx - 1 ? w0 + wcap : nx - 1) : nx - 1;
        for (int i0 = w0 + wave * kWave; i0 < seg_end; i0 += kThreads) {
            const int i = i0 + lane;
            uint32_t bits = 0u;
            if (i < seg_end) {
                const double h = s_knots[i + 1] - s_knots[i];
                double m2 = 0.0;
#pragma unroll
                for (int ax = 0; ax < 2; ++ax) {
                    const double b1 = sp.coef[(4 * ax + 1) * ld + i], c2 = 2.0 * sp.coef[(4 * ax + 2) * ld + i], d3 = 3.0 * sp.coef[(4 * ax + 3) * ld + i];
                    double m = fmax(fabs(b1), fabs(fma(fma(d3, h, c2), h, b1)));  // g(u) = b1 + c2 u + d3 u^2 at u = 0, h
                    const double uv = -c2 / (2.0 * d3);
                    if (uv > 0.0 && uv < h) m = fmax(m, fabs(fma(fma(d3, uv, c2), uv, b1)));
                    if (!(m == m) || !(h == h)) m = __builtin_nan("");
                    m2 = fma(m, m, m2);
                }
                bits = __float_as_uint(float_above(sqrt(m2)) * 1.0000005f);
            }
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) {
                const uint32_t o2 = (uint32_t)__shfl_xor((int)bits, off, kWave);
                bits = o2 > bits ? o2 : bits;
            }
            if (lane == 0) atomicMax((unsigned int*)&s_cnt[5], bits);
        }
    }
    int horizon_cap = 0;  // final_time_step - time_step_now (:173-174)
    int rows = 0;         // obstacle rows the collision horizon can touch: rows of the table below final_time_step
    if (n_obs > 0) {
        horizon_cap = fts - t_now;
        int h = horizon_cap < points_cap(p) ? horizon_cap : points_cap(p);
        if (h < 0) h = 0;
        rows = (h + stride - 1) / stride;
        if (rows_stage < rows) rows = rows_stage;
    }
    if constexpr (kWalk) {
        for (int c = tid; c < 2 * nt * nv; c += kThreads) s_collmask[c] = 0u;
    } else {
        for (int c = tid; c < C; c += kThreads) s_coll[c] = 0;
    }
    for (int r = tid; r < rows; r += kThreads) {
        s_box[r] = make_uint4(kOrdPosInf, kOrdNegInf, kOrdPosInf, kOrdNegInf);  // empty
    }
    if constexpr (!kWalk)
        for (int i = tid; i < 2 * gs * hp_max; i += kThreads) { s_dmax2[i] = 0.0f; s_ddmax2[i] = 0.0f; }
    // points whose frame the collision stage can touch: 0 .. hp-1 (pose k needs point k+1 for its heading)
    const int pose_limit = rows * stride < horizon_cap ? rows * stride : horizon_cap;  // poses k < pose_limit (and k < M)
    int hp = pose_limit > 0 ? pose_limit + 1 : 0;
    if (hp > hp_max) hp = hp_max;
    const double veh_hl = 0.5 * p.veh_l, veh_hw = 0.5 * p.veh_w;
    const double r_ego = sqrt(fma(veh_hl, veh_hl, veh_hw * veh_hw));
    // (no barrier: phase A0 below needs only what the staging barrier made visible - the LUT / speed bound above and the solves below
    // are independent and share one barrier interval; the solves go to the LAST threads, the LUT to the first)
    FP_STAMP(1);

    const double* v_samples = s_vs;
    int item_base = 0, hit_base = 0;  // list counters at the start of the current stage (block-uniform)

    // ---------------------------------------------------------------- phase A0 (once): masks / M / cost sums of every profile
    // (0) lane per lon profile: the boundary-value solve (two divisions), kept for the whole kernel; meta = {N, no flags}.
    // (1) lane per (lon profile, time point): speed / acceleration masks by LDS atomic OR, the truncation index M (first point
    //     off the spline, a pure range test) by LDS atomic MIN - the same brute force over every point as the reference, with
    //     every lane busy (a wavefront per profile would idle a third of its lanes and all of its second half);
    //     one lane per slice: the power sums S_k = sum_{i<N} t_i^k (k = 0..10) in closed form (Faulhaber, power_sums_closed).
    // (2) lane per profile: the six cost sums in closed form.  Each summand is the square of a polynomial in t
    //     (s_d - v_target: cubic, s_dd: quadratic, s_ddd: linear, d: quintic, d_dd: cubic, d_ddd: quadratic), so
    //     sum_i p(t_i)^2 = sum_k c_k S_k with c = p (*) p (coefficient convolution) - no per-point work and no reductions.
    //     Conditioning is benign on t in [0, 10] (terms ~1e2..1e4 against sums ~1e1..1e3: ~1e-12 absolute), far inside the
    //     1e-6 cost bar, and the expressions are even in the lateral boundary data, so mirrored candidates still tie bit-exactly.
    for (int e = kThreads - 1 - tid; e < n_it * nv; e += kThreads) {
        const int eq = div_by<NV, NT * NV>(e, 1.0f / (float)nv);
        const int it = it_lo + eq, iv = e - mul24(eq, nv);
        const double T = s_ts[it];
        const Quartic q = quartic_bvp(pS0(), pSd0(), pSdd0(), v_samples[iv], 0.0, T);
        s_qlon[2 * (mul24(it, nv) + iv)] = q.a3;
        s_qlon[2 * (mul24(it, nv) + iv) + 1] = q.a4;
        const int N = arange_len(T, tick);
        s_lon_meta[mul24(it, nv) + iv] = make_int2(N, 0);
        if (iv == 0) s_nslice[it] = N;
        // Proof that the profile violates nothing, from the extrema of the polynomials over the continuous interval [0, t_last]
        // (a superset of the grid): |s_dd| (quadratic: ends + vertex), s_d (cubic: ends + the roots of s_dd), and s inside the
        // spline's range (s_d >= 0 makes s monotone: its ends suffice).  Margins of 1e-9 dwarf the rounding of the Horner
        // evaluations the point-by-point scan would do; anything not proven (including NaNs) falls back to that scan, so the
        // outcome is the scan's outcome either way.
        if (N > 0) {
            const double a2 = pSdd0() * 0.5, tl = (double)(N - 1) * tick;
            auto acc = [&](double t) { return fma(fma(12.0 * q.a4, t, 6.0 * q.a3), t, 2.0 * a2); };
            auto vel = [&](double t) { return fma(fma(fma(4.0 * q.a4, t, 3.0 * q.a3), t, 2.0 * a2), t, pSd0()); };
            const double qa = 12.0 * q.a4, qb = 6.0 * q.a3, qc = 2.0 * a2;  // s_dd = qa t^2 + qb t + qc
            double amax = fmax(fabs(acc(0.0)), fabs(acc(tl)));
            const double tv = -qb / (2.0 * qa);
            if (tv > 0.0 && tv < tl) amax = fmax(amax, fabs(acc(tv)));
            bool proven = amax <= p.max_accel * (1.0 - 1e-9) - 1e-12;
            double vmax = fmax(vel(0.0), vel(tl)), vmin = fmin(vel(0.0), vel(tl));
            const double disc = fma(qb, qb, -4.0 * qa * qc);
            if (disc >= 0.0) {  // stable quadratic roots; a root that is NaN / inf / outside (0, t_last) is not an interior extremum
                const double sq = sqrt(disc), qq = -0.5 * (qb + (qb >= 0.0 ? sq : -sq));
                const double r1 = qq / qa, r2 = qc / qq;
                if (r1 > 0.0 && r1 < tl) { vmax = fmax(vmax, vel(r1)); vmin = fmin(vmin, vel(r1)); }
                if (r2 > 0.0 && r2 < tl) { vmax = fmax(vmax, vel(r2)); vmin = fmin(vmin, vel(r2)); }
            } else if (!(disc < 0.0)) proven = false;  // NaN
            const double pad = amax * 1e-6;  // |s_d(t) - s_d(r)| <= max|s_dd| |t - r|: covers a root off by a microsecond
            proven = proven && vmax + pad <= p.max_speed * (1.0 - 1e-9) - 1e-12 && vmin - pad >= 0.0;
            const double s_end = fma(fma(fma(fma(q.a4, tl, q.a3), tl, a2), tl, pSd0()), tl, pS0());
            const double eps = 1e-9 * (1.0 + fabs(knot_last) + fabs(knot0));
            proven = proven && pS0() >= knot0 + eps && s_end < knot_last - eps;
            if (!proven) atomicOr((unsigned int*)&s_cnt[2], 1u << ((it - it_lo) & 31));
        }
    }
    // bound of |d(t)| over ALL slices and lateral samples (the once-per-ego group test needs it), by the wavefronts the solves above
    // leave idle.  In the Hermite form of the quintic with a resting end state,
    //   d(t) = d_end + (d0 - d_end) h0 + d_d0 T h1 + d_dd0 T^2 h2  (tau = t / T),  h0 = 1 - 10 tau^3 + 15 tau^4 - 6 tau^5 in [0, 1],
    //   |h1| = |tau - 6 tau^3 + 8 tau^4 - 3 tau^5| <= 16/81,  h2 = tau^2 (1 - tau)^3 / 2 <= 54/3125:
    //   |d| <= max(|d0|, |d_end|) + 0.1976 |d_d0| T + 0.01729 |d_dd0| T^2.
    // Wave maximum of the fp32 bit patterns (a NaN is the largest pattern and keeps every obstacle), ONE LDS atomic per wavefront -
    // same-address LDS atomics of many lanes serialise badly.
#if defined(FP_ABL_NO_DALL)
    if (false) {
#else
    if (n_obs > 0 && wave >= 1) {
#endif
        for (int e0 = (wave - 1) * kWave; e0 < n_it * nd; e0 += (kWaves - 1) * kWave) {
            const int e = e0 + lane;
            uint32_t bits = 0u;
            if (e < n_it * nd) {
                const int eq = div_by<ND, NT * ND + kWave>(e < n_it * nd ? e : 0, 1.0f / (float)nd);
                const double T = s_ts[it_lo + eq], de = s_ds[e - mul24(eq, nd)];
                const double d0e = pD0();
                double bd = (fabs(d0e) > fabs(de) ? fabs(d0e) : fabs(de)) + 0.1976 * fabs(pDd0()) * T + 0.01729 * fabs(pDdd0()) * T * T;
                if (!(d0e == d0e) || !(de == de)) bd = __builtin_nan("");
                bits = __float_as_uint(float_above(bd) * 1.0000005f);
            }
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) {
                const uint32_t o2 = (uint32_t)__shfl_xor((int)bits, off, kWave);
                bits = o2 > bits ? o2 : bits;
            }
            if (lane == 0 && bits) atomicMax((unsigned int*)&s_cnt[6], bits);
        }
    }
    if (tid == 0) { s_k[0] = knot0; s_k[1] = inv_bucket_w; s_k[2] = veh_hl; s_k[3] = veh_hw; s_k[4] = r_ego; }
    if (kEgoLds && !kEgoEarly && tid == 1) { s_k[5] = s0; s_k[6] = s_d0; s_k[7] = s_dd0; s_k[8] = d0; s_k[9] = d_d0; s_k[10] = d_dd0; }
    {   // power sums of every slice (they need nothing but the time samples): a few threads of the middle
        const int i = tid - kThreads / 2;
        if (i >= 0 && i < n_it) power_sums_closed(arange_len(s_ts[it_lo + i], tick), tick, s_pows + (it_lo + i) * 11);
    }
    __syncthreads();
    FP_STAMP(2);
    auto eS0 = [&]() -> double { if constexpr (kEgoLds) return s_k[5]; else return s0; };
    auto eSd0 = [&]() -> double { if constexpr (kEgoLds) return s_k[6]; else return s_d0; };
    auto eSdd0 = [&]() -> double { if constexpr (kEgoLds) return s_k[7]; else return s_dd0; };
    auto eD0 = [&]() -> double { if constexpr (kEgoLds) return s_k[8]; else return d0; };
    auto eDd0 = [&]() -> double { if constexpr (kEgoLds) return s_k[9]; else return d_d0; };
    auto eDdd0 = [&]() -> double { if constexpr (kEgoLds) return s_k[10]; else return d_dd0; };
    if (n_obs > 0 && pose_limit > 0) {
        // ---- arclength range of every checked pose row over the lon profiles of ALL slices of this workgroup: lane = (row, slice),
        // loop over the end-speed samples, then LDS atomic min / max on order-preserving fp32 bit patterns (relative to the first knot).
        // (The truncation index M is still being reduced in this phase: poses off the spline count too, the circle clamps the range.)
        const float inv_rows = 1.0f / (float)(rows > 0 ? rows : 1);
        for (int e = tid; e < mul24(n_it, rows); e += kThreads) {
            const int itl = div_small(e, inv_rows), r = e - mul24(itl, rows);
            const int it = it_lo + itl;
            const int k = mul24(r, stride);
            if (k >= pose_limit || k >= s_nslice[it]) continue;
            const double t = (double)k * tick;
            double lo = __builtin_inf(), hi = -__builtin_inf();
            bool bad = false;
            for (int iv = 0; iv < nv; ++iv) {
                const double a3 = s_qlon[2 * (mul24(it, nv) + iv)], a4 = s_qlon[2 * (mul24(it, nv) + iv) + 1];
                const double sv = fma(fma(fma(fma(a4, t, a3), t, eSdd0() * 0.5), t, eSd0()), t, eS0()) - knot0;
                bad = bad || !(sv == sv);
                lo = fmin(lo, sv); hi = fmax(hi, sv);
            }
            uint32_t* bx = (uint32_t*)&s_box[r];
            if (bad) { atomicMin(bx + 0, kOrdNegInf); atomicMax(bx + 1, kOrdPosInf); }  // NaN: keep everything
            else if (lo <= hi) { atomicMin(bx + 0, f32_ordered((float)lo)); atomicMax(bx + 1, f32_ordered((float)hi)); }
        }
    }
    // [section MASKS]
#if defined(FP_ABL_NO_SCAN)  // timing ablation: no point-by-point mask scan (results are wrong)
    const uint32_t scan_mask = 0u;
#else
    const uint32_t scan_mask = (uint32_t)s_cnt[2];
#endif
    if (tid == 0) FP_COUNT(9, __popc(scan_mask));
    for (int it = it_lo; it < it_hi; ++it) {
        if (!((scan_mask >> ((it - it_lo) & 31)) & 1u)) continue;  // every lon profile of the slice is proven clean
        const int N = arange_len(s_ts[it], tick);
        const float inv_n = 1.0f / (float)N;
        for (int e = tid; e < nv * N; e += kThreads) {
            const int iv = div_small(e, inv_n), i = e - mul24(iv, N);
            const double a3 = s_qlon[2 * (mul24(it, nv) + iv)], a4 = s_qlon[2 * (mul24(it, nv) + iv) + 1];
            const double a2 = eSdd0() * 0.5;
            const double t = (double)i * tick;
            const double s = fma(fma(fma(fma(a4, t, a3), t, a2), t, eSd0()), t, eS0());
            const double s_d = fma(fma(fma(4.0 * a4, t, 3.0 * a3), t, 2.0 * a2), t, eSd0());
            const double s_dd = fma(fma(12.0 * a4, t, 6.0 * a3), t, 2.0 * a2);
            const uint32_t bad = (s_d > p.max_speed ? FP_FLAG_SPEED : 0u) | (fabs(s_dd) > p.max_accel ? FP_FLAG_ACCEL : 0u);
            if (bad) atomicOr((unsigned int*)&s_lon_meta[mul24(it, nv) + iv].y, bad);
            if (!(s >= knot0) || !(s < knot_last)) atomicMin(&s_lon_meta[mul24(it, nv) + iv].x, i);  // calc_position -> None (cubic_spline.py:56-59)
        }
    }
    // [/section MASKS]
    for (int e0 = (kWaves - 1 - wave) * kWave; e0 < n_it * (nv + nd); e0 += kThreads) {  // (the last wavefronts: the ranges above keep the first busy)  // (whole wavefronts: the row maxima below are wave reductions)
        const int e = e0 + lane;
        const bool live = e < n_it * (nv + nd);
        const int eq = live ? div_by<NV + ND, NT * (NV + ND)>(e, 1.0f / (float)(nv + nd)) : 0;
        const int it = it_lo + eq, sub = live ? e - mul24(eq, nv + nd) : 0;
        const double T = s_ts[it];
        const double* S = s_pows + it * 11;
        if (!live) {
        } else if (sub < nv) {
            const Quartic q{eS0(), eSd0(), eSdd0() * 0.5, s_qlon[2 * (mul24(it, nv) + sub)], s_qlon[2 * (mul24(it, nv) + sub) + 1]};
            double lon[3];
            lon_cost_sums(q, target_speed, S, lon);
            double* o = s_lon_sum + 3 * (mul24(it, nv) + sub);
            o[0] = lon[0]; o[1] = lon[1]; o[2] = lon[2];
        } else {
            const int id = sub - nv;
            const Quintic q = quintic_bvp(eD0(), eDd0(), eDdd0(), s_ds[id], 0.0, 0.0, T);
            double lat[3];
            lat_cost_sums(q, S, lat);
            double* o = s_lat_sum + 3 * (mul24(id, nt) + it);
            o[0] = lat[0]; o[1] = lat[1]; o[2] = lat[2];
        }
    }
    // lat polynomial coefficients of one slice (a boundary-value solve costs two divisions): nd threads per slice instead of once
    // per trajectory point
    const float inv_nd_g = 1.0f / (float)nd;
    auto fill_group_lat = [&](int it0) {  // the group of slices that starts at it0: [slice in group][lateral sample]
        if constexpr (kWalk) {  // every slice of this workgroup, [slice][lateral sample] (absolute slice index)
            for (int e = kThreads - 1 - tid; e < mul24(n_it, nd); e += kThreads) {
                const int itl = div_small(e, inv_nd_g), id = e - mul24(itl, nd);
                const Quintic q = quintic_bvp(eD0(), eDd0(), eDdd0(), s_ds[id], 0.0, 0.0, s_ts[it_lo + itl]);
                double* o = s_qlat + 3 * (mul24(it_lo + itl, nd) + id);
                o[0] = q.a3; o[1] = q.a4; o[2] = q.a5;
            }
            return;
        }
        const int idg = kThreads - 1 - tid;  // the last threads: they have the least prep work
        if (idg < mul24(gs, nd)) {
            const int itl = kGroup ? div_small(idg, inv_nd_g) : 0, id = idg - mul24(itl, nd);
            if (it0 + itl < it_hi) {
                const Quintic q = quintic_bvp(eD0(), eDd0(), eDdd0(), s_ds[id], 0.0, 0.0, s_ts[it0 + itl]);
                s_qlat[3 * idg] = q.a3; s_qlat[3 * idg + 1] = q.a4; s_qlat[3 * idg + 2] = q.a5;
            }
        }
    };
    fill_group_lat(it_lo);
    __syncthreads();
    FP_STAMP(3);
    if constexpr (kWalk) {
        // ---- lateral fan bounds of EVERY slice (walk): lane = (slice, point inside the collision horizon): fan half-width max|d| and
        // largest lateral step max|d(i + 1) - d(i)| over the lateral samples, float_above: never below the fp64 value.  For a fixed
        // horizon the quintic is AFFINE in its end offset (Hermite form, see the lateral bound above): d(t; d_end) = b(t) + d_end g(t),
        // so |d| and |d(i + 1) - d(i)| are convex in d_end and their maxima over the samples sit at the smallest and the largest
        // sample - two profiles are evaluated instead of nd (the rounding of the other samples' own evaluations, ~1e-16 relative,
        // is far inside float_above's 2^-22).  (The first threads: the circles below take the last ones.)
        if (n_obs > 0 && hp > 0) {
            int id_lo = 0, id_hi = 0;  // (any order of d_samples: the reference passes np.linspace, the ABI does not require it)
            for (int id = 1; id < nd; ++id) {
                id_lo = s_ds[id] < s_ds[id_lo] ? id : id_lo;
                id_hi = s_ds[id] > s_ds[id_hi] ? id : id_hi;
            }
            bool d_nan = false;
            for (int id = 0; id < nd; ++id) d_nan = d_nan || !(s_ds[id] == s_ds[id]);
            const float inv_hp = 1.0f / (float)hp;
            for (int e = tid; e < mul24(n_it, hp); e += kThreads) {
                const int itl = div_small(e, inv_hp), i = e - mul24(itl, hp);
                const int it = it_lo + itl;
                const int np_i = hp < s_nslice[it] ? hp : s_nslice[it];
                if (i >= np_i) continue;
                const double t = (double)i * tick, tn = (double)(i + 1) * tick;
                float dm = 0.0f, ddm = 0.0f;
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const double* ql = s_qlat + 3 * (mul24(it, nd) + (side ? id_hi : id_lo));
                    const double a3 = ql[0], a4 = ql[1], a5 = ql[2];
                    const double d = fma(fma(fma(fma(fma(a5, t, a4), t, a3), t, eDdd0() * 0.5), t, eDd0()), t, eD0());
                    const double dn = fma(fma(fma(fma(fma(a5, tn, a4), tn, a3), tn, eDdd0() * 0.5), tn, eDd0()), tn, eD0());
                    // (a NaN offset poisons the bound: as an unsigned bit pattern NaN is the largest)
                    const float fd = float_above(fabs(d)), fdd = float_above(fabs(dn - d));
                    dm = __float_as_uint(fd) > __float_as_uint(dm) ? fd : dm;
                    ddm = __float_as_uint(fdd) > __float_as_uint(ddm) ? fdd : ddm;
                }
                if (d_nan) dm = ddm = __builtin_nanf("");  // a NaN sample: its own profile is NaN everywhere -> every pair passes
                s_fan_d[mul24(it, hp_max) + i] = FanType<kSlim>::above(dm);
                s_fan_dd[mul24(it, hp_max) + i] = i + 1 < np_i ? FanType<kSlim>::above(ddm) : (fan_t)0.0f;
            }
        }
    }
    // ---- ranges -> circles: every reference point of the row lies within  v_max * (half the range)  of the line's point at
    // the middle of the range (v_max >= |P'(s)| everywhere); + ego reach + the largest lateral offset.  One test per
    // (row, obstacle) item then prunes the item for every profile of every slice at once.
    if (n_obs > 0 && kThreads - 1 - nd - tid >= 0 && kThreads - 1 - nd - tid < rows) {  // (threads next to fill_group_lat's first)
        const int r = kThreads - 1 - nd - tid;
        const uint4 bx = s_box[r];
        const double knot0 = s_knots[0], knot_last = s_knots[nx - 1];  // (fresh reads: the prologue's copies need not live this long)
        const float lo_f = f32_from_ordered(bx.x), hi_f = f32_from_ordered(bx.y);
        const double v_max = (double)__uint_as_float((uint32_t)s_cnt[5]), d_all = (double)__uint_as_float((uint32_t)s_cnt[6]);
        // empty row (no valid pose): radius -1 rejects every obstacle; an infinite range or a NaN bound keeps every obstacle
        double rad = -1.0, cx = sp.coef[w0], cy = sp.coef[4 * ld + w0];  // a knot: fallback centre of an empty row circle
        if (hi_f >= lo_f) {
            // the ends were rounded to nearest fp32: half an ulp each, covered by slack; points off the spline do not exist
            const double slack = ((double)fabsf(lo_f) + (double)fabsf(hi_f)) * 1.2e-7 + 1e-6;
            double s_lo = knot0 + (double)lo_f - slack, s_hi = knot0 + (double)hi_f + slack;
            s_lo = fmax(s_lo, knot0); s_hi = fmin(s_hi, knot_last);
            if (!(s_hi >= s_lo)) s_hi = s_lo = fmin(fmax(knot0 + (double)lo_f, knot0), knot_last);  // (a range that only grazes the end)
            double s_mid = 0.5 * (s_lo + s_hi);
            if (!(s_mid < knot_last)) s_mid = knot0 + 0.5 * (knot_last - knot0);  // infinite range: any centre will do
            const int seg = lut_segment(s_knots, s_lut, nx, s_mid, s_k[0], s_k[1], n_buckets);
            double tx, ty;
            frame_at(seg, s_mid - s_knots[seg], cx, cy, tx, ty);
            rad = (v_max * (0.5 * (s_hi - s_lo)) * (1.0 + 1e-9) + s_k[4] + d_all) * (1.0 + 1e-9) + 1e-6;
            if (!(hi_f <= 3.0e38f) || !(lo_f >= -3.0e38f)) rad = __builtin_inf();
            // (coefficient window: v_max bounds |P'| over the window's segments only - a range that leaves them keeps every obstacle)
            if (windowed && (!(s_lo >= s_knots[w0]) || !(s_hi <= s_knots[w0 + wcap < nx - 1 ? w0 + wcap : nx - 1]))) rad = __builtin_inf();
        }
        s_grp[r].hl = cx;
        s_grp[r].hw = cy;
        s_grp[r].r = rad;
    }
    __syncthreads();  // the final assembly reads the sums (with no obstacles there is no other barrier in between)
    FP_STAMP(4);

    // ---------------------------------------------------------------- collision
    // Once per ego:   G  lane = (row, obstacle) item against the circle that encloses the row's reference points of ALL lon
    //                    profiles of ALL slices of this workgroup (+ ego reach + the largest lateral offset)     -> s_items
    // Per slice i_T:  phase A  frames + lateral offsets;  prep  fan half-widths
    //                 B  lane = (surviving item, lon profile): fattened circle around the reference point + separating axes
    //                    n_k and t_k of the lateral fan                                                          -> s_hits
    //                 N  lane = (hit, lateral sample): exact circle + 4-axis separating-axis test                -> s_coll
    // The time horizon T moves a profile's first seconds only a little (the end speed spreads them by tens of metres), so one
    // group test per ego keeps about as few items as one per slice did - at a seventh of the work and one barrier less per slice.
    // Lists are appended with one LDS atomic per wavefront (ballot + popcount); the two counters only ever grow, each stage works
    // on [base, counter) and every thread tracks the bases itself, so nothing is reset between stages.  Capacity: the item list is
    // filled optimistically by the whole table; if the survivors do not fit (rare) the table is cut into chunks of kItemCap items
    // and the slice loop runs once per chunk; the pair range of B is cut into passes of kHitCap pairs.
#if defined(FP_ABL_NO_COLL)
    const bool collide = false;
#else
    const bool collide = n_obs > 0 && hp > 0;
#endif
    if (collide) {
#if defined(FP_ABL_NO_GBN)   // timing ablations (tools/variants.sh): results are wrong, only the clock is read
        const int n_items = 0;
#else
        const int n_items = rows * n_obs;
#endif
        const float inv_nobs = 1.0f / (float)n_obs, inv_nvf = 1.0f / (float)nv, inv_ndf = 1.0f / (float)nd;
        int chunk = n_items;
        for (int i0 = 0; i0 < n_items;) {
            // ---- G (once per ego and item chunk): the poses come straight from the scene table (L2 for all but the first ego of a
            // scene), kPoseFlight reads in flight per lane.  x = NaN / valid = 0: no state at this step (state_at_time -> None).
            // A survivor's pose goes to the list with its orientation still an angle.
            const int i1 = i0 + chunk < n_items ? i0 + chunk : n_items;
            for (int b0 = i0; b0 < i1; b0 += kPoseFlight * kThreads) {
                double4 ps[kPoseFlight];
                fetch_poses(b0, ps);
#pragma unroll
                for (int u = 0; u < kPoseFlight; ++u) {
                    const int e = b0 + u * kThreads + tid;
                    bool keep = false;
                    if (e < i1) {
                        const int r = div_by<NOBS, ROWS * NOBS>(e, inv_nobs), j = e - mul24(r, n_obs);
                        const int k = mul24(r, stride);
                        const ObsDim g = s_grp[r];
                        const double dx = ps[u].x - g.hl, dy = ps[u].y - g.hw, R = g.r + s_dim.rad[j];
                        keep = k < pose_limit && ps[u].w != 0.0 && (ps[u].x == ps[u].x) && !(g.r < 0.0) && !(fma(dx, dx, dy * dy) > R * R);
                    }
                    const unsigned long long m = __ballot(keep);
                    if (m) {
                        const int first = __ffsll((long long)m) - 1;
                        int base = 0;
                        if (lane == first) base = atomicAdd(&s_cnt[0], __popcll(m));
                        const int pos = __builtin_amdgcn_readlane(base, first) - item_base + __popcll(m & ((1ull << lane) - 1ull));
                        if (keep && pos < kItemCap) {
                            s_items[pos] = (unsigned short)e;
                            s_spose.set(pos, ps[u].x, ps[u].y, ps[u].z, 0.0);
                        }
                    }
                }
            }
            __syncthreads();
    FP_STAMP(7);
            const int item_end = s_cnt[0];
            const int n_surv = item_end - item_base;
            item_base = item_end;
            if (tid == 0) FP_COUNT(0, n_surv);
            if (n_surv > kItemCap) {  // block-uniform: does not fit, redo [i0, ...) in chunks whose survivors always fit
                chunk = kItemCap;
                __syncthreads();  // every thread has read s_cnt[0] before the redone pass adds to it again
                continue;
            }
            // the survivors' orientations -> (cos, sin), with shapely's snap (frenet_device.h); every item belongs to one chunk
            for (int si = tid; si < n_surv; si += kThreads) {
                double sn, cs;
                sincos_snapped(s_spose.hi[si].x, sn, cs);
                s_spose.hi[si] = make_double2(cs, sn);
            }
            // (the slice loop's first barrier orders these writes before stage B reads them)
            // No survivor (block-uniform: empty surroundings, obstacles out of reach): nothing can collide, the slices are skipped.
            // (Skipping only the rows without a survivor was tried: a lane that skips its point saves nothing while its wavefront's
            // other lanes work, and the bookkeeping cost 3 % on dense scenes.)
            if constexpr (kWalk) {
                // ---- WALK: every wavefront takes lon profiles (slice, end speed) from a counter and handles each one on its own, in four
                // wave-local steps with no workgroup barrier between them (a wavefront's LDS accesses execute in program order):
                //   frames   lane = trajectory point inside the collision horizon: reference-line frame -> the wavefront's frame table
                //   prep     lane = checked pose row: the fan's half-width along the reference normal (fp32, rounded outwards)
                //   B        lane = surviving item, in list (= time) order, 64 at a time: fattened circle + separating axes n_k, t_k;
                //            the passing items are compacted into the wavefront's hit list (ballot + popcount, no atomics)
                //   N        lane = (hit, lateral sample), floor(64 / nd) hits per round: exact circle + 4-axis separating-axis test.
                // The collision state of the profile's nd candidates is ONE wave-uniform bit mask: a lane skips candidates that have
                // collided, and the walk of a profile ENDS when all nd have (later items cannot change anything) - items are in time
                // order, so a blocked profile ends after its first hits instead of testing every pose of the horizon.
                if (n_surv > 0) {
                    __syncthreads();  // the survivors' (cos, sin), the item list and the lateral bounds are visible to every wavefront
                    const SplitTab<Frame> wfr = s_frames.at(mul24(wave, hp_max));
                    float* wwf = s_wfat + mul24(wave, rows_max > 0 ? rows_max : 1);
                    hit_t* wh = (hit_t*)s_hits + wave * kWave;
                    constexpr int kHpwC = ND > 0 ? kWave / ND : 1;
                    const int hpw = kShape ? kHpwC : kWave / nd;  // hits per narrow-phase round
                    const int hh_l = div_by<ND, kWave>(lane, inv_ndf), id_l = lane - mul24(hh_l, nd);
                    const unsigned long long full = nd >= 64 ? ~0ull : ((1ull << nd) - 1ull);
                    const unsigned long long lt_mask = (1ull << lane) - 1ull;
                    const int n_prof = mul24(n_it, nv);
                    const float inv_stride = 1.0f / (float)stride;
                    // Profiles are dealt to the wavefronts round-robin (wave w takes w, w + 8, ...: a mix of end speeds each); the first
                    // wavefront to run out waits ~3 of the walk's 24 us for the last one.  Dealing them from a shared LDS counter instead was
                    // built and measured: no difference (153.1 / 154.4 against 155.1 / 154.1 us per step in same-box runs) - the waiting
                    // wavefronts cost no issue slots and the CU's other workgroups fill them.  (Its C++ spelling, `if (lane == 0) t =
                    // atomicAdd(...)` + readfirstlane inside a loop whose only exit is a break, was restructured by the compiler so that lanes
                    // 1..63 spun on their own - a hang; it took an inline-asm block with EXEC narrowed by hand.)  The loop below has a
                    // wave-uniform scalar trip count and no divergent branch around a cross-lane operation.
                    for (int pq = __builtin_amdgcn_readfirstlane(wave); pq < n_prof; pq += kWaves) {
                        const int itl = div_by<NV, NT * NV>(pq, inv_nvf), iv = pq - mul24(itl, nv);
                        const int it = it_lo + itl, qg = mul24(it, nv) + iv;
                        const int M = __builtin_amdgcn_readfirstlane(s_lon_meta[qg].x);  // (M <= N; wave-uniform -> scalar registers)
                        // a trajectory of fewer than two points has no heading: no pair (M == 1 is the assembly's business)
                        if (M < 2) continue;
                        const int n_pts = __builtin_amdgcn_readfirstlane(s_nslice[it]);
                        const int np = hp < n_pts ? hp : n_pts;
                        const int npm = np < M ? np : M;
                        const double a3 = s_qlon[2 * qg], a4 = s_qlon[2 * qg + 1];
    // [section FRAMES]
                        for (int i = lane; i < npm; i += kWave) {
                            const double t = (double)i * tick;
                            const double s = fma(fma(fma(fma(a4, t, a3), t, eSdd0() * 0.5), t, eSd0()), t, eS0());
                            const int seg = lut_segment(s_knots, s_lut, nx, s, s_k[0], s_k[1], n_buckets);
                            Frame fr;
                            frame_at(seg, s - s_knots[seg], fr.px, fr.py, fr.tx, fr.ty);
                            wfr.set(i, fr.px, fr.py, fr.tx, fr.ty);
                        }
    // [/section FRAMES]
                        wave_lds_sync();
                        const fan_t* dm = s_fan_d + mul24(it, hp_max);
                        const fan_t* ddm = s_fan_dd + mul24(it, hp_max);
                        {   // prep (see the slice loop's comment for the bound): conservative, fp32
                            const float r_ego_f = float_above(s_k[4]), hl_f = float_above(s_k[2]), hw_f = float_above(s_k[3]);
    // [section PREP]
                            for (int r = lane; r < rows; r += kWave) {
                                const int k = mul24(r, stride);
                                const bool row_ok = k < n_pts && k < hp;
                                float wl = r_ego_f;
                                if (k + 1 < M && k + 1 < hp) {
                                    const Frame f0 = wfr.get(k), f1 = wfr.get(k + 1);
                                    const double dpx = f1.px - f0.px, dpy = f1.py - f0.py;
                                    const double a_n = fma(dpy, f0.tx, -dpx * f0.ty);     // dP . n_k,  n_k = (-ty, tx)
                                    const double a_t = fma(dpx, f0.tx, dpy * f0.ty);      // dP . t_k
                                    const double nn = fma(f1.tx, f0.tx, f1.ty * f0.ty);   // n_{k+1} . n_k
                                    const double nt_ = fma(f1.tx, f0.ty, -f1.ty * f0.tx); // n_{k+1} . t_k
                                    const double dm1 = (double)dm[k + 1];
                                    const float num = (float)(fabs(a_n) + (double)ddm[k] + dm1 * fabs(1.0 - nn));
                                    const float den = (float)(fabs(a_t) - dm1 * fabs(nt_));
                                    if (den > 1e-30f) {  // v_rcp_f32: 1 ulp; the factor covers it and the two roundings to nearest
                                        const float sigma = vmin_f32(1.0f, num * __builtin_amdgcn_rcpf(den) * (1.0f + 4e-6f) + 1e-9f);
                                        wl = vmin_f32(r_ego_f, (hl_f * sigma + hw_f) * (1.0f + 1e-6f));
                                    }
                                }
                                wwf[r] = row_ok ? ((float)dm[k] + wl) * (1.0f + 1e-6f) + 1e-9f : 0.0f;
                            }
    // [/section PREP]
                        }
                        wave_lds_sync();
                        // candidates of this profile that collided in an earlier item chunk (crowded scenes only; else 0)
                        unsigned long long coll = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_collmask[2 * qg + 1]) << 32) |
                                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)s_collmask[2 * qg]);
                        const double* qlat = s_qlat + 3 * mul24(it, nd);
#if defined(FP_ABL_NO_BN)
                        const int n_walk = 0;
#else
                        const int n_walk = n_surv;
#endif
                        for (int c0 = 0; c0 < n_walk && coll != full; c0 += kWave) {
                            const int si = c0 + lane;
                            bool pass = false;
                            uint32_t code = 0;
                            if (si < n_walk) {
                                const int item = s_items[si];
                                const int r = div_by<NOBS, ROWS * NOBS>(item, inv_nobs), j = item - mul24(r, n_obs);
                                const int k = mul24(r, stride);
                                const ObsPose op = s_spose.get(si);
                                const ObsDim od = s_dim.get(j);
                                const Frame fr = wfr.get(k);  // (stale beyond the profile's M: the test below excludes those poses)
                                const double r_ego_b = s_k[4];
                                const double fat = (r_ego_b + od.r + (double)dm[k]) * (1.0 + 1e-12);
                                const double dx = op.x - fr.px, dy = op.y - fr.py;
                                // (1) circle around the reference point; (2) separating axis n_k: lateral offset of the obstacle centre
                                // vs the fan half-width + the obstacle's own reach along n_k; (3) separating axis t_k: every ego centre
                                // of the fan lies ON the normal line, so along t_k the fan reaches no further than the ego box's
                                // half-diagonal.  A NaN pose passes all three (-> "collision").
                                const double w = fma(dy, fr.tx, -dx * fr.ty), u = fma(dx, fr.tx, dy * fr.ty);
                                const double a_n = fabs(fma(op.s, fr.tx, -op.c * fr.ty)), a_t = fabs(fma(op.c, fr.tx, op.s * fr.ty));
                                const double reach = fma(od.hl, a_n, od.hw * a_t), reach_t = fma(od.hl, a_t, od.hw * a_n);
                                pass = k < M && !(fma(dx, dx, dy * dy) > fat * fat) && !(fabs(w) > (double)wwf[r] + reach) &&
                                       !(fabs(u) > r_ego_b * (1.0 + 1e-12) + reach_t);
                                code = (uint32_t)k | ((uint32_t)si << 8);  // k < 128, si < 512 (slim: si < 256, 16 bits)
                            }
                            const unsigned long long m = __ballot(pass);
                            if (lane == 0) { FP_COUNT(1, 1); FP_COUNT(2, __popcll(m)); }
                            if (!m) continue;
                            const int n_hits = __popcll(m);
                            if (pass) wh[__popcll(m & lt_mask)] = (hit_t)code;
                            wave_lds_sync();
                            // ---- N: exact narrow phase, hpw hits x nd lateral samples per round
#if defined(FP_ABL_NO_N)
                            const int n_exact = 0;
#else
                            const int n_exact = n_hits;
#endif
                            for (int h0 = 0; h0 < n_exact; h0 += hpw) {
                                const int h = h0 + hh_l;
                                bool hit = false;
                                if (hh_l < hpw && h < n_exact && !((coll >> id_l) & 1ull)) {
                                    const uint32_t hc = wh[h];
                                    const int k = hc & 0xFF, si2 = hc >> 8;
                                    const int j = (int)s_items[si2] - mul24(div_by<STRIDE, FP_MAX_POINTS>(k, inv_stride), n_obs);  // item = row * n_obs + obstacle
                                    FP_COUNT(3, 1);
                                    // heading of pose k: forward difference, or the previous one for the last point (:127-129)
                                    const int ka_ = (k + 1 < M) ? k : k - 1;
                                    const Frame f0 = wfr.get(ka_), f1 = wfr.get(ka_ + 1);
                                    const double* ql = qlat + mul24(id_l, 3);
                                    const double b3 = ql[0], b4 = ql[1], b5 = ql[2];
                                    const double ta = (double)ka_ * tick, tb = (double)(ka_ + 1) * tick;
                                    const double da = fma(fma(fma(fma(fma(b5, ta, b4), ta, b3), ta, eDdd0() * 0.5), ta, eDd0()), ta, eD0());
                                    const double db = fma(fma(fma(fma(fma(b5, tb, b4), tb, b3), tb, eDdd0() * 0.5), tb, eDd0()), tb, eD0());
                                    double xa, ya, xb, yb;
                                    frenet_to_cartesian(f0.px, f0.py, f0.tx, f0.ty, da, xa, ya);
                                    frenet_to_cartesian(f1.px, f1.py, f1.tx, f1.ty, db, xb, yb);
                                    Obb ego;
                                    step_heading(xb - xa, yb - ya, ego.c, ego.s);
                                    ego.x = (ka_ == k) ? xa : xb;
                                    ego.y = (ka_ == k) ? ya : yb;
                                    ego.hl = s_k[2];
                                    ego.hw = s_k[3];
                                    const ObsPose op = s_spose.get(si2);
                                    const ObsDim od = s_dim.get(j);
                                    if (!(ego.x == ego.x) || !(ego.y == ego.y) || !(ego.c == ego.c)) {
                                        hit = true;  // polygon construction fails in the reference -> collision (:178-182)
                                    } else {
                                        const double R = (s_k[4] + od.r) * (1.0 + 1e-12);
                                        const double dx = op.x - ego.x, dy = op.y - ego.y;
                                        hit = fma(dx, dx, dy * dy) <= R * R && obb_overlap(ego, Obb{op.x, op.y, op.c, op.s, od.hl, od.hw});
                                        if constexpr (POLY) {
                                            if (hit) {  // a polygon column: its box was a necessary condition, the ring decides
                                                const int nvx = ((const int32_t*)(smem + L.nvert))[j];
                                                if (nvx > 0) hit = ring_overlap(ego, Obb{op.x, op.y, op.c, op.s, od.hl, od.hw}, ring_base + (ring_col0 + j) * 2 * (size_t)bt.poly_stride, nvx, s_dim.rin[j]);
                                            }
                                        }
                                    }
                                    if (hit) FP_COUNT(5, 1);
                                }
                                if (lane == 0) FP_COUNT(6, 1);
                                unsigned long long hm = __ballot(hit);
                                // fold the round's hits onto the nd lateral samples (scalar arithmetic on the ballot)
                                if (nd >= kWave) coll |= hm;
                                else
                                    for (; hm; hm >>= nd) coll |= hm & full;
                                if (coll == full) {
#if defined(FP_COUNTERS)  // when a blocked profile dies: the pose index of the last hit of the deciding round, in bins of 8 steps (slots 10..15)
                                    if (lane == 0) {
                                        FP_COUNT(7, 1);
                                        const int hl = (h0 + hpw < n_exact ? h0 + hpw : n_exact) - 1;
                                        const int kd = (int)(wh[hl] & 0xFF) >> 3;
                                        FP_COUNT(10 + (kd > 5 ? 5 : kd), 1);
                                    }
#endif
                                    break;
                                }
                            }
                        }
                        if (lane == 0) { s_collmask[2 * qg] = (uint32_t)coll; s_collmask[2 * qg + 1] = (uint32_t)(coll >> 32); }
                    }
#if defined(FP_PHASE_STAMPS)  // when the first and the last wavefront finished their profiles (columns 11, 12): the walk's imbalance
                    if (lane == 0) { atomicMin(&s_walk_t[0], (unsigned int)(wall_clock64() - t_begin)); atomicMax(&s_walk_t[1], (unsigned int)(wall_clock64() - t_begin)); }
#endif
                    // every wavefront's collision masks are visible to the assembly (and every wavefront is done with this chunk's item
                    // list before G overwrites it: crowded scenes walk the profiles again over the next chunk's survivors)
                    __syncthreads();
#if defined(FP_PHASE_STAMPS)
                    if (threadIdx.x == 0 && ka.r.best_traj && (perm || blockIdx.x % nsplit == 0)) {
                        double* row = ka.r.best_traj + ((size_t)(perm ? perm[blockIdx.x] : blockIdx.x / nsplit) * FP_ARR_COUNT + 15) * (ka.r.traj_stride > 0 ? ka.r.traj_stride : FP_DEFAULT_STRIDE) + 112;
                        row[11] = (double)s_walk_t[0]; row[12] = (double)s_walk_t[1];
                    }
#endif
                }
                i0 = i1;
                continue;
            }
            int par = 0;  // which of the two fan-bound buffers this group fills (the other one is zeroed meanwhile)
            for (int it0 = it_lo; n_surv > 0 && it0 < it_hi; it0 += gs) {
                // ---- a GROUP of g slices (g = 1 in the throughput instances): profile index q = (slice in group) * nv + iv and
                // idg = (slice in group) * nd + id; the lon tables ([nt][nv]) are contiguous over a group, so q indexes them from it0 * nv
                const int g = kGroup ? (it_hi - it0 < gs ? it_hi - it0 : gs) : 1;
                const int nvg = mul24(g, nv), ndg = mul24(g, nd);
                const int q0 = mul24(it0, nv);
                float* s_dmax = s_dmax2 + mul24(par, mul24(gs, hp_max));    // [g][hp_max] (zeroed during the previous group)
                float* s_ddmax = s_ddmax2 + mul24(par, mul24(gs, hp_max));
                // ---------------------------------------------------- phase A (per group): one lane per (profile, point)
                // reference-line frames of the points the collision horizon can touch (i < hp), lateral offsets, fan bounds
                int n_max = s_nslice[it0];  // points per slice, len(np.arange(0, T, tick)); the longest slice of the group sizes the passes
                if constexpr (kGroup)
                    for (int j = 1; j < g; ++j) n_max = s_nslice[it0 + j] > n_max ? s_nslice[it0 + j] : n_max;
                const int np = hp < n_max ? hp : n_max;
                const float inv_np = 1.0f / (float)np;
    // [section FRAMES]
                for (int e = tid; e < mul24(nvg, np); e += kThreads) {
                    const int q = div_small(e, inv_np), i = e - mul24(q, np);
                    const int M = s_lon_meta[q0 + q].x;
                    if (i < M) {  // the point is on the spline (M <= the slice's N)
                        const Quartic ql{eS0(), eSd0(), eSdd0() * 0.5, s_qlon[2 * (q0 + q)], s_qlon[2 * (q0 + q) + 1]};
                        const double t = (double)i * tick;
                        const double s = fma(fma(fma(fma(ql.a4, t, ql.a3), t, ql.a2), t, ql.a1), t, ql.a0);
                        const int seg = lut_segment(s_knots, s_lut, nx, s, s_k[0], s_k[1], n_buckets);
                        Frame fr;
                        frame_at(seg, s - s_knots[seg], fr.px, fr.py, fr.tx, fr.ty);
                        s_frames.set(mul24(q, hp_max) + i, fr.px, fr.py, fr.tx, fr.ty);
                    }
                }
    // [/section FRAMES]
    // [section LAT]
                for (int e = tid; e < mul24(ndg, np); e += kThreads) {
                    const int idg = div_small(e, inv_np), i = e - mul24(idg, np);
                    const int itl = kGroup ? div_small(idg, inv_nd_g) : 0;
                    int np_i = np;  // this slice's points inside the horizon
                    if constexpr (kGroup) {
                        np_i = s_nslice[it0 + itl];
                        np_i = hp < np_i ? hp : np_i;
                        if (i >= np_i) continue;
                    }
                    const double* ql = s_qlat + mul24(idg, 3);
                    const Quintic q{eD0(), eDd0(), eDdd0() * 0.5, ql[0], ql[1], ql[2]};
                    const double t = (double)i * tick;
                    const double d = fma(fma(fma(fma(fma(q.a5, t, q.a4), t, q.a3), t, q.a2), t, q.a1), t, q.a0);
                    s_lat[mul24(idg, hp_max) + i] = d;
                    // fan half-width max|d| and largest lateral step max|d(i+1) - d(i)| over the lateral samples: LDS atomic max on
                    // the bit patterns (non-negative floats order like unsigned integers); float_above: never below the fp64 value
                    atomicMax((unsigned int*)&s_dmax[mul24(itl, hp_max) + i], __float_as_uint(float_above(fabs(d))));
                    if (i + 1 < np_i) {
                        const double tn = (double)(i + 1) * tick;
                        const double dn = fma(fma(fma(fma(fma(q.a5, tn, q.a4), tn, q.a3), tn, q.a2), tn, q.a1), tn, q.a0);
                        atomicMax((unsigned int*)&s_ddmax[mul24(itl, hp_max) + i], __float_as_uint(float_above(fabs(dn - d))));
                    }
                }
            // [/section LAT]
                SLICE_SYNC();
                if (it0 == it_lo + (kGroup ? 0 : 3)) FP_STAMP(11);
                // ---- prep: wfat = lateral half-width of the whole fan along the reference normal n_k, per checked pose (row r, lon
                // profile q): every ego centre is P_k + d n_k with |d| <= dmax[k];  the ego box, whose heading deviates from the
                // reference tangent by alpha, reaches hw cos(alpha) + hl |sin(alpha)| <= min(r_ego, hw + hl sigma) along n_k, where
                // sigma >= |sin(alpha)| for every lateral sample follows from the heading vector
                //   h = (P_{k+1} - P_k) + d_{k+1} n_{k+1} - d_k n_k:  |h.n_k| <= |dP.n_k| + max|d_{k+1} - d_k| + max|d_{k+1}| |1 - n_{k+1}.n_k|,
                //   |h| >= |h.t_k| >= |dP.t_k| - max|d_{k+1}| |n_{k+1}.t_k|.   n_k then serves as a (conservative) separating axis.
                {
                    float* z_dmax = s_dmax2 + mul24(par ^ 1, mul24(gs, hp_max));   // zero the other buffer for the next group
                    float* z_ddmax = s_ddmax2 + mul24(par ^ 1, mul24(gs, hp_max));
                    for (int i = tid; i < mul24(gs, hp_max); i += kThreads) { z_dmax[i] = 0.0f; z_ddmax[i] = 0.0f; }
                    fill_group_lat(it0 + gs);  // phase A of this group is done with the table (barrier above)
                    // A conservative bound, so it is computed in fp32 (the ratio is a reciprocal instead of an fp64 division).
                    const float r_ego_f = float_above(s_k[4]), hl_f = float_above(s_k[2]), hw_f = float_above(s_k[3]);
                    const float inv_nv = 1.0f / (float)nv, inv_rows_p = 1.0f / (float)(rows > 0 ? rows : 1);
    // [section PREP]
                    for (int e = tid; e < mul24(g, mul24(rows, nv)); e += kThreads) {
                        // e = ((slice in group) * rows + r) * nv + iv
                        const int rr = div_by<NV, (NT > 0 ? NT : 1) * ROWS * NV>(e, inv_nv), iv = e - mul24(rr, nv);
                        const int itl = kGroup ? div_small(rr, inv_rows_p) : 0, r = rr - mul24(itl, rows);
                        const int q = mul24(itl, nv) + iv;
                        const int k = mul24(r, stride);
                        const bool row_ok = k < (kGroup ? s_nslice[it0 + itl] : n_max) && k < hp;
                        const int M = s_lon_meta[q0 + q].x;  // (M <= N)
                        const float* dm = s_dmax + mul24(itl, hp_max);
                        float wl = r_ego_f;
                        if (k + 1 < M && k + 1 < hp) {
                            const Frame f0 = s_frames.get(mul24(q, hp_max) + k), f1 = s_frames.get(mul24(q, hp_max) + k + 1);
                            const double dpx = f1.px - f0.px, dpy = f1.py - f0.py;
                            const double a_n = fma(dpy, f0.tx, -dpx * f0.ty);     // dP . n_k,  n_k = (-ty, tx)
                            const double a_t = fma(dpx, f0.tx, dpy * f0.ty);      // dP . t_k
                            const double nn = fma(f1.tx, f0.tx, f1.ty * f0.ty);   // n_{k+1} . n_k
                            const double nt_ = fma(f1.tx, f0.ty, -f1.ty * f0.tx); // n_{k+1} . t_k
                            const double dm1 = (double)dm[k + 1];
                            const float num = (float)(fabs(a_n) + (double)s_ddmax[mul24(itl, hp_max) + k] + dm1 * fabs(1.0 - nn));
                            const float den = (float)(fabs(a_t) - dm1 * fabs(nt_));
                            if (den > 1e-30f) {  // v_rcp_f32: 1 ulp; the factor covers it and the two roundings to nearest
                                const float sigma = vmin_f32(1.0f, num * __builtin_amdgcn_rcpf(den) * (1.0f + 4e-6f) + 1e-9f);
                                wl = vmin_f32(r_ego_f, (hl_f * sigma + hw_f) * (1.0f + 1e-6f));
                            }
                        }
                        s_wfat[mul24(q, hp_max) + k] = row_ok ? (dm[k] + wl) * (1.0f + 1e-6f) + 1e-9f : 0.0f;
                    }
        // [/section PREP]
                }
                SLICE_SYNC();
                if (it0 == it_lo + (kGroup ? 0 : 3)) FP_STAMP(12);
                const float inv_nvg = 1.0f / (float)nvg;
#if defined(FP_ABL_NO_BN)
                const int n_pairs = 0;
#else
                const int n_pairs = mul24(n_surv, nvg);
#endif
                for (int p0 = 0; p0 < n_pairs; p0 += kHitCap) {
                    const int p1 = p0 + kHitCap < n_pairs ? p0 + kHitCap : n_pairs;
                    for (int qq0 = p0 + wave * kWave; qq0 < p1; qq0 += kThreads) {
                        const int pr = qq0 + lane;
                        bool pass = false;
                        uint32_t code = 0;
                        if (pr < p1) {
                            const int si = kGroup ? div_small(pr, inv_nvg) : div_by<NV, kItemCapMax * NV>(pr, inv_nvf), q = pr - mul24(si, nvg);
                            const int itl = kGroup ? div_small(q, inv_nvf) : 0;
                            const int item = s_items[si];
                            const int r = div_by<NOBS, ROWS * NOBS>(item, inv_nobs), j = item - mul24(r, n_obs);
                            const int k = mul24(r, stride);
                            const ObsPose op = s_spose.get(si);
                            const ObsDim od = s_dim.get(j);
                            const Frame fr = s_frames.get(mul24(q, hp_max) + k);
                            // Poses beyond a profile's M hold stale frames, and a trajectory of fewer than two points has no heading: no
                            // pair (the narrow phase relies on it)
                            const int Mp = s_lon_meta[q0 + q].x;
                            const double r_ego_b = s_k[4];
                            const double fat = (r_ego_b + od.r + (double)s_dmax[mul24(itl, hp_max) + k]) * (1.0 + 1e-12);
                            const double dx = op.x - fr.px, dy = op.y - fr.py;
                            // (1) circle around the reference point; (2) separating axis n_k: lateral offset of the obstacle centre
                            // vs the fan half-width + the obstacle's own reach along n_k; (3) separating axis t_k: every ego centre
                            // of the fan lies ON the normal line, so along t_k the fan reaches no further than the ego box's
                            // half-diagonal.  A NaN pose passes all three (-> "collision").
                            const double w = fma(dy, fr.tx, -dx * fr.ty), u = fma(dx, fr.tx, dy * fr.ty);
                            const double a_n = fabs(fma(op.s, fr.tx, -op.c * fr.ty)), a_t = fabs(fma(op.c, fr.tx, op.s * fr.ty));
                            const double reach = fma(od.hl, a_n, od.hw * a_t), reach_t = fma(od.hl, a_t, od.hw * a_n);
                            pass = k < Mp && Mp >= 2 && !(fma(dx, dx, dy * dy) > fat * fat) && !(fabs(w) > (double)s_wfat[mul24(q, hp_max) + k] + reach) &&
                                   !(fabs(u) > r_ego_b * (1.0 + 1e-12) + reach_t);
                            code = (uint32_t)q | ((uint32_t)k << 8) | ((uint32_t)si << 16);  // q <= 255, k < 128, si < 512
                        }
                        const unsigned long long m = __ballot(pass);
                        if (lane == 0) { FP_COUNT(1, 1); FP_COUNT(2, __popcll(m)); }
                        if (m) {
                            const int first = __ffsll((long long)m) - 1;
                            int base = 0;
                            if (lane == first) base = atomicAdd(&s_cnt[1], __popcll(m));
                            const int pos = __builtin_amdgcn_readlane(base, first) - hit_base + __popcll(m & ((1ull << lane) - 1ull));
                            if (pass) s_hits[pos] = code;  // at most kHitCap pairs per pass: always fits
                        }
                    }
                    SLICE_SYNC();
                    if (it0 == it_lo + (kGroup ? 0 : 3)) FP_STAMP(13);
                    const int hit_end = s_cnt[1];
                    const int n_hits = hit_end - hit_base;
                    hit_base = hit_end;
                    // ---- N: exact narrow phase
#if defined(FP_ABL_NO_N)
                    const int n_exact = 0;
#else
                    const int n_exact = n_hits * nd;
#endif
                    for (int x = tid; x < n_exact; x += kThreads) {
                        const int h = div_by<ND, kHitCap * ND>(x, inv_ndf), id = x - mul24(h, nd);
                        // everything the filter needs rides in the hit word: one LDS read, then the candidate's collision byte
                        const uint32_t code = s_hits[h];
                        const int q = code & 0xFF, k = (code >> 8) & 0xFF, si = code >> 16;
                        const int itl = kGroup ? div_small(q, inv_nvf) : 0;
                        const int cand = mul24(mul24(id, nt), nv) + q0 + q;  // (id * nt + it) * nv + iv with it * nv + iv = q0 + q
                        FP_COUNT(3, 1);
                        if (s_coll[cand]) FP_COUNT(4, 1);
                        if (!s_coll[cand]) {
                            const int j = (int)s_items[si] - mul24(div_by<STRIDE, FP_MAX_POINTS>(k, 1.0f / (float)stride), n_obs);  // item = row * n_obs + obstacle, k = row * stride
                            const int M = s_lon_meta[q0 + q].x;
                            // heading of pose k: forward difference, or the previous one for the last point (:127-129)
                            const int ka_ = (k + 1 < M) ? k : k - 1;
                            const Frame f0 = s_frames.get(mul24(q, hp_max) + ka_), f1 = s_frames.get(mul24(q, hp_max) + ka_ + 1);
                            const double* lt = s_lat + mul24(mul24(itl, nd) + id, hp_max) + ka_;
                            const double da = lt[0], db = lt[1];
                            double xa, ya, xb, yb;
                            frenet_to_cartesian(f0.px, f0.py, f0.tx, f0.ty, da, xa, ya);
                            frenet_to_cartesian(f1.px, f1.py, f1.tx, f1.ty, db, xb, yb);
                            Obb ego;
                            step_heading(xb - xa, yb - ya, ego.c, ego.s);
                            ego.x = (ka_ == k) ? xa : xb;
                            ego.y = (ka_ == k) ? ya : yb;
                            ego.hl = s_k[2];
                            ego.hw = s_k[3];
                            const ObsPose op = s_spose.get(si);
                            const ObsDim od = s_dim.get(j);
                            bool hit;
                            if (!(ego.x == ego.x) || !(ego.y == ego.y) || !(ego.c == ego.c)) {
                                hit = true;  // polygon construction fails in the reference -> collision (:178-182)
                            } else {
                                const double R = (s_k[4] + od.r) * (1.0 + 1e-12);
                                const double dx = op.x - ego.x, dy = op.y - ego.y;
                                hit = fma(dx, dx, dy * dy) <= R * R && obb_overlap(ego, Obb{op.x, op.y, op.c, op.s, od.hl, od.hw});
                                        if constexpr (POLY) {
                                            if (hit) {  // a polygon column: its box was a necessary condition, the ring decides
                                                const int nvx = ((const int32_t*)(smem + L.nvert))[j];
                                                if (nvx > 0) hit = ring_overlap(ego, Obb{op.x, op.y, op.c, op.s, od.hl, od.hw}, ring_base + (ring_col0 + j) * 2 * (size_t)bt.poly_stride, nvx, s_dim.rin[j]);
                                            }
                                        }
                            }
                            if (hit) { s_coll[cand] = 1; FP_COUNT(5, 1); FP_COUNT(8 + (k >> 3), 1); }
                        }
                    }
                    if (p1 < n_pairs) SLICE_SYNC();  // the next pass overwrites the hit list
                }
                SLICE_SYNC();  // frames / lat / dmax are rewritten by the next group
                if (it0 == it_lo + (kGroup ? 0 : 3)) FP_STAMP(14);
                par ^= 1;
            }
            i0 = i1;
            if (i0 < n_items) {  // another item chunk: its slice loop starts over (rare: crowded scenes)
                for (int i = tid; i < 2 * mul24(gs, hp_max); i += kThreads) { s_dmax2[i] = 0.0f; s_ddmax2[i] = 0.0f; }
                fill_group_lat(it_lo);
                __syncthreads();
            }
        }
    }

    FP_STAMP(8);
#if defined(FP_COUNTERS)
    __syncthreads();
    if (tid < 16 && ka.r.best_traj) ka.r.best_traj[((size_t)b * FP_ARR_COUNT + 14) * (ka.r.traj_stride > 0 ? ka.r.traj_stride : FP_DEFAULT_STRIDE) + 112 + tid] = (double)s_dbg[tid];
#endif
    // ---------------------------------------------------------------- per-candidate assembly + argmin
    // From here on the kernel's arguments are read through `la`: for the four-per-CU instances (80 SGPRs) a view of the argument segment the
    // compiler cannot fold into the loads at the kernel's start - result pointers, cost weights and ticket buffers are then scalar loads HERE
    // instead of values carried through the walk in spilled SGPRs (each spill is a v_writelane / v_readlane pair: VALU issue slots); for
    // the other instances the same loads as before.
    int la_off = 0;
    if constexpr (OCC > 6) asm volatile("" : "+s"(la_off));
    const LatticeKernarg& la = *(const LatticeKernarg*)((const unsigned char*)__builtin_amdgcn_kernarg_segment_ptr() + la_off);
    const KernelArgs& kb = la.ka;
    const fp_params& pa = la.ka.p;
    Best mine{0.0, -1};
    const float inv_nv_a = 1.0f / (float)nv, inv_nt_a = 1.0f / (float)nt;
    // [section ASM]
    for (int c = tid; c < C; c += kThreads) {
        // c = (id * nt + it) * nv + iv
        const int q1 = div_by<NV, ND * NV * NT>(c, inv_nv_a), iv = c - mul24(q1, nv);
        const int id = div_by<NT, ND * NT>(q1, inv_nt_a), it = q1 - mul24(id, nt);
        if (it < it_lo || it >= it_hi) continue;  // another workgroup's slice (latency mode)
        const int N = s_nslice[it];
        const double* ls = s_lon_sum + mul24(3, mul24(it, nv) + iv);
        const double* ds = s_lat_sum + mul24(3, mul24(id, nt) + it);
        const int2 meta = s_lon_meta[mul24(it, nv) + iv];
        const int M = meta.x;
        uint32_t flags = (uint32_t)meta.y;
        bool hit;
        if constexpr (kWalk) hit = (s_collmask[2 * (mul24(it, nv) + iv) + (id >> 5)] >> (id & 31)) & 1u;
        else hit = s_coll[c] != 0;
        if (n_obs > 0 && M == 1 && horizon_cap >= 1) hit = true;  // traj.yaw is empty -> IndexError -> collision (:178-182)
        if (hit) flags |= FP_FLAG_COLLISION;
        if (M < N) flags |= FP_FLAG_TRUNCATED;
        if (kb.curv_tbl) flags |= kb.curv_tbl[(size_t)b * C + c];  // optional curvature checks, computed by curvature_flags_kernel
        double cost = combine_cost(pa, N, ls, ds);  // cost_function.py:41-50, same grouping as the reference
        uint32_t word = flags | ((uint32_t)N << FP_FLAG_N_SHIFT) | ((uint32_t)M << FP_FLAG_M_SHIFT);
        if (N <= 0 || N > points_cap(pa)) {  // a time sample the ABI's limits exclude (unvalidated in FP_MEM_DEVICE mode): no trajectory,
            cost = __builtin_nan("");        // exactly what the lane-per-candidate kernel reports (traj_eval)
            word = flags = FP_FLAG_SPEED | FP_FLAG_ACCEL | FP_FLAG_COLLISION;
        }
        if constexpr (FISS) {  // (read by the ego's search workgroup in THIS launch: agent-scope stores, see the template's comment)
            __hip_atomic_store((unsigned long long*)&kb.r.cost_tbl[(size_t)b * C + c], (unsigned long long)__double_as_longlong(cost), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&kb.r.flag_tbl[(size_t)b * C + c], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (kb.r.cost_tbl) kb.r.cost_tbl[(size_t)b * C + c] = cost;
            if (kb.r.flag_tbl) kb.r.flag_tbl[(size_t)b * C + c] = word;
        }
        if (!(flags & FP_FLAG_INFEASIBLE) && cost == cost) mine = best_merge(mine, Best{cost, c});
    }
    if constexpr (FISS) {  // every thread's table stores are acknowledged before the barrier below - and so before the ticket / the flag
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
    }
    // [/section ASM]
    mine = wave_best(mine);
    if (lane == 0) s_best[wave] = mine;
    __syncthreads();
    FP_STAMP(9);
    if (nsplit > 1) {
        // Latency mode: this workgroup holds the argmin of its slices.  The LAST workgroup of the ego to arrive (ticket counter;
        // the partial argmins travel as device-coherent atomic stores / loads) merges the partial argmins in part order - the result does not depend on
        // the arrival order - and carries on as the ego's only workgroup: results, epilogue.  No merge launch.
        if (tid == 0) {
            Best r = s_best[0];
            for (int w = 1; w < kWaves; ++w) r = best_merge(r, s_best[w]);
            // device-coherent (agent scope) stores, acknowledged before the ticket is taken: no whole-L2 write-back fence
            // Ordering: relaxed agent-scope atomics go to the device-coherent L2 in program order per wavefront; s_waitcnt(0) holds
            // the ticket back until both stores are acknowledged by L2, and the merging workgroup reads the partials with
            // agent-scope atomic loads (served by L2, never by a stale L1 / scalar cache line).  That is a hardware argument
            // (gfx950: one coherent L2 per agent for atomics with agent scope), not a C++ memory-model one - a release / acquire
            // pair would be the portable spelling, but on this chip it writes back the whole L2 per workgroup (measured: 15-35 %
            // of the launch).  The signal fences pin the COMPILER to the same order (no motion of the stores, the wait, the ticket
            // or the loads across each other).  A kernel that faults leaves the ticket counters non-zero, but a GPU fault ends
            // the process on this stack, and with it the ctx that owns the counters.
            Best* mine = la.part_best + blockIdx.x;
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __hip_atomic_store(&mine->cost, r.cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&mine->idx, r.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __builtin_amdgcn_s_waitcnt(0);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            const int ticket = __hip_atomic_fetch_add(&la.part_count[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            s_cnt[3] = ticket;
        }
        __syncthreads();
        if (s_cnt[3] != nsplit - 1) return;
        FP_STAMP_LAST(5);
        if (tid == 0) {
            la.part_count[b] = 0;  // ready for the next launch
            __atomic_signal_fence(__ATOMIC_SEQ_CST);  // the loads below stay behind the ticket
            Best* pb = la.part_best + ((size_t)blockIdx.x - part);  // the ego's first workgroup
            auto part = [&](int w) {
                return Best{__hip_atomic_load(&pb[w].cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                            __hip_atomic_load(&pb[w].idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
            };
            Best r = part(0);
            for (int w = 1; w < nsplit; ++w) r = best_merge(r, part(w));
            s_best[0] = r;
        }
        __syncthreads();
    } else if (tid == 0) {
        Best r = s_best[0];
        for (int w = 1; w < kWaves; ++w) r = best_merge(r, s_best[w]);
        s_best[0] = r;
    }
    if (tid == 0) {
        const Best r = s_best[0];
        // (the hand-over first: its s_waitcnt would otherwise also wait for the stores below, and best_idx / best_cost may be
        // device-mapped HOST memory - a link round trip in front of every flag)
        if constexpr (OCC > 4) {
            if (kb.epi_flag) {  // hand the ego to the appended epilogue workgroups: index first, acknowledged by L2, then the flag
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                __hip_atomic_store(&kb.idx_shadow[b], r.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                __builtin_amdgcn_s_waitcnt(0);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                __hip_atomic_store(&kb.epi_flag[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
            } else if (kb.idx_shadow) kb.idx_shadow[b] = r.idx;
        } else if (kb.idx_shadow) kb.idx_shadow[b] = r.idx;
#if defined(FP_TL)
        __hip_atomic_store((unsigned long long*)&kb.r.best_cost[b], (unsigned long long)__double_as_longlong((double)(wall_clock64() & 0xFFFFFFFFFFll)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&kb.r.best_idx[b], (int)(t_begin & 0x7FFFFFFFll), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
#endif
        if constexpr (FISS) {
            if (la.ft.flag) {  // the ego's rows of the tables are complete (both halves of a tail-split ego: the ticket came after theirs)
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                __hip_atomic_store(&la.ft.flag[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
            }
        }
#if !defined(FP_TL)
        kb.r.best_idx[b] = r.idx;
        kb.r.best_cost[b] = r.idx >= 0 ? r.cost : __builtin_nan("");
#endif
        if (kb.r.stats) {
            int32_t* st = kb.r.stats + (size_t)b * 4;
            st[0] = 0; st[1] = C; st[2] = C; st[3] = C;
        }
    }
    // 10 ns ticks: feeds the next launches' order (a tail ego leaves twice its last part's time: about what it would take whole)
    if (tid == 0 && la.dur && (nsplit == 1 || in_tail)) la.dur[b] = (int)(wall_clock64() - t_begin) * nsplit;
    if (nsplit > 1) FP_STAMP_LAST(10); else FP_STAMP(10);
#if defined(FP_PHASE_STAMPS)  // column 15: the workgroup's absolute start (10 ns ticks, low 40 bits) - the launch's occupancy over time
    if (threadIdx.x == 0 && ka.r.best_traj && (perm || blockIdx.x % nsplit == 0)) ka.r.best_traj[((size_t)b * FP_ARR_COUNT + 15) * (ka.r.traj_stride > 0 ? ka.r.traj_stride : FP_DEFAULT_STRIDE) + 112 + 15] = (double)(t_begin & 0xFFFFFFFFFFll);
#endif
    // ---------------------------------------------------------------- winner epilogue (what plan() returns), on request
    // The workgroup that found the argmin writes its series itself: no second launch, no re-staging of the ego's tables, and the
    // stores hide behind the other workgroups' arithmetic.  ONE wavefront does it (two time points per lane, neighbours by lane
    // shuffles: no barrier, no scratch); the other seven are done and leave - their issue slots go to the CU's other workgroup.
#if defined(FP_ABL_NO_WINNER)
    return;
#endif
    // fp_plan_step: the ego's hand-over to its next state (point 1 of the winner, stop rules) by the thread that published the argmin -
    // after the series below where this workgroup writes them (they describe the trajectory from the OLD state).  Every other
    // workgroup of the ego (latency mode, tail split) has read the state before it posted its partial argmin.
    // (The three-workgroup instances never do it: the hand-over needs ~110 registers - atan2, hypot, the spline search through global
    // memory - against their 80, and an instance that spills there also changes its code elsewhere; their launcher reports it and
    // advance_kernel follows the launch.)
    if constexpr (OCC > 4) return;  // (this variant is only launched when the series come from winner_traj_kernel)
    if (!ka.r.best_traj) {
        if (ka.has_loop && tid == 0) advance_ego(ka, b, s_best[0].idx, nullptr, ka.loop);
        return;
    }
    __syncthreads();
    if (wave != 0) return;
    const int win = s_best[0].idx;
    double d_end = __builtin_nan(""), v_end = d_end, T_end = d_end;
    if (win >= 0) {
        const int q1 = div_by<NV, ND * NV * NT>(win, inv_nv_a), iv = win - mul24(q1, nv);
        const int id = div_by<NT, ND * NT>(q1, inv_nt_a), it = q1 - mul24(id, nt);
        d_end = s_ds[id]; v_end = s_vs[iv]; T_end = s_ts[it];
    }
    if (ld != NX) {  // (coefficient window: the series run over the WHOLE horizon - straight from the batch's table)
        const int fg = in_frame[b];
        winner_series_wave(ka, b, b, win >= 0, d_end, v_end, T_end, lane, SplineLds{bt.knots + (size_t)fg * NX, bt.coef + (size_t)fg * 8 * NX, nx, NX}, eg);
    } else {
        winner_series_wave(ka, b, b, win >= 0, d_end, v_end, T_end, lane, sp, eg);
    }
    FP_STAMP_LAST(6);
    if (ka.has_loop && tid == 0) advance_ego(ka, b, win, nullptr, ka.loop);
}

// Returns hipErrorInvalidValue when the problem does not fit this kernel (caller falls back to the lane-per-candidate kernel).
// LDS budget of one workgroup (the CU has 160 KB; beyond ~82 KB only one workgroup fits per CU)
constexpr int kLdsLimit = 150 * 1024;
constexpr int kLdsQuarter = 40 * 1024 - 512;  // (a margin below 160 KB / 4 for the allocation granule)

static bool fused_shape(const fp_params& p, const fp_batch& b, int* rows_out, int* hp_out)
{
    if (p.nd > kWave || p.nv > 255 || b.n_obs > 4095) return false;
    const int stride = p.check_stride;
    int rows = 0, hp = 0;
    if (b.n_obs > 0) {
        rows = (points_cap(p) + stride - 1) / stride;
        const int rows_tab = (b.T_obs + stride - 1) / stride;
        if (rows_tab < rows) rows = rows_tab;
        hp = rows * stride + 1;
        if (hp > points_cap(p)) hp = points_cap(p);
        if (rows > 4095 || (long)rows * b.n_obs > 65535) return false;
    }
    *rows_out = rows;
    *hp_out = hp;
    return true;
}

// Largest number of time-horizon slices one workgroup can hold at once (the grouped instances, GS = 0): LDS budget and the 8-bit
// profile index of the hit word.  0 when the problem does not fit the fused kernel at all.
int lattice_group_fit(const fp_params& p, const fp_batch& b)
{
    int rows = 0, hp = 0;
    if (!fused_shape(p, b, &rows, &hp)) return 0;
    int gs = 0;
    for (int g = 1; g <= p.nt; ++g) {
        if (g * p.nv > 256 || make_layout(b.NX, b.n_obs, rows, hp, p.nd, p.nv, p.nt, item_cap(4), g, FP_GROUP_THREADS / kWave, b.obs_nvert ? b.poly_stride : 0).total > kLdsLimit) break;
        gs = g;
    }
    return gs;
}

static std::atomic<long> g_launches_per_cu[3];
long lattice_launches_per_cu(int which) { return which >= 0 && which < 3 ? g_launches_per_cu[which].load(std::memory_order_relaxed) : 0; }

hipError_t launch_lattice_fused(const KernelArgs& ka, hipStream_t stream, void* part_scratch, int nsplit, bool* winner_done, const int* perm, int* dur,
                                int group, const InlineIn* inl, int tail, bool* step_done, const FissTail* ft, bool* search_done)
{
    if (step_done) *step_done = false;
    if (search_done) *search_done = false;
    static const InlineIn kNoInline{};
    const InlineIn& in = inl ? *inl : kNoInline;
    if (winner_done) *winner_done = false;
    const fp_params& p = ka.p;
    const fp_batch& b = ka.b;
    int rows = 0, hp = 0;
    if (!fused_shape(p, b, &rows, &hp)) return hipErrorInvalidValue;
    if (!part_scratch || nsplit < 1) nsplit = 1;
    // Three workgroups per CU (the OCC = 6 variant) when the launch has more egos than two per CU can hold at once, nobody needs the
    // series from this kernel and a workgroup's LDS fits a third of the CU's 160 KB; else two per CU (OCC = 4).
    // slices per barrier interval (the grouped instances): as asked for, as far as one workgroup's LDS holds them
    int gs = group < 1 ? 1 : (group > p.nt ? p.nt : group);
    if (gs > 1) {
        const int fit = lattice_group_fit(p, b);
        gs = fit < 1 ? 1 : (gs > fit ? fit : gs);
    }
    const int pstride = b.obs_nvert && b.n_obs > 0 ? b.poly_stride : 0;  // (polygon columns: their counts - and rings, when they fit - live in LDS)
    // Coefficient window (the kernel's wcap): when the whole spline does not fit a residency's LDS share, the largest window that does -
    // if it is at least kWinMin segments (below that too many points would read global memory)
    constexpr int kWinMin = 32;
    auto window_for = [&](int cap_bytes, int occ, int ps, bool slim) {
        const int base = make_layout(b.NX, b.n_obs, rows, hp, p.nd, p.nv, p.nt, item_cap(occ), 1, kThreads / kWave, ps, slim, 0).total;
        const int w = (cap_bytes - base - 16) / 64;
        if (gs > 1) return w >= b.NX ? b.NX : 0;  // (no WIN instance with grouped slices)
        return w >= b.NX ? b.NX : (w >= kWinMin ? w : 0);  // 0: not even a useful window fits
    };
    const int w6 = window_for(52 * 1024, 6, pstride, false);
    const Layout L6 = make_layout(b.NX, b.n_obs, rows, hp, p.nd, p.nv, p.nt, item_cap(6), 1, kThreads / kWave, pstride, false, w6 > 0 ? w6 : b.NX);
    // the series of a three-per-CU launch: by epilogue workgroups appended to the grid (ka.epi_flag + ka.idx_shadow from the caller), if
    // there are fewer of them than resident slots (see the kernel); else the caller launches winner_traj_kernel behind this launch
    constexpr int kEpiPairs = kEpiPairsC;
    const int kEpiLds = kEpiLdsBytes + (b.NX <= kEpiSplineNX ? kEpiPairs * 9 * b.NX * 8 : 0);
#if defined(FP_PHASE_STAMPS) || defined(FP_COUNTERS)  // (the stamps travel in the series block: the three-workgroup variant leaves the series themselves unwritten)
    const bool can_epi = false;
    bool three = gs == 1 && nsplit == 1 && b.B > ka.resident2 && L6.total <= 52 * 1024;
#else
    const bool can_epi = ka.r.best_traj && ka.epi_flag && ka.idx_shadow && !ka.has_loop;
    // (series asked of THIS kernel pin it to the two-per-CU instances; series offered to the epilogue workgroups do not)
    bool three = gs == 1 && nsplit == 1 && (!ka.r.best_traj || ka.epi_flag) && b.B > ka.resident2 && L6.total <= 52 * 1024;  // (a margin below 160 KB / 3 for the allocation granule)
#endif
    if (ka.lds_cu_kb < 160) three = false;  // (the 52 KB / 40 KB layouts are thirds / quarters of gfx950's 160 KB; a CU with less LDS keeps two per CU)
    if (in.on) three = false;  // (inline inputs are read by the two-workgroup instances only; they belong to tiny batches anyway)
#if defined(FP_NO_OCC6)  // (A/B diagnostic)
    three = false;
#endif
    if (ka.occ_cap == 2) three = false;  // (fp_ctx_set_option("lattice_occupancy"))
    const bool epilogue = three && can_epi;
    // the FISS+ search in appended workgroups: three-per-CU launches that write their tables, lattices the 1024-sample search instance holds
    const int C_all = p.nd * p.nv * p.nt;
    bool search = three && ft && ft->flag && ka.r.cost_tbl && ka.r.flag_tbl && !ka.r.best_traj && !ka.has_loop && C_all > 4 * kWave && C_all <= 1024 &&
                        ft->opts.kind == FP_FISS_PLUS && !(b.obs_nvert && b.n_obs > 0);
    static const FissTail kNoFiss{};
    FissTail fx = search ? *ft : kNoFiss;
    if (search) fx.NB = 512;  // (a multiple of 64 x the 8 wavefronts of the appended workgroups)
    const int search_lds = search ? fsp::fissplus_lds_bytes(C_all, fx.NB) : 0;
    // FOUR workgroups per CU when the slim layout (make_layout) and the appended workgroups' LDS fit a quarter of the CU (BASELINE.json's
    // dense shape: reference lines of up to ~80 knots); 64 VGPRs a lane, the ego's start state re-read from LDS (kEgoLds)
    const int w8 = window_for(kLdsQuarter, 8, 0, true);
    const Layout L8 = make_layout(b.NX, b.n_obs, rows, hp, p.nd, p.nv, p.nt, item_cap(8), 1, kThreads / kWave, 0, true, w8 > 0 ? w8 : b.NX);
    bool four = three && !pstride && L8.total <= kLdsQuarter && (!epilogue || kEpiLds <= kLdsQuarter) && (!search || search_lds <= kLdsQuarter) && (long)b.B * 2 > (long)ka.resident2 * 3 &&
                (!b.skip || ka.occ_cap == 4);  // (a closed-loop batch: its finished egos leave at once, what runs rarely fills three per CU - measured 68 -> 71-75 us per cycle with four; "lattice_occupancy" 4 asks for it anyway)
#if defined(FP_NO_OCC8)  // (A/B diagnostic)
    four = false;
#endif
    if (ka.occ_cap == 2 || ka.occ_cap == 3) four = false;
    KernelArgs kx = ka;  // (epilogue workgroups offered but not taken: the caller's winner_traj_kernel writes the series)
    if (ka.epi_flag && !epilogue) { kx.r.best_traj = nullptr; kx.epi_flag = nullptr; }
    Layout L = four ? L8 : three ? L6 : make_layout(b.NX, b.n_obs, rows, hp, p.nd, p.nv, p.nt, item_cap(4), gs, (gs > 1 ? FP_GROUP_THREADS : kThreads) / kWave, pstride);
    if (L.total > kLdsLimit) return hipErrorInvalidValue;
    if (epilogue && L.total < kEpiLds) L.total = kEpiLds;  // (every workgroup of a launch gets the same dynamic LDS)
    if (search && L.total < search_lds) L.total = search_lds;
    const int wcap = in.on ? b.NX : four ? (w8 > 0 ? w8 : b.NX) : three ? (w6 > 0 ? w6 : b.NX) : b.NX;  // (two per CU: the whole table, as before)
    const bool windowed = wcap < b.NX;
    if (windowed && search) { search = false; fx = kNoFiss; }  // (no WIN instance with appended search workgroups: the search follows in its own launch)
    if (nsplit > p.nt) nsplit = p.nt;
    // part_scratch: [ticket counters: kTicketBytes, zero between launches][partial argmins: Best x B x nsplit]
    int* part_count = (int*)part_scratch;
    Best* part_best = part_scratch ? (Best*)((char*)part_scratch + kTicketBytes) : nullptr;
    if (nsplit != 1) perm = nullptr;
    // Tail split: a launch of several rounds of workgroups (one per ego) ends on the egos that happened to start last - with ~50 us
    // per ego and the last workgroup starting ~40 us before the end, a fifth of the launch runs on a draining chip.  The last `tail`
    // dispatch slots are cut in two (time-horizon slices it_lo .. it_hi per part, ticket + merge like the latency mode): each half
    // repeats the ego's prologue, so only a quarter of a round's worth of slots is cut (tail < 0: auto).  Results do not depend on it.
    int tail_from = -1;
#if !defined(FP_PHASE_STAMPS) && !defined(FP_COUNTERS)
    if (tail != 0 && nsplit == 1 && gs == 1 && part_scratch && p.nt >= 2 && b.S > 0 && b.n_obs > 0 && (size_t)b.B * 4 <= kTicketBytes) {
        const int resident = (four ? 4 : three ? 3 : 2) * (tail < 0 ? -tail : 0);  // workgroups the device holds at once (auto: tail = -compute units)
        // (three per CU: 128 ... 384 of 768 slots measured within 1 %; 576: no gain; 768: slower.  Four per CU, launch order = the batch's own
        // history, round 6: 96-192 of 1024 within 1 % of each other and of no cut at all, 512: 4 % slower - an eighth of a round, which also
        // halves the inputs staged twice)
        int n_tail = tail > 0 ? tail : (b.B > resident ? resident / (four ? 8 : 4) : 0);
        if (n_tail > b.B - resident && tail < 0) n_tail = b.B - resident;
        if (n_tail > b.B) n_tail = b.B;
        if (n_tail > 0) tail_from = b.B - n_tail;
    }
#endif
    const unsigned lattice_grid = tail_from >= 0 ? (unsigned)(2 * b.B - tail_from) : (unsigned)(b.B * nsplit);
    const int epi_from = epilogue || search ? (int)lattice_grid : -1;
    const unsigned grid = lattice_grid + (epilogue ? (unsigned)((b.B + kEpiPairs - 1) / kEpiPairs) : 0u) + (search ? (unsigned)b.B : 0u);
    hipError_t e;
    if (p.curvature_mask) {  // optional curvature checks: their own launch, ORed into the flag words by the assembly stage
        if (!ka.curv_tbl) return hipErrorInvalidValue;
        e = launch_curvature_flags(ka, const_cast<uint8_t*>(ka.curv_tbl), stream);
        if (e != hipSuccess) return e;
    }
    // the instance whose compile-time shape is this problem's (BASELINE.json's two lattice shapes), else the run-time one
    auto go = [&](auto kernel, int* configured, int threads = kThreads) -> hipError_t {
        hipError_t err = ensure_dynamic_lds((const void*)kernel, L.total, configured);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), L.total, stream, kx, rows, hp, nsplit, part_best, part_count, perm, dur, gs, tail_from, epi_from, wcap, fx, in);
        return hipGetLastError();
    };
    FP_LDS_SLOTS(cfg_generic);
    FP_LDS_SLOTS(cfg_997);
    FP_LDS_SLOTS(cfg_555);
    FP_LDS_SLOTS(cfg_generic6);
    FP_LDS_SLOTS(cfg_9976);
    FP_LDS_SLOTS(cfg_9978);
    FP_LDS_SLOTS(cfg_9978f);
    FP_LDS_SLOTS(cfg_generic8);
    FP_LDS_SLOTS(cfg_generic8f);
    FP_LDS_SLOTS(cfg_5556);
    FP_LDS_SLOTS(cfg_generic_g);
    FP_LDS_SLOTS(cfg_997_g);
    FP_LDS_SLOTS(cfg_555_g);
    FP_LDS_SLOTS(cfg_9976f);
    FP_LDS_SLOTS(cfg_generic6f);
    FP_LDS_SLOTS(cfg_poly);
    FP_LDS_SLOTS(cfg_poly6);
    FP_LDS_SLOTS(cfg_poly9976);
    FP_LDS_SLOTS(cfg_poly6w);
    FP_LDS_SLOTS(cfg_poly9976w);
    FP_LDS_SLOTS(cfg_poly_g);
    FP_LDS_SLOTS(cfg_9978w);
    FP_LDS_SLOTS(cfg_generic8w);
    FP_LDS_SLOTS(cfg_9976w);
    FP_LDS_SLOTS(cfg_generic6w);
    auto is = [&](int nd, int nv, int nt, int stride, int n_obs, int r) {
        return p.nd == nd && p.nv == nv && p.nt == nt && p.check_stride == stride && b.n_obs == n_obs && rows == r;
    };
#if defined(FP_NO_SHAPES)  // (A/B diagnostic: the run-time instance for every shape)
    e = gs > 1 ? go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 4, 0, FP_GROUP_THREADS, true>, cfg_generic_g, FP_GROUP_THREADS) : go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 4, 1, 512, true>, cfg_generic);
#else
    if (b.obs_nvert && b.n_obs > 0) {  // convex-polygon columns: the run-time-shape instances with the polygon narrow phase
        if (gs > 1) e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 4, 0, FP_GROUP_THREADS, true>, cfg_poly_g, FP_GROUP_THREADS);
        else if (three && windowed && is(9, 9, 7, 2, 50, 25)) e = go(lattice_fused_kernel<9, 9, 7, 2, 50, 25, 6, 1, 512, true, false, true>, cfg_poly9976w);
        else if (three && windowed) e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 6, 1, 512, true, false, true>, cfg_poly6w);
        else if (three && is(9, 9, 7, 2, 50, 25)) e = go(lattice_fused_kernel<9, 9, 7, 2, 50, 25, 6, 1, 512, true>, cfg_poly9976);
        else if (three) e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 6, 1, 512, true>, cfg_poly6);
        else e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 4, 1, 512, true>, cfg_poly);
    } else if (gs > 1) {
        if (is(9, 9, 7, 2, 50, 25)) e = go(lattice_fused_kernel<9, 9, 7, 2, 50, 25, 4, 0, FP_GROUP_THREADS>, cfg_997_g, FP_GROUP_THREADS);
        else if (is(5, 5, 5, 2, 10, 50)) e = go(lattice_fused_kernel<5, 5, 5, 2, 10, 50, 4, 0, FP_GROUP_THREADS>, cfg_555_g, FP_GROUP_THREADS);
        else e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 4, 0, FP_GROUP_THREADS>, cfg_generic_g, FP_GROUP_THREADS);
    } else if (search) {
        if (four && is(9, 9, 7, 2, 50, 25)) e = go(lattice_fused_kernel<9, 9, 7, 2, 50, 25, 8, 1, 512, false, true>, cfg_9978f);
        else if (four) e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 8, 1, 512, false, true>, cfg_generic8f);
        else if (is(9, 9, 7, 2, 50, 25)) e = go(lattice_fused_kernel<9, 9, 7, 2, 50, 25, 6, 1, 512, false, true>, cfg_9976f);
        else e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 6, 1, 512, false, true>, cfg_generic6f);
    } else if (three && windowed) {  // long reference lines: the instances that keep a window of the coefficient columns in LDS
        if (four && is(9, 9, 7, 2, 50, 25)) e = go(lattice_fused_kernel<9, 9, 7, 2, 50, 25, 8, 1, 512, false, false, true>, cfg_9978w);
        else if (four) e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 8, 1, 512, false, false, true>, cfg_generic8w);
        else if (is(9, 9, 7, 2, 50, 25)) e = go(lattice_fused_kernel<9, 9, 7, 2, 50, 25, 6, 1, 512, false, false, true>, cfg_9976w);
        else e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 6, 1, 512, false, false, true>, cfg_generic6w);
    } else if (three) {
        if (four && is(9, 9, 7, 2, 50, 25)) e = go(lattice_fused_kernel<9, 9, 7, 2, 50, 25, 8, 1, 512>, cfg_9978);
        else if (four) e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 8, 1, 512>, cfg_generic8);
        else if (is(9, 9, 7, 2, 50, 25)) e = go(lattice_fused_kernel<9, 9, 7, 2, 50, 25, 6, 1, 512>, cfg_9976);
        else if (is(5, 5, 5, 2, 10, 50)) e = go(lattice_fused_kernel<5, 5, 5, 2, 10, 50, 6, 1, 512>, cfg_5556);
        else e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 6, 1, 512>, cfg_generic6);
    } else {
        if (is(9, 9, 7, 2, 50, 25)) e = go(lattice_fused_kernel<9, 9, 7, 2, 50, 25, 4, 1, 512>, cfg_997);
        else if (is(5, 5, 5, 2, 10, 50)) e = go(lattice_fused_kernel<5, 5, 5, 2, 10, 50, 4, 1, 512>, cfg_555);
        else e = go(lattice_fused_kernel<0, 0, 0, 0, 0, 0, 4, 1, 512>, cfg_generic);
    }
#endif
    if (e != hipSuccess) return e;
    g_launches_per_cu[four ? 2 : three ? 1 : 0].fetch_add(1, std::memory_order_relaxed);
    if (winner_done) *winner_done = kx.r.best_traj != nullptr && (!three || epilogue);
    if (step_done) *step_done = ka.has_loop != 0 && !three;  // (ka.has_loop: the two-per-CU instances hand the egos over themselves)
    if (search_done) *search_done = search;
    return hipSuccess;
}

}  // namespace fp
