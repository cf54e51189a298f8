// frenet_frame.hip - Frenet frame construction and Cartesian -> Frenet projection on the device.
//
//   frames_build_kernel   one workgroup per centerline: chord-length knots, natural cubic spline per axis
//                         (reference common/geometry/cubic_spline.py:19-43,157-168)
//   from_state_kernel     one workgroup per ego: argmin over the 0.1 m-resampled reference line + the projection rules of
//                         FrenetState.from_state (reference common/scenario/frenet.py:32-99)
#include "frenet_device.h"
#include "frenet_kernels.h"

namespace fp {

namespace {
constexpr int kFrameThreads = 128;
constexpr double kPi = 3.141592653589793;
}

// LDS: s[NX], h[NX], then per axis: a[NX], c[NX], cp[NX], dp[NX]
__global__ __launch_bounds__(kFrameThreads) void frames_build_kernel(int NX, const int32_t* n_of, const double* points, double* knots_out,
                                                                      double* coef_out)
{
    extern __shared__ __attribute__((aligned(16))) double fl[];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = n_of[f];
    double* s = fl;
    double* h = s + NX;
    const double* pts = points + (size_t)f * NX * 2;
    // segment lengths in parallel, cumulative sum in order (np.cumsum is sequential too)
    for (int i = tid; i < n - 1; i += kFrameThreads) h[i] = hypot(pts[2 * (i + 1)] - pts[2 * i], pts[2 * (i + 1) + 1] - pts[2 * i + 1]);
    __syncthreads();
    if (tid == 0) {
        double acc = 0.0;
        s[0] = 0.0;
        for (int i = 0; i < n - 1; ++i) { acc += h[i]; s[i + 1] = acc; }
    }
    __syncthreads();
    for (int i = tid; i < n - 1; i += kFrameThreads) h[i] = s[i + 1] - s[i];  // np.diff(x) of the knot vector (:21)
    __syncthreads();
    // Thomas sweep per axis on lanes 0 (x) and 64 (y): natural end conditions c_0 = c_{n-1} = 0 (:118-142)
    if (tid == 0 || tid == kWave) {
        const int ax = tid == 0 ? 0 : 1;
        double* a = h + NX + ax * 4 * NX;
        double* c = a + NX;
        double* cp = c + NX;
        double* dp = cp + NX;
        for (int i = 0; i < n; ++i) a[i] = pts[2 * i + ax];
        cp[0] = 0.0;
        dp[0] = 0.0;
        for (int i = 1; i < n - 1; ++i) {
            const double rhs = 3.0 * (a[i + 1] - a[i]) / h[i] - 3.0 * (a[i] - a[i - 1]) / h[i - 1];
            const double den = 2.0 * (h[i - 1] + h[i]) - h[i - 1] * cp[i - 1];
            cp[i] = h[i] / den;
            dp[i] = (rhs - h[i - 1] * dp[i - 1]) / den;
        }
        c[n - 1] = 0.0;
        for (int i = n - 2; i >= 1; --i) c[i] = dp[i] - cp[i] * c[i + 1];
        c[0] = 0.0;
    }
    __syncthreads();
    double* ko = knots_out + (size_t)f * NX;
    double* co = coef_out + (size_t)f * 8 * NX;
    for (int i = tid; i < NX; i += kFrameThreads) ko[i] = i < n ? s[i] : __builtin_inf();
    for (int e = tid; e < 2 * NX; e += kFrameThreads) {
        const int ax = e / NX, i = e - ax * NX;
        const double* a = h + NX + ax * 4 * NX;
        const double* c = a + NX;
        double av = 0.0, bv = 0.0, cv = 0.0, dv = 0.0;
        if (i < n) {
            av = a[i];
            cv = c[i];
            if (i < n - 1) {  // :39-43
                dv = (c[i + 1] - c[i]) / (3.0 * h[i]);
                bv = 1.0 / h[i] * (a[i + 1] - a[i]) - h[i] / 3.0 * (2.0 * c[i] + c[i + 1]);
            }
        }
        double* row = co + (size_t)ax * 4 * NX;
        row[i] = av; row[NX + i] = bv; row[2 * NX + i] = cv; row[3 * NX + i] = dv;
    }
}

hipError_t launch_frames_build(int F, int NX, const int32_t* n, const double* points, double* knots, double* coef, hipStream_t stream)
{
    const int bytes = (int)sizeof(double) * (2 * NX + 8 * NX);
    hipLaunchKernelGGL(frames_build_kernel, dim3(F), dim3(kFrameThreads), bytes, stream, NX, n, points, knots, coef);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
struct RefPoint {
    double x, y, yaw;
};

__device__ __forceinline__ RefPoint ref_point(const SplineLds& sp, double s)
{
    // calc_position / calc_yaw at s (cubic_spline.py:170-232); s is inside [0, s_last) by construction
    RefPoint r;
    const int seg = spline_segment(sp, s, -1);
    const double* c = sp.coef + seg;
    const int ld = sp.ld;
    const double dx = s - sp.knots[seg];
    r.x = fma(fma(fma(c[3 * ld], dx, c[2 * ld]), dx, c[ld]), dx, c[0]);
    r.y = fma(fma(fma(c[7 * ld], dx, c[6 * ld]), dx, c[5 * ld]), dx, c[4 * ld]);
    const double gx = fma(fma(3.0 * c[3 * ld], dx, 2.0 * c[2 * ld]), dx, c[ld]);
    const double gy = fma(fma(3.0 * c[7 * ld], dx, 2.0 * c[6 * ld]), dx, c[5 * ld]);
    r.yaw = atan2(gy, gx);
    return r;
}

__global__ __launch_bounds__(256) void from_state_kernel(fp_batch bt, const double* states, double* ego_out)
{
    __shared__ double red_v[4];
    __shared__ int red_i[4];
    __shared__ double red_s[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int f = bt.frame_of[b];
    const int nx = bt.nx[f];
    const double* knots = bt.knots + (size_t)f * bt.NX;
    SplineLds sp{knots, bt.coef + (size_t)f * 8 * bt.NX, nx, bt.NX};
    const double sx = states[(size_t)b * 4], sy = states[(size_t)b * 4 + 1], syaw = states[(size_t)b * 4 + 2], sv = states[(size_t)b * 4 + 3];
    const double s_last = knots[nx - 1];
    const int n_ref = (int)ceil(s_last / 0.1);  // len(np.arange(0, s_last, 0.1))  (frenet_optimal_planner.py:274)
    // nearest resampled point: np.argmin -> first minimum (frenet.py:34-36)
    double best = __builtin_inf();
    int bi = 0x7fffffff;
    for (int i = tid; i < n_ref; i += blockDim.x) {
        const RefPoint p = ref_point(sp, (double)i * 0.1);
        const double dd = hypot(p.x - sx, p.y - sy);
        if (dd < best) { best = dd; bi = i; }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off, kWave);
        const int oi = __shfl_xor(bi, off, kWave);
        if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { red_v[wave] = best; red_i[wave] = bi; }
    __syncthreads();
    for (int w = 0; w < (int)blockDim.x / kWave; ++w)
        if (red_v[w] < best || (red_v[w] == best && red_i[w] < bi)) { best = red_v[w]; bi = red_i[w]; }
    const int nearest = bi;
    // find_next_point_idx (:38-56)
    const RefPoint pn = ref_point(sp, (double)nearest * 0.1);
    const double heading = atan2(pn.y - sy, pn.x - sx);
    double angle = fabs(syaw - heading);
    angle = fmin(2.0 * kPi - angle, angle);
    int next = angle > kPi / 2.0 ? nearest + 1 : nearest;
    if (next < 1) next = 1;
    else if (next >= n_ref) next = n_ref - 1;
    const int prev = next - 1 > 0 ? next - 1 : 0;
    // s = sum of the polyline segment lengths before prev (:86-88)
    double acc = 0.0;
    for (int i = tid; i < prev; i += blockDim.x) {
        const RefPoint p0 = ref_point(sp, (double)i * 0.1), p1 = ref_point(sp, (double)(i + 1) * 0.1);
        acc += hypot(p1.x - p0.x, p1.y - p0.y);
    }
    acc = wave_sum_f64(acc);
    __syncthreads();
    if (lane == 0) red_s[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)blockDim.x / kWave; ++w) s += red_s[w];
        const RefPoint pp = ref_point(sp, (double)prev * 0.1), px = ref_point(sp, (double)next * 0.1);
        const double n_x = px.x - pp.x, n_y = px.y - pp.y;
        const double x_x = sx - pp.x, x_y = sy - pp.y;
        const double x_yaw = atan2(x_y, x_x);
        const double proj = (x_x * n_x + x_y * n_y) / (n_x * n_x + n_y * n_y);
        double d = hypot(x_x - proj * n_x, x_y - proj * n_y);
        const double wp_yaw = pp.yaw;
        double delta = syaw - wp_yaw;  // unifyAngleRange (math_utils.py:28-34)
        while (delta > kPi) delta -= 2.0 * kPi;
        while (delta < -kPi) delta += 2.0 * kPi;
        if (wp_yaw <= x_yaw) d = -d;  // :82-83
        double* o = ego_out + (size_t)b * 6;
        o[0] = s; o[1] = sv * cos(delta); o[2] = 0.0;
        o[3] = d; o[4] = sv * sin(delta); o[5] = 0.0;
    }
}

hipError_t launch_from_state(const fp_batch& bt, const double* states, double* ego, hipStream_t stream)
{
    hipLaunchKernelGGL(from_state_kernel, dim3(bt.B), dim3(256), 0, stream, bt, states, ego);
    return hipGetLastError();
}

}  // namespace fp
