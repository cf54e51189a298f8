// VALU issue cost per wave64 instruction on gfx950, by instruction class (tools/microbench/issue_rate.hip).
// N waves per SIMD, each running a long unrolled stream of ONE instruction over 8 independent register sets; cycles per instruction per
// SIMD = kernel time x clock / (instructions per wave x waves per SIMD).  Build + run on the GPU box:
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/issue_rate tools/microbench/issue_rate.hip && /tmp/issue_rate
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x
constexpr int kIters = 2000;

#define OUT_IDX (blockIdx.x * blockDim.x + threadIdx.x)
// three-operand op on 8 independent accumulators: op a, a, b, c
#define K3(name, T, op)                                                                                                     \
    __global__ __launch_bounds__(1024) void name(double* out, int n)                                                        \
    {                                                                                                                       \
        T a0 = (T)threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = (T)1.0000001, c = (T)1e-9;       \
        for (int it = 0; it < n; ++it) {                                                                                    \
            REP8(asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n"   \
                              op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"   \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                              : "v"(b), "v"(c));)                                                                           \
        }                                                                                                                   \
        out[OUT_IDX] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                                                     \
    }
// two-operand op: op a, a, b   (clob: extra clobber string, e.g. "vcc")
#define K2(name, T, op)                                                                                                     \
    __global__ __launch_bounds__(1024) void name(double* out, int n)                                                        \
    {                                                                                                                       \
        T a0 = (T)threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = (T)3;                            \
        for (int it = 0; it < n; ++it) {                                                                                    \
            REP8(asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n"                   \
                              op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"                   \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                              : "v"(b));)                                                                                   \
        }                                                                                                                   \
        out[OUT_IDX] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                                                     \
    }
// one-operand op: op a, a
#define K1(name, T, op)                                                                                                     \
    __global__ __launch_bounds__(1024) void name(double* out, int n)                                                        \
    {                                                                                                                       \
        T a0 = (T)(threadIdx.x + 1), a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;                                \
        for (int it = 0; it < n; ++it) {                                                                                    \
            REP8(asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n"                                   \
                              op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"                                   \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)           \
        }                                                                                                                   \
        out[OUT_IDX] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                                                     \
    }
// compare: op vcc, a, b
#define KC(name, T, op)                                                                                                     \
    __global__ __launch_bounds__(1024) void name(double* out, int n)                                                        \
    {                                                                                                                       \
        T a0 = (T)threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = (T)3;                            \
        for (int it = 0; it < n; ++it) {                                                                                    \
            REP8(asm volatile(op " vcc, %0, %8\n" op " vcc, %1, %8\n" op " vcc, %2, %8\n" op " vcc, %3, %8\n"               \
                              op " vcc, %4, %8\n" op " vcc, %5, %8\n" op " vcc, %6, %8\n" op " vcc, %7, %8\n"               \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                              : "v"(b) : "vcc");)                                                                           \
        }                                                                                                                   \
        out[OUT_IDX] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                                                     \
    }

K3(k_fma_f64, double, "v_fma_f64")
K2(k_mul_f64, double, "v_mul_f64")
K2(k_add_f64, double, "v_add_f64")
K2(k_max_f64, double, "v_max_f64")
KC(k_cmp_f64, double, "v_cmp_gt_f64")
K1(k_rsq_f64, double, "v_rsq_f64")
K1(k_rcp_f64, double, "v_rcp_f64")
K1(k_sqrt_f64, double, "v_sqrt_f64")
K3(k_fma_f32, float, "v_fma_f32")
K2(k_mul_f32, float, "v_mul_f32")
K2(k_add_f32, float, "v_add_f32")
K2(k_max_f32, float, "v_max_f32")
KC(k_cmp_f32, float, "v_cmp_gt_f32")
K1(k_rcp_f32, float, "v_rcp_f32")
K1(k_rsq_f32, float, "v_rsq_f32")
K3(k_pk_fma_f32, double, "v_pk_fma_f32")
K2(k_pk_mul_f32, double, "v_pk_mul_f32")
K2(k_pk_add_f32, double, "v_pk_add_f32")
K2(k_add_u32, int, "v_add_u32")
K2(k_mul_lo_u32, int, "v_mul_lo_u32")
K2(k_mul_u24, int, "v_mul_u32_u24")
K2(k_and_b32, int, "v_and_b32")
K2(k_lshl_b32, int, "v_lshlrev_b32")
K3(k_mad_u24, int, "v_mad_u32_u24")
K1(k_mov_b32, int, "v_mov_b32")

__global__ __launch_bounds__(1024) void k_cvt_f32_f64(double* out, int n)
{
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
    float f0 = 0, f1 = 0, f2 = 0, f3 = 0;
    for (int it = 0; it < n; ++it) {
        REP8(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                          "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                          : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    }
    out[OUT_IDX] = f0 + f1 + f2 + f3;
}
__global__ __launch_bounds__(1024) void k_cndmask(double* out, int n)
{
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 3;
    for (int it = 0; it < n; ++it) {
        REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                          "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    }
    out[OUT_IDX] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
// v_cndmask variants: mask in an SGPR pair (e64), and a v_cmp feeding every select (the compiler's usual pairing)
__global__ __launch_bounds__(1024) void k_cndmask_sgpr(double* out, int n)
{
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 3;
    unsigned long long m = 0x5555555555555555ull + blockIdx.x;
    for (int it = 0; it < n; ++it) {
        REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                          "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m));)
    }
    out[OUT_IDX] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ __launch_bounds__(1024) void k_cmp_cndmask(double* out, int n)
{
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, b = 3;
    for (int it = 0; it < n; ++it) {
        REP8(asm volatile("v_cmp_gt_i32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_gt_i32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n"
                          "v_cmp_gt_i32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_gt_i32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
    }
    out[OUT_IDX] = (double)(a0 + a1 + a2 + a3);
}
__global__ __launch_bounds__(1024) void k_cmp_i32(double* out, int n)
{
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, b = 3;
    for (int it = 0; it < n; ++it) {
        REP8(asm volatile("v_cmp_gt_i32 vcc, %0, %4\n v_cmp_gt_i32 vcc, %1, %4\n v_cmp_gt_i32 vcc, %2, %4\n v_cmp_gt_i32 vcc, %3, %4\n"
                          "v_cmp_gt_i32 vcc, %0, %4\n v_cmp_gt_i32 vcc, %1, %4\n v_cmp_gt_i32 vcc, %2, %4\n v_cmp_gt_i32 vcc, %3, %4\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
    }
    out[OUT_IDX] = (double)(a0 + a1 + a2 + a3);
}
__global__ __launch_bounds__(1024) void k_readfirstlane(double* out, int n)
{
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int it = 0; it < n; ++it) {
        REP8(asm volatile("v_readfirstlane_b32 %0, %4\n v_readfirstlane_b32 %1, %5\n v_readfirstlane_b32 %2, %6\n v_readfirstlane_b32 %3, %7\n"
                          "v_readfirstlane_b32 %0, %4\n v_readfirstlane_b32 %1, %5\n v_readfirstlane_b32 %2, %6\n v_readfirstlane_b32 %3, %7\n"
                          : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    }
    out[OUT_IDX] = (double)(s0 + s1 + s2 + s3);
}
__global__ __launch_bounds__(1024) void k_bpermute(double* out, int n)
{
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, idx = ((threadIdx.x + 1) & 63) * 4;
    for (int it = 0; it < n; ++it) {
        REP8(asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n"
                          "s_waitcnt lgkmcnt(0)\n"
                          "ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n"
                          "s_waitcnt lgkmcnt(0)\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(idx));)
    }
    out[OUT_IDX] = (double)(a0 + a1 + a2 + a3);
}
__global__ __launch_bounds__(1024) void k_dpp_mov(double* out, int n)
{
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    for (int it = 0; it < n; ++it) {
        REP8(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                          "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                          "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                          "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
    out[OUT_IDX] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
K2(k_min_f64, double, "v_min_f64")

__global__ __launch_bounds__(1024) void k_s_add(double* out, int n)
{
    int a0 = blockIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    for (int it = 0; it < n; ++it) {
        REP8(asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n s_add_u32 %4, %4, 3\n s_add_u32 %5, %5, 3\n s_add_u32 %6, %6, 3\n s_add_u32 %7, %7, 3\n"
                          : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3), "+s"(a4), "+s"(a5), "+s"(a6), "+s"(a7) : : "scc");)
    }
    out[OUT_IDX] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ __launch_bounds__(1024) void k_ds_read_b128(double* out, int n)
{
    __shared__ double2 sm[2048];
    sm[threadIdx.x] = make_double2(threadIdx.x, 1);
    sm[threadIdx.x + 1024] = make_double2(threadIdx.x, 2);
    __syncthreads();
    double acc = 0;
    int idx = threadIdx.x;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {  // 64 b128 reads per iteration, 16-byte stride across lanes (conflict free)
            const double2 a = sm[(idx + 64 * u) & 2047];
            acc += a.x;
        }
        idx = (idx + 1) & 1023;
    }
    out[OUT_IDX] = acc;
}
// a mix like the lattice kernel's stream: fp64 fma + i32 add + s_add per step (issue slots shared or not?)
__global__ __launch_bounds__(1024) void k_mix(double* out, int n)
{
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, b = 1.0000001, c = 1e-9;
    int i0 = 1, i1 = 2, i2 = 3, i3 = 4, s0 = 0, s1 = 1;
    for (int it = 0; it < n; ++it) {
        REP8(asm volatile("v_fma_f64 %0, %0, %10, %11\n v_add_u32 %4, %4, 3\n s_add_u32 %8, %8, 3\n v_fma_f64 %1, %1, %10, %11\n v_add_u32 %5, %5, 3\n s_add_u32 %9, %9, 3\n"
                          "v_fma_f64 %2, %2, %10, %11\n v_add_u32 %6, %6, 3\n s_add_u32 %8, %8, 3\n v_fma_f64 %3, %3, %10, %11\n v_add_u32 %7, %7, 3\n s_add_u32 %9, %9, 3\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+s"(s0), "+s"(s1) : "v"(b), "v"(c) : "scc");)
    }
    out[OUT_IDX] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3 + s0 + s1;
}
// a DEPENDENT chain of fp64 FMAs: latency per instruction with one wave per SIMD
__global__ __launch_bounds__(1024) void k_chain_f64(double* out, int n)
{
    double a0 = threadIdx.x, b = 1.0000001, c = 1e-9;
    for (int it = 0; it < n; ++it) {
        REP8(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                          "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                          : "+v"(a0) : "v"(b), "v"(c));)
    }
    out[OUT_IDX] = a0;
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;  // Hz
    printf("%s: %d CUs, clock %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    double* out;
    (void)hipMalloc(&out, (size_t)cus * 2 * 1024 * sizeof(double));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct T { const char* name; void (*k)(double*, int); int per_iter; };
    const T tests[] = {{"v_fma_f64", k_fma_f64, 64}, {"v_mul_f64", k_mul_f64, 64}, {"v_add_f64", k_add_f64, 64}, {"v_max_f64", k_max_f64, 64}, {"v_cmp_gt_f64", k_cmp_f64, 64},
                       {"v_rsq_f64", k_rsq_f64, 64}, {"v_rcp_f64", k_rcp_f64, 64}, {"v_sqrt_f64", k_sqrt_f64, 64}, {"v_cvt_f32_f64", k_cvt_f32_f64, 64},
                       {"v_fma_f32", k_fma_f32, 64}, {"v_mul_f32", k_mul_f32, 64}, {"v_add_f32", k_add_f32, 64}, {"v_max_f32", k_max_f32, 64}, {"v_cmp_gt_f32", k_cmp_f32, 64},
                       {"v_rcp_f32", k_rcp_f32, 64}, {"v_rsq_f32", k_rsq_f32, 64}, {"v_pk_fma_f32", k_pk_fma_f32, 64}, {"v_pk_mul_f32", k_pk_mul_f32, 64}, {"v_pk_add_f32", k_pk_add_f32, 64},
                       {"v_add_u32", k_add_u32, 64}, {"v_mul_lo_u32", k_mul_lo_u32, 64}, {"v_mul_u32_u24", k_mul_u24, 64}, {"v_mad_u32_u24", k_mad_u24, 64}, {"v_and_b32", k_and_b32, 64},
                       {"v_lshlrev_b32", k_lshl_b32, 64}, {"v_mov_b32", k_mov_b32, 64}, {"v_cndmask_b32 vcc", k_cndmask, 64}, {"v_cndmask_b32_e64 sgpr mask", k_cndmask_sgpr, 64}, {"v_cmp_gt_i32 + v_cndmask (per pair)", k_cmp_cndmask, 32}, {"v_cmp_gt_i32", k_cmp_i32, 64},
                       {"v_readfirstlane_b32", k_readfirstlane, 64}, {"ds_bpermute_b32 (4 in flight)", k_bpermute, 64}, {"v_mov_b32_dpp wave_shr:1", k_dpp_mov, 64}, {"v_min_f64", k_min_f64, 64}, {"s_add_u32", k_s_add, 64},
                       {"ds_read_b128", k_ds_read_b128, 64}, {"mix: f64 fma + i32 add + s_add (per triple)", k_mix, 32}, {"v_fma_f64 dependent chain", k_chain_f64, 64}};
    for (int waves_per_simd : {8, 4, 1}) {
        printf("---- %d wave(s) per SIMD\n", waves_per_simd);
        // one workgroup per CU: waves_per_simd x 4 wavefronts
        const int blocks = cus, threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
        const int per_cu = 256 * waves_per_simd / threads;
        for (const T& t : tests) {
            hipLaunchKernelGGL(t.k, dim3(blocks * per_cu), dim3(threads), 0, 0, out, 50);
            (void)hipDeviceSynchronize();
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(t.k, dim3(blocks * per_cu), dim3(threads), 0, 0, out, kIters);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double inst_per_wave = (double)kIters * t.per_iter;
            const double cyc = best * 1e-3 * clk / (inst_per_wave * waves_per_simd);
            printf("%-44s %8.3f ms   %6.2f cycles per wave-instruction per SIMD\n", t.name, best, cyc);
        }
    }
    return 0;
}
