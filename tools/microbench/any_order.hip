// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx".)
// A kernel of `wgs` workgroups, each spinning `us` microseconds on the 100 MHz wall clock.  Two launches back to back on one stream:
// serialised -> ~2 x us, overlapped -> ~us.  Build: hipcc --offload-arch=gfx950 -O2 -o any_order any_order.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long ticks, int* out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (out && threadIdx.x == 0) out[blockIdx.x] = 1;
}
static double run(hipStream_t s, int n, int wgs, long long ticks, int flags, int* out)
{
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
        if (flags < 0) hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, s, ticks, out);
        else hipExtLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, s, nullptr, nullptr, (unsigned)flags, ticks, out);
    }
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
int main()
{
    hipStream_t s;
    hipStreamCreate(&s);
    int* out;
    hipMalloc(&out, 1 << 20);
    for (int w = 0; w < 3; ++w) run(s, 4, 64, 10000, -1, out);
    for (int wgs : {64, 1024}) {
        for (long long us : {100ll, 500ll}) {
            const long long ticks = us * 100;
            printf("wgs %4d x %3lld us, 8 launches: plain %8.1f us | ext flags=0 %8.1f us | ext AnyOrder %8.1f us\n", wgs, us, run(s, 8, wgs, ticks, -1, out),
                   run(s, 8, wgs, ticks, 0, out), run(s, 8, wgs, ticks, hipExtAnyOrderLaunch, out));
        }
    }
    return 0;
}
