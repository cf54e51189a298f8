// frenet_fissplus.hip - the FISS+ coarse search as its own launch: one workgroup per ego behind the lattice launch (the walk itself:
// frenet_fissplus.h).
#include "frenet_fissplus.h"

namespace fp {

using fsp::fissplus_lds_bytes;

template <int W, int NW, int CMAX>
__global__ __launch_bounds__(W * kWave, W >= 8 ? 8 : 1) void fissplus_search_kernel(FissArgs fa, int NB)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    fsp::fissplus_search_ego<W, NW, CMAX>(fa, NB, (int)blockIdx.x, smem);
}

#ifndef FP_SEARCH_WAVES
#define FP_SEARCH_WAVES 4   // wavefronts per workgroup of the mid-size instances (C <= 2048)
#endif
#ifndef FP_SEARCH_NB
#define FP_SEARCH_NB 512    // buckets (a multiple of 64 x FP_SEARCH_WAVES; 0: 256 up to 1024 samples)
#endif

hipError_t launch_fissplus_search(const FissArgs& fa, hipStream_t stream)
{
    const int C = fa.ka.p.nd * fa.ka.p.nv * fa.ka.p.nt;
    FP_LDS_SLOTS(cfg_1);
    FP_LDS_SLOTS(cfg_4);
    FP_LDS_SLOTS(cfg_4s);
    FP_LDS_SLOTS(cfg_4w);
#ifndef FP_SEARCH_SMALL
#define FP_SEARCH_SMALL (4 * kWave)  // lattices up to this size: one wavefront
#endif
    if (C <= FP_SEARCH_SMALL) {  // single wavefront: the single-ego plan cycle (5 x 5 x 5)
        const int NB = C <= kWave ? kWave : 2 * kWave;
        const int bytes = fissplus_lds_bytes(C, NB);
        hipError_t e = ensure_dynamic_lds((const void*)fissplus_search_kernel<1, 1, 256>, bytes, cfg_1);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((fissplus_search_kernel<1, 1, 256>), dim3(fa.ka.b.B), dim3(kWave), bytes, stream, fa, NB);
    } else if (C <= 2048) {
        const int NB = FP_SEARCH_NB > 0 ? FP_SEARCH_NB : (C <= 1024 ? 256 : 512);
        const int bytes = fissplus_lds_bytes(C, NB);
        if (C <= 1024) {
            hipError_t e = ensure_dynamic_lds((const void*)fissplus_search_kernel<FP_SEARCH_WAVES, 1, 1024>, bytes, cfg_4s);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fissplus_search_kernel<FP_SEARCH_WAVES, 1, 1024>), dim3(fa.ka.b.B), dim3(FP_SEARCH_WAVES * kWave), bytes, stream, fa, NB);
        } else {
            hipError_t e = ensure_dynamic_lds((const void*)fissplus_search_kernel<FP_SEARCH_WAVES, 1, 2048>, bytes, cfg_4);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fissplus_search_kernel<FP_SEARCH_WAVES, 1, 2048>), dim3(fa.ka.b.B), dim3(FP_SEARCH_WAVES * kWave), bytes, stream, fa, NB);
        }
    } else {
        const int NB = 1024;
        const int bytes = fissplus_lds_bytes(C, NB);
        hipError_t e = ensure_dynamic_lds((const void*)fissplus_search_kernel<4, 2, 4096>, bytes, cfg_4w);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((fissplus_search_kernel<4, 2, 4096>), dim3(fa.ka.b.B), dim3(4 * kWave), bytes, stream, fa, NB);
    }
    return hipGetLastError();
}

}  // namespace fp
