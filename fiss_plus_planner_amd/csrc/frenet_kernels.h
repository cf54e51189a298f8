// frenet_kernels.h - host-visible launch interface of the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/frenet_gpu.h"

namespace fp {

// Everything a kernel needs, passed by value as the kernel argument block.
// All pointers are device addresses.
struct KernelArgs {
    fp_params p;
    fp_batch b;
    fp_result r;
};

hipError_t launch_lattice_percand(const KernelArgs& ka, hipStream_t stream);
hipError_t launch_eval_trajs(const KernelArgs& ka, int K, const double* end_states, double* cost, uint32_t* flags, double* traj,
                             int stride, hipStream_t stream);

}  // namespace fp
