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

// Production dense-lattice kernel (profile sharing + compacted collision).  Returns hipErrorInvalidValue when the
// problem does not fit it (LDS budget / index widths); launch_lattice then uses the lane-per-candidate kernel.
hipError_t launch_lattice_fused(const KernelArgs& ka, hipStream_t stream);
hipError_t launch_lattice_percand(const KernelArgs& ka, hipStream_t stream);
// Dispatcher used by the ABI.  which: 0 = auto (fused, else per-candidate), 1 = per-candidate, 2 = fused only.
hipError_t launch_lattice(const KernelArgs& ka, hipStream_t stream, int which);
// Winner epilogue: recompute the full series of trajectory best_idx[b] for every ego (one lane per time point).
hipError_t launch_winner_traj(const KernelArgs& ka, hipStream_t stream);
hipError_t launch_eval_trajs(const KernelArgs& ka, int K, const double* end_states, double* cost, uint32_t* flags, double* traj,
                             int stride, hipStream_t stream);

}  // namespace fp
