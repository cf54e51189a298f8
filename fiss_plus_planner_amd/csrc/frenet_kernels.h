// frenet_kernels.h - host-visible launch interface of the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/frenet_gpu.h"

namespace fp {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and not free: remember the largest size configured per
// (kernel, device).  `slot` is a function-local static array of kMaxDevices ints initialised to -1.  Several host threads may
// launch on one device (ShardedEngine(shards_per_device > 1), two contexts on two streams): the check-and-set is serialised, so
// the recorded size is always the size the runtime was last told (a smaller request can never land after a larger one).
constexpr int kMaxDevices = 64;
inline std::mutex& dynamic_lds_mutex()
{
    static std::mutex m;
    return m;
}
inline hipError_t ensure_dynamic_lds(const void* kernel, int bytes, int* slot)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(dynamic_lds_mutex());
    if (dev < 0 || dev >= kMaxDevices) return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (bytes > slot[dev]) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        slot[dev] = bytes;
    }
    return hipSuccess;
}
#define FP_LDS_SLOTS(name) static int name[fp::kMaxDevices] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}

// Everything a kernel needs, passed by value as the kernel argument block.
// All pointers are device addresses.
struct KernelArgs {
    fp_params p;
    fp_batch b;
    fp_result r;
    // [B][C] FP_FLAG_CURVATURE / KAPPA_D / KAPPA_DD of every lattice candidate (flat FOP order), written by
    // launch_curvature_flags ahead of the fused lattice kernel when p.curvature_mask is set; nullptr otherwise
    const uint8_t* curv_tbl = nullptr;
    // device copy of r.best_idx, written by the lattice kernels beside it and read by winner_traj_kernel (optional): the caller's
    // best_idx may be device-mapped HOST memory (bench.py's results need no copy that way), where the epilogue's first read costs
    // the link's round trip (winner kernel 10.9 -> 9.2 us on BASELINE configs[2])
    int32_t* idx_shadow = nullptr;
    // fp_plan_step (fused lattice kernel only): the workgroup that finds an ego's argmin also hands the ego over to its next state
    // (advance_ego, frenet_advance.h) - has_loop != 0: `loop` holds the caller's fp_loop_io (device addresses), b.skip = loop.done
    fp_loop_io loop = {};
    int has_loop = 0;
    // Epilogue workgroups appended to a multi-round fused lattice launch (three workgroups per CU): [B] ints, zero between launches.
    // The workgroup that publishes an ego's argmin sets its flag; the appended workgroups - dispatched last, i.e. into the slots the
    // draining launch leaves empty - wait for it, write the winner's series (r.best_traj) and clear it.  Needs idx_shadow.
    int32_t* epi_flag = nullptr;
    // The ctx's hand-over error word (device-mapped pinned host memory, or nullptr): an appended workgroup whose wait for its ego's flag
    // runs out leaves a code here instead of trapping (1: winner-series epilogue, 2: FISS+ search); the host reads it at the next call.
    int32_t* err_word = nullptr;
    // host-side launch hint (fp_ctx_set_option("lattice_occupancy")): 0 = auto, 2 / 3 = at most that many lattice workgroups per CU
    int occ_cap = 0;
    // host-side launch hints from the ctx: lattice workgroups the device holds at once at TWO per CU (fp_ctx: resident_groups = 2 x compute
    // units, or what fp_ctx_set_option("resident_groups") says - a process under a CU mask, a test that models a smaller device) and the
    // LDS of one CU in KB.  A launch takes the three- / four-per-CU instances when it has more egos than resident2 / 1.5 x resident2 and
    // the CU's LDS holds three / four of their layouts.
    int resident2 = 512;
    int lds_cu_kb = 160;
    // test hook (fp_ctx_set_option("handover_timeout_us"), FP_TEST_HOOKS only): how long an appended workgroup waits for its ego's flag
    // before it gives up; 0 = the production value (2 s).  A microsecond makes REAL waits run out, which is how the tests drive the
    // device side of the time-out path (error word, workgroup leaves without output, the launch completes) instead of injecting its result.
    int handover_timeout_us = 0;
};

// Inline inputs (latency regime of the FP_MEM_HOST entry, fused lattice kernel only): the per-ego arrays of a tiny batch travel
// INSIDE the kernel argument block - kernel arguments live in device memory on this platform (the host writes them through the
// BAR), so the kernel reads them like any other HBM data and no copy, copy kernel or dependency is enqueued for them.  on != 0:
// the fields d_samples, t_samples, v_samples, target_speed, ego, frame_of, scene_of, t_now of KernelArgs.b hold byte OFFSETS into
// bytes[] instead of addresses (skip must be NULL).
// publish != nullptr: the lattice kernel's first workgroup also copies the first n8 8-byte words of bytes[] to that device address -
// the later kernels of a multi-kernel call (fp_plan_fiss: search, refinement) read the same arrays from there, stream-ordered behind
// the lattice kernel, and the call enqueues no copy of any kind.
constexpr int kInlineMax = 1024;
struct InlineIn {
    int on = 0;
    int n8 = 0;
    void* publish = nullptr;
    unsigned char bytes[kInlineMax];
};

// Production dense-lattice kernel (profile sharing + compacted collision).  Returns hipErrorInvalidValue when the
// problem does not fit it (LDS budget / index widths); launch_lattice then uses the lane-per-candidate kernel.
// Arguments of the FISS / FISS+ batch kernels: the lattice arguments plus the dense tables and the fp_fiss_io arrays.
struct FissArgs {
    KernelArgs ka;
    fp_fiss_opts opts;
    fp_fiss_io io;
    const double* cost_tbl;    // [B][C] flat FOP order
    const uint32_t* flag_tbl;  // [B][C]
    int walk_jump = 1;         // FISS+ walk: skip the iterations below the first feasible sample's minimax level (frenet_fissplus.hip)
};

// The FISS+ search as workgroups appended to the fused lattice launch (lattice_fused_kernel's FISS instances): what the search needs
// beside the lattice's KernelArgs.  flag: [B] ints, zero between launches - an ego's lattice workgroup sets its flag when the ego's
// rows of the dense tables are written (agent-scope stores), the ego's search workgroup waits for it and clears it.
struct FissTail {
    fp_fiss_opts opts;
    fp_fiss_io io;
    int NB = 0;          // buckets of the ranking (fsp::fissplus_search_ego)
    int walk_jump = 1;
    int32_t* flag = nullptr;
};

// One wavefront per ego: coarse FISS / FISS+ search over the dense tables.
hipError_t launch_fiss_search(const FissArgs& fa, hipStream_t stream);
// The FISS+ walk in rank space (frenet_fissplus.hip); launch_fiss_search dispatches to it for FP_FISS_PLUS.
hipError_t launch_fissplus_search(const FissArgs& fa, hipStream_t stream);
// One workgroup per ego: FISS+ refinement rounds + cost-ordered validation of the refined trajectories.  table_kb = LDS budget
// of the per-ego fp32 pose-obstacle pair table (0: no table, pairs are read from the scene table).
// perm / dur: launch order and duration feedback, like launch_lattice_fused.
hipError_t launch_fiss_refine(const FissArgs& fa, hipStream_t stream, int table_kb, const int* perm = nullptr, int* dur = nullptr);

// part_scratch: device buffer of kTicketBytes (ticket counters, int per ego, ZERO before the first launch; the kernel leaves
// them zero) + B * nsplit * 16 bytes (partial argmins), or nullptr; nsplit > 1 = latency mode (B <= kTicketBytes / 4).
constexpr size_t kTicketBytes = 64 * 1024;
// *winner_done (optional): the kernel also wrote ka.r.best_traj / best_flags (asked for, and not in latency mode).
// perm (optional, nsplit == 1): workgroup i works on ego perm[i] (a permutation of [0, B): the launch order); dur (optional): every
// workgroup leaves its ego's duration there (10 ns ticks).
// group: time-horizon slices per barrier interval of the collision stages (<= 1: one at a time; n: up to n, as far as
// lattice_group_fit allows - for small lattices whose slices are latency bound).
// tail (needs part_scratch, nsplit == 1): the last `tail` dispatch slots of a multi-round launch are cut in two workgroups each (the
// launch's tail drains faster); < 0: auto for a device of -tail compute units (a quarter of a round of resident workgroups, only when the
// launch has more egos than stay resident); 0: off.
// *step_done (optional, ka.has_loop set): the launched instance hands the egos over to their next states itself (fp_plan_step); false:
// the caller launches advance_kernel behind it.
// ft / search_done (optional): the FISS+ search of every ego in workgroups appended to the launch (three-per-CU launches of lattices up
// to 1024 samples whose tables ka.r.cost_tbl / flag_tbl are written); *search_done = false: the caller launches the search kernel.
hipError_t launch_lattice_fused(const KernelArgs& ka, hipStream_t stream, void* part_scratch, int nsplit, bool* winner_done = nullptr,
                                const int* perm = nullptr, int* dur = nullptr, int group = 1, const InlineIn* inl = nullptr, int tail = 0,
                                bool* step_done = nullptr, const FissTail* ft = nullptr, bool* search_done = nullptr);
int lattice_group_fit(const fp_params& p, const fp_batch& b);
// launches of the fused lattice kernel by workgroups per CU ([0] two, [1] three, [2] four) since the library was loaded: process-wide
// counters behind fp_ctx_get_option("lattice_launches_2 / _3 / _4") - what the tests use to know which instance family they ran
long lattice_launches_per_cu(int which);
hipError_t launch_lattice_percand(const KernelArgs& ka, hipStream_t stream);
// Curvature flags of every lattice candidate -> out [B][C] (one workgroup per ego, one lane per candidate, spline in LDS).
hipError_t launch_curvature_flags(const KernelArgs& ka, uint8_t* out, hipStream_t stream);
// Dispatcher used by the ABI.  which: 0 = auto (fused, else per-candidate), 1 = per-candidate, 2 = fused only.
// inl (optional, see InlineIn): only with which == 2 semantics guaranteed by the caller (the problem fits the fused kernel).
hipError_t launch_lattice(const KernelArgs& ka, hipStream_t stream, int which, void* part_scratch, int nsplit, bool* winner_done = nullptr,
                          const int* perm = nullptr, int* dur = nullptr, int group = 1, const InlineIn* inl = nullptr, int tail = 0,
                          const FissTail* ft = nullptr, bool* search_done = nullptr);
// Winner epilogue: recompute the full series of trajectory best_idx[b] for every ego (one lane per time point).
// end_states = nullptr: series of lattice candidate ka.r.best_idx[b]; else [B][3] explicit (d, v, T) end states (NaN = none).
hipError_t launch_winner_traj(const KernelArgs& ka, const double* end_states, hipStream_t stream);
// FopPlusPlanner.plan from the dense tables + the FOP argmin (one wavefront per ego): out [B][2] = {popped, tie}, stats [B][4].
// skip (optional, fp_batch.skip): egos the lattice kernel did not plan get out = {0, 0} and keep their Stats.
hipError_t launch_fopplus_count(int B, int C, const double* cost_tbl, const uint32_t* flag_tbl, const int32_t* best_idx, const double* best_cost,
                                int32_t* out, int32_t* stats, const int32_t* skip, hipStream_t stream);
// fp_result.audit: near-tie / thin-contact bits of every ego from its dense tables (ka.r.cost_tbl / flag_tbl / best_idx / best_cost must be
// set; best_idx / best_cost are rewritten where a near tie is settled by point-by-point sums).  One workgroup per ego.
hipError_t launch_audit(const KernelArgs& ka, uint32_t* audit, hipStream_t stream);
// Frenet frame construction / Cartesian -> Frenet projection (frenet_frame.hip).
hipError_t launch_frames_build(int F, int NX, const int32_t* n, const double* points, double* knots, double* coef, hipStream_t stream);
hipError_t launch_from_state(const fp_batch& bt, const double* states, double* ego, hipStream_t stream);
// Closed-loop bookkeeping between two plan cycles (one lane per ego).
hipError_t launch_advance(const KernelArgs& ka, const int32_t* best_idx, const double* end_state, const fp_loop_io& io, hipStream_t stream);
// Series of EVERY lattice candidate: ka.r.best_traj [B*C][16][traj_stride], ka.r.best_flags [B*C] (N, M, truncated).
hipError_t launch_materialize_all(const KernelArgs& ka, hipStream_t stream);
hipError_t launch_eval_trajs(const KernelArgs& ka, int K, const double* end_states, double* cost, uint32_t* flags, double* traj,
                             int stride, int sparse, hipStream_t stream);

}  // namespace fp
