// frenet_abi.hip - the C ABI of libfrenetgpu.so (include/frenet_gpu.h).
//
// Host-side responsibilities only: argument validation, the per-ctx staging used by FP_MEM_HOST calls, error
// strings.  No planning arithmetic happens on the host: if there is no usable GPU every entry point fails with
// FP_ENODEV / FP_EHIP.
//
// FP_MEM_HOST staging (latency matters for the B = 1 drop-in planners): small arrays are packed into ONE pinned host
// block that mirrors the head of the device arena, so a call costs one H2D copy, the kernels, one D2H copy - instead of
// one transfer per array.  Arrays above kSmallMax bytes are copied directly (no extra host pass over big batches).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "frenet_kernels.h"

// Toolchain pin.  The kernels were validated on ROCm 7.2.0's clang 22 (AMD clang 22.0.0git roc-7.2.0): no kernel may spill a VGPR
// (tests/test_abi_cpu.py:test_no_kernel_spills_a_vgpr - this compiler's spill placement in divergent loop exits is wrong), the build
// needs -mllvm -disable-machine-licm, and the four-per-CU instances sit exactly on their 64-VGPR / 80-SGPR budgets.  Another major
// version is another code generator: rebuild with -DFP_ANY_COMPILER only together with tools/resource_usage.py and the GPU test suite.
#if !defined(FP_ANY_COMPILER) && defined(__clang_major__) && __clang_major__ != 22
#error "libfrenetgpu was validated on ROCm 7.2.0 (clang 22); build with EXTRA=-DFP_ANY_COMPILER after re-checking spills (tools/resource_usage.py) and the GPU tests"
#endif

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return fail(FP_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e));     \
    } while (0)
#define FP_TRY(expr)                  \
    do {                              \
        int _rc = (expr);             \
        if (_rc != FP_OK) return _rc; \
    } while (0)
#define LAUNCH_TRY(expr, what)                                                                          \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(FP_EHIP, what " launch failed: %s", hipGetErrorString(_e));   \
    } while (0)

constexpr size_t kAlign = 256;
constexpr size_t kSmallRegion = 4u << 20;  // device bytes mirrored by the pinned host block
constexpr size_t kSmallMax = 64u << 10;    // arrays up to this size travel through the pinned block
constexpr size_t kZeroCopyInMax = 256u << 10;  // latency regime: inputs the kernels read straight from the pinned block (all of a call's arrays together)

inline size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

// grow-only device buffer
struct DeviceBuf {
    char* base = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return FP_OK;
        if (base) {
            hipError_t e = hipFree(base);
            base = nullptr;
            cap = 0;
            if (e != hipSuccess) return fail(FP_EHIP, "hipFree failed: %s", hipGetErrorString(e));
        }
        const size_t want = bytes + bytes / 4 + (1u << 20);
        hipError_t e = hipMalloc((void**)&base, want);
        if (e != hipSuccess) return fail(FP_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        cap = want;
        return FP_OK;
    }
};

}  // namespace

// launches between two fetches of the duration table: 8 while a batch's order is young, doubling with every order that lands up to 64 -
// a fetch and the upload of the order it yields are two copy commands on the launch stream (~13 us of stream time each: a copy engine
// hand-over), 2.5 % of a 2048-ego step at one pair per 8 launches; a resident batch's durations drift slowly
constexpr int kOrderRefresh = 8, kOrderRefreshMax = 64;

// State of one kernel's feedback-directed launch order (see lattice_order_before).
struct LaunchOrder {
    DeviceBuf buf;              // [dur: int x cap][perm: int x cap]
    int* host = nullptr;        // pinned: [dur x cap][perm x cap]
    int cap = 0;                // egos the buffers hold
    int valid_B = 0;            // the uploaded order is a permutation of [0, valid_B)
    int dur_B = 0;              // batch size of the durations in flight / on the device
    int since = 0;              // launches since the last fetch was enqueued
    int interval = 8;           // launches between two fetches: kOrderRefresh at first, doubled after every order that landed (up to kOrderRefreshMax)
    bool pending = false;       // a fetch is in flight (event)
    hipEvent_t event = nullptr;
    void release()
    {
        if (buf.base) (void)hipFree(buf.base);
        if (host) (void)hipHostFree(host);
        if (event) (void)hipEventDestroy(event);
    }
};

// The feedback orders of one kernel, one per RESIDENT BATCH (keyed by the address of the batch's ego-state array): an order is only ever
// applied to the batch it was learnt on.  (Until round 6 there was one order per kernel and batch SIZE: a caller cycling through
// several resident batches of one size dispatched each in the order another one had left behind - for the refinement kernel, whose
// launch is one round of workgroups with only the egos that found a trajectory doing work, a foreign order is worse than none:
// rocprofv3 82 us per launch against 55 with its own.)  kOrderSlots batches, least recently used one replaced.
constexpr int kOrderSlots = 8;
struct OrderSet {
    LaunchOrder slot[kOrderSlots];
    const void* key[kOrderSlots] = {};
    unsigned long used[kOrderSlots] = {};
    unsigned long tick = 0;
    LaunchOrder& of(const void* k)
    {
        int lru = 0;
        for (int i = 0; i < kOrderSlots; ++i) {
            if (key[i] == k && used[i]) { used[i] = ++tick; return slot[i]; }
            if (used[i] < used[lru]) lru = i;
        }
        LaunchOrder& o = slot[lru];
        if (o.pending) { (void)hipEventSynchronize(o.event); o.pending = false; }  // (the fetch in flight belongs to the batch that leaves)
        o.valid_B = o.dur_B = 0;
        o.since = 0;
        o.interval = kOrderRefresh;
        key[lru] = k;
        used[lru] = ++tick;
        return o;
    }
    void forget() { for (auto& o : slot) o.valid_B = 0; }
    void release() { for (auto& o : slot) o.release(); }
};

struct fp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  // used by FP_MEM_HOST calls
    DeviceBuf arena;               // [ small region (kSmallRegion) | large region ] staging of FP_MEM_HOST calls
    char* pinned = nullptr;        // kSmallRegion bytes of pinned host memory mirroring the small region
    DeviceBuf scratch;             // intermediate tables of multi-kernel entry points (fp_plan_fiss)
    // frame + scene tables of tagged FP_MEM_HOST calls (fp_batch.tables_tag): a few sets, least recently used one replaced - two
    // planners that take turns on one ctx (FOP and FISS+ on the same scenario, say) both keep theirs
    struct TableSet {
        DeviceBuf buf;
        int key[7] = {0, 0, 0, 0, 0, 0, 0};    // {tag, F, NX, S, T_obs, n_obs, poly_stride (0: rectangles only)} of what buf holds (tag 0: nothing)
        size_t off[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // nx, knots, coef, obs_pose, obs_dims, final_time_step, obs_poly, obs_nvert
        unsigned long used = 0;                // tick of the last call that read it
    };
    static constexpr int kTableSets = 4;
    TableSet tables[kTableSets];
    unsigned long tables_tick = 0;
    DeviceBuf parts;               // latency-mode lattice launch: [ticket counters, fixed-size region][partial argmins]
    int lattice_kernel = 0;        // fp_ctx_set_option("lattice_kernel")
    int lattice_split = 0;         // fp_ctx_set_option("lattice_split"): 0 auto, 1 never, 2 always
    int inline_inputs = 1;         // fp_ctx_set_option("inline_inputs"): fp_plan_dense(FP_MEM_HOST) of a tiny batch with cached tables passes the per-ego arrays inside the lattice kernel's argument block
    int stage_kernel = 1;          // fp_ctx_set_option("stage_kernel"): the latency regime's inputs reach the device by a copy kernel instead of a copy command
    int zero_copy_in = 0;          // fp_ctx_set_option("zero_copy_in"): FP_MEM_HOST calls of a handful of egos read inputs from pinned host memory: 0 never (default: even a few hundred bytes read over the link cost every kernel of the call a round trip - measured slower than the copy kernel), 1 the per-ego arrays of a call whose tables are cached (tables_tag), 2 everything
    int lattice_occupancy = 0;     // fp_ctx_set_option("lattice_occupancy"): 0 auto, 2 / 3: at most that many lattice workgroups per CU
    int lattice_tail = 0;          // fp_ctx_set_option("lattice_tail"): 0 auto, 1 never, n >= 2: the last n dispatch slots of a multi-round launch are cut in two workgroups
    int lattice_group = 0;         // fp_ctx_set_option("lattice_group"): 0 auto, 1 never, n >= 2: up to n slices per barrier interval
    int refine_table_kb = 96;      // fp_ctx_set_option("refine_table_kb")
    int fiss_stages = 3;           // fp_ctx_set_option("fiss_stages"): timing diagnostic, 3 = the whole pipeline
    int fiss_jump = 1;             // fp_ctx_set_option("fiss_jump"): FISS+ walk skips ahead to the first feasible sample's level
    int validate = 0;              // fp_ctx_set_option("validate"): FP_MEM_DEVICE calls range-check the batch's index arrays first
    DeviceBuf validate_buf;        // two ints on the device: first failing check, index
    int* validate_host = nullptr;  // ... and their pinned mirror
    int lattice_winner = 0;        // fp_ctx_set_option("lattice_winner"): 0 auto, 1 inside the lattice kernel, 2 its own launch
    int fiss_fused = 1;            // fp_ctx_set_option("fiss_fused"): 1 = the FISS+ search of a multi-round batch runs in workgroups appended to the lattice launch; 0 = always its own launch
    int resident_groups = 512;     // lattice workgroups the device holds at once: 2 per CU (128-VGPR budget, 512 threads each); fp_ctx_set_option("resident_groups")
    int lds_cu_kb = 160;           // LDS of one compute unit (hipDeviceProp_t::maxSharedMemoryPerMultiProcessor)
    // feedback-directed launch order of the multi-round lattice launch (fp_ctx_set_option("lattice_order")): every workgroup
    // leaves its ego's duration in dur_dev; now and then the host fetches them (async copy + event, never a wait), sorts the egos
    // longest-first and uploads the order the following launches dispatch in.  A stale or missing order only costs speed.
    int lattice_order = 1;
    int lattice_launches = 0, lattice_ordered_launches = 0;  // fp_ctx_get_option counters
    OrderSet order_lattice, order_refine;
    DeviceBuf idx_shadow;          // [B] device copy of best_idx for the winner kernel of a dense call (KernelArgs::idx_shadow)
    DeviceBuf epi_flags;           // [B] hand-over flags of the epilogue workgroups appended to a multi-round lattice launch (KernelArgs::epi_flag); zero between launches
    DeviceBuf curv_buf;            // [B][C] curvature flag bytes of the lattice (fp_params.curvature_mask), written ahead of the fused kernel
    // Workgroups appended to a lattice launch (winner-series epilogue, FISS+ search) wait for flags that other workgroups of the SAME
    // launch set; that they cannot starve rests on the workgroup distributors starting workgroups in index order - observed on gfx942 /
    // gfx950, documented nowhere.  appended_ok: the device is one of those architectures and no hand-over of this ctx ever timed out.
    // hand_err: one int of device-mapped, coherent host memory; a wait that runs out leaves a code there instead of trapping
    // (KernelArgs::err_word), and the next call on the ctx reports it, resets the flags and stops offering appended workgroups.
    bool appended_ok = false;
    int32_t* hand_err = nullptr;
    int handover_timeout_us = 0;   // fp_ctx_set_option("handover_timeout_us") (FP_TEST_HOOKS): KernelArgs::handover_timeout_us
    // fp_ctx_set_option("overlap"): consecutive INDEPENDENT FP_MEM_DEVICE dense calls alternate between two internal streams (this ctx's
    // and a twin ctx's, each with its own scratch: tickets, hand-over flags, launch order), so the draining tail of one launch runs beside
    // the ramp of the next.  See fp_plan_dense / fp_ctx_join in include/frenet_gpu.h for what the caller's stream is ordered after.
    int overlap = 0;
    bool is_twin = false;
    fp_ctx* twin = nullptr;
    hipEvent_t ov_fork = nullptr, ov_done[2] = {nullptr, nullptr};
    bool ov_pending[2] = {false, false};
    int ov_next = 0;
    const void* ov_last_p[12] = {};  // the output arrays of the call in flight on the OTHER internal stream (independence check)
    int overlapped_calls = 0;      // fp_ctx_get_option("overlapped_calls"): dense calls that started without waiting for their predecessor
};

namespace {

// Latency regime: the pinned window goes to the arena by a copy KERNEL (16 bytes per lane, read straight from the pinned host block)
// instead of a copy command: for ~100 KB the copy engine takes ~9 us and the kernel behind it starts ~8 us after the copy ends
// (cross-engine dependency); a kernel-to-kernel dependency on the same queue costs ~2-3 us and the copy itself ~4 us.
__global__ __launch_bounds__(256) void stage_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, int n16)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// One FP_MEM_HOST call: reserve(), in()/in_mut() for every input, flush_in(), out() for every output, launch kernels,
// fetch_out().
class HostStage {
  public:
    explicit HostStage(fp_ctx* ctx) : ctx_(ctx) {}

    // zero_copy_out (latency regime, a handful of egos): small outputs are written by the kernels straight into the pinned
    // host block (it is device-visible) - no D2H command at all after the launch, only the stream synchronisation.  Larger
    // batches keep the outputs in HBM (kernels of the same call read each other's outputs) and fetch them with one copy.
    // The same regime reads its INPUTS from the pinned block too (zero-copy in): the host's memcpy into the block is the whole
    // transfer, the kernels fetch what they touch over the link (a few dependent reads, ~1 us more each than from HBM) - no copy
    // command, no blit kernel, no gap behind them: a single-ego plan cycle is ~15 us shorter.  Only while the inputs of the call
    // stay below kZeroCopyInMax bytes in total (a kernel re-reads parts of them; beyond that the copy engine wins).
    int reserve(size_t large_bytes, bool zero_copy_out = false)
    {
        FP_TRY(ctx_->arena.reserve(kSmallRegion + large_bytes + kAlign));
        zero_copy_out_ = zero_copy_out;
        zero_copy_in_ = zero_copy_out && ctx_->zero_copy_in == 2;
        small_ = 0;
        small_dirty_ = false;
        large_ = kSmallRegion;
        outs_.clear();
        small_out_lo_ = small_out_hi_ = 0;
        return FP_OK;
    }
    // bytes a `count`-element array may add to the large region
    template <typename T>
    static size_t need(size_t count) { return align_up(sizeof(T) * count) + kAlign; }
    fp_ctx* ctx() const { return ctx_; }
    bool latency() const { return zero_copy_out_; }
    // (latency regime with the big tables resident on the device: the per-ego arrays that are left are a few hundred bytes - the
    // kernels read them from the pinned block, one more microsecond in their first round of loads, and no copy is enqueued at all)
    void small_inputs_only() { if (zero_copy_out_ && ctx_->zero_copy_in != 0) zero_copy_in_ = true; }

    template <typename T>
    int in(const T* host, size_t count, const T** dev)
    {
        const size_t bytes = sizeof(T) * count;
        if (count == 0) { *dev = (const T*)(ctx_->arena.base + large_); return FP_OK; }
        if (zero_copy_in_ && align_up(small_) + bytes <= kZeroCopyInMax) {
            small_ = align_up(small_);
            memcpy(ctx_->pinned + small_, host, bytes);
            *dev = (const T*)(ctx_->pinned + small_);
            small_ += bytes;
            return FP_OK;
        }
        if (zero_copy_in_) {  // does not fit the zero-copy window: its own copy (the mirrored window below is not flushed in this mode)
            large_ = align_up(large_);
            T* d = (T*)(ctx_->arena.base + large_);
            large_ += bytes;
            HIP_TRY(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, ctx_->stream));
            *dev = d;
            return FP_OK;
        }
        // (latency regime: bigger arrays too - one transfer for the whole call instead of one more copy command per array)
        if (bytes <= (zero_copy_out_ ? kZeroCopyInMax : kSmallMax) && align_up(small_) + bytes <= kSmallRegion) {
            small_ = align_up(small_);
            memcpy(ctx_->pinned + small_, host, bytes);
            *dev = (const T*)(ctx_->arena.base + small_);
            small_ += bytes;
            small_dirty_ = true;  // the arena's copy of the window has to be brought up to date (flush_in)
            return FP_OK;
        }
        large_ = align_up(large_);
        T* d = (T*)(ctx_->arena.base + large_);
        large_ += bytes;
        HIP_TRY(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, ctx_->stream));
        *dev = d;
        return FP_OK;
    }
    template <typename T>
    int in_mut(T* host, size_t count, T** dev)  // in/out array: staged in, copied straight back by fetch_out
    {
        const size_t bytes = sizeof(T) * count;
        if (zero_copy_out_ && count > 0 && bytes <= kSmallMax && align_up(small_) + bytes <= kSmallRegion) {
            // latency regime: the kernels read and update the array in the pinned host block itself - no copy either way
            small_ = align_up(small_);
            memcpy(ctx_->pinned + small_, host, bytes);
            *dev = (T*)(ctx_->pinned + small_);
            outs_.push_back({host, small_, (char*)*dev, bytes, true});
            small_ += bytes;
            return FP_OK;
        }
        const T* c = nullptr;
        FP_TRY(in((const T*)host, count, &c));
        *dev = const_cast<T*>(c);
        outs_.push_back({host, 0, (char*)*dev, bytes, false});
        return FP_OK;
    }
    int flush_in()  // the output window of the small region starts after the inputs
    {
        if (small_ > 0 && small_dirty_ && !zero_copy_in_) {  // (arrays the kernels address in the pinned block itself need no mirror)
            if (zero_copy_out_ && ctx_->stage_kernel) {
                const int n16 = (int)((small_ + 15) / 16);  // (both blocks are 256-byte aligned and kSmallRegion long)
                const int blocks = (n16 + 255) / 256;
                hipLaunchKernelGGL(stage_copy_kernel, dim3(blocks < 512 ? blocks : 512), dim3(256), 0, ctx_->stream, (uint4*)ctx_->arena.base, (const uint4*)ctx_->pinned, n16);
                HIP_TRY(hipGetLastError());
            } else {
                HIP_TRY(hipMemcpyAsync(ctx_->arena.base, ctx_->pinned, small_, hipMemcpyHostToDevice, ctx_->stream));
            }
        }
        small_out_lo_ = small_out_hi_ = align_up(small_);
        return FP_OK;
    }
    template <typename T>
    T* out(T* host, size_t count)
    {
        const size_t bytes = sizeof(T) * count;
        if (!host || count == 0) return nullptr;
        if (bytes <= kSmallMax && align_up(small_out_hi_) + bytes <= kSmallRegion) {
            small_out_hi_ = align_up(small_out_hi_);
            T* d = (T*)((zero_copy_out_ ? ctx_->pinned : ctx_->arena.base) + small_out_hi_);
            outs_.push_back({host, small_out_hi_, (char*)d, bytes, true});
            small_out_hi_ += bytes;
            return d;
        }
        T* d = temp<T>(count);
        outs_.push_back({host, 0, (char*)d, bytes, false});
        return d;
    }
    template <typename T>
    T* temp(size_t count)  // device-only scratch inside the arena
    {
        large_ = align_up(large_);
        T* d = (T*)(ctx_->arena.base + large_);
        large_ += sizeof(T) * count;
        return d;
    }
    int fetch_out()
    {
        if (small_out_hi_ > small_out_lo_ && !zero_copy_out_)
            HIP_TRY(hipMemcpyAsync(ctx_->pinned + small_out_lo_, ctx_->arena.base + small_out_lo_, small_out_hi_ - small_out_lo_,
                                   hipMemcpyDeviceToHost, ctx_->stream));
        for (const Out& o : outs_)
            if (!o.via_pinned) HIP_TRY(hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, ctx_->stream));
        HIP_TRY(hipStreamSynchronize(ctx_->stream));
        for (const Out& o : outs_)
            if (o.via_pinned) memcpy(o.host, ctx_->pinned + o.small_off, o.bytes);
        return FP_OK;
    }

  private:
    struct Out {
        void* host;
        size_t small_off;  // offset in the small region (via_pinned)
        char* dev;
        size_t bytes;
        bool via_pinned;
    };
    fp_ctx* ctx_;
    bool zero_copy_out_ = false, zero_copy_in_ = false, small_dirty_ = false;
    size_t small_ = 0, large_ = 0, small_out_lo_ = 0, small_out_hi_ = 0;
    std::vector<Out> outs_;
};

int check_params(const fp_params* p)
{
    if (!p) return fail(FP_EINVAL, "params is NULL");
    if (p->nd < 1 || p->nv < 1 || p->nt < 1) return fail(FP_EINVAL, "lattice sizes must be >= 1 (got %d,%d,%d)", p->nd, p->nv, p->nt);
    if ((long)p->nd * p->nv * p->nt > FP_MAX_CAND) return fail(FP_ELIMIT, "nd*nv*nt = %ld exceeds FP_MAX_CAND", (long)p->nd * p->nv * p->nt);
    if (p->check_stride < 1) return fail(FP_EINVAL, "check_stride must be >= 1");
    if (!(p->tick_t > 0)) return fail(FP_EINVAL, "tick_t must be > 0");
    if (p->points_max < 0 || p->points_max > FP_MAX_POINTS) return fail(FP_ELIMIT, "points_max=%d outside 0..FP_MAX_POINTS", p->points_max);
    if (p->curvature_mask && (!(p->max_curvature >= 0) || !(p->max_kappa_d >= 0) || !(p->max_kappa_dd >= 0)))
        return fail(FP_EINVAL, "curvature_mask is set but max_curvature / max_kappa_d / max_kappa_dd are not all >= 0");
    return FP_OK;
}

// Series layout of a call: columns per row (0 = FP_DEFAULT_STRIDE).  Host calls can check it against the batch's time samples.
int traj_stride_of(int32_t requested, int* stride)
{
    if (requested < 0) return fail(FP_EINVAL, "traj_stride must be >= 0");
    *stride = requested > 0 ? requested : FP_DEFAULT_STRIDE;
    return FP_OK;
}
int check_stride_host(const fp_params* p, const fp_batch* b, int stride)
{
    for (int k = 0; k < p->nt; ++k)
        if (ceil(b->t_samples[k] / p->tick_t) > stride)
            return fail(FP_EINVAL, "traj_stride=%d is smaller than the %g points of t_samples[%d]=%g", stride, ceil(b->t_samples[k] / p->tick_t), k, b->t_samples[k]);
    return FP_OK;
}

// Points per trajectory of a FP_MEM_HOST call (fp_params.points_max is the device callers' announcement; host calls look themselves):
// the largest ceil(T / tick_t) over the time samples (+ `extra_T`: explicit end states / refinement bounds of the call), 0 when the
// fast paths' FP_FAST_POINTS hold it.
int host_points_max(const fp_params* p, const fp_batch* b, const double* extra_T, size_t n_extra, size_t extra_step)
{
    double n = 0.0;
    for (int k = 0; k < p->nt; ++k) n = fmax(n, ceil(b->t_samples[k] / p->tick_t));
    for (size_t i = 0; i < n_extra; ++i) {
        const double v = ceil(extra_T[i * extra_step] / p->tick_t);
        if (v == v) n = fmax(n, v);
    }
    return n > FP_FAST_POINTS ? (n > FP_MAX_POINTS ? FP_MAX_POINTS : (int)n) : 0;
}
bool big_points(const fp_params& p) { return p.points_max > FP_FAST_POINTS; }

int check_batch(const fp_batch* b)
{
    if (!b) return fail(FP_EINVAL, "batch is NULL");
    if (b->B < 0 || b->F < 1 || b->NX < 2) return fail(FP_EINVAL, "bad batch sizes B=%d F=%d NX=%d", b->B, b->F, b->NX);
    if (b->NX > FP_MAX_KNOTS) return fail(FP_ELIMIT, "NX=%d exceeds FP_MAX_KNOTS", b->NX);
    if (!b->d_samples || !b->t_samples || !b->v_samples || !b->target_speed || !b->ego || !b->frame_of || !b->scene_of ||
        !b->t_now || !b->nx || !b->knots || !b->coef)
        return fail(FP_EINVAL, "batch has a NULL array");
    if (b->S > 0 && b->n_obs > 0 && (!b->obs_pose || !b->obs_dims || !b->final_time_step))
        return fail(FP_EINVAL, "batch has obstacles but a NULL obstacle array");
    if (b->S > 0 && b->n_obs > 0 && b->obs_nvert) {
        if (!b->obs_poly) return fail(FP_EINVAL, "obs_nvert is set but obs_poly is NULL");
        if (b->poly_stride < 3 || b->poly_stride > FP_MAX_POLY_VERTS) return fail(FP_ELIMIT, "poly_stride=%d outside 3..FP_MAX_POLY_VERTS", b->poly_stride);
    }
    return FP_OK;
}

// Host-side validation of the index arrays (only possible for FP_MEM_HOST calls).
int check_batch_host(const fp_params* p, const fp_batch* b)
{
    for (int i = 0; i < b->B; ++i) {
        if (b->frame_of[i] < 0 || b->frame_of[i] >= b->F) return fail(FP_EINVAL, "frame_of[%d]=%d out of range", i, b->frame_of[i]);
        if (b->scene_of[i] >= b->S) return fail(FP_EINVAL, "scene_of[%d]=%d out of range", i, b->scene_of[i]);
        if (b->t_now[i] < 0) return fail(FP_EINVAL, "t_now[%d]=%d is negative", i, b->t_now[i]);
    }
    for (int f = 0; f < b->F; ++f)
        if (b->nx[f] < 2 || b->nx[f] > b->NX) return fail(FP_EINVAL, "nx[%d]=%d out of range", f, b->nx[f]);
    for (int k = 0; k < p->nt; ++k) {
        const double n = b->t_samples[k] / p->tick_t;
        if (!(n > 0) || n > FP_MAX_POINTS) return fail(FP_ELIMIT, "t_samples[%d]=%g needs more than FP_MAX_POINTS points", k, b->t_samples[k]);
    }
    if (b->S > 0 && b->n_obs > 0 && b->obs_nvert) {  // polygon columns: vertex count, ring orientation, and the box the broad phases test
        for (long c = 0; c < (long)b->S * b->n_obs; ++c) {
            const int n = b->obs_nvert[c];
            if (n == 0) continue;
            if (n < 3 || n > b->poly_stride) return fail(FP_EINVAL, "obs_nvert[%ld]=%d outside 3..poly_stride=%d", c, n, b->poly_stride);
            const double* v = b->obs_poly + (size_t)c * 2 * b->poly_stride;
            double mx = 0.0, my = 0.0, area2 = 0.0;
            for (int i = 0; i < n; ++i) {
                const double* a = v + 2 * i;
                const double* q = v + 2 * ((i + 1) % n);
                if (!(a[0] == a[0]) || !(a[1] == a[1])) return fail(FP_EINVAL, "obs_poly column %ld has a NaN vertex", c);
                mx = fmax(mx, fabs(a[0])); my = fmax(my, fabs(a[1]));
                area2 += a[0] * q[1] - a[1] * q[0];
            }
            if (!(area2 > 0.0)) return fail(FP_EINVAL, "obs_poly column %ld is not counter-clockwise (twice its signed area = %g)", c, area2);
            // convex: the narrow phase treats the ring as the intersection of its edge half-planes - a reflex vertex would silently shrink
            // the obstacle to its kernel (missed collisions).  Every turn must be to the left; collinear vertices pass (the tolerance
            // covers the rounding of a cross product of coordinates ~ mx, my).
            const double turn_tol = 64.0 * 2.220446049250313e-16 * (mx + my) * (mx + my);
            for (int i = 0; i < n; ++i) {
                const double* a = v + 2 * i;
                const double* q = v + 2 * ((i + 1) % n);
                const double* r = v + 2 * ((i + 2) % n);
                const double cr = (q[0] - a[0]) * (r[1] - q[1]) - (q[1] - a[1]) * (r[0] - q[0]);
                if (!(cr >= -turn_tol)) return fail(FP_EINVAL, "obs_poly column %ld is not convex (right turn at vertex %d, cross product %g): cut it into convex pieces (obstacles.shape_columns)", c, (i + 1) % n, cr);
            }
            if (!(b->obs_dims[2 * c] >= 2.0 * mx) || !(b->obs_dims[2 * c + 1] >= 2.0 * my))
                return fail(FP_EINVAL, "obs_dims of polygon column %ld (%g x %g) does not contain its vertices (needs %g x %g)", c, b->obs_dims[2 * c], b->obs_dims[2 * c + 1], 2.0 * mx, 2.0 * my);
        }
    }
    return FP_OK;
}

size_t batch_need(const fp_params* p, const fp_batch* b)
{
    const size_t B = (size_t)b->B, fn = (size_t)b->F * b->NX, so = (size_t)b->S * b->n_obs;
    return HostStage::need<double>(p->nd) + HostStage::need<double>(p->nt) + HostStage::need<double>(B * p->nv) + HostStage::need<double>(B) +
           HostStage::need<double>(B * 6) + 4 * HostStage::need<int32_t>(B) + HostStage::need<int32_t>(b->F) + HostStage::need<double>(fn) +
           HostStage::need<double>(fn * 8) + HostStage::need<double>(so * b->T_obs * 4) + HostStage::need<double>(so * 2) +
           HostStage::need<int32_t>(b->S) + (b->obs_nvert ? HostStage::need<double>(so * 2 * b->poly_stride) + HostStage::need<int32_t>(so) : 0);
}

// Inline inputs of a multi-kernel call (fp::InlineIn::publish): besides the eight per-ego arrays of the batch the blob carries
// the call's other small inputs (`extra`), the lattice kernel copies it to a device mirror, and `dev_pub` / the extras' `dev`
// get the mirror's addresses - what the kernels behind the lattice kernel read.
struct InlineExtra { const void* src; size_t bytes; const void** dev; };
int stage_batch(HostStage& hs, const fp_params* p, const fp_batch* b, fp_batch* dev, fp::InlineIn* inl = nullptr,
                fp_batch* dev_pub = nullptr, const InlineExtra* extra = nullptr, int n_extra = 0)
{
    *dev = *b;
    const bool has_obs = b->S > 0 && b->n_obs > 0;
    const bool has_poly = has_obs && b->obs_nvert != nullptr;
    if (!has_poly) { dev->obs_poly = nullptr; dev->obs_nvert = nullptr; dev->poly_stride = 0; }
    bool tables_resident = false;
    if (b->tables_tag != 0) {
        // the frame / scene tables of a tagged call live in their own device buffer across calls (fp_batch.tables_tag)
        fp_ctx* ctx = hs.ctx();
        const int key[7] = {b->tables_tag, b->F, b->NX, b->S, b->T_obs, has_obs ? b->n_obs : 0, has_poly ? b->poly_stride : 0};
        const void* src[8] = {b->nx, b->knots, b->coef, b->obs_pose, b->obs_dims, b->final_time_step, b->obs_poly, b->obs_nvert};
        const size_t bytes[8] = {sizeof(int32_t) * (size_t)b->F, sizeof(double) * (size_t)b->F * b->NX, sizeof(double) * (size_t)b->F * 8 * b->NX,
                                 has_obs ? sizeof(double) * (size_t)b->S * b->T_obs * b->n_obs * 4 : 0, has_obs ? sizeof(double) * (size_t)b->S * b->n_obs * 2 : 0,
                                 has_obs ? sizeof(int32_t) * (size_t)b->S : 0, has_poly ? sizeof(double) * (size_t)b->S * b->n_obs * 2 * b->poly_stride : 0,
                                 has_poly ? sizeof(int32_t) * (size_t)b->S * b->n_obs : 0};
        fp_ctx::TableSet* ts = nullptr;
        fp_ctx::TableSet* lru = &ctx->tables[0];
        for (auto& t : ctx->tables) {
            if (memcmp(key, t.key, sizeof(key)) == 0) { ts = &t; break; }
            if (t.used < lru->used) lru = &t;
        }
        if (!ts) {  // upload into the least recently used set
            ts = lru;
            size_t total = 0;
            for (int i = 0; i < 8; ++i) { ts->off[i] = total; total += align_up(bytes[i]); }
            ts->key[0] = 0;  // (nothing valid while the upload is being set up)
            if (total + kAlign > ts->buf.cap) {
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                FP_TRY(ts->buf.reserve(total + kAlign));
            }
            for (int i = 0; i < 8; ++i)
                if (bytes[i]) HIP_TRY(hipMemcpyAsync(ts->buf.base + ts->off[i], src[i], bytes[i], hipMemcpyHostToDevice, ctx->stream));
            memcpy(ts->key, key, sizeof(key));
        }
        ts->used = ++ctx->tables_tick;
        char* tb = ts->buf.base;
        dev->nx = (const int32_t*)(tb + ts->off[0]);
        dev->knots = (const double*)(tb + ts->off[1]);
        dev->coef = (const double*)(tb + ts->off[2]);
        dev->obs_pose = (const double*)(tb + ts->off[3]);
        dev->obs_dims = (const double*)(tb + ts->off[4]);
        dev->final_time_step = (const int32_t*)(tb + ts->off[5]);
        if (has_poly) { dev->obs_poly = (const double*)(tb + ts->off[6]); dev->obs_nvert = (const int32_t*)(tb + ts->off[7]); }
        tables_resident = true;
        hs.small_inputs_only();
    }
#define PUSH(field, count) FP_TRY(hs.in(b->field, (size_t)(count), &dev->field))
    // Inline inputs (fp::InlineIn): with the tables resident, what is left of a tiny batch fits the kernel's argument block
    bool inlined = false;
    if (inl && tables_resident && hs.latency() && !b->skip) {
        const void* src[8] = {b->d_samples, b->t_samples, b->v_samples, b->target_speed, b->ego, b->frame_of, b->scene_of, b->t_now};
        const size_t bytes[8] = {sizeof(double) * (size_t)p->nd, sizeof(double) * (size_t)p->nt, sizeof(double) * (size_t)b->B * p->nv, sizeof(double) * (size_t)b->B,
                                 sizeof(double) * (size_t)b->B * 6, sizeof(int32_t) * (size_t)b->B, sizeof(int32_t) * (size_t)b->B, sizeof(int32_t) * (size_t)b->B};
        size_t off[8], total = 0;
        for (int i = 0; i < 8; ++i) { off[i] = total; total += (bytes[i] + 7) & ~(size_t)7; }
        size_t total_x = total;
        for (int i = 0; i < n_extra; ++i) total_x += (extra[i].bytes + 7) & ~(size_t)7;
        if ((dev_pub ? total_x : total) <= (size_t)fp::kInlineMax) {
            for (int i = 0; i < 8; ++i) memcpy(inl->bytes + off[i], src[i], bytes[i]);
            dev->d_samples = (const double*)off[0]; dev->t_samples = (const double*)off[1]; dev->v_samples = (const double*)off[2];
            dev->target_speed = (const double*)off[3]; dev->ego = (const double*)off[4]; dev->frame_of = (const int32_t*)off[5];
            dev->scene_of = (const int32_t*)off[6]; dev->t_now = (const int32_t*)off[7];
            inl->on = 1;
            inlined = true;
            if (dev_pub) {  // the device mirror the lattice kernel fills for the kernels behind it
                const unsigned char* mirror = hs.temp<unsigned char>(fp::kInlineMax);
                *dev_pub = *dev;
                dev_pub->d_samples = (const double*)(mirror + off[0]); dev_pub->t_samples = (const double*)(mirror + off[1]);
                dev_pub->v_samples = (const double*)(mirror + off[2]); dev_pub->target_speed = (const double*)(mirror + off[3]);
                dev_pub->ego = (const double*)(mirror + off[4]); dev_pub->frame_of = (const int32_t*)(mirror + off[5]);
                dev_pub->scene_of = (const int32_t*)(mirror + off[6]); dev_pub->t_now = (const int32_t*)(mirror + off[7]);
                size_t o = total;
                for (int i = 0; i < n_extra; ++i) {
                    memcpy(inl->bytes + o, extra[i].src, extra[i].bytes);
                    *extra[i].dev = mirror + o;
                    o += (extra[i].bytes + 7) & ~(size_t)7;
                }
                inl->publish = (void*)mirror;
                inl->n8 = (int)((total_x + 7) / 8);
            }
        }
    }
    if (!inlined) {
        PUSH(d_samples, p->nd);
        PUSH(t_samples, p->nt);
        PUSH(v_samples, (size_t)b->B * p->nv);
        PUSH(target_speed, b->B);
        PUSH(ego, (size_t)b->B * 6);
        PUSH(frame_of, b->B);
        PUSH(scene_of, b->B);
        PUSH(t_now, b->B);
    }
    if (!tables_resident) {
        PUSH(nx, b->F);
        PUSH(knots, (size_t)b->F * b->NX);
        PUSH(coef, (size_t)b->F * 8 * b->NX);
        PUSH(obs_pose, has_obs ? (size_t)b->S * b->T_obs * b->n_obs * 4 : 0);
        PUSH(obs_dims, has_obs ? (size_t)b->S * b->n_obs * 2 : 0);
        PUSH(final_time_step, has_obs ? b->S : 0);
        if (has_poly) {
            PUSH(obs_poly, (size_t)b->S * b->n_obs * 2 * b->poly_stride);
            PUSH(obs_nvert, (size_t)b->S * b->n_obs);
        }
    }
    if (b->skip) PUSH(skip, b->B);
#undef PUSH
    if (!has_obs) dev->n_obs = 0;
    if (!has_obs && dev_pub && inlined) dev_pub->n_obs = 0;
    return FP_OK;
}

// Launch order of a multi-round lattice launch (more egos than resident workgroups, one workgroup per ego).  Called right before
// the launch: hands out the permutation to dispatch in (nullptr = index order) and the array the workgroups leave their durations
// in.  Host work happens only when a fetched duration table has arrived (hipEventQuery, no wait): an argsort of B ints.
int launch_order_before(fp_ctx* ctx, OrderSet& set, int resident, const fp_batch* b, int nsplit, hipStream_t stream, const int** perm, int** dur,
                        const int* hint = nullptr, LaunchOrder** slot = nullptr)
{
    *perm = nullptr;
    *dur = nullptr;
    if (slot) *slot = nullptr;
    const bool lattice = &set == &ctx->order_lattice;
    if (lattice) ++ctx->lattice_launches;
    if (nsplit != 1 || b->B <= resident) return FP_OK;
    // Priority: the order learnt on THIS batch (exact history) > fp_batch.launch_order (the caller's input-only hint) > index order.
    // Durations are collected under a hint too, so a hinted batch moves on to its own order once it has one.
    if (!ctx->lattice_order) {
        if (hint) { *perm = hint; if (lattice) ++ctx->lattice_ordered_launches; }
        return FP_OK;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
    const bool capturing = cap != hipStreamCaptureStatusNone;
    LaunchOrder& o = set.of(b->ego);
    if (slot) *slot = &o;
    if (b->B > o.cap) {
        if (capturing) {  // no (re)allocation inside a capture: the hint, or index order
            if (hint) { *perm = hint; if (lattice) ++ctx->lattice_ordered_launches; }
            if (slot) *slot = nullptr;
            return FP_OK;
        }
        if (o.pending) { HIP_TRY(hipEventSynchronize(o.event)); o.pending = false; }
        HIP_TRY(hipStreamSynchronize(stream));
        const int cap_new = b->B + b->B / 4;
        FP_TRY(o.buf.reserve((size_t)cap_new * 2 * sizeof(int)));
        if (o.host) (void)hipHostFree(o.host);
        o.host = nullptr;
        HIP_TRY(hipHostMalloc((void**)&o.host, (size_t)cap_new * 2 * sizeof(int), hipHostMallocDefault));
        if (!o.event) HIP_TRY(hipEventCreateWithFlags(&o.event, hipEventDisableTiming));
        o.cap = cap_new;
        o.valid_B = o.dur_B = 0;
        o.since = 0;
    }
    int* d_dur = (int*)o.buf.base;
    int* d_perm = d_dur + o.cap;
    if (!capturing && o.pending && hipEventQuery(o.event) == hipSuccess) {
        o.pending = false;
        const int n = o.dur_B;
        if (n == b->B) {  // longest first; ties in index order (std::stable_sort keeps the result deterministic)
            const int* h_dur = o.host;
            int* h_perm = o.host + o.cap;
            for (int i = 0; i < n; ++i) h_perm[i] = i;
            std::stable_sort(h_perm, h_perm + n, [h_dur](int x, int y) { return h_dur[x] > h_dur[y]; });
            HIP_TRY(hipMemcpyAsync(d_perm, h_perm, (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream));
            if (o.valid_B == n) o.interval = o.interval * 2 < kOrderRefreshMax ? o.interval * 2 : kOrderRefreshMax;
            o.valid_B = n;
        }
    } else if (o.pending) {
        (void)hipGetLastError();  // hipErrorNotReady is not an error
    }
    if (o.valid_B == b->B) *perm = d_perm;
    else if (hint) *perm = hint;
    *dur = d_dur;
    if (lattice && *perm) ++ctx->lattice_ordered_launches;
    return FP_OK;
}

// Right after the launch: every kOrderRefresh launches enqueue the fetch of the durations this launch leaves behind.
int launch_order_after(LaunchOrder* op, const fp_batch* b, const int* dur, hipStream_t stream)
{
    if (!dur || !op) return FP_OK;
    LaunchOrder& o = *op;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
    if (cap != hipStreamCaptureStatusNone || o.pending) return FP_OK;
    const bool first = o.valid_B != b->B;  // no order for this batch size yet: fetch at once
    if (first) o.interval = kOrderRefresh;
    if (!first && ++o.since < o.interval) return FP_OK;
    o.since = 0;
    HIP_TRY(hipMemcpyAsync(o.host, dur, (size_t)b->B * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipEventRecord(o.event, stream));
    o.dur_B = b->B;
    o.pending = true;
    return FP_OK;
}

// Latency mode: a small batch cannot fill 256 CUs with one workgroup per ego, so the time-horizon slices of every ego are
// spread over nt workgroups.  Returns the split factor and makes sure the partial-argmin buffer exists.
int lattice_split_for(fp_ctx* ctx, const fp_params* p, const fp_batch* b, hipStream_t stream, int* nsplit, void** parts, int* group, int* tail)
{
    *nsplit = 1;
    *parts = nullptr;
    *group = 1;
    *tail = 0;
    // auto: as many workgroups per ego as keep ALL workgroups resident at once (measured on MI355X: 1.6-2.5x faster up to that
    // point, slower beyond it - a second round of workgroups costs more than the shorter critical path saves); at most one
    // workgroup per slice
    int parts_per_ego = p->nt;
    if (ctx->lattice_split == 0) {
        const long fit = b->B > 0 ? ctx->resident_groups / (long)b->B : 0;
        parts_per_ego = (int)(fit < p->nt ? fit : p->nt);
        // without obstacles the slices carry no work: one workgroup per ego
        if (parts_per_ego < p->nt && !(b->S > 0 && b->n_obs > 0)) parts_per_ego = 1;
    } else if (ctx->lattice_split == 1) {
        parts_per_ego = 1;
    }
    // Grouped slices (launch_lattice_fused, `group`): when the batch leaves room for no more than TWO workgroups per ego but every
    // ego can have a CU to itself, one 1024-thread workgroup per ego takes all its slices at once - four barrier intervals
    // instead of 4 nt, one prologue instead of two, no ticket and no merge.  Measured on MI355X (BASELINE configs[1], 5 x 5 x 5,
    // 10 obstacles): 256 egos 36.3 us against 37.9 split in two; 128 egos (split in four) 34.5 against 31.3 - so only in that
    // regime.  Needs the whole lattice's per-slice tables in one workgroup's LDS.
    const bool obstacles = b->S > 0 && b->n_obs > 0;
    if (ctx->lattice_group >= 2) {
        *group = ctx->lattice_group;
    } else if (ctx->lattice_group == 0 && ctx->lattice_split == 0 && obstacles && p->nt > 2 && parts_per_ego == 2 &&
               (long)b->B * 2 <= ctx->resident_groups && fp::lattice_group_fit(*p, *b) >= p->nt) {
        *group = p->nt;
        parts_per_ego = 1;
    }
    // Tail split (launch_lattice_fused, `tail`): a launch of several rounds of one-workgroup egos cuts its last slots in two
    const bool want_tail = parts_per_ego < 2 && *group == 1 && ctx->lattice_tail != 1 && obstacles && p->nt >= 2 &&
                           (ctx->lattice_tail >= 2 || b->B > ctx->resident_groups);
    if ((parts_per_ego < 2 && !want_tail) || p->nt < 2) return FP_OK;
    // want implies a small batch or an explicit request: the counters get a fixed region in front (kTicketBytes) so that they
    // never share bytes with the partial argmins of a call with another B
    if ((size_t)b->B * 4 > fp::kTicketBytes) return FP_OK;  // (an explicitly requested split of a huge batch: not worth it)
    const size_t need = fp::kTicketBytes + (size_t)b->B * p->nt * 16 + kAlign;
    if (need > ctx->parts.cap) {
        if (want_tail) {  // an optimisation only: never (re)allocate inside a stream capture for it
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
            if (cap != hipStreamCaptureStatusNone) return FP_OK;
        }
        HIP_TRY(hipStreamSynchronize(stream));  // the buffer is reallocated: drain its users
        FP_TRY(ctx->parts.reserve(need));
        // ticket counters start at zero; every launch leaves them at zero
        HIP_TRY(hipMemsetAsync(ctx->parts.base, 0, fp::kTicketBytes, stream));
    }
    *nsplit = parts_per_ego < 2 ? 1 : parts_per_ego;
    *parts = ctx->parts.base;
    if (want_tail) *tail = ctx->lattice_tail >= 2 ? ctx->lattice_tail : -(ctx->resident_groups / 2);  // (auto: the launcher knows the residency)
    return FP_OK;
}

// Device copy of best_idx for a dense call whose winner series come from winner_traj_kernel (KernelArgs::idx_shadow); nullptr
// when the buffer would have to grow inside a stream capture (the epilogue then reads the caller's array).
int32_t* idx_shadow_for(fp_ctx* ctx, size_t B, hipStream_t stream)
{
    if (B * sizeof(int32_t) > ctx->idx_shadow.cap) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap != hipStreamCaptureStatusNone) return nullptr;
        if (hipStreamSynchronize(stream) != hipSuccess || ctx->idx_shadow.reserve(B * sizeof(int32_t)) != FP_OK) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    return (int32_t*)ctx->idx_shadow.base;
}

// Hand-over flags for the epilogue workgroups a multi-round fused lattice launch appends to its grid (KernelArgs::epi_flag): zero when
// allocated, and every launch leaves them zero.  nullptr when the buffer would have to grow inside a stream capture (the series then
// come from winner_traj_kernel).
int32_t* epi_flags_for(fp_ctx* ctx, size_t B, hipStream_t stream)
{
    if (B * sizeof(int32_t) > ctx->epi_flags.cap) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap != hipStreamCaptureStatusNone) return nullptr;
        if (hipStreamSynchronize(stream) != hipSuccess || ctx->epi_flags.reserve(B * sizeof(int32_t)) != FP_OK ||
            hipMemsetAsync(ctx->epi_flags.base, 0, ctx->epi_flags.cap, stream) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    return (int32_t*)ctx->epi_flags.base;
}

// Who writes the winner's series of a dense call that does not write them inside the lattice workgroups: with "lattice_winner" = 0
// (auto) the epilogue workgroups appended to the lattice grid (launch_lattice_fused takes them when the launch is a three-per-CU one;
// otherwise it reports winner_done = false and winner_traj_kernel follows).
void offer_epilogue(fp_ctx* ctx, const fp::KernelArgs& ka, fp::KernelArgs* kl, size_t B, hipStream_t stream)
{
    if (ctx->lattice_winner != 0 || !ctx->appended_ok || !ka.r.best_traj || !ka.idx_shadow) return;
    int32_t* flags = epi_flags_for(ctx, B, stream);
    if (!flags) return;
    kl->r.best_traj = ka.r.best_traj;
    kl->epi_flag = flags;
}

// A hand-over of an earlier launch on this ctx timed out (fp_ctx::hand_err): the flags are reset in stream order, appended workgroups
// are not offered again on this ctx (winner_traj_kernel / the search kernel take over), and the caller learns that the outputs of that
// earlier call were incomplete.  FP_OK when nothing happened.
bool handover_failed(const fp_ctx* ctx) { return ctx->hand_err && *(volatile int32_t*)ctx->hand_err != 0; }
int handover_recover(fp_ctx* ctx, hipStream_t stream)
{
    const int code = *(volatile int32_t*)ctx->hand_err;
    ctx->appended_ok = false;
    // Inside a stream capture nothing may synchronise (and nothing runs): appended workgroups are off from here on, the word stays set,
    // and the first call outside a capture reports and clears it.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
    if (cap != hipStreamCaptureStatusNone) return FP_OK;
    // (the launch that timed out may still hold workgroups about to give up too: drain the stream before the word is cleared, or one of
    // them would set it again and a later, healthy call would report a failure of its own)
    (void)hipStreamSynchronize(stream);
    *(volatile int32_t*)ctx->hand_err = 0;
    if (ctx->epi_flags.base) HIP_TRY(hipMemsetAsync(ctx->epi_flags.base, 0, ctx->epi_flags.cap, stream));
    return fail(FP_EHIP, "a %s workgroup appended to an earlier lattice launch on this ctx waited 2 s for its ego's results and gave up: that call's "
                         "%s incomplete; the ctx now runs them in their own launches (call again)",
                code == 2 ? "FISS+ search" : "winner-series", code == 2 ? "FISS+ outputs are" : "series are");
}
int handover_check(fp_ctx* ctx, hipStream_t stream) { return handover_failed(ctx) ? handover_recover(ctx, stream) : FP_OK; }

// Optional curvature checks: the fused lattice kernel reads them from a [B][C] byte table that launch_lattice fills first.
int lattice_curv_scratch(fp_ctx* ctx, const fp_params* p, const fp_batch* b, hipStream_t stream, const uint8_t** out)
{
    *out = nullptr;
    if (!p->curvature_mask) return FP_OK;
    const size_t need = (size_t)b->B * p->nd * p->nv * p->nt + kAlign;
    if (need > ctx->curv_buf.cap) {
        HIP_TRY(hipStreamSynchronize(stream));  // the buffer is reallocated: drain its users
        FP_TRY(ctx->curv_buf.reserve(need));
    }
    *out = (const uint8_t*)ctx->curv_buf.base;
    return FP_OK;
}

// Where the winner's series is written: by the lattice kernel's own workgroup (no second launch: what a single plan() call wants)
// or by winner_traj_kernel right behind it.  In a multi-round launch (more egos than resident workgroups) the epilogue - one
// wavefront's dependent chain of ~4 us - holds the workgroup's slot while the next ego waits for it; as its own launch the same
// work runs on idle SIMDs next to nothing else.
bool winner_inside_lattice(const fp_ctx* ctx, const fp_batch* b)
{
    if (ctx->lattice_winner == 1) return true;
    if (ctx->lattice_winner == 2) return false;
    return b->B <= ctx->resident_groups;
}

fp_result no_result() { return fp_result{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0}; }

// FopPlusPlanner counts pops over the dense tables: when the caller did not ask for them they live in the ctx's scratch buffer.
int fopplus_tables(fp_ctx* ctx, size_t B, size_t C, fp_result* r, hipStream_t stream)
{
    const size_t need = align_up(sizeof(double) * B * C) + align_up(sizeof(uint32_t) * B * C);
    if (need > ctx->scratch.cap) {  // growing the buffer frees the old one: not inside a stream capture, and not under its users
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap != hipStreamCaptureStatusNone) return fail(FP_EINVAL, "result.fopplus: the ctx's table scratch must grow - run one call of this size outside the stream capture first");
        HIP_TRY(hipStreamSynchronize(stream));
    }
    FP_TRY(ctx->scratch.reserve(need));
    char* sp = (char*)ctx->scratch.base;
    if (!r->cost_tbl) r->cost_tbl = (double*)sp;
    if (!r->flag_tbl) r->flag_tbl = (uint32_t*)(sp + align_up(sizeof(double) * B * C));
    return FP_OK;
}

// The checks of check_batch_host on arrays that live in device memory (fp_ctx_set_option("validate", 1)): one lane per ego / frame /
// time sample, the first failing check wins.  err[0] = 0: clean.
__global__ void validate_batch_kernel(fp_params p, fp_batch b, int* err)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int code = 0;
    if (i < b.B) {
        const int f = b.frame_of[i], sc = b.scene_of[i];
        if (f < 0 || f >= b.F) code = 1;
        else if (sc >= b.S) code = 2;
        else if (b.t_now[i] < 0) code = 3;
    }
    if (!code && i < b.F && (b.nx[i] < 2 || b.nx[i] > b.NX)) code = 4;
    if (!code && i < p.nt) {
        const double n = b.t_samples[i] / p.tick_t;
        if (!(n > 0) || n > FP_MAX_POINTS) code = 5;
    }
    // the launch-order hint: a permutation of 0 .. B-1 (an entry out of range would be read as an ego; a repeated one plans an ego twice and
    // another not at all).  marks: [B] zeros behind the error words.
    if (!code && b.launch_order && i < b.B) {
        const int e = b.launch_order[i];
        if (e < 0 || e >= b.B) code = 11;
        else if (atomicAdd(&err[2 + e], 1) != 0) code = 12;
    }
    // polygon columns, one lane per column: a vertex count outside {0} u [3, poly_stride] would walk off the ring table; a ring that is
    // not counter-clockwise and convex, or that reaches outside the box of obs_dims, would be tested as a smaller shape than it is
    // (check_batch_host's checks, same tolerances)
    if (!code && b.obs_nvert && b.S > 0 && b.n_obs > 0) {
        const long cols = (long)b.S * b.n_obs;
        for (long c = i; c < cols && !code; c += (long)gridDim.x * blockDim.x) {
            const int n = b.obs_nvert[c];
            if (n == 0) continue;
            if (n < 3 || n > b.poly_stride) code = 6;
            else {
                const double* v = b.obs_poly + (size_t)c * 2 * b.poly_stride;
                double mx = 0.0, my = 0.0, area2 = 0.0;
                bool nan = false;
                for (int k = 0; k < n; ++k) {
                    const double* a = v + 2 * k;
                    const double* q = v + 2 * ((k + 1) % n);
                    nan = nan || !(a[0] == a[0]) || !(a[1] == a[1]);
                    mx = fmax(mx, fabs(a[0])); my = fmax(my, fabs(a[1]));
                    area2 += a[0] * q[1] - a[1] * q[0];
                }
                const double turn_tol = 64.0 * 2.220446049250313e-16 * (mx + my) * (mx + my);
                bool convex = true;
                for (int k = 0; k < n; ++k) {
                    const double* a = v + 2 * k;
                    const double* q = v + 2 * ((k + 1) % n);
                    const double* r = v + 2 * ((k + 2) % n);
                    convex = convex && ((q[0] - a[0]) * (r[1] - q[1]) - (q[1] - a[1]) * (r[0] - q[0]) >= -turn_tol);
                }
                if (nan) code = 7;
                else if (!(area2 > 0.0)) code = 8;
                else if (!convex) code = 9;
                else if (!(b.obs_dims[2 * c] >= 2.0 * mx) || !(b.obs_dims[2 * c + 1] >= 2.0 * my)) code = 10;
            }
            if (code) { if (atomicCAS(&err[0], 0, code) == 0) err[1] = (int)c; return; }
        }
    }
    if (code && atomicCAS(&err[0], 0, code) == 0) err[1] = i;
}

static size_t batch_marks(const fp_batch* b) { return b->launch_order && b->B > 0 ? (size_t)b->B : 0; }

int device_validate(fp_ctx* ctx, const fp_params* p, const fp_batch* b, hipStream_t stream)
{
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t marks = batch_marks(b);
    FP_TRY(ctx->validate_buf.reserve((2 + marks) * sizeof(int)));
    if (!ctx->validate_host) HIP_TRY(hipHostMalloc((void**)&ctx->validate_host, 2 * sizeof(int), hipHostMallocDefault));
    int* d_err = (int*)ctx->validate_buf.base;
    HIP_TRY(hipMemsetAsync(d_err, 0, (2 + marks) * sizeof(int), stream));
    int n = b->B > b->F ? b->B : b->F;
    if (p->nt > n) n = p->nt;
    if (b->obs_nvert && b->S > 0 && b->n_obs > 0) {  // (polygon columns: a lane per column, grid-stride beyond 64 K lanes)
        const long cols = (long)b->S * b->n_obs;
        const int want = (int)(cols < 65536 ? cols : 65536);
        if (want > n) n = want;
    }
    hipLaunchKernelGGL(validate_batch_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, *p, *b, d_err);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ctx->validate_host, d_err, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));  // the price of the option: the call waits for the stream
    const int code = ctx->validate_host[0], at = ctx->validate_host[1];
    switch (code) {
        case 0: return FP_OK;
        case 1: return fail(FP_EINVAL, "frame_of[%d] out of range (device batch, F=%d)", at, b->F);
        case 2: return fail(FP_EINVAL, "scene_of[%d] out of range (device batch, S=%d)", at, b->S);
        case 3: return fail(FP_EINVAL, "t_now[%d] is negative (device batch)", at);
        case 4: return fail(FP_EINVAL, "nx[%d] out of range (device batch, NX=%d)", at, b->NX);
        case 6: return fail(FP_EINVAL, "obs_nvert[%d] outside {0} and 3..poly_stride=%d (device batch)", at, b->poly_stride);
        case 7: return fail(FP_EINVAL, "obs_poly column %d has a NaN vertex (device batch)", at);
        case 8: return fail(FP_EINVAL, "obs_poly column %d is not counter-clockwise (device batch)", at);
        case 9: return fail(FP_EINVAL, "obs_poly column %d is not convex (device batch): cut it into convex pieces (obstacles.shape_columns)", at);
        case 10: return fail(FP_EINVAL, "obs_dims of polygon column %d does not contain its vertices (device batch)", at);
        case 11: return fail(FP_EINVAL, "launch_order[%d] is outside 0 .. B-1 (device batch, B=%d)", at, b->B);
        case 12: return fail(FP_EINVAL, "launch_order[%d] repeats an ego: not a permutation (device batch)", at);
        default: return fail(FP_ELIMIT, "t_samples[%d] needs more than FP_MAX_POINTS points (device batch)", at);
    }
}

// Orders `stream` after every overlapped dense call still in flight on the ctx's internal streams ("overlap").
int overlap_join(fp_ctx* ctx, hipStream_t stream)
{
    for (int k = 0; k < 2; ++k)
        if (ctx->ov_pending[k]) {
            HIP_TRY(hipStreamWaitEvent(stream, ctx->ov_done[k], 0));
            ctx->ov_pending[k] = false;
        }
    return FP_OK;
}

int common_checks(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, int mem, void* stream = nullptr)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    // any entry point but the overlapped dense path itself first joins what that path left in flight: its inputs may be those calls' outputs
    if (ctx->ov_pending[0] || ctx->ov_pending[1]) FP_TRY(overlap_join(ctx, mem == FP_MEM_DEVICE ? (hipStream_t)stream : ctx->stream));
    FP_TRY(check_params(params));
    FP_TRY(check_batch(batch));
    if (mem != FP_MEM_HOST && mem != FP_MEM_DEVICE) return fail(FP_EINVAL, "mem must be FP_MEM_HOST or FP_MEM_DEVICE");
    if (mem == FP_MEM_DEVICE && ctx->validate && batch->B > 0) FP_TRY(device_validate(ctx, params, batch, (hipStream_t)stream));
    return FP_OK;
}

// "overlap": the options the twin ctx must share with its owner (everything a dense call reads)
void overlap_sync_options(fp_ctx* twin, const fp_ctx* ctx)
{
    twin->lattice_kernel = ctx->lattice_kernel; twin->lattice_split = ctx->lattice_split; twin->lattice_occupancy = ctx->lattice_occupancy;
    twin->lattice_tail = ctx->lattice_tail; twin->lattice_group = ctx->lattice_group; twin->lattice_winner = ctx->lattice_winner;
    twin->lattice_order = ctx->lattice_order; twin->resident_groups = ctx->resident_groups; twin->lds_cu_kb = ctx->lds_cu_kb;
    twin->validate = ctx->validate;
    twin->refine_table_kb = ctx->refine_table_kb; twin->fiss_stages = ctx->fiss_stages; twin->fiss_jump = ctx->fiss_jump; twin->fiss_fused = ctx->fiss_fused;
    twin->handover_timeout_us = ctx->handover_timeout_us;
    if (!ctx->appended_ok) twin->appended_ok = false;
}

// The arrays a call writes (NULL entries write nothing), for the independence check of "overlap"
struct OverlapOuts {
    const void* p[12] = {};
    bool disjoint(const void* const (&o)[12]) const
    {
        for (const void* x : p)
            for (const void* y : o)
                if (x && x == y) return false;
        return true;
    }
};
OverlapOuts overlap_outs(const fp_result& a)
{
    OverlapOuts o;
    const void* v[] = {a.best_idx, a.best_cost, a.cost_tbl, a.flag_tbl, a.stats, a.best_flags, a.best_traj, a.fopplus, a.audit};
    for (int i = 0; i < 9; ++i) o.p[i] = v[i];
    return o;
}
OverlapOuts overlap_outs(const fp_fiss_io& a)
{
    OverlapOuts o;
    const void* v[] = {a.prev_best_idx, a.best_ijk, a.best_cost, a.end_state, a.refined, a.stats, a.trace, a.best_flags, a.best_traj};
    for (int i = 0; i < 9; ++i) o.p[i] = v[i];
    return o;
}

// "overlap": runs `impl(target ctx, its internal stream)` on the internal stream the previous overlapped call did NOT use.  The internal
// stream is ordered after everything the caller has enqueued on `stream` so far; `stream` is ordered after the call BEFORE this one (the
// deferred join: two calls in flight at most).  A call that writes an array its predecessor writes is not independent: it joins first
// and runs behind it.
template <class Impl>
int overlapped_call(fp_ctx* ctx, hipStream_t stream, const OverlapOuts& outs, Impl&& impl)
{
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->twin) {
        fp_ctx* tw = nullptr;
        FP_TRY(fp_ctx_create(ctx->device, &tw));
        tw->is_twin = true;
        ctx->twin = tw;
        for (hipEvent_t* e : {&ctx->ov_fork, &ctx->ov_done[0], &ctx->ov_done[1]}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    const int k = ctx->ov_next;
    fp_ctx* target = k ? ctx->twin : ctx;
    if (k) overlap_sync_options(ctx->twin, ctx);
    const bool independent = !ctx->ov_pending[k ^ 1] || outs.disjoint(ctx->ov_last_p);
    if (!independent) FP_TRY(overlap_join(ctx, stream));
    else if (ctx->ov_pending[k ^ 1]) ++ctx->overlapped_calls;
    HIP_TRY(hipEventRecord(ctx->ov_fork, stream));
    HIP_TRY(hipStreamWaitEvent(target->stream, ctx->ov_fork, 0));
    const bool keep[2] = {ctx->ov_pending[0], ctx->ov_pending[1]};
    ctx->ov_pending[0] = ctx->ov_pending[1] = false;  // (the implementation's common_checks must not join: that is this function's business)
    const int rc = impl(target, target->stream);
    ctx->ov_pending[0] = keep[0]; ctx->ov_pending[1] = keep[1];
    if (rc != FP_OK) {  // nothing (or not everything) was enqueued: leave the caller's stream ordered after whatever is in flight
        (void)overlap_join(ctx, stream);
        return rc;
    }
    HIP_TRY(hipEventRecord(ctx->ov_done[k], target->stream));
    if (ctx->ov_pending[k ^ 1]) {  // the deferred join of the predecessor
        HIP_TRY(hipStreamWaitEvent(stream, ctx->ov_done[k ^ 1], 0));
        ctx->ov_pending[k ^ 1] = false;
    }
    ctx->ov_pending[k] = true;
    memcpy(ctx->ov_last_p, outs.p, sizeof(outs.p));
    ctx->ov_next = k ^ 1;
    return FP_OK;
}

// Is this call one the "overlap" option applies to?  (FP_MEM_DEVICE, on the owner ctx, not inside a stream capture: nothing forks there.)
bool overlap_applies(fp_ctx* ctx, int mem, const fp_batch* batch, void* stream)
{
    if (!ctx->overlap || mem != FP_MEM_DEVICE || ctx->is_twin || !batch || batch->B <= 0) return false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) (void)hipGetLastError();
    return cap == hipStreamCaptureStatusNone;
}

}  // namespace

extern "C" {

int fp_abi_version(void) { return FP_ABI_VERSION; }

// The diagnostic macros this library was compiled with: "" for a production build.  The timing ablations (FP_ABL_*: results are
// WRONG by design), the instance switches (FP_NO_*, FP_SLICE_LOOP, ...) and the stamp / counter builds (FP_PHASE_STAMPS, FP_COUNTERS,
// FP_TL: series blocks carry clock ticks) all live in the production sources behind #if and build into the same file name under the same
// ABI version; this is how a caller tells them apart.  Two sources: the Makefile's EXTRA verbatim, and - however the macro got onto the
// command line - the list below.
const char* fp_build_flags(void)
{
    static const std::string flags = [] {
        std::string o;
#if defined(FP_BUILD_EXTRA)
        o = FP_BUILD_EXTRA;
#endif
        auto add = [&o](const char* n) {
            if (o.find(n) == std::string::npos) { if (!o.empty()) o += " "; o += "-D"; o += n; }
        };
        (void)add;
#define FP_FLAG_IF(m) add(#m);
#if defined(FP_PHASE_STAMPS)
        FP_FLAG_IF(FP_PHASE_STAMPS)
#endif
#if defined(FP_TL)
        FP_FLAG_IF(FP_TL)
#endif
#if defined(FP_RABL)
        FP_FLAG_IF(FP_RABL)
#endif
#if defined(FP_COUNTERS)
        FP_FLAG_IF(FP_COUNTERS)
#endif
#if defined(FP_ABL_MAT_NO_STORE)
        FP_FLAG_IF(FP_ABL_MAT_NO_STORE)
#endif
#if defined(FP_ABL_MAT_STORE_ONLY)
        FP_FLAG_IF(FP_ABL_MAT_STORE_ONLY)
#endif
#if defined(FP_ABL_MAT_NO_MATH)
        FP_FLAG_IF(FP_ABL_MAT_NO_MATH)
#endif
#if defined(FP_ABL_SEARCH_NOWALK)
        FP_FLAG_IF(FP_ABL_SEARCH_NOWALK)
#endif
#if defined(FP_ABL_SEARCH_NOSORT)
        FP_FLAG_IF(FP_ABL_SEARCH_NOSORT)
#endif
#if defined(FP_ABL_NO_N)
        FP_FLAG_IF(FP_ABL_NO_N)
#endif
#if defined(FP_ABL_NO_BN)
        FP_FLAG_IF(FP_ABL_NO_BN)
#endif
#if defined(FP_ABL_NO_GBN)
        FP_FLAG_IF(FP_ABL_NO_GBN)
#endif
#if defined(FP_ABL_NO_COLL)
        FP_FLAG_IF(FP_ABL_NO_COLL)
#endif
#if defined(FP_ABL_NO_WINNER)
        FP_FLAG_IF(FP_ABL_NO_WINNER)
#endif
#if defined(FP_ABL_NO_VMAX)
        FP_FLAG_IF(FP_ABL_NO_VMAX)
#endif
#if defined(FP_ABL_NO_SPLINE_COPY)
        FP_FLAG_IF(FP_ABL_NO_SPLINE_COPY)
#endif
#if defined(FP_ABL_NO_SLICE_SYNC)
        FP_FLAG_IF(FP_ABL_NO_SLICE_SYNC)
#endif
#if defined(FP_ABL_NO_SCAN)
        FP_FLAG_IF(FP_ABL_NO_SCAN)
#endif
#if defined(FP_ABL_NO_DALL)
        FP_FLAG_IF(FP_ABL_NO_DALL)
#endif
#if defined(FP_SLICE_LOOP)
        FP_FLAG_IF(FP_SLICE_LOOP)
#endif
#if defined(FP_NO_SHAPES)
        FP_FLAG_IF(FP_NO_SHAPES)
#endif
#if defined(FP_NO_OCC6)
        FP_FLAG_IF(FP_NO_OCC6)
#endif
#if defined(FP_NO_OCC8)
        FP_FLAG_IF(FP_NO_OCC8)
#endif
#if defined(FP_MAT_PER_CANDIDATE)
        FP_FLAG_IF(FP_MAT_PER_CANDIDATE)
#endif
#if defined(FP_MAT_NO_XCD)
        FP_FLAG_IF(FP_MAT_NO_XCD)
#endif
#if defined(FP_MAT_OCC)
        FP_FLAG_IF(FP_MAT_OCC)
#endif
#if defined(FP_WINNER_OCC)
        FP_FLAG_IF(FP_WINNER_OCC)
#endif
#if defined(FP_REFINE_OCC)
        FP_FLAG_IF(FP_REFINE_OCC)
#endif
#if defined(FP_SEARCH_WAVES)
        FP_FLAG_IF(FP_SEARCH_WAVES)
#endif
#if defined(FP_SEARCH_SMALL)
        FP_FLAG_IF(FP_SEARCH_SMALL)
#endif
#if defined(FP_SEARCH_NB)
        FP_FLAG_IF(FP_SEARCH_NB)
#endif
#if defined(FP_POSE_FLIGHT)
        FP_FLAG_IF(FP_POSE_FLIGHT)
#endif
#if defined(FP_GROUP_THREADS)
        FP_FLAG_IF(FP_GROUP_THREADS)
#endif
#if defined(FP_TEST_HOOKS)
        FP_FLAG_IF(FP_TEST_HOOKS)
#endif
#undef FP_FLAG_IF
        return o;
    }();
    return flags.c_str();
}

// The compiler this library was built with (the kernels depend on properties of ONE validated toolchain: no VGPR spill in any kernel -
// this compiler places spill stores before the exec restore of a divergent loop's exit -, -disable-machine-licm, the SGPR budgets).
const char* fp_build_compiler(void) { return __clang_version__; }

const char* fp_last_error(void) { return g_last_error.c_str(); }

int fp_device_count(int* count)
{
    if (!count) return fail(FP_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(FP_ENODEV, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *count = n;
    return FP_OK;
}

int fp_device_info(int device, char* buf, int buflen, int* compute_units, int64_t* hbm_bytes)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (buf && buflen > 0) snprintf(buf, (size_t)buflen, "%s (%s)", prop.name, prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return FP_OK;
}

int fp_ctx_create(int device, fp_ctx** out)
{
    if (!out) return fail(FP_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(FP_ENODEV, "no HIP device visible: the engine has no CPU fallback");
    if (device < 0 || device >= n) return fail(FP_EINVAL, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    fp_ctx* ctx = new (std::nothrow) fp_ctx();
    if (!ctx) return fail(FP_ENOMEM, "out of host memory");
    ctx->device = device;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->pinned, kSmallRegion, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->hand_err, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) memset(ctx->hand_err, 0, 64);
    if (e != hipSuccess) {
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        if (ctx->hand_err) (void)hipHostFree(ctx->hand_err);
        delete ctx;
        return fail(FP_EHIP, "ctx resources: %s", hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        if (prop.multiProcessorCount > 0) ctx->resident_groups = 2 * prop.multiProcessorCount;
        if (prop.maxSharedMemoryPerMultiProcessor > 0) ctx->lds_cu_kb = (int)(prop.maxSharedMemoryPerMultiProcessor / 1024);
        // (in-order workgroup dispatch per XCD was verified on these two; anything else gets the series / the search in their own launches)
        ctx->appended_ok = strncmp(prop.gcnArchName, "gfx950", 6) == 0 || strncmp(prop.gcnArchName, "gfx942", 6) == 0;
    }
    *out = ctx;
    return FP_OK;
}

int fp_ctx_destroy(fp_ctx* ctx)
{
    if (!ctx) return FP_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->twin) {  // ("overlap": drain both internal streams before anything they use is freed)
        (void)hipStreamSynchronize(ctx->twin->stream);
        (void)hipStreamSynchronize(ctx->stream);
        (void)fp_ctx_destroy(ctx->twin);
        for (hipEvent_t e : {ctx->ov_fork, ctx->ov_done[0], ctx->ov_done[1]})
            if (e) (void)hipEventDestroy(e);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->arena.base) (void)hipFree(ctx->arena.base);
    if (ctx->scratch.base) (void)hipFree(ctx->scratch.base);
    if (ctx->parts.base) (void)hipFree(ctx->parts.base);
    for (auto& t : ctx->tables)
        if (t.buf.base) (void)hipFree(t.buf.base);
    if (ctx->curv_buf.base) (void)hipFree(ctx->curv_buf.base);
    if (ctx->idx_shadow.base) (void)hipFree(ctx->idx_shadow.base);
    if (ctx->epi_flags.base) (void)hipFree(ctx->epi_flags.base);
    ctx->order_lattice.release();
    ctx->order_refine.release();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->hand_err) (void)hipHostFree(ctx->hand_err);
    if (ctx->validate_host) (void)hipHostFree(ctx->validate_host);
    delete ctx;
    return FP_OK;
}

int fp_ctx_set_option(fp_ctx* ctx, const char* name, int value)
{
    if (!ctx || !name) return fail(FP_EINVAL, "ctx/name is NULL");
    if (strcmp(name, "lattice_kernel") == 0) {
        if (value < 0 || value > 2) return fail(FP_EINVAL, "lattice_kernel must be 0 (auto), 1 (per-candidate) or 2 (fused)");
        ctx->lattice_kernel = value;
        return FP_OK;
    }
    if (strcmp(name, "lattice_order") == 0) {
        if (value < 0 || value > 1) return fail(FP_EINVAL, "lattice_order must be 0 or 1");
        ctx->lattice_order = value;
        ctx->order_lattice.forget(); ctx->order_refine.forget();
        return FP_OK;
    }
    if (strcmp(name, "refine_table_kb") == 0) {
        if (value < 0 || value > 96) return fail(FP_EINVAL, "refine_table_kb must be in [0, 96]");
        ctx->refine_table_kb = value;
        return FP_OK;
    }
    if (strcmp(name, "lattice_winner") == 0) {
        if (value < 0 || value > 2) return fail(FP_EINVAL, "lattice_winner must be 0 (auto), 1 (inside the lattice kernel) or 2 (own launch)");
        ctx->lattice_winner = value;
        return FP_OK;
    }
    if (strcmp(name, "fiss_fused") == 0) {
        if (value < 0 || value > 1) return fail(FP_EINVAL, "fiss_fused must be 0 or 1");
        ctx->fiss_fused = value;
        return FP_OK;
    }
    if (strcmp(name, "appended_workgroups") == 0) {  // 0: never offer appended workgroups on this ctx (what a timed-out hand-over sets)
        if (value < 0 || value > 1) return fail(FP_EINVAL, "appended_workgroups must be 0 or 1");
        ctx->appended_ok = value != 0;
        return FP_OK;
    }
    if (strcmp(name, "handover_timeout_us") == 0) {  // test hook: appended workgroups give up after this many microseconds (0 = 2 s)
        if (!getenv("FP_TEST_HOOKS")) return fail(FP_EINVAL, "handover_timeout_us is a test hook: set FP_TEST_HOOKS=1 in the environment to use it");
        if (value < 0 || value > 10000000) return fail(FP_EINVAL, "handover_timeout_us must be in 0..10^7");
        ctx->handover_timeout_us = value;
        return FP_OK;
    }
    if (strcmp(name, "handover_inject") == 0) {  // test hook: what a timed-out hand-over leaves in the ctx's error word (1 series, 2 search)
        if (!getenv("FP_TEST_HOOKS")) return fail(FP_EINVAL, "handover_inject is a test hook: set FP_TEST_HOOKS=1 in the environment to use it");
        if (value < 1 || value > 2 || !ctx->hand_err) return fail(FP_EINVAL, "handover_inject must be 1 or 2");
        *(volatile int32_t*)ctx->hand_err = value;
        return FP_OK;
    }
    if (strcmp(name, "fiss_stages") == 0) {
        if (value < 1 || value > 3) return fail(FP_EINVAL, "fiss_stages must be 1 (lattice only), 2 (+ search) or 3 (all)");
        ctx->fiss_stages = value;
        return FP_OK;
    }
    if (strcmp(name, "validate") == 0) {
        if (value < 0 || value > 1) return fail(FP_EINVAL, "validate must be 0 or 1");
        ctx->validate = value;
        return FP_OK;
    }
    if (strcmp(name, "fiss_jump") == 0) {
        if (value < 0 || value > 1) return fail(FP_EINVAL, "fiss_jump must be 0 or 1");
        ctx->fiss_jump = value;
        return FP_OK;
    }
    if (strcmp(name, "inline_inputs") == 0) {
        if (value < 0 || value > 1) return fail(FP_EINVAL, "inline_inputs must be 0 or 1");
        ctx->inline_inputs = value;
        return FP_OK;
    }
    if (strcmp(name, "stage_kernel") == 0) {
        if (value < 0 || value > 1) return fail(FP_EINVAL, "stage_kernel must be 0 or 1");
        ctx->stage_kernel = value;
        return FP_OK;
    }
    if (strcmp(name, "zero_copy_in") == 0) {
        if (value < 0 || value > 2) return fail(FP_EINVAL, "zero_copy_in must be 0, 1 or 2");
        ctx->zero_copy_in = value;
        return FP_OK;
    }
    if (strcmp(name, "lattice_group") == 0) {
        if (value < 0) return fail(FP_EINVAL, "lattice_group must be 0 (auto), 1 (never) or the number of slices per group");
        ctx->lattice_group = value;
        return FP_OK;
    }
    if (strcmp(name, "lattice_occupancy") == 0) {
        if (value != 0 && (value < 2 || value > 4)) return fail(FP_EINVAL, "lattice_occupancy must be 0 (auto), 2, 3 or 4 (workgroups per compute unit at most)");
        ctx->lattice_occupancy = value;
        return FP_OK;
    }
    if (strcmp(name, "overlap") == 0) {
        if (value < 0 || value > 1) return fail(FP_EINVAL, "overlap must be 0 or 1");
        if (ctx->is_twin) return fail(FP_EINVAL, "overlap: not on a twin ctx");
        ctx->overlap = value;  // (calls in flight stay in flight: fp_ctx_join / the next entry point joins them)
        return FP_OK;
    }
    if (strcmp(name, "resident_groups") == 0) {
        // lattice workgroups the device holds at once at two per CU (default: 2 x the device's compute units).  Lower it when the process
        // runs under a CU mask (HSA_CU_MASK / ROC_GLOBAL_CU_MASK: the runtime still reports every CU), or to model a smaller device: the
        // latency-mode split, the tail split, the three- / four-per-CU instances and the launch order all key on it.  0 = the device's value.
        if (value < 0 || (value & 1)) return fail(FP_EINVAL, "resident_groups must be 0 (the device's 2 x compute units) or an even number >= 2");
        if (value == 0) {
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
            value = 2 * prop.multiProcessorCount;
        }
        ctx->resident_groups = value;
        return FP_OK;
    }
    if (strcmp(name, "lattice_tail") == 0) {
        if (value < 0) return fail(FP_EINVAL, "lattice_tail must be 0 (auto), 1 (never) or the number of egos cut in two");
        ctx->lattice_tail = value;
        return FP_OK;
    }
    if (strcmp(name, "lattice_split") == 0) {
        if (value < 0 || value > 2) return fail(FP_EINVAL, "lattice_split must be 0 (auto), 1 (never) or 2 (always)");
        ctx->lattice_split = value;
        return FP_OK;
    }
    return fail(FP_EINVAL, "unknown option '%s'", name);
}

int fp_ctx_get_option(fp_ctx* ctx, const char* name, int* value)
{
    if (!ctx || !name || !value) return fail(FP_EINVAL, "ctx/name/value is NULL");
    const struct { const char* n; int v; } tab[] = {
        {"lattice_kernel", ctx->lattice_kernel}, {"lattice_split", ctx->lattice_split}, {"lattice_group", ctx->lattice_group}, {"lattice_tail", ctx->lattice_tail}, {"lattice_occupancy", ctx->lattice_occupancy}, {"resident_groups", ctx->resident_groups}, {"zero_copy_in", ctx->zero_copy_in}, {"stage_kernel", ctx->stage_kernel}, {"inline_inputs", ctx->inline_inputs}, {"lattice_order", ctx->lattice_order},
        {"refine_table_kb", ctx->refine_table_kb}, {"fiss_stages", ctx->fiss_stages}, {"fiss_jump", ctx->fiss_jump}, {"validate", ctx->validate}, {"lattice_winner", ctx->lattice_winner}, {"fiss_fused", ctx->fiss_fused}, {"appended_workgroups", ctx->appended_ok ? 1 : 0}, {"handover_failed", ctx->hand_err ? *(volatile int32_t*)ctx->hand_err : 0}, {"overlap", ctx->overlap}, {"overlapped_calls", ctx->overlapped_calls}, {"lattice_launches", ctx->lattice_launches + (ctx->twin ? ctx->twin->lattice_launches : 0)},
        {"lattice_ordered_launches", ctx->lattice_ordered_launches + (ctx->twin ? ctx->twin->lattice_ordered_launches : 0)},
        {"lattice_launches_2", (int)fp::lattice_launches_per_cu(0)}, {"lattice_launches_3", (int)fp::lattice_launches_per_cu(1)}, {"lattice_launches_4", (int)fp::lattice_launches_per_cu(2)}};
    for (const auto& t : tab)
        if (strcmp(name, t.n) == 0) { *value = t.v; return FP_OK; }
    return fail(FP_EINVAL, "unknown option '%s'", name);
}

static int plan_dense_impl(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_result* result, int mem, void* stream);

int fp_ctx_join(fp_ctx* ctx, void* stream)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    return overlap_join(ctx, (hipStream_t)stream);
}

int fp_plan_dense(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_result* result, int mem, void* stream)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    if (!result || !overlap_applies(ctx, mem, batch, stream)) return plan_dense_impl(ctx, params, batch, result, mem, stream);
    return overlapped_call(ctx, (hipStream_t)stream, overlap_outs(*result),
                           [&](fp_ctx* target, hipStream_t is) { return plan_dense_impl(target, params, batch, result, FP_MEM_DEVICE, is); });
}

static int plan_dense_impl(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_result* result, int mem, void* stream)
{
    FP_TRY(common_checks(ctx, params, batch, mem, stream));
    if (!result || !result->best_idx || !result->best_cost) return fail(FP_EINVAL, "result.best_idx/best_cost must not be NULL");
    if (result->best_traj && !result->best_flags) return fail(FP_EINVAL, "result.best_traj requires result.best_flags");
    if (result->audit && result->fopplus) return fail(FP_EINVAL, "result.audit settles FrenetOptimalPlanner's argmin: not together with result.fopplus");
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    FP_TRY(handover_check(ctx, mem == FP_MEM_DEVICE ? (hipStream_t)stream : ctx->stream));
    const size_t C = (size_t)params->nd * params->nv * params->nt, B = (size_t)batch->B;
    int stride;
    FP_TRY(traj_stride_of(result->traj_stride, &stride));
    fp::KernelArgs ka;
    ka.p = *params;
    ka.err_word = ctx->hand_err;
    ka.occ_cap = ctx->lattice_occupancy; ka.resident2 = ctx->resident_groups; ka.lds_cu_kb = ctx->lds_cu_kb; ka.handover_timeout_us = ctx->handover_timeout_us;
    if (mem == FP_MEM_DEVICE) {
        ka.b = *batch;
        if (!(batch->S > 0 && batch->n_obs > 0)) ka.b.n_obs = 0;
        ka.r = *result;
        if ((result->fopplus || result->audit) && (!ka.r.cost_tbl || !ka.r.flag_tbl)) FP_TRY(fopplus_tables(ctx, B, C, &ka.r, (hipStream_t)stream));
        FP_TRY(lattice_curv_scratch(ctx, params, batch, (hipStream_t)stream, &ka.curv_tbl));
        int nsplit, group, tail; void* parts;
        FP_TRY(lattice_split_for(ctx, params, batch, (hipStream_t)stream, &nsplit, &parts, &group, &tail));
        bool winner_done = false;
        const int* perm; int* dur;
        LaunchOrder* oslot = nullptr;
        FP_TRY(launch_order_before(ctx, ctx->order_lattice, ctx->resident_groups, batch, nsplit, (hipStream_t)stream, &perm, &dur, batch->launch_order, &oslot));
        // (the audit pass may move the winner: the series are written after it, by their own launch)
        const bool inside = winner_inside_lattice(ctx, batch) && !result->audit && !big_points(ka.p);
        if (result->best_traj && !inside) ka.idx_shadow = idx_shadow_for(ctx, B, (hipStream_t)stream);
        fp::KernelArgs kl = ka;
        if (!inside) kl.r.best_traj = nullptr;
        if (!inside && !result->audit && !big_points(ka.p)) offer_epilogue(ctx, ka, &kl, B, (hipStream_t)stream);
        LAUNCH_TRY(fp::launch_lattice(kl, (hipStream_t)stream, ctx->lattice_kernel, parts, nsplit, &winner_done, perm, dur, group, nullptr, tail), "lattice kernel");
        FP_TRY(launch_order_after(oslot, batch, dur, (hipStream_t)stream));
        if (result->audit) LAUNCH_TRY(fp::launch_audit(ka, result->audit, (hipStream_t)stream), "audit kernel");
        if (result->best_traj && !winner_done) LAUNCH_TRY(fp::launch_winner_traj(ka, nullptr, (hipStream_t)stream), "winner epilogue");
        if (result->fopplus)
            LAUNCH_TRY(fp::launch_fopplus_count((int)B, (int)C, ka.r.cost_tbl, ka.r.flag_tbl, ka.r.best_idx, ka.r.best_cost, result->fopplus, ka.r.stats,
                                                ka.b.skip, (hipStream_t)stream), "FOP+ count kernel");
        return FP_OK;
    }
    FP_TRY(check_batch_host(params, batch));
    if (result->best_traj) FP_TRY(check_stride_host(params, batch, stride));
    ka.p.points_max = host_points_max(params, batch, nullptr, 0, 0);
    const bool big = big_points(ka.p);
    const size_t traj_doubles = result->best_traj ? B * FP_ARR_COUNT * (size_t)stride : 0;
    HostStage hs(ctx);
    FP_TRY(hs.reserve(batch_need(params, batch) + HostStage::need<int32_t>(B * 4) + 2 * HostStage::need<double>(B) + HostStage::need<int32_t>(B) +
                      HostStage::need<double>(B * C) + HostStage::need<uint32_t>(B * C) + HostStage::need<uint32_t>(B) +
                      HostStage::need<double>(traj_doubles) + HostStage::need<int32_t>(B * 2) + HostStage::need<uint32_t>(B),
                      /*zero_copy_out=*/B <= 8));
    // (inline inputs need the fused kernel with the winner's series inside it: no other kernel of this call may read the batch)
    fp::InlineIn inl;
    const bool try_inline = ctx->inline_inputs && B <= 8 && !params->curvature_mask && ctx->lattice_kernel != 1 && !big &&
                            (!result->best_traj || winner_inside_lattice(ctx, batch)) && !result->audit && fp::lattice_group_fit(*params, *batch) >= 1;
    FP_TRY(stage_batch(hs, params, batch, &ka.b, try_inline ? &inl : nullptr));
    FP_TRY(hs.flush_in());
    ka.r.best_idx = hs.out(result->best_idx, B);
    ka.r.best_cost = hs.out(result->best_cost, B);
    ka.r.stats = hs.out(result->stats, B * 4);
    ka.r.cost_tbl = hs.out(result->cost_tbl, B * C);
    ka.r.flag_tbl = hs.out(result->flag_tbl, B * C);
    ka.r.best_flags = hs.out(result->best_flags, B);
    ka.r.best_traj = hs.out(result->best_traj, traj_doubles);
    ka.r.traj_stride = result->traj_stride;
    ka.r.traj_sparse = result->traj_sparse;
    int32_t* d_fopplus = hs.out(result->fopplus, B * 2);
    uint32_t* d_audit = hs.out(result->audit, B);
    if ((result->fopplus || result->audit) && (!ka.r.cost_tbl || !ka.r.flag_tbl)) FP_TRY(fopplus_tables(ctx, B, C, &ka.r, ctx->stream));
    // sparse rows are only partly written by the kernels: the host block comes back with the caller's own bytes elsewhere
    if (result->traj_sparse && ka.r.best_traj) HIP_TRY(hipMemcpyAsync(ka.r.best_traj, result->best_traj, traj_doubles * sizeof(double), hipMemcpyDefault, ctx->stream));
    FP_TRY(lattice_curv_scratch(ctx, params, batch, ctx->stream, &ka.curv_tbl));
    int nsplit, group, tail; void* parts;
    FP_TRY(lattice_split_for(ctx, params, batch, ctx->stream, &nsplit, &parts, &group, &tail));
    bool winner_done = false;
    const int* perm; int* dur;
    LaunchOrder* oslot = nullptr;
    FP_TRY(launch_order_before(ctx, ctx->order_lattice, ctx->resident_groups, batch, nsplit, ctx->stream, &perm, &dur, nullptr, &oslot));
    fp::KernelArgs kl = ka;
    if (!winner_inside_lattice(ctx, batch) || result->audit || big) {
        kl.r.best_traj = nullptr;
        if (result->best_traj && !result->audit && !inl.on) {
            ka.idx_shadow = kl.idx_shadow = idx_shadow_for(ctx, B, ctx->stream);
            if (!big) offer_epilogue(ctx, ka, &kl, B, ctx->stream);
        }
    }
    LAUNCH_TRY(fp::launch_lattice(kl, ctx->stream, ctx->lattice_kernel, parts, nsplit, &winner_done, perm, dur, group, inl.on ? &inl : nullptr, tail), "lattice kernel");
    FP_TRY(launch_order_after(oslot, batch, dur, ctx->stream));
    if (d_audit) LAUNCH_TRY(fp::launch_audit(ka, d_audit, ctx->stream), "audit kernel");
    if (result->best_traj && !winner_done) {
        if (inl.on) return fail(FP_EHIP, "internal: inline inputs without the series inside the lattice kernel");
        LAUNCH_TRY(fp::launch_winner_traj(ka, nullptr, ctx->stream), "winner epilogue");
    }
    if (d_fopplus)
        LAUNCH_TRY(fp::launch_fopplus_count((int)B, (int)C, ka.r.cost_tbl, ka.r.flag_tbl, ka.r.best_idx, ka.r.best_cost, d_fopplus, ka.r.stats, ka.b.skip, ctx->stream),
                   "FOP+ count kernel");
    FP_TRY(hs.fetch_out());
    if (handover_failed(ctx)) {  // (the call has synchronised: a timed-out hand-over of ITS launch is known now - run it again without appended workgroups)
        (void)handover_recover(ctx, ctx->stream);
        return fp_plan_dense(ctx, params, batch, result, mem, stream);
    }
    return FP_OK;
}

int fp_winner_trajs(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const int32_t* best_idx, uint32_t* best_flags,
                    double* best_traj, int32_t traj_stride, int32_t traj_sparse, int mem, void* stream)
{
    FP_TRY(common_checks(ctx, params, batch, mem, stream));
    if (!best_idx || !best_flags || !best_traj) return fail(FP_EINVAL, "best_idx/best_flags/best_traj must not be NULL");
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    int stride;
    FP_TRY(traj_stride_of(traj_stride, &stride));
    const size_t B = (size_t)batch->B, traj_doubles = B * FP_ARR_COUNT * (size_t)stride;
    fp::KernelArgs ka;
    ka.p = *params;
    ka.r = no_result();
    ka.r.traj_stride = traj_stride;
    ka.r.traj_sparse = traj_sparse;
    if (mem == FP_MEM_DEVICE) {
        ka.b = *batch;
        ka.r.best_idx = const_cast<int32_t*>(best_idx);
        ka.r.best_flags = best_flags;
        ka.r.best_traj = best_traj;
        LAUNCH_TRY(fp::launch_winner_traj(ka, nullptr, (hipStream_t)stream), "winner epilogue");
        return FP_OK;
    }
    FP_TRY(check_batch_host(params, batch));
    FP_TRY(check_stride_host(params, batch, stride));
    ka.p.points_max = host_points_max(params, batch, nullptr, 0, 0);
    const int C = params->nd * params->nv * params->nt;
    for (size_t i = 0; i < B; ++i)
        if (best_idx[i] >= C) return fail(FP_EINVAL, "best_idx[%zu]=%d out of range", i, best_idx[i]);
    HostStage hs(ctx);
    FP_TRY(hs.reserve(batch_need(params, batch) + HostStage::need<int32_t>(B) + HostStage::need<uint32_t>(B) + HostStage::need<double>(traj_doubles)));
    FP_TRY(stage_batch(hs, params, batch, &ka.b));
    const int32_t* d_idx = nullptr;
    FP_TRY(hs.in(best_idx, B, &d_idx));
    FP_TRY(hs.flush_in());
    ka.r.best_idx = const_cast<int32_t*>(d_idx);
    ka.r.best_flags = hs.out(best_flags, B);
    ka.r.best_traj = hs.out(best_traj, traj_doubles);
    if (traj_sparse) HIP_TRY(hipMemcpyAsync(ka.r.best_traj, best_traj, traj_doubles * sizeof(double), hipMemcpyDefault, ctx->stream));
    LAUNCH_TRY(fp::launch_winner_traj(ka, nullptr, ctx->stream), "winner epilogue");
    return hs.fetch_out();
}

int fp_materialize_all(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, uint32_t* flags, double* traj, int32_t traj_stride,
                       int32_t traj_sparse, int mem, void* stream)
{
    FP_TRY(common_checks(ctx, params, batch, mem, stream));
    if (!flags || !traj) return fail(FP_EINVAL, "flags/traj must not be NULL");
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    int stride;
    FP_TRY(traj_stride_of(traj_stride, &stride));
    const size_t BC = (size_t)batch->B * params->nd * params->nv * params->nt, traj_doubles = BC * FP_ARR_COUNT * (size_t)stride;
    if (BC > 0x7fffffffu) return fail(FP_ELIMIT, "B*C = %zu exceeds the grid size limit", BC);
    fp::KernelArgs ka;
    ka.p = *params;
    ka.r = no_result();
    ka.r.traj_stride = traj_stride;
    ka.r.traj_sparse = traj_sparse;
    if (mem == FP_MEM_DEVICE) {
        ka.b = *batch;
        ka.r.best_flags = flags;
        ka.r.best_traj = traj;
        LAUNCH_TRY(fp::launch_materialize_all(ka, (hipStream_t)stream), "materialise kernel");
        return FP_OK;
    }
    FP_TRY(check_batch_host(params, batch));
    FP_TRY(check_stride_host(params, batch, stride));
    ka.p.points_max = host_points_max(params, batch, nullptr, 0, 0);
    HostStage hs(ctx);
    FP_TRY(hs.reserve(batch_need(params, batch) + HostStage::need<uint32_t>(BC) + HostStage::need<double>(traj_doubles)));
    FP_TRY(stage_batch(hs, params, batch, &ka.b));
    FP_TRY(hs.flush_in());
    ka.r.best_flags = hs.out(flags, BC);
    ka.r.best_traj = hs.out(traj, traj_doubles);
    if (traj_sparse) HIP_TRY(hipMemcpyAsync(ka.r.best_traj, traj, traj_doubles * sizeof(double), hipMemcpyDefault, ctx->stream));
    LAUNCH_TRY(fp::launch_materialize_all(ka, ctx->stream), "materialise kernel");
    return hs.fetch_out();
}

}  // extern "C"

namespace {

// fp_plan_fiss, and - with `loop` (FP_MEM_DEVICE only) - fp_plan_fiss_step: the pipeline plus the egos' hand-over to their next states.
int plan_fiss_impl(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_fiss_opts* opts, const fp_fiss_io* io, const fp_loop_io* loop,
                   int mem, void* stream_v)
{
    FP_TRY(common_checks(ctx, params, batch, mem, stream_v));
    if (!opts || !io) return fail(FP_EINVAL, "opts/io is NULL");
    if (opts->kind != FP_FISS && opts->kind != FP_FISS_PLUS) return fail(FP_EINVAL, "opts.kind must be FP_FISS or FP_FISS_PLUS");
    if (opts->max_refine_iters < 0 || opts->max_refine_iters * 7 > 64) return fail(FP_ELIMIT, "max_refine_iters must be in 0..9");
    if ((long)params->nd * params->nv * params->nt > FP_MAX_CAND_SEARCH)
        return fail(FP_ELIMIT, "nd*nv*nt = %ld exceeds FP_MAX_CAND_SEARCH (the device-side search walk): use fp_plan_dense's tables and a host walk", (long)params->nd * params->nv * params->nt);
    if (!io->samp_min || !io->samp_max || !io->samp_res || !io->prev_best_idx || !io->best_ijk || !io->best_cost || !io->end_state ||
        !io->refined || !io->stats)
        return fail(FP_EINVAL, "fp_fiss_io has a NULL mandatory array");
    if (io->best_traj && !io->best_flags) return fail(FP_EINVAL, "io.best_traj requires io.best_flags");
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t B = (size_t)batch->B, C = (size_t)params->nd * params->nv * params->nt;
    const int R = opts->kind == FP_FISS_PLUS ? opts->max_refine_iters : 0;
    hipStream_t stream = mem == FP_MEM_DEVICE ? (hipStream_t)stream_v : ctx->stream;
    FP_TRY(handover_check(ctx, stream));
    // (host entry: prev_best_idx is in / out - should the appended search of this call time out, the call is repeated from the caller's values)
    std::vector<int32_t> prev_in;
    if (mem == FP_MEM_HOST && ctx->appended_ok && ctx->fiss_fused && B > (size_t)ctx->resident_groups) prev_in.assign(io->prev_best_idx, io->prev_best_idx + B * 3);

    // dense tables + per-ego dense results live in the scratch buffer in both modes
    const size_t scratch_need = align_up(sizeof(double) * B * C) + align_up(sizeof(uint32_t) * B * C) + align_up(sizeof(int32_t) * B) +
                                align_up(sizeof(double) * B) + 4 * kAlign;
    if (scratch_need > ctx->scratch.cap) HIP_TRY(hipStreamSynchronize(stream));  // the buffer may be reallocated: drain its users
    FP_TRY(ctx->scratch.reserve(scratch_need));
    char* sp = ctx->scratch.base;
    fp::FissArgs fa;
    fa.ka.p = *params;
    fa.ka.err_word = ctx->hand_err;
    fa.ka.occ_cap = ctx->lattice_occupancy; fa.ka.resident2 = ctx->resident_groups; fa.ka.lds_cu_kb = ctx->lds_cu_kb; fa.ka.handover_timeout_us = ctx->handover_timeout_us;
    fa.opts = *opts;
    fa.opts.max_refine_iters = R;
    fa.ka.r = no_result();
    fa.ka.r.cost_tbl = (double*)sp;
    sp += align_up(sizeof(double) * B * C);
    fa.ka.r.flag_tbl = (uint32_t*)sp;
    sp += align_up(sizeof(uint32_t) * B * C);
    fa.ka.r.best_idx = (int32_t*)sp;
    sp += align_up(sizeof(int32_t) * B);
    fa.ka.r.best_cost = (double*)sp;
    fa.cost_tbl = fa.ka.r.cost_tbl;
    fa.flag_tbl = fa.ka.r.flag_tbl;
    int stride;
    FP_TRY(traj_stride_of(io->traj_stride, &stride));
    const size_t traj_doubles = io->best_traj ? B * FP_ARR_COUNT * (size_t)stride : 0;
    const size_t trace_doubles = (io->trace && R > 0) ? B * (size_t)R * 7 * 4 : 0;
    HostStage hs(ctx);
    fp::InlineIn inl;   // (host entry, latency regime: see below)
    fp_batch lat_b;     // the batch as the lattice kernel addresses it (byte offsets into inl.bytes when inl.on)
    if (mem == FP_MEM_DEVICE) {
        fa.ka.b = *batch;
        if (loop) fa.ka.b.skip = loop->done;
        if (!(batch->S > 0 && batch->n_obs > 0)) fa.ka.b.n_obs = 0;
        fa.io = *io;
        if (!trace_doubles) fa.io.trace = nullptr;
    } else {
        FP_TRY(check_batch_host(params, batch));
        fa.ka.p.points_max = host_points_max(params, batch, R > 0 ? io->samp_max + 2 : nullptr, R > 0 ? B : 0, 3);  // (refined trajectories reach T = samp_max)
        if (io->best_traj) {
            for (size_t i = 0; i < B; ++i)  // refined trajectories reach T = samp_max of their ego
                if (ceil(io->samp_max[3 * i + 2] / params->tick_t) > stride) return fail(FP_EINVAL, "traj_stride=%d is smaller than the points of samp_max[%zu].T=%g", stride, i, io->samp_max[3 * i + 2]);
            FP_TRY(check_stride_host(params, batch, stride));
        }
        FP_TRY(hs.reserve(batch_need(params, batch) + 4 * HostStage::need<double>(B * 3) + 2 * HostStage::need<int32_t>(B * 3) +
                          HostStage::need<double>(B) + 2 * HostStage::need<int32_t>(B * 4) + HostStage::need<uint32_t>(B) +
                          HostStage::need<double>(trace_doubles) + HostStage::need<double>(traj_doubles) + HostStage::need<unsigned char>(fp::kInlineMax),
                          /*zero_copy_out=*/B <= 8));
        // Latency regime with the tables resident (fp_batch.tables_tag): the per-ego arrays and the three sampling-range arrays ride
        // inside the lattice kernel's argument block, which also leaves them in a device mirror for the search and refinement
        // kernels - no copy kernel, no dependency in front of the lattice kernel (a single-ego FISS+ cycle: ~5 us of ~80).
        fa.io = *io;
        const bool try_inline = ctx->inline_inputs && B <= 8 && !params->curvature_mask && ctx->lattice_kernel != 1 && !big_points(fa.ka.p) &&
                                fp::lattice_group_fit(*params, *batch) >= 1;
        const InlineExtra ex[3] = {{io->samp_min, sizeof(double) * B * 3, (const void**)&fa.io.samp_min},
                                   {io->samp_max, sizeof(double) * B * 3, (const void**)&fa.io.samp_max},
                                   {io->samp_res, sizeof(double) * B * 3, (const void**)&fa.io.samp_res}};
        FP_TRY(stage_batch(hs, params, batch, &lat_b, try_inline ? &inl : nullptr, &fa.ka.b, ex, 3));
        if (!inl.on) {
            fa.ka.b = lat_b;
            FP_TRY(hs.in(io->samp_min, B * 3, &fa.io.samp_min));
            FP_TRY(hs.in(io->samp_max, B * 3, &fa.io.samp_max));
            FP_TRY(hs.in(io->samp_res, B * 3, &fa.io.samp_res));
        }
        FP_TRY(hs.in_mut(io->prev_best_idx, B * 3, &fa.io.prev_best_idx));
        FP_TRY(hs.flush_in());
        fa.io.best_ijk = hs.out(io->best_ijk, B * 3);
        fa.io.best_cost = hs.out(io->best_cost, B);
        fa.io.end_state = hs.out(io->end_state, B * 3);
        fa.io.refined = hs.out(io->refined, B);
        fa.io.stats = hs.out(io->stats, B * 4);
        fa.io.trace = trace_doubles ? hs.out(io->trace, trace_doubles) : nullptr;
        fa.io.best_flags = hs.out(io->best_flags, B);
        fa.io.best_traj = hs.out(io->best_traj, traj_doubles);
        if (io->traj_sparse && fa.io.best_traj) HIP_TRY(hipMemcpyAsync(fa.io.best_traj, io->best_traj, traj_doubles * sizeof(double), hipMemcpyDefault, ctx->stream));
    }
    FP_TRY(lattice_curv_scratch(ctx, params, batch, stream, &fa.ka.curv_tbl));
    int nsplit, group, tail; void* parts;
    FP_TRY(lattice_split_for(ctx, params, batch, stream, &nsplit, &parts, &group, &tail));
    const int* perm; int* dur;
    LaunchOrder* oslot = nullptr;
    FP_TRY(launch_order_before(ctx, ctx->order_lattice, ctx->resident_groups, batch, nsplit, stream, &perm, &dur, mem == FP_MEM_DEVICE ? batch->launch_order : nullptr, &oslot));
    bool search_done = false;
    if (inl.on) {
        fp::KernelArgs kl = fa.ka;
        kl.b = lat_b;
        LAUNCH_TRY(fp::launch_lattice(kl, stream, 2, parts, nsplit, nullptr, perm, dur, group, &inl, tail), "lattice kernel");
    } else {
        // the FISS+ search as workgroups appended to the lattice launch (launch_lattice_fused takes the offer for three-per-CU launches;
        // "fiss_fused" 0, the stage-timing diagnostic and every other shape: the search kernel follows in its own launch)
        fp::FissTail ft;
        ft.opts = fa.opts; ft.io = fa.io; ft.walk_jump = ctx->fiss_jump;
        ft.flag = (ctx->fiss_fused && ctx->appended_ok && ctx->fiss_stages >= 3 && opts->kind == FP_FISS_PLUS) ? epi_flags_for(ctx, B, stream) : nullptr;
        LAUNCH_TRY(fp::launch_lattice(fa.ka, stream, ctx->lattice_kernel, parts, nsplit, nullptr, perm, dur, group, nullptr, tail, ft.flag ? &ft : nullptr, &search_done),
                   "lattice kernel");
    }
    FP_TRY(launch_order_after(oslot, batch, dur, stream));
    if (ctx->fiss_stages < 2) return mem == FP_MEM_HOST ? hs.fetch_out() : FP_OK;  // timing diagnostic: outputs are not produced
    fa.walk_jump = ctx->fiss_jump;
    if (!search_done) LAUNCH_TRY(fp::launch_fiss_search(fa, stream), "search kernel");
    bool handed_over = false;
    if (R > 0 && ctx->fiss_stages >= 3) {
        // three refinement workgroups per CU are resident at once (fiss_refine_kernel: 168 VGPRs, ~52 KB LDS)
        const int* rperm; int* rdur;
        LaunchOrder* oslot_r = nullptr;
        FP_TRY(launch_order_before(ctx, ctx->order_refine, ctx->resident_groups / 2 * 3, batch, 1, stream, &rperm, &rdur, nullptr, &oslot_r));
        fp::FissArgs fr = fa;
        if (loop) {  // fp_plan_fiss_step: the refinement workgroup that settles an ego's trajectory hands the ego over itself
            fr.ka.loop = *loop;
            fr.ka.has_loop = 1;
            handed_over = true;
        }
        LAUNCH_TRY(fp::launch_fiss_refine(fr, stream, ctx->refine_table_kb, rperm, rdur), "refinement kernel");
        FP_TRY(launch_order_after(oslot_r, batch, rdur, stream));
    }
    if (fa.io.best_traj && R <= 0) {  // with refinement rounds the refinement kernel writes the series itself
        fp::KernelArgs kw = fa.ka;
        kw.r.best_flags = fa.io.best_flags;
        kw.r.best_traj = fa.io.best_traj;
        kw.r.traj_stride = fa.io.traj_stride;
        kw.r.traj_sparse = fa.io.traj_sparse;
        LAUNCH_TRY(fp::launch_winner_traj(kw, fa.io.end_state, stream), "winner epilogue");
    }
    if (loop && !handed_over) LAUNCH_TRY(fp::launch_advance(fa.ka, nullptr, fa.io.end_state, *loop, stream), "advance kernel");  // (FISS, or no refinement rounds)
    if (mem != FP_MEM_HOST) return FP_OK;
    FP_TRY(hs.fetch_out());
    if (handover_failed(ctx)) {  // (see fp_plan_dense)
        (void)handover_recover(ctx, ctx->stream);
        if (!prev_in.empty()) memcpy(io->prev_best_idx, prev_in.data(), prev_in.size() * sizeof(int32_t));
        return plan_fiss_impl(ctx, params, batch, opts, io, loop, mem, stream_v);
    }
    return FP_OK;
}

}  // namespace

extern "C" {

int fp_plan_fiss(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_fiss_opts* opts, const fp_fiss_io* io, int mem,
                 void* stream_v)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    if (!io || !overlap_applies(ctx, mem, batch, stream_v)) return plan_fiss_impl(ctx, params, batch, opts, io, nullptr, mem, stream_v);
    // ("overlap": two FISS / FISS+ pipelines of independent batches side by side - the one-round search and refinement launches of one
    // run beside the other's lattice kernel; prev_best_idx is in/out, so calls that share it run one behind the other)
    return overlapped_call(ctx, (hipStream_t)stream_v, overlap_outs(*io),
                           [&](fp_ctx* target, hipStream_t is) { return plan_fiss_impl(target, params, batch, opts, io, nullptr, FP_MEM_DEVICE, is); });
}

int fp_plan_fiss_step(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_fiss_opts* opts, const fp_fiss_io* io,
                      const fp_loop_io* loop, int mem, void* stream)
{
    if (!loop || !loop->ego || !loop->t_now || !loop->done || !loop->cycles || !loop->goal_xy) return fail(FP_EINVAL, "fp_loop_io has a NULL mandatory array");
    if (loop->goal_poly && (!loop->goal_nv || loop->goal_max_vertices < 3)) return fail(FP_EINVAL, "fp_loop_io.goal_poly needs goal_nv and goal_max_vertices >= 3");
    if (mem != FP_MEM_DEVICE) {  // host buffers: the two staged calls (every array travels anyway)
        if (!batch || !io) return fail(FP_EINVAL, "batch/io is NULL");
        fp_batch bb = *batch;
        bb.skip = loop->done;
        FP_TRY(plan_fiss_impl(ctx, params, &bb, opts, io, nullptr, mem, stream));
        return fp_advance(ctx, params, batch, nullptr, io->end_state, loop, mem, stream);
    }
    return plan_fiss_impl(ctx, params, batch, opts, io, loop, mem, stream);
}

int fp_advance(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const int32_t* best_idx, const double* end_state,
               const fp_loop_io* io, int mem, void* stream)
{
    FP_TRY(common_checks(ctx, params, batch, mem, stream));
    if ((best_idx == nullptr) == (end_state == nullptr)) return fail(FP_EINVAL, "exactly one of best_idx / end_state must be given");
    if (!io || !io->ego || !io->t_now || !io->done || !io->cycles || !io->goal_xy) return fail(FP_EINVAL, "fp_loop_io has a NULL mandatory array");
    if (io->goal_poly && (!io->goal_nv || io->goal_max_vertices < 3)) return fail(FP_EINVAL, "fp_loop_io.goal_poly needs goal_nv and goal_max_vertices >= 3");
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t B = (size_t)batch->B;
    fp::KernelArgs ka;
    ka.p = *params;
    ka.r = no_result();
    if (mem == FP_MEM_DEVICE) {
        ka.b = *batch;
        LAUNCH_TRY(fp::launch_advance(ka, best_idx, end_state, *io, (hipStream_t)stream), "advance kernel");
        return FP_OK;
    }
    FP_TRY(check_batch_host(params, batch));
    HostStage hs(ctx);
    const size_t goal_v = io->goal_poly && io->goal_max_vertices >= 3 ? (size_t)io->goal_max_vertices : 0;
    FP_TRY(hs.reserve(batch_need(params, batch) + HostStage::need<double>(B * 6) + 5 * HostStage::need<int32_t>(B) + HostStage::need<double>(B * 2) +
                      2 * HostStage::need<double>(B * 3) + HostStage::need<double>(B * goal_v * 2) + HostStage::need<double>(B * 6)));
    FP_TRY(stage_batch(hs, params, batch, &ka.b));
    fp_loop_io dio = *io;
    FP_TRY(hs.in_mut(io->ego, B * 6, &dio.ego));
    FP_TRY(hs.in_mut(io->t_now, B, &dio.t_now));
    FP_TRY(hs.in_mut(io->done, B, &dio.done));
    FP_TRY(hs.in_mut(io->cycles, B, &dio.cycles));
    FP_TRY(hs.in(io->goal_xy, B * 2, &dio.goal_xy));
    if (io->goal_poly && io->goal_nv && io->goal_max_vertices >= 3) {
        FP_TRY(hs.in(io->goal_poly, B * (size_t)io->goal_max_vertices * 2, &dio.goal_poly));
        FP_TRY(hs.in(io->goal_nv, B, &dio.goal_nv));
        if (io->goal_intervals) FP_TRY(hs.in(io->goal_intervals, B * 6, &dio.goal_intervals));
    } else {
        dio.goal_poly = nullptr;
    }
    const int32_t* d_idx = nullptr;
    const double* d_es = nullptr;
    if (best_idx) FP_TRY(hs.in(best_idx, B, &d_idx));
    if (end_state) FP_TRY(hs.in(end_state, B * 3, &d_es));
    FP_TRY(hs.flush_in());
    dio.cart_state = hs.out(io->cart_state, B * 3);
    if (dio.cart_state) HIP_TRY(hipMemsetAsync(dio.cart_state, 0xFF, sizeof(double) * B * 3, ctx->stream));  // NaN for egos that do not move
    LAUNCH_TRY(fp::launch_advance(ka, d_idx, d_es, dio, ctx->stream), "advance kernel");
    return hs.fetch_out();
}

int fp_plan_step(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_result* result, const fp_loop_io* io, int mem, void* stream)
{
    FP_TRY(common_checks(ctx, params, batch, mem, stream));
    if (!result || !result->best_idx || !result->best_cost) return fail(FP_EINVAL, "result.best_idx/best_cost must not be NULL");
    if (result->best_traj && !result->best_flags) return fail(FP_EINVAL, "result.best_traj requires result.best_flags");
    if (result->fopplus || result->audit) return fail(FP_EINVAL, "fp_plan_step: result.fopplus / result.audit must be NULL (use fp_plan_dense + fp_advance)");
    if (!io || !io->ego || !io->t_now || !io->done || !io->cycles || !io->goal_xy) return fail(FP_EINVAL, "fp_loop_io has a NULL mandatory array");
    if (io->goal_poly && (!io->goal_nv || io->goal_max_vertices < 3)) return fail(FP_EINVAL, "fp_loop_io.goal_poly needs goal_nv and goal_max_vertices >= 3");
    if (batch->B == 0) return FP_OK;
    if (mem != FP_MEM_DEVICE) {
        // host buffers: the two staged calls (every array travels anyway; the fused launch is for resident loops)
        fp_batch bb = *batch;
        bb.skip = io->done;
        FP_TRY(fp_plan_dense(ctx, params, &bb, result, mem, stream));
        return fp_advance(ctx, params, batch, result->best_idx, nullptr, io, mem, stream);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    FP_TRY(handover_check(ctx, (hipStream_t)stream));
    const size_t B = (size_t)batch->B;
    fp::KernelArgs ka;
    ka.p = *params;
    ka.err_word = ctx->hand_err;
    ka.occ_cap = ctx->lattice_occupancy; ka.resident2 = ctx->resident_groups; ka.lds_cu_kb = ctx->lds_cu_kb; ka.handover_timeout_us = ctx->handover_timeout_us;
    ka.b = *batch;
    ka.b.skip = io->done;
    if (!(batch->S > 0 && batch->n_obs > 0)) ka.b.n_obs = 0;
    ka.r = *result;
    FP_TRY(lattice_curv_scratch(ctx, params, batch, (hipStream_t)stream, &ka.curv_tbl));
    int nsplit, group, tail; void* parts;
    FP_TRY(lattice_split_for(ctx, params, batch, (hipStream_t)stream, &nsplit, &parts, &group, &tail));
    const int* perm; int* dur;
    LaunchOrder* oslot = nullptr;
    FP_TRY(launch_order_before(ctx, ctx->order_lattice, ctx->resident_groups, batch, nsplit, (hipStream_t)stream, &perm, &dur, batch->launch_order, &oslot));
    // the hand-over rides in the lattice launch unless that launch cannot write the series it is asked for itself (the standalone
    // epilogue reads the ego's state, so it has to run BEFORE the state moves on) or the lane-per-candidate kernel is asked for
    const bool series_elsewhere = result->best_traj && (!winner_inside_lattice(ctx, batch) || big_points(ka.p));
    bool try_fused = !series_elsewhere && ctx->lattice_kernel != 1, launched = false, fused = false;
    bool winner_done = false;
    if (try_fused) {
        fp::KernelArgs kl = ka;
        kl.loop = *io;
        kl.has_loop = 1;
        hipError_t e = fp::launch_lattice_fused(kl, (hipStream_t)stream, parts, nsplit, &winner_done, perm, dur, group, nullptr, tail, &fused);
        if (e == hipErrorInvalidValue) {  // the problem does not fit the fused kernel
            (void)hipGetLastError();
        } else if (e != hipSuccess) {
            return fail(FP_EHIP, "lattice kernel: %s", hipGetErrorString(e));
        } else {
            launched = true;  // (fused: the instance handed the egos over itself; else advance_kernel follows below)
        }
    }
    if (!launched) {
        if (series_elsewhere) ka.idx_shadow = idx_shadow_for(ctx, B, (hipStream_t)stream);
        fp::KernelArgs kl = ka;
        if (series_elsewhere) kl.r.best_traj = nullptr;
        LAUNCH_TRY(fp::launch_lattice(kl, (hipStream_t)stream, ctx->lattice_kernel, parts, nsplit, &winner_done, perm, dur, group, nullptr, tail), "lattice kernel");
    }
    FP_TRY(launch_order_after(oslot, batch, dur, (hipStream_t)stream));
    if (result->best_traj && !winner_done) LAUNCH_TRY(fp::launch_winner_traj(ka, nullptr, (hipStream_t)stream), "winner epilogue");
    if (!fused) LAUNCH_TRY(fp::launch_advance(ka, ka.r.best_idx, nullptr, *io, (hipStream_t)stream), "advance kernel");
    return FP_OK;
}

int fp_frames_build(fp_ctx* ctx, int32_t F, int32_t NX, const int32_t* n, const double* points, double* knots, double* coef, int mem,
                    void* stream)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    if (F < 0 || NX < 2 || NX > FP_MAX_KNOTS) return fail(FP_EINVAL, "bad sizes F=%d NX=%d", F, NX);
    if (!n || !points || !knots || !coef) return fail(FP_EINVAL, "NULL array");
    if (mem != FP_MEM_HOST && mem != FP_MEM_DEVICE) return fail(FP_EINVAL, "mem must be FP_MEM_HOST or FP_MEM_DEVICE");
    if (F == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (mem == FP_MEM_DEVICE) {
        LAUNCH_TRY(fp::launch_frames_build(F, NX, n, points, knots, coef, (hipStream_t)stream), "frame build");
        return FP_OK;
    }
    for (int f = 0; f < F; ++f)
        if (n[f] < 2 || n[f] > NX) return fail(FP_EINVAL, "n[%d]=%d out of range", f, n[f]);
    const size_t fn = (size_t)F * NX;
    HostStage hs(ctx);
    FP_TRY(hs.reserve(HostStage::need<int32_t>(F) + HostStage::need<double>(fn * 2) + HostStage::need<double>(fn) + HostStage::need<double>(fn * 8)));
    const int32_t* d_n = nullptr;
    const double* d_pts = nullptr;
    FP_TRY(hs.in(n, (size_t)F, &d_n));
    FP_TRY(hs.in(points, fn * 2, &d_pts));
    FP_TRY(hs.flush_in());
    double* d_k = hs.out(knots, fn);
    double* d_c = hs.out(coef, fn * 8);
    LAUNCH_TRY(fp::launch_frames_build(F, NX, d_n, d_pts, d_k, d_c, ctx->stream), "frame build");
    return hs.fetch_out();
}

int fp_from_state(fp_ctx* ctx, const fp_batch* batch, const double* states, double* ego, int mem, void* stream)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    if (!batch || !states || !ego) return fail(FP_EINVAL, "NULL argument");
    if (batch->B < 0 || batch->F < 1 || batch->NX < 2 || batch->NX > FP_MAX_KNOTS) return fail(FP_EINVAL, "bad batch sizes");
    if (!batch->frame_of || !batch->nx || !batch->knots || !batch->coef) return fail(FP_EINVAL, "batch frame arrays must not be NULL");
    if (mem != FP_MEM_HOST && mem != FP_MEM_DEVICE) return fail(FP_EINVAL, "mem must be FP_MEM_HOST or FP_MEM_DEVICE");
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (mem == FP_MEM_DEVICE) {
        LAUNCH_TRY(fp::launch_from_state(*batch, states, ego, (hipStream_t)stream), "from_state");
        return FP_OK;
    }
    const size_t B = (size_t)batch->B, fn = (size_t)batch->F * batch->NX;
    for (size_t i = 0; i < B; ++i)
        if (batch->frame_of[i] < 0 || batch->frame_of[i] >= batch->F) return fail(FP_EINVAL, "frame_of[%zu] out of range", i);
    HostStage hs(ctx);
    FP_TRY(hs.reserve(HostStage::need<int32_t>(B) + HostStage::need<int32_t>(batch->F) + HostStage::need<double>(fn) + HostStage::need<double>(fn * 8) +
                      HostStage::need<double>(B * 4) + HostStage::need<double>(B * 6)));
    fp_batch db = *batch;
    FP_TRY(hs.in(batch->frame_of, B, &db.frame_of));
    FP_TRY(hs.in(batch->nx, (size_t)batch->F, &db.nx));
    FP_TRY(hs.in(batch->knots, fn, &db.knots));
    FP_TRY(hs.in(batch->coef, fn * 8, &db.coef));
    const double* d_states = nullptr;
    FP_TRY(hs.in(states, B * 4, &d_states));
    FP_TRY(hs.flush_in());
    double* d_ego = hs.out(ego, B * 6);
    LAUNCH_TRY(fp::launch_from_state(db, d_states, d_ego, ctx->stream), "from_state");
    return hs.fetch_out();
}

int fp_eval_trajs(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, int32_t K, const double* end_states, double* cost,
                  uint32_t* flags, double* traj, int32_t traj_stride, int32_t traj_sparse, int mem, void* stream)
{
    FP_TRY(common_checks(ctx, params, batch, mem, stream));
    if (K < 1 || !end_states) return fail(FP_EINVAL, "K must be >= 1 and end_states non-NULL");
    int stride;
    FP_TRY(traj_stride_of(traj_stride, &stride));
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t BK = (size_t)batch->B * K;
    fp::KernelArgs ka;
    ka.p = *params;
    ka.r = no_result();
    if (mem == FP_MEM_DEVICE) {
        ka.b = *batch;
        if (!(batch->S > 0 && batch->n_obs > 0)) ka.b.n_obs = 0;
        LAUNCH_TRY(fp::launch_eval_trajs(ka, K, end_states, cost, flags, traj, stride, traj_sparse, (hipStream_t)stream), "eval kernel");
        return FP_OK;
    }
    FP_TRY(check_batch_host(params, batch));
    for (size_t i = 0; i < BK; ++i) {
        const double n = end_states[3 * i + 2] / params->tick_t;
        if (n != n) continue;  // NaN end state = "no trajectory": the kernels emit NaN cost / all-NaN series
        if (!(n > 0) || n > FP_MAX_POINTS) return fail(FP_ELIMIT, "end_states[%zu].T=%g needs 1..FP_MAX_POINTS points", i, end_states[3 * i + 2]);
        if (traj && ceil(n) > stride) return fail(FP_EINVAL, "traj_stride=%d is smaller than the %g points of end_states[%zu].T=%g", stride, ceil(n), i, end_states[3 * i + 2]);
    }
    ka.p.points_max = host_points_max(params, batch, end_states + 2, BK, 3);
    const size_t traj_doubles = traj ? BK * FP_ARR_COUNT * (size_t)stride : 0;
    HostStage hs(ctx);
    FP_TRY(hs.reserve(batch_need(params, batch) + HostStage::need<double>(BK * 3) + HostStage::need<double>(BK) + HostStage::need<uint32_t>(BK) +
                      HostStage::need<double>(traj_doubles)));
    FP_TRY(stage_batch(hs, params, batch, &ka.b));
    const double* d_end = nullptr;
    FP_TRY(hs.in(end_states, BK * 3, &d_end));
    FP_TRY(hs.flush_in());
    double* d_cost = cost ? hs.out(cost, BK) : hs.temp<double>(BK);
    uint32_t* d_flags = flags ? hs.out(flags, BK) : hs.temp<uint32_t>(BK);
    double* d_traj = hs.out(traj, traj_doubles);
    if (traj_sparse && d_traj) HIP_TRY(hipMemcpyAsync(d_traj, traj, traj_doubles * sizeof(double), hipMemcpyDefault, ctx->stream));
    LAUNCH_TRY(fp::launch_eval_trajs(ka, K, d_end, d_cost, d_flags, d_traj, stride, traj_sparse, ctx->stream), "eval kernel");
    return hs.fetch_out();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------- fp_group
// One persistent host thread per ctx, fed through a one-deep mailbox (include/frenet_gpu.h).  posted / done are sequence numbers:
// the submitter copies a call into the worker's slot and increments `posted`; the worker runs it and sets `done` = the number it
// ran.  A worker spins on `posted` for kSpinRounds pause instructions after its last call (~50 us: the next step of a running loop
// arrives well inside that), then sleeps on a condition variable; the submitter notifies only when it finds the worker asleep.
namespace {

inline void cpu_pause()
{
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
    __builtin_ia32_pause();
#endif
}

constexpr int kSpinRounds = 4000;
constexpr int kMaxCopies = 8;

struct GroupWorker {
    fp_ctx* ctx = nullptr;
    std::thread th;
    // the mailbox (written by the submitter while done == posted, read by the worker after it saw posted move)
    int kind = 0;  // 1 dense, 2 step, 3 fiss, 4 fiss + advance
    fp_params params;
    fp_batch batch;
    fp_result result;
    fp_loop_io loop;
    fp_fiss_opts fopts;
    fp_fiss_io fio;
    void* stream = nullptr;
    fp_copy copies[kMaxCopies];
    int n_copies = 0;
    std::atomic<unsigned long long> posted{0}, done{0};
    std::atomic<bool> asleep{false}, stop{false};
    std::mutex m;
    std::condition_variable cv;
    int rc = FP_OK;        // of the last call (read by fp_group_wait after done == posted)
    std::string err;
};

void group_worker_main(GroupWorker* w)
{
    (void)hipSetDevice(w->ctx->device);
    unsigned long long seen = 0;
    for (;;) {
        int spins = 0;
        while (w->posted.load(std::memory_order_acquire) == seen && !w->stop.load(std::memory_order_relaxed)) {
            if (++spins < kSpinRounds) { cpu_pause(); continue; }
            std::unique_lock<std::mutex> lk(w->m);
            w->asleep.store(true, std::memory_order_seq_cst);
            w->cv.wait(lk, [&] { return w->posted.load(std::memory_order_seq_cst) != seen || w->stop.load(std::memory_order_seq_cst); });
            w->asleep.store(false, std::memory_order_seq_cst);
            spins = 0;
        }
        if (w->stop.load(std::memory_order_relaxed) && w->posted.load(std::memory_order_acquire) == seen) return;
        seen = w->posted.load(std::memory_order_acquire);
        int rc = FP_OK;
        switch (w->kind) {
            case 1: rc = fp_plan_dense(w->ctx, &w->params, &w->batch, &w->result, FP_MEM_DEVICE, w->stream); break;
            case 2: rc = fp_plan_step(w->ctx, &w->params, &w->batch, &w->result, &w->loop, FP_MEM_DEVICE, w->stream); break;
            case 3: rc = fp_plan_fiss(w->ctx, &w->params, &w->batch, &w->fopts, &w->fio, FP_MEM_DEVICE, w->stream); break;
            case 4: rc = fp_plan_fiss_step(w->ctx, &w->params, &w->batch, &w->fopts, &w->fio, &w->loop, FP_MEM_DEVICE, w->stream); break;
            default: break;
        }
        for (int i = 0; rc == FP_OK && i < w->n_copies; ++i) {
            const hipError_t e = hipMemcpyAsync(w->copies[i].dst, w->copies[i].src, w->copies[i].bytes, hipMemcpyDefault, (hipStream_t)w->stream);
            if (e != hipSuccess) rc = fail(FP_EHIP, "fp_group copy %d failed: %s", i, hipGetErrorString(e));
        }
        w->rc = rc;
        if (rc != FP_OK) w->err = g_last_error;  // (thread-local: the worker's own)
        w->done.store(seen, std::memory_order_release);
    }
}

void group_wait_worker(GroupWorker& w)
{
    const unsigned long long want = w.posted.load(std::memory_order_relaxed);
    while (w.done.load(std::memory_order_acquire) != want) cpu_pause();
}

}  // namespace

struct fp_group {
    std::vector<GroupWorker*> workers;
};

extern "C" {

int fp_group_create(fp_ctx* const* ctxs, int32_t n, fp_group** out)
{
    if (!out) return fail(FP_EINVAL, "out is NULL");
    *out = nullptr;
    if (!ctxs || n < 1 || n > 1024) return fail(FP_EINVAL, "fp_group_create: need 1..1024 contexts");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(FP_EINVAL, "fp_group_create: ctxs[%d] is NULL", i);
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(FP_EINVAL, "fp_group_create: ctxs[%d] and ctxs[%d] are the same ctx (one worker per ctx)", j, i);
    }
    fp_group* g = new (std::nothrow) fp_group();
    if (!g) return fail(FP_ENOMEM, "out of host memory");
    for (int i = 0; i < n; ++i) {
        GroupWorker* w = new (std::nothrow) GroupWorker();
        if (!w) { fp_group_destroy(g); return fail(FP_ENOMEM, "out of host memory"); }
        w->ctx = ctxs[i];
        try {  // (no exception may leave an extern "C" function: vector growth and thread creation both can throw)
            g->workers.push_back(w);
        } catch (...) {
            delete w;
            fp_group_destroy(g);
            return fail(FP_ENOMEM, "out of host memory");
        }
        try {
            w->th = std::thread(group_worker_main, w);
        } catch (const std::exception& ex) {  // std::system_error: the process is out of threads
            fp_group_destroy(g);  // (joins the workers that did start; this one is not joinable)
            return fail(FP_EHIP, "fp_group_create: worker thread %d could not be started: %s", i, ex.what());
        }
    }
    *out = g;
    return FP_OK;
}

int fp_group_destroy(fp_group* g)
{
    if (!g) return FP_OK;
    for (GroupWorker* w : g->workers) {
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->stop.store(true, std::memory_order_seq_cst);
        }
        w->cv.notify_one();
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    delete g;
    return FP_OK;
}

int fp_group_submit(fp_group* g, const fp_shard_call* calls)
{
    if (!g || !calls) return fail(FP_EINVAL, "group/calls is NULL");
    const int n = (int)g->workers.size();
    for (int i = 0; i < n; ++i) {  // validate the whole round before any of it is posted
        const fp_shard_call& c = calls[i];
        if (!c.params) continue;
        if (!c.batch) return fail(FP_EINVAL, "fp_group_submit: calls[%d].batch is NULL", i);
        if (!c.result && !(c.fiss_opts && c.fiss_io)) return fail(FP_EINVAL, "fp_group_submit: calls[%d] needs result or fiss_opts + fiss_io", i);
        if (c.result && (c.fiss_opts || c.fiss_io)) return fail(FP_EINVAL, "fp_group_submit: calls[%d] sets both result and the FISS structs", i);
        if (c.n_copies < 0 || c.n_copies > kMaxCopies || (c.n_copies > 0 && !c.copies)) return fail(FP_EINVAL, "fp_group_submit: calls[%d].n_copies must be 0..%d", i, kMaxCopies);
    }
    // A shard whose earlier call failed takes no new work until fp_group_wait has reported (and cleared) the failure - and the round is
    // refused as a whole, before any of it is posted: the other shards must not run ahead of a shard that silently skipped its call.
    for (int i = 0; i < n; ++i) {
        if (!calls[i].params) continue;
        GroupWorker& w = *g->workers[i];
        group_wait_worker(w);  // the mailbox is one deep
        if (w.rc != FP_OK) return fail(w.rc, "fp_group_submit: shard %d still holds the error of an earlier call (%s); nothing of this round was posted - fp_group_wait reports and clears it", i, w.err.c_str());
    }
    for (int i = 0; i < n; ++i) {
        const fp_shard_call& c = calls[i];
        if (!c.params) continue;
        GroupWorker& w = *g->workers[i];
        w.params = *c.params;
        w.batch = *c.batch;
        if (c.result) w.result = *c.result;
        if (c.loop) w.loop = *c.loop;
        if (c.fiss_opts) { w.fopts = *c.fiss_opts; w.fio = *c.fiss_io; }
        w.kind = c.result ? (c.loop ? 2 : 1) : (c.loop ? 4 : 3);
        w.stream = c.stream;
        w.n_copies = c.n_copies;
        for (int k = 0; k < c.n_copies; ++k) w.copies[k] = c.copies[k];
        w.posted.fetch_add(1, std::memory_order_seq_cst);
        if (w.asleep.load(std::memory_order_seq_cst)) {
            std::lock_guard<std::mutex> lk(w.m);
            w.cv.notify_one();
        }
    }
    return FP_OK;
}

int fp_group_wait(fp_group* g)
{
    if (!g) return fail(FP_EINVAL, "group is NULL");
    int rc = FP_OK;
    for (size_t i = 0; i < g->workers.size(); ++i) {
        GroupWorker& w = *g->workers[i];
        group_wait_worker(w);
        if (w.rc != FP_OK) {
            if (rc == FP_OK) rc = fail(w.rc, "shard %zu: %s", i, w.err.c_str());
            w.rc = FP_OK;
            w.err.clear();
        }
    }
    return rc;
}

}  // extern "C"
