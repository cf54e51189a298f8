// frenet_abi.hip - the C ABI of libfrenetgpu.so (include/frenet_gpu.h).
//
// Host-side responsibilities only: argument validation, the per-ctx device staging
// arena used by FP_MEM_HOST calls, error strings.  No planning arithmetic happens on
// the host: if there is no usable GPU every entry point fails with FP_ENODEV/FP_EHIP.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "frenet_kernels.h"

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

// A grow-only device arena; host-memory calls carve their staging copies out of it.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return FP_OK;
        if (base) {
            hipError_t e = hipFree(base);
            base = nullptr;
            cap = 0;
            if (e != hipSuccess) return fail(FP_EHIP, "hipFree failed: %s", hipGetErrorString(e));
        }
        size_t want = bytes + bytes / 4 + (1u << 20);
        hipError_t e = hipMalloc((void**)&base, want);
        if (e != hipSuccess) return fail(FP_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        cap = want;
        return FP_OK;
    }
    void reset() { used = 0; }
    void* take(size_t bytes)
    {
        size_t off = (used + 255) & ~size_t(255);
        used = off + bytes;
        return base + off;
    }
    static size_t padded(size_t bytes) { return ((bytes + 255) & ~size_t(255)) + 256; }
};

}  // namespace

struct fp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  // used by FP_MEM_HOST calls
    Arena arena;                   // staging copies of FP_MEM_HOST calls
    Arena scratch;                 // intermediate tables of multi-kernel entry points (fp_plan_fiss)
    int lattice_kernel = 0;        // fp_ctx_set_option("lattice_kernel")
};

namespace {

int check_params(const fp_params* p)
{
    if (!p) return fail(FP_EINVAL, "params is NULL");
    if (p->nd < 1 || p->nv < 1 || p->nt < 1) return fail(FP_EINVAL, "lattice sizes must be >= 1 (got %d,%d,%d)", p->nd, p->nv, p->nt);
    if ((long)p->nd * p->nv * p->nt > FP_MAX_CAND) return fail(FP_ELIMIT, "nd*nv*nt = %ld exceeds FP_MAX_CAND", (long)p->nd * p->nv * p->nt);
    if (p->check_stride < 1) return fail(FP_EINVAL, "check_stride must be >= 1");
    if (!(p->tick_t > 0)) return fail(FP_EINVAL, "tick_t must be > 0");
    return FP_OK;
}

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
    return FP_OK;
}

struct Staged {
    fp_batch dev;
    size_t bytes = 0;
};

size_t batch_bytes(const fp_params* p, const fp_batch* b)
{
    size_t n = 0;
    n += Arena::padded(sizeof(double) * p->nd) + Arena::padded(sizeof(double) * p->nt);
    n += Arena::padded(sizeof(double) * (size_t)b->B * p->nv) + Arena::padded(sizeof(double) * b->B);
    n += Arena::padded(sizeof(double) * (size_t)b->B * 6) + 3 * Arena::padded(sizeof(int32_t) * b->B);
    n += Arena::padded(sizeof(int32_t) * b->F) + Arena::padded(sizeof(double) * (size_t)b->F * b->NX) +
         Arena::padded(sizeof(double) * (size_t)b->F * 8 * b->NX);
    n += Arena::padded(sizeof(double) * (size_t)b->S * b->T_obs * b->n_obs * 4) + Arena::padded(sizeof(double) * (size_t)b->S * b->n_obs * 2) +
         Arena::padded(sizeof(int32_t) * (b->S > 0 ? b->S : 1)) + Arena::padded(sizeof(int32_t) * b->B);
    return n;
}

template <typename T>
int push(fp_ctx* ctx, const T* host, size_t count, const T** dev_out)
{
    T* d = (T*)ctx->arena.take(sizeof(T) * (count ? count : 1));
    if (count) HIP_TRY(hipMemcpyAsync(d, host, sizeof(T) * count, hipMemcpyHostToDevice, ctx->stream));
    *dev_out = d;
    return FP_OK;
}

int stage_batch(fp_ctx* ctx, const fp_params* p, const fp_batch* b, fp_batch* dev)
{
    *dev = *b;
    int rc;
#define PUSH(field, count) if ((rc = push(ctx, b->field, (size_t)(count), &dev->field)) != FP_OK) return rc
    PUSH(d_samples, p->nd);
    PUSH(t_samples, p->nt);
    PUSH(v_samples, (size_t)b->B * p->nv);
    PUSH(target_speed, b->B);
    PUSH(ego, (size_t)b->B * 6);
    PUSH(frame_of, b->B);
    PUSH(scene_of, b->B);
    PUSH(t_now, b->B);
    PUSH(nx, b->F);
    PUSH(knots, (size_t)b->F * b->NX);
    PUSH(coef, (size_t)b->F * 8 * b->NX);
    const bool has_obs = b->S > 0 && b->n_obs > 0;
    PUSH(obs_pose, has_obs ? (size_t)b->S * b->T_obs * b->n_obs * 4 : 0);
    PUSH(obs_dims, has_obs ? (size_t)b->S * b->n_obs * 2 : 0);
    PUSH(final_time_step, has_obs ? b->S : 0);
    if (b->skip) { PUSH(skip, b->B); }
#undef PUSH
    if (!has_obs) dev->n_obs = 0;
    return FP_OK;
}

}  // namespace

extern "C" {

int fp_abi_version(void) { return FP_ABI_VERSION; }

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
    if (e != hipSuccess) {
        delete ctx;
        return fail(FP_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return FP_OK;
}

int fp_ctx_destroy(fp_ctx* ctx)
{
    if (!ctx) return FP_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->arena.base) (void)hipFree(ctx->arena.base);
    if (ctx->scratch.base) (void)hipFree(ctx->scratch.base);
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
    return fail(FP_EINVAL, "unknown option '%s'", name);
}

int fp_plan_dense(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_result* result, int mem, void* stream)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    int rc;
    if ((rc = check_params(params)) != FP_OK) return rc;
    if ((rc = check_batch(batch)) != FP_OK) return rc;
    if (!result || !result->best_idx || !result->best_cost) return fail(FP_EINVAL, "result.best_idx/best_cost must not be NULL");
    if (result->best_traj && !result->best_flags) return fail(FP_EINVAL, "result.best_traj requires result.best_flags");
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t C = (size_t)params->nd * params->nv * params->nt;
    const size_t B = (size_t)batch->B;
    fp::KernelArgs ka;
    ka.p = *params;
    if (mem == FP_MEM_DEVICE) {
        ka.b = *batch;
        if (!(batch->S > 0 && batch->n_obs > 0)) ka.b.n_obs = 0;
        ka.r = *result;
        hipError_t e = fp::launch_lattice(ka, (hipStream_t)stream, ctx->lattice_kernel);
        if (e != hipSuccess) return fail(FP_EHIP, "lattice kernel launch failed: %s", hipGetErrorString(e));
        if (result->best_traj) {
            e = fp::launch_winner_traj(ka, nullptr, (hipStream_t)stream);
            if (e != hipSuccess) return fail(FP_EHIP, "winner epilogue launch failed: %s", hipGetErrorString(e));
        }
        return FP_OK;
    }
    if (mem != FP_MEM_HOST) return fail(FP_EINVAL, "mem must be FP_MEM_HOST or FP_MEM_DEVICE");
    if ((rc = check_batch_host(params, batch)) != FP_OK) return rc;
    size_t need = batch_bytes(params, batch) + Arena::padded(sizeof(int32_t) * B) + Arena::padded(sizeof(double) * B) +
                  Arena::padded(sizeof(int32_t) * B * 4);
    if (result->cost_tbl) need += Arena::padded(sizeof(double) * B * C);
    if (result->flag_tbl) need += Arena::padded(sizeof(uint32_t) * B * C);
    const size_t traj_doubles = result->best_traj ? B * FP_ARR_COUNT * (size_t)FP_MAX_POINTS : 0;
    need += Arena::padded(sizeof(uint32_t) * B) + Arena::padded(sizeof(double) * traj_doubles);
    if ((rc = ctx->arena.reserve(need)) != FP_OK) return rc;
    ctx->arena.reset();
    if ((rc = stage_batch(ctx, params, batch, &ka.b)) != FP_OK) return rc;
    ka.r.best_idx = (int32_t*)ctx->arena.take(sizeof(int32_t) * B);
    ka.r.best_cost = (double*)ctx->arena.take(sizeof(double) * B);
    ka.r.stats = result->stats ? (int32_t*)ctx->arena.take(sizeof(int32_t) * B * 4) : nullptr;
    ka.r.cost_tbl = result->cost_tbl ? (double*)ctx->arena.take(sizeof(double) * B * C) : nullptr;
    ka.r.flag_tbl = result->flag_tbl ? (uint32_t*)ctx->arena.take(sizeof(uint32_t) * B * C) : nullptr;
    ka.r.best_flags = result->best_flags ? (uint32_t*)ctx->arena.take(sizeof(uint32_t) * B) : nullptr;
    ka.r.best_traj = result->best_traj ? (double*)ctx->arena.take(sizeof(double) * traj_doubles) : nullptr;
    hipError_t e = fp::launch_lattice(ka, ctx->stream, ctx->lattice_kernel);
    if (e != hipSuccess) return fail(FP_EHIP, "lattice kernel launch failed: %s", hipGetErrorString(e));
    if (result->best_traj) {
        e = fp::launch_winner_traj(ka, nullptr, ctx->stream);
        if (e != hipSuccess) return fail(FP_EHIP, "winner epilogue launch failed: %s", hipGetErrorString(e));
        HIP_TRY(hipMemcpyAsync(result->best_traj, ka.r.best_traj, sizeof(double) * traj_doubles, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(result->best_flags, ka.r.best_flags, sizeof(uint32_t) * B, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipMemcpyAsync(result->best_idx, ka.r.best_idx, sizeof(int32_t) * B, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(result->best_cost, ka.r.best_cost, sizeof(double) * B, hipMemcpyDeviceToHost, ctx->stream));
    if (result->stats) HIP_TRY(hipMemcpyAsync(result->stats, ka.r.stats, sizeof(int32_t) * B * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (result->cost_tbl) HIP_TRY(hipMemcpyAsync(result->cost_tbl, ka.r.cost_tbl, sizeof(double) * B * C, hipMemcpyDeviceToHost, ctx->stream));
    if (result->flag_tbl) HIP_TRY(hipMemcpyAsync(result->flag_tbl, ka.r.flag_tbl, sizeof(uint32_t) * B * C, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FP_OK;
}

int fp_winner_trajs(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const int32_t* best_idx, uint32_t* best_flags,
                    double* best_traj, int mem, void* stream)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    int rc;
    if ((rc = check_params(params)) != FP_OK) return rc;
    if ((rc = check_batch(batch)) != FP_OK) return rc;
    if (!best_idx || !best_flags || !best_traj) return fail(FP_EINVAL, "best_idx/best_flags/best_traj must not be NULL");
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t B = (size_t)batch->B;
    const size_t traj_doubles = B * FP_ARR_COUNT * (size_t)FP_MAX_POINTS;
    fp::KernelArgs ka;
    ka.p = *params;
    ka.r = fp_result{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (mem == FP_MEM_DEVICE) {
        ka.b = *batch;
        ka.r.best_idx = const_cast<int32_t*>(best_idx);
        ka.r.best_flags = best_flags;
        ka.r.best_traj = best_traj;
        hipError_t e = fp::launch_winner_traj(ka, nullptr, (hipStream_t)stream);
        if (e != hipSuccess) return fail(FP_EHIP, "winner epilogue launch failed: %s", hipGetErrorString(e));
        return FP_OK;
    }
    if (mem != FP_MEM_HOST) return fail(FP_EINVAL, "mem must be FP_MEM_HOST or FP_MEM_DEVICE");
    if ((rc = check_batch_host(params, batch)) != FP_OK) return rc;
    const int C = params->nd * params->nv * params->nt;
    for (size_t i = 0; i < B; ++i)
        if (best_idx[i] >= C) return fail(FP_EINVAL, "best_idx[%zu]=%d out of range", i, best_idx[i]);
    size_t need = batch_bytes(params, batch) + Arena::padded(sizeof(int32_t) * B) + Arena::padded(sizeof(uint32_t) * B) +
                  Arena::padded(sizeof(double) * traj_doubles);
    if ((rc = ctx->arena.reserve(need)) != FP_OK) return rc;
    ctx->arena.reset();
    if ((rc = stage_batch(ctx, params, batch, &ka.b)) != FP_OK) return rc;
    const int32_t* d_idx = nullptr;
    if ((rc = push(ctx, best_idx, B, &d_idx)) != FP_OK) return rc;
    ka.r.best_idx = const_cast<int32_t*>(d_idx);
    ka.r.best_flags = (uint32_t*)ctx->arena.take(sizeof(uint32_t) * B);
    ka.r.best_traj = (double*)ctx->arena.take(sizeof(double) * traj_doubles);
    hipError_t e = fp::launch_winner_traj(ka, nullptr, ctx->stream);
    if (e != hipSuccess) return fail(FP_EHIP, "winner epilogue launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(best_traj, ka.r.best_traj, sizeof(double) * traj_doubles, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(best_flags, ka.r.best_flags, sizeof(uint32_t) * B, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FP_OK;
}

int fp_plan_fiss(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const fp_fiss_opts* opts, const fp_fiss_io* io, int mem,
                 void* stream_v)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    int rc;
    if ((rc = check_params(params)) != FP_OK) return rc;
    if ((rc = check_batch(batch)) != FP_OK) return rc;
    if (!opts || !io) return fail(FP_EINVAL, "opts/io is NULL");
    if (opts->kind != FP_FISS && opts->kind != FP_FISS_PLUS) return fail(FP_EINVAL, "opts.kind must be FP_FISS or FP_FISS_PLUS");
    if (opts->max_refine_iters < 0 || opts->max_refine_iters * 7 > 64) return fail(FP_ELIMIT, "max_refine_iters must be in 0..9");
    if (!io->samp_min || !io->samp_max || !io->samp_res || !io->prev_best_idx || !io->best_ijk || !io->best_cost || !io->end_state ||
        !io->refined || !io->stats)
        return fail(FP_EINVAL, "fp_fiss_io has a NULL mandatory array");
    if (io->best_traj && !io->best_flags) return fail(FP_EINVAL, "io.best_traj requires io.best_flags");
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t B = (size_t)batch->B, C = (size_t)params->nd * params->nv * params->nt;
    const int R = opts->kind == FP_FISS_PLUS ? opts->max_refine_iters : 0;
    hipStream_t stream = mem == FP_MEM_DEVICE ? (hipStream_t)stream_v : ctx->stream;
    if (mem != FP_MEM_DEVICE && mem != FP_MEM_HOST) return fail(FP_EINVAL, "mem must be FP_MEM_HOST or FP_MEM_DEVICE");

    // dense tables + per-ego dense results live in the scratch arena in both modes
    const size_t scratch_need = Arena::padded(sizeof(double) * B * C) + Arena::padded(sizeof(uint32_t) * B * C) +
                                Arena::padded(sizeof(int32_t) * B) + Arena::padded(sizeof(double) * B);
    if (scratch_need > ctx->scratch.cap) HIP_TRY(hipStreamSynchronize(stream));  // the arena may be reallocated: drain its users
    if ((rc = ctx->scratch.reserve(scratch_need)) != FP_OK) return rc;
    ctx->scratch.reset();
    fp::FissArgs fa;
    fa.ka.p = *params;
    fa.opts = *opts;
    fa.opts.max_refine_iters = R;
    fa.ka.r = fp_result{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    fa.ka.r.cost_tbl = (double*)ctx->scratch.take(sizeof(double) * B * C);
    fa.ka.r.flag_tbl = (uint32_t*)ctx->scratch.take(sizeof(uint32_t) * B * C);
    fa.ka.r.best_idx = (int32_t*)ctx->scratch.take(sizeof(int32_t) * B);
    fa.ka.r.best_cost = (double*)ctx->scratch.take(sizeof(double) * B);
    fa.cost_tbl = fa.ka.r.cost_tbl;
    fa.flag_tbl = fa.ka.r.flag_tbl;
    const size_t traj_doubles = io->best_traj ? B * FP_ARR_COUNT * (size_t)FP_MAX_POINTS : 0;
    const size_t trace_doubles = (io->trace && R > 0) ? B * (size_t)R * 7 * 4 : 0;
    if (mem == FP_MEM_DEVICE) {
        fa.ka.b = *batch;
        if (!(batch->S > 0 && batch->n_obs > 0)) fa.ka.b.n_obs = 0;
        fa.io = *io;
    } else {
        if ((rc = check_batch_host(params, batch)) != FP_OK) return rc;
        size_t need = batch_bytes(params, batch) + 3 * Arena::padded(sizeof(double) * B * 3) + 2 * Arena::padded(sizeof(int32_t) * B * 3) +
                      Arena::padded(sizeof(double) * B) + Arena::padded(sizeof(double) * B * 3) + Arena::padded(sizeof(int32_t) * B) +
                      Arena::padded(sizeof(int32_t) * B * 4) + Arena::padded(sizeof(double) * trace_doubles) +
                      Arena::padded(sizeof(uint32_t) * B) + Arena::padded(sizeof(double) * traj_doubles);
        if ((rc = ctx->arena.reserve(need)) != FP_OK) return rc;
        ctx->arena.reset();
        if ((rc = stage_batch(ctx, params, batch, &fa.ka.b)) != FP_OK) return rc;
        fa.io = *io;
        if ((rc = push(ctx, io->samp_min, B * 3, &fa.io.samp_min)) != FP_OK) return rc;
        if ((rc = push(ctx, io->samp_max, B * 3, &fa.io.samp_max)) != FP_OK) return rc;
        if ((rc = push(ctx, io->samp_res, B * 3, &fa.io.samp_res)) != FP_OK) return rc;
        const int32_t* d_prev = nullptr;
        if ((rc = push(ctx, (const int32_t*)io->prev_best_idx, B * 3, &d_prev)) != FP_OK) return rc;
        fa.io.prev_best_idx = const_cast<int32_t*>(d_prev);
        fa.io.best_ijk = (int32_t*)ctx->arena.take(sizeof(int32_t) * B * 3);
        fa.io.best_cost = (double*)ctx->arena.take(sizeof(double) * B);
        fa.io.end_state = (double*)ctx->arena.take(sizeof(double) * B * 3);
        fa.io.refined = (int32_t*)ctx->arena.take(sizeof(int32_t) * B);
        fa.io.stats = (int32_t*)ctx->arena.take(sizeof(int32_t) * B * 4);
        fa.io.trace = trace_doubles ? (double*)ctx->arena.take(sizeof(double) * trace_doubles) : nullptr;
        fa.io.best_flags = io->best_flags ? (uint32_t*)ctx->arena.take(sizeof(uint32_t) * B) : nullptr;
        fa.io.best_traj = io->best_traj ? (double*)ctx->arena.take(sizeof(double) * traj_doubles) : nullptr;
    }
    if (!(io->trace && R > 0)) fa.io.trace = nullptr;
    hipError_t e = fp::launch_lattice(fa.ka, stream, ctx->lattice_kernel);
    if (e != hipSuccess) return fail(FP_EHIP, "lattice kernel launch failed: %s", hipGetErrorString(e));
    e = fp::launch_fiss_search(fa, stream);
    if (e != hipSuccess) return fail(FP_EHIP, "search kernel launch failed: %s", hipGetErrorString(e));
    if (R > 0) {
        e = fp::launch_fiss_refine(fa, stream);
        if (e != hipSuccess) return fail(FP_EHIP, "refinement kernel launch failed: %s", hipGetErrorString(e));
    }
    if (fa.io.best_traj) {
        fp::KernelArgs kw = fa.ka;
        kw.r.best_flags = fa.io.best_flags;
        kw.r.best_traj = fa.io.best_traj;
        e = fp::launch_winner_traj(kw, fa.io.end_state, stream);
        if (e != hipSuccess) return fail(FP_EHIP, "winner epilogue launch failed: %s", hipGetErrorString(e));
    }
    if (mem == FP_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(io->prev_best_idx, fa.io.prev_best_idx, sizeof(int32_t) * B * 3, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(io->best_ijk, fa.io.best_ijk, sizeof(int32_t) * B * 3, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(io->best_cost, fa.io.best_cost, sizeof(double) * B, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(io->end_state, fa.io.end_state, sizeof(double) * B * 3, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(io->refined, fa.io.refined, sizeof(int32_t) * B, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(io->stats, fa.io.stats, sizeof(int32_t) * B * 4, hipMemcpyDeviceToHost, stream));
        if (trace_doubles) HIP_TRY(hipMemcpyAsync(io->trace, fa.io.trace, sizeof(double) * trace_doubles, hipMemcpyDeviceToHost, stream));
        if (io->best_flags) HIP_TRY(hipMemcpyAsync(io->best_flags, fa.io.best_flags, sizeof(uint32_t) * B, hipMemcpyDeviceToHost, stream));
        if (io->best_traj) HIP_TRY(hipMemcpyAsync(io->best_traj, fa.io.best_traj, sizeof(double) * traj_doubles, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    return FP_OK;
}

int fp_advance(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, const int32_t* best_idx, const double* end_state,
               const fp_loop_io* io, int mem, void* stream)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    int rc;
    if ((rc = check_params(params)) != FP_OK) return rc;
    if ((rc = check_batch(batch)) != FP_OK) return rc;
    if ((best_idx == nullptr) == (end_state == nullptr)) return fail(FP_EINVAL, "exactly one of best_idx / end_state must be given");
    if (!io || !io->ego || !io->t_now || !io->done || !io->cycles || !io->goal_xy) return fail(FP_EINVAL, "fp_loop_io has a NULL mandatory array");
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t B = (size_t)batch->B;
    fp::KernelArgs ka;
    ka.p = *params;
    ka.r = fp_result{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (mem == FP_MEM_DEVICE) {
        ka.b = *batch;
        hipError_t e = fp::launch_advance(ka, best_idx, end_state, *io, (hipStream_t)stream);
        if (e != hipSuccess) return fail(FP_EHIP, "advance kernel launch failed: %s", hipGetErrorString(e));
        return FP_OK;
    }
    if (mem != FP_MEM_HOST) return fail(FP_EINVAL, "mem must be FP_MEM_HOST or FP_MEM_DEVICE");
    if ((rc = check_batch_host(params, batch)) != FP_OK) return rc;
    size_t need = batch_bytes(params, batch) + Arena::padded(sizeof(double) * B * 6) + 3 * Arena::padded(sizeof(int32_t) * B) +
                  Arena::padded(sizeof(double) * B * 2) + 2 * Arena::padded(sizeof(double) * B * 3) + Arena::padded(sizeof(int32_t) * B);
    if ((rc = ctx->arena.reserve(need)) != FP_OK) return rc;
    ctx->arena.reset();
    if ((rc = stage_batch(ctx, params, batch, &ka.b)) != FP_OK) return rc;
    fp_loop_io dio = *io;
    const double* c_ego = nullptr; const int32_t *c_tn = nullptr, *c_done = nullptr, *c_cyc = nullptr, *c_idx = nullptr; const double* c_es = nullptr;
    if ((rc = push(ctx, (const double*)io->ego, B * 6, &c_ego)) != FP_OK) return rc;
    if ((rc = push(ctx, (const int32_t*)io->t_now, B, &c_tn)) != FP_OK) return rc;
    if ((rc = push(ctx, (const int32_t*)io->done, B, &c_done)) != FP_OK) return rc;
    if ((rc = push(ctx, (const int32_t*)io->cycles, B, &c_cyc)) != FP_OK) return rc;
    if ((rc = push(ctx, io->goal_xy, B * 2, &dio.goal_xy)) != FP_OK) return rc;
    if (best_idx && (rc = push(ctx, best_idx, B, &c_idx)) != FP_OK) return rc;
    if (end_state && (rc = push(ctx, end_state, B * 3, &c_es)) != FP_OK) return rc;
    dio.ego = const_cast<double*>(c_ego); dio.t_now = const_cast<int32_t*>(c_tn);
    dio.done = const_cast<int32_t*>(c_done); dio.cycles = const_cast<int32_t*>(c_cyc);
    dio.cart_state = io->cart_state ? (double*)ctx->arena.take(sizeof(double) * B * 3) : nullptr;
    if (dio.cart_state) HIP_TRY(hipMemsetAsync(dio.cart_state, 0xFF, sizeof(double) * B * 3, ctx->stream));  // NaN for egos that do not move
    hipError_t e = fp::launch_advance(ka, c_idx, c_es, dio, ctx->stream);
    if (e != hipSuccess) return fail(FP_EHIP, "advance kernel launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(io->ego, dio.ego, sizeof(double) * B * 6, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(io->t_now, dio.t_now, sizeof(int32_t) * B, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(io->done, dio.done, sizeof(int32_t) * B, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(io->cycles, dio.cycles, sizeof(int32_t) * B, hipMemcpyDeviceToHost, ctx->stream));
    if (io->cart_state) HIP_TRY(hipMemcpyAsync(io->cart_state, dio.cart_state, sizeof(double) * B * 3, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FP_OK;
}

int fp_frames_build(fp_ctx* ctx, int32_t F, int32_t NX, const int32_t* n, const double* points, double* knots, double* coef, int mem,
                    void* stream)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    if (F < 0 || NX < 2 || NX > FP_MAX_KNOTS) return fail(FP_EINVAL, "bad sizes F=%d NX=%d", F, NX);
    if (!n || !points || !knots || !coef) return fail(FP_EINVAL, "NULL array");
    if (F == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (mem == FP_MEM_DEVICE) {
        hipError_t e = fp::launch_frames_build(F, NX, n, points, knots, coef, (hipStream_t)stream);
        if (e != hipSuccess) return fail(FP_EHIP, "frame build launch failed: %s", hipGetErrorString(e));
        return FP_OK;
    }
    if (mem != FP_MEM_HOST) return fail(FP_EINVAL, "mem must be FP_MEM_HOST or FP_MEM_DEVICE");
    for (int f = 0; f < F; ++f)
        if (n[f] < 2 || n[f] > NX) return fail(FP_EINVAL, "n[%d]=%d out of range", f, n[f]);
    const size_t fn = (size_t)F * NX;
    int rc;
    if ((rc = ctx->arena.reserve(Arena::padded(sizeof(int32_t) * F) + Arena::padded(sizeof(double) * fn * 2) + Arena::padded(sizeof(double) * fn) +
                                 Arena::padded(sizeof(double) * fn * 8))) != FP_OK)
        return rc;
    ctx->arena.reset();
    const int32_t* d_n = nullptr; const double* d_pts = nullptr;
    if ((rc = push(ctx, n, (size_t)F, &d_n)) != FP_OK) return rc;
    if ((rc = push(ctx, points, fn * 2, &d_pts)) != FP_OK) return rc;
    double* d_k = (double*)ctx->arena.take(sizeof(double) * fn);
    double* d_c = (double*)ctx->arena.take(sizeof(double) * fn * 8);
    hipError_t e = fp::launch_frames_build(F, NX, d_n, d_pts, d_k, d_c, ctx->stream);
    if (e != hipSuccess) return fail(FP_EHIP, "frame build launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(knots, d_k, sizeof(double) * fn, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(coef, d_c, sizeof(double) * fn * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FP_OK;
}

int fp_from_state(fp_ctx* ctx, const fp_batch* batch, const double* states, double* ego, int mem, void* stream)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    if (!batch || !states || !ego) return fail(FP_EINVAL, "NULL argument");
    if (batch->B < 0 || batch->F < 1 || batch->NX < 2 || batch->NX > FP_MAX_KNOTS) return fail(FP_EINVAL, "bad batch sizes");
    if (!batch->frame_of || !batch->nx || !batch->knots || !batch->coef) return fail(FP_EINVAL, "batch frame arrays must not be NULL");
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (mem == FP_MEM_DEVICE) {
        hipError_t e = fp::launch_from_state(*batch, states, ego, (hipStream_t)stream);
        if (e != hipSuccess) return fail(FP_EHIP, "from_state launch failed: %s", hipGetErrorString(e));
        return FP_OK;
    }
    if (mem != FP_MEM_HOST) return fail(FP_EINVAL, "mem must be FP_MEM_HOST or FP_MEM_DEVICE");
    const size_t B = (size_t)batch->B, fn = (size_t)batch->F * batch->NX;
    for (size_t i = 0; i < B; ++i)
        if (batch->frame_of[i] < 0 || batch->frame_of[i] >= batch->F) return fail(FP_EINVAL, "frame_of[%zu] out of range", i);
    int rc;
    if ((rc = ctx->arena.reserve(Arena::padded(sizeof(int32_t) * B) + Arena::padded(sizeof(int32_t) * batch->F) + Arena::padded(sizeof(double) * fn) +
                                 Arena::padded(sizeof(double) * fn * 8) + Arena::padded(sizeof(double) * B * 4) + Arena::padded(sizeof(double) * B * 6))) != FP_OK)
        return rc;
    ctx->arena.reset();
    fp_batch db = *batch;
    if ((rc = push(ctx, batch->frame_of, B, &db.frame_of)) != FP_OK) return rc;
    if ((rc = push(ctx, batch->nx, (size_t)batch->F, &db.nx)) != FP_OK) return rc;
    if ((rc = push(ctx, batch->knots, fn, &db.knots)) != FP_OK) return rc;
    if ((rc = push(ctx, batch->coef, fn * 8, &db.coef)) != FP_OK) return rc;
    const double* d_states = nullptr;
    if ((rc = push(ctx, states, B * 4, &d_states)) != FP_OK) return rc;
    double* d_ego = (double*)ctx->arena.take(sizeof(double) * B * 6);
    hipError_t e = fp::launch_from_state(db, d_states, d_ego, ctx->stream);
    if (e != hipSuccess) return fail(FP_EHIP, "from_state launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(ego, d_ego, sizeof(double) * B * 6, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FP_OK;
}

int fp_eval_trajs(fp_ctx* ctx, const fp_params* params, const fp_batch* batch, int32_t K, const double* end_states, double* cost,
                  uint32_t* flags, double* traj, int32_t stride, int mem, void* stream)
{
    if (!ctx) return fail(FP_EINVAL, "ctx is NULL");
    int rc;
    if ((rc = check_params(params)) != FP_OK) return rc;
    if ((rc = check_batch(batch)) != FP_OK) return rc;
    if (K < 1 || !end_states) return fail(FP_EINVAL, "K must be >= 1 and end_states non-NULL");
    if (traj && stride < FP_MAX_POINTS) return fail(FP_EINVAL, "traj stride must be >= FP_MAX_POINTS (%d)", FP_MAX_POINTS);
    if (batch->B == 0) return FP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t BK = (size_t)batch->B * K;
    fp::KernelArgs ka;
    ka.p = *params;
    ka.r = fp_result{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (mem == FP_MEM_DEVICE) {
        ka.b = *batch;
        if (!(batch->S > 0 && batch->n_obs > 0)) ka.b.n_obs = 0;
        hipError_t e = fp::launch_eval_trajs(ka, K, end_states, cost, flags, traj, stride, (hipStream_t)stream);
        if (e != hipSuccess) return fail(FP_EHIP, "eval kernel launch failed: %s", hipGetErrorString(e));
        return FP_OK;
    }
    if (mem != FP_MEM_HOST) return fail(FP_EINVAL, "mem must be FP_MEM_HOST or FP_MEM_DEVICE");
    if ((rc = check_batch_host(params, batch)) != FP_OK) return rc;
    for (size_t i = 0; i < BK; ++i) {
        const double n = end_states[3 * i + 2] / params->tick_t;
        if (n != n) continue;  // NaN end state = "no trajectory": the kernels emit NaN cost / all-NaN series
        if (!(n > 0) || n > FP_MAX_POINTS) return fail(FP_ELIMIT, "end_states[%zu].T=%g needs 1..FP_MAX_POINTS points", i, end_states[3 * i + 2]);
    }
    const size_t traj_doubles = traj ? BK * FP_ARR_COUNT * (size_t)stride : 0;
    size_t need = batch_bytes(params, batch) + Arena::padded(sizeof(double) * BK * 3) + Arena::padded(sizeof(double) * BK) +
                  Arena::padded(sizeof(uint32_t) * BK) + Arena::padded(sizeof(double) * traj_doubles);
    if ((rc = ctx->arena.reserve(need)) != FP_OK) return rc;
    ctx->arena.reset();
    if ((rc = stage_batch(ctx, params, batch, &ka.b)) != FP_OK) return rc;
    const double* d_end = nullptr;
    if ((rc = push(ctx, end_states, BK * 3, &d_end)) != FP_OK) return rc;
    double* d_cost = (double*)ctx->arena.take(sizeof(double) * BK);
    uint32_t* d_flags = (uint32_t*)ctx->arena.take(sizeof(uint32_t) * BK);
    double* d_traj = traj ? (double*)ctx->arena.take(sizeof(double) * traj_doubles) : nullptr;
    hipError_t e = fp::launch_eval_trajs(ka, K, d_end, d_cost, d_flags, d_traj, stride, ctx->stream);
    if (e != hipSuccess) return fail(FP_EHIP, "eval kernel launch failed: %s", hipGetErrorString(e));
    if (cost) HIP_TRY(hipMemcpyAsync(cost, d_cost, sizeof(double) * BK, hipMemcpyDeviceToHost, ctx->stream));
    if (flags) HIP_TRY(hipMemcpyAsync(flags, d_flags, sizeof(uint32_t) * BK, hipMemcpyDeviceToHost, ctx->stream));
    if (traj) HIP_TRY(hipMemcpyAsync(traj, d_traj, sizeof(double) * traj_doubles, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FP_OK;
}

}  // extern "C"
