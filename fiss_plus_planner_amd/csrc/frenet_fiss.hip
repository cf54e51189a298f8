// frenet_fiss.hip - FISS / FISS+ coarse search for a whole batch: one wavefront per ego walks the dense tables.
//
// The reference generates candidates lazily while it moves over the (d, v, t) index grid.  A candidate's cost_final,
// its constraint / collision outcome and its cost_est are pure functions of its index, so the walk
// can run over tables the lattice kernel already produced:  J = cost_final,  F = flag word,  E = cost_est.
// The walk itself is sequential and data dependent; it is kept wave-uniform (every lane follows the same control
// flow, scalar state lives in uniform registers) and only its two search primitives use the 64 lanes:
//     queue head    = argmin of J over "in queue" entries          (fiss_planner.py:207 / :229, heapq order)
//     initial guess = argmin of E over not-yet-generated entries, LAST minimum  (fiss_planner.py:140-150)
//     frontier pop  = argmin of J over frontier entries            (fiss_plus_planner.py:113)
// each a strided scan over an LDS key array (+inf = absent) + a DPP wave minimum + ballot for the owner.
//
// Restated: fiss_planner.py:33-99 (cost_est), :101-138 (generate_trajectory -> table lookup), :140-188, :190-270;
//           fiss_plus_planner.py:30-59, :80-148.
#include "frenet_device.h"
#include "frenet_kernels.h"

namespace fp {

namespace {

constexpr uint8_t kGen = 1, kInQ = 2;

struct Walk {
    const double* J;
    const double* E;
    const uint8_t* F;
    uint8_t* st;
    double* keyQ;   // J where "in queue", +inf elsewhere      -> queue head = argmin
    double* keyF;   // J where "on the frontier", +inf elsewhere
    double* keyG;   // E where not yet generated, +inf elsewhere -> initial guess = LAST argmin
    int nd, nv, nt, C, lane;
    int num_iter, num_generated, num_validated, num_checks;

    __device__ __forceinline__ int raster(int i, int j, int k) const { return (i * nv + j) * nt + k; }

    // generate_trajectory (fiss_planner.py:101-138): a table lookup + bookkeeping.  Wave-uniform.
    __device__ __forceinline__ bool generate(int q, double& cost)
    {
        cost = J[q];
        const uint8_t s = st[q];
        if (s & kGen) return false;
        st[q] = s | kGen | kInQ;  // candidate_trajs.put((cost_final, idx))
        keyQ[q] = cost;
        keyG[q] = __builtin_inf();
        ++num_generated;
        return true;
    }

    // argmin over a key array (+inf = absent).  The scan issues all of a lane's loads before comparing (independent LDS
    // reads), the wave minimum uses DPP, the owner is found with a ballot.  Exact ties: lowest raster index
    // (PREFER_HIGH = false; documented divergence from the reference's ValueError) or highest (find_initial_guess).
    template <bool PREFER_HIGH>
    __device__ __forceinline__ int argmin_key(const double* key) const
    {
        double best = __builtin_inf();
        int bq = -1;
#pragma unroll 4
        for (int q = lane; q < C; q += kWave) {
            const double v = key[q];
            if (PREFER_HIGH ? (v <= best && v < __builtin_inf()) : (v < best)) { best = v; bq = q; }
        }
        const double m = wave_min_f64(best);
        if (!(m < __builtin_inf())) return -1;
        const unsigned long long owners = __ballot(best == m && bq >= 0);
        if (__popcll(owners) == 1) return __builtin_amdgcn_readlane(bq, __ffsll((long long)owners) - 1);
        // exact tie across lanes: integer min / max of the candidates' raster indices
        int cand = (best == m && bq >= 0) ? bq : (PREFER_HIGH ? -1 : 0x7fffffff);
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const int o = __shfl_xor(cand, off, kWave);
            cand = PREFER_HIGH ? (o > cand ? o : cand) : (o < cand ? o : cand);
        }
        return cand;
    }

    __device__ __forceinline__ int head_queue() const { return argmin_key<false>(keyQ); }
    __device__ __forceinline__ int head_frontier() const { return argmin_key<false>(keyF); }
    // find_initial_guess (fiss_planner.py:140-150): `cost_est <= min_cost` keeps the LAST minimum
    __device__ __forceinline__ int initial_guess() const { return argmin_key<true>(keyG); }

    // validation of the queue head (fiss_planner.py:229-258): returns 1 = answer, 0 = rejected
    __device__ __forceinline__ int validate(int q)
    {
        st[q] &= (uint8_t)~kInQ;
        keyQ[q] = __builtin_inf();
        ++num_validated;
        const uint8_t f = F[q];
        if (f & (FP_FLAG_SPEED | FP_FLAG_ACCEL)) return 0;
        ++num_checks;
        return (f & FP_FLAG_COLLISION) ? 0 : 1;
    }
};

}  // namespace

__global__ __launch_bounds__(kWave) void fiss_search_kernel(FissArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const fp_params& p = fa.ka.p;
    const fp_batch& bt = fa.ka.b;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nd = p.nd, nv = p.nv, nt = p.nt, C = nd * nv * nt;
    if (bt.skip && bt.skip[b]) {  // finished ego of a closed-loop batch
        if (lane == 0) {
            int32_t* out = fa.io.best_ijk + (size_t)b * 3;
            out[0] = out[1] = out[2] = -1;
            fa.io.best_cost[b] = __builtin_nan("");
            double* es = fa.io.end_state + (size_t)b * 3;
            es[0] = es[1] = es[2] = __builtin_nan("");
            fa.io.refined[b] = 0;
            int32_t* s4 = fa.io.stats + (size_t)b * 4;
            s4[0] = s4[1] = s4[2] = s4[3] = 0;
        }
        return;
    }
    double* J = (double*)smem;
    double* E = J + C;
    double* keyQ = E + C;
    double* keyF = keyQ + C;
    double* keyG = keyF + C;
    uint8_t* F = (uint8_t*)(keyG + C);
    uint8_t* st = F + C;

    // ---- tables: FOP flat order (i_d, i_T, i_v) -> FISS raster (i_d, i_v, i_t); cost_est (fiss_planner.py:33-99)
    const double* smin = fa.io.samp_min + (size_t)b * 3;
    const double* smax = fa.io.samp_max + (size_t)b * 3;
    const int* prev = fa.io.prev_best_idx + (size_t)b * 3;
    const int p0 = prev[0], p1 = prev[1], p2 = prev[2];
    const double lat_norm = fmax(smin[0] * smin[0], smax[0] * smax[0]);
    const double vr = smax[1] - smin[1], tr = smax[2] - smin[2];
    const double max_sqr_dist = (double)(nd * nd + nv * nv + nt * nt);
    const double* vs = bt.v_samples + (size_t)b * nv;
    for (int q = lane; q < C; q += kWave) {
        const int k = q % nt, j = (q / nt) % nv, i = q / (nt * nv);
        const size_t flat = (size_t)b * C + (size_t)(i * nt + k) * nv + j;
        J[q] = fa.cost_tbl[flat];
        F[q] = (uint8_t)(fa.flag_tbl[flat] & 0xFFu);
        st[q] = 0;
        const double d = bt.d_samples[i], v = vs[j], t = bt.t_samples[k];
        const double ev = smax[1] - v;
        const double est_lat = (d * d) / lat_norm;
        const double est_speed = (ev * ev) / (vr * vr);
        const double est_time = 1.0 - (t - smin[2]) / tr;
        double est = est_lat + est_time + est_speed;
        if (p0 >= 0) {
            const int a = i - p0, bb = j - p1, c = k - p2;
            est += fa.opts.w_heuristic * (double)(a * a + bb * bb + c * c) / max_sqr_dist;
        }
        E[q] = est;
        keyQ[q] = __builtin_inf();
        keyF[q] = __builtin_inf();
        keyG[q] = est <= __builtin_inf() ? est : __builtin_inf();  // a NaN estimate can never satisfy `<=`
    }
    __syncthreads();

    // No feasible candidate anywhere in the lattice: the walk would generate and validate every sample, one per outer
    // iteration, and give up (fiss_planner.py:203-206).  Its outcome is closed form: num_iter = C + 1, generated =
    // validated = C, collision checks = samples that pass the constraints.  (Each iteration pops exactly one candidate.)
    {
        int feasible = 0, pass_constraints = 0;
        for (int q = lane; q < C; q += kWave) {
            feasible += (F[q] & FP_FLAG_INFEASIBLE) == 0;
            pass_constraints += (F[q] & (FP_FLAG_SPEED | FP_FLAG_ACCEL)) == 0;
        }
        if (__ballot(feasible != 0) == 0ull) {
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) pass_constraints += __shfl_xor(pass_constraints, off, kWave);
            if (lane == 0) {
                int32_t* out = fa.io.best_ijk + (size_t)b * 3;
                out[0] = out[1] = out[2] = -1;
                fa.io.best_cost[b] = __builtin_nan("");
                double* es = fa.io.end_state + (size_t)b * 3;
                es[0] = es[1] = es[2] = __builtin_nan("");
                fa.io.refined[b] = 0;
                int32_t* s4 = fa.io.stats + (size_t)b * 4;
                s4[0] = C + 1; s4[1] = C; s4[2] = C; s4[3] = pass_constraints;
            }
            return;
        }
    }

    Walk w{J, E, F, st, keyQ, keyF, keyG, nd, nv, nt, C, lane, 0, 0, 0, 0};
    const int sizes[3] = {nd, nv, nt};
    int best = -1;
    const bool plus = fa.opts.kind == FP_FISS_PLUS;
    for (;;) {
        ++w.num_iter;
        int q = w.head_queue();
        const int generated_before = w.num_generated;
        const int head_before = q;
        if (q < 0) {
            q = w.initial_guess();
            if (q < 0) break;  // every sample searched, nothing feasible (:203-206)
        }
        int idx[3] = {q / (nt * nv), (q / nt) % nv, q % nt};
        double cost_center, cost;
        if (!plus) {
            // explore_next_sample (:174-188) until it lands on a generated sample
            while (!(st[w.raster(idx[0], idx[1], idx[2])] & kGen)) {
                w.generate(w.raster(idx[0], idx[1], idx[2]), cost_center);  // find_gradients (:152-172)
                double grad[3];
#pragma unroll
                for (int dim = 0; dim < 3; ++dim) {
                    int nb[3] = {idx[0], idx[1], idx[2]};
                    if (idx[dim] < sizes[dim] - 1) {
                        nb[dim] += 1;
                        w.generate(w.raster(nb[0], nb[1], nb[2]), cost);
                        grad[dim] = cost - cost_center;
                        if (grad[dim] >= 0 && idx[dim] == 0) grad[dim] = 0.0;
                    } else {
                        nb[dim] -= 1;
                        if (nb[dim] < 0) nb[dim] = sizes[dim] - 1;  // python negative index on a size-1 axis
                        w.generate(w.raster(nb[0], nb[1], nb[2]), cost);
                        grad[dim] = cost_center - cost;
                        if (grad[dim] <= 0 && idx[dim] == sizes[dim] - 1) grad[dim] = 0.0;
                    }
                }
#pragma unroll
                for (int dim = 0; dim < 3; ++dim) {
                    idx[dim] += grad[dim] > 0.0 ? -1 : +1;
                    idx[dim] = idx[dim] < 0 ? 0 : (idx[dim] > sizes[dim] - 1 ? sizes[dim] - 1 : idx[dim]);
                }
            }
        } else {
            // explore_neighbors + frontier (fiss_plus_planner.py:30-59, :106-116)
            for (;;) {
                w.generate(w.raster(idx[0], idx[1], idx[2]), cost_center);
#pragma unroll
                for (int dim = 0; dim < 3; ++dim) {
#pragma unroll
                    for (int step = -1; step <= 1; step += 2) {
                        const int n = idx[dim] + step;
                        if (n < 0 || n > sizes[dim] - 1) continue;
                        int nb[3] = {idx[0], idx[1], idx[2]};
                        nb[dim] = n;
                        const int nq = w.raster(nb[0], nb[1], nb[2]);
                        if (w.generate(nq, cost) && cost <= cost_center) keyF[nq] = cost;  // frontier_idxs.put((cost, idx))
                    }
                }
                const int nq = w.head_frontier();
                if (nq < 0) break;
                keyF[nq] = __builtin_inf();
                idx[0] = nq / (nt * nv); idx[1] = (nq / nt) % nv; idx[2] = nq % nt;
            }
        }
        // the queue head only changes when the exploration generated something
        q = (head_before >= 0 && w.num_generated == generated_before) ? head_before : w.head_queue();
        if (q < 0) break;
        if (w.validate(q)) { best = q; break; }
    }
    if (lane == 0) {
        int32_t* out = fa.io.best_ijk + (size_t)b * 3;
        int32_t* pv = fa.io.prev_best_idx + (size_t)b * 3;
        if (best >= 0) {
            const int i = best / (nt * nv), j = (best / nt) % nv, k = best % nt;
            out[0] = i; out[1] = j; out[2] = k;
            pv[0] = i; pv[1] = j; pv[2] = k;  // prev_best_idx persists across cycles (:252 / :140)
            fa.io.best_cost[b] = J[best];
            double* es = fa.io.end_state + (size_t)b * 3;
            es[0] = bt.d_samples[i]; es[1] = vs[j]; es[2] = bt.t_samples[k];
        } else {
            out[0] = out[1] = out[2] = -1;
            fa.io.best_cost[b] = __builtin_nan("");
            double* es = fa.io.end_state + (size_t)b * 3;
            es[0] = es[1] = es[2] = __builtin_nan("");
        }
        fa.io.refined[b] = 0;
        int32_t* s4 = fa.io.stats + (size_t)b * 4;
        s4[0] = w.num_iter; s4[1] = w.num_generated; s4[2] = w.num_validated; s4[3] = w.num_checks;
    }
}

hipError_t launch_fiss_search(const FissArgs& fa, hipStream_t stream)
{
    const int C = fa.ka.p.nd * fa.ka.p.nv * fa.ka.p.nt;
    const int bytes = C * (5 * 8 + 1 + 1) + 16;
    hipLaunchKernelGGL(fiss_search_kernel, dim3(fa.ka.b.B), dim3(kWave), bytes, stream, fa);
    return hipGetLastError();
}

}  // namespace fp
