// frenet_fissplus.h - FISS+ coarse search (fiss_plus_planner.py:30-59, :80-148) for a whole batch, in RANK space.
//
// A candidate's cost_final, its constraint / collision outcome and its cost_est are pure functions of its index,
// so the walk runs over the tables the lattice kernel produced.  The walk is sequential and data dependent - a blocked scene pops
// every candidate of the lattice one after the other and the LONGEST walk of the batch is the kernel's duration - so everything
// here is about the dependent chain of ONE outer iteration.
//
// Both priority queues of the reference order by (cost_final, raster index): a total order that is fixed before the walk starts.
// The prologue ranks the lattice once by that key; from then on nothing is addressed by lattice index any more:
//   * "generated", "in the candidate queue" and "on the frontier" are BIT SETS over ranks held in registers - lane L owns ranks
//     32 L .. 32 L + 31 (a second word per lane above 2048 candidates);
//   * one 16-byte LDS record per rank carries the ranks of the six axis neighbours, the frontier bound and the flag bits, so
//     explore_neighbors (fiss_plus_planner.py:30-59) is ONE broadcast LDS read followed by lane-local bit arithmetic:
//         new      = neighbours & ~generated                 (is_new, fiss_planner.py:104-105)
//         queue   |= new & finite                            (candidate_trajs.put, fiss_planner.py:136)
//         frontier|= new & finite & (rank < bound)           (`is_new and cost <= cost_center`, fiss_plus_planner.py:44-45, :53-54:
//                                                             cost <= centre cost <=> rank below the end of the centre's tie run)
//   * exploring a sample whose neighbours are all generated is the identity, so the head of the queue is explored without asking
//     whether it was explored before; when that adds nothing to the frontier the head itself is the next candidate to validate
//     (anything cheaper that appeared would have been put on the frontier), and the iteration needs no second queue search;
//   * Stats: generated = population count of the "new" words, accumulated per lane and summed once at the end.
// Exact ties resolve to the LOWEST raster index (documented divergence: the reference raises ValueError on tied heap entries);
// non-finite costs are generated but never queued.
//
// The ranking is a bucket sort: 256 buckets linear in the cost (monotone in the key, so buckets are rank ranges), LDS atomics for
// the histogram and the scatter, then every candidate counts the smaller keys inside its own bucket (a handful).  The comparison
// network this replaces cost ~55 barrier-separated LDS passes.
// The walk of ONE ego is a device function (fissplus_search_ego): fissplus_search_kernel (frenet_fissplus.hip) runs it one workgroup per
// ego behind the lattice launch; lattice_fused_kernel's FISS instances run it in workgroups APPENDED to the lattice grid - they become
// resident in the slots the draining launch leaves empty and wait for their ego's dense tables (frenet_lattice_fused.hip).
#pragma once
#include "frenet_device.h"
#include "frenet_kernels.h"

namespace fp {
namespace fsp {

// sort payload: raster index << 3 | bits (ordering by payload = ordering by raster index)
constexpr uint32_t kPayCfail = 1u;  // fails check_constraints
constexpr uint32_t kPayColl = 2u;   // collides
constexpr uint32_t kPayNan = 4u;    // cost_final is NaN (a NaN centre puts nothing on the frontier)

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lowest set rank of a bit set over ranks (word k of lane L holds ranks 2048 k + 32 L .. + 31); -1 when empty.  Wave-uniform.
template <int NW>
__device__ __forceinline__ int lowest_rank(const uint32_t (&w)[NW])
{
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const unsigned long long nz = __ballot(w[k] != 0u);
        if (nz) {
            const int L = __ffsll((long long)nz) - 1;
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)w[k], L);
            return (k << 11) + (L << 5) + __ffs((int)v) - 1;
        }
    }
    return -1;
}

__device__ __forceinline__ double uniform_f64(double v)  // v is the same in every lane: tell the compiler
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int NW>
__device__ __forceinline__ void reset_rank(uint32_t (&w)[NW], int r, int lane)  // r wave-uniform
{
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] &= ~(((r >> 5) == lane + 64 * k) ? (1u << (r & 31)) : 0u);
}

template <int NW>
__device__ __forceinline__ void set_rank(uint32_t (&w)[NW], int r, int lane)
{
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] |= ((r >> 5) == lane + 64 * k) ? (1u << (r & 31)) : 0u;
}

template <int NW>
__device__ __forceinline__ bool any_rank(const uint32_t (&w)[NW])
{
    uint32_t a = w[0];
#pragma unroll
    for (int k = 1; k < NW; ++k) a |= w[k];
    return __ballot(a != 0u) != 0ull;
}

// explore_neighbors (fiss_plus_planner.py:30-59) of the sample whose record is `rec` (the same 16 bytes in every lane).
template <int NW>
__device__ __forceinline__ void explore(const uint4& rec, int lane, uint32_t (&G)[NW], uint32_t (&Q)[NW], uint32_t (&Fr)[NW],
                                        const uint32_t (&fin)[NW], int& ngen)
{
    const uint32_t n[6] = {rec.x & 0xFFFFu, rec.x >> 16, rec.y & 0xFFFFu, rec.y >> 16, rec.z & 0xFFFFu, rec.z >> 16};
    const int lim1 = (int)(rec.w & 0xFFFFu);  // ranks below lim1 cost no more than the centre
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const uint32_t me = (uint32_t)(lane + 64 * k);
        uint32_t m = 0u;
#pragma unroll
        for (int e = 0; e < 6; ++e) m |= ((n[e] >> 5) == me) ? (1u << (n[e] & 31u)) : 0u;  // 0xFFFF (no neighbour) matches no lane
        const uint32_t gnew = m & ~G[k];
        G[k] |= m;
        ngen += __popc(gnew);
        const uint32_t qnew = gnew & fin[k];
        Q[k] |= qnew;
        int rel = lim1 - (int)(me << 5);
        rel = rel < 0 ? 0 : (rel > 32 ? 32 : rel);
        const uint32_t below = rel == 0 ? 0u : (0xFFFFFFFFu >> (32 - rel));
        Fr[k] |= qnew & below;
    }
}

// W wavefronts build the tables and rank the lattice (W = 1: small lattices, wave-level synchronisation only); the walk itself is
// one wavefront's - the others leave before it starts.  NW = 32-bit words of rank bits per lane (C <= 2048 NW).  NB = buckets
// (a multiple of 64 W).
constexpr int kMisc = 256;  // bytes of counters / per-wave extremes / jump state in front of the tables
// CMAX: the most samples the instance is launched for (sizes the per-thread register copies of the jump's relaxation).
// COHERENT: the dense tables were written by ANOTHER workgroup of the SAME launch (the fused lattice + search launch): they are read
// with agent-scope loads (the writer used agent-scope stores; an XCD's L2 is not coherent with the others' inside a launch).
template <int W, int NW, int CMAX, bool COHERENT = false>
__device__ __forceinline__ void fissplus_search_ego(const FissArgs& fa, int NB, int b, unsigned char* smem)
{
    constexpr int T = W * kWave;
    const fp_params& p = fa.ka.p;
    const fp_batch& bt = fa.ka.b;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    // Timing diagnostic (tools/refine_stamps.py, -DFP_PHASE_STAMPS): thread 0 leaves 10 ns ticks since the workgroup started in
    // columns 96..111 of the last row of the ego's series block (sparse layout, stride 128, N <= 96)
#if defined(FP_PHASE_STAMPS)
    const long long t_begin = wall_clock64();
#define FP_SSTAMP(k) do { if (threadIdx.x == 0 && fa.io.best_traj) fa.io.best_traj[((size_t)b * FP_ARR_COUNT + 15) * fa.io.traj_stride + 96 + (k)] = (double)(wall_clock64() - t_begin); } while (0)
#else
#define FP_SSTAMP(k) do { } while (0)
#endif
    const int nd = p.nd, nv = p.nv, nt = p.nt, C = nd * nv * nt;
    auto group_sync = [&]() { if constexpr (W == 1) wave_lds_sync(); else __syncthreads(); };
    auto write_none = [&](int s0, int s1, int s2, int s3) {
        int32_t* out = fa.io.best_ijk + (size_t)b * 3;
        out[0] = out[1] = out[2] = -1;
        fa.io.best_cost[b] = __builtin_nan("");
        double* es = fa.io.end_state + (size_t)b * 3;
        es[0] = es[1] = es[2] = __builtin_nan("");
        fa.io.refined[b] = 0;
        int32_t* s4 = fa.io.stats + (size_t)b * 4;
        s4[0] = s0; s4[1] = s1; s4[2] = s2; s4[3] = s3;
    };
    if (bt.skip && bt.skip[b]) {  // finished ego of a closed-loop batch
        if (tid == 0) write_none(0, 0, 0, 0);
        return;
    }
    // LDS: misc [128 B] | X: double [C32] (keys by raster index, later cost_est by rank, one pad per 32) | Y: 16 C bytes (bucket-ordered
    // keys + payloads, later the rank records) | order, rank, lim u16 [C] each | hist int [NB + 2] | cursor int [NB + 2]
    const int C32 = C + (C >> 5) + 1;
    const int C8 = (C + 7) & ~7;
    int* s_cnt = (int*)smem;                    // [0] feasible, [1] pass constraints, [4..4 + W) wave totals of the scan (W <= 8)
    double* s_wmin = (double*)(smem + 64);      // [W]
    double* s_wmax = (double*)(smem + 128);     // [W]
    double* X = (double*)(smem + kMisc);
    unsigned char* Y = (unsigned char*)(X + ((C32 + 1) & ~1));
    double* bkey = (double*)Y;
    uint16_t* bq = (uint16_t*)(Y + (size_t)8 * C8);
    uint4* REC = (uint4*)Y;
    uint16_t* order = (uint16_t*)(Y + (size_t)16 * C8);
    uint16_t* rank = order + C8;
    uint16_t* limtab = rank + C8;
    int* hist = (int*)(limtab + C8);
    int* cursor = hist + NB + 2;

    if (tid < 16) s_cnt[tid] = 0;
    for (int i = tid; i < NB + 2; i += T) hist[i] = 0;
    group_sync();

    // ---- T1: tables, FOP flat order (i_d, i_T, i_v) -> FISS raster (i_d, i_v, i_t)
    const int nvt = nv * nt;
    int feasible = 0, pass_constraints = 0;
    double kmin = __builtin_inf(), kmax = -__builtin_inf();
    for (int q = tid; q < C; q += T) {
        const int i = q / nvt, rem = q - i * nvt, j = rem / nt, k = rem - j * nt;
        const size_t flat = (size_t)b * C + (size_t)(i * nt + k) * nv + j;
        double cost;
        uint32_t f;
        if constexpr (COHERENT) {
            cost = __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)&fa.cost_tbl[flat], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            f = __hip_atomic_load(&fa.flag_tbl[flat], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            cost = fa.cost_tbl[flat];
            f = fa.flag_tbl[flat];
        }
        const bool finite = cost < __builtin_inf() && cost > -__builtin_inf();
        X[q] = finite ? cost : __builtin_inf();  // NaN and +-inf: never queued, ranked behind every finite key by raster index
        if (finite) { kmin = fmin(kmin, cost); kmax = fmax(kmax, cost); }
        feasible += (f & FP_FLAG_INFEASIBLE) == 0;
        pass_constraints += (f & FP_FLAG_CONSTRAINTS) == 0;
        rank[q] = (uint16_t)(((uint32_t)q << 3) | ((f & FP_FLAG_CONSTRAINTS) ? kPayCfail : 0u) | ((f & FP_FLAG_COLLISION) ? kPayColl : 0u) |
                             (cost != cost ? kPayNan : 0u));  // (the payload, parked here until the scatter)
    }
    // No feasible candidate anywhere in the lattice: the walk would generate and validate every sample, one per outer
    // iteration, and give up (fiss_planner.py:203-206).  Its outcome is closed form: num_iter = C + 1, generated =
    // validated = C, collision checks = samples that pass the constraints.  (Each iteration pops exactly one candidate.)
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        pass_constraints += __shfl_xor(pass_constraints, off, kWave);
        feasible += __shfl_xor(feasible, off, kWave);
    }
    kmin = wave_min_f64(kmin);
    kmax = -wave_min_f64(-kmax);
    if (W > 1) {
        if (lane == 0) {
            atomicAdd(&s_cnt[0], feasible);
            atomicAdd(&s_cnt[1], pass_constraints);
            s_wmin[wave] = kmin;
            s_wmax[wave] = kmax;
        }
        __syncthreads();
        feasible = s_cnt[0];
        pass_constraints = s_cnt[1];
#pragma unroll
        for (int w = 0; w < W; ++w) { kmin = fmin(kmin, s_wmin[w]); kmax = fmax(kmax, s_wmax[w]); }
    } else {
        wave_lds_sync();
    }
    if (feasible == 0) {
        if (tid == 0) write_none(C + 1, C, C, pass_constraints);
        return;
    }

    FP_SSTAMP(1);
    // ---- T2: histogram over NB buckets linear in the key (+ bucket NB: the non-finite keys)
    const double scale = kmax > kmin ? (double)NB / (kmax - kmin) : 0.0;
    auto bucket_of = [&](double key) -> int {
        if (!(key < __builtin_inf())) return NB;
        const int v = (int)((key - kmin) * scale);  // monotone in the key: subtraction, product and truncation all are
        return v < NB - 1 ? v : NB - 1;
    };
    for (int q = tid; q < C; q += T) atomicAdd(&hist[bucket_of(X[q])], 1);
    group_sync();
    // ---- T3: exclusive scan of the histogram -> bucket starts (kept in hist) and scatter cursors
    {
        const int BPT = NB / T;  // buckets per thread (NB is a multiple of T)
        int local = 0;
        for (int u = 0; u < BPT; ++u) local += hist[tid * BPT + u];
        int incl = local;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        if (W > 1) {
            if (lane == kWave - 1) s_cnt[4 + wave] = incl;
            __syncthreads();
            for (int w = 0; w < wave; ++w) incl += s_cnt[4 + w];
        }
        int run = incl - local;
        for (int u = 0; u < BPT; ++u) {
            const int h = hist[tid * BPT + u];
            hist[tid * BPT + u] = run;
            cursor[tid * BPT + u] = run;
            run += h;
        }
        if (tid == T - 1) {  // run = number of finite keys
            hist[NB] = run;
            cursor[NB] = run;
            hist[NB + 1] = C;
        }
    }
    group_sync();
    FP_SSTAMP(3);
    // ---- T4: scatter into bucket order (the payloads were parked in the rank array, which T5 fills)
    for (int q = tid; q < C; q += T) {
        const double key = X[q];
        const int slot = atomicAdd(&cursor[bucket_of(key)], 1);
        bkey[slot] = key;
        bq[slot] = rank[q];
    }
    group_sync();
    FP_SSTAMP(4);
    // ---- T5: rank inside the bucket by counting; length of the tie run behind the candidate
    const int nfin = hist[NB];
    for (int q = tid; q < C; q += T) {
        const double key = X[q];
        const int bk = bucket_of(key);
        const int s0 = hist[bk], s1 = hist[bk + 1];
        int less = 0, eq_after = 0;
        uint32_t mine = 0;
        for (int s = s0; s < s1; ++s) {
            const double ok = bkey[s];
            const uint32_t op = bq[s];
            const bool same = (op >> 3) == (uint32_t)q;
            mine = same ? op : mine;
            less += (ok < key) || (ok == key && (op >> 3) < (uint32_t)q);
            eq_after += (ok == key) && (op >> 3) > (uint32_t)q;
        }
        const int r = s0 + less;
        order[r] = (uint16_t)mine;
        rank[q] = (uint16_t)r;
        // frontier bound of this candidate as a centre: ranks < lim1 have cost <= its cost.  A NaN centre: none; an infinite
        // centre: every finite one (only finite candidates are ever queued)
        limtab[r] = (uint16_t)(r < nfin ? r + eq_after + 1 : ((mine & kPayNan) ? 0 : nfin));
    }
    group_sync();
    FP_SSTAMP(5);
    // ---- T6: per rank: cost_est (fiss_planner.py:33-99) and the record {six neighbour ranks, bound, payload bits}
    {
        const double* smin = fa.io.samp_min + (size_t)b * 3;
        const double* smax = fa.io.samp_max + (size_t)b * 3;
        const int* prev = fa.io.prev_best_idx + (size_t)b * 3;
        const int p0 = prev[0], p1 = prev[1], p2 = prev[2];
        const double lat_norm = fmax(smin[0] * smin[0], smax[0] * smax[0]);
        const double vr = smax[1] - smin[1], tr = smax[2] - smin[2];
        const double max_sqr_dist = (double)(nd * nd + nv * nv + nt * nt);
        const double* vs = bt.v_samples + (size_t)b * nv;
        for (int r = tid; r < C; r += T) {
            const uint32_t pay = order[r];
            const int q = (int)(pay >> 3);
            const int i = q / nvt, rem = q - i * nvt, j = rem / nt, k = rem - j * nt;
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
            X[r + (r >> 5)] = est <= __builtin_inf() ? est : __builtin_inf();  // a NaN estimate can never satisfy `<=`
            const uint32_t none = 0xFFFFu;
            const uint32_t n0 = i > 0 ? rank[q - nvt] : none, n1 = i < nd - 1 ? rank[q + nvt] : none;
            const uint32_t n2 = j > 0 ? rank[q - nt] : none, n3 = j < nv - 1 ? rank[q + nt] : none;
            const uint32_t n4 = k > 0 ? rank[q - 1] : none, n5 = k < nt - 1 ? rank[q + 1] : none;
            REC[r] = make_uint4(n0 | (n1 << 16), n2 | (n3 << 16), n4 | (n5 << 16), (uint32_t)limtab[r] | ((pay & 7u) << 16));
        }
    }
    group_sync();
    FP_SSTAMP(6);
#if defined(FP_ABL_SEARCH_NOWALK)  // timing ablation: prologue + ranking only
    return;
#endif

    // ---- the walk (fiss_plus_planner.py:80-148): wavefront 0's
    uint32_t G[NW], Q[NW], Fr[NW], fin[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        G[k] = Q[k] = Fr[k] = 0u;
        int rel = nfin - ((lane + 64 * k) << 5);
        rel = rel < 0 ? 0 : (rel > 32 ? 32 : rel);
        fin[k] = rel == 0 ? 0u : (0xFFFFFFFFu >> (32 - rel));
    }
    int ngen = 0;  // per-lane share of num_trajs_generated
    int num_iter = 0, num_validated = 0, num_checks = 0;
    int best = -1;
    int rs = -1;   // head of the candidate queue (rank), -1: empty
    int last_pop = -1;
    bool start_nan = false;  // the last initial guess started on a sample whose cost_final is NaN
    // one outer iteration of plan() (:80-148) -> 0: go on, 1: found (best), 2: gave up
    auto iterate = [&]() -> int {
        ++num_iter;
        bool slow = false;
        if (rs < 0) {
            // find_initial_guess (fiss_planner.py:140-150): argmin of cost_est over the samples not generated yet, `<=` keeps the
            // LAST minimum in raster order.  Lane L scans its own 32 ranks per word.
            double bv = __builtin_inf();
            int bqi = -1, br = -1;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int base = (lane + 64 * k) << 5;
#pragma unroll 8
                for (int bit = 0; bit < 32; ++bit) {
                    const int r = base + bit;
                    if (r < C && !((G[k] >> bit) & 1u)) {
                        const double v = X[r + (r >> 5)];
                        const int q = (int)(order[r] >> 3);
                        if (v < __builtin_inf() && (v < bv || (v == bv && q > bqi))) { bv = v; bqi = q; br = r; }
                    }
                }
            }
            const double m = uniform_f64(wave_min_f64(bv));
            if (!(m < __builtin_inf())) return 2;  // every sample searched, nothing feasible (:203-206)
            int cand = (bv == m) ? bqi : -1;
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) {
                const int o = __shfl_xor(cand, off, kWave);
                cand = o > cand ? o : cand;
            }
            cand = __builtin_amdgcn_readfirstlane(cand);
            const unsigned long long owner = __ballot(bv == m && bqi == cand);
            rs = __builtin_amdgcn_readlane(br, __ffsll((long long)owner) - 1);
            // generate_trajectory of the centre (the neighbours' turn comes in explore)
            set_rank<NW>(G, rs, lane);
            if (rs < nfin) set_rank<NW>(Q, rs, lane);
            ngen += lane == 0;
            slow = true;  // (an unqueued non-finite centre is not the candidate that gets validated)
        }
        rs = __builtin_amdgcn_readfirstlane(rs);
        const uint4 rec = REC[rs];
        if (slow) start_nan = (((uint32_t)__builtin_amdgcn_readfirstlane((int)rec.w) >> 16) & kPayNan) != 0u;
        explore<NW>(rec, lane, G, Q, Fr, fin, ngen);
        if (any_rank<NW>(Fr)) {  // frontier_idxs not empty (:110-113): pop the cheapest, explore it, until the frontier is empty
            slow = true;
            do {
                const int c = lowest_rank<NW>(Fr);
                reset_rank<NW>(Fr, c, lane);
                const uint4 rc = REC[c];
                explore<NW>(rc, lane, G, Q, Fr, fin, ngen);
            } while (any_rank<NW>(Fr));
        }
        // validation of the queue head (fiss_plus_planner.py:122-148)
        int pr = rs;
        uint32_t pw = rec.w;
        if (slow) {
            pr = lowest_rank<NW>(Q);
            if (pr < 0) return 2;
            if (pr != rs) pw = REC[pr].w;
        }
        reset_rank<NW>(Q, pr, lane);
        last_pop = pr;
        const uint32_t bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)pw) >> 16;
        ++num_validated;
        if (!(bits & kPayCfail)) {
            ++num_checks;
            if (!(bits & kPayColl)) { best = pr; return 1; }
        }
        rs = lowest_rank<NW>(Q);
        return 0;
    };
    auto finish = [&]() {
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) ngen += __shfl_xor(ngen, off, kWave);
        if (lane == 0) {
            if (best >= 0) {
                const int q = (int)(order[best] >> 3);
                const int i = q / nvt, rem = q - i * nvt, j = rem / nt, k = rem - j * nt;
                int32_t* out = fa.io.best_ijk + (size_t)b * 3;
                int32_t* pv = fa.io.prev_best_idx + (size_t)b * 3;
                out[0] = i; out[1] = j; out[2] = k;
                pv[0] = i; pv[1] = j; pv[2] = k;  // prev_best_idx persists across cycles (:140)
                fa.io.best_cost[b] = fa.cost_tbl[(size_t)b * C + (size_t)(i * nt + k) * nv + j];
                double* es = fa.io.end_state + (size_t)b * 3;
                es[0] = bt.d_samples[i]; es[1] = bt.v_samples[(size_t)b * nv + j]; es[2] = bt.t_samples[k];
                fa.io.refined[b] = 0;
                int32_t* s4 = fa.io.stats + (size_t)b * 4;
                s4[0] = num_iter; s4[1] = ngen; s4[2] = num_validated; s4[3] = num_checks;
            } else {
                write_none(num_iter, ngen, num_validated, num_checks);
            }
        }
    };

    // Jump scratch (the histogram's bytes, dead since T5): G / Q words after iteration 1, G / Q words after the jump, levels by rank.
    uint32_t* jG1 = (uint32_t*)hist;
    uint32_t* jG2 = jG1 + 64 * NW;
    uint32_t* jQ2 = jG2 + 64 * NW;
    uint16_t* lam = (uint16_t*)(jQ2 + 64 * NW);
    int* s_j = (int*)(smem + 192);  // [0] state after iteration 1 (0 jump, 1 finished, 2 serial only) [1] its pop [2] beta [3..5] counts
    if (wave == 0) {
        const int st = iterate();
        if (st != 0) finish();
        if (lane == 0) {
            s_j[0] = st != 0 ? 1 : ((fa.walk_jump && !start_nan) ? 0 : 2);
            s_j[1] = last_pop;
            s_j[2] = 0xFFFF;
            s_j[3] = s_j[4] = s_j[5] = 0;
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) jG1[lane + 64 * k] = G[k];
    }
    group_sync();
    FP_SSTAMP(7);
    const int after_first = s_j[0];
    if (after_first == 1) return;
    if (after_first == 0) {
        // ---- THE JUMP.  From the second iteration on the walk is a region growing: the head of the queue is the cheapest
        // generated sample that was not validated yet, exploring it generates its neighbours, and whatever the descent over the
        // frontier explores early costs no more than its centre.  So for any cost level: by the first iteration whose queue head
        // costs at least that level, exactly the samples that are connected to the generated set by samples BELOW the level have
        // been validated, they are all explored, nothing else is, and the frontier is empty - whatever the order was in which the
        // walk took them.  With level(v) = the smallest level that connects v (the minimax path cost from the generated set:
        // lam(v) = max(cost(v), min over neighbours lam(u)), a fixed point every lane relaxes for its own samples) and beta = the
        // smallest level of a FEASIBLE sample, no feasible sample is validated before the walk's state is
        //     validated = {first pop} + {lam < beta},  generated = generated + neighbourhood of {lam < beta},
        //     queue = generated & finite - validated,  num_iter = num_validated = |validated|
        // and the walk resumes there: typically the feasible sample IS the bottleneck and the next iteration ends the search.
        // Costs are compared as tie-run ends (the frontier bound of the records), so a level never splits a run of equal costs.
        // Not taken when the initial guess landed on a NaN cost: its first pop was validated without being explored.
        for (int r = tid; r < C; r += T) {
            const uint32_t w3 = REC[r].w;
            const bool seeded = (jG1[r >> 5] >> (r & 31)) & 1u;
            lam[r] = (uint16_t)((seeded && r < nfin) ? (w3 & 0xFFFFu) : 0xFFFFu);
        }
        group_sync();
        // Every sweep is a chain of dependent LDS round trips behind a barrier, so a thread keeps the records of its own samples in
        // registers (one round trip per sweep instead of two) and sweeps twice per barrier: the levels only ever fall towards the
        // fixed point, in any order of the updates, and the loop ends after a barrier interval in which nobody changed anything.
        constexpr int kPer = (CMAX + T - 1) / T;  // samples per thread
        if constexpr (kPer <= 4) {
            uint32_t nx[kPer], ny[kPer], nz[kPer], lv[kPer];
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int r = tid + u * T;
                lv[u] = 0xFFFFFFFFu;  // not mine / not finite
                nx[u] = ny[u] = nz[u] = 0u;
                if (r < nfin) {
                    const uint4 rec = REC[r];
                    // a missing neighbour (0xFFFF) reads the sample itself
                    auto fix = [&](uint32_t w) {
                        const uint32_t lo = w & 0xFFFFu, hi = w >> 16;
                        return (lo < (uint32_t)C ? lo : (uint32_t)r) | ((hi < (uint32_t)C ? hi : (uint32_t)r) << 16);
                    };
                    nx[u] = fix(rec.x); ny[u] = fix(rec.y); nz[u] = fix(rec.z);
                    lv[u] = rec.w & 0xFFFFu;
                }
            }
            for (;;) {
                int changed = 0;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                    for (int u = 0; u < kPer; ++u) {
                        if (lv[u] == 0xFFFFFFFFu) continue;
                        const int r = tid + u * T;
                        const uint32_t cur = lam[r];
                        if (cur == lv[u]) continue;  // at its floor
                        const uint32_t a0 = lam[nx[u] & 0xFFFFu], a1 = lam[nx[u] >> 16], a2 = lam[ny[u] & 0xFFFFu], a3 = lam[ny[u] >> 16],
                                       a4 = lam[nz[u] & 0xFFFFu], a5 = lam[nz[u] >> 16];
                        uint32_t mn = a0 < a1 ? a0 : a1;
                        const uint32_t m2 = a2 < a3 ? a2 : a3, m3 = a4 < a5 ? a4 : a5;
                        mn = mn < m2 ? mn : m2;
                        mn = mn < m3 ? mn : m3;
                        const uint32_t v = mn > lv[u] ? mn : lv[u];
                        if (v < cur) { lam[r] = (uint16_t)v; changed = 1; }
                    }
                }
                if (!__syncthreads_or(changed)) break;
            }
        } else {
            for (;;) {
                int changed = 0;
                for (int r = tid; r < nfin; r += T) {
                    const uint32_t cur = lam[r];
                    const uint4 rec = REC[r];
                    const uint32_t lev = rec.w & 0xFFFFu;
                    if (cur == lev) continue;  // at its floor
                    const uint32_t n[6] = {rec.x & 0xFFFFu, rec.x >> 16, rec.y & 0xFFFFu, rec.y >> 16, rec.z & 0xFFFFu, rec.z >> 16};
                    uint32_t mn = 0xFFFFu;
#pragma unroll
                    for (int e = 0; e < 6; ++e) {
                        const uint32_t v = lam[n[e] < (uint32_t)C ? n[e] : r];
                        mn = v < mn ? v : mn;
                    }
                    const uint32_t v = mn > lev ? mn : lev;
                    if (v < cur) { lam[r] = (uint16_t)v; changed = 1; }
                }
                if (!__syncthreads_or(changed)) break;
            }
        }
        FP_SSTAMP(8);
        {   // beta: the smallest level of a feasible sample
            uint32_t mine = 0xFFFFu;
            for (int r = tid; r < nfin; r += T)
                if (((REC[r].w >> 16) & (kPayCfail | kPayColl)) == 0u) { const uint32_t v = lam[r]; mine = v < mine ? v : mine; }
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(mine, off, kWave); mine = o < mine ? o : mine; }
            if (lane == 0) atomicMin(&s_j[2], (int)mine);
        }
        group_sync();
        const uint32_t beta = (uint32_t)s_j[2];
        const int first_pop = s_j[1];
        int n_pop = 0, n_chk = 0, n_gen = 0;
        for (int r0 = wave * kWave; r0 < ((C + kWave - 1) & ~(kWave - 1)); r0 += T) {  // whole wavefronts: the words come from ballots
            const int r = r0 + lane;
            bool gen = false, pop = false, ok = false;
            if (r < C) {
                const uint4 rec = REC[r];
                const bool in_r = r < nfin && lam[r] < beta;
                const uint32_t n[6] = {rec.x & 0xFFFFu, rec.x >> 16, rec.y & 0xFFFFu, rec.y >> 16, rec.z & 0xFFFFu, rec.z >> 16};
                bool near = false;
#pragma unroll
                for (int e = 0; e < 6; ++e) near |= n[e] < (uint32_t)nfin && lam[n[e]] < beta;
                gen = in_r || near || ((jG1[r >> 5] >> (r & 31)) & 1u);
                pop = in_r || r == first_pop;
                ok = !((rec.w >> 16) & kPayCfail);
            }
            const unsigned long long bg = __ballot(gen), bp = __ballot(pop), bc = __ballot(pop && ok);
            const unsigned long long bq2 = __ballot(gen && !pop && r < nfin);
            if (lane == 0) {
                jG2[r0 >> 5] = (uint32_t)bg; jG2[(r0 >> 5) + 1] = (uint32_t)(bg >> 32);
                jQ2[r0 >> 5] = (uint32_t)bq2; jQ2[(r0 >> 5) + 1] = (uint32_t)(bq2 >> 32);
            }
            n_pop += __popcll(bp); n_chk += __popcll(bc); n_gen += __popcll(bg);
        }
        if (lane == 0) { atomicAdd(&s_j[3], n_pop); atomicAdd(&s_j[4], n_chk); atomicAdd(&s_j[5], n_gen); }
        group_sync();
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int widx = lane + 64 * k;
                const bool have = (widx << 5) < ((C + kWave - 1) & ~(kWave - 1));
                G[k] = have ? jG2[widx] : 0u;
                Q[k] = have ? jQ2[widx] : 0u;
                Fr[k] = 0u;
            }
            num_iter = num_validated = s_j[3];
            num_checks = s_j[4];
            ngen = lane == 0 ? s_j[5] : 0;
            rs = lowest_rank<NW>(Q);
        }
        FP_SSTAMP(9);
    }
    if (wave != 0) return;
    int st;
    do { st = iterate(); } while (st == 0);
    finish();
    FP_SSTAMP(10);
#if defined(FP_PHASE_STAMPS)
    if (lane == 0 && fa.io.best_traj) fa.io.best_traj[((size_t)b * FP_ARR_COUNT + 15) * fa.io.traj_stride + 96 + 12] = (double)(t_begin & 0xFFFFFFFFFFll);
    if (lane == 0 && fa.io.best_traj) fa.io.best_traj[((size_t)b * FP_ARR_COUNT + 15) * fa.io.traj_stride + 96 + 11] = (double)(num_iter - (after_first == 0 ? s_j[3] : 0));  // iterations walked one by one
#endif
}


// LDS bytes of one ego's search (C samples, NB buckets)
__host__ __device__ inline int fissplus_lds_bytes(int C, int NB)
{
    const int C32 = C + (C >> 5) + 1;
    const int C8 = (C + 7) & ~7;
    const int NW = C <= 2048 ? 1 : 2;
    const int sort_scratch = 4 * 2 * (NB + 2);                 // histogram + cursors
    const int jump_scratch = 3 * 4 * 64 * NW + 2 * C8;         // three word arrays + the levels (they reuse the sort's bytes)
    return kMisc + 8 * ((C32 + 1) & ~1) + 16 * C8 + 3 * 2 * C8 + (sort_scratch > jump_scratch ? sort_scratch : jump_scratch) + 16;
}

}  // namespace fsp
}  // namespace fp
