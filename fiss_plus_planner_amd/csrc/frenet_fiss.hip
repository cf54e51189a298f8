// frenet_fiss.hip - FISS / FISS+ coarse search for a whole batch: one wavefront per ego walks the dense tables.
//
// The reference generates candidates lazily while it moves over the (d, v, t) index grid.  A candidate's cost_final,
// its constraint / collision outcome and its cost_est are pure functions of its index, so the walk
// can run over tables the lattice kernel already produced:  J = cost_final,  F = flag word,  E = cost_est.
// The walk itself is sequential and data dependent - a walk through a blocked scene pops hundreds of candidates one after the
// other, and the LONGEST walk of the batch is the kernel's duration - so what counts is the length of the dependent chain of one
// pop.  It is kept wave-uniform (every lane follows the same control flow, scalar state lives in uniform registers), and the
// two priority queues of the reference cost a handful of scalar instructions per operation:
//     queue head    = argmin of J over "in queue" entries          (fiss_planner.py:207 / :229, heapq order)
//     frontier pop  = argmin of J over frontier entries            (fiss_plus_planner.py:113)
// Both order by (J, raster index), a total order that is fixed before the walk starts.  The prologue sorts the lattice once by
// that key (bitonic network in LDS, the whole wavefront); afterwards a queue is a BIT SET over ranks - lane L holds ranks
// 64 L .. 64 L + 63 in one 64-bit register - with insert = set a bit, pop = clear it, argmin = lowest set bit (ballot + two
// find-first-set).  No key array, no rescans, no reductions.  Exact ties resolve to the LOWEST raster index (documented
// divergence from the reference's ValueError on tied heap entries); NaN / infinite costs are never popped.
//     initial guess = argmin of E over not-yet-generated entries, LAST minimum  (fiss_planner.py:140-150)
// is needed only when the queue runs dry: a strided scan + DPP wave minimum.
//
// Restated: fiss_planner.py:33-99 (cost_est), :101-138 (generate_trajectory -> table lookup), :140-188, :190-270;
//           fiss_plus_planner.py:30-59, :80-148.
#include "frenet_device.h"
#include "frenet_kernels.h"
#include "frenet_winner.h"
#include "frenet_advance.h"

namespace fp {

namespace {

constexpr uint8_t kGen = 1;

// Bit set over ranks 0 .. 4095: lane L owns ranks 64 L .. 64 L + 63.  All arguments wave-uniform.
struct RankSet {
    unsigned long long word;
    __device__ __forceinline__ void clear() { word = 0ull; }
    __device__ __forceinline__ void set(int r, int lane) { if (lane == (r >> 6)) word |= 1ull << (r & 63); }
    __device__ __forceinline__ void reset(int r, int lane) { if (lane == (r >> 6)) word &= ~(1ull << (r & 63)); }
    __device__ __forceinline__ int lowest() const  // -1 when empty
    {
        const unsigned long long nz = __ballot(word != 0ull);
        if (!nz) return -1;
        const int L = __ffsll((long long)nz) - 1;
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)word, L);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(word >> 32), L);
        return (L << 6) + (lo ? __ffs((int)lo) - 1 : 32 + __ffs((int)hi) - 1);
    }
};

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sort (key[i], idx[i]), i < n (any n <= P, P a power of two), ascending by key, ties by idx, by the W wavefronts of a workgroup.
// Bitonic network in the form whose compare-exchanges ALL put the smaller element at the lower index (a merge starts with the
// mirrored partner i ^ (k - 1), then half-cleaners i ^ j): elements n .. P-1 are +inf that is never stored - a partner beyond n
// means no exchange.  Two (P = 1024, 256 threads) compare-exchanges per thread and stage, a barrier per stage: ~20 us where a lone
// wavefront needs ~100.
template <int W>
__device__ void block_bitonic_sort(double* key, uint16_t* idx, int n, int P, int tid)
{
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool mirror = j == (k >> 1);
            for (int t = tid; t < (P >> 1); t += W * kWave) {
                const int a = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // t with a zero inserted at bit log2(j)
                const int b = mirror ? (a ^ (k - 1)) : (a | j);
                if (b < n) {
                    const double ka = key[a], kb = key[b];
                    const uint16_t ia = idx[a], ib = idx[b];
                    if (ka > kb || (ka == kb && ia > ib)) { key[a] = kb; key[b] = ka; idx[a] = ib; idx[b] = ia; }
                }
            }
            __syncthreads();
        }
    }
}

// The same sort for P = 64 E <= 256 elements with the data in REGISTERS: lane l holds elements l, l + 64, ... (E of them); a
// compare-exchange at distance j < 64 is a lane shuffle, at distance >= 64 a swap between two registers of the lane.  No LDS
// round trip per step: the single-ego planners (5 x 5 x 5 lattice, P = 128) pay ~1.5 us for it instead of ~7.
template <int E>
__device__ __forceinline__ void wave_bitonic_sort_regs(double* skey, uint16_t* order, int lane)
{
    double key[E];
    int idx[E];
#pragma unroll
    for (int m = 0; m < E; ++m) { key[m] = skey[m * kWave + lane]; idx[m] = order[m * kWave + lane]; }
#pragma unroll
    for (int k = 2; k <= E * kWave; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= kWave) {
                const int dm = j / kWave;
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    if ((m & dm) == 0) {  // pair (m, m + dm), both mine
                        const int e = m * kWave + lane;
                        const bool up = (e & k) == 0;
                        const bool gt = key[m] > key[m + dm] || (key[m] == key[m + dm] && idx[m] > idx[m + dm]);
                        if (gt == up) {
                            const double tk = key[m]; key[m] = key[m + dm]; key[m + dm] = tk;
                            const int ti = idx[m]; idx[m] = idx[m + dm]; idx[m + dm] = ti;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    const int e = m * kWave + lane;
                    const double ok = __shfl_xor(key[m], j, kWave);
                    const int oi = __shfl_xor(idx[m], j, kWave);
                    const bool up = (e & k) == 0, lower = (e & j) == 0;
                    const bool mine_gt = key[m] > ok || (key[m] == ok && idx[m] > oi);
                    // the lower element of an ascending pair keeps the smaller one
                    const bool take_other = (lower == up) ? mine_gt : !mine_gt;
                    if (take_other) { key[m] = ok; idx[m] = oi; }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < E; ++m) order[m * kWave + lane] = (uint16_t)idx[m];
}

struct Walk {
    const double* J;
    const uint8_t* F;
    uint8_t* st;
    const uint16_t* rank;  // raster index -> rank in (J, raster index) order
    const uint16_t* order; // rank -> raster index (the sort's payload)
    RankSet Q;      // "in queue"        -> queue head = lowest rank   (fiss_planner.py:207 / :229, heapq order)
    RankSet Fr;     // "on the frontier" -> frontier pop = lowest rank (fiss_plus_planner.py:113)
    const double* keyG;  // E (+inf where it is NaN) -> initial guess = LAST argmin over the not-yet-generated samples (needed only when the queue runs dry)
    const uint16_t* ijk;  // raster index -> i | j << sh_j | k << sh_k (bit fields sized for nd, nv, nt: at most 15 bits in all)
    int nd, nv, nt, C, lane;
    int num_iter, num_generated, num_validated, num_checks;

    __device__ __forceinline__ int raster(int i, int j, int k) const { return (i * nv + j) * nt + k; }

    // generate_trajectory (fiss_planner.py:101-138): a table lookup + bookkeeping.  Wave-uniform.
    __device__ __forceinline__ bool generate(int q, double& cost)
    {
        cost = J[q];
        const uint8_t s = st[q];
        if (s & kGen) return false;
        st[q] = s | kGen;  // candidate_trajs.put((cost_final, idx))
        if (cost < __builtin_inf()) Q.set(__builtin_amdgcn_readfirstlane((int)rank[q]), lane);  // NaN / inf: never popped
        ++num_generated;
        return true;
    }

    // find_initial_guess (fiss_planner.py:140-150): `cost_est <= min_cost` keeps the LAST minimum.  A strided scan of the whole
    // array (independent LDS reads), DPP wave minimum, ballot for the owner; exact ties across lanes -> highest raster index.
    __device__ __forceinline__ int initial_guess() const
    {
        double best = __builtin_inf();
        int bq = -1;
#pragma unroll 4
        for (int q = lane; q < C; q += kWave) {
            const double v = (st[q] & kGen) ? __builtin_inf() : keyG[q];  // only samples that are not generated yet
            if (v <= best && v < __builtin_inf()) { best = v; bq = q; }
        }
        const double m = wave_min_f64(best);
        if (!(m < __builtin_inf())) return -1;
        int cand = (best == m && bq >= 0) ? bq : -1;
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const int o = __shfl_xor(cand, off, kWave);
            cand = o > cand ? o : cand;
        }
        return __builtin_amdgcn_readfirstlane(cand);
    }
};

}  // namespace

// W wavefronts build the tables and sort (W = 1: small lattices, the sort runs in registers; W = 4: a barrier-stepped network over
// the whole workgroup); the walk itself is one wavefront's - the others leave before it starts.
template <int W>
__global__ __launch_bounds__(W * kWave) void fiss_search_kernel(FissArgs fa, int P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const fp_params& p = fa.ka.p;
    const fp_batch& bt = fa.ka.b;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1);
    const int nd = p.nd, nv = p.nv, nt = p.nt, C = nd * nv * nt;
    const int PS = W == 1 ? P : C;  // stored sort entries (the wide sort keeps its padding virtual)
    auto group_sync = [&]() { if constexpr (W == 1) wave_lds_sync(); else __syncthreads(); };
    if (bt.skip && bt.skip[b]) {  // finished ego of a closed-loop batch
        if (tid == 0) {
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
    // LDS: counters [4] | J [C] | sort keys [PS] | keyG [C] | order [PS] | rank [C] | ijk [C] | F [C] | st [C]
    // (PS = P, a power of two >= C, for the one-wavefront sorts; C for the workgroup sort)
    int* s_count = (int*)smem;
    double* J = (double*)(smem + 16);
    double* skey = J + C;
    double* keyG = skey + PS;
    uint16_t* order = (uint16_t*)(keyG + C);
    uint16_t* rank = order + PS;
    uint16_t* ijk = rank + C;
    uint8_t* F = (uint8_t*)(ijk + C);
    uint8_t* st = F + C;
    if (W > 1) {
        if (tid < 4) s_count[tid] = 0;
        __syncthreads();
    }
    // bit fields of the packed index: ceil(log2 nd) + ceil(log2 nv) + ceil(log2 nt) <= log2(FP_MAX_CAND) + 3 = 15
    const int sh_j = nd > 1 ? 32 - __clz(nd - 1) : 0;
    const int sh_k = sh_j + (nv > 1 ? 32 - __clz(nv - 1) : 0);
    const uint32_t mask_i = (1u << sh_j) - 1u, mask_j = (1u << (sh_k - sh_j)) - 1u;

    // ---- tables: FOP flat order (i_d, i_T, i_v) -> FISS raster (i_d, i_v, i_t); cost_est (fiss_planner.py:33-99)
    const double* smin = fa.io.samp_min + (size_t)b * 3;
    const double* smax = fa.io.samp_max + (size_t)b * 3;
    const int* prev = fa.io.prev_best_idx + (size_t)b * 3;
    const int p0 = prev[0], p1 = prev[1], p2 = prev[2];
    const double lat_norm = fmax(smin[0] * smin[0], smax[0] * smax[0]);
    const double vr = smax[1] - smin[1], tr = smax[2] - smin[2];
    const double max_sqr_dist = (double)(nd * nd + nv * nv + nt * nt);
    const double* vs = bt.v_samples + (size_t)b * nv;
    int feasible = 0, pass_constraints = 0;
    for (int q = tid; q < PS; q += W * kWave) {
        if (q >= C) {  // padding of the sort: behind every real entry
            skey[q] = __builtin_inf();
            order[q] = (uint16_t)q;
            continue;
        }
        const int k = q % nt, j = (q / nt) % nv, i = q / (nt * nv);
        const size_t flat = (size_t)b * C + (size_t)(i * nt + k) * nv + j;
        const double cost = fa.cost_tbl[flat];
        const uint8_t f = (uint8_t)(fa.flag_tbl[flat] & 0xFFu);
        J[q] = cost;
        skey[q] = cost < __builtin_inf() ? cost : __builtin_inf();  // NaN sorts with the infinities: never popped anyway
        order[q] = (uint16_t)q;
        F[q] = f;
        feasible += (f & FP_FLAG_INFEASIBLE) == 0;
        pass_constraints += (f & FP_FLAG_CONSTRAINTS) == 0;
        // state byte: bit 0 = generated; bits 2..7 = which of the six axis neighbours (-d, +d, -v, +v, -t, +t) exist
        st[q] = (uint8_t)(((i > 0) << 2) | ((i < nd - 1) << 3) | ((j > 0) << 4) | ((j < nv - 1) << 5) | ((k > 0) << 6) | ((k < nt - 1) << 7));
        ijk[q] = (uint16_t)((uint32_t)i | ((uint32_t)j << sh_j) | ((uint32_t)k << sh_k));
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
        keyG[q] = est <= __builtin_inf() ? est : __builtin_inf();  // a NaN estimate can never satisfy `<=`
    }

    // No feasible candidate anywhere in the lattice: the walk would generate and validate every sample, one per outer
    // iteration, and give up (fiss_planner.py:203-206).  Its outcome is closed form: num_iter = C + 1, generated =
    // validated = C, collision checks = samples that pass the constraints.  (Each iteration pops exactly one candidate.)
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        pass_constraints += __shfl_xor(pass_constraints, off, kWave);
        feasible += __shfl_xor(feasible, off, kWave);
    }
    if (W > 1) {  // totals of the workgroup
        if (lane == 0) { atomicAdd(&s_count[0], feasible); atomicAdd(&s_count[1], pass_constraints); }
        __syncthreads();
        feasible = s_count[0];
        pass_constraints = s_count[1];
    }
    if (feasible == 0) {
        if (tid == 0) {
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
    group_sync();
    // ---- the total order of the walk's two queues: (J, raster index)
#if !defined(FP_ABL_SEARCH_NOSORT)  // (timing ablation: without the sort the walk runs in raster order - wrong results)
    if constexpr (W == 1) {
        if (P == kWave) wave_bitonic_sort_regs<1>(skey, order, lane);
        else if (P == 2 * kWave) wave_bitonic_sort_regs<2>(skey, order, lane);
        else wave_bitonic_sort_regs<4>(skey, order, lane);  // (the launcher gives lattices above 256 candidates to W = 4)
    } else {
        block_bitonic_sort<W>(skey, order, C, P, tid);
    }
#endif
    group_sync();
    // rank -> (raster index | flag byte << 16): one LDS read tells a pop which candidate it is and whether it is feasible.  The
    // words take over the sort keys' bytes (dead from here on).
    uint32_t* order32 = (uint32_t*)skey;
    for (int r0 = 0; r0 < PS; r0 += W * kWave) {
        const int r = r0 + tid;
        const int q = r < PS ? order[r] : C;
        uint32_t word = 0xFFFFu;
        if (q < C) { rank[q] = (uint16_t)r; word = (uint32_t)q | ((uint32_t)F[q] << 16); }
        group_sync();  // (the words of this batch overlay keys whose ranks were read above)
        if (r < PS) order32[r] = word;
    }
    group_sync();
    if (W > 1 && tid >= kWave) return;  // the walk is one wavefront's

    Walk w{J, F, st, rank, order, {}, {}, keyG, ijk, nd, nv, nt, C, lane, 0, 0, 0, 0};
#if defined(FP_ABL_SEARCH_NOWALK)  // timing ablation: prologue + sort only
    return;
#endif
    w.Q.clear();
    w.Fr.clear();
    const int sizes[3] = {nd, nv, nt};
    int best = -1;
    const bool plus = fa.opts.kind == FP_FISS_PLUS;
    // lane l < 6 owns the neighbour along axis l / 2 in direction +-1 (FISS+ exploration)
    const int my_dim = lane >> 1, my_step = (lane & 1) ? +1 : -1;
    const int my_stride = my_step * (my_dim == 0 ? nv * nt : (my_dim == 1 ? nt : 1));
    for (;;) {
        ++w.num_iter;
        const int rh = w.Q.lowest();
        int q;
        uint32_t head_word = 0;
        if (rh < 0) {
            q = w.initial_guess();
            if (q < 0) break;  // every sample searched, nothing feasible (:203-206)
        } else {
            head_word = (uint32_t)__builtin_amdgcn_readfirstlane((int)order32[rh]);  // peek the most likely candidate (:207-209)
            q = (int)(head_word & 0xFFFFu);
        }
        q = __builtin_amdgcn_readfirstlane(q);  // wave-uniform: scalar addressing and branches from here on
        if (!plus) {
            const uint32_t packed = __builtin_amdgcn_readfirstlane((int)ijk[q]);
            int idx[3] = {(int)(packed & mask_i), (int)((packed >> sh_j) & mask_j), (int)(packed >> sh_k)};
            double cost_center, cost;
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
            // explore_neighbors + frontier (fiss_plus_planner.py:30-59, :106-116).  One LDS round trip per centre: its packed
            // index, state, cost and rank together with those of the six axis neighbours (distinct cells, lanes 0..5, addresses
            // clamped - the range check needs the unpacked index and is applied afterwards).  The centre is generated first
            // (wave-uniform); the bookkeeping counts are order-independent.
            int cq = q;             // raster index of the centre
            int n_front = 0;        // entries on the frontier (wave-uniform): an empty frontier needs no bit-set search
            for (;;) {
                int nq = cq + my_stride;
                nq = nq < 0 ? 0 : (nq > C - 1 ? C - 1 : nq);
                const uint8_t cs = st[cq];
                const double cost_center = J[cq];
                const int cr = rank[cq];
                const uint8_t s = st[nq];
                const double c = J[nq];
                const int nr = rank[nq];
                if (!(cs & kGen)) {  // generate_trajectory of the centre
                    st[cq] = cs | kGen;
                    if (cost_center < __builtin_inf()) w.Q.set(__builtin_amdgcn_readfirstlane(cr), lane);
                    ++w.num_generated;
                }
                const bool is_new = lane < 6 && ((cs >> (2 + (lane < 6 ? lane : 0))) & 1) && !(s & kGen);  // the neighbour exists and is not generated
                const bool to_frontier = is_new && c <= cost_center;  // frontier_idxs.put((cost, idx))
                if (is_new) st[nq] = s | kGen;
                const unsigned long long fresh = __ballot(is_new);
                unsigned long long queued = __ballot(is_new && c < __builtin_inf());  // NaN / inf: generated, never popped
                const unsigned long long front = __ballot(to_frontier);
                w.num_generated += __popcll(fresh);
                n_front += __popcll(front & queued);
                while (queued) {  // register side of the (at most six) inserts
                    const int l = __ffsll((long long)queued) - 1;
                    queued &= queued - 1;
                    const int ur = __builtin_amdgcn_readlane(nr, l);
                    w.Q.set(ur, lane);
                    if ((front >> l) & 1ull) w.Fr.set(ur, lane);
                }
                if (n_front == 0) break;
                --n_front;
                const int rf = w.Fr.lowest();
                w.Fr.reset(rf, lane);
                cq = __builtin_amdgcn_readfirstlane((int)(order32[rf] & 0xFFFFu));
            }
        }
        // validation of the queue head (fiss_planner.py:229-258).  (Reusing the head found at the top of the iteration when nothing
        // was queued on the way was tried: the extra uniform branch costs more than the ballot + two find-first-set it saves.)
        const int rp = w.Q.lowest();
        if (rp < 0) break;
        w.Q.reset(rp, lane);
        // (usually the head the iteration started with: nothing cheaper was generated on the way)
        const uint32_t pop_word = rp == rh ? head_word : (uint32_t)__builtin_amdgcn_readfirstlane((int)order32[rp]);
        const int popped = (int)(pop_word & 0xFFFFu);
        const uint32_t f = pop_word >> 16;
        ++w.num_validated;
        if (f & FP_FLAG_CONSTRAINTS) continue;
        ++w.num_checks;
        if (!(f & FP_FLAG_COLLISION)) { best = popped; break; }
    }
    if (lane == 0) {
        int32_t* out = fa.io.best_ijk + (size_t)b * 3;
        int32_t* pv = fa.io.prev_best_idx + (size_t)b * 3;
        if (best >= 0) {
            const uint32_t pk = ijk[best];
            const int i = (int)(pk & mask_i), j = (int)((pk >> sh_j) & mask_j), k = (int)(pk >> sh_k);
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

// ---------------------------------------------------------------------------
// FISS+ refinement (fiss_plus_planner.py:207-326), one workgroup of kRefineWaves wavefronts per ego.
//   * costs in closed form: the power sums S_k(N) = sum_{i<N} t_i^k of a horizon come from Faulhaber's polynomials, so the cost of ANY
//     end state is O(1) (lon_cost_sums / lat_cost_sums): per round lanes 0..5 price the six probes clip(x -/+ res_dim e_dim)
//     (:213-232), the finite-difference gradient and the decayed step are wave-uniform arithmetic on shuffled lane values
//     (:262-271), lane 0 prices the trajectory at the new x.  The coarse winner is priced by the same function, so a probe clipped
//     back onto x ties with it exactly (`cost > coarse cost` ends the loop, :303-304).  Every wavefront runs the rounds
//     redundantly (bit-identical), so all of them hold the same candidate list.
//   * validation is lazy and in cost order like refined_trajs.get() (:301-323): the next kRefineWaves trajectories in pop order
//     are checked speculatively side by side, one whole wavefront each (lane = time point for the masks / truncation / Cartesian
//     points, then lane = (checked pose, obstacle) pair), and the verdicts are consumed in pop order.
//   * collision checks never wait on L2 / HBM per pair: the obstacle rows the horizon can touch are staged once per ego as an
//     fp32 table relative to the first knot; a conservative fp32 circle test (branch-free, four table reads in flight per lane)
//     feeds a per-wavefront survivor queue, survivors take the exact fp64 circle + separating-axis test one per lane.
//   The kernel is instruction-issue bound at low occupancy (160+ VGPRs): 3 workgroups per CU need <= 168 VGPRs and <= 53 KB of
//   LDS, which is what sizes the survivor queues.
// ---------------------------------------------------------------------------
namespace {

constexpr int kQueue = 5 * kWave;  // survivor queue entries per wavefront (refinement kernel): drained above kWave, filled 4 kWave at a time
constexpr int kHot = 12;           // remembered collision pairs per ego; each is tried with its 2 time steps either side: 5 kHot <= kWave lanes

struct RefineLds {
    double* S;      // [FP_FAST_POINTS + 1][11]
    double2* xy;    // [FP_FAST_POINTS]
    double* knots;  // [nx]
    double* coef;   // [8][nx]
    // conservative fp32 broad phase, staged once per ego (nullptr: table over the LDS budget, pairs are read from the scene table)
    float4* pt;     // [rows][n_obs] {x - ox, y - oy, (padded bounding-circle sum)^2 or -1 when absent, (step | obstacle << 8) as int bits}
    float2* xyf;    // [FP_FAST_POINTS] this wavefront's poses relative to (ox, oy), fp32
    uint16_t* queue;  // [kQueue] this wavefront's broad-phase survivors (pair table indices)
    // pairs (pair table indices) at which earlier trajectories of this ego collided: the refined trajectories are neighbours in end
    // state space, so the next one most likely collides at the same obstacle a step or two away - tried first (kHot entries + count)
    int* hot;
    double ox, oy;  // first knot of the reference line: keeps the fp32 coordinates small
    int scene, t_now, horizon_cap;  // per-ego scene facts read once (scene < 0: none; horizon_cap = final_time_step - t_now)
};

__device__ __forceinline__ double analytic_cost(const fp_params& p, const double* eg, double target_speed, const double* x, const double* Stab, int n_max = FP_FAST_POINTS)
{
    const double T = x[2];
    const int N = (T == T) ? arange_len(T, p.tick_t) : 0;
    if (N <= 0 || N > n_max) return __builtin_nan("");
    const Quartic lon = quartic_bvp(eg[0], eg[1], eg[2], x[1], 0.0, T);
    const Quintic lat = quintic_bvp(eg[3], eg[4], eg[5], x[0], 0.0, 0.0, T);
    double S[11], ls[3], ds[3];
    (void)Stab;
    power_sums_closed(N, p.tick_t, S);  // S_k(N) = sum_{i<N} t_i^k in closed form (Faulhaber): no table
    lon_cost_sums(lon, target_speed, S, ls);
    lat_cost_sums(lat, S, ds);
    return combine_cost(p, N, ls, ds);
}

// The same cost with the two halves of the dependent chain in two lanes: lane L (role 0) solves the longitudinal boundary-value problem
// and sums its three cost terms, lane L + 8 (role 1) does the lateral ones (both need the power sums of the horizon: each computes
// them), then lane L fetches the lateral sums and recombines.  Same arithmetic per term, so the value is analytic_cost's bit for bit;
// the chain of one evaluation is ~45 % shorter (the refinement rounds are R + 1 evaluations one after the other on a lone wavefront).
// Valid in role-0 lanes; x must be the same in lanes L and L + 8.
__device__ __forceinline__ double analytic_cost_two_lanes(const fp_params& p, const double* eg, double target_speed, const double* x, int role, int n_max = FP_FAST_POINTS)
{
    const double T = x[2];
    const int N = (T == T) ? arange_len(T, p.tick_t) : 0;
    const bool ok = N > 0 && N <= n_max;
    double S[11], part[3] = {0.0, 0.0, 0.0};
    if (ok) {
        power_sums_closed(N, p.tick_t, S);
        if (role == 0) {
            const Quartic lon = quartic_bvp(eg[0], eg[1], eg[2], x[1], 0.0, T);
            lon_cost_sums(lon, target_speed, S, part);
        } else {
            const Quintic lat = quintic_bvp(eg[3], eg[4], eg[5], x[0], 0.0, 0.0, T);
            lat_cost_sums(lat, S, part);
        }
    }
    double ds[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) ds[m] = __shfl_down(part[m], 8, kWave);
    return ok ? combine_cost(p, N, part, ds) : __builtin_nan("");
}

// Optional curvature checks (frenet_optimal_planner.py:145-150) of ONE trajectory whose M Cartesian points sit in xy[] (LDS), by
// the whole wavefront: lane l holds elements l and l + 64 of every chain (M <= 128).  Same difference chains as CurvTrack /
// winner_series: yaw_k = atan2 of segment k (the last point repeats the previous heading, :129), c = diff(yaw) / ds,
// c_d = diff(c) / dt, c_dd = diff(c_d) / dt; element i + 1 comes from the neighbouring lane.
template <int NCH>
__device__ __forceinline__ uint32_t wave_curvature_flags(const fp_params& p, const double2* xy, int M, int lane)
{
    if (M < 2) return 0u;
    double yaw[NCH], ds[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
        const int i = lane + h * kWave;
        const int k = i < M - 1 ? i : M - 2;  // segment whose heading element i carries
        const double2 a = xy[k], b2 = xy[k + 1];
        yaw[h] = atan2(b2.y - a.y, b2.x - a.x);
        ds[h] = hypot(b2.x - a.x, b2.y - a.y);
    }
    auto next = [&](const double* v, int h) {  // element (lane + 64 h) + 1 of a chain
        const double dn = __shfl_down(v[h], 1, kWave);
        return (h + 1 < NCH && lane == kWave - 1) ? lane_value(v[h + 1 < NCH ? h + 1 : h], 0) : dn;
    };
    uint32_t flags = 0;
    double c[NCH], cd[NCH];
    bool bad_c = false, bad_cd = false, bad_cdd = false;
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
        c[h] = (next(yaw, h) - yaw[h]) / ds[h];
        bad_c |= lane + h * kWave < M - 1 && fabs(c[h]) > p.max_curvature;
    }
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
        cd[h] = (next(c, h) - c[h]) / p.tick_t;
        bad_cd |= lane + h * kWave < M - 2 && fabs(cd[h]) > p.max_kappa_d;
    }
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
        const double cdd = (next(cd, h) - cd[h]) / p.tick_t;
        bad_cdd |= lane + h * kWave < M - 3 && fabs(cdd) > p.max_kappa_dd;
    }
    if (__ballot(bad_c)) flags |= FP_FLAG_CURVATURE;
    if (__ballot(bad_cd)) flags |= FP_FLAG_KAPPA_D;
    if (__ballot(bad_cdd)) flags |= FP_FLAG_KAPPA_DD;
    return flags;
}

// constraint + collision flags of ONE trajectory, computed by the whole wavefront (all arguments wave-uniform).  NCH = points per lane:
// 2 (trajectories of up to FP_FAST_POINTS = 128 points: the instance every default setting takes) or 4 (up to FP_MAX_POINTS = 256).
template <int NCH>
__device__ uint32_t wave_traj_flags(const KernelArgs& ka, int b, const double* eg, const double* x, const RefineLds& L, int nx, int lane, double* stamp = nullptr,
                                long long t_begin = 0)
{
#define FP_WSTAMP(k) do { if (stamp && lane == 0) stamp[k] = (double)(wall_clock64() - t_begin); } while (0)

    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    const double T = x[2];
    const int N = arange_len(T, p.tick_t);
    const Quartic lon = quartic_bvp(eg[0], eg[1], eg[2], x[1], 0.0, T);
    const Quintic lat = quintic_bvp(eg[3], eg[4], eg[5], x[0], 0.0, 0.0, T);
    SplineLds sp{L.knots, L.coef, nx, nx};
    const double knot0 = L.knots[0], knot_last = L.knots[nx - 1];
    const double seg_scale = (double)(nx - 1) / (knot_last - knot0);  // (NaN / inf / <= 0: spline_segment divides per point instead)
    constexpr int kPts = NCH * kWave;
    unsigned long long off_m[NCH];
    bool bad_speed = false, bad_accel = false;
    const int need_xy = p.curvature_mask ? kPts : ((L.scene >= 0 && bt.n_obs > 0) ? L.horizon_cap : -1);
#pragma unroll
    for (int half = 0; half < NCH; ++half) {
        const int i = lane + half * kWave;
        bool off = false;
        if (i < N) {
            const double t = (double)i * p.tick_t;
            double s, s_d, s_dd, s_ddd;
            quartic_eval(lon, t, s, s_d, s_dd, s_ddd);
            bad_speed |= s_d > p.max_speed;
            bad_accel |= fabs(s_dd) > p.max_accel;
            off = !(s >= knot0) || !(s < knot_last);
            // Cartesian points: the collision checks read poses k < min(M, horizon) and the point after each (heading); the optional
            // curvature checks read all of them.  The rest of the trajectory only needs its masks and the range test above.
            if (!off && i <= need_xy) {
                const double d = fma(fma(fma(fma(fma(lat.a5, t, lat.a4), t, lat.a3), t, lat.a2), t, lat.a1), t, lat.a0);
                const int seg = spline_segment(sp, s, -1, seg_scale < 1e300 ? seg_scale : 0.0);
                double px, py, tx, ty, cx, cy;
                spline_frame(sp, seg, s - L.knots[seg], px, py, tx, ty);
                frenet_to_cartesian(px, py, tx, ty, d, cx, cy);
                L.xy[i] = make_double2(cx, cy);
                if (L.pt) L.xyf[i] = make_float2((float)(cx - L.ox), (float)(cy - L.oy));
            }
        }
        off_m[half] = __ballot(off);
    }
    FP_WSTAMP(12);
    uint32_t flags = 0;
    if (__ballot(bad_speed)) flags |= FP_FLAG_SPEED;
    if (__ballot(bad_accel)) flags |= FP_FLAG_ACCEL;
    int M = N;
#pragma unroll
    for (int half = NCH - 1; half >= 0; --half)
        if (off_m[half]) M = half * kWave + __ffsll((long long)off_m[half]) - 1;
    if (M < N) flags |= FP_FLAG_TRUNCATED;
    flags |= ((uint32_t)N << FP_FLAG_N_SHIFT) | ((uint32_t)M << FP_FLAG_M_SHIFT);
    if (p.curvature_mask && M >= 2) {  // optional checks (:145-150): they read the points this wavefront just wrote
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        flags |= wave_curvature_flags<NCH>(p, L.xy, M, lane);
    }
    // collision (frenet_optimal_planner.py:168-195)
    const int sc = L.scene;
    const int n_obs = sc >= 0 ? bt.n_obs : 0;
    if (n_obs <= 0) return flags;
    const int t_now = L.t_now;
    const int horizon_cap = L.horizon_cap;
    if (M == 1 && horizon_cap >= 1) return flags | FP_FLAG_COLLISION;  // traj.yaw is empty -> IndexError -> collision (:178-182)
    if (M < 2) return flags;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double* scene = bt.obs_pose + (size_t)sc * bt.T_obs * n_obs * 4;
    const double* gd = bt.obs_dims + (size_t)sc * n_obs * 2;
    const double veh_hl = 0.5 * p.veh_l, veh_hw = 0.5 * p.veh_w;
    const double r_ego = sqrt(fma(veh_hl, veh_hl, veh_hw * veh_hw));
    const int k_end = M < horizon_cap ? M : horizon_cap;
    const int stride = p.check_stride;
    // a non-finite checked pose makes the reference's polygon construction fail -> "collision" (:178-182)
    {
        bool broken = false;
#pragma unroll
        for (int half = 0; half < NCH; ++half) {
            const int i = lane + half * kWave;
            if (i < k_end && (i % stride) == 0) {
                const double2 q = L.xy[i];
                broken |= !(q.x == q.x) || !(q.y == q.y);
            }
        }
        if (__ballot(broken)) return flags | FP_FLAG_COLLISION;
    }
    const int in_table = bt.T_obs - t_now;
    const int k_lim = k_end < in_table ? k_end : in_table;  // beyond the table state_at_time() is None for every obstacle
    const int n_poses = k_lim > 0 ? (k_lim + stride - 1) / stride : 0;
    const int P = n_poses * n_obs;
    // exact test of pair (checked row r, obstacle j): bounding circles, then the separating-axis test
    auto pair_hits = [&](int kk, int j, const double4& ps) -> bool {  // kk = checked time step
        if (ps.w == 0.0) return false;
        const double hl = 0.5 * gd[2 * j], hw = 0.5 * gd[2 * j + 1];
        const double R = (r_ego + sqrt(fma(hl, hl, hw * hw))) * (1.0 + 1e-12);
        const double2 pc = L.xy[kk];
        const double dx = ps.x - pc.x, dy = ps.y - pc.y;
        if (!(fma(dx, dx, dy * dy) <= R * R)) return false;
        const int a2 = (kk + 1 < M) ? kk : kk - 1;  // heading: forward difference, previous one for the last point (:127-129)
        const double2 pa = L.xy[a2], pb = L.xy[a2 + 1];
        Obb ego;
        step_heading(pb.x - pa.x, pb.y - pa.y, ego.c, ego.s);
        ego.x = pc.x; ego.y = pc.y; ego.hl = veh_hl; ego.hw = veh_hw;
        double oc, os;
        sincos_snapped(ps.z, os, oc);
        return shape_overlap(ego, Obb{ps.x, ps.y, oc, os, hl, hw}, bt.obs_nvert, bt.obs_poly, bt.poly_stride, (size_t)sc * n_obs + j);
    };
    if (L.pt) {
        // every (checked pose, obstacle) pair is independent: lanes run over the flattened pair table; the fp32 circle test is
        // conservative (radius padded far beyond the fp32 rounding).  Survivors are compacted into this wavefront's queue and
        // take the exact fp64 test one per lane, so their pose reads (L2 / HBM, ~1-2 us) are all in flight together
        constexpr int kU = 4;  // table reads in flight per lane: branch-free (clamped index) so the LDS latencies overlap
        // (0) the pairs at which earlier trajectories of this ego collided, with two checked steps either side (same obstacle): one
        // exact pass; a hit ends the trajectory before its ~1250 pairs are scanned.  (A collision is a collision whichever pair shows it.)
        const int n_hot = L.hot[kHot] < kHot ? L.hot[kHot] : kHot;
        if (n_hot > 0) {
            const int hh = lane / 5, dr = lane - 5 * hh - 2;
            bool hit = false;
            if (hh < n_hot) {
                const int e = L.hot[hh] + dr * n_obs;
                if (e >= 0 && e < P) {
                    const uint32_t kj = (uint32_t)__float_as_int(L.pt[e].w);
                    const int kk = kj & 0xFF, j = kj >> 8;
                    hit = pair_hits(kk, j, *(const double4*)(scene + ((size_t)(kk + t_now) * n_obs + j) * 4));
                }
            }
            if (__ballot(hit)) return flags | FP_FLAG_COLLISION;
        }
        int qn = 0;
        for (int e0 = 0; e0 < P + kU * kWave; e0 += kU * kWave) {  // one extra trip drains what is left: the (large) exact test
            if (e0 < P) {                                            // is inlined once
                float4 q[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int e = e0 + u * kWave + lane;
                    q[u] = L.pt[e < P ? e : P - 1];
                }
                float2 c[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) c[u] = L.xyf[__float_as_int(q[u].w) & 0xFF];
                bool pass[kU], any = false;
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const float dx = q[u].x - c[u].x, dy = q[u].y - c[u].y;
                    pass[u] = (dx * dx + dy * dy <= q[u].z) && (e0 + u * kWave + lane < P);
                    any |= pass[u];
                }
                if (__ballot(any)) {  // (most trips keep nothing: one ballot instead of four compactions)
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const unsigned long long m = __ballot(pass[u]);
                        const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (pass[u]) L.queue[qn + below] = (uint16_t)(e0 + u * kWave + lane);
                        qn += __popcll(m);
                    }
                }
            }
            if (e0 >= P) FP_WSTAMP(13);
            // survivors are tested as soon as a trip leaves any (the table is in time order: a trajectory that collides early is
            // done before the rest of its horizon is scanned)
            if (qn > 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                bool hit = false;
                int hit_e = -1;
                for (int slot = lane; slot < qn; slot += kWave) {  // survivors: exact fp64 test, their pose reads in flight together
                    const int e = L.queue[slot];
                    const uint32_t kj = (uint32_t)__float_as_int(L.pt[e].w);
                    const int kk = kj & 0xFF, j = kj >> 8;
                    if (pair_hits(kk, j, *(const double4*)(scene + ((size_t)(kk + t_now) * n_obs + j) * 4))) { hit = true; hit_e = e; }
                }
                qn = 0;
                __builtin_amdgcn_wave_barrier();
                const unsigned long long hm = __ballot(hit);
                if (hm) {
                    if (lane == __ffsll((long long)hm) - 1) {  // remember the pair for the ego's next trajectories
                        const int pos = atomicAdd(&L.hot[kHot], 1);
                        if (pos < kHot) L.hot[pos] = hit_e;
                    }
                    return flags | FP_FLAG_COLLISION;
                }
            }
        }
        return flags;
    }
    // no table: coalesced 32-byte pose reads straight from the scene table, four in flight per lane
    constexpr int kInFlight = 4;
    for (int e0 = 0; e0 < P; e0 += kInFlight * kWave) {
        bool hit = false;
        double4 pq[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const int e = e0 + u * kWave + lane;
            pq[u] = make_double4(0.0, 0.0, 0.0, 0.0);
            if (e < P) {
                const int r = e / n_obs, j = e - r * n_obs;
                pq[u] = *(const double4*)(scene + ((size_t)(r * stride + t_now) * n_obs + j) * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const int e = e0 + u * kWave + lane;
            if (e < P) {
                const int r = e / n_obs, j = e - r * n_obs;
                hit |= pair_hits(r * stride, j, pq[u]);
            }
        }
        if (__ballot(hit)) return flags | FP_FLAG_COLLISION;
    }
    return flags;
}

}  // namespace

constexpr int kRefineWaves = 4;  // trajectories validated speculatively side by side (one wavefront each)

// LDS layout of the refinement kernel, in doubles (every double2 region starts 16-byte aligned):
//   kRefineWaves x (fp64 poses, fp32 relative poses) | knots + coef (9 NX, padded even)
//   | pair table (float4 per entry) | verdicts (32 B) | kRefineWaves survivor queues
constexpr int kRefineS = 0;  // (round 2 kept a table S[N][k] of power sums here: 11 KB that cost the fourth workgroup per CU)
__host__ __device__ constexpr int refine_spline_off(int pts) { return kRefineS + 3 * pts * kRefineWaves; }
__host__ __device__ constexpr int refine_pt_off(int NX, int pts) { return refine_spline_off(pts) + ((9 * NX + 1) & ~1); }
__host__ __device__ constexpr int refine_lds_bytes(int NX, int pt_entries, int pts)
{
    return (int)sizeof(double) * (refine_pt_off(NX, pts) + 2 * pt_entries) + 32 + kRefineWaves * 2 * kQueue + 4 * (kHot + 4);
}

#ifndef FP_REFINE_OCC
#define FP_REFINE_OCC 3   // workgroups per CU (= waves per SIMD) the register budget is sized for: 4 fits the LDS (40.7 KB) but spills 61 VGPRs, measured slower
#endif
// NCH = 2: trajectories of up to 128 points (three workgroups per CU); NCH = 4: up to FP_MAX_POINTS = 256 (tick_t below ~0.04 s at the default
// horizons: twice the per-wavefront point scratch, two workgroups per CU - the register budget of a loop nest twice as deep).
template <int NCH>
__global__ __launch_bounds__(kWave * kRefineWaves, NCH == 2 ? FP_REFINE_OCC : 2) void fiss_refine_kernel(FissArgs fa, int pt_rows_max, const int* perm, int* dur)
{
    constexpr int kPts = NCH * kWave;  // points per trajectory this instance holds
#if defined(FP_PHASE_STAMPS)
    const long long t_begin = wall_clock64();
#else
    const long long t_begin = dur ? wall_clock64() : 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const KernelArgs& ka = fa.ka;
    const fp_params& p = ka.p;
    const fp_batch& bt = ka.b;
    const int b = perm ? perm[blockIdx.x] : (int)blockIdx.x;  // launch order: longest egos first when the host has one
    // Timing diagnostic (tools/refine_stamps.py, -DFP_PHASE_STAMPS): thread 0 leaves 10 ns ticks since the workgroup started in
    // columns 112.. of the last row of the ego's series block (sparse layout, stride 128); column 112 = the absolute start.
#if defined(FP_PHASE_STAMPS)
#define FP_RSTAMP(k) do { if (threadIdx.x == 0 && fa.io.best_traj) fa.io.best_traj[((size_t)b * FP_ARR_COUNT + 15) * fa.io.traj_stride + 112 + (k)] = (k) == 0 ? (double)(t_begin & 0xFFFFFFFFFFll) : (double)(wall_clock64() - t_begin); } while (0)
#else
#define FP_RSTAMP(k) do { } while (0)
#endif
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    constexpr int kThreads = kWave * kRefineWaves;
    const int32_t* ijk = fa.io.best_ijk + (size_t)b * 3;
    const int R = fa.opts.max_refine_iters;
    if (ijk[0] < 0 || R <= 0) {  // nothing found by the coarse search: plan() returns None
        if (tid == 0 && dur) dur[b] = 0;
        if (fa.io.best_traj) {   // NaN series, flag word 0
            KernelArgs kw = ka;
            kw.r.best_traj = fa.io.best_traj;
            kw.r.best_flags = fa.io.best_flags;
            kw.r.traj_stride = fa.io.traj_stride;
            kw.r.traj_sparse = fa.io.traj_sparse;
            const double none = __builtin_nan("");
            if (wave == 0) winner_series_wave(kw, b, b, false, none, none, none, lane, SplineLds{nullptr, nullptr, 0, 0});
        }
        // fp_plan_fiss_step: the hand-over of an ego without a trajectory (R > 0: with R == 0 this kernel is not launched) - "no solution"
        if (ka.has_loop && tid == 0 && R > 0) {
            int off = (int)offsetof(FissArgs, ka);
            asm volatile("" : "+s"(off) : : "memory");
            const KernelArgs& kl = *(const KernelArgs*)((const unsigned char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
            const double none = __builtin_nan("");
            advance_ego_to(kl, b, none, none, none, kl.loop);
        }
        return;
    }
    const int f = bt.frame_of[b];
    const int nx = bt.nx[f];
    const int verdict_off = refine_lds_bytes(bt.NX, pt_rows_max * bt.n_obs, kPts) - 32 - kRefineWaves * 2 * kQueue - 4 * (kHot + 4);
    RefineLds L;
    L.S = (double*)smem;
    L.xy = (double2*)(L.S + kRefineS) + wave * kPts;  // one Cartesian scratch row per wavefront
    L.xyf = (float2*)(L.S + kRefineS + 2 * kPts * kRefineWaves) + wave * kPts;
    L.knots = L.S + refine_spline_off(kPts);
    L.queue = (uint16_t*)(smem + verdict_off + 32) + wave * kQueue;
    L.hot = (int*)(smem + verdict_off + 32 + kRefineWaves * 2 * kQueue);  // [kHot] pairs + [1] count
    if (tid == 0) L.hot[kHot] = 0;  // (the barrier behind the rounds orders it before the first validation)
    L.coef = L.knots + nx;
    // Order of the prologue: wavefront 0 runs the refinement rounds (they only need the ego state: the power sums of a probe's
    // horizon come in closed form) WHILE wavefronts 1..3 stage the spline and the pair table; then the candidate list goes through
    // LDS to every wavefront.  The rounds are ~half of the kernel's instructions: running them once instead of
    // four times is what matters (the kernel is instruction-issue bound), hiding them behind the staging is a bonus.
    const double* gk = bt.knots + (size_t)f * bt.NX;
    const double* gc = bt.coef + (size_t)f * 8 * bt.NX;
    L.pt = nullptr;
    L.ox = gc[0];                       // x, y of the first knot
    L.oy = gc[(size_t)4 * bt.NX];
    const int sc0 = bt.scene_of[b];
    L.scene = sc0;
    L.t_now = bt.t_now[b];
    L.horizon_cap = sc0 >= 0 ? bt.final_time_step[sc0] - L.t_now : 0;
    int pt_rows = 0;
    if (sc0 >= 0 && bt.n_obs > 0 && pt_rows_max > 0) {
        int h = L.horizon_cap;
        if (h > kPts) h = kPts;
        if (h > bt.T_obs - L.t_now) h = bt.T_obs - L.t_now;
        pt_rows = h > 0 ? (h + p.check_stride - 1) / p.check_stride : 0;  // <= pt_rows_max by construction
        L.pt = (float4*)(L.S + refine_pt_off(bt.NX, kPts));
    }
    const double nan = __builtin_nan("");
    double eg[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) eg[m] = bt.ego[(size_t)b * 6 + m];
    const double target_speed = bt.target_speed[b];
    double x[3], res[3], lo[3], hi[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        x[m] = fa.io.end_state[(size_t)b * 3 + m];
        res[m] = fa.io.samp_res[(size_t)b * 3 + m];
        lo[m] = fa.io.samp_min[(size_t)b * 3 + m];
        hi[m] = fa.io.samp_max[(size_t)b * 3 + m];
    }
    const double coarse_x[3] = {x[0], x[1], x[2]};
    double* cand = (double*)(smem + verdict_off + 32);  // [64][4] + {ncand, coarse cost}: the queues' bytes, free until validation
    if (wave > 0) {
        const int stid = tid - kWave;
        constexpr int kStagers = kThreads - kWave;
        for (int i = stid; i < nx; i += kStagers) L.knots[i] = gk[i];
        for (int i = stid; i < 8 * nx; i += kStagers) {
            const int r = i / nx, c = i - r * nx;
            L.coef[r * nx + c] = gc[(size_t)r * bt.NX + c];
        }
        // fp32 broad-phase table of every (checked row, obstacle) pair the collision horizon can touch, once per ego: the
        // validation loop may visit the same pair for up to 21 trajectories
        if (L.pt) {
            const int t0 = L.t_now;
            const double* scene = bt.obs_pose + (size_t)sc0 * bt.T_obs * bt.n_obs * 4;
            const double* gd = bt.obs_dims + (size_t)sc0 * bt.n_obs * 2;
            const double veh_hl = 0.5 * p.veh_l, veh_hw = 0.5 * p.veh_w;
            const double r_ego = sqrt(fma(veh_hl, veh_hl, veh_hw * veh_hw));
            for (int e = stid; e < pt_rows * bt.n_obs; e += kStagers) {
                const int r = e / bt.n_obs, j = e - r * bt.n_obs;
                const double4 ps = *(const double4*)(scene + ((size_t)(r * p.check_stride + t0) * bt.n_obs + j) * 4);
                const double hl = 0.5 * gd[2 * j], hw = 0.5 * gd[2 * j + 1];
                const double rx = ps.x - L.ox, ry = ps.y - L.oy;
                // padding: 1 cm + 2e-6 of the coordinate magnitude dwarfs the fp32 rounding of rx, ry, the pose and the sum
                const double R = r_ego + sqrt(fma(hl, hl, hw * hw)) + 1e-2 + 2e-6 * (fabs(rx) + fabs(ry));
                float R2 = (float)(R * R * (1.0 + 1e-5));
                if (ps.w == 0.0 || !(R2 >= 0.0f)) R2 = -1.0f;  // absent at this step (or NaN size): never passes
                L.pt[e] = make_float4((float)rx, (float)ry, R2, __int_as_float((r * p.check_stride) | (j << 8)));
            }
        }
#if defined(FP_RABL) && FP_RABL == 3  // timing ablation: staging only
    } else if (false) {
#else
    } else {
#endif
        // The evaluations form a dependent chain (~2.3 us each on a lone wavefront), so the cost of the CURRENT point x rides along
        // with the six probes around it (lanes >= 6 price x itself): that is the coarse winner's cost in round 0 and the previous
        // round's step afterwards - R + 1 evaluations in the chain instead of 2 R + 1.
        double coarse_cost0 = nan;
        bool have_coarse = false;
        int pending = -1;  // the lane whose trajectory (= x) still waits for its cost
        double my_x[3] = {nan, nan, nan}, my_cost = nan;  // lane c = refinement trajectory c (generation order)
        int ncand = 0;
        for (int r = 0; r < R; ++r) {
            // lanes 0..5: the six probes, lane 6: x itself (role 0: longitudinal half + recombination); lanes 8..14: their lateral halves
            const int pk = lane & 7, dim = pk >> 1;
            double xp[3] = {x[0], x[1], x[2]};
            if (pk < 6) {
                xp[dim] += (pk & 1) ? res[dim] : -res[dim];
#pragma unroll
                for (int m = 0; m < 3; ++m) xp[m] = fmin(fmax(xp[m], lo[m]), hi[m]);  // np.clip
            }
            const bool bad = lane < 6 && (!(xp[0] == xp[0]) || !(xp[1] == xp[1]) || !(xp[2] == xp[2]));
            if (__ballot(bad)) break;
            const double cp = analytic_cost_two_lanes(p, eg, target_speed, xp, (lane >> 3) & 1, kPts);  // valid in lanes 0..6 (6: the cost of x itself)
            {
                const double cx0 = __shfl(cp, 6, kWave);
                if (!have_coarse) { coarse_cost0 = cx0; have_coarse = true; }
                if (lane == pending) my_cost = cx0;
                pending = -1;
            }
            for (int k = 0; k < 6; ++k) {  // hand probe k to lane ncand + k
                const double c = __shfl(cp, k, kWave), a0 = __shfl(xp[0], k, kWave), a1 = __shfl(xp[1], k, kWave), a2 = __shfl(xp[2], k, kWave);
                if (lane == ncand + k) { my_cost = c; my_x[0] = a0; my_x[1] = a1; my_x[2] = a2; }
            }
            ncand += 6;
            double g[3], nrm2 = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const double Jl = __shfl(cp, 2 * m, kWave), Jr = __shfl(cp, 2 * m + 1, kWave);
                const double xl = __shfl(xp[m], 2 * m, kWave), xr = __shfl(xp[m], 2 * m + 1, kWave);
                g[m] = (Jr - Jl) / (xr - xl);
                nrm2 += g[m] * g[m];
            }
            const double nrm = sqrt(nrm2);
            double xn[3];
            bool nan_step = false;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                res[m] *= fa.opts.decaying_factor;  // decays in place, like the reference's aliasing of sampling_res (:282)
                xn[m] = x[m] - res[m] * g[m] / nrm;
                xn[m] = fmin(fmax(xn[m], lo[m]), hi[m]);
                nan_step |= !(xn[m] == xn[m]);
            }
            if (nan_step) break;  // zero gradient: the reference raises inside np.arange(nan); refinement stops here
            if (lane == ncand) { my_x[0] = xn[0]; my_x[1] = xn[1]; my_x[2] = xn[2]; }
            pending = ncand;  // priced with the next round's probes, or below
            ncand += 1;
            x[0] = xn[0]; x[1] = xn[1]; x[2] = xn[2];
        }
        if (!have_coarse || pending >= 0) {  // (x is the coarse winner while no round got as far as its evaluation)
            const double cl = analytic_cost(p, eg, target_speed, x, L.S, kPts);
            if (!have_coarse) coarse_cost0 = cl;
            if (lane == pending) my_cost = cl;
        }
        cand[lane * 4] = my_x[0]; cand[lane * 4 + 1] = my_x[1]; cand[lane * 4 + 2] = my_x[2]; cand[lane * 4 + 3] = my_cost;
        if (lane == 0) { cand[4 * kWave] = (double)ncand; cand[4 * kWave + 1] = coarse_cost0; }
    }
    if (wave == 0) FP_RSTAMP(1);
    __syncthreads();
    FP_RSTAMP(2);
    double my_x[3] = {cand[lane * 4], cand[lane * 4 + 1], cand[lane * 4 + 2]};  // lane c = refinement trajectory c (generation order)
    const double my_cost = cand[lane * 4 + 3];
    const int ncand = (int)cand[4 * kWave];
    const double coarse_cost = cand[4 * kWave + 1];
    __syncthreads();  // the queues take their bytes back
#if defined(FP_RABL) && FP_RABL == 1  // timing ablation: rounds + staging only
    return;
#endif
    // refined_trajs.get() in cost order (ties: generation order), :301-323.  Every wavefront holds the same candidate list;
    // the next kRefineWaves trajectories in pop order are checked SPECULATIVELY side
    // by side, one whole wavefront each, and the verdicts are then consumed in pop order exactly like the sequential loop -
    // validated / checks count only what the reference would have popped before its first collision-free trajectory.
    uint32_t* verdict = (uint32_t*)(smem + verdict_off);  // [2][kRefineWaves], double-buffered across groups
    int validated = 0, checks = 0, winner = -1;
    bool alive = lane < ncand;
    // Pop order = ascending (cost, generation index): a strict total order unless a cost is NaN, so every lane counts the
    // candidates in front of its own once (register reads, no cross-lane reduction per pop: four 6-step butterflies cost ~1.2 us
    // of every group).  With a NaN cost in the list the butterfly below decides, as it always did.
    int rank = kWave;
    const bool ranked = !__ballot(alive && !(my_cost == my_cost));
    if (ranked) {
        rank = 0;
        for (int j = 0; j < ncand; ++j) {
            const double cj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_cost), j), __builtin_amdgcn_readlane(__double2loint(my_cost), j));
            rank += (cj < my_cost || (cj == my_cost && j < lane)) ? 1 : 0;
        }
        if (!alive) rank = kWave;
    }
    bool open = true;  // (ranked pops) nothing has ended the loop yet
    for (int grp = 0; winner < 0; ++grp) {
        int pop[kRefineWaves];
        if (ranked) {
#pragma unroll
            for (int u = 0; u < kRefineWaves; ++u) {
                const unsigned long long m = __ballot(rank == grp * kRefineWaves + u);
                int bl = (open && m) ? __ffsll((long long)m) - 1 : -1;
                if (bl >= 0) {
                    const double bc = __shfl(my_cost, bl, kWave);
                    if (bc > coarse_cost) bl = -1;  // `cost > coarse cost` ends the loop (:303-304)
                }
                if (bl < 0) open = false;
                pop[u] = bl;
            }
        } else
#pragma unroll
        for (int u = 0; u < kRefineWaves; ++u) {
            double bc = alive ? my_cost : __builtin_inf();
            int bl = alive ? lane : kWave;
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) {
                const double oc = __shfl_xor(bc, off, kWave);
                const int ol = __shfl_xor(bl, off, kWave);
                if (ol < kWave && (bl >= kWave || oc < bc || (oc == bc && ol < bl))) { bc = oc; bl = ol; }
            }
            if (bl >= kWave || bc > coarse_cost) bl = -1;  // queue empty / `cost > coarse cost` ends the loop (:303-304)
            pop[u] = bl;
            if (bl < 0) alive = false;  // nothing further is ever popped
            if (lane == bl) alive = false;
        }
        if (pop[0] < 0) break;
        int mine = pop[0];
#pragma unroll
        for (int u = 1; u < kRefineWaves; ++u) mine = (wave == u) ? pop[u] : mine;
        if (mine >= 0) {
            const double cx[3] = {__shfl(my_x[0], mine, kWave), __shfl(my_x[1], mine, kWave), __shfl(my_x[2], mine, kWave)};
#if defined(FP_PHASE_STAMPS)
            const uint32_t fl = wave_traj_flags<NCH>(ka, b, eg, cx, L, nx, lane, (wave == 0 && grp == 0 && fa.io.best_traj) ? fa.io.best_traj + ((size_t)b * FP_ARR_COUNT + 15) * fa.io.traj_stride + 112 : nullptr, t_begin);
#else
            const uint32_t fl = wave_traj_flags<NCH>(ka, b, eg, cx, L, nx, lane);
#endif
            if (lane == 0) verdict[(grp & 1) * kRefineWaves + wave] = fl;
        }
        __syncthreads();
        FP_RSTAMP(3 + (grp < 6 ? grp : 6));
        bool done = false;
#pragma unroll
        for (int u = 0; u < kRefineWaves; ++u) {
            if (done) continue;
            if (pop[u] < 0) { done = true; continue; }
            const uint32_t fl = verdict[(grp & 1) * kRefineWaves + u];
            ++validated;
            if (fl & FP_FLAG_CONSTRAINTS) continue;
            ++checks;
            if (!(fl & FP_FLAG_COLLISION)) { winner = pop[u]; done = true; }
        }
        if (done) break;
#if defined(FP_RABL) && FP_RABL >= 4  // timing ablation: at most FP_RABL - 3 validation groups
        if (grp >= FP_RABL - 4) break;
#endif
    }
    if (wave == 0) {
        if (fa.io.trace && lane < R * 7) {
            double* tr = fa.io.trace + ((size_t)b * R * 7 + lane) * 4;
            const bool have = lane < ncand;
            tr[0] = have ? my_x[0] : nan; tr[1] = have ? my_x[1] : nan; tr[2] = have ? my_x[2] : nan; tr[3] = have ? my_cost : nan;
        }
        if (lane == 0) {
            int32_t* s4 = fa.io.stats + (size_t)b * 4;
            s4[1] += ncand;
            s4[2] += validated;
            s4[3] += checks;
            fa.io.best_cost[b] = coarse_cost;  // same evaluator as the refined costs
        }
        if (winner >= 0 && lane == winner) {
            fa.io.refined[b] = 1;
            fa.io.best_cost[b] = my_cost;
            double* es = fa.io.end_state + (size_t)b * 3;
            es[0] = my_x[0]; es[1] = my_x[1]; es[2] = my_x[2];
        }
    }
    if (tid == 0 && dur) dur[b] = (int)(wall_clock64() - t_begin);  // 10 ns ticks: feeds the next launches' order
    // winner epilogue (what plan() returns) on request: the series of the refined trajectory, or of the coarse winner when no
    // refined one survived.  Every wavefront holds the same candidate list; wavefront 0 writes the series (two time points per
    // lane, no barrier), the others are done.
#if defined(FP_RABL) && FP_RABL == 2  // timing ablation: no series
    return;
#endif
    if ((fa.io.best_traj || ka.has_loop) && wave == 0) {
        double fx[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) fx[m] = winner >= 0 ? __shfl(my_x[m], winner, kWave) : coarse_x[m];
        if (fa.io.best_traj) {
            KernelArgs kw = ka;
            kw.r.best_traj = fa.io.best_traj;
            kw.r.best_flags = fa.io.best_flags;
            kw.r.traj_stride = fa.io.traj_stride;
            kw.r.traj_sparse = fa.io.traj_sparse;
            FP_RSTAMP(10);
            bool long_series = false;
            if constexpr (NCH > 2) {  // more points than the one-chunk writer holds: the chunked writer (frenet_winner.h), same arithmetic per element
                const int n_pts = (fx[2] == fx[2]) ? arange_len(fx[2], p.tick_t) : 0;
                long_series = n_pts > kSeriesChunk && n_pts <= points_cap(p) && (fx[0] == fx[0]) && (fx[1] == fx[1]);
                if (long_series) winner_series_wave_long(kw, b, b, fx[0], fx[1], fx[2], lane, SplineLds{L.knots, L.coef, nx, nx});
            }
            if (!long_series) winner_series_wave(kw, b, b, true, fx[0], fx[1], fx[2], lane, SplineLds{L.knots, L.coef, nx, nx});
            FP_RSTAMP(11);
            FP_RSTAMP(0);
        }
        // fp_plan_fiss_step: the workgroup that settled the ego's trajectory hands the ego over to its next state itself (after the
        // series, which describe the trajectory from the OLD state; every other wavefront of the workgroup is done with the state,
        // and no other workgroup reads it) - no advance_kernel launch behind the pipeline
        // (the hand-over's arguments are re-read from the kernel's argument segment HERE, through a laundered pointer: as ordinary kernel
        // arguments the compiler loads the loop structure's eleven fields at the kernel's start and holds them - 47 more spilled SGPRs
        // and, through their lanes, two spilled VGPRs in a kernel that is at its register budget)
        if (ka.has_loop && lane == 0) {
            int off = (int)offsetof(FissArgs, ka);
            asm volatile("" : "+s"(off) : : "memory");  // (an offset the compiler cannot see through: the loads below depend on it and stay here)
            const KernelArgs& kl = *(const KernelArgs*)((const unsigned char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
            advance_ego_to(kl, b, fx[0], fx[1], fx[2], kl.loop);
        }
    }
}

hipError_t launch_fiss_refine(const FissArgs& fa, hipStream_t stream, int table_kb, const int* perm, int* dur)
{
    // trajectories of more than FP_FAST_POINTS points (fp_params.points_max): the four-points-per-lane instance
    const bool big = fa.ka.p.points_max > FP_FAST_POINTS;
    const int pts = big ? FP_MAX_POINTS : FP_FAST_POINTS;
    // pair table of the checked obstacle rows when it fits in a modest LDS budget (keeps >= 3 workgroups per CU)
    int pt_rows = 0;
    if (fa.ka.b.n_obs > 0) {
        const int stride = fa.ka.p.check_stride;
        int rows = (pts + stride - 1) / stride;
        const int rows_tab = (fa.ka.b.T_obs + stride - 1) / stride;
        if (rows_tab < rows) rows = rows_tab;
        if (fa.ka.b.n_obs < (1 << 23) && (long)rows * fa.ka.b.n_obs * 16 <= (long)table_kb * 1024) pt_rows = rows;
    }
    const int bytes = refine_lds_bytes(fa.ka.b.NX, pt_rows * fa.ka.b.n_obs, pts);
    FP_LDS_SLOTS(configured);
    FP_LDS_SLOTS(configured_big);
    if (big) {
        hipError_t e = ensure_dynamic_lds((const void*)fiss_refine_kernel<4>, bytes, configured_big);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fiss_refine_kernel<4>, dim3(fa.ka.b.B), dim3(kWave * kRefineWaves), bytes, stream, fa, pt_rows, perm, dur);
        return hipGetLastError();
    }
    hipError_t e = ensure_dynamic_lds((const void*)fiss_refine_kernel<2>, bytes, configured);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fiss_refine_kernel<2>, dim3(fa.ka.b.B), dim3(kWave * kRefineWaves), bytes, stream, fa, pt_rows, perm, dur);
    return hipGetLastError();
}

hipError_t launch_fiss_search(const FissArgs& fa, hipStream_t stream)
{
    if (fa.opts.kind == FP_FISS_PLUS) return launch_fissplus_search(fa, stream);  // rank-space walk (frenet_fissplus.hip)
    const int C = fa.ka.p.nd * fa.ka.p.nv * fa.ka.p.nt;
    int P = kWave;  // (at least one element per lane: the register sort of small lattices)
    while (P < C) P <<= 1;  // C <= FP_MAX_CAND = 4096: at most 64 words of rank bits, one per lane
    FP_LDS_SLOTS(configured);
    FP_LDS_SLOTS(configured_wide);
    if (P <= 4 * kWave) {  // the register sorts of one wavefront
        const int bytes = C * (8 + 8 + 2 + 2 + 1 + 1) + P * (8 + 2) + 32;
        hipError_t e = ensure_dynamic_lds((const void*)fiss_search_kernel<1>, bytes, configured);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fiss_search_kernel<1>, dim3(fa.ka.b.B), dim3(kWave), bytes, stream, fa, P);
    } else {  // four wavefronts set up and sort (no padding stored: 567 candidates -> 18 KB, eight egos per CU)
        const int bytes = C * (8 + 8 + 2 + 2 + 1 + 1) + C * (8 + 2) + 32;
        hipError_t e = ensure_dynamic_lds((const void*)fiss_search_kernel<4>, bytes, configured_wide);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fiss_search_kernel<4>, dim3(fa.ka.b.B), dim3(4 * kWave), bytes, stream, fa, P);
    }
    return hipGetLastError();
}

}  // namespace fp
