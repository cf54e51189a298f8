"""Host-side search logic of the FOP+/FISS/FISS+ planners over GPU-computed dense tables.

The reference generates candidates lazily while it walks the (d, v, t) index grid.  A candidate's
cost_final, its constraint/collision outcome and its cost_est depend only on its index (SURVEY.md
section 3.4), so the walks can run over three dense [nd, nv, nt] tables that the GPU produced in one
pass:  J = cost_final,  F = flag word,  E = cost_est.  Every function here reproduces the reference's
visiting order and its four Stats counters; nothing here evaluates a trajectory.

Tie rule: the reference keeps (cost, numpy index) tuples in its queues, so an exact cost tie raises
ValueError there (fiss_planner.py:136,229).  Here ties resolve to the lower raster index (i_d, i_v, i_t).
"""
from __future__ import annotations

import heapq

import numpy as np

FLAG_SPEED, FLAG_ACCEL, FLAG_COLLISION = 1, 2, 4
FLAG_CONSTRAINTS = 1 | 2 | 16 | 32 | 64   # speed, acceleration + the optional curvature checks (include/frenet_gpu.h FP_FLAG_CONSTRAINTS)
FLAG_INFEASIBLE = FLAG_CONSTRAINTS | FLAG_COLLISION


def tables_to_dvt(cost_flat: np.ndarray, flags_flat: np.ndarray, nd: int, nv: int, nt: int):
    """Dense kernel output is in FOP order (i_d, i_T, i_v) (frenet_optimal_planner.py:75-100);
    FISS indexes (i_d, i_v, i_t) (fiss_planner.py:47-69)."""
    J = np.ascontiguousarray(cost_flat.reshape(nd, nt, nv).transpose(0, 2, 1))
    F = np.ascontiguousarray(flags_flat.reshape(nd, nt, nv).transpose(0, 2, 1))
    return J, F


def cost_est_table(d_samples, v_samples, t_samples, samp_min, samp_max, prev_best_idx=None, w_heuristic=10.0):
    """sample_end_frenet_states (fiss_planner.py:33-99): estimated cost of every lattice end state."""
    nd, nv, nt = len(d_samples), len(v_samples), len(t_samples)
    max_sqr_dist = nd ** 2 + nv ** 2 + nt ** 2
    lat_norm = max(samp_min[0] ** 2, samp_max[0] ** 2)
    est_lat = np.asarray(d_samples) ** 2 / lat_norm
    est_speed = (samp_max[1] - np.asarray(v_samples)) ** 2 / (samp_max[1] - samp_min[1]) ** 2
    est_time = 1.0 - (np.asarray(t_samples) - samp_min[2]) / (samp_max[2] - samp_min[2])
    est = (est_lat[:, None, None] + est_time[None, None, :]) + est_speed[None, :, None]
    if prev_best_idx is not None:
        i, j, k = np.meshgrid(np.arange(nd), np.arange(nv), np.arange(nt), indexing="ij")
        heu = (i - prev_best_idx[0]) ** 2 + (j - prev_best_idx[1]) ** 2 + (k - prev_best_idx[2]) ** 2
        est = est + w_heuristic * heu / max_sqr_dist
    return est


class _Walk:
    """State shared by the FISS and FISS+ walks: which samples are generated, the candidate queue, Stats."""

    def __init__(self, J, F, E):
        self.J, self.F, self.E = J, F, E
        self.sizes = J.shape
        self.generated = np.zeros(J.shape, dtype=bool)
        self.queue = []  # heap of (cost, raster index)
        self.order = []  # index triples in generation order = the reference's trajs_per_timestep (fiss_planner.py:131)
        self.num_iter = self.num_generated = self.num_validated = self.num_collision_checks = 0

    def raster(self, idx):
        return (idx[0] * self.sizes[1] + idx[1]) * self.sizes[2] + idx[2]

    def unraster(self, q):
        return (q // (self.sizes[1] * self.sizes[2]), (q // self.sizes[2]) % self.sizes[1], q % self.sizes[2])

    def generate(self, idx):
        """generate_trajectory (fiss_planner.py:101-138) -> (is_new, cost_final)."""
        idx = tuple(int(v) for v in idx)
        if self.generated[idx]:
            return False, self.J[idx]
        self.num_generated += 1
        self.generated[idx] = True
        self.order.append(idx)
        heapq.heappush(self.queue, (float(self.J[idx]), self.raster(idx)))
        return True, self.J[idx]

    def initial_guess(self):
        """find_initial_guess (fiss_planner.py:140-150): `<=` keeps the LAST minimum of cost_est."""
        cand = np.where(self.generated, np.inf, self.E)
        if self.generated.all():
            return None
        flat = cand.ravel()
        m = flat.min()
        if not np.isfinite(m) and not (flat <= np.inf).any():
            return None
        q = int(np.nonzero(flat == m)[0][-1])
        return self.unraster(q)

    def validate_head(self):
        """Pop the cheapest generated candidate and check it (fiss_planner.py:229-258).
        Returns (idx, ok)."""
        cost, q = heapq.heappop(self.queue)
        idx = self.unraster(q)
        self.num_validated += 1
        f = int(self.F[idx])
        if f & FLAG_CONSTRAINTS:
            return idx, False
        self.num_collision_checks += 1
        return idx, not (f & FLAG_COLLISION)

    @property
    def stats(self):
        return (self.num_iter, self.num_generated, self.num_validated, self.num_collision_checks)


def fiss_search(J, F, E, generated_order=None):
    """FissPlanner.plan coarse search (fiss_planner.py:190-270) -> (best idx triple or None, stats).
    generated_order: optional list that receives the index triples in generation order."""
    w = _Walk(J, F, E)
    if generated_order is not None:
        w.order = generated_order
    sizes = w.sizes
    while True:
        w.num_iter += 1
        if not w.queue:
            idx = w.initial_guess()
            if idx is None:
                return None, w.stats
        else:
            idx = w.unraster(w.queue[0][1])
        # explore_next_sample (:174-188) until it lands on a generated sample
        idx = list(idx)
        while not w.generated[tuple(idx)]:
            _, cost_center = w.generate(idx)  # find_gradients (:152-172)
            grad = [0.0, 0.0, 0.0]
            for dim in range(3):
                nb = list(idx)
                if idx[dim] < sizes[dim] - 1:
                    nb[dim] += 1
                    _, c = w.generate(nb)
                    grad[dim] = c - cost_center
                    if grad[dim] >= 0 and idx[dim] == 0:
                        grad[dim] = 0.0
                else:
                    nb[dim] -= 1
                    if nb[dim] < 0:
                        nb[dim] = sizes[dim] - 1  # python negative index on a size-1 axis
                    _, c = w.generate(nb)
                    grad[dim] = cost_center - c
                    if grad[dim] <= 0 and idx[dim] == sizes[dim] - 1:
                        grad[dim] = 0.0
            for dim in range(3):
                idx[dim] += -1 if grad[dim] > 0.0 else +1
                idx[dim] = min(max(idx[dim], 0), sizes[dim] - 1)
        if not w.queue:
            return None, w.stats
        cand, ok = w.validate_head()
        if ok:
            return cand, w.stats


def fissplus_search(J, F, E, generated_order=None):
    """FissPlusPlanner.plan coarse search (fiss_plus_planner.py:80-148) -> (best idx triple or None, stats).
    generated_order: optional list that receives the index triples in generation order."""
    w = _Walk(J, F, E)
    if generated_order is not None:
        w.order = generated_order
    sizes = w.sizes
    frontier = []
    while True:
        w.num_iter += 1
        if not w.queue:
            idx = w.initial_guess()
            if idx is None:
                return None, w.stats
        else:
            idx = w.unraster(w.queue[0][1])
        while True:
            # explore_neighbors (:30-59)
            _, cost_center = w.generate(idx)
            for dim in range(3):
                for step in (-1, +1):
                    n = idx[dim] + step
                    if n < 0 or n > sizes[dim] - 1:
                        continue
                    nb = list(idx)
                    nb[dim] = n
                    is_new, c = w.generate(nb)
                    if is_new and c <= cost_center:
                        heapq.heappush(frontier, (float(c), w.raster(nb)))
            if not frontier:
                break
            _, q = heapq.heappop(frontier)
            idx = w.unraster(q)
        if not w.queue:
            return None, w.stats
        cand, ok = w.validate_head()
        if ok:
            return cand, w.stats


class _ByCost:
    """Heap entry ordered by cost only, like FrenetTrajectory objects in the reference's PriorityQueue."""
    __slots__ = ("cost", "idx")

    def __init__(self, cost, idx):
        self.cost, self.idx = cost, idx

    def __lt__(self, other):
        return self.cost < other.cost


def fopplus_search(cost_flat, flags_flat):
    """FopPlusPlanner.plan (fop_plus_planner.py:16-41): validate in heapq order, stop at the first survivor.
    -> (flat index or None, stats).  heapq on cost-only keys reproduces CPython's order on exact ties."""
    heap = []
    for i, c in enumerate(cost_flat):
        heapq.heappush(heap, _ByCost(float(c), i))
    popped = 0
    while heap:
        popped += 1
        it = heapq.heappop(heap)
        if not (int(flags_flat[it.idx]) & FLAG_INFEASIBLE):
            return it.idx, (popped, len(cost_flat), popped, popped)
    return None, (popped, len(cost_flat), popped, popped)


def refine_step(J_l, J_r, x_l, x_r, x, res, decay, samp_min, samp_max):
    """The arithmetic of one gradient_decent round (fiss_plus_planner.py:262-271) after the six probes:
    returns (x_new_clipped, res_new) or (None, res_new) when the step is undefined (zero gradient)."""
    d_J = np.asarray(J_r) - np.asarray(J_l)
    d_x = np.array([x_r[m][m] - x_l[m][m] for m in range(3)])
    with np.errstate(divide="ignore", invalid="ignore"):
        grad = d_J / d_x
        res_new = np.asarray(res) * decay
        x_new = np.asarray(x) - res_new * grad / np.linalg.norm(grad)
    if np.isnan(x_new).any():
        return None, res_new
    return np.clip(x_new, samp_min, samp_max), res_new
