// two_opt_tabu_list.hpp -- alg_2opt_tabu WITH a tabu list (src/tabusearch.c:127-165) without reading the list per pair
// Part of the GRID engine; included by two_opt_grid.hip only (one translation unit).
//
// The reference looks at four stamps of the n(n-1)/2-entry list for every pair of every sweep (:137-149): 800 MB of
// (mostly scattered) reads per sweep at n = 10 000 for a list that holds a few hundred non-zero entries.  What those
// reads decide depends on the non-zero entries alone:
//   * a pair (a, b) is skipped iff one of (a,b), (a,a1), (b,b1), (a,b1) is LIVE (non-zero and iter - stamp <= tenure,
//     :83-92).  iter and tenure are constant within a call and a call only ever clears EXPIRED stamps, so the set of
//     live stamps -- and with it every decision -- is fixed for the call;
//   * the side effects are the lazy clears of the expired stamps the scan looks at (:87-90) and the number of pairs
//     that reach the delta expression (:150; what tsp_two_opt_stats.evals reports).
// So a sweep is split in two.  (A) The arg-min runs through the sorted sweep (k_sweep, two_opt_sweep.hpp) exactly as
// without a list; a pair that would become a lane's best is first put through the reference's own check_tenure chain
// and dropped when it is tabu -- a few hundred look-ups per sweep.  (B) tabu_side(), a few extra workgroups of the same
// launch, walks the compact list of non-zero stamps that the tabu handle maintains and reproduces the side effects in
// closed form:
//   clears   an expired stamp on a non-adjacent pair is always looked at (it is that pair's first check); one on a tour
//            edge x -> succ x only through the (a,a1) check of a pair (x, b) whose first check passes, the (b,b1) check
//            of a pair (a, x) whose first two pass, or the (a,b1) check of the pair (succ x, pred x);
//   evals    non-adjacent pairs minus the skipped ones, the latter counted once each by the FIRST live check of the
//            chain: P1 live (a,b); P2 live (a,a1) [rows of nodes whose tour edge is live]; P3 live (b,b1) [their
//            columns]; P4 live (a,b1) [two pairs per live stamp] -- O(1) per list entry, see tabu_side().
// Tours outside the sorted sweep (metrics without the bound) get the same treatment through the tiled step: k_step<..., TLIST>
// for (A), k_tabu_side as a launch of its own for (B).
// Nothing here changes a result: tests/test_gpu_tabu_list.py runs it against the oracle with dense random lists
// (stamps on tour edges, live and expired) and compares tours, counters and the whole stamp array.
#pragma once
#include "two_opt_step.hpp"

namespace tsp {

constexpr int kTabuSideBlocks = 8;   // extra workgroups of a k_sweep launch that run tabu_side()

// ---- the compact list of non-zero stamps -----------------------------------------------------------------------
// list[k] = (u, v), u < v; no duplicates; a superset of the non-zero stamps (entries whose stamp has been cleared since
// stay until the next compaction).  *count may exceed cap after a scan: the list is then unusable (dense path).

// first index of row i of the reference's triangular layout (utility.c:17-30)
__device__ __forceinline__ long long udir_row_start(long long i, long long n) { return i * n + i + 1 - (i + 1) * (i + 2) / 2; }

__device__ __forceinline__ int2 udir_pair(long long e, int n) {
    // row i holds n - 1 - i entries; start(i) = i (2n - i - 3) / 2 <= e
    const double b = 2.0 * n - 3.0;
    long long i = (long long)((b - sqrt(fmax(0.0, b * b - 8.0 * (double)e))) * 0.5);
    i = max(0ll, min(i, (long long)n - 2));
    while (i > 0 && udir_row_start(i, n) > e) --i;
    while (i < n - 2 && udir_row_start(i + 1, n) <= e) ++i;
    return make_int2((int)i, (int)(i + 1 + (e - udir_row_start(i, n))));
}

__global__ __launch_bounds__(256) void k_tabu_scan(const int *__restrict__ stamp, long long count, int n,
                                                   int2 *__restrict__ list, int cap, int *__restrict__ list_n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) {
        if (stamp[e] != 0) {
            const int k = atomicAdd(list_n, 1);
            if (k < cap) list[k] = udir_pair(e, n);
        }
    }
}

// one workgroup: entries whose stamp is zero leave the list (order is free)
__global__ __launch_bounds__(1024) void k_tabu_compact(const int *__restrict__ stamp, int n, int2 *list, int *list_n) {
    __shared__ int s_keep[1024 / 64];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = *list_n;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < m; k0 += 1024) {
        const int k = k0 + tid;
        int2 e = make_int2(0, 0);
        bool keep = false;
        if (k < m) { e = list[k]; keep = stamp[udir_pos(e.x, e.y, n)] != 0; }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_keep[wave] = __popcll(bal);
        __syncthreads();   // every read of this chunk is done: writes below land at indices <= k0 + tid
        int before = s_base, total = 0;
        for (int w = 0; w < 1024 / 64; ++w) { const int c = s_keep[w]; before += (w < wave) ? c : 0; total += c; }
        if (keep) list[before + __popcll(bal & ((1ull << lane) - 1ull))] = e;
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) *list_n = s_base;
}

// after a run: evals counted every non-adjacent pair of every sweep; take the skipped ones off (tabusearch.c:150)
// (side[1 .. 4]: live tour edges of a sweep whose C(|F|, 2) term the CLUSTER engine had not taken off yet -- it does so two
// sweeps later, so the last two sweeps' are left; the GRID engine leaves them at zero)
__global__ void k_tabu_fix_evals(TourState *st, unsigned long long *side) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // read-and-zero as one returning atomic each: as plain code the compiler loads the four words with one scalar load
        // and issues the zeroing vector stores before that load has returned (seen in the ISA: the stores won the race)
        long long skipped = (long long)atomicExch(side, 0ull);
        for (int p = 1; p <= kTabuSideSlots; ++p) {
            const long long f = (long long)atomicExch(side + p, 0ull);
            skipped -= f * (f - 1) / 2;
        }
        st->evals -= skipped;
    }
}

// ---- (B): side effects of one sweep ------------------------------------------------------------------------------
struct TabuTour {
    const int *order, *pos;
    int n;
    __device__ __forceinline__ int succ(int v) const { int p = pos[v] + 1; if (p == n) p = 0; return order[p]; }
    __device__ __forceinline__ int pred(int v) const { int p = pos[v]; p = p == 0 ? n - 1 : p - 1; return order[p]; }
};

struct TabuView {
    int *stamp;
    int n, iter, tenure;
    // live = what check_tenure answers 1 for (tabusearch.c:83-92); an expired stamp another thread is clearing reads as
    // expired or as zero, not live either way
    __device__ __forceinline__ bool live_value(int s) const { return s != 0 && !(iter - s > tenure); }
    __device__ __forceinline__ bool live(int x, int y) const { return live_value(stamp[udir_pos(x, y, n)]); }
};

// Would the reference's scan of this tour look at the expired stamp of the tour edge x -> y (y = succ x)?
__device__ inline bool tabu_edge_looked_at(const TabuTour &t, const TabuView &tv, int x, int y) {
    const int n = t.n, px = t.pred(x);
    // as (a,a1) of a pair (x, b), b > x, not adjacent (b != a1, b1 != a): reached when the pair's own stamp is not live
    for (int b = x + 1; b < n; ++b)
        if (b != y && b != px && !tv.live(x, b)) return true;
    // as (b,b1) of a pair (a, x), a < x: reached when (a,x) and (a,a1) are not live
    for (int a = 0; a < x; ++a)
        if (a != px && a != y && !tv.live(a, x) && !tv.live(a, t.succ(a))) return true;
    // as (a,b1) of the pair (a, b) = (y, pred x)
    if (y < px && px != t.succ(y) && !tv.live(y, px) && !tv.live(y, t.succ(y)) && !tv.live(px, x)) return true;
    return false;
}

// One sweep's side effects.  Called by the nsb workgroups past the sweep's own; side[0] = skipped pairs of the run,
// side[1] / side[2] = this sweep's count of live tour edges and the arrival ticket (both zero between launches).
// With E = the live stamps on non-adjacent pairs (u < v), F = the nodes whose tour edge x -> succ x is live, and
// f(v) = [v in F], the skipped pairs of a sweep are
//   P1            |E|
//   P2 + P3       |F| (n - 3)  -  sum_E f(u)  -  sum_E f(v)  +  sum_E f(u) f(v)  -  C(|F|, 2)  +  |{x in F : succ x in F}|
//                 (rows and columns of the F nodes without the pairs an earlier check of the chain already skipped:
//                 members of E in those rows / columns, pairs of two F nodes -- the adjacent ones were never pairs)
//   P4            per live stamp (a, w) and orientation: the pair (a, pred w) if it is a pair (a < pred w, not adjacent)
//                 whose three earlier checks are not live
// scratch: 2 * (NT / 64) ints of LDS, 8-byte aligned.
template <int NT>
__device__ inline void tabu_side(int *scratch, const TabuTour t, const TabuView tv, const int2 *__restrict__ list, int m, int sb,
                                 int nsb, unsigned long long *side) {
    long long *s_sum = reinterpret_cast<long long *>(scratch);
    const int tid = threadIdx.x, n = t.n;
    long long cnt = 0;
    int nf = 0;
    for (int k = sb * NT + tid; k < m; k += nsb * NT) {
        const int2 e = list[k];
        const int u = e.x, v = e.y;
        int *sp = tv.stamp + udir_pos(u, v, n);
        const int s = *sp;
        if (s == 0) continue;   // cleared since it joined the list
        const int su = t.succ(u), sv = t.succ(v);
        const bool uv = su == v, vu = sv == u;   // the stamped edge is a tour edge u -> v / v -> u
        if (!tv.live_value(s)) {
            // (tabusearch.c:87-90) cleared by the first look; a non-adjacent pair's own stamp is its first check
            if (!uv && !vu) *sp = 0;
            else if (tabu_edge_looked_at(t, tv, uv ? u : v, uv ? v : u)) *sp = 0;
            continue;
        }
        if (!uv && !vu) {
            const int fu = tv.live(u, su) ? 1 : 0, fv = tv.live(v, sv) ? 1 : 0;
            cnt += 1 - fu - fv + fu * fv;
        } else {
            nf += 1;
            cnt += n - 3;
            const int y = uv ? v : u, sy = uv ? sv : su;
            if (tv.live(y, sy)) cnt += 1;
        }
        // P4: the stamp is (a, b1) of the pair (a, b = pred w), {a, w} = {u, v}
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int a = o ? v : u, w = o ? u : v, sa = o ? sv : su;
            const int b = t.pred(w);
            if (b != a && a < b && b != sa && !tv.live(a, b) && !tv.live(a, sa) && !tv.live(b, w)) cnt += 1;
        }
    }
    const long long tot = block_sum<long long>(cnt, s_sum);
    const long long totf = block_sum<long long>((long long)nf, s_sum);
    if (tid == 0) {
        if (tot) atomicAdd(side, (unsigned long long)tot);
        if (totf) atomicAdd(side + 1, (unsigned long long)totf);
        __threadfence();
        if (atomicAdd(side + 2, 1ull) == (unsigned long long)(nsb - 1)) {   // the last of the side workgroups
            __threadfence();
            const long long f = (long long)atomicAdd(side + 1, 0ull);
            atomicAdd(side, (unsigned long long)(-(f * (f - 1) / 2)));
            atomicExch(side + 1, 0ull);
            atomicExch(side + 2, 0ull);
        }
    }
}

// The side effects as a launch of their own, in front of a tiled step (k_step<..., TLIST>): tours outside the sorted sweep.
__global__ __launch_bounds__(kScanThreads) void k_tabu_side(const int *__restrict__ order, const int *__restrict__ pos, int n,
                                                            const TourState *__restrict__ st, int *stamp, const int2 *__restrict__ list,
                                                            const int *__restrict__ list_n, int list_cap, int iter, int tenure,
                                                            unsigned long long *side) {
    __shared__ long long s_scratch[kScanThreads / 64];
    if (st->done) return;
    const TabuTour tt{order, pos, n};
    const TabuView tv{stamp, n, iter, tenure};
    tabu_side<kScanThreads>(reinterpret_cast<int *>(s_scratch), tt, tv, list, min(*list_n, list_cap), (int)blockIdx.x, (int)gridDim.x, side);
}

}  // namespace tsp
