// two_opt_grid.hip -- GRID engine: 2-opt on tours resident in HBM, many workgroups per tour.
//
// Tour state in HBM is two int arrays per tour, order[p] (node at tour position p) and pos[v]
// (position of node v); succ(v) = order[pos[v]+1].  Coordinates are per instance.
//
// One step of a descent = one launch (two for the sorted sweep); the host only queues steps and polls `done`,
// every decision of the reference's loops is taken on the device, through a per-tour control block.  Four kernels
// implement a step (headers of the same name stem):
//   two_opt_tiled.hpp  k_step      every pair of the scanned range visited, tile by tile; the block that arrives
//                                  last picks the winner and reverses the segment.  Tabu runs, non-sqrt metrics,
//                                  best improvement below TSP_SORTED_MIN_N nodes, first improvement as TSP_FIRST_V1.
//   two_opt_sweep.hpp  k_move_recs + k_sweep   best improvement on sqrt metrics: nodes ranked along a Hilbert
//                                  curve, whole 64 x 64 blocks of pairs decided by the box form of the new-edge bound.
//   two_opt_first.hpp  k_first     first improvement: small fixed grid, moves carried out of place by the next launch.
//   two_opt_exh.hpp    k_move_pos + k_exh      best improvement with EVERY delta expression executed (TSP_NO_FILTER=1 on the
//                                  integer-coordinate metrics; bench.py's timed region): tour-position order, one new
//                                  distance per pair.
//   two_opt_step.hpp               what they share: arguments, in-launch hand-off, MoveView, apply_step.
// The two selection rules:
//   FIRST  = alg_2opt       (src/heuristics.c:438-502): first improving pair after the cursor in
//            (i<j) order, applied at once, scan resumes right after it; stop after a sweep that
//            did not lower obj_best (:492).
//   BEST   = alg_2opt_tabu  (src/tabusearch.c:107-178): arg-min delta over the whole sweep, strict
//            '<' so ties go to the first pair (:151); stop when the minimum is >= 0 (:158); cost
//            recomputed as a sum over edges in node order (:168-172).
//
// The pair (i,j) always denotes removing (i,succ i) and (j,succ j) and reversing the FORWARD path
// succ(i)..j (src/utility.c:708-717): that path is the cyclic position range pos[i]+1 .. pos[j],
// so reversing exactly that range keeps succ() identical to the reference's after every move.
#include "two_opt_exh.hpp"
#include "two_opt_first.hpp"
#include "two_opt_step.hpp"
#include "two_opt_sweep.hpp"
#include "two_opt_tiled.hpp"

#include <algorithm>
#include <time.h>

#pragma clang fp contract(off)
namespace tsp {

__global__ void k_stamp_scatter(int *__restrict__ stamp, const int *__restrict__ idx, const int *__restrict__ val,
                                int count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) stamp[idx[t]] = val[t];
}
__global__ void k_stamp_gather(const int *__restrict__ stamp, const int *__restrict__ idx, int *__restrict__ val,
                               int count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) val[t] = stamp[idx[t]];
}


// ---- drivers on resident tours: the state stays on the device between the steps of tabu() / HEU_VNS ----

// A new alg_2opt / alg_2opt_tabu call on the tour that is already there: cursor at the start of a sweep, obj_best as it
// stands (FIRST adds deltas to it, heuristics.c:442), nothing pending.  The executed-work and sweep counters run on.
__global__ void k_rearm(TourState *states, int B, int first_chunk) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    TourState *st = states + b;
    st->ci = 0; st->cj = 0; st->chunk_rows = first_chunk; st->done = 0;
    st->seen_cost = st->obj;
    st->parity = 0; st->pending = 0;
}

// One trial of tabu()'s kick (src/tabusearch.c:262-295) on tour 0 of the handle: the host has drawn a and b.  Rejected
// without touching the stamps when the edges share a node (:271-273); else the four check_tenure calls of the && chain in
// the reference's order, lazy clears included (:83-92, :282-285); if every edge is free the 2-exchange is carried out
// (edges[a].j = b, edges[a1].j = b1, reverse_path(b, a1): positions pos[a]+1 .. pos[b] reversed) and the two removed edges
// are stamped with iter (:306-309 -- the policy update in between does not read them).  result = {accepted, a1, b1, 0}.
// list != nullptr: the handle's list of non-zero stamps is current; the two stamped edges join it unless they are in it
// already (as entries whose stamp had been cleared).
__device__ __forceinline__ void tabu_kick_body(int *order, int *pos, int *stamp, int n, int a, int b, int iter, int tenure, int *result,
                                               int2 *list, int *list_n, int list_cap) {
    __shared__ int s_acc, s_pa, s_pb, s_a1, s_b1, s_have[2];
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int pa = pos[a], pb = pos[b];
        const int a1 = order[pa + 1 == n ? 0 : pa + 1], b1 = order[pb + 1 == n ? 0 : pb + 1];
        int acc = 0;
        if (!(a == b || a1 == b || b1 == a)) {
            acc = !stamp_is_tabu(stamp + udir_pos(a, a1, n), iter, tenure) && !stamp_is_tabu(stamp + udir_pos(b, b1, n), iter, tenure) &&
                  !stamp_is_tabu(stamp + udir_pos(a, b, n), iter, tenure) && !stamp_is_tabu(stamp + udir_pos(a1, b1, n), iter, tenure);
            if (acc) { stamp[udir_pos(a, a1, n)] = iter; stamp[udir_pos(b, b1, n)] = iter; }
        }
        result[0] = acc; result[1] = a1; result[2] = b1; result[3] = 0;
        s_acc = acc; s_pa = pa; s_pb = pb; s_a1 = a1; s_b1 = b1; s_have[0] = 0; s_have[1] = 0;
    }
    __syncthreads();
    if (!s_acc) return;
    if (list && iter != 0) {   // a stamp of value 0 is no entry
        const int e0x = min(a, s_a1), e0y = max(a, s_a1), e1x = min(b, s_b1), e1y = max(b, s_b1);
        const int m = min(*list_n, list_cap);
        for (int k = tid; k < m; k += kApplyThreads) {
            const int2 e = list[k];
            if (e.x == e0x && e.y == e0y) s_have[0] = 1;
            if (e.x == e1x && e.y == e1y) s_have[1] = 1;
        }
        __syncthreads();
        if (tid == 0) {
            int k = *list_n;   // past list_cap the count keeps running and the host stops using the list
            if (!s_have[0]) { if (k < list_cap) list[k] = make_int2(e0x, e0y); ++k; }
            if (!s_have[1] && !(e0x == e1x && e0y == e1y)) { if (k < list_cap) list[k] = make_int2(e1x, e1y); ++k; }
            *list_n = k;
        }
    }
    const int pa = s_pa, pb = s_pb;
    int L = pb - pa; if (L < 0) L += n;
    for (int t = tid; t < (L >> 1); t += kApplyThreads) {
        int p = pa + 1 + t; if (p >= n) p -= n;
        int q = pb - t; if (q < 0) q += n;
        const int u = order[p], w = order[q];
        order[p] = w; order[q] = u;
        pos[w] = p; pos[u] = q;
    }
}
__global__ __launch_bounds__(kApplyThreads) void k_tabu_kick(int *order, int *pos, int *stamp, int n, int a, int b, int iter,
                                                             int tenure, int *result, int2 *list, int *list_n, int list_cap) {
    tabu_kick_body(order, pos, stamp, n, a, b, iter, tenure, result, list, list_n, list_cap);
}

// The tail of one tabu() iteration, queued right behind the CLUSTER launch of its alg_2opt_tabu and decided here: when that launch
// finished the descent (done, no give-up), the incumbent is compared with the finished tour's cost (tabusearch.c:241-249; the host
// passes the incumbent's cost), the tour is kept as the new incumbent if it is better (order -> snap), and the kick's trial runs as
// k_tabu_kick would (:262-309).  result[4] = 1: all of that has happened (result[5]: the incumbent improved); 0: nothing was
// touched -- the host goes the two-wait way.
__global__ __launch_bounds__(kApplyThreads) void k_tabu_post(const TourState *state, const int *err, int *order, int *pos, int *stamp, int n,
                                                             int a, int b, int iter, int tenure, int *result, int2 *list, int *list_n,
                                                             int list_cap, double best_obj, int *snap) {
    __shared__ int s_go, s_better;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int go = state->done && !(err && *err);
        s_go = go; s_better = go && state->obj < best_obj;
        result[4] = go; result[5] = s_better; result[6] = 0; result[7] = 0;
        if (!go) { result[0] = 0; result[1] = 0; result[2] = 0; result[3] = 0; }
    }
    __syncthreads();
    if (!s_go) return;
    if (s_better) {
        for (int p = tid; p < n; p += kApplyThreads) snap[p] = order[p];
        __syncthreads();
    }
    tabu_kick_body(order, pos, stamp, n, a, b, iter, tenure, result, list, list_n, list_cap);
}

// ---- K iterations of tabu() per wait for the device ------------------------------------------------------------------------
// The same tail for launch k of a chain of iterations queued without a wait in between (tsp_dev_tours::cl_chain).  chain[0] is
// the stop word: once an iteration could not be completed on the device -- its descent did not finish in its launch, the
// exchange gave up, or the kick's trial was rejected (the host must draw again) -- every later launch of the chain is a no-op:
// nobody re-arms the control block, which says `done`.  The incumbent's cost lives in chain[2..3] (double).
// res = {accepted, a1, b1, 0, ran, improved, why-not (1 descent unfinished / give-up), 0, cost (double)}.
// The kernel also does what k_tabu_fix_evals does after a run (the skipped pairs come off the evaluation count, the side words
// go back to zero) and, when the iteration is complete and another one follows, the re-arm for it (k_rearm's stores): one
// launch between two CLUSTER launches instead of three.
__global__ __launch_bounds__(kApplyThreads) void k_tabu_post_chain(TourState *state, const int *err, int *order, int *pos, int *stamp, int n,
                                                                   int a, int b, int iter, int tenure, int *chain, int k, int2 *list,
                                                                   int *list_n, int list_cap, int *snap, unsigned long long *side,
                                                                   int rearm_chunk /* > 0: another iteration follows */) {
    __shared__ int s_go, s_better;
    const int tid = threadIdx.x;
    int *res = chain + 4 + 10 * k;
    if (tid == 0) {
        double *best = reinterpret_cast<double *>(chain + 2);
        int go = 0, better = 0, why = 0;
        if (!chain[0]) {
            {   // k_tabu_fix_evals: read-and-zero as returning atomics (see there)
                long long skipped = (long long)atomicExch(side, 0ull);
                for (int p = 1; p <= kTabuSideSlots; ++p) {
                    const long long f = (long long)atomicExch(side + p, 0ull);
                    skipped -= f * (f - 1) / 2;
                }
                state->evals -= skipped;
            }
            go = state->done && !(err && *err);
            if (!go) { why = 1; chain[0] = 1; state->done = 1; }   // the host finishes this iteration its own way
            else if (state->obj < *best) { better = 1; *best = state->obj; }
        }
        s_go = go; s_better = better;
        res[0] = 0; res[1] = 0; res[2] = 0; res[3] = 0; res[4] = go; res[5] = better; res[6] = why; res[7] = 0;
        *reinterpret_cast<double *>(res + 8) = go ? state->obj : 0.0;
    }
    __syncthreads();
    if (!s_go) return;
    if (s_better) {
        for (int p = tid; p < n; p += kApplyThreads) snap[p] = order[p];
        __syncthreads();
    }
    tabu_kick_body(order, pos, stamp, n, a, b, iter, tenure, res, list, list_n, list_cap);
    if (tid == 0) {
        if (!res[0]) chain[0] = 1;   // rejected: the host draws the next trial (tabusearch.c:262-287); done stays set
        else if (rearm_chunk > 0) {  // the next iteration's alg_2opt_tabu starts on the kicked tour (k_rearm)
            state->ci = 0; state->cj = 0; state->chunk_rows = rearm_chunk; state->done = 0;
            state->seen_cost = state->obj;
            state->parity = 0; state->pending = 0;
        }
    }
}

// HEU_VNS's kick (src/vns.c:11-100) on tour 0: with tour[] the walk from node 0, positions p1 < p2 < p3 (host draws),
// the new tour is tour[0..p1] tour[p2+1..p3] tour[p1+1..p2] tour[p3+1..n-1] (:60-62: a->d, e->b, c->f; the reference reads
// tour[n] for f when p3 == n-1, here the walk closes on tour[0]).  Out of place into the second copy, node 0 at position 0.
__global__ void k_vns_kick(const int *__restrict__ order, const int *__restrict__ pos, int *__restrict__ order2,
                           int *__restrict__ pos2, int n, int p1, int p2, int p3) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int z = pos[0];
    const int len2 = p3 - p2, len1 = p2 - p1;
    int src;
    if (k <= p1) src = k;
    else if (k <= p1 + len2) src = p2 + 1 + (k - p1 - 1);
    else if (k <= p1 + len2 + len1) src = p1 + 1 + (k - p1 - len2 - 1);
    else src = k;
    int q = z + src; if (q >= n) q -= n;
    const int v = order[q];
    order2[k] = v;
    pos2[v] = k;
}

}  // namespace tsp

// =============================================================================================
// host side
// =============================================================================================
using namespace tsp;

namespace {

constexpr int kBestRJ = 2;
constexpr int kFirstRJ = 1;

template <int MODE>
dim3 scan_grid(const tsp_dev_tours *t) {
    const int n = t->n;
    const int RJ = MODE == TSP_2OPT_BEST ? kBestRJ : kFirstRJ;
    const int gx = (n + kScanThreads * RJ - 1) / (kScanThreads * RJ);
    int gy;
    if (MODE == TSP_2OPT_BEST) gy = (n - 1 + t->best_rows_per_block - 1) / t->best_rows_per_block;
    else gy = (std::min(t->first_max_rows, n - 1) + t->first_rows_per_block - 1) / t->first_rows_per_block;
    return dim3(gx, gy, t->B);
}


// BEST sweeps of this handle go through k_move_recs + k_sweep (no tabu list, metric with the new-edge bound)
bool sorted_sweep_possible(const tsp_dev_tours *t) {
    return t->inst->d_sperm && t->d_gmax && t->d_order2 && t->inst->prune_margin < 1e299 &&
           t->inst->ng <= 32768;   // k_sweep packs (r, c) into one int and counts group pairs (ng (ng + 1) / 2) in an int
}
bool sorted_sweep(const tsp_dev_tours *t) { return sorted_sweep_possible(t) && t->n >= t->sorted_min_n; }

// This step of this run goes through k_move_recs + k_sweep (with a list: only when tabu_list_prepare said so)
bool sorted_run(const tsp_dev_tours *t, int mode, const tsp_dev_tabu *tabu) {
    if (mode != TSP_2OPT_BEST) return false;
    return tabu ? (t->tabu_list_run && sorted_sweep_possible(t) && t->n >= 8) : sorted_sweep(t);
}

// BEST sweeps of this handle execute every delta expression through k_move_pos + k_exh (bounds off, integer-coordinate metric)
bool exh_run(const tsp_dev_tours *t, int mode, const tsp_dev_tabu *tabu) {
    return mode == TSP_2OPT_BEST && !tabu && t->exh_blocks > 0 && t->d_pxy && t->inst->filter_margin > 1e299 &&
           (t->inst->wtype == WT_EUC_2D_ICOORD || t->inst->wtype == WT_CEIL_2D_ICOORD || t->inst->wtype == WT_ATT_ICOORD);
}

constexpr long long kTabuListMax = 16384;   // more non-zero stamps than this: the dense scan (k_step<TABU>)
// Entries whose stamp has been cleared stay in the list until it is compacted, and every sweep of a run with a list passes over
// them (CLUSTER engine: an entry per thread, the cluster waits for the slowest).  The list is compacted once this many entries
// have been appended since the last time (2 048 until round 4: at tenure 200 three entries in four were dead ones).
constexpr long long kTabuCompactSlack = 256;

// Before a run with a list: bring the handle's list of non-zero stamps up to date (a scan after the host wrote
// stamps; a compaction once enough cleared entries have piled up) and say whether the run can work from it.
int tabu_list_prepare(tsp_dev_tours *t, tsp_dev_tabu *tb, bool *usable, long long room = 0 /* entries the caller is about to append */) {
    *usable = false;
    // any size and any metric: the alternative reads four stamps per pair (60 us per sweep at n = 299 against 19); tours outside
    // the sorted sweep (metrics without the bound) take the tiled step with the side effects as a launch of their own
    if (t->B != 1 || !tb->d_list || t->n < 4) return TSP_OK;
    hipStream_t s = t->inst->ctx->stream;
    bool readback = false;
    if (!tb->list_valid) {
        TSP_HIP_TRY(hipMemsetAsync(tb->d_list_n, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_tabu_scan, dim3(2048), dim3(256), 0, s, tb->d_stamp, tb->count, t->n, tb->d_list, tb->list_cap, tb->d_list_n);
        readback = true;
    } else if (tb->list_ub + room > tb->list_compact_at) {
        hipLaunchKernelGGL(k_tabu_compact, dim3(1), dim3(1024), 0, s, tb->d_stamp, t->n, tb->d_list, tb->d_list_n);
        readback = true;
    }
    if (readback) {
        TSP_HIP_TRY(hipMemcpyAsync(tb->h_list_n, tb->d_list_n, sizeof(int), hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
        TSP_HIP_TRY(hipGetLastError());
        const long long m = *tb->h_list_n;
        tb->list_valid = m <= tb->list_cap;   // an overflowing scan leaves no list: the next run scans again
        tb->list_ub = m;
        tb->list_compact_at = m + std::max(kTabuCompactSlack, room);
    }
    *usable = tb->list_valid && tb->list_ub <= kTabuListMax;
    return TSP_OK;
}

// After sorted sweeps: pending move carried out, tour back in the first copy of order/pos.
void launch_flush(tsp_dev_tours *t) {
    if (!t->d_order2) return;
    hipStream_t s = t->inst->ctx->stream;
    const dim3 g((t->n + 255) / 256, t->B);
    hipLaunchKernelGGL(k_flush_move, g, dim3(256), 0, s, t->d_order, t->d_pos, t->d_order2, t->d_pos2, t->d_state, t->n);
    hipLaunchKernelGGL(k_flush_copy, g, dim3(256), 0, s, t->d_order, t->d_pos, t->d_order2, t->d_pos2, t->d_state, t->n);
    hipLaunchKernelGGL(k_flush_state, dim3((t->B + 255) / 256), dim3(256), 0, s, t->d_state, t->B);
}

StepArgs make_args(tsp_dev_tours *t, int mode, tsp_dev_tabu *tabu, int iter, int tenure) {
    StepArgs a;
    a.coord = t->inst->d_coord; a.orders = t->d_order; a.poss = t->d_pos; a.states = t->d_state;
    a.partials = t->d_partial; a.tickets = t->d_ticket; a.slot_evals = t->d_slot_evals;
    a.row_tickets = t->d_row_ticket; a.row_slots = t->d_row_slot; a.row_evals = t->d_row_evals;
    a.max_tile_rows = t->max_tile_rows;
    a.tabu = tabu ? tabu->d_stamp : nullptr;
    // materialising the records costs a launch: worth it once a sweep has thousands of tiles
    a.recs = (mode == TSP_2OPT_BEST && t->use_recs && t->n >= 4096) ? t->d_rec : nullptr;
    a.partial_per_tour = t->partial_per_tour;
    a.n = t->n;
    a.rows_per_block = mode == TSP_2OPT_BEST ? t->best_rows_per_block : t->first_rows_per_block;
    a.first_min_rows = std::min(t->first_min_rows, std::max(1, t->n - 1));
    a.first_max_rows = 0;
    a.count_evals = t->count_evals;
    a.iter = iter; a.tenure = tenure;
    a.margin = t->inst->filter_margin;
    a.prune = t->inst->prune_margin;
    a.sum_margin = t->inst->sum_margin;
    a.pairtab = t->d_pairtab;
    a.cl_tickets = t->d_cl_ticket;
    a.orders2 = t->d_order2; a.poss2 = t->d_pos2;
    a.gbox = t->inst->d_gbox; a.gmax = t->d_gmax;
    a.ng = t->inst->ng; a.n_slots = t->inst->n_slots;
    a.flat_slots = t->sweep_blocks;
    a.tabu_list = tabu ? tabu->d_list : nullptr; a.tabu_list_n = tabu ? tabu->d_list_n : nullptr;
    a.tabu_list_cap = tabu ? tabu->list_cap : 0; a.tabu_pairs = tabu ? tabu->d_tabu_pairs : nullptr;
    return a;
}

template <int WT, bool INT>
int launch_step(tsp_dev_tours *t, int mode, tsp_dev_tabu *tabu, int iter, int tenure) {
    hipStream_t s = t->inst->ctx->stream;
    StepArgs a = make_args(t, mode, tabu, iter, tenure);
    if constexpr (exh_metric<WT>()) {
        if (exh_run(t, mode, tabu)) {
            hipLaunchKernelGGL((k_move_pos<WT, INT>), dim3((t->n + kExhPad + kScanThreads - 1) / kScanThreads, t->B), dim3(kScanThreads), 0, s,
                               t->inst->d_coord, t->d_order, t->d_pos, t->d_order2, t->d_pos2, t->d_state, t->d_pxy, t->d_pe, t->d_pid, t->n);
            a.flat_slots = t->exh_blocks;
            const int wt_ = t->exh_blocks * (kScanThreads / 64);
            const dim3 g(t->exh_blocks, 1, t->B);
            if (t->exh_lds > 65536 - 1024 && !t->exh_lds_granted) {   // one workgroup per CU: more than half of the CU's LDS
                const void *f = t->exh_rj == 16 ? reinterpret_cast<const void *>(k_exh<WT, INT, 16>) : (t->exh_rj == 8 ? reinterpret_cast<const void *>(k_exh<WT, INT, 8>) : reinterpret_cast<const void *>(k_exh<WT, INT, 4>));
                if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, t->exh_lds) != hipSuccess) { (void)hipGetLastError(); t->exh_lds = 65536 - 1024; }
                t->exh_lds_granted = true;
            }
            if (t->exh_rj == 16) hipLaunchKernelGGL((k_exh<WT, INT, 16>), g, dim3(kScanThreads), (size_t)t->exh_lds, s, a, t->d_pxy, t->d_pe, t->d_pid, wt_, t->exh_prio, make_int4(t->exh_share[0], t->exh_share[1], t->exh_share[2], t->exh_share[3]), t->exh_gens);
            else if (t->exh_rj == 8) hipLaunchKernelGGL((k_exh<WT, INT, 8>), g, dim3(kScanThreads), (size_t)t->exh_lds, s, a, t->d_pxy, t->d_pe, t->d_pid, wt_, t->exh_prio, make_int4(t->exh_share[0], t->exh_share[1], t->exh_share[2], t->exh_share[3]), t->exh_gens);
            else if (t->exh_rj == 2) hipLaunchKernelGGL((k_exh<WT, INT, 2>), g, dim3(kScanThreads), (size_t)t->exh_lds, s, a, t->d_pxy, t->d_pe, t->d_pid, wt_, t->exh_prio, make_int4(t->exh_share[0], t->exh_share[1], t->exh_share[2], t->exh_share[3]), t->exh_gens);
            else if (t->exh_rj == 1) hipLaunchKernelGGL((k_exh<WT, INT, 1>), g, dim3(kScanThreads), (size_t)t->exh_lds, s, a, t->d_pxy, t->d_pe, t->d_pid, wt_, t->exh_prio, make_int4(t->exh_share[0], t->exh_share[1], t->exh_share[2], t->exh_share[3]), t->exh_gens);
            else hipLaunchKernelGGL((k_exh<WT, INT, 4>), g, dim3(kScanThreads), (size_t)t->exh_lds, s, a, t->d_pxy, t->d_pe, t->d_pid, wt_, t->exh_prio, make_int4(t->exh_share[0], t->exh_share[1], t->exh_share[2], t->exh_share[3]), t->exh_gens);
            return TSP_OK;
        }
    }
    if constexpr (has_root_filter<WT>()) {
        if (sorted_run(t, mode, tabu)) {
            a.recs = t->d_rec;
            const int ns = t->inst->n_slots;
            hipLaunchKernelGGL((k_move_recs<WT, INT>), dim3((std::max(ns, t->n) + kScanThreads - 1) / kScanThreads, t->B),
                               dim3(kScanThreads), 0, s, t->inst->d_coord, t->d_order, t->d_pos, t->d_order2, t->d_pos2,
                               t->d_state, t->inst->d_sperm, t->d_rec, t->d_gmax, t->n, t->inst->ng, ns);
            if (tabu)
                hipLaunchKernelGGL((k_sweep<WT, INT, true>), dim3(t->sweep_blocks + kTabuSideBlocks, 1, t->B), dim3(kScanThreads), 0, s, a);
            else
                hipLaunchKernelGGL((k_sweep<WT, INT>), dim3(t->sweep_blocks, 1, t->B), dim3(kScanThreads), 0, s, a);
            return TSP_OK;
        }
    }
    if (mode == TSP_2OPT_BEST) {
        const dim3 g = scan_grid<TSP_2OPT_BEST>(t);
        if (a.recs)
            hipLaunchKernelGGL((k_recs<WT, INT>), dim3((t->n + kScanThreads - 1) / kScanThreads, t->B), dim3(kScanThreads), 0, s,
                               t->inst->d_coord, t->d_order, t->d_pos, t->d_state, t->d_rec, t->n);
        if (tabu && t->tabu_list_run) {   // from the list of non-zero stamps, on a tour outside the sorted sweep
            hipLaunchKernelGGL(k_tabu_side, dim3(kTabuSideBlocks), dim3(kScanThreads), 0, s, t->d_order, t->d_pos, t->n, t->d_state,
                               tabu->d_stamp, tabu->d_list, tabu->d_list_n, tabu->list_cap, iter, tenure, tabu->d_tabu_pairs);
            hipLaunchKernelGGL((k_step<WT, INT, TSP_2OPT_BEST, kBestRJ, false, true>), g, dim3(kScanThreads), 0, s, a);
        } else if (tabu)
            hipLaunchKernelGGL((k_step<WT, INT, TSP_2OPT_BEST, kBestRJ, true>), g, dim3(kScanThreads), 0, s, a);
        else
            hipLaunchKernelGGL((k_step<WT, INT, TSP_2OPT_BEST, kBestRJ, false>), g, dim3(kScanThreads), 0, s, a);
    } else if (!t->first_v1) {
        a.states = t->d_state_base; a.slot = t->slot;
        a.first_max_rows = t->first_max_rows2;
        const int tj = kScanThreads * t->first_rj;
        const dim3 g((t->n + tj - 1) / tj, t->first_grid_rows, t->B);
        if (t->first_rj == 2) hipLaunchKernelGGL((k_first<WT, INT, 2>), g, dim3(kScanThreads), 0, s, a);
        else hipLaunchKernelGGL((k_first<WT, INT, 1>), g, dim3(kScanThreads), 0, s, a);
        t->slot ^= 1;                                  // the launch's last block has written the other slot
        t->d_state = t->d_state_base + (size_t)t->slot * t->B;
    } else {
        const dim3 g = scan_grid<TSP_2OPT_FIRST>(t);
        a.first_max_rows = (int)g.y * t->first_rows_per_block;
        hipLaunchKernelGGL((k_step<WT, INT, TSP_2OPT_FIRST, kFirstRJ, false>), g, dim3(kScanThreads), 0, s, a);
    }
    return TSP_OK;
}

int launch_step_rt(tsp_dev_tours *t, int mode, tsp_dev_tabu *tabu, int iter, int tenure) {
    int rc = TSP_OK;
    TSP_DISPATCH_METRIC(t->inst->wtype, t->inst->integer_cost,
                        { rc = launch_step<WTC, INTC>(t, mode, tabu, iter, tenure); });
    return rc;
}

// Tickets for the first step of a run in `mode` (the tours may have been left by a run in the other mode).
void launch_arm(tsp_dev_tours *t, int mode) {
    hipStream_t s = t->inst->ctx->stream;
    if (mode == TSP_2OPT_BEST) {
        const dim3 g = scan_grid<TSP_2OPT_BEST>(t);
        hipLaunchKernelGGL((k_arm<TSP_2OPT_BEST>), dim3(t->B), dim3(kScanThreads), 0, s, t->d_state, t->d_ticket,
                           t->d_row_ticket, t->max_tile_rows, t->n,
                           t->best_rows_per_block, (int)g.x, (int)g.y, kScanThreads * kBestRJ);
    } else if (!t->first_v1) {
        (void)hipMemsetAsync(t->d_ticket, 0, sizeof(int) * (size_t)t->B, s);   // k_first counts up from zero
        // both control-block slots start equal (a run on tours that are already done must find `done` in both)
        (void)hipMemcpyAsync(t->d_state_base + (size_t)(1 - t->slot) * t->B, t->d_state, sizeof(TourState) * (size_t)t->B,
                             hipMemcpyDeviceToDevice, s);
    } else {
        const dim3 g = scan_grid<TSP_2OPT_FIRST>(t);
        hipLaunchKernelGGL((k_arm<TSP_2OPT_FIRST>), dim3(t->B), dim3(kScanThreads), 0, s, t->d_state, t->d_ticket,
                           t->d_row_ticket, t->max_tile_rows, t->n,
                           t->first_rows_per_block, (int)g.x, (int)g.y, kScanThreads * kFirstRJ);
    }
}

// out[b] (stride in bytes) = recomputed tour cost
void launch_tour_cost(tsp_dev_tours *t, double *d_out, size_t stride_bytes) {
    hipStream_t s = t->inst->ctx->stream;
    TSP_DISPATCH_METRIC(t->inst->wtype, t->inst->integer_cost, {
        hipLaunchKernelGGL((k_tour_cost<WTC, INTC>), dim3(t->B), dim3(kApplyThreads), 0, s, t->inst->d_coord,
                           t->d_order, t->d_pos, t->n, d_out, stride_bytes);
    });
}

double wall_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}


}  // namespace

// Shared with the other translation units of the library.
int tsp_grid_run(tsp_dev_tours *t, int mode, tsp_dev_tabu *tabu, int iter, int tenure, int64_t max_steps,
                 double time_limit_s, int sync, int *all_done) {
    if (!t || (mode != TSP_2OPT_FIRST && mode != TSP_2OPT_BEST)) return TSP_DEV_E_ARG;
    if (!sync && max_steps < 0) return TSP_DEV_E_ARG;   // "until done" needs the polls of a synchronous run: it would queue launches forever
    hipStream_t s = t->inst->ctx->stream;
    const double t0 = wall_s();
    if (tabu && (iter < 0 || tenure < 0)) tabu = nullptr;   // check_tenure answers 0 before it reads anything (tabusearch.c:84)
    t->tabu_list_run = false;
    if (tabu && mode == TSP_2OPT_BEST) {
        bool usable = false;
        const int rc = tabu_list_prepare(t, tabu, &usable);
        if (rc) return rc;
        t->tabu_list_run = usable && TSP_SW(t->inst, TABU_DENSE, 0) == 0;
        tabu->last_run_list = t->tabu_list_run;
        if (t->tabu_list_run) TSP_HIP_TRY(hipMemsetAsync(tabu->d_tabu_pairs, 0, kTabuSideWords * sizeof(unsigned long long), s));
    }
    const int64_t batch = 64;
    // launches between two looks at `done`: short descents (a kicked local optimum, a small instance) should not
    // pay for dozens of launches that find their tour finished, long ones not for many polls
    int64_t burst = sync ? 8 : batch;
    int64_t queued = 0;
    int status = TSP_OK;
    bool finished = false;
    if (all_done) *all_done = 0;
    launch_arm(t, mode);
    for (;;) {
        int64_t todo = burst;
        burst = std::min(batch, burst * 2);
        if (max_steps >= 0) todo = std::min<int64_t>(todo, max_steps - queued);
        if (todo <= 0) break;
        // A full batch of identical launches is replayed from a captured hipGraph (the kernel arguments
        // never change: the descent is driven by the device-resident control block); partial batches and
        // tabu runs (iter/tenure change per call) are launched directly.
        bool replayed = false;
        if (todo == batch && !tabu && t->use_graph && (mode == TSP_2OPT_BEST || t->first_v1)) {   // k_first alternates its slot argument
            hipGraphExec_t &exec = t->graph_exec[mode];
            if (!exec) {
                hipGraph_t g = nullptr;
                if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    for (int64_t k = 0; k < batch; ++k) launch_step_rt(t, mode, nullptr, 0, 0);
                    if (hipStreamEndCapture(s, &g) == hipSuccess && g) {
                        if (hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
                        (void)hipGraphDestroy(g);
                    }
                }
                if (!exec) { (void)hipGetLastError(); t->use_graph = 0; }
            }
            if (exec) { TSP_HIP_TRY(hipGraphLaunch(exec, s)); replayed = true; }
        }
        if (!replayed)
            for (int64_t k = 0; k < todo; ++k) {
                int rc = launch_step_rt(t, mode, tabu, iter, tenure);
                if (rc) return rc;
            }
        queued += todo;
        TSP_HIP_TRY(hipGetLastError());
        if (!sync) continue;
        TSP_HIP_TRY(hipMemcpyAsync(t->h_state, t->d_state, sizeof(TourState) * (size_t)t->B, hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
        bool done = true;
        for (int b = 0; b < t->B; ++b) done = done && t->h_state[b].done;
        if (done) { finished = true; if (all_done) *all_done = 1; break; }
        if (time_limit_s > 0 && wall_s() - t0 > time_limit_s) { status = TSP_TIME_LIMIT_EXCEEDED; break; }
    }
    if (tabu && t->tabu_list_run)   // the sweeps counted every non-adjacent pair; the skipped ones come off
        hipLaunchKernelGGL(k_tabu_fix_evals, dim3(1), dim3(64), 0, s, t->d_state, tabu->d_tabu_pairs);
    if (sorted_run(t, mode, tabu) || exh_run(t, mode, tabu) || (mode == TSP_2OPT_FIRST && !t->first_v1)) {
        launch_flush(t);
        TSP_HIP_TRY(hipGetLastError());
        if (sync == 1) TSP_HIP_TRY(hipStreamSynchronize(s));   // sync == 2: the caller queues more work and waits once
    }
    if ((status == TSP_TIME_LIMIT_EXCEEDED || (sync && !finished)) && mode == TSP_2OPT_BEST) {
        // the reference recomputes the cost on every exit path (tabusearch.c:168-172); a run capped by max_steps too
        launch_tour_cost(t, &t->d_state[0].obj, sizeof(TourState));
        TSP_HIP_TRY(hipStreamSynchronize(s));
    }
    return status;
}

// After another engine has rewritten order[] (two_opt_lds.hip): rebuild pos[], and give BEST runs
// that were cut short their recomputed cost (tabusearch.c:168-172).
// Queued, not waited for (everything that follows is ordered behind it on the engine's stream) -- except for the recomputed
// cost of a run that was cut short, which the caller reads.
int tsp_grid_after_external_run(tsp_dev_tours *t, int mode, int timed_out, bool pos_written) {
    hipStream_t s = t->inst->ctx->stream;
    if (!pos_written) hipLaunchKernelGGL(k_build_pos, dim3((t->n + 255) / 256, t->B), dim3(256), 0, s, t->d_order, t->d_pos, t->n);
    if (timed_out && mode == TSP_2OPT_BEST) {
        launch_tour_cost(t, &t->d_state[0].obj, sizeof(TourState));
        TSP_HIP_TRY(hipStreamSynchronize(s));
    }
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

tsp_dev_tours *tsp_scratch_tours(tsp_dev_inst *inst, int B, bool *owned, int *rc);   // api.hip
// two_opt_cluster.hip
int tsp_cluster_run(tsp_dev_tours *t, int mode, int C, int64_t max_steps, double time_limit_s, int *all_done, int *fell_through,
                    tsp_dev_tabu *tabu, int iter, int tenure);
bool tsp_cluster_fits(const tsp_dev_tours *t, int mode);
bool tsp_cluster_sorted(const tsp_dev_tours *t, int mode);
int tsp_cluster_size(const tsp_dev_tours *t, int mode);

// alg_2opt_tabu with a list on tour 0 of a handle: the CLUSTER engine when the tour fits its sorted scan and the handle's
// list of non-zero stamps is usable (one launch per descent, 11 us per sweep at n = 10 000), else the GRID engine (which
// works from the list as well when it can, and reads four stamps per pair when it cannot).  sync as tsp_grid_run.
int tsp_tabu_run(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter, int tenure, double time_limit_s, int sync, int *all_done) {
    struct PlanGuard { tsp_dev_tours *t; ~PlanGuard() { t->cl_tabu_plan = false; } } plan_guard{t};
    t->cl_tabu_plan = tabu && iter >= 0 && tenure >= 0;   // the cluster's sorted scan at any size (see cl_plan)
    // after a give-up (a workgroup was not resident) the CLUSTER engine is left out for a while: tsp_dev_ctx::cl_skip
    auto cluster_allowed = [&]() {
        tsp_dev_ctx *cx = t->inst->ctx;
        if (TSP_SW(t->inst, ENGINE, 0) == 3 || cx->cl_skip <= 0) return true;
        --cx->cl_skip;
        return false;
    };
    if (tabu && iter >= 0 && tenure >= 0 && t->B == 1 && TSP_SW(t->inst, TABU_DENSE, 0) == 0 && TSP_SW(t->inst, ENGINE, 0) != 1 &&
        tsp_cluster_fits(t, TSP_2OPT_BEST) && tsp_cluster_sorted(t, TSP_2OPT_BEST) && cluster_allowed()) {
        bool usable = false;
        int rc = tabu_list_prepare(t, tabu, &usable);
        if (rc) return rc;
        if (usable) {
            hipStream_t s = t->inst->ctx->stream;
            TSP_HIP_TRY(hipMemsetAsync(tabu->d_tabu_pairs, 0, kTabuSideWords * sizeof(unsigned long long), s));
            int fell = 0;
            const int status = tsp_cluster_run(t, TSP_2OPT_BEST, tsp_cluster_size(t, TSP_2OPT_BEST), -1, time_limit_s, all_done, &fell,
                                               tabu, iter, tenure);
            t->cl_tabu_plan = false;
            if (!fell) {
                if (status < 0) return status;
                tabu->last_run_list = true;
                hipLaunchKernelGGL(k_tabu_fix_evals, dim3(1), dim3(64), 0, s, t->d_state, tabu->d_tabu_pairs);
                TSP_HIP_TRY(hipGetLastError());
                if (sync == 1) TSP_HIP_TRY(hipStreamSynchronize(s));
                return status;
            }
            // A workgroup was not resident.  The tour and the control block in HBM are as the last launch that COMPLETED left
            // them (a run of more than 4096 sweeps, or a time-limited one, is several launches), the side words of the list
            // accounting have been put back to that point too (tsp_cluster_run), and stamps the failed launch cleared stay
            // cleared (expired either way: the same decisions, the same clears are due again).  The skipped pairs of the
            // completed launches come off the evaluation count now -- tsp_grid_run starts its own count at zero -- and the
            // rest of the descent goes through the GRID engine.
            hipLaunchKernelGGL(k_tabu_fix_evals, dim3(1), dim3(64), 0, s, t->d_state, tabu->d_tabu_pairs);
            TSP_HIP_TRY(hipGetLastError());
        }
    }
    if ((!tabu || iter < 0 || tenure < 0) && t->B == 1 && TSP_SW(t->inst, ENGINE, 0) != 1 && tsp_cluster_fits(t, TSP_2OPT_BEST) && cluster_allowed()) {
        // no list (check_tenure answers 0 before it reads anything, tabusearch.c:84): the plain best-improvement descent
        hipStream_t s = t->inst->ctx->stream;
        int fell = 0;
        const int status = tsp_cluster_run(t, TSP_2OPT_BEST, tsp_cluster_size(t, TSP_2OPT_BEST), -1, time_limit_s, all_done, &fell,
                                           nullptr, 0, 0);
        if (!fell) {
            if (status >= 0 && sync == 1) TSP_HIP_TRY(hipStreamSynchronize(s));
            return status;
        }
    }
    return tsp_grid_run(t, TSP_2OPT_BEST, tabu, iter, tenure, -1, time_limit_s, sync, all_done);
}
int tsp_perm_cost_device(tsp_dev_inst *inst, const int *d_perm, long long stride, int B, double *d_out, size_t out_stride_bytes);   // api.hip

// ---- resident-tour drivers (called by the extern "C" wrappers in api.hip) ---------------------------------------
int tsp_grid_rearm(tsp_dev_tours *t, int mode) {
    hipStream_t s = t->inst->ctx->stream;
    const int chunk = std::min(t->first_min_rows, std::max(1, t->n - 1));
    hipLaunchKernelGGL(k_rearm, dim3((t->B + 255) / 256), dim3(256), 0, s, t->d_state, t->B, chunk);
    (void)mode;
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

static int kick_buffers(tsp_dev_tours *t) {
    if (!t->d_kick_result) {
        TSP_HIP_TRY(hipMalloc(&t->d_kick_result, 8 * sizeof(int)));
        TSP_HIP_TRY(hipHostMalloc(&t->h_kick_result, 8 * sizeof(int)));
    }
    return TSP_OK;
}

int tsp_grid_tabu_kick(tsp_dev_tours *t, tsp_dev_tabu *tabu, int a, int b, int iter, int tenure, int *accepted) {
    if (!t || !tabu || t->B != 1 || a < 0 || b < 0 || a >= t->n || b >= t->n) return TSP_DEV_E_ARG;
    int rc = kick_buffers(t);
    if (rc) return rc;
    hipStream_t s = t->inst->ctx->stream;
    hipLaunchKernelGGL(k_tabu_kick, dim3(1), dim3(kApplyThreads), 0, s, t->d_order, t->d_pos, tabu->d_stamp, t->n, a, b, iter,
                       tenure, t->d_kick_result, tabu->list_valid ? tabu->d_list : nullptr, tabu->d_list_n, tabu->list_cap);
    TSP_HIP_TRY(hipMemcpyAsync(t->h_kick_result, t->d_kick_result, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    if (t->h_kick_result[0] && tabu->list_valid) {
        tabu->list_ub += 2;
        if (tabu->list_ub > tabu->list_cap) tabu->list_valid = false;   // entries may have been dropped: scan before the next use
    }
    if (accepted) *accepted = t->h_kick_result[0];
    return TSP_OK;
}

int tsp_grid_vns_kick(tsp_dev_tours *t, int p1, int p2, int p3, double *obj) {
    if (!t || t->B != 1 || !(0 <= p1 && p1 < p2 && p2 < p3 && p3 < t->n)) return TSP_DEV_E_ARG;
    hipStream_t s = t->inst->ctx->stream;
    const int n = t->n;
    hipLaunchKernelGGL(k_vns_kick, dim3((n + 255) / 256), dim3(256), 0, s, t->d_order, t->d_pos, t->d_order2, t->d_pos2, n, p1, p2, p3);
    TSP_HIP_TRY(hipMemcpyAsync(t->d_order, t->d_order2, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s));
    TSP_HIP_TRY(hipMemcpyAsync(t->d_pos, t->d_pos2, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s));
    // the reference recomputes the cost edge by edge along the new tour from node 0 (vns.c:77-86): order[] IS that walk now
    int rc = tsp_perm_cost_device(t->inst, t->d_order, n, 1, &t->d_state[0].obj, sizeof(TourState));
    if (rc) return rc;
    if (obj) {
        TSP_HIP_TRY(hipMemcpyAsync(&t->h_state[0].obj, &t->d_state[0].obj, sizeof(double), hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
        *obj = t->h_state[0].obj;
    }
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

int tsp_grid_snapshot(tsp_dev_tours *t, bool restore) {
    if (!t) return TSP_DEV_E_ARG;
    hipStream_t s = t->inst->ctx->stream;
    const size_t bn = (size_t)t->B * t->n;
    if (!restore) {
        if (!t->d_order_snap) TSP_HIP_TRY(hipMalloc(&t->d_order_snap, bn * sizeof(int)));
        TSP_HIP_TRY(hipMemcpyAsync(t->d_order_snap, t->d_order, bn * sizeof(int), hipMemcpyDeviceToDevice, s));
        TSP_HIP_TRY(hipMemcpyAsync(t->h_state, t->d_state, sizeof(TourState) * (size_t)t->B, hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
        t->h_obj_snap.resize((size_t)t->B);
        for (int b = 0; b < t->B; ++b) t->h_obj_snap[b] = t->h_state[b].obj;
        return TSP_OK;
    }
    if (!t->d_order_snap || (int)t->h_obj_snap.size() != t->B) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipMemcpyAsync(t->d_order, t->d_order_snap, bn * sizeof(int), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_build_pos, dim3((t->n + 255) / 256, t->B), dim3(256), 0, s, t->d_order, t->d_pos, t->n);
    for (int b = 0; b < t->B; ++b)
        TSP_HIP_TRY(hipMemcpyAsync(&t->d_state[b].obj, &t->h_obj_snap[b], sizeof(double), hipMemcpyHostToDevice, s));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

// One iteration of tabu() (src/tabusearch.c:238-309) with two waits for the device instead of five: alg_2opt_tabu on the
// resident tour; the poll that finds it finished also brings its cost (:168-172), so the host decides about the incumbent
// (:241-249) and queues, behind the run's last launches, the device-to-device snapshot and the first trial of the kick with
// the nodes a, b it has drawn (:264-265; nothing else draws in between), and waits once for all of it.
int tsp_grid_tabu_iteration(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter, int tenure, double time_limit_s, int a, int b,
                            double *best_obj, double *obj, int *improved, int *accepted) {
    if (!t || !tabu || t->B != 1 || tabu->inst != t->inst || !best_obj || a < 0 || b < 0 || a >= t->n || b >= t->n) return TSP_DEV_E_ARG;
    int rc = tsp_grid_rearm(t, TSP_2OPT_BEST);
    if (!rc) rc = kick_buffers(t);
    if (rc) return rc;
    int done = 0;
    hipStream_t s = t->inst->ctx->stream;
    // The CLUSTER engine finishes the descent in one launch (almost always): the incumbent's update and the kick are queued
    // behind that launch and decided on the device, and the iteration is ONE wait for the device.  Anything else (another
    // engine, a second launch, a give-up, the time limit) leaves result[4] = 0 and takes the two waits below.
    struct Post { tsp_dev_tours *t; tsp_dev_tabu *tabu; int a, b, iter, tenure; double best; } post{t, tabu, a, b, iter, tenure, *best_obj};
    const size_t bn = (size_t)t->n;
    if (!t->d_order_snap) TSP_HIP_TRY(hipMalloc(&t->d_order_snap, bn * sizeof(int)));
    t->h_kick_result[4] = 0;
    t->cl_post_ctx = &post; t->cl_post_ran = false;
    t->cl_post = [](void *ctx, hipStream_t st, const int *d_err) {
        Post *q = static_cast<Post *>(ctx);
        tsp_dev_tours *tt = q->t;
        hipLaunchKernelGGL(k_tabu_post, dim3(1), dim3(kApplyThreads), 0, st, tt->d_state, d_err, tt->d_order, tt->d_pos, q->tabu->d_stamp, tt->n,
                           q->a, q->b, q->iter, q->tenure, tt->d_kick_result, q->tabu->list_valid ? q->tabu->d_list : nullptr, q->tabu->d_list_n,
                           q->tabu->list_cap, q->best, tt->d_order_snap);
        (void)hipMemcpyAsync(tt->h_kick_result, tt->d_kick_result, 8 * sizeof(int), hipMemcpyDeviceToHost, st);
    };
    const int status = tsp_tabu_run(t, tabu, iter, tenure, time_limit_s, 2, &done);
    const bool post_ran = t->cl_post_ran;
    t->cl_post = nullptr; t->cl_post_ctx = nullptr; t->cl_post_ran = false;
    if (status < 0) return status;
    if (improved) *improved = 0;
    if (accepted) *accepted = 0;
    if (post_ran && t->h_kick_result[4]) {   // the run's own wait has brought the result back
        TSP_HIP_TRY(hipGetLastError());
        const double cost = t->h_state[0].obj;
        if (obj) *obj = cost;
        if (t->h_kick_result[5]) {
            *best_obj = cost;
            if (improved) *improved = 1;
            t->h_obj_snap.assign(1, cost);
        }
        if (t->h_kick_result[0] && tabu->list_valid) {
            tabu->list_ub += 2;
            if (tabu->list_ub > tabu->list_cap) tabu->list_valid = false;
        }
        if (accepted) *accepted = t->h_kick_result[0];
        return status;
    }
    if (status != TSP_OK || !done) {   // time limit: the cost was recomputed by the run; no kick (tabusearch.c:255-258)
        TSP_HIP_TRY(hipMemcpyAsync(t->h_state, t->d_state, sizeof(TourState), hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
        const double c = t->h_state[0].obj;
        if (obj) *obj = c;
        if (c < *best_obj) {   // the incumbent is updated before the status is looked at (:241-249, :255)
            *best_obj = c;
            if (improved) *improved = 1;
            const int rc2 = tsp_grid_snapshot(t, false);
            if (rc2) return rc2;
        }
        return status;
    }
    const double cost = t->h_state[0].obj;   // the poll that saw `done` carried the recomputed cost
    if (obj) *obj = cost;
    if (cost < *best_obj) {
        *best_obj = cost;
        if (improved) *improved = 1;
        const size_t bn = (size_t)t->n;
        if (!t->d_order_snap) TSP_HIP_TRY(hipMalloc(&t->d_order_snap, bn * sizeof(int)));
        TSP_HIP_TRY(hipMemcpyAsync(t->d_order_snap, t->d_order, bn * sizeof(int), hipMemcpyDeviceToDevice, s));
        t->h_obj_snap.assign(1, cost);
    }
    hipLaunchKernelGGL(k_tabu_kick, dim3(1), dim3(kApplyThreads), 0, s, t->d_order, t->d_pos, tabu->d_stamp, t->n, a, b, iter,
                       tenure, t->d_kick_result, tabu->list_valid ? tabu->d_list : nullptr, tabu->d_list_n, tabu->list_cap);
    TSP_HIP_TRY(hipMemcpyAsync(t->h_kick_result, t->d_kick_result, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    if (t->h_kick_result[0] && tabu->list_valid) {
        tabu->list_ub += 2;
        if (tabu->list_ub > tabu->list_cap) tabu->list_valid = false;
    }
    if (accepted) *accepted = t->h_kick_result[0];
    return status;
}

// `count` iterations of tabu() (src/tabusearch.c:238-309) in ONE wait for the device: the CLUSTER engine's launches of iterations
// iter0 .. iter0 + count - 1 are queued back to back, each followed by k_tabu_post_chain (incumbent + the kick's first trial with
// the host-drawn ab[2k], ab[2k + 1]), each preceded by a re-arm that the chain's stop word can veto.  *completed = iterations that
// ran to their kick's trial; the last of them may have had its trial rejected (*last_accepted = 0: the host draws further
// trials, tsp_dev_tours_tabu_kick, before it goes on), and an iteration whose descent did not finish in its launch is not counted:
// the host runs it through tsp_dev_tours_tabu_iteration.  Returns 0 with *completed = 0 when the chain does not apply (another
// engine, a long list): nothing was touched.
int tsp_grid_tabu_iterations(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter0, int count, const int *tenure, int pairs, const int *ab, double time_limit_s,
                             double *best_obj, double *obj, int *improved, int *trials, int *completed, int *last_accepted) {
    constexpr int kMaxChain = 128, kMaxPairs = 256;
    if (!t || !tabu || t->B != 1 || tabu->inst != t->inst || !best_obj || !tenure || !ab || !completed || count < 1) return TSP_DEV_E_ARG;
    *completed = 0;
    if (last_accepted) *last_accepted = 0;
    count = std::min(count, kMaxChain);
    // pairs > 0: the kick's trials in the order tabu() draws them -- a rejected one is followed by the next pair, inside the launch
    // (the first `count` pairs are the iterations' first trials only as long as none is rejected); 0: ab[2k], ab[2k + 1] is the one
    // trial of iteration k and the chain stops where it is rejected
    if (pairs < 0 || (pairs > 0 && pairs < count)) return TSP_DEV_E_ARG;
    pairs = std::min(pairs, kMaxPairs);
    const int npairs = pairs > 0 ? pairs : count;
    for (int k = 0; k < count; ++k) if (tenure[k] < 0) return TSP_DEV_E_ARG;
    for (int k = 0; k < npairs; ++k)
        if (ab[2 * k] < 0 || ab[2 * k] >= t->n || ab[2 * k + 1] < 0 || ab[2 * k + 1] >= t->n) return TSP_DEV_E_ARG;
    if (iter0 < 0) return TSP_DEV_E_ARG;
    // the chain rides on the CLUSTER engine's tabu variant: the conditions of tsp_tabu_run's first branch, and room in the list
    // of non-zero stamps for the two entries every accepted kick appends (no scan, no compaction inside a chain)
    if (TSP_SW(t->inst, TABU_DENSE, 0) != 0 || TSP_SW(t->inst, ENGINE, 0) == 1 || !tsp_cluster_fits(t, TSP_2OPT_BEST) ||
        !tsp_cluster_sorted(t, TSP_2OPT_BEST))
        return TSP_OK;
    {
        tsp_dev_ctx *cx = t->inst->ctx;
        if (TSP_SW(t->inst, ENGINE, 0) != 3 && cx->cl_skip > 0) return TSP_OK;   // after a give-up: the single-iteration path counts the back-off down
    }
    t->cl_tabu_plan = true;
    struct PlanGuard { tsp_dev_tours *t; ~PlanGuard() { t->cl_tabu_plan = false; } } plan_guard{t};
    bool usable = false;
    int rc = tabu_list_prepare(t, tabu, &usable, 2ll * count);   // (compacts now if the chain's entries would not fit before the next compaction)
    if (rc) return rc;
    if (!usable || tabu->list_ub + 2ll * count > std::min<long long>(kTabuListMax, tabu->list_cap) || tabu->list_ub + 2ll * count > tabu->list_compact_at)
        return TSP_OK;
    rc = kick_buffers(t);
    if (rc) return rc;
    hipStream_t s = t->inst->ctx->stream;
    // + in-kernel chains: the tenure per iteration, {a, b, a and b in rank order} per kick trial, the index of the next trial
    const int par_at = 4 + 10 * kMaxChain, ab_at = par_at + kMaxChain, pp_at = ab_at + 4 * kMaxPairs;
    const size_t chain_ints = (size_t)pp_at + 4;
    if (!t->d_chain) {
        TSP_HIP_TRY(hipMalloc(&t->d_chain, chain_ints * sizeof(int)));
        TSP_HIP_TRY(hipHostMalloc(&t->h_chain, chain_ints * sizeof(int)));
    }
    if (!t->d_order_snap) TSP_HIP_TRY(hipMalloc(&t->d_order_snap, (size_t)t->n * sizeof(int)));
    memset(t->h_chain, 0, chain_ints * sizeof(int));
    memcpy(t->h_chain + 2, best_obj, sizeof(double));
    const bool in_kernel = TSP_SW(t->inst, TABU_INKERNEL, 1) != 0;
    if (in_kernel) {
        for (int k = 0; k < count; ++k) t->h_chain[par_at + k] = tenure[k];
        for (int k = 0; k < npairs; ++k) {
            int *q = t->h_chain + ab_at + 4 * k;
            q[0] = ab[2 * k]; q[1] = ab[2 * k + 1];
            q[2] = t->inst->h_sinv[(size_t)ab[2 * k]]; q[3] = t->inst->h_sinv[(size_t)ab[2 * k + 1]];   // the two nodes inside the rank-order replica
        }
    } else if (pairs > 0) return TSP_OK;   // (queued chains take one trial per iteration: the caller falls back)
    TSP_HIP_TRY(hipMemcpyAsync(t->d_chain, t->h_chain, chain_ints * sizeof(int), hipMemcpyHostToDevice, s));
    rc = tsp_grid_rearm(t, TSP_2OPT_BEST);
    if (rc) return rc;
    TSP_HIP_TRY(hipMemsetAsync(tabu->d_tabu_pairs, 0, kTabuSideWords * sizeof(unsigned long long), s));
    if (in_kernel) {
        // The iterations run INSIDE the CLUSTER launch (k_cluster_two_opt, TABU variant, chain_n > 0): between two descents the
        // kernel itself keeps the incumbent, decides the kick's first trial and carries it out on the replicas -- what
        // k_tabu_post_chain does between two launches of a queued chain, without the write-back, the replica load and the two
        // kernel boundaries (28 + 4.6 + ~8 us of an iteration of ~140 at n = 10 000).  A launch that runs out of sweeps in the
        // middle of an iteration writes its state back and the next launch goes on at chain[1].
        t->cl_ik_n = count; t->cl_ik_par = par_at; t->cl_ik_pairs = pairs; t->cl_ik_ab = ab_at; t->cl_ik_pp = pp_at;
        int done = 0, fell = 0;
        const int status = tsp_cluster_run(t, TSP_2OPT_BEST, tsp_cluster_size(t, TSP_2OPT_BEST), -1, time_limit_s, &done, &fell, tabu, iter0, tenure[0]);
        t->cl_ik_n = 0;
        t->h_state_fresh = false;
        if (status < 0 && !fell) return status;
        if (!fell) {
            hipLaunchKernelGGL(k_tabu_fix_evals, dim3(1), dim3(64), 0, s, t->d_state, tabu->d_tabu_pairs);
            TSP_HIP_TRY(hipGetLastError());
        }
        TSP_HIP_TRY(hipMemcpyAsync(t->h_chain, t->d_chain, chain_ints * sizeof(int), hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
        (void)hipGetLastError();
        int nc = 0;
        for (int k = 0; k < count; ++k) {
            const int *res = t->h_chain + 4 + 10 * k;
            if (!res[4]) break;
            double c;
            memcpy(&c, res + 8, sizeof c);
            if (obj) obj[k] = c;
            if (improved) improved[k] = res[5];
            if (trials) trials[k] = res[3];
            if (res[5]) t->h_obj_snap.assign(1, c);
            if (res[0] && tabu->list_valid) {
                tabu->list_ub += 2;
                if (tabu->list_ub > tabu->list_cap) tabu->list_valid = false;
            }
            if (last_accepted) *last_accepted = res[0];
            ++nc;
        }
        if (fell) {
            // The exchange gave up.  In the launch's first exchanges nothing of the search state has been touched (the list may have
            // two entries more than stamps: a superset is what it has to be): the host takes the other path.  Later -- a workgroup
            // that had been resident stopped answering in the middle of a chain -- the kicks of the completed iterations are in
            // the stamps while the tour in HBM is the one the launch started from.
            if (nc == 0) { if (tabu->list_valid) tabu->list_ub += 2; return TSP_OK; }
            // Degraded, not aborted: the incumbent (tour and cost) and the stamps are intact, so the search goes on FROM THE
            // INCUMBENT -- a valid tabu search, no longer the reference's trajectory (which a device shared to the point of a
            // give-up has lost anyway: its runs end on the wall clock).  The event is counted and left in tsp_dev_last_error().
            tsp::set_last_error("k_cluster_two_opt: the exchange gave up in the middle of a chain of tabu() iterations; the search goes on from the incumbent",
                                hipErrorLaunchFailure, __FILE__, __LINE__);
            t->inst->ctx->cl_chain_losses += 1;
            memcpy(best_obj, t->h_chain + 2, sizeof(double));
            if (!t->h_obj_snap.empty()) {
                const int rc2 = tsp_grid_snapshot(t, /*restore=*/true);
                if (rc2) return rc2;
            }
            tabu->list_valid = false;   // (entries may be missing for stamps of the lost iterations: rebuilt by a scan before the next use)
            *completed = nc;
            if (last_accepted) *last_accepted = 1;
            return TSP_OK;
        }
        tabu->last_run_list = true;
        memcpy(best_obj, t->h_chain + 2, sizeof(double));
        *completed = nc;
        if (status == TSP_TIME_LIMIT_EXCEEDED && nc < count) {
            // iteration nc was cut short: its tour and recomputed cost are in HBM, and the incumbent is updated before the status is
            // looked at (tabusearch.c:241-249, :255) -- reported in slot nc, which is not counted as completed
            TSP_HIP_TRY(hipMemcpyAsync(t->h_state, t->d_state, sizeof(TourState), hipMemcpyDeviceToHost, s));
            TSP_HIP_TRY(hipStreamSynchronize(s));
            const double c = t->h_state[0].obj;
            if (obj) obj[nc] = c;
            if (improved) improved[nc] = 0;
            if (c < *best_obj) {
                *best_obj = c;
                if (improved) improved[nc] = 1;
                const int rc2 = tsp_grid_snapshot(t, false);
                if (rc2) return rc2;
            }
        }
        return status == TSP_TIME_LIMIT_EXCEEDED ? status : TSP_OK;
    }
    struct Chain { tsp_dev_tours *t; tsp_dev_tabu *tabu; int iter0, count; const int *tenure, *ab; } ch{t, tabu, iter0, count, tenure, ab};
    auto post = [](void *ctx, hipStream_t st, int k, const int *d_err) {   // ONE launch between two CLUSTER launches: evaluation count, incumbent, kick, re-arm
        Chain *q = static_cast<Chain *>(ctx);
        tsp_dev_tours *tt = q->t;
        const int chunk = k + 1 < q->count ? std::min(tt->first_min_rows, std::max(1, tt->n - 1)) : 0;
        hipLaunchKernelGGL(k_tabu_post_chain, dim3(1), dim3(kApplyThreads), 0, st, tt->d_state, d_err, tt->d_order, tt->d_pos, q->tabu->d_stamp,
                           tt->n, q->ab[2 * k], q->ab[2 * k + 1], q->iter0 + k, q->tenure[k], tt->d_chain, k, q->tabu->d_list, q->tabu->d_list_n,
                           q->tabu->list_cap, tt->d_order_snap, q->tabu->d_tabu_pairs, chunk);
    };
    t->cl_post_ctx = &ch; t->cl_post_ran = false;
    t->cl_post_k = post;
    t->cl_post = [](void *ctx, hipStream_t st, const int *d_err) { static_cast<Chain *>(ctx)->t->cl_post_k(ctx, st, 0, d_err); };
    t->cl_chain = [](void *ctx, hipStream_t, int k, int *iter, int *ten) {
        Chain *q = static_cast<Chain *>(ctx);
        if (k >= q->count) return false;
        *iter = q->iter0 + k; *ten = q->tenure[k];   // (the re-arm was the previous post kernel's last act, unless it stopped the chain)
        return true;
    };
    int done = 0, fell = 0;
    const int status = tsp_cluster_run(t, TSP_2OPT_BEST, tsp_cluster_size(t, TSP_2OPT_BEST), -1, time_limit_s, &done, &fell, tabu, iter0, tenure[0]);
    const bool post_ran = t->cl_post_ran;
    const int launched = t->cl_chain_launched;
    t->cl_post = nullptr; t->cl_post_k = nullptr; t->cl_chain = nullptr; t->cl_post_ctx = nullptr; t->cl_post_ran = false; t->cl_chain_launched = 0;
    t->h_state_fresh = false;
    if (!post_ran) return status < 0 && !fell ? status : TSP_OK;   // nothing was launched: the tour is as it was
    // the results: one copy, one wait (the run's own wait came before the chain's last kernels were known to be through only if
    // the chain was a single launch; the copy below is ordered behind all of them either way)
    TSP_HIP_TRY(hipMemcpyAsync(t->h_chain, t->d_chain, chain_ints * sizeof(int), hipMemcpyDeviceToHost, s));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    (void)hipGetLastError();
    tabu->last_run_list = true;
    int nc = 0;
    for (int k = 0; k < launched && k < count; ++k) {
        const int *res = t->h_chain + 4 + 10 * k;
        if (!res[4]) break;
        double c;
        memcpy(&c, res + 8, sizeof c);
        if (obj) obj[k] = c;
        if (improved) improved[k] = res[5];
        if (trials) trials[k] = 1;
        if (res[5]) t->h_obj_snap.assign(1, c);
        if (res[0] && tabu->list_valid) {
            tabu->list_ub += 2;
            if (tabu->list_ub > tabu->list_cap) tabu->list_valid = false;
        }
        if (last_accepted) *last_accepted = res[0];
        ++nc;
    }
    memcpy(best_obj, t->h_chain + 2, sizeof(double));
    *completed = nc;
    if (fell && nc == 0) return TSP_OK;   // the exchange gave up in the first launch: nothing completed, the host takes the other path
    return status == TSP_TIME_LIMIT_EXCEEDED ? status : TSP_OK;
}

// alg_2opt_tabu on the resident tour 0 (no upload, no download): *obj = the recomputed cost (tabusearch.c:168-172)
int tsp_grid_resident_tabu(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter, int tenure, double time_limit_s, double *obj) {
    if (!t || t->B != 1 || (tabu && tabu->inst != t->inst)) return TSP_DEV_E_ARG;
    int rc = tsp_grid_rearm(t, TSP_2OPT_BEST);
    if (rc) return rc;
    int done = 0;
    const int status = tsp_tabu_run(t, tabu, iter, tenure, time_limit_s, 2, &done);
    if (status < 0) return status;
    if (status != TSP_OK || !done) {   // cut short: the recomputed cost was written after the last poll
        hipStream_t s = t->inst->ctx->stream;
        TSP_HIP_TRY(hipMemcpyAsync(t->h_state, t->d_state, sizeof(TourState), hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
    }
    if (obj) *obj = t->h_state[0].obj;   // the poll that saw `done` carried the cost (tabusearch.c:168-172)
    return status;
}

extern "C" {

int tsp_dev_tours_create(tsp_dev_inst *inst, int B, tsp_dev_tours **out) {
    if (!inst || !out || B < 1) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    tsp_dev_tours *t = new tsp_dev_tours();
    t->inst = inst; t->B = B; t->n = inst->n;
    struct Guard { tsp_dev_tours *t; ~Guard() { if (t) tsp_dev_tours_destroy(t); } } guard{t};   // an early error return frees what exists
    // FIRST-mode chunk geometry.  A step costs a launch (~10 us of latency) plus the evaluation of
    // rows x n pairs per tour; only the pairs up to the first improving one are useful.  One tour:
    // latency dominates, 32 rows is the measured optimum at n = 10000.  Many tours: keep the smallest
    // chunk near 2M pairs per launch so that dense-improvement phases do not pay for rows they discard.
    const long long pairs_budget = 2000000;
    int auto_min = (int)std::min<long long>(32, std::max<long long>(4, pairs_budget / ((long long)B * inst->n)));
    auto_min = auto_min >= 32 ? 32 : (auto_min >= 16 ? 16 : (auto_min >= 8 ? 8 : 4));
    t->first_min_rows = std::max(1, TSP_SW(inst, FIRST_MIN_ROWS, auto_min));
    t->first_rows_per_block =
        std::min(kMaxRowsPerBlock, std::max(1, TSP_SW(inst, FIRST_ROWS_PER_BLOCK, std::min(8, t->first_min_rows))));
    t->first_max_rows = std::max(t->first_min_rows, TSP_SW(inst, FIRST_MAX_ROWS, 2048));
    {   // Every launch dispatches the grid of the LARGEST chunk (blocks beyond a tour's current chunk
        // return at once, but dispatching them is not free): with many tours keep that grid near
        // 16k blocks so that a step stays latency-sized.
        const long long gx = (inst->n + kScanThreads * kFirstRJ - 1) / (kScanThreads * kFirstRJ);
        const long long budget = std::max<long long>(1, 16384 / (gx * B));
        const int cap = (int)std::max<long long>(t->first_min_rows, budget * t->first_rows_per_block);
        t->first_max_rows = std::min(t->first_max_rows, cap);
    }
    {   // k_first: a fixed grid of gy tile rows (<= 64: a block numbers the working blocks with one wave); a block
        // takes ceil(chunk / gy) <= kMaxRowsPerBlock rows.  Many tours: keep the launch near 16k blocks.
        t->first_rj = TSP_SW(inst, FIRST_RJ, 2) == 2 ? 2 : 1;
        const long long gx = (inst->n + kScanThreads * t->first_rj - 1) / (kScanThreads * t->first_rj);
        // every working block takes a ticket on one word (~12 ns each): few, fat blocks
        const int gy = (int)std::max<long long>(1, std::min<long long>(TSP_SW(inst, FIRST_GRID_ROWS, 8), 16384 / (gx * B)));
        t->first_grid_rows = std::min(64, gy);
        t->first_max_rows2 = std::max(t->first_min_rows, std::min(TSP_SW(inst, FIRST_MAX_ROWS, 2048), t->first_grid_rows * kMaxRowsPerBlock));
        t->first_v1 = TSP_SW(inst, FIRST_V1, 0);
    }
    t->best_rows_per_block = std::min(kMaxRowsPerBlock, std::max(1, TSP_SW(inst, BEST_ROWS_PER_BLOCK, 32)));
    t->count_evals = TSP_SW(inst, COUNT_EVALS, 1);
    t->use_graph = TSP_SW(inst, USE_GRAPH, 0);
    const size_t bn = (size_t)B * inst->n;
    const dim3 gb = scan_grid<TSP_2OPT_BEST>(t), gf = scan_grid<TSP_2OPT_FIRST>(t);
    t->partial_per_tour = std::max((size_t)gb.x * gb.y, (size_t)gf.x * gf.y);
    t->partial_per_tour = std::max(t->partial_per_tour, (size_t)((inst->n + kScanThreads - 1) / kScanThreads) * t->first_grid_rows);
    t->sorted_min_n = TSP_SW(inst, SORTED_MIN_N, 1000);
    t->cl_sorted_min_n = TSP_SW(inst, SORTED_MIN_N, 8);
    size_t rec_per_tour = (size_t)inst->n;
    size_t cl_words = (size_t)B * 64 * 64;   // k_first: one arrival counter per tour x tile row, 64 ints apart
    if (inst->d_sperm) {
        // k_sweep blocks per tour: whole clusters; about eight group pairs per block, between one and three blocks
        // per CU for one tour (measured over n = 500 .. 15 000: 256 blocks are best below ~4000 nodes, 512-768 above)
        const long long gp = (long long)inst->ng * (inst->ng + 1) / 2;
        const int one_tour = (int)std::min<long long>(768, std::max<long long>(256, gp / 8));
        const int want = TSP_SW(inst, SWEEP_BLOCKS, std::max(16, one_tour / B));
        t->sweep_blocks = std::max(1, want / kSweepCluster) * kSweepCluster;
        rec_per_tour = std::max(rec_per_tour, (size_t)inst->n_slots);
        t->partial_per_tour = std::max(t->partial_per_tour, (size_t)t->sweep_blocks);
        cl_words = std::max(cl_words, (size_t)B * (t->sweep_blocks / kSweepCluster) * 64);
        const long long npairs = (long long)inst->ng * (inst->ng + 1) / 2;
        if (npairs <= (1ll << 24) && inst->ng <= 32768 && TSP_SW(inst, SWEEP_TABLE, 1)) {
            // group pairs by box distance, dealt to the clusters in turn (see k_sweep)
            const int ng = inst->ng, Q = t->sweep_blocks / kSweepCluster;
            const long long ntests = (npairs + Q - 1) / Q;
            std::vector<std::pair<double, int>> pr((size_t)npairs);
            size_t w = 0;
            for (int r = 0; r < ng; ++r)
                for (int c = r; c < ng; ++c) {
                    const double4 &rb = inst->h_gbox[r], &cb = inst->h_gbox[c];
                    const double gx = std::max(0.0, std::max(rb.x - cb.y, cb.x - rb.y)), gy = std::max(0.0, std::max(rb.z - cb.w, cb.z - rb.w));
                    pr[w++] = {gx * gx + gy * gy, (r << 16) | c};
                }
            std::sort(pr.begin(), pr.end());
            std::vector<int> tab((size_t)Q * ntests, -1);
            for (long long k = 0; k < npairs; ++k) tab[(size_t)(k % Q) * ntests + (size_t)(k / Q)] = pr[(size_t)k].second;
            TSP_HIP_TRY(hipMalloc(&t->d_pairtab, tab.size() * sizeof(int)));
            TSP_HIP_TRY(hipMemcpy(t->d_pairtab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        TSP_HIP_TRY(hipMalloc(&t->d_gmax, (size_t)B * (inst->ng + 1) * sizeof(double)));

    }
    if (inst->filter_margin > 1e299 && TSP_SW(inst, EXH_POS, 1) && inst->n >= 5 &&
        (inst->wtype == WT_EUC_2D_ICOORD || inst->wtype == WT_CEIL_2D_ICOORD || inst->wtype == WT_ATT_ICOORD)) {
        // the exhaustive sweep in position order (two_opt_exh.hpp): EXH_WAVES waves per SIMD of one tour's grid, all resident
        const int waves = std::max(1, std::min(8, TSP_SW(inst, EXH_WAVES, 4)));
        t->exh_blocks = std::max(1, std::min(2048, inst->ctx->num_cus * waves / std::max(1, B > 4 ? 4 : B)));
        // One tour: the grid is `waves` workgroups of four waves per CU, and every CU must get exactly that many -- the kernel is
        // VALU-bound and the waves' shares are equal, so a CU that the dispatcher handed six workgroups finishes 1.5 x later than
        // the mean (measured: waves leaving their rows between 13 and 40 us, mean 24.6).  Each workgroup therefore asks for
        // 1 / waves of the CU's LDS (it uses none of it): one more does not fit.
        if (B == 1 && inst->ctx->lds_bytes >= 65536 && TSP_SW(inst, EXH_EVEN, 1))
            t->exh_lds = (waves == 1 ? (inst->ctx->lds_bytes * 3) / 5 : std::min(65536, inst->ctx->lds_bytes / waves)) - 1024;   // one per CU: more than half of it
        const int rj = TSP_SW(inst, EXH_RJ, 4);
        t->exh_prio = TSP_SW(inst, EXH_PRIO, 0);
        t->exh_rj = (rj == 16 || rj == 8 || rj == 2 || rj == 1) ? rj : 4;
        t->partial_per_tour = std::max(t->partial_per_tour, (size_t)t->exh_blocks);
        const size_t pn = (size_t)B * (inst->n + kExhPad);
        TSP_HIP_TRY(hipMalloc(&t->d_pxy, pn * sizeof(double2)));
        TSP_HIP_TRY(hipMalloc(&t->d_pe, pn * sizeof(int)));
        TSP_HIP_TRY(hipMalloc(&t->d_pid, pn * sizeof(int)));
        {   // rows per wave for each of the `waves` equal parts of the grid (k_exh: the older a workgroup, the larger its share);
            // TSP_EXH_SHARES = per-cent figures (or "0": equal shares), defaults measured on MI355X for 2 / 3 / 4 workgroups per CU
            t->exh_share[0] = t->exh_share[1] = t->exh_share[2] = t->exh_share[3] = 0;
            t->exh_gens = 0;
            int pc[4] = {0, 0, 0, 0};
            bool on = B == 1 && waves >= 2 && waves <= 4;
            if (waves == 4) { pc[0] = 52; pc[1] = 28; pc[2] = 13; pc[3] = 7; }   // (tools/exh_shares.sh: 44.7 us per sweep against 45.7 with 53 / 26 / 13 / 8)
            if (waves == 3) { pc[0] = 56; pc[1] = 29; pc[2] = 15; }
            if (waves == 2) { pc[0] = 62; pc[1] = 38; }
            const char *e = getenv("TSP_EXH_SHARES");
            if (e && *e) {
                pc[0] = pc[1] = pc[2] = pc[3] = 0;
                const int got = sscanf(e, "%d,%d,%d,%d", &pc[0], &pc[1], &pc[2], &pc[3]);
                on = on && got == waves && pc[0] > 0;
            }
            const int W = 64 * t->exh_rj, WEFF = W - 1, strips = (inst->n + WEFF - 1) / WEFF;
            long long total = 0;
            for (int sidx = 0; sidx < strips; ++sidx) total += std::min(inst->n - 1, sidx * WEFF + WEFF - 1);
            const long long wtot = (long long)t->exh_blocks * (kScanThreads / 64);
            const int sum = pc[0] + pc[1] + pc[2] + pc[3];
            if (on && sum > 0 && wtot % waves == 0) {
                const long long wq = wtot / waves;
                t->exh_gens = waves;
                for (int q = 0; q < waves; ++q) t->exh_share[q] = (int)std::max<long long>(1, (total * pc[q] + sum * wq - 1) / (sum * wq));
            }
        }
    }
    TSP_HIP_TRY(hipMalloc(&t->d_cl_ticket, cl_words * sizeof(int)));
    TSP_HIP_TRY(hipMemset(t->d_cl_ticket, 0, cl_words * sizeof(int)));   // every launch leaves them at zero
    TSP_HIP_TRY(hipMalloc(&t->d_order, bn * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_order2, bn * sizeof(int)));   // second copies: moves are carried out of place
    TSP_HIP_TRY(hipMalloc(&t->d_pos2, bn * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_order0, bn * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_pos, bn * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_state_base, 2 * (size_t)B * sizeof(TourState)));   // two slots per tour (k_first)
    t->d_state = t->d_state_base; t->slot = 0;
    TSP_HIP_TRY(hipMalloc(&t->d_partial, (size_t)B * t->partial_per_tour * sizeof(Partial)));
    TSP_HIP_TRY(hipMalloc(&t->d_slot_evals, (size_t)B * t->partial_per_tour * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_ticket, (size_t)B * sizeof(int)));
    t->use_recs = TSP_SW(inst, BEST_RECS, 1);
    TSP_HIP_TRY(hipMalloc(&t->d_rec, (size_t)B * rec_per_tour * sizeof(NodeRec)));
    t->max_tile_rows = std::max((int)gb.y, (int)gf.y);
    TSP_HIP_TRY(hipMalloc(&t->d_row_ticket, (size_t)B * t->max_tile_rows * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_row_evals, (size_t)B * t->max_tile_rows * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_row_slot, (size_t)B * t->max_tile_rows * sizeof(Partial)));
    TSP_HIP_TRY(hipHostMalloc(&t->h_state, (size_t)B * sizeof(TourState)));
    guard.t = nullptr;
    *out = t;
    return TSP_OK;
}

void tsp_dev_tours_destroy(tsp_dev_tours *t) {
    if (!t) return;
    (void)hipSetDevice(t->inst->ctx->device);
    (void)hipStreamSynchronize(t->inst->ctx->stream);
    (void)hipFree(t->d_order); (void)hipFree(t->d_order0); (void)hipFree(t->d_pos);
    (void)hipFree(t->d_state_base); (void)hipFree(t->d_partial); (void)hipFree(t->d_slot_evals); (void)hipFree(t->d_ticket); (void)hipFree(t->d_rec);
    (void)hipFree(t->d_gmax); (void)hipFree(t->d_order2); (void)hipFree(t->d_pos2); (void)hipFree(t->d_pairtab); (void)hipFree(t->d_cl_ticket);
    (void)hipFree(t->d_row_ticket); (void)hipFree(t->d_row_evals); (void)hipFree(t->d_row_slot);
    (void)hipFree(t->d_pxy); (void)hipFree(t->d_pe); (void)hipFree(t->d_pid);
    (void)hipFree(t->d_cl_slots); (void)hipFree(t->d_cl_pairtab); (void)hipFree(t->d_cl_stats);
    (void)hipFree(t->d_chain); (void)hipHostFree(t->h_chain); (void)hipFree(t->d_order_snap); (void)hipFree(t->d_kick_result); (void)hipHostFree(t->h_kick_result); (void)hipHostFree(t->h_cl_err);
    (void)hipHostFree(t->h_state);
    for (int m = 0; m < 2; ++m) if (t->graph_exec[m]) (void)hipGraphExecDestroy(t->graph_exec[m]);
    delete t;
}

int tsp_dev_tours_reset(tsp_dev_tours *t) {
    if (!t) return TSP_DEV_E_ARG;
    hipStream_t s = t->inst->ctx->stream;
    const int n = t->n, B = t->B;
    TSP_HIP_TRY(hipMemcpyAsync(t->d_order, t->d_order0, (size_t)B * n * sizeof(int), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_build_pos, dim3((n + 255) / 256, B), dim3(256), 0, s, t->d_order, t->d_pos, n);
    // control blocks: cursor at sweep start, smallest chunk, obj = uploaded value
    for (int b = 0; b < B; ++b) {
        TourState z;
        memset(&z, 0, sizeof z);
        z.chunk_rows = std::min(t->first_min_rows, std::max(1, n - 1));
        z.obj = t->h_obj0[b];
        z.seen_cost = t->h_obj0[b];
        t->h_state[b] = z;
    }
    t->slot = 0; t->d_state = t->d_state_base;
    if (t->d_cl_stats) TSP_HIP_TRY(hipMemsetAsync(t->d_cl_stats, 0, sizeof(long long) * (size_t)B * 256 * 4, s));
    TSP_HIP_TRY(hipMemcpyAsync(t->d_state, t->h_state, sizeof(TourState) * (size_t)B, hipMemcpyHostToDevice, s));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

int tsp_dev_tours_upload(tsp_dev_tours *t, const int *succ, int succ_stride, int64_t tour_stride, const double *obj) {
    if (!t || !succ || succ_stride < 1) return TSP_DEV_E_ARG;
    const int n = t->n, B = t->B;
    hipStream_t s = t->inst->ctx->stream;
    std::vector<int> &order = t->h_order_buf;   // lives until the reset below has synchronised
    order.resize((size_t)B * n);
    std::vector<char> seen((size_t)n);
    for (int b = 0; b < B; ++b) {
        const int *sp = succ + (size_t)b * tour_stride;
        std::fill(seen.begin(), seen.end(), 0);
        int v = 0;
        for (int p = 0; p < n; ++p) {
            if (v < 0 || v >= n || seen[v]) return TSP_DEV_E_NOT_A_TOUR;
            seen[v] = 1;
            order[(size_t)b * n + p] = v;
            v = sp[(size_t)v * succ_stride];
        }
        if (v != 0) return TSP_DEV_E_NOT_A_TOUR;
    }
    t->h_obj0.assign((size_t)B, 0.0);
    if (obj) for (int b = 0; b < B; ++b) t->h_obj0[b] = obj[b];
    TSP_HIP_TRY(hipMemcpyAsync(t->d_order0, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice, s));
    return tsp_dev_tours_reset(t);   // one synchronisation for the upload and the reset
}

int tsp_dev_tours_download(tsp_dev_tours *t, int *succ, int succ_stride, int64_t tour_stride, double *obj,
                           tsp_two_opt_stats *stats) {
    if (!t) return TSP_DEV_E_ARG;
    const int n = t->n, B = t->B;
    hipStream_t s = t->inst->ctx->stream;
    TSP_HIP_TRY(hipMemcpyAsync(t->h_state, t->d_state, sizeof(TourState) * (size_t)B, hipMemcpyDeviceToHost, s));
    std::vector<int> order;
    if (succ) {
        order.resize((size_t)B * n);
        TSP_HIP_TRY(hipMemcpyAsync(order.data(), t->d_order, order.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    }
    std::vector<long long> part;   // the CLUSTER engine's executed-work counters, one slot per workgroup
    if (stats && t->d_cl_stats) {
        part.resize((size_t)B * 256 * 4);
        TSP_HIP_TRY(hipMemcpyAsync(part.data(), t->d_cl_stats, part.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
    }
    TSP_HIP_TRY(hipStreamSynchronize(s));
    for (int b = 0; b < B; ++b) {
        if (succ) {
            int *sp = succ + (size_t)b * tour_stride;
            const int *op = order.data() + (size_t)b * n;
            for (int p = 0; p < n; ++p) sp[(size_t)op[p] * succ_stride] = op[p + 1 == n ? 0 : p + 1];
        }
        const TourState &z = t->h_state[b];
        if (obj) obj[b] = z.obj;
        if (stats) {
            tsp_two_opt_stats &o = stats[b];
            o.sweeps = z.sweeps; o.evals = z.evals; o.moves = z.moves; o.reversed = z.reversed;
            o.pairs_scanned = z.pairs_scanned; o.steps = z.steps;
            o.seconds = 0.0; o.device_ms = t->device_ms;   // the last tsp_dev_tours_run_engine (host-tour calls overwrite both)
            long long cs[4] = {z.lane_pairs, z.tier1_pairs, z.exact_pairs, z.staged_recs};
            if (!part.empty())
                for (int w = 0; w < 256; ++w)
                    for (int k = 0; k < 4; ++k) cs[k] += part[((size_t)b * 256 + w) * 4 + k];
            const bool counted = cs[0] > 0 || cs[2] > 0;   // the CLUSTER engine counts what it executes
            o.lane_pairs = counted ? cs[0] : z.pairs_scanned;
            o.tier1_pairs = counted ? cs[1] : -1; o.exact_pairs = counted ? cs[2] : -1;
            o.staged_recs = counted ? cs[3] : -1;
        }
    }
    return TSP_OK;
}

int tsp_dev_tours_run(tsp_dev_tours *t, int mode, int64_t max_steps, double time_limit_s, int sync, int *all_done) {
    if (!t) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    return tsp_grid_run(t, mode, nullptr, 0, 0, max_steps, time_limit_s, sync, all_done);
}

int tsp_dev_tours_time_scan(tsp_dev_tours *t, int reps, float *mean_ms, int64_t *evals_per_launch) {
    if (!t || reps < 1) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    hipStream_t s = t->inst->ctx->stream;
    const int n = t->n;
    hipEvent_t e0, e1;
    TSP_HIP_TRY(hipEventCreate(&e0));
    TSP_HIP_TRY(hipEventCreate(&e1));
    double total = 0.0;
    launch_arm(t, TSP_2OPT_BEST);
    launch_step_rt(t, TSP_2OPT_BEST, nullptr, 0, 0);   // warm
    TSP_HIP_TRY(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) launch_step_rt(t, TSP_2OPT_BEST, nullptr, 0, 0);   // back to back on the engine's stream
    TSP_HIP_TRY(hipEventRecord(e1, s));
    TSP_HIP_TRY(hipEventSynchronize(e1));
    if (sorted_sweep(t) || exh_run(t, TSP_2OPT_BEST, nullptr)) { launch_flush(t); TSP_HIP_TRY(hipStreamSynchronize(s)); }
    {
        float ms = 0.f;
        TSP_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        total = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (mean_ms) *mean_ms = (float)(total / reps);
    if (evals_per_launch) *evals_per_launch = ((int64_t)n * (n - 1) / 2 - n) * t->B;
    return TSP_OK;
}

int tsp_dev_tours_device_ms(tsp_dev_tours *t, double *ms) {
    if (!t || !ms) return TSP_DEV_E_ARG;
    *ms = t->device_ms;
    return TSP_OK;
}

int tsp_dev_tours_describe(tsp_dev_tours *t, int mode, char *buf, int cap) {
    if (!t || !buf || cap < 1 || (mode != TSP_2OPT_FIRST && mode != TSP_2OPT_BEST)) return TSP_DEV_E_ARG;
    if (exh_run(t, mode, nullptr))
        snprintf(buf, (size_t)cap, "k_move_pos + k_exh<RJ=%d> x %d workgroups (every delta expression, tour-position order)", t->exh_rj, t->exh_blocks);
    else if (sorted_run(t, mode, nullptr))
        snprintf(buf, (size_t)cap, "k_move_recs + k_sweep x %d workgroups (sorted sweep, box bound)", t->sweep_blocks);
    else if (mode == TSP_2OPT_BEST)
        snprintf(buf, (size_t)cap, "%sk_step<BEST> (tiled sweep%s)", t->use_recs && t->n >= 4096 ? "k_recs + " : "",
                 t->inst->filter_margin > 1e299 ? ", every delta expression" : ", bound tiers");
    else
        snprintf(buf, (size_t)cap, "%s", t->first_v1 ? "k_step<FIRST>" : "k_first");
    return TSP_OK;
}

int tsp_dev_tours_best(tsp_dev_tours *t, int true_cost, int64_t *packed) {
    if (!t || !packed) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    hipStream_t s = t->inst->ctx->stream;
    std::vector<double> cost((size_t)t->B);
    if (true_cost) {
        double *d_c = static_cast<double *>(tsp_io_pool(t->inst, sizeof(double) * (size_t)t->B));
        if (!d_c) return TSP_DEV_E_NOMEM;
        launch_tour_cost(t, d_c, sizeof(double));
        TSP_HIP_TRY(hipMemcpyAsync(cost.data(), d_c, sizeof(double) * (size_t)t->B, hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
    } else {
        TSP_HIP_TRY(hipMemcpyAsync(t->h_state, t->d_state, sizeof(TourState) * (size_t)t->B, hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
        for (int b = 0; b < t->B; ++b) cost[b] = t->h_state[b].obj;
    }
    int64_t best = INT64_MAX;
    for (int b = 0; b < t->B; ++b) {
        const int64_t p = ((int64_t)cost[b] << 24) | (int64_t)b;
        best = std::min(best, p);
    }
    *packed = best;
    return TSP_OK;
}

// ---- tabu stamps (tabusearch.c:195, :306-309) and alg_2opt_tabu ------------------------------

int tsp_dev_tabu_create(tsp_dev_inst *inst, tsp_dev_tabu **out) {
    if (!inst || !out) return TSP_DEV_E_ARG;
    // the stamp of pair (i, j) sits at the int index x_udir_pos(i, j, n) (src/utility.c:17-30), here and in every kernel:
    // n (n - 1) / 2 must stay below 2^31 (n <= 65 536, the limit the reference's int arithmetic has too)
    if ((long long)inst->n * (inst->n - 1) / 2 >= (1ll << 31)) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    tsp_dev_tabu *tb = new tsp_dev_tabu();
    tb->inst = inst;
    tb->count = (long long)inst->n * (inst->n - 1) / 2;
    struct Guard { tsp_dev_tabu *tb; ~Guard() { if (tb) tsp_dev_tabu_destroy(tb); } } guard{tb};
    TSP_HIP_TRY(hipMalloc(&tb->d_stamp, sizeof(int) * (size_t)std::max<long long>(1, tb->count)));
    TSP_HIP_TRY(hipMemsetAsync(tb->d_stamp, 0, sizeof(int) * (size_t)tb->count, inst->ctx->stream));
    tb->list_cap = (int)std::max<long long>(16, std::min<long long>(tb->count, 1ll << 18));
    TSP_HIP_TRY(hipMalloc(&tb->d_list, sizeof(int2) * (size_t)tb->list_cap));
    TSP_HIP_TRY(hipMalloc(&tb->d_list_n, sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&tb->d_tabu_pairs, 2 * kTabuSideWords * sizeof(unsigned long long)));   // the side words + their snapshot (CLUSTER engine, multi-launch runs)
    TSP_HIP_TRY(hipHostMalloc(&tb->h_list_n, sizeof(int)));
    TSP_HIP_TRY(hipMemsetAsync(tb->d_list_n, 0, sizeof(int), inst->ctx->stream));
    TSP_HIP_TRY(hipMemsetAsync(tb->d_tabu_pairs, 0, 2 * kTabuSideWords * sizeof(unsigned long long), inst->ctx->stream));
    TSP_HIP_TRY(hipStreamSynchronize(inst->ctx->stream));
    tb->list_valid = true;   // no stamp is set: the empty list is complete
    tb->list_ub = 0; tb->list_compact_at = kTabuCompactSlack;
    guard.tb = nullptr;
    *out = tb;
    return TSP_OK;
}

void tsp_dev_tabu_destroy(tsp_dev_tabu *tb) {
    if (!tb) return;
    (void)hipSetDevice(tb->inst->ctx->device);
    (void)hipStreamSynchronize(tb->inst->ctx->stream);
    (void)hipFree(tb->d_stamp); (void)hipFree(tb->d_list); (void)hipFree(tb->d_list_n); (void)hipFree(tb->d_tabu_pairs);
    (void)hipHostFree(tb->h_list_n);
    delete tb;
}

static int stamp_io(tsp_dev_tabu *tb, const int *idx, int *val, int count, bool scatter) {
    if (!tb || !idx || !val || count < 0) return TSP_DEV_E_ARG;
    if (count == 0) return TSP_OK;
    for (int k = 0; k < count; ++k)
        if (idx[k] < 0 || idx[k] >= tb->count) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(tb->inst->ctx->device));
    hipStream_t s = tb->inst->ctx->stream;
    int *d_idx = static_cast<int *>(tsp_io_pool(tb->inst, sizeof(int) * 2 * (size_t)count)), *d_val = d_idx + count;
    if (!d_idx) return TSP_DEV_E_NOMEM;
    TSP_HIP_TRY(hipMemcpyAsync(d_idx, idx, sizeof(int) * (size_t)count, hipMemcpyHostToDevice, s));
    if (scatter) {
        tb->list_valid = false;   // the list of non-zero stamps is rebuilt by a scan before the next run
        TSP_HIP_TRY(hipMemcpyAsync(d_val, val, sizeof(int) * (size_t)count, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_stamp_scatter, dim3((count + 255) / 256), dim3(256), 0, s, tb->d_stamp, d_idx, d_val, count);
    } else {
        hipLaunchKernelGGL(k_stamp_gather, dim3((count + 255) / 256), dim3(256), 0, s, tb->d_stamp, d_idx, d_val, count);
        TSP_HIP_TRY(hipMemcpyAsync(val, d_val, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost, s));
    }
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

int tsp_dev_tabu_set(tsp_dev_tabu *tb, const int *idx, const int *value, int count) {
    return stamp_io(tb, idx, const_cast<int *>(value), count, true);
}
int tsp_dev_tabu_get(tsp_dev_tabu *tb, const int *idx, int *value, int count) {
    return stamp_io(tb, idx, value, count, false);
}
int tsp_dev_tabu_list_info(tsp_dev_tabu *tb, int *entries, int *used_by_last_run) {
    if (!tb) return TSP_DEV_E_ARG;
    if (entries) *entries = tb->list_valid ? (int)tb->list_ub : -1;
    if (used_by_last_run) *used_by_last_run = tb->last_run_list ? 1 : 0;
    return TSP_OK;
}
int tsp_dev_tabu_upload(tsp_dev_tabu *tb, const int *stamps) {
    if (!tb || !stamps) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(tb->inst->ctx->device));
    tb->list_valid = false;
    TSP_HIP_TRY(hipMemcpyAsync(tb->d_stamp, stamps, sizeof(int) * (size_t)tb->count, hipMemcpyHostToDevice,
                               tb->inst->ctx->stream));
    TSP_HIP_TRY(hipStreamSynchronize(tb->inst->ctx->stream));
    return TSP_OK;
}
int tsp_dev_tabu_download(tsp_dev_tabu *tb, int *stamps) {
    if (!tb || !stamps) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(tb->inst->ctx->device));
    TSP_HIP_TRY(hipMemcpyAsync(stamps, tb->d_stamp, sizeof(int) * (size_t)tb->count, hipMemcpyDeviceToHost,
                               tb->inst->ctx->stream));
    TSP_HIP_TRY(hipStreamSynchronize(tb->inst->ctx->stream));
    return TSP_OK;
}

int tsp_dev_two_opt_tabu(tsp_dev_inst *inst, tsp_dev_tabu *tabu, int iter, int tenure, int *succ, int succ_stride,
                         double *obj, int *stored_prev, double time_limit_s, tsp_two_opt_stats *stats) {
    if (!inst || !succ || !obj || succ_stride < 1) return TSP_DEV_E_ARG;
    if (tabu && tabu->inst != inst) return TSP_DEV_E_ARG;
    const double t0 = wall_s();
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    bool owned = false;
    int rc = TSP_OK;
    tsp_dev_tours *t = tsp_scratch_tours(inst, 1, &owned, &rc);
    if (rc) return rc;
    rc = tsp_dev_tours_upload(t, succ, succ_stride, inst->n, obj);
    if (rc) return rc;
    int done = 0;
    const int status = tsp_tabu_run(t, tabu, iter, tenure, time_limit_s, 1, &done);
    if (status < 0) return status;
    rc = tsp_dev_tours_download(t, succ, succ_stride, inst->n, obj, stats);
    if (rc) return rc;
    if (stored_prev)  // tabusearch.c:173-175
        for (int v = 0; v < inst->n; ++v) stored_prev[succ[(size_t)v * succ_stride]] = v;
    if (stats) stats->seconds = wall_s() - t0;
    return status;
}

#ifdef TSP_STAMPS
// diagnostic: mean 100 MHz ticks per segment of the last block of a step; resets the sums
// diagnostic build: the timeline of the last exhaustive sweep (two_opt_exh.hpp), microseconds after the first wave's start:
// out[0] last wave start, [1] / [2] / [3] first / mean / last wave out of its rows, [4] last candidate published, [5] apply done,
// [6] shader clock during the rows (MHz), [7] waves seen
int tsp_dev_debug_exh_stamps(double *out8) {
    std::vector<unsigned long long> w(8192 * 4);
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(tsp::g_exh_w), w.size() * 8) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(tsp::g_exh_t), sizeof h) != hipSuccess) return -1;
    unsigned long long t0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
    double esum = 0, csum = 0, rsum = 0, hsum = 0, hcsum = 0;
    int nw = 0;
    for (int k = 0; k < 8192; ++k) {
        if (!w[4 * k + 3]) continue;
        ++nw;
        t0 = std::min(t0, w[4 * k]); s1 = std::max(s1, w[4 * k]);
        e0 = std::min(e0, w[4 * k + 1]); e1 = std::max(e1, w[4 * k + 1]);
        esum += (double)w[4 * k + 1]; csum += (double)(w[4 * k + 2] & 0xffffffffull); rsum += (double)(w[4 * k + 1] - w[4 * k]);
        hsum += (double)((w[4 * k + 3] >> 8) & 0xffff); hcsum += (double)((w[4 * k + 3] >> 24) & 0xfffffffffull);
    }
    if (!nw) return 0;
    out8[0] = (double)(s1 - t0) / 100.0; out8[1] = (double)(e0 - t0) / 100.0; out8[2] = (esum / nw - (double)t0) / 100.0;
    out8[3] = (double)(e1 - t0) / 100.0; out8[4] = ((double)h[4] - (double)t0) / 100.0; out8[5] = ((double)h[5] - (double)t0) / 100.0;
    out8[6] = rsum > 0 ? csum / rsum * 100.0 : 0.0; out8[7] = nw;
    out8[8] = hsum / nw; out8[9] = hcsum / nw; out8[10] = csum / nw;   // per wave: bookkeeping branches taken, shader cycles in them, cycles in the rows
    if (const char *dump = getenv("TSP_EXH_DUMP")) {   // per wave: exit time (us after the first start), XCC, HW_ID
        if (FILE *fp = fopen(dump, "w")) {
            for (int k = 0; k < 8192; ++k)
                if (w[4 * k + 3]) fprintf(fp, "%d %.2f %.2f %llu %llu\n", k, (double)(w[4 * k] - t0) / 100.0, (double)(w[4 * k + 1] - t0) / 100.0, w[4 * k + 3] >> 60, w[4 * k + 2] >> 32);
            fclose(fp);
        }
    }
    std::fill(w.begin(), w.end(), 0ull);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_exh_w), w.data(), w.size() * 8);
    const unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_exh_t), z, sizeof z);
    return nw;
}

int tsp_dev_debug_stamps(double *out16) {
    unsigned long long h[16], nn = 0;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(tsp::g_stamp_sum), sizeof h) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&nn, HIP_SYMBOL(tsp::g_stamp_n), sizeof nn) != hipSuccess) return -1;
    for (int k = 0; k < 16; ++k) out16[k] = nn ? (double)h[k] / (double)nn : 0.0;
    unsigned long long cc = 0, rr = 0;
    (void)hipMemcpyFromSymbol(&cc, HIP_SYMBOL(tsp::g_clk_core), sizeof cc);
    (void)hipMemcpyFromSymbol(&rr, HIP_SYMBOL(tsp::g_clk_real), sizeof rr);
    out16[15] = rr ? (double)cc / (double)rr * 100.0 : 0.0;   // MHz during the row loops
    out16[14] = nn ? (double)rr / 100.0 : 0.0;                // total row-loop block-microseconds
    unsigned long long z[16] = {0}, zn = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_stamp_sum), z, sizeof z);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_stamp_n), &zn, sizeof zn);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_clk_core), &zn, sizeof zn);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_clk_real), &zn, sizeof zn);
    return (int)nn;
}
#endif

}  // extern "C"
