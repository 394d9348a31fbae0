// two_opt_tiled.hpp -- the tiled step kernels: k_recs, k_arm, k_step (every pair of the scanned range visited)
// Part of the GRID engine; included by two_opt_grid.hip only (one translation unit).
#pragma once
#include "two_opt_step.hpp"

#pragma clang fp contract(off)

namespace tsp {

// ---- node records of a whole tour (BEST sweeps) ------------------------------------------------------------
// A BEST sweep has ~n^2 / (rows x columns) tiles and every tile needs the NodeRec of its rows and columns:
// deriving them per tile costs ~n^2 / 32 scattered gathers per sweep, which became the bottleneck once the
// pair loop was pruned.  So each BEST step first materialises all n records (one small launch, 4 n gathers),
// and the tiles read them as contiguous 48-byte loads.  FIRST steps touch few tiles and keep deriving.
template <int WT, bool INT>
__global__ __launch_bounds__(kScanThreads) void k_recs(const double2 *__restrict__ coord, const int *__restrict__ orders,
                                                       const int *__restrict__ poss, const TourState *__restrict__ states,
                                                       NodeRec *__restrict__ recs, int n) {
    const int tour = blockIdx.y;
    if (states[tour].done) return;
    const int v = blockIdx.x * kScanThreads + threadIdx.x;
    if (v >= n) return;
    const size_t base = (size_t)tour * n;
    recs[base + v] = load_node<WT, INT>(coord, orders + base, poss + base, n, v);
}

// Arms the tickets for the first step of a run (later steps are armed by the apply).
// FIRST: per-tour countdown of the active blocks.  BEST: count-up tickets (per tile row, then per
// tour) start at zero.
template <int MODE>
__global__ __launch_bounds__(kScanThreads) void k_arm(const TourState *__restrict__ states, int *__restrict__ tickets,
                                                      int *__restrict__ row_tickets, int max_tile_rows,
                                                      int n, int rpb, int gx, int gy, int TJ) {
    __shared__ int s_i[kScanThreads / 64];
    const TourState *st = states + blockIdx.x;
    if constexpr (MODE == TSP_2OPT_BEST) {
        for (int k = threadIdx.x; k < max_tile_rows; k += kScanThreads) row_tickets[(size_t)blockIdx.x * max_tile_rows + k] = 0;
        if (threadIdx.x == 0) tickets[blockIdx.x] = 0;
    } else {
        int row_lo, row_hi;
        active_rows<MODE>(st, n, row_lo, row_hi);
        const int c = count_active_blocks(row_lo, row_hi, rpb, gx, gy, TJ, s_i);
        if (threadIdx.x == 0) tickets[blockIdx.x] = st->done ? 0 : c;
    }
}

// ---- step kernel ------------------------------------------------------------------------------
// Block (bx, by, tour): rows r0 .. r0+rows_per_block of the tour's active row range, columns
// bx*256*RJ .. +256*RJ.  Prologue: the block derives the NodeRec of its rows (into LDS) and of its
// columns (RJ per lane, registers) from order/pos/coord -- three dependent loads and one sqrt per
// node, amortised over rows x columns evaluations.  Main loop: lanes own columns, the row record
// is a wave-uniform LDS broadcast; ~70 fp64 instructions per evaluation, no memory traffic.
// TLIST: a best-improvement run with a tabu list worked from its non-zero entries (two_opt_tabu_list.hpp) on a tour outside
// the sorted sweep (metrics without the bound): a pair that would become a lane's best goes through the check_tenure chain
// first; the scan's side effects come from k_tabu_side, launched in front of every step.
template <int WT, bool INT, int MODE, int RJ, bool TABU, bool TLIST = false>
__global__ __launch_bounds__(kScanThreads) void k_step(const StepArgs a) {
    static_assert(!TLIST || (!TABU && MODE == TSP_2OPT_BEST), "the list variant replaces the per-pair look-ups of a best-improvement sweep");
    constexpr int TJ = kScanThreads * RJ;
#ifdef TSP_STAMPS
    __shared__ unsigned long long stamps[16];
#endif
    TSP_STAMP(0);
    const int tour = blockIdx.z;
    const int n = a.n;
    const TourState *st = a.states + tour;
    if (st->done) return;
    int row_lo, row_hi, ci = -1, cj = -1;
    active_rows<MODE>(st, n, row_lo, row_hi);
    if constexpr (MODE == TSP_2OPT_FIRST) { ci = st->ci; cj = st->cj; }
    const int r0 = row_lo + blockIdx.y * a.rows_per_block;
    if (r0 >= row_hi) return;                 // beyond the active chunk
    const int r1 = min(r0 + a.rows_per_block, row_hi);
    const int c0 = blockIdx.x * TJ;
    if (c0 + TJ - 1 <= r0) return;            // every column <= every row: nothing with j > i, no ticket
    const size_t slot_idx = (size_t)tour * a.partial_per_tour + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int tid = threadIdx.x;
    const int *order = a.orders + (size_t)tour * n;
    const int *pos = a.poss + (size_t)tour * n;

    __shared__ NodeRec s_rows[kMaxRowsPerBlock];
    int jc[RJ];
    NodeRec rj[RJ];
    if (MODE == TSP_2OPT_BEST && a.recs) {
        const NodeRec *rec = a.recs + (size_t)tour * n;
        if (tid < r1 - r0) s_rows[tid] = rec[r0 + tid];
#pragma unroll
        for (int k = 0; k < RJ; ++k) {
            jc[k] = c0 + tid + k * kScanThreads;
            rj[k] = rec[min(jc[k], n - 1)];
            if (jc[k] >= n) jc[k] = -1;  // never > i
        }
    } else {
        if (tid < r1 - r0) s_rows[tid] = load_node<WT, INT>(a.coord, order, pos, n, r0 + tid);
#pragma unroll
        for (int k = 0; k < RJ; ++k) {
            jc[k] = c0 + tid + k * kScanThreads;
            rj[k] = load_node<WT, INT>(a.coord, order, pos, n, min(jc[k], n - 1));
            if (jc[k] >= n) jc[k] = -1;  // never > i
        }
    }
    __syncthreads();
    TSP_STAMP(1);
#ifdef TSP_STAMPS
    const unsigned long long clk0 = clock64(), rt0 = wall_clock64();
#endif

    double bd = 0.0;
    int bi = -1, bj = -1;
    int n_eval = 0;  // TABU: pairs that reach the delta expression (tabusearch.c:150)
    // rows in groups of RU: RU x RJ independent delta evaluations per lane keep the fp64 pipe fed
    // (the root refinement is a long dependent chain); FIRST leaves after the group with a hit.
    // Sqrt metrics: every pair first gets the raw-root lower bound (straight-line code, all RJ columns
    // interleaved); the exact evaluation runs under one divergent branch for the few lanes that need it.
    constexpr int RU = (MODE == TSP_2OPT_FIRST) ? 4 : 1;
    constexpr bool FILTER = has_root_filter<WT>();
    // Interior tiles of a BEST sweep on integer-valued costs need no per-pair predicate at all: every column
    // is a valid node above every row of the tile, and an adjacent pair has delta == 0 exactly
    // (d(a,b) = d(a,a1), d(a1,b1) = d(b,b1) and integer sums are exact), which the strict '<' never takes
    // (heuristics.c:471 / tabusearch.c:134 exist to skip exactly those).  Non-integer costs keep the test:
    // there (x + y) - x - y can round to a tiny negative.
    constexpr bool EXACT_SUMS = INT || WT == WT_CEIL_2D || WT == WT_CEIL_2D_ICOORD;
    const bool plain_tile = MODE == TSP_2OPT_BEST && !TABU && !TLIST && FILTER && EXACT_SUMS && c0 >= r1 && c0 + TJ <= n;
    // bounds switched off (TSP_NO_FILTER=1, margins 1e300): every delta expression is executed, and nothing else -- no
    // bound arithmetic that could never exclude a pair.  This is the exhaustive sweep bench.py's roofline.exhaustive times.
    const bool exhaustive = MODE == TSP_2OPT_BEST && !TABU && !TLIST && FILTER && a.margin > 1e299;
    if (exhaustive) {
        for (int ib = r0; ib < r1; ib += 2) {   // two rows in flight: the root is a long dependent chain
            const int i0 = ib, i1 = min(ib + 1, r1 - 1);
            const NodeRec ra = s_rows[i0 - r0], rb = s_rows[i1 - r0];
            const bool two = ib + 1 < r1;
#pragma unroll
            for (int k = 0; k < RJ; ++k) {
                const double da = pair_delta<WT, INT>(ra, rj[k]), db = pair_delta<WT, INT>(rb, rj[k]);
                const int j = jc[k];
                const bool oka = plain_tile || (j > i0 && j != ra.succ && rj[k].succ != i0);   // heuristics.c:471 / tabusearch.c:134
                const bool okb = two && (plain_tile || (j > i1 && j != rb.succ && rj[k].succ != i1));
                if (oka && da < bd) { bd = da; bi = i0; bj = j; }
                if (okb && db < bd) { bd = db; bi = i1; bj = j; }
            }
        }
    } else if (plain_tile) {
        for (int i = r0; i < r1; ++i) {
            const NodeRec ri = s_rows[i - r0];
            const double row_bias = ri.ds + a.margin;
            const double row_t = ri.ds + bd + a.prune;   // a stale (larger) bd only prunes less
            bool need[RJ];
            bool any = false;
#pragma unroll
            for (int k = 0; k < RJ; ++k) {
                need[k] = new_edge_can_improve<WT>(ri.x, ri.y, rj[k].x, rj[k].y, row_t + rj[k].ds);
                any = any || need[k];
            }
            if (any) {
                bool any2 = false;
#pragma unroll
                for (int k = 0; k < RJ; ++k) {
                    const double lower = approx_root_dist<WT>(ri.x, ri.y, rj[k].x, rj[k].y) +
                                         approx_root_dist<WT>(ri.xs, ri.ys, rj[k].xs, rj[k].ys) - row_bias - rj[k].ds;
                    need[k] = need[k] & (lower < bd);
                    any2 = any2 || need[k];
                }
                if (any2) {
#pragma unroll
                    for (int k = 0; k < RJ; ++k) {
                        const double delta = pair_delta<WT, INT>(ri, rj[k]);
                        if (need[k] && delta < bd) { bd = delta; bi = i; bj = jc[k]; }
                    }
                }
            }
        }
    } else
    for (int ib = r0; ib < r1; ib += RU) {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int i = min(ib + u, r1 - 1);
            const bool row_ok = ib + u < r1;
            const NodeRec ri = s_rows[i - r0];
            bool ok[RJ];
            bool any_ok = false;
#pragma unroll
            for (int k = 0; k < RJ; ++k) {
                const int j = jc[k];
                ok[k] = row_ok && j > i && j != ri.succ && rj[k].succ != i;  // heuristics.c:471 / tabusearch.c:134
                if constexpr (MODE == TSP_2OPT_FIRST) ok[k] = ok[k] && (i > ci || j > cj);
                if constexpr (TABU) {
                    if (ok[k]) {
                        const int a1 = ri.succ, b1 = rj[k].succ;
                        if (stamp_is_tabu(a.tabu + udir_pos(i, j, n), a.iter, a.tenure) ||
                            stamp_is_tabu(a.tabu + udir_pos(i, a1, n), a.iter, a.tenure) ||
                            stamp_is_tabu(a.tabu + udir_pos(j, b1, n), a.iter, a.tenure) ||
                            stamp_is_tabu(a.tabu + udir_pos(i, b1, n), a.iter, a.tenure))
                            ok[k] = false;  // tabusearch.c:137-149
                    }
                    n_eval += ok[k] ? 1 : 0;
                }
                if constexpr (FILTER) {
                    // the new edge alone must be short enough to pay for the two removed edges
                    const double bound = (MODE == TSP_2OPT_FIRST) ? 0.0 : bd;
                    ok[k] = ok[k] & new_edge_can_improve<WT>(ri.x, ri.y, rj[k].x, rj[k].y, bound + ri.ds + rj[k].ds + a.prune);
                }
                any_ok = any_ok || ok[k];
            }
            if constexpr (FILTER) {
                if (any_ok) {
                    // survivors: a pair whose raw-root delta cannot get below the bound is not evaluated exactly
                    any_ok = false;
                    const double bound = (MODE == TSP_2OPT_FIRST) ? 0.0 : bd;
#pragma unroll
                    for (int k = 0; k < RJ; ++k) {
                        ok[k] = ok[k] & (pair_delta_approx<WT>(ri, rj[k]) - a.margin < bound);
                        any_ok = any_ok || ok[k];
                    }
                }
            }
            if (!FILTER || any_ok) {
#pragma unroll
                for (int k = 0; k < RJ; ++k) {
                    const double delta = pair_delta<WT, INT>(ri, rj[k]);
                    if constexpr (MODE == TSP_2OPT_FIRST) {
                        if (ok[k] && delta < 0 && bi < 0) { bd = delta; bi = i; bj = jc[k]; }  // keep the first in (i, j) order
                    } else if constexpr (TLIST) {
                        if (ok[k] && delta < bd) {   // tabusearch.c:137-149 (i < jc[k]: a = i), lazy clears included
                            const int j = jc[k], a1 = ri.succ, b1 = rj[k].succ;
                            const bool tb = stamp_is_tabu(a.tabu + udir_pos(i, j, n), a.iter, a.tenure) ||
                                            stamp_is_tabu(a.tabu + udir_pos(i, a1, n), a.iter, a.tenure) ||
                                            stamp_is_tabu(a.tabu + udir_pos(j, b1, n), a.iter, a.tenure) ||
                                            stamp_is_tabu(a.tabu + udir_pos(i, b1, n), a.iter, a.tenure);
                            if (!tb) { bd = delta; bi = i; bj = j; }
                        }
                    } else {
                        if (ok[k] && delta < bd) { bd = delta; bi = i; bj = jc[k]; }
                    }
                }
            }
        }
        if constexpr (MODE == TSP_2OPT_FIRST) {
            if (__any(bi >= 0)) break;  // later rows only hold later pairs
        }
    }

    u64 key = make_key(bi, bj);
    __shared__ double s_d[kScanThreads / 64];
    __shared__ u64 s_k[kScanThreads / 64];
    __shared__ int s_cnt[kScanThreads / 64];
    __shared__ int s_last;
#ifdef TSP_STAMPS
    if (tid == 0) { atomicAdd(&g_clk_core, clock64() - clk0); atomicAdd(&g_clk_real, wall_clock64() - rt0); }
#endif
    TSP_STAMP(2);
    block_argmin<MODE == TSP_2OPT_BEST>(bd, key, s_d, s_k);
    TSP_STAMP(3);
    int tot_eval = 0;
    if constexpr (TABU) tot_eval = block_sum<int>(n_eval, s_cnt);
    constexpr bool HIER = MODE == TSP_2OPT_BEST;
    const int skipped = skipped_in_tile_row(r0, gridDim.x, TJ);
    if (tid == 0) {
        publish_partial(a.partials + slot_idx, bd, key_i(key), key_j(key));
        if constexpr (TABU)
            __hip_atomic_store((gi32 *)(a.slot_evals + slot_idx), tot_eval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left this CU before the ticket
        TSP_STAMP(4);
        if constexpr (HIER) {
            const int old = __hip_atomic_fetch_add((gi32 *)(a.row_tickets + (size_t)tour * a.max_tile_rows + blockIdx.y), 1,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old + 1 == (int)gridDim.x - skipped);
        } else {
            const int old = __hip_atomic_fetch_sub((gi32 *)(a.tickets + tour), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == 1);
        }
    }
    __syncthreads();
    if constexpr (HIER) {
        if (!s_last) return;
        __syncthreads();   // everyone has read s_last before it is rewritten
        // last block of this tile row: its first wave folds the row's candidates into one
        if (tid < 64) {
            double d = 0.0;
            u64 k2 = kNoKey;
            int ev = 0;
            const size_t row_base = (size_t)tour * a.partial_per_tour + (size_t)blockIdx.y * gridDim.x;
            for (int bx = skipped + tid; bx < (int)gridDim.x; bx += 64) {
                double pd; int pi, pj;
                read_partial(a.partials + row_base + bx, pd, pi, pj);
                const u64 kk = make_key(pi, pj);
                if (better(pd, kk, d, k2)) { d = pd; k2 = kk; }
                if constexpr (TABU)
                    ev += __hip_atomic_load((gi32 *)(a.slot_evals + row_base + bx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            wave_argmin<true>(d, k2);
            if constexpr (TABU) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) ev += __shfl_xor(ev, off);
            }
            if (tid == 0) {
                const size_t ridx = (size_t)tour * a.max_tile_rows + blockIdx.y;
                publish_partial(a.row_slots + ridx, d, key_i(k2), key_j(k2));
                if constexpr (TABU)
                    __hip_atomic_store((gi32 *)(a.row_evals + ridx), ev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int old = __hip_atomic_fetch_add((gi32 *)(a.tickets + tour), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = (old + 1 == (int)gridDim.y);
                if (s_last) __hip_atomic_store((gi32 *)(a.tickets + tour), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
    if (!s_last) return;
    TSP_STAMP(5);
#ifdef TSP_STAMPS
    apply_step<WT, INT, MODE, RJ, TABU>(a, tour, row_lo, row_hi, stamps);   // TLIST: as without a list (k_tabu_fix_evals takes the skipped pairs off)
#else
    apply_step<WT, INT, MODE, RJ, TABU>(a, tour, row_lo, row_hi);
#endif
}

}  // namespace tsp
