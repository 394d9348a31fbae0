// two_opt_first.hpp -- first-improvement steps, second form: k_first
// Part of the GRID engine; included by two_opt_grid.hip only (one translation unit).
#pragma once
#include "two_opt_step.hpp"

#pragma clang fp contract(off)

namespace tsp {

// ---- first improvement (alg_2opt), second form: k_first ------------------------------------------------------
// Same decisions as k_step<FIRST> (first improving pair after the cursor in (i<j) order, heuristics.c:452-486),
// three things done differently, all about the latency of a step:
//  * the grid is fixed and small (gy = 8 tile rows by 512 columns for one tour); a block takes ceil(chunk / gy) rows.
//    No launch dispatches thousands of blocks that return at once, and -- arrivals on one word are served one
//    after the other, ~12 ns each -- few blocks take tickets, on two levels (per tile row, then per tour);
//  * the move is carried out by the NEXT launch, out of place, by the working blocks (MoveView, as in k_move_recs):
//    the scan reads the tour through the closed form of the pending reversal, the last block only notes the move;
//  * two control-block slots per tour (see the kernel): a launch never writes what its own blocks may still read.
template <int WT, bool INT>
__device__ __forceinline__ NodeRec load_node_view(const double2 *coord, const MoveView &mv, int n, int v) {
    int q = mv.pos_of(v) + 1;
    if (q == n) q = 0;
    const int s = mv.node_at(q);
    const double2 c = coord[v], cs = coord[s];
    NodeRec r;
    r.x = c.x; r.y = c.y; r.xs = cs.x; r.ys = cs.y;
    r.ds = dist_xy<WT, INT>(c.x, c.y, cs.x, cs.y);
    r.succ = s; r.id = v;
    return r;
}

template <int WT, bool INT, int RJ>
__global__ __launch_bounds__(kScanThreads) void k_first(const StepArgs a) {
    constexpr int TJ = kScanThreads * RJ;
    constexpr bool FILTER = has_root_filter<WT>();
#ifdef TSP_STAMPS
    __shared__ unsigned long long stamps[16];
#endif
    TSP_STAMP(0);
    const int tour = blockIdx.z, n = a.n, tid = threadIdx.x;
    // Two control blocks per tour: this launch reads slot a.slot, which nobody writes while it runs (a block
    // dispatched late must not mistake the next step's cursor for its own), and its last block writes the other.
    const TourState *st = a.states + (size_t)a.slot * gridDim.z + tour;
    TourState *st_out = a.states + (size_t)(1 - a.slot) * gridDim.z + tour;
    if (st->done) return;
    const size_t base = (size_t)tour * n;
    const MoveView mv = move_view(st, a.orders + base, a.poss + base, a.orders2 + base, a.poss2 + base, n);
    const int ci = st->ci, cj = st->cj;
    const int row_lo = ci, row_hi = min(ci + st->chunk_rows, n - 1);
    const int gx = gridDim.x, gy = gridDim.y;
    const int rpb = max(1, (row_hi - row_lo + gy - 1) / gy);   // rows per block in this step (<= kMaxRowsPerBlock)
    const int r0 = row_lo + (int)blockIdx.y * rpb;
    if (r0 >= row_hi) return;                 // beyond the active chunk
    const int r1 = min(r0 + rpb, row_hi);
    const int c0 = (int)blockIdx.x * TJ;
    if (c0 + TJ - 1 <= r0) return;            // every column <= every row: nothing with j > i, no ticket
    // The working blocks of this step number themselves (tile rows first):
    // `active` of them take a ticket, and block `widx` carries out slices widx, widx + active, ... of the pending
    // move -- only ticket holders touch the other copy, so all of it is written before the last block moves on.
    const int tile_rows = (row_hi - row_lo + rpb - 1) / rpb;
    int active = 0, widx = 0, rowblocks = 0;   // wave-uniform: a scalar loop over the (few) tile rows
    for (int by = 0; by < tile_rows; ++by) {
        const int mine = gx - skipped_in_tile_row(row_lo + by * rpb, gx, TJ);
        if (by < (int)blockIdx.y) widx += mine;
        if (by == (int)blockIdx.y) rowblocks = mine;
        active += mine;
    }
    widx += (int)blockIdx.x - skipped_in_tile_row(r0, gx, TJ);
    if (mv.L > 0) {
        int *o_new = (st->parity ? a.orders : a.orders2) + base, *p_new = (st->parity ? a.poss : a.poss2) + base;
        for (int k = widx * kScanThreads + tid; k < n; k += active * kScanThreads) {
            const int v = mv.node_at(k);
            o_new[k] = v;
            p_new[v] = k;
        }
    }

    __shared__ NodeRec s_rows[kMaxRowsPerBlock];
    if (tid < r1 - r0) s_rows[tid] = load_node_view<WT, INT>(a.coord, mv, n, r0 + tid);
    int jc[RJ];
    NodeRec rj[RJ];
#pragma unroll
    for (int k = 0; k < RJ; ++k) {
        jc[k] = c0 + tid + k * kScanThreads;
        rj[k] = load_node_view<WT, INT>(a.coord, mv, n, min(jc[k], n - 1));
        if (jc[k] >= n) jc[k] = -1;  // never > i
    }
    __syncthreads();
    TSP_STAMP(1);

    double bd = 0.0;
    int bi = -1, bj = -1;
    for (int ib = r0; ib < r1; ib += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(ib + u, r1 - 1);
            const NodeRec ri = s_rows[i - r0];
#pragma unroll
            for (int k = 0; k < RJ; ++k) {   // a lane's columns in increasing order: the first hit is the first in (i, j) order
                // heuristics.c:471 skip rule, the cursor, and (sqrt metrics) the bounds with bound = 0
                bool ok = ib + u < r1 && jc[k] > i && jc[k] != ri.succ && rj[k].succ != i && (i > ci || jc[k] > cj);
                if constexpr (FILTER) {
                    ok = ok & new_edge_can_improve<WT>(ri.x, ri.y, rj[k].x, rj[k].y, ri.ds + rj[k].ds + a.prune);
                    if (ok) ok = pair_delta_approx<WT>(ri, rj[k]) - a.margin < 0.0;
                }
                if (ok) {
                    const double delta = pair_delta<WT, INT>(ri, rj[k]);
                    if (delta < 0 && bi < 0) { bd = delta; bi = i; bj = jc[k]; }
                }
            }
        }
        if (__any(bi >= 0)) break;  // later rows only hold later pairs
    }

    TSP_STAMP(2);
    u64 key = make_key(bi, bj);
    __shared__ double s_d[kScanThreads / 64];
    __shared__ u64 s_k[kScanThreads / 64];
    __shared__ long long s_ll[kScanThreads / 64];
    __shared__ int s_last;
    block_argmin<false>(bd, key, s_d, s_k);
    TSP_STAMP(3);
    const Partial *part = a.partials + (size_t)tour * a.partial_per_tour;
    if (tid == 0) {
        publish_partial(a.partials + (size_t)tour * a.partial_per_tour + (size_t)blockIdx.y * gx + blockIdx.x, bd, key_i(key), key_j(key));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left this CU before the ticket
        TSP_STAMP(4);
        // arrivals per tile row first (one word per row, 256 B apart), then the rows on the tour's word: arrivals on
        // one word are served one after the other
        gi32 *rt = (gi32 *)(a.cl_tickets + ((size_t)tour * 64 + blockIdx.y) * 64);
        s_last = 0;
        if (__hip_atomic_fetch_add(rt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == rowblocks) {
            __hip_atomic_store(rt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int old = __hip_atomic_fetch_add((gi32 *)(a.tickets + tour), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old + 1 == tile_rows);
            if (s_last) __hip_atomic_store((gi32 *)(a.tickets + tour), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!s_last) return;
    TSP_STAMP(5);

    // ---- last block of the step: winner, counters, cursor, the move noted for the next launch
    const int nslots = tile_rows * gx;
    bd = 0.0;
    key = kNoKey;
    constexpr int PU = 4;
    for (int s0 = tid; s0 < nslots; s0 += PU * kScanThreads) {
        double pd[PU]; int pi[PU], pj[PU]; bool live[PU];
#pragma unroll
        for (int k = 0; k < PU; ++k) {
            const int sl = s0 + k * kScanThreads;
            const int by = sl / gx, bx = sl - by * gx;
            live[k] = sl < nslots && bx >= skipped_in_tile_row(row_lo + by * rpb, gx, TJ);
            pd[k] = 0.0; pi[k] = -1; pj[k] = -1;
            if (live[k]) read_partial(part + sl, pd[k], pi[k], pj[k]);
        }
#pragma unroll
        for (int k = 0; k < PU; ++k) {
            const u64 kk = make_key(pi[k], pj[k]);
            if (live[k] && kk < key) { bd = pd[k]; key = kk; }
        }
    }
    block_argmin<false>(bd, key, s_d, s_k);
    TSP_STAMP(6);
    const bool found = key != kNoKey;
    const int wi = found ? key_i(key) : -1, wj = found ? key_j(key) : -1;
    int pa = 0, pb = 0, L = 0;
    if (found) { pa = mv.pos_of(wi); pb = mv.pos_of(wj); L = pb - pa; if (L < 0) L += n; }
    // pairs between the old and the new cursor the reference would have skipped as adjacent (heuristics.c:471),
    // on the tour the scan saw: row r's adjacent columns are succ(r) and pred(r), when they are > r
    int ni = wi, nj = wj;  // new cursor
    if (!found) { ni = row_hi - 1; nj = n - 1; }
    long long adj = 0;
    if (a.count_evals) {
        const u64 lo = make_key(ci, cj), hi = make_key(ni, nj);
        long long c = 0;
        for (int r = ci + tid; r <= ni; r += kScanThreads) {
            const int p = mv.pos_of(r);
            const int sc = mv.node_at(p + 1 == n ? 0 : p + 1), q = mv.node_at(p == 0 ? n - 1 : p - 1);
            const u64 ks = make_key(r, sc), kq = make_key(r, q);
            c += (sc > r && ks > lo && ks <= hi) ? 1 : 0;
            c += (q > r && kq > lo && kq <= hi) ? 1 : 0;
        }
        adj = block_sum<long long>(c, s_ll);
    }
    TSP_STAMP(7);
    if (tid == 0) {
        int done = 0, n_ci = 0, n_cj = 0, n_chunk = st->chunk_rows, sweep_end = 0;
        double obj = st->obj, seen = st->seen_cost;
        if (found) {
            obj += bd;                               // heuristics.c:486
            n_ci = wi; n_cj = wj; n_chunk = a.first_min_rows;
        } else {
            n_chunk = min(st->chunk_rows * 2, a.first_max_rows);
            if (row_hi >= n - 1) {                   // sweep complete
                sweep_end = 1;
                if (obj >= seen) done = 1;           // heuristics.c:492
                else { seen = obj; n_ci = 0; n_cj = 0; }
            } else { n_ci = row_hi - 1; n_cj = n - 1; }
        }
        const long long r_old = pair_rank(ci, cj, n);
        TourState z = *st;
        z.steps += 1;
        z.pairs_scanned += pair_rank(row_hi - 1, n - 1, n) - r_old;
        z.evals += pair_rank(ni, nj, n) - r_old - adj;
        if (found) { z.moves += 1; z.reversed += L - 1; }   // successors rewritten by utility.c:710-717
        z.sweeps += sweep_end;
        z.ci = n_ci; z.cj = n_cj; z.chunk_rows = n_chunk; z.seen_cost = seen;
        z.obj = obj;
        z.done = done;
        z.parity = st->parity ^ (mv.L > 0 ? 1 : 0);   // this launch has filled the other copy
        z.pending = found ? 1 : 0; z.mv_pa = pa; z.mv_pb = pb;
        *st_out = z;
        // finished: later launches alternate between the slots and must find `done` in both (a block of this
        // launch that reads it now returns, as it would have anyway: every working block is past its ticket)
        if (done) *const_cast<TourState *>(st) = z;
#ifdef TSP_STAMPS
        stamps[8] = wall_clock64(); stamps[9] = stamps[8];
        for (int k = 1; k < 10; ++k) atomicAdd(&g_stamp_sum[k], stamps[k] - stamps[k - 1]);
        atomicAdd(&g_stamp_n, 1ull);
#endif
    }
}

}  // namespace tsp
