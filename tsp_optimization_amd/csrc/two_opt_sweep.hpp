// two_opt_sweep.hpp -- the sorted best-improvement sweep: k_move_recs, k_sweep, the flush kernels
// Part of the GRID engine; included by two_opt_grid.hip only (one translation unit).
#pragma once
#include "two_opt_step.hpp"
#include "two_opt_tabu_list.hpp"

#pragma clang fp contract(off)

namespace tsp {

// ---- sorted sweep (BEST, sqrt metrics) ------------------------------------------------------------------------
// The new-edge bound says a pair (a, b) can only beat `bound` if |ab| < bound + d(a,succ a) + d(b,succ b) + margin.
// With the nodes ranked along a Hilbert curve, 64 consecutive ranks form a compact group, and the same bound
// with the groups' bounding boxes and their longest tour edges decides 64 x 64 pairs at once: on a constructed
// tour 80-95 % of the group pairs of a sweep never reach the pair loop.  Nothing about the result changes:
// every decision the reference takes (strict '<', first pair in (i<j) order among equal deltas) is taken on
// exact values with the nodes' own ids; the order in which pairs are visited is free in a best-improvement sweep.
//
// k_move_recs: (1) the pending move, out of place; (2) the record of every node in rank order for the sweep that
// follows, on the tour AFTER that move; (3) each group's longest edge.  Writes no control state: the sweep's last
// block notes that the other copy is current from now on (apply_step, FLAT).
template <int WT, bool INT>
__global__ __launch_bounds__(kScanThreads) void k_move_recs(const double2 *__restrict__ coord, int *orders, int *poss,
                                                            int *orders2, int *poss2, const TourState *__restrict__ states,
                                                            const int *__restrict__ sperm, NodeRec *__restrict__ recs,
                                                            double *__restrict__ gmax, int n, int ng, int n_slots) {
    const int tour = blockIdx.y;
    const TourState *st = states + tour;
    if (st->done) return;
    const size_t base = (size_t)tour * n;
    const MoveView mv = move_view(st, orders + base, poss + base, orders2 + base, poss2 + base, n);
    const int k = blockIdx.x * kScanThreads + threadIdx.x;
    // every load (of the current copy, which this kernel never writes) comes before the first store: the compiler
    // cannot know that the two copies do not overlap and would otherwise finish the move before starting the records
    const bool copies = mv.L > 0 && k < n;
    const int moved = copies ? mv.node_at(k) : 0;   // node at new position k
    const int v = k < n_slots ? sperm[k] : -1;
    NodeRec r;
    if (v >= 0) {
        int ps = mv.pos_of(v) + 1; if (ps == n) ps = 0;
        const int sc = mv.node_at(ps);
        const double2 c = coord[v], cs = coord[sc];
        r.x = c.x; r.y = c.y; r.xs = cs.x; r.ys = cs.y;
        r.ds = dist_xy<WT, INT>(c.x, c.y, cs.x, cs.y);
        r.succ = sc; r.id = v;
    } else {   // padding: far away from everything, never passes the new-edge test
        r.x = r.y = r.xs = r.ys = 1e30; r.ds = 0.0; r.succ = -1; r.id = -1;
    }
    if (copies) {
        int *o_new = (st->parity ? orders : orders2) + base, *p_new = (st->parity ? poss : poss2) + base;
        o_new[k] = moved;
        p_new[moved] = k;
    }
    if (k >= n_slots) return;
    recs[(size_t)tour * n_slots + k] = r;
    // the group's longest edge: lengths are >= 0, so they order like their bits and max = ~min(~bits)
    const u64 mb = ~wave_min_u64(~(u64)__double_as_longlong(r.ds));
    if ((threadIdx.x & 63) == 0) gmax[(size_t)tour * (ng + 1) + (k >> 6)] = __longlong_as_double((long long)mb);
}

// End of a run through the sorted sweep: bring the tour back into the first copy of order/pos, where every other
// path expects it.  Three tiny launches, each reading a control block nobody writes meanwhile.
__global__ void k_flush_move(int *orders, int *poss, int *orders2, int *poss2, const TourState *__restrict__ states, int n) {
    const int tour = blockIdx.y;
    const TourState *st = states + tour;
    if (!st->pending) return;
    const size_t base = (size_t)tour * n;
    const MoveView mv = move_view(st, orders + base, poss + base, orders2 + base, poss2 + base, n);
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int *o_new = (st->parity ? orders : orders2) + base, *p_new = (st->parity ? poss : poss2) + base;
    const int v = mv.node_at(k);
    o_new[k] = v;
    p_new[v] = k;
}
__global__ void k_flush_copy(int *orders, int *poss, const int *orders2, const int *poss2,
                             const TourState *__restrict__ states, int n) {
    const int tour = blockIdx.y;
    const TourState *st = states + tour;
    if ((st->parity ^ st->pending) == 0) return;   // the tour already sits in the first copy
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const size_t base = (size_t)tour * n;
    orders[base + k] = orders2[base + k];
    poss[base + k] = poss2[base + k];
}
__global__ void k_flush_state(TourState *states, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { states[b].parity = 0; states[b].pending = 0; }
}

// v of lane l (l wave-uniform) in every lane, through two v_readlane_b32
__device__ __forceinline__ double lane_bcast(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// Group pair number t (0 <= t < ng (ng + 1) / 2, rows first) -> (row group r, column group c >= r).
__device__ __forceinline__ void group_pair(int t, int ng, int &r, int &c) {
    const double b = 2.0 * ng + 1.0;
    int rr = (int)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    rr = max(0, min(rr, ng - 1));
    // first pair of row group r: off(r) = r ng - r (r - 1) / 2
    while (rr > 0 && (long long)rr * ng - (long long)rr * (rr - 1) / 2 > t) --rr;
    while ((long long)(rr + 1) * ng - (long long)(rr + 1) * rr / 2 <= t) ++rr;
    r = rr;
    c = rr + (int)(t - ((long long)rr * ng - (long long)rr * (rr - 1) / 2));
}

constexpr int kSweepCluster = 4;      // blocks that test the same group pairs and deal the survivors among themselves (2-4 measured best)
constexpr int kSweepRows = 16;        // rows of one unit of wave work (64 / kSweepRows units per group pair)
constexpr int kSweepStage = 8;        // group pairs whose records a block holds in LDS at a time (6 KB each)
constexpr int kSweepListCap = 1024;   // survivors one block can hold (more are processed in further passes)

// k_sweep.  Blocks come in clusters of kSweepCluster.  The group pairs (row group r, column group c >= r; c == r: the
// pairs inside the group) are listed by box distance in a host-built table and dealt to the clusters in turn; every
// block of a cluster runs the same box tests, one per thread and round, so all of them see the same ordered survivor
// list and block j keeps ranks j, j + C, ...  The survivors of a sweep are very unevenly spread (a group that holds
// one long edge survives against everything, near pairs always survive and cost the most): the table plus the deal
// spreads them evenly over the chip without a queue or a second launch.
// The 128 records of each of a block's pairs are fetched in one burst into LDS; a wave takes kSweepRows rows of a
// pair at a time against one column per lane: rows that cannot reach the column box are dropped by one ballot, the
// others go four per step with their x, y, edge length broadcast out of registers (v_readlane), so the common path
// (tier 0) touches neither memory nor LDS.  The waves of a block share nothing until the block's arg-min.
// TABU: a call with a tabu list (two_opt_tabu_list.hpp): a pair that would become a lane's best goes through the
// reference's check_tenure chain first, and the workgroups past a.flat_slots reproduce the scan's side effects.
template <int WT, bool INT, bool TABU = false>
__global__ __launch_bounds__(kScanThreads, 3) void k_sweep(const StepArgs a) {   // 3 waves per SIMD: 768 blocks resident
    constexpr bool ATT10 = WT == WT_ATT || WT == WT_ATT_ICOORD;
    constexpr int NW = kScanThreads / 64;
#ifdef TSP_STAMPS
    __shared__ unsigned long long stamps[16];
#endif
    TSP_STAMP(0);
    const int tour = blockIdx.z;
    const TourState *st = a.states + tour;
    if (st->done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ng = a.ng;
    const NodeRec *rec = a.recs + (size_t)tour * a.n_slots;
    const double *gmax = a.gmax + (size_t)tour * (ng + 1);
    const double prune2 = 2.0 * a.prune;   // doubled: keeps ties (a lane does not meet its pairs in key order)
    const int Q = a.flat_slots / kSweepCluster, q = (int)blockIdx.x / kSweepCluster, j = (int)blockIdx.x % kSweepCluster;
    const int npairs = ng * (ng + 1) / 2;
    const int ntests = (npairs + Q - 1) / Q;   // stride blocks of group pairs; the last one may be partial

    __shared__ NodeRec s_stage[kSweepStage][128];   // per staged group pair: 64 row records, 64 column records
    __shared__ double4 s_cbox[kSweepStage];
    __shared__ double s_cgmax[kSweepStage];
    __shared__ int s_list[kSweepListCap];   // r << 16 | c
    __shared__ int s_wcount[NW];
    if constexpr (TABU) {
        if ((int)blockIdx.x >= a.flat_slots) {
            // k_move_recs has carried the pending move out: the other copy is the tour this sweep scans
            const bool second = (st->parity ^ st->pending) != 0;
            const size_t base = (size_t)tour * a.n;
            const TabuTour tt{(second ? a.orders2 : a.orders) + base, (second ? a.poss2 : a.poss) + base, a.n};
            const TabuView tv{a.tabu, a.n, a.iter, a.tenure};
            tabu_side<kScanThreads>(reinterpret_cast<int *>(&s_stage[0][0]), tt, tv, a.tabu_list, min(*a.tabu_list_n, a.tabu_list_cap), (int)blockIdx.x - a.flat_slots,
                                    (int)gridDim.x - a.flat_slots, a.tabu_pairs);   // tabu_pairs: the handle's three side words
            return;
        }
    }
    double bd = 0.0;
    u64 key = kNoKey;

    int m0 = 0;
    int seen = 0;          // survivors of the cluster so far (same in every block of the cluster)
    while (m0 < ntests) {
        // ---- tests: rounds of one group pair per thread until the block's list may be full or the pairs run out
        // (the pass ends on a condition every block of the cluster evaluates alike, or their ranks would part)
        int kept = 0;      // entries in s_list (same value in every thread)
        const int seen0 = seen;
        int e_first = -1;   // the first round's table entry is on its way while the control block is read
        if (a.pairtab && m0 + tid < ntests) e_first = a.pairtab[(size_t)q * ntests + m0 + tid];
        bool first_round = true;
        while (m0 < ntests && (seen - seen0) / kSweepCluster + kScanThreads / kSweepCluster + 2 <= kSweepListCap) {
            const int m = m0 + tid;
            bool surv = false;
            int r = 0, c = 0;
            bool valid = false;
            if (a.pairtab) {
                // host-built table: the group pairs in order of box distance, dealt to the clusters in turn, so that
                // every cluster (and, rank by rank, every block of it) gets its share of the near pairs, which
                // always survive and cost the most
                const int e = first_round ? e_first : (m < ntests ? a.pairtab[(size_t)q * ntests + m] : -1);
                valid = e >= 0; r = e >> 16; c = e & 0xffff;
            } else if (m < ntests && Q * m + (q + 29 * m) % Q < npairs) {
                // pair number: stride Q with a rotation per stride block (a plain stride would hand a cluster a
                // lattice in (r, c) that can sit on the diagonal, where every pair survives)
                group_pair(Q * m + (q + 29 * m) % Q, ng, r, c);
                valid = true;
            }
            first_round = false;
            if (valid) {
                const double4 rb = a.gbox[r], cb = a.gbox[c];
                const double gx = fmax(0.0, fmax(rb.x - cb.y, cb.x - rb.y)), gy = fmax(0.0, fmax(rb.z - cb.w, cb.z - rb.w));
                // bound 0: nothing is known about this sweep yet
                const double T = gmax[r] + gmax[c] + prune2;
                surv = gx * gx + gy * gy < (ATT10 ? 10.0 * T * T : T * T);
            }
            const unsigned long long bal = __ballot(surv);
            if (lane == 0) s_wcount[wave] = __popcll(bal);
            __syncthreads();
            int before = seen, total = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const int cw = s_wcount[w]; before += (w < wave) ? cw : 0; total += cw; }
            const int rank = before + __popcll(bal & ((1ull << lane) - 1ull));   // place in the cluster's list
            // entries of this block among ranks [seen, seen + total): those with rank % C == j
            const int first_mine = seen + ((j - seen % kSweepCluster + kSweepCluster) % kSweepCluster);
            if (surv && rank % kSweepCluster == j) s_list[kept + (rank - first_mine) / kSweepCluster] = (r << 16) | c;
            kept += (seen + total > first_mine) ? (seen + total - first_mine + kSweepCluster - 1) / kSweepCluster : 0;
            seen += total;
            m0 += kScanThreads;
            __syncthreads();
        }

        TSP_STAMP(1);
        // ---- the block's survivors, kSweepStage group pairs at a time: all 128 records of each pair are fetched
        // by the whole block in one burst (one memory latency per chunk instead of one per unit of wave work),
        // then every wave takes 16 rows of a group pair at a time out of LDS
        for (int e0 = 0; e0 < kept; e0 += kSweepStage) {
            const int ne = min(kSweepStage, kept - e0);
            if (tid < ne) {   // the column groups' boxes and longest edges, for the row culling below
                const int c = s_list[e0 + tid] & 0xffff;
                s_cbox[tid] = a.gbox[c];
                s_cgmax[tid] = gmax[c];
            }
            {   // a record is three 16-byte pieces; all loads of a thread are issued before its first LDS store
                constexpr int PER = kSweepStage * 128 * 3 / kScanThreads;
                const double2 *src = reinterpret_cast<const double2 *>(rec);
                double2 *dst = reinterpret_cast<double2 *>(&s_stage[0][0]);
                double2 tmp[PER];
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int x = tid + k * kScanThreads;      // piece x of the chunk
                    const int rcd = x / 3, part = x - rcd * 3;  // record 0 .. ne * 128 - 1
                    tmp[k] = make_double2(0.0, 0.0);
                    if (rcd < ne * 128) {
                        const int e = s_list[e0 + (rcd >> 7)];
                        const int g = (rcd & 64) ? (e & 0xffff) : (e >> 16);   // 0..63 rows of r, 64..127 columns of c
                        tmp[k] = src[(g * 64 + (rcd & 63)) * 3 + part];
                    }
                }
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int x = tid + k * kScanThreads;
                    if (x < ne * 128 * 3) dst[x] = tmp[k];
                }
            }
            __syncthreads();
        constexpr int UPP = 64 / kSweepRows;   // units per group pair
        for (int ht = wave; ht < UPP * ne; ht += NW) {
            const int pe = ht / UPP;             // staged pair
            const int e = s_list[e0 + pe];
            const int r = e >> 16, cgp = e & 0xffff, row0 = (ht % UPP) * kSweepRows;
            const NodeRec *rows = &s_stage[pe][row0];
            const NodeRec rj = s_stage[pe][64 + lane];
            const double cds = rj.ds + prune2, cds2 = rj.ds + a.sum_margin;
            double bound = bd;   // the lane's own best so far
            // Rows that cannot reach the column group's box at all are dropped for the whole wave (one row per
            // lane, one ballot): about 60 % of the rows of a surviving group pair.  bound = 0 here: the test must
            // hold for every lane, and a lane that has found nothing yet has no better bound.
            unsigned alive;
            double hx, hy, hd;   // lane l holds row l & 31: x, y, length of its tour edge
            {
                const double4 cb = s_cbox[pe];
                const NodeRec &rr = rows[lane & (kSweepRows - 1)];
                hx = rr.x; hy = rr.y; hd = rr.ds;
                const double gx = fmax(0.0, fmax(cb.x - rr.x, rr.x - cb.y)), gy = fmax(0.0, fmax(cb.z - rr.y, rr.y - cb.w));
                const double T = rr.ds + s_cgmax[pe] + prune2;
                const bool reach = lane < kSweepRows && gx * gx + gy * gy < (ATT10 ? 10.0 * T * T : T * T);
                alive = __builtin_amdgcn_readfirstlane((unsigned)__ballot(reach));
            }
            // tiers 1 and 2 for four rows (need[u]: tier 0 could not exclude row idx[u] for this lane)
            auto rare4 = [&](const int (&idx)[4], const bool (&need)[4]) {
                // tier 1, both new edges, still without a root: |ab| + |a1 b1| < T2 = bound + d(a,a1) + d(b,b1) +
                // margin  <=>  w = T2^2 - s1 - s2 > 0 and 4 s1 s2 < w^2.  All four rows in straight-line code.
                bool ok[4];
                bool any2 = false;
                NodeRec ri[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) ri[u] = rows[idx[u]];   // all LDS reads in flight before the first use
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double dx1 = ri[u].x - rj.x, dy1 = ri[u].y - rj.y;
                    const double dx = ri[u].xs - rj.xs, dy = ri[u].ys - rj.ys, T2 = ri[u].ds + bound + cds2;
                    const double sc = ATT10 ? 0.1 : 1.0;
                    const double p1 = sc * fma(dx1, dx1, dy1 * dy1), p2 = sc * fma(dx, dx, dy * dy);
                    const double w = T2 * T2 - p1 - p2;
                    // one slot pair once (inside a group: row slot below column slot), never adjacent nodes;
                    // '&' on purpose: straight-line code, no branch per condition
                    ok[u] = need[u] & (T2 > 0.0) & (w > 0.0) & (4.0 * p1 * p2 < w * w) &
                            ((cgp > r) | (row0 + idx[u] < lane)) & (ri[u].id >= 0) & (rj.id >= 0) &
                            (rj.id != ri[u].succ) & (rj.succ != ri[u].id);
                    any2 = any2 | ok[u];
                }
                if (any2) {   // tier 2: the exact delta
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (ok[u]) {
                            // the reference evaluates with the lower node id as `a` (tabusearch.c:128-150, i < j); with
                            // non-integer costs the other operand order can differ by an ulp and flip a tie
                            const double delta = (INT || ri[u].id < rj.id) ? pair_delta<WT, INT>(ri[u], rj)
                                                                            : pair_delta<WT, INT>(rj, ri[u]);
                            const u64 kk = make_key(min(ri[u].id, rj.id), max(ri[u].id, rj.id));
                            if (delta < bd || (delta == bd && delta < 0.0 && kk < key)) {
                                bool tabu = false;
                                if constexpr (TABU) {   // tabusearch.c:137-149, a = the lower node id, lazy clears included
                                    const bool row_lo = ri[u].id < rj.id;
                                    const int i = row_lo ? ri[u].id : rj.id, jn = row_lo ? rj.id : ri[u].id;
                                    const int a1 = row_lo ? ri[u].succ : rj.succ, b1 = row_lo ? rj.succ : ri[u].succ;
                                    tabu = stamp_is_tabu(a.tabu + udir_pos(i, jn, a.n), a.iter, a.tenure) ||
                                           stamp_is_tabu(a.tabu + udir_pos(i, a1, a.n), a.iter, a.tenure) ||
                                           stamp_is_tabu(a.tabu + udir_pos(jn, b1, a.n), a.iter, a.tenure) ||
                                           stamp_is_tabu(a.tabu + udir_pos(i, b1, a.n), a.iter, a.tenure);
                                }
                                if (!tabu) { bd = delta; key = kk; bound = bd; }
                            }
                        }
                    }
                }
            };
            // four live rows at a time (a short last group repeats its last row: the same pair twice changes nothing)
            while (alive) {
                int idx[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (alive) { idx[u] = __builtin_ctz(alive); alive &= alive - 1; }
                    else idx[u] = idx[u > 0 ? u - 1 : 0];
                }
                bool need[4];
                bool any = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    // tier 0, the new edge (a, b) alone: |ab| < bound + d(a,a1) + d(b,b1) + margin.  The row's x, y and
                    // edge length come out of lane idx[u]'s registers (v_readlane): no LDS round trip in the hot path.
                    const double rx = lane_bcast(hx, idx[u]), ry = lane_bcast(hy, idx[u]), rd = lane_bcast(hd, idx[u]);
                    const double dx = rx - rj.x, dy = ry - rj.y, T = rd + bound + cds;
                    need[u] = fma(dx, dx, dy * dy) < (ATT10 ? 10.0 : 1.0) * T * fabs(T);   // T <= 0: never
                    any = any || need[u];
                }
                if (any) rare4(idx, need);
            }
        }
            __syncthreads();   // the stage (and, after the last chunk, s_list) is rewritten next
        }
    }

    __shared__ double s_d[NW];
    __shared__ u64 s_k[NW];
    __shared__ int s_last;
    TSP_STAMP(2);
    block_argmin<true>(bd, key, s_d, s_k);
    TSP_STAMP(3);
    if (tid == 0) {
        const size_t slot_idx = (size_t)tour * a.partial_per_tour + blockIdx.x;
        publish_partial(a.partials + slot_idx, bd, key_i(key), key_j(key));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left this CU before the ticket
        TSP_STAMP(4);
        // arrivals on one word are served one after the other (~12 ns each): count per cluster first (one word per
        // cluster, 256 B apart), then the clusters on the tour's word
        gi32 *ct = (gi32 *)(a.cl_tickets + ((size_t)tour * Q + q) * 64);
        s_last = 0;
        if (__hip_atomic_fetch_add(ct, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == kSweepCluster) {
            __hip_atomic_store(ct, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int old = __hip_atomic_fetch_add((gi32 *)(a.tickets + tour), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old + 1 == Q);
            if (s_last) __hip_atomic_store((gi32 *)(a.tickets + tour), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!s_last) return;
    TSP_STAMP(5);
#ifdef TSP_STAMPS
    apply_step<WT, INT, TSP_2OPT_BEST, 2, false, true>(a, tour, 0, a.n - 1, stamps, reinterpret_cast<double *>(&s_stage[0][0]));
#else
    apply_step<WT, INT, TSP_2OPT_BEST, 2, false, true>(a, tour, 0, a.n - 1, reinterpret_cast<double *>(&s_stage[0][0]));
#endif
    static_assert(sizeof(NodeRec) * kSweepStage * 128 >= 4096 * sizeof(double), "the staging area doubles as the cost chunk");
}

}  // namespace tsp
