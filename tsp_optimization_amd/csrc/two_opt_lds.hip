// two_opt_lds.hip -- LDS engine: one 512-thread workgroup per tour, the whole 2-opt descent inside
// one launch.  Same semantics, same control block and same counters as the GRID engine
// (two_opt_grid.hip); what changes is where the state lives and who synchronises:
//
//   LDS (160 KiB per CU on MI355X):  coord[n] double2 | order[n], pos[n] uint16 | 32 row records
//   => 20 n bytes + 2.5 KiB: tours of up to ~8000 nodes stay on chip for the entire descent; a
//   step costs two block barriers instead of a kernel boundary, so the many-small-tours cases
//   (GRASP multi-start, population refinement) are no longer launch-latency bound.
//
// Per step the block scans rows [ci, ci+R) (FIRST, R adaptive 1..32) or every row in blocks of 32
// (BEST): lanes own columns j = tid + 512 m, derive the column's NodeRec from LDS once per row
// block and evaluate it against the row records (LDS broadcast).  The winner is a block arg-min;
// the move is a parallel swap loop on the LDS arrays.
#include "two_opt_common.hpp"

#include <algorithm>
#include <time.h>

#pragma clang fp contract(off)

namespace tsp {

#ifdef TSP_STAMPS
__device__ unsigned long long g_lds_prof[8];
__device__ unsigned long long g_lds_scan[8];   // tour 0, thread 0: cycles deriving column records / in the row loop / in votes, batches, row iterations   // tour 0, thread 0: cycles in scan / arg-min / counters / move / control, steps
#define LDS_T(k) do { const unsigned long long t_ = clock64(); prof[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define LDS_T(k) do { } while (0)
#endif

constexpr int kLdsThreads = 512;
constexpr int kLdsRows = 32;
using idx_t = unsigned short;

__host__ __device__ inline size_t lds_bytes_needed(int n, bool edge_cache = false) {
    // rows | coord | order | pos | (edge lengths by position) | reduction scratch
    return sizeof(NodeRec) * kLdsRows + sizeof(double2) * (size_t)n + 2 * sizeof(idx_t) * (size_t)n +
           (edge_cache ? sizeof(float) * (size_t)n + 16 : 0) + 1024 + 512;   // + the rows' float records (fp32 first tier)
}

// Node record from the LDS arrays.  CACHE: d(v, succ v) is dsp[pos v] -- the tour's edge lengths by position, kept
// as floats (integer costs < 2^24 on the integer-coordinate variants, exact), so that a step derives thousands of
// column records without a single root; a move reverses the same sub-array of dsp and recomputes its two new edges.
template <int WT, bool INT, bool CACHE>
__device__ __forceinline__ NodeRec lds_node(const double2 *coord, const idx_t *order, const idx_t *pos, const float *dsp,
                                            int n, int v) {
    const int p = (int)pos[v];
    int q = p + 1;
    if (q == n) q = 0;
    const int s = (int)order[q];
    const double2 c = coord[v], cs = coord[s];
    NodeRec r;
    r.x = c.x; r.y = c.y; r.xs = cs.x; r.ys = cs.y;
    if constexpr (CACHE) r.ds = (double)dsp[p];
    else r.ds = dist_xy<WT, INT>(c.x, c.y, cs.x, cs.y);
    r.succ = s; r.id = v;
    return r;
}

template <int WT, bool INT, int MODE, bool CACHE, bool F32>
__global__ __launch_bounds__(kLdsThreads) void k_lds_two_opt(const double2 *__restrict__ coord_g,
                                                             int *__restrict__ orders_g,
                                                             TourState *__restrict__ states, int n, int rmin,
                                                             int rmax, int count_evals, int max_iters, double margin, double prune,
                                                             int probe, int probe2, double org_x, double org_y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NodeRec *s_rows = reinterpret_cast<NodeRec *>(smem);
    double2 *coord = reinterpret_cast<double2 *>(smem + sizeof(NodeRec) * kLdsRows);
    idx_t *order = reinterpret_cast<idx_t *>(coord + n);
    idx_t *pos = order + n;
    char *scratch = reinterpret_cast<char *>(pos + n);
    scratch += (16 - (reinterpret_cast<size_t>(scratch) & 15)) & 15;
    float *dsp = reinterpret_cast<float *>(scratch);   // CACHE: edge length leaving tour position p
    if constexpr (CACHE) {
        scratch += sizeof(float) * (size_t)n;
        scratch += (16 - (reinterpret_cast<size_t>(scratch) & 15)) & 15;
    }
    double *s_d = reinterpret_cast<double *>(scratch);            // 16
    u64 *s_k = reinterpret_cast<u64 *>(scratch + 128);            // 16
    long long *s_ll = reinterpret_cast<long long *>(scratch + 256);  // 16
    double *s_chunk = reinterpret_cast<double *>(scratch + 384);  // 64 doubles (fcost cost recompute)
    int4 *s_win = reinterpret_cast<int4 *>(scratch + 384);        // 8 (FIRST, the probe's vote): shares the chunk, which only BEST uses
    int *s_flag = reinterpret_cast<int *>(scratch + 384 + 384);   // FIRST: second probe round
    int *s_vote = reinterpret_cast<int *>(scratch + 384 + 384 + 8);   // FIRST, scan: two words, the stamp of the last vote (by parity) that saw a hit in the block's first row
    float4 *s_rowsf = reinterpret_cast<float4 *>(scratch + 1024);  // x, y (relative to the instance corner), edge length of the block's rows as floats
    // Integer coordinates of bounded span are exact as floats relative to the instance corner: the new-edge bound runs in fp32
    // first (its rounding paid for in slack: s may come out 2^-22 low, T -- below 2^23 -- is taken 2 units high), four rows per
    // trip, before the row's full record and the fp64 tiers are looked at -- as in the CLUSTER engine's tiles scan
    // (a variant of its own, chosen by the host from 2 500 nodes on (128 random tours, on / off: n = 2 000 48.9 / 46.6 ms, 3 000 99.5 / 101.5, 5 000 200 / 210): rand5000 x 128 random individuals 216 -> 205 ms, but att532 x 256
    //  GRASP starts 2.95 -> 3.7 ms -- short rows, dense hits: the trip of four rows costs more than it prunes; and with both forms
    //  in one kernel behind a run-time switch both were 5-8 % slower)
    constexpr bool F32T0 = F32 && (WT == WT_EUC_2D_ICOORD || WT == WT_CEIL_2D_ICOORD || WT == WT_ATT_ICOORD);
    constexpr bool ATT10 = WT == WT_ATT || WT == WT_ATT_ICOORD;

    const int tour = blockIdx.x;
    const int tid = threadIdx.x;
    TourState *st = states + tour;
    if (st->done) return;
    int *order_g = orders_g + (size_t)tour * n;

    for (int v = tid; v < n; v += kLdsThreads) {
        coord[v] = coord_g[v];
        const int w = order_g[v];
        order[v] = (idx_t)w;
        pos[w] = (idx_t)v;
    }
    // control block in registers (every thread keeps an identical copy)
    int ci = st->ci, cj = st->cj, chunk = min(max(st->chunk_rows, 1), kLdsRows), done = 0;
    double obj = st->obj, seen = st->seen_cost;
    long long sweeps = st->sweeps, evals = st->evals, moves = st->moves, reversed = st->reversed,
              scanned = st->pairs_scanned, steps = st->steps;
    bool probe_on = true;                       // FIRST: the last hit lay within `probe` pairs of the cursor
    bool after_hit = true;                      // FIRST: the step before found a move (TSP_LDS_PROBE2=1: the second round only then)
    // Evaluation count: pairs between the old and the new cursor minus the adjacent ones among them (heuristics.c:471).  The
    // adjacent pairs telescope over the steps of a sweep (two_opt_cluster.hip has the derivation): with A(K) = tour edges whose
    // pair key is <= K, a sweep's total is A_now(cursor) - A(cursor at the start) - sum over its moves of
    // d = 1 + [(a1,b1) <= hi] - [(i,a1) <= hi] - [(j,b1) <= hi], hi = (i,j) the move's pair: O(1) per move, one pass over the
    // tour per launch, instead of a pass over the step's rows (and ballots in the probes) in every step.
    long long adj_seen = 0, adjD = 0, adjA0 = 0;
    bool sweep_open = false;
    auto edges_upto = [&](u64 K) -> long long {
        long long cnt = 0;
        for (int p = tid; p < n; p += kLdsThreads) {
            const int u = (int)order[p], v = (int)order[p + 1 == n ? 0 : p + 1];
            cnt += make_key(min(u, v), max(u, v)) <= K ? 1 : 0;
        }
        return block_sum<long long>(cnt, s_ll);
    };
    long long r_cur = pair_rank(ci, cj, n);     // rank of the cursor in scan order
    int vote_seq = 0;
    if (tid < 2) s_vote[tid] = 0;
    __syncthreads();
    if constexpr (MODE == TSP_2OPT_FIRST) {
        if (count_evals && (ci != 0 || cj != 0)) { adjA0 = edges_upto(make_key(ci, cj)); sweep_open = true; }
    }
    if constexpr (CACHE) {
        for (int p = tid; p < n; p += kLdsThreads) {
            const double2 c = coord[order[p]], cs = coord[order[p + 1 == n ? 0 : p + 1]];
            dsp[p] = (float)dist_xy<WT, INT>(c.x, c.y, cs.x, cs.y);
        }
        __syncthreads();
    }

#ifdef TSP_STAMPS
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
    unsigned long long scn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int iter = 0; iter < max_iters && !done; ++iter) {
        LDS_T(6);   // control block of the step before
        int row_lo = 0, row_hi = n - 1;
        double bd = 0.0;
        u64 key = kNoKey;
        // scan cursor: where the rows x columns scan below starts (the probe moves it on when it finds nothing)
        int si = ci, sj = cj;
        bool probe_hit = false;
        int4 win = make_int4(0, 0, 0, 0);   // probe hit: positions of the pair's nodes and their successors
        if constexpr (MODE == TSP_2OPT_FIRST) {
            // ---- probe: the next 512 pairs in scan order, one per thread --------------------------------------
            // The distance from the cursor to the next improving pair is very skewed: on a random individual of rand5000
            // (40 519 moves) half of the hits lie within 64 pairs of the cursor and 77 % within 512, while a rows x columns
            // batch evaluates thousands of pairs before its first vote.  So every step first looks at the 512 pairs
            // that follow the cursor (heuristics.c:452-454 order; they span at most a few rows), thread t at the t-th:
            // the hit of the lowest thread is the reference's next move, and the adjacent pairs the reference skips on
            // its way there (:471) are counted from the same ballots -- no arg-min, no counting pass, no second barrier.
            // (rows of fewer than 128 columns: 512 pairs could span more than the four row changes below -- no probe there)
            if (probe > 0 && probe_on && ci <= n - 134) {
#ifdef TSP_STAMPS
                const unsigned long long pq0 = clock64();
                unsigned long long pq1 = pq0;
#endif
                int i = ci, j = cj + 1 + tid;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (j >= n) { j = j - n + i + 2; i += 1; }
                const bool act = j < n && i < n - 1;
                bool hit = false, adjp = false;
                double delta = 0.0;
                int win_a1 = 0, win_b1 = 0;
                if (act) {
                    const NodeRec ri = lds_node<WT, INT, CACHE>(coord, order, pos, dsp, n, i);
                    const NodeRec rj = lds_node<WT, INT, CACHE>(coord, order, pos, dsp, n, j);
                    win_a1 = ri.succ; win_b1 = rj.succ;
                    adjp = j == ri.succ || rj.succ == i;   // heuristics.c:471
#ifdef TSP_STAMPS
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    pq1 = clock64();
#endif
                    // one pair per lane: a wave takes as long as its slowest lane, so the bound tiers of the rows x columns
                    // scan would only add their cost to the exact evaluation of the lanes that pass them
                    if (!adjp) {
                        delta = pair_delta<WT, INT>(ri, rj);
                        hit = delta < 0;
                    }
                }
                const unsigned long long hb = __ballot(hit);
#ifdef TSP_STAMPS
                const unsigned long long pq2 = clock64();
#endif
                const int wv = tid >> 6, ln = tid & 63;
                if (ln == 0) s_k[wv] = hb;
                if (tid == 0) *s_flag = 1 << 20;   // second round: smallest m with a hit so far
                if (hb && ln == __builtin_ctzll(hb)) {
                    s_d[wv] = delta; s_k[8 + wv] = make_key(i, j);
                    s_win[wv] = make_int4((int)pos[i], (int)pos[j], win_a1, win_b1);   // the move needs no further look at the tour
                }
                // the pair of the last thread: where the scan goes on after a probe without a hit
                int ei = ci, ej = cj + kLdsThreads;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (ej >= n) { ej = ej - n + ei + 2; ei += 1; }
                __syncthreads();
#ifdef TSP_STAMPS
                const unsigned long long pq3 = clock64();
                if (tour == 0 && tid == 0) { scn[5] += pq1 - pq0; scn[6] += pq2 - pq1; scn[7] += pq3 - pq2; }
#endif
                {   // every wave reads the eight wave results once (lane w: wave w) and reduces them in registers
                    constexpr int NWV = kLdsThreads / 64;
                    const int lw = ln & (NWV - 1);
                    const unsigned long long h = s_k[lw];
                    const double dw = s_d[lw];
                    const u64 kw = s_k[8 + lw];
                    const int4 ww = s_win[lw];
                    const unsigned long long hm = __ballot(ln < NWV && h != 0ull);
                    probe_hit = hm != 0ull;
                    const int fw = probe_hit ? __builtin_ctzll(hm) : NWV;   // first wave with a hit
                    if (probe_hit) {
                        bd = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dw), fw), __builtin_amdgcn_readlane(__double2loint(dw), fw));
                        const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)kw, fw);
                        const unsigned khi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(kw >> 32), fw);
                        key = ((u64)khi << 32) | klo;
                        win.x = __builtin_amdgcn_readlane(ww.x, fw); win.y = __builtin_amdgcn_readlane(ww.y, fw);
                        win.z = __builtin_amdgcn_readlane(ww.z, fw); win.w = __builtin_amdgcn_readlane(ww.w, fw);
                    }
                }
                if (!probe_hit) { si = ei; sj = ej; }
                // ---- second round: the 4 096 pairs after those, up to eight per thread (thread t at 512 (m + 1) + t, m = 0 .. 7).
                // 57 % of the hits the first 512 pairs miss lie here (oracle trace, see above), and a rows x columns scan
                // with its votes, arg-min and counting pass costs three times this.  A wave stops after the first m in which
                // one of its lanes has a hit (every pair of a later m comes later in scan order), and after the first m any wave
                // has reported one (s_flag, an LDS minimum); the adjacent pairs up to the winner are counted per thread into
                // adj_acc (summed once per launch).  Rows of >= 591 columns only: the 4 608 pairs then span at most nine rows
                // and never reach the end of the sweep.
                if (!probe_hit && probe2 && (probe2 > 1 || after_hit) && ci <= n - 600) {   // (4 and 8 rounds measure alike, 12 and more lose: configs[4] 227 / 227 / 229 / 233 / 242 ms at 4 / 8 / 12 / 16 / 24)
                    constexpr int NWV = kLdsThreads / 64, R2 = 8;
                    long long *s_t2 = s_ll + 8;                                  // per wave: smallest pair number with a hit
                    double *s_d2 = s_d + 8;
                    int4 *s_win2 = s_win + 8;
                    u64 *s_key2 = reinterpret_cast<u64 *>(s_win + 16);
                    const int ln = tid & 63, wv = tid >> 6;
                    bool hit = false;
                    int hm = 0, hi_ = 0, hj_ = 0, ha1 = 0, hb1 = 0;
                    double hd = 0.0;
                    for (int m = 0; m < R2; ++m) {
                        if (m > *(volatile int *)s_flag) break;   // an earlier m has a hit somewhere in the workgroup
                        int i = ci, j = cj + 1 + kLdsThreads * (m + 1) + tid;
                        while (j >= n) { j = j - n + i + 2; i += 1; }
                        const NodeRec ri = lds_node<WT, INT, CACHE>(coord, order, pos, dsp, n, i);
                        const NodeRec rj = lds_node<WT, INT, CACHE>(coord, order, pos, dsp, n, j);
                        const bool adjp = j == ri.succ || rj.succ == i;   // heuristics.c:471
                        if (!adjp) {
                            const double delta = pair_delta<WT, INT>(ri, rj);
                            if (delta < 0) { hit = true; hm = m; hi_ = i; hj_ = j; ha1 = ri.succ; hb1 = rj.succ; hd = delta; }
                        }
                        if (__any(hit)) { if (ln == 0) atomicMin(s_flag, m); break; }
                    }
                    const unsigned long long hb2 = __ballot(hit);
                    if (ln == 0) s_t2[wv] = hb2 ? (long long)kLdsThreads * (0 + 1) : (long long)1 << 40;   // patched below by the hit lane
                    if (hb2 && ln == __builtin_ctzll(hb2)) {
                        s_t2[wv] = (long long)kLdsThreads * (hm + 1) + tid;
                        s_d2[wv] = hd; s_key2[wv] = make_key(hi_, hj_);
                        s_win2[wv] = make_int4((int)pos[hi_], (int)pos[hj_], ha1, hb1);
                    }
                    __syncthreads();
                    {
                        const int lw = ln & (NWV - 1);
                        const long long tw = s_t2[lw];
                        const double dw = s_d2[lw];
                        const u64 kw = s_key2[lw];
                        const int4 ww = s_win2[lw];
                        const u64 tk = wave_min_u64(((u64)tw << 3) | (u64)lw);   // lanes 8 .. 63 repeat lanes 0 .. 7
                        const long long twin = (long long)(tk >> 3);
                        const int fw = (int)(tk & 7);
                        if (twin < ((long long)1 << 40)) {
                            probe_hit = true;
                            bd = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dw), fw), __builtin_amdgcn_readlane(__double2loint(dw), fw));
                            const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)kw, fw);
                            const unsigned khi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(kw >> 32), fw);
                            key = ((u64)khi << 32) | klo;
                            win.x = __builtin_amdgcn_readlane(ww.x, fw); win.y = __builtin_amdgcn_readlane(ww.y, fw);
                            win.z = __builtin_amdgcn_readlane(ww.z, fw); win.w = __builtin_amdgcn_readlane(ww.w, fw);
                        } else {   // nothing in these 2 560 pairs: the scan goes on behind them
                            int ei2 = ci, ej2 = cj + kLdsThreads * (R2 + 1);
                            while (ej2 >= n) { ej2 = ej2 - n + ei2 + 2; ei2 += 1; }
                            si = ei2; sj = ej2;
                        }
                    }
                }
            }
            row_lo = si; row_hi = min(si + chunk, n - 1);
#ifdef TSP_STAMPS
            LDS_T(4);
            if (probe_hit) prof[5] += 1;
#endif
        }

        // ---- scan ------------------------------------------------------------------------------
        for (int rb = row_lo; rb < row_hi && !probe_hit; rb += kLdsRows) {
            const int nr = min(kLdsRows, row_hi - rb);
            __syncthreads();
            if (tid < nr) {
                const NodeRec rr = lds_node<WT, INT, CACHE>(coord, order, pos, dsp, n, rb + tid);
                s_rows[tid] = rr;
                if constexpr (F32T0) s_rowsf[tid] = make_float4((float)(rr.x - org_x), (float)(rr.y - org_y), (float)rr.ds, 0.f);
            }
            __syncthreads();
            // Columns four at a time: a column record is a chain of dependent LDS reads (pos -> order / edge length ->
            // successor's coordinates); the four chains advance level by level, so a thread waits for LDS three
            // times per four columns instead of three times per column (two waves per SIMD hide little).
            // FIRST: the winner is the smallest key, and a hit in the block's first row ends the search: no later
            // batch of columns can hold a smaller one (dense-improvement phases -- random individuals -- find it
            // in the first batch after the cursor), so batches are 512 columns there and the block votes after each
            // (512 then 2048 was measured: no better; two batches per vote: 3 % better on configs[3] and [4]).
            constexpr int U = MODE == TSP_2OPT_FIRST ? 2 : 4;   // FIRST: 1024 columns per vote (measured: 1 -> 2: -3 %, 4: +2 %)
            // batches that hold no column above the rows (or, for the cursor's row alone, above the cursor) are skipped
            const int jbase = MODE == TSP_2OPT_FIRST ? ((max(rb, (nr == 1 && rb == si) ? sj : 0) + 1) / kLdsThreads) * kLdsThreads : 0;
            const bool vote = n - jbase > 2 * kLdsThreads;   // a vote is a barrier: not for two batches
            for (int j0 = jbase + tid; j0 - tid < n; j0 += U * kLdsThreads) {
#ifdef TSP_STAMPS
                const unsigned long long q0 = clock64();
#endif
                int jj[U], pp[U], sc[U];
                bool act[U];
                double2 cxy[U], cs[U];
                double dsv[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    jj[k] = j0 + k * kLdsThreads;
                    act[k] = jj[k] < n && jj[k] > rb;   // no row of this block is below a column <= rb
                    const int jc = act[k] ? jj[k] : 0;
                    pp[k] = (int)pos[jc];
                    cxy[k] = coord[jc];
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int q = pp[k] + 1 == n ? 0 : pp[k] + 1;
                    sc[k] = (int)order[q];
                    if constexpr (CACHE) dsv[k] = (double)dsp[pp[k]];
                }
#pragma unroll
                for (int k = 0; k < U; ++k) cs[k] = coord[sc[k]];
#ifdef TSP_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long q1 = clock64();
#endif
                if constexpr (F32T0) {
                    NodeRec rjv[U];
                    float cxf[U], cyf[U], cdf[U];
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        rjv[k].x = cxy[k].x; rjv[k].y = cxy[k].y; rjv[k].xs = cs[k].x; rjv[k].ys = cs[k].y;
                        if constexpr (CACHE) rjv[k].ds = dsv[k];
                        else rjv[k].ds = dist_xy<WT, INT>(cxy[k].x, cxy[k].y, cs[k].x, cs[k].y);
                        rjv[k].succ = sc[k]; rjv[k].id = jj[k];
                        cxf[k] = (float)(cxy[k].x - org_x); cyf[k] = (float)(cxy[k].y - org_y); cdf[k] = (float)rjv[k].ds;
                    }
                    constexpr int RQ = 4;   // rows per trip: four LDS reads in flight, one vote
                    for (int r0 = 0; r0 < nr; r0 += RQ) {
                        float4 rf[RQ];
#pragma unroll
                        for (int u = 0; u < RQ; ++u) rf[u] = s_rowsf[min(r0 + u, nr - 1)];
                        const float bf = (float)((MODE == TSP_2OPT_FIRST ? 0.0 : bd) + 2.0 * prune + 2.0);
                        bool need[RQ][U];
                        bool any = false;
#pragma unroll
                        for (int u = 0; u < RQ; ++u) {
#pragma unroll
                            for (int k = 0; k < U; ++k) {
                                const float dx = rf[u].x - cxf[k], dy = rf[u].y - cyf[k], T = rf[u].z + cdf[k] + bf;
                                need[u][k] = act[k] && r0 + u < nr && fmaf(dx, dx, dy * dy) < (ATT10 ? 10.0f : 1.0f) * 1.000002f * T * fabsf(T);   // T <= 0: never
                                any = any || need[u][k];
                            }
                        }
                        if (!__any(any)) continue;
#pragma unroll
                        for (int u = 0; u < RQ; ++u) {
#pragma unroll
                            for (int k = 0; k < U; ++k) {
                                if (!__any(need[u][k])) continue;
                                const int i = rb + r0 + u, j = jj[k];
                                const NodeRec ri = s_rows[r0 + u];
                                const NodeRec &rj = rjv[k];
                                const u64 kq = make_key(i, j);
                                bool ok = need[u][k] && j > i && j != ri.succ && rj.succ != i;   // heuristics.c:471 / tabusearch.c:134
                                if constexpr (MODE == TSP_2OPT_FIRST) ok = ok && (i > si || j > sj) && kq < key;
                                const double bound = (MODE == TSP_2OPT_FIRST) ? 0.0 : bd;
                                ok = ok && new_edge_can_improve<WT>(ri.x, ri.y, rj.x, rj.y, bound + ri.ds + rj.ds + 2.0 * prune);
                                if (ok) {
                                    const double lower = pair_delta_approx<WT>(ri, rj) - margin;
                                    ok = MODE == TSP_2OPT_FIRST ? lower < bound : lower <= bound;
                                }
                                if (ok) {
                                    const double delta = pair_delta<WT, INT>(ri, rj);
                                    if constexpr (MODE == TSP_2OPT_FIRST) {
                                        if (delta < 0) { bd = delta; key = kq; }
                                    } else {
                                        if (better(delta, kq, bd, key)) { bd = delta; key = kq; }
                                    }
                                }
                            }
                        }
                    }
                } else {
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (!act[k]) continue;
                    const int j = jj[k];
                    NodeRec rj;
                    rj.x = cxy[k].x; rj.y = cxy[k].y; rj.xs = cs[k].x; rj.ys = cs[k].y;
                    if constexpr (CACHE) rj.ds = dsv[k];
                    else rj.ds = dist_xy<WT, INT>(cxy[k].x, cxy[k].y, cs[k].x, cs[k].y);
                    rj.succ = sc[k]; rj.id = j;
                    for (int r = 0; r < nr; ++r) {
                        const int i = rb + r;
                        const NodeRec ri = s_rows[r];
                        bool ok = j > i && j != ri.succ && rj.succ != i;  // heuristics.c:471 / tabusearch.c:134
                        if constexpr (MODE == TSP_2OPT_FIRST) ok = ok && (i > si || j > sj);
                        const u64 kq = make_key(i, j);
                        if constexpr (MODE == TSP_2OPT_FIRST) ok = ok && kq < key;
                        if constexpr (has_root_filter<WT>()) {
                            const double bound = (MODE == TSP_2OPT_FIRST) ? 0.0 : bd;
                            // a lane's keys are not visited in increasing order here, so BEST must keep ties
                            // (an equal delta with a smaller key wins): skip only when provably greater
                            // (the new-edge bound is written with its margin doubled so that "<" also keeps ties)
                            ok = ok && new_edge_can_improve<WT>(ri.x, ri.y, rj.x, rj.y, bound + ri.ds + rj.ds + 2.0 * prune);
                            if (ok) {
                                const double lower = pair_delta_approx<WT>(ri, rj) - margin;
                                ok = MODE == TSP_2OPT_FIRST ? lower < bound : lower <= bound;
                            }
                        }
                        if (ok) {
                            const double delta = pair_delta<WT, INT>(ri, rj);
                            if constexpr (MODE == TSP_2OPT_FIRST) {
                                if (delta < 0) { bd = delta; key = kq; }
                            } else {
                                if (better(delta, kq, bd, key)) { bd = delta; key = kq; }
                            }
                        }
                    }
                }
                }
#ifdef TSP_STAMPS
                const unsigned long long q2 = clock64();
                bool stop = false;
                if constexpr (MODE == TSP_2OPT_FIRST) {
                    if (vote) {
                        const int stamp = ++vote_seq;
                        const bool anyh = __any(key != kNoKey && key_i(key) == rb);
                        if ((tid & 63) == 0 && anyh) s_vote[stamp & 1] = stamp;
                        __syncthreads();
                        stop = s_vote[stamp & 1] == stamp;
                    }
                }
                if (tour == 0 && tid == 0) { const unsigned long long q3 = clock64(); scn[0] += q1 - q0; scn[1] += q2 - q1; scn[2] += q3 - q2; scn[3] += 1; scn[4] += nr; }
                if (stop) break;
#else
                if constexpr (MODE == TSP_2OPT_FIRST) {
                    // The vote: a hit in the block's first row ends the search.  One flag word and one barrier (__syncthreads_or is a
                    // workgroup reduction: several); the word holds the stamp of the last vote that saw a hit -- no zeroing -- and
                    // votes alternate between two words: nobody writes vote k + 2 before everybody has read vote k.
                    if (vote) {
                        const int stamp = ++vote_seq;
                        const bool anyh = __any(key != kNoKey && key_i(key) == rb);
                        if ((tid & 63) == 0 && anyh) s_vote[stamp & 1] = stamp;
                        __syncthreads();
                        if (s_vote[stamp & 1] == stamp) break;
                    }
                }
#endif
            }
        }
        LDS_T(0);
        if (!probe_hit) block_argmin<MODE == TSP_2OPT_BEST>(bd, key, s_d, s_k);
        LDS_T(1);
        const bool found = key != kNoKey && (MODE == TSP_2OPT_FIRST || bd < 0);
        const int wi = found ? key_i(key) : -1, wj = found ? key_j(key) : -1;

        // ---- reference-equivalent evaluation count (FIRST): see adj_seen above ------------------------------
        int ni = wi, nj = wj;
        if constexpr (MODE == TSP_2OPT_FIRST) {
            if (!found) { ni = row_hi - 1; nj = n - 1; }
        }

        LDS_T(2);
        // ---- move: reverse positions pa+1 .. pb (cyclic) --------------------------------------------
        int L = 0;
        if (found) {
            // a probe hit carries the positions and successors along: nobody reads the tour between its vote and the swaps,
            // so the barrier that otherwise separates those reads from the swaps is not needed
            const int pa = probe_hit ? win.x : (int)pos[wi], pb = probe_hit ? win.y : (int)pos[wj];
            if constexpr (MODE == TSP_2OPT_FIRST) {
                if (count_evals) {   // d of this move, on the tour as it is before the move
                    const int a1 = probe_hit ? win.z : (int)order[pa + 1 == n ? 0 : pa + 1], b1 = probe_hit ? win.w : (int)order[pb + 1 == n ? 0 : pb + 1];
                    const u64 hi = make_key(wi, wj);
                    auto le = [&](int u, int v) { return make_key(min(u, v), max(u, v)) <= hi ? 1 : 0; };
                    adjD += 1 + le(a1, b1) - le(wi, a1) - le(wj, b1);
                }
            }
            float new_edge = 0.f;
            if constexpr (CACHE) {
                // the two new edges (a, b) and (succ a, succ b) will leave positions pa and pb, which the reversal of the
                // inner edge lengths does not touch: two threads price them on the old tour, beside the swaps
                if (tid < 2) {
                    const int u = tid == 0 ? wi : (probe_hit ? win.z : (int)order[pa + 1 == n ? 0 : pa + 1]);
                    const int v = tid == 0 ? wj : (probe_hit ? win.w : (int)order[pb + 1 == n ? 0 : pb + 1]);
                    const double2 c = coord[u], cs = coord[v];
                    new_edge = (float)dist_xy<WT, INT>(c.x, c.y, cs.x, cs.y);
                }
            }
            if (!probe_hit) __syncthreads();  // everyone has read pa/pb (and finished the adjacency reads)
            L = pb - pa; if (L < 0) L += n;
            const int half = L >> 1;
            if constexpr (CACHE) {
                // one loop for both reversals (tour positions pa+1 .. pb; the edge lengths between them, positions
                // pa+1 .. pb-1, which keep their values and change places): all four reads of a trip are in flight together
                const int inner = (L - 1) >> 1;
                // two trips of a thread at a time: the eight reads of both are in flight together (a random tour's reversal is
                // ~800 swaps, i.e. two trips per thread, and a trip is a chain of LDS latencies)
                for (int t0 = tid; t0 < half; t0 += 2 * kLdsThreads) {
                    const int t1 = t0 + kLdsThreads;
                    const bool two = t1 < half;
                    int p0 = pa + 1 + t0; if (p0 >= n) p0 -= n;
                    int q0 = pb - t0; if (q0 < 0) q0 += n;
                    int r0 = q0 - 1; if (r0 < 0) r0 += n;
                    int p1 = pa + 1 + t1; if (p1 >= n) p1 -= n;
                    int q1 = pb - t1; if (q1 < 0) q1 += n;
                    int r1 = q1 - 1; if (r1 < 0) r1 += n;
                    if (!two) { p1 = p0; q1 = q0; r1 = r0; }
                    const idx_t u0 = order[p0], w0 = order[q0], u1 = order[p1], w1 = order[q1];
                    const float du0 = dsp[p0], dw0 = dsp[r0], du1 = dsp[p1], dw1 = dsp[r1];
                    order[p0] = w0; order[q0] = u0;
                    pos[w0] = (idx_t)p0; pos[u0] = (idx_t)q0;
                    if (t0 < inner) { dsp[p0] = dw0; dsp[r0] = du0; }
                    if (two) {
                        order[p1] = w1; order[q1] = u1;
                        pos[w1] = (idx_t)p1; pos[u1] = (idx_t)q1;
                        if (t1 < inner) { dsp[p1] = dw1; dsp[r1] = du1; }
                    }
                }
                if (tid < 2) dsp[tid == 0 ? pa : pb] = new_edge;
            } else {
                for (int t = tid; t < half; t += kLdsThreads) {
                    int p = pa + 1 + t; if (p >= n) p -= n;
                    int q = pb - t; if (q < 0) q += n;
                    const idx_t u = order[p], w = order[q];
                    order[p] = w; order[q] = u;
                    pos[w] = (idx_t)p; pos[u] = (idx_t)q;
                }
            }
        }
        __syncthreads();

        LDS_T(3);
        // ---- control block ----------------------------------------------------------------------------
        steps += 1;
        if constexpr (MODE == TSP_2OPT_BEST) {
            sweeps += 1;
            evals += (long long)n * (n - 1) / 2 - n;
            scanned += (long long)n * (n - 1) / 2;
            if (found) { moves += 1; reversed += L - 1; }
            else {
                done = 1;
                // recomputed cost, node order (tabusearch.c:168-172)
                if constexpr (INT || WT == WT_CEIL_2D) {
                    double c = 0.0;
                    for (int v = tid; v < n; v += kLdsThreads) c += lds_node<WT, INT, CACHE>(coord, order, pos, dsp, n, v).ds;
                    obj = block_sum<double>(c, s_d);
                } else {
                    double acc = 0.0;
                    for (int base = 0; base < n; base += 64) {  // sequential order, 64 edges at a time
                        __syncthreads();
                        if (tid < 64 && base + tid < n) s_chunk[tid] = lds_node<WT, INT, CACHE>(coord, order, pos, dsp, n, base + tid).ds;
                        __syncthreads();
                        const int m = min(64, n - base);
                        for (int t = 0; t < m; ++t) acc += s_chunk[t];  // every thread adds the same values in order
                    }
                    obj = acc;
                }
            }
        } else {
            const long long r_new = pair_rank(ni, nj, n);
            scanned += probe_hit ? r_new - r_cur : (found ? pair_rank(row_hi - 1, n - 1, n) : r_new) - r_cur;
            evals += r_new - r_cur;   // the adjacent pairs among them come off per sweep / per launch (adj_seen)
            sweep_open = true;
            after_hit = found;
            if (found) probe_on = r_new - r_cur <= probe;   // (also switching it off after a step without a hit: measured, slower)
            r_cur = r_new;
            if (found) {
                obj += bd;                              // heuristics.c:486
                moves += 1; reversed += L - 1;
                ci = wi; cj = wj; chunk = rmin;
            } else {
                chunk = min(chunk * 2, rmax);
                if (row_hi >= n - 1) {                  // sweep complete
                    adj_seen += (long long)n - adjA0 - adjD;   // every tour edge has been passed
                    adjD = 0; adjA0 = 0; sweep_open = false;
                    sweeps += 1;
                    if (obj >= seen) done = 1;          // heuristics.c:492
                    else { seen = obj; ci = 0; cj = 0; r_cur = 0; }
                } else { ci = row_hi - 1; cj = n - 1; }
            }
        }
    }

#ifdef TSP_STAMPS
    if (tour == 0 && tid == 0) { for (int k = 0; k < 7; ++k) g_lds_prof[k] += prof[k]; g_lds_prof[7] += steps - st->steps; for (int k = 0; k < 8; ++k) g_lds_scan[k] += scn[k]; }
#endif
    // ---- write back ---------------------------------------------------------------------------------------
    __syncthreads();
    if constexpr (MODE == TSP_2OPT_FIRST) {
        if (count_evals) {
            if (sweep_open) adj_seen += edges_upto(make_key(ci, cj)) - adjA0 - adjD;   // the sweep goes on in the next launch
            evals -= adj_seen;
        }
    }
    for (int v = tid; v < n; v += kLdsThreads) order_g[v] = (int)order[v];
    if (tid == 0) {
        st->ci = ci; st->cj = cj; st->chunk_rows = chunk; st->done = done; st->obj = obj; st->seen_cost = seen;
        st->sweeps = sweeps; st->evals = evals; st->moves = moves; st->reversed = reversed;
        st->pairs_scanned = scanned; st->steps = steps;
    }
}

}  // namespace tsp

using namespace tsp;

namespace {
double wall_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

template <int WT, bool INT, int MODE, bool CACHE, bool F32>
hipError_t launch_lds_k(tsp_dev_tours *t, int rmin, int rmax, int max_iters) {
    hipStream_t s = t->inst->ctx->stream;
    const size_t bytes = lds_bytes_needed(t->n, CACHE);
    auto k = k_lds_two_opt<WT, INT, MODE, CACHE, F32>;
    static size_t granted_dev[64] = {0};   // per kernel variant and device
    size_t &granted = granted_dev[t->inst->ctx->device & 63];
    if (bytes > granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        granted = bytes;
    }
    hipLaunchKernelGGL(k, dim3(t->B), dim3(kLdsThreads), bytes, s, t->inst->d_coord, t->d_order, t->d_state, t->n,
                       rmin, rmax, t->count_evals, max_iters, t->inst->filter_margin, t->inst->prune_margin, TSP_SW(t->inst, LDS_PROBE, 65536), TSP_SW(t->inst, LDS_PROBE2, 1),
                       t->inst->org_x, t->inst->org_y);
    return hipGetLastError();
}

template <int WT, bool INT>
hipError_t launch_lds(tsp_dev_tours *t, int mode, int rmin, int rmax, int max_iters) {
    // integer-coordinate variants: integer edge lengths < 2^21, exact as floats
    constexpr bool CAN_CACHE = WT == WT_EUC_2D_ICOORD || WT == WT_CEIL_2D_ICOORD || WT == WT_ATT_ICOORD;
    if constexpr (CAN_CACHE) {
        const bool f32 = t->n >= TSP_SW(t->inst, LDS_F32_MIN_N, 2500);   // the fp32 first tier of the rows x columns scan
        if (lds_bytes_needed(t->n, true) <= (size_t)t->inst->ctx->lds_bytes && TSP_SW(t->inst, LDS_EDGE_CACHE, 1)) {
            if (f32) return mode == TSP_2OPT_FIRST ? launch_lds_k<WT, INT, TSP_2OPT_FIRST, true, true>(t, rmin, rmax, max_iters)
                                                   : launch_lds_k<WT, INT, TSP_2OPT_BEST, true, true>(t, rmin, rmax, max_iters);
            return mode == TSP_2OPT_FIRST ? launch_lds_k<WT, INT, TSP_2OPT_FIRST, true, false>(t, rmin, rmax, max_iters)
                                          : launch_lds_k<WT, INT, TSP_2OPT_BEST, true, false>(t, rmin, rmax, max_iters);
        }
        if (f32) return mode == TSP_2OPT_FIRST ? launch_lds_k<WT, INT, TSP_2OPT_FIRST, false, true>(t, rmin, rmax, max_iters)
                                               : launch_lds_k<WT, INT, TSP_2OPT_BEST, false, true>(t, rmin, rmax, max_iters);
    }
    return mode == TSP_2OPT_FIRST ? launch_lds_k<WT, INT, TSP_2OPT_FIRST, false, false>(t, rmin, rmax, max_iters)
                                  : launch_lds_k<WT, INT, TSP_2OPT_BEST, false, false>(t, rmin, rmax, max_iters);
}
}  // namespace

// implemented in two_opt_grid.hip
int tsp_grid_after_external_run(tsp_dev_tours *t, int mode, int timed_out, bool pos_written = false);

#ifdef TSP_STAMPS
extern "C" int tsp_dev_debug_lds(unsigned long long *out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(tsp::g_lds_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_lds_prof), z, sizeof z);
    return 0;
}
extern "C" int tsp_dev_debug_lds_scan(unsigned long long *out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(tsp::g_lds_scan), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_lds_scan), z, sizeof z);
    return 0;
}
#endif

bool tsp_lds_fits(const tsp_dev_inst *inst) {
    return inst && inst->n <= 65535 && lds_bytes_needed(inst->n) <= (size_t)inst->ctx->lds_bytes;
}

// Runs the tours of `t` to their local optima with the LDS engine (launches of bounded length so
// that a time limit can be honoured between them).
int tsp_lds_run(tsp_dev_tours *t, int mode, double time_limit_s, int *all_done) {
    if (!t || !tsp_lds_fits(t->inst)) return TSP_DEV_E_ARG;
    hipStream_t s = t->inst->ctx->stream;
    const double t0 = wall_s();
    // smallest chunk ~4000 pairs, at least the four rows of one trip of the row loop (measured: att532 best at 8 rows; rand5000
    // x 128: 1 / 2 / 4 / 8 rows = 196.7 / 193.1 / 189.7 / 196.7 ms)
    const int auto_rmin = std::max(4, std::min(16, (4000 + t->n / 2) / t->n));
    const int rmin = std::max(1, std::min(kLdsRows, TSP_SW(t->inst, LDS_MIN_ROWS, auto_rmin)));
    const int rmax = kLdsRows;
    const int max_iters = mode == TSP_2OPT_FIRST ? 8192 : 256;
    if (all_done) *all_done = 0;
    int status = TSP_OK;
    for (;;) {
        hipError_t e = hipSuccess;
        TSP_DISPATCH_METRIC(t->inst->wtype, t->inst->integer_cost, { e = launch_lds<WTC, INTC>(t, mode, rmin, rmax, max_iters); });
        if (e != hipSuccess) { tsp::set_last_error("k_lds_two_opt launch", e, __FILE__, __LINE__); return TSP_DEV_E_HIP; }
        TSP_HIP_TRY(hipMemcpyAsync(t->h_state, t->d_state, sizeof(TourState) * (size_t)t->B, hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
        bool done = true;
        for (int b = 0; b < t->B; ++b) done = done && t->h_state[b].done;
        if (done) { if (all_done) *all_done = 1; break; }
        if (time_limit_s > 0 && wall_s() - t0 > time_limit_s) { status = TSP_TIME_LIMIT_EXCEEDED; break; }
    }
    // pos[] in HBM follows the order[] written back; BEST stopped early gets its recomputed cost
    const int rc = tsp_grid_after_external_run(t, mode, status == TSP_TIME_LIMIT_EXCEEDED);
    return rc ? rc : status;
}
