// two_opt_exh.hpp -- the exhaustive best-improvement sweep in tour-position order: k_move_pos + k_exh.
// Part of the GRID engine; included by two_opt_grid.hip only (one translation unit).
//
// What it is for.  alg_2opt_tabu (src/tabusearch.c:127-157) executes the delta expression for every non-adjacent pair
// of every sweep: 49 985 000 at n = 10 000.  The product's sweeps decide most pairs by bounds (two_opt_sweep.hpp,
// two_opt_cluster.hip); THIS kernel executes every one of them exactly -- it is the sweep SURVEY.md 8(d) defines the
// evals/s headline on ("every one of 49 985 000 pairs evaluated per sweep"), bench.py's timed region, and what a caller
// gets with TSP_NO_FILTER=1 on the integer-coordinate metrics.
//
// Why position order.  delta(i, j) = d(a, b) + d(a1, b1) - d(a, a1) - d(b, b1) with a1 = succ a, b1 = succ b.  In the
// reference's node-id order the four distances of a pair are unrelated to its neighbours'.  Number the nodes by tour
// position instead -- u_p = node at position p, D(p, q) = d(u_p, u_q), e[p] = d(u_p, u_{p+1}) -- and
//     delta(p, q) = D(p, q) + D(p + 1, q + 1) - e[p] - e[q]:
// the second new edge of pair (p, q) IS the first new edge of pair (p + 1, q + 1).  A lane that owns column q and walks
// down the rows therefore needs ONE new distance per pair (the other arrives from its left neighbour, which computed it
// one row earlier) instead of two, and both removed edges come from arrays built once per sweep.  A best-improvement
// sweep may visit the pairs in any order: the decision is the arg-min of (delta, (i, j)) with the nodes' own ids as the
// tie-break (strict '<' at tabusearch.c:151 keeps the first pair in (i < j) order), which is what every lane keeps.
// Every delta is still computed exactly (integer-valued distances from the exact roots of tsp_dist.hpp's int_root).
//
// Layout.  k_move_pos (one thread per position) carries the previous sweep's move out of place (as k_move_recs does)
// and writes, in position order and padded: pxy[p] = coordinates of u_p (position n repeats position 0; further pads lie
// far outside the instance), pe[p] = e[p] as int32, pid[p] = u_p.
// k_exh: the pair-columns are cut into strips of W - 1 (W = 64 RJ columns of D per wave, RJ adjacent columns per lane);
// a strip's rows are its units of work, and the units of all strips, laid end to end, are dealt to the waves in equal
// contiguous ranges (every wave does the same number of row steps; at most two strips per wave).  Per row step a lane
// computes D(p, q) for its RJ columns (row operands are wave-uniform: scalar loads, no LDS), forms
//     sum = [D(p - 1, q - 1) - e[q - 1]] + D(p, q)           (the bracket: own register, or v_add_u32_dpp wave_shr:1)
// and compares it with wbd + e[p - 1], wbd = the best delta any lane of the WAVE has seen so far: a wave-uniform bound, so the
// common path is one integer minimum per pair and one compare against a scalar per row step, and a rarely taken branch
// does the exact bookkeeping (delta, tie-break on node ids) for the few pairs that reach that bound -- about ln(pairs of the
// wave) times per wave.  Measured and not kept: a per-lane bound (a wave's 64 RJ pairs per row step held one that beat its
// own lane's best almost every step: 47.7 us per sweep at RJ = 4); a bound shared by all waves through one word per tour
// (atomicMin on improvement, re-read every 16 rows: 4 096 waves on one L2 line cost more than the pruning saves, 69.6 us).
#pragma once
#include "two_opt_step.hpp"

#pragma clang fp contract(off)

#ifndef TSP_EXH_EXP
#define TSP_EXH_EXP 0
#endif

namespace tsp {

#ifdef TSP_STAMPS
// diagnostic build: per wave {100 MHz wall clock at start, at the end of its rows, shader cycles in between, 0}; [4] last
// candidate published, [5] apply done in g_exh_t (tools/diag_exh.py reads them).  Plain stores to slots of their own: 4 096
// atomics on one word at the start of a kernel would be the thing measured.
__device__ unsigned long long g_exh_w[8192 * 4];
__device__ unsigned long long g_exh_t[8];
#endif

constexpr int kExhPad = 1152;          // positions past n that k_move_pos fills (>= the widest strip + 2)
constexpr int kExhCluster = 32;        // blocks per first-level arrival counter
constexpr int kRowBatch = 4;           // rows whose operands one scalar load instruction fetches

template <int WT>
constexpr bool exh_metric() { return WT == WT_EUC_2D_ICOORD || WT == WT_CEIL_2D_ICOORD || WT == WT_ATT_ICOORD; }

// the exact integer-valued distance of tsp_dist.hpp's int_root as an int32
template <int WT>
__device__ __forceinline__ int exh_dist(double cx, double cy, double rx, double ry) {
    const double dx = cx - rx, dy = cy - ry;                  // exact: integer operands
    const double s = __builtin_fma(dx, dx, dy * dy);          // exact: < 2^53, so fused or not is the same number
    if constexpr (WT == WT_EUC_2D_ICOORD) {
        const double k = floor(__builtin_amdgcn_sqrt(s) + 0.25);
        const double e = __builtin_fma(-k, k, s);
        return __double2int_rz(k) + (e > k ? 1 : 0);          // s > k^2 + k: the root rounds to k + 1
    } else if constexpr (WT == WT_CEIL_2D_ICOORD) {
        const double k = floor(__builtin_amdgcn_sqrt(s) + 0.75);
        const double e = __builtin_fma(-k, k, s);
        return __double2int_rz(k) + (e > 0.0 ? 1 : 0);
    } else {
        const double k = floor(__builtin_amdgcn_sqrt(s * 0.1) + 0.75);
        const double e = __builtin_fma(-10.0 * k, k, s);
        return __double2int_rz(k) + (e > 0.0 ? 1 : 0);
    }
}

// The same for the RJ columns of a lane against one row, STAGE BY STAGE across the columns: the wave issues in order, so the RJ
// dependent chains (sub -> mul -> fma -> sqrt -> add -> floor -> fma -> compare -> convert -> add-with-carry) only overlap if
// their instructions are interleaved in the stream.  Left to itself the scheduler keeps most of a chain together (fewer live
// registers) and a lone wave then runs the chains one after the other (RJ = 16, one wave per SIMD: 2 960 cycles per row step
// against 1 056 of issue).  The scheduling barriers pin the stage order.
template <int WT, int RJ>
__device__ __forceinline__ void exh_dist_row(const double (&cx)[RJ], const double (&cy)[RJ], double rx, double ry, int (&D)[RJ]) {
    double s[RJ], k[RJ], e[RJ];
#pragma unroll
    for (int q = 0; q < RJ; ++q) { const double dx = cx[q] - rx, dy = cy[q] - ry; s[q] = __builtin_fma(dx, dx, dy * dy); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < RJ; ++q) k[q] = __builtin_amdgcn_sqrt(WT == WT_ATT_ICOORD ? s[q] * 0.1 : s[q]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < RJ; ++q) k[q] = floor(k[q] + (WT == WT_EUC_2D_ICOORD ? 0.25 : 0.75));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < RJ; ++q) e[q] = __builtin_fma(WT == WT_ATT_ICOORD ? -10.0 * k[q] : -k[q], k[q], s[q]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < RJ; ++q) D[q] = __double2int_rz(k[q]) + ((WT == WT_EUC_2D_ICOORD ? e[q] > k[q] : e[q] > 0.0) ? 1 : 0);
    __builtin_amdgcn_sched_barrier(0);
}

// (1) the pending move, out of place; (2) the tour AFTER that move in position order: coordinates, edge lengths, ids.
template <int WT, bool INT>
__global__ __launch_bounds__(kScanThreads) void k_move_pos(const double2 *__restrict__ coord, int *orders, int *poss, int *orders2,
                                                           int *poss2, const TourState *__restrict__ states, double2 *__restrict__ pxy,
                                                           int *__restrict__ pe, int *__restrict__ pid, int n) {
    const int tour = blockIdx.y;
    const TourState *st = states + tour;
    if (st->done) return;
    const size_t base = (size_t)tour * n, pbase = (size_t)tour * (n + kExhPad);
    const MoveView mv = move_view(st, orders + base, poss + base, orders2 + base, poss2 + base, n);
    const int k = blockIdx.x * kScanThreads + threadIdx.x;
    if (k >= n + kExhPad) return;
    // every load of the current copy (which this kernel never writes) comes before the first store
    const int u = k < n ? mv.node_at(k) : (k == n ? mv.node_at(0) : -1);
    const int v = k < n ? mv.node_at(k + 1 == n ? 0 : k + 1) : -1;
    const double2 c0 = coord[0];
    double2 cu = make_double2(c0.x - 6.0e6, c0.y - 6.0e6);   // pads: farther from every node than any tour edge is long
    int len = 0;
    if (u >= 0) cu = coord[u];
    if (v >= 0) { const double2 cv = coord[v]; len = (int)dist_xy<WT, INT>(cu.x, cu.y, cv.x, cv.y); }
    if (mv.L > 0 && k < n) {
        int *o_new = (st->parity ? orders : orders2) + base, *p_new = (st->parity ? poss : poss2) + base;
        o_new[k] = u;
        p_new[u] = k;
    }
    pxy[pbase + k] = cu;
    pe[pbase + k] = len;
    pid[pbase + k] = u;
}

// (RJ = 8 is built for three waves per SIMD: 170 VGPRs; RJ = 16 is meant for one or two waves per SIMD -- many independent
// distances per wave instead of many waves)
template <int WT, bool INT, int RJ>
__global__ __launch_bounds__(kScanThreads, (RJ == 8 ? 3 : 1)) void k_exh(const StepArgs a, const double2 *__restrict__ pxy_all, const int *__restrict__ pe_all,
                                                      const int *__restrict__ pid_all, int waves_total, int prio_on, int4 share, int gens) {
    // the position arrays are kernel arguments of their own, restrict-qualified: the row operands are wave-uniform loads, and
    // the compiler only issues them as scalar loads (s_load: no vector-memory slot, no VGPRs) when it can prove that the
    // kernel's own stores and atomics (candidate slots, tickets) never touch them
    static_assert(exh_metric<WT>(), "integer-coordinate metrics only");
    constexpr int W = 64 * RJ, WEFF = W - 1;
    const int tour = blockIdx.z;
    const TourState *st = a.states + tour;
    if (st->done) return;
    const int n = a.n, tid = threadIdx.x, lane = tid & 63;
    const size_t pbase = (size_t)tour * (n + kExhPad);
    const double2 *__restrict__ pxy = pxy_all + pbase;
    const int *__restrict__ pe = pe_all + pbase;
    const int *__restrict__ pid = pid_all + pbase;

#ifdef TSP_STAMPS
    const unsigned long long stamp_r0 = wall_clock64(), stamp_c0 = clock64();
    unsigned long long stamp_hits = 0, stamp_hit_cycles = 0;
#endif
    // ---- this wave's share: units [u_lo, u_hi) of the strips' rows laid end to end --------------------------------
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (kScanThreads / 64) + (tid >> 6));
    const int strips = (n + WEFF - 1) / WEFF;                       // pair-columns 0 .. n-1
    long long total = 0;
    for (int s = 0; s < strips; ++s) total += min(n - 1, s * WEFF + WEFF - 1);   // pair-rows p' < q' <= Q0 + WEFF - 1, p' <= n - 2
    // A SIMD serves its oldest wave first, and the workgroups of a CU are as old as their place in the grid: the first quarter of
    // the grid leaves its rows at 13.7 us, the others at 20.6 / 28.2 / 35.4 (equal shares; tools/diag_exh.py) -- while four waves
    // are active the SIMD's issue slots go 53 / 26 / 13 / 8 %.  share = the rows per wave of each part of the grid (quarters at four workgroups per CU) in those
    // proportions (host: tsp_dev_tours_create), so that the four waves of a SIMD finish together; share.x == 0: equal shares.
    long long per = (total + waves_total - 1) / waves_total;
    long long u_lo = per * gw;
    if (share.x > 0 && gens > 0) {   // gens = workgroups per CU = equal parts of the grid with a share of their own (2 .. 4)
        const int wq = waves_total / gens, g = min(gens - 1, gw / wq), idx = gw - g * wq;
        const int sh[4] = {share.x, share.y, share.z, share.w};
        u_lo = 0;
        for (int q = 0; q < g; ++q) u_lo += (long long)sh[q] * wq;
        per = sh[g];
        u_lo += per * idx;
    }
    long long u_hi = min(total, u_lo + per);
    u_lo = min(u_lo, total);

    int bd = -1, bp = -1, bq = -1;     // integer costs: delta < 0  <=>  delta <= -1; (bd, no pair) loses every tie
    int wbd = -1;                      // wave-uniform: the lowest delta any lane of this wave has seen

    // exact bookkeeping for a pair that reached the wave's best: delta, then -- on a tie only -- the reference's tie-break on
    // node ids (two dependent global loads: not on the path of a strict improvement)
    auto consider = [&](int sum, int erow, int pp, int qq, bool valid) {
        const int d = sum - erow;
        if (valid && d <= bd) {
            bool take = d < bd || bp < 0;
            if (!take) {
                const int i1 = pid[pp], j1 = pid[qq], i0 = pid[bp], j0 = pid[bq];
                take = make_key(min(i1, j1), max(i1, j1)) < make_key(min(i0, j0), max(i0, j0));
            }
            if (take) { bd = d; bp = pp; bq = qq; }
        }
    };

    long long cum = 0;
    for (int s = 0; s < strips && u_lo < u_hi; ++s) {
        const int rows_s = min(n - 1, s * WEFF + WEFF - 1);
        if (u_lo >= cum + rows_s) { cum += rows_s; continue; }
        // segment of strip s: pair-rows [pa, pb)
        const int pa = (int)(u_lo - cum), pb = (int)min<long long>(rows_s, u_hi - cum);
        const long long u_lo_now = u_lo;
        u_lo = cum + pb;
        cum += rows_s;
        const int Q0 = s * WEFF;
        double cx[RJ], cy[RJ];
        int ce[RJ], S[RJ], qk[RJ];
#pragma unroll
        for (int k = 0; k < RJ; ++k) {
            qk[k] = Q0 + RJ * lane + k;
            const double2 c = pxy[qk[k]];
            cx[k] = c.x; cy[k] = c.y; ce[k] = pe[qk[k]];
        }
        {   // row pa: distances only
            const double2 r = pxy[pa];
#pragma unroll
            for (int k = 0; k < RJ; ++k) S[k] = exh_dist<WT>(cx[k], cy[k], r.x, r.y) - ce[k];
        }
        // rows p = pa + 1 .. pb: D(p, .), then the pairs (p - 1, q - 1).  Up to p = Q0 every column of the strip lies above
        // the row (q_k > p for every evaluated pair); beyond it the pairs on and below the diagonal are masked.
        auto step = [&](int p, const double2 r_in, const int erow, auto pred_c) {
            constexpr bool PRED = decltype(pred_c)::value;
            int D[RJ], sum[RJ];
#if TSP_EXH_EXP == 1
            double2 r;   // experiment: row operands in VGPRs
            asm volatile("v_mov_b32 %0, %1" : "=v"(((int *)&r)[0]) : "s"(((const int *)&r_in)[0]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(((int *)&r)[1]) : "s"(((const int *)&r_in)[1]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(((int *)&r)[2]) : "s"(((const int *)&r_in)[2]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(((int *)&r)[3]) : "s"(((const int *)&r_in)[3]));
#else
            const double2 r = r_in;
#endif
            exh_dist_row<WT, RJ>(cx, cy, r.x, r.y, D);
            // lane 0's first column has no left neighbour in this wave (the DPP hands it 0): its pair belongs to the strip on
            // the left; should D alone ever pass the test below, the bookkeeping drops it (valid == false)
            sum[0] = __builtin_amdgcn_update_dpp(0, S[RJ - 1], 0x138 /* wave_shr:1 */, 0xf, 0xf, true) + D[0];
#pragma unroll
            for (int k = 1; k < RJ; ++k) sum[k] = S[k - 1] + D[k];
            const int thr = wbd + erow;   // scalar
            bool hit = false;
#pragma unroll
            for (int k = 0; k < RJ; ++k) hit = hit || (sum[k] <= thr && (!PRED || qk[k] > p));
#pragma unroll
            for (int k = 0; k < RJ; ++k) S[k] = D[k] - ce[k];
            if (__builtin_expect(__any(hit), 0)) {
#ifdef TSP_STAMPS
                const unsigned long long sc0 = clock64();
#endif
#pragma unroll
                for (int k = 0; k < RJ; ++k) consider(sum[k], erow, p - 1, qk[k] - 1, (k > 0 || lane > 0) && qk[k] > p);
                const int nb = (int)(unsigned)(wave_min_u64((u64)((unsigned)bd ^ 0x80000000u)) ^ 0x80000000u);
                wbd = min(wbd, nb);
#ifdef TSP_STAMPS
                stamp_hits += 1; stamp_hit_cycles += clock64() - sc0;
#endif
            }
        };
        // The row operands are wave-uniform: scalar loads, kRowBatch rows per load instruction, and the NEXT batch is on its way
        // while this one is worked (a scalar load that misses the CU's constant cache takes longer than one row step: with a
        // prefetch distance of one row the waves spent a quarter of their cycles in s_waitcnt, SQ_WAIT_ANY).  Positions up to
        // n + kRowBatch exist: k_move_pos pads.
        struct RowsXY { double2 r[kRowBatch]; };
        struct RowsE { int e[kRowBatch]; };
        const int p_plain = min(pb, Q0);
        int p = pa + 1;
        RowsXY nx = *reinterpret_cast<const RowsXY *>(pxy + p);
        RowsE ne = *reinterpret_cast<const RowsE *>(pe + p - 1);
        for (; p <= pb; p += kRowBatch) {
            // A SIMD serves its oldest wave first: left alone, the four waves of a SIMD leave their rows one after the other
            // (13, 21, 29, 36 us measured) and the last one runs alone, latency-bound, at half the issue rate.  Every wave has
            // the same number of rows, so a wave raises its priority with the share of them it still has to do: whoever is
            // behind goes first, and the waves of a SIMD finish together.
            if (prio_on) {
                const int lvl = (int)min<long long>(3, (u_hi - u_lo_now - (p - pa - 1)) * 4 / per);
                if (lvl >= 3) __builtin_amdgcn_s_setprio(3);
                else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
                else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            const RowsXY cx4 = nx;
            const RowsE ce4 = ne;
            nx = *reinterpret_cast<const RowsXY *>(pxy + p + kRowBatch);
            ne = *reinterpret_cast<const RowsE *>(pe + p + kRowBatch - 1);
            if (p + kRowBatch - 1 <= p_plain) {
#pragma unroll
                for (int u = 0; u < kRowBatch; ++u) step(p + u, cx4.r[u], ce4.e[u], std::false_type{});
            } else {
#pragma unroll
                for (int u = 0; u < kRowBatch; ++u)
                    if (p + u <= pb) step(p + u, cx4.r[u], ce4.e[u], std::true_type{});   // the predicate is harmless above the diagonal
            }
        }
    }

    // ---- the wave's, the block's, the tour's arg-min (delta, (i, j)) ------------------------------------------------
#ifdef TSP_STAMPS
    if (lane == 0 && tour == 0) {
        const int w = (int)blockIdx.x * (kScanThreads / 64) + (tid >> 6);
        if (w < 8192) { g_exh_w[4 * w] = stamp_r0; g_exh_w[4 * w + 1] = wall_clock64(); unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_exh_w[4 * w + 2] = (clock64() - stamp_c0) | ((unsigned long long)hwid << 32); g_exh_w[4 * w + 3] = 1 | (stamp_hits << 8) | ((unsigned long long)(xcc & 0xf) << 60) | ((stamp_hit_cycles & 0xfffffffffull) << 24); }
    }
#endif
    // only the lanes that hold the wave's lowest delta need their pair's node ids (usually one lane: one pair of loads)
    double d = 0.0;
    u64 key = kNoKey;
    {
        const int wmin = (int)(unsigned)(wave_min_u64((u64)((unsigned)bd ^ 0x80000000u)) ^ 0x80000000u);
        if (bp >= 0 && bd == wmin) {
            const int i = pid[bp], j = pid[bq];
            d = (double)bd;
            key = make_key(min(i, j), max(i, j));
        }
    }
    __shared__ double s_d[kScanThreads / 64];
    __shared__ u64 s_k[kScanThreads / 64];
    __shared__ int s_last;
    block_argmin<true>(d, key, s_d, s_k);
    if (tid == 0) {
        publish_partial(a.partials + (size_t)tour * a.partial_per_tour + blockIdx.x, d, key_i(key), key_j(key));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left this CU before the ticket
#ifdef TSP_STAMPS
        atomicMax(&g_exh_t[4], wall_clock64());
#endif
        // arrivals on one word are served one after the other: count per cluster of blocks first, then the clusters
        const int Q = ((int)gridDim.x + kExhCluster - 1) / kExhCluster, q = (int)blockIdx.x / kExhCluster;
        const int members = min(kExhCluster, (int)gridDim.x - q * kExhCluster);
        gi32 *ct = (gi32 *)(a.cl_tickets + ((size_t)tour * 64 + q) * 64);
        s_last = 0;
        if (__hip_atomic_fetch_add(ct, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == members) {
            __hip_atomic_store(ct, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int old = __hip_atomic_fetch_add((gi32 *)(a.tickets + tour), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old + 1 == Q);
            if (s_last) __hip_atomic_store((gi32 *)(a.tickets + tour), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!s_last) return;
#ifdef TSP_STAMPS
    __shared__ unsigned long long stamps[16];
    apply_step<WT, INT, TSP_2OPT_BEST, 2, false, true>(a, tour, 0, n - 1, stamps, nullptr);
    if (tid == 0) atomicMax(&g_exh_t[5], wall_clock64());
#else
    apply_step<WT, INT, TSP_2OPT_BEST, 2, false, true>(a, tour, 0, n - 1, nullptr);
#endif
}

}  // namespace tsp
