// two_opt_grid.hip -- GRID engine: 2-opt on tours resident in HBM, many workgroups per tour.
//
// Tour state in HBM is two int arrays per tour, order[p] (node at tour position p) and pos[v]
// (position of node v); succ(v) = order[pos[v]+1].  Coordinates are per instance.
//
// One step = ONE launch of k_step: every block evaluates its tile of (i<j) node pairs against the
// current tour and publishes one candidate; the block that arrives last (per-tour countdown
// ticket) picks the winner, reverses the tour segment and advances the tour's control block for
// the next launch.  The host only queues steps and polls `done`; every decision of the
// reference's loops is taken on the device:
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
#include "two_opt_common.hpp"

#include <algorithm>
#include <time.h>

#pragma clang fp contract(off)

namespace tsp {

constexpr int kMaxRowsPerBlock = 256;

// Diagnostic build only (-DTSP_STAMPS): 100 MHz wall-clock stamps of the last block of each step,
// accumulated into a buffer nothing else reads (cdna_hip_programming.md section 7, in-kernel stamps).
#ifdef TSP_STAMPS
__device__ unsigned long long g_stamp_sum[16];
__device__ unsigned long long g_stamp_n;
__device__ unsigned long long g_blk[1024][8];
__device__ unsigned long long g_blk2[1024][4];   // wave 0 of each block: row quads, quads entering tier 1, tier 2, diagonal half-units   // per k_sweep block (plain accumulation): launches, kept, ticks tests->staged, ticks tests->loop end
__device__ unsigned long long g_sw_cnt[8];   // k_sweep: survivors kept, max kept, row quads, with tier 1, with exact, blocks, loop ticks, max loop ticks
__device__ unsigned long long g_clk_core, g_clk_real;   // row-loop time of every block: shader cycles vs 100 MHz ticks
#define TSP_STAMP(k) do { if (threadIdx.x == 0) stamps[k] = wall_clock64(); } while (0)
#else
#define TSP_STAMP(k) do { } while (0)
#endif

__global__ void k_build_pos(const int *__restrict__ orders, int *__restrict__ poss, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const size_t base = (size_t)blockIdx.y * n;
    poss[base + orders[base + p]] = p;
}

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

// ---- in-launch hand-off of the block candidates ---------------------------------------------
// Producer (lane 0 of each block): two 8-byte write-through (sc1) stores, drain, then one relaxed
// agent-scope countdown on the tour's ticket.  Consumer (the block whose decrement returned 1):
// sc1 loads after the block barrier that the decrementing wave joins.  Every slot is written once
// and read once per launch and launches are separated by kernel boundaries, so no stale copy of a
// slot can sit in the reader's caches (cdna_hip_programming.md G16 / MI355X_MICROARCH.md
// "Valid forms", first row).  order/pos/state are only written by the last block, after every
// other block of the tour has finished, and are next read in the following launch.
using gu64 = __attribute__((address_space(1))) unsigned long long;
using gi32 = __attribute__((address_space(1))) int;

__device__ __forceinline__ void publish_partial(Partial *slot, double delta, int i, int j) {
    gu64 *g = (gu64 *)slot;
    __hip_atomic_store(g, (u64)__double_as_longlong(delta), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 1, ((u64)(unsigned)j << 32) | (u64)(unsigned)i, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void read_partial(const Partial *slot, double &delta, int &i, int &j) {
    gu64 *g = (gu64 *)slot;
    const u64 a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    delta = __longlong_as_double((long long)a);
    i = (int)(b & 0xffffffffu);
    j = (int)(b >> 32);
}

// Blocks of a step that own at least one column above their first row (the others return at
// once and take no ticket).  Block (bx, by) is skipped iff bx < (r0(by) + 1) / TJ.
__device__ __forceinline__ int skipped_in_tile_row(int r0, int gx, int TJ) { return min(gx, (r0 + 1) / TJ); }

__device__ __forceinline__ int count_active_blocks(int row_lo, int row_hi, int rpb, int gx, int gy, int TJ,
                                                   int *scratch) {
    const int tile_rows = min((row_hi - row_lo + rpb - 1) / rpb, gy);
    int c = 0;
    for (int by = threadIdx.x; by < tile_rows; by += (int)blockDim.x)
        c += gx - skipped_in_tile_row(row_lo + by * rpb, gx, TJ);
    return block_sum<int>(c, scratch);
}

template <int MODE>
__device__ __forceinline__ void active_rows(const TourState *st, int n, int &row_lo, int &row_hi) {
    row_lo = 0; row_hi = n - 1;
    if constexpr (MODE == TSP_2OPT_FIRST) {
        row_lo = st->ci;
        row_hi = min(st->ci + st->chunk_rows, n - 1);
    }
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

// ---- tour cost ----------------------------------------------------------------------------
// Sum over nodes of d(v, succ v) in node order (tabusearch.c:168-172), by one whole block.
template <int WT, bool INT>
__device__ __forceinline__ double tour_cost_block(const double2 *coord, const int *order, const int *pos, int n,
                                                  double *s_d /*>=16*/, double *s_chunk /*4096 unless INT*/) {
    const int tid = threadIdx.x;
    if constexpr (INT || WT == WT_CEIL_2D) {  // integer-valued terms: any order is exact
        double c = 0.0;
        for (int v = tid; v < n; v += (int)blockDim.x) c += load_node<WT, INT>(coord, order, pos, n, v).ds;
        return block_sum<double>(c, s_d);
    } else {  // same sequential order as the reference, staged through LDS
        double acc = 0.0;
        for (int base = 0; base < n; base += 4096) {
            __syncthreads();
            for (int t = tid; t < 4096 && base + t < n; t += (int)blockDim.x)
                s_chunk[t] = load_node<WT, INT>(coord, order, pos, n, base + t).ds;
            __syncthreads();
            if (tid == 0) {
                const int m = min(4096, n - base);
                for (int t = 0; t < m; ++t) acc += s_chunk[t];
            }
        }
        __syncthreads();
        if (tid == 0) s_d[0] = acc;
        __syncthreads();
        return s_d[0];
    }
}

// out[b] = recomputed cost of tour b (BEST runs that stop early; multi-start "true cost")
template <int WT, bool INT>
__global__ __launch_bounds__(kApplyThreads) void k_tour_cost(const double2 *__restrict__ coord,
                                                             const int *__restrict__ orders,
                                                             const int *__restrict__ poss, int n,
                                                             double *__restrict__ out, size_t out_stride_bytes) {
    __shared__ double s_d[kApplyThreads / 64];
    __shared__ double s_chunk[(INT || WT == WT_CEIL_2D) ? 1 : 4096];
    const size_t base = (size_t)blockIdx.x * n;
    const double c = tour_cost_block<WT, INT>(coord, orders + base, poss + base, n, s_d, s_chunk);
    if (threadIdx.x == 0)
        *reinterpret_cast<double *>(reinterpret_cast<char *>(out) + blockIdx.x * out_stride_bytes) = c;
}

// ---- apply: executed by the last block of a tour's step ------------------------------------
struct StepArgs {
    const double2 *coord;
    int *orders;
    int *poss;
    TourState *states;
    Partial *partials;
    int *tickets;      // per tour
    int *row_tickets;  // per tour x tile row (BEST: two-level hand-off)
    Partial *row_slots; // per tour x tile row
    int *row_evals;    // per tour x tile row (tabu runs)
    int max_tile_rows;
    int *slot_evals;   // tabu runs only
    int *tabu;
    const NodeRec *recs;   // BEST: materialised by k_recs before the step; nullptr = derive per tile
    size_t partial_per_tour;
    int n, rows_per_block, first_min_rows, first_max_rows, count_evals, iter, tenure;
    int slot;          // k_first: which of the tour's two control blocks this launch reads (the other is written)
    double margin;     // root filter (tsp_dist.hpp); 1e300 = every pair is evaluated exactly
    double prune;      // new-edge bound margin (tsp_dist.hpp); 1e300 = never prune
    // sorted sweep (k_sweep): records in Hilbert-rank order, group boxes, per-group longest edge, shared bound
    double sum_margin; // k_sweep tier 1: rounding of the two new distances + fp slack (doubled: keeps ties)
    const int *pairtab;    // k_sweep: group pairs (r << 16 | c, -1 = none) per cluster, or nullptr (computed)
    int *cl_tickets;       // k_sweep: per tour x cluster arrival counters, 64 ints apart
    int *orders2, *poss2;  // k_sweep / k_move_recs: the second copy of order/pos (TourState::parity says which is current)
    const double4 *gbox;
    const double *gmax;
    unsigned long long *gbest;
    int ng, n_slots, flat_slots;
};

template <int WT, bool INT, int MODE, int RJ, bool TABU, bool FLAT = false>
__device__ __forceinline__ void apply_step(const StepArgs &a, int tour, int row_lo, int row_hi
#ifdef TSP_STAMPS
                                           , unsigned long long *stamps
#endif
) {
    constexpr int TJ = kScanThreads * RJ;
    const int n = a.n, rpb = a.rows_per_block;
    const int gx = gridDim.x, gy = gridDim.y;
    TourState *st = a.states + tour;
    const int tid = threadIdx.x;
    int *order = a.orders + (size_t)tour * n;
    int *pos = a.poss + (size_t)tour * n;
    int cur_parity = 0;
    if constexpr (FLAT) {
        // k_move_recs has just carried the previous step's move out into the other copy: that one is current now
        cur_parity = st->parity ^ st->pending;
        if (cur_parity) { order = a.orders2 + (size_t)tour * n; pos = a.poss2 + (size_t)tour * n; }
    }
    const Partial *part = a.partials + (size_t)tour * a.partial_per_tour;

    __shared__ double s_d[kScanThreads / 64];
    __shared__ u64 s_k[kScanThreads / 64];
    __shared__ long long s_ll[kScanThreads / 64];
    __shared__ int s_i32[kScanThreads / 64];

    // BEST: one pre-reduced candidate per tile row; FLAT (sorted sweep): one candidate per block, all live
    constexpr bool HIER = MODE == TSP_2OPT_BEST && !FLAT;
    const int ci = MODE == TSP_2OPT_FIRST ? st->ci : 0, cj = MODE == TSP_2OPT_FIRST ? st->cj : 0;
    const int tile_rows = min((row_hi - row_lo + rpb - 1) / rpb, gy);
    const int nslots = FLAT ? a.flat_slots : (HIER ? tile_rows : tile_rows * gx);
    if constexpr (HIER) part = a.row_slots + (size_t)tour * a.max_tile_rows;

    // 1. winner over the blocks that published a candidate (loads batched: they are sc1 loads
    //    that go to memory, so eight slots per lane are kept in flight)
    double bd = 0.0;
    u64 key = kNoKey;
    long long tabu_evals = 0;
    constexpr int PU = 4;
    for (int s0 = tid; s0 < nslots; s0 += PU * kScanThreads) {
        double pd[PU]; int pi[PU], pj[PU]; bool live[PU];
#pragma unroll
        for (int k = 0; k < PU; ++k) {
            const int s = s0 + k * kScanThreads;
            const int by = s / gx, bx = s - by * gx;
            live[k] = s < nslots && (HIER || FLAT || bx >= skipped_in_tile_row(row_lo + by * rpb, gx, TJ));
            pd[k] = 0.0; pi[k] = -1; pj[k] = -1;
            if (live[k]) read_partial(part + s, pd[k], pi[k], pj[k]);
        }
#pragma unroll
        for (int k = 0; k < PU; ++k) {
            const u64 kk = make_key(pi[k], pj[k]);
            const bool take = live[k] && ((MODE == TSP_2OPT_BEST) ? better(pd[k], kk, bd, key) : (kk < key));
            if (take) { bd = pd[k]; key = kk; }
            if constexpr (TABU) {
                if (live[k])
                    tabu_evals += __hip_atomic_load(
                        (gi32 *)(a.row_evals + (size_t)tour * a.max_tile_rows + s0 + k * kScanThreads),
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    block_argmin<MODE == TSP_2OPT_BEST>(bd, key, s_d, s_k);
    TSP_STAMP(6);
    if constexpr (TABU) tabu_evals = block_sum<long long>(tabu_evals, s_ll);
    const bool found = key != kNoKey && (MODE == TSP_2OPT_FIRST || bd < 0);
    const int wi = found ? key_i(key) : -1, wj = found ? key_j(key) : -1;
    int pa = 0, pb = 0;
    if (found) { pa = pos[wi]; pb = pos[wj]; }

    // 2. FIRST: how many pairs between the old and the new cursor the reference would have skipped
    //    as adjacent (heuristics.c:471), on the tour the scan saw.  Row r's adjacent columns are
    //    succ(r) and pred(r), when they are > r.
    long long adj = 0;
    int ni = wi, nj = wj;  // new cursor
    if constexpr (MODE == TSP_2OPT_FIRST) {
        if (!found) { ni = row_hi - 1; nj = n - 1; }
        if (a.count_evals) {
            const u64 lo = make_key(ci, cj), hi = make_key(ni, nj);
            long long c = 0;
            for (int r = ci + tid; r <= ni; r += kScanThreads) {
                const int p = pos[r];
                const int s = order[p + 1 == n ? 0 : p + 1], q = order[p == 0 ? n - 1 : p - 1];
                const u64 ks = make_key(r, s), kq = make_key(r, q);
                c += (s > r && ks > lo && ks <= hi) ? 1 : 0;
                c += (q > r && kq > lo && kq <= hi) ? 1 : 0;
            }
            adj = block_sum<long long>(c, s_ll);
        }
    }
    __syncthreads();  // every read of the old order/pos is done
    TSP_STAMP(7);

    // 3. the move: reverse positions pa+1 .. pb (cyclic)
    int L = 0;
    if (found) { L = pb - pa; if (L < 0) L += n; }
    if (found && !FLAT) {   // FLAT: the move is left to the next launch of k_move_recs (all blocks, not one)
        const int half = L >> 1;
        constexpr int U = 4;
        for (int t0 = tid; t0 < half; t0 += U * kScanThreads) {
            int p[U], q[U], u[U], w[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int t = t0 + k * kScanThreads;
                p[k] = pa + 1 + t; if (p[k] >= n) p[k] -= n;
                q[k] = pb - t; if (q[k] < 0) q[k] += n;
                u[k] = 0; w[k] = 0;
                if (t < half) { u[k] = order[p[k]]; w[k] = order[q[k]]; }
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (t0 + k * kScanThreads < half) {
                    order[p[k]] = w[k]; order[q[k]] = u[k];
                    pos[w[k]] = p[k]; pos[u[k]] = q[k];
                }
            }
        }
    }

    TSP_STAMP(8);
    // 4. BEST at the local optimum: recomputed cost
    double final_cost = 0.0;
    if constexpr (MODE == TSP_2OPT_BEST) {
        if (!found) {
            __shared__ double s_chunk[(INT || WT == WT_CEIL_2D) ? 1 : 4096];
            final_cost = tour_cost_block<WT, INT>(a.coord, order, pos, n, s_d, s_chunk);
        }
    }

    // 5. next cursor / chunk, and the ticket for the next launch
    int done = 0, n_ci = 0, n_cj = 0, n_chunk = st->chunk_rows, sweep_end = 0;
    double obj = st->obj, seen = st->seen_cost;
    if constexpr (MODE == TSP_2OPT_BEST) {
        if (found) { obj = st->obj; } else { done = 1; obj = final_cost; }
    } else {
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
    }
    int next_lo = 0, next_hi = n - 1;
    if constexpr (MODE == TSP_2OPT_FIRST) { next_lo = n_ci; next_hi = min(n_ci + n_chunk, n - 1); }
    int next_active = 0;
    if constexpr (HIER) {   // count-up tickets back to zero for the next launch
        for (int k = tid; k < tile_rows; k += kScanThreads)
            __hip_atomic_store((gi32 *)(a.row_tickets + (size_t)tour * a.max_tile_rows + k), 0, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (!FLAT) {
        next_active = count_active_blocks(next_lo, next_hi, rpb, gx, gy, TJ, s_i32);
    }

    if (tid == 0) {
        st->steps += 1;
        if constexpr (MODE == TSP_2OPT_BEST) {
            st->sweeps += 1;
            st->evals += TABU ? tabu_evals : (long long)n * (n - 1) / 2 - n;  // non-adjacent pairs (n >= 4)
            st->pairs_scanned += (long long)n * (n - 1) / 2;
            if (found) { st->moves += 1; st->reversed += L - 1; }
        } else {
            const long long r_old = pair_rank(ci, cj, n);
            st->pairs_scanned += pair_rank(row_hi - 1, n - 1, n) - r_old;
            st->evals += pair_rank(ni, nj, n) - r_old - adj;
            if (found) { st->moves += 1; st->reversed += L - 1; }   // successors rewritten by utility.c:710-717
            st->sweeps += sweep_end;
            st->ci = n_ci; st->cj = n_cj; st->chunk_rows = n_chunk; st->seen_cost = seen;
        }
        st->obj = obj;
        st->done = done;
        if constexpr (FLAT) { st->parity = cur_parity; st->pending = found ? 1 : 0; st->mv_pa = pa; st->mv_pb = pb; }
        if constexpr (!HIER && !FLAT)
            __hip_atomic_store((gi32 *)(a.tickets + tour), done ? 0 : next_active, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
#ifdef TSP_STAMPS
        stamps[9] = wall_clock64();
        for (int k = 1; k < 10; ++k) atomicAdd(&g_stamp_sum[k], stamps[k] - stamps[k - 1]);
        atomicAdd(&g_stamp_n, 1ull);
#endif
    }
}

// ---- step kernel ------------------------------------------------------------------------------
// Block (bx, by, tour): rows r0 .. r0+rows_per_block of the tour's active row range, columns
// bx*256*RJ .. +256*RJ.  Prologue: the block derives the NodeRec of its rows (into LDS) and of its
// columns (RJ per lane, registers) from order/pos/coord -- three dependent loads and one sqrt per
// node, amortised over rows x columns evaluations.  Main loop: lanes own columns, the row record
// is a wave-uniform LDS broadcast; ~70 fp64 instructions per evaluation, no memory traffic.
template <int WT, bool INT, int MODE, int RJ, bool TABU>
__global__ __launch_bounds__(kScanThreads) void k_step(const StepArgs a) {
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
    const bool plain_tile = MODE == TSP_2OPT_BEST && !TABU && FILTER && EXACT_SUMS && c0 >= r1 && c0 + TJ <= n;
    if (plain_tile) {
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
    apply_step<WT, INT, MODE, RJ, TABU>(a, tour, row_lo, row_hi, stamps);
#else
    apply_step<WT, INT, MODE, RJ, TABU>(a, tour, row_lo, row_hi);
#endif
}

// ---- sorted sweep (BEST, sqrt metrics) ------------------------------------------------------------------------
// The new-edge bound says a pair (a, b) can only beat `bound` if |ab| < bound + d(a,succ a) + d(b,succ b) + margin.
// With the nodes ranked along a Hilbert curve, 64 consecutive ranks form a compact group, and the same bound
// with the groups' bounding boxes and their longest tour edges decides 64 x 64 pairs at once: on a constructed
// tour 80-95 % of the group pairs of a sweep never reach the pair loop.  Nothing about the result changes:
// every decision the reference takes (strict '<', first pair in (i<j) order among equal deltas) is taken on
// exact values with the nodes' own ids; the order in which pairs are visited is free in a best-improvement sweep.
//
// The move a sweep chose is carried out by the NEXT launch, by all of its blocks: order/pos exist twice, the
// reversal of positions pa+1 .. pb is a gather from the current copy into the other one
//     new_order[p] = old_order[mirror(p)],  mirror(p) = pa + 1 + (L - 1 - t) for t = (p - pa - 1) mod n < L, else p
// and the records of the next sweep are built from the same closed form, so nothing waits for the copy.
struct MoveView {
    const int *order, *pos;   // the current copy
    int n, pa1, L;            // pending reversal: positions pa1 .. pa1 + L - 1 (cyclic); L == 0: none
    __device__ __forceinline__ int mirror(int p) const {
        int t = p - pa1; if (t < 0) t += n;
        if (t >= L) return p;
        int q = pa1 + (L - 1 - t); if (q >= n) q -= n;
        return q;
    }
    __device__ __forceinline__ int node_at(int p) const { return order[mirror(p)]; }   // node at new position p
    __device__ __forceinline__ int pos_of(int v) const { return mirror(pos[v]); }       // the mirror is an involution
};

__device__ __forceinline__ MoveView move_view(const TourState *st, const int *o1, const int *p1, const int *o2,
                                              const int *p2, int n) {
    MoveView m;
    const bool second = st->parity != 0;
    m.order = second ? o2 : o1; m.pos = second ? p2 : p1; m.n = n;
    m.L = 0; m.pa1 = 0;
    if (st->pending) {
        int L = st->mv_pb - st->mv_pa; if (L < 0) L += n;
        m.L = L; m.pa1 = st->mv_pa + 1 == n ? 0 : st->mv_pa + 1;
    }
    return m;
}

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
    if (mv.L > 0 && k < n) {   // new position k
        int *o_new = (st->parity ? orders : orders2) + base, *p_new = (st->parity ? poss : poss2) + base;
        const int v = mv.node_at(k);
        o_new[k] = v;
        p_new[v] = k;
    }
    if (k >= n_slots) return;
    const int v = sperm[k];
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
    recs[(size_t)tour * n_slots + k] = r;
    double m = r.ds;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) gmax[(size_t)tour * (ng + 1) + (k >> 6)] = m;
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

// Non-positive deltas order like their bit patterns read as unsigned (more negative = larger).
__device__ __forceinline__ double gbest_load(unsigned long long *g) {
    return __longlong_as_double((long long)__hip_atomic_load((gu64 *)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
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

constexpr int kSweepCluster = 8;      // blocks that test the same group pairs and deal the survivors among themselves
constexpr int kSweepRows = 16;        // rows of one unit of wave work (64 / kSweepRows units per group pair)
constexpr int kSweepStage = 8;        // group pairs whose records a block holds in LDS at a time (6 KB each)
constexpr int kSweepListCap = 1024;   // survivors one block can hold (more are processed in further passes)

// k_sweep.  Blocks come in clusters of kSweepCluster.  Cluster q tests the group pairs t = q, q + Q, q + 2Q, ...
// (row group r against column group c >= r; c == r: the pairs inside the group) with the groups' boxes and longest
// edges -- one test per thread and round, every block of the cluster the same tests, so that all of them see the
// same ordered survivor list and block j keeps entries j, j + C, ...: the survivors of a sweep are very unevenly
// spread over the row groups (a group that holds one long edge survives against everything), the strided sample
// plus the deal spreads them evenly over the chip without a queue or a second launch.
// A wave then takes half a surviving group pair at a time: rows 32h .. 32h+31 of r against one column of c per
// lane.  The 32 row records are staged in the wave's own LDS strip and read back as wave-uniform broadcasts; the
// waves of a block share nothing until the block's arg-min.  Rows go four at a time so that the LDS reads and the
// fp64 chains of different rows overlap.
template <int WT, bool INT>
__global__ __launch_bounds__(kScanThreads, 3) void k_sweep(const StepArgs a) {   // 3 waves per SIMD: 768 blocks resident
    constexpr bool ATT10 = WT == WT_ATT || WT == WT_ATT_ICOORD;
    constexpr int NW = kScanThreads / 64;
#ifdef TSP_STAMPS
    __shared__ unsigned long long stamps[16];
#endif
    TSP_STAMP(0);
#ifdef TSP_STAMPS
    const unsigned long long bt0 = wall_clock64();
    unsigned long long bt3 = 0;
    unsigned long long dq = 0, dq1 = 0, dq2 = 0, ddiag = 0;
#endif
    const int tour = blockIdx.z;
    const TourState *st = a.states + tour;
    if (st->done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ng = a.ng;
    const NodeRec *rec = a.recs + (size_t)tour * a.n_slots;
    const double *gmax = a.gmax + (size_t)tour * (ng + 1);
    const double prune2 = 2.0 * a.prune;   // doubled: keeps ties (a lane does not meet its pairs in key order)
    const int Q = (int)gridDim.x / kSweepCluster, q = (int)blockIdx.x / kSweepCluster, j = (int)blockIdx.x % kSweepCluster;
    const int npairs = ng * (ng + 1) / 2;
    const int ntests = (npairs + Q - 1) / Q;   // stride blocks of group pairs; the last one may be partial

    __shared__ NodeRec s_stage[kSweepStage][128];   // per staged group pair: 64 row records, 64 column records
    __shared__ double4 s_cbox[kSweepStage];
    __shared__ double s_cgmax[kSweepStage];
    __shared__ int s_list[kSweepListCap];   // r << 16 | c
    __shared__ int s_wcount[NW];
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
#ifdef TSP_STAMPS
        const unsigned long long lt0 = wall_clock64();
        if (tid == 0 && blockIdx.x < 1024) { g_blk[blockIdx.x][0] += 1; g_blk[blockIdx.x][1] += kept; g_blk[blockIdx.x][2] += lt0 - bt0; g_blk[blockIdx.x][6] = bt0; }
#endif
        // ---- the block's survivors, kSweepStage group pairs at a time: all 128 records of each pair are fetched
        // by the whole block in one burst (one memory latency per chunk instead of one per unit of wave work),
        // then every wave takes half a group pair at a time out of LDS
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
#ifdef TSP_STAMPS
            if (tid == 0 && blockIdx.x < 1024 && e0 == 0) { bt3 = wall_clock64(); g_blk[blockIdx.x][3] += bt3 - lt0; }
#endif
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
                            const double delta = pair_delta<WT, INT>(ri[u], rj);
                            const u64 kk = make_key(min(ri[u].id, rj.id), max(ri[u].id, rj.id));
                            if (delta < bd || (delta == bd && delta < 0.0 && kk < key)) { bd = delta; key = kk; bound = bd; }
                        }
                    }
                }
            };
            // four live rows at a time (a short last group repeats its last row: the same pair twice changes nothing)
#ifdef TSP_STAMPS
            ddiag += (cgp == r) ? 1 : 0;
#endif
            while (alive) {
#ifdef TSP_STAMPS
                dq += 1;
                const unsigned long long ch0 = clock64();
#endif
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
#ifdef TSP_STAMPS
                const unsigned long long c2 = clock64();
                dq1 += c2 - ch0;
                if (any) { rare4(idx, need); dq2 += clock64() - c2; }
#else
                if (any) rare4(idx, need);
#endif
            }
        }
#ifdef TSP_STAMPS
            const unsigned long long bt4 = wall_clock64();
#endif
            __syncthreads();   // the stage (and, after the last chunk, s_list) is rewritten next
#ifdef TSP_STAMPS
            if (tid == 0 && blockIdx.x < 1024 && e0 == 0) { const unsigned long long bt5 = wall_clock64(); g_blk[blockIdx.x][4] += bt4 - bt3; g_blk[blockIdx.x][5] += bt5 - bt4; g_blk[blockIdx.x][7] = bt5; }
#endif
        }
    }

    __shared__ double s_d[NW];
    __shared__ u64 s_k[NW];
    __shared__ int s_last;
#ifdef TSP_STAMPS
    if (tid == 0 && blockIdx.x < 1024) { g_blk2[blockIdx.x][0] += dq; g_blk2[blockIdx.x][1] += dq1; g_blk2[blockIdx.x][2] += dq2; g_blk2[blockIdx.x][3] += ddiag; }
#endif
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
    apply_step<WT, INT, TSP_2OPT_BEST, 2, false, true>(a, tour, 0, a.n - 1, stamps);
#else
    apply_step<WT, INT, TSP_2OPT_BEST, 2, false, true>(a, tour, 0, a.n - 1);
#endif
}

// ---- first improvement (alg_2opt), second form: k_first ------------------------------------------------------
// Same decisions as k_step<FIRST> (first improving pair after the cursor in (i<j) order, heuristics.c:452-486),
// three things done differently, all about the latency of a step:
//  * the grid is fixed and small (gy tile rows); a block takes ceil(chunk / gy) rows, so the 32-row chunk that
//    follows every hit is spread over the whole chip one row per block instead of 8 rows on a few CUs, and no
//    launch dispatches thousands of blocks that return at once;
//  * the move is carried out by the NEXT launch, out of place, by all blocks (MoveView, as in k_move_recs): the
//    scan reads the tour through the closed form of the pending reversal, the last block only notes the move;
//  * the ticket counts up to a number every block works out for itself (no count left behind by the apply).
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
    // The working blocks of this step number themselves (tile rows first; gy <= 64, one lane per tile row):
    // `active` of them take a ticket, and block `widx` carries out slices widx, widx + active, ... of the pending
    // move -- only ticket holders touch the other copy, so all of it is written before the last block moves on.
    const int tile_rows = (row_hi - row_lo + rpb - 1) / rpb;
    __shared__ int s_active, s_widx, s_rowblocks;
    if (tid < 64) {
        const int mine = tid < tile_rows ? gx - skipped_in_tile_row(row_lo + tid * rpb, gx, TJ) : 0;
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t2 = __shfl_up(incl, off); if (tid >= off) incl += t2; }
        if (tid == (int)blockIdx.y) { s_widx = incl - mine + ((int)blockIdx.x - skipped_in_tile_row(r0, gx, TJ)); s_rowblocks = mine; }
        if (tid == 63) s_active = incl;
    }
    __syncthreads();
    const int active = s_active;
    if (mv.L > 0) {
        int *o_new = (st->parity ? a.orders : a.orders2) + base, *p_new = (st->parity ? a.poss : a.poss2) + base;
        for (int k = s_widx * kScanThreads + tid; k < n; k += active * kScanThreads) {
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

    u64 key = make_key(bi, bj);
    __shared__ double s_d[kScanThreads / 64];
    __shared__ u64 s_k[kScanThreads / 64];
    __shared__ long long s_ll[kScanThreads / 64];
    __shared__ int s_last;
    block_argmin<false>(bd, key, s_d, s_k);
    const Partial *part = a.partials + (size_t)tour * a.partial_per_tour;
    if (tid == 0) {
        publish_partial(a.partials + (size_t)tour * a.partial_per_tour + (size_t)blockIdx.y * gx + blockIdx.x, bd, key_i(key), key_j(key));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left this CU before the ticket
        // arrivals per tile row first (one word per row, 256 B apart), then the rows on the tour's word: arrivals on
        // one word are served one after the other
        gi32 *rt = (gi32 *)(a.cl_tickets + ((size_t)tour * 64 + blockIdx.y) * 64);
        s_last = 0;
        if (__hip_atomic_fetch_add(rt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == s_rowblocks) {
            __hip_atomic_store(rt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int old = __hip_atomic_fetch_add((gi32 *)(a.tickets + tour), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old + 1 == tile_rows);
            if (s_last) __hip_atomic_store((gi32 *)(a.tickets + tour), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!s_last) return;

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
    }
}

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

int env_int(const char *name, int dflt);

// BEST sweeps of this handle go through k_recs_sorted + k_sweep (no tabu list, metric with the new-edge bound)
bool sorted_sweep(const tsp_dev_tours *t) {
    return t->inst->d_sperm && t->d_gmax && t->d_order2 && t->n >= t->sorted_min_n && t->inst->prune_margin < 1e299 &&
           t->inst->ng < 65535;   // k_sweep packs (r, c) into one int and counts group pairs in 31 bits
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
    a.gbox = t->inst->d_gbox; a.gmax = t->d_gmax; a.gbest = t->d_gbest;
    a.ng = t->inst->ng; a.n_slots = t->inst->n_slots;
    a.flat_slots = t->sweep_blocks;
    return a;
}

template <int WT, bool INT>
int launch_step(tsp_dev_tours *t, int mode, tsp_dev_tabu *tabu, int iter, int tenure) {
    hipStream_t s = t->inst->ctx->stream;
    StepArgs a = make_args(t, mode, tabu, iter, tenure);
    if constexpr (has_root_filter<WT>()) {
        if (mode == TSP_2OPT_BEST && !tabu && sorted_sweep(t)) {
            a.recs = t->d_rec;
            const int ns = t->inst->n_slots;
            hipLaunchKernelGGL((k_move_recs<WT, INT>), dim3((std::max(ns, t->n) + kScanThreads - 1) / kScanThreads, t->B),
                               dim3(kScanThreads), 0, s, t->inst->d_coord, t->d_order, t->d_pos, t->d_order2, t->d_pos2,
                               t->d_state, t->inst->d_sperm, t->d_rec, t->d_gmax, t->n, t->inst->ng, ns);
            hipLaunchKernelGGL((k_sweep<WT, INT>), dim3(t->sweep_blocks, 1, t->B), dim3(kScanThreads), 0, s, a);
            return TSP_OK;
        }
    }
    if (mode == TSP_2OPT_BEST) {
        const dim3 g = scan_grid<TSP_2OPT_BEST>(t);
        if (a.recs)
            hipLaunchKernelGGL((k_recs<WT, INT>), dim3((t->n + kScanThreads - 1) / kScanThreads, t->B), dim3(kScanThreads), 0, s,
                               t->inst->d_coord, t->d_order, t->d_pos, t->d_state, t->d_rec, t->n);
        if (tabu)
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

int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

// Shared with the other translation units of the library.
int tsp_grid_run(tsp_dev_tours *t, int mode, tsp_dev_tabu *tabu, int iter, int tenure, int64_t max_steps,
                 double time_limit_s, int sync, int *all_done) {
    if (!t || (mode != TSP_2OPT_FIRST && mode != TSP_2OPT_BEST)) return TSP_DEV_E_ARG;
    hipStream_t s = t->inst->ctx->stream;
    const double t0 = wall_s();
    const int64_t batch = 64;
    int64_t queued = 0;
    int status = TSP_OK;
    if (all_done) *all_done = 0;
    launch_arm(t, mode);
    for (;;) {
        int64_t todo = batch;
        if (max_steps >= 0) todo = std::min<int64_t>(batch, max_steps - queued);
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
        if (done) { if (all_done) *all_done = 1; break; }
        if (time_limit_s > 0 && wall_s() - t0 > time_limit_s) { status = TSP_TIME_LIMIT_EXCEEDED; break; }
    }
    if ((mode == TSP_2OPT_BEST && !tabu && sorted_sweep(t)) || (mode == TSP_2OPT_FIRST && !t->first_v1)) {
        launch_flush(t);
        TSP_HIP_TRY(hipGetLastError());
        if (sync) TSP_HIP_TRY(hipStreamSynchronize(s));
    }
    if (status == TSP_TIME_LIMIT_EXCEEDED && mode == TSP_2OPT_BEST) {
        // the reference recomputes the cost on every exit path (tabusearch.c:168-172)
        launch_tour_cost(t, &t->d_state[0].obj, sizeof(TourState));
        TSP_HIP_TRY(hipStreamSynchronize(s));
    }
    return status;
}

// After another engine has rewritten order[] (two_opt_lds.hip): rebuild pos[], and give BEST runs
// that were cut short their recomputed cost (tabusearch.c:168-172).
int tsp_grid_after_external_run(tsp_dev_tours *t, int mode, int timed_out) {
    hipStream_t s = t->inst->ctx->stream;
    hipLaunchKernelGGL(k_build_pos, dim3((t->n + 255) / 256, t->B), dim3(256), 0, s, t->d_order, t->d_pos, t->n);
    if (timed_out && mode == TSP_2OPT_BEST) launch_tour_cost(t, &t->d_state[0].obj, sizeof(TourState));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

tsp_dev_tours *tsp_scratch_tours(tsp_dev_inst *inst, int B, bool *owned, int *rc);   // api.hip

extern "C" {

int tsp_dev_tours_create(tsp_dev_inst *inst, int B, tsp_dev_tours **out) {
    if (!inst || !out || B < 1) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    tsp_dev_tours *t = new tsp_dev_tours();
    t->inst = inst; t->B = B; t->n = inst->n;
    // FIRST-mode chunk geometry.  A step costs a launch (~10 us of latency) plus the evaluation of
    // rows x n pairs per tour; only the pairs up to the first improving one are useful.  One tour:
    // latency dominates, 32 rows is the measured optimum at n = 10000.  Many tours: keep the smallest
    // chunk near 2M pairs per launch so that dense-improvement phases do not pay for rows they discard.
    const long long pairs_budget = 2000000;
    int auto_min = (int)std::min<long long>(32, std::max<long long>(4, pairs_budget / ((long long)B * inst->n)));
    auto_min = auto_min >= 32 ? 32 : (auto_min >= 16 ? 16 : (auto_min >= 8 ? 8 : 4));
    t->first_min_rows = std::max(1, env_int("TSP_FIRST_MIN_ROWS", auto_min));
    t->first_rows_per_block =
        std::min(kMaxRowsPerBlock, std::max(1, env_int("TSP_FIRST_ROWS_PER_BLOCK", std::min(8, t->first_min_rows))));
    t->first_max_rows = std::max(t->first_min_rows, env_int("TSP_FIRST_MAX_ROWS", 2048));
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
        t->first_rj = env_int("TSP_FIRST_RJ", 2) == 2 ? 2 : 1;
        const long long gx = (inst->n + kScanThreads * t->first_rj - 1) / (kScanThreads * t->first_rj);
        // every working block takes a ticket on one word (~12 ns each): few, fat blocks
        const int gy = (int)std::max<long long>(1, std::min<long long>(env_int("TSP_FIRST_GRID_ROWS", 8), 16384 / (gx * B)));
        t->first_grid_rows = std::min(64, gy);
        t->first_max_rows2 = std::max(t->first_min_rows, std::min(env_int("TSP_FIRST_MAX_ROWS", 2048), t->first_grid_rows * kMaxRowsPerBlock));
        t->first_v1 = env_int("TSP_FIRST_V1", 0);
    }
    t->best_rows_per_block = std::min(kMaxRowsPerBlock, std::max(1, env_int("TSP_BEST_ROWS_PER_BLOCK", 32)));
    t->count_evals = env_int("TSP_COUNT_EVALS", 1);
    t->use_graph = env_int("TSP_USE_GRAPH", 0);
    const size_t bn = (size_t)B * inst->n;
    const dim3 gb = scan_grid<TSP_2OPT_BEST>(t), gf = scan_grid<TSP_2OPT_FIRST>(t);
    t->partial_per_tour = std::max((size_t)gb.x * gb.y, (size_t)gf.x * gf.y);
    t->partial_per_tour = std::max(t->partial_per_tour, (size_t)((inst->n + kScanThreads - 1) / kScanThreads) * t->first_grid_rows);
    t->sorted_min_n = env_int("TSP_SORTED_MIN_N", 4096);
    size_t rec_per_tour = (size_t)inst->n;
    size_t cl_words = (size_t)B * 64 * 64;   // k_first: one arrival counter per tour x tile row, 64 ints apart
    if (inst->d_sperm) {
        // k_sweep blocks per tour: whole clusters, about two waves per SIMD on the chip for one tour
        const int want = env_int("TSP_SWEEP_BLOCKS", std::min(768, std::max(16, 768 / B)));
        t->sweep_blocks = std::max(1, want / kSweepCluster) * kSweepCluster;
        rec_per_tour = std::max(rec_per_tour, (size_t)inst->n_slots);
        t->partial_per_tour = std::max(t->partial_per_tour, (size_t)t->sweep_blocks);
        cl_words = std::max(cl_words, (size_t)B * (t->sweep_blocks / kSweepCluster) * 64);
        const long long npairs = (long long)inst->ng * (inst->ng + 1) / 2;
        if (npairs <= (1ll << 24) && inst->ng < 65535 && env_int("TSP_SWEEP_TABLE", 1)) {
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
        TSP_HIP_TRY(hipMalloc(&t->d_gbest, (size_t)B * sizeof(unsigned long long)));

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
    t->use_recs = env_int("TSP_BEST_RECS", 1);
    TSP_HIP_TRY(hipMalloc(&t->d_rec, (size_t)B * rec_per_tour * sizeof(NodeRec)));
    t->max_tile_rows = std::max((int)gb.y, (int)gf.y);
    TSP_HIP_TRY(hipMalloc(&t->d_row_ticket, (size_t)B * t->max_tile_rows * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_row_evals, (size_t)B * t->max_tile_rows * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_row_slot, (size_t)B * t->max_tile_rows * sizeof(Partial)));
    TSP_HIP_TRY(hipHostMalloc(&t->h_state, (size_t)B * sizeof(TourState)));
    *out = t;
    return TSP_OK;
}

void tsp_dev_tours_destroy(tsp_dev_tours *t) {
    if (!t) return;
    (void)hipSetDevice(t->inst->ctx->device);
    (void)hipStreamSynchronize(t->inst->ctx->stream);
    (void)hipFree(t->d_order); (void)hipFree(t->d_order0); (void)hipFree(t->d_pos);
    (void)hipFree(t->d_state_base); (void)hipFree(t->d_partial); (void)hipFree(t->d_slot_evals); (void)hipFree(t->d_ticket); (void)hipFree(t->d_rec);
    (void)hipFree(t->d_gmax); (void)hipFree(t->d_gbest); (void)hipFree(t->d_order2); (void)hipFree(t->d_pos2); (void)hipFree(t->d_pairtab); (void)hipFree(t->d_cl_ticket);
    (void)hipFree(t->d_row_ticket); (void)hipFree(t->d_row_evals); (void)hipFree(t->d_row_slot);
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
    TSP_HIP_TRY(hipMemcpyAsync(t->d_state, t->h_state, sizeof(TourState) * (size_t)B, hipMemcpyHostToDevice, s));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

int tsp_dev_tours_upload(tsp_dev_tours *t, const int *succ, int succ_stride, int64_t tour_stride, const double *obj) {
    if (!t || !succ || succ_stride < 1) return TSP_DEV_E_ARG;
    const int n = t->n, B = t->B;
    hipStream_t s = t->inst->ctx->stream;
    std::vector<int> order((size_t)B * n);
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
    TSP_HIP_TRY(hipStreamSynchronize(s));
    return tsp_dev_tours_reset(t);
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
    if (sorted_sweep(t)) { launch_flush(t); TSP_HIP_TRY(hipStreamSynchronize(s)); }
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

int tsp_dev_tours_best(tsp_dev_tours *t, int true_cost, int64_t *packed) {
    if (!t || !packed) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    hipStream_t s = t->inst->ctx->stream;
    std::vector<double> cost((size_t)t->B);
    if (true_cost) {
        double *d_c = nullptr;
        TSP_HIP_TRY(hipMalloc(&d_c, sizeof(double) * (size_t)t->B));
        launch_tour_cost(t, d_c, sizeof(double));
        TSP_HIP_TRY(hipMemcpyAsync(cost.data(), d_c, sizeof(double) * (size_t)t->B, hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
        (void)hipFree(d_c);
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
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    tsp_dev_tabu *tb = new tsp_dev_tabu();
    tb->inst = inst;
    tb->count = (long long)inst->n * (inst->n - 1) / 2;
    TSP_HIP_TRY(hipMalloc(&tb->d_stamp, sizeof(int) * (size_t)tb->count));
    TSP_HIP_TRY(hipMemsetAsync(tb->d_stamp, 0, sizeof(int) * (size_t)tb->count, inst->ctx->stream));
    TSP_HIP_TRY(hipStreamSynchronize(inst->ctx->stream));
    *out = tb;
    return TSP_OK;
}

void tsp_dev_tabu_destroy(tsp_dev_tabu *tb) {
    if (!tb) return;
    (void)hipSetDevice(tb->inst->ctx->device);
    (void)hipStreamSynchronize(tb->inst->ctx->stream);
    (void)hipFree(tb->d_stamp);
    delete tb;
}

static int stamp_io(tsp_dev_tabu *tb, const int *idx, int *val, int count, bool scatter) {
    if (!tb || !idx || !val || count < 0) return TSP_DEV_E_ARG;
    if (count == 0) return TSP_OK;
    for (int k = 0; k < count; ++k)
        if (idx[k] < 0 || idx[k] >= tb->count) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(tb->inst->ctx->device));
    hipStream_t s = tb->inst->ctx->stream;
    int *d_idx = nullptr, *d_val = nullptr;
    TSP_HIP_TRY(hipMalloc(&d_idx, sizeof(int) * (size_t)count));
    TSP_HIP_TRY(hipMalloc(&d_val, sizeof(int) * (size_t)count));
    TSP_HIP_TRY(hipMemcpyAsync(d_idx, idx, sizeof(int) * (size_t)count, hipMemcpyHostToDevice, s));
    if (scatter) {
        TSP_HIP_TRY(hipMemcpyAsync(d_val, val, sizeof(int) * (size_t)count, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_stamp_scatter, dim3((count + 255) / 256), dim3(256), 0, s, tb->d_stamp, d_idx, d_val, count);
    } else {
        hipLaunchKernelGGL(k_stamp_gather, dim3((count + 255) / 256), dim3(256), 0, s, tb->d_stamp, d_idx, d_val, count);
        TSP_HIP_TRY(hipMemcpyAsync(val, d_val, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost, s));
    }
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    (void)hipFree(d_idx); (void)hipFree(d_val);
    return TSP_OK;
}

int tsp_dev_tabu_set(tsp_dev_tabu *tb, const int *idx, const int *value, int count) {
    return stamp_io(tb, idx, const_cast<int *>(value), count, true);
}
int tsp_dev_tabu_get(tsp_dev_tabu *tb, const int *idx, int *value, int count) {
    return stamp_io(tb, idx, value, count, false);
}
int tsp_dev_tabu_upload(tsp_dev_tabu *tb, const int *stamps) {
    if (!tb || !stamps) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(tb->inst->ctx->device));
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
    const int status = tsp_grid_run(t, TSP_2OPT_BEST, tabu, iter, tenure, -1, time_limit_s, 1, &done);
    if (status < 0) return status;
    rc = tsp_dev_tours_download(t, succ, succ_stride, inst->n, obj, stats);
    if (rc) return rc;
    if (stored_prev)  // tabusearch.c:173-175
        for (int v = 0; v < inst->n; ++v) stored_prev[succ[(size_t)v * succ_stride]] = v;
    if (stats) stats->seconds = wall_s() - t0;
    return status;
}

#ifdef TSP_STAMPS
int tsp_dev_debug_sweep2(unsigned long long *out4096) {
    if (hipMemcpyFromSymbol(out4096, HIP_SYMBOL(tsp::g_blk2), 4096 * sizeof(unsigned long long)) != hipSuccess) return -1;
    static unsigned long long z[4096];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_blk2), z, sizeof z);
    return 0;
}
int tsp_dev_debug_sweep(unsigned long long *out8192) {
    if (hipMemcpyFromSymbol(out8192, HIP_SYMBOL(tsp::g_blk), 8192 * sizeof(unsigned long long)) != hipSuccess) return -1;
    static unsigned long long z[8192];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_blk), z, sizeof z);
    return 0;
}
// diagnostic: mean 100 MHz ticks per segment of the last block of a step; resets the sums
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
