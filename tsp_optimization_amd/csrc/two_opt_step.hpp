// two_opt_step.hpp -- what every step kernel of the GRID engine shares: launch arguments, the in-launch hand-off,
// the tour seen through a pending move, the tour cost, and the apply executed by a step's last block
// Part of the GRID engine; included by two_opt_grid.hip only (one translation unit).
#pragma once
#include "two_opt_common.hpp"

#pragma clang fp contract(off)

namespace tsp {


constexpr int kMaxRowsPerBlock = 256;

// Diagnostic build only (-DTSP_STAMPS): 100 MHz wall-clock stamps of the last block of each step,
// accumulated into a buffer nothing else reads (cdna_hip_programming.md section 7, in-kernel stamps).
#ifdef TSP_STAMPS
__device__ unsigned long long g_stamp_sum[16];
__device__ unsigned long long g_stamp_n;
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
    int ng, n_slots, flat_slots;
    // k_sweep<TABU> (two_opt_tabu_list.hpp): the compact list of non-zero stamps and the sweep's skipped-pair counter
    const int2 *tabu_list;
    const int *tabu_list_n;
    int tabu_list_cap;
    unsigned long long *tabu_pairs;
};

template <int WT, bool INT, int MODE, int RJ, bool TABU, bool FLAT = false>
__device__ __forceinline__ void apply_step(const StepArgs &a, int tour, int row_lo, int row_hi
#ifdef TSP_STAMPS
                                           , unsigned long long *stamps
#endif
                                           , double *chunk_buf = nullptr   // FLAT: 4096 doubles of LDS the caller no longer needs
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
            // the sequential cost of non-integer lengths is staged through 32 KB of LDS: the sweep lends its staging
            // area (its own 32 KB would cost the float-cost variants two thirds of their resident blocks)
            __shared__ double s_chunk[(INT || WT == WT_CEIL_2D || FLAT) ? 1 : 4096];
            final_cost = tour_cost_block<WT, INT>(a.coord, order, pos, n, s_d, FLAT ? chunk_buf : s_chunk);
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

}  // namespace tsp
