// two_opt_grid.hip -- GRID engine: 2-opt on tours resident in HBM, many workgroups per tour.
//
// One step = k_scan (evaluate a range of (i<j) node pairs against the current tour, one
// candidate per block) + k_apply (one workgroup per tour: pick the winner, reverse the tour
// segment, refresh the touched node records, advance the tour's control block).  The host only
// queues steps and polls `done`; every decision of the reference's loops is taken on the device:
//   FIRST  = alg_2opt       (src/heuristics.c:438-502): first improving pair after the cursor in
//            (i<j) order, applied at once, scan resumes right after it; stop after a sweep that
//            did not lower obj_best (:492).
//   BEST   = alg_2opt_tabu  (src/tabusearch.c:107-178): arg-min delta over the whole sweep, strict
//            '<' so ties go to the first pair (:151); stop when the minimum is >= 0 (:158); cost
//            recomputed as a sum over edges in node order (:168-172).
//
// The pair (i,j) always denotes removing (i,succ i) and (j,succ j) and reversing the FORWARD path
// succ(i)..j (src/utility.c:708-717).  Tours are kept as order[]/pos[] arrays; the reversed path
// is the cyclic position range pos[i]+1 .. pos[j], so orientation is preserved by construction.
#include "tsp_internal.hpp"

#pragma clang fp contract(off)

namespace tsp {

using u64 = unsigned long long;
constexpr u64 kNoKey = ~0ull;

__device__ __forceinline__ u64 make_key(int i, int j) {
    return i < 0 ? kNoKey : (((u64)(unsigned)i << 32) | (u64)(unsigned)j);
}

// (delta, key) lexicographic minimum == "first pair in scan order among the minimal deltas"
__device__ __forceinline__ bool better(double d1, u64 k1, double d2, u64 k2) {
    return d1 < d2 || (d1 == d2 && k1 < k2);
}

template <bool BY_DELTA>
__device__ __forceinline__ void wave_argmin(double &d, u64 &k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double od = __shfl_xor(d, off);
        const u64 ok = __shfl_xor(k, off);
        const bool take = BY_DELTA ? better(od, ok, d, k) : (ok < k);
        if (take) { d = od; k = ok; }
    }
}

// number of pairs (r,c), r<c, that precede or equal (i,j) in lexicographic order
__device__ __forceinline__ long long pair_rank(long long i, long long j, long long n) {
    return i * (n - 1) - i * (i - 1) / 2 + (j - i);
}

// src/utility.c:17-30 for i != j
__device__ __forceinline__ long long udir_pos(int i, int j, int n) {
    if (i > j) { const int t = i; i = j; j = t; }
    return (long long)i * n + j - ((long long)(i + 1) * (i + 2)) / 2;
}

// src/tabusearch.c:83-92, including the lazy clearing write.  Within one call iter and tenure
// are constant, so concurrent clears of an expired stamp all store 0: a benign race.
__device__ __forceinline__ bool stamp_is_tabu(int *stamp, int iter, int tenure) {
    if (iter < 0 || tenure < 0) return false;
    const int v = *stamp;
    if (v == 0) return false;
    if (iter - v > tenure) { *stamp = 0; return false; }
    return true;
}

// ---- build pos[] and rec[] from order[] ---------------------------------------------------
template <int WT, bool INT>
__global__ void k_build(const double2 *__restrict__ coord, const int *__restrict__ orders,
                        int *__restrict__ poss, Rec *__restrict__ recs, int n) {
    const int tour = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int *order = orders + (size_t)tour * n;
    const int v = order[p];
    const int w = order[p + 1 == n ? 0 : p + 1];
    const double2 cv = coord[v], cw = coord[w];
    poss[(size_t)tour * n + v] = p;
    Rec r;
    r.x = cv.x; r.y = cv.y; r.xs = cw.x; r.ys = cw.y;
    r.ds = dist_xy<WT, INT>(cv.x, cv.y, cw.x, cw.y);
    r.succ = w; r.pad = 0;
    recs[(size_t)tour * n + v] = r;
}

// ---- scan ---------------------------------------------------------------------------------
// Block (bx, by, tour): rows r0 .. r0+rows_per_block of the tour's active row range, columns
// bx*256*RJ .. +256*RJ.  Lanes own columns (coalesced 48-byte record loads, held in registers
// for all rows of the block); the row record is wave-uniform (scalar loads).
template <int WT, bool INT, int MODE, int RJ, bool TABU>
__global__ __launch_bounds__(kScanThreads) void k_scan(const Rec *__restrict__ recs,
                                                       const TourState *__restrict__ states,
                                                       Partial *__restrict__ partials, int n,
                                                       int rows_per_block, size_t partial_per_tour,
                                                       int *__restrict__ tabu, int iter, int tenure,
                                                       int *__restrict__ slot_evals) {
    const int tour = blockIdx.z;
    const TourState *st = states + tour;
    if (st->done) return;
    int row_lo = 0, row_hi = n - 1, ci = -1, cj = -1;
    if constexpr (MODE == TSP_2OPT_FIRST) {
        ci = st->ci; cj = st->cj;
        row_lo = ci;
        row_hi = min(ci + st->chunk_rows, n - 1);
    }
    const int r0 = row_lo + blockIdx.y * rows_per_block;
    if (r0 >= row_hi) return;  // beyond the active chunk: no slot is read for this block
    const int r1 = min(r0 + rows_per_block, row_hi);
    const int c0 = blockIdx.x * (kScanThreads * RJ);
    Partial *slot = partials + (size_t)tour * partial_per_tour + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int tid = threadIdx.x;
    if (c0 + kScanThreads * RJ - 1 <= r0) {  // every column <= every row: nothing with j > i
        if (tid == 0) {
            Partial p; p.delta = 0.0; p.i = -1; p.j = -1; *slot = p;
            if constexpr (TABU) slot_evals[slot - partials] = 0;
        }
        return;
    }
    const Rec *rec = recs + (size_t)tour * n;
    int n_eval = 0;  // TABU: pairs that reach the delta expression (tabusearch.c:150)

    int jc[RJ];
    Rec rj[RJ];
#pragma unroll
    for (int k = 0; k < RJ; ++k) {
        jc[k] = c0 + tid + k * kScanThreads;
        rj[k] = rec[min(jc[k], n - 1)];
        if (jc[k] >= n) jc[k] = -1;  // never > i
    }

    double bd = 0.0;
    int bi = -1, bj = -1;
    for (int i = r0; i < r1; ++i) {
        const Rec ri = rec[i];
#pragma unroll
        for (int k = 0; k < RJ; ++k) {
            const int j = jc[k];
            bool ok = j > i && j != ri.succ && rj[k].succ != i;  // heuristics.c:471 / tabusearch.c:134
            if constexpr (MODE == TSP_2OPT_FIRST) ok = ok && (i > ci || j > cj) && bi < 0;
            if constexpr (TABU) {
                if (ok) {
                    const int a1 = ri.succ, b1 = rj[k].succ;
                    if (stamp_is_tabu(tabu + udir_pos(i, j, n), iter, tenure) ||
                        stamp_is_tabu(tabu + udir_pos(i, a1, n), iter, tenure) ||
                        stamp_is_tabu(tabu + udir_pos(j, b1, n), iter, tenure) ||
                        stamp_is_tabu(tabu + udir_pos(i, b1, n), iter, tenure))
                        ok = false;  // tabusearch.c:137-149
                }
                n_eval += ok ? 1 : 0;
            }
            // heuristics.c:474 / tabusearch.c:150, same association: ((d(a,b)+d(a1,b1))-d(a,a1))-d(b,b1)
            const double delta = dist_xy<WT, INT>(ri.x, ri.y, rj[k].x, rj[k].y) +
                                 dist_xy<WT, INT>(ri.xs, ri.ys, rj[k].xs, rj[k].ys) - ri.ds - rj[k].ds;
            if constexpr (MODE == TSP_2OPT_FIRST) {
                if (ok && delta < 0) { bd = delta; bi = i; bj = j; }
            } else {
                if (ok && delta < bd) { bd = delta; bi = i; bj = j; }
            }
        }
        if constexpr (MODE == TSP_2OPT_FIRST) {
            if (__any(bi >= 0)) break;  // later rows only hold later pairs
        }
    }

    u64 key = make_key(bi, bj);
    wave_argmin<MODE == TSP_2OPT_BEST>(bd, key);
    __shared__ double s_d[kScanThreads / 64];
    __shared__ u64 s_k[kScanThreads / 64];
    if ((tid & 63) == 0) { s_d[tid >> 6] = bd; s_k[tid >> 6] = key; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < kScanThreads / 64; ++w) {
            const bool take = (MODE == TSP_2OPT_BEST) ? better(s_d[w], s_k[w], bd, key) : (s_k[w] < key);
            if (take) { bd = s_d[w]; key = s_k[w]; }
        }
        Partial p;
        p.delta = bd;
        p.i = key == kNoKey ? -1 : (int)(key >> 32);
        p.j = key == kNoKey ? -1 : (int)(key & 0xffffffffu);
        *slot = p;
    }
    if constexpr (TABU) {
        __shared__ int s_cnt;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) n_eval += __shfl_xor(n_eval, off);
        if ((tid & 63) == 0) atomicAdd(&s_cnt, n_eval);
        __syncthreads();
        if (tid == 0) slot_evals[slot - partials] = s_cnt;
    }
}

// ---- apply --------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T block_sum(T v, T *scratch /* >= 16 */) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    const int tid = threadIdx.x;
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    T tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += scratch[w];
    return tot;
}

// Sum over nodes of d(v, succ v) in node order (tabusearch.c:168-172), by one whole block.
template <bool INT>
__device__ __forceinline__ double tour_cost_block(const Rec *rec, int n, double *s_d /*>=16*/, double *s_chunk /*4096*/) {
    const int tid = threadIdx.x;
    if constexpr (INT) {  // integer-valued terms: any order is exact
        double c = 0.0;
        for (int v = tid; v < n; v += (int)blockDim.x) c += rec[v].ds;
        return block_sum<double>(c, s_d);
    } else {              // same sequential order as the reference, staged through LDS
        double acc = 0.0;
        for (int base = 0; base < n; base += 4096) {
            __syncthreads();
            for (int t = tid; t < 4096 && base + t < n; t += (int)blockDim.x) s_chunk[t] = rec[base + t].ds;
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

// obj = recomputed cost, for BEST runs that stop before the local optimum (time limit)
template <bool INT>
__global__ __launch_bounds__(kApplyThreads) void k_tour_cost(const Rec *__restrict__ recs, TourState *__restrict__ states, int n) {
    __shared__ double s_d[kApplyThreads / 64];
    __shared__ double s_chunk[INT ? 1 : 4096];
    const double c = tour_cost_block<INT>(recs + (size_t)blockIdx.x * n, n, s_d, s_chunk);
    if (threadIdx.x == 0) states[blockIdx.x].obj = c;
}

template <int WT, bool INT, int MODE>
__global__ __launch_bounds__(kApplyThreads) void k_apply(Rec *__restrict__ recs, int *__restrict__ orders,
                                                         int *__restrict__ poss, TourState *__restrict__ states,
                                                         const Partial *__restrict__ partials, int n,
                                                         int rows_per_block, int scan_gx, int scan_gy,
                                                         size_t partial_per_tour, int first_min_rows,
                                                         int first_max_rows, int count_evals,
                                                         const int *__restrict__ slot_evals) {
    const int tour = blockIdx.x;
    TourState *st = states + tour;
    if (st->done) return;
    const int tid = threadIdx.x;
    Rec *rec = recs + (size_t)tour * n;
    int *order = orders + (size_t)tour * n;
    int *pos = poss + (size_t)tour * n;
    const Partial *part = partials + (size_t)tour * partial_per_tour;

    __shared__ double s_d[kApplyThreads / 64];
    __shared__ u64 s_k[kApplyThreads / 64];
    __shared__ long long s_ll[kApplyThreads / 64];
    __shared__ double s_delta;
    __shared__ int s_i, s_j, s_pa, s_pb;

    int row_lo = 0, row_hi = n - 1, ci = 0, cj = 0;
    if constexpr (MODE == TSP_2OPT_FIRST) {
        ci = st->ci; cj = st->cj;
        row_lo = ci;
        row_hi = min(ci + st->chunk_rows, n - 1);
    }
    int tile_rows = (row_hi - row_lo + rows_per_block - 1) / rows_per_block;
    tile_rows = min(tile_rows, scan_gy);
    const int nslots = tile_rows * scan_gx;

    // 1. winner over the scan blocks
    double bd = 0.0;
    u64 key = kNoKey;
    for (int s = tid; s < nslots; s += kApplyThreads) {
        const Partial p = part[s];
        const u64 k = make_key(p.i, p.j);
        const bool take = (MODE == TSP_2OPT_BEST) ? better(p.delta, k, bd, key) : (k < key);
        if (take) { bd = p.delta; key = k; }
    }
    wave_argmin<MODE == TSP_2OPT_BEST>(bd, key);
    if ((tid & 63) == 0) { s_d[tid >> 6] = bd; s_k[tid >> 6] = key; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kApplyThreads / 64; ++w) {
            const bool take = (MODE == TSP_2OPT_BEST) ? better(s_d[w], s_k[w], bd, key) : (s_k[w] < key);
            if (take) { bd = s_d[w]; key = s_k[w]; }
        }
        int wi = -1, wj = -1;
        if (key != kNoKey && (MODE == TSP_2OPT_FIRST || bd < 0)) {
            wi = (int)(key >> 32); wj = (int)(key & 0xffffffffu);
        }
        s_i = wi; s_j = wj;
        if (wi >= 0) {
            s_pa = pos[wi]; s_pb = pos[wj];
            const Rec ra = rec[wi], rb = rec[wj];
            s_delta = dist_xy<WT, INT>(ra.x, ra.y, rb.x, rb.y) + dist_xy<WT, INT>(ra.xs, ra.ys, rb.xs, rb.ys) -
                      ra.ds - rb.ds;
        }
    }
    __syncthreads();
    const int wi = s_i, wj = s_j;
    const bool found = wi >= 0;

    // 2. FIRST: how many of the pairs between the old and the new cursor the reference would have
    //    skipped as adjacent (counted on the tour the scan saw, i.e. before the move)
    long long adj = 0;
    int ni = wi, nj = wj;  // new cursor
    if constexpr (MODE == TSP_2OPT_FIRST) {
        if (!found) { ni = row_hi - 1; nj = n - 1; }
        if (count_evals) {
            const u64 lo = ((u64)(unsigned)ci << 32) | (unsigned)cj, hi = ((u64)(unsigned)ni << 32) | (unsigned)nj;
            long long c = 0;
            for (int v = tid; v < n; v += kApplyThreads) {
                const int s = rec[v].succ;
                const u64 k = ((u64)(unsigned)min(v, s) << 32) | (unsigned)max(v, s);
                c += (k > lo && k <= hi) ? 1 : 0;
            }
            adj = block_sum<long long>(c, s_ll);
        }
    }

    // 3. the move: reverse positions pa+1 .. pb (cyclic), then refresh the records of pa .. pb
    int L = 0;
    if (found) {
        const int pa = s_pa, pb = s_pb;
        L = pb - pa; if (L < 0) L += n;
        const int half = L >> 1;
        for (int t = tid; t < half; t += kApplyThreads) {
            int p = pa + 1 + t; if (p >= n) p -= n;
            int q = pb - t; if (q < 0) q += n;
            const int u = order[p], w = order[q];
            order[p] = w; order[q] = u;
            pos[w] = p; pos[u] = q;
        }
        __syncthreads();
        for (int t = tid; t <= L; t += kApplyThreads) {
            int p = pa + t; if (p >= n) p -= n;
            const int q = p + 1 == n ? 0 : p + 1;
            const int v = order[p], w = order[q];
            const double vx = rec[v].x, vy = rec[v].y, wx = rec[w].x, wy = rec[w].y;
            rec[v].xs = wx; rec[v].ys = wy;
            rec[v].ds = dist_xy<WT, INT>(vx, vy, wx, wy);
            rec[v].succ = w;
        }
    }

    // 4. BEST at the local optimum: cost = sum over nodes of d(v, succ v), node order (tabusearch.c:168-172)
    double final_cost = 0.0;
    long long tabu_evals = -1;
    if constexpr (MODE == TSP_2OPT_BEST) {
        if (slot_evals) {  // tabu list active: the scan blocks counted what reached the delta expression
            long long c = 0;
            for (int s = tid; s < nslots; s += kApplyThreads) c += slot_evals[(size_t)tour * partial_per_tour + s];
            tabu_evals = block_sum<long long>(c, s_ll);
        }
        if (!found) {
            __shared__ double s_chunk[INT ? 1 : 4096];
            final_cost = tour_cost_block<INT>(rec, n, s_d, s_chunk);
        }
    }

    // 5. control block
    if (tid == 0) {
        st->steps += 1;
        if constexpr (MODE == TSP_2OPT_BEST) {
            st->sweeps += 1;
            st->evals += tabu_evals >= 0 ? tabu_evals : (long long)n * (n - 1) / 2 - n;  // every non-adjacent pair (n >= 4)
            st->pairs_scanned += (long long)n * (n - 1) / 2;
            if (found) { st->moves += 1; st->reversed += L - 1; }
            else { st->done = 1; st->obj = final_cost; }
        } else {
            const long long r_old = pair_rank(ci, cj, n);
            const long long r_end = pair_rank(row_hi - 1, n - 1, n);
            st->pairs_scanned += r_end - r_old;
            st->evals += pair_rank(ni, nj, n) - r_old - adj;
            if (found) {
                st->obj += s_delta;                    // heuristics.c:486
                st->moves += 1;
                st->reversed += L - 1;         // successors rewritten by reverse_path's walk (utility.c:710-717)
                st->ci = wi; st->cj = wj;
                st->chunk_rows = first_min_rows;
            } else {
                st->chunk_rows = min(st->chunk_rows * 2, first_max_rows);
                if (row_hi >= n - 1) {                 // sweep complete
                    st->sweeps += 1;
                    if (st->obj >= st->seen_cost) st->done = 1;   // heuristics.c:492
                    else { st->seen_cost = st->obj; st->ci = 0; st->cj = 0; }
                } else {
                    st->ci = row_hi - 1; st->cj = n - 1;
                }
            }
        }
    }
}

// packed (cost, tour) minimum over the tours of one handle
template <int WT, bool INT>
__global__ __launch_bounds__(kApplyThreads) void k_best_tour(const Rec *__restrict__ recs,
                                                             const TourState *__restrict__ states, int n, int B,
                                                             int true_cost, long long *__restrict__ out) {
    __shared__ double s_d[kApplyThreads / 64];
    long long best = 0x7fffffffffffffffLL;
    for (int t = 0; t < B; ++t) {
        double c;
        if (true_cost) {
            double acc = 0.0;
            for (int v = threadIdx.x; v < n; v += kApplyThreads) acc += recs[(size_t)t * n + v].ds;
            c = block_sum<double>(acc, s_d);
        } else {
            c = states[t].obj;
        }
        const long long packed = ((long long)c << 24) | (long long)t;
        best = packed < best ? packed : best;
    }
    if (threadIdx.x == 0) *out = best;
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

template <int MODE>
dim3 scan_grid(const tsp_dev_tours *t, int RJ) {
    const int n = t->n;
    const int gx = (n + kScanThreads * RJ - 1) / (kScanThreads * RJ);
    int gy;
    if (MODE == TSP_2OPT_BEST) gy = (n - 1 + t->best_rows_per_block - 1) / t->best_rows_per_block;
    else gy = (std::min(t->first_max_rows, n - 1) + t->first_rows_per_block - 1) / t->first_rows_per_block;
    return dim3(gx, gy, t->B);
}

constexpr int kBestRJ = 2;
constexpr int kFirstRJ = 1;

template <int WT, bool INT>
int launch_step(tsp_dev_tours *t, int mode, tsp_dev_tabu *tabu, int iter, int tenure, int count_evals) {
    hipStream_t s = t->inst->ctx->stream;
    const int n = t->n;
    if (mode == TSP_2OPT_BEST) {
        const dim3 g = scan_grid<TSP_2OPT_BEST>(t, kBestRJ);
        if (tabu)
            hipLaunchKernelGGL((k_scan<WT, INT, TSP_2OPT_BEST, kBestRJ, true>), g, dim3(kScanThreads), 0, s, t->d_rec,
                               t->d_state, t->d_partial, n, t->best_rows_per_block, t->partial_per_tour,
                               tabu->d_stamp, iter, tenure, t->d_slot_evals);
        else
            hipLaunchKernelGGL((k_scan<WT, INT, TSP_2OPT_BEST, kBestRJ, false>), g, dim3(kScanThreads), 0, s,
                               t->d_rec, t->d_state, t->d_partial, n, t->best_rows_per_block, t->partial_per_tour,
                               (int *)nullptr, 0, 0, (int *)nullptr);
        hipLaunchKernelGGL((k_apply<WT, INT, TSP_2OPT_BEST>), dim3(t->B), dim3(kApplyThreads), 0, s, t->d_rec,
                           t->d_order, t->d_pos, t->d_state, t->d_partial, n, t->best_rows_per_block, (int)g.x,
                           (int)g.y, t->partial_per_tour, 0, 0, count_evals,
                           tabu ? (const int *)t->d_slot_evals : (const int *)nullptr);
    } else {
        const dim3 g = scan_grid<TSP_2OPT_FIRST>(t, kFirstRJ);
        const int rmin = std::min(t->first_rows_per_block * 2, std::max(1, n - 1));
        const int rmax = (int)g.y * t->first_rows_per_block;
        hipLaunchKernelGGL((k_scan<WT, INT, TSP_2OPT_FIRST, kFirstRJ, false>), g, dim3(kScanThreads), 0, s, t->d_rec,
                           t->d_state, t->d_partial, n, t->first_rows_per_block, t->partial_per_tour, (int *)nullptr,
                           0, 0, (int *)nullptr);
        hipLaunchKernelGGL((k_apply<WT, INT, TSP_2OPT_FIRST>), dim3(t->B), dim3(kApplyThreads), 0, s, t->d_rec,
                           t->d_order, t->d_pos, t->d_state, t->d_partial, n, t->first_rows_per_block, (int)g.x,
                           (int)g.y, t->partial_per_tour, rmin, rmax, count_evals, (const int *)nullptr);
    }
    return TSP_OK;
}

int launch_step_rt(tsp_dev_tours *t, int mode, tsp_dev_tabu *tabu, int iter, int tenure, int count_evals) {
    int rc = TSP_OK;
    TSP_DISPATCH_METRIC(t->inst->wtype, t->inst->integer_cost,
                        { rc = launch_step<WTC, INTC>(t, mode, tabu, iter, tenure, count_evals); });
    return rc;
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
    hipStream_t s = t->inst->ctx->stream;
    const double t0 = wall_s();
    const int64_t batch = 64;
    int64_t queued = 0;
    int status = TSP_OK;
    if (all_done) *all_done = 0;
    for (;;) {
        int64_t todo = batch;
        if (max_steps >= 0) todo = std::min<int64_t>(batch, max_steps - queued);
        if (todo <= 0) break;
        for (int64_t k = 0; k < todo; ++k) {
            int rc = launch_step_rt(t, mode, tabu, iter, tenure, 1);
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
    if (status == TSP_TIME_LIMIT_EXCEEDED && mode == TSP_2OPT_BEST) {
        // the reference recomputes the cost on every exit path (tabusearch.c:168-172)
        if (t->inst->integer_cost)
            hipLaunchKernelGGL((k_tour_cost<true>), dim3(t->B), dim3(kApplyThreads), 0, s, t->d_rec, t->d_state, t->n);
        else
            hipLaunchKernelGGL((k_tour_cost<false>), dim3(t->B), dim3(kApplyThreads), 0, s, t->d_rec, t->d_state, t->n);
        TSP_HIP_TRY(hipStreamSynchronize(s));
    }
    return status;
}

extern "C" {

int tsp_dev_tours_create(tsp_dev_inst *inst, int B, tsp_dev_tours **out) {
    if (!inst || !out || B < 1) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    tsp_dev_tours *t = new tsp_dev_tours();
    t->inst = inst; t->B = B; t->n = inst->n;
    const size_t bn = (size_t)B * inst->n;
    const dim3 gb = scan_grid<TSP_2OPT_BEST>(t, kBestRJ), gf = scan_grid<TSP_2OPT_FIRST>(t, kFirstRJ);
    t->partial_per_tour = std::max((size_t)gb.x * gb.y, (size_t)gf.x * gf.y);
    TSP_HIP_TRY(hipMalloc(&t->d_order, bn * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_order0, bn * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_pos, bn * sizeof(int)));
    TSP_HIP_TRY(hipMalloc(&t->d_rec, bn * sizeof(Rec)));
    TSP_HIP_TRY(hipMalloc(&t->d_state, (size_t)B * sizeof(TourState)));
    TSP_HIP_TRY(hipMalloc(&t->d_partial, (size_t)B * t->partial_per_tour * sizeof(Partial)));
    TSP_HIP_TRY(hipMalloc(&t->d_slot_evals, (size_t)B * t->partial_per_tour * sizeof(int)));
    TSP_HIP_TRY(hipHostMalloc(&t->h_state, (size_t)B * sizeof(TourState)));
    *out = t;
    return TSP_OK;
}

void tsp_dev_tours_destroy(tsp_dev_tours *t) {
    if (!t) return;
    (void)hipSetDevice(t->inst->ctx->device);
    (void)hipStreamSynchronize(t->inst->ctx->stream);
    (void)hipFree(t->d_order); (void)hipFree(t->d_order0); (void)hipFree(t->d_pos); (void)hipFree(t->d_rec);
    (void)hipFree(t->d_state); (void)hipFree(t->d_partial); (void)hipFree(t->d_slot_evals);
    (void)hipHostFree(t->h_state);
    delete t;
}

int tsp_dev_tours_reset(tsp_dev_tours *t) {
    if (!t) return TSP_DEV_E_ARG;
    hipStream_t s = t->inst->ctx->stream;
    const int n = t->n, B = t->B;
    TSP_HIP_TRY(hipMemcpyAsync(t->d_order, t->d_order0, (size_t)B * n * sizeof(int), hipMemcpyDeviceToDevice, s));
    TSP_DISPATCH_METRIC(t->inst->wtype, t->inst->integer_cost, {
        hipLaunchKernelGGL((k_build<WTC, INTC>), dim3((n + 255) / 256, B), dim3(256), 0, s, t->inst->d_coord,
                           t->d_order, t->d_pos, t->d_rec, n);
    });
    // control blocks: cursor at sweep start, smallest chunk, obj = uploaded value
    std::vector<TourState> init((size_t)B);
    for (int b = 0; b < B; ++b) {
        TourState z;
        memset(&z, 0, sizeof z);
        z.chunk_rows = std::min(t->first_rows_per_block * 2, std::max(1, n - 1));
        z.obj = t->h_obj0[b];
        z.seen_cost = t->h_obj0[b];
        init[b] = z;
    }
    memcpy(t->h_state, init.data(), sizeof(TourState) * (size_t)B);
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
    std::vector<double> o((size_t)B, 0.0);
    if (obj) for (int b = 0; b < B; ++b) o[b] = obj[b];
    TSP_HIP_TRY(hipMemcpyAsync(t->d_order0, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice, s));
    t->h_obj0 = o;
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
    const dim3 g = scan_grid<TSP_2OPT_BEST>(t, kBestRJ);
    for (int r = 0; r < reps; ++r) {
        TSP_HIP_TRY(hipEventRecord(e0, s));
        TSP_DISPATCH_METRIC(t->inst->wtype, t->inst->integer_cost, {
            hipLaunchKernelGGL((k_scan<WTC, INTC, TSP_2OPT_BEST, kBestRJ, false>), g, dim3(kScanThreads), 0, s,
                               t->d_rec, t->d_state, t->d_partial, n, t->best_rows_per_block, t->partial_per_tour,
                               (int *)nullptr, 0, 0, (int *)nullptr);
        });
        TSP_HIP_TRY(hipEventRecord(e1, s));
        TSP_HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        TSP_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (mean_ms) *mean_ms = (float)(total / reps);
    if (evals_per_launch) *evals_per_launch = ((int64_t)n * (n - 1) / 2 - n) * t->B;
    return TSP_OK;
}

int tsp_dev_tours_best(tsp_dev_tours *t, int true_cost, int64_t *packed) {
    if (!t || !packed) return TSP_DEV_E_ARG;
    hipStream_t s = t->inst->ctx->stream;
    long long *d_out = nullptr;
    TSP_HIP_TRY(hipMalloc(&d_out, sizeof(long long)));
    TSP_DISPATCH_METRIC(t->inst->wtype, t->inst->integer_cost, {
        hipLaunchKernelGGL((k_best_tour<WTC, INTC>), dim3(1), dim3(kApplyThreads), 0, s, t->d_rec, t->d_state, t->n,
                           t->B, true_cost, d_out);
    });
    long long h = 0;
    TSP_HIP_TRY(hipMemcpyAsync(&h, d_out, sizeof h, hipMemcpyDeviceToHost, s));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    (void)hipFree(d_out);
    *packed = h;
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
    tsp_dev_tours *t = nullptr;
    int rc = tsp_dev_tours_create(inst, 1, &t);
    if (rc) return rc;
    rc = tsp_dev_tours_upload(t, succ, succ_stride, inst->n, obj);
    if (rc) { tsp_dev_tours_destroy(t); return rc; }
    int done = 0;
    const int status = tsp_grid_run(t, TSP_2OPT_BEST, tabu, iter, tenure, -1, time_limit_s, 1, &done);
    if (status < 0) { tsp_dev_tours_destroy(t); return status; }
    rc = tsp_dev_tours_download(t, succ, succ_stride, inst->n, obj, stats);
    tsp_dev_tours_destroy(t);
    if (rc) return rc;
    if (stored_prev)  // tabusearch.c:173-175
        for (int v = 0; v < inst->n; ++v) stored_prev[succ[(size_t)v * succ_stride]] = v;
    if (stats) stats->seconds = wall_s() - t0;
    return status;
}

}  // extern "C"
