// two_opt_common.hpp -- device helpers shared by the GRID and LDS 2-opt engines.
#pragma once
#include "tsp_internal.hpp"

#pragma clang fp contract(off)

namespace tsp {

using u64 = unsigned long long;
constexpr u64 kNoKey = ~0ull;

// scan-order key of a pair: lexicographic (i, j) == the reference's loop order (heuristics.c:452-454)
__device__ __forceinline__ u64 make_key(int i, int j) {
    return i < 0 ? kNoKey : (((u64)(unsigned)i << 32) | (u64)(unsigned)j);
}
__device__ __forceinline__ int key_i(u64 k) { return k == kNoKey ? -1 : (int)(k >> 32); }
__device__ __forceinline__ int key_j(u64 k) { return k == kNoKey ? -1 : (int)(k & 0xffffffffu); }

// (delta, key) lexicographic minimum == "first pair in scan order among the minimal deltas"
// (tabusearch.c:151 keeps the first pair because its '<' is strict)
__device__ __forceinline__ bool better(double d1, u64 k1, double d2, u64 k2) {
    return d1 < d2 || (d1 == d2 && k1 < k2);
}

// BY_DELTA: arg-min of (delta, key); else: min key (first improving pair), delta rides along.
template <bool BY_DELTA>
__device__ __forceinline__ void wave_argmin(double &d, u64 &k) {
    // Almost every wave of a scan has no candidate at all (one move per sweep): one ballot instead of
    // eighteen cross-lane moves.  A wave with no key keeps (d, kNoKey) in every lane, which is what the
    // reduction would have produced for the key; callers only use d together with a valid key.
    if (!__any(k != kNoKey)) return;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double od = __shfl_xor(d, off);
        const u64 ok = __shfl_xor(k, off);
        const bool take = BY_DELTA ? better(od, ok, d, k) : (ok < k);
        if (take) { d = od; k = ok; }
    }
}

// Block-wide version; every thread gets the winner.  scratch: >= blockDim/64 entries each.
template <bool BY_DELTA>
__device__ __forceinline__ void block_argmin(double &d, u64 &k, double *s_d, u64 *s_k) {
    wave_argmin<BY_DELTA>(d, k);
    const int tid = threadIdx.x;
    __syncthreads();
    if ((tid & 63) == 0) { s_d[tid >> 6] = d; s_k[tid >> 6] = k; }
    __syncthreads();
    d = s_d[0]; k = s_k[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
        const bool take = BY_DELTA ? better(s_d[w], s_k[w], d, k) : (s_k[w] < k);
        if (take) { d = s_d[w]; k = s_k[w]; }
    }
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T *scratch /* >= blockDim/64 */) {
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

// What a delta evaluation needs to know about one end of a pair: the node, its successor and
// the length of the tour edge between them (heuristics.c:466-467,474).
struct alignas(16) NodeRec {
    double x, y;    // node (lat/lon radians for GEO)
    double xs, ys;  // succ(node)
    double ds;      // calc_dist(node, succ(node))
    int succ;
    int id;         // the node itself (records stored out of node order carry it: sorted sweep)
};
static_assert(sizeof(NodeRec) == 48, "NodeRec must be 48 bytes");

// succ(v) = order[pos[v] + 1]; ORD/POS may be int or unsigned short arrays (global or LDS).
template <int WT, bool INT, typename COORD, typename ORD>
__device__ __forceinline__ NodeRec load_node(const COORD *coord, const ORD *order, const ORD *pos, int n, int v) {
    int q = (int)pos[v] + 1;
    if (q == n) q = 0;
    const int s = (int)order[q];
    const double2 c = coord[v], cs = coord[s];
    NodeRec r;
    r.x = c.x; r.y = c.y; r.xs = cs.x; r.ys = cs.y;
    r.ds = dist_xy<WT, INT>(c.x, c.y, cs.x, cs.y);
    r.succ = s; r.id = v;
    return r;
}

// heuristics.c:474 / tabusearch.c:150, same association: ((d(a,b) + d(a1,b1)) - d(a,a1)) - d(b,b1)
template <int WT, bool INT>
__device__ __forceinline__ double pair_delta(const NodeRec &a, const NodeRec &b) {
    return dist_xy<WT, INT>(a.x, a.y, b.x, b.y) + dist_xy<WT, INT>(a.xs, a.ys, b.xs, b.ys) - a.ds - b.ds;
}

// delta~ of the same pair from raw roots (see has_root_filter in tsp_dist.hpp); ds of both nodes is exact.
template <int WT>
__device__ __forceinline__ double pair_delta_approx(const NodeRec &a, const NodeRec &b) {
    return approx_root_dist<WT>(a.x, a.y, b.x, b.y) + approx_root_dist<WT>(a.xs, a.ys, b.xs, b.ys) - a.ds - b.ds;
}

}  // namespace tsp
