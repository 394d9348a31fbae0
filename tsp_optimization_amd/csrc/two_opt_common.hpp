// two_opt_common.hpp -- device helpers shared by the GRID and LDS 2-opt engines.
#pragma once
#include "tsp_internal.hpp"

#pragma clang fp contract(off)

namespace tsp {

using u64 = unsigned long long;
constexpr u64 kNoKey = ~0ull;
// runs with a tabu list: the handle's side words = {pairs the run's sweeps skipped as tabu, live tour edges of the sweeps whose number
// is 0 / 1 / 2 / 3 mod 4 (CLUSTER engine; the GRID engine uses [1], [2] within a launch)}
constexpr int kTabuSideSlots = 4;
constexpr int kTabuSideWords = 1 + kTabuSideSlots;

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

// min over the 64 lanes of an unsigned 64-bit key, through DPP row operations (a __shfl_xor is an LDS crossbar
// round trip per step; the step kernels and the nearest-neighbour construction are chains of such reductions):
// xor 1, xor 2, half-row mirror, row mirror inside each row of 16, then lane 15 -> next row, lane 31 -> upper half;
// lane 63 holds the result and hands it to everyone.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_min_step(unsigned long long v) {
    const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o < v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    v = dpp_min_step<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_min_step<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_min_step<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_min_step<0x140, 0xf>(v);   // row_mirror
    v = dpp_min_step<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_min_step<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// a double as an unsigned key of the same order (no NaNs here)
__device__ __forceinline__ u64 ordered_bits(double d) {
    const u64 b = (u64)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | (1ull << 63));
}
__device__ __forceinline__ double from_ordered_bits(u64 k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & ~(1ull << 63)) : ~k));
}

// BY_DELTA: arg-min of (delta, key); else: min key (first improving pair), delta rides along.
// Every lane ends up with the winner.
template <bool BY_DELTA>
__device__ __forceinline__ void wave_argmin(double &d, u64 &k) {
    // Almost every wave of a scan has no candidate at all (one move per sweep): one ballot and out.  A wave with no
    // key keeps (d, kNoKey) in every lane, which is what the reduction would have produced for the key; callers
    // only use d together with a valid key.
    if (!__any(k != kNoKey)) return;
    if constexpr (BY_DELTA) {
        const u64 dk = ordered_bits(d == 0.0 ? 0.0 : d);   // -0.0 and 0.0 are the same delta
        const u64 md = wave_min_u64(dk);
        k = wave_min_u64(dk == md ? k : kNoKey);           // among the minimal deltas, the first pair in scan order
        d = from_ordered_bits(md);
    } else {
        const u64 mk = wave_min_u64(k);
        const int src = __builtin_ctzll(__ballot(k == mk));
        const int lo = __builtin_amdgcn_readlane(__double2loint(d), src), hi = __builtin_amdgcn_readlane(__double2hiint(d), src);
        d = __hiloint2double(hi, lo);
        k = mk;
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

// sum over the 64 lanes (same DPP walk as wave_min_u64; a lane outside a step's row mask adds nothing)
template <typename T, int CTRL, int ROW_MASK>
__device__ __forceinline__ T dpp_add_step(T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit payloads");
    if constexpr (sizeof(T) == 4) {
        int b;
        __builtin_memcpy(&b, &v, 4);
        const int o = __builtin_amdgcn_update_dpp(0, b, CTRL, ROW_MASK, 0xf, false);
        T t;
        __builtin_memcpy(&t, &o, 4);
        return v + t;
    } else {
        unsigned long long b;
        __builtin_memcpy(&b, &v, 8);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        T t;
        __builtin_memcpy(&t, &o, 8);
        return v + t;
    }
}
template <typename T>
__device__ __forceinline__ T wave_sum_to_lane63(T v) {
    v = dpp_add_step<T, 0xB1, 0xf>(v);
    v = dpp_add_step<T, 0x4E, 0xf>(v);
    v = dpp_add_step<T, 0x141, 0xf>(v);
    v = dpp_add_step<T, 0x140, 0xf>(v);
    v = dpp_add_step<T, 0x142, 0xa>(v);
    v = dpp_add_step<T, 0x143, 0xc>(v);
    return v;   // complete in lane 63 only
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T *scratch /* >= blockDim/64 */) {
    v = wave_sum_to_lane63(v);
    const int tid = threadIdx.x;
    __syncthreads();
    if ((tid & 63) == 63) scratch[tid >> 6] = v;
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
