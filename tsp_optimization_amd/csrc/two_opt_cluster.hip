// two_opt_cluster.hip -- CLUSTER engine: C workgroups per tour, the whole 2-opt descent inside one launch.
//
// Every workgroup of a tour's cluster keeps a full replica of the tour (coordinates, order[], pos[]) in its CU's LDS,
// scans its share of the pair space, and the cluster exchanges ONE candidate per workgroup and step through HBM/L2:
// tagged 8-byte granules, written once with write-through (sc1) stores and swept by one wave of every workgroup until
// all tags carry the step's epoch (cdna_hip_programming.md Guideline 16, form R2: the data is the flag, no fence).
// Every workgroup then picks the same winner and applies the same move to its own replica, so the replicas never
// diverge and a step needs no grid barrier, no kernel boundary and no global tour state.  C = 1 is the degenerate
// case (no exchange); C = #CUs puts the whole chip on one tour (BASELINE configs[2]); B tours x C workgroups with
// B C <= #CUs serves multi-start / population shards that would otherwise leave CUs idle (configs[3], [4] on 8 GPUs).
//
// Selection rules and semantics are those of the other engines (two_opt_grid.hip header): FIRST = alg_2opt
// (src/heuristics.c:438-502), BEST = alg_2opt_tabu with skip_edge == NULL (src/tabusearch.c:107-178).
// Two scans:
//   tiles   every pair of the scanned rows visited, tiles of up to 32 rows x 512 columns dealt round-robin to the
//           cluster's workgroups, the rows per tile shrinking until every workgroup has one (FIRST on any metric;
//           BEST on metrics without the new-edge bound);
//   sorted  BEST on sqrt metrics: nodes renumbered along the Hilbert curve (tsp_dev_inst_create), 64 consecutive
//           ranks = one group; the box form of the new-edge bound decides 64 x 64 pairs at once.  The surviving
//           group pairs (host-built table, nearest first, dealt round-robin) are staged eight at a time (their node
//           records derived once into LDS, rows culled against the column box in the same pass), the live rows go
//           four at a time through an fp32 tier 0, and the pairs that survive it are queued per wave and taken
//           through tiers 1 and 2 sixty-four at a time.  Group bound: gmax2[g] = longest tour edge INCIDENT to a
//           node of g (either direction), which a move changes for at most four groups -- no O(n) rebuild per step.
//   FIRST on the sorted replica (round 3): a first-improvement step whose last hits lay hundreds of rows apart -- the tail of
//           every descent, the sweeps that find (almost) nothing, HEU_VNS's last sweeps of a round -- takes the sorted scan
//           too, with key = the first improving pair after the cursor (min pair index among delta < 0; bound 0 in the box
//           test): ONE step decides the whole rest of the sweep (a tiles sweep over all rows of rand10000 that finds nothing
//           costs 217 us).  Dense phases keep the probe and the tiles scan, which then read the replica through the id maps.
// Residency: the cluster protocol needs all B C workgroups on the chip at once; the host launches at most one per CU
// and every spin is bounded (a workgroup that gives up raises `err`, everybody leaves, the host falls back to GRID).
#include "two_opt_common.hpp"

#include <algorithm>
#include <time.h>
#include <vector>

#pragma clang fp contract(off)

namespace tsp {

constexpr int kClThreads = 512;
constexpr int kClWaves = kClThreads / 64;
constexpr int kClRows = 32;        // rows of one tile
constexpr int kClListCap = kClThreads;   // surviving group pairs a workgroup holds at a time (one round of box tests at least)

constexpr int kClQueue = 128;      // per-wave ring of pairs waiting for tiers 1 and 2 (entries; a power of two >= 128)
constexpr int kClSlotGranules = 4; // granules per workgroup and parity in the exchange area
constexpr int kClCopies = 8;       // copies of the exchange area (TSP_CLUSTER_COPIES): every producer writes all, a workgroup sweeps the one of its XCD
constexpr unsigned kClSpinLimit = 1u << 20;   // sweeps of the exchange area before a workgroup gives up (~ seconds): the backstop
constexpr int kClSpinMs = 50;                 // ... and the time a workgroup waits for its peers' tags (TSP_CLUSTER_SPIN_MS): a step
                                              // takes microseconds, so tens of milliseconds without a tag mean a peer is not resident
using idx_t = unsigned short;
using gu64 = __attribute__((address_space(1))) unsigned long long;
using gi32c = __attribute__((address_space(1))) int;

// Diagnostic build only (-DTSP_STAMPS): 100 MHz wall-clock time per phase, summed over a run by thread 0 of every
// workgroup of tour 0, in a buffer nothing else reads (cdna_hip_programming.md section 7, in-kernel stamps).
#ifdef TSP_STAMPS
__device__ unsigned long long g_cl_prof[256][8];
__device__ unsigned long long g_cl_wstat[256][4];   // per workgroup of tour 0: lane pairs, tier-1 pairs, delta expressions, staged records
__device__ unsigned long long g_cl_b0[8];   // sweeps with a winner, sum of -b0, sum of -delta of the winner, sweeps with b0 == delta, with b0 >= delta / 2, >= delta / 1.25
__device__ unsigned long long g_cl_tail[256][12];   // per workgroup of tour 0, tails of tabu() iterations inside a launch: ticks {cost + wait, decision (first workgroup), release, exchange, acquire, kick, -, tails; per sweep: the list's side effects (thread 64), the first workgroup's read of the live-edge count, the exchange proper}
__device__ unsigned long long g_cl_cnt[8];   // sorted scan, all workgroups: units, live rows, row quads, tier-1 blocks, tier-2 pairs, survivors
#define CL_T(k) do { if (tid == 0) { const unsigned long long t_ = wall_clock64(); prof[k] += t_ - tprev; tprev = t_; } } while (0)
#else
#define CL_T(k) do { } while (0)
#endif

struct ClusterArgs {
    const double2 *coord;   // internal numbering: node order (tiles) or Hilbert rank order incl. padding (sorted)
    int *orders;            // B x n: node (caller's numbering) at tour position p
    int *poss;              // B x n: position of node v, written back with the tour
    long long *stats_part;  // B x 256 x 4: executed-work counters per workgroup {lane pairs, tier-1 pairs, delta expressions, staged records}
    unsigned epoch0;        // exchange epochs of this launch start above this (even): no zeroing of the exchange area between launches
    TourState *states;
    const int *gid;         // sorted: internal id -> node
    const int *iid;         // sorted: node -> internal id
    const double4 *gbox;    // sorted: group boxes {min x, max x, min y, max y}
    const int *pairtab;     // sorted: C x ntests group pairs (r << 16 | c, -1 = none), nearest first, dealt in turn
    unsigned long long *slots;   // B x 2 x kClCopies x C x kClSlotGranules
    int copies;                  // copies in use (1 ... kClCopies)
    int xcd_local;               // block -> (tour, c) keeps a tour's workgroups on one XCD (host: the grid is a multiple of 8 C)
    int *err;
    int n, nid, ng, ntests, C, max_iters, rmin, rmax, rcap, rbs, count_evals;
    // BEST, sorted scan, with a tabu list (two_opt_tabu_list.hpp has the method): stamps, the compact list of the non-zero
    // ones, and the handle's side words {skipped pairs of the run, live tour edges of sweep number mod 3 = 0 / 1 / 2}
    int *tabu;
    const int2 *tabu_list;
    const int *tabu_list_n;
    int tabu_list_cap, iter, tenure;
    unsigned long long *tabu_side;
    // iterations of tabu() inside the launch (TABU variant; chain_n == 0: one descent, as ever): chain = {stop word, next iteration
    // of the chain, the incumbent's cost (double), 10 result words per iteration}, chain_par = the tenure per iteration; the kick's trials are host-drawn (chain_ab), snap = the incumbent's tour (nodes by position)
    int *chain;
    const int *chain_par;
    int chain_n;
    int chain_pairs;        // > 0: the kick's trials are taken IN ORDER from chain_ab (a rejected trial is followed by the next pair, as tabu()
                            // draws them: tabusearch.c:262-287), chain_pp = where the next one is; 0: trial k belongs to iteration k, one each
    const int *chain_ab;    // {a, b, a and b inside the replica} per pair
    int *chain_pp;
    int *snap;
    int probe;              // FIRST: largest distance (pairs) of the last hit after which a step starts with the probe; 0 = never
    int defer_moves;        // carry the swaps of a move out during the next step's exchange (TSP_CLUSTER_DEFER)
    int use_b0;             // BEST, sorted scan: start every sweep from the bound the previous exchange yields (TSP_CLUSTER_B0)
    int fs_rows;            // FIRST on the sorted replica: a step takes the box-pruned scan when the running mean of the rows between hits is at least this
    int fs_exit;            // FIRST, plain replica: the launch ends when that mean reaches this (the host goes on with the rank-order variant); 0 = never
    int fs_leave;           // FIRST, rank-order replica: the launch ends when the mean falls below this (back to the plain variant); 0 = never
    int stage_pairs;        // sorted: group pairs whose records are staged in LDS at a time
    unsigned spin_limit;    // sweeps of the exchange area before a workgroup gives up ...
    unsigned long long spin_ticks;   // ... or this much time (100 MHz ticks) without the peers' tags, whichever comes first
    int dbg;                // diagnostics (TSP_CLUSTER_DEBUG): 1 rebuild every group bound per step, 2 no row culling, 4 no box test
    double org_x, org_y;    // float replicas hold coordinates relative to this corner (exact: bounded integers)
    double margin, prune, sum_margin;
};

template <typename CT> struct ClCoord;
template <> struct ClCoord<double2> {
    static __device__ __forceinline__ double2 make(double2 c, double, double) { return c; }
    static __device__ __forceinline__ double4 box(double4 b, double, double) { return b; }
};
template <> struct ClCoord<float2> {
    static __device__ __forceinline__ float2 make(double2 c, double ox, double oy) {
        return make_float2((float)(c.x - ox), (float)(c.y - oy));
    }
    // group boxes live in the replica's frame too ({min x, max x, min y, max y}; integer coordinates: exact)
    static __device__ __forceinline__ double4 box(double4 b, double ox, double oy) {
        return make_double4(b.x - ox, b.y - ox, b.z - oy, b.w - oy);
    }
};

// Staged node records of the sorted scan.  Integer-coordinate instances (float replica): every field is a bounded
// integer, exact in a float, 24 bytes; otherwise the 48-byte NodeRec as it is.
struct alignas(8) StageRecF {
    float x, y, xs, ys, ds;
    unsigned short succ, id;   // 0xffff = none (padding)
};
static_assert(sizeof(StageRecF) == 24, "StageRecF must be 24 bytes");
template <typename CT> struct ClStage;
template <> struct ClStage<float2> {
    using rec = StageRecF;
    static constexpr bool kF32 = true;
    static __device__ __forceinline__ rec pack(const NodeRec &r) {
        rec o;
        o.x = (float)r.x; o.y = (float)r.y; o.xs = (float)r.xs; o.ys = (float)r.ys; o.ds = (float)r.ds;
        o.succ = (unsigned short)r.succ; o.id = (unsigned short)r.id;   // -1 -> 0xffff
        return o;
    }
    static __device__ __forceinline__ NodeRec unpack(const rec &o) {
        NodeRec r;
        r.x = (double)o.x; r.y = (double)o.y; r.xs = (double)o.xs; r.ys = (double)o.ys; r.ds = (double)o.ds;
        r.succ = o.succ == 0xffffu ? -1 : (int)o.succ; r.id = o.id == 0xffffu ? -1 : (int)o.id;
        return r;
    }
    static __device__ __forceinline__ void xyd(const rec &o, double &x, double &y, double &d) { x = (double)o.x; y = (double)o.y; d = (double)o.ds; }
    static __device__ __forceinline__ void xydf(const rec &o, float &x, float &y, float &d) { x = o.x; y = o.y; d = o.ds; }
};
template <> struct ClStage<double2> {
    using rec = NodeRec;
    static constexpr bool kF32 = false;
    static __device__ __forceinline__ rec pack(const NodeRec &r) { return r; }
    static __device__ __forceinline__ NodeRec unpack(const rec &o) { return o; }
    static __device__ __forceinline__ void xyd(const rec &o, double &x, double &y, double &d) { x = o.x; y = o.y; d = o.ds; }
    static __device__ __forceinline__ void xydf(const rec &o, float &x, float &y, float &d) { x = (float)o.x; y = (float)o.y; d = (float)o.ds; }
};

__host__ __device__ inline size_t cl_align16(size_t x) { return (x + 15) & ~(size_t)15; }

// LDS carve-up (host and device agree through this one function)
struct ClLayout {
    size_t coord, order, pos, gbox, gmax, stage, list, items, queue, rows, scratch, total;
};
constexpr int kClMaxStagePairs = 8;   // <= kClWaves: one wave stages one pair
static_assert(kClMaxStagePairs <= kClWaves && kClMaxStagePairs * 128 <= 1024, "one wave per staged pair; a unit holds a 10-bit stage slot");
__host__ __device__ inline ClLayout cl_layout(int n, int nid, int ng, size_t coord_elem, bool sorted, int stage_pairs, bool tiles_too = false) {
    ClLayout L;
    size_t o = 0;
    L.coord = o; o = cl_align16(o + coord_elem * (size_t)nid);
    L.order = o; o = cl_align16(o + sizeof(idx_t) * (size_t)n);
    L.pos = o; o = cl_align16(o + sizeof(idx_t) * (size_t)nid);
    L.gbox = L.gmax = L.stage = L.list = L.items = L.queue = L.rows = o;
    if (sorted) {
        const size_t rec = coord_elem == sizeof(float2) ? sizeof(StageRecF) : sizeof(NodeRec);
        L.gbox = o; o = cl_align16(o + sizeof(double4) * (size_t)ng);
        L.gmax = o; o = cl_align16(o + sizeof(double) * (size_t)ng);
        L.stage = o; o = cl_align16(o + rec * 128 * (size_t)stage_pairs);
        L.list = o; o = cl_align16(o + sizeof(int) * kClListCap);
        L.items = o; o = cl_align16(o + sizeof(unsigned short) * 256 * (size_t)stage_pairs);   // (row, quarter) units: at most 4 per staged row
        L.queue = o; o = cl_align16(o + sizeof(unsigned) * kClQueue * kClWaves);
        if (tiles_too) { L.rows = o; o = cl_align16(o + sizeof(NodeRec) * kClRows); }   // first improvement: both scans
    } else {
        L.rows = o; o = cl_align16(o + sizeof(NodeRec) * kClRows);
    }
    L.scratch = o; o += 1024;
    L.total = o;
    return L;
}

template <int WT, bool INT, typename CT>
__device__ __forceinline__ NodeRec cl_node(const CT *coord, const idx_t *order, const idx_t *pos, int n, int v) {
    const int p = (int)pos[v];
    const int q = p + 1 == n ? 0 : p + 1;
    const int s = (int)order[q];
    const CT c = coord[v], cs = coord[s];
    NodeRec r;
    r.x = (double)c.x; r.y = (double)c.y; r.xs = (double)cs.x; r.ys = (double)cs.y;
    r.ds = dist_xy<WT, INT>(r.x, r.y, r.xs, r.ys);
    r.succ = s; r.id = v;
    return r;
}

// A pending reversal of the tour positions pa+1 .. pb (L of them, cyclic; L == 0: none): the move has been decided, the swaps
// have not been carried out yet.  Position p holds afterwards what its mirror inside the range holds now.
struct ClView {
    int pa, pb, L;
};
__device__ __forceinline__ int cl_mirror(const ClView &m, int p, int n) {
    int d = p - m.pa - 1;
    if (d < 0) d += n;
    if (d >= m.L) return p;
    int q = m.pb - d;
    if (q < 0) q += n;
    return q;
}
// cl_node on the tour as it will be once the pending reversal has been carried out
template <int WT, bool INT, typename CT>
__device__ __forceinline__ NodeRec cl_node_v(const CT *coord, const idx_t *order, const idx_t *pos, int n, int v, const ClView &m) {
    if (m.L == 0) return cl_node<WT, INT, CT>(coord, order, pos, n, v);   // wave-uniform
    const int p = cl_mirror(m, (int)pos[v], n);
    const int q = p + 1 == n ? 0 : p + 1;
    const int s = (int)order[cl_mirror(m, q, n)];
    const CT c = coord[v], cs = coord[s];
    NodeRec r;
    r.x = (double)c.x; r.y = (double)c.y; r.xs = (double)cs.x; r.ys = (double)cs.y;
    r.ds = dist_xy<WT, INT>(r.x, r.y, r.xs, r.ys);
    r.succ = s; r.id = v;
    return r;
}

template <int WT, bool INT, typename CT>
__device__ __forceinline__ double cl_dist(const CT *coord, int u, int v) {
    const CT a = coord[u], b = coord[v];
    return dist_xy<WT, INT>((double)a.x, (double)a.y, (double)b.x, (double)b.y);
}

// a / b for 0 <= a < 2^22, 0 < b < 2^22 without the integer-division expansion (~40 instructions each; a first-improvement
// step needs three of them before its first tile): float quotient, one correction either way
__device__ __forceinline__ int cl_div(int a, int b) {
    int q = (int)((float)a / (float)b);
    if (q * b > a) --q;
    if ((q + 1) * b <= a) ++q;
    return q;
}

// min / max over each row of 16 lanes (DPP: xor 1, xor 2, half-row mirror, row mirror), every lane of the row gets it
template <int CTRL> __device__ __forceinline__ float cl_dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ double cl_dpp_mov(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float cl_min2(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double cl_min2(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float cl_max2(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double cl_max2(double a, double b) { return fmax(a, b); }
template <typename T> __device__ __forceinline__ T cl_row16_min(T v) {
    v = cl_min2(v, cl_dpp_mov<0xB1>(v)); v = cl_min2(v, cl_dpp_mov<0x4E>(v)); v = cl_min2(v, cl_dpp_mov<0x141>(v)); v = cl_min2(v, cl_dpp_mov<0x140>(v));
    return v;
}
template <typename T> __device__ __forceinline__ T cl_row16_max(T v) {
    v = cl_max2(v, cl_dpp_mov<0xB1>(v)); v = cl_max2(v, cl_dpp_mov<0x4E>(v)); v = cl_max2(v, cl_dpp_mov<0x141>(v)); v = cl_max2(v, cl_dpp_mov<0x140>(v));
    return v;
}
template <typename T> __device__ __forceinline__ T cl_readlane(T v, int l);
template <> __device__ __forceinline__ float cl_readlane<float>(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
template <> __device__ __forceinline__ double cl_readlane<double>(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// max over the 64 lanes of a non-negative double (orders like its bits)
__device__ __forceinline__ double cl_wave_max_nonneg(double v) {
    const u64 mb = ~wave_min_u64(~(u64)__double_as_longlong(v));
    return __longlong_as_double((long long)mb);
}

// One workgroup's candidate of a step.  key: the pair in the caller's node numbering (lower id first; kNoKey = none),
// ipair: the same two nodes in internal numbering, (node of the lower id) << 16 | (the other one).
struct ClCand {
    double d;
    u64 key;
    unsigned ipair;
    // out (best improvement, sorted scan): this lane's share of the cluster's candidates (internal pair, delta; 0 = none), from
    // which the caller derives the bound the next sweep starts from (cl_next_bound)
    unsigned c_ip[4] = {0u, 0u, 0u, 0u};
    double c_d[4] = {0.0, 0.0, 0.0, 0.0};
};

// The bound the next sweep starts from.  The winner's move reverses the tour positions pa+1 .. pb and changes the successor of
// exactly those nodes and of a = the winner's first node.  A candidate (u, v) of another workgroup with neither node among them
// keeps succ u, succ v, its adjacency and therefore its delta: the next sweep's minimum is at most that.  Every workgroup holds
// all C candidates after an exchange and computes the same value.  One wave; pos[] as it stands when the winner's positions are read.
__device__ __forceinline__ double cl_next_bound(const ClCand &cd, const idx_t *pos, int n, bool inside_ok, unsigned (&mem_ip)[4], double (&mem_d)[4]) {
    double b0 = 0.0;
    if (cd.key != kNoKey && cd.d < 0.0) {
        const int wi = (int)(cd.ipair >> 16), wj = (int)(cd.ipair & 0xffffu);
        const int pa = (int)pos[wi], pb = (int)pos[wj];
        int L = pb - pa; if (L < 0) L += n;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // the candidate workgroup 64 q + lane has just sent, and the best one it sent in earlier sweeps that no move has touched
            // since (its own new candidate is at least as good -- until that one is the move, or touched by it)
            double keep_d = 0.0;
            unsigned keep_ip = 0u;
#pragma unroll
            for (int src = 0; src < 2; ++src) {
                const double d = src ? mem_d[q] : cd.c_d[q];
                const unsigned ip = src ? mem_ip[q] : cd.c_ip[q];
                if (d < 0.0) {
                    const int ui = (int)(ip >> 16), uj = (int)(ip & 0xffffu);
                    int du = (int)pos[ui] - pa - 1, dv = (int)pos[uj] - pa - 1;
                    if (du < 0) du += n;
                    if (dv < 0) dv += n;
                    const bool untouched = ui != wi && uj != wi && du >= L && dv >= L;
                    // both nodes strictly inside the reversed positions (pa+1 .. pb-1): their successor edges are reversed with the
                    // segment, not removed, and the same exchange of those two edges is the pair of the two old successors on the new
                    // tour -- the same four lengths, the same delta when the terms are integers (no tabu list: its four stamps differ)
                    const bool inside = inside_ok && du <= L - 2 && dv <= L - 2;
                    if ((untouched || inside) && d < b0) b0 = d;
                    if (untouched && d < keep_d) { keep_d = d; keep_ip = ip; }
                }
            }
            mem_d[q] = keep_d; mem_ip[q] = keep_ip;
        }
    }
    return from_ordered_bits(wave_min_u64(ordered_bits(b0)));
}

// ---- the exchange: one candidate per workgroup and step -------------------------------------------------------
// Producer: lane 0 of the workgroup's first wave, NG 8-byte sc1 stores {epoch, payload}.  Consumer: the first wave of
// every workgroup sweeps all C candidates with sc1 loads until every tag carries the epoch.  The area is laid out
// granule-major ([parity][granule][workgroup]): one wave-instruction of the sweep reads 64 consecutive granules = four
// whole 128-byte lines, and a candidate is as few granules as the variant allows --
//   granule 0  the pair in the caller's numbering (two 16-bit ids; 0xffffffff = no candidate)
//   granule 1  the pair in internal numbering (sorted scan only: elsewhere it is granule 0)
//   then       the delta: one granule holding it as an int32 when the metric's values are bounded integers (SMALLD:
//              the integer-coordinate variants, |delta| < 2^23), else two granules with the halves of the double.
// Two parities: a workgroup can be at most one step ahead of the slowest one (it cannot finish step s + 1 before
// everybody has published s + 1, i.e. has finished reading s), so parity s is never rewritten while somebody still
// reads it.  Returns false when the sweep gave up (a peer is not resident): *err is raised.
template <bool BEST, bool SORTED, bool SMALLD>
__device__ __forceinline__ bool cl_exchange(gu64 *area, int C, int c, unsigned ep, ClCand &cd, int *err, unsigned spin_limit,
                                            unsigned long long spin_ticks, int copies, int mycopy) {
    constexpr int NG = 1 + (SORTED ? 1 : 0) + (SMALLD ? 1 : 2);
    constexpr int GD = SORTED ? 2 : 1;   // first delta granule
    const int lane = threadIdx.x & 63;
    const u64 tag = (u64)ep << 32;
    // Copies: a round among 256 workgroups costs 2.6 us when all of them sweep the same 16 - 48 lines and 2.2 us when the
    // sweepers of each XCD have lines of their own (tools/ubench/xchg.hip); the stores of the copies are one instruction
    gu64 *par = area + (size_t)(ep & 1u) * kClCopies * C * kClSlotGranules + (size_t)(lane < copies ? lane : 0) * C * kClSlotGranules;
    if (lane < copies) {
        const u64 db = (u64)__double_as_longlong(cd.d);
        const unsigned kp = cd.key == kNoKey ? 0xffffffffu : (((unsigned)key_i(cd.key) & 0xffffu) << 16) | ((unsigned)key_j(cd.key) & 0xffffu);
        __hip_atomic_store(par + c, tag | kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (SORTED) __hip_atomic_store(par + C + c, tag | cd.ipair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (SMALLD) {
            __hip_atomic_store(par + GD * C + c, tag | (u64)(unsigned)(int)cd.d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(par + GD * C + c, tag | (db & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(par + (GD + 1) * C + c, tag | (db >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    par = area + (size_t)(ep & 1u) * kClCopies * C * kClSlotGranules + (size_t)mycopy * C * kClSlotGranules;
    constexpr int Q = 4;   // C <= 256: at most four candidates per lane
    u64 g[Q][NG];
    bool have[Q];          // candidate q of this lane is complete: later passes leave it alone (fewer requests on the hot lines)
#pragma unroll
    for (int q = 0; q < Q; ++q) have[q] = q * 64 + lane >= C;
    unsigned spins = 0;
    unsigned long long t_wait = 0;   // when the 64th sweep without all tags began (a sweep is ~1 us: most exchanges never read the clock)
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (q * 64 < C) {   // wave-uniform
                if (!have[q]) {
                    const int k = q * 64 + lane;
#pragma unroll
                    for (int w = 0; w < NG; ++w)
                        g[q][w] = __hip_atomic_load(par + (size_t)w * C + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool all = true;
#pragma unroll
                    for (int w = 0; w < NG; ++w) all = all && (g[q][w] >> 32) == (u64)ep;
                    have[q] = all;
                    ok = ok && all;
                }
            }
        }
        if (__all(ok)) break;
        ++spins;
        bool late = spins > spin_limit;
        if ((spins & 63u) == 0) {
            const unsigned long long now = wall_clock64();
            if (t_wait == 0) t_wait = now;
            late = late || now - t_wait > spin_ticks;
        }
        if (late ||
            ((spins & 1023u) == 0 && __hip_atomic_load((gi32c *)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            if (lane == 0) __hip_atomic_store((gi32c *)err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    double bd = 0.0;
    u64 key = kNoKey;
    unsigned ip = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int k = q * 64 + lane;
        if (k < C) {
            const unsigned kp = (unsigned)g[q][0];
            if (kp != 0xffffffffu) {
                const u64 kk = make_key((int)(kp >> 16), (int)(kp & 0xffffu));
                double d;
                if constexpr (SMALLD) d = (double)(int)(unsigned)g[q][GD];
                else d = __longlong_as_double((long long)((g[q][GD] & 0xffffffffull) | (g[q][GD + 1] << 32)));
                const bool take = BEST ? better(d, kk, bd, key) : (kk < key);
                if (key == kNoKey || take) { bd = d; key = kk; ip = SORTED ? (unsigned)g[q][1] : kp; }
            }
        }
    }
    double wd = bd;
    u64 wk = key;
    wave_argmin<BEST>(wd, wk);
    // the winner's internal pair rides along: a pair is evaluated by exactly one lane of one workgroup
    const unsigned long long owners = __ballot(key == wk && wk != kNoKey);
    unsigned wip = 0;
    if (owners) wip = (unsigned)__builtin_amdgcn_readlane((int)ip, __builtin_ctzll(owners));
    cd.d = wd; cd.key = wk; cd.ipair = wip;
    if constexpr (BEST && SORTED) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int k = q * 64 + lane;
            cd.c_ip[q] = 0u; cd.c_d[q] = 0.0;
            if (k < C && (unsigned)g[q][0] != 0xffffffffu) {
                double d;
                if constexpr (SMALLD) d = (double)(int)(unsigned)g[q][GD];
                else d = __longlong_as_double((long long)((g[q][GD] & 0xffffffffull) | (g[q][GD + 1] << 32)));
                cd.c_ip[q] = (unsigned)g[q][1]; cd.c_d[q] = d;
            }
        }
    }
    return true;
}

// TABU: a best-improvement run with a tabu list (sorted scan only) -- a variant of its own, so that the plain descent carries
// none of it (as run-time branches on a pointer the list code cost the plain sweep 2.5 %: 11.3 -> 11.6 us at n = 10 000)
template <int WT, bool INT, int MODE, typename CT, bool SORTED, bool TABU = false>
__global__ __launch_bounds__(kClThreads) void k_cluster_two_opt(const ClusterArgs a) {
    static_assert(!TABU || (SORTED && MODE == TSP_2OPT_BEST), "tabu lists ride on the sorted best-improvement scan");
    static_assert(!SORTED || has_root_filter<WT>(), "the sorted scan needs the new-edge bound of a sqrt metric");
    constexpr bool BEST = MODE == TSP_2OPT_BEST;
    constexpr bool FS = SORTED && !BEST;   // first improvement on the sorted replica: internal ids = Hilbert ranks, caller ids through a.gid / a.iid
    constexpr bool ATT10 = WT == WT_ATT || WT == WT_ATT_ICOORD;
    // bounded integer distances (< 2^21): a delta is an integer below 2^23 in magnitude and travels as an int32
    constexpr bool kSmallD = WT == WT_EUC_2D_ICOORD || WT == WT_CEIL_2D_ICOORD || WT == WT_ATT_ICOORD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = a.n, nid = a.nid, ng = a.ng, C = a.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups are dealt to the XCDs in turn (block b: XCD b % 8, tools/ubench/xchg.hip).  Several tours: the C workgroups
    // of a tour come from ONE XCD when the grid divides that way -- an exchange among 16 workgroups behind one L2 takes
    // 0.63 us, 0.9 us when they span all eight (a placement hint: any mapping is correct)
    int tour = (int)blockIdx.x / C, c = (int)blockIdx.x % C;
    if (a.xcd_local) {
        const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        tour = x * ((int)gridDim.x / 8 / C) + j / C;
        c = j % C;
    }
    using SR = ClStage<CT>;
    const ClLayout L = cl_layout(n, nid, ng, sizeof(CT), SORTED, a.stage_pairs, FS);
    CT *coord = reinterpret_cast<CT *>(smem + L.coord);
    idx_t *order = reinterpret_cast<idx_t *>(smem + L.order);
    idx_t *pos = reinterpret_cast<idx_t *>(smem + L.pos);
    double4 *gbox = reinterpret_cast<double4 *>(smem + L.gbox);
    double *gmax = reinterpret_cast<double *>(smem + L.gmax);
    typename SR::rec *stage = reinterpret_cast<typename SR::rec *>(smem + L.stage);
    unsigned short *s_items = reinterpret_cast<unsigned short *>(smem + L.items);
    unsigned *s_queue = reinterpret_cast<unsigned *>(smem + L.queue);
    int *s_list = reinterpret_cast<int *>(smem + L.list);
    NodeRec *s_rows = reinterpret_cast<NodeRec *>(smem + L.rows);
    char *scratch = smem + L.scratch;
    double *s_d = reinterpret_cast<double *>(scratch);                 // 16
    u64 *s_k = reinterpret_cast<u64 *>(scratch + 128);                 // 16
    long long *s_ll = reinterpret_cast<long long *>(scratch + 256);    // 16
    int *s_wcount = reinterpret_cast<int *>(scratch + 384);            // 16 ints
    double *s_win_d = reinterpret_cast<double *>(scratch + 448);
    u64 *s_win_k = reinterpret_cast<u64 *>(scratch + 456);
    unsigned *s_win_ip = reinterpret_cast<unsigned *>(scratch + 464);
    int *s_fail = reinterpret_cast<int *>(scratch + 468);
    int *s_vote = reinterpret_cast<int *>(scratch + 472);   // tiles scan, first improvement: the stamp of the last vote that saw a hit
    static_assert(kClWaves <= 8, "scratch carve-up: eight wave winners");
    unsigned *s_ip = reinterpret_cast<unsigned *>(scratch + 480);      // 8: the waves' winners (internal pairs)
    double *s_b0 = reinterpret_cast<double *>(scratch + 256 + 15 * 8);  // the next sweep's bound (s_ll[15]: block sums use 0 .. 7, the final counters come after the loop)
    int *s_nitems = reinterpret_cast<int *>(scratch + 476);   // the workgroup's own winner (before the exchange)
    double *s_chunk = reinterpret_cast<double *>(scratch + 512);       // 64 doubles (fcost cost recompute)
    float4 *s_rowsf = reinterpret_cast<float4 *>(scratch + 512);       // tiles scan, float replica: x, y, edge length of the tile's rows (shares the chunk: that is used at the end of a run only)

    TourState *st = a.states + tour;
    if (st->done) return;
    int *order_g = a.orders + (size_t)tour * n;
    gu64 *area = (gu64 *)a.slots + (size_t)tour * 2 * kClCopies * C * kClSlotGranules;
    // the iteration of tabu() this launch is at (its number and tenure decide what is tabu): a launch that carries a chain of
    // iterations goes on where the launch before it stopped (chain[1])
    int cur_iter = a.iter, cur_ten = a.tenure, ck = 0, kpp = 0;   // kpp: the pair the next kick trial takes
    int4 nx_pair = make_int4(0, 0, 0, 0);   // ... fetched ahead ({a, b, a and b inside the replica}: asked for when the kick before it is through)
    int nx_pp = -1;
    double inc_best = 0.0;
    if constexpr (TABU) {
        if (a.chain_n > 0) {
            ck = a.chain[1];
            cur_iter = a.iter + ck; cur_ten = a.chain_par[ck];
            inc_best = *reinterpret_cast<const double *>(a.chain + 2);
            kpp = a.chain_pairs > 0 ? *a.chain_pp : ck;
            if (kpp < (a.chain_pairs > 0 ? a.chain_pairs : a.chain_n)) { nx_pair = reinterpret_cast<const int4 *>(a.chain_ab)[kpp]; nx_pp = kpp; }
        }
    }

    // the replica: eight loads per thread in flight at a time (a launch loads 120 KB per workgroup at n = 10 000; one load per
    // thread and trip was ~40 dependent memory latencies -- most of the fixed cost of a short resident run)
    constexpr int LU = 8;
    for (int v0 = tid; v0 < nid; v0 += LU * kClThreads) {
        double2 cv[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) { const int v = v0 + u * kClThreads; cv[u] = a.coord[v < nid ? v : 0]; }
#pragma unroll
        for (int u = 0; u < LU; ++u) { const int v = v0 + u * kClThreads; if (v < nid) coord[v] = ClCoord<CT>::make(cv[u], a.org_x, a.org_y); }
    }
    for (int p0 = tid; p0 < n; p0 += LU * kClThreads) {
        int w[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) { const int p = p0 + u * kClThreads; w[u] = order_g[p < n ? p : 0]; }
        if constexpr (SORTED) {
#pragma unroll
            for (int u = 0; u < LU; ++u) w[u] = a.iid[w[u]];
        }
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int p = p0 + u * kClThreads;
            if (p < n) { order[p] = (idx_t)w[u]; pos[w[u]] = (idx_t)p; }
        }
    }
    if constexpr (SORTED) {
        for (int g = tid; g < ng; g += kClThreads) gbox[g] = ClCoord<CT>::box(a.gbox[g], a.org_x, a.org_y);
    }
    if (tid == 0) { *s_fail = 0; *s_nitems = 0; *s_vote = 0; }
    int vote_seq = 0;
    int ci = st->ci, cj = st->cj, chunk = min(max(st->chunk_rows, 1), a.rmax), done = 0;
    int hit_rows = st->hit_rows;   // FIRST: running mean of the rows between hits (kept from call to call: HEU_VNS's rounds look alike)
    // caller's node id <-> id inside the replica (the same thing unless the replica is in rank order)
    auto to_int = [&](int v) -> int { if constexpr (SORTED) return a.iid[v]; else return v; };
    auto to_ext = [&](int v) -> int { if constexpr (SORTED) return a.gid[v]; else return v; };
    double obj = st->obj, seen = st->seen_cost;
    long long sweeps = st->sweeps, evals = st->evals, moves = st->moves, reversed = st->reversed,
              scanned = st->pairs_scanned, steps = st->steps;
    __syncthreads();

    // longest tour edge incident to a node of group g, for the groups g0, g0 + stride, ... (one wave per group)
    auto group_bounds = [&](int g0, int stride, int gend) {
        // four groups at a time: their chains of dependent LDS reads (position, two neighbours, three coordinates) advance together
        // (a launch of the sorted scan starts with all ng groups: 20 per wave at n = 10 000, the largest fixed item of a short launch)
        constexpr int GU = 4;
        for (int gb = g0; gb < gend; gb += stride * GU) {
            int v[GU], p[GU], su[GU], pr[GU];
            double m[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                v[u] = (gb + u * stride) * 64 + lane;
                p[u] = (int)pos[min(v[u], n - 1)];
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                su[u] = (int)order[p[u] + 1 == n ? 0 : p[u] + 1];
                pr[u] = (int)order[p[u] == 0 ? n - 1 : p[u] - 1];
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int vc = min(v[u], n - 1);
                m[u] = fmax(cl_dist<WT, INT, CT>(coord, vc, su[u]), cl_dist<WT, INT, CT>(coord, vc, pr[u]));
                if (v[u] >= n) m[u] = 0.0;
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int g = gb + u * stride;
                if (g < gend) {   // wave-uniform
                    const double mm = cl_wave_max_nonneg(m[u]);
                    if (lane == 0) gmax[g] = mm;
                }
            }
        }
    };
    // first improvement on the rank-order replica: the group bounds are only read by a box-pruned step; moves made by tiles /
    // probe steps leave them stale (gmax_dirty) and the next box-pruned step rebuilds them all
    bool gmax_dirty = FS;
    if constexpr (SORTED && BEST) {
        group_bounds(wave, kClWaves, ng);
        __syncthreads();
    }

    // iterations of tabu() inside the launch, integer costs: the tour's cost is kept by every workgroup from move to move and kick
    // to kick (integer terms: the sum is exact in any order and equals the reference's recomputation, tabusearch.c:168-172)
    double run_obj = 0.0;
    if constexpr (TABU && INT) {
        if (a.chain_n > 0) {
            double cc = 0.0;
            for (int v = tid; v < n; v += kClThreads) cc += cl_node<WT, INT, CT>(coord, order, pos, n, v).ds;
            run_obj = block_sum<double>(cc, s_d);
            __syncthreads();
        }
    }
    const double prune2 = 2.0 * a.prune;   // doubled: keeps ties (a lane does not meet its pairs in key order)
    bool failed = false;
    // the workgroup's share of the group-pair table never changes: when it fits one round of box tests (one entry per
    // thread) it is read once, not once per step (a global load at the head of every step's critical path otherwise)
    int tab_reg = -1;
    // ... and when it fits ONE WAVE (n = 10 000 on 256 workgroups: 49 group pairs each), every wave runs all the tests itself on
    // the same numbers: the survivors are a ballot in every wave's registers -- no compaction through LDS, no barrier between the
    // tests and the staging of the survivors
    const bool tests_per_wave = SORTED && a.ntests <= 64;
    if constexpr (SORTED) {
        if (tests_per_wave) { if (lane < a.ntests) tab_reg = a.pairtab[(size_t)c * a.ntests + lane]; }
        else if (a.ntests <= kClThreads && tid < a.ntests) tab_reg = a.pairtab[(size_t)c * a.ntests + tid];
    }
    // executed work, per wave (wave-uniform values): rows or row-lanes through tier 0, pairs queued for tier 1, delta
    // expressions, staged records; summed into the tour's control block at the end of the launch
    long long w_lane = 0, w_t1 = 0, w_ex = 0, w_st = 0;
    long long tabu_cnt = 0;   // this thread's share of the pairs the reference's scan skips as tabu (linear terms)
#ifdef TSP_STAMPS
    unsigned long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = wall_clock64();
#endif
    int mycopy = 0;   // the copy of the exchange area this workgroup sweeps: its XCD's (any assignment is correct: every copy holds everything)
    if (a.copies > 1) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        mycopy = (int)(xcc & 0xfu) % a.copies;
    }
    unsigned xep = a.epoch0;   // exchange epochs (a step decided by the probe has none); slots hold smaller ones from earlier launches
    bool probe_on = true;    // FIRST: the last hit lay close to the cursor (every workgroup keeps the same value)
    if constexpr (!BEST) {
        // Arrival rendezvous.  A first-improvement launch whose steps are all decided by the probe exchanges nothing, so the
        // cluster's first workgroup could reach its write-back of order[] / pos[] / the control block while a workgroup that
        // was dispatched late is still loading them: one (empty) exchange after the replica load -- nobody passes it before
        // everybody has loaded.  (Best improvement exchanges in every step: its first step is the rendezvous.)
        if (C > 1) {
            if (wave == 0) {
                ClCand none{0.0, kNoKey, 0u};
                const bool okx = cl_exchange<BEST, SORTED, kSmallD>(area, C, c, ++xep, none, a.err, a.spin_limit, a.spin_ticks, a.copies, mycopy);
                if (lane == 0) *s_fail = okx ? 0 : 1;
            }
            __syncthreads();
            if (*s_fail) failed = true;
        }
    }
    // FIRST, evaluation count: adjacent pairs passed so far in this launch (adj_seen), the moves' corrections of the current
    // sweep (adjD), the tour edges that lay behind the cursor when the launch took the sweep over (adjA0)
    long long adj_seen = 0, adjD = 0, adjA0 = 0;
    bool sweep_open = false;
    auto edges_upto = [&](u64 K) -> long long {   // tour edges whose pair is <= K in scan order (block-wide; first workgroup only)
        long long cnt = 0;
        for (int p = tid; p < n; p += kClThreads) {
            const int u = to_ext((int)order[p]), v = to_ext((int)order[p + 1 == n ? 0 : p + 1]);
            cnt += make_key(min(u, v), max(u, v)) <= K ? 1 : 0;
        }
        return block_sum<long long>(cnt, s_ll);
    };
    if constexpr (!BEST) {
        if (a.count_evals && c == 0 && !failed && (ci != 0 || cj != 0)) {
            adjA0 = edges_upto(make_key(ci, cj));
            sweep_open = true;
        }
    }
    double b0 = 0.0;      // BEST, sorted scan: bound the sweep starts from (<= 0; see cl_next_bound)
    unsigned b0_mem_ip[4] = {0u, 0u, 0u, 0u};   // (first wave: the candidates it remembers from earlier sweeps, four workgroups per lane)
    double b0_mem_d[4] = {0.0, 0.0, 0.0, 0.0};
    // Deferred moves.  The swaps of a move that an EXCHANGE step decided are not carried out at once: the next step scans the tour
    // through the closed form of the pending reversal (ClView), and the swaps are done by waves 1 .. 7 while wave 0 runs that
    // step's exchange -- time they would spend waiting for it.  (Not with a tabu list: those waves work on the list then.  Not for
    // C == 1: no exchange to hide behind.  A move decided by the probe is carried out at once, after any pending one.)
    ClView view{0, 0, 0};
    const bool defer_on = !TABU && C > 1 && a.defer_moves;
    auto swaps = [&](const ClView &m, int t0, int stride) {   // the reversal itself, by the threads t0, t0 + stride, ...
        const int half = m.L >> 1;
        for (int t = t0; t < half; t += stride) {
            int p = m.pa + 1 + t; if (p >= n) p -= n;
            int q = m.pb - t; if (q < 0) q += n;
            const idx_t u = order[p], w = order[q];
            order[p] = w; order[q] = u;
            pos[w] = (idx_t)p; pos[u] = (idx_t)q;
        }
    };
    auto flush_view = [&]() {   // workgroup-wide: carry the pending reversal out now
        if (view.L != 0) {      // the same in every thread
            swaps(view, tid, kClThreads);
            view.L = 0;
            __syncthreads();
        }
    };
    // the group bounds of the four nodes whose incident edges the 2-exchange (wi, wj) changes, from the tour BEFORE the swaps with
    // those four edges patched (see the move below); waves 1 .. 4, one group each
    auto patch_bounds = [&](int wi, int wj, int pa, int pb) {
        const int a1 = (int)order[pa + 1 == n ? 0 : pa + 1], b1 = (int)order[pb + 1 == n ? 0 : pb + 1];
        if (wave >= 1 && wave <= 4) {   // (wave 0 computes the next sweep's bound meanwhile)
            const int g = wave == 1 ? (wi >> 6) : (wave == 2 ? (wj >> 6) : (wave == 3 ? (a1 >> 6) : (b1 >> 6)));
            const int v = g * 64 + lane;
            double m = 0.0;
            if (v < n) {
                const int p = (int)pos[v];
                int su = (int)order[p + 1 == n ? 0 : p + 1], pr = (int)order[p == 0 ? n - 1 : p - 1];
                if (v == wi) su = wj;            // a -> b
                else if (v == wj) su = wi;       // b's old successor b1 gives way to a
                if (v == a1) pr = b1;            // a1's old predecessor a gives way to b1
                else if (v == b1) pr = a1;       // b1's old predecessor b gives way to a1
                m = fmax(cl_dist<WT, INT, CT>(coord, v, su), cl_dist<WT, INT, CT>(coord, v, pr));
            }
            m = cl_wave_max_nonneg(m);
            if (lane == 0) gmax[g] = m;
        }
    };
    // tabu lists: the first list entry this thread accounts for in every sweep, and its ids inside the replica (see the side effects)
    int2 tl_e = make_int2(0, 0);
    int tl_iu = 0, tl_iv = 0;
    bool tl_have = false;
    // ... and what that entry contributed in the last sweep, with the four tour neighbours it was computed from.  Inside one
    // iteration of tabu() nothing becomes tabu (stamps are written by kicks only) and what has expired stays expired: an entry
    // whose neighbours are the same contributes the same, and an entry whose stamp is zero stays out -- no load at all.
    // tl_state: 0 look, 1 the stamp is zero (until the next kick), 2 live: tl_cnt / tl_edge hold for the neighbours tl_nb0 / tl_nb1
    int tl_state = 0, tl_cnt = 0;
    unsigned tl_nb0 = 0, tl_nb1 = 0;
    bool tl_edge = false;
#ifdef TSP_STAMPS
    unsigned tl_n_skip = 0, tl_n_hit = 0, tl_n_full = 0;
#endif
    int tl_m = 0;   // entries of the list (the count runs on past the capacity, as on the host): read once, kept up with the kicks of a chain
    if constexpr (TABU) tl_m = *a.tabu_list_n;
    bool leave = false;   // FIRST: hand the descent to the other variant of this kernel (the host launches it)
    for (int iter = 0; iter < a.max_iters && !done && !failed && !leave; ++iter) {
        int row_lo = 0, row_hi = n - 1;
        double bd = (SORTED && BEST) ? b0 : 0.0;   // every lane starts from the bound; a lane without a pair keeps key == kNoKey
        u64 key = kNoKey;
        unsigned ipair = 0;
        const int slot_cur = TABU ? (int)(sweeps % 4) : 0, slot_read = (slot_cur + 2) & 3;   // tabu lists: live tour edges per sweep; the slot of two sweeps ago
        int si = ci, sj = cj;    // FIRST: where the tiles scan starts (the probe moves it on when it finds nothing)
        bool probe_hit = false;
        if constexpr (!BEST) {
            // ---- probe: the next 512 pairs in scan order, one per thread (see two_opt_lds.hip) -----------------
            // Every workgroup of the cluster runs it on its own replica and gets the same answer: a step the probe
            // decides needs no exchange at all.  It only runs while hits come close together (the distance of the
            // last one, the same number in every workgroup): a sparse descent would pay for it at every step.
            if (a.probe > 0 && probe_on && ci <= n - 134) {
                int i = ci, j = cj + 1 + tid;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (j >= n) { j = j - n + i + 2; i += 1; }
                const bool act = j < n && i < n - 1;
                bool hit = false, adjp = false;
                double delta = 0.0;
                unsigned pip = 0;
                if (act) {
                    const NodeRec ri = cl_node_v<WT, INT, CT>(coord, order, pos, n, to_int(i), view);
                    const NodeRec rj = cl_node_v<WT, INT, CT>(coord, order, pos, n, to_int(j), view);
                    adjp = rj.id == ri.succ || rj.succ == ri.id;   // heuristics.c:471
                    if (!adjp) {
                        delta = pair_delta<WT, INT>(ri, rj);
                        hit = delta < 0;
                    }
                    pip = ((unsigned)ri.id << 16) | (unsigned)rj.id;
                }
                w_lane += __popcll(__ballot(act));
                w_ex += __popcll(__ballot(act && !adjp));
                const unsigned long long hb = __ballot(hit);
                if (lane == 0) s_k[wave] = hb;
                if (hb && lane == __builtin_ctzll(hb)) { s_d[wave] = delta; s_k[8 + wave] = make_key(i, j); s_ip[wave] = pip; }
                int ei = ci, ej = cj + kClThreads;   // the last thread's pair: where the scan goes on after a probe without a hit
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (ej >= n) { ej = ej - n + ei + 2; ei += 1; }
                __syncthreads();
                {   // every wave reads the eight wave results once (lane w: wave w) and reduces them in registers
                    const int lw = lane & (kClWaves - 1);
                    const unsigned long long h = s_k[lw];
                    const double dw = s_d[lw];
                    const u64 kw = s_k[8 + lw];
                    const unsigned long long hm = __ballot(lane < kClWaves && h != 0ull);
                    probe_hit = hm != 0ull;
                    const int fw = probe_hit ? __builtin_ctzll(hm) : kClWaves;   // first wave with a hit
                    if (probe_hit) {
                        bd = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dw), fw), __builtin_amdgcn_readlane(__double2loint(dw), fw));
                        const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)kw, fw);
                        const unsigned khi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(kw >> 32), fw);
                        key = ((u64)khi << 32) | klo;
                        ipair = (unsigned)__builtin_amdgcn_readlane((int)s_ip[lw], fw);   // the pair inside the replica
                    }
                }
                if (!probe_hit) { si = ei; sj = ej; __syncthreads(); }   // the vote's scratch is rewritten by the arg-min below
            }
            row_lo = si; row_hi = min(si + chunk, n - 1);
        }
        // which scan this step takes (the same answer in every workgroup: it depends on replicated state alone)
        bool do_sorted = SORTED && BEST;
        if constexpr (FS) {
            do_sorted = a.fs_rows > 0 && hit_rows >= a.fs_rows && !probe_hit;
            if (do_sorted) row_hi = n - 1;   // the box-pruned scan decides the whole rest of the sweep
        }
        const u64 startkey = make_key(si, sj);   // FIRST on the sorted replica: pairs up to here are behind the cursor

        if constexpr (SORTED) if (do_sorted) {
            if constexpr (FS) {
                if (gmax_dirty) {   // wave-uniform, the same in every workgroup
                    flush_view();
                    group_bounds(wave, kClWaves, ng);
                    gmax_dirty = false;
                    __syncthreads();
                }
            }
            // ---- sorted scan: box tests on this workgroup's share of the group pairs, then the survivors ------
            const int *tab = a.pairtab + (size_t)c * a.ntests;
            const int P = a.stage_pairs;
            int m0 = 0;
            while (m0 < a.ntests) {
                int kept = 0;
                unsigned long long wave_surv = 0ull;   // tests_per_wave: the surviving entries of tab_reg, by lane
                while (m0 < a.ntests && kept + kClThreads <= kClListCap) {
                    const int m = m0 + tid;
                    const int e = a.ntests <= kClThreads ? tab_reg : (m < a.ntests ? tab[m] : -1);
                    bool surv = false;
                    if (e >= 0) {
                        const int r = e >> 16, cg = e & 0xffff;
                        const double4 rb = gbox[r], cb = gbox[cg];
                        const double gx = fmax(0.0, fmax(rb.x - cb.y, cb.x - rb.y)), gy = fmax(0.0, fmax(rb.z - cb.w, cb.z - rb.w));
                        const double T = gmax[r] + gmax[cg] + prune2 + (BEST ? b0 : 0.0);   // b0: what is known about this sweep before it starts
                        surv = T > 0.0 && gx * gx + gy * gy < (ATT10 ? 10.0 * T * T : T * T);
                        if (a.dbg & 4) surv = true;
                    }
                    if (tests_per_wave) {   // (the list's counter was zeroed behind the arg-min barrier of the step before)
                        wave_surv = __ballot(surv);
                        kept = __popcll(wave_surv);
                        m0 = a.ntests;
                        break;
                    }
                    const unsigned long long bal = __ballot(surv);
                    if (lane == 0) s_wcount[wave] = __popcll(bal);
                    if (tid == 0) *s_nitems = 0;   // for the first chunk of survivors below (its readers are behind two barriers)
                    __syncthreads();
                    int before = 0, total = 0;
#pragma unroll
                    for (int w = 0; w < kClWaves; ++w) { const int cw = s_wcount[w]; before += (w < wave) ? cw : 0; total += cw; }
                    if (surv) s_list[kept + before + __popcll(bal & ((1ull << lane) - 1ull))] = e;
                    kept += total;
                    m0 += kClThreads;
                    __syncthreads();
                }
                CL_T(0);
                // The survivors, P group pairs at a time.  Work is dealt in three grains, because a step costs the time of its
                // slowest wave and the pairs that reach the expensive tiers are concentrated in a few group pairs (a group
                // against itself or its neighbours):
                //  (1) stage: one wave per pair derives the records of both groups (lane = slot; slot block 2 pe = the row group,
                //      2 pe + 1 = the column group), takes the box and the longest successor edge of each QUARTER of the column
                //      group from them, and tests every row against the four quarters: a (row, quarter) pair that can hold a
                //      candidate becomes a unit of the list;
                //  (2) units, one per 16 lanes, two trips per wave and turn with their LDS reads issued together: tier 0 against the
                //      quarter's 16 columns; the pairs that survive it are queued per wave;
                //  (3) queued pairs, 64 at a time (one per lane): tiers 1 and 2 run on full waves, not on the few lanes of
                //      a unit that happen to need them.
                for (int e0 = 0; e0 < kept; e0 += P) {
                    const int ne = min(P, kept - e0);
                    if (wave == 0) w_st += ne * 128;
                    if (e0 > 0) {   // (the first chunk: zeroed beside the box tests, whose second barrier stands between)
                        if (tid == 0) *s_nitems = 0;
                        __syncthreads();
                    }
                    if (wave < ne) {   // one wave per staged pair (ne <= kClMaxStagePairs <= kClWaves): both groups' records, lane = slot
                        const int pe = wave;
                        int e;
                        if (tests_per_wave) {   // the (e0 + pe)-th survivor of this wave's own ballot
                            unsigned long long mm = wave_surv;
                            for (int i = 0; i < e0 + pe; ++i) mm &= mm - 1ull;
                            e = __builtin_amdgcn_readlane(tab_reg, __builtin_ctzll(mm));
                        } else {
                            e = s_list[e0 + pe];
                        }
                        const int rg = e >> 16, cgp = e & 0xffff;
                        const int vr = rg * 64 + lane, vc = cgp * 64 + lane;
                        NodeRec rr, rc;
                        rr.x = rr.y = rr.xs = rr.ys = 1e30; rr.ds = 0.0; rr.succ = -1; rr.id = -1;   // padding: far from everything
                        rc = rr;
                        if (vr < n) rr = cl_node_v<WT, INT, CT>(coord, order, pos, n, vr, view);
                        if (vc < n) rc = cl_node_v<WT, INT, CT>(coord, order, pos, n, vc, view);
                        stage[(2 * pe) * 64 + lane] = SR::pack(rr);
                        stage[(2 * pe + 1) * 64 + lane] = SR::pack(rc);
                        // The columns' four quarters (16 consecutive slots: neighbours on the curve): box and longest successor edge,
                        // from the records just derived (row reductions of 16 lanes).  A row is tested against each quarter -- not
                        // against the whole group's box with the group's persistent bound: 62 % fewer lanes reach tier 0 -- and its
                        // item carries the quarters it can reach.
                        using QT = std::conditional_t<SR::kF32, float, double>;   // float replica: coordinates and lengths are exact floats
                        const bool creal = vc < n;
                        const QT qx0 = cl_row16_min<QT>(creal ? (QT)rc.x : (QT)1e30), qx1 = cl_row16_max<QT>(creal ? (QT)rc.x : (QT)-1e30);
                        const QT qy0 = cl_row16_min<QT>(creal ? (QT)rc.y : (QT)1e30), qy1 = cl_row16_max<QT>(creal ? (QT)rc.y : (QT)-1e30);
                        const QT qdm = cl_row16_max<QT>(creal ? (QT)rc.ds : (QT)0);
                        unsigned qm = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const QT x0 = cl_readlane<QT>(qx0, 16 * k), x1 = cl_readlane<QT>(qx1, 16 * k);
                            const QT y0 = cl_readlane<QT>(qy0, 16 * k), y1 = cl_readlane<QT>(qy1, 16 * k), dm = cl_readlane<QT>(qdm, 16 * k);
                            bool reach;
                            if constexpr (SR::kF32) {   // as tier 0 below: the rounding of the squares paid for in slack, T taken 2 units high
                                const float rx = (float)rr.x, ry = (float)rr.y;
                                const float gx = fmaxf(0.f, fmaxf(x0 - rx, rx - x1)), gy = fmaxf(0.f, fmaxf(y0 - ry, ry - y1));
                                const float T = (float)rr.ds + dm + (float)((BEST ? b0 : 0.0) + prune2 + 2.0);
                                reach = fmaf(gx, gx, gy * gy) < (ATT10 ? 10.0f : 1.0f) * 1.000002f * T * fabsf(T);   // T <= 0: never
                            } else {
                                const double gx = fmax(0.0, fmax(x0 - rr.x, rr.x - x1)), gy = fmax(0.0, fmax(y0 - rr.y, rr.y - y1));
                                const double T = rr.ds + dm + prune2 + (BEST ? b0 : 0.0);
                                reach = T > 0.0 && gx * gx + gy * gy < (ATT10 ? 10.0 * T * T : T * T);
                            }
                            qm |= (reach ? 1u : 0u) << k;
                        }
                        if (a.dbg & 2) qm = vr < n ? 0xfu : 0u;
                        // units: (row, quarter) pairs, quarter-major within this wave's share of the list; a unit is the row's stage
                        // slot (10 bits), the quarter (2), "the pair is a group with itself" (1)
                        const unsigned long long m0 = __ballot(qm & 1u), m1 = __ballot(qm & 2u), m2 = __ballot(qm & 4u), m3 = __ballot(qm & 8u);
                        const int c0 = __popcll(m0), c1 = __popcll(m1), c2 = __popcll(m2), c3 = __popcll(m3);
                        if (c0 + c1 + c2 + c3) {
                            int base = 0;
                            if (lane == 0) base = atomicAdd(s_nitems, c0 + c1 + c2 + c3);
                            base = __builtin_amdgcn_readfirstlane(base);
                            const unsigned long long below = (1ull << lane) - 1ull;
                            const unsigned u = (unsigned)((2 * pe) * 64 + lane) | (rg == cgp ? 1u << 14 : 0u);
                            if (qm & 1u) s_items[base + __popcll(m0 & below)] = (unsigned short)u;
                            if (qm & 2u) s_items[base + c0 + __popcll(m1 & below)] = (unsigned short)(u | (1u << 10));
                            if (qm & 4u) s_items[base + c0 + c1 + __popcll(m2 & below)] = (unsigned short)(u | (2u << 10));
                            if (qm & 8u) s_items[base + c0 + c1 + c2 + __popcll(m3 & below)] = (unsigned short)(u | (3u << 10));
                        }
                    }
                    __syncthreads();
                    CL_T(8);
                    {
                        const int nit = *s_nitems;
#ifdef TSP_STAMPS
                        if (tid == 0) prof[11] += nit;
#endif
                        unsigned *q = s_queue + wave * kClQueue;
                        int qhead = 0, qtail = 0;   // wave-uniform
                        // tiers 1 and 2 for `cnt` queued pairs, one per lane
                        auto dense = [&](int cnt) {
                            bool ok = false;
                            if (lane < cnt) {
                                const unsigned en = q[(qhead + lane) & (kClQueue - 1)];
                                const NodeRec ri = SR::unpack(stage[en & 0xffffu]), rj = SR::unpack(stage[en >> 16]);
                                // tier 1, both new edges, no root: |ab| + |a1 b1| < T2 = bound + d(a,a1) + d(b,b1) + margin
                                //   <=>  w = T2^2 - s1 - s2 > 0 and 4 s1 s2 < w^2   (bound: this lane's best so far)
                                const double dx1 = ri.x - rj.x, dy1 = ri.y - rj.y;
                                const double dx = ri.xs - rj.xs, dy = ri.ys - rj.ys, T2 = ri.ds + (BEST ? bd : 0.0) + rj.ds + a.sum_margin;
                                const double sc = ATT10 ? 0.1 : 1.0;
                                const double p1 = sc * fma(dx1, dx1, dy1 * dy1), p2 = sc * fma(dx, dx, dy * dy);
                                const double w = T2 * T2 - p1 - p2;
                                ok = (T2 > 0.0) & (w > 0.0) & (4.0 * p1 * p2 < w * w) &
                                     (rj.id != ri.succ) & (rj.succ != ri.id);   // never adjacent nodes
                                if (ok) {   // tier 2: the exact delta, lower node id first (tabusearch.c:150 with i < j)
                                    int gi = 0, gj = 0;
                                    double delta;
                                    if constexpr (INT) {
                                        delta = pair_delta<WT, INT>(ri, rj);   // integer-valued terms: the sum is exact in any order
                                    } else {
                                        gi = a.gid[ri.id]; gj = a.gid[rj.id];
                                        delta = gi < gj ? pair_delta<WT, INT>(ri, rj) : pair_delta<WT, INT>(rj, ri);
                                    }
                                    if (BEST ? (delta < bd || (delta == bd && delta < 0.0)) : (delta < 0.0)) {
                                        if constexpr (INT) { if (a.dbg & 8) { gi = ri.id; gj = rj.id; } else { gi = a.gid[ri.id]; gj = a.gid[rj.id]; } }
                                        const u64 kk = make_key(min(gi, gj), max(gi, gj));
                                        // best improvement: arg-min of (delta, pair); first improvement: the first improving pair after the cursor
                                        if (BEST ? (delta < bd || kk < key) : (kk > startkey && kk < key)) {
                                            bool is_tabu = false;
                                            if constexpr (TABU) {   // tabusearch.c:137-149 on node ids, a = the lower one, lazy clears included
                                                const bool lo = gi < gj;
                                                const int i = lo ? gi : gj, jn = lo ? gj : gi;
                                                const int a1 = a.gid[lo ? ri.succ : rj.succ], b1 = a.gid[lo ? rj.succ : ri.succ];
                                                is_tabu = stamp_is_tabu(a.tabu + udir_pos(i, jn, n), cur_iter, cur_ten) ||
                                                          stamp_is_tabu(a.tabu + udir_pos(i, a1, n), cur_iter, cur_ten) ||
                                                          stamp_is_tabu(a.tabu + udir_pos(jn, b1, n), cur_iter, cur_ten) ||
                                                          stamp_is_tabu(a.tabu + udir_pos(i, b1, n), cur_iter, cur_ten);
                                            }
                                            if (!is_tabu) {
                                                bd = delta; key = kk;
                                                ipair = gi < gj ? (((unsigned)ri.id << 16) | (unsigned)rj.id)
                                                                : (((unsigned)rj.id << 16) | (unsigned)ri.id);
                                            }
                                        }
                                    }
                                }
                            }
                            qhead += cnt;
                            w_ex += __popcll(__ballot(ok));
#ifdef TSP_STAMPS
                            if (tid == 0) { prof[6] += 1; prof[7] += cnt; }
#endif
                        };
                        // The units, one per 16 lanes and trip, TR trips per wave and turn with their LDS reads issued together (2 measured
                        // best: 3 / 4 / 6 / 8 trips cost +0.5 / +1.5 / +2 / +5 % of a descent), turns dealt to the waves in turn
#ifndef TSP_CL_TR
#define TSP_CL_TR 2
#endif
                        constexpr int TR = TSP_CL_TR;
                        const int g = lane >> 4, l15 = lane & 15;
                        for (int u0 = wave * 4 * TR; u0 < nit; u0 += kClWaves * 4 * TR) {
                            const int uend = min(nit, u0 + 4 * TR);
                            w_lane += 16 * (uend - u0);
                            int un[TR];
#pragma unroll
                            for (int t = 0; t < TR; ++t) {
                                const int ui = u0 + 4 * t + g;
                                un[t] = ui < uend ? (int)s_items[ui] : -1;
                            }
                            int ridx[TR], cidx[TR];
                            bool need[TR];
#pragma unroll
                            for (int t = 0; t < TR; ++t) {
                                const int uu = max(un[t], 0);
                                ridx[t] = uu & 1023;
                                cidx[t] = (ridx[t] & ~127) + 64 + (((uu >> 10) & 3) << 4) + l15;
                            }
                            // tier 0, the new edge (a, b) alone: |ab| < bound + d(a,a1) + d(b,b1) + margin
                            if constexpr (SR::kF32) {
                                // integer coordinates (exact as floats): a float test with its rounding paid for in slack -- s may
                                // come out low by 2^-22 relative, T (< 2^23) is taken 2 units high; half the issue cost of fp64
                                float rx[TR], ry[TR], rd[TR], cxf[TR], cyf[TR], cdf[TR];
#pragma unroll
                                for (int t = 0; t < TR; ++t) {
                                    SR::xydf(stage[ridx[t]], rx[t], ry[t], rd[t]);
                                    SR::xydf(stage[cidx[t]], cxf[t], cyf[t], cdf[t]);
                                }
                                const float bf = (float)((BEST ? bd : 0.0) + prune2 + 2.0);
#pragma unroll
                                for (int t = 0; t < TR; ++t) {
                                    const float dx = rx[t] - cxf[t], dy = ry[t] - cyf[t], T = rd[t] + cdf[t] + bf;
                                    need[t] = fmaf(dx, dx, dy * dy) < (ATT10 ? 10.0f : 1.0f) * 1.000002f * T * fabsf(T);   // T <= 0: never
                                }
                            } else {
                                double rx[TR], ry[TR], rd[TR], cx[TR], cy[TR], cd[TR];
#pragma unroll
                                for (int t = 0; t < TR; ++t) {
                                    SR::xyd(stage[ridx[t]], rx[t], ry[t], rd[t]);
                                    SR::xyd(stage[cidx[t]], cx[t], cy[t], cd[t]);
                                }
#pragma unroll
                                for (int t = 0; t < TR; ++t) {
                                    const double dx = rx[t] - cx[t], dy = ry[t] - cy[t], T = rd[t] + cd[t] + (BEST ? bd : 0.0) + prune2;
                                    need[t] = fma(dx, dx, dy * dy) < (ATT10 ? 10.0 : 1.0) * T * fabs(T);
                                }
                            }
#pragma unroll
                            for (int t = 0; t < TR; ++t) {
                                // one slot pair once (inside a group: row slot below column slot); padding columns lie 1e30 away
                                const bool nd = need[t] & (un[t] >= 0) & (!((un[t] >> 14) & 1) | ((ridx[t] & 63) < (cidx[t] & 63)));
                                const unsigned long long m = __ballot(nd);
                                if (m) {
                                    if (nd) q[(qtail + __popcll(m & ((1ull << lane) - 1ull))) & (kClQueue - 1)] = (unsigned)ridx[t] | ((unsigned)cidx[t] << 16);
                                    qtail += __popcll(m);
                                    w_t1 += __popcll(m);
                                    if (qtail - qhead >= 64) dense(64);
                                }
                            }
                        }
                        if (qtail > qhead) dense(qtail - qhead);
                    }
                    CL_T(10);
                    // the stage (and, after a round's last chunk, the list) is rewritten next -- not after the last chunk of the last
                    // round: what follows is the arg-min, whose per-wave slots nobody is reading
                    if (!(e0 + P >= kept && m0 >= a.ntests)) __syncthreads();
                }
            }
        }
        if constexpr (!SORTED || FS) if (!do_sorted) {
            // ---- tiles: rpt rows x 512 columns, tile t = row block * nb + column batch, dealt round-robin.  A step costs
            // the latency of its slowest workgroup, so the rows per tile shrink until every workgroup of the cluster has a
            // tile (a first-improvement step right after a move scans a few dozen rows: 20 tiles of 32 rows would leave
            // 236 of 256 workgroups idle and make the 20 busy ones eight times slower than they need be)
            const int nb = (n + kClThreads - 1) / kClThreads;
            const int nrows = row_hi - row_lo;
            const int rbs = a.rbs;    // C / nb: row blocks that give every workgroup at most one tile
            const int rpt = rbs > 0 ? max(1, min(kClRows, cl_div(nrows + rbs - 1, rbs))) : kClRows;
            const int nrb = cl_div(nrows + rpt - 1, rpt);
            int hit_rb = nrb;   // FIRST: first row block in which this workgroup has found an improving pair
            bool any_hit = false;
            for (int t = c; t < nrb * nb && !probe_hit; t += C) {
                const int rbi = cl_div(t, nb), b = t - rbi * nb;
                if (!BEST && rbi > hit_rb) break;   // later rows only hold later pairs
                const int rb = row_lo + rbi * rpt;
                const int nr = min(rpt, row_hi - rb);
                const int j = b * kClThreads + tid;
                // no column of this batch above the first row (or, in the cursor's row alone, above the cursor)
                const int jmin = (!BEST && nr == 1 && rb == si) ? max(rb, sj) : rb;
                if (b * kClThreads + kClThreads - 1 <= jmin) continue;
                CL_T(8);
                // the row records of the tile before are still being read by slower waves -- not for a step's first tile: the last
                // readers of s_rows finished before the arg-min barrier of the step before
                if (t != c) __syncthreads();
                constexpr bool F32T0 = std::is_same<CT, float2>::value && has_root_filter<WT>();
                if (tid < nr) {
                    const NodeRec rr = cl_node_v<WT, INT, CT>(coord, order, pos, n, to_int(rb + tid), view);
                    s_rows[tid] = rr;
                    if constexpr (F32T0) s_rowsf[tid] = make_float4((float)rr.x, (float)rr.y, (float)rr.ds, 0.f);
                }
                NodeRec rj;
                const bool act = j < n && j > rb;
                if (act) rj = cl_node_v<WT, INT, CT>(coord, order, pos, n, to_int(j), view);
                __syncthreads();
                CL_T(9);
#ifdef TSP_STAMPS
                if (tid == 0) prof[11] += 1;
#endif
                w_lane += (long long)nr * __popcll(__ballot(act));
                if constexpr (F32T0) {
                    // Integer coordinates (exact as floats): the new-edge bound in fp32 FIRST, its rounding paid for in slack (s may
                    // come out 2^-22 low, T -- below 2^23 -- is taken 2 units high, as in the sorted scan), before anything else is
                    // looked at: on a constructed tour it sends > 99 % of the pairs home after six fp32 operations, and the row's
                    // full record, the adjacency / cursor / key tests and the fp64 tiers are for the survivors only.  A sweep that
                    // finds nothing (every first-improvement descent ends with one, HEU_VNS runs five or six per round) is this loop.
                    const float cxf = act ? (float)rj.x : 0.f, cyf = act ? (float)rj.y : 0.f, cdf = act ? (float)rj.ds : 0.f;
#ifndef TSP_CL_RQ
#define TSP_CL_RQ 4
#endif
                    constexpr int RQ = TSP_CL_RQ;   // rows per trip: their LDS reads in flight together, one vote
                    for (int r0 = 0; r0 < nr; r0 += RQ) {
                        float4 rf[RQ];
                        bool need[RQ];
#pragma unroll
                        for (int u = 0; u < RQ; ++u) rf[u] = s_rowsf[min(r0 + u, nr - 1)];
                        const float bf = (float)((BEST ? bd : 0.0) + prune2 + 2.0);
                        bool any = false;
#pragma unroll
                        for (int u = 0; u < RQ; ++u) {
                            const float dx = rf[u].x - cxf, dy = rf[u].y - cyf, T = rf[u].z + cdf + bf;
                            need[u] = act && r0 + u < nr && fmaf(dx, dx, dy * dy) < (ATT10 ? 10.0f : 1.0f) * 1.000002f * T * fabsf(T);   // T <= 0: never
                            any = any || need[u];
                        }
                        if (!__any(any)) continue;
#pragma unroll
                        for (int u = 0; u < RQ; ++u) {
                            const unsigned long long nm = __ballot(need[u]);
                            if (!nm) continue;
                            w_t1 += __popcll(nm);
                            bool ok = false;
                            const int i = rb + r0 + u;
                            const NodeRec ri = s_rows[r0 + u];
                            const u64 kq = make_key(i, j);
                            if (need[u]) {
                                ok = j > i && rj.id != ri.succ && rj.succ != ri.id;   // heuristics.c:471 / tabusearch.c:134
                                if constexpr (!BEST) ok = ok && (i > si || j > sj) && kq < key;
                                const double bound = BEST ? bd : 0.0;
                                ok = ok && new_edge_can_improve<WT>(ri.x, ri.y, rj.x, rj.y, bound + ri.ds + rj.ds + prune2);
                                if (ok) {
                                    const double lower = pair_delta_approx<WT>(ri, rj) - a.margin;
                                    ok = BEST ? lower <= bound : lower < bound;
                                }
                            }
                            w_ex += __popcll(__ballot(ok));
                            if (ok) {
                                const double delta = pair_delta<WT, INT>(ri, rj);
                                if constexpr (!BEST) {
                                    if (delta < 0) { bd = delta; key = kq; ipair = ((unsigned)ri.id << 16) | (unsigned)rj.id; }
                                } else {
                                    if (better(delta, kq, bd, key)) { bd = delta; key = kq; ipair = ((unsigned)ri.id << 16) | (unsigned)rj.id; }
                                }
                            }
                        }
                    }
                } else {
                    for (int r = 0; r < nr; ++r) {
                        const int i = rb + r;
                        const NodeRec ri = s_rows[r];
                        bool ok = act && j > i && rj.id != ri.succ && rj.succ != ri.id;   // heuristics.c:471 / tabusearch.c:134
                        if constexpr (!BEST) ok = ok && (i > si || j > sj);
                        const u64 kq = make_key(i, j);
                        if constexpr (!BEST) ok = ok && kq < key;
                        if constexpr (has_root_filter<WT>()) {
                            const double bound = BEST ? bd : 0.0;
                            ok = ok && new_edge_can_improve<WT>(ri.x, ri.y, rj.x, rj.y, bound + ri.ds + rj.ds + prune2);
                            w_t1 += __popcll(__ballot(ok));
                            if (ok) {
                                const double lower = pair_delta_approx<WT>(ri, rj) - a.margin;
                                ok = BEST ? lower <= bound : lower < bound;
                            }
                        }
                        w_ex += __popcll(__ballot(ok));
                        if (ok) {
                            const double delta = pair_delta<WT, INT>(ri, rj);
                            if constexpr (!BEST) {
                                if (delta < 0) { bd = delta; key = kq; ipair = ((unsigned)ri.id << 16) | (unsigned)rj.id; }
                            } else {
                                if (better(delta, kq, bd, key)) { bd = delta; key = kq; ipair = ((unsigned)ri.id << 16) | (unsigned)rj.id; }
                            }
                        }
                    }
                }
                CL_T(10);
                if constexpr (!BEST) {
                    // has anybody here found an improving pair?  Then this workgroup's later row blocks are skipped.  Asked only when
                    // there is a later tile, and through one flag word and one barrier (a stamp no earlier vote has used: no zeroing)
                    if (!any_hit && t + C < nrb * nb) {
                        const int stamp = ++vote_seq;
                        const bool anyh = __any(key != kNoKey);
                        if (lane == 0 && anyh) *s_vote = stamp;
                        __syncthreads();
                        if (*s_vote == stamp) { any_hit = true; hit_rb = rbi; }
                    }
                }
            }
        }

        CL_T(1);
        // ---- the workgroup's candidate, the cluster's winner -----------------------------------------------------
        // wave arg-min (the winner's internal pair rides along: a pair is evaluated by exactly one lane), the eight wave
        // winners through LDS to the first wave, which reduces them, runs the exchange and hands the result back
        ClCand xcd{0.0, kNoKey, 0u};   // wave 0: the exchange's result with this lane's share of the candidates
        if (!probe_hit) {
            const u64 mykey = key;
            wave_argmin<BEST>(bd, key);
            const unsigned long long owners = __ballot(mykey == key && key != kNoKey);
            if (owners) ipair = (unsigned)__builtin_amdgcn_readlane((int)ipair, __builtin_ctzll(owners));
            if (lane == 0) { s_d[wave] = bd; s_k[wave] = key; s_ip[wave] = ipair; }
        }
        if (!probe_hit) __syncthreads();
        if (tests_per_wave && tid == kClThreads - 1) *s_nitems = 0;   // the next sorted scan's unit counter (its last readers are behind the barrier above, its next writers behind the exchange's)
        CL_T(2);
        if (wave == 0 && !probe_hit) {
            double d = 0.0;
            u64 k2 = kNoKey;
            unsigned ip = 0;
            if (lane < kClWaves) { d = s_d[lane]; k2 = s_k[lane]; ip = s_ip[lane]; }
            const u64 mine = k2;
            wave_argmin<BEST>(d, k2);
            const unsigned long long owners = __ballot(mine == k2 && k2 != kNoKey);
            unsigned wip = 0;
            if (owners) wip = (unsigned)__builtin_amdgcn_readlane((int)ip, __builtin_ctzll(owners));
            ClCand cd{d, k2, wip};
            bool okx = true;
#ifdef TSP_STAMPS
            const unsigned long long tx0 = wall_clock64();
#endif
            if (C > 1) okx = cl_exchange<BEST, SORTED, kSmallD>(area, C, c, ++xep, cd, a.err, a.spin_limit, a.spin_ticks, a.copies, mycopy);
#ifdef TSP_STAMPS
            if (tid == 0 && tour == 0 && c < 256) g_cl_tail[c][10] += wall_clock64() - tx0;
#endif
            if (lane == 0) { *s_win_d = cd.d; *s_win_k = cd.key; *s_win_ip = cd.ipair; *s_fail = okx ? 0 : 1; }
#ifdef TSP_STAMPS
            if constexpr (SORTED && BEST) {
                if (lane == 0 && c == 0 && tour == 0 && cd.key != kNoKey) {   // how good the bound this sweep started from was
                    atomicAdd(&g_cl_b0[0], 1ull);
                    atomicAdd(&g_cl_b0[1], (unsigned long long)(-b0));
                    atomicAdd(&g_cl_b0[2], (unsigned long long)(-cd.d));
                    if (b0 == cd.d) atomicAdd(&g_cl_b0[3], 1ull);
                    if (-b0 * 2.0 >= -cd.d) atomicAdd(&g_cl_b0[4], 1ull);
                    if (-b0 * 1.25 >= -cd.d) atomicAdd(&g_cl_b0[5], 1ull);
                }
            }
#endif
            xcd = cd;
        } else if (!probe_hit && view.L != 0) {
            swaps(view, tid - 64, kClThreads - 64);   // the pending reversal, while wave 0 exchanges
        }
        if constexpr (TABU) {
            if (wave != 0) {
                // ---- side effects of the reference's scan of the tabu list (two_opt_tabu_list.hpp, tabu_side), by the seven
                // waves that would otherwise wait for the first one's exchange: thread t of them, cluster-wide, takes the list
                // entries t, t + C x 448, ...; succ / pred come from the replica (internal ids; the stamps are indexed by
                // node ids).  The move of this sweep is applied after the barrier that follows.
                auto nsucc = [&](int v) { int p2 = (int)pos[a.iid[v]] + 1; if (p2 == n) p2 = 0; return a.gid[(int)order[p2]]; };
                auto npred = [&](int v) { int p2 = (int)pos[a.iid[v]]; p2 = p2 == 0 ? n - 1 : p2 - 1; return a.gid[(int)order[p2]]; };
                auto live_v = [&](int sv) { return sv != 0 && !(cur_iter - sv > cur_ten); };
                auto live = [&](int x, int y) { return live_v(a.tabu[udir_pos(x, y, n)]); };
                // One more live tour edge of this sweep (the C(|F|, 2) term is cluster-wide: the first workgroup reads the count after
                // the NEXT exchange).  The add must have been performed before this workgroup publishes its next candidate: a
                // RETURNING atomic whose result the wave waits for -- the barrier behind the exchange comes after it, the publish
                // after the barrier.  (This was atomicAdd + __threadfence(): the fence writes this XCD's L2 back and invalidates it,
                // 1.5 - 2 us per live tour edge and sweep, and stamped edges that a move's unchecked second new edge brings back
                // while their stamp is live are common -- that fence, not the loads, was most of what the list cost a sweep.)
                auto count_live_edge = [&]() {
                    const unsigned long long was = atomicAdd(a.tabu_side + 1 + slot_cur, 1ull);
                    asm volatile("" ::"v"((unsigned)was));
                };
#ifdef TSP_STAMPS
                const unsigned long long ts0 = wall_clock64();
#endif
                if (c == 0 && tid == 64) {
                    // The C(|F|, 2) term of the sweep TWO back (F = its live tour edges, counted cluster-wide in that sweep's slot):
                    // every workgroup's adds of that sweep came before the candidate it published for the sweep after it, and that
                    // exchange is complete.  Read (and zeroed) here, beside the first wave's exchange, where it costs the step
                    // nothing -- as the first thread's act behind the exchange it was 0.6 us of every sweep on the workgroup the
                    // cluster waits for.  Four slots in turn: the slot is next added to two sweeps on, by workgroups that have
                    // seen this workgroup's next candidate, which it publishes behind the barrier this wave is still in front of.
                    unsigned long long *fp = a.tabu_side + 1 + slot_read;
                    const long long f = (long long)atomicAdd(fp, 0ull);
                    if (f) {
                        tabu_cnt -= f * (f - 1) / 2;
                        const unsigned long long was = atomicExch(fp, 0ull);
                        asm volatile("" ::"v"((unsigned)was));
                    }
                }
                const int m = min(tl_m, a.tabu_list_cap);
                // The entries go round the workgroups 1 .. C - 1, one per thread: the first workgroup, which reads the live-edge count
                // and (in a chain of iterations) decides the kicks, is the one the cluster waits for in every sweep -- its exchange
                // shares the CU's memory path with whatever its other waves load (rand10000: 137.6 -> 131.3 us per iteration)
                const bool spread = C > 1 && !(a.dbg & 64);
                const int kstep = spread ? (C - 1) * (kClThreads - 64) : C * (kClThreads - 64);
                const int k0 = spread ? (c == 0 ? m : (c - 1) + (C - 1) * (tid - 64)) : c * (kClThreads - 64) + tid - 64;
                for (int k = k0; k < m; k += kstep) {
                    // An entry is a chain of dependent reads -- the entry, its stamp, the node maps, the neighbours' stamps -- and the
                    // whole cluster waits for the longest one (this loop, not the exchange beside it, was 7 of a sweep's 10 us
                    // with a list of a few hundred live stamps).  So: a thread keeps its first entry, that entry's ids inside the
                    // replica in registers from sweep to sweep (a list only grows inside a launch: entries never move), the
                    // entry's stamp and the ids of its four tour neighbours are fetched together, and then all six stamps the
                    // accounting can ask for (live() has no side effect: reading one the reference's && chain would not have
                    // reached changes nothing) -- two round trips to memory instead of six to eight.  What is left of them costs
                    // 1.5 us of a sweep (the workgroup that publishes last has only the rest of the exchange to hide its entry
                    // behind).  Measured and not kept: the entry's last contribution remembered with the neighbours it was
                    // computed from (no load at all in most sweeps: +6 us per iteration); the first round trip issued before
                    // the scan (+3 us: every wait for a load in the scan then waits for it too); the first workgroup left out of
                    // the group-pair table (no difference).
                    const bool mine = k == k0 && !(a.dbg & 128);
                    if (mine && tl_state == 1) {
#ifdef TSP_STAMPS
                        ++tl_n_skip;
#endif
                        continue;
                    }
                    int2 e;
                    int iu, iv;
                    if (mine && tl_have) { e = tl_e; iu = tl_iu; iv = tl_iv; }
                    else {
                        e = a.tabu_list[k];
                        iu = a.iid[e.x]; iv = a.iid[e.y];
                        if (mine) { tl_e = e; tl_iu = iu; tl_iv = iv; tl_have = true; }
                    }
                    const int u = e.x, v = e.y;
                    int *sp = a.tabu + udir_pos(u, v, n);
                    const int pu = (int)pos[iu], pv = (int)pos[iv];
                    const int su_i = (int)order[pu + 1 == n ? 0 : pu + 1], s2_i = (int)order[pv + 1 == n ? 0 : pv + 1];
                    const int pru_i = (int)order[pu == 0 ? n - 1 : pu - 1], prv_i = (int)order[pv == 0 ? n - 1 : pv - 1];
                    const unsigned nb0 = ((unsigned)su_i << 16) | (unsigned)s2_i, nb1 = ((unsigned)pru_i << 16) | (unsigned)prv_i;
                    if (mine && tl_state == 2 && nb0 == tl_nb0 && nb1 == tl_nb1) {
                        tabu_cnt += tl_cnt;
                        if (tl_edge) count_live_edge();
#ifdef TSP_STAMPS
                        ++tl_n_hit;
#endif
                        continue;
                    }
#ifdef TSP_STAMPS
                    ++tl_n_full;
#endif
                    const int sv = *sp;
                    const int su = a.gid[su_i], s2 = a.gid[s2_i], pru = a.gid[pru_i], prv = a.gid[prv_i];
                    if (sv == 0) { if (mine) tl_state = 1; continue; }
                    const bool uv = su == v, vu = s2 == u;
                    if (!live_v(sv)) {
                        if (mine) tl_state = (!uv && !vu) ? 1 : 0;
                        if (!uv && !vu) *sp = 0;
                        else {
                            // expired stamp on the tour edge x -> y: cleared iff the reference's chain reaches it
                            const int x = uv ? u : v, y = uv ? v : u, px = npred(x);
                            bool looked = false;
                            for (int b = x + 1; b < n && !looked; ++b) looked = b != y && b != px && !live(x, b);
                            for (int q = 0; q < x && !looked; ++q) looked = q != px && q != y && !live(q, x) && !live(q, nsucc(q));
                            if (!looked) looked = y < px && px != nsucc(y) && !live(y, px) && !live(y, nsucc(y)) && !live(px, x);
                            if (looked) { *sp = 0; if (mine) tl_state = 1; }
                        }
                        continue;
                    }
                    // the stamps of (u, succ u), (v, succ v) and, per orientation, (aa, pred w), (pred w, w): fetched together
                    const int s_usu = a.tabu[udir_pos(u, su, n)], s_vs2 = a.tabu[udir_pos(v, s2, n)];
                    const int s_uprv = prv != u ? a.tabu[udir_pos(u, prv, n)] : 0, s_prvv = a.tabu[udir_pos(prv, v, n)];
                    const int s_vpru = pru != v ? a.tabu[udir_pos(v, pru, n)] : 0, s_pruu = a.tabu[udir_pos(pru, u, n)];
                    const bool l_usu = live_v(s_usu), l_vs2 = live_v(s_vs2);
                    int add = 0;
                    bool edge = false;
                    if (!uv && !vu) {
                        const int fu = l_usu ? 1 : 0, fv = l_vs2 ? 1 : 0;
                        add += 1 - fu - fv + fu * fv;
                    } else {
                        add += n - 3;
                        if (uv ? l_vs2 : l_usu) add += 1;   // live(y, succ y), y = the edge's head
                        edge = true;
                    }
                    // o = 0: aa = u, w = v, b = pred v;  o = 1: aa = v, w = u, b = pred u
                    if (prv != u && u < prv && prv != su && !live_v(s_uprv) && !l_usu && !live_v(s_prvv)) add += 1;
                    if (pru != v && v < pru && pru != s2 && !live_v(s_vpru) && !l_vs2 && !live_v(s_pruu)) add += 1;
                    tabu_cnt += add;
                    if (edge) {
                        // live tour edges of this sweep, cluster-wide (the C(|F|, 2) term): complete before this workgroup
                        // publishes its NEXT candidate; the first workgroup reads the count after that exchange
                        count_live_edge();
                    }
                    if (mine) { tl_state = 2; tl_nb0 = nb0; tl_nb1 = nb1; tl_cnt = add; tl_edge = edge; }
                }
#ifdef TSP_STAMPS
                if (tid == 64 && tour == 0 && c < 256) g_cl_tail[c][8] += wall_clock64() - ts0;
#endif
            }
        }
        if (!probe_hit) {
            __syncthreads();
            if (*s_fail) { failed = true; break; }
            bd = *s_win_d; key = *s_win_k; ipair = *s_win_ip;   // rewritten after the barriers of the move below
            view.L = 0;   // carried out by waves 1 .. 7 before this barrier (an exchange step with a pending reversal)
        }
        CL_T(3);
        const bool found = key != kNoKey && (!BEST || bd < 0);
        const int wi = found ? (int)(ipair >> 16) : -1, wj = found ? (int)(ipair & 0xffffu) : -1;   // ids inside the replica
        const int fi = found ? key_i(key) : -1, fj = found ? key_j(key) : -1;                       // the caller's ids of the same pair

        // ---- reference-equivalent evaluation count (FIRST; kept by the cluster's first workgroup) ----------------
        // evals = pairs between the old and the new cursor MINUS the adjacent ones among them (heuristics.c:471 skips those).
        // The adjacent pairs are not counted step by step (that was a block-wide pass over the rows of the step by the
        // first workgroup, for which the whole cluster then waited at the next exchange): over the steps of a sweep the
        // counts telescope.  With A_t(K) = tour edges {u, v} whose pair key is <= K at step t, a step from cursor lo to hi
        // skips A_t(hi) - A_t(lo) pairs, the next step starts at lo' = hi, and a move only changes four edges, so
        //     sum over the steps = A_now(cursor) - A(cursor at the start) - sum over the moves of d_t,
        //     d_t = [(i,j) <= hi] + [(a1,b1) <= hi] - [(i,a1) <= hi] - [(j,b1) <= hi],  hi = (i,j) the move's pair:
        // O(1) per move (adjD), one pass over the tour per launch (A_now) instead of one per step.
        int ni = fi, nj = fj;
        if constexpr (!BEST) {
            if (!found) { ni = row_hi - 1; nj = n - 1; }
        }

        CL_T(4);
        // ---- move: reverse positions pa+1 .. pb (cyclic), src/utility.c:708-717 ------------------------------------
        int Lr = 0;
        bool deferred_now = false;
        if (found) {
            flush_view();   // a probe step: the reversal an exchange step left pending comes first (no-op after an exchange step)
            const int pa = pos[wi], pb = pos[wj];
            if constexpr (!BEST) {
                if (a.count_evals && c == 0) {   // d_t of this move, on the tour as it is before the move
                    const int a1e = to_ext((int)order[pa + 1 == n ? 0 : pa + 1]), b1e = to_ext((int)order[pb + 1 == n ? 0 : pb + 1]);
                    const u64 hi = make_key(fi, fj);
                    auto le = [&](int u, int v) { return make_key(min(u, v), max(u, v)) <= hi ? 1 : 0; };
                    adjD += 1 + le(a1e, b1e) - le(fi, a1e) - le(fj, b1e);
                }
            }
            if constexpr (SORTED && BEST) {
                if (a.use_b0 && wave == 0 && C > 1) {   // the bound the next sweep starts from (the exchanging wave holds the candidates)
                    if (a.dbg & 16) { for (int q = 0; q < 4; ++q) b0_mem_d[q] = 0.0; }   // diagnostics: no memory of earlier sweeps
                    const double nb0 = cl_next_bound(xcd, pos, n, (INT && !TABU) && !(a.dbg & 32), b0_mem_ip, b0_mem_d);
                    if (lane == 0) *s_b0 = nb0;
                }
            }
            if constexpr (FS) { if (!do_sorted) gmax_dirty = true; }
            if constexpr (SORTED) if (!gmax_dirty) {
                // The move changes the incident edges of four nodes only -- a: (a, a1) becomes (a, b); a1: (a, a1) becomes (a1, b1);
                // b: (b, b1) becomes (a, b); b1: (b, b1) becomes (a1, b1) -- every other node keeps its two tour neighbours whatever
                // the orientation.  So the bounds of their four groups are rebuilt from the OLD tour with those four edges patched,
                // beside the reads of pa / pb and before the swaps: no barrier between the swaps and a rebuild.
                if (a.dbg & 1) gmax_dirty = true;   // diagnostics: every bound rebuilt after the swaps (below)
                else patch_bounds(wi, wj, pa, pb);
            }
            Lr = pb - pa; if (Lr < 0) Lr += n;
            const bool dbg_rebuild = SORTED && BEST && gmax_dirty;
            if (defer_on && !probe_hit && !dbg_rebuild) {
                view.pa = pa; view.pb = pb; view.L = Lr;   // carried out during the next step's exchange
                deferred_now = true;
            } else {
                __syncthreads();   // everyone has read pa / pb (and finished the bound rebuild)
                ClView now{pa, pb, Lr};
                swaps(now, tid, kClThreads);
                if constexpr (SORTED && BEST) {
                    if (gmax_dirty) {   // TSP_CLUSTER_DEBUG & 1: every bound from the new tour
                        __syncthreads();
                        group_bounds(wave, kClWaves, ng);
                        gmax_dirty = false;
                    }
                }
            }
        }
        // End of the step.  Nothing was written since the exchange's barrier when the move was deferred on the plain replica (no
        // swaps, no group bounds, no bound for the next sweep): the next step's first shared writes are the probe's / the arg-min's
        // per-wave slots and, behind a barrier of their own, the row records -- none of which this step still reads.
        if (!(deferred_now && !SORTED)) __syncthreads();
        if constexpr (SORTED && BEST) b0 = (a.use_b0 && found && C > 1) ? *s_b0 : 0.0;

        CL_T(5);
        // ---- control block ------------------------------------------------------------------------------------------
        steps += 1;
        if constexpr (BEST) {
            sweeps += 1;
            evals += (long long)n * (n - 1) / 2 - n;
            scanned += (long long)n * (n - 1) / 2;
            bool have_run = false;   // a chain of iterations on integer costs: every workgroup keeps the cost as it goes (exact)
            if constexpr (TABU && INT) have_run = a.chain_n > 0;
            if (found) { moves += 1; reversed += Lr - 1; run_obj += bd; }
            else {
                done = 1;
                // recomputed cost in node order (tabusearch.c:168-172); only the first workgroup reports it
                if (have_run) obj = run_obj;
                else if (c == 0) {
                    if constexpr (INT || WT == WT_CEIL_2D) {
                        double cc = 0.0;
                        for (int v = tid; v < n; v += kClThreads) cc += cl_node<WT, INT, CT>(coord, order, pos, n, v).ds;
                        obj = block_sum<double>(cc, s_d);
                    } else {
                        double acc = 0.0;
                        for (int base = 0; base < n; base += 64) {   // sequential order, 64 edges at a time
                            __syncthreads();
                            if (tid < 64 && base + tid < n) {
                                int v = base + tid;
                                if constexpr (SORTED) v = a.iid[v];
                                s_chunk[tid] = cl_node<WT, INT, CT>(coord, order, pos, n, v).ds;
                            }
                            __syncthreads();
                            const int m = min(64, n - base);
                            for (int t = 0; t < m; ++t) acc += s_chunk[t];
                        }
                        obj = acc;
                    }
                }
            }
            if constexpr (TABU) {
                if (a.chain_n > 0 && done) {   // (the same in every workgroup: `done` follows from the exchange's result)
                    // ---- the tail of iteration ck of tabu() (src/tabusearch.c:241-309), inside the launch ----------------------
                    // What k_tabu_post_chain does between two launches of a queued chain: the incumbent (the first workgroup keeps
                    // it), the first trial of the kick with the host-drawn nodes, and the next descent starts on the kicked replica
                    // -- no write-back, no replica load, no kernel boundary.  The first workgroup decides the trial (reading the
                    // stamps; its lazy clears are those of check_tenure, :83-92) and appends the two removed edges to the list;
                    // its decision travels through one more exchange, and EVERY workgroup then carries the 2-exchange out on its
                    // replica and stamps the two edges itself.
                    // Memory between XCDs: the stamps are read and lazily cleared with plain loads and stores through the L2 of the
                    // workgroup's XCD.  Inside one iteration a stale value is harmless (an expired stamp reads as expired or as 0:
                    // iter and tenure are constant), across iterations it is not (the tenure changes).  So at this boundary every
                    // workgroup writes its XCD's dirty lines back BEFORE it publishes (release), and drops what it holds AFTER the
                    // exchange (acquire): whatever was cleared in this iteration is in memory before anybody reads a stamp for the
                    // next one, and the stamps of the kick are written by each workgroup into its own L2 (the same value by all) and
                    // written back at once when the next iteration's tenure is 0 (see there).
                    // Measured and not kept: the first trial's inputs (successors, the four stamps, the list look-up) read ahead by an
                    // idle wave of the first workgroup during every sweep's exchange -- its decision 4.8 -> 1.7 us, the iteration
                    // unchanged (111.2 us with two-sided fences at the boundary, 108.4 with the one-sided ones): the boundary's exchange,
                    // behind 256 simultaneous L2 write-backs, is what the tail waits for.
                    const int npairs = a.chain_pairs > 0 ? a.chain_pairs : a.chain_n;
                    int *res = a.chain + 4 + 10 * ck;
                    int *s_kick = reinterpret_cast<int *>(s_chunk);   // {accepted, have0, have1, a1, b1, entries appended, pair taken}: the cost's chunks are through
                    int better_inc = 0, trials = 0;
                    __syncthreads();
#ifdef TSP_STAMPS
                    unsigned long long tt[7]; tt[0] = wall_clock64();
#define CL_TT(k) do { if (tid == 0) tt[k] = wall_clock64(); } while (0)
#else
#define CL_TT(k) do { } while (0)
#endif
                    if (c == 0) {
                        if (obj < inc_best) {
                            better_inc = 1; inc_best = obj;
                            for (int p = tid; p < n; p += kClThreads) a.snap[p] = a.gid[(int)order[p]];
                        }
                        int2 *list = const_cast<int2 *>(a.tabu_list);
                        if (kpp >= npairs) {   // the earlier iterations' further trials have used the pairs up: no trial, the caller draws on
                            if (tid == 0) { s_kick[0] = 0; s_kick[3] = 0; s_kick[4] = 0; s_kick[5] = 0; s_kick[6] = kpp; }
                            __syncthreads();
                        } else
                        for (int pp = kpp;; ++pp) {   // the trials of this iteration's kick, in the order tabu() draws them
                            const int4 pr = pp == nx_pp ? nx_pair : reinterpret_cast<const int4 *>(a.chain_ab)[pp];
                            const int ka = pr.x, kb = pr.y, ia = pr.z, ib = pr.w;   // (ia, ib: the same two nodes inside the replica)
                            if (tid == 0) {
                                const int pa = (int)pos[ia], pb = (int)pos[ib];
                                s_kick[3] = a.gid[(int)order[pa + 1 == n ? 0 : pa + 1]]; s_kick[4] = a.gid[(int)order[pb + 1 == n ? 0 : pb + 1]];
                                s_kick[1] = 0; s_kick[2] = 0; s_kick[5] = 0; s_kick[6] = pp;
                            }
                            __syncthreads();
                            const int a1 = s_kick[3], b1 = s_kick[4];
                            // the two edges a kick would stamp, looked up in the list of non-zero stamps by everybody (tabu_kick_body)
                            // while the first thread decides the trial: both are one round trip to memory
                            const int e0x = min(ka, a1), e0y = max(ka, a1), e1x = min(kb, b1), e1y = max(kb, b1);
                            if (tid == 0) {
                                int acc = 0;
                                if (!(ka == kb || a1 == kb || b1 == ka)) {
                                    // check_tenure on (a, a1), (b, b1), (a, b), (a1, b1) in the reference's order (:282-285): the four stamps
                                    // are read together, the && chain -- and with it the lazy clears -- runs on the values
                                    int *sp[4] = {a.tabu + udir_pos(ka, a1, n), a.tabu + udir_pos(kb, b1, n), a.tabu + udir_pos(ka, kb, n), a.tabu + udir_pos(a1, b1, n)};
                                    const int sv[4] = {*sp[0], *sp[1], *sp[2], *sp[3]};
                                    acc = 1;
#pragma unroll
                                    for (int q = 0; q < 4 && acc; ++q) {
                                        if (cur_iter < 0 || cur_ten < 0 || sv[q] == 0) continue;
                                        if (cur_iter - sv[q] > cur_ten) *sp[q] = 0;
                                        else acc = 0;
                                    }
                                }
                                s_kick[0] = acc;
                            }
                            if (cur_iter != 0) {   // a stamp of value 0 is no entry
                                const int m = min(tl_m, a.tabu_list_cap);
                                for (int k = tid; k < m; k += kClThreads) {
                                    const int2 e = list[k];
                                    if (e.x == e0x && e.y == e0y) s_kick[1] = 1;
                                    if (e.x == e1x && e.y == e1y) s_kick[2] = 1;
                                }
                            }
                            __syncthreads();
                            if (tid == 0 && s_kick[0] && cur_iter != 0) {
                                int k = tl_m;   // past the capacity the count keeps running and the host stops using the list
                                if (!s_kick[1]) { if (k < a.tabu_list_cap) list[k] = make_int2(e0x, e0y); ++k; }
                                if (!s_kick[2] && !(e0x == e1x && e0y == e1y)) { if (k < a.tabu_list_cap) list[k] = make_int2(e1x, e1y); ++k; }
                                *const_cast<int *>(a.tabu_list_n) = k;
                                s_kick[5] = k - tl_m;
                            }
                            ++trials;
                            const bool again = !s_kick[0] && a.chain_pairs > 0 && pp + 1 < npairs;   // (the same in every thread: read behind the barrier)
                            __syncthreads();
                            if (!again) break;
                        }
                    }
                    CL_TT(1);
                    if (wave == 0) {
                        // the decision travels as a candidate: the two successors a1, b1 (node ids) in the pair field, a and b inside the
                        // replica in the internal-pair field, and -(1 + the entries the list has grown by + 4 x the pair that was taken) as the delta
                        // -- nobody has to look
                        // anything up before the swaps or before the next sweep's pass over the list
                        ClCand kc{0.0, kNoKey, 0u};
                        if (c == 0 && s_kick[0]) { kc.d = -1.0 - (double)(s_kick[5] + 4 * s_kick[6]); kc.key = make_key(s_kick[3], s_kick[4]); const int4 pr = s_kick[6] == nx_pp ? nx_pair : reinterpret_cast<const int4 *>(a.chain_ab)[s_kick[6]]; kc.ipair = ((unsigned)pr.z << 16) | (unsigned)pr.w; }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // this XCD's cleared stamps (and the first workgroup's list entries) are in memory
                        CL_TT(2);
                        bool okx = true;
                        if (C > 1) okx = cl_exchange<BEST, SORTED, kSmallD>(area, C, c, ++xep, kc, a.err, a.spin_limit, a.spin_ticks, a.copies, mycopy);
                        CL_TT(3);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // nothing this CU or its L2 holds of the stamps / the list outlives the boundary
                        CL_TT(4);
                        if (lane == 0) { *s_win_d = kc.d; *s_win_k = kc.key; *s_win_ip = kc.ipair; *s_fail = okx ? 0 : 1; }
                    }
                    __syncthreads();
                    if (*s_fail) { failed = true; break; }
                    const bool kicked = *s_win_k != kNoKey;
                    const int kword = kicked ? (int)(-*s_win_d) - 1 : 0, ptaken = kword >> 2;
                    if (kicked) tl_m += kword & 3;
                    tl_state = 0;   // the next iteration's number and tenure decide anew what is live
                    const int ia = kicked ? (int)(*s_win_ip >> 16) : 0, ib = kicked ? (int)(*s_win_ip & 0xffffu) : 0;
                    const int ka1 = kicked ? key_i(*s_win_k) : 0, kb1 = kicked ? key_j(*s_win_k) : 0;
                    if (kicked) {
                        const int pa = (int)pos[ia], pb = (int)pos[ib];
                        if (tid == 0) {   // :306-309
                            const int4 pr = ptaken == nx_pp ? nx_pair : reinterpret_cast<const int4 *>(a.chain_ab)[ptaken];
                            a.tabu[udir_pos(pr.x, ka1, n)] = cur_iter; a.tabu[udir_pos(pr.y, kb1, n)] = cur_iter;
                            // These copies stay dirty in this XCD's L2 until the next boundary's release.  That is too late in one case:
                            // the NEXT iteration's tenure is 0 -- the stamp expires at once, another XCD's workgroup clears it during that
                            // iteration (a dirty 0 there), and whichever copy reaches memory last wins (found by the randomised chains
                            // of tools/stress_parity.py).  Then, and only then, they are written back now (the fence writes the whole
                            // L2 back: 4 us; the reference's tenures start at 2 % of n).
                            if (ck + 1 < a.chain_n && a.chain_par[ck + 1] == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        }
                        if constexpr (INT) {   // the running cost (exact: integer terms)
                            const int a1i = (int)order[pa + 1 == n ? 0 : pa + 1], b1i = (int)order[pb + 1 == n ? 0 : pb + 1];
                            run_obj += cl_dist<WT, INT, CT>(coord, ia, ib) + cl_dist<WT, INT, CT>(coord, a1i, b1i) -
                                       cl_dist<WT, INT, CT>(coord, ia, a1i) - cl_dist<WT, INT, CT>(coord, ib, b1i);
                        }
                        patch_bounds(ia, ib, pa, pb);
                        int Lk = pb - pa; if (Lk < 0) Lk += n;
                        __syncthreads();   // everyone has read pa / pb
                        ClView now{pa, pb, Lk};
                        swaps(now, tid, kClThreads);
                        __syncthreads();
                    }
#ifdef TSP_STAMPS
                    if (tid == 0 && tour == 0 && c < 256) {
                        tt[5] = wall_clock64();
                        g_cl_tail[c][0] += tt[0] - tprev; g_cl_tail[c][1] += tt[1] - tt[0]; g_cl_tail[c][2] += tt[2] - tt[1]; g_cl_tail[c][3] += tt[3] - tt[2];
                        g_cl_tail[c][4] += tt[4] - tt[3]; g_cl_tail[c][5] += tt[5] - tt[4]; g_cl_tail[c][7] += 1;
                        tprev = tt[5];
                    }
#endif
                    if (c == 0 && tid == 0) {   // res as k_tabu_post_chain leaves it; read by the host after the launch
                        res[0] = kicked ? 1 : 0; res[1] = kicked ? ka1 : s_kick[3]; res[2] = kicked ? kb1 : s_kick[4]; res[3] = trials;
                        res[4] = 1; res[5] = better_inc; res[6] = 0; res[7] = 0;
                        *reinterpret_cast<double *>(res + 8) = obj;
                        *reinterpret_cast<double *>(a.chain + 2) = inc_best;
                        a.chain[1] = ck + 1;
                        if (a.chain_pairs > 0) *a.chain_pp = kpp + trials;
                        if (!kicked) a.chain[0] = 1;   // rejected (and no pair left to try): the host draws the next trial (tabusearch.c:262-287)
                    }
                    ck += 1;
                    if ((a.dbg & 2048) && c == 1 && ck >= 2) return;   // test hook: a workgroup stops answering in the middle of a chain
                    kpp = a.chain_pairs > 0 ? ptaken + 1 : ck;
                    if (kicked && kpp < npairs) { nx_pair = reinterpret_cast<const int4 *>(a.chain_ab)[kpp]; nx_pp = kpp; }   // used eleven sweeps on
                    if (kicked && ck < a.chain_n) {   // the next iteration's alg_2opt_tabu, on the kicked tour
                        done = 0;
                        cur_iter = a.iter + ck; cur_ten = a.chain_par[ck];
                        b0 = 0.0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { b0_mem_d[q] = 0.0; b0_mem_ip[q] = 0u; }
                    }
                }
            }
        } else {
            const long long r_old = pair_rank(ci, cj, n), r_new = pair_rank(ni, nj, n);
            scanned += probe_hit ? kClThreads : pair_rank(row_hi - 1, n - 1, n) - r_old;
            evals += r_new - r_old;   // the adjacent pairs among them come off per sweep / per launch (adj_seen)
            sweep_open = true;
            if (found) probe_on = r_new - r_old <= a.probe;
            if (found) {
                obj += bd;                              // heuristics.c:486
                moves += 1; reversed += Lr - 1;
                // the next chunk: twice the rows this hit was away from the cursor -- dense phases (a random tour: a hit
                // in almost every row) scan a row or two per step, sparse ones keep the chunk that found something
                chunk = max(a.rmin, min(a.rcap, 2 * (fi - ci + 1)));
                hit_rows = (3 * hit_rows + (fi - ci) + 2) >> 2;   // running mean of the rows between hits
                ci = fi; cj = fj;
            } else {
                hit_rows = max(hit_rows, row_hi - ci);  // nothing within these rows: the next hit is at least that far
                chunk = min(chunk * 2, a.rmax);
                if (row_hi >= n - 1) {                  // sweep complete
                    adj_seen += (long long)n - adjA0 - adjD;   // every tour edge has been passed
                    adjD = 0; adjA0 = 0; sweep_open = false;
                    sweeps += 1;
                    if (obj >= seen) done = 1;          // heuristics.c:492
                    else { seen = obj; ci = 0; cj = 0; }
                } else { ci = row_hi - 1; cj = n - 1; }
            }
            if constexpr (FS) leave = !done && a.fs_leave > 0 && hit_rows < a.fs_leave;
            else leave = !done && a.fs_exit > 0 && hit_rows >= a.fs_exit;
        }
    }

#ifdef TSP_STAMPS
    if constexpr (TABU) { if (tour == 0 && tid == 64 && c < 256) { g_cl_tail[c][6] += tl_n_full; g_cl_tail[c][9] += tl_n_hit; g_cl_tail[c][11] += tl_n_skip; } }
    if (tour == 0 && tid == 0 && c < 256) { for (int k = 0; k < 6; ++k) g_cl_prof[c][k] += prof[k]; g_cl_prof[c][6] += prof[8] + prof[10]; g_cl_prof[c][7] += steps - st->steps; }
    if (tour == 0 && tid == 0) { for (int k = 8; k < 12; ++k) atomicAdd(&g_cl_cnt[k - 8], prof[k]); atomicAdd(&g_cl_cnt[4], prof[6]); atomicAdd(&g_cl_cnt[5], prof[7]); }
#endif
    if (!failed) flush_view();   // a reversal the last exchange step left pending
    if (!failed) {
        // executed-work counters: every workgroup adds its share to a slot of its own (read and summed by tsp_dev_tours_download).
        // As 8 x 256 atomics on four words of one cache line they were the largest single item of a short launch (~70 us of 86).
        __syncthreads();
        if (lane == 0) { s_ll[wave] = w_lane; s_ll[8 + wave] = w_t1; }
        if (lane == 0) { s_k[wave] = (u64)w_ex; s_k[8 + wave] = (u64)w_st; }
        __syncthreads();
        if (tid < 4) {
            long long tot = 0;
            for (int w = 0; w < kClWaves; ++w)
                tot += tid == 0 ? s_ll[w] : (tid == 1 ? s_ll[8 + w] : (tid == 2 ? (long long)s_k[w] : (long long)s_k[8 + w]));
            a.stats_part[((size_t)tour * 256 + c) * 4 + tid] += tot;
#ifdef TSP_STAMPS
            if (tour == 0 && c < 256) g_cl_wstat[c][tid] += (unsigned long long)tot;
#endif
        }
        __syncthreads();
    }
    if constexpr (TABU) {
        if (!failed) {   // uniform: every thread of the workgroup takes part in the sum
            __syncthreads();
            const long long tot = block_sum<long long>(tabu_cnt, s_ll);
            if (tid == 0 && tot) atomicAdd(a.tabu_side, (unsigned long long)tot);
        }
    }
    // ---- write back (first workgroup of the cluster; a failed run leaves the tour in HBM untouched) ---------------------
    if (failed || c != 0) return;
    __syncthreads();
    if constexpr (!BEST) {
        if (a.count_evals) {
            if (sweep_open) adj_seen += edges_upto(make_key(ci, cj)) - adjA0 - adjD;   // the sweep goes on in the next launch
            evals -= adj_seen;
        }
    }
    int *pos_g = a.poss + (size_t)tour * n;
    for (int p0 = tid; p0 < n; p0 += LU * kClThreads) {
        int v[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) { const int p = p0 + u * kClThreads; v[u] = (int)order[p < n ? p : 0]; }
        if constexpr (SORTED) {
#pragma unroll
            for (int u = 0; u < LU; ++u) v[u] = a.gid[v[u]];
        }
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int p = p0 + u * kClThreads;
            if (p < n) { order_g[p] = v[u]; pos_g[v[u]] = p; }
        }
    }
    if (tid == 0) {
        st->ci = ci; st->cj = cj; st->chunk_rows = chunk; st->done = done; st->obj = obj; st->seen_cost = seen;
        st->hit_rows = hit_rows;
        st->sweeps = sweeps; st->evals = evals; st->moves = moves; st->reversed = reversed;
        st->pairs_scanned = scanned; st->steps = steps;
        st->parity = 0; st->pending = 0;
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

// LDS one workgroup may have: what the device grants (160 KiB on gfx950)
size_t cl_lds_limit(const tsp_dev_ctx *ctx) { return ctx->lds_bytes > 0 ? (size_t)ctx->lds_bytes : (size_t)64 * 1024; }

// ICOORD variants: coordinates are integers of bounded span -> exact as floats relative to the instance corner
template <int WT>
constexpr bool cl_float_coords() {
    return WT == WT_EUC_2D_ICOORD || WT == WT_CEIL_2D_ICOORD || WT == WT_ATT_ICOORD;
}

struct ClPlan {
    bool ok = false, sorted = false, float_coords = false;
    size_t lds = 0;
    int nid = 0, stage_pairs = 0;
};

// Which scan a run in `mode` uses on this handle, and whether the replica fits in LDS.
// want_fs: first improvement on the rank-order replica (the variant a sparse phase of a single tour's descent is handed to)
ClPlan cl_plan(const tsp_dev_tours *t, int mode, bool want_fs = false) {
    ClPlan p;
    const tsp_dev_inst *inst = t->inst;
    if (inst->n > 65534) return p;
    const int wt = inst->wtype;
    p.float_coords = wt == WT_EUC_2D_ICOORD || wt == WT_CEIL_2D_ICOORD || wt == WT_ATT_ICOORD;
    // (runs with a tabu list take the sorted scan at any size: their list code rides on it, and the alternative reads four
    // stamps per pair -- two_opt_tabu_list.hpp)
    p.sorted = inst->d_sperm && inst->prune_margin < 1e299 && inst->ng <= 32768 &&
               (mode == TSP_2OPT_BEST ? (inst->n >= t->cl_sorted_min_n || (t->cl_tabu_plan && inst->n >= 8))
                                      // first improvement: sparse phases of larger instances (a sweep of the tiles scan that finds
                                      // nothing is 217 us at n = 10 000; a box-pruned step 12 us).  Measured on HEU_VNS rounds
                                      // (tools/vns_time.py): no gain at n = 1 002 / 2 000, -7 % at 5 000, -19 % at 10 000
                                      : (want_fs && TSP_SW(inst, CLUSTER_FIRST_SORTED, 3000) > 0 && inst->n >= TSP_SW(inst, CLUSTER_FIRST_SORTED, 3000)));
    p.nid = p.sorted ? inst->ng * 64 : inst->n;
    const size_t ce = p.float_coords ? sizeof(float2) : sizeof(double2);
    // as many staged group pairs as fit (at least one), at most kClMaxStagePairs
    for (p.stage_pairs = p.sorted ? kClMaxStagePairs : 0;; --p.stage_pairs) {
        p.lds = cl_layout(inst->n, p.nid, inst->ng, ce, p.sorted, p.stage_pairs, p.sorted && mode == TSP_2OPT_FIRST).total;
        p.ok = p.lds <= cl_lds_limit(inst->ctx);
        if (p.ok || p.stage_pairs <= 1) break;
    }
    return p;
}

template <int WT, bool INT, int MODE, typename CT, bool SORTED, bool TABU = false>
hipError_t cl_launch_k(tsp_dev_tours *t, const ClusterArgs &a, size_t lds) {
    hipStream_t s = t->inst->ctx->stream;
    auto k = k_cluster_two_opt<WT, INT, MODE, CT, SORTED, TABU>;
    static size_t granted_dev[64] = {0};   // per kernel variant and device: the attribute call is not free and a resident driver launches thousands of times
    size_t &granted = granted_dev[t->inst->ctx->device & 63];
    if (lds > granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        granted = lds;
    }
    // Co-residency is asked for, not assumed: the exchange needs every workgroup of the launch on the chip at once.  The
    // runtime's occupancy figure for THIS kernel variant with THIS much LDS (registers, LDS, waves: whatever limits it) times
    // the CUs must cover the grid -- else the launch is refused here and the caller goes on with another engine, instead of
    // finding out by the exchange's time-out.  (Asked once per variant, device and LDS size.)  What the figure cannot know -- a
    // device shared with another process, CUs masked off -- remains the time-out's business (kClSpinMs).
    static size_t occ_lds[64] = {0};
    static int occ_blocks[64] = {0};
    const int dv = t->inst->ctx->device & 63;
    if (occ_lds[dv] != lds + 1) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, kClThreads, lds) != hipSuccess) { (void)hipGetLastError(); nb = 1; }
        occ_blocks[dv] = nb; occ_lds[dv] = lds + 1;
    }
    const long long grid = (long long)t->B * a.C;
    if (a.C > 1 && (long long)occ_blocks[dv] * std::max(1, t->inst->ctx->num_cus) < grid && !TSP_SW(t->inst, CLUSTER_ALLOW_OVERSUB, 0))
        return hipErrorCooperativeLaunchTooLarge;
    if (TSP_SW(t->inst, CLUSTER_COOP, 0) && a.C > 1) {   // measurement switch: the same grid as a cooperative launch (the runtime checks residency itself)
        ClusterArgs copy = a;
        void *args[] = {&copy};
        const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k), dim3((unsigned)grid), dim3(kClThreads), args, (unsigned)lds, s);
        return e == hipSuccess ? hipGetLastError() : e;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kClThreads), lds, s, a);
    return hipGetLastError();
}

template <int WT, bool INT>
hipError_t cl_launch(tsp_dev_tours *t, int mode, const ClPlan &p, const ClusterArgs &a) {
    using CT = std::conditional_t<cl_float_coords<WT>(), float2, double2>;
    if (mode == TSP_2OPT_FIRST) {
        if constexpr (has_root_filter<WT>()) {
            if (p.sorted) return cl_launch_k<WT, INT, TSP_2OPT_FIRST, CT, true>(t, a, p.lds);
        }
        return cl_launch_k<WT, INT, TSP_2OPT_FIRST, CT, false>(t, a, p.lds);
    }
    if constexpr (has_root_filter<WT>()) {
        if (p.sorted && a.tabu) return cl_launch_k<WT, INT, TSP_2OPT_BEST, CT, true, true>(t, a, p.lds);
        if (p.sorted) return cl_launch_k<WT, INT, TSP_2OPT_BEST, CT, true>(t, a, p.lds);
    }
    return cl_launch_k<WT, INT, TSP_2OPT_BEST, CT, false>(t, a, p.lds);
}
}  // namespace

// implemented in two_opt_grid.hip
int tsp_grid_after_external_run(tsp_dev_tours *t, int mode, int timed_out, bool pos_written = false);

#ifdef TSP_STAMPS
// diagnostic: per workgroup of tour 0, 100 MHz ticks per phase {tests, scan, block arg-min, exchange, counters, move, -, steps}; resets
extern "C" int tsp_dev_debug_cluster(unsigned long long *out /* 256 x 8 */) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tsp::g_cl_prof), sizeof(unsigned long long) * 256 * 8) != hipSuccess) return -1;
    static unsigned long long z[256 * 8];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_cl_prof), z, sizeof z);
    return 0;
}
extern "C" int tsp_dev_debug_cluster_tail(unsigned long long *out /* 256 x 12 */) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tsp::g_cl_tail), sizeof(unsigned long long) * 256 * 12) != hipSuccess) return -1;
    static unsigned long long z[256 * 12];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_cl_tail), z, sizeof z);
    return 0;
}
extern "C" int tsp_dev_debug_cluster_wstat(unsigned long long *out /* 256 x 4 */) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tsp::g_cl_wstat), sizeof(unsigned long long) * 256 * 4) != hipSuccess) return -1;
    static unsigned long long z[256 * 4];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_cl_wstat), z, sizeof z);
    return 0;
}
extern "C" int tsp_dev_debug_cluster_b0(unsigned long long *out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(tsp::g_cl_b0), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_cl_b0), z, sizeof z);
    return 0;
}
extern "C" int tsp_dev_debug_cluster_counts(unsigned long long *out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(tsp::g_cl_cnt), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tsp::g_cl_cnt), z, sizeof z);
    return 0;
}
#endif

bool tsp_cluster_fits(const tsp_dev_tours *t, int mode) { return t && cl_plan(t, mode).ok; }
bool tsp_cluster_sorted(const tsp_dev_tours *t, int mode) { return t && cl_plan(t, mode).sorted; }

// Cluster size for B tours on this device: one workgroup per CU at most, whole clusters only.
int tsp_cluster_size(const tsp_dev_tours *t, int mode) {
    const int cus = std::max(1, t->inst->ctx->num_cus);
    int C = std::max(1, std::min(256, cus / std::max(1, t->B)));
    const ClPlan p = cl_plan(t, mode);
    if (p.sorted) {   // no more workgroups than a few group pairs each
        const long long npairs = (long long)t->inst->ng * (t->inst->ng + 1) / 2;
        C = (int)std::max<long long>(1, std::min<long long>(C, (npairs + 3) / 4));
    }
    if (!p.sorted) {
        const long long nb = (t->n + kClThreads - 1) / kClThreads, nrb = (t->n - 1 + kClRows - 1) / kClRows;
        C = (int)std::max<long long>(1, std::min<long long>(C, nb * nrb));
        // first improvement: a step scans a few dozen rows, and the exchange gets slower with every workgroup that takes part
        // (measured, one tour: n = 532 / 1002 / 2000 are 10 % faster on 64 workgroups than on 256, n >= 5000 on 256)
        if (mode == TSP_2OPT_FIRST) C = (int)std::max<long long>(1, std::min<long long>(C, 32 * nb));
    }
    return std::max(1, TSP_SW(t->inst, CLUSTER_BLOCKS, C));
}

// Runs the tours of `t` to their local optima with C workgroups per tour.  Returns TSP_DEV_E_HIP with
// *fell_through = 1 when the cluster protocol gave up (a workgroup was not resident): the tours in HBM are
// then exactly as uploaded by the last launch that completed and the caller may continue with another engine.
// max_steps >= 0 caps the steps per tour (a capped best-improvement run gets its recomputed cost like a timed-out one).
int tsp_cluster_run(tsp_dev_tours *t, int mode, int C, int64_t max_steps, double time_limit_s, int *all_done, int *fell_through,
                    tsp_dev_tabu *tabu, int iter, int tenure) {
    if (fell_through) *fell_through = 0;
    if (all_done) *all_done = 0;
    if (!t || C < 1 || C > 256) return TSP_DEV_E_ARG;
    const ClPlan p = cl_plan(t, mode);
    if (!p.ok) return TSP_DEV_E_ARG;
    tsp_dev_inst *inst = t->inst;
    hipStream_t s = inst->ctx->stream;
    const int n = t->n, B = t->B;
    // all workgroups must be resident (TSP_CLUSTER_ALLOW_OVERSUB=1 lifts the check: the tests use it to drive the give-up path)
    if (C > 1 && (long long)B * C > std::max(1, inst->ctx->num_cus) && !TSP_SW(inst, CLUSTER_ALLOW_OVERSUB, 0)) return TSP_DEV_E_ARG;

    // First improvement of a single tour: two variants of the kernel hand the descent to each other between launches -- the plain
    // replica (probe + tiles at full speed) while hits come close together, the replica in rank order with the box-pruned step
    // once the running mean of the rows between hits (TourState::hit_rows, kept by both) passes fs_rows, back below fs_rows / 4.
    ClPlan pf;   // the rank-order variant's plan
    bool fs_avail = false;
    const int fs_rows = std::max(0, TSP_SW(inst, CLUSTER_FS_ROWS, 120));   // 80 .. 200 measure alike (rand10000: alg_2opt 20.7 ms, VNS round 3.4 ms)
    if (mode == TSP_2OPT_FIRST && B == 1 && fs_rows > 0 && max_steps < 0) {
        pf = cl_plan(t, mode, /*want_fs=*/true);
        const long long npairs = (long long)inst->ng * (inst->ng + 1) / 2;
        fs_avail = pf.ok && pf.sorted && C <= (npairs + 3) / 4;
    }
    // per-instance tables of the sorted scan: coordinates in rank order (padding far away), node -> rank
    if ((p.sorted || fs_avail) && !inst->d_rcoord) {
        const int nid_sorted = inst->ng * 64;
        std::vector<double2> rc((size_t)nid_sorted);
        std::vector<int> sperm((size_t)inst->n_slots);
        TSP_HIP_TRY(hipMemcpy(sperm.data(), inst->d_sperm, sizeof(int) * sperm.size(), hipMemcpyDeviceToHost));
        for (int k = 0; k < nid_sorted; ++k) {
            const int v = sperm[k];
            rc[k] = v >= 0 ? make_double2(inst->h_xy[2 * (size_t)v], inst->h_xy[2 * (size_t)v + 1]) : make_double2(1e30, 1e30);
        }
        TSP_HIP_TRY(hipMalloc(&inst->d_rcoord, sizeof(double2) * rc.size()));
        TSP_HIP_TRY(hipMemcpy(inst->d_rcoord, rc.data(), sizeof(double2) * rc.size(), hipMemcpyHostToDevice));
        TSP_HIP_TRY(hipMalloc(&inst->d_sinv, sizeof(int) * (size_t)n));
        TSP_HIP_TRY(hipMemcpy(inst->d_sinv, inst->h_sinv.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    }
    // per-handle: exchange area, error word, and the group-pair table dealt to C workgroups
    if (t->cl_C != C || !t->d_cl_slots) {
        (void)hipFree(t->d_cl_slots); t->d_cl_slots = nullptr;
        (void)hipFree(t->d_cl_pairtab); t->d_cl_pairtab = nullptr;
        t->cl_ntests = 0;
        const size_t words = (size_t)B * 2 * kClCopies * C * kClSlotGranules + 2;   // + the error word
        TSP_HIP_TRY(hipMalloc(&t->d_cl_slots, sizeof(unsigned long long) * words));
        t->cl_slot_words = words;
        TSP_HIP_TRY(hipMemsetAsync(t->d_cl_slots, 0, sizeof(unsigned long long) * words, s));
        t->cl_epoch = 0;
        t->cl_C = C;
    }
    if ((p.sorted || fs_avail) && !t->d_cl_pairtab) {
        const int ng = inst->ng;
        const long long npairs = (long long)ng * (ng + 1) / 2;
        const long long ntests = (npairs + C - 1) / C;
        std::vector<std::pair<double, int>> pr((size_t)npairs);
        size_t w = 0;
        for (int r = 0; r < ng; ++r)
            for (int cg = r; cg < ng; ++cg) {
                const double4 &rb = inst->h_gbox[r], &cb = inst->h_gbox[cg];
                const double gx = std::max(0.0, std::max(rb.x - cb.y, cb.x - rb.y)), gy = std::max(0.0, std::max(rb.z - cb.w, cb.z - rb.w));
                pr[w++] = {gx * gx + gy * gy, (r << 16) | cg};
            }
        std::sort(pr.begin(), pr.end());
        std::vector<int> tab((size_t)C * ntests, -1);
        bool dealt = false;
        if (C > 1 && B == 1 && TSP_SW(inst, CLUSTER_LPT, 1)) {
            // Deal by estimated cost instead of in turn.  A step costs the time of its slowest workgroup (1.9 us of an 11.4 us
            // best-improvement step at n = 10 000 were spent waiting for it), and what a group pair costs is decided by the tour:
            // whether it survives the box test and how many of its rows survive the culling.  Both are estimated here on the
            // tour the handle holds now (Euclidean lengths: a cost model, not a decision), the survivors are dealt heaviest
            // first to the least loaded workgroup (LPT), the others fill the tables up in turn.  Every pair is still tested in
            // every step; only who tests it changes.
            std::vector<int> order((size_t)n);
            // on the engine's stream (created non-blocking: a null-stream copy is not ordered behind work queued on it -- a kick
            // that was not waited for could still be rewriting the tour), then one wait
            TSP_HIP_TRY(hipMemcpyAsync(order.data(), t->d_order, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
            TSP_HIP_TRY(hipStreamSynchronize(s));
            const double sc = (inst->wtype_public == TSP_ATT) ? 1.0 / sqrt(10.0) : 1.0;
            auto X = [&](int v) { return inst->h_xy[2 * (size_t)v]; };
            auto Y = [&](int v) { return inst->h_xy[2 * (size_t)v + 1]; };
            auto len = [&](int u, int v) { return sc * sqrt((X(u) - X(v)) * (X(u) - X(v)) + (Y(u) - Y(v)) * (Y(u) - Y(v))) + 1.0; };
            std::vector<double> ds((size_t)n, 0.0), inc((size_t)n, 0.0), gmx((size_t)ng, 0.0);
            bool tour_ok = true;
            for (int q = 0; q < n && tour_ok; ++q) tour_ok = order[q] >= 0 && order[q] < n;
            if (tour_ok) {
                for (int q = 0; q < n; ++q) {
                    const int v = order[q], su = order[q + 1 == n ? 0 : q + 1], pv = order[q == 0 ? n - 1 : q - 1];
                    ds[v] = len(v, su);
                    inc[v] = std::max(ds[v], len(v, pv));
                }
                std::vector<int> sperm((size_t)inst->n_slots);
                TSP_HIP_TRY(hipMemcpy(sperm.data(), inst->d_sperm, sizeof(int) * sperm.size(), hipMemcpyDeviceToHost));   // per-instance, written once at creation
                for (int g = 0; g < ng; ++g)
                    for (int k = 0; k < 64; ++k) { const int v = sperm[(size_t)g * 64 + k]; if (v >= 0) gmx[g] = std::max(gmx[g], inc[v]); }
                struct Item { double cost; int e; };
                std::vector<Item> heavy, light;
                for (long long k = 0; k < npairs; ++k) {
                    const int e = pr[(size_t)k].second, r = e >> 16, cg = e & 0xffff;
                    const double T = gmx[r] + gmx[cg] + 2.0;
                    double cost = 0.0;
                    if (sc * sc * pr[(size_t)k].first < T * T) {
                        const double4 &cb = inst->h_gbox[cg];
                        int live = 0;
                        for (int q = 0; q < 64; ++q) {
                            const int v = sperm[(size_t)r * 64 + q];
                            if (v < 0) continue;
                            const double gx = std::max(0.0, std::max(cb.x - X(v), X(v) - cb.y)), gy = std::max(0.0, std::max(cb.z - Y(v), Y(v) - cb.w));
                            const double Tr = ds[v] + gmx[cg] + 2.0;
                            live += sc * sc * (gx * gx + gy * gy) < Tr * Tr;
                        }
                        cost = 8.0 + live;   // staging the pair's 128 records + its live rows against 64 columns
                    }
                    (cost > 0.0 ? heavy : light).push_back({cost, e});
                }
                std::stable_sort(heavy.begin(), heavy.end(), [](const Item &x, const Item &y) { return x.cost > y.cost; });
                std::vector<double> load((size_t)C, 0.0);
                std::vector<int> cnt((size_t)C, 0);
                // least loaded workgroup with room: a heap keyed by load
                std::vector<std::pair<double, int>> heap;
                for (int w = 0; w < C; ++w) heap.push_back({0.0, w});
                auto cmp = [](const std::pair<double, int> &x, const std::pair<double, int> &y) { return x.first > y.first || (x.first == y.first && x.second > y.second); };
                std::make_heap(heap.begin(), heap.end(), cmp);
                for (const Item &it : heavy) {
                    std::pop_heap(heap.begin(), heap.end(), cmp);
                    auto top = heap.back(); heap.pop_back();
                    const int w = top.second;
                    tab[(size_t)w * ntests + (size_t)cnt[w]++] = it.e;
                    load[w] += it.cost;
                    if (cnt[w] < ntests) { heap.push_back({load[w], w}); std::push_heap(heap.begin(), heap.end(), cmp); }
                }
                int w = 0;
                for (const Item &it : light) {   // the rest in turn, wherever there is room
                    while (cnt[w] >= ntests) w = (w + 1) % C;
                    tab[(size_t)w * ntests + (size_t)cnt[w]++] = it.e;
                    w = (w + 1) % C;
                }
                dealt = true;
            }
        }
        if (!dealt)
            for (long long k = 0; k < npairs; ++k) tab[(size_t)(k % C) * ntests + (size_t)(k / C)] = pr[(size_t)k].second;
        TSP_HIP_TRY(hipMalloc(&t->d_cl_pairtab, tab.size() * sizeof(int)));
        TSP_HIP_TRY(hipMemcpy(t->d_cl_pairtab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
        t->cl_ntests = (int)ntests;
    }

    ClusterArgs a;
    auto set_plan = [&](const ClPlan &q) {   // what depends on the numbering of the replica
        a.coord = q.sorted ? inst->d_rcoord : inst->d_coord;
        a.gid = q.sorted ? inst->d_sperm : nullptr;
        a.iid = q.sorted ? inst->d_sinv : nullptr;
        a.nid = q.nid;
        a.stage_pairs = q.stage_pairs;
    };
    set_plan(p);
    a.orders = t->d_order; a.states = t->d_state;
    a.gbox = inst->d_gbox;
    a.pairtab = t->d_cl_pairtab;
    a.slots = t->d_cl_slots;
    {
        const int want = TSP_SW(inst, CLUSTER_COPIES, 0);   // 0: by cluster size
        a.copies = want > 0 ? std::min(want, kClCopies) : (C >= 128 ? kClCopies : 1);
    }
    a.poss = t->d_pos;
    if (!t->d_cl_stats) {
        TSP_HIP_TRY(hipMalloc(&t->d_cl_stats, sizeof(long long) * (size_t)B * 256 * 4));
        TSP_HIP_TRY(hipMemsetAsync(t->d_cl_stats, 0, sizeof(long long) * (size_t)B * 256 * 4, s));
    }
    a.stats_part = t->d_cl_stats;
    a.err = reinterpret_cast<int *>(t->d_cl_slots + (t->cl_slot_words - 2));
    a.n = n; a.ng = inst->ng; a.ntests = t->cl_ntests; a.C = C;
    a.xcd_local = (B > 1 && C <= 32 && (B * C) % (8 * C) == 0 && TSP_SW(inst, CLUSTER_XCD_LOCAL, 1)) ? 1 : 0;
    a.count_evals = t->count_evals;
    a.tabu = nullptr; a.tabu_list = nullptr; a.tabu_list_n = nullptr; a.tabu_list_cap = 0; a.iter = iter; a.tenure = tenure; a.tabu_side = nullptr;
    if (tabu) {   // the caller has brought the handle's list up to date (tsp_tabu_list_prepare) and zeroed the side words
        if (mode != TSP_2OPT_BEST || !p.sorted || B != 1 || !tabu->list_valid) return TSP_DEV_E_ARG;
        a.tabu = tabu->d_stamp; a.tabu_list = tabu->d_list; a.tabu_list_n = tabu->d_list_n; a.tabu_list_cap = tabu->list_cap;
        a.tabu_side = tabu->d_tabu_pairs;
    }
    a.chain = nullptr; a.chain_par = nullptr; a.chain_n = 0; a.snap = nullptr; a.chain_pairs = 0; a.chain_ab = nullptr; a.chain_pp = nullptr;
    if (tabu && t->cl_ik_n > 0) {   // iterations of tabu() inside the launch (tsp_grid_tabu_iterations has filled the words)
        if (!t->d_chain || !t->d_order_snap || max_steps >= 0) return TSP_DEV_E_ARG;
        a.chain = t->d_chain; a.chain_par = t->d_chain + t->cl_ik_par; a.chain_n = t->cl_ik_n; a.snap = t->d_order_snap;
        a.chain_pairs = t->cl_ik_pairs; a.chain_ab = t->d_chain + t->cl_ik_ab; a.chain_pp = t->d_chain + t->cl_ik_pp;
    }
    a.probe = TSP_SW(inst, CLUSTER_PROBE, 4096);
    a.use_b0 = TSP_SW(inst, CLUSTER_B0, 1);
    a.defer_moves = TSP_SW(inst, CLUSTER_DEFER, 1);
    a.fs_rows = fs_rows;
    a.fs_exit = fs_avail ? fs_rows : 0;
    a.fs_leave = fs_avail ? std::max(1, fs_rows / 4) : 0;
    a.dbg = TSP_SW(inst, CLUSTER_DEBUG, 0);
    a.spin_limit = (unsigned)std::max(16, TSP_SW(inst, CLUSTER_SPIN_LIMIT, (int)kClSpinLimit));
    a.spin_ticks = (unsigned long long)std::max(1, TSP_SW(inst, CLUSTER_SPIN_MS, kClSpinMs)) * 100000ull;   // 100 MHz
    a.org_x = inst->org_x; a.org_y = inst->org_y;
    a.margin = inst->filter_margin; a.prune = inst->prune_margin; a.sum_margin = inst->sum_margin;
    // FIRST chunk geometry: the chunk adapts to the distance between hits (see the kernel's control block); the largest
    // keeps every workgroup busy for a few tiles
    {   // rows that cost (almost) nothing more than one: every workgroup at most one tile of at most four rows -- one trip of
        // the tile's row loop (measured, rand10000 on 256 workgroups: 24 / 36 / 48 / 52 rows = 16.8 / 16.4 / 16.3 / 17.0 ms per
        // descent; round 2, when a tile ended with a workgroup reduction, had its optimum at two rows per tile)
        const int nb = (n + kClThreads - 1) / kClThreads;
        a.rmin = std::max(1, std::min(2048, TSP_SW(inst, CLUSTER_MIN_ROWS, std::max(1, TSP_SW(inst, CLUSTER_TILE_ROWS, 4) * std::max(1, C / nb)))));
        a.rcap = std::max(a.rmin, TSP_SW(inst, CLUSTER_HIT_CAP, 4) * a.rmin);   // largest chunk right after a hit
        a.rbs = C / nb;
    }
    a.rmax = std::max(a.rmin, std::min(2048, TSP_SW(inst, CLUSTER_MAX_ROWS, C == 1 ? kClRows : std::max(kClRows, 8 * C))));
    // steps per launch: a time limit is honoured between launches (the reference checks it per sweep / per pair), so a
    // limited run is cut into launches of a millisecond or two (a relaunch reloads the replicas: ~0.1 ms)
    const int launch_iters = time_limit_s > 0 ? (mode == TSP_2OPT_FIRST ? 256 : 128) : (mode == TSP_2OPT_FIRST ? 16384 : 4096);

    const double t0 = wall_s();
    int status = TSP_OK;
    int64_t queued = 0;
    int launches_done = 0;
    bool on_fs = false;   // the variant of the launch in flight
    for (;;) {
        if (fs_avail) {   // t->h_state: as the last launch (or the upload) left it
            const int hr = t->h_state[0].hit_rows;
            on_fs = on_fs ? hr >= a.fs_leave : hr >= a.fs_exit;
            set_plan(on_fs ? pf : p);
            // inside the rank-order variant a tiles step reads through the id maps (+1.8 us at n = 10 000) and covers ~400 rows
            // per round of tiles: the box-pruned step takes over from half the switching distance on
            a.fs_rows = on_fs ? std::max(1, fs_rows / 2) : fs_rows;
        }
        a.max_iters = launch_iters;
        if (max_steps >= 0) {
            if (queued >= max_steps) break;
            a.max_iters = (int)std::min<int64_t>(launch_iters, max_steps - queued);
        }
        queued += a.max_iters;
        // exchange epochs run on from launch to launch (a tag of an earlier launch never equals a later epoch), so the area is
        // zeroed only when it is new, after a failed launch, and before the 32-bit epoch would wrap
        const unsigned epochs = (unsigned)a.max_iters + (unsigned)a.chain_n;   // (a chain of iterations: one more exchange per kick)
        if ((unsigned long long)t->cl_epoch + epochs + 6ull >= 0xffffffffull) {
            TSP_HIP_TRY(hipMemsetAsync(t->d_cl_slots, 0, sizeof(unsigned long long) * t->cl_slot_words, s));
            t->cl_epoch = 0;
        }
        a.epoch0 = t->cl_epoch;
        t->cl_epoch += (epochs + 5u) & ~1u;   // even: the parity of an epoch picks the half of the area (+1: the arrival rendezvous)
        hipError_t e = hipSuccess;
        TSP_DISPATCH_METRIC(inst->wtype, inst->integer_cost, { e = cl_launch<WTC, INTC>(t, mode, on_fs ? pf : p, a); });
        if (e == hipSuccess && launches_done == 0 && t->cl_post) {   // a driver's follow-up, decided on the device (see tsp_dev_tours::cl_post)
            t->cl_post(t->cl_post_ctx, s, a.err);
            t->cl_post = nullptr; t->cl_post_ran = true;
            // the driver's next iterations, queued behind this one without a wait (tsp_dev_tours::cl_chain): the same kernel on
            // the tour the follow-up leaves, with the epochs running on; a stop word on the device turns the launches that
            // follow a failed iteration into no-ops (their re-arm is vetoed, the control block says `done`)
            t->cl_chain_launched = 1;
            if (t->cl_chain && max_steps < 0 && !fs_avail) {
                for (int k = 1; e == hipSuccess && t->cl_chain(t->cl_post_ctx, s, k, &a.iter, &a.tenure); ++k) {
                    if ((unsigned long long)t->cl_epoch + (unsigned)a.max_iters + 6ull >= 0xffffffffull) {
                        TSP_HIP_TRY(hipMemsetAsync(t->d_cl_slots, 0, sizeof(unsigned long long) * t->cl_slot_words, s));
                        t->cl_epoch = 0;
                    }
                    a.epoch0 = t->cl_epoch;
                    t->cl_epoch += ((unsigned)a.max_iters + 5u) & ~1u;
                    TSP_DISPATCH_METRIC(inst->wtype, inst->integer_cost, { e = cl_launch<WTC, INTC>(t, mode, p, a); });
                    if (e != hipSuccess) break;
                    if (t->cl_post_k) t->cl_post_k(t->cl_post_ctx, s, k, a.err);
                    t->cl_chain_launched = k + 1;
                }
                if (e != hipSuccess) { (void)hipGetLastError(); e = hipSuccess; }   // the chain simply ends here: the iterations queued so far stand
            }
            t->cl_chain = nullptr; t->cl_post_k = nullptr;
        }
        if (e != hipSuccess) {
            // the attribute or the launch was refused (an LDS size this device does not grant): nothing ran, the tours in
            // HBM are as they were -- the caller may go on with another engine
            tsp::set_last_error("k_cluster_two_opt launch", e, __FILE__, __LINE__);
            (void)hipGetLastError();
            if (fell_through && launches_done == 0) *fell_through = 1;
            return TSP_DEV_E_HIP;
        }
        TSP_HIP_TRY(hipMemcpyAsync(t->h_state, t->d_state, sizeof(TourState) * (size_t)B, hipMemcpyDeviceToHost, s));
        if (!t->h_cl_err) TSP_HIP_TRY(hipHostMalloc(&t->h_cl_err, sizeof(int)));
        TSP_HIP_TRY(hipMemcpyAsync(t->h_cl_err, a.err, sizeof(int), hipMemcpyDeviceToHost, s));   // pinned: no staging copy
        TSP_HIP_TRY(hipStreamSynchronize(s));
        const int err = *t->h_cl_err;
        if (err) {
            (void)hipMemsetAsync(t->d_cl_slots, 0, sizeof(unsigned long long) * t->cl_slot_words, s);   // the error word too
            t->cl_epoch = 0;
            {   // remember it: the next AUTO decisions on this device leave the CLUSTER engine out (64 calls, doubling up to 4096
                // with every further give-up), so that a driver making thousands of calls on a shared device stalls once
                tsp_dev_ctx *cx = inst->ctx;
                cx->cl_giveups += 1;
                cx->cl_backoff = std::min(4096, std::max(64, 2 * cx->cl_backoff));
                cx->cl_skip = cx->cl_backoff;
            }
            if (tabu) {
                // the failed launch may have consumed or added to the side words of the tabu-list accounting: back to what
                // they were after the last launch that completed (zero before the first)
                if (launches_done > 0) (void)hipMemcpyAsync(tabu->d_tabu_pairs, tabu->d_tabu_pairs + kTabuSideWords, kTabuSideWords * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s);
                else (void)hipMemsetAsync(tabu->d_tabu_pairs, 0, kTabuSideWords * sizeof(unsigned long long), s);
            }
            tsp::set_last_error("k_cluster_two_opt: a workgroup of the cluster was not resident (exchange gave up)",
                                hipErrorLaunchFailure, __FILE__, __LINE__);
            if (fell_through) *fell_through = 1;
            return TSP_DEV_E_HIP;
        }
        launches_done += 1;
        inst->ctx->cl_backoff = 0;
        bool done = true;
        for (int b = 0; b < B; ++b) done = done && t->h_state[b].done;
        if (done) { if (all_done) *all_done = 1; break; }
        if (time_limit_s > 0 && wall_s() - t0 > time_limit_s) { status = TSP_TIME_LIMIT_EXCEEDED; break; }
        // another launch follows: keep the tabu-list side words as they stand after this one (see the give-up path)
        if (tabu) TSP_HIP_TRY(hipMemcpyAsync(tabu->d_tabu_pairs + kTabuSideWords, tabu->d_tabu_pairs, kTabuSideWords * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    }
    bool unfinished = status == TSP_TIME_LIMIT_EXCEEDED;
    for (int b = 0; b < B; ++b) unfinished = unfinished || !t->h_state[b].done;
    const int rc = tsp_grid_after_external_run(t, mode, unfinished, /*pos_written=*/true);
    t->h_state_fresh = rc == 0 && !(unfinished && mode == TSP_2OPT_BEST);   // (a cut-short best-improvement run has its cost recomputed after the poll)
    return rc ? rc : status;
}
