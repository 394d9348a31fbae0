// tsp_internal.hpp -- data laid out in HBM and the opaque handle types behind include/tsp_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/tsp_hip.h"
#include "tsp_dist.hpp"

namespace tsp {

// One scan block's candidate.  i < 0 means "none".
struct alignas(16) Partial {
    double delta;
    int i, j;
};
static_assert(sizeof(Partial) == 16, "Partial must be 16 bytes");

// Per-tour control block, read by every scan block (scalar loads) and written by the apply
// kernel's thread 0.  The descent is driven entirely by this block: the host only polls `done`.
struct alignas(16) TourState {
    // FIRST mode cursor: every pair up to and including (ci, cj) in (i<j) lexicographic order has
    // been scanned in the current sweep; (0,0) at the start of a sweep.
    int ci, cj;
    int chunk_rows;   // FIRST: number of rows the next scan covers
    int done;         // local optimum reached (or nothing to do)
    double obj;       // FIRST: running obj_best (+= delta); BEST: recomputed cost once done
    double seen_cost; // FIRST: best_cost of heuristics.c:442,492,495
    long long sweeps, evals, moves, reversed, pairs_scanned, steps;
    // sorted sweep: order/pos exist twice; `parity` says which copy holds the tour, `pending` that the move
    // (reverse positions mv_pa+1 .. mv_pb) chosen by the last sweep has not been carried out yet
    int parity, pending, mv_pa, mv_pb;
    // what the device really executed (CLUSTER engine; added to by every workgroup of the tour with atomics):
    // pairs a lane evaluated a bound or the delta for, pairs that reached tier 1, delta expressions executed, node
    // records derived for the sorted scan
    long long lane_pairs, tier1_pairs, exact_pairs, staged_recs;
    // FIRST on the CLUSTER engine: running mean of the rows between two hits (decides between the tiles scan and the
    // box-pruned scan of a step; survives k_rearm, so that a driver's next call starts with what the last one learnt)
    int hit_rows, pad0, pad1, pad2;
};

constexpr int kScanThreads = 256;
constexpr int kApplyThreads = 1024;

}  // namespace tsp

// ---- switches: TSP_<NAME> environment variables.  None of them changes a result.  They are read ONCE per
// instance handle (tsp_dev_inst_create) into this table; nothing on a call path calls getenv. ------------------------------
#define TSP_SWITCH_LIST(X)                                                                                                  \
    X(ENGINE) X(CLUSTER_BLOCKS) X(CLUSTER_MIN_ROWS) X(CLUSTER_MAX_ROWS) X(CLUSTER_HIT_CAP) X(CLUSTER_SPIN_LIMIT)            \
    X(CLUSTER_SPIN_MS) X(CLUSTER_COPIES) X(CLUSTER_TILE_ROWS) X(CLUSTER_XCD_LOCAL) X(CLUSTER_ALLOW_OVERSUB) X(CLUSTER_DEBUG) X(CLUSTER_PROBE) X(CLUSTER_FIRST_SORTED) X(TABU_DENSE)     \
    X(LDS_PROBE) X(LDS_PROBE2) X(LDS_MIN_ROWS) X(LDS_EDGE_CACHE) X(NO_ICOORD) X(NO_FILTER) X(NO_PRUNE) X(SORTED_MIN_N)      \
    X(SWEEP_BLOCKS) X(SWEEP_TABLE) X(BEST_ROWS_PER_BLOCK) X(BEST_RECS) X(FIRST_V1) X(FIRST_GRID_ROWS) X(FIRST_RJ)          \
    X(FIRST_MIN_ROWS) X(FIRST_MAX_ROWS) X(FIRST_ROWS_PER_BLOCK) X(COUNT_EVALS) X(USE_GRAPH) X(CONSTRUCT_GLOBAL)            \
    X(CONSTRUCT_NN) X(LDS_PAIR) X(CLUSTER_FS_ROWS) X(CLUSTER_LPT) X(CLUSTER_B0) X(LDS_F32_MIN_N) X(CLUSTER_DEFER)          \
    X(EXH_POS) X(EXH_WAVES) X(EXH_RJ) X(EXH_EVEN) X(EXH_PRIO) X(CLUSTER_COOP) X(TABU_INKERNEL)
namespace tsp {
enum SwitchId {
#define TSP_SW_ENUM(name) SW_##name,
    TSP_SWITCH_LIST(TSP_SW_ENUM)
#undef TSP_SW_ENUM
    SW_COUNT
};
struct Switches {
    int v[SW_COUNT];
    bool has[SW_COUNT];
    int get(SwitchId k, int dflt) const { return has[k] ? v[k] : dflt; }
};
void read_switches(Switches *sw);   // api.hip
}  // namespace tsp
#define TSP_SW(inst_, NAME, dflt) ((inst_)->sw.get(tsp::SW_##NAME, (dflt)))

// ---- opaque handles ------------------------------------------------------------------------

struct tsp_dev_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    int lds_bytes = 0;        // LDS one workgroup may be granted (hipDeviceAttributeMaxSharedMemoryPerBlock; 160 KiB on gfx950)
    // CLUSTER engine under AUTO: after a give-up (a workgroup was not resident: the device is shared or CU-masked) the next
    // `cl_skip` AUTO decisions go to the other engines; the back-off doubles with every further give-up (64 .. 4096 calls)
    int cl_skip = 0, cl_backoff = 0;
    long long cl_giveups = 0;
    long long cl_chain_losses = 0;   // give-ups in the middle of a chain of tabu() iterations (the search went on from the incumbent)
};

namespace tsp { struct NodeRec; }
struct tsp_dev_tours;
struct tsp_dev_inst {
    tsp_dev_ctx *ctx = nullptr;
    int n = 0;
    int wtype = 0;        // kernel variant: unknown -> EUC_2D; *_ICOORD when the coordinates allow it
    int wtype_public = 0; // the caller's weight type
    tsp_dev_tours *scratch1 = nullptr;  // reusable single-tour handle of the host-tour entry points (B == 1)
    tsp_dev_tours *scratch_b = nullptr; // ... and the handle of the last batch size used
    int scratch_b_count = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // reusable timing events
    double filter_margin = 1e300; // root filter margin of the 2-opt scans (tsp_dist.hpp); 1e300 = off
    double prune_margin = 1e300;  // new-edge bound margin (tsp_dist.hpp); 1e300 = off
    double sum_margin = 1e300;    // sorted sweep, both new edges: |ab| + |a1 b1| < bound + d(a,a1) + d(b,b1) + this
    int integer_cost = 1;
    double2 *d_coord = nullptr; // n x (x,y) or (lat,lon) for GEO
    // Sorted sweep (sqrt metrics): nodes ranked along a Hilbert curve, 64 consecutive ranks = one group.
    // d_sperm[slot] = node (or -1 for padding), n_slots = (ng + 1) * 64 (one extra all-padding group);
    // d_gbox[g] = {min x, max x, min y, max y} of group g's nodes (the padding group sits far away).
    int *d_sperm = nullptr;
    double4 *d_gbox = nullptr;
    std::vector<double4> h_gbox;   // host copy (pair table of the sorted sweep)
    int ng = 0, n_slots = 0;
    double org_x = 0.0, org_y = 0.0;   // min corner of the coordinates
    double cost_bound = 1e300;         // no distance of the instance exceeds this (bounding-box diagonal + rounding)
    std::vector<int> h_sinv;           // node -> rank slot
    void *io_pool = nullptr;           // call-local device scratch of the small host-array entry points (fitness, spot
    size_t io_pool_bytes = 0;          // distances, stamp reads/writes), grown on demand: no hipMalloc per call
    void *cons_pool = nullptr;         // scratch of tsp_dev_construct, grown on demand
    size_t cons_pool_bytes = 0;
    void *d_sxy = nullptr;             // coordinates in rank order (float2 / double2), built on first use by the large greedy
    double2 *d_rcoord = nullptr;       // CLUSTER engine, sorted scan: coordinates in rank order, padding far away (ng * 64)
    int *d_sinv = nullptr;             // CLUSTER engine, sorted scan: node -> rank slot
    std::vector<double> h_xy;   // host copy of the raw coordinates (2n)
    tsp::Switches sw;           // the TSP_* switches as they stood when this handle was created
};

struct tsp_dev_tabu {
    tsp_dev_inst *inst = nullptr;
    long long count = 0;
    int *d_stamp = nullptr;
    // compact list of the non-zero stamps (two_opt_tabu_list.hpp): what a sweep with a list works from
    int2 *d_list = nullptr;
    int *d_list_n = nullptr;                     // entries (device); may exceed list_cap after a scan = unusable
    unsigned long long *d_tabu_pairs = nullptr;  // pairs the sweeps of the current run skipped as tabu
    int *h_list_n = nullptr;                     // pinned
    int list_cap = 0;
    bool list_valid = false;                     // the list covers every non-zero stamp
    long long list_ub = 0;                       // host's upper bound of the entry count
    long long list_compact_at = 0;               // compaction when list_ub passes this
    bool last_run_list = false;                  // the last run with this handle worked from the list
};

struct tsp_dev_tours {
    tsp_dev_inst *inst = nullptr;
    int B = 0;
    int n = 0;
    // device
    int *d_order = nullptr;          // B x n : node at tour position p
    int *d_pos = nullptr;            // B x n : position of node v
    int *d_order2 = nullptr, *d_pos2 = nullptr;   // second copies (sorted sweep: moves are applied out of place)
    tsp::TourState *d_state = nullptr;       // the current control blocks: d_state_base + slot * B
    tsp::TourState *d_state_base = nullptr;  // 2 x B: k_first reads one slot and writes the other
    int slot = 0;
    tsp::Partial *d_partial = nullptr;
    size_t partial_per_tour = 0;
    tsp::NodeRec *d_rec = nullptr;   // B x max(n, n_slots) node records, rebuilt before every BEST step (k_recs*)
    double *d_gmax = nullptr;        // B x (ng + 1): longest tour edge leaving a node of the group (sorted sweep)
    int sorted_min_n = 0;            // BEST sweeps of instances with n >= this use the sorted sweep (GRID engine)
    int cl_sorted_min_n = 8;         // ... the sorted scan (CLUSTER engine: ahead of its tiles scan from n = 52 up, tools/small_best.py)
    int sweep_blocks = 512;          // k_sweep blocks per tour
    int *d_cl_ticket = nullptr;      // k_sweep: arrival counters per tour x cluster
    // exhaustive sweep in position order (two_opt_exh.hpp): the tour's coordinates, edge lengths and ids by position (padded)
    double2 *d_pxy = nullptr;
    int *d_pe = nullptr, *d_pid = nullptr;
    bool exh_lds_granted = false;
    int exh_share[4] = {0, 0, 0, 0};   // k_exh: rows per wave of each part of the grid (0: equal shares)
    int exh_gens = 0;                  // ... parts (= workgroups per CU)
    int exh_lds = 0;                 // k_exh: dynamic LDS a workgroup asks for (unused; it pins the number of workgroups per CU)
    int exh_blocks = 0, exh_rj = 4, exh_prio = 1;  // k_exh: workgroups per tour (0: the tiled k_step executes the exhaustive sweep), columns per lane
    int *d_pairtab = nullptr;        // group pairs per cluster of k_sweep blocks (host-built), or nullptr
    int *d_ticket = nullptr;         // per tour: scan blocks still to arrive in the current step
    int *d_row_ticket = nullptr;     // per tour x tile row (BEST two-level hand-off)
    tsp::Partial *d_row_slot = nullptr;
    int *d_row_evals = nullptr;
    int max_tile_rows = 0;
    int *d_slot_evals = nullptr;     // per scan block: pairs evaluated (tabu runs only)
    // CLUSTER engine (two_opt_cluster.hip): exchange area (+ error word) and the group-pair table dealt to cl_C workgroups
    unsigned long long *d_cl_slots = nullptr;
    size_t cl_slot_words = 0;
    int cl_C = 0;
    long long *d_cl_stats = nullptr; // B x 256 x 4: the CLUSTER engine's executed-work counters per workgroup (summed at download)
    bool cl_tabu_plan = false;       // the CLUSTER engine is being asked about / run for a descent with a tabu list
    unsigned cl_epoch = 0;           // exchange epochs handed out so far (they run on from launch to launch)
    int *d_cl_pairtab = nullptr;
    int cl_ntests = 0;
    // drivers on resident tours: the incumbent kept on the device (tsp_dev_tours_snapshot / _restore), kick results
    int *d_order_snap = nullptr;
    std::vector<double> h_obj_snap;
    int *d_kick_result = nullptr;    // 8 ints: {accepted, a1, b1, 0, acted, improved, 0, 0}
    int *h_kick_result = nullptr;    // pinned
    // work a driver wants queued right behind the FIRST launch of a CLUSTER run, before the run's wait for the device
    // (tabu(): incumbent snapshot + kick decided on the device from the finished descent -- one wait per iteration, not two)
    void (*cl_post)(void *ctx, hipStream_t s, const int *d_err) = nullptr;
    void *cl_post_ctx = nullptr;
    bool cl_post_ran = false;
    // ... and FURTHER runs of the same kind queued behind it without a wait in between (K iterations of tabu() per wait for the
    // device): cl_chain(ctx, s, k, &iter, &tenure) queues what precedes launch k >= 1 (a re-arm that a stop word on the device
    // can veto) and says with which iter / tenure it runs, or returns false when the chain ends; after launch k the run calls
    // cl_post_k(ctx, s, k, d_err).  cl_chain_launched = launches queued in all (1 + the chained ones).
    bool (*cl_chain)(void *ctx, hipStream_t s, int k, int *iter, int *tenure) = nullptr;
    void (*cl_post_k)(void *ctx, hipStream_t s, int k, const int *d_err) = nullptr;
    int cl_chain_launched = 0;
    // device words of a chain: [0] stop (set by a post kernel: every later launch of the chain is a no-op), [2..3] the incumbent's
    // cost (double), then per launch 8 ints of result + 1 double of cost
    int *d_chain = nullptr;
    int *h_chain = nullptr;          // pinned mirror
    // ... or INSIDE one launch (TSP_TABU_INKERNEL, the default): the kernel runs cl_ik_n iterations with the kick's first trials
    // the tenures and the host-drawn kick trials behind the result words of d_chain, keeps the incumbent in d_order_snap and writes the same result words
    int cl_ik_n = 0;
    int cl_ik_par = 0;               // offsets into d_chain: the tenures, ...
    int cl_ik_pairs = 0, cl_ik_ab = 0, cl_ik_pp = 0;   // ... the kick trials (count; 0 = one per iteration), and where the next trial's index lives
    bool h_state_fresh = false;   // h_state holds what d_state holds (set by a CLUSTER run's last poll, cleared by whatever queues work after it)
    int *h_cl_err = nullptr;         // pinned: the CLUSTER engine's error word, read with every poll
    bool tabu_list_run = false;      // the current tsp_grid_run goes through k_sweep<TABU> (two_opt_tabu_list.hpp)
    // reset point (device copies of the uploaded tours)
    int *d_order0 = nullptr;
    std::vector<double> h_obj0;
    std::vector<int> h_order_buf;    // staging of tsp_dev_tours_upload
    // pinned host mirror of the states, for polling
    tsp::TourState *h_state = nullptr;
    // scan geometry
    int first_rows_per_block = 8;
    int first_grid_rows = 32;        // k_first: tile rows of its fixed grid
    int first_rj = 1;                // k_first: columns per lane
    int first_max_rows2 = 2048;      // k_first: largest chunk
    int first_v1 = 0;                // use k_step<FIRST> (the first form) instead of k_first
    int first_min_rows = 8;
    int first_max_rows = 2048;
    int best_rows_per_block = 32;
    int count_evals = 1;             // FIRST: keep the reference-equivalent evaluation counter
    int use_recs = 1;                // BEST: materialise the node records once per step (k_recs)
    int use_graph = 0;               // replay full batches of steps from a captured hipGraph
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};   // per mode
    // accumulated device time
    double device_ms = 0.0;
};

// ---- error plumbing ------------------------------------------------------------------------
namespace tsp {
void set_last_error(const char *what, hipError_t e, const char *file, int line);

// Call-local device scratch: released when the scope ends, so an early error return leaks nothing.
template <typename T>
struct DevBuf {
    T *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(&p, sizeof(T) * count); }
    operator T *() const { return p; }
};
}
// Per-instance scratch for the host-array entry points (not for concurrent use, like every handle): >= bytes, 256-aligned.
void *tsp_io_pool(tsp_dev_inst *inst, size_t bytes);

#define TSP_HIP_TRY(expr)                                                    \
    do {                                                                     \
        hipError_t e__ = (expr);                                             \
        if (e__ != hipSuccess) {                                             \
            tsp::set_last_error(#expr, e__, __FILE__, __LINE__);             \
            return TSP_DEV_E_HIP;                                            \
        }                                                                    \
    } while (0)
