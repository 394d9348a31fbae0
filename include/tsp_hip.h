/*
 * include/tsp_hip.h -- C ABI of libtsp_hip.so, the MI355X (gfx950) 2-opt local-search engine.
 *
 * This is the drop-in boundary for the heuristics path of deno750/TSP_Optimization.  The
 * reference has no FFI layer; the seam is the set of C functions that its solver dispatch and
 * meta-heuristics call.  Each entry point below names the reference
 * function whose body it replaces.  Host code stays C: it keeps the reference's `instance`
 * struct and calls these functions with the struct's own arrays (see INTEGRATION.md and
 * tsp_optimization_amd/host/).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - Return value: 0 = ok, 1 = WRONG_STARTING_NODE, 2 = TIME_LIMIT_EXCEEDED (the reference's
 *     include/heuristics.h:6-7); negative = TSP_DEV_E_* (the reference has no such class: its
 *     unrecoverable errors go through LOG_E -> exit(1), include/utility.h:33).
 *   - `xy` is the reference's `point` array: n x {double x, double y} (include/utility.h:126-129),
 *     so `(const double *)inst->nodes` can be passed as is.
 *   - Tours are successor lists.  `succ` + `succ_stride` (in ints) address them: stride 1 for a
 *     plain int array, stride 2 with succ = &inst->solution.edges[0].j for the reference's
 *     `edge {int i; int j;}` array (include/utility.h:134-137).  Batched tours are
 *     `tour_stride` ints apart.
 *   - Weight types are numbered like the reference's enum weight_type (include/utility.h:45-52).
 *   - Nothing here falls back to the CPU: without a usable HIP device every call fails with
 *     TSP_DEV_E_NODEVICE.
 */
#ifndef TSP_HIP_H
#define TSP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* weight types: include/utility.h:45-52 */
enum { TSP_EUC_2D = 0, TSP_MAX_2D = 1, TSP_MAN_2D = 2, TSP_CEIL_2D = 3, TSP_GEO = 4, TSP_ATT = 5 };

/* status codes */
enum {
    TSP_OK = 0,
    TSP_WRONG_STARTING_NODE = 1,  /* include/heuristics.h:6 */
    TSP_TIME_LIMIT_EXCEEDED = 2,  /* include/heuristics.h:7 */
    TSP_DEV_E_NODEVICE = -1,      /* no HIP device / HIP runtime error at init */
    TSP_DEV_E_HIP = -2,           /* a HIP call failed (tsp_dev_last_error() has the text) */
    TSP_DEV_E_ARG = -3,           /* bad argument (NULL, n < 3, unknown mode ...) */
    TSP_DEV_E_NOT_A_TOUR = -4,    /* a successor list is not one Hamiltonian cycle */
    TSP_DEV_E_NOMEM = -5,
    TSP_DEV_E_COMM = -6           /* RCCL could not be opened or a collective failed (tsp_dev_comm_last_error()) */
};

/* 2-opt move selection */
enum {
    TSP_2OPT_FIRST = 0, /* alg_2opt: first improvement in (i<j) node order, applied immediately
                           (src/heuristics.c:438-502) */
    TSP_2OPT_BEST = 1   /* alg_2opt_tabu: best improvement, ties -> first pair (src/tabusearch.c:107-178) */
};

/* constructive heuristics */
enum {
    TSP_CONSTRUCT_GREEDY = 0, /* greedy(): src/heuristics.c:18-78 */
    TSP_CONSTRUCT_GRASP = 1   /* grasp():  src/heuristics.c:82-156 */
};

/* 2-opt execution engine (a performance choice; results are identical) */
enum {
    TSP_ENGINE_AUTO = 0,
    TSP_ENGINE_GRID = 1, /* many workgroups per tour, tour state in HBM, one launch (two for the sorted
                            best-improvement sweep) per step */
    TSP_ENGINE_LDS = 2,  /* one workgroup per tour, whole descent inside one launch, state in LDS */
    TSP_ENGINE_CLUSTER = 3 /* C workgroups per tour (B C <= #CUs), each with a replica of the tour in LDS, whole
                            descent inside one launch; one candidate per workgroup and step exchanged through L2 */
};

typedef struct tsp_dev_ctx tsp_dev_ctx;     /* one device + stream */
typedef struct tsp_dev_inst tsp_dev_inst;   /* node coordinates resident in HBM */
typedef struct tsp_dev_tours tsp_dev_tours; /* B tours of one instance resident in HBM */
typedef struct tsp_dev_tabu tsp_dev_tabu;   /* n(n-1)/2 tabu stamps resident in HBM */
typedef struct tsp_dev_comm tsp_dev_comm;   /* one rank of an RCCL communicator (multi-start across GPUs) */

typedef struct {
    int64_t sweeps;        /* completed passes over the (i<j) pair space                         */
    int64_t evals;         /* delta evaluations the reference would have executed (non-skipped
                              pairs, src/heuristics.c:474 / src/tabusearch.c:150)                */
    int64_t moves;         /* applied 2-opt moves                                                */
    int64_t reversed;      /* tour positions rewritten by segment reversals                      */
    int64_t pairs_scanned; /* pairs the device actually evaluated (>= evals in FIRST mode: a
                              chunk is scanned past the first improving pair)                    */
    int64_t steps;         /* steps: launches of a step kernel (GRID) or chunk iterations (LDS)   */
    double seconds;        /* wall time of the call, host clock                                  */
    double device_ms;      /* device time of the call, HIP events on the engine's stream         */
    /* What the device executed to decide those pairs (the decisions are the reference's; the work is not).  CLUSTER
     * engine: counted; other engines: lane_pairs = pairs_scanned, the rest -1 (not counted).                      */
    int64_t lane_pairs;    /* pairs for which a lane evaluated a lower bound of delta or delta itself             */
    int64_t tier1_pairs;   /* ... that the first bound could not exclude                                           */
    int64_t exact_pairs;   /* delta expressions actually executed (src/heuristics.c:474 / src/tabusearch.c:150)   */
    int64_t staged_recs;   /* node records derived for the sorted scan (one rounded root each)                    */
} tsp_two_opt_stats;

/* ---- context ---------------------------------------------------------------------------- */

/* Opens device `device` (hipSetDevice) and creates the engine's stream. */
int tsp_dev_open(int device, tsp_dev_ctx **out);
void tsp_dev_close(tsp_dev_ctx *ctx);
/* Text of the last failing HIP call on this thread ("" if none). */
const char *tsp_dev_last_error(void);
/* Number of HIP devices visible (0 if none / no runtime).  Does not create a context. */
int tsp_dev_count(void);
int tsp_dev_synchronize(tsp_dev_ctx *ctx);
/* The hipStream_t the engine launches on (as void*), for callers that order work against it. */
void *tsp_dev_stream(tsp_dev_ctx *ctx);

/* ---- instance: replaces the operand side of calc_dist (src/distutil.c:73-92) -------------- */

int tsp_dev_inst_create(tsp_dev_ctx *ctx, const double *xy, int n, int weight_type,
                        int integer_cost, tsp_dev_inst **out);
void tsp_dev_inst_destroy(tsp_dev_inst *inst);
int tsp_dev_inst_size(const tsp_dev_inst *inst);
/* Diagnostics.  The TSP_* environment switches (none changes a result) are read once, when an instance handle is
 * created; tours / tabu handles take theirs from their instance when THEY are created.  This re-reads them for `inst` (tests
 * and measurement scripts that run one form of a kernel against another on the same instance); handles created from it
 * earlier keep what they were created with, except for the cluster / tabu-path choices that are looked up per run. */
int tsp_dev_inst_reload_switches(tsp_dev_inst *inst);

/* calc_dist(i,j) for `count` index pairs, evaluated on the device (parity / spot checks). */
int tsp_dev_dist_pairs(tsp_dev_inst *inst, const int *i, const int *j, int count, double *out);

/* Full n x n matrix of calc_dist (row-major, diagonal 0) -- the north star's "distance-matrix
 * build"; the reference recomputes distances on every call and has no such array.
 * out_host may be NULL (timing only).  Elements are int32 when as_int32 != 0 (valid only with
 * integer costs), else double.  *kernel_ms receives the kernel's device time if not NULL. */
int tsp_dev_dist_matrix(tsp_dev_inst *inst, void *out_host, int as_int32, float *kernel_ms);

/* Self-test: out[k] = the hardware's approximate v_sqrt_f64(in[k]).  The exact integer-root
 * variants used for integer coordinates rely on its error bound; tests measure it through this. */
int tsp_dev_selftest_raw_sqrt(tsp_dev_ctx *ctx, const double *in, int count, double *out);

/* ---- construction: greedy() / grasp() for B starting nodes at once ------------------------ */
/* kind = TSP_CONSTRUCT_*.  starts[B].  urand: B x n doubles in [0,1], the values URAND()
 * (include/utility.h:36) would return for start b, in draw order (grasp draws exactly n per
 * call, src/heuristics.c:127); ignored for greedy.  Outputs: successor lists and obj (obj is
 * the reference's reported value, i.e. GRASP's closing edge counted twice, :135,:152).
 * status_out[B] (may be NULL) receives 0 or TSP_WRONG_STARTING_NODE per start. */
int tsp_dev_construct(tsp_dev_inst *inst, int kind, int B, const int *starts, const double *urand,
                      int *succ, int succ_stride, int64_t tour_stride, double *obj, int *status_out);

/* HEU_extramileage (src/heuristics.c:208-314): farthest pair, then cheapest insertion of every other
 * node; writes the successor list and the reference's obj (2*d(A,B) + the sum of the extra mileages). */
int tsp_dev_extramileage(tsp_dev_inst *inst, int *succ, int succ_stride, double *obj);

/* ---- 2-opt on host-resident tours: replaces alg_2opt / alg_2opt_tabu(skip_edge==NULL) ------ */
/* B tours in/out.  obj[B] in/out: FIRST adds the applied deltas to the incoming value like
 * `obj_best += delta` (src/heuristics.c:486); BEST overwrites it with the recomputed tour cost
 * (src/tabusearch.c:168-172).  time_limit_s <= 0 = unlimited.  stats may be NULL, else B entries. */
int tsp_dev_two_opt(tsp_dev_inst *inst, int mode, int engine, int B, int *succ, int succ_stride,
                    int64_t tour_stride, double *obj, double time_limit_s, tsp_two_opt_stats *stats);

/* ---- tabu stamps + alg_2opt_tabu with skip_edge != NULL (src/tabusearch.c:107-178) --------- */
int tsp_dev_tabu_create(tsp_dev_inst *inst, tsp_dev_tabu **out); /* all stamps 0 (CALLOC, :195) */
void tsp_dev_tabu_destroy(tsp_dev_tabu *tabu);
/* stamps[idx[k]] = value[k]; idx = x_udir_pos(i,j,n) (src/utility.c:17-30), as :306-309 */
int tsp_dev_tabu_set(tsp_dev_tabu *tabu, const int *idx, const int *value, int count);
int tsp_dev_tabu_get(tsp_dev_tabu *tabu, const int *idx, int *value, int count);
int tsp_dev_tabu_upload(tsp_dev_tabu *tabu, const int *stamps);   /* n(n-1)/2 ints */
int tsp_dev_tabu_download(tsp_dev_tabu *tabu, int *stamps);
/* Diagnostics of the handle's compact list of non-zero stamps, from which runs with a list work (a few hundred entries
 * of the reference's n(n-1)/2, :195): *entries = upper bound of its length, or -1 while it is out of date (the host
 * wrote stamps; the next run scans); *used_by_last_run = 1 when the last alg_2opt_tabu call worked from the list, 0
 * when it read the stamps pair by pair (list too long, tour outside the sorted sweep).  Either may be NULL. */
int tsp_dev_tabu_list_info(tsp_dev_tabu *tabu, int *entries, int *used_by_last_run);
/* One call of alg_2opt_tabu(inst, skip_edge, stored_prev, iter, tenure) on one tour.
 * stored_prev (may be NULL) receives the predecessor array (:173-175). */
int tsp_dev_two_opt_tabu(tsp_dev_inst *inst, tsp_dev_tabu *tabu, int iter, int tenure, int *succ,
                         int succ_stride, double *obj, int *stored_prev, double time_limit_s,
                         tsp_two_opt_stats *stats);

/* ---- tour cost: fitness() for B permutations (src/genetic.c:51-60) ------------------------- */
int tsp_dev_perm_cost(tsp_dev_inst *inst, int B, const int *perm, int64_t perm_stride, double *cost);

/* ---- device-resident tours (what bench.py times: inputs already in HBM) -------------------- */
int tsp_dev_tours_create(tsp_dev_inst *inst, int B, tsp_dev_tours **out);
void tsp_dev_tours_destroy(tsp_dev_tours *t);
/* Upload B successor lists (+ their obj values) and remember them as the reset point. */
int tsp_dev_tours_upload(tsp_dev_tours *t, const int *succ, int succ_stride, int64_t tour_stride,
                         const double *obj);
/* Restore the uploaded tours on the device (device-to-device). */
int tsp_dev_tours_reset(tsp_dev_tours *t);
int tsp_dev_tours_download(tsp_dev_tours *t, int *succ, int succ_stride, int64_t tour_stride,
                           double *obj, tsp_two_opt_stats *stats);
/* Run at most max_steps GRID-engine steps (one step = one scan of the selection rule's range and at
 * most one move per tour) in `mode`; max_steps < 0 = until every tour is at its local optimum (needs sync != 0:
 * an unbounded run without polls is refused with TSP_DEV_E_ARG).
 * Does not wait for completion unless `sync` != 0.  *all_done (if not NULL, sync only). */
int tsp_dev_tours_run(tsp_dev_tours *t, int mode, int64_t max_steps, double time_limit_s, int sync,
                      int *all_done);
/* The same on a chosen engine (TSP_ENGINE_*), always waiting for completion: device-resident tours run to their
 * local optima (or for at most max_steps steps per tour when max_steps >= 0; not with TSP_ENGINE_LDS).  A capped or
 * timed-out best-improvement run leaves the recomputed cost in obj like a finished one (src/tabusearch.c:168-172).
 * TSP_ENGINE_AUTO picks CLUSTER where it applies, else GRID. */
int tsp_dev_tours_run_engine(tsp_dev_tours *t, int mode, int engine, int64_t max_steps, double time_limit_s,
                             int *all_done);
/* ---- drivers on resident tours: what tabu() (src/tabusearch.c:188-320) and HEU_VNS (src/vns.c:103-166) do between two
 * 2-opt calls, on the device, so that an iteration moves no tour and no stamp across PCIe ------------------------------ */
/* alg_2opt / alg_2opt_tabu(NULL) on the tours as they are: the cursor starts a new sweep, obj_best continues (FIRST adds its
 * deltas to the value the control block holds, src/heuristics.c:442,486).  obj[B] (may be NULL) receives the result. */
int tsp_dev_tours_two_opt(tsp_dev_tours *t, int mode, int engine, double time_limit_s, double *obj);
/* One alg_2opt_tabu(inst, skip_edge, prev, iter, tenure) on resident tour 0 with resident stamps (B == 1). */
int tsp_dev_tours_two_opt_tabu(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter, int tenure, double time_limit_s, double *obj);
/* One trial of tabu()'s kick (src/tabusearch.c:262-309) with the host-drawn nodes a, b: rejected if the two edges share a
 * node or one of (a,a1) (b,b1) (a,b) (a1,b1) is in the tabu list (check_tenure with its lazy clears, in that order); else
 * the 2-exchange is carried out and (a,a1), (b,b1) are stamped with iter.  *accepted = 1 / 0. */
int tsp_dev_tours_tabu_kick(tsp_dev_tours *t, tsp_dev_tabu *tabu, int a, int b, int iter, int tenure, int *accepted);
/* One iteration of tabu() (src/tabusearch.c:238-309) in one wait for the device (two when the descent does not finish in the
 * CLUSTER engine's first launch): alg_2opt_tabu on resident tour 0; if its
 * cost is below *best_obj the tour becomes the incumbent (as tsp_dev_tours_snapshot; *best_obj updated, *improved = 1,
 * :241-249); then ONE trial of the kick with the host-drawn a, b (as tsp_dev_tours_tabu_kick; *accepted) -- further
 * trials, if that one is rejected, go through tsp_dev_tours_tabu_kick.  Returns the run's status; with a time limit hit
 * no kick is made (:255-258).  obj / improved / accepted may be NULL. */
int tsp_dev_tours_tabu_iteration(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter, int tenure, double time_limit_s, int a, int b,
                                 double *best_obj, double *obj, int *improved, int *accepted);
/* `count` (<= 128) iterations of tabu() in ONE wait for the device: iteration iter0 + k runs alg_2opt_tabu with tenure[k], updates
 * the incumbent and makes the FIRST trial of its kick with the host-drawn nodes ab[2k], ab[2k + 1] (what
 * tsp_dev_tours_tabu_iteration does for one iteration) -- the iterations run inside one launch of the CLUSTER engine (or, TSP_TABU_INKERNEL=0, as launches queued back
 * to back) and a word on the device stops the chain as soon as an iteration cannot be completed there.  *completed = iterations that ran up to their kick's trial; obj[k] /
 * improved[k] are filled for those.  *last_accepted = 0: the trial of iteration iter0 + *completed - 1 was rejected (edges that
 * share a node, or tabu): the caller draws further trials for it (tsp_dev_tours_tabu_kick) and goes on; the pairs ab[2k ..] of
 * the iterations that did not run have not been consumed (the caller serves them first: the libc stream stays the
 * reference's).  An iteration whose descent did not finish inside its launch (or whose exchange gave up) is not counted: the
 * caller runs it through tsp_dev_tours_tabu_iteration with its own a, b.  *completed = 0 with return 0: the chain does not
 * apply here (another engine, a list too long for it); nothing was touched. */
int tsp_dev_tours_tabu_iterations(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter0, int count, const int *tenure, const int *ab,
                                  double time_limit_s, double *best_obj, double *obj, int *improved, int *completed, int *last_accepted);
/* The same with the kick's FURTHER trials on the device too (src/tabusearch.c:262-287 draws pairs until one is accepted): ab holds
 * `pairs` (count <= pairs <= 256) node pairs in the order tabu() would draw them, and the iterations take them in that order --
 * iteration iter0 + k starts with the pair after the last one iteration iter0 + k - 1 took, and a rejected trial is followed by
 * the next pair, all inside the launch (the CLUSTER engine's tabu variant runs the iterations itself: incumbent, trials, kick
 * and the next descent on the replicas).  trials[k] = pairs iteration k took (0 when the earlier iterations had used them all up: no trial was made); the caller has consumed
 * sum(trials[0 .. *completed - 1]) pairs and serves the rest of its look-ahead first.  *last_accepted = 0 only when the pairs ran
 * out in the middle of an iteration's trials: the caller draws on (tsp_dev_tours_tabu_kick).  Everything else as above.  Where
 * the iterations cannot run inside a launch (TSP_TABU_INKERNEL=0, another engine) *completed = 0 and the caller takes
 * tsp_dev_tours_tabu_iterations. */
int tsp_dev_tours_tabu_iterations_ex(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter0, int count, const int *tenure, int pairs, const int *ab,
                                     double time_limit_s, double *best_obj, double *obj, int *improved, int *trials, int *completed,
                                     int *last_accepted);
/* kick() of src/vns.c:11-100 with the three host-drawn, sorted tour positions p1 < p2 < p3 (positions of the walk from
 * node 0): segments tour[p1+1..p2] and tour[p2+1..p3] swap places; the recomputed cost (:77-86) goes to the control block
 * and to *obj (may be NULL). */
int tsp_dev_tours_vns_kick(tsp_dev_tours *t, int p1, int p2, int p3, double *obj);
/* Page-lock / release a caller's host array that is passed to the library again and again (uploads at PCIe speed). */
int tsp_dev_host_register(void *p, size_t bytes);
int tsp_dev_host_unregister(void *p);
/* Incumbent on the device: remember the current tours + costs / go back to them (src/vns.c:148-158, tabusearch.c:241-249). */
int tsp_dev_tours_snapshot(tsp_dev_tours *t);
int tsp_dev_tours_restore(tsp_dev_tours *t);
/* Launch `reps` best-improvement steps back to back on the current tours (they continue the descent)
 * with HIP events around the run on the engine's stream; returns the mean duration of a step's
 * launches in *mean_ms and the reference-equivalent evaluations per step in *evals_per_launch.
 * Roofline measurement. */
int tsp_dev_tours_time_scan(tsp_dev_tours *t, int reps, float *mean_ms, int64_t *evals_per_launch);
/* Device time of the last tsp_dev_tours_run_engine / tsp_dev_tours_two_opt on this handle (HIP events on the engine's stream
 * around the run; the same value tsp_dev_tours_download reports as stats.device_ms), without a copy or a wait. */
int tsp_dev_tours_device_ms(tsp_dev_tours *t, double *ms);
/* Diagnostics: the kernels one GRID-engine step of `mode` launches for this handle, as text (bench.py names the kernel its
 * roofline describes from this; tests check that a switch selected the path they mean to test). */
int tsp_dev_tours_describe(tsp_dev_tours *t, int mode, char *buf, int cap);
/* min over tours of (cost, tour index) packed as (int64(cost) << 24 | index); the value the
 * multi-start all-reduce(min) combines across ranks.  true_cost != 0 recomputes the cost from
 * the tour (GRASP's reported value carries an offset).  Written to *packed. */
int tsp_dev_tours_best(tsp_dev_tours *t, int true_cost, int64_t *packed);


/* ---- multi-start across the GPUs of a node: generalises HEU_Grasp_iter's "keep the best start"
 * (src/heuristics.c:510-544, :534-539) to starts sharded k % world over the ranks.  The path shards across tours only, so
 * the whole exchange is ONE RCCL all-reduce(min) of the packed (cost << 24 | start id) and ONE broadcast of the winner's
 * successor list (4n bytes) from the rank that owns it.  librccl is dlopen()ed by the first of these calls; errors of this
 * group return TSP_DEV_E_COMM with the text in tsp_dev_comm_last_error(). ------------------------------------------------ */
#define TSP_COMM_ID_BYTES 128
const char *tsp_dev_comm_last_error(void);
/* 1 if librccl could be opened (dlopen) and holds every symbol this group needs, else 0.  Forms no communicator: a rank can
 * ask before it enters the collective tsp_dev_comm_init_rank, in which a rank that cannot load RCCL would leave the others waiting. */
int tsp_dev_comm_available(void);
/* One process per GPU: rank 0 obtains the id (ncclGetUniqueId) and hands its TSP_COMM_ID_BYTES to every rank by a side
 * channel of the caller's choice; then every rank calls init_rank with its own context (collective: returns when all have). */
int tsp_dev_comm_unique_id(char *id);
int tsp_dev_comm_init_rank(tsp_dev_ctx *ctx, int world, int rank, const char *id, tsp_dev_comm **out);
/* One process, ndev devices (ncclCommInitAll): out[k] is rank k on ctxs[k]'s device; use the *_group calls below. */
int tsp_dev_comm_init_all(tsp_dev_ctx *const *ctxs, int ndev, tsp_dev_comm **out);
void tsp_dev_comm_destroy(tsp_dev_comm *comm);
int tsp_dev_comm_info(const tsp_dev_comm *comm, int *rank, int *world, int *rccl_version);
/* (cost, start id) -> the int64 whose minimum is (lowest cost, then lowest start id): cost << 24 | start_id.  Costs that
 * are not non-negative integers below 2^39 (--fcost, GEO) cannot be packed: TSP_DEV_E_ARG, nothing is written. */
int tsp_dev_multistart_pack(double cost, int start_id, int64_t *packed);
/* all-reduce(min) of one int64 per rank over RCCL; every rank receives the minimum. */
int tsp_dev_multistart_allreduce(tsp_dev_comm *comm, int64_t packed_local, int64_t *packed_best);
/* all-reduce(min) of one double per rank: the first of the TWO reductions that carry costs the packed word cannot (--fcost,
 * src/utility.c:285; `< bestobj` on doubles, src/heuristics.c:534) -- min of the cost, then tsp_dev_multistart_allreduce of the
 * start id among the ranks whose cost equals that minimum (ties -> lowest start, the strict `<` in stream order). */
int tsp_dev_multistart_allreduce_f64(tsp_dev_comm *comm, double cost_local, double *cost_best);
/* Every collective of this group waits at most TSP_COMM_TIMEOUT_S (300) seconds for its peers; after that the communicator is
 * aborted and the call returns TSP_DEV_E_COMM (RCCL itself would wait for ever for a rank that died before the collective). */
/* broadcast of n ints (a successor list, `succ_stride` ints apart: 2 for &inst->solution.edges[0].j) from rank `root`. */
int tsp_dev_multistart_bcast_tour(tsp_dev_comm *comm, int root, int *succ, int succ_stride, int n);
/* The same two collectives for all ndev communicators of ONE process (ncclGroupStart / ncclGroupEnd around them).
 * packed_local[k] / packed_best[k] belong to rank k; the tour travels from rank `root`'s device to every device and is
 * read back from `read_back_rank`'s (so that a caller can check what a non-root rank received). */
int tsp_dev_multistart_allreduce_group(tsp_dev_comm *const *comms, int ndev, const int64_t *packed_local, int64_t *packed_best);
int tsp_dev_multistart_allreduce_f64_group(tsp_dev_comm *const *comms, int ndev, const double *cost_local, double *cost_best);
int tsp_dev_multistart_bcast_tour_group(tsp_dev_comm *const *comms, int ndev, int root, const int *succ_root, int succ_stride,
                                        int n, int read_back_rank, int *succ_out);

#ifdef __cplusplus
}
#endif
#endif
