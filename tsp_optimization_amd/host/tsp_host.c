/*
 * tsp_host.c -- the reference's heuristics entry points as thin C shims over libtsp_hip.so.
 *
 * Every function that the reference implements as an O(n^2) or O(n^2)-per-move CPU loop
 * (greedy, grasp, their multistart wrappers, alg_2opt, alg_2opt_tabu, fitness) marshals the
 * instance's own arrays into one call of the C ABI (include/tsp_hip.h) and writes the result back
 * in place, keeping the reference's data contract: edges[k].i == k, edges[k].j == succ(k),
 * obj_best updated as the reference updates it.  No distance, scan or reversal is computed here.
 */
#define _DEFAULT_SOURCE
#include "tsp_host.h"

#include <fcntl.h>
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "tsp_hip.h"

/* ---- the libc stream, looked ahead ----------------------------------------------------------------------------------------
 * A chain of tabu() iterations (tsp_dev_tours_tabu_iterations_ex) needs the kick's draws of EVERY iteration before it is
 * launched, and how many of them it takes is known only afterwards (a rejected trial is followed by another, the chain may stop
 * early).  The values it did not take were drawn too early, not wrongly -- so the generator is put back: libc's state is
 * remembered before the draws (host_random_mark) and afterwards restored and advanced by exactly the values that were consumed
 * (host_random_rewind).  Between two chains, and when tabu() returns, random() stands where the reference's run would have left
 * it: a caller that draws on, or reseeds, sees nothing of the look-ahead.
 *
 * The state is copied through the documented switch of state arrays (initstate / setstate, random(3)): setstate() returns the
 * array that was current, with the generator's position written into its first word (glibc: random_r.c, MAX_TYPES = 5, array
 * lengths 8 / 32 / 64 / 128 / 256 bytes by type).  The first use checks the round trip (mark, three draws, rewind, the same
 * three draws); where it does not hold, the fallback is a window of values handed back (host_random_unget) that every draw of
 * this file -- URAND(), rand_choice(), the skipped draws of a shard -- serves first: the sequence consumed is still the
 * reference's, only a reseed by the caller in between would not clear it. */
#define AHEAD_CAP 1024
static long g_ahead[AHEAD_CAP];
static int g_ahead_n = 0;   /* values waiting, oldest first (fallback only) */

static long host_random(void) {
    if (g_ahead_n > 0) {
        const long v = g_ahead[0];
        memmove(g_ahead, g_ahead + 1, sizeof(long) * (size_t)(--g_ahead_n));
        return v;
    }
    return random();
}
static void host_random_unget(const long *v, int count) {   /* v[0] is the next value to be served */
    if (count <= 0) return;
    if (g_ahead_n + count > AHEAD_CAP) LOG_E("look-ahead window of the libc stream overflows");
    memmove(g_ahead + count, g_ahead, sizeof(long) * (size_t)g_ahead_n);
    memcpy(g_ahead, v, sizeof(long) * (size_t)count);
    g_ahead_n += count;
}
int tsp_host_random_lookahead(void) { return g_ahead_n; }

typedef struct { char *where; size_t len; char copy[256]; } random_mark;
static char g_scratch_state[256];
static int g_mark_ok = -1;   /* -1 not tried, 0 the round trip does not hold here, 1 it does */

static int random_mark_take(random_mark *m) {
    static const size_t len_of[5] = {8, 32, 64, 128, 256};
    char *cur = initstate(1u, g_scratch_state, sizeof g_scratch_state);   /* parks the generator on a scratch array; cur = the caller's */
    if (!cur) return 0;
    unsigned info;
    memcpy(&info, cur, sizeof info);
    m->where = cur; m->len = len_of[info % 5u];
    memcpy(m->copy, cur, m->len);
    return setstate(cur) != NULL;
}
static int random_mark_back(const random_mark *m, long advance) {
    if (!initstate(1u, g_scratch_state, sizeof g_scratch_state)) return 0;
    memcpy(m->where, m->copy, m->len);
    if (!setstate(m->where)) return 0;
    for (long k = 0; k < advance; k++) (void)random();
    return 1;
}
static int host_random_mark(random_mark *m) {
    if (g_mark_ok < 0) {
        random_mark t;
        long a[3], b[3];
        g_mark_ok = 0;
        if (random_mark_take(&t)) {
            for (int k = 0; k < 3; k++) a[k] = random();
            if (random_mark_back(&t, 1)) {
                b[0] = a[0]; b[1] = random(); b[2] = random();
                const int same = a[1] == b[1] && a[2] == b[2];
                if (random_mark_back(&t, 0) && random() == a[0] && same && random_mark_back(&t, 0)) g_mark_ok = 1;
            }
        }
    }
    return g_mark_ok == 1 && g_ahead_n == 0 && random_mark_take(m);
}
/* the generator back to the mark and `consumed` values on; raw[0 .. drawn) are the values drawn since the mark (fallback: handed back) */
static void host_random_rewind(const random_mark *m, int marked, const long *raw, int drawn, int consumed) {
    if (marked && random_mark_back(m, consumed)) return;
    host_random_unget(raw + consumed, drawn - consumed);
}
#undef URAND
#define URAND() (((double)host_random()) / RAND_MAX)   /* include/utility.h:36 (the fallback's window first) */

#define GRASP_ITER_TIME_LIM 120 /* src/heuristics.c:11 */
#define TABU_CHAIN 128          /* most iterations of tabu() per wait for the device */
#define MULTISTART_BATCH 256    /* GRASP starts constructed per device call in HEU_Grasp_iter */

/* ---- device context and instance cache -------------------------------------------------------- */

typedef struct {
    const point *nodes; /* identity of the host array the upload was made from */
    int n, wtype, integer_cost;
    double checksum;    /* guards against a reused pointer with different content */
    tsp_dev_inst *dev;
    unsigned long stamp;
} cache_slot;

#define CACHE_SLOTS 8
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static tsp_dev_ctx *g_ctx = NULL;
static tsp_dev_comm *g_comm = NULL;   /* one process per GPU: this process's rank of the RCCL communicator */
/* CLI extensions of this build (-starts, -gpus): instance_params is the reference's struct and takes no new fields */
static int g_cli_starts = -1 /* default: 256 GRASP starts / 128 individuals */, g_cli_gpus = 0;
static cache_slot g_cache[CACHE_SLOTS];
/* The device-side stamp array alg_2opt_tabu works on, kept from call to call (one per device instance: 4 n (n-1) / 2 bytes of
 * HBM are not allocated and freed per call).  It holds no host state: every call uploads the caller's skip_edge array and
 * downloads it again, and nothing of the caller's memory is page-locked or remembered -- the reference's tabu() CALLOCs
 * and FREEs that array per run (tabusearch.c:195, :318), and a later allocation may well get the same address. */
static struct { tsp_dev_inst *dev; tsp_dev_tabu *tb; } g_tabu_cache;

static void tabu_cache_drop(void) {
    if (g_tabu_cache.tb) tsp_dev_tabu_destroy(g_tabu_cache.tb);
    memset(&g_tabu_cache, 0, sizeof g_tabu_cache);
}

static unsigned long g_clock = 0;
static __thread long long t_sweeps, t_evals, t_moves, t_grasp_iter_starts;
static __thread double t_driver_loop_s;   /* seconds the last tsp_host_tabu / tsp_host_vns spent in its iteration loop (not the initial solution) */
static __thread double t_device_ms;

static void dev_fail(const char *what, int rc) {
    LOG_E("%s failed with %d %s (this build has no CPU path: an MI355X and libtsp_hip.so are required)", what, rc,
          tsp_dev_last_error());
}

/* *rc_out == NULL: a failure ends the process (LOG_E, the reference's convention); else it is returned (a rank that is about
 * to enter a collective must not exit before it: see the failure agreement of the multi-GPU section) */
static tsp_dev_ctx *ctx_try_locked(int *rc_out) {
    if (!g_ctx) {
        const char *d = getenv("TSP_DEVICE");
        if (!d || !*d) d = getenv("LOCAL_RANK");   /* one process per GPU under torchrun / mpirun style launchers */
        int rc = tsp_dev_open(d ? atoi(d) : 0, &g_ctx);
        if (rc) {
            g_ctx = NULL;
            if (!rc_out) dev_fail("tsp_dev_open", rc);
            *rc_out = rc;
            return NULL;
        }
        atexit(tsp_host_shutdown);
    }
    return g_ctx;
}


static double nodes_checksum(const point *p, int n) {
    double s = 0.0;
    for (int k = 0; k < n; k++) s += p[k].x * 1.000001 + p[k].y;
    return s;
}

/* Device copy of inst->nodes (uploaded on first use, then reused while the host array is unchanged). */
static tsp_dev_inst *dev_inst_try_locked(instance *inst, int *rc_out) {
    if (!inst->nodes || inst->num_nodes < 3) LOG_E("instance has no nodes (or fewer than 3)");
    const double sum = nodes_checksum(inst->nodes, inst->num_nodes);
    const int ic = inst->params.integer_cost ? 1 : 0;
    int victim = 0;
    for (int k = 0; k < CACHE_SLOTS; k++) {
        cache_slot *c = &g_cache[k];
        if (c->dev && c->nodes == inst->nodes && c->n == inst->num_nodes && c->wtype == (int)inst->weight_type &&
            c->integer_cost == ic && c->checksum == sum) {
            c->stamp = ++g_clock;
            return c->dev;
        }
        if (g_cache[k].stamp < g_cache[victim].stamp) victim = k;
    }
    cache_slot *c = &g_cache[victim];
    if (c->dev && g_tabu_cache.dev == c->dev) tabu_cache_drop();
    if (c->dev) tsp_dev_inst_destroy(c->dev);
    memset(c, 0, sizeof *c);
    tsp_dev_ctx *ctx = ctx_try_locked(rc_out);
    if (!ctx) return NULL;
    int rc = tsp_dev_inst_create(ctx, (const double *)inst->nodes, inst->num_nodes, (int)inst->weight_type, ic, &c->dev);
    if (rc) {
        c->dev = NULL;
        if (!rc_out) dev_fail("tsp_dev_inst_create", rc);
        *rc_out = rc;
        return NULL;
    }
    c->nodes = inst->nodes; c->n = inst->num_nodes; c->wtype = (int)inst->weight_type; c->integer_cost = ic;
    c->checksum = sum; c->stamp = ++g_clock;
    return c->dev;
}

static tsp_dev_inst *dev_inst_locked(instance *inst) { return dev_inst_try_locked(inst, NULL); }

void tsp_host_shutdown(void) {
    pthread_mutex_lock(&g_lock);
    tabu_cache_drop();
    if (g_comm) { tsp_dev_comm_destroy(g_comm); g_comm = NULL; }
    for (int k = 0; k < CACHE_SLOTS; k++)
        if (g_cache[k].dev) { tsp_dev_inst_destroy(g_cache[k].dev); memset(&g_cache[k], 0, sizeof g_cache[k]); }
    if (g_ctx) { tsp_dev_close(g_ctx); g_ctx = NULL; }
    pthread_mutex_unlock(&g_lock);
}

long long tsp_host_last_grasp_iter_starts(void) { return t_grasp_iter_starts; }
double tsp_host_last_driver_loop_seconds(void) { return t_driver_loop_s; }

void tsp_host_last_stats(long long *sweeps, long long *evals, long long *moves, double *device_ms) {
    if (sweeps) *sweeps = t_sweeps;
    if (evals) *evals = t_evals;
    if (moves) *moves = t_moves;
    if (device_ms) *device_ms = t_device_ms;
}

static void keep_stats(const tsp_two_opt_stats *st) {
    t_sweeps = st->sweeps; t_evals = st->evals; t_moves = st->moves; t_device_ms = st->device_ms;
}

static double limit_of(const instance *inst) { return inst->params.time_limit > 0 ? (double)inst->params.time_limit : -1.0; }

/* ---- src/distutil.c:73 -------------------------------------------------------------------------- */
double calc_dist(int i, int j, instance *inst) {
    double d = 0.0;
    pthread_mutex_lock(&g_lock);
    int rc = tsp_dev_dist_pairs(dev_inst_locked(inst), &i, &j, 1, &d);
    pthread_mutex_unlock(&g_lock);
    if (rc) dev_fail("tsp_dev_dist_pairs", rc);
    return d;
}

/* ---- small host helpers (src/utility.c) ------------------------------------------------------------ */
int x_udir_pos(int i, int j, int num_nodes) { /* :17-30 */
    if (i == j) LOG_E("Indexes passed are equal!");
    if (i > num_nodes - 1 || j > num_nodes - 1) LOG_E("Indexes passed greater than the number of nodes");
    if (i > j) { int t = i; i = j; j = t; }
    return i * num_nodes + j - ((i + 1) * (i + 2)) / 2;
}

double get_elapsed_time(struct timeval start, struct timeval end) { /* :701-706 */
    return (double)(end.tv_sec - start.tv_sec) + 1e-6 * (double)(end.tv_usec - start.tv_usec);
}

/* :708-722.  Host-side list surgery for callers that kick a tour themselves (tabu, VNS drivers). */
void reverse_path(instance *inst, int start_node, int end_node, int *prev) {
    edge *e = inst->solution.edges;
    for (int cur = start_node;;) {
        const int p = prev[cur];
        e[cur].j = p;
        cur = p;
        if (p == end_node) break;
    }
    for (int k = 0; k < inst->num_nodes; k++) prev[e[k].j] = k;
}

void copy_instance(instance *dst, instance *src) { /* :724-743 */
    *dst = *src;
    dst->name = NULL; dst->comment = NULL; dst->params.file_path = NULL; dst->params.method.name = NULL;
    dst->thread_seeds = NULL;
    if (src->nodes) {
        dst->nodes = malloc(sizeof(point) * (size_t)src->num_nodes);
        memcpy(dst->nodes, src->nodes, sizeof(point) * (size_t)src->num_nodes);
    }
    if (src->ind) {
        dst->ind = malloc(sizeof(int) * (size_t)src->num_columns);
        memcpy(dst->ind, src->ind, sizeof(int) * (size_t)src->num_columns);
    }
    if (src->solution.edges) {
        dst->solution.edges = malloc(sizeof(edge) * (size_t)src->num_nodes);
        memcpy(dst->solution.edges, src->solution.edges, sizeof(edge) * (size_t)src->num_nodes);
    }
}

int rand_choice(int from, int to) { return from + (int)(URAND() * (to - from)); } /* :752-753 */

void free_instance(instance *inst) { /* :340-349 */
    free(inst->params.file_path); inst->params.file_path = NULL;
    free(inst->name); inst->name = NULL;
    free(inst->comment); inst->comment = NULL;
    free(inst->nodes); inst->nodes = NULL;
    free(inst->ind); inst->ind = NULL;
    free(inst->thread_seeds); inst->thread_seeds = NULL;
    free(inst->solution.edges); inst->solution.edges = NULL;
    free(inst->solution.xbest); inst->solution.xbest = NULL;
}

/* ---- constructive heuristics ------------------------------------------------------------------------- */

static void stamp_edge_sources(instance *inst) {
    for (int k = 0; k < inst->num_nodes; k++) inst->solution.edges[k].i = k;
}

static int construct_one(instance *inst, int kind, int start, const double *urand) {
    if (start >= inst->num_nodes) return WRONG_STARTING_NODE; /* heuristics.c:20 / :84 */
    double obj = 0.0;
    int status = 0;
    pthread_mutex_lock(&g_lock);
    int rc = tsp_dev_construct(dev_inst_locked(inst), kind, 1, &start, urand, &inst->solution.edges[0].j, 2,
                               2 * (int64_t)inst->num_nodes, &obj, &status);
    pthread_mutex_unlock(&g_lock);
    if (rc < 0) dev_fail("tsp_dev_construct", rc);
    if (status) return status;
    stamp_edge_sources(inst);
    inst->solution.obj_best = obj;
    return 0;
}

int greedy(instance *inst, int starting_node) { return construct_one(inst, TSP_CONSTRUCT_GREEDY, starting_node, NULL); }

int grasp(instance *inst, int starting_node) {
    if (starting_node >= inst->num_nodes) return WRONG_STARTING_NODE;
    /* the reference draws one URAND() per loop iteration, n per call (heuristics.c:127) */
    double *u = malloc(sizeof(double) * (size_t)inst->num_nodes);
    for (int k = 0; k < inst->num_nodes; k++) u[k] = URAND();
    int st = construct_one(inst, TSP_CONSTRUCT_GRASP, starting_node, u);
    free(u);
    return st;
}

int HEU_greedy(instance *inst) { return greedy(inst, 0); } /* :160 */
int HEU_Grasp(instance *inst) { return grasp(inst, 0); }   /* :505 */

/* :168-205 -- all n starting nodes, in batched device calls of at most GREEDY_ITER_BATCH starts (O(n) host memory per start
 * in flight, like the reference's O(n) state, instead of an n x n array); the first strictly better start wins (:193) and the
 * time limit is honoured between batches (:183 checks it per start) */
#define GREEDY_ITER_BATCH 1024
int HEU_Greedy_iter(instance *inst) {
    const int n = inst->num_nodes;
    const int B = n < GREEDY_ITER_BATCH ? n : GREEDY_ITER_BATCH;
    int *starts = malloc(sizeof(int) * (size_t)B);
    int *succ = malloc(sizeof(int) * (size_t)B * n);
    double *obj = malloc(sizeof(double) * (size_t)B);
    if (!starts || !succ || !obj) LOG_E("HEU_Greedy_iter: out of memory");
    struct timeval t0, t1;
    gettimeofday(&t0, 0);
    double best = DBL_MAX;
    int status = 0;
    for (int k0 = 0; k0 < n; k0 += B) {
        const int m = n - k0 < B ? n - k0 : B;
        for (int k = 0; k < m; k++) starts[k] = k0 + k;
        pthread_mutex_lock(&g_lock);
        int rc = tsp_dev_construct(dev_inst_locked(inst), TSP_CONSTRUCT_GREEDY, m, starts, NULL, succ, 1, n, obj, NULL);
        pthread_mutex_unlock(&g_lock);
        if (rc < 0) dev_fail("tsp_dev_construct", rc);
        for (int k = 0; k < m; k++)
            if (obj[k] < best) {
                best = obj[k];
                for (int v = 0; v < n; v++) { inst->solution.edges[v].i = v; inst->solution.edges[v].j = succ[(size_t)k * n + v]; }
            }
        gettimeofday(&t1, 0);
        if (inst->params.time_limit > 0 && get_elapsed_time(t0, t1) > inst->params.time_limit && k0 + B < n) { status = TIME_LIMIT_EXCEEDED; break; }
    }
    inst->solution.obj_best = best;
    free(starts); free(succ); free(obj);
    return status;
}

/* :510-544 -- wall-clock bounded random restarts.  The RNG stream is consumed in the reference's
 * order (start node :519, then the n draws of grasp()); starts are built MULTISTART_BATCH at a time. */
int HEU_Grasp_iter(instance *inst, int time_lim) {
    const int n = inst->num_nodes;
    const int limit = time_lim > 0 ? time_lim : GRASP_ITER_TIME_LIM;
    struct timeval t0, t1;
    gettimeofday(&t0, 0);
    const int B = MULTISTART_BATCH;
    int *starts = malloc(sizeof(int) * B);
    double *u = malloc(sizeof(double) * (size_t)B * n);
    int *succ = malloc(sizeof(int) * (size_t)B * n);
    double *obj = malloc(sizeof(double) * B);
    double best = DBL_MAX;
    edge *best_edges = calloc((size_t)n, sizeof(edge));
    t_grasp_iter_starts = 0;
    for (;;) {
        for (int b = 0; b < B; b++) {
            starts[b] = (int)(URAND() * (n - 1));
            for (int k = 0; k < n; k++) u[(size_t)b * n + k] = URAND();
        }
        gettimeofday(&t1, 0);
        if (get_elapsed_time(t0, t1) >= limit) break;
        pthread_mutex_lock(&g_lock);
        int rc = tsp_dev_construct(dev_inst_locked(inst), TSP_CONSTRUCT_GRASP, B, starts, u, succ, 1, n, obj, NULL);
        pthread_mutex_unlock(&g_lock);
        if (rc < 0) dev_fail("tsp_dev_construct", rc);
        for (int b = 0; b < B; b++)
            if (obj[b] < best) {
                best = obj[b];
                for (int v = 0; v < n; v++) { best_edges[v].i = v; best_edges[v].j = succ[(size_t)b * n + v]; }
            }
        t_grasp_iter_starts += B;
    }
    inst->solution.obj_best = best;
    memcpy(inst->solution.edges, best_edges, sizeof(edge) * (size_t)n);
    free(starts); free(u); free(succ); free(obj); free(best_edges);
    return TIME_LIMIT_EXCEEDED; /* the reference only ever leaves this loop through the time limit (:522-524) */
}

/* :208-314 -- farthest pair + n-2 cheapest insertions, one scan/apply launch pair per inserted node */
int HEU_extramileage(instance *inst) {
    double obj = 0.0;
    pthread_mutex_lock(&g_lock);
    int rc = tsp_dev_extramileage(dev_inst_locked(inst), &inst->solution.edges[0].j, 2, &obj);
    pthread_mutex_unlock(&g_lock);
    if (rc < 0) dev_fail("tsp_dev_extramileage", rc);
    stamp_edge_sources(inst);
    inst->solution.obj_best = obj;
    return 0;
}

/* ---- refinement ------------------------------------------------------------------------------------------ */

/* src/heuristics.c:438-502 */
int alg_2opt(instance *inst) {
    tsp_two_opt_stats st;
    memset(&st, 0, sizeof st);
    double obj = inst->solution.obj_best;
    pthread_mutex_lock(&g_lock);
    int rc = tsp_dev_two_opt(dev_inst_locked(inst), TSP_2OPT_FIRST, TSP_ENGINE_AUTO, 1, &inst->solution.edges[0].j, 2,
                             2 * (int64_t)inst->num_nodes, &obj, limit_of(inst), &st);
    pthread_mutex_unlock(&g_lock);
    if (rc < 0) dev_fail("tsp_dev_two_opt", rc);
    inst->solution.obj_best = obj;
    keep_stats(&st);
    if (rc == TIME_LIMIT_EXCEEDED) LOG_I("2-opt heuristics time exceeded");
    return rc;
}

/* src/tabusearch.c:107-178.  skip_edge is the caller's host array of n(n-1)/2 stamps, which the caller also writes
 * between two calls (tabu() stamps two edges per iteration, check_tenure clears lazily), so it travels to the device
 * and back around every call (2 x 200 MB at n = 10 000).  What a loop like the reference's tabu() does not pay again and
 * again is the device allocation.  A driver that wants no copies at all keeps the stamps resident: tsp_host_tabu /
 * HEU_Tabu_* below. */
int alg_2opt_tabu(instance *inst, int *skip_edge, int *stored_prev, const int iter, const int tenure) {
    tsp_two_opt_stats st;
    memset(&st, 0, sizeof st);
    double obj = inst->solution.obj_best;
    pthread_mutex_lock(&g_lock);
    tsp_dev_inst *d = dev_inst_locked(inst);
    tsp_dev_tabu *tb = NULL;
    int rc = 0;
    if (skip_edge) {
        if (g_tabu_cache.tb && g_tabu_cache.dev != d) tabu_cache_drop();
        if (!g_tabu_cache.tb) {
            rc = tsp_dev_tabu_create(d, &g_tabu_cache.tb);
            if (!rc) g_tabu_cache.dev = d;
        }
        tb = g_tabu_cache.tb;
        if (!rc) rc = tsp_dev_tabu_upload(tb, skip_edge);
    }
    if (!rc) rc = tsp_dev_two_opt_tabu(d, tb, iter, tenure, &inst->solution.edges[0].j, 2, &obj, stored_prev,
                                       limit_of(inst), &st);
    if (tb && rc >= 0) { int rc2 = tsp_dev_tabu_download(tb, skip_edge); if (rc2) rc = rc2; }
    pthread_mutex_unlock(&g_lock);
    if (rc < 0) dev_fail("tsp_dev_two_opt_tabu", rc);
    inst->solution.obj_best = obj;
    keep_stats(&st);
    if (rc == TIME_LIMIT_EXCEEDED) LOG_I("2-opt heuristics time exceeded");
    return rc;
}

/* src/heuristics.c:547-594: the constructive status is overwritten by the 2-opt status, as there */
int HEU_2opt_grasp(instance *inst) { (void)HEU_Grasp(inst); return alg_2opt(inst); }
int HEU_2opt_grasp_iter(instance *inst) { (void)HEU_Grasp_iter(inst, inst->params.time_limit / 5); return alg_2opt(inst); }
int HEU_2opt_greedy(instance *inst) { (void)HEU_greedy(inst); return alg_2opt(inst); }
int HEU_2opt_greedy_iter(instance *inst) { (void)HEU_Greedy_iter(inst); return alg_2opt(inst); }
int HEU_2opt_extramileage(instance *inst) { (void)HEU_extramileage(inst); return alg_2opt(inst); }

/* ---- VNS (src/vns.c) ------------------------------------------------------------------------------------- */

/* vns.c:11-100: three random tour positions, segments b..c and d..e swap places, cost recomputed.
 * The draws and the list surgery are host logic; the cost is one batched fitness on the device.
 * The reference reads tour[idx3+1] one past its array when idx3 == n-1 (:57): here that index wraps
 * to the tour's first node (the only value that keeps the successor list a tour). */
int kick(instance *inst) {
    const int n = inst->num_nodes;
    edge *e = inst->solution.edges;
    int *tour = malloc(sizeof(int) * (size_t)n);
    for (int k = 0, v = 0; k < n; k++) { tour[k] = v; v = e[v].j; }
    int p1 = rand_choice(0, n), p2 = p1, p3 = p1;
    while (p2 == p1 || abs(p1 - p2) <= 1) p2 = rand_choice(0, n);
    while (p3 == p1 || p3 == p2 || abs(p1 - p3) <= 1 || abs(p2 - p3) <= 1) p3 = rand_choice(0, n);
    int t;
    if (p1 > p2) { t = p1; p1 = p2; p2 = t; }
    if (p1 > p3) { t = p1; p1 = p3; p3 = t; }
    if (p2 > p3) { t = p2; p2 = p3; p3 = t; }
    const int a = tour[p1], b = tour[p1 + 1], c = tour[p2], d = tour[p2 + 1], g = tour[p3];
    const int f = tour[p3 + 1 == n ? 0 : p3 + 1];
    e[a].j = d; e[g].j = b; e[c].j = f;                      /* :60-62 */
    for (int k = 0, v = 0; k < n; k++) { tour[k] = v; v = e[v].j; }
    double cost = 0.0;
    fitness_batch(inst, tour, 1, &cost);                     /* :77-86, same edge order as fitness() */
    inst->solution.obj_best = cost;
    for (int k = 0; k < n; k++) { e[tour[k]].i = tour[k]; e[tour[k]].j = tour[k + 1 == n ? 0 : k + 1]; }
    free(tour);
    return 0;
}

/* The three tour positions of kick() (vns.c:23-47), drawn and sorted on the host: the draws are the reference's. */
static void kick_positions(int n, int *p1, int *p2, int *p3) {
    int a = rand_choice(0, n), b = a, c = a;
    while (b == a || abs(a - b) <= 1) b = rand_choice(0, n);
    while (c == a || c == b || abs(a - c) <= 1 || abs(b - c) <= 1) c = rand_choice(0, n);
    int t;
    if (a > b) { t = a; a = b; b = t; }
    if (a > c) { t = a; a = c; c = t; }
    if (b > c) { t = b; b = c; c = t; }
    *p1 = a; *p2 = b; *p3 = c;
}

static void download_into(instance *inst, tsp_dev_tours *t) {
    double obj = 0.0;
    int rc = tsp_dev_tours_download(t, &inst->solution.edges[0].j, 2, 2 * (int64_t)inst->num_nodes, &obj, NULL);
    if (rc) dev_fail("tsp_dev_tours_download", rc);
    stamp_edge_sources(inst);
    inst->solution.obj_best = obj;
}

/* vns.c:103-166.  After the initial solution the tour lives on the device: a round is kick (segment swap + recomputed
 * cost, one kernel each), alg_2opt on the resident tour, one 8-byte read of the cost, and a device-to-device copy that
 * either remembers the new incumbent or goes back to the old one (:148-158).  Only the draws are host work. */
int tsp_host_vns(instance *inst, long long max_rounds) {
    const int n = inst->num_nodes;
    const int time_limit = inst->params.time_limit > 0 ? inst->params.time_limit : DEFAULT_TIME_LIM;
    struct timeval t0, t1;
    gettimeofday(&t0, 0);
    int status = HEU_2opt_greedy_iter(inst);                 /* :116 */
    double best_obj = inst->solution.obj_best;
    pthread_mutex_lock(&g_lock);
    tsp_dev_inst *d = dev_inst_locked(inst);
    tsp_dev_tours *t = NULL;
    int rc = tsp_dev_tours_create(d, 1, &t);
    if (!rc) rc = tsp_dev_tours_upload(t, &inst->solution.edges[0].j, 2, 2 * (int64_t)n, &best_obj);
    if (!rc) rc = tsp_dev_tours_snapshot(t);                 /* the incumbent, :119-122 */
    pthread_mutex_unlock(&g_lock);
    if (rc) dev_fail("tsp_host_vns: resident tour", rc);
    long long rounds = 0;
    struct timeval tl0, tl1;
    gettimeofday(&tl0, 0);
    for (long long round = 0; max_rounds < 0 || round < max_rounds; round++, rounds++) {
        gettimeofday(&t1, 0);
        if (get_elapsed_time(t0, t1) > time_limit) { status = TIME_LIMIT_EXCEEDED; break; }
        int p1, p2, p3;
        kick_positions(n, &p1, &p2, &p3);                    /* :138 */
        double obj = 0.0;
        pthread_mutex_lock(&g_lock);
        rc = tsp_dev_tours_vns_kick(t, p1, p2, p3, NULL);
        if (!rc) { rc = tsp_dev_tours_two_opt(t, TSP_2OPT_FIRST, TSP_ENGINE_AUTO, limit_of(inst), &obj); if (rc > 0) { status = rc; rc = 0; } else status = 0; }   /* :143 */
        if (!rc) {
            if (obj < best_obj) { best_obj = obj; rc = tsp_dev_tours_snapshot(t); }     /* :148-155 */
            else rc = tsp_dev_tours_restore(t);                                            /* :157-158 */
        }
        pthread_mutex_unlock(&g_lock);
        if (rc) dev_fail("tsp_host_vns: round", rc);
        if (inst->params.verbose >= 3 && obj == best_obj) LOG_I("Updated incumbent: %0.0f", best_obj);
    }
    gettimeofday(&tl1, 0);
    t_driver_loop_s = get_elapsed_time(tl0, tl1);
    pthread_mutex_lock(&g_lock);
    rc = tsp_dev_tours_restore(t);
    pthread_mutex_unlock(&g_lock);
    if (rc) dev_fail("tsp_dev_tours_restore", rc);
    download_into(inst, t);
    inst->solution.obj_best = best_obj;
    pthread_mutex_lock(&g_lock);
    tsp_dev_tours_destroy(t);
    pthread_mutex_unlock(&g_lock);
    (void)rounds;
    return status;
}

int HEU_VNS(instance *inst) { return tsp_host_vns(inst, -1); }

/* ---- tabu search (src/tabusearch.c:188-320) ---------------------------------------------------------------- */

/* The tour and the stamps stay on the device for the whole search (the reference CALLOCs num_columns ints, :195): an
 * iteration is alg_2opt_tabu on the resident tour, an 8-byte read of its cost, a device-to-device copy when the incumbent
 * improves (:241-249), and the kick (:262-309) -- one tiny launch per trial that tests (a,a1) (b,b1) (a,b) (a1,b1) against
 * the tabu list with check_tenure's lazy clears, carries the 2-exchange out and stamps the removed edges; the host only
 * draws a and b and steps the tenure policy. */
/* the tenure policies, applied after an iteration's kick (:298-304).  draw == 0: a dry run that predicts the tenures of a chain
 * (step_policy :33-37 and linear_policy :47-59 draw nothing; random_policy :69-72 draws, and a chain never spans such a step) */
static void tabu_policy_step(int policy, int iter, int *tenure, int *rising, int lo, int hi, int draw) {
    if (policy == 0) {                                   /* step_policy :33-37 */
        if (iter % 100 == 0) *tenure = (*tenure == lo) ? hi : lo;
    } else if (policy == 1) {                            /* linear_policy :47-59 */
        if (*tenure > hi) *tenure = hi;
        if (*tenure < lo) *tenure = lo;
        if (*tenure == hi || *tenure == lo) *rising = !*rising;
        if (*rising) (*tenure)++; else (*tenure)--;
    } else if (draw) {                                   /* random_policy :69-72 */
        if (iter == 1 || iter % 100 == 0) *tenure = rand_choice(lo, hi + 1);
    }
}

int tsp_host_tabu(instance *inst, int policy, long long max_iterations) {
    const int n = inst->num_nodes;
    struct timeval t0, t1;
    gettimeofday(&t0, 0);
    int status = HEU_2opt_greedy_iter(inst);                 /* :200 */
    if (status) LOG_E("An error occurred in HEU_2opt_greedy_iter");
    double obj0 = inst->solution.obj_best;
    pthread_mutex_lock(&g_lock);
    tsp_dev_inst *d = dev_inst_locked(inst);
    tsp_dev_tabu *tb = NULL;
    tsp_dev_tours *t = NULL;
    int rc = tsp_dev_tabu_create(d, &tb);
    if (!rc) rc = tsp_dev_tours_create(d, 1, &t);
    if (!rc) rc = tsp_dev_tours_upload(t, &inst->solution.edges[0].j, 2, 2 * (int64_t)n, &obj0);
    pthread_mutex_unlock(&g_lock);
    if (rc) dev_fail("tsp_host_tabu: resident state", rc);
    double best_obj = DBL_MAX;
    int have_best = 0;
    int lo = (int)ceil(n * 0.02), hi = (int)round(n * 0.1);  /* :213-214, MIN/MAX_TENURE_RATE :12-13 */
    if (lo == hi) hi += 2; else if (hi < lo) { int tt = lo; lo = hi; hi = tt; }
    int tenure = lo, rising = 0;
    const char *chain_env = getenv("TSP_TABU_CHAIN");
    const int chain_max = chain_env && *chain_env ? (atoi(chain_env) < 1 ? 1 : (atoi(chain_env) > TABU_CHAIN ? TABU_CHAIN : atoi(chain_env))) : TABU_CHAIN;
    int iter = 1;
    /* a chain stops at the first rejected kick, and the launches queued behind that point are wasted (no-ops, ~10 us each): the
     * chain is made one and a half times as long as the chains have lately run (rand10000, step policy: one first trial in
     * twelve is rejected, chains run ten iterations on average; one in six at n = 299) */
    double mean_run = 8.0;
    const int trace = getenv("TSP_HOST_TRACE") != NULL;
    /* iterations inside the launch, the kick's further trials included (the default), or queued launches with one trial each */
    const int in_kernel = !(getenv("TSP_TABU_INKERNEL") && atoi(getenv("TSP_TABU_INKERNEL")) == 0);
    struct timeval tl0, tl1;
    gettimeofday(&tl0, 0);
    long long st_chains = 0, st_queued = 0, st_done = 0, st_single = 0, st_retrials = 0;   /* TSP_HOST_STATS=1: printed at the end */
    while (max_iterations < 0 || iter <= max_iterations) {
        gettimeofday(&t1, 0);
        if (inst->params.time_limit > 0 && get_elapsed_time(t0, t1) > inst->params.time_limit) { status = TIME_LIMIT_EXCEEDED; break; }
        /* A chain of K iterations per wait for the device.  K ends where the host must act between two iterations: at the cap on
         * the iterations, and -- random policy -- at an iteration whose policy step draws (:69-72: that draw comes after the
         * iteration's kick draws, so the iteration is the chain's last).  Step and linear policies draw nothing: the tenure of
         * every iteration of the chain is known now. */
        /* (inside the launch a rejected trial does not end the chain: the longest chain from the start) */
        int K = (chain_env && *chain_env) || in_kernel ? chain_max : (int)(1.5 * mean_run + 2.0);
        if (K > chain_max) K = chain_max;
        if (max_iterations >= 0 && K > max_iterations - iter + 1) K = (int)(max_iterations - iter + 1);
        if (policy == 2)
            for (int k = 0; k < K - 1; k++)
                if (iter + k == 1 || (iter + k) % 100 == 0) { K = k + 1; break; }
        int tenures[TABU_CHAIN], ab[4 * TABU_CHAIN], improved[TABU_CHAIN] = {0}, trials[TABU_CHAIN];
        long raw[4 * TABU_CHAIN];
        double objs[TABU_CHAIN];
        {
            int ten = tenure, ris = rising;
            for (int k = 0; k < K; k++) {
                tenures[k] = ten;
                tabu_policy_step(policy, iter + k, &ten, &ris, lo, hi, 0);
            }
        }
        /* The draws of the kicks (:264-265, and :262-287 for the trials that follow a rejected one), in the order the iterations
         * would make them: K first trials and a reserve for the further ones (one first trial in twelve is rejected at n = 10 000,
         * one in six at n = 299).  With the iterations inside the launch the device takes the pairs in order, further trials
         * included, and reports how many each iteration took; what it did not take goes back to the front of the window. */
        const int P = in_kernel ? (K + K / 3 + 4 > 2 * TABU_CHAIN ? 2 * TABU_CHAIN : K + K / 3 + 4) : K;
        random_mark mark;
        const int marked = host_random_mark(&mark);
        for (int k = 0; k < 2 * P; k++) { raw[k] = host_random(); ab[k] = (int)((((double)raw[k]) / RAND_MAX) * n); }
        int completed = 0, accepted = 0, consumed = 0;
        pthread_mutex_lock(&g_lock);
        if (in_kernel)
            rc = tsp_dev_tours_tabu_iterations_ex(t, tb, iter, K, tenures, P, ab, limit_of(inst), &best_obj, objs, improved, trials, &completed, &accepted);   /* :238-249, :262-309 */
        else
            rc = tsp_dev_tours_tabu_iterations(t, tb, iter, K, tenures, ab, limit_of(inst), &best_obj, objs, improved, &completed, &accepted);
        pthread_mutex_unlock(&g_lock);
        if (rc < 0) dev_fail("tsp_dev_tours_tabu_iterations", rc);
        st_chains++; st_queued += K; st_done += completed;
        for (int k = 0; k < completed; k++) {
            if (improved[k]) have_best = 1;
            consumed += in_kernel ? trials[k] : 1;
            if (in_kernel) st_retrials += trials[k] - 1;
        }
        if (rc > 0 && completed < K && improved[completed]) have_best = 1;   /* cut short by the time limit: the incumbent is updated before the status is looked at (:241-249, :255) */
        if (rc) { status = rc; LOG_I("2-opt move returned status %d", rc); host_random_rewind(&mark, marked, raw, 2 * P, 2 * consumed); break; }
        if (completed == 0) {
            /* the chain did not apply, or its first iteration could not be finished on the device: this iteration the one-wait
             * way, with the draws it has; the later iterations' draws wait for their turn */
            host_random_rewind(&mark, marked, raw, 2 * P, 2);
            double obj = 0.0;
            int imp = 0;
            pthread_mutex_lock(&g_lock);
            rc = tsp_dev_tours_tabu_iteration(t, tb, iter, tenure, limit_of(inst), ab[0], ab[1], &best_obj, &obj, &imp, &accepted);
            pthread_mutex_unlock(&g_lock);
            if (imp) have_best = 1;
            if (rc < 0) dev_fail("tsp_dev_tours_tabu_iteration", rc);
            if (rc) { status = rc; LOG_I("2-opt move returned status %d", rc); break; }
            completed = 1; st_single++;
            objs[0] = obj;
        } else {
            host_random_rewind(&mark, marked, raw, 2 * P, 2 * consumed);
        }
        if (trace) {   /* TSP_HOST_TRACE=1: one line per iteration (a divergence between two runs shows at its first iteration) */
            if (iter == 1) fprintf(stderr, "[tabu-trace] start obj %.0f\n", obj0);
            fprintf(stderr, "[tabu-chain] from %d: %d iterations asked, %d pairs, %d completed, %d pairs taken, last kick %s\n", iter, K, P, completed,
                    consumed, accepted ? "accepted" : "rejected");
            for (int k = 0; k < completed; k++)
                fprintf(stderr, "[tabu-trace] iter %d tenure %d trials %d obj %.0f%s\n", iter + k, tenures[k], in_kernel && !st_single ? trials[k] : 1,
                        objs[k], k == completed - 1 ? (accepted ? " +" : " -") : " +");
        }
        mean_run = 0.75 * mean_run + 0.25 * completed;
        /* every completed iteration but the last had its kick accepted (the chain stops at a rejection) */
        for (int k = 0; k < completed - 1; k++) tabu_policy_step(policy, iter + k, &tenure, &rising, lo, hi, 1);
        const int last = iter + completed - 1;
        while (!accepted) {                                  /* :262-287: draws until a pair of free, disjoint edges comes up */
            const int a = rand_choice(0, n), b = rand_choice(0, n);
            pthread_mutex_lock(&g_lock);
            rc = tsp_dev_tours_tabu_kick(t, tb, a, b, last, tenure, &accepted);   /* + :288-290 move, :306-309 stamps */
            pthread_mutex_unlock(&g_lock);
            st_retrials++;
            if (rc) dev_fail("tsp_dev_tours_tabu_kick", rc);
        }
        tabu_policy_step(policy, last, &tenure, &rising, lo, hi, 1);
        iter = last + 1;
    }
    gettimeofday(&tl1, 0);
    t_driver_loop_s = get_elapsed_time(tl0, tl1);
    if (getenv("TSP_HOST_STATS"))
        fprintf(stderr, "[tabu] %d iterations: %lld chains, %lld iterations queued, %lld completed in chains, %lld run singly, %lld further kick trials\n",
                iter - 1, st_chains, st_queued, st_done, st_single, st_retrials);
    pthread_mutex_lock(&g_lock);
    rc = have_best ? tsp_dev_tours_restore(t) : 0;
    pthread_mutex_unlock(&g_lock);
    if (rc) dev_fail("tsp_dev_tours_restore", rc);
    download_into(inst, t);
    if (have_best) inst->solution.obj_best = best_obj;
    pthread_mutex_lock(&g_lock);
    tsp_dev_tours_destroy(t);
    tsp_dev_tabu_destroy(tb);
    pthread_mutex_unlock(&g_lock);
    return status;
}

int HEU_Tabu_step(instance *inst) { return tsp_host_tabu(inst, 0, -1); }
int HEU_Tabu_lin(instance *inst) { return tsp_host_tabu(inst, 1, -1); }
int HEU_Tabu_rand(instance *inst) { return tsp_host_tabu(inst, 2, -1); }


/* ---- genetic algorithm (src/genetic.c) --------------------------------------------------------------------- */
#define GA_POPULATION 1000      /* genetic.c:12 */
#define GA_MUTATION_RATE 0.1    /* :13 */
#define GA_PARENT_RATE 0.6      /* :14 */
#define GA_HEURISTIC_INIT 0.0   /* :16 */
#define GA_CROSSOVER_SPLIT 0.0  /* :17 */
#define GA_TWO_OPT_MUT 0.00     /* :18 */

typedef struct { int *genes; double fit; } ga_member;   /* genetic.c:22-25 */

/* :62-68 -- the difference of two doubles returned as int: ties and sub-unit gaps compare equal */
static int ga_by_fitness_desc(const void *l, const void *r) {
    return (int)(((const ga_member *)r)->fit - ((const ga_member *)l)->fit);
}

/* rank roulette of :96-128 / :309-326: slot = floor((-1 + sqrt(1 + 8 u)) / 2), then the next free slot upwards */
static int ga_roulette_pick(double rank_sum, char *taken, int count_slots) {
    const double u = rand_choice(1, rank_sum);
    int slot = (int)((-1 + sqrt(1 + 8 * u)) / 2.0);
    while (slot < count_slots - 1 && taken[slot]) slot++;
    if (taken[slot]) return -1;
    taken[slot] = 1;
    return slot;
}

/* :78-131 */
static void ga_select_parents(ga_member *pop, int *parents, int want, int pop_size) {
    char *taken = calloc((size_t)pop_size, 1);
    qsort(pop, (size_t)pop_size, sizeof(ga_member), ga_by_fitness_desc);
    const double rank_sum = pop_size * (pop_size + 1) / 2;
    for (int got = 0; got < want;) {
        const int s = ga_roulette_pick(rank_sum, taken, pop_size);
        if (s >= 0) parents[got++] = s;
    }
    free(taken);
}

/* :143-229 */
static void ga_crossover(int n, const int *p1, const int *p2, int *child, char *seen) {
    memset(seen, 0, (size_t)n);
    const double u = URAND();
    if (u < GA_CROSSOVER_SPLIT) {                       /* method 1, :150-175 */
        const int cut = rand_choice(0, n);
        int w = 0;
        for (int k = 0; k < n; k++) {
            if (k <= cut) { seen[p1[k]] = 1; child[w] = p1[k]; }
            else { if (seen[p2[k]]) continue; child[w] = p2[k]; }
            w++;
        }
        if (w < n) for (int k = 0; k <= cut; k++) { if (seen[p2[k]]) continue; child[w++] = p2[k]; }
        return;
    }
    int lo = rand_choice(0, n), hi = rand_choice(0, n);  /* method 2, :176-226 */
    if (lo > hi) { const int t = lo; lo = hi; hi = t; }
    if (lo == hi) { if (lo > 0) lo -= 1; else hi += 1; }
    int placed = 0;
    for (int k = lo; k <= hi; k++) { seen[p1[k]] = 1; child[k] = p1[k]; placed++; }
    for (int src = hi + 1, dst = hi + 1; placed < n; src++) {
        const int g = p2[src % n];
        if (!seen[g]) { child[dst % n] = g; placed++; dst++; }
    }
}

/* :375-446 without its 2-opt branch (handled by the caller); returns 1 if that branch was drawn */
static int ga_mutate_one(int n, int *genes, double two_opt_prob) {
    const double u = URAND();
    if (!(u < GA_MUTATION_RATE)) return 0;
    const double method = URAND();
    if (!(method > two_opt_prob)) return 1;
    int lo = rand_choice(0, n - 1), hi = rand_choice(0, n - 1);
    if (lo > hi) { const int t = lo; lo = hi; hi = t; }
    if (lo == hi) { if (lo > 0) lo -= 1; else hi += 1; }
    for (int k = 0, a = lo, b = hi; k < (hi - lo) / 2; k++, a++, b--) { const int t = genes[a]; genes[a] = genes[b]; genes[b] = t; }
    return 0;
}

/* :266-331 -- including the reference's aliasing: `total` holds shallow copies, so a population slot that
 * was already overwritten can be copied again later with its old fitness */
static void ga_choose_survivors(int n, ga_member *pop, int pop_size, const ga_member *kids, int kid_count) {
    const int total_n = pop_size + kid_count;
    ga_member *total = calloc((size_t)total_n, sizeof(ga_member));
    char *taken = calloc((size_t)total_n, 1);
    int w = 0;
    for (int k = 0; k < kid_count; k++) total[w++] = kids[k];
    for (int k = 0; k < pop_size; k++) total[w++] = pop[k];
    qsort(total, (size_t)total_n, sizeof(ga_member), ga_by_fitness_desc);
    const double rank_sum = total_n * (total_n + 1) / 2;
    for (int got = 0; got < pop_size;) {
        const int s = ga_roulette_pick(rank_sum, taken, total_n);
        if (s < 0) continue;
        memmove(pop[got].genes, total[s].genes, sizeof(int) * (size_t)n);
        pop[got].fit = total[s].fit;
        got++;
    }
    free(total); free(taken);
}

/* Mutation method 3 for a batch of offspring on several devices: the offspring are independent (each is alg_2opt on a private
 * copy, genetic.c:426-443), so GPU g takes a contiguous block of them; one thread per GPU and generation, the context and the
 * device instance of a GPU live for the whole run. */
typedef struct { int dev, n, count; instance *inst; tsp_dev_ctx *ctx; tsp_dev_inst *dinst; int *succ; double *obj; int rc; } ga_gpu;

static void *ga_gpu_run(void *arg) {
    ga_gpu *g = arg;
    g->rc = 0;
    if (!g->ctx) g->rc = tsp_dev_open(g->dev, &g->ctx);
    if (!g->rc && !g->dinst) g->rc = tsp_dev_inst_create(g->ctx, (const double *)g->inst->nodes, g->n, (int)g->inst->weight_type,
                                                         g->inst->params.integer_cost ? 1 : 0, &g->dinst);
    if (!g->rc && g->count > 0) g->rc = tsp_dev_two_opt(g->dinst, TSP_2OPT_FIRST, TSP_ENGINE_AUTO, g->count, g->succ, 1, g->n, g->obj, 2.0 /* :432 */, NULL);
    return NULL;
}

static void ga_two_opt_over_gpus(ga_gpu *pool, int gpus, int n, int n2, int *succ, double *obj) {
    pthread_t th[64];
    for (int g = 0; g < gpus; g++) {
        const int lo = (int)((long long)n2 * g / gpus), hi = (int)((long long)n2 * (g + 1) / gpus);
        pool[g].count = hi - lo; pool[g].succ = succ + (size_t)lo * n; pool[g].obj = obj + lo;
        if (pthread_create(&th[g], NULL, ga_gpu_run, &pool[g])) LOG_E("pthread_create failed");
    }
    for (int g = 0; g < gpus; g++) pthread_join(th[g], NULL);
    for (int g = 0; g < gpus; g++) if (pool[g].rc < 0) dev_fail("tsp_dev_two_opt (GA mutation, one of the GPUs)", pool[g].rc);
}

/* :448-565 with a cap on the number of generations in addition to the time limit */
int tsp_host_genetic_gpus(instance *inst, long long max_generations, double two_opt_prob, int gpus) {
    const int n = inst->num_nodes;
    struct timeval t0, t1;
    gettimeofday(&t0, 0);
    const int pop_size = GA_POPULATION, parent_count = (int)(pop_size * GA_PARENT_RATE), kid_count = parent_count;
    int *pop_slab = calloc((size_t)pop_size * n, sizeof(int)), *kid_slab = calloc((size_t)kid_count * n, sizeof(int));
    ga_member *pop = calloc((size_t)pop_size, sizeof(ga_member)), *kids = calloc((size_t)kid_count, sizeof(ga_member));
    double *fit = malloc(sizeof(double) * (size_t)pop_size);
    int *two_opt_kids = malloc(sizeof(int) * (size_t)kid_count);
    int *two_opt_succ = malloc(sizeof(int) * (size_t)kid_count * n);
    double *two_opt_obj = malloc(sizeof(double) * (size_t)kid_count);
    for (int k = 0; k < pop_size; k++) {
        pop[k].genes = pop_slab + (size_t)k * n;
        const double u = URAND();                            /* :463 */
        if (u < GA_HEURISTIC_INIT) {
            const int start = rand_choice(0, n);
            grasp(inst, start);
            for (int q = 0, v = start; q < n; q++) { pop[k].genes[q] = v; v = inst->solution.edges[v].j; }
        } else {                                             /* random_generation :349-364 */
            for (int q = 0; q < n; q++) pop[k].genes[q] = q;
            for (int q = 0; q < n; q++) {
                const int a = rand_choice(0, n), b = rand_choice(0, n);
                const int t = pop[k].genes[a]; pop[k].genes[a] = pop[k].genes[b]; pop[k].genes[b] = t;
            }
        }
    }
    fitness_batch(inst, pop_slab, pop_size, fit);            /* :481, whole population in one launch */
    for (int k = 0; k < pop_size; k++) pop[k].fit = fit[k];
    for (int k = 0; k < kid_count; k++) kids[k].genes = kid_slab + (size_t)k * n;
    const int time_limit = inst->params.time_limit > 0 ? inst->params.time_limit : DEFAULT_TIME_LIM;
    int *parents = calloc((size_t)parent_count, sizeof(int));
    char *seen = malloc((size_t)n);
    double incumbent = DBL_MAX;
    int status = 0;
    if (gpus > 64 || (gpus > 0 && tsp_dev_count() < gpus)) LOG_E("HEU_Genetic: %d GPUs asked for, %d visible", gpus, tsp_dev_count());
    ga_gpu *pool = gpus > 0 ? calloc((size_t)gpus, sizeof(ga_gpu)) : NULL;
    for (int g = 0; g < gpus; g++) { pool[g].dev = g; pool[g].n = n; pool[g].inst = inst; }
    for (long long gen = 0; max_generations < 0 || gen < max_generations; gen++) {
        gettimeofday(&t1, 0);
        if (get_elapsed_time(t0, t1) > time_limit) { status = TIME_LIMIT_EXCEEDED; break; }
        double best = DBL_MAX; int best_k = 0;               /* fitness_metrics :333-347 */
        for (int k = 0; k < pop_size; k++) if (pop[k].fit < best) { best = pop[k].fit; best_k = k; }
        if (best < incumbent) {                              /* :518-526, from_chromosome_to_edges :33-42 */
            incumbent = best;
            inst->solution.obj_best = best;
            const int *g = pop[best_k].genes;
            for (int q = 0; q < n; q++) { inst->solution.edges[g[q]].i = g[q]; inst->solution.edges[g[q]].j = g[q + 1 == n ? 0 : q + 1]; }
        }
        ga_select_parents(pop, parents, parent_count, pop_size);
        for (int k = 0; k < parent_count; k++)               /* procreate :240-256 */
            ga_crossover(n, pop[parents[k]].genes, pop[parents[(k + 1) % parent_count]].genes, kids[k].genes, seen);
        fitness_batch(inst, kid_slab, kid_count, fit);       /* :251, all offspring in one launch (no draws in between) */
        for (int k = 0; k < kid_count; k++) kids[k].fit = fit[k];
        /* mutation :375-446.  Method 3 (alg_2opt on a private copy, :426-443) draws nothing and touches only its own
         * offspring, so the offspring that drew it are refined together after the loop: one batched device call per
         * generation instead of one copy_instance + alg_2opt per offspring. */
        int n2 = 0;
        for (int k = 0; k < kid_count; k++)
            if (ga_mutate_one(n, kids[k].genes, two_opt_prob)) two_opt_kids[n2++] = k;
        if (n2 > 0) {
            for (int m = 0; m < n2; m++) {
                const int *g = kids[two_opt_kids[m]].genes;
                int *sp = two_opt_succ + (size_t)m * n;
                for (int q = 0; q < n; q++) sp[g[q]] = g[q + 1 == n ? 0 : q + 1];     /* from_chromosome_to_edges :33-42 */
                two_opt_obj[m] = inst->solution.obj_best;                              /* copy_instance keeps obj_best; alg_2opt adds deltas to it */
            }
            if (gpus > 0) ga_two_opt_over_gpus(pool, gpus, n, n2, two_opt_succ, two_opt_obj);
            else {
                pthread_mutex_lock(&g_lock);
                int rc = tsp_dev_two_opt(dev_inst_locked(inst), TSP_2OPT_FIRST, TSP_ENGINE_AUTO, n2, two_opt_succ, 1, n, two_opt_obj,
                                         2.0 /* :432 */, NULL);
                pthread_mutex_unlock(&g_lock);
                if (rc < 0) dev_fail("tsp_dev_two_opt (GA mutation)", rc);
            }
            for (int m = 0; m < n2; m++) {                                             /* :436-441: the walk from node 0 */
                int *g = kids[two_opt_kids[m]].genes;
                const int *sp = two_opt_succ + (size_t)m * n;
                for (int q = 0, v = 0; q < n; q++) { g[q] = v; v = sp[v]; }
            }
        }
        ga_choose_survivors(n, pop, pop_size, kids, kid_count);
    }
    free(pop_slab); free(kid_slab); free(pop); free(kids); free(fit); free(parents); free(seen);
    free(two_opt_kids); free(two_opt_succ); free(two_opt_obj);
    for (int g = 0; g < gpus; g++) {
        if (pool[g].dinst) tsp_dev_inst_destroy(pool[g].dinst);
        if (pool[g].ctx) tsp_dev_close(pool[g].ctx);
    }
    free(pool);
    return status;
}

int tsp_host_genetic_ex(instance *inst, long long max_generations, double two_opt_prob) {
    return tsp_host_genetic_gpus(inst, max_generations, two_opt_prob, g_cli_gpus > 1 ? g_cli_gpus : 0);   /* -gpus G of the CLI */
}
int tsp_host_genetic(instance *inst, long long max_generations) { return tsp_host_genetic_ex(inst, max_generations, GA_TWO_OPT_MUT); }
int HEU_Genetic(instance *inst) { return tsp_host_genetic(inst, -1); }

/* genetic.c:51-60 for `count` chromosomes of n nodes each */
int fitness_batch(instance *inst, const int *chromosomes, int count, double *fitness_out) {
    pthread_mutex_lock(&g_lock);
    int rc = tsp_dev_perm_cost(dev_inst_locked(inst), count, chromosomes, inst->num_nodes, fitness_out);
    pthread_mutex_unlock(&g_lock);
    if (rc < 0) dev_fail("tsp_dev_perm_cost", rc);
    return rc;
}

/* ---- multi-start across GPUs ----------------------------------------------------------------------
 * Generalises HEU_Grasp_iter's loop (heuristics.c:510-544: random start :519, grasp(), keep the strictly better one :534):
 * every start is refined by alg_2opt, start k runs on rank k % world, and the ranks agree on the winner with ONE RCCL
 * all-reduce(min) of (true cost << 24 | k) and ONE broadcast of its successor list (tsp_dev_multistart_* of the C ABI).
 * The same skeleton serves the population job of BASELINE configs[4]: random individuals (genetic.c:349-364), individual k
 * on rank k % world, alg_2opt on each (the mutation-3 path, genetic.c:426-443).
 *
 * Failure agreement.  Between its shard and the collective a rank never exits: whatever went wrong on it (no device, a HIP
 * error in the shard, a cost the packed word cannot carry) is CONTRIBUTED to the reduction as a value that wins the minimum
 * (EPI_FAIL / -inf), so every rank learns of it in the same collective and every rank returns the same negative code --
 * nobody is left waiting inside ncclAllReduce for a peer that has gone.  Only then do the callers LOG_E. */

enum { SHARD_GRASP = 0, SHARD_POPULATION = 1 };
#define NO_RESULT_PACKED ((int64_t)1 << 62)
#define EPI_FAIL ((int64_t)-1)          /* = multistart.PACK_ERROR of the Python launcher: smaller than every packed value */

static int shard_count(int total, int rank, int world) { return total > rank ? (total - rank + world - 1) / world : 0; }

/* The libc stream of the whole job in the reference's draw order; keeps the draws of the units k % world == rank (every rank
 * walks the whole stream so that unit k is the same everywhere).
 *   GRASP starts: node (heuristics.c:519), then the n URAND() of grasp() (:127)      -> node[], u[] (n doubles per start)
 *   individuals : random_generation (genetic.c:349-364): identity, then n swaps of two rand_choice(0, n) positions -> perm[] */
static int draw_shard(int kind, int n, int total, int rank, int world, int **gid_out, int **node_out, double **u_out, int **perm_out) {
    const int mine = shard_count(total, rank, world), cap = mine ? mine : 1;
    int *gid = malloc(sizeof(int) * (size_t)cap), *node = NULL, *perm = NULL, *scratch = NULL;
    double *u = NULL;
    if (kind == SHARD_GRASP) { node = malloc(sizeof(int) * (size_t)cap); u = malloc(sizeof(double) * (size_t)cap * n); }
    else { perm = malloc(sizeof(int) * (size_t)cap * n); scratch = malloc(sizeof(int) * (size_t)n); }
    if (!gid || (kind == SHARD_GRASP ? (!node || !u) : (!perm || !scratch))) LOG_E("multistart: out of memory");
    int m = 0;
    for (int k = 0; k < total; k++) {
        const int keep = k % world == rank;
        if (kind == SHARD_GRASP) {
            const int nd = (int)(URAND() * (n - 1));
            if (keep) { node[m] = nd; for (int q = 0; q < n; q++) u[(size_t)m * n + q] = URAND(); }
            else for (int q = 0; q < n; q++) (void)random();
        } else {
            int *g = keep ? perm + (size_t)m * n : scratch;          /* the swaps depend on nothing but the draws */
            if (keep) for (int q = 0; q < n; q++) g[q] = q;
            for (int q = 0; q < n; q++) {
                const int a = rand_choice(0, n), b = rand_choice(0, n);
                if (keep) { const int t = g[a]; g[a] = g[b]; g[b] = t; }
            }
        }
        if (keep) gid[m++] = k;
    }
    free(scratch);
    *gid_out = gid;
    if (node_out) *node_out = node;
    if (u_out) *u_out = u;
    if (perm_out) *perm_out = perm;
    return mine;
}

/* One shard on one device instance; best (cost, global id, tour) of the shard, first strictly better unit in id order
 * (heuristics.c:534).  GRASP: construct + alg_2opt + TRUE cost (fitness of the walk from node 0: the reported obj carries
 * GRASP's double-counted closing edge).  Population: fitness of the individual, alg_2opt adding its deltas to it.
 * costs_out / succ_out / stats_out (may be NULL) are indexed by the GLOBAL unit id. */
static int refine_shard(tsp_dev_inst *d, int kind, int n, int mine, const int *gid, const int *node, const double *u, const int *perm,
                        double limit, double *best_cost, int *best_k, int *best_succ, double *costs_out, int *succ_out,
                        tsp_two_opt_stats *stats_out) {
    *best_cost = DBL_MAX; *best_k = -1;
    if (mine <= 0) return 0;
    int *succ = malloc(sizeof(int) * (size_t)mine * n);
    int *walk = malloc(sizeof(int) * (size_t)mine * n);
    double *obj = malloc(sizeof(double) * (size_t)mine), *cost = malloc(sizeof(double) * (size_t)mine);
    tsp_two_opt_stats *st = stats_out ? calloc((size_t)mine, sizeof *st) : NULL;
    if (!succ || !walk || !obj || !cost || (stats_out && !st)) LOG_E("multistart: out of memory");
    int rc;
    if (kind == SHARD_GRASP) {
        rc = tsp_dev_construct(d, TSP_CONSTRUCT_GRASP, mine, node, u, succ, 1, n, obj, NULL);
        if (rc >= 0) rc = tsp_dev_two_opt(d, TSP_2OPT_FIRST, TSP_ENGINE_AUTO, mine, succ, 1, n, obj, limit, st);
        if (rc >= 0) {
            for (int b = 0; b < mine; b++) { int v = 0; for (int q = 0; q < n; q++) { walk[(size_t)b * n + q] = v; v = succ[(size_t)b * n + v]; } }
            const int rc2 = tsp_dev_perm_cost(d, mine, walk, n, cost);
            if (rc2) rc = rc2;
        }
    } else {
        for (int b = 0; b < mine; b++) {                           /* from_chromosome_to_edges, genetic.c:33-42 */
            const int *g = perm + (size_t)b * n;
            int *sp = succ + (size_t)b * n;
            for (int q = 0; q < n; q++) sp[g[q]] = g[q + 1 == n ? 0 : q + 1];
        }
        rc = tsp_dev_perm_cost(d, mine, perm, n, cost);            /* fitness, genetic.c:51-60 */
        if (rc >= 0) rc = tsp_dev_two_opt(d, TSP_2OPT_FIRST, TSP_ENGINE_AUTO, mine, succ, 1, n, cost, limit, st);   /* obj_best += delta */
    }
    if (rc >= 0) {
        int bm = 0;
        for (int b = 0; b < mine; b++) {
            if (cost[b] < *best_cost) { *best_cost = cost[b]; *best_k = gid[b]; bm = b; }
            if (costs_out) costs_out[gid[b]] = cost[b];
            if (succ_out) memcpy(succ_out + (size_t)gid[b] * n, succ + (size_t)b * n, sizeof(int) * (size_t)n);
            if (stats_out) stats_out[gid[b]] = st[b];
        }
        memcpy(best_succ, succ + (size_t)bm * n, sizeof(int) * (size_t)n);
    }
    free(succ); free(walk); free(obj); free(cost); free(st);
    return rc;
}

/* ---- the collectives of the epilogue: RCCL through libtsp_hip.so, unless the caller plugged in its own ---------------- */
static tsp_host_collectives g_coll;    /* all NULL: RCCL */
static int g_coll_set = 0;

void tsp_host_set_collectives(const tsp_host_collectives *c) {
    pthread_mutex_lock(&g_lock);
    if (c) { g_coll = *c; g_coll_set = 1; } else { memset(&g_coll, 0, sizeof g_coll); g_coll_set = 0; }
    pthread_mutex_unlock(&g_lock);
}

static long long proc_start_ticks(long pid) {   /* field 22 of /proc/<pid>/stat: start time in clock ticks since boot */
    char path[64], buf[1024];
    snprintf(path, sizeof path, "/proc/%ld/stat", pid);
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    const size_t got = fread(buf, 1, sizeof buf - 1, fp);
    fclose(fp);
    buf[got] = 0;
    char *p = strrchr(buf, ')');                 /* the command name may hold blanks and brackets */
    if (!p) return -1;
    long long v = -1;
    int field = 2;
    for (char *tok = strtok(p + 1, " "); tok; tok = strtok(NULL, " "))
        if (++field == 22) { v = atoll(tok); break; }
    return v;
}

/* What rank 0 leaves for the other ranks of its job: the RCCL id plus who wrote it.  A reader accepts the record only while
 * that very process (pid AND start time) is alive -- the record of a run that crashed, or of a pid that has since been
 * reused, is stale by construction, whatever its age.  (One node: SURVEY 8(e) shards over the GPUs of one host.) */
typedef struct { char magic[8]; char id[TSP_COMM_ID_BYTES]; long long pid, start_ticks; } id_record;
static const char k_id_magic[8] = {'T', 'S', 'P', 'R', 'I', 'D', '0', '2'};

static void id_file_path(char *path, size_t cap) {
    const char *f = getenv("TSP_RCCL_ID_FILE"), *port = getenv("MASTER_PORT");
    if (f && *f) snprintf(path, cap, "%s", f);
    else snprintf(path, cap, "/tmp/tsp_rccl_id.%ld.%ld.%s", (long)getuid(), (long)getppid(), port && *port ? port : "0");
}

static int id_record_is_live(const id_record *r) {
    return !memcmp(r->magic, k_id_magic, sizeof k_id_magic) && r->pid > 0 && proc_start_ticks((long)r->pid) == r->start_ticks &&
           r->start_ticks >= 0;
}

static int read_id_record(const char *path, id_record *r) {
    const int fd = open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return 0;
    const ssize_t got = read(fd, r, sizeof *r);
    close(fd);
    return got == (ssize_t)sizeof *r;
}

/* Diagnostics (and the CPU tests): what a rank would make of the id file at `path` (NULL: this process's default path) --
 * 0 nothing readable there (a symlink is never followed), 1 a record whose writer is alive, 2 a stale record. */
int tsp_host_rccl_id_file_state(const char *path) {
    char dflt[512];
    if (!path) { id_file_path(dflt, sizeof dflt); path = dflt; }
    id_record r;
    if (!read_id_record(path, &r)) return 0;
    return id_record_is_live(&r) ? 1 : 2;
}

/* rank 0, BEFORE its shard: whatever lies at the path is either stale (removed) or belongs to a running job (refused) */
static void id_file_claim(void) {
    char path[512];
    id_file_path(path, sizeof path);
    id_record old;
    if (read_id_record(path, &old) && id_record_is_live(&old) && old.pid != (long long)getpid())
        LOG_E("%s belongs to a running job (pid %lld): give this job its own TSP_RCCL_ID_FILE or MASTER_PORT", path, old.pid);
    (void)unlink(path);
}

/* One process per GPU: the communicator of this process, formed on first use from RANK / WORLD_SIZE.  Rank 0 obtains the
 * RCCL id and leaves it in a file (TSP_RCCL_ID_FILE, default /tmp/tsp_rccl_id.<uid>.<launcher pid>.<MASTER_PORT>; created
 * with O_EXCL | O_NOFOLLOW, mode 0600, then renamed into place), the others wait for a record whose writer is alive; the
 * file is removed once every rank is in.  Returns NULL (text in *why) instead of exiting: the caller is about to enter a
 * collective-free failure path of its own. */
static tsp_dev_comm *comm_try_locked(int rank, int world, char *why, size_t why_cap) {
    if (g_comm) return g_comm;
    if (!g_ctx) { snprintf(why, why_cap, "no device context"); return NULL; }
    char path[512];
    id_file_path(path, sizeof path);
    id_record rec;
    memset(&rec, 0, sizeof rec);
    if (rank == 0) {
        int rc = tsp_dev_comm_unique_id(rec.id);
        if (rc) { snprintf(why, why_cap, "tsp_dev_comm_unique_id failed with %d %s", rc, tsp_dev_comm_last_error()); return NULL; }
        memcpy(rec.magic, k_id_magic, sizeof k_id_magic);
        rec.pid = (long long)getpid();
        rec.start_ticks = proc_start_ticks((long)getpid());
        char tmp[600];
        snprintf(tmp, sizeof tmp, "%s.%ld.tmp", path, (long)getpid());
        (void)unlink(tmp);
        const int fd = open(tmp, O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
        if (fd < 0 || write(fd, &rec, sizeof rec) != (ssize_t)sizeof rec) { if (fd >= 0) close(fd); snprintf(why, why_cap, "cannot write the RCCL id to %.300s", tmp); return NULL; }
        close(fd);
        if (rename(tmp, path)) { snprintf(why, why_cap, "cannot publish the RCCL id as %.300s", path); return NULL; }
    } else {
        struct timeval t0, t1;
        gettimeofday(&t0, 0);
        for (;;) {
            if (read_id_record(path, &rec) && id_record_is_live(&rec)) break;
            gettimeofday(&t1, 0);
            if (get_elapsed_time(t0, t1) > 120.0) { snprintf(why, why_cap, "rank %d: no live RCCL id in %.300s after 120 s (is rank 0 running?)", rank, path); return NULL; }
            struct timespec ts = {0, 20 * 1000 * 1000};
            nanosleep(&ts, NULL);
        }
    }
    int rc = tsp_dev_comm_init_rank(g_ctx, world, rank, rec.id, &g_comm);
    if (rank == 0) (void)unlink(path);   /* init_rank is collective: every rank has read it (or this rank failed and nobody should) */
    if (rc) { snprintf(why, why_cap, "tsp_dev_comm_init_rank failed with %d %s %s", rc, tsp_dev_comm_last_error(), tsp_dev_last_error()); g_comm = NULL; return NULL; }
    return g_comm;
}

static int coll_allreduce_i64(int rank, int world, int64_t local, int64_t *out, char *why, size_t cap) {
    if (g_coll_set) return g_coll.allreduce_min_i64(g_coll.self, local, out);
    tsp_dev_comm *cm = comm_try_locked(rank, world, why, cap);
    if (!cm) return TSP_DEV_E_COMM;
    const int rc = tsp_dev_multistart_allreduce(cm, local, out);
    if (rc) snprintf(why, cap, "all-reduce(min) failed with %d %s %s", rc, tsp_dev_comm_last_error(), tsp_dev_last_error());
    return rc;
}
static int coll_allreduce_f64(int rank, int world, double local, double *out, char *why, size_t cap) {
    if (g_coll_set) return g_coll.allreduce_min_f64(g_coll.self, local, out);
    tsp_dev_comm *cm = comm_try_locked(rank, world, why, cap);
    if (!cm) return TSP_DEV_E_COMM;
    const int rc = tsp_dev_multistart_allreduce_f64(cm, local, out);
    if (rc) snprintf(why, cap, "all-reduce(min, double) failed with %d %s %s", rc, tsp_dev_comm_last_error(), tsp_dev_last_error());
    return rc;
}
static int coll_bcast(int rank, int world, int root, int *buf, int stride, int n, char *why, size_t cap) {
    if (g_coll_set) return g_coll.bcast_i32(g_coll.self, root, buf, stride, n);
    tsp_dev_comm *cm = comm_try_locked(rank, world, why, cap);
    if (!cm) return TSP_DEV_E_COMM;
    const int rc = tsp_dev_multistart_bcast_tour(cm, root, buf, stride, n);
    if (rc) snprintf(why, cap, "broadcast failed with %d %s %s", rc, tsp_dev_comm_last_error(), tsp_dev_last_error());
    return rc;
}

static __thread char t_epilogue_error[512];
const char *tsp_host_multistart_last_error(void) { return t_epilogue_error; }

/* The epilogue, COLLECTIVE over `world` ranks: every rank passes the outcome of its shard -- shard_rc (< 0: the shard failed),
 * *best / *best_k (best_k < 0: empty shard), the shard's best tour in inst->solution.edges -- and every rank returns with the
 * global winner in *best / *best_k / inst->solution, or with the SAME negative code on every rank:
 *   TSP_HOST_E_PEER     some rank's shard failed or holds a cost the reduction cannot carry (this rank's text, if it was
 *                       this rank, in tsp_host_multistart_last_error());
 *   TSP_DEV_E_*         the transport itself failed on THIS rank (its peers leave their collective by its timeout).
 * Integer costs (the reference's default): ONE all-reduce(min) of (cost << 24 | id), a failed rank contributes EPI_FAIL.
 * --fcost (src/utility.c:285; `< bestobj` on doubles, heuristics.c:534): TWO reductions -- min of the double (a failed rank
 * contributes -inf), then min of the id among the ranks that hold that cost (ties -> lowest id: the strict `<` in stream order).
 * Then ONE broadcast of the winner's successor list from rank id % world, in place (stride 2). */
int tsp_host_multistart_epilogue(instance *inst, int rank, int world, int shard_rc, double *best, int *best_k) {
    const int n = inst->num_nodes;
    char why[400] = "";
    t_epilogue_error[0] = 0;
    int failed = shard_rc < 0;
    if (failed) snprintf(t_epilogue_error, sizeof t_epilogue_error, "rank %d: shard failed with %d %s", rank, shard_rc, tsp_dev_last_error());
    int rc = 0, owner = -1;
    pthread_mutex_lock(&g_lock);
    if (inst->params.integer_cost) {
        int64_t mine = NO_RESULT_PACKED, win = 0;
        if (failed) mine = EPI_FAIL;
        else if (*best_k >= 0 && tsp_dev_multistart_pack(*best, *best_k, &mine)) {
            mine = EPI_FAIL;
            snprintf(t_epilogue_error, sizeof t_epilogue_error, "rank %d: cost %f of unit %d does not fit the packed all-reduce (cost << 24 | id)", rank, *best, *best_k);
        }
        rc = coll_allreduce_i64(rank, world, mine, &win, why, sizeof why);
        if (!rc && win < 0) rc = TSP_HOST_E_PEER;
        if (!rc) {
            if (win == NO_RESULT_PACKED) { *best = DBL_MAX; *best_k = -1; }
            else { *best = (double)(win >> 24); *best_k = (int)(win & 0xffffff); owner = *best_k % world; }
        }
    } else {
        double mine = failed || isnan(*best) ? -INFINITY : (*best_k >= 0 ? *best : INFINITY), cmin = 0.0;
        if (!failed && isnan(*best)) snprintf(t_epilogue_error, sizeof t_epilogue_error, "rank %d: cost of unit %d is not a number", rank, *best_k);
        rc = coll_allreduce_f64(rank, world, mine, &cmin, why, sizeof why);
        if (!rc && cmin == -INFINITY) rc = TSP_HOST_E_PEER;
        if (!rc) {
            int64_t idm = *best_k >= 0 && mine == cmin ? (int64_t)*best_k : NO_RESULT_PACKED, idw = 0;
            rc = coll_allreduce_i64(rank, world, idm, &idw, why, sizeof why);
            if (!rc && idw < 0) rc = TSP_HOST_E_PEER;
            if (!rc) {
                if (idw == NO_RESULT_PACKED) { *best = DBL_MAX; *best_k = -1; }
                else { *best = cmin; *best_k = (int)idw; owner = *best_k % world; }
            }
        }
    }
    if (!rc && owner >= 0) rc = coll_bcast(rank, world, owner, &inst->solution.edges[0].j, 2, n, why, sizeof why);
    pthread_mutex_unlock(&g_lock);
    if (rc && rc != TSP_HOST_E_PEER) snprintf(t_epilogue_error, sizeof t_epilogue_error, "rank %d: %s", rank, why);
    if (rc == TSP_HOST_E_PEER && !t_epilogue_error[0])
        snprintf(t_epilogue_error, sizeof t_epilogue_error, "rank %d: a peer reported a failed shard (or a cost the reduction cannot carry)", rank);
    if (!rc && *best_k >= 0) { stamp_edge_sources(inst); inst->solution.obj_best = *best; }
    return rc;
}

/* One rank's share, no communication, and no exit on a device failure: the units k % world == rank on this process's
 * device, the shard's best in inst->solution / *best / *best_k.  < 0: the failure, for the epilogue to carry. */
static int shard_try(instance *inst, int kind, int total, int rank, int world, double *best, int *best_k, double *costs_out,
                     int *succ_out, tsp_two_opt_stats *stats_out) {
    const int n = inst->num_nodes;
    int *gid, *node = NULL, *perm = NULL;
    double *u = NULL;
    const int mine = draw_shard(kind, n, total, rank, world, &gid, &node, &u, &perm);   /* the stream is walked whatever happens next */
    int *best_succ = malloc(sizeof(int) * (size_t)n);
    if (!best_succ) LOG_E("multistart: out of memory");
    *best = DBL_MAX; *best_k = -1;
    pthread_mutex_lock(&g_lock);
    int rc = 0;
    tsp_dev_inst *d = dev_inst_try_locked(inst, &rc);
    if (d) rc = refine_shard(d, kind, n, mine, gid, node, u, perm, limit_of(inst), best, best_k, best_succ, costs_out, succ_out, stats_out);
    pthread_mutex_unlock(&g_lock);
    if (rc >= 0 && *best_k >= 0) {
        for (int v = 0; v < n; v++) { inst->solution.edges[v].i = v; inst->solution.edges[v].j = best_succ[v]; }
        inst->solution.obj_best = *best;
    }
    free(gid); free(node); free(u); free(perm); free(best_succ);
    return rc < 0 ? rc : 0;
}

static int shard_args_ok(const instance *inst, int total, int rank, int world) {
    return inst && inst->nodes && inst->num_nodes >= 3 && total >= 1 && total < (1 << 24) && world >= 1 && rank >= 0 && rank < world;
}

int tsp_host_multistart_shard(instance *inst, int starts, int rank, int world, double *best_true_cost, int *best_start) {
    if (!shard_args_ok(inst, starts, rank, world)) return -1;
    double best; int best_k;
    const int rc = shard_try(inst, SHARD_GRASP, starts, rank, world, &best, &best_k, NULL, NULL, NULL);
    if (rc < 0) dev_fail("multistart", rc);
    if (best_true_cost) *best_true_cost = best;
    if (best_start) *best_start = best_k;
    return 0;
}

int tsp_host_population_shard(instance *inst, int individuals, int rank, int world, double *best_cost, int *best_individual,
                              double *costs_out, int *succ_out, tsp_two_opt_stats *stats_out) {
    if (!shard_args_ok(inst, individuals, rank, world)) return -1;
    double best; int best_k;
    const int rc = shard_try(inst, SHARD_POPULATION, individuals, rank, world, &best, &best_k, costs_out, succ_out, stats_out);
    if (rc < 0) dev_fail("population", rc);
    if (best_cost) *best_cost = best;
    if (best_individual) *best_individual = best_k;
    return 0;
}

/* shard + (world > 1, or TSP_FORCE_COMM=1 for the tests) the collective epilogue; exits -- on EVERY rank -- only after the
 * ranks have agreed that something failed */
static int sharded_job(instance *inst, int kind, int total, int rank, int world, double *best_out, int *best_k_out, double *costs_out,
                       int *succ_out, tsp_two_opt_stats *stats_out) {
    if (!shard_args_ok(inst, total, rank, world)) return -1;
    const char *force = getenv("TSP_FORCE_COMM");   /* tests: run the collectives with a single rank too */
    const int collective = world > 1 || (force && *force == '1');
    if (collective && rank == 0 && !g_coll_set) id_file_claim();   /* before the shard: the peers look for the id after theirs */
    double best; int best_k;
    int rc = shard_try(inst, kind, total, rank, world, &best, &best_k, costs_out, succ_out, stats_out);
    if (collective) {
        rc = tsp_host_multistart_epilogue(inst, rank, world, rc, &best, &best_k);
        if (rc) LOG_E("multi-GPU %s: the ranks agreed to fail (%d): %s", kind == SHARD_GRASP ? "multi-start" : "population", rc, t_epilogue_error);
    } else if (rc < 0) dev_fail(kind == SHARD_GRASP ? "multistart" : "population", rc);
    if (best_out) *best_out = best;
    if (best_k_out) *best_k_out = best_k;
    return 0;
}

/* BASELINE config 4 end to end.  world == 1: this process refines every start.  world > 1 (one process per GPU, COLLECTIVE:
 * every rank must call it): this rank refines its shard on its device, then the ranks run the epilogue above, so that EVERY
 * rank returns the global winner in inst->solution, *best_true_cost and *best_start. */
int HEU_2opt_grasp_multistart(instance *inst, int starts, int rank, int world, double *best_true_cost, int *best_start) {
    return sharded_job(inst, SHARD_GRASP, starts, rank, world, best_true_cost, best_start, NULL, NULL, NULL);
}

/* BASELINE config 5 end to end: `individuals` random individuals (genetic.c:349-364, libc stream), individual k on rank
 * k % world, alg_2opt on each (genetic.c:426-443 without its 2 s limit), the best refined individual on every rank. */
int HEU_2opt_population_multistart(instance *inst, int individuals, int rank, int world, double *best_cost, int *best_individual,
                                   double *costs_out, int *succ_out, tsp_two_opt_stats *stats_out) {
    return sharded_job(inst, SHARD_POPULATION, individuals, rank, world, best_cost, best_individual, costs_out, succ_out, stats_out);
}

/* The same jobs in ONE process on `gpus` devices (0 .. gpus-1): one thread, context and device instance per GPU, shard
 * k % gpus on GPU k, then ncclCommInitAll + the grouped reductions + broadcast.  The libc stream is drawn once,
 * up front, by the calling thread (random() is process-global, heuristics.c:519). */
typedef struct {
    instance *inst; int kind, dev, rank, world, total, mine; int *gid, *node, *perm; double *u; double limit;
    tsp_dev_ctx *ctx; tsp_dev_inst *dinst; double best; int best_k; int *best_succ; int rc; double seconds;
    double *costs_out; int *succ_out; tsp_two_opt_stats *stats_out;
} gpu_job;

static void *gpu_job_run(void *arg) {
    gpu_job *j = arg;
    struct timeval t0, t1;
    gettimeofday(&t0, 0);
    const int n = j->inst->num_nodes;
    j->rc = tsp_dev_open(j->dev, &j->ctx);
    if (!j->rc) j->rc = tsp_dev_inst_create(j->ctx, (const double *)j->inst->nodes, n, (int)j->inst->weight_type,
                                            j->inst->params.integer_cost ? 1 : 0, &j->dinst);
    if (!j->rc) j->rc = refine_shard(j->dinst, j->kind, n, j->mine, j->gid, j->node, j->u, j->perm, j->limit, &j->best, &j->best_k,
                                     j->best_succ, j->costs_out, j->succ_out, j->stats_out);
    gettimeofday(&t1, 0);
    j->seconds = get_elapsed_time(t0, t1);
    return NULL;
}

static int job_over_gpus(instance *inst, int kind, int total, int gpus, double *best_out, int *best_k_out, double *shard_seconds,
                         double *costs_out, int *succ_out, tsp_two_opt_stats *stats_out) {
    const int n = inst->num_nodes;
    if (!shard_args_ok(inst, total, 0, 1) || gpus < 1 || gpus > 64) return -1;
    if (tsp_dev_count() < gpus) LOG_E("%d GPUs asked for, %d visible", gpus, tsp_dev_count());
    gpu_job *jobs = calloc((size_t)gpus, sizeof(gpu_job));
    /* one pass over the stream per GPU would draw it `gpus` times: draw the whole job once and deal the units out */
    int *gid_all, *node_all = NULL, *perm_all = NULL;
    double *u_all = NULL;
    (void)draw_shard(kind, n, total, 0, 1, &gid_all, &node_all, &u_all, &perm_all);
    for (int g = 0; g < gpus; g++) {
        gpu_job *j = &jobs[g];
        j->mine = shard_count(total, g, gpus);
        const int cap = j->mine ? j->mine : 1;
        j->gid = malloc(sizeof(int) * (size_t)cap);
        if (kind == SHARD_GRASP) { j->node = malloc(sizeof(int) * (size_t)cap); j->u = malloc(sizeof(double) * (size_t)cap * n); }
        else j->perm = malloc(sizeof(int) * (size_t)cap * n);
        for (int m = 0; m < j->mine; m++) {
            const int k = g + m * gpus;
            j->gid[m] = k;
            if (kind == SHARD_GRASP) { j->node[m] = node_all[k]; memcpy(j->u + (size_t)m * n, u_all + (size_t)k * n, sizeof(double) * (size_t)n); }
            else memcpy(j->perm + (size_t)m * n, perm_all + (size_t)k * n, sizeof(int) * (size_t)n);
        }
    }
    free(gid_all); free(node_all); free(u_all); free(perm_all);
    pthread_t *th = calloc((size_t)gpus, sizeof(pthread_t));
    for (int g = 0; g < gpus; g++) {
        gpu_job *j = &jobs[g];
        j->inst = inst; j->kind = kind; j->dev = g; j->rank = g; j->world = gpus; j->total = total; j->limit = limit_of(inst);
        j->best = DBL_MAX; j->best_k = -1; j->best_succ = malloc(sizeof(int) * (size_t)n);
        j->costs_out = costs_out; j->succ_out = succ_out; j->stats_out = stats_out;
        if (pthread_create(&th[g], NULL, gpu_job_run, j)) LOG_E("pthread_create failed");
    }
    for (int g = 0; g < gpus; g++) pthread_join(th[g], NULL);
    /* one process: a failed shard is known to this thread, nobody is inside a collective yet */
    for (int g = 0; g < gpus; g++)
        if (jobs[g].rc < 0) dev_fail("shard on one of the GPUs", jobs[g].rc);
    /* the epilogue: the reductions and the broadcast of the winner's tour from its owner, over RCCL, grouped */
    tsp_dev_ctx **ctxs = calloc((size_t)gpus, sizeof(tsp_dev_ctx *));
    tsp_dev_comm **comms = calloc((size_t)gpus, sizeof(tsp_dev_comm *));
    int64_t *loc = calloc((size_t)gpus, sizeof(int64_t)), *win = calloc((size_t)gpus, sizeof(int64_t));
    double *cloc = calloc((size_t)gpus, sizeof(double)), *cwin = calloc((size_t)gpus, sizeof(double));
    for (int g = 0; g < gpus; g++) ctxs[g] = jobs[g].ctx;
    int rc = tsp_dev_comm_init_all(ctxs, gpus, comms);
    double best = DBL_MAX;
    int best_k = -1;
    if (!rc && inst->params.integer_cost) {
        for (int g = 0; g < gpus; g++) {
            loc[g] = NO_RESULT_PACKED;
            if (jobs[g].best_k >= 0 && tsp_dev_multistart_pack(jobs[g].best, jobs[g].best_k, &loc[g]))
                LOG_E("cost %f of unit %d does not fit the packed all-reduce (cost << 24 | id)", jobs[g].best, jobs[g].best_k);
        }
        rc = tsp_dev_multistart_allreduce_group(comms, gpus, loc, win);
        for (int g = 1; g < gpus && !rc; g++) if (win[g] != win[0]) rc = TSP_DEV_E_COMM;   /* every rank must hold the same minimum */
        if (!rc && win[0] != NO_RESULT_PACKED) { best = (double)(win[0] >> 24); best_k = (int)(win[0] & 0xffffff); }
    } else if (!rc) {
        for (int g = 0; g < gpus; g++) cloc[g] = jobs[g].best_k >= 0 ? jobs[g].best : INFINITY;
        rc = tsp_dev_multistart_allreduce_f64_group(comms, gpus, cloc, cwin);
        for (int g = 0; g < gpus && !rc; g++) loc[g] = jobs[g].best_k >= 0 && jobs[g].best == cwin[g] ? (int64_t)jobs[g].best_k : NO_RESULT_PACKED;
        if (!rc) rc = tsp_dev_multistart_allreduce_group(comms, gpus, loc, win);
        for (int g = 1; g < gpus && !rc; g++) if (win[g] != win[0] || cwin[g] != cwin[0]) rc = TSP_DEV_E_COMM;
        if (!rc && win[0] != NO_RESULT_PACKED) { best = cwin[0]; best_k = (int)win[0]; }
    }
    if (!rc && best_k >= 0) {
        const int owner = best_k % gpus;
        int *tour = malloc(sizeof(int) * (size_t)n);
        rc = tsp_dev_multistart_bcast_tour_group(comms, gpus, owner, jobs[owner].best_succ, 1, n, (owner + 1) % gpus, tour);
        if (!rc) for (int v = 0; v < n; v++) { inst->solution.edges[v].i = v; inst->solution.edges[v].j = tour[v]; }
        free(tour);
        inst->solution.obj_best = best;
    }
    if (rc) LOG_E("multi-GPU collective failed with %d %s %s", rc, tsp_dev_comm_last_error(), tsp_dev_last_error());
    if (best_out) *best_out = best;
    if (best_k_out) *best_k_out = best_k;
    for (int g = 0; g < gpus; g++) {
        if (shard_seconds) shard_seconds[g] = jobs[g].seconds;
        tsp_dev_comm_destroy(comms[g]);
        if (jobs[g].dinst) tsp_dev_inst_destroy(jobs[g].dinst);
        if (jobs[g].ctx) tsp_dev_close(jobs[g].ctx);
        free(jobs[g].gid); free(jobs[g].node); free(jobs[g].u); free(jobs[g].perm); free(jobs[g].best_succ);
    }
    free(jobs); free(th); free(ctxs); free(comms); free(loc); free(win); free(cloc); free(cwin);
    return 0;
}

int tsp_host_multistart_gpus(instance *inst, int starts, int gpus, double *best_true_cost, int *best_start, double *shard_seconds) {
    return job_over_gpus(inst, SHARD_GRASP, starts, gpus, best_true_cost, best_start, shard_seconds, NULL, NULL, NULL);
}

int tsp_host_population_gpus(instance *inst, int individuals, int gpus, double *best_cost, int *best_individual, double *shard_seconds,
                             double *costs_out, int *succ_out, tsp_two_opt_stats *stats_out) {
    return job_over_gpus(inst, SHARD_POPULATION, individuals, gpus, best_cost, best_individual, shard_seconds, costs_out, succ_out, stats_out);
}

/* ---- src/solver.c:262-299 ------------------------------------------------------------------------------- */
int TSP_heuc(instance *inst) {
    const char *ws = getenv("WORLD_SIZE"), *rk = getenv("RANK");
    const int world_env = ws && atoi(ws) > 1 && rk ? atoi(ws) : 1, rank_env = world_env > 1 ? atoi(rk) : 0;
    if (inst->params.seed >= 0) srandom((unsigned)inst->params.seed);
    inst->num_columns = (long)inst->num_nodes * (inst->num_nodes - 1) / 2;
    inst->solution.edges = calloc((size_t)inst->num_nodes, sizeof(edge));
    struct timeval t0, t1;
    gettimeofday(&t0, 0);
    switch (inst->params.method.id) { /* solver.c:114-151, heuristics of this round */
    case SOLVE_GREEDY: HEU_greedy(inst); break;
    case SOLVE_GREEDY_ITER: HEU_Greedy_iter(inst); break;
    case SOLVE_GRASP: HEU_Grasp(inst); break;
    case SOLVE_GRASP_ITER: HEU_Grasp_iter(inst, inst->params.time_limit); break;
    case SOLVE_2OPT_GRASP: HEU_2opt_grasp(inst); break;
    case SOLVE_2OPT_GRASP_ITER: HEU_2opt_grasp_iter(inst); break;
    case SOLVE_2OPT_GREEDY: HEU_2opt_greedy(inst); break;
    case SOLVE_2OPT_GREEDY_ITER: HEU_2opt_greedy_iter(inst); break;
    case SOLVE_EXTR_MIL: HEU_extramileage(inst); break;
    case SOLVE_2OPT_EXTR_MIL: HEU_2opt_extramileage(inst); break;
    case SOLVE_VNS: HEU_VNS(inst); break;
    case SOLVE_GENETIC: HEU_Genetic(inst); break;
    case SOLVE_2OPT_GRASP_MULTI: {
        /* one process, -gpus G devices: threads + ncclCommInitAll; one process per GPU (RANK / WORLD_SIZE in the
         * environment, as torchrun / mpirun set them): ncclCommInitRank, device LOCAL_RANK; else this process alone */
        double cost = 0.0; int start = -1;
        const int starts = g_cli_starts > 0 ? g_cli_starts : 256;
        if (g_cli_gpus >= 1 && world_env == 1) tsp_host_multistart_gpus(inst, starts, g_cli_gpus, &cost, &start, NULL);
        else HEU_2opt_grasp_multistart(inst, starts, rank_env, world_env, &cost, &start);
        if (inst->params.verbose >= 1 && !inst->params.perf_prof && rank_env == 0) LOG_I("best start %d of %d, true cost %f", start, starts, cost);
        break;
    }
    case SOLVE_2OPT_POP_MULTI: {   /* BASELINE configs[4]; the same three ways */
        double cost = 0.0; int who = -1;
        const int individuals = g_cli_starts > 0 ? g_cli_starts : 128;
        if (g_cli_gpus >= 1 && world_env == 1) tsp_host_population_gpus(inst, individuals, g_cli_gpus, &cost, &who, NULL, NULL, NULL, NULL);
        else HEU_2opt_population_multistart(inst, individuals, rank_env, world_env, &cost, &who, NULL, NULL, NULL);
        if (inst->params.verbose >= 1 && !inst->params.perf_prof && rank_env == 0) LOG_I("best individual %d of %d, cost %f", who, individuals, cost);
        break;
    }
    case SOLVE_TABU_STEP:
    case SOLVE_TABU_LIN:
    case SOLVE_TABU_RAND:
        if (inst->params.time_limit <= 0) LOG_E("TABU_* runs until the time limit: pass -t <seconds> (tabusearch.c:231)");
        tsp_host_tabu(inst, inst->params.method.id - SOLVE_TABU_STEP, -1);
        break;
    default:
        LOG_E("method %s is outside this build's scope (2-opt hot path: GREEDY, GREEDY_ITER, EXTR_MILE, GRASP, GRASP_ITER, 2OPT_EXTR_MIL, "
              "2OPT_GRASP, 2OPT_GRASP_ITER, 2OPT_GRASP_MULTI, 2OPT_POP_MULTI, 2OPT_GREEDY, 2OPT_GREEDY_ITER, VNS, TABU_STEP, TABU_LIN, TABU_RAND, GENETIC)",
              inst->params.method.name ? inst->params.method.name : "?");
    }
    gettimeofday(&t1, 0);
    const double elapsed = get_elapsed_time(t0, t1);
    inst->solution.time_to_solve = elapsed;
    if (rank_env != 0) return 0;   /* one process per GPU: every rank holds the winner, rank 0 reports it */
    export_tour(inst);
    if (inst->params.perf_prof) printf("%0.2f", inst->solution.obj_best);          /* solver.c:291-292 */
    else printf("\n\n\nTIME TO SOLVE %0.6fs\n\n\n", elapsed);                       /* solver.c:295 */
    return 0;
}

/* ---- CLI edge ---------------------------------------------------------------------------------------------- */

/* The reference matches -method by a cascade of strncmp prefixes in which later matches override
 * earlier ones (src/utility.c:100-277); the table keeps that order and those lengths. */
static const struct { const char *prefix; int len; solver_type id; const char *name; } k_methods[] = {
    {"GREEDY", 6, SOLVE_GREEDY, "GREEDY HEURISTIC"},
    {"GREEDY_ITER", 11, SOLVE_GREEDY_ITER, "GREEDY ITERATIVE HEURISTIC"},
    {"EXTR_MIL", 6, SOLVE_EXTR_MIL, "EXTRA MILEAGE HEURISTIC"},
    {"GRASP", 5, SOLVE_GRASP, "GRASP HEURISTIC"},
    {"GRASP_ITER", 10, SOLVE_GRASP_ITER, "GRASP ITERATIVE HEURISTIC"},
    {"2OPT_GRASP", 9, SOLVE_2OPT_GRASP, "2-OPT HEURISTIC WITH GRASP INITIALIZATION"},
    {"2OPT_GRASP_ITER", 15, SOLVE_2OPT_GRASP_ITER, "2-OPT HEURISTIC WITH ITERATIVE GRASP INITIALIZATION"},
    {"2OPT_GRASP_MULTI", 16, SOLVE_2OPT_GRASP_MULTI, "GRASP MULTI-START, 2-OPT ON EVERY START, SHARDED OVER THE GPUS (extension)"},
    {"2OPT_POP_MULTI", 14, SOLVE_2OPT_POP_MULTI, "RANDOM POPULATION, 2-OPT ON EVERY INDIVIDUAL, SHARDED OVER THE GPUS (extension)"},
    {"2OPT_GREEDY", 11, SOLVE_2OPT_GREEDY, "2-OPT HEURISTIC WITH GREEDY INITIALIZATION"},
    {"2OPT_GREEDY_ITER", 16, SOLVE_2OPT_GREEDY_ITER, "2-OPT HEURISTIC WITH ITERATIVE GREEDY INITIALIZATION"},
    {"2OPT_EXTR_MIL", 13, SOLVE_2OPT_EXTR_MIL, "2-OPT HEURISTIC WITH EXTRA MILEAGE INITIALIZATION"},
    {"VNS", 3, SOLVE_VNS, "VNS META-HEURISTIC"},
    {"TABU_STEP", 9, SOLVE_TABU_STEP, "TABU SEARCH META-HEURISTIC WITH STEP POLICY"},
    {"TABU_LIN", 8, SOLVE_TABU_LIN, "TABU SEARCH META-HEURISTIC WITH LINEAR POLICY"},
    {"TABU_RAND", 9, SOLVE_TABU_RAND, "TABU SEARCH META-HEURISTIC WITH RANDOM POLICY"},
    {"GENETIC", 7, SOLVE_GENETIC, "GENETIC ALGORITHM META-HEURISTIC"},
};

static char *dup_string(const char *s) {
    char *d = malloc(strlen(s) + 1);
    strcpy(d, s);
    return d;
}

void parse_comand_line(int argc, const char *argv[], instance *inst) { /* src/utility.c:47-338 */
    if (argc <= 1) { printf("Type \"%s --help\" to see available comands\n", argv[0]); exit(1); }
    memset(inst, 0, sizeof *inst);
    inst->params.method.id = SOLVE_2OPT_GREEDY; /* the reference's default is a CPLEX method (:54); not buildable here */
    inst->params.method.edge_type = UDIR_EDGE;
    inst->params.method.name = (char *)"2-OPT HEURISTIC WITH GREEDY INITIALIZATION";
    inst->params.time_limit = -1;
    inst->params.num_threads = -1;
    inst->params.verbose = 1;
    inst->params.integer_cost = 1;
    inst->params.seed = (int)time(NULL);
    int need_help = 0, show_methods = 0;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        const int has_value = i < argc - 1;
        if (!strcmp(a, "-f")) { if (!has_value) { need_help = 1; continue; } inst->params.file_path = dup_string(argv[++i]); continue; }
        if (!strcmp(a, "-t")) { if (!has_value) { need_help = 1; continue; } inst->params.time_limit = atoi(argv[++i]); continue; }
        if (!strcmp(a, "-threads")) { if (!has_value) { need_help = 1; continue; } inst->params.num_threads = atoi(argv[++i]); continue; }
        if (!strcmp(a, "-verbose")) { if (!has_value) { need_help = 1; continue; } inst->params.verbose = atoi(argv[++i]); continue; }
        if (!strcmp(a, "-seed")) { if (!has_value) { need_help = 1; continue; } inst->params.seed = atoi(argv[++i]); continue; }
        if (!strcmp(a, "-starts")) { if (!has_value) { need_help = 1; continue; } g_cli_starts = atoi(argv[++i]); continue; }   /* extension */
        if (!strcmp(a, "-gpus")) { if (!has_value) { need_help = 1; continue; } g_cli_gpus = atoi(argv[++i]); continue; }       /* extension */
        if (!strcmp(a, "-method")) {
            if (!has_value) { need_help = 1; continue; }
            const char *m = argv[++i];
            int hit = 0;
            for (size_t k = 0; k < sizeof k_methods / sizeof k_methods[0]; k++)
                if (!strncmp(m, k_methods[k].prefix, (size_t)k_methods[k].len)) {
                    inst->params.method.id = k_methods[k].id;
                    inst->params.method.name = (char *)k_methods[k].name;
                    inst->params.method.use_cplex = 0;
                    hit = 1;
                }
            if (!hit) LOG_E("method %s needs CPLEX, which this build does not link (heuristics path only)", m);
            continue;
        }
        if (!strcmp(a, "--fcost")) { inst->params.integer_cost = 0; continue; }
        if (!strcmp(a, "--methods")) { show_methods = 1; continue; }
        if (!strcmp(a, "--perfprof")) { inst->params.perf_prof = 1; continue; }
        if (!strcmp(a, "--v") || !strcmp(a, "--version")) { printf("Version %s\n", "mi355x-r1"); exit(0); }
        need_help = 1;
    }
    if (show_methods) {
        for (size_t k = 0; k < sizeof k_methods / sizeof k_methods[0]; k++) printf("%-18s %s\n", k_methods[k].prefix, k_methods[k].name);
        exit(0);
    }
    if (need_help) {
        printf("-f <file's path>          To pass the problem's path\n");
        printf("-t <time>                 The time limit in seconds\n");
        printf("-threads <num threads>    The number of threads to use\n");
        printf("-verbose <level>          The verbosity level of the debugging printing\n");
        printf("-method <type>            The method used to solve the problem. Use \"--methods\" to see the list of available methods\n");
        printf("-seed <seed>              The seed for random generation\n");
        printf("-starts <S>               2OPT_GRASP_MULTI: number of GRASP starts (default 256); 2OPT_POP_MULTI: random individuals (default 128)\n");
        printf("-gpus <G>                 2OPT_GRASP_MULTI / 2OPT_POP_MULTI / GENETIC: GPUs of this process (or launch one process per GPU with RANK / WORLD_SIZE set)\n");
        printf("--fcost                   Whether you want float costs in the problem\n");
        printf("--perfprof                Print only the objective (machine mode)\n");
        printf("--v, --version            Software's current version\n");
        exit(0);
    }
}

void parse_instance(instance *inst) { /* src/utility.c:351-453: TSPLIB NODE_COORD_SECTION files */
    if (!inst->params.file_path) LOG_E("You didn't pass any file!");
    FILE *fp = fopen(inst->params.file_path, "r");
    if (!fp) LOG_E("Unable to open file!");
    inst->num_nodes = -1;
    inst->weight_type = (weight_type)-1;
    inst->num_columns = -1;
    char line[256];
    const char *sep = " :\n\t\r";
    int in_coords = 0;
    while (fgets(line, sizeof line, fp)) {
        char *key = strtok(line, sep);
        if (!key) continue;
        if (!strncmp(key, "EOF", 3)) break;
        if (!strncmp(key, "NAME", 4)) { char *v = strtok(NULL, sep); if (v) { free(inst->name); inst->name = dup_string(v); } in_coords = 0; continue; }
        if (!strncmp(key, "COMMENT", 7)) { in_coords = 0; continue; }
        if (!strncmp(key, "TYPE", 4)) {
            char *v = strtok(NULL, sep);
            if (!v || strncmp(v, "TSP", 3)) LOG_E(" format error:  only TYPE == TSP implemented so far!");
            in_coords = 0; continue;
        }
        if (!strncmp(key, "DIMENSION", 9)) {
            char *v = strtok(NULL, sep);
            inst->num_nodes = v ? atoi(v) : -1;
            if (inst->num_nodes > 0) inst->nodes = calloc((size_t)inst->num_nodes, sizeof(point));
            in_coords = 0; continue;
        }
        if (!strncmp(key, "EDGE_WEIGHT_TYPE", 16)) {
            char *v = strtok(NULL, sep);
            if (v) {
                if (!strncmp(v, "EUC_2D", 6)) inst->weight_type = EUC_2D;
                if (!strncmp(v, "MAX_2D", 6)) inst->weight_type = MAX_2D;
                if (!strncmp(v, "MAN_2D", 6)) inst->weight_type = MAN_2D;
                if (!strncmp(v, "CEIL_2D", 7)) inst->weight_type = CEIL_2D;
                if (!strncmp(v, "GEO", 3)) inst->weight_type = GEO;
                if (!strncmp(v, "ATT", 3)) inst->weight_type = ATT;
                if (!strncmp(v, "EXPLICIT", 8)) LOG_E("Wrong edge weight type, this program resolve only 2D TSP case with coordinate type.");
            }
            in_coords = 0; continue;
        }
        if (!strncmp(key, "NODE_COORD_SECTION", 18)) { in_coords = 1; continue; }
        if (!strncmp(key, "EDGE_WEIGHT_SECTION", 19)) { in_coords = 0; continue; }
        if (in_coords) {
            const int id = atoi(key) - 1;
            if (!inst->nodes || id < 0 || id >= inst->num_nodes) LOG_E(" ... unknown node in NODE_COORD_SECTION section");
            char *a = strtok(NULL, sep), *b = strtok(NULL, sep);
            if (!a || !b) LOG_E(" ... malformed coordinate line");
            inst->nodes[id].x = atof(a);
            inst->nodes[id].y = atof(b);
        }
    }
    fclose(fp);
    if (inst->num_nodes <= 0 || !inst->nodes) LOG_E("no DIMENSION in %s", inst->params.file_path);
    if (!inst->name) inst->name = dup_string("unnamed");
}

void export_tour(instance *inst) { /* src/utility.c:523-555: TSPLIB TOUR + OBJECTIVE/TIME lines */
    if (inst->params.perf_prof) return;
    mkdir("../tour", 0777);
    char path[1024];
    snprintf(path, sizeof path, "../tour/%s.tour", inst->name);
    FILE *f = fopen(path, "w");
    if (!f) { printf("Unable to save the tour file\n"); return; }
    fprintf(f, "NAME : %s.tour\nTYPE : TOUR\nDIMENSION : %d\nOBJECTIVE : %f\nTIME : %f\nTOUR_SECTION\n", inst->name,
            inst->num_nodes, inst->solution.obj_best, inst->solution.time_to_solve);
    int v = 0;
    for (int k = 0; k < inst->num_nodes; k++) { fprintf(f, "%d\n", v + 1); v = inst->solution.edges[v].j; }
    fprintf(f, "-1\nEOF");
    fclose(f);
}
