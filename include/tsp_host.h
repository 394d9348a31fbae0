/*
 * tsp_host.h -- C host mirror of the reference's heuristics entry points (same names, same argument
 * meaning, same status codes), implemented on top of the C ABI in include/tsp_hip.h.
 *
 * A caller written against the reference's include/utility.h + include/heuristics.h +
 * include/solver.h + include/tabusearch.h recompiles against this header unchanged for the
 * functions listed here.  The data model below restates the reference's structs field for field
 * (include/utility.h:113-160) because their layout IS the interface: `inst->nodes` is handed to the
 * device library as `const double *xy`, `&inst->solution.edges[0].j` as a stride-2 successor list.
 * The only difference: <cplex.h> is not included (the heuristics path never needed it).
 *
 * There is no CPU implementation behind these functions: each of them runs on the MI355X through
 * libtsp_hip.so and terminates the process with an [ERROR] line (the reference's LOG_E convention,
 * include/utility.h:33) if no device is available.
 */
#ifndef TSP_HOST_H
#define TSP_HOST_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "tsp_hip.h"   /* tsp_two_opt_stats, TSP_DEV_E_* (extensions at the end of this header) */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- logging / allocation idioms of the reference (include/utility.h:10-36) ---------------- */
#define LOG_I(...) do { fprintf(stdout, "[INFO]  "); fprintf(stdout, __VA_ARGS__); fprintf(stdout, "\n"); } while (0)
#define LOG_E(...) do { fprintf(stderr, "[ERROR] "); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); \
                        fflush(NULL); exit(1); } while (0)
#define URAND() (((double)random()) / RAND_MAX)   /* include/utility.h:36 */
#define DEFAULT_TIME_LIM 900                      /* include/utility.h:41 */

/* status codes: include/heuristics.h:6-7 */
#define WRONG_STARTING_NODE 1
#define TIME_LIMIT_EXCEEDED 2

/* ---- data model: include/utility.h:45-160 --------------------------------------------------- */
typedef enum { EUC_2D, MAX_2D, MAN_2D, CEIL_2D, GEO, ATT } weight_type;                 /* :45-52 */

typedef enum {                                                                           /* :56-84 */
    SOLVE_MTZ, SOLVE_MTZL, SOLVE_MTZI, SOLVE_MTZLI, SOLVE_MTZ_IND, SOLVE_GG, SOLVE_LOOP, SOLVE_CALLBACK,
    SOLVE_UCUT, SOLVE_HARD_FIXING, SOLVE_HARD_FIXING2, SOLVE_SOFT_FIXING,
    SOLVE_GREEDY, SOLVE_GREEDY_ITER, SOLVE_EXTR_MIL, SOLVE_GRASP, SOLVE_GRASP_ITER,
    SOLVE_2OPT_GRASP, SOLVE_2OPT_GRASP_ITER, SOLVE_2OPT_GREEDY, SOLVE_2OPT_GREEDY_ITER, SOLVE_2OPT_EXTR_MIL,
    SOLVE_VNS, SOLVE_TABU_STEP, SOLVE_TABU_LIN, SOLVE_TABU_RAND, SOLVE_GENETIC,
    SOLVE_2OPT_GRASP_MULTI,  /* extension of this build (after the reference's last value): BASELINE configs[3] */
    SOLVE_2OPT_POP_MULTI     /* extension: BASELINE configs[4], random individuals + alg_2opt each, sharded over the GPUs */
} solver_type;

typedef enum { UDIR_EDGE, DIR_EDGE } edge_type;                                          /* :99-102 */

typedef struct { solver_type id; edge_type edge_type; char *name; int use_cplex; } sol_method;   /* :105-110 */

typedef struct {                                                                         /* :113-123 */
    char *file_path;
    int num_threads;
    int time_limit;
    sol_method method;
    int verbose;
    int integer_cost;
    int seed;
    int perf_prof;
    int callback_2opt;
} instance_params;

typedef struct { double x; double y; } point;                                            /* :126-129 */
typedef struct { int i; int j; } edge;                                                   /* :134-137 */

typedef struct {                                                                         /* :139-144 */
    double obj_best;
    edge *edges;
    double time_to_solve;
    double *xbest;
} solution;

typedef struct {                                                                         /* :147-160 */
    instance_params params;
    char *name;
    char *comment;
    point *nodes;
    int num_nodes;
    weight_type weight_type;
    long num_columns;
    int *ind;
    unsigned int *thread_seeds;
    solution solution;
} instance;

/* ---- src/distutil.c ---------------------------------------------------------------------- */
double calc_dist(int i, int j, instance *inst);                                          /* distutil.h:137 */

/* ---- src/utility.c helpers on the path ------------------------------------------------------ */
int x_udir_pos(int i, int j, int num_nodes);                                             /* utility.c:17  */
double get_elapsed_time(struct timeval start, struct timeval end);                       /* utility.c:701 */
void reverse_path(instance *inst, int start_node, int end_node, int *prev);              /* utility.c:708 */
void copy_instance(instance *dst, instance *src);                                        /* utility.c:724 */
int rand_choice(int from, int to);                                                       /* utility.c:752 */
void free_instance(instance *inst);                                                      /* utility.c:340 */

/* ---- src/heuristics.c ------------------------------------------------------------------------ */
int greedy(instance *inst, int starting_node);                                           /* :18  */
int grasp(instance *inst, int starting_node);                                            /* :82  */
int HEU_greedy(instance *inst);                                                          /* :160 */
int HEU_Greedy_iter(instance *inst);                                                     /* :168 */
int HEU_extramileage(instance *inst);                                                    /* :208 */
int alg_2opt(instance *inst);                                                            /* :438 */
int HEU_Grasp(instance *inst);                                                           /* :505 */
int HEU_Grasp_iter(instance *inst, int time_lim);                                        /* :510 */
int HEU_2opt_grasp(instance *inst);                                                      /* :547 */
int HEU_2opt_grasp_iter(instance *inst);                                                 /* :559 */
int HEU_2opt_greedy(instance *inst);                                                     /* :572 */
int HEU_2opt_greedy_iter(instance *inst);                                                /* :584 */
int HEU_2opt_extramileage(instance *inst);                                               /* :596 */

/* ---- src/tabusearch.c (externally linked there, not in its header) --------------------------- */
int alg_2opt_tabu(instance *inst, int *skip_edge, int *stored_prev, const int iter, const int tenure); /* :107 */

/* ---- src/tabusearch.c / src/vns.c drivers (wall-clock bounded like the reference's) -------------- */
int HEU_Tabu_step(instance *inst);                                                       /* tabusearch.c:323 */
int HEU_Tabu_lin(instance *inst);                                                        /* :328 */
int HEU_Tabu_rand(instance *inst);                                                       /* :333 */
int kick(instance *inst);                                                                /* vns.c:11  */
int HEU_VNS(instance *inst);                                                             /* vns.c:103 */
int HEU_Genetic(instance *inst);                                                         /* genetic.c:448 */

/* ---- src/genetic.c : fitness of `count` chromosomes at once (the reference scores one at a time, :51) */
int fitness_batch(instance *inst, const int *chromosomes, int count, double *fitness_out);

/* ---- src/solver.c ---------------------------------------------------------------------------- */
int TSP_heuc(instance *inst);                                                            /* :262 */

/* ---- CLI edge (src/utility.c:47, :351; only what the heuristics path needs) ------------------- */
void parse_comand_line(int argc, const char *argv[], instance *inst);
void parse_instance(instance *inst);
void export_tour(instance *inst);

/* ---- extensions of this build (not in the reference) ------------------------------------------ */
/* Multi-start of BASELINE configs[3]: `starts` GRASP tours drawn exactly like HEU_Grasp_iter draws
 * them (heuristics.c:519 then :127), each refined by alg_2opt on the device, best TRUE cost kept in
 * inst->solution (ties -> lowest start).  rank/world shard the starts (k % world == rank).
 * world > 1 is COLLECTIVE (one process per GPU, device LOCAL_RANK; every rank must call it): after its shard the ranks agree
 * on the winner with one RCCL all-reduce(min) of (true cost << 24 | start) -- with --fcost: min of the double cost, then min
 * of the start among its holders -- and one broadcast of its tour (tsp_dev_multistart_* of include/tsp_hip.h); the RCCL id
 * travels from rank 0 through the file TSP_RCCL_ID_FILE.  A rank whose shard fails does not exit before the collective: it
 * contributes a value that wins the minimum, every rank sees it, and every rank ends with an [ERROR] line and status 1. */
int HEU_2opt_grasp_multistart(instance *inst, int starts, int rank, int world, double *best_true_cost,
                              int *best_start);
/* One rank's share of it and nothing else: the starts k % world == rank, the shard's best in inst->solution (no communication). */
int tsp_host_multistart_shard(instance *inst, int starts, int rank, int world, double *best_true_cost, int *best_start);
/* The same job in ONE process on devices 0 .. gpus-1 (one thread per GPU, ncclCommInitAll, grouped collectives);
 * shard_seconds[gpus] (may be NULL) receives every GPU's construct + 2-opt time. */
int tsp_host_multistart_gpus(instance *inst, int starts, int gpus, double *best_true_cost, int *best_start,
                             double *shard_seconds);

/* The population job of BASELINE configs[4], same three forms: `individuals` random individuals generated exactly as
 * random_generation does (genetic.c:349-364: identity, then n swaps of two rand_choice(0, n) positions, libc stream; every
 * rank walks the whole stream), individual k on rank k % world, each refined by alg_2opt starting from its fitness
 * (the mutation-3 path, genetic.c:426-443, without its 2 s limit unless -t is given); the best refined individual (first
 * strictly better in id order, heuristics.c:534) ends in inst->solution on every rank.
 * Optional outputs, indexed by the GLOBAL individual id and filled for the individuals THIS process refined (all of them
 * in the one-process forms): costs_out[individuals], succ_out[individuals x n] (successor lists), stats_out[individuals]. */
int HEU_2opt_population_multistart(instance *inst, int individuals, int rank, int world, double *best_cost, int *best_individual,
                                   double *costs_out, int *succ_out, tsp_two_opt_stats *stats_out);
int tsp_host_population_shard(instance *inst, int individuals, int rank, int world, double *best_cost, int *best_individual,
                              double *costs_out, int *succ_out, tsp_two_opt_stats *stats_out);
int tsp_host_population_gpus(instance *inst, int individuals, int gpus, double *best_cost, int *best_individual,
                             double *shard_seconds, double *costs_out, int *succ_out, tsp_two_opt_stats *stats_out);

/* The collective epilogue on its own (what the two *_multistart entries run after their shard), COLLECTIVE over `world`
 * ranks.  In: shard_rc (< 0: this rank's shard failed), *best / *best_id (best_id < 0: empty shard), the shard's best tour in
 * inst->solution.edges.  Out, on every rank: the global winner in *best / *best_id / inst->solution and 0 -- or the same
 * negative code everywhere: TSP_HOST_E_PEER when any rank reported a failure (text of this rank's own failure, if any, in
 * tsp_host_multistart_last_error()), a TSP_DEV_E_* when the transport failed on this rank. */
#define TSP_HOST_E_PEER (-7)
int tsp_host_multistart_epilogue(instance *inst, int rank, int world, int shard_rc, double *best, int *best_id);
const char *tsp_host_multistart_last_error(void);
/* The three collectives of the epilogue.  Default (NULL): RCCL through libtsp_hip.so.  A host program that already has a
 * communicator of its own (MPI, a test harness over gloo) can carry them instead; return 0 or a negative code. */
typedef struct tsp_host_collectives {
    int (*allreduce_min_i64)(void *self, int64_t local, int64_t *out);
    int (*allreduce_min_f64)(void *self, double local, double *out);
    int (*bcast_i32)(void *self, int root, int *buf, int stride, int n);   /* n ints `stride` apart, in place */
    void *self;
} tsp_host_collectives;
void tsp_host_set_collectives(const tsp_host_collectives *c);
/* The id file of the one-process-per-GPU form (TSP_RCCL_ID_FILE, default /tmp/tsp_rccl_id.<uid>.<launcher pid>.<MASTER_PORT>)
 * holds {magic, the 128-byte RCCL id, rank 0's pid, rank 0's start time}: a reader accepts it only while that very process is
 * alive, so the file a crashed run left behind is stale whatever its age.  State of the file at `path` (NULL: the default
 * path): 0 nothing readable (symlinks are not followed), 1 live, 2 stale. */
int tsp_host_rccl_id_file_state(const char *path);
/* Starts built by the last HEU_Grasp_iter / HEU_2opt_grasp_iter call of this thread (whole batches of 256: the clock is
 * read between batches, heuristics.c:519-525 reads it per start). */
long long tsp_host_last_grasp_iter_starts(void);
/* The two drivers with a cap on the number of rounds / iterations in addition to the time limit
 * (max < 0 = time limit only, which is what HEU_VNS / HEU_Tabu_* pass).  policy: 0 step, 1 linear,
 * 2 random.  The reference's loops are bounded by the wall clock alone, which no test can reproduce. */
int tsp_host_vns(instance *inst, long long max_rounds);
int tsp_host_tabu(instance *inst, int policy, long long max_iterations);
int tsp_host_genetic(instance *inst, long long max_generations);
/* the same with the probability of mutation method 3 (alg_2opt on the offspring, genetic.c:426-443) as an argument:
 * TWO_OPT_MUTATION_PROB is 0.00 in the reference (genetic.c:18); tests raise it so that the branch is executed */
int tsp_host_genetic_ex(instance *inst, long long max_generations, double two_opt_prob);
/* ... and with the offspring that drew mutation 3 spread over devices 0 .. gpus-1 (a contiguous block of them per GPU, one thread per GPU
 * and generation, context and instance per GPU kept for the whole run); gpus <= 0: this process's device as above.  The CLI's -gpus G. */
int tsp_host_genetic_gpus(instance *inst, long long max_generations, double two_opt_prob, int gpus);
/* Values of the libc stream that this library has drawn ahead of their turn and not consumed yet (tsp_host_tabu queues chains
 * of iterations and needs every iteration's first kick draws up front; what a chain leaves unconsumed is served first by every
 * later draw of this library: URAND() of this build, rand_choice()).  0 whenever a capped run (max_iterations) has returned. */
int tsp_host_random_lookahead(void);
/* Seconds the last tsp_host_tabu / tsp_host_vns call of this thread spent in its iteration loop (the initial
 * HEU_2opt_greedy_iter excluded): iterations / this = the driver's rate. */
double tsp_host_last_driver_loop_seconds(void);
/* Counters of the last alg_2opt / alg_2opt_tabu call of this thread. */
void tsp_host_last_stats(long long *sweeps, long long *evals, long long *moves, double *device_ms);
/* Releases the cached device context / instances (optional; also done at exit). */
void tsp_host_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
