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

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

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
    SOLVE_2OPT_GRASP_MULTI   /* extension of this build (after the reference's last value): BASELINE configs[3] */
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
/* Multi-start of BASELINE config 4: `starts` GRASP tours drawn exactly like HEU_Grasp_iter draws
 * them (heuristics.c:519 then :127), each refined by alg_2opt on the device, best TRUE cost kept in
 * inst->solution (ties -> lowest start).  rank/world shard the starts (k % world == rank). */
int HEU_2opt_grasp_multistart(instance *inst, int starts, int rank, int world, double *best_true_cost,
                              int *best_start);
/* One rank's share of it and nothing else: the starts k % world == rank, the shard's best in inst->solution (no communication). */
int tsp_host_multistart_shard(instance *inst, int starts, int rank, int world, double *best_true_cost, int *best_start);
/* HEU_2opt_grasp_multistart with world > 1 is COLLECTIVE (one process per GPU, device LOCAL_RANK; every rank must call it):
 * after its shard the ranks agree on the winner with one RCCL all-reduce(min) of
 * (true cost << 24 | start) and one broadcast of its tour (tsp_dev_multistart_* of include/tsp_hip.h); the RCCL id travels
 * from rank 0 through the file TSP_RCCL_ID_FILE.  The same job in ONE process on devices 0 .. gpus-1 (one thread per GPU,
 * ncclCommInitAll, grouped collectives); shard_seconds[gpus] (may be NULL) receives every GPU's construct + 2-opt time. */
int tsp_host_multistart_gpus(instance *inst, int starts, int gpus, double *best_true_cost, int *best_start,
                             double *shard_seconds);
/* The two drivers with a cap on the number of rounds / iterations in addition to the time limit
 * (max < 0 = time limit only, which is what HEU_VNS / HEU_Tabu_* pass).  policy: 0 step, 1 linear,
 * 2 random.  The reference's loops are bounded by the wall clock alone, which no test can reproduce. */
int tsp_host_vns(instance *inst, long long max_rounds);
int tsp_host_tabu(instance *inst, int policy, long long max_iterations);
int tsp_host_genetic(instance *inst, long long max_generations);
/* the same with the probability of mutation method 3 (alg_2opt on the offspring, genetic.c:426-443) as an argument:
 * TWO_OPT_MUTATION_PROB is 0.00 in the reference (genetic.c:18); tests raise it so that the branch is executed */
int tsp_host_genetic_ex(instance *inst, long long max_generations, double two_opt_prob);
/* Counters of the last alg_2opt / alg_2opt_tabu call of this thread. */
void tsp_host_last_stats(long long *sweeps, long long *evals, long long *moves, double *device_ms);
/* Releases the cached device context / instances (optional; also done at exit). */
void tsp_host_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
