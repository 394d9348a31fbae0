/*
 * oracle/tsp_oracle.h -- CPU restatement of the reference's 2-opt hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (tsp_optimization_amd/,
 * include/, the C-ABI library, the host mirror, the CLI) may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the timed CPU comparator.
 *
 * Parity status: PINNED against the reference's own published known answers
 * (results/constructive_heuristics_new.csv and
 * results/constructive_heuristics_2opt_new.csv, seed 123; committed as
 * tests/golden/reference_results.json) and against the counters SURVEY.md
 * Appendix B recorded from the unmodified reference.  The reference itself is
 * NOT buildable in this image (every source includes <cplex.h>, which the image
 * lacks, and stand-in headers are not allowed), so there is no oracle/_ref.
 *
 * The two driver restatements (orc_vns, orc_tabu) are pinned only through their components
 * (orc_two_opt_first / orc_two_opt_best / orc_perm_cost): the reference's loops run until a wall-clock
 * limit, so neither its result tables nor any fixture can pin the loops themselves.
 *
 * Every function cites the reference file:line whose behaviour it restates.
 * Data is passed as flat arrays (xy = n x {x,y} doubles, succ = successor list)
 * rather than the reference's `instance` struct.
 */
#ifndef TSP_ORACLE_H
#define TSP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* weight types, numbered like the reference's enum (include/utility.h:45-52) */
enum { ORC_EUC_2D = 0, ORC_MAX_2D = 1, ORC_MAN_2D = 2, ORC_CEIL_2D = 3, ORC_GEO = 4, ORC_ATT = 5 };

/* status codes (include/heuristics.h:6-7) */
enum { ORC_OK = 0, ORC_WRONG_STARTING_NODE = 1, ORC_TIME_LIMIT_EXCEEDED = 2 };

typedef struct {
    long long sweeps;      /* completed passes over the (i<j) pair space            */
    long long evals;       /* executions of the delta expression (non-skipped pairs) */
    long long moves;       /* applied 2-opt moves (= reverse_path calls)            */
    long long reversed;    /* nodes whose successor was rewritten by reversals      */
    double seconds;        /* wall time of the call                                  */
} orc_stats;

typedef struct {           /* one applied move, for trajectory parity */
    int i, j;
    double delta;
} orc_move;

/* src/distutil.c:73-92 (+ :4-71) */
double orc_dist(const double *xy, int i, int j, int wtype, int integer_cost);

/* Fill an n x n row-major matrix with orc_dist(i,j); diagonal = 0. */
void orc_dist_matrix(const double *xy, int n, int wtype, int integer_cost, double *out);

/* src/heuristics.c:18-78 */
int orc_greedy(const double *xy, int n, int wtype, int integer_cost, int start, int *succ, double *obj);

/* src/heuristics.c:82-156; draws from libc random() exactly like URAND() (include/utility.h:36).
 * If urand != NULL the n draws are taken from urand[0..n-1] instead (same values, no libc state). */
int orc_grasp(const double *xy, int n, int wtype, int integer_cost, int start, const double *urand,
              int *succ, double *obj);

/* src/heuristics.c:168-205 (all n starting nodes, keep the best) */
/* HEU_Grasp_iter (src/heuristics.c:510-544) for exactly `starts` starts of the libc stream */
int orc_grasp_iter_prefix(const double *xy, int n, int wtype, int integer_cost, long long starts, int *succ, double *obj,
                          long long *best_start);
int orc_greedy_iter(const double *xy, int n, int wtype, int integer_cost, int *succ, double *obj);

/* src/heuristics.c:208-314 (farthest pair + cheapest insertion) */
int orc_extramileage(const double *xy, int n, int wtype, int integer_cost, int *succ, double *obj);

/* src/heuristics.c:438-502 (first improvement, moves applied immediately).
 * clock_per_pair != 0 also calls gettimeofday once per pair as the reference does (:456). */
int orc_two_opt_first(const double *xy, int n, int wtype, int integer_cost, int *succ, double *obj,
                      double time_limit, int clock_per_pair, orc_stats *st, orc_move *trace,
                      long long trace_cap);

/* Harness helper: the first max_moves moves of orc_two_opt_first (stops mid-sweep right after the last one). */
int orc_two_opt_first_moves(const double *xy, int n, int wtype, int integer_cost, int *succ, double *obj,
                            long long max_moves, orc_stats *st);

/* src/tabusearch.c:107-178 (best improvement; tabu == NULL gives plain best-improvement 2-opt).
 * max_sweeps < 0 = until local optimum (max_sweeps is a harness knob for bounded timing samples). */
int orc_two_opt_best(const double *xy, int n, int wtype, int integer_cost, int *succ, double *obj,
                     int *tabu, int iter, int tenure, int *stored_prev, double time_limit,
                     long long max_sweeps, orc_stats *st, orc_move *trace, long long trace_cap);

/* src/utility.c:708-722 */
void orc_reverse_path(int n, int *succ, int start_node, int end_node, int *prev);

/* src/genetic.c:51-60 : cost of a permutation */
double orc_perm_cost(const double *xy, int n, int wtype, int integer_cost, const int *perm);
/* src/tabusearch.c:168-172 : cost of a successor list, summed in node order */
double orc_succ_cost(const double *xy, int n, int wtype, int integer_cost, const int *succ);
/* src/genetic.c:33-42 and :436-441 */
void orc_perm_to_succ(int n, const int *perm, int *succ);
void orc_succ_to_perm(int n, const int *succ, int *perm);
/* src/genetic.c:349-364, libc random() via rand_choice (src/utility.c:752) */
void orc_random_perm(int n, int *perm);

/* src/utility.c:17-30 */
int orc_udir_pos(int i, int j, int n);

/* src/vns.c:11-100 : random 3-edge reconnection + full cost recompute; draws from libc random().
 * The reference reads tour[idx3+1] one past its array when idx3 == n-1 (:57); this restatement
 * (like the product) wraps that index to the tour's first node. */
void orc_vns_kick(const double *xy, int n, int wtype, int integer_cost, int *succ, double *obj);

/* src/vns.c:103-166 with the initial solution passed in (the reference builds it with
 * HEU_2opt_greedy_iter, :116) and a harness cap on the number of kick+2opt rounds instead of the
 * wall clock.  improved (may be NULL) counts the rounds that lowered the incumbent. */
int orc_vns(const double *xy, int n, int wtype, int integer_cost, int *succ, double *obj, long long rounds,
            long long *improved);

/* src/tabusearch.c:188-320 with the initial solution passed in (:200) and a cap on the number of
 * iterations instead of the wall clock.  policy: 0 = step (:33), 1 = linear (:47), 2 = random (:69). */
int orc_tabu(const double *xy, int n, int wtype, int integer_cost, int policy, int *succ, double *obj,
             long long iterations, long long *total_moves);

/* src/genetic.c:448-565 (population 1000, rank-roulette selection, OX-like crossover, reversal mutation,
 * rank-roulette survivors incl. the reference's chromosome aliasing) with a generation cap; the incumbent
 * tour / cost are written whenever a generation's best improves on them (:518-526). */
int orc_genetic(const double *xy, int n, int wtype, int integer_cost, long long generations, int *succ, double *obj);
/* the same with the probability of mutation method 3 (2-opt, genetic.c:426-443) as an argument; 0.00 in the reference */
int orc_genetic_ex(const double *xy, int n, int wtype, int integer_cost, long long generations, double two_opt_prob,
                   int *succ, double *obj);

/* libc RNG access so that tests can reproduce the reference's stream (src/solver.c:264-266) */
void orc_srandom(unsigned seed);
double orc_urand(void);

/* src/utility.c:351-453 : TSPLIB NODE_COORD reader.  Returns n (>0) or a negative error.
 * xy may be NULL to query n only; cap = capacity of xy in nodes. */
int orc_parse_tsplib(const char *path, double *xy, int cap, int *wtype);

/* 64-bit FNV-1a over the n ints of a successor list (fixture compaction helper) */
unsigned long long orc_fnv1a(const int *v, int n);

#ifdef __cplusplus
}
#endif
#endif
