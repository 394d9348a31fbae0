/*
 * tsp_main.c -- `tsp` command line for the heuristics path, same flags and output protocol as the
 * reference's src/main.c:14-31 + src/utility.c:47-338 (-f -t -verbose -method -seed --fcost
 * --perfprof --methods --help --v).  `--perfprof` prints the bare objective ("%0.2f", no newline),
 * which is what the reference's experiment drivers parse (other_codes/constructive_comparison.py:34-43).
 */
#include "tsp_host.h"

int main(int argc, const char *argv[]) {
    instance inst;
    parse_comand_line(argc, argv, &inst);
    parse_instance(&inst);
    if (inst.params.verbose >= 1 && !inst.params.perf_prof) {
        printf("\n======== INSTANCE ========\nname: %s\nn nodes: %d\nmethod: %s\n\n", inst.name, inst.num_nodes,
               inst.params.method.name);
    }
    TSP_heuc(&inst);
    inst.params.method.name = NULL; /* points into a constant table */
    free_instance(&inst);
    return 0;
}
