// two_opt_lds.hip -- LDS engine: one workgroup per tour, the whole 2-opt descent in one launch.
#include "tsp_internal.hpp"

#pragma clang fp contract(off)

using namespace tsp;

bool tsp_lds_fits(const tsp_dev_inst *inst) { (void)inst; return false; }

int tsp_lds_two_opt(tsp_dev_inst *inst, int mode, int B, int *succ, int succ_stride, int64_t tour_stride,
                    double *obj, double time_limit_s, tsp_two_opt_stats *stats) {
    (void)inst; (void)mode; (void)B; (void)succ; (void)succ_stride; (void)tour_stride; (void)obj;
    (void)time_limit_s; (void)stats;
    return TSP_DEV_E_ARG;
}
