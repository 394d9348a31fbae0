// tsp_dist.hpp -- device-side distance functions (gfx950), bit-compatible with the reference's
// calc_dist (src/distutil.c:73-92) for every metric whose arithmetic is IEEE basic operations
// + sqrt (EUC_2D, ATT, CEIL_2D, MAN_2D, MAX_2D).  GEO needs cos/acos, where ocml and glibc
// differ in the last ulp: tolerance tier.
//
// Rules that make the results equal to x86-64 gcc's (which never fuses a*b+c without -mfma):
//   - this translation unit is compiled with -ffp-contract=off and the pragma below repeats it;
//   - sqrt() on double is the correctly rounded ocml/LLVM expansion (no -ffast-math);
//   - nint(x) = (double)(long)(x + 0.5) == trunc(x + 0.5) for the non-negative values that occur.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#pragma clang fp contract(off)

namespace tsp {

enum : int { WT_EUC_2D = 0, WT_MAX_2D = 1, WT_MAN_2D = 2, WT_CEIL_2D = 3, WT_GEO = 4, WT_ATT = 5,
             // Same metrics on instances whose coordinates are all integers of bounded size: the
             // squared distance s is then an exact integer and the rounded root can be fixed up
             // exactly from a low-precision v_sqrt_f64 (see int_root below).  Chosen by the host
             // (tsp_dev_inst_create) only when the bound holds; results are identical by construction.
             WT_EUC_2D_ICOORD = 6, WT_CEIL_2D_ICOORD = 7, WT_ATT_ICOORD = 8 };

// Largest coordinate span for which the *_ICOORD variants are used: distances < 2^21 keep
// k*k, 10*k*k and s exact in fp64 and the raw v_sqrt_f64 error (<= 2^-23 relative per the ISA
// manual) below 0.25, so rint() of it is within one of the true root.
#define TSP_ICOORD_MAX_DIST 2097151.0

// include/distutil.h:6-7
#define TSP_GEO_PI 3.14159265358979323846264
#define TSP_GEO_RADIUS 6378.388

// src/distutil.c:4-6
__device__ __forceinline__ double nint_pos(double v) { return trunc(v + 0.5); }

// src/distutil.c:51-58 : degrees.minutes -> radians; (long) truncates toward zero like trunc()
__device__ __host__ __forceinline__ double geo_radians(double v) {
    double deg = (double)(long long)v;
    double frac = v - deg;
    return TSP_GEO_PI * (deg + 5.0 * frac / 3.0) / 180.0;
}

// Exact integer roots of an exact integer s >= 0 (s < 2^43) from the hardware's approximate sqrt:
//   MODE 0: nint(sqrt(s))      = the k with k*k - k <  s <= k*k + k          (EUC_2D, integer costs)
//   MODE 1: ceil(sqrt(s))      = the smallest k with k*k >= s                (CEIL_2D)
//   MODE 2: ATT's rounded-up sqrt(s/10) = the smallest k with 10*k*k >= s    (ATT, integer costs;
//           distutil.c:26-27 turns nint into a ceiling, and 10*k*k == s is exact in fp64)
// g = v_sqrt_f64 of the (scaled) argument is within 0.25 of the true root r because r < 2^21 and the
// instruction's relative error is <= 2^-23 (ISA manual; tests/test_gpu_parity.py measures it at
// < 2^-25 on gfx950).  Then k0 = floor(g + c) is k* - 1 or k* (c = 0.25 for the rounded root,
// 0.75 for the ceilings), and one exact residual test decides which.  No branch, 6 instructions.
template <int MODE>
__device__ __forceinline__ double int_root(double s) {
    const double g = __builtin_amdgcn_sqrt(MODE == 2 ? s * 0.1 : s);
    double k = floor(g + (MODE == 0 ? 0.25 : 0.75));
    if constexpr (MODE == 0) {
        const double e = fma(-k, k, s);            // exact: s - k^2
        k += (e > k) ? 1.0 : 0.0;                  // s > k^2 + k: the root rounds to k + 1
    } else if constexpr (MODE == 1) {
        const double e = fma(-k, k, s);
        k += (e > 0.0) ? 1.0 : 0.0;                // k^2 < s
    } else {
        const double e = fma(-10.0 * k, k, s);     // exact: s - 10 k^2
        k += (e > 0.0) ? 1.0 : 0.0;
    }
    return k;
}

// Distance between two nodes given their coordinate pairs.  For WT_GEO the pairs are
// (latitude, longitude) in radians as produced by geo_radians() at upload time; for every other
// type they are the raw (x, y).  Unknown weight types fall back to EUC_2D (src/distutil.c:90-91),
// which the host maps before choosing the template instance.
template <int WT, bool INT>
__device__ __forceinline__ double dist_xy(double ax, double ay, double bx, double by) {
    if constexpr (WT == WT_EUC_2D_ICOORD || WT == WT_CEIL_2D_ICOORD || WT == WT_ATT_ICOORD) {
        static_assert(INT || WT == WT_CEIL_2D_ICOORD, "integer-coordinate variants exist for integer costs only");
        const double dx = ax - bx, dy = ay - by;        // exact: integer operands
        const double s = dx * dx + dy * dy;             // exact: < 2^43
        return int_root<WT == WT_EUC_2D_ICOORD ? 0 : (WT == WT_CEIL_2D_ICOORD ? 1 : 2)>(s);
    } else if constexpr (WT == WT_ATT) {                       // src/distutil.c:20-31
        const double dx = ax - bx, dy = ay - by;
        const double r = sqrt((dx * dx + dy * dy) / 10.0);
        if constexpr (!INT) return r;
        const double t = nint_pos(r);
        return t < r ? t + 1.0 : t;
    } else if constexpr (WT == WT_MAN_2D) {             // src/distutil.c:33-37, dy = |by - by|
        const double dx = fabs(ax - bx), dy = fabs(by - by);
        return INT ? nint_pos(dx + dy) : dx + dy;
    } else if constexpr (WT == WT_MAX_2D) {             // src/distutil.c:39-45, dy = |by - by|
        double dx = fabs(ax - bx), dy = fabs(by - by);
        if constexpr (INT) { dx = nint_pos(dx); dy = nint_pos(dy); }
        return dx > dy ? dx : dy;
    } else if constexpr (WT == WT_CEIL_2D) {            // src/distutil.c:47-49
        const double dx = ax - bx, dy = ay - by;
        return ceil(sqrt(dx * dx + dy * dy));
    } else if constexpr (WT == WT_GEO) {                // src/distutil.c:60-71
        const double q1 = cos(ay - by);
        const double q2 = cos(ax - bx);
        const double q3 = cos(ax + bx);
        const double d = TSP_GEO_RADIUS * acos(0.5 * ((1.0 + q1) * q2 - (1.0 - q1) * q3)) + 1.0;
        return INT ? nint_pos(d) : d;
    } else {                                            // src/distutil.c:13-18
        const double dx = ax - bx, dy = ay - by;
        const double d = sqrt(dx * dx + dy * dy);
        return INT ? nint_pos(d) : d;
    }
}

// ---- cheap lower bound used to skip exact evaluations ------------------------------------------------
// For the sqrt-based metrics the exact distance costs a correctly rounded root (or an exact integer
// fix-up); the hardware's raw v_sqrt_f64 is ~4x cheaper and within r * 2^-23 of the true root r.  A
// 2-opt scan only needs the exact delta of pairs that could beat the current bound, so it first forms
// delta~ from raw roots and skips the pair when  delta~ - margin >= bound.  `margin` (computed per
// instance on the host, tsp_dev_inst_create) covers the two raw-root errors plus the rounding that the
// exact metric applies (nint: 0.5 per distance, ceil/ATT: < 1 per distance).  Every decision is still
// taken on exact values; the filter only removes pairs that provably cannot change it.
template <int WT>
constexpr bool has_root_filter() {
    return WT == WT_EUC_2D || WT == WT_CEIL_2D || WT == WT_ATT || WT == WT_EUC_2D_ICOORD ||
           WT == WT_CEIL_2D_ICOORD || WT == WT_ATT_ICOORD;
}

// ---- new-edge bound: no root at all ----------------------------------------------------------------------
// delta = d(a,b) + d(a1,b1) - d(a,a1) - d(b,b1) >= d(a,b) - d(a,a1) - d(b,b1), and every sqrt metric satisfies
// d(a,b) >= r - 1/2 with r = sqrt(s) (nint; ceil and ATT even give d >= r).  So delta < bound forces
//     r < T,   T = bound + d(a,a1) + d(b,b1) + prune_margin        (prune_margin = 1/2 + fp slack),
// i.e. s < T*T (ATT: s < 10*T*T).  A pair that fails this cannot beat the bound whatever its second edge is;
// it costs two subtractions, a multiply, an fma, an add, a multiply and a compare.  On a tour whose edges are
// short compared with the instance (any constructed tour) well over 99 % of the pairs end here.
template <int WT>
__device__ __forceinline__ bool new_edge_can_improve(double ax, double ay, double bx, double by, double T) {
    const double dx = ax - bx, dy = ay - by;
    const double s = fma(dx, dx, dy * dy);
    const double t2 = (WT == WT_ATT || WT == WT_ATT_ICOORD) ? 10.0 * T * T : T * T;
    return T > 0.0 && s < t2;
}

template <int WT>
__device__ __forceinline__ double approx_root_dist(double ax, double ay, double bx, double by) {
    const double dx = ax - bx, dy = ay - by;
    const double s = fma(dx, dx, dy * dy);   // a bound, not a reference value: the fused form is fine here
    return __builtin_amdgcn_sqrt((WT == WT_ATT || WT == WT_ATT_ICOORD) ? s * 0.1 : s);
}

// Runtime dispatch over (weight type, integer cost) -> template instance.
#define TSP_DISPATCH_METRIC(WT_RT, INT_RT, ...)                                               \
    do {                                                                                      \
        const int wt__ = (WT_RT);                                                             \
        const bool int__ = (INT_RT) != 0;                                                     \
        auto call__ = [&](auto wt_c, auto int_c) {                                            \
            constexpr int WTC = decltype(wt_c)::value;                                        \
            constexpr bool INTC = decltype(int_c)::value;                                     \
            __VA_ARGS__                                                                       \
        };                                                                                    \
        auto pick_int__ = [&](auto wt_c) {                                                    \
            if (int__) call__(wt_c, std::true_type{}); else call__(wt_c, std::false_type{});  \
        };                                                                                    \
        switch (wt__) {                                                                       \
        case tsp::WT_EUC_2D_ICOORD: call__(std::integral_constant<int, tsp::WT_EUC_2D_ICOORD>{}, std::true_type{}); break; \
        case tsp::WT_CEIL_2D_ICOORD: call__(std::integral_constant<int, tsp::WT_CEIL_2D_ICOORD>{}, std::true_type{}); break; \
        case tsp::WT_ATT_ICOORD: call__(std::integral_constant<int, tsp::WT_ATT_ICOORD>{}, std::true_type{}); break; \
        case tsp::WT_ATT: pick_int__(std::integral_constant<int, tsp::WT_ATT>{}); break;       \
        case tsp::WT_MAN_2D: pick_int__(std::integral_constant<int, tsp::WT_MAN_2D>{}); break; \
        case tsp::WT_MAX_2D: pick_int__(std::integral_constant<int, tsp::WT_MAX_2D>{}); break; \
        case tsp::WT_CEIL_2D: pick_int__(std::integral_constant<int, tsp::WT_CEIL_2D>{}); break; \
        case tsp::WT_GEO: pick_int__(std::integral_constant<int, tsp::WT_GEO>{}); break;       \
        default: pick_int__(std::integral_constant<int, tsp::WT_EUC_2D>{}); break;             \
        }                                                                                     \
    } while (0)

}  // namespace tsp
