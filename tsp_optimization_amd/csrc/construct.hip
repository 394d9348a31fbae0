// construct.hip -- greedy() / grasp() (src/heuristics.c:18-156) for B starting nodes at once, and the
// n x n distance matrix build.
//
// Construction: one workgroup per start.  Each of the n steps is a block-wide arg-min of
// calc_dist(cur, k) over the unvisited k, lowest index winning ties (the reference's strict '<'
// scan in index order, :51 / :117).  GRASP's runner-up is "the running minimum just before the
// final one" (:117-122), i.e. the arg-min over unvisited k < best -- a second, shorter arg-min that
// is only needed on the ~10 % of steps whose draw is >= GRASP_RAND (:127-128).
#include "tsp_internal.hpp"

#include <cfloat>

#pragma clang fp contract(off)

namespace tsp {

constexpr int kConsThreads = 1024;
constexpr double kGraspPickBest = 0.9;  // src/heuristics.c:10

struct ArgMin {
    double d;
    int k;
};

__device__ __forceinline__ bool lt(const ArgMin &a, const ArgMin &b) { return a.d < b.d || (a.d == b.d && a.k < b.k); }

// block-wide (d, k) lexicographic minimum; k == INT_MAX means "none"
__device__ __forceinline__ ArgMin block_argmin(ArgMin v, double *s_d, int *s_k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ArgMin o;
        o.d = __shfl_xor(v.d, off);
        o.k = __shfl_xor(v.k, off);
        if (lt(o, v)) v = o;
    }
    const int tid = threadIdx.x;
    __syncthreads();
    if ((tid & 63) == 0) { s_d[tid >> 6] = v.d; s_k[tid >> 6] = v.k; }
    __syncthreads();
    ArgMin r;
    r.d = s_d[0]; r.k = s_k[0];
    for (int w = 1; w < kConsThreads / 64; ++w) {
        ArgMin o;
        o.d = s_d[w]; o.k = s_k[w];
        if (lt(o, r)) r = o;
    }
    return r;
}

template <int WT, bool INT, bool IS_GRASP>
__global__ __launch_bounds__(kConsThreads) void k_construct(const double2 *__restrict__ coord, int n,
                                                           const int *__restrict__ starts,
                                                           const double *__restrict__ urand,
                                                           unsigned char *__restrict__ visited_all,
                                                           int *__restrict__ succ_all, double *__restrict__ obj,
                                                           int *__restrict__ status) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int start = starts[b];
    if (start < 0 || start >= n) {  // heuristics.c:20 / :84 (negative starts would index out of bounds there)
        if (tid == 0) { status[b] = TSP_WRONG_STARTING_NODE; obj[b] = 0.0; }
        return;
    }
    unsigned char *visited = visited_all + (size_t)b * n;
    int *succ = succ_all + (size_t)b * n;
    const double *u = IS_GRASP ? urand + (size_t)b * n : nullptr;
    __shared__ double s_d[kConsThreads / 64];
    __shared__ int s_k[kConsThreads / 64];

    for (int k = tid; k < n; k += kConsThreads) visited[k] = (k == start) ? 1 : 0;
    __syncthreads();

    double total = 0.0;
    int cur = start;
    for (int step = 0;; ++step) {
        const double2 c = coord[cur];
        ArgMin mine;
        mine.d = DBL_MAX; mine.k = 0x7fffffff;
        for (int k = tid; k < n; k += kConsThreads) {
            if (k == cur || visited[k]) continue;
            const double2 o = coord[k];
            const double d = dist_xy<WT, INT>(c.x, c.y, o.x, o.y);
            if (d < mine.d) { mine.d = d; mine.k = k; }
        }
        const ArgMin best = block_argmin(mine, s_d, s_k);
        int pick = best.k == 0x7fffffff ? -1 : best.k;
        double pick_d = best.d;
        if constexpr (IS_GRASP) {
            const double draw = u[step];
            if (!(draw < kGraspPickBest) && pick >= 0) {
                ArgMin m2;
                m2.d = DBL_MAX; m2.k = 0x7fffffff;
                for (int k = tid; k < pick; k += kConsThreads) {
                    if (k == cur || visited[k]) continue;
                    const double2 o = coord[k];
                    const double d = dist_xy<WT, INT>(c.x, c.y, o.x, o.y);
                    if (d < m2.d) { m2.d = d; m2.k = k; }
                }
                const ArgMin runner = block_argmin(m2, s_d, s_k);
                if (runner.k != 0x7fffffff) { pick = runner.k; pick_d = runner.d; }
            }
        }
        if (pick < 0) {  // every node visited: close the cycle
            if (tid == 0) succ[cur] = start;
            if constexpr (IS_GRASP) {  // heuristics.c:135
                const double2 s0 = coord[start];
                total += dist_xy<WT, INT>(c.x, c.y, s0.x, s0.y);
            }
            break;
        }
        if (tid == 0) { succ[cur] = pick; visited[pick] = 1; }
        total += pick_d;
        cur = pick;
        __syncthreads();  // visited[pick] before the next scan
    }
    {   // heuristics.c:74 / :152
        const double2 c = coord[cur], s0 = coord[start];
        total += dist_xy<WT, INT>(c.x, c.y, s0.x, s0.y);
    }
    if (tid == 0) { obj[b] = total; status[b] = TSP_OK; }
}

// LDS-resident variant (n <= 16384, 16 n bytes of LDS): four waves, one per SIMD, so the per-step
// reduction work is not multiplied by sixteen co-resident waves fighting for the same issue slots (that,
// not memory, bounds k_construct: ~2.5 us per step at any n).  Thread t owns the candidates t + 256 m and
// keeps their "unvisited" flags in a 64-bit mask; coordinates sit in LDS.  One barrier per arg-min: each
// wave reduces (d, k) with shuffles, the lane holding the wave's winner writes (d, k, x, y) into the wave's
// slot of a rotating LDS table, and after the barrier every thread scans the four slots, so the picked
// node's coordinates arrive with the result.  Same arg-min and tie-break as k_construct.
constexpr int kConsLdsThreads = 256;
constexpr int kConsLdsMaxN = 64 * kConsLdsThreads;
struct alignas(16) ConsSlot { double d, x, y; int k; int pad; };

__device__ __forceinline__ ConsSlot cons_argmin(ArgMin mine, double mx, double my, ConsSlot *table) {
    ArgMin w = mine;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ArgMin o;
        o.d = __shfl_xor(w.d, off);
        o.k = __shfl_xor(w.k, off);
        if (lt(o, w)) w = o;
    }
    const int tid = threadIdx.x;
    if (mine.k == w.k && (w.k != 0x7fffffff || (tid & 63) == 0)) {  // the winner's lane (lane 0 if the wave has none)
        ConsSlot sl;
        sl.d = w.d; sl.k = w.k; sl.x = mx; sl.y = my; sl.pad = 0;
        table[tid >> 6] = sl;
    }
    __syncthreads();
    ConsSlot r = table[0];
#pragma unroll
    for (int q = 1; q < kConsLdsThreads / 64; ++q) {
        const ConsSlot o = table[q];
        if (o.d < r.d || (o.d == r.d && o.k < r.k)) r = o;
    }
    return r;
}

template <int WT, bool INT, bool IS_GRASP>
__global__ __launch_bounds__(kConsLdsThreads) void k_construct_lds(const double2 *__restrict__ coord, int n,
                                                                  const int *__restrict__ starts,
                                                                  const double *__restrict__ urand,
                                                                  int *__restrict__ succ_all, double *__restrict__ obj,
                                                                  int *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char cons_smem[];
    ConsSlot *s_tab = reinterpret_cast<ConsSlot *>(cons_smem);                          // 4 tables x 4 slots
    double2 *s_xy = reinterpret_cast<double2 *>(cons_smem + 4 * (kConsLdsThreads / 64) * sizeof(ConsSlot));
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int start = starts[b];
    if (start < 0 || start >= n) {
        if (tid == 0) { status[b] = TSP_WRONG_STARTING_NODE; obj[b] = 0.0; }
        return;
    }
    int *succ = succ_all + (size_t)b * n;
    const double *u = IS_GRASP ? urand + (size_t)b * n : nullptr;
    const int M = (n + kConsLdsThreads - 1) / kConsLdsThreads;

    unsigned long long alive = 0;
    for (int m = 0; m < M; ++m) {
        const int k = tid + m * kConsLdsThreads;
        if (k < n) { s_xy[k] = coord[k]; if (k != start) alive |= 1ull << m; }
    }
    const double2 c_start = coord[start];
    double curx = c_start.x, cury = c_start.y;
    __syncthreads();

    double total = 0.0;
    int cur = start, rot = 0;
    for (int step = 0;; ++step) {
        ArgMin mine;
        mine.d = DBL_MAX; mine.k = 0x7fffffff;
        double mx = 0.0, my = 0.0;
        // candidates four at a time: the LDS reads and the four distance chains overlap (one wave per SIMD
        // has no other wave to hide latency behind); groups whose four candidates are all visited are skipped
        for (int g = 0; g < M; g += 4) {
            const unsigned nib = (unsigned)(alive >> g) & 0xFu;
            if (!nib) continue;
            double2 c[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] = s_xy[min(tid + (g + q) * kConsLdsThreads, n - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double d = dist_xy<WT, INT>(curx, cury, c[q].x, c[q].y);
                if (((nib >> q) & 1u) && d < mine.d) {   // g, q ascend: the lowest index wins ties
                    mine.d = d; mine.k = tid + (g + q) * kConsLdsThreads; mx = c[q].x; my = c[q].y;
                }
            }
        }
        ConsSlot best = cons_argmin(mine, mx, my, s_tab + rot * (kConsLdsThreads / 64));
        rot = (rot + 1) & 3;
        if constexpr (IS_GRASP) {
            const double draw = u[step];
            if (!(draw < kGraspPickBest) && best.k != 0x7fffffff) {
                ArgMin m2;
                m2.d = DBL_MAX; m2.k = 0x7fffffff;
                double m2x = 0.0, m2y = 0.0;
                for (int g = 0; g < M && tid + g * kConsLdsThreads < best.k; g += 4) {
                    const unsigned nib = (unsigned)(alive >> g) & 0xFu;
                    if (!nib) continue;
                    double2 c[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) c[q] = s_xy[min(tid + (g + q) * kConsLdsThreads, n - 1)];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int k = tid + (g + q) * kConsLdsThreads;
                        const double d = dist_xy<WT, INT>(curx, cury, c[q].x, c[q].y);
                        if (((nib >> q) & 1u) && k < best.k && d < m2.d) { m2.d = d; m2.k = k; m2x = c[q].x; m2y = c[q].y; }
                    }
                }
                const ConsSlot runner = cons_argmin(m2, m2x, m2y, s_tab + rot * (kConsLdsThreads / 64));
                rot = (rot + 1) & 3;
                if (runner.k != 0x7fffffff) best = runner;
            }
        }
        if (best.k == 0x7fffffff) {  // every node visited: close the cycle
            if (tid == 0) succ[cur] = start;
            if constexpr (IS_GRASP) total += dist_xy<WT, INT>(curx, cury, c_start.x, c_start.y);  // heuristics.c:135
            total += dist_xy<WT, INT>(curx, cury, c_start.x, c_start.y);                           // :74 / :152
            break;
        }
        const int pick = best.k;
        if ((pick & (kConsLdsThreads - 1)) == tid) {   // the owner retires the node and records the edge
            alive &= ~(1ull << (pick / kConsLdsThreads));
            succ[cur] = pick;
        }
        total += best.d;
        cur = pick; curx = best.x; cury = best.y;
    }
    if (tid == 0) { obj[b] = total; status[b] = TSP_OK; }
}

// ---- distance matrix ------------------------------------------------------------------------
// Block = kDmRows rows x 1024 columns.  Each lane keeps the coordinates of its 4 consecutive
// columns in registers for all rows of the block, computes 4 entries per row and streams them out
// with one 16-byte (int32) or two 16-byte (double) non-temporal stores: a wave writes 1 KiB / 2 KiB
// contiguous per row.  The row's coordinates are wave-uniform (scalar loads).  Write-bandwidth
// bound: 4 n^2 (int32) or 8 n^2 (double) bytes to HBM against 16 n bytes of coordinates.
constexpr int kDmRows = 16;

template <int WT, bool INT, typename OUT>
__global__ __launch_bounds__(256) void k_dist_matrix(const double2 *__restrict__ coord, int n, OUT *__restrict__ out) {
    constexpr bool I32 = sizeof(OUT) == 4;
    const int i0 = blockIdx.y * kDmRows;
    const int base = blockIdx.x * 1024;
    if (base >= n) return;
    // int32: lane owns 4 consecutive columns (one 16-byte store).  double: lane owns two column
    // pairs 512 apart (two 16-byte stores), so that every store instruction of a wave is 1 KiB contiguous.
    int col[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        col[k] = I32 ? base + threadIdx.x * 4 + k : base + threadIdx.x * 2 + (k & 1) + (k >> 1) * 512;
    double2 cj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cj[k] = coord[min(col[k], n - 1)];
    const int i1 = min(i0 + kDmRows, n);
    for (int i = i0; i < i1; ++i) {
        const double2 ci = coord[i];
        OUT v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double d = (col[k] == i) ? 0.0 : dist_xy<WT, INT>(ci.x, ci.y, cj[k].x, cj[k].y);
            v[k] = (OUT)d;
        }
        OUT *row = out + (size_t)i * n;
        if constexpr (I32) {
            if (col[3] < n && (((size_t)i * n + col[0]) * 4) % 16 == 0) {
                typedef int v4i __attribute__((ext_vector_type(4)));
                v4i pk = {(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
                __builtin_nontemporal_store(pk, reinterpret_cast<v4i *>(row + col[0]));
            } else {
                for (int k = 0; k < 4; ++k) if (col[k] < n) row[col[k]] = v[k];
            }
        } else {
            typedef double v2d __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (col[2 * h + 1] < n && (((size_t)i * n + col[2 * h]) * 8) % 16 == 0) {
                    v2d a = {(double)v[2 * h], (double)v[2 * h + 1]};
                    __builtin_nontemporal_store(a, reinterpret_cast<v2d *>(row + col[2 * h]));
                } else {
                    for (int k = 2 * h; k < 2 * h + 2; ++k) if (col[k] < n) row[col[k]] = v[k];
                }
            }
        }
    }
}

}  // namespace tsp

using namespace tsp;

extern "C" {

int tsp_dev_construct(tsp_dev_inst *inst, int kind, int B, const int *starts, const double *urand, int *succ,
                      int succ_stride, int64_t tour_stride, double *obj, int *status_out) {
    if (!inst || !starts || !succ || !obj || B < 1 || succ_stride < 1) return TSP_DEV_E_ARG;
    if (kind != TSP_CONSTRUCT_GREEDY && kind != TSP_CONSTRUCT_GRASP) return TSP_DEV_E_ARG;
    if (kind == TSP_CONSTRUCT_GRASP && !urand) return TSP_DEV_E_ARG;
    const int n = inst->n;
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    hipStream_t s = inst->ctx->stream;
    int *d_starts = nullptr, *d_succ = nullptr, *d_status = nullptr;
    double *d_urand = nullptr, *d_obj = nullptr;
    unsigned char *d_vis = nullptr;
    TSP_HIP_TRY(hipMalloc(&d_starts, sizeof(int) * (size_t)B));
    TSP_HIP_TRY(hipMalloc(&d_status, sizeof(int) * (size_t)B));
    TSP_HIP_TRY(hipMalloc(&d_obj, sizeof(double) * (size_t)B));
    TSP_HIP_TRY(hipMalloc(&d_succ, sizeof(int) * (size_t)B * n));
    TSP_HIP_TRY(hipMalloc(&d_vis, (size_t)B * n));
    TSP_HIP_TRY(hipMemcpyAsync(d_starts, starts, sizeof(int) * (size_t)B, hipMemcpyHostToDevice, s));
    TSP_HIP_TRY(hipMemsetAsync(d_succ, 0, sizeof(int) * (size_t)B * n, s));  // CALLOC'd edges, solver.c:270
    if (kind == TSP_CONSTRUCT_GRASP) {
        TSP_HIP_TRY(hipMalloc(&d_urand, sizeof(double) * (size_t)B * n));
        TSP_HIP_TRY(hipMemcpyAsync(d_urand, urand, sizeof(double) * (size_t)B * n, hipMemcpyHostToDevice, s));
    }
    const char *no_lds = getenv("TSP_CONSTRUCT_GLOBAL");
    const size_t lds_bytes = 4 * (kConsLdsThreads / 64) * sizeof(ConsSlot) + sizeof(double2) * (size_t)n;
    const bool use_lds = !(no_lds && *no_lds == '1') && n <= kConsLdsMaxN && lds_bytes <= (size_t)160 * 1024;
    hipError_t attr_err = hipSuccess;
    TSP_DISPATCH_METRIC(inst->wtype, inst->integer_cost, {
        if (use_lds) {
            if (kind == TSP_CONSTRUCT_GRASP) {
                auto kf = k_construct_lds<WTC, INTC, true>;
                attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                hipLaunchKernelGGL(kf, dim3(B), dim3(kConsLdsThreads), lds_bytes, s, inst->d_coord, n, d_starts, d_urand,
                                   d_succ, d_obj, d_status);
            } else {
                auto kf = k_construct_lds<WTC, INTC, false>;
                attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                hipLaunchKernelGGL(kf, dim3(B), dim3(kConsLdsThreads), lds_bytes, s, inst->d_coord, n, d_starts,
                                   (const double *)nullptr, d_succ, d_obj, d_status);
            }
        } else if (kind == TSP_CONSTRUCT_GRASP)
            hipLaunchKernelGGL((k_construct<WTC, INTC, true>), dim3(B), dim3(kConsThreads), 0, s, inst->d_coord, n,
                               d_starts, d_urand, d_vis, d_succ, d_obj, d_status);
        else
            hipLaunchKernelGGL((k_construct<WTC, INTC, false>), dim3(B), dim3(kConsThreads), 0, s, inst->d_coord, n,
                               d_starts, (const double *)nullptr, d_vis, d_succ, d_obj, d_status);
    });
    TSP_HIP_TRY(attr_err);
    std::vector<int> h_succ((size_t)B * n), h_status((size_t)B);
    TSP_HIP_TRY(hipMemcpyAsync(h_succ.data(), d_succ, sizeof(int) * (size_t)B * n, hipMemcpyDeviceToHost, s));
    TSP_HIP_TRY(hipMemcpyAsync(h_status.data(), d_status, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, s));
    TSP_HIP_TRY(hipMemcpyAsync(obj, d_obj, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, s));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    int worst = TSP_OK;
    for (int b = 0; b < B; ++b) {
        if (status_out) status_out[b] = h_status[b];
        if (h_status[b] != TSP_OK) { worst = h_status[b]; continue; }  // edges untouched, like :20
        int *sp = succ + (size_t)b * tour_stride;
        for (int v = 0; v < n; ++v) sp[(size_t)v * succ_stride] = h_succ[(size_t)b * n + v];
    }
    (void)hipFree(d_starts); (void)hipFree(d_status); (void)hipFree(d_obj); (void)hipFree(d_succ);
    (void)hipFree(d_vis); (void)hipFree(d_urand);
    return (B == 1) ? worst : TSP_OK;
}

int tsp_dev_dist_matrix(tsp_dev_inst *inst, void *out_host, int as_int32, float *kernel_ms) {
    if (!inst) return TSP_DEV_E_ARG;
    if (as_int32 && !inst->integer_cost && inst->wtype_public != TSP_CEIL_2D) return TSP_DEV_E_ARG;
    const int n = inst->n;
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    hipStream_t s = inst->ctx->stream;
    const size_t bytes = (size_t)n * n * (as_int32 ? 4 : 8);
    void *d_out = nullptr;
    TSP_HIP_TRY(hipMalloc(&d_out, bytes));
    hipEvent_t e0, e1;
    TSP_HIP_TRY(hipEventCreate(&e0));
    TSP_HIP_TRY(hipEventCreate(&e1));
    const dim3 grid((n + 1023) / 1024, (n + kDmRows - 1) / kDmRows);
    const int reps = out_host ? 1 : 10;  // timing-only calls: warm once, then average back-to-back launches
    float ms = 0.f;
    auto launch = [&]() {
        TSP_DISPATCH_METRIC(inst->wtype, inst->integer_cost, {
            if (as_int32)
                hipLaunchKernelGGL((k_dist_matrix<WTC, INTC, int>), grid, dim3(256), 0, s, inst->d_coord, n, (int *)d_out);
            else
                hipLaunchKernelGGL((k_dist_matrix<WTC, INTC, double>), grid, dim3(256), 0, s, inst->d_coord, n,
                                   (double *)d_out);
        });
    };
    if (!out_host) launch();
    TSP_HIP_TRY(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) launch();
    TSP_HIP_TRY(hipEventRecord(e1, s));
    TSP_HIP_TRY(hipEventSynchronize(e1));
    TSP_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    ms /= (float)reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (out_host) {
        TSP_HIP_TRY(hipMemcpyAsync(out_host, d_out, bytes, hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
    }
    TSP_HIP_TRY(hipGetLastError());
    (void)hipFree(d_out);
    if (kernel_ms) *kernel_ms = ms;
    return TSP_OK;
}

}  // extern "C"
