// extramileage.hip -- HEU_extramileage (src/heuristics.c:208-314): farthest pair, then n-2 cheapest
// insertions.  The reference's O(n^3) triple loop becomes one scan + one apply launch per inserted
// node: every (unvisited node c, tour edge slot j) pair is priced in parallel,
//     deltacost = (d(a,c) + d(c,b)) - d(a,b)            (:270-273, same association)
// and the arg-min with the reference's tie-break (strict '<' in a loop over c ascending, then j
// ascending, :262-280 => the lexicographically first (c, j) among the minima) picks the insertion.
// Slots keep the reference's edges_visited order: the replaced slot gets (a,c), the new last slot
// gets (c,b) (:296-297), because later ties are broken by slot index.
#include "two_opt_common.hpp"

#include <cfloat>

#pragma clang fp contract(off)

namespace tsp {

constexpr int kXmRows = 32;

struct XmState {
    int num_visited;  // also the number of edge slots
    int done;
    double obj;
};

struct alignas(16) XmSlot {  // one tour edge (a -> b) with its endpoints' coordinates and its length
    double ax, ay, bx, by, len;
    int a, b;
};
static_assert(sizeof(XmSlot) == 48, "XmSlot must be 48 bytes");

// ---- farthest pair (:226-235): first (i<j) in loop order among the maxima ---------------------
template <int WT, bool INT>
__global__ __launch_bounds__(kScanThreads) void k_xm_far(const double2 *__restrict__ coord, int n,
                                                         Partial *__restrict__ partials) {
    const int r0 = blockIdx.y * kXmRows, c0 = blockIdx.x * kScanThreads;
    const int tid = threadIdx.x;
    Partial *slot = partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    double bd = 0.0;  // max_dist starts at 0 and needs a strictly larger distance
    u64 key = kNoKey;
    if (c0 + kScanThreads - 1 > r0) {
        const int j = c0 + tid;
        const double2 cj = coord[min(j, n - 1)];
        const int r1 = min(r0 + kXmRows, n - 1);
        for (int i = r0; i < r1; ++i) {
            const double2 ci = coord[i];
            const double d = dist_xy<WT, INT>(ci.x, ci.y, cj.x, cj.y);
            if (j > i && j < n && d > bd) { bd = d; key = make_key(i, j); }
        }
    }
    // block arg-max of (dist, then smallest key): negate to reuse the arg-min helpers
    double nd = -bd;
    __shared__ double s_d[kScanThreads / 64];
    __shared__ u64 s_k[kScanThreads / 64];
    block_argmin<true>(nd, key, s_d, s_k);
    if (tid == 0) { Partial p; p.delta = nd; p.i = key_i(key); p.j = key_j(key); *slot = p; }
}

template <int WT, bool INT>
__global__ __launch_bounds__(kApplyThreads) void k_xm_init(const double2 *__restrict__ coord, int n,
                                                           const Partial *__restrict__ partials, int nslots,
                                                           XmSlot *__restrict__ slots, unsigned char *__restrict__ visited,
                                                           int *__restrict__ succ, XmState *__restrict__ st) {
    __shared__ double s_d[kApplyThreads / 64];
    __shared__ u64 s_k[kApplyThreads / 64];
    const int tid = threadIdx.x;
    double nd = 0.0;
    u64 key = kNoKey;
    for (int s = tid; s < nslots; s += kApplyThreads) {
        const Partial p = partials[s];
        const u64 k = make_key(p.i, p.j);
        if (k != kNoKey && better(p.delta, k, nd, key)) { nd = p.delta; key = k; }
    }
    block_argmin<true>(nd, key, s_d, s_k);
    for (int v = tid; v < n; v += kApplyThreads) { visited[v] = 0; succ[v] = 0; }  // CALLOC'd edges, solver.c:270
    __syncthreads();
    if (tid == 0) {
        int A = 0, B = 1;  // :213-214 when every distance is 0
        if (key != kNoKey) { A = key_i(key); B = key_j(key); }
        const double2 ca = coord[A], cb = coord[B];
        const double d = dist_xy<WT, INT>(ca.x, ca.y, cb.x, cb.y);
        XmSlot e1, e2;
        e1.a = A; e1.b = B; e1.ax = ca.x; e1.ay = ca.y; e1.bx = cb.x; e1.by = cb.y; e1.len = d;
        e2.a = B; e2.b = A; e2.ax = cb.x; e2.ay = cb.y; e2.bx = ca.x; e2.by = ca.y; e2.len = d;
        slots[0] = e1; slots[1] = e2;
        succ[A] = B; succ[B] = A;
        visited[A] = 1; visited[B] = 1;
        st->num_visited = 2; st->done = 0;
        st->obj = 2 * d;  // :251
    }
}

// ---- one insertion: price every (unvisited node, slot) ---------------------------------------
template <int WT, bool INT>
__global__ __launch_bounds__(kScanThreads) void k_xm_scan(const double2 *__restrict__ coord, int n,
                                                          const XmSlot *__restrict__ slots,
                                                          const unsigned char *__restrict__ visited,
                                                          const XmState *__restrict__ st, Partial *__restrict__ partials) {
    if (st->done) return;
    const int nv = st->num_visited;
    const int j0 = blockIdx.y * kXmRows;
    if (j0 >= nv) return;
    const int tid = threadIdx.x;
    const int c = blockIdx.x * kScanThreads + tid;
    __shared__ XmSlot s_slots[kXmRows];
    const int nr = min(kXmRows, nv - j0);
    if (tid < nr) s_slots[tid] = slots[j0 + tid];
    const bool live = c < n && !visited[min(c, n - 1)];
    const double2 cc = coord[min(c, n - 1)];
    __syncthreads();
    double bd = DBL_MAX;
    u64 key = kNoKey;
    for (int r = 0; r < nr; ++r) {
        const XmSlot e = s_slots[r];
        const double delta = dist_xy<WT, INT>(e.ax, e.ay, cc.x, cc.y) + dist_xy<WT, INT>(cc.x, cc.y, e.bx, e.by) - e.len;
        if (live && delta < bd) { bd = delta; key = make_key(c, j0 + r); }
    }
    __shared__ double s_d[kScanThreads / 64];
    __shared__ u64 s_k[kScanThreads / 64];
    block_argmin<true>(bd, key, s_d, s_k);
    if (tid == 0) {
        Partial p; p.delta = bd; p.i = key_i(key); p.j = key_j(key);
        partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = p;
    }
}

template <int WT, bool INT>
__global__ __launch_bounds__(kApplyThreads) void k_xm_apply(const double2 *__restrict__ coord, int n, int gx,
                                                            const Partial *__restrict__ partials,
                                                            XmSlot *__restrict__ slots, unsigned char *__restrict__ visited,
                                                            int *__restrict__ succ, XmState *__restrict__ st) {
    if (st->done) return;
    __shared__ double s_d[kApplyThreads / 64];
    __shared__ u64 s_k[kApplyThreads / 64];
    const int tid = threadIdx.x;
    const int nv = st->num_visited;
    const int nslots = ((nv + kXmRows - 1) / kXmRows) * gx;
    double bd = DBL_MAX;
    u64 key = kNoKey;
    for (int s = tid; s < nslots; s += kApplyThreads) {
        const Partial p = partials[s];
        const u64 k = make_key(p.i, p.j);
        if (k != kNoKey && better(p.delta, k, bd, key)) { bd = p.delta; key = k; }
    }
    block_argmin<true>(bd, key, s_d, s_k);
    if (tid == 0) {
        if (key == kNoKey || nv >= n) { st->done = 1; return; }  // :283-285
        const int c = key_i(key), j = key_j(key);
        const XmSlot old = slots[j];
        const double2 cc = coord[c];
        XmSlot e1, e2;  // :288-297
        e1.a = old.a; e1.b = c; e1.ax = old.ax; e1.ay = old.ay; e1.bx = cc.x; e1.by = cc.y;
        e1.len = dist_xy<WT, INT>(old.ax, old.ay, cc.x, cc.y);
        e2.a = c; e2.b = old.b; e2.ax = cc.x; e2.ay = cc.y; e2.bx = old.bx; e2.by = old.by;
        e2.len = dist_xy<WT, INT>(cc.x, cc.y, old.bx, old.by);
        slots[j] = e1;
        slots[nv] = e2;
        succ[old.a] = c; succ[c] = old.b;
        visited[c] = 1;
        st->obj += bd;  // :301
        st->num_visited = nv + 1;
        if (nv + 1 >= n) st->done = 1;
    }
}

}  // namespace tsp

using namespace tsp;

extern "C" int tsp_dev_extramileage(tsp_dev_inst *inst, int *succ, int succ_stride, double *obj) {
    if (!inst || !succ || !obj || succ_stride < 1) return TSP_DEV_E_ARG;
    const int n = inst->n;
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    hipStream_t s = inst->ctx->stream;
    const int gx = (n + kScanThreads - 1) / kScanThreads, gy = (n + kXmRows - 1) / kXmRows;
    DevBuf<Partial> d_part;
    DevBuf<XmSlot> d_slots;
    DevBuf<unsigned char> d_vis;
    DevBuf<int> d_succ;
    DevBuf<XmState> d_st;
    TSP_HIP_TRY(d_part.alloc((size_t)gx * gy));
    TSP_HIP_TRY(d_slots.alloc((size_t)(n + 1)));
    TSP_HIP_TRY(d_vis.alloc((size_t)n));
    TSP_HIP_TRY(d_succ.alloc((size_t)n));
    TSP_HIP_TRY(d_st.alloc(1));
    TSP_DISPATCH_METRIC(inst->wtype, inst->integer_cost, {
        hipLaunchKernelGGL((k_xm_far<WTC, INTC>), dim3(gx, gy), dim3(kScanThreads), 0, s, inst->d_coord, n, d_part);
        hipLaunchKernelGGL((k_xm_init<WTC, INTC>), dim3(1), dim3(kApplyThreads), 0, s, inst->d_coord, n, d_part, gx * gy,
                           d_slots, d_vis, d_succ, d_st);
        for (int step = 2; step < n; ++step) {
            hipLaunchKernelGGL((k_xm_scan<WTC, INTC>), dim3(gx, gy), dim3(kScanThreads), 0, s, inst->d_coord, n, d_slots,
                               d_vis, d_st, d_part);
            hipLaunchKernelGGL((k_xm_apply<WTC, INTC>), dim3(1), dim3(kApplyThreads), 0, s, inst->d_coord, n, gx, d_part,
                               d_slots, d_vis, d_succ, d_st);
        }
    });
    std::vector<int> h_succ((size_t)n);
    XmState h_st;
    TSP_HIP_TRY(hipMemcpyAsync(h_succ.data(), d_succ, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
    TSP_HIP_TRY(hipMemcpyAsync(&h_st, d_st, sizeof h_st, hipMemcpyDeviceToHost, s));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    for (int v = 0; v < n; ++v) succ[(size_t)v * succ_stride] = h_succ[v];
    *obj = h_st.obj;
    return TSP_OK;
}
