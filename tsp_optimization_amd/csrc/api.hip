// api.hip -- context / instance handles and the host-tour entry points of include/tsp_hip.h.
#include "tsp_internal.hpp"

#include <stdlib.h>
#include <mutex>

#include <algorithm>
#include <time.h>

#pragma clang fp contract(off)

using namespace tsp;

// implemented in two_opt_grid.hip / two_opt_lds.hip
int tsp_grid_run(tsp_dev_tours *t, int mode, tsp_dev_tabu *tabu, int iter, int tenure, int64_t max_steps,
                 double time_limit_s, int sync, int *all_done);
int tsp_lds_run(tsp_dev_tours *t, int mode, double time_limit_s, int *all_done);
bool tsp_lds_fits(const tsp_dev_inst *inst);
// two_opt_cluster.hip
bool tsp_cluster_fits(const tsp_dev_tours *t, int mode);
bool tsp_cluster_sorted(const tsp_dev_tours *t, int mode);
int tsp_cluster_size(const tsp_dev_tours *t, int mode);
int tsp_cluster_run(tsp_dev_tours *t, int mode, int C, int64_t max_steps, double time_limit_s, int *all_done, int *fell_through,
                    tsp_dev_tabu *tabu = nullptr, int iter = 0, int tenure = 0);

void *tsp_io_pool(tsp_dev_inst *inst, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > inst->io_pool_bytes) {
        (void)hipStreamSynchronize(inst->ctx->stream);   // nobody may still be using the old block
        (void)hipFree(inst->io_pool);
        inst->io_pool = nullptr; inst->io_pool_bytes = 0;
        const size_t want = std::max(bytes, (size_t)1 << 16);
        if (hipMalloc(&inst->io_pool, want) != hipSuccess) { inst->io_pool = nullptr; return nullptr; }
        inst->io_pool_bytes = want;
    }
    return inst->io_pool;
}

namespace tsp {
void read_switches(Switches *sw) {
    static const char *const names[SW_COUNT] = {
#define TSP_SW_NAME(name) "TSP_" #name,
        TSP_SWITCH_LIST(TSP_SW_NAME)
#undef TSP_SW_NAME
    };
    for (int k = 0; k < SW_COUNT; ++k) {
        const char *v = getenv(names[k]);
        sw->has[k] = v && *v;
        sw->v[k] = sw->has[k] ? atoi(v) : 0;
    }
}
static thread_local char g_last_error[512] = "";
void set_last_error(const char *what, hipError_t e, const char *file, int line) {
    snprintf(g_last_error, sizeof g_last_error, "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
}
}  // namespace tsp

// Single-tour calls (alg_2opt in a VNS / tabu / GA loop) reuse one tours handle and one event pair per
// instance instead of eight hipMallocs per call.  Handles are per instance: not for concurrent use.
tsp_dev_tours *tsp_scratch_tours(tsp_dev_inst *inst, int B, bool *owned, int *rc) {
    *owned = false; *rc = TSP_OK;
    if (B == 1) {
        if (!inst->scratch1) { *rc = tsp_dev_tours_create(inst, 1, &inst->scratch1); if (*rc) return nullptr; }
        return inst->scratch1;
    }
    // batches: the handle of the last batch size is kept too (multi-start loops call with the same B again and
    // again; creating a handle is ~20 allocations and, for the sorted sweep, a host-built table)
    if (inst->scratch_b && inst->scratch_b_count != B) { tsp_dev_tours_destroy(inst->scratch_b); inst->scratch_b = nullptr; }
    if (!inst->scratch_b) {
        *rc = tsp_dev_tours_create(inst, B, &inst->scratch_b);
        if (*rc) { inst->scratch_b = nullptr; return nullptr; }
        inst->scratch_b_count = B;
    }
    return inst->scratch_b;
}

namespace {
// Rank of cell (x, y) of a 65536 x 65536 grid along the Hilbert curve (the classic xy -> d walk).
unsigned hilbert_rank16(unsigned x, unsigned y) {
    unsigned d = 0;
    for (unsigned s = 1u << 15; s > 0; s >>= 1) {
        const unsigned rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) { x = s - 1 - x; y = s - 1 - y; }
            const unsigned t = x; x = y; y = t;
        }
    }
    return d;
}
double wall_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

template <int WT, bool INT>
__global__ void k_dist_pairs(const double2 *__restrict__ coord, const int *__restrict__ pi,
                             const int *__restrict__ pj, int count, double *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const double2 a = coord[pi[t]], b = coord[pj[t]];
    out[t] = dist_xy<WT, INT>(a.x, a.y, b.x, b.y);
}

// self-test helper: the raw v_sqrt_f64 the integer-root variants build on
__global__ void k_raw_sqrt(const double *__restrict__ in, int count, double *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) out[t] = __builtin_amdgcn_sqrt(in[t]);
}

// genetic.c:51-60 : one wave per permutation would leave the sum order free; the reference adds
// edge by edge, so one block stages the edge lengths and thread 0 adds them in order when the
// costs are not integer-valued.
template <int WT, bool INT>
__global__ __launch_bounds__(256) void k_perm_cost(const double2 *__restrict__ coord, const int *__restrict__ perm,
                                                   long long perm_stride, int n, double *cost, size_t cost_stride_bytes) {
    const int *p = perm + (size_t)blockIdx.x * perm_stride;
    double *out = reinterpret_cast<double *>(reinterpret_cast<char *>(cost) + blockIdx.x * cost_stride_bytes);
    __shared__ double s_part[256];
    __shared__ double s_chunk[2048];
    const int tid = threadIdx.x;
    if constexpr (INT) {
        double acc = 0.0;
        for (int k = tid; k < n; k += 256) {
            const double2 a = coord[p[k]], b = coord[p[k + 1 == n ? 0 : k + 1]];
            acc += dist_xy<WT, INT>(a.x, a.y, b.x, b.y);
        }
        s_part[tid] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) s_part[tid] += s_part[tid + s];
            __syncthreads();
        }
        if (tid == 0) *out = s_part[0];
    } else {
        double acc = 0.0;
        for (int base = 0; base < n; base += 2048) {
            __syncthreads();
            for (int t = tid; t < 2048 && base + t < n; t += 256) {
                const int k = base + t;
                const double2 a = coord[p[k]], b = coord[p[k + 1 == n ? 0 : k + 1]];
                s_chunk[t] = dist_xy<WT, INT>(a.x, a.y, b.x, b.y);
            }
            __syncthreads();
            if (tid == 0) {
                const int m = min(2048, n - base);
                for (int t = 0; t < m; ++t) acc += s_chunk[t];
            }
        }
        if (tid == 0) *out = acc;
    }
}
}  // namespace

using tsp::DevBuf;

// fitness() of B permutations that already sit on the device (src/genetic.c:51-60); out[b] at out_stride_bytes
int tsp_perm_cost_device(tsp_dev_inst *inst, const int *d_perm, long long stride, int B, double *d_out, size_t out_stride_bytes) {
    hipStream_t s = inst->ctx->stream;
    TSP_DISPATCH_METRIC(inst->wtype, inst->integer_cost, {
        hipLaunchKernelGGL((k_perm_cost<WTC, INTC>), dim3(B), dim3(256), 0, s, inst->d_coord, d_perm, stride, inst->n, d_out,
                           out_stride_bytes);
    });
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

// two_opt_grid.hip: drivers on resident tours
int tsp_grid_rearm(tsp_dev_tours *t, int mode);
int tsp_grid_tabu_kick(tsp_dev_tours *t, tsp_dev_tabu *tabu, int a, int b, int iter, int tenure, int *accepted);
int tsp_grid_vns_kick(tsp_dev_tours *t, int p1, int p2, int p3, double *obj);
int tsp_grid_snapshot(tsp_dev_tours *t, bool restore);
int tsp_grid_resident_tabu(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter, int tenure, double time_limit_s, double *obj);
int tsp_grid_tabu_iteration(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter, int tenure, double time_limit_s, int a, int b,
                            double *best_obj, double *obj, int *improved, int *accepted);
int tsp_grid_tabu_iterations(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter0, int count, const int *tenure, int pairs, const int *ab, double time_limit_s,
                             double *best_obj, double *obj, int *improved, int *trials, int *completed, int *last_accepted);

extern "C" {

const char *tsp_dev_last_error(void) { return tsp::g_last_error; }

namespace {
// The HIP runtime's initialisation reseeds libc's random() generator (tools/rng_probe.py: srandom(123), a first device call,
// random() -> not the value the seed promises, and a different one in every process).  The reference's drivers seed once in
// main and draw with random() / rand() throughout (utility.h:36), and tabu() / HEU_VNS begin with device work here
// (HEU_2opt_greedy_iter) before their first draw: the process's generator is parked on a scratch state while the runtime comes
// up and put back exactly as the caller left it.
// (One keeper at a time, process-wide: tsp_host_population_gpus opens a context per GPU from a thread each, and two interleaved
// switches would leave the process on one of the scratch arrays.  The scratch array is static for the same reason: libc never
// points into a dead stack frame.)
std::mutex g_open_mutex;
struct LibcRandomKeeper {
    std::lock_guard<std::mutex> lock;
    char *old;
    LibcRandomKeeper() : lock(g_open_mutex) {
        static char scratch[256];
        old = initstate(1u, scratch, sizeof scratch);
    }
    ~LibcRandomKeeper() { if (old) (void)setstate(old); }
};
}  // namespace

int tsp_dev_count(void) {
    LibcRandomKeeper keep_libc_random;   // (this may be the process's first HIP call)
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int tsp_dev_open(int device, tsp_dev_ctx **out) {
    if (!out) return TSP_DEV_E_ARG;
    LibcRandomKeeper keep_libc_random;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0 || device < 0 || device >= count) {
        tsp::set_last_error("hipGetDeviceCount", e == hipSuccess ? hipErrorNoDevice : e, __FILE__, __LINE__);
        return TSP_DEV_E_NODEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) return TSP_DEV_E_NODEVICE;
    tsp_dev_ctx *c = new tsp_dev_ctx();
    struct Guard { tsp_dev_ctx *c; ~Guard() { delete c; } } guard{c};   // an early error return frees the context
    c->device = device;
    hipDeviceProp_t prop;
    TSP_HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount;
    c->lds_bytes = (int)prop.sharedMemPerBlock;
    {   // what one workgroup may be granted with hipFuncAttributeMaxDynamicSharedMemorySize (160 KiB on gfx950)
        int optin = 0;
        if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && optin > c->lds_bytes)
            c->lds_bytes = optin;
    }
    TSP_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {   // the rest of what a process's first device work sets up (allocator, a first copy on the stream), still under the keeper
        void *w = nullptr;
        int zero = 0;
        TSP_HIP_TRY(hipMalloc(&w, 256));
        TSP_HIP_TRY(hipMemcpyAsync(w, &zero, sizeof zero, hipMemcpyHostToDevice, c->stream));
        TSP_HIP_TRY(hipStreamSynchronize(c->stream));
        TSP_HIP_TRY(hipFree(w));
    }
    guard.c = nullptr;
    *out = c;
    return TSP_OK;
}

void tsp_dev_close(tsp_dev_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    delete ctx;
}

int tsp_dev_synchronize(tsp_dev_ctx *ctx) {
    if (!ctx) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return TSP_OK;
}

void *tsp_dev_stream(tsp_dev_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int tsp_dev_inst_create(tsp_dev_ctx *ctx, const double *xy, int n, int weight_type, int integer_cost,
                        tsp_dev_inst **out) {
    if (!ctx || !xy || !out || n < 3) return TSP_DEV_E_ARG;   // three nodes: one tour, nothing to improve (the reference runs it)
    TSP_HIP_TRY(hipSetDevice(ctx->device));
    tsp_dev_inst *inst = new tsp_dev_inst();
    inst->ctx = ctx;
    inst->n = n;
    struct Guard { tsp_dev_inst *i; ~Guard() { if (i) tsp_dev_inst_destroy(i); } } guard{inst};   // an early error return frees what exists
    // unknown types (the reference's parser leaves -1) use EUC_2D: src/distutil.c:90-91
    inst->wtype = (weight_type >= 0 && weight_type <= 5) ? weight_type : TSP_EUC_2D;
    inst->integer_cost = integer_cost ? 1 : 0;
    inst->wtype_public = inst->wtype;
    inst->h_xy.assign(xy, xy + 2 * (size_t)n);
    tsp::read_switches(&inst->sw);
    {   // integer coordinates of bounded span: switch to the exact integer-root variants (tsp_dist.hpp)
        bool all_int = true;
        double lox = xy[0], hix = xy[0], loy = xy[1], hiy = xy[1];
        for (int v = 0; v < n; ++v) {
            const double x = xy[2 * v], y = xy[2 * v + 1];
            if (!std::isfinite(x) || !std::isfinite(y)) return TSP_DEV_E_ARG;   // NaN / inf coordinates: nothing downstream is defined
            all_int = all_int && fabs(x) < 4.0e15 && fabs(y) < 4.0e15 && x == (double)(long long)x && y == (double)(long long)y;
            lox = x < lox ? x : lox; hix = x > hix ? x : hix; loy = y < loy ? y : loy; hiy = y > hiy ? y : hiy;
        }
        const double span = sqrt((hix - lox) * (hix - lox) + (hiy - loy) * (hiy - loy));
        inst->cost_bound = span + 2.0;
        {   // root filter margin: two raw roots (each within r * 2^-23 of r <= span, taken at 2^-22 for slack)
            // + the rounding the exact metric adds to each of the two distances (nint 0.5, ceil/ATT < 1)
            const bool nof = TSP_SW(inst, NO_FILTER, 0) == 1, nop = TSP_SW(inst, NO_PRUNE, 0) == 1;
            const bool sqrt_metric = inst->wtype == TSP_EUC_2D || inst->wtype == TSP_CEIL_2D || inst->wtype == TSP_ATT;
            inst->filter_margin = 1e300;   // "off": no pair is ever skipped
            if (sqrt_metric && !nof && span < 1e100) {
                const double rounding = (inst->integer_cost || inst->wtype == TSP_CEIL_2D) ? 2.0 : 0.0;
                inst->filter_margin = 2.0 * span * 0x1p-22 + rounding + 1e-6 + span * 0x1p-40;
            }
            // new-edge bound: nint() can shorten the new edge by at most 1/2 (ceil / ATT never shorten it);
            // the slack covers the rounding of s, T*T and the sums for coordinates of this magnitude
            inst->prune_margin = 1e300;
            if (sqrt_metric && !nop && !nof && span < 1e100)
                inst->prune_margin = ((inst->integer_cost && inst->wtype == TSP_EUC_2D) ? 0.5 : 0.0) + 1e-6 + span * 0x1p-36;
            // both new edges: nint() shortens each by at most 1/2; slack doubled so that '<' keeps ties.  The test
            // multiplies squared distances: only for spans whose fourth power is far from overflow.
            if (inst->prune_margin < 1e299 && span < 1e60)
                inst->sum_margin = ((inst->integer_cost && inst->wtype == TSP_EUC_2D) ? 1.0 : 0.0) + 2e-6 + span * 0x1p-34;
        }
        if (all_int && span < TSP_ICOORD_MAX_DIST && TSP_SW(inst, NO_ICOORD, 0) != 1) {
            if (inst->wtype == TSP_EUC_2D && inst->integer_cost) inst->wtype = tsp::WT_EUC_2D_ICOORD;
            else if (inst->wtype == TSP_ATT && inst->integer_cost) inst->wtype = tsp::WT_ATT_ICOORD;
            else if (inst->wtype == TSP_CEIL_2D) { inst->wtype = tsp::WT_CEIL_2D_ICOORD; }
        }
    }
    std::vector<double2> c((size_t)n);
    for (int v = 0; v < n; ++v) {
        if (inst->wtype == TSP_GEO) { c[v].x = geo_radians(xy[2 * v]); c[v].y = geo_radians(xy[2 * v + 1]); }
        else { c[v].x = xy[2 * v]; c[v].y = xy[2 * v + 1]; }
    }
    TSP_HIP_TRY(hipMalloc(&inst->d_coord, sizeof(double2) * (size_t)n));
    TSP_HIP_TRY(hipMemcpyAsync(inst->d_coord, c.data(), sizeof(double2) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (inst->sum_margin < 1e299) {
        // Sorted sweep: rank the nodes along a Hilbert curve; 64 consecutive ranks form a group whose bounding
        // box lets whole 64 x 64 blocks of pairs be decided by the new-edge bound at once (two_opt_grid.hip).
        const int ng = (n + 63) / 64, n_slots = (ng + 1) * 64;
        double lox = xy[0], hix = xy[0], loy = xy[1], hiy = xy[1];
        for (int v = 0; v < n; ++v) {
            lox = std::min(lox, xy[2 * v]); hix = std::max(hix, xy[2 * v]);
            loy = std::min(loy, xy[2 * v + 1]); hiy = std::max(hiy, xy[2 * v + 1]);
        }
        const double sx = hix > lox ? 65535.0 / (hix - lox) : 0.0, sy = hiy > loy ? 65535.0 / (hiy - loy) : 0.0;
        std::vector<std::pair<unsigned, int>> key((size_t)n);
        for (int v = 0; v < n; ++v)
            key[v] = {hilbert_rank16((unsigned)((xy[2 * v] - lox) * sx), (unsigned)((xy[2 * v + 1] - loy) * sy)), v};
        std::sort(key.begin(), key.end());
        std::vector<int> sperm((size_t)n_slots, -1);
        std::vector<double4> gbox((size_t)ng + 1);
        for (int g = 0; g <= ng; ++g) gbox[g] = make_double4(1e30, 1e30, 1e30, 1e30);   // {min x, max x, min y, max y}
        for (int k = 0; k < n; ++k) {
            const int v = key[k].second, g = k / 64;
            sperm[k] = v;
            double4 &b = gbox[g];
            if ((k & 63) == 0) b = make_double4(xy[2 * v], xy[2 * v], xy[2 * v + 1], xy[2 * v + 1]);
            b.x = std::min(b.x, xy[2 * v]); b.y = std::max(b.y, xy[2 * v]);
            b.z = std::min(b.z, xy[2 * v + 1]); b.w = std::max(b.w, xy[2 * v + 1]);
        }
        inst->ng = ng; inst->n_slots = n_slots; inst->h_gbox = gbox; inst->org_x = lox; inst->org_y = loy;
        inst->h_sinv.assign((size_t)n, 0);
        for (int k = 0; k < n; ++k) inst->h_sinv[sperm[k]] = k;
        TSP_HIP_TRY(hipMalloc(&inst->d_sperm, sizeof(int) * (size_t)n_slots));
        TSP_HIP_TRY(hipMalloc(&inst->d_gbox, sizeof(double4) * ((size_t)ng + 1)));
        TSP_HIP_TRY(hipMemcpyAsync(inst->d_sperm, sperm.data(), sizeof(int) * (size_t)n_slots, hipMemcpyHostToDevice, ctx->stream));
        TSP_HIP_TRY(hipMemcpyAsync(inst->d_gbox, gbox.data(), sizeof(double4) * ((size_t)ng + 1), hipMemcpyHostToDevice, ctx->stream));
        TSP_HIP_TRY(hipStreamSynchronize(ctx->stream));   // the staging vectors die with this scope
    }
    TSP_HIP_TRY(hipStreamSynchronize(ctx->stream));
    guard.i = nullptr;
    *out = inst;
    return TSP_OK;
}

void tsp_dev_inst_destroy(tsp_dev_inst *inst) {
    if (!inst) return;
    (void)hipSetDevice(inst->ctx->device);
    (void)hipStreamSynchronize(inst->ctx->stream);
    if (inst->scratch1) tsp_dev_tours_destroy(inst->scratch1);
    if (inst->scratch_b) tsp_dev_tours_destroy(inst->scratch_b);
    if (inst->ev0) { (void)hipEventDestroy(inst->ev0); (void)hipEventDestroy(inst->ev1); }
    (void)hipFree(inst->d_coord); (void)hipFree(inst->d_sperm); (void)hipFree(inst->d_gbox); (void)hipFree(inst->d_sxy); (void)hipFree(inst->cons_pool);
    (void)hipFree(inst->d_rcoord); (void)hipFree(inst->d_sinv); (void)hipFree(inst->io_pool);
    delete inst;
}

int tsp_dev_inst_size(const tsp_dev_inst *inst) { return inst ? inst->n : 0; }

int tsp_dev_inst_reload_switches(tsp_dev_inst *inst) {
    if (!inst) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    TSP_HIP_TRY(hipStreamSynchronize(inst->ctx->stream));
    // the cached single-tour / batch handles were laid out under the old switches
    if (inst->scratch1) { tsp_dev_tours_destroy(inst->scratch1); inst->scratch1 = nullptr; }
    if (inst->scratch_b) { tsp_dev_tours_destroy(inst->scratch_b); inst->scratch_b = nullptr; inst->scratch_b_count = 0; }
    tsp::read_switches(&inst->sw);
    return TSP_OK;
}

int tsp_dev_dist_pairs(tsp_dev_inst *inst, const int *i, const int *j, int count, double *out) {
    if (!inst || !i || !j || !out || count < 0) return TSP_DEV_E_ARG;
    if (count == 0) return TSP_OK;
    for (int k = 0; k < count; ++k)
        if (i[k] < 0 || i[k] >= inst->n || j[k] < 0 || j[k] >= inst->n) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    hipStream_t s = inst->ctx->stream;
    // one block of the instance's scratch: out | i | j
    char *pool = static_cast<char *>(tsp_io_pool(inst, (size_t)count * 16));
    if (!pool) return TSP_DEV_E_NOMEM;
    double *d_o = reinterpret_cast<double *>(pool);
    int *d_i = reinterpret_cast<int *>(pool + (size_t)count * 8), *d_j = d_i + count;
    TSP_HIP_TRY(hipMemcpyAsync(d_i, i, sizeof(int) * (size_t)count, hipMemcpyHostToDevice, s));
    TSP_HIP_TRY(hipMemcpyAsync(d_j, j, sizeof(int) * (size_t)count, hipMemcpyHostToDevice, s));
    TSP_DISPATCH_METRIC(inst->wtype, inst->integer_cost, {
        hipLaunchKernelGGL((k_dist_pairs<WTC, INTC>), dim3((count + 255) / 256), dim3(256), 0, s, inst->d_coord, d_i,
                           d_j, count, d_o);
    });
    TSP_HIP_TRY(hipMemcpyAsync(out, d_o, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

int tsp_dev_selftest_raw_sqrt(tsp_dev_ctx *ctx, const double *in, int count, double *out) {
    if (!ctx || !in || !out || count < 1) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<double> d_in, d_out;
    TSP_HIP_TRY(d_in.alloc((size_t)count));
    TSP_HIP_TRY(d_out.alloc((size_t)count));
    TSP_HIP_TRY(hipMemcpyAsync(d_in, in, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_raw_sqrt, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, d_in, count, d_out);
    TSP_HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    TSP_HIP_TRY(hipStreamSynchronize(ctx->stream));
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

int tsp_dev_perm_cost(tsp_dev_inst *inst, int B, const int *perm, int64_t perm_stride, double *cost) {
    if (!inst || !perm || !cost || B < 1 || perm_stride < inst->n) return TSP_DEV_E_ARG;
    const int n = inst->n;
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < n; ++k) {
            const int v = perm[(size_t)b * perm_stride + k];
            if (v < 0 || v >= n) return TSP_DEV_E_ARG;
        }
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    hipStream_t s = inst->ctx->stream;
    const size_t words = (size_t)(B - 1) * perm_stride + n;
    char *pool = static_cast<char *>(tsp_io_pool(inst, (size_t)B * 8 + words * 4));   // cost | permutations
    if (!pool) return TSP_DEV_E_NOMEM;
    double *d_c = reinterpret_cast<double *>(pool);
    int *d_p = reinterpret_cast<int *>(pool + (size_t)B * 8);
    TSP_HIP_TRY(hipMemcpyAsync(d_p, perm, sizeof(int) * words, hipMemcpyHostToDevice, s));
    int rc = tsp_perm_cost_device(inst, d_p, (long long)perm_stride, B, d_c, sizeof(double));
    if (rc) return rc;
    TSP_HIP_TRY(hipMemcpyAsync(cost, d_c, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, s));
    TSP_HIP_TRY(hipStreamSynchronize(s));
    TSP_HIP_TRY(hipGetLastError());
    return TSP_OK;
}

int tsp_dev_two_opt(tsp_dev_inst *inst, int mode, int engine, int B, int *succ, int succ_stride,
                    int64_t tour_stride, double *obj, double time_limit_s, tsp_two_opt_stats *stats) {
    if (!inst || !succ || !obj || B < 1 || succ_stride < 1) return TSP_DEV_E_ARG;
    if (mode != TSP_2OPT_FIRST && mode != TSP_2OPT_BEST) return TSP_DEV_E_ARG;
    if (engine < TSP_ENGINE_AUTO || engine > TSP_ENGINE_CLUSTER) return TSP_DEV_E_ARG;
    if (engine == TSP_ENGINE_LDS && !tsp_lds_fits(inst)) return TSP_DEV_E_ARG;
    const double t0 = wall_s();
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    hipStream_t s = inst->ctx->stream;
    bool owned = false;
    int rc = TSP_OK;
    tsp_dev_tours *t = tsp_scratch_tours(inst, B, &owned, &rc);
    if (rc) return rc;
    rc = tsp_dev_tours_upload(t, succ, succ_stride, tour_stride, obj);
    if (rc) { if (owned) tsp_dev_tours_destroy(t); return rc; }
    int cluster_C = 1;
    if (engine == TSP_ENGINE_CLUSTER) {
        if (!tsp_cluster_fits(t, mode)) { if (owned) tsp_dev_tours_destroy(t); return TSP_DEV_E_ARG; }
        cluster_C = tsp_cluster_size(t, mode);
    }
    if (engine == TSP_ENGINE_AUTO) {
        // Measured on MI355X (tools/cluster_time.py, tools/shard_time.py).  Results are identical.
        //  * CLUSTER: tours that fit in a CU's LDS and leave CUs idle (B tours on 256 CUs): C = #CUs / B workgroups per
        //    tour, whole descent in one launch.  A step costs one exchange through L2 instead of one or two kernel boundaries.
        //  * LDS (one workgroup per tour): first-improvement batches of more tours than an eighth of the CUs.
        //  * GRID: everything else (tours beyond LDS, tabu runs, large best-improvement batches).
        const int force = TSP_SW(inst, ENGINE, 0);
        const bool lds_ok = tsp_lds_fits(inst);
        const int C = tsp_cluster_fits(t, mode) ? tsp_cluster_size(t, mode) : 0;
        // "few tours" / "most of the chip idle" in units of this device's CUs: 1/32 of them (8 on the 256 CUs of an MI355X)
        const int few = std::max(1, inst->ctx->num_cus / 32);
        bool lds = lds_ok && mode == TSP_2OPT_FIRST && B >= few;
        // few tours: CLUSTER whatever the cluster size (measured on single tours from berlin52 to rand10000, tools/cluster_time.py:
        // 1.4-2.6 x faster than GRID in both rules, level with LDS at n = 52, 1.6 x faster at n = 299); eight or more tours: one
        // workgroup per tour (LDS, first improvement) unless that would leave most of the chip idle (C >= 8), best-improvement
        // batches on CLUSTER only with the sorted scan
        // (best-improvement batches: the cluster's sorted scan beats the GRID engine's lock-step launches about 2 x at every batch
        // size that fits the chip, one workgroup per tour included -- tools/best_batch.py)
        // (first-improvement batches the LDS engine cannot hold -- n > ~8000 -- go to the cluster at any size: 64 tours of
        // rand10000 on 4 workgroups each 117 ms, GRID 240 ms -- tools/first_batch_big.py)
        // (first-improvement batches: from four workgroups per tour on the cluster is ahead of one workgroup per tour -- 64 random
        // tours of rand5000: LDS 215 ms, 4 workgroups each 187 ms; 128 tours on 2 each 232 ms against 216 -- tools/pop_time.py)
        bool cluster = C >= 1 && (B < few || (mode == TSP_2OPT_BEST ? tsp_cluster_sorted(t, mode) : (C >= std::max(1, few / 2) || !lds_ok)));
        if (cluster && force != 3 && inst->ctx->cl_skip > 0) { --inst->ctx->cl_skip; cluster = false; }   // backing off after a give-up
        if (force == 1) { lds = false; cluster = false; }
        if (force == 2 && lds_ok) { lds = true; cluster = false; }
        if (force == 3 && C >= 1) cluster = true;
        engine = cluster ? TSP_ENGINE_CLUSTER : (lds ? TSP_ENGINE_LDS : TSP_ENGINE_GRID);
        cluster_C = std::max(1, C);
    }
    if (!inst->ev0) { TSP_HIP_TRY(hipEventCreate(&inst->ev0)); TSP_HIP_TRY(hipEventCreate(&inst->ev1)); }
    hipEvent_t e0 = inst->ev0, e1 = inst->ev1;
    TSP_HIP_TRY(hipEventRecord(e0, s));
    int done = 0;
    int status;
    if (engine == TSP_ENGINE_CLUSTER) {
        int fell = 0;
        status = tsp_cluster_run(t, mode, cluster_C, -1, time_limit_s, &done, &fell);
        if (fell) {   // a workgroup was not resident (the device is shared): same descent, one launch per step
            rc = tsp_dev_tours_reset(t);
            if (rc) { if (owned) tsp_dev_tours_destroy(t); return rc; }
            status = tsp_grid_run(t, mode, nullptr, 0, 0, -1, time_limit_s, 1, &done);
        }
    } else if (engine == TSP_ENGINE_LDS) {
        status = tsp_lds_run(t, mode, time_limit_s, &done);
    } else {
        status = tsp_grid_run(t, mode, nullptr, 0, 0, -1, time_limit_s, 1, &done);
    }
    TSP_HIP_TRY(hipEventRecord(e1, s));
    TSP_HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    TSP_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (status < 0) { if (owned) tsp_dev_tours_destroy(t); return status; }
    std::vector<double> new_obj((size_t)B);
    rc = tsp_dev_tours_download(t, succ, succ_stride, tour_stride, new_obj.data(), stats);
    if (rc == TSP_OK) {
        for (int b = 0; b < B; ++b) {
            // BEST stopped by the time limit has no recomputed cost yet: do what tabusearch.c:168-172 does
            obj[b] = new_obj[b];
        }
        if (stats) for (int b = 0; b < B; ++b) { stats[b].seconds = wall_s() - t0; stats[b].device_ms = ms; }
    }
    if (owned) tsp_dev_tours_destroy(t);
    return rc ? rc : status;
}

static int run_engine_untimed(tsp_dev_tours *t, int mode, int engine, int64_t max_steps, double time_limit_s, int *all_done) {
    if (engine == TSP_ENGINE_CLUSTER || engine == TSP_ENGINE_AUTO) {
        const int few = std::max(1, t->inst->ctx->num_cus / 32);   // as in tsp_dev_two_opt
        bool want = engine == TSP_ENGINE_CLUSTER ||
                    (tsp_cluster_fits(t, mode) && (t->B < few || (mode == TSP_2OPT_BEST ? tsp_cluster_sorted(t, mode) : (tsp_cluster_size(t, mode) >= std::max(1, few / 2) || !tsp_lds_fits(t->inst)))));
        if (want && engine == TSP_ENGINE_AUTO && t->inst->ctx->cl_skip > 0) { --t->inst->ctx->cl_skip; want = false; }   // backing off after a give-up
        if (want) {
            if (!tsp_cluster_fits(t, mode)) return TSP_DEV_E_ARG;
            int fell = 0;
            const int status = tsp_cluster_run(t, mode, tsp_cluster_size(t, mode), max_steps, time_limit_s, all_done, &fell);
            if (!fell) return status;
            if (engine == TSP_ENGINE_CLUSTER) return status;   // asked for explicitly: report, do not substitute
        }
        engine = TSP_ENGINE_GRID;
    }
    if (engine == TSP_ENGINE_LDS) {
        if (!tsp_lds_fits(t->inst) || max_steps >= 0) return TSP_DEV_E_ARG;
        return tsp_lds_run(t, mode, time_limit_s, all_done);
    }
    if (engine != TSP_ENGINE_GRID) return TSP_DEV_E_ARG;
    return tsp_grid_run(t, mode, nullptr, 0, 0, max_steps, time_limit_s, 1, all_done);
}

int tsp_dev_tours_run_engine(tsp_dev_tours *t, int mode, int engine, int64_t max_steps, double time_limit_s, int *all_done) {
    if (!t || (mode != TSP_2OPT_FIRST && mode != TSP_2OPT_BEST)) return TSP_DEV_E_ARG;
    tsp_dev_inst *inst = t->inst;
    TSP_HIP_TRY(hipSetDevice(inst->ctx->device));
    if (all_done) *all_done = 0;
    // device time of the run, HIP events on the engine's stream (reported as stats.device_ms by the next download)
    if (!inst->ev0) { TSP_HIP_TRY(hipEventCreate(&inst->ev0)); TSP_HIP_TRY(hipEventCreate(&inst->ev1)); }
    TSP_HIP_TRY(hipEventRecord(inst->ev0, inst->ctx->stream));
    const int status = run_engine_untimed(t, mode, engine, max_steps, time_limit_s, all_done);
    if (status < 0) return status;
    TSP_HIP_TRY(hipEventRecord(inst->ev1, inst->ctx->stream));
    TSP_HIP_TRY(hipEventSynchronize(inst->ev1));
    float ms = 0.f;
    TSP_HIP_TRY(hipEventElapsedTime(&ms, inst->ev0, inst->ev1));
    t->device_ms = ms;
    return status;
}

// ---- drivers on resident tours: tabu() and HEU_VNS keep the tour (and the stamps) on the device between their steps ----

int tsp_dev_tours_two_opt(tsp_dev_tours *t, int mode, int engine, double time_limit_s, double *obj) {
    if (!t || (mode != TSP_2OPT_FIRST && mode != TSP_2OPT_BEST)) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    int rc = tsp_grid_rearm(t, mode);
    if (rc) return rc;
    int done = 0;
    t->h_state_fresh = false;
    const int status = tsp_dev_tours_run_engine(t, mode, engine, -1, time_limit_s, &done);
    if (status < 0) return status;
    if (!t->h_state_fresh) {   // (the CLUSTER engine's last poll has brought the control blocks back already)
        hipStream_t s = t->inst->ctx->stream;
        TSP_HIP_TRY(hipMemcpyAsync(t->h_state, t->d_state, sizeof(tsp::TourState) * (size_t)t->B, hipMemcpyDeviceToHost, s));
        TSP_HIP_TRY(hipStreamSynchronize(s));
    }
    t->h_state_fresh = false;
    if (obj) for (int b = 0; b < t->B; ++b) obj[b] = t->h_state[b].obj;
    return status;
}

int tsp_dev_tours_two_opt_tabu(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter, int tenure, double time_limit_s, double *obj) {
    if (!t) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    return tsp_grid_resident_tabu(t, tabu, iter, tenure, time_limit_s, obj);
}

int tsp_dev_tours_tabu_kick(tsp_dev_tours *t, tsp_dev_tabu *tabu, int a, int b, int iter, int tenure, int *accepted) {
    if (!t) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    return tsp_grid_tabu_kick(t, tabu, a, b, iter, tenure, accepted);
}

int tsp_dev_tours_tabu_iteration(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter, int tenure, double time_limit_s, int a, int b,
                                 double *best_obj, double *obj, int *improved, int *accepted) {
    if (!t) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    return tsp_grid_tabu_iteration(t, tabu, iter, tenure, time_limit_s, a, b, best_obj, obj, improved, accepted);
}

int tsp_dev_tours_tabu_iterations(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter0, int count, const int *tenure, const int *ab,
                                  double time_limit_s, double *best_obj, double *obj, int *improved, int *completed, int *last_accepted) {
    if (!t) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    return tsp_grid_tabu_iterations(t, tabu, iter0, count, tenure, 0, ab, time_limit_s, best_obj, obj, improved, nullptr, completed, last_accepted);
}

int tsp_dev_tours_tabu_iterations_ex(tsp_dev_tours *t, tsp_dev_tabu *tabu, int iter0, int count, const int *tenure, int pairs, const int *ab,
                                     double time_limit_s, double *best_obj, double *obj, int *improved, int *trials, int *completed,
                                     int *last_accepted) {
    if (!t || pairs < count) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    return tsp_grid_tabu_iterations(t, tabu, iter0, count, tenure, pairs, ab, time_limit_s, best_obj, obj, improved, trials, completed, last_accepted);
}

int tsp_dev_tours_vns_kick(tsp_dev_tours *t, int p1, int p2, int p3, double *obj) {
    if (!t) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    return tsp_grid_vns_kick(t, p1, p2, p3, obj);
}

/* Page-locks a caller's host array so that the uploads / downloads of a loop that passes the same array again and again
 * (alg_2opt_tabu's skip_edge: 4 n^2 / 2 bytes each way per call) run at PCIe speed. */
int tsp_dev_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return TSP_OK;
}
int tsp_dev_host_unregister(void *p) {
    if (!p) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipHostUnregister(p));
    return TSP_OK;
}

int tsp_dev_tours_snapshot(tsp_dev_tours *t) {
    if (!t) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    return tsp_grid_snapshot(t, false);
}

int tsp_dev_tours_restore(tsp_dev_tours *t) {
    if (!t) return TSP_DEV_E_ARG;
    TSP_HIP_TRY(hipSetDevice(t->inst->ctx->device));
    return tsp_grid_snapshot(t, true);
}

}  // extern "C"
