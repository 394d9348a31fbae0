// comm.hip -- the multi-start epilogue across GPUs, reachable from C: one RCCL all-reduce(min) of the packed
// (cost << 24 | start id) per rank and one broadcast of the winner's successor list from the rank that owns it
// (the generalisation of HEU_Grasp_iter's "keep the best start", src/heuristics.c:510-544).
//
// The path shards across tours only, so these two latency-bound collectives (8 bytes, 4n bytes) are the only traffic
// over xGMI (two reductions when the costs are not integers: min of the double, then min of the start id among its holders).
// librccl is opened lazily (dlopen) the first time a communicator is asked for: a single-GPU caller never pays for loading it,
// and libtsp_hip.so has neither a link-time nor a build-time dependency on it (<rccl/rccl.h> is used when present, see
// below).  Every collective waits for its peers with a timeout (comm_wait).  Two ways to form the communicator:
//   tsp_dev_comm_init_rank   one process per GPU (torchrun / mpirun style): rank 0 calls tsp_dev_comm_unique_id and hands
//                            the 128 bytes to the other ranks by any side channel (libtsp_host.so: a file);
//   tsp_dev_comm_init_all    one process, several devices (ncclCommInitAll); the collectives of all its communicators are
//                            then issued together by the *_group entry points (ncclGroupStart / ncclGroupEnd).
#include "tsp_internal.hpp"

#include <dlfcn.h>
#include <time.h>
#include <unistd.h>

// RCCL is reached through dlopen / dlsym only, so its header is needed for nothing but a handful of types and enumerators.
// A ROCm install without the rccl development package still builds this library (every tsp_dev_comm_* entry then works
// as long as librccl.so.1 itself can be opened at run time); where the header exists it is used and checks the local values.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(ncclInt32 == 2 && ncclInt64 == 4 && ncclFloat64 == 8 && ncclMin == 3 && ncclSuccess == 0,
              "RCCL's enumerators are not the ones this file declares for header-less builds");
#else
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt32 = 2, ncclInt64 = 4, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclMin = 3 } ncclRedOp_t;
}
#endif

#include <mutex>

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
thread_local char g_comm_error[384] = "";

void rccl_open() {
    // the soname first: a process that already holds an RCCL (PyTorch-ROCm ships one) gets that very copy back
    const char *names[] = {getenv("TSP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        if (!nm || !*nm) continue;
        g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) return;
    bool all = true;
    auto sym = [&](const char *n) { void *p = dlsym(g_rccl.handle, n); all = all && p; return p; };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(sym("ncclCommInitAll"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(sym("ncclBroadcast"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    g_rccl.GetVersion = reinterpret_cast<decltype(g_rccl.GetVersion)>(sym("ncclGetVersion"));
    g_rccl.ok = all;
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(g_rccl.handle, "ncclCommAbort"));   // optional
}

const Rccl *rccl() {
    std::call_once(g_rccl_once, rccl_open);
    if (!g_rccl.ok) {
        snprintf(g_comm_error, sizeof g_comm_error, "librccl could not be opened (%s): multi-GPU multi-start needs RCCL",
                 g_rccl.handle ? "a symbol is missing" : "librccl.so.1 not found; set TSP_RCCL_LIB");
        return nullptr;
    }
    return &g_rccl;
}

}  // namespace

// RCCL prints a version banner on stdout when a communicator is formed.  The reference's CLI protocol is a bare number on
// stdout (src/solver.c:291-292), so while a communicator is being formed stdout is pointed at stderr.
struct StdoutToStderr {
    int saved = -1;
    StdoutToStderr() {
        fflush(stdout);
        saved = dup(1);
        if (saved >= 0) (void)dup2(2, 1);
    }
    ~StdoutToStderr() {
        fflush(stdout);
        if (saved >= 0) { (void)dup2(saved, 1); close(saved); }
    }
};

struct tsp_dev_comm {
    tsp_dev_ctx *ctx = nullptr;     // must outlive the communicator; device and stream are kept by value for the teardown
    int device = 0;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool broken = false;            // a collective timed out or failed: the communicator is aborted, not destroyed
    long long *d_word = nullptr;    // the packed best (or, as its bit pattern, the double cost) of this rank, reduced in place
    int *d_tour = nullptr;          // broadcast buffer, grown on demand
    size_t tour_cap = 0;
};

#define TSP_NCCL_TRY(expr)                                                                              \
    do {                                                                                                \
        ncclResult_t r__ = (expr);                                                                      \
        if (r__ != ncclSuccess) {                                                                       \
            snprintf(g_comm_error, sizeof g_comm_error, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,   \
                     R->GetErrorString(r__));                                                           \
            return TSP_DEV_E_COMM;                                                                      \
        }                                                                                               \
    } while (0)

// Inside ncclGroupStart / ncclGroupEnd: remember the first failure and keep going, so that the group is always closed (a
// return from between the two would leave this thread's group open and every later RCCL call silently queued).
#define TSP_NCCL_NOTE(rcvar, expr)                                                                      \
    do {                                                                                                \
        ncclResult_t r__ = (expr);                                                                      \
        if (r__ != ncclSuccess && (rcvar) == TSP_OK) {                                                  \
            snprintf(g_comm_error, sizeof g_comm_error, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,   \
                     R->GetErrorString(r__));                                                           \
            (rcvar) = TSP_DEV_E_COMM;                                                                   \
        }                                                                                               \
    } while (0)

namespace {
int comm_alloc(tsp_dev_comm *c) {
    TSP_HIP_TRY(hipSetDevice(c->device));
    TSP_HIP_TRY(hipMalloc(&c->d_word, sizeof(long long)));
    return TSP_OK;
}
int comm_tour_buf(tsp_dev_comm *c, size_t n) {
    if (n <= c->tour_cap) return TSP_OK;
    TSP_HIP_TRY(hipSetDevice(c->device));
    TSP_HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_tour); c->d_tour = nullptr; c->tour_cap = 0;
    TSP_HIP_TRY(hipMalloc(&c->d_tour, sizeof(int) * n));
    c->tour_cap = n;
    return TSP_OK;
}

double comm_timeout_s() {
    static const double t = [] {
        const char *e = getenv("TSP_COMM_TIMEOUT_S");
        const double v = e && *e ? atof(e) : 300.0;
        return v > 0 ? v : 300.0;
    }();
    return t;
}

// Wait for the collective queued on the communicator's stream, but not for ever: RCCL has no timeout of its own, and a peer
// that died before it entered the collective would leave this rank in hipStreamSynchronize until somebody kills it.  After
// TSP_COMM_TIMEOUT_S (300) seconds the communicator is aborted and the call fails with TSP_DEV_E_COMM, so that the caller
// exits non-zero like the peer did.
int comm_wait(tsp_dev_comm *c, const char *what) {
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    long spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return TSP_OK;
        if (q != hipErrorNotReady) { TSP_HIP_TRY(q); }
        if (++spins > 2000) { timespec nap = {0, 200 * 1000}; nanosleep(&nap, nullptr); }   // the usual wait is tens of microseconds
        if ((spins & 255) == 0) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double el = double(t1.tv_sec - t0.tv_sec) + 1e-9 * double(t1.tv_nsec - t0.tv_nsec);
            if (el > comm_timeout_s()) {
                snprintf(g_comm_error, sizeof g_comm_error, "rank %d of %d: %s did not complete within %.0f s (a peer never entered "
                         "it?); communicator aborted", c->rank, c->world, what, comm_timeout_s());
                if (g_rccl.ok && g_rccl.CommAbort && c->comm) { (void)g_rccl.CommAbort(c->comm); c->comm = nullptr; }
                c->broken = true;
                return TSP_DEV_E_COMM;
            }
        }
    }
}

// all-reduce(min) of one 8-byte word per rank (int64 or double); every rank receives the minimum
int allreduce_word(tsp_dev_comm *c, const void *local, void *best, ncclDataType_t type) {
    if (!c || !best || c->broken) return TSP_DEV_E_ARG;
    const Rccl *R = rccl();
    if (!R) return TSP_DEV_E_COMM;
    TSP_HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    long long w;
    memcpy(&w, local, sizeof w);
    TSP_HIP_TRY(hipMemcpyAsync(c->d_word, &w, sizeof w, hipMemcpyHostToDevice, s));
    TSP_NCCL_TRY(R->AllReduce(c->d_word, c->d_word, 1, type, ncclMin, c->comm, s));
    TSP_HIP_TRY(hipMemcpyAsync(&w, c->d_word, sizeof w, hipMemcpyDeviceToHost, s));
    int rc = comm_wait(c, "the all-reduce(min)");
    if (rc) return rc;
    memcpy(best, &w, sizeof w);
    return TSP_OK;
}

int allreduce_word_group(tsp_dev_comm *const *cs, int ndev, const void *local, void *best, ncclDataType_t type) {
    if (!cs || !local || !best || ndev < 1 || ndev > 64) return TSP_DEV_E_ARG;
    const Rccl *R = rccl();
    if (!R) return TSP_DEV_E_COMM;
    for (int k = 0; k < ndev; ++k) {
        if (!cs[k] || cs[k]->world != ndev || cs[k]->broken) return TSP_DEV_E_ARG;
        TSP_HIP_TRY(hipSetDevice(cs[k]->device));
        long long w;
        memcpy(&w, static_cast<const char *>(local) + 8 * (size_t)k, sizeof w);
        TSP_HIP_TRY(hipMemcpy(cs[k]->d_word, &w, sizeof w, hipMemcpyHostToDevice));
    }
    int rc = TSP_OK;
    TSP_NCCL_TRY(R->GroupStart());
    for (int k = 0; k < ndev; ++k)
        TSP_NCCL_NOTE(rc, R->AllReduce(cs[k]->d_word, cs[k]->d_word, 1, type, ncclMin, cs[k]->comm, cs[k]->stream));
    TSP_NCCL_NOTE(rc, R->GroupEnd());
    if (rc) return rc;
    for (int k = 0; k < ndev; ++k) {
        TSP_HIP_TRY(hipSetDevice(cs[k]->device));
        rc = comm_wait(cs[k], "the grouped all-reduce(min)");
        if (rc) return rc;
        long long w = 0;
        TSP_HIP_TRY(hipMemcpy(&w, cs[k]->d_word, sizeof w, hipMemcpyDeviceToHost));
        memcpy(static_cast<char *>(best) + 8 * (size_t)k, &w, sizeof w);
    }
    return TSP_OK;
}
}  // namespace

extern "C" {

const char *tsp_dev_comm_last_error(void) { return g_comm_error; }

int tsp_dev_comm_available(void) { return rccl() ? 1 : 0; }

int tsp_dev_comm_unique_id(char *id) {
    if (!id) return TSP_DEV_E_ARG;
    const Rccl *R = rccl();
    if (!R) return TSP_DEV_E_COMM;
    static_assert(sizeof(ncclUniqueId) == TSP_COMM_ID_BYTES, "TSP_COMM_ID_BYTES must be RCCL's NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId u;
    TSP_NCCL_TRY(R->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return TSP_OK;
}

int tsp_dev_comm_init_rank(tsp_dev_ctx *ctx, int world, int rank, const char *id, tsp_dev_comm **out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return TSP_DEV_E_ARG;
    const Rccl *R = rccl();
    if (!R) return TSP_DEV_E_COMM;
    TSP_HIP_TRY(hipSetDevice(ctx->device));
    tsp_dev_comm *c = new tsp_dev_comm();
    c->ctx = ctx; c->device = ctx->device; c->stream = ctx->stream; c->rank = rank; c->world = world;
    struct Guard { tsp_dev_comm *c; ~Guard() { if (c) tsp_dev_comm_destroy(c); } } guard{c};
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    {
        StdoutToStderr quiet;
        TSP_NCCL_TRY(R->CommInitRank(&c->comm, world, u, rank));
    }
    int rc = comm_alloc(c);
    if (rc) return rc;
    guard.c = nullptr;
    *out = c;
    return TSP_OK;
}

int tsp_dev_comm_init_all(tsp_dev_ctx *const *ctxs, int ndev, tsp_dev_comm **out) {
    if (!ctxs || !out || ndev < 1 || ndev > 64) return TSP_DEV_E_ARG;
    const Rccl *R = rccl();
    if (!R) return TSP_DEV_E_COMM;
    int devs[64];
    ncclComm_t comms[64];
    for (int k = 0; k < ndev; ++k) {
        if (!ctxs[k]) return TSP_DEV_E_ARG;
        devs[k] = ctxs[k]->device;
        for (int q = 0; q < k; ++q) if (devs[q] == devs[k]) return TSP_DEV_E_ARG;   // one rank per device
        out[k] = nullptr;
    }
    {
        StdoutToStderr quiet;
        TSP_NCCL_TRY(R->CommInitAll(comms, ndev, devs));
    }
    for (int k = 0; k < ndev; ++k) {
        tsp_dev_comm *c = new tsp_dev_comm();
        c->ctx = ctxs[k]; c->device = ctxs[k]->device; c->stream = ctxs[k]->stream; c->comm = comms[k]; c->rank = k; c->world = ndev;
        out[k] = c;
    }
    for (int k = 0; k < ndev; ++k) {
        int rc = comm_alloc(out[k]);
        if (rc) { for (int q = 0; q < ndev; ++q) { tsp_dev_comm_destroy(out[q]); out[q] = nullptr; } return rc; }
    }
    return TSP_OK;
}

// The context the communicator was formed on must still be open (its stream carries the collectives): destroy the
// communicator first, then tsp_dev_close.  Nothing of the context is dereferenced here -- device and stream were copied.
void tsp_dev_comm_destroy(tsp_dev_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (!c->broken) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.ok) (void)g_rccl.CommDestroy(c->comm);
    (void)hipFree(c->d_word); (void)hipFree(c->d_tour);
    delete c;
}

int tsp_dev_comm_info(const tsp_dev_comm *c, int *rank, int *world, int *rccl_version) {
    if (!c) return TSP_DEV_E_ARG;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (rccl_version) { *rccl_version = 0; if (g_rccl.ok) (void)g_rccl.GetVersion(rccl_version); }
    return TSP_OK;
}

int tsp_dev_multistart_pack(double cost, int start_id, int64_t *packed) {
    if (!packed) return TSP_DEV_E_ARG;
    // the integer minimum must order like (cost, start id): costs that are not non-negative integers below 2^39 cannot be packed
    if (!(cost >= 0.0) || cost >= 549755813888.0 || cost != (double)(long long)cost || start_id < 0 || start_id >= (1 << 24))
        return TSP_DEV_E_ARG;
    *packed = ((int64_t)cost << 24) | (int64_t)start_id;
    return TSP_OK;
}

int tsp_dev_multistart_allreduce(tsp_dev_comm *c, int64_t packed_local, int64_t *packed_best) {
    return allreduce_word(c, &packed_local, packed_best, ncclInt64);
}

int tsp_dev_multistart_allreduce_f64(tsp_dev_comm *c, double cost_local, double *cost_best) {
    return allreduce_word(c, &cost_local, cost_best, ncclFloat64);
}

int tsp_dev_multistart_bcast_tour(tsp_dev_comm *c, int root, int *succ, int succ_stride, int n) {
    if (!c || !succ || n < 1 || succ_stride < 1 || root < 0 || root >= c->world || c->broken) return TSP_DEV_E_ARG;
    const Rccl *R = rccl();
    if (!R) return TSP_DEV_E_COMM;
    int rc = comm_tour_buf(c, (size_t)n);
    if (rc) return rc;
    TSP_HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    std::vector<int> h((size_t)n);
    if (c->rank == root) {
        for (int v = 0; v < n; ++v) h[v] = succ[(size_t)v * succ_stride];
        TSP_HIP_TRY(hipMemcpyAsync(c->d_tour, h.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
    }
    TSP_NCCL_TRY(R->Broadcast(c->d_tour, c->d_tour, (size_t)n, ncclInt32, root, c->comm, s));
    TSP_HIP_TRY(hipMemcpyAsync(h.data(), c->d_tour, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
    rc = comm_wait(c, "the broadcast of the winner's tour");
    if (rc) return rc;
    if (c->rank != root)
        for (int v = 0; v < n; ++v) succ[(size_t)v * succ_stride] = h[v];
    return TSP_OK;
}

int tsp_dev_multistart_allreduce_group(tsp_dev_comm *const *cs, int ndev, const int64_t *packed_local, int64_t *packed_best) {
    return allreduce_word_group(cs, ndev, packed_local, packed_best, ncclInt64);
}

int tsp_dev_multistart_allreduce_f64_group(tsp_dev_comm *const *cs, int ndev, const double *cost_local, double *cost_best) {
    return allreduce_word_group(cs, ndev, cost_local, cost_best, ncclFloat64);
}

int tsp_dev_multistart_bcast_tour_group(tsp_dev_comm *const *cs, int ndev, int root, const int *succ_root, int succ_stride,
                                        int n, int read_back_rank, int *succ_out) {
    if (!cs || !succ_root || !succ_out || ndev < 1 || ndev > 64 || n < 1 || succ_stride < 1 || root < 0 || root >= ndev ||
        read_back_rank < 0 || read_back_rank >= ndev)
        return TSP_DEV_E_ARG;
    const Rccl *R = rccl();
    if (!R) return TSP_DEV_E_COMM;
    std::vector<int> h((size_t)n);
    for (int v = 0; v < n; ++v) h[v] = succ_root[(size_t)v * succ_stride];
    for (int k = 0; k < ndev; ++k) {
        if (!cs[k] || cs[k]->world != ndev || cs[k]->broken) return TSP_DEV_E_ARG;
        int rc = comm_tour_buf(cs[k], (size_t)n);
        if (rc) return rc;
    }
    TSP_HIP_TRY(hipSetDevice(cs[root]->device));
    TSP_HIP_TRY(hipMemcpy(cs[root]->d_tour, h.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    int rc = TSP_OK;
    TSP_NCCL_TRY(R->GroupStart());
    for (int k = 0; k < ndev; ++k)
        TSP_NCCL_NOTE(rc, R->Broadcast(cs[k]->d_tour, cs[k]->d_tour, (size_t)n, ncclInt32, root, cs[k]->comm, cs[k]->stream));
    TSP_NCCL_NOTE(rc, R->GroupEnd());
    if (rc) return rc;
    for (int k = 0; k < ndev; ++k) {
        TSP_HIP_TRY(hipSetDevice(cs[k]->device));
        rc = comm_wait(cs[k], "the grouped broadcast of the winner's tour");
        if (rc) return rc;
    }
    TSP_HIP_TRY(hipSetDevice(cs[read_back_rank]->device));
    TSP_HIP_TRY(hipMemcpy(h.data(), cs[read_back_rank]->d_tour, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    for (int v = 0; v < n; ++v) succ_out[v] = h[v];
    return TSP_OK;
}

}  // extern "C"
