"""Multi-start across the GPUs of one node: shard the starts, no data-path collective, one
all-reduce(min) of a packed (cost, start id) over RCCL (torch.distributed backend "nccl"; "gloo"
in the CPU tests), then the winner's tour is broadcast from the rank that owns it.

This is the generalisation of HEU_Grasp_iter's loop (src/heuristics.c:510-544: random start, grasp(), keep the
best) that BASELINE configs[3] and [4] define: every start is refined by alg_2opt, start k runs on rank
k % world, and the reported winner is the one a serial loop keeping the first strictly better start (:534)
would report.  The packing is the one SURVEY.md section 5 proposes: cost in the high bits, start id in the low 24,
so that the integer minimum is (lowest cost, then lowest start id).

`run_sharded` is the launcher: bench.py drives it with the device engine (one process per GPU), the CPU tests
drive the same function over gloo with the engine replaced by the golden table.
"""
import time

import numpy as np

ID_BITS = 24
ID_MASK = (1 << ID_BITS) - 1


def shard_starts(num_starts, rank, world):
    """Global start ids owned by `rank`: k with k % world == rank (round-robin keeps ragged counts balanced)."""
    return list(range(rank, num_starts, world))


def owner_of(start_id, world):
    return start_id % world


class UnpackableCost(ValueError):
    """The all-reduce(min) carries (cost << 24 | start id): only non-negative integer costs below 2^39 fit (the reference's
    default integer_cost = 1; not --fcost, not GEO's tolerance tier)."""


def pack(cost, start_id):
    """Same rule as tsp_dev_multistart_pack of the C ABI.  Raises UnpackableCost -- callers that sit between two
    collectives use try_pack and agree on the failure through the reduction itself."""
    ok = cost == cost and 0 <= cost < float(1 << 39)      # not NaN, in range (before int(): int(nan) raises)
    c = int(cost) if ok else -1
    if not (ok and c == cost and 0 <= start_id <= ID_MASK):
        raise UnpackableCost("cost %r / start %r" % (cost, start_id))
    return (c << ID_BITS) | start_id


PACK_ERROR = -1      # smaller than every packed value: a rank that could not pack wins the MIN and every rank sees it


def try_pack(cost, start_id):
    try:
        return pack(cost, start_id)
    except UnpackableCost:
        return PACK_ERROR


def unpack(packed):
    return packed >> ID_BITS, packed & ID_MASK


NO_RESULT = (1 << 62)


def local_best(costs, start_ids):
    """Packed minimum over this rank's starts (NO_RESULT for an empty shard; PACK_ERROR if a cost cannot be packed)."""
    best = NO_RESULT
    for c, k in zip(costs, start_ids):
        best = min(best, try_pack(c, k))
    return best


def allreduce_best(packed, device=None):
    """One all-reduce(min) of an int64 scalar; returns (cost, start id).  Needs an initialised process group."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([packed], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return unpack(int(t.item()))


def broadcast_winner(succ, winner_start, world, device=None):
    """Broadcast the winner's successor list (int32 tensor of n) from the rank that owns it."""
    import torch.distributed as dist
    dist.broadcast(succ, src=owner_of(winner_start, world))
    return succ


class _TorchCollectives:
    """The launcher's three collectives over torch.distributed (backend "nccl" = RCCL on the GPUs, "gloo" in the CPU tests)."""

    def __init__(self, device=None):
        self.device = device

    def allreduce_min(self, packed):
        import torch
        import torch.distributed as dist
        t = torch.tensor([packed], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    def allreduce_min_f64(self, cost):
        import torch
        import torch.distributed as dist
        t = torch.tensor([cost], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item())

    def bcast_tour(self, root, tour):
        import torch
        import torch.distributed as dist
        buf = torch.from_numpy(tour).to(self.device) if self.device is not None else torch.from_numpy(tour)
        dist.broadcast(buf, src=root)
        tour[:] = buf.cpu().numpy()
        return tour


class _NoCollectives:
    """world == 1 without a communicator: the reductions of one value are that value."""

    def allreduce_min(self, packed):
        return packed

    def allreduce_min_f64(self, cost):
        return cost

    def bcast_tour(self, root, tour):
        return tour


def run_sharded(refine, num_starts, n, rank=0, world=1, device=None, comm=None, integer_costs=True):
    """The multi-start launcher (the Python twin of the C host's sharded_job / tsp_host_multistart_epilogue).

    comm: an engine.Comm (the C ABI's RCCL communicator, tsp_dev_multistart_allreduce / _allreduce_f64 / _bcast_tour) -- the
    collectives then run through libtsp_hip.so exactly as the C host runs them; None: torch.distributed (world > 1) or nothing.

    refine(ids) -> (costs, tours): for the global start ids `ids` (this rank's shard, ascending) the true tour cost of
    every refined start and the refined tours as an int32 array [len(ids), n] of successor lists.
    Returns {"cost", "start", "tour" (np.int32 [n], the winner's, on EVERY rank), "seconds" (this rank's refine time),
    "local_starts"}.

    integer_costs (the reference's default): ONE all_reduce(MIN) of the packed (cost << 24 | start) and ONE broadcast of n
    int32 from the winner's owner.  integer_costs=False (--fcost, src/utility.c:285): TWO reductions -- MIN of the double cost,
    then MIN of the start id among the ranks that hold it (ties -> lowest start: the strict `<` of src/heuristics.c:534 in
    stream order) -- and the broadcast.  A rank that cannot contribute (refine raised, a cost that cannot be packed) does not
    leave before the collective: it contributes PACK_ERROR / -inf, which wins the minimum, so that EVERY rank raises.
    """
    ids = shard_starts(num_starts, rank, world)
    coll = comm if comm is not None else (_TorchCollectives(device) if world > 1 else _NoCollectives())
    t0 = time.perf_counter()
    failure = None
    try:
        costs, tours = refine(ids)
        tours = np.ascontiguousarray(tours, dtype=np.int32).reshape(len(ids), n)
    except Exception as e:      # noqa: BLE001 -- carried to every rank through the reduction, re-raised below
        failure, costs, tours = e, [], np.zeros((0, n), dtype=np.int32)
    seconds = time.perf_counter() - t0
    if integer_costs:
        packed = PACK_ERROR if failure is not None else local_best(costs, ids)
        win = coll.allreduce_min(packed)
        if win == PACK_ERROR:
            if failure is not None:
                raise failure
            raise UnpackableCost("a rank failed its shard or holds a cost the packed all-reduce cannot carry")
        assert win != NO_RESULT
        cost, start = unpack(win)
    else:
        mine, mine_k = float("inf"), NO_RESULT
        for c, k in zip(costs, ids):
            if c != c:
                failure = failure or ValueError("cost of start %d is not a number" % k)
            elif c < mine:
                mine, mine_k = float(c), k
        cmin = coll.allreduce_min_f64(float("-inf") if failure is not None else mine)
        if cmin == float("-inf"):
            if failure is not None:
                raise failure
            raise RuntimeError("a rank failed its shard")
        start = coll.allreduce_min(mine_k if mine == cmin else NO_RESULT)
        assert start != NO_RESULT
        cost = cmin
    tour = np.zeros(n, dtype=np.int32)
    if owner_of(start, world) == rank:
        tour[:] = tours[ids.index(start)]
    coll.bcast_tour(owner_of(start, world), tour)
    return {"cost": cost, "start": start, "tour": tour, "seconds": seconds, "local_starts": len(ids)}


# ---- the two multi-start workloads of BASELINE.json, on the device engine --------------------------------------

def succ_to_perm(succ):
    """Successor list -> the permutation that starts at node 0 (the chromosome walk of src/genetic.c:436-441)."""
    n = len(succ)
    s = succ.tolist()
    perm = [0] * n
    v = 0
    for k in range(n):
        perm[k] = v
        v = s[v]
    return np.asarray(perm, dtype=np.int32)


def succ_to_perm_batch(succ):
    """The same walk for a batch [B, n] of successor lists, one vectorised step per tour position."""
    succ = np.asarray(succ)
    B, n = succ.shape
    perm = np.zeros((B, n), dtype=np.int32)
    rows = np.arange(B)
    v = np.zeros(B, dtype=np.int64)
    for k in range(1, n):
        v = succ[rows, v]
        perm[:, k] = v
    return perm


def perm_to_succ(perm):
    """Permutation -> successor list (from_chromosome_to_edges, src/genetic.c:33-42)."""
    perm = np.asarray(perm, dtype=np.int32)
    succ = np.empty(len(perm), dtype=np.int32)
    succ[perm] = np.roll(perm, -1)
    return succ


class LibcRandom:
    """The process-global libc random() stream the reference draws from (srandom(seed) in src/solver.c:264-266;
    URAND() = random() / RAND_MAX, include/utility.h:36)."""
    RAND_MAX = 2147483647

    def __init__(self, seed):
        import ctypes
        self._libc = ctypes.CDLL(None)
        self._libc.random.restype = ctypes.c_long
        self._libc.srandom(ctypes.c_uint(seed))

    def urand(self):
        return self._libc.random() / self.RAND_MAX

    def random_perm(self, n):
        """random_generation of src/genetic.c:349-364: the identity permutation, then n swaps of two rand_choice(0, n)
        positions (src/utility.c:752: from + (int)(URAND() * (to - from)))."""
        perm = list(range(n))
        for _ in range(n):
            p = int(self.urand() * n)
            q = int(self.urand() * n)
            perm[p], perm[q] = perm[q], perm[p]
        return np.asarray(perm, dtype=np.int32)


def grasp_stream(urand, n, num_starts):
    """Start nodes and URAND streams of `num_starts` GRASP starts in the reference's draw order: one draw for the
    start node (src/heuristics.c:519), then the n draws of grasp() (:127).  `urand` is a callable returning the
    next URAND() of the libc stream (the caller seeds it).  Every rank draws the whole stream and keeps its shard."""
    starts = np.zeros(num_starts, dtype=np.int32)
    stream = np.zeros((num_starts, n))
    for k in range(num_starts):
        starts[k] = int(urand() * (n - 1))
        stream[k] = [urand() for _ in range(n)]
    return starts, stream


def config4_refiner(E, inst, starts, stream):
    """BASELINE configs[3]: GRASP tours for the shard's starts + alg_2opt each, true cost through the fitness kernel."""
    def refine(ids):
        ids = list(ids)
        succ, obj, _ = inst.construct(E.GRASP, starts[ids], stream[ids])
        rc, s2, o2, st = inst.two_opt(succ, obj, mode=E.FIRST)
        if rc != E.OK:
            raise E.TspDeviceError("alg_2opt on the shard returned status %d" % rc)
        # true cost = sum over nodes of d(v, succ v): one batched spot-distance call (integer costs: any order of the sum is
        # exact; the walk from node 0 that perm_cost needs would be n dependent gathers on the host: 5 of this function's 8 ms)
        n = s2.shape[1]
        d = inst.dist_pairs(np.tile(np.arange(n, dtype=np.int32), len(ids)), s2.reshape(-1))
        true_cost = d.reshape(len(ids), n).sum(axis=1)
        refine.stats = st
        return true_cost, s2
    return refine


def config5_refiner(E, inst, perms):
    """BASELINE configs[4]: the shard's random individuals (src/genetic.c:349-364) each refined by alg_2opt
    (the mutation-3 path, src/genetic.c:426-443, without its 2 s limit)."""
    def refine(ids):
        ids = list(ids)
        p = perms[ids]
        succ = np.empty_like(p)
        np.put_along_axis(succ, p, np.roll(p, -1, axis=1), axis=1)          # perm_to_succ for the whole shard
        cost = inst.perm_cost(p)
        rc, s2, o2, st = inst.two_opt(succ, cost, mode=E.FIRST)
        if rc != E.OK:
            raise E.TspDeviceError("alg_2opt on the shard returned status %d" % rc)
        refine.stats = st
        return o2, s2
    return refine
