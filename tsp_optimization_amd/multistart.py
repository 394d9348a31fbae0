"""Multi-start across the GPUs of one node: shard the starts, no data-path collective, one
all-reduce(min) of a packed (cost, start id) over RCCL (torch.distributed backend "nccl"; "gloo"
in the CPU tests), then the winner's tour is broadcast from the rank that owns it.

The packing is the one SURVEY.md section 5 proposes: cost in the high bits, start id in the low 24,
so that the integer minimum is (lowest cost, then lowest start id) -- the same winner a serial
loop that keeps the first strictly better start (heuristics.c:534) would report.
"""
ID_BITS = 24
ID_MASK = (1 << ID_BITS) - 1


def shard_starts(num_starts, rank, world):
    """Global start ids owned by `rank`: k with k % world == rank (round-robin keeps ragged counts balanced)."""
    return list(range(rank, num_starts, world))


def owner_of(start_id, world):
    return start_id % world


def pack(cost, start_id):
    c = int(cost)
    assert c == cost and c >= 0, "packed reduction needs non-negative integer costs"
    assert 0 <= start_id <= ID_MASK
    return (c << ID_BITS) | start_id


def unpack(packed):
    return packed >> ID_BITS, packed & ID_MASK


NO_RESULT = (1 << 62)


def local_best(costs, start_ids):
    """Packed minimum over this rank's starts (NO_RESULT for an empty shard)."""
    best = NO_RESULT
    for c, k in zip(costs, start_ids):
        best = min(best, pack(c, k))
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
