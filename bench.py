#!/usr/bin/env python3
"""bench.py -- 2-opt edge-pair evaluations per second and time-to-local-optimum on MI355X (BASELINE.json's metric).

Workload (BASELINE.json configs[2], the one the >= 1e9 evals/s target is quoted on): synthetic random EUC_2D,
n = 10000, numpy default_rng(10000) integer coordinates in [0,1e6)^2, integer costs, single start.

One *step* = ONE FULL best-improvement 2-opt descent (alg_2opt_tabu with skip_edge == NULL, src/tabusearch.c:107-178)
of the nearest-neighbour tour greedy(start = rank) to its local optimum: 1428 sweeps for rank 0, every sweep deciding
all n(n-1)/2 - n = 49 985 000 non-adjacent pairs, picking the arg-min, reversing the segment.  The tour is resident in
HBM (tsp_dev_tours) before the timed region starts; a step restores it device-to-device and runs the product's default
engine for resident tours (TSP_ENGINE_AUTO -> CLUSTER: 256 workgroups, one descent = one launch).  The final tour is
checked against the committed golden vector after the timed region.

What the line reports, kept strictly apart:
  value                               pair evaluations the device EXECUTED per second: pairs for which a lane
                                      evaluated a lower bound of delta or delta itself (tier counters of the kernel);
  evals.exact_delta_per_s             delta expressions (tabusearch.c:150) actually executed per second;
  evals.reference_equivalent_pairs_per_s   pairs DECIDED per second (what the reference would have executed for the
                                      same, bit-identical decisions) -- most are decided 64 x 64 at a time by a box bound;
  roofline                            bound "valu": counted floating-point lane-operations of the executed tiers / kernel
                                      time against the fp64 vector peak, frac <= 1; the same for the exhaustive tiled
                                      sweep (every delta expression executed) in roofline.exhaustive.
With N GPUs each rank refines its own start (weak scaling, no data-path collective); after the timed region the ranks
run the sharded multi-start configs (BASELINE configs[3], [4]) with one RCCL all-reduce(min) + one broadcast each.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_NODES = 10000
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
FP64_LANE_OPS_PEAK = 39.3e12     # 78.6 TFLOP/s fp64 vector (FMA = 2 flops) = 39.3 T lane-instructions/s
# counted floating-point lane-operations per unit of executed work (fp32 operations count 1/2)
OPS_TIER0_F32 = 11 * 0.5         # dx, dy, dx*dx, fma, two adds for T, T*|T|, two scalings, compare
OPS_TIER1 = 26.0                 # both new edges without a root: 4 sub, 3 add, 2 x (mul + fma + scale), w, 4 p1 p2, w^2, 3 compares
OPS_EXACT = 35.0                 # 31 fp64 operations + 4 roots per delta expression
OPS_STAGED = 24.0                # one node record: rounded root distance (12) + row culling against a box (12)
OPS_BOXTEST = 14.0               # one group pair: box gap (6), bound (3), squares and compare (5)


def rand_instance(n):
    return np.random.default_rng(n).integers(0, 1_000_000, size=(n, 2)).astype(np.float64)


def fnv1a(v):
    """The hash of a successor list the golden fixtures carry (FNV-1a walk over the bytes of the int32 array, with the
    basis the fixture generator has used since round 1: 1469598103934665603)."""
    h = 1469598103934665603
    for b in np.ascontiguousarray(v, dtype=np.int32).tobytes():
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


def sharded_configs(E, MS, ctx, rank, world, device, comm=None):
    """BASELINE configs[3] and [4] sharded k % world over the ranks: construct + 2-opt per rank, one all_reduce(MIN) of
    the packed (cost, start), one broadcast of the winner's tour; every rank checks the winner against the goldens.
    comm: the C ABI's RCCL communicator (engine.Comm) -- the two collectives then go through libtsp_hip.so
    (tsp_dev_multistart_allreduce / _bcast_tour), as the C host's HEU_2opt_grasp_multistart runs them."""
    res = {}
    how = ("none (1 GPU)" if world == 1 and comm is None else
           "all_reduce(min) int64 + broadcast 4n bytes over RCCL, " +
           ("through the C ABI (tsp_dev_multistart_allreduce / tsp_dev_multistart_bcast_tour)" if comm is not None
            else "through torch.distributed"))

    def per_rank(seconds):
        if world == 1:
            return [seconds]
        import torch
        import torch.distributed as dist
        mine = torch.tensor([seconds], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        return [float(x.item()) for x in every]

    def wall(seconds):
        if world == 1:
            return seconds
        import torch
        import torch.distributed as dist
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # configs[3]: att532, 256 GRASP starts (seed 123, stream order of heuristics.c:519 then :127) + alg_2opt each
    from tsp_optimization_amd import tsplib
    xy, wt = tsplib.parse(os.path.join(ROOT, "tests", "golden", "instances", "att532.tsp"))
    n = len(xy)
    inst = E.Instance(ctx, xy, wt, 1)
    rng = MS.LibcRandom(123)
    starts, stream = MS.grasp_stream(rng.urand, n, 256)
    refine = MS.config4_refiner(E, inst, starts, stream)
    refine(MS.shard_starts(256, rank, world)[:4])   # warm
    t0 = time.perf_counter()
    out = MS.run_sharded(refine, 256, n, rank, world, device, comm=comm)
    t_all = wall(time.perf_counter() - t0)
    table = golden("oracle_vectors.json")["att532_multistart256"]
    # checks are recorded, never raised: a rank that threw between two collectives would leave the others waiting
    ok4 = (out["cost"], out["start"]) == (28998, 122)                  # best true cost 28998 at start 122
    tour4 = fnv1a(out["tour"]) == table[122]["hash"]
    res["config4_att532_grasp256_2opt"] = {
        "starts": 256, "starts_per_rank": out["local_starts"], "wall_s": t_all, "refine_s_max_over_ranks": wall(out["seconds"]),
        "refine_s_per_rank": per_rank(out["seconds"]),
        "best_true_cost": out["cost"], "best_start": out["start"], "reference_best": [28998, 122],
        "winner_is_the_reference_winner": bool(ok4), "winner_tour_matches_golden": bool(tour4),
        "collectives": how}
    inst.close()

    # configs[4]: synthetic n = 5000, 128 random individuals (genetic.c:349-364, seed 123) each refined by alg_2opt
    xy = rand_instance(5000)
    inst = E.Instance(ctx, xy, E.EUC_2D, 1)
    rng = MS.LibcRandom(123)
    perms = np.stack([rng.random_perm(5000) for _ in range(128)])
    refine = MS.config5_refiner(E, inst, perms)
    t0 = time.perf_counter()
    out = MS.run_sharded(refine, 128, 5000, rank, world, device, comm=comm)
    t_all = wall(time.perf_counter() - t0)
    gold = golden("oracle_vectors_big.json")["config5_rand5000_pop128"]["individuals"]
    best = min(gold, key=lambda r: (r["cost"], r["k"]))
    ok5 = (out["cost"], out["start"]) == (int(best["cost"]), best["k"])
    tour5 = fnv1a(out["tour"]) == best["hash"]
    mine = MS.shard_starts(128, rank, world)
    ev = int(sum(x["evals"] for x in refine.stats))
    local5 = all(int(refine.stats[i]["evals"]) == gold[k]["ev"] and int(refine.stats[i]["moves"]) == gold[k]["mv"]
                 for i, k in enumerate(mine))
    all5 = wall(0.0 if local5 else 1.0) == 0.0        # max over ranks of "some local individual differs"
    res["config5_rand5000_population128_2opt"] = {
        "individuals": 128, "individuals_per_rank": out["local_starts"], "wall_s": t_all,
        "refine_s_max_over_ranks": wall(out["seconds"]), "refine_s_per_rank": per_rank(out["seconds"]), "collectives": how,
        "best_cost": out["cost"], "best_individual": out["start"],
        "golden_best": [int(best["cost"]), best["k"]], "rank0_reference_equivalent_evals": ev,
        "winner_is_the_golden_winner": bool(ok5), "winner_tour_matches_golden": bool(tour5),
        "every_individual_on_every_rank_matches_golden_counters": bool(all5)}
    inst.close()
    return res


def cpu_baselines(xy, wt, succ0, obj0, cores):
    """The oracle (oracle/tsp_oracle.c, the reference restated; kind "port": the reference needs cplex.h and cannot be
    built here) timed on this box's host cores on bounded samples of the same workloads."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    out = {}
    sweeps = 15
    _, _, _, cst, _, _ = O.two_opt_best(xy, wt, succ0, max_sweeps=sweeps)
    out["best_improvement"] = {"evals_per_s": cst["evals"] / cst["seconds"], "seconds": cst["seconds"], "cores": 1,
                               "sample": "%d sweeps of alg_2opt_tabu on the rand10000 greedy tour" % sweeps}
    # alg_2opt as the reference runs it: gettimeofday per pair (heuristics.c:456-462), and with the clock hoisted
    for label, cpp in (("first_improvement_faithful_clock_per_pair", 1), ("first_improvement_clock_hoisted", 0)):
        _, _, _, st, _ = O.two_opt_first(xy, wt, succ0, obj0, time_limit=5.0, clock_per_pair=cpp)
        out[label] = {"evals_per_s": st["evals"] / st["seconds"], "seconds": st["seconds"], "cores": 1,
                      "sample": "the first %.0f s of alg_2opt on the rand10000 greedy tour (%d evaluations)" % (st["seconds"], st["evals"])}
    # configs[3] on all host cores: 256 GRASP starts of att532 (libc stream, sequential) then alg_2opt each in a thread pool
    axy, awt = O.parse_tsplib(os.path.join(ROOT, "tests", "golden", "instances", "att532.tsp"))
    O.srandom(123)
    t0 = time.perf_counter()
    tours = []
    for k in range(256):
        node = int(O.urand() * (len(axy) - 1))
        _, s, o = O.grasp(axy, awt, start=node)
        tours.append((s, o))
    t1 = time.perf_counter()

    def one(k):
        _, s2, _, st, _ = O.two_opt_first(axy, awt, tours[k][0], tours[k][1])
        return O.succ_cost(axy, awt, s2), st["evals"]
    with ThreadPoolExecutor(max_workers=cores) as ex:   # ctypes releases the GIL; the descent keeps no global state
        rows = list(ex.map(one, range(256)))
    t2 = time.perf_counter()
    k = min(range(256), key=lambda i: (rows[i][0], i))
    out["config4_att532_grasp256_2opt_all_cores"] = {
        "seconds": t2 - t0, "grasp_s": t1 - t0, "two_opt_s": t2 - t1, "cores": cores, "evals": int(sum(r[1] for r in rows)),
        "best_true_cost": rows[k][0], "best_start": k,
        "note": "the reference is single-threaded; this is an outer loop over the starts on all host cores"}
    # configs[4] on all host cores, capped: the first 2 x cores of the 128 random individuals of rand5000 (genetic.c:349-364,
    # seed 123), alg_2opt each (about 2 s of CPU per individual), extrapolated to the population of 128
    from tsp_optimization_amd import multistart as MS
    xy5 = rand_instance(5000)
    rng = MS.LibcRandom(123)
    sample = min(128, 2 * cores)
    perms = [rng.random_perm(5000) for _ in range(sample)]
    gold = golden("oracle_vectors_big.json")["config5_rand5000_pop128"]["individuals"]

    def one5(k):
        succ = MS.perm_to_succ(perms[k])
        c0 = O.succ_cost(xy5, O.EUC_2D, succ)
        _, s2, o2, st, _ = O.two_opt_first(xy5, O.EUC_2D, succ, c0)
        return o2, st["evals"], st["moves"]
    t3 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        rows5 = list(ex.map(one5, range(sample)))
    t4 = time.perf_counter()
    out["config5_rand5000_population128_2opt_all_cores"] = {
        "sample": "%d of the 128 individuals (the first ones, whole descents), %d threads" % (sample, cores),
        "seconds_sample": t4 - t3, "seconds_extrapolated_to_128": (t4 - t3) * 128.0 / sample, "cores": cores,
        "evals_sample": int(sum(r[1] for r in rows5)), "moves_sample": int(sum(r[2] for r in rows5)),
        "sample_matches_golden": bool(all(int(rows5[k][0]) == int(gold[k]["cost"]) and rows5[k][1] == gold[k]["ev"] for k in range(sample))),
        "note": "extrapolated linearly (the individuals cost alike: 1.4e8 evaluations and 40 k moves each); the reference is "
                "single-threaded and runs mutation 3 with a 2 s limit per individual (genetic.c:432)"}
    return out



def launch_ranks(argv, n_ranks):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks (one process per GPU) ourselves.

    The parent touches no GPU (no HIP call, no torch.cuda call -- it imports neither the engine nor torch) and never
    exec()s: it starts N child processes of this very script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set (what torch.distributed.run would set), relays rank 0's JSON line as its own last line of stdout and
    exits non-zero if any child did.  The other ranks' stdout goes to stderr, prefixed."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sk:                 # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               TSP_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n_ranks):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv,
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n_ranks)),
                                      stdout=subprocess.PIPE, stderr=None, text=True))
    lines = [[] for _ in procs]

    def pump(r):
        for ln in procs[r].stdout:
            lines[r].append(ln.rstrip("\n"))
            if r != 0:
                sys.stderr.write("[rank %d] %s" % (r, ln))
    threads = [threading.Thread(target=pump, args=(r,), daemon=True) for r in range(n_ranks)]
    for t in threads:
        t.start()
    limit = float(os.environ.get("TSP_BENCH_LAUNCH_TIMEOUT", "3000"))
    t0 = time.time()
    codes = [None] * n_ranks
    first_failure = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        failed = any(c not in (None, 0) for c in codes)
        if failed and first_failure is None:
            first_failure = next(c for c in codes if c not in (None, 0))   # the rank that failed by itself, not the ones ended below
        if failed or time.time() - t0 > limit:
            # a rank died (or the job hangs): the others would wait in a collective for ever -- end exactly the processes
            # this parent started (by PID), give them a moment, then kill
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            if not failed:
                codes = [c if c else 124 for c in codes]
            break
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=10)
    json_line = None
    for ln in lines[0]:
        if ln.startswith("{") and ln.endswith("}"):
            json_line = ln
        else:
            sys.stderr.write("[rank 0] %s\n" % ln)
    rc = first_failure if first_failure is not None else next((c for c in codes if c), 0)
    if json_line is None and rc == 0:
        rc = 1
    sys.stderr.flush()
    if json_line is not None:
        print(json_line, flush=True)
    return rc


def launcher_selftest_child():
    """TSP_BENCH_SELFTEST=1 (CPU tests of the self-launcher, no GPU): every rank joins a gloo group, the ranks count
    themselves with one all-reduce and run the multi-start launcher on the golden table; rank 0 prints the line."""
    import torch
    import torch.distributed as dist
    from tsp_optimization_amd import multistart as MS
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    one = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(one)
    table = golden("oracle_vectors.json")["att532_multistart256"]

    def refine(ids):
        return [table[k]["opt_true"] for k in ids], np.stack([np.full(532, k, dtype=np.int32) for k in ids])
    out = MS.run_sharded(refine, len(table), 532, rank, world)
    fail_rank = os.environ.get("TSP_BENCH_SELFTEST_FAIL_RANK")
    dist.barrier()
    dist.destroy_process_group()
    if fail_rank is not None and int(fail_rank) == rank:
        sys.exit(3)
    if rank == 0:
        print("not the json line")
        print(json.dumps({"selftest": True, "n_gpus": world, "ranks_seen": int(one.item()), "cost": out["cost"],
                          "start": out["start"], "tour_ok": bool((out["tour"] == out["start"]).all())}), flush=True)
    else:
        print("rank %d done" % rank, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip everything but the timed region and the roofline")
    ap.add_argument("--no-variants", action="store_true", help="skip the exhaustive sweeps (keeps a profile clean)")
    args = ap.parse_args()

    # --gpus N without a launcher around this process: start the N ranks here, BEFORE anything touches the GPU
    force_dist = os.environ.get("TSP_BENCH_FORCE_DIST") == "1"
    selftest = os.environ.get("TSP_BENCH_SELFTEST") == "1"
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist or selftest):
        sys.exit(launch_ranks(sys.argv[1:], max(1, args.gpus)))
    if selftest:
        return launcher_selftest_child()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    device = None
    # TSP_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (used to test it on a 1-GPU box)
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        import datetime
        dist.init_process_group(backend="nccl", device_id=device, timeout=datetime.timedelta(seconds=300))
    else:
        local_rank = 0

    from tsp_optimization_amd import engine as E
    from tsp_optimization_amd import multistart as MS
    if E.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")

    def barrier_sync():
        ctx.synchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    xy = rand_instance(N_NODES)
    wt = E.EUC_2D
    ctx = E.Context(local_rank)
    # The RCCL communicator of the C ABI (what the C host's HEU_2opt_grasp_multistart uses): rank 0's id travels to the
    # other ranks as a byte tensor over the torch group.  A failure on rank 0 travels as an all-zero id, so that every
    # rank takes the same branch and nobody waits in a collective the others never enter.
    comm = None
    dist_info = {}
    if dist is not None:
        import torch

        def all_agree(flag):   # min over the ranks of a 0 / 1 flag: every rank takes the same branch
            t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())
        # tsp_dev_comm_init_rank is collective: only enter it when EVERY rank can load RCCL through the C ABI and rank 0 has an id
        if all_agree(E.comm_available()):
            idt = torch.zeros(E.COMM_ID_BYTES, dtype=torch.uint8, device=device)
            if rank == 0:
                try:
                    idt.copy_(torch.frombuffer(bytearray(E.comm_unique_id()), dtype=torch.uint8))
                except Exception as e:   # noqa: BLE001 -- reported in the line; travels to the other ranks as an all-zero id
                    dist_info["c_abi_comm_error"] = repr(e)
            dist.broadcast(idt, src=0)
            raw = idt.cpu().numpy().tobytes()
            if any(raw):
                try:
                    comm = E.Comm(ctx, world, rank, raw)
                except Exception as e:   # noqa: BLE001
                    dist_info["c_abi_comm_error"] = repr(e)
                if not all_agree(comm is not None):   # some rank failed after the rendezvous: nobody uses the C communicator
                    if comm is not None:
                        comm.close()
                    comm = None
            if comm is not None:
                dist_info["c_abi_comm"] = "tsp_dev_comm_init_rank over RCCL %d" % comm.rccl_version()
        else:
            dist_info["c_abi_comm_error"] = "librccl could not be opened through the C ABI on some rank"
        one = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(one)
        dist_info["rccl_ranks_seen"] = int(one.item())
    inst = E.Instance(ctx, xy, wt, 1)
    start_node = rank % N_NODES
    succ0, obj0, status = inst.construct(E.GREEDY, np.array([start_node], dtype=np.int32))
    assert status[0] == 0
    tours = E.Tours(inst, 1)
    tours.upload(succ0[0], obj0[0])          # resident in HBM from here on (and remembered as the reset point)
    pairs_per_sweep = N_NODES * (N_NODES - 1) // 2 - N_NODES

    def one_descent():
        tours.reset()                        # device-to-device restore of the start tour
        rc, done = tours.run_engine(E.BEST, engine=E.ENGINE_AUTO)
        assert rc == 0 and done

    for _ in range(args.warmup):
        one_descent()
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_descent()
    barrier_sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    s_fin, o_fin, st_fin = tours.download()
    st = st_fin[0]
    # every rank ran `steps` descents of its own start; totals over the job (counters are per descent, identical each step)
    tot = {k: float(st[k]) for k in ("lane_pairs", "tier1_pairs", "exact_pairs", "staged_recs", "evals", "sweeps")}
    if dist is not None:
        import torch
        v = torch.tensor([tot[k] for k in sorted(tot)], dtype=torch.float64, device=device)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        tot = dict(zip(sorted(tot), [float(x) for x in v.tolist()]))
    counted = st["exact_pairs"] >= 0
    lane_pairs_job = tot["lane_pairs"] * args.steps
    value = lane_pairs_job / elapsed

    # multi-start epilogue of the timed workload: one RCCL all-reduce(min) of the packed (true cost, rank)
    cost_now, _, packed = tours.best(true_cost=True)
    best_cost, best_rank = int(cost_now), rank
    if dist is not None:
        best_cost, best_rank = MS.allreduce_best(MS.pack(int(cost_now), rank), device=device)
        if comm is not None:   # the same reduction through the C ABI's communicator: both must agree
            c_cost, c_rank = MS.unpack(comm.allreduce_min(MS.pack(int(cost_now), rank)))
            dist_info["c_abi_allreduce_agrees_with_torch"] = bool((c_cost, c_rank) == (best_cost, best_rank))

    out = {
        "metric": "2opt_edge_pair_evals_per_sec",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[2]: synthetic random EUC_2D n=10000 (numpy default_rng(10000), integer coords in "
                        "[0,1e6)^2), single-start 2-opt; step = one full best-improvement descent (alg_2opt_tabu with "
                        "skip_edge == NULL, tabusearch.c:107-178) of greedy(start=rank) to its local optimum on a tour "
                        "resident in HBM, default engine (TSP_ENGINE_AUTO)",
            "n": N_NODES, "starts_per_gpu": 1, "pairs_per_sweep": pairs_per_sweep, "sweeps_per_step_rank0": int(st["sweeps"]),
            "step": "device-to-device restore of the start tour + %d sweeps (box / bound / exact tiers) + arg-min + segment "
                    "reversal each" % int(st["sweeps"]),
            "parallelism": "multi-start x%d (one start per GPU, no data-path collective)" % world,
        },
        "evals": {
            "definition": "value = pairs for which a lane evaluated a lower bound of delta or delta itself (kernel tier "
                          "counters), per second, whole job.  NOT counted in value: pairs decided 64 x 64 at a time by the "
                          "box form of the new-edge bound or by row culling.  reference_equivalent_pairs_per_s counts every "
                          "pair the reference would have evaluated for the same (bit-identical) decisions.",
            "counted_on_device": bool(counted),
            "per_step_rank0": {"lane_pairs_tier0": int(st["lane_pairs"]), "tier1_pairs": int(st["tier1_pairs"]),
                               "exact_delta_expressions": int(st["exact_pairs"]), "staged_node_records": int(st["staged_recs"]),
                               "reference_equivalent_pairs": int(st["evals"]), "sweeps": int(st["sweeps"]), "moves": int(st["moves"])},
            "lane_pairs_per_s": value,
            "exact_delta_per_s": tot["exact_pairs"] * args.steps / elapsed if counted else None,
            "reference_equivalent_pairs_per_s": tot["evals"] * args.steps / elapsed,
        },
        "time_to_local_optimum_s": elapsed / args.steps,
        "multistart_best": dict({"cost": best_cost, "rank": best_rank,
                                 "collective": ("all_reduce(min) int64 over RCCL (torch.distributed and the C ABI's "
                                                "tsp_dev_multistart_allreduce)") if dist is not None else "none (1 GPU)"},
                                **dist_info),
    }

    if rank == 0:
        big = golden("oracle_vectors_big.json")["rand10000_best"]["final"]
        out["parity"] = {"final_cost": float(o_fin[0]), "golden_cost": big["cost"], "sweeps": int(st["sweeps"]),
                         "golden_sweeps": big["stats"]["sweeps"], "moves": int(st["moves"]),
                         "final_tour_matches_golden": bool(fnv1a(s_fin[0]) == big["hash"] and o_fin[0] == big["cost"]
                                                           and st["sweeps"] == big["stats"]["sweeps"]
                                                           and st["evals"] == big["stats"]["evals"]
                                                           and st["moves"] == big["stats"]["moves"]),
                         "golden": "tests/golden/oracle_vectors_big.json (oracle: full CPU descent, 23 min)"}

    def guarded(section, fn):
        """Optional (rank-0, collective-free) sections must never cost the JSON line."""
        try:
            fn()
        except Exception as e:   # noqa: BLE001 -- reported, not swallowed
            out[section] = {"error": repr(e)}

    def roofline_section():
        # roofline of the dominant kernel of the timed region: one k_cluster_two_opt launch per descent, HIP events on the
        # engine's stream around a further descent (stats.device_ms), counted lane-operations of the executed tiers
        one_descent()
        _, _, st2 = tours.download()
        kernel_ms = st2[0]["device_ms"]
        sweeps = st2[0]["sweeps"]
        ops = (st2[0]["lane_pairs"] * OPS_TIER0_F32 + max(0, st2[0]["tier1_pairs"]) * OPS_TIER1 +
               max(0, st2[0]["exact_pairs"]) * OPS_EXACT + max(0, st2[0]["staged_recs"]) * OPS_STAGED +
               sweeps * (inst_groups(N_NODES) * (inst_groups(N_NODES) + 1) // 2) * OPS_BOXTEST)
        achieved = ops / (kernel_ms * 1e-3)
        traffic = None
        tj = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tj):
            with open(tj) as f:
                traffic = json.load(f).get("r03_cluster_descent_n10000_hbm_bytes_per_launch")
        exhaustive = {}
        if not args.no_variants:
            # the same sweep with every delta expression executed (tiled kernel k_recs + k_step, bounds off): the kernel
            # whose work IS the reference's 49 985 000 evaluations per launch, VALU-throughput bound
            os.environ["TSP_NO_FILTER"] = "1"
            inst_v = E.Instance(ctx, xy, wt, 1)
            tours_v = E.Tours(inst_v, 1)
            del os.environ["TSP_NO_FILTER"]
            tours_v.upload(succ0[0], obj0[0])
            ms_v, ev_v = tours_v.time_scan(reps=40)
            exhaustive = {"kernel": "tsp::k_recs + tsp::k_step<EUC_2D icoord, BEST> with the bounds off (TSP_NO_FILTER=1): every "
                                    "non-adjacent pair gets the exact delta", "kernel_ms": ms_v, "delta_expressions_per_launch": ev_v,
                          "exact_delta_per_s": ev_v / (ms_v * 1e-3), "bound": "valu", "ops_per_eval": OPS_EXACT,
                          "achieved": ev_v * OPS_EXACT / (ms_v * 1e-3) / 1e12, "peak": FP64_LANE_OPS_PEAK / 1e12,
                          "unit": "T lane-op/s", "frac": ev_v * OPS_EXACT / (ms_v * 1e-3) / FP64_LANE_OPS_PEAK,
                          "rocprof": "profiles/r03_kernel_stats_exhaustive.csv"}
            tours_v.close()
            inst_v.close()
        out["roofline"] = {
            "kernel": "tsp::k_cluster_two_opt<EUC_2D integer-coordinate variant, BEST, float replica, sorted scan> -- one launch = "
                      "one whole descent (%d sweeps) on 256 workgroups; kernel_ms = HIP events on the engine's stream around one "
                      "further descent after the timed region (rocprof mean: profiles/r03_kernel_stats.csv)" % sweeps,
            "bound": "valu", "achieved": achieved / 1e12, "peak": FP64_LANE_OPS_PEAK / 1e12,
            "unit": "T lane-op/s (fp64 vector lane-instructions; peak = 78.6 TFLOP/s / 2; fp32 operations count 1/2)",
            "frac": achieved / FP64_LANE_OPS_PEAK, "traffic": traffic,
            "kernel_ms": kernel_ms, "us_per_sweep": 1e3 * kernel_ms / sweeps, "counted_lane_ops_per_launch": ops,
            "ops_per_unit": {"tier0_pair_f32": OPS_TIER0_F32, "tier1_pair": OPS_TIER1, "exact_delta": OPS_EXACT,
                             "staged_record": OPS_STAGED, "group_pair_box_test": OPS_BOXTEST},
            "operand_bandwidth": {"note": "SURVEY 8(d) also asks for evals/s x 72 B against the HBM peak: with the operands "
                                          "on chip it is not a roofline (the tour is read once per launch), reported for completeness",
                                  "reference_equivalent_GBps": st2[0]["evals"] * 72.0 / (kernel_ms * 1e-3) / 1e9,
                                  "executed_lane_pairs_GBps": st2[0]["lane_pairs"] * 72.0 / (kernel_ms * 1e-3) / 1e9,
                                  "hbm_peak_GBps": HBM_PEAK_GBS},
            "note": "the step is a chain of latencies (LDS gathers at two waves per SIMD, one all-to-all exchange through L2 per "
                    "sweep, block barriers), not a throughput kernel: the executed arithmetic is ~1-2 % of the VALU peak by "
                    "design -- see DESIGN.md 4.8; roofline.exhaustive is the same sweep with nothing pruned",
            "exhaustive": exhaustive,
        }

    def first_section():
        # the north star's named function, alg_2opt (first improvement, heuristics.c:438-502), on a resident tour: the
        # CLUSTER engine's first-improvement kernels (the plain replica for dense phases, the replica in rank order with the
        # box-pruned step for sparse ones; they hand the descent to each other between launches)
        tf = E.Tours(inst, 1)
        tf.upload(succ0[0], obj0[0])
        ms = []
        for r in range(4):
            tf.reset()
            rc, done = tf.run_engine(E.FIRST, engine=E.ENGINE_AUTO)
            assert rc == 0 and done
            _, of, stf = tf.download()
            if r:
                ms.append(stf[0]["device_ms"])
        stf = stf[0]
        t_nohit = []
        for r in range(12):           # at the local optimum: the sweep that finds nothing (every descent ends with one)
            ctx.synchronize()
            t1 = time.perf_counter()
            tf.two_opt(E.FIRST)
            t_nohit.append(time.perf_counter() - t1)
        tf.close()
        ops = (stf["lane_pairs"] * OPS_TIER0_F32 + max(0, stf["tier1_pairs"]) * OPS_TIER1 + max(0, stf["exact_pairs"]) * OPS_EXACT +
               max(0, stf["staged_recs"]) * OPS_STAGED)
        kms = float(np.mean(ms))
        out["roofline"]["first"] = {
            "kernel": "tsp::k_cluster_two_opt<EUC_2D integer-coordinate variant, FIRST, float replica, plain | rank order>: one descent = "
                      "a few launches (hand-overs between the two variants); device_ms = HIP events on the engine's stream around "
                      "all of them (rocprof: profiles/r03_kernel_stats_first.csv)",
            "bound": "valu", "device_ms": kms, "steps": int(stf["steps"]), "us_per_step": 1e3 * kms / max(1, stf["steps"]),
            "sweeps": int(stf["sweeps"]), "reference_evals": int(stf["evals"]), "moves": int(stf["moves"]),
            "reference_counters_match": bool((stf["sweeps"], stf["evals"], stf["moves"]) == (10, 499850987, 2704) and of[0] == 77370387),
            "counted_lane_ops_per_descent": ops, "achieved": ops / (kms * 1e-3) / 1e12, "peak": FP64_LANE_OPS_PEAK / 1e12,
            "frac": ops / (kms * 1e-3) / FP64_LANE_OPS_PEAK, "unit": "T lane-op/s",
            "reference_equivalent_evals_per_s": stf["evals"] / (kms * 1e-3),
            "sweep_that_finds_nothing_us_per_call": 1e6 * float(np.mean(t_nohit[2:])),
            "note": "a chain of 2 704 dependent moves: 5.8 us per step, 3.1 of it one all-to-all exchange (DESIGN.md 4.8, 4.8c)"}

    if rank == 0:
        guarded("roofline", roofline_section)
        if "error" not in out.get("roofline", {"error": 1}):
            guarded("roofline_first", first_section)
            ex = out["roofline"].get("exhaustive") or {}
            # the three rates the metric can mean, side by side at the top level
            out["value_definition"] = ("lane_pairs: pairs for which a lane executed a lower bound of delta or delta itself, per second, "
                                       "timed region, whole job")
            out["value_note"] = ("value counts the pairs that reach a lane's tier 0: better pruning LOWERS it (round 2: 1.45e11 in 16.2 ms per descent; "
                                 "round 3's quarter units send 62 % fewer pairs there: 6.4e10 in 12.2 ms, the same decisions).  Compare ms_per_step, "
                                 "time_to_local_optimum and reference_equivalent_pairs_per_s across rounds, not value")
            out["delta_evals_per_s_exhaustive"] = ex.get("exact_delta_per_s")
            out["reference_equivalent_pairs_per_s"] = out["evals"]["reference_equivalent_pairs_per_s"]

    def extras_section():
        extras = {}
        t1 = time.perf_counter()
        rc, s1, o1, st1 = inst.two_opt(succ0[0], obj0[0], mode=E.FIRST)
        dt1 = time.perf_counter() - t1
        extras["first_improvement_alg_2opt"] = {
            "time_to_local_optimum_s": dt1, "device_ms": st1["device_ms"], "final_cost": o1,
            "reference_final_cost": 77370387, "cost_match": bool(o1 == 77370387 and obj0[0] == 88104308),
            "sweeps": st1["sweeps"], "reference_evals": st1["evals"], "moves": st1["moves"],
            "reference_counters_match": bool((st1["sweeps"], st1["evals"], st1["moves"]) == (10, 499850987, 2704)),
            "reference_equivalent_evals_per_s": st1["evals"] / dt1, "lane_pairs_executed": st1["lane_pairs"],
            "exact_delta_expressions_executed": st1["exact_pairs"], "steps": st1["steps"],
            "note": "host tour in, host tour out (PCIe-inclusive: 200 KB each way)"}
        t2 = time.perf_counter()
        rc, s2, o2, st2 = inst.two_opt(succ0[0], obj0[0], mode=E.BEST)
        dt2 = time.perf_counter() - t2
        extras["best_improvement_alg_2opt_tabu_host_tour"] = {
            "time_to_local_optimum_s": dt2, "device_ms": st2["device_ms"], "final_cost": o2, "sweeps": st2["sweeps"],
            "moves": st2["moves"], "note": "host tour in, host tour out (PCIe-inclusive)"}
        out["time_to_local_optimum"] = extras
        # the genuinely HBM-bound kernel of the path: n x n calc_dist matrix (4 n^2 bytes written)
        _, dm_ms = inst.dist_matrix(as_int32=True, fetch=False)
        _, dm64_ms = inst.dist_matrix(as_int32=False, fetch=False)
        out["distance_matrix_build"] = {
            "kernel": "tsp::k_dist_matrix (n=10000)", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "int32": {"kernel_ms": dm_ms, "bytes_written": 4 * N_NODES * N_NODES,
                      "achieved": 4 * N_NODES * N_NODES / (dm_ms * 1e-3) / 1e9,
                      "frac": 4 * N_NODES * N_NODES / (dm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "f64": {"kernel_ms": dm64_ms, "bytes_written": 8 * N_NODES * N_NODES,
                    "achieved": 8 * N_NODES * N_NODES / (dm64_ms * 1e-3) / 1e9,
                    "frac": 8 * N_NODES * N_NODES / (dm64_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "timing": "HIP events around 12 back-to-back launches that rotate over 3 output buffers (1.2 / 2.4 GB in all): no launch "
                      "stores into lines the 256 MB Infinity Cache may still hold"}
        tj = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tj):
            with open(tj) as f:
                r3 = json.load(f).get("r03_dist_matrix")
            if r3:   # rocprofv3 evidence of the same kernel (profiles/r03_*): mean duration and WRITE_SIZE per launch
                for k in ("int32", "f64"):
                    out["distance_matrix_build"][k].update({
                        "rocprof_mean_us": r3[k]["rocprof_mean_us"], "write_size_bytes": r3[k]["write_size_bytes"],
                        "hbm_frac_write_size_over_rocprof_time": r3[k]["write_size_bytes"] / (r3[k]["rocprof_mean_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS})
                out["distance_matrix_build"]["rocprof"] = r3["source"]

    def tabu_section():
        # alg_2opt_tabu WITH a tabu list (tabusearch.c:127-165): the reference reads four stamps of the n(n-1)/2-int list
        # per pair -- 16 B x 49 985 000 = 800 MB per sweep, the one sweep of the path that is HBM-bound when executed
        # as written (k_step<TABU>, TSP_TABU_DENSE=1).  The default works from the compact list of non-zero stamps
        # (two_opt_tabu_list.hpp) and gives the same tours, counters and stamp array.
        live = 400                                     # what tabu() holds at tenure 200 (2 stamps per iteration)
        rng = np.random.default_rng(7)
        idx = rng.choice(N_NODES * (N_NODES - 1) // 2, size=live, replace=False).astype(np.int32)
        res = {}
        ref = None
        for label, dense in (("from_the_list_of_nonzero_stamps", "0"), ("four_stamp_reads_per_pair", "1")):
            os.environ["TSP_TABU_DENSE"] = dense
            inst.reload_switches()                     # switches are read when a handle is created, not per call
            tb = E.Tabu(inst)
            tb.set(idx, np.full(live, 5, dtype=np.int32))
            tt = E.Tours(inst, 1)
            tt.upload(succ0[0], obj0[0])
            ctx.synchronize()
            t1 = time.perf_counter()
            rc, o = tt.two_opt_tabu(tb, 6, 200)
            dt = time.perf_counter() - t1
            s_t, _, st_t = tt.download()
            key = (fnv1a(s_t[0]), o, st_t[0]["sweeps"], st_t[0]["evals"], st_t[0]["moves"])
            ref = ref or key
            res[label] = {"descent_s": dt, "sweeps": st_t[0]["sweeps"], "us_per_sweep": 1e6 * dt / max(1, st_t[0]["sweeps"]),
                          "reference_evals": st_t[0]["evals"], "final_cost": o, "same_result_as_the_other_path": bool(key == ref),
                          "stamp_bytes_the_reference_reads_per_sweep": 16 * pairs_per_sweep,
                          "those_bytes_per_s_GBps": 16.0 * pairs_per_sweep * st_t[0]["sweeps"] / dt / 1e9}
            tb.close(); tt.close()
        os.environ.pop("TSP_TABU_DENSE", None)
        inst.reload_switches()
        res["four_stamp_reads_per_pair"]["hbm_frac"] = res["four_stamp_reads_per_pair"]["those_bytes_per_s_GBps"] / HBM_PEAK_GBS
        res["live_stamps"] = live
        # iterations of tabu() (tabusearch.c:238-309) on resident state: alg_2opt_tabu + incumbent + kick per iteration
        tb = E.Tabu(inst)
        tt = E.Tours(inst, 1)
        tt.upload(s_t[0], o)                      # the local optimum reached above
        krng = np.random.default_rng(11)
        best, iters, tenure = float("inf"), 400, 200
        def one(it):
            nonlocal best
            a, b = int(krng.integers(0, N_NODES)), int(krng.integers(0, N_NODES))
            rc, obj, best, improved, acc = tt.tabu_iteration(tb, it, tenure, a, b, best)
            while not acc:
                acc = tt.tabu_kick(tb, int(krng.integers(0, N_NODES)), int(krng.integers(0, N_NODES)), it, tenure)
        for it in range(1, 41):
            one(it)
        ctx.synchronize()
        t1 = time.perf_counter()
        for it in range(41, 41 + iters):
            one(it)
        ctx.synchronize()
        dt = time.perf_counter() - t1
        _, _, st_d = tt.download()
        res["tabu_iterations_on_resident_state"] = {"iterations": iters, "seconds": dt, "iterations_per_s": iters / dt,
                                                    "tenure": tenure, "incumbent": best, "list_entries": tb.list_info()[0],
                                                    "worked_from_the_list": tb.list_info()[1],
                                                    "sweeps_per_iteration": (st_d[0]["sweeps"]) / float(iters + 40)}
        tb.close(); tt.close()
        out["alg_2opt_tabu_with_a_list"] = res

    if rank == 0 and world == 1 and not args.no_extras:
        guarded("time_to_local_optimum", extras_section)
        guarded("alg_2opt_tabu_with_a_list", tabu_section)

    if not args.no_extras:
        out["other_configs"] = sharded_configs(E, MS, ctx, rank, world, device, comm)
        out["scaling_note"] = {
            "timed_region": "one start per GPU, no data-path collective: weak scaling, flat by construction",
            "sharded_configs": "configs[3] / [4] shard the starts / individuals k % world; every tour is a chain of dependent moves "
                               "(rand5000: 40 k per individual), so more GPUs shorten the chain only by giving a tour more workgroups",
            "expected_2opt_ms_per_rank_measured_on_one_gpu": {
                "world": [1, 2, 4, 8],
                "config4_att532_256_starts": [2.9, 2.9, 2.9, 2.4],
                "config5_rand5000_128_individuals": [190, 183, 154, 145],
                "engine": ["LDS (1 workgroup per tour)", "CLUSTER (4 per tour)", "CLUSTER (8)", "CLUSTER (16)"],
                "source": "tools/pop_time.py, tools/shard_time.py (DESIGN.md section 5): 1.31 x on 8 GPUs for configs[4], 1.2 x for configs[3]"}}
        out["all_checks_ok"] = bool(out.get("parity", {}).get("final_tour_matches_golden", True) and
                                    all(v for d in out["other_configs"].values() for k, v in d.items()
                                        if k.startswith(("winner_", "every_"))))

    def cpu_section():
        cores = max(1, min(16, len(os.sched_getaffinity(0))))   # a 1-GPU box's CPU share
        base = cpu_baselines(xy, wt, succ0[0], obj0[0], cores)
        out["cpu_baseline"] = {
            "value": base["best_improvement"]["evals_per_s"], "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": base["best_improvement"]["sample"] + " by oracle/tsp_oracle.c (gcc -O2), one thread, %.1f s: every "
                      "delta expression executed (compare with evals.reference_equivalent_pairs_per_s and "
                      "roofline.exhaustive.exact_delta_per_s); the reference is single-threaded and cannot be built here "
                      "(needs cplex.h)" % base["best_improvement"]["seconds"],
            "others": {k: v for k, v in base.items() if k != "best_improvement"},
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        guarded("cpu_baseline", cpu_section)

    tours.close()
    inst.close()
    if comm is not None:
        comm.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is block-buffered on a pipe and would otherwise be
        # flushed at exit, after this line: push it out first so that the JSON line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def inst_groups(n):
    return (n + 63) // 64


if __name__ == "__main__":
    main()
