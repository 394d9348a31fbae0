#!/usr/bin/env python3
"""bench.py -- 2-opt edge-pair evaluations per second and time-to-local-optimum on MI355X (BASELINE.json's metric).

Workload (BASELINE.json configs[2], the one the >= 1e9 evals/s target is quoted on): synthetic random EUC_2D,
n = 10000, numpy default_rng(10000) integer coordinates in [0,1e6)^2, integer costs, single start.

The metric as SURVEY.md 8(d) defines it: "steady-state full sweeps in best-improvement mode (every one of 49 985 000 pairs
evaluated per sweep; evals/s = pairs_evaluated / kernel time), plus end-to-end time-to-local-optimum for both modes".

One *step* of the timed region = ONE FULL best-improvement 2-opt descent (alg_2opt_tabu with skip_edge == NULL,
src/tabusearch.c:107-178) of the nearest-neighbour tour greedy(start = rank) to its local optimum WITH EVERY DELTA EXPRESSION
EXECUTED, as the reference executes them: 1428 sweeps for rank 0, each evaluating all n(n-1)/2 - n = 49 985 000 non-adjacent
pairs exactly (k_move_pos + k_exh, csrc/two_opt_exh.hpp: tour-position order, one new distance per pair), picking the
arg-min, reversing the segment.  The tour is resident in HBM before the timed region starts; a step restores it
device-to-device and runs the descent.  The final tour is checked against the committed golden vector after the timed region.
    value      = delta expressions executed per second, whole job (= sweeps x 49 985 000 x steps x ranks / elapsed)
    roofline   = that kernel pair: algorithmic operations per launch (35 per delta) over the mean launch
                 duration measured with HIP events on the engine's stream over the timed region
The product's default engines decide most pairs by bounds instead of executing them (identical decisions, 5 x faster to the
local optimum): those descents are reported as time_to_local_optimum (both move-selection rules, parity-checked), never as value.
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
# what k_exh issues per delta expression (csrc/two_opt_exh.hpp, RJ = 4: 55 VALU instructions per row step of 4 pairs per lane,
# the four v_sqrt_f64 among them holding the issue port for 4 slots each): ONE distance per pair instead of four
SLOTS_EXH = (55 + 3 * 4) / 4.0
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


def host_lib():
    """libtsp_host.so (the C host mirror: the reference's own function names on its `instance` struct) and the ctypes view of that
    struct (include/tsp_host.h == the reference's include/utility.h:113-160)."""
    import ctypes as C
    from tsp_optimization_amd.build import lib_path
    if "lib" not in host_lib.__dict__:
        class _Edge(C.Structure):
            _fields_ = [("i", C.c_int), ("j", C.c_int)]

        class _Method(C.Structure):
            _fields_ = [("id", C.c_int), ("edge_type", C.c_int), ("name", C.c_char_p), ("use_cplex", C.c_int)]

        class _Params(C.Structure):
            _fields_ = [("file_path", C.c_char_p), ("num_threads", C.c_int), ("time_limit", C.c_int), ("method", _Method), ("verbose", C.c_int),
                        ("integer_cost", C.c_int), ("seed", C.c_int), ("perf_prof", C.c_int), ("callback_2opt", C.c_int)]

        class _Solution(C.Structure):
            _fields_ = [("obj_best", C.c_double), ("edges", C.POINTER(_Edge)), ("time_to_solve", C.c_double), ("xbest", C.POINTER(C.c_double))]

        class _Instance(C.Structure):
            _fields_ = [("params", _Params), ("name", C.c_char_p), ("comment", C.c_char_p), ("nodes", C.POINTER(C.c_double)), ("num_nodes", C.c_int),
                        ("weight_type", C.c_int), ("num_columns", C.c_long), ("ind", C.POINTER(C.c_int)), ("thread_seeds", C.POINTER(C.c_uint)),
                        ("solution", _Solution)]
        host_lib.lib, host_lib.Instance, host_lib.Edge = C.CDLL(lib_path("libtsp_host.so")), _Instance, _Edge
    return host_lib.lib, host_lib.Instance


def host_instance(xy, wt, time_limit=-1):
    """An `instance` whose nodes / edges point into numpy arrays the caller keeps alive -> (instance, edges [n, 2] int32)."""
    import ctypes as C
    _, _Instance = host_lib()
    n = len(xy)
    edges = np.zeros((n, 2), dtype=np.int32)
    hi = _Instance()
    hi.params.time_limit = time_limit; hi.params.integer_cost = 1; hi.params.seed = 123; hi.params.perf_prof = 1
    hi.nodes = xy.ctypes.data_as(C.POINTER(C.c_double)); hi.num_nodes = n; hi.weight_type = wt
    hi.num_columns = n * (n - 1) // 2
    hi.solution.edges = edges.ctypes.data_as(C.POINTER(host_lib.Edge))
    return hi, edges


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

    # configs[4]: synthetic n = 5000, 128 random individuals (genetic.c:349-364, seed 123) each refined by alg_2opt -- through the C
    # host (north star: "host code stays in C"): HEU_2opt_population_multistart of libtsp_host.so draws the individuals on the
    # libc stream, refines the ones k % world == rank on this rank's device and, with world > 1, runs the collective epilogue
    # (all-reduce(min) + broadcast over RCCL through the C ABI, failure agreement included) itself
    import ctypes as C
    H, _Instance = host_lib()
    xy5 = rand_instance(5000)
    n5, P5 = 5000, 128
    hi, edges = host_instance(xy5, E.EUC_2D)
    costs5 = np.full(P5, np.nan)
    succ5 = np.zeros((P5, n5), dtype=np.int32)
    st5 = (E.Stats * P5)()
    bc, bk = C.c_double(0), C.c_int(-1)
    H.HEU_2opt_population_multistart.argtypes = [C.POINTER(_Instance), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                                 C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(E.Stats)]
    C.CDLL(None).srandom(123)
    if world > 1:   # the C host's rendezvous file must not be the one torch's launcher uses for anything: its own name per job
        os.environ.setdefault("TSP_RCCL_ID_FILE", "/tmp/tsp_bench_rccl_id.%d.%s" % (os.getuid(), os.environ.get("MASTER_PORT", "0")))
    t0 = time.perf_counter()
    rc5 = H.HEU_2opt_population_multistart(C.byref(hi), P5, rank, world, C.byref(bc), C.byref(bk), costs5.ctypes.data_as(C.POINTER(C.c_double)),
                                           succ5.ctypes.data_as(C.POINTER(C.c_int)), st5)
    t_mine = time.perf_counter() - t0
    t_all = wall(t_mine)
    gold = golden("oracle_vectors_big.json")["config5_rand5000_pop128"]["individuals"]
    best = min(gold, key=lambda r: (r["cost"], r["k"]))
    ok5 = rc5 == 0 and (bc.value, bk.value) == (best["cost"], best["k"])
    tour5 = fnv1a(edges[:, 1]) == best["hash"]
    mine = MS.shard_starts(P5, rank, world)
    local5 = all(costs5[k] == gold[k]["cost"] and fnv1a(succ5[k]) == gold[k]["hash"] and st5[k].evals == gold[k]["ev"] and
                 st5[k].moves == gold[k]["mv"] and st5[k].sweeps == gold[k]["sw"] for k in mine)
    all5 = wall(0.0 if local5 else 1.0) == 0.0        # max over ranks of "some local individual differs"
    res["config5_rand5000_population128_2opt"] = {
        "individuals": P5, "individuals_per_rank": len(mine), "wall_s": t_all, "shard_s_per_rank": per_rank(t_mine),
        "entry": "HEU_2opt_population_multistart(inst, 128, rank, world, ...) of libtsp_host.so (C host: libc draws, fitness, alg_2opt on the "
                 "shard, RCCL epilogue through tsp_dev_multistart_* when world > 1); wall_s includes the 1.28 M libc draws and the device "
                 "instance the C host creates for itself",
        "collectives": "none (1 GPU)" if world == 1 else "all_reduce(min) int64 + broadcast 4n bytes over RCCL, issued by the C host (tsp_host_multistart_epilogue)",
        "best_cost": bc.value, "best_individual": bk.value, "golden_best": [int(best["cost"]), best["k"]],
        "rank0_reference_equivalent_evals": int(sum(st5[k].evals for k in mine)),
        "winner_is_the_golden_winner": bool(ok5), "winner_tour_matches_golden": bool(tour5),
        "every_individual_on_every_rank_matches_golden_cost_tour_and_counters": bool(all5)}
    H.tsp_host_shutdown()
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
    ap.add_argument("--no-variants", action="store_true", help="skip the product's own descents (time_to_local_optimum): keeps a profile of the timed kernel clean")
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
    tours = E.Tours(inst, 1)                 # the product's default engines (time_to_local_optimum below)
    tours.upload(succ0[0], obj0[0])
    # the timed workload: the same instance with every bound off -- every delta expression of every sweep is executed
    os.environ["TSP_NO_FILTER"] = "1"
    inst_x = E.Instance(ctx, xy, wt, 1)
    tours_x = E.Tours(inst_x, 1)
    del os.environ["TSP_NO_FILTER"]
    tours_x.upload(succ0[0], obj0[0])        # resident in HBM from here on (and remembered as the reset point)
    kernel_x = tours_x.describe(E.BEST)
    pairs_per_sweep = N_NODES * (N_NODES - 1) // 2 - N_NODES
    dev_ms = []

    def one_descent():
        tours_x.reset()                      # device-to-device restore of the start tour
        rc, done = tours_x.run_engine(E.BEST, engine=E.ENGINE_GRID)
        assert rc == 0 and done
        dev_ms.append(tours_x.device_ms())   # HIP events on the engine's stream around the run's launches (no wait, no copy)

    for _ in range(args.warmup):
        one_descent()
    barrier_sync()
    del dev_ms[:]
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

    s_fin, o_fin, st_fin = tours_x.download()
    st = st_fin[0]
    # every rank ran `steps` descents of its own start; totals over the job (counters are per descent, identical each step)
    tot = {k: float(st[k]) for k in ("evals", "sweeps")}
    if dist is not None:
        import torch
        v = torch.tensor([tot[k] for k in sorted(tot)], dtype=torch.float64, device=device)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        tot = dict(zip(sorted(tot), [float(x) for x in v.tolist()]))
    value = tot["evals"] * args.steps / elapsed          # delta expressions executed per second, whole job
    launch_ms = float(np.sum(dev_ms)) / max(1, int(st["sweeps"]) * len(dev_ms))   # mean of one sweep's launches (k_move_pos + k_exh)

    # multi-start epilogue of the timed workload: one RCCL all-reduce(min) of the packed (true cost, rank)
    cost_now, _, packed = tours_x.best(true_cost=True)
    best_cost, best_rank = int(cost_now), rank
    if dist is not None:
        best_cost, best_rank = MS.allreduce_best(MS.pack(int(cost_now), rank), device=device)
        if comm is not None:   # the same reduction through the C ABI's communicator: both must agree
            c_cost, c_rank = MS.unpack(comm.allreduce_min(MS.pack(int(cost_now), rank)))
            dist_info["c_abi_allreduce_agrees_with_torch"] = bool((c_cost, c_rank) == (best_cost, best_rank))

    ops_per_launch = pairs_per_sweep * OPS_EXACT
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
                        "resident in HBM, EVERY delta expression of every sweep executed (SURVEY.md 8(d)'s roofline run)",
            "n": N_NODES, "starts_per_gpu": 1, "pairs_per_sweep": pairs_per_sweep, "sweeps_per_step_rank0": int(st["sweeps"]),
            "step": "device-to-device restore of the start tour + %d sweeps, each: %s, arg-min, segment reversal" % (int(st["sweeps"]), kernel_x),
            "parallelism": "multi-start x%d (one start per GPU, no data-path collective)" % world,
        },
        "value_definition": "delta expressions (src/tabusearch.c:150) EXECUTED per second over the timed region, whole job: every one of the "
                            "49 985 000 non-adjacent pairs of every sweep gets its exact delta; nothing is decided by a bound",
        "evals": {"per_step_rank0": {"delta_expressions": int(st["evals"]), "sweeps": int(st["sweeps"]), "moves": int(st["moves"])},
                  "delta_per_s": value, "us_per_sweep_wall": 1e6 * elapsed / args.steps / max(1, int(st["sweeps"]))},
        "multistart_best": dict({"cost": best_cost, "rank": best_rank,
                                 "collective": ("all_reduce(min) int64 over RCCL (torch.distributed and the C ABI's "
                                                "tsp_dev_multistart_allreduce)") if dist is not None else "none (1 GPU)"},
                                **dist_info),
        # the dominant kernel of the timed region, measured live: HIP events on the engine's stream over the timed descents
        "roofline": {
            "kernel": "tsp::k_exh<EUC_2D integer-coordinate variant, RJ> (+ tsp::k_move_pos, 4.7 us of the pair): one launch pair = one sweep = "
                      "49 985 000 delta expressions; " + kernel_x,
            "bound": "valu",
            "launch_ms": launch_ms, "launches_timed": int(st["sweeps"]) * len(dev_ms),
            "work_per_launch": {"delta_expressions": pairs_per_sweep, "valu_issue_slots_per_delta": SLOTS_EXH,
                                "what": "what the kernel issues per delta expression: ONE exact distance (2 sub, mul, fma, v_sqrt_f64 = 4 issue "
                                        "slots, add, floor, fma, compare, convert, add-with-carry = 14 slots) + 2.75 slots of integer sums / minimum / "
                                        "compare -- 55 VALU instructions per 4 pairs (csrc/two_opt_exh.hpp, ISA)"},
            "achieved": pairs_per_sweep * SLOTS_EXH / (launch_ms * 1e-3) / 1e12, "peak": FP64_LANE_OPS_PEAK / 1e12,
            "unit": "T lane-instruction slots/s (peak = fp64 vector 78.6 TFLOP/s / 2 = 39.3 T lane-instructions/s at 2.4 GHz)",
            "frac": pairs_per_sweep * SLOTS_EXH / (launch_ms * 1e-3) / FP64_LANE_OPS_PEAK,
            "survey_8d_model": {
                "ops_per_delta": OPS_EXACT, "ops_per_launch": ops_per_launch,
                "achieved": ops_per_launch / (launch_ms * 1e-3) / 1e12, "frac": ops_per_launch / (launch_ms * 1e-3) / FP64_LANE_OPS_PEAK,
                "note": "SURVEY 8(d) prices a delta at 35 fp64 operations = FOUR distances.  In tour-position order the second new edge of pair "
                        "(p, q) is the first new edge of pair (p + 1, q + 1): a sweep needs n^2 / 2 distinct distances, not 2 n^2, and k_exh "
                        "computes each once (every delta is still formed and compared exactly).  Priced by the 35-operation model the kernel "
                        "therefore reaches or passes the peak -- which says that the model over-counts the necessary work, not that the VALUs "
                        "do the impossible; `frac` above is the kernel's real utilisation of the issue port.  Round 3's tiled sweep (two "
                        "distances per delta executed) stood at 0.50 by this model"},
            "operand_bandwidth": {"note": "SURVEY 8(d) also asks for evals/s x 72 B against the HBM peak: with the operands on chip it is not a "
                                          "roofline (the tour is read once per sweep: 240 KB), reported for completeness",
                                  "GBps": value * 72.0 / 1e9, "hbm_peak_GBps": HBM_PEAK_GBS},
            "rocprof": "profiles/r04_kernel_stats.csv, profiles/r04_pmc_sq_wave_counters.json, profiles/r04_pmc_FETCH_SIZE.json, profiles/r04_pmc_WRITE_SIZE.json",
        },
        "time_to_local_optimum_s": elapsed / args.steps,
    }
    tj = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tj):
        with open(tj) as f:
            out["roofline"]["traffic"] = json.load(f).get("r04_exhaustive_sweep_n10000_hbm_bytes_per_launch")
    else:
        out["roofline"]["traffic"] = None

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

    def product_section():
        # time-to-local-optimum of the PRODUCT: the default engine for resident tours (TSP_ENGINE_AUTO -> CLUSTER: 256 workgroups, one
        # descent = one launch) takes the same decisions as the exhaustive descent above -- same final tour -- but decides most pairs
        # 64 x 64 at a time by bounds instead of executing their delta
        ms, wl = [], []
        for r in range(6):
            tours.reset()
            ctx.synchronize()
            t1 = time.perf_counter()
            rc, done = tours.run_engine(E.BEST, engine=E.ENGINE_AUTO)
            wl.append(time.perf_counter() - t1)
            assert rc == 0 and done
            ms.append(tours.device_ms())
        s_p, o_p, st_p = tours.download()
        st2 = st_p[0]
        big = golden("oracle_vectors_big.json")["rand10000_best"]["final"]
        kernel_ms = float(np.mean(ms[1:]))
        sweeps = st2["sweeps"]
        ops = (st2["lane_pairs"] * OPS_TIER0_F32 + max(0, st2["tier1_pairs"]) * OPS_TIER1 +
               max(0, st2["exact_pairs"]) * OPS_EXACT + max(0, st2["staged_recs"]) * OPS_STAGED +
               sweeps * (inst_groups(N_NODES) * (inst_groups(N_NODES) + 1) // 2) * OPS_BOXTEST)
        traffic = None
        tj2 = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tj2):
            with open(tj2) as f:
                traffic = json.load(f).get("r03_cluster_descent_n10000_hbm_bytes_per_launch")
        out["time_to_local_optimum"] = {"best_improvement_alg_2opt_tabu": {
            "engine": "TSP_ENGINE_AUTO -> CLUSTER: tsp::k_cluster_two_opt<EUC_2D integer-coordinate variant, BEST, float replica, sorted scan>, "
                      "one launch = the whole descent on 256 workgroups",
            "device_ms": kernel_ms, "wall_ms": 1e3 * float(np.mean(wl[1:])), "sweeps": int(sweeps), "us_per_sweep": 1e3 * kernel_ms / sweeps,
            "final_cost": float(o_p[0]), "moves": int(st2["moves"]),
            "final_tour_matches_golden": bool(fnv1a(s_p[0]) == big["hash"] and o_p[0] == big["cost"] and st2["sweeps"] == big["stats"]["sweeps"]
                                              and st2["evals"] == big["stats"]["evals"] and st2["moves"] == big["stats"]["moves"]),
            "pairs_decided_per_s": st2["evals"] / (kernel_ms * 1e-3),
            "executed_per_descent": {"lane_pairs_tier0": int(st2["lane_pairs"]), "tier1_pairs": int(st2["tier1_pairs"]),
                                     "exact_delta_expressions": int(st2["exact_pairs"]), "staged_node_records": int(st2["staged_recs"])},
            "valu_frac_of_executed_work": ops / (kernel_ms * 1e-3) / FP64_LANE_OPS_PEAK,
            "hbm_bytes_per_launch": traffic,
            "note": "a chain of latencies by design (LDS gathers, one all-to-all exchange through L2 per sweep, block barriers), DESIGN.md 4.8: "
                    "it decides 5.9e12 pairs per second while executing ~1 k delta expressions per sweep -- which is why it is not the "
                    "evals/s headline",
            "rocprof": "profiles/r03_kernel_stats.csv (12.11 ms mean), profiles/r03_pmc_sq_wave_counters.json"}}

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
        out["time_to_local_optimum"]["first_improvement_alg_2opt"] = {
            "engine": "TSP_ENGINE_AUTO -> CLUSTER: tsp::k_cluster_two_opt<EUC_2D integer-coordinate variant, FIRST, float replica, plain | rank order>: "
                      "one descent = a few launches (hand-overs between the two variants); device_ms = HIP events on the engine's stream "
                      "around all of them (rocprof: profiles/r03_kernel_stats_first.csv)",
            "device_ms": kms, "steps": int(stf["steps"]), "us_per_step": 1e3 * kms / max(1, stf["steps"]),
            "sweeps": int(stf["sweeps"]), "reference_evals": int(stf["evals"]), "moves": int(stf["moves"]),
            "reference_counters_match": bool((stf["sweeps"], stf["evals"], stf["moves"]) == (10, 499850987, 2704) and of[0] == 77370387),
            "valu_frac_of_executed_work": ops / (kms * 1e-3) / FP64_LANE_OPS_PEAK,
            "reference_equivalent_evals_per_s": stf["evals"] / (kms * 1e-3),
            "sweep_that_finds_nothing_us_per_call": 1e6 * float(np.mean(t_nohit[2:])),
            "note": "a chain of 2 704 dependent moves: 5.8 us per step, 3.1 of it one all-to-all exchange (DESIGN.md 4.8, 4.8c)"}

    if rank == 0 and not args.no_variants:
        guarded("time_to_local_optimum", product_section)
        if "error" not in out.get("time_to_local_optimum", {"error": 1}):
            guarded("time_to_local_optimum_first", first_section)

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
        out.setdefault("time_to_local_optimum", {})["host_tour_in_host_tour_out"] = extras
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
                    d = out["distance_matrix_build"][k]
                    # the roofline figure: measured HBM write traffic over the profiler's kernel time.  The HIP-event rate of
                    # back-to-back launches reads up to 7 % higher (head / tail overlap of consecutive launches, not bandwidth)
                    d.update({"rocprof_mean_us": r3[k]["rocprof_mean_us"], "write_size_bytes": r3[k]["write_size_bytes"],
                              "hip_event_frac_this_run": d["frac"],
                              "frac": r3[k]["write_size_bytes"] / (r3[k]["rocprof_mean_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS})
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
        # tabu() itself (tabusearch.c:188-320, step policy) through the C host: tsp_host_tabu of libtsp_host.so keeps tour, stamps and
        # incumbent on the device, draws on the host's libc stream, and queues chains of iterations per wait for the device
        # (tsp_dev_tours_tabu_iterations_ex: the iterations of a chain run inside one launch).  Rate = iterations / the seconds the
        # call spent in its iteration loop (tsp_host_last_driver_loop_seconds): the initial HEU_2opt_greedy_iter (:200) is not in it.
        import ctypes as C
        H, _Instance = host_lib()
        H.tsp_host_tabu.argtypes = [C.POINTER(_Instance), C.c_int, C.c_longlong]

        H.tsp_host_last_driver_loop_seconds.restype = C.c_double

        def tabu_rate(chain, iters):
            if chain is None:
                os.environ.pop("TSP_TABU_CHAIN", None)
            else:
                os.environ["TSP_TABU_CHAIN"] = str(chain)
            hi, _edges = host_instance(xy, wt, time_limit=3600)
            C.CDLL(None).srandom(123)
            H.tsp_host_tabu(C.byref(hi), 0, iters)
            secs = H.tsp_host_last_driver_loop_seconds()    # the iteration loop alone: the initial HEU_2opt_greedy_iter (:200) is 0.8 s of its own
            os.environ.pop("TSP_TABU_CHAIN", None)
            return {"iterations": iters, "seconds": secs, "iterations_per_s": iters / secs, "incumbent": hi.solution.obj_best}
        iters = 1500
        tabu_rate(None, 60)                      # warm
        chained = tabu_rate(None, iters)
        H.tsp_host_shutdown()                    # (switches are read when a device instance is made: drop the cached one)
        os.environ["TSP_TABU_INKERNEL"] = "0"
        tabu_rate(None, 60)
        queued = tabu_rate(None, iters)
        os.environ.pop("TSP_TABU_INKERNEL", None)
        H.tsp_host_shutdown()
        tabu_rate(None, 60)
        single = tabu_rate(1, iters)
        res["tabu_iterations_on_resident_state"] = dict(
            chained, driver="tsp_host_tabu(inst, step policy, cap on the iterations) of libtsp_host.so, seed 123: chains of up to 128 iterations "
                            "INSIDE one CLUSTER launch (incumbent, the kick's trials, the kick and the next descent in the kernel; "
                            "tsp_dev_tours_tabu_iterations_ex)",
            queued_launches=dict(queued, what="TSP_TABU_INKERNEL=0: the launches of a chain queued back to back, one small kernel between two"),
            one_iteration_per_wait=single,
            same_incumbent_all_ways=bool(chained["incumbent"] == single["incumbent"] == queued["incumbent"]))
        H.tsp_host_shutdown()
        out["alg_2opt_tabu_with_a_list"] = res

    if rank == 0 and world == 1 and not args.no_extras:
        guarded("time_to_local_optimum_host_tours", extras_section)
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
                "source": "tools/pop_time.py, tools/shard_time.py (DESIGN.md section 5): 1.31 x on 8 GPUs for configs[4], 1.2 x for configs[3]; "
                          "configs[4] runs through the C host (HEU_2opt_population_multistart), configs[3] through the Python launcher on the C communicator"}}
        out["all_checks_ok"] = bool(out.get("parity", {}).get("final_tour_matches_golden", True) and
                                    all(v for d in out["other_configs"].values() for k, v in d.items()
                                        if k.startswith(("winner_", "every_"))))

    def cpu_section():
        cores = max(1, min(16, len(os.sched_getaffinity(0))))   # a 1-GPU box's CPU share
        base = cpu_baselines(xy, wt, succ0[0], obj0[0], cores)
        out["cpu_baseline"] = {
            "value": base["best_improvement"]["evals_per_s"], "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": base["best_improvement"]["sample"] + " by oracle/tsp_oracle.c (gcc -O2), one thread, %.1f s: every "
                      "delta expression executed (the same unit as `value`); the reference is single-threaded and cannot be built here "
                      "(needs cplex.h)" % base["best_improvement"]["seconds"],
            "others": {k: v for k, v in base.items() if k != "best_improvement"},
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        guarded("cpu_baseline", cpu_section)

    tours.close()
    tours_x.close()
    inst_x.close()
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
