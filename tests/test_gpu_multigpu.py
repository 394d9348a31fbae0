"""GPU: the multi-start epilogue across GPUs -- the C ABI's RCCL communicator (tsp_dev_comm_*, tsp_dev_multistart_*), the C host's
HEU_2opt_grasp_multistart / tsp_host_multistart_gpus on it, the `tsp` CLI's 2OPT_GRASP_MULTI, and `bench.py --gpus N` starting its
own ranks.  A 1-GPU box can only form single-rank communicators (RCCL refuses two ranks on one device): these tests drive every
call of the path through RCCL with world = 1; the world = 2 logic runs over gloo in tests/test_cpu_abi_and_multistart.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import golden, INSTANCES

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_comm_single_rank_allreduce_and_broadcast():
    from tsp_optimization_amd import engine as E
    from tsp_optimization_amd import multistart as M
    ctx = E.Context(0)
    uid = E.comm_unique_id()
    assert len(uid) == E.COMM_ID_BYTES and any(uid)
    comm = E.Comm(ctx, 1, 0, uid)
    assert comm.rccl_version() > 20000
    p = M.pack(28998, 122)
    assert comm.allreduce_min(p) == p and comm.allreduce_min(M.NO_RESULT) == M.NO_RESULT
    tour = np.arange(532, dtype=np.int32)[::-1].copy()
    assert (comm.bcast_tour(0, tour) == np.arange(532, dtype=np.int32)[::-1]).all()
    # the launcher with the C communicator in place of torch.distributed
    table = golden("oracle_vectors.json")["att532_multistart256"]
    out = M.run_sharded(lambda ids: ([table[k]["opt_true"] for k in ids], np.stack([np.full(532, k, dtype=np.int32) for k in ids])),
                        256, 532, 0, 1, comm=comm)
    assert (out["cost"], out["start"]) == (28998, 122) and (out["tour"] == 122).all()
    with pytest.raises(M.UnpackableCost):
        M.run_sharded(lambda ids: ([0.5] * len(ids), np.zeros((len(ids), 4), dtype=np.int32)), 3, 4, 0, 1, comm=comm)
    comm.close()
    ctx.close()


def _cli(args, env=None):
    from tsp_optimization_amd.build import lib_path
    r = subprocess.run([lib_path("tsp")] + args, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("how", ["one_process", "forced_comm_rank", "threads_per_gpu"])
def test_cli_grasp_multistart_256_finds_the_reference_winner(how):
    """BASELINE configs[3] through the C host and the CLI: 256 GRASP starts of att532 (seed 123) + alg_2opt each -> best true cost
    28998 (start 122).  forced_comm_rank: the one-process-per-GPU path with a single rank -- RCCL id through the
    id file, ncclCommInitRank, all-reduce(min), broadcast.  threads_per_gpu: -gpus 1 = one thread, context and instance per
    GPU, ncclCommInitAll and the grouped collectives."""
    f = os.path.join(INSTANCES, "att532.tsp")
    args = ["-f", f, "-method", "2OPT_GRASP_MULTI", "-starts", "256", "-seed", "123", "--perfprof", "-verbose", "-1"]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if how == "forced_comm_rank":
        env["TSP_FORCE_COMM"] = "1"
    if how == "threads_per_gpu":
        args += ["-gpus", "1"]
    assert _cli(args, env) == "28998.00"


def test_host_multistart_gpus_returns_winner_tour_and_shard_times():
    import ctypes as C
    from tsp_optimization_amd.build import lib_path
    from oracle import oracle as O
    from test_gpu_host_cli import HostInstance, Instance
    L = C.CDLL(lib_path("libtsp_host.so"))
    L.tsp_host_multistart_gpus.argtypes = [C.POINTER(Instance), C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                           C.POINTER(C.c_double)]
    h = HostInstance("att532")
    cost, start, secs = C.c_double(0), C.c_int(-1), (C.c_double * 1)()
    C.CDLL(None).srandom(123)
    assert L.tsp_host_multistart_gpus(C.byref(h.c), 64, 1, C.byref(cost), C.byref(start), secs) == 0
    table = golden("oracle_vectors.json")["att532_multistart256"]
    want = min(range(64), key=lambda k: (table[k]["opt_true"], k))
    assert (cost.value, start.value) == (table[want]["opt_true"], want) and secs[0] > 0
    assert O.succ_cost(h.xy, h.wt, h.succ) == cost.value == h.obj    # the tour that came back over the broadcast is the winner's
    L.tsp_host_shutdown()


def test_bench_starts_its_own_ranks_and_runs_the_collectives_over_rccl():
    """`python bench.py --gpus 1` with TSP_BENCH_FORCE_DIST=1: the parent spawns the rank itself, the rank joins an RCCL process
    group AND forms the C ABI's communicator, and configs[3] / [4] run their all-reduce(min) + broadcast through libtsp_hip.so."""
    env = dict(os.environ, TSP_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-variants"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["parity"]["final_tour_matches_golden"] and out["all_checks_ok"]
    ms = out["multistart_best"]
    assert ms["rccl_ranks_seen"] == 1 and "RCCL" in ms["collective"] and ms["c_abi_allreduce_agrees_with_torch"]
    assert ms["c_abi_comm"].startswith("tsp_dev_comm_init_rank over RCCL")
    c4 = out["other_configs"]["config4_att532_grasp256_2opt"]
    assert "tsp_dev_multistart_allreduce" in c4["collectives"] and c4["winner_is_the_reference_winner"]
    assert len(c4["refine_s_per_rank"]) == 1


# ---- round 4: the population job of BASELINE configs[4] behind the C host, --fcost through the epilogue, failure agreement ----

def _host_lib():
    import ctypes as C
    from tsp_optimization_amd.build import lib_path
    from tsp_optimization_amd import engine as E
    from helpers import Instance
    L = C.CDLL(lib_path("libtsp_host.so"))
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.tsp_host_population_gpus.argtypes = [C.POINTER(Instance), C.c_int, C.c_int, dp, ip, dp, dp, ip, C.POINTER(E.Stats)]
    L.HEU_2opt_population_multistart.argtypes = [C.POINTER(Instance), C.c_int, C.c_int, C.c_int, dp, ip, dp, ip, C.POINTER(E.Stats)]
    L.HEU_2opt_grasp_multistart.argtypes = [C.POINTER(Instance), C.c_int, C.c_int, C.c_int, dp, ip]
    L.tsp_host_multistart_gpus.argtypes = [C.POINTER(Instance), C.c_int, C.c_int, dp, ip, dp]
    L.tsp_host_multistart_epilogue.argtypes = [C.POINTER(Instance), C.c_int, C.c_int, C.c_int, dp, ip]
    L.tsp_host_multistart_last_error.restype = C.c_char_p
    L.tsp_host_genetic_gpus.argtypes = [C.POINTER(Instance), C.c_longlong, C.c_double, C.c_int]
    L.calc_dist.restype = C.c_double
    L.calc_dist.argtypes = [C.c_int, C.c_int, C.POINTER(Instance)]
    return L


@pytest.mark.parametrize("how", ["threads_per_gpu", "forced_comm_rank", "one_process"])
def test_population_128_through_the_c_host_equals_the_golden_table(how, monkeypatch):
    """BASELINE configs[4] behind the C host (north star: "genetic population members shard ... across the GPUs", "host code
    stays in C"): 128 random individuals of rand5000 drawn as random_generation draws them (genetic.c:349-364, seed 123), each
    refined by alg_2opt (:426-443) -- every row of the oracle's 128-row table (cost, tour hash, sweeps, evaluations, moves)
    and the winner (57357536, individual 9), whose tour is the one that came back through RCCL's broadcast.
    threads_per_gpu: tsp_host_population_gpus(gpus = 1) = thread + context per GPU, ncclCommInitAll, grouped collectives;
    forced_comm_rank: HEU_2opt_population_multistart with TSP_FORCE_COMM=1 = the one-process-per-GPU path with one rank
    (id file, ncclCommInitRank, all-reduce(min), broadcast); one_process: no collective."""
    import ctypes as C
    from oracle import oracle as O
    from tsp_optimization_amd import engine as E
    from helpers import HostInstance
    g = golden("oracle_vectors_big.json")["config5_rand5000_pop128"]
    L = _host_lib()
    h = HostInstance("rand5000")
    n, P = h.n, g["population"]
    costs, succ, st = np.zeros(P), np.zeros((P, n), dtype=np.int32), (E.Stats * P)()
    cost, who, secs = C.c_double(0), C.c_int(-1), (C.c_double * 1)()
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    C.CDLL(None).srandom(g["seed"])
    if how == "threads_per_gpu":
        rc = L.tsp_host_population_gpus(C.byref(h.c), P, 1, C.byref(cost), C.byref(who), secs, costs.ctypes.data_as(dp), succ.ctypes.data_as(ip), st)
        assert secs[0] > 0
    else:
        if how == "forced_comm_rank":
            monkeypatch.setenv("TSP_FORCE_COMM", "1")
        else:
            monkeypatch.delenv("TSP_FORCE_COMM", raising=False)
        for k in ("WORLD_SIZE", "RANK", "MASTER_PORT"):
            monkeypatch.delenv(k, raising=False)
        rc = L.HEU_2opt_population_multistart(C.byref(h.c), P, 0, 1, C.byref(cost), C.byref(who), costs.ctypes.data_as(dp), succ.ctypes.data_as(ip), st)
    assert rc == 0
    for k, row in enumerate(g["individuals"]):
        assert costs[k] == row["cost"] and O.fnv1a(succ[k]) == row["hash"], k
        assert (st[k].sweeps, st[k].evals, st[k].moves, st[k].reversed) == (row["sw"], row["ev"], row["mv"], row["reversed"]), k
    best = min(g["individuals"], key=lambda r: (r["cost"], r["k"]))
    assert (best["cost"], best["k"]) == (57357536, 9)
    assert (cost.value, who.value) == (57357536, 9) and h.obj == 57357536 and O.fnv1a(h.succ) == best["hash"]
    assert (h.edges[:, 0] == np.arange(n)).all()
    L.tsp_host_shutdown()


def _rand_tsplib(tmp_path, n):
    from helpers import rand_instance
    xy = rand_instance(n)
    f = tmp_path / ("rand%d.tsp" % n)
    f.write_text("NAME : rand%d\nTYPE : TSP\nDIMENSION : %d\nEDGE_WEIGHT_TYPE : EUC_2D\nNODE_COORD_SECTION\n" % (n, n) +
                 "".join("%d %d %d\n" % (k + 1, x, y) for k, (x, y) in enumerate(xy)) + "EOF\n")
    return str(f)


@pytest.mark.parametrize("how", ["one_process", "forced_comm_rank", "threads_per_gpu"])
def test_cli_population_multi_prints_the_golden_winner(how, tmp_path):
    """`tsp -f rand5000.tsp -method 2OPT_POP_MULTI -starts 128 -seed 123 [-gpus 1]`: the CLI method of configs[4], three ways."""
    args = ["-f", _rand_tsplib(tmp_path, 5000), "-method", "2OPT_POP_MULTI", "-starts", "128", "-seed", "123", "--perfprof", "-verbose", "-1"]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TSP_FORCE_COMM"):
        env.pop(k, None)
    if how == "forced_comm_rank":
        env["TSP_FORCE_COMM"] = "1"
    if how == "threads_per_gpu":
        args += ["-gpus", "1"]
    assert _cli(args, env) == "57357536.00"


@pytest.mark.parametrize("how", ["threads_per_gpu", "forced_comm_rank"])
def test_fcost_multistart_goes_through_the_two_reduction_epilogue(how, monkeypatch):
    """--fcost (src/utility.c:285): costs are doubles, `< bestobj` (heuristics.c:534) still decides.  att532 x 256 GRASP starts
    + alg_2opt with integer_cost = 0 through RCCL: min of the double, then min of the start among its holders, then the
    broadcast -- cost bit-identical to the oracle's float table (start 67), tour = that row's."""
    import ctypes as C
    from oracle import oracle as O
    from helpers import HostInstance
    g = golden("oracle_vectors_fcost.json")
    L = _host_lib()
    h = HostInstance("att532", integer_cost=0)
    cost, start = C.c_double(0), C.c_int(-1)
    C.CDLL(None).srandom(123)
    if how == "threads_per_gpu":
        assert L.tsp_host_multistart_gpus(C.byref(h.c), 256, 1, C.byref(cost), C.byref(start), None) == 0
    else:
        monkeypatch.setenv("TSP_FORCE_COMM", "1")
        for k in ("WORLD_SIZE", "RANK", "MASTER_PORT"):
            monkeypatch.delenv(k, raising=False)
        assert L.HEU_2opt_grasp_multistart(C.byref(h.c), 256, 0, 1, C.byref(cost), C.byref(start)) == 0
    best = g["best"]
    row = g["starts"][best["k"]]
    assert (cost.value.hex(), start.value) == (best["cost_hex"], best["k"]) and best["k"] == 67
    assert O.fnv1a(h.succ) == row["hash"] and h.obj == cost.value
    assert cost.value != int(cost.value)                             # really a cost the packed word cannot carry
    L.tsp_host_shutdown()


def test_cli_fcost_multistart_prints_the_float_winner():
    f = os.path.join(INSTANCES, "att532.tsp")
    env = dict(os.environ, TSP_FORCE_COMM="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = _cli(["-f", f, "-method", "2OPT_GRASP_MULTI", "-starts", "256", "-seed", "123", "--fcost", "--perfprof", "-verbose", "-1"], env)
    assert out == "%0.2f" % golden("oracle_vectors_fcost.json")["best"]["cost"]


def test_epilogue_over_rccl_carries_a_failed_shard_to_every_rank(monkeypatch):
    """The failure agreement on the real transport (one rank: what a 1-GPU box can form): a shard that failed is contributed
    to ncclAllReduce as the value that wins the minimum; the epilogue returns TSP_HOST_E_PEER instead of exiting before the
    collective.  (World 2 over gloo: tests/test_cpu_epilogue.py.)"""
    import ctypes as C
    from helpers import HostInstance
    L = _host_lib()
    for k in ("WORLD_SIZE", "RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    for ic in (1, 0):
        h = HostInstance("berlin52", integer_cost=ic)
        assert L.calc_dist(0, 1, C.byref(h.c)) > 0                    # opens this process's device context
        h.set_tour(np.roll(np.arange(h.n, dtype=np.int32), -1), 123.0)
        best, bk = C.c_double(123.0), C.c_int(3)
        assert L.tsp_host_multistart_epilogue(C.byref(h.c), 0, 1, 0, C.byref(best), C.byref(bk)) == 0
        assert (best.value, bk.value, h.obj) == (123.0, 3, 123.0)
        assert L.tsp_host_multistart_epilogue(C.byref(h.c), 0, 1, -2, C.byref(best), C.byref(bk)) == -7
        assert b"shard failed" in L.tsp_host_multistart_last_error()
    L.tsp_host_shutdown()


def test_genetic_mutation3_batch_over_gpus_matches_oracle():
    """tsp_host_genetic_gpus(..., gpus = 1): the offspring that drew mutation 3 (genetic.c:426-443) are refined by per-GPU
    worker threads with their own context and instance (the CLI's `-method GENETIC -gpus G`); same incumbent as the oracle GA."""
    import ctypes as C
    from oracle import oracle as O
    from helpers import HostInstance
    L = _host_lib()
    h = HostInstance("berlin52")
    h.c.params.time_limit = 600
    O.srandom(123)
    rc = L.tsp_host_genetic_gpus(C.byref(h.c), 25, 0.5, 1)
    O.srandom(123)
    es, eo = O.genetic(h.xy, h.wt, 25, two_opt_prob=0.5)
    assert rc == 0 and h.obj == eo and (h.succ == es).all()
    L.tsp_host_shutdown()
