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
