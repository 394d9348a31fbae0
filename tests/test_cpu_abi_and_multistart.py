"""CPU (no GPU): the C-ABI libraries load and export every symbol the headers declare; the
multi-start sharding / packed all-reduce runs with world_size 2 over gloo; the host logic of the
binding refuses to run without a device instead of falling back."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from helpers import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return True


def _declared(header):
    txt = open(os.path.join(ROOT, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tsp_dev_\w+|tsp_host_\w+)\s*\(", txt)))


def test_libtsp_hip_exports_every_declared_symbol(built):
    from tsp_optimization_amd import engine as E
    L = C.CDLL(E.lib_path())
    names = _declared("include/tsp_hip.h")
    assert len(names) >= 29
    assert sorted(names) == sorted(E.EXPORTED)
    for n in names:
        assert hasattr(L, n), n


def test_libtsp_host_exports_the_reference_entry_points(built):
    from tsp_optimization_amd.build import lib_path
    L = C.CDLL(lib_path("libtsp_host.so"))
    for n in ["calc_dist", "greedy", "grasp", "HEU_greedy", "HEU_Greedy_iter", "HEU_Grasp", "HEU_Grasp_iter",
              "alg_2opt", "alg_2opt_tabu", "HEU_2opt_grasp", "HEU_2opt_grasp_iter", "HEU_2opt_greedy",
              "HEU_2opt_greedy_iter", "reverse_path", "copy_instance", "rand_choice", "x_udir_pos",
              "get_elapsed_time", "free_instance", "TSP_heuc", "parse_comand_line", "parse_instance",
              "export_tour", "fitness_batch", "HEU_2opt_grasp_multistart", "tsp_host_multistart_gpus", "tsp_host_multistart_shard",
              "HEU_2opt_population_multistart", "tsp_host_population_gpus", "tsp_host_population_shard",
              "tsp_host_multistart_epilogue", "tsp_host_multistart_last_error", "tsp_host_set_collectives",
              "tsp_host_rccl_id_file_state", "tsp_host_last_grasp_iter_starts", "tsp_host_random_lookahead", "tsp_host_last_driver_loop_seconds", "tsp_host_genetic_gpus", "tsp_host_genetic_ex",
              "tsp_host_vns", "tsp_host_tabu", "HEU_VNS", "HEU_Tabu_step", "HEU_Tabu_lin", "HEU_Tabu_rand", "HEU_Genetic",
              "HEU_extramileage", "HEU_2opt_extramileage", "kick",
              "tsp_host_last_stats", "tsp_host_shutdown"]:
        assert hasattr(L, n), n


def test_no_device_means_error_not_fallback(built):
    """In this container there is no GPU: opening a context must fail loudly."""
    from tsp_optimization_amd import engine as E
    if E.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(E.TspDeviceError):
        E.Context(0)


def test_cli_without_device_exits_with_error(built):
    from tsp_optimization_amd import engine as E
    if E.device_count() > 0:
        pytest.skip("a GPU is visible")
    from tsp_optimization_amd.build import lib_path
    r = subprocess.run([lib_path("tsp"), "-f", os.path.join(ROOT, "tests/golden/instances/berlin52.tsp"),
                        "-method", "2OPT_GREEDY", "-seed", "123", "--perfprof", "-verbose", "-1"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "[ERROR]" in r.stderr and r.stdout == ""


def test_cli_help_and_methods(built):
    from tsp_optimization_amd.build import lib_path
    r = subprocess.run([lib_path("tsp"), "--methods"], capture_output=True, text=True)
    assert r.returncode == 0 and "2OPT_GREEDY_ITER" in r.stdout
    r = subprocess.run([lib_path("tsp")], capture_output=True, text=True)
    assert r.returncode == 1 and "--help" in r.stdout            # src/utility.c:49-52
    r = subprocess.run([lib_path("tsp"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "-method <type>" in r.stdout


def test_reference_ctest_cases_for_the_parser(built):
    """The reference's own CTest cases (test/CMakeLists.txt:6-19): no args fails, --help and --v succeed, a
    missing file fails, a file whose header fields are shuffled still parses, a file without DIMENSION
    fails.  Without a GPU the shuffled file gets past the parser and stops at the device."""
    from tsp_optimization_amd.build import lib_path
    from tsp_optimization_amd import engine as E
    tsp = lib_path("tsp")
    inst = os.path.join(ROOT, "tests/golden/instances")
    run = lambda *a: subprocess.run([tsp] + list(a), capture_output=True, text=True)
    assert run().returncode == 1
    assert run("--help").returncode == 0
    r = run("--v")
    assert r.returncode == 0 and r.stdout.startswith("Version")
    r = run("-f", "hello.txt", "-method", "GREEDY")
    assert r.returncode == 1 and "Unable to open file" in r.stderr
    r = run("-f", os.path.join(inst, "fail_att48.tsp"), "-method", "GREEDY", "-verbose", "3")
    assert r.returncode == 1 and "unknown node" in r.stderr            # no DIMENSION before the coordinates
    r = run("-f", os.path.join(inst, "shuffled_prop_att48.tsp"), "-method", "GREEDY", "--perfprof")
    if E.device_count() == 0:
        assert r.returncode == 1 and "tsp_dev_open" in r.stderr        # parsed; no device to run on
    else:
        assert r.returncode == 0 and float(r.stdout) > 0


def test_oracle_parser_on_the_reference_ctest_files():
    from oracle import oracle as O
    inst = os.path.join(ROOT, "tests/golden/instances")
    xy, wt = O.parse_tsplib(os.path.join(inst, "shuffled_prop_att48.tsp"))
    ref, wt_ref = O.parse_tsplib(os.path.join(inst, "att48.tsp"))
    assert xy.shape == (48, 2) and (xy == ref).all() and wt_ref == O.ATT and wt == O.ATT
    with pytest.raises(ValueError):
        O.parse_tsplib(os.path.join(inst, "fail_att48.tsp"))


def test_pack_orders_by_cost_then_start():
    from tsp_optimization_amd import multistart as M
    assert M.unpack(M.pack(28998, 122)) == (28998, 122)
    assert M.pack(28998, 200) < M.pack(28999, 0)
    assert M.pack(28998, 5) < M.pack(28998, 6)
    assert M.local_best([], []) == M.NO_RESULT
    assert M.shard_starts(10, 1, 4) == [1, 5, 9] and M.owner_of(9, 4) == 1


def test_pack_rejects_costs_the_allreduce_cannot_carry(built):
    """--fcost / GEO costs are not integers: pack raises (no assert), try_pack returns the error value that wins the MIN so
    that every rank learns of it through the reduction itself; the C ABI's tsp_dev_multistart_pack applies the same rule."""
    from tsp_optimization_amd import multistart as M
    from tsp_optimization_amd import engine as E
    for bad in (1.5, -1.0, float(1 << 39), float("nan")):
        with pytest.raises(M.UnpackableCost):
            M.pack(bad, 3)
        with pytest.raises(ValueError):
            E.multistart_pack(bad, 3)
        assert M.try_pack(bad, 3) == M.PACK_ERROR
    with pytest.raises(M.UnpackableCost):
        M.pack(10.0, 1 << 24)
    assert M.PACK_ERROR < M.pack(0, 0) and M.local_best([7.0, 7.5], [0, 1]) == M.PACK_ERROR
    for c, k in ((28998, 122), (0, 0), ((1 << 39) - 1, (1 << 24) - 1)):
        assert E.multistart_pack(c, k) == M.pack(c, k)
    with pytest.raises(M.UnpackableCost):                      # single rank: raised, not asserted
        M.run_sharded(lambda ids: ([0.5] * len(ids), np.zeros((len(ids), 4), dtype=np.int32)), 3, 4)


WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from tsp_optimization_amd import multistart as M
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
table = json.load(open(os.path.join(sys.argv[1], "tests/golden/oracle_vectors.json")))["att532_multistart256"]
n = 532
calls = []
def refine(ids):
    # the engine of one rank, replaced by the golden table (oracle-generated): true cost per start, and a stand-in tour
    # that encodes the start id so that the broadcast can be checked
    calls.append(list(ids))
    return [table[k]["opt_true"] for k in ids], np.stack([np.full(n, k, dtype=np.int32) for k in ids])
out = M.run_sharded(refine, len(table), n, rank, world)       # the launcher bench.py drives on the GPUs
res = {"rank": rank, "n_mine": out["local_starts"], "cost": out["cost"], "start": out["start"],
       "tour_ok": bool((out["tour"] == out["start"]).all()), "ids_ok": calls == [M.shard_starts(len(table), rank, world)]}
# --fcost: the same launcher with costs the packed word cannot carry -> two reductions (min of the double, min of the start
# among its holders); table = the oracle's 256 starts of att532 with integer_cost = 0
ftab = json.load(open(os.path.join(sys.argv[1], "tests/golden/oracle_vectors_fcost.json")))["starts"]
fref = lambda ids: ([float.fromhex(ftab[k]["cost_hex"]) for k in ids], np.stack([np.full(n, k, dtype=np.int32) for k in ids]))
fo = M.run_sharded(fref, len(ftab), n, rank, world, integer_costs=False)
res["fcost"] = [fo["cost"].hex(), fo["start"], bool((fo["tour"] == fo["start"]).all())]
# ties between the ranks: the lowest start wins whatever rank holds it
tie = M.run_sharded(lambda ids: ([2.5 if k in (5, 2) else 9.75 for k in ids], np.stack([np.full(n, k, dtype=np.int32) for k in ids])),
                    8, n, rank, world, integer_costs=False)
res["fcost_tie"] = [tie["cost"], tie["start"], bool((tie["tour"] == 2).all())]
# a rank whose refine raises: the failure travels through the reduction, EVERY rank raises, none waits in a collective
def bad(ids):
    if rank == 1:
        raise RuntimeError("shard of rank 1 broke")
    return fref(ids)
for name, ic in (("fail_int", True), ("fail_f", False)):
    try:
        M.run_sharded(bad, len(ftab), n, rank, world, integer_costs=ic)
        res[name] = "returned"
    except Exception as e:
        res[name] = type(e).__name__
print(json.dumps(res))
dist.destroy_process_group()
'''


def test_multistart_allreduce_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    exp = golden("survey_appendix_b.json")["att532"]["multistart256"]
    table = golden("oracle_vectors.json")["att532_multistart256"]
    assert table[exp["best_start"]]["opt_true"] == exp["best_true"]
    for o in outs:
        assert (o["cost"], o["start"]) == (exp["best_true"], exp["best_start"])   # 28998 at start 122
        assert o["n_mine"] == 128 and o["tour_ok"] and o["ids_ok"]                # every rank holds the winner's tour
    fbest = golden("oracle_vectors_fcost.json")["best"]
    for r, o in enumerate(outs):
        assert o["fcost"] == [fbest["cost_hex"], fbest["k"], True]                # bit-identical double, start 67
        assert o["fcost_tie"] == [2.5, 2, True]
        assert o["fail_int"] == ("RuntimeError" if r == 1 else "UnpackableCost")  # both raised; nobody hung
        assert o["fail_f"] == "RuntimeError"


def test_bench_self_launcher_world2_gloo():
    """`python bench.py --gpus 2` with no launcher around it starts the two ranks itself (children spawned before anything
    touches a GPU), relays rank 0's JSON line as its own LAST stdout line and returns the children's status.
    TSP_BENCH_SELFTEST=1 replaces the device work by the golden table and the backend by gloo."""
    import json
    env = dict(os.environ, TSP_BENCH_SELFTEST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, lines                               # everything else went to stderr
    out = json.loads(lines[0])
    assert out == {"selftest": True, "n_gpus": 2, "ranks_seen": 2, "cost": 28998, "start": 122, "tour_ok": True}
    assert "[rank 1] rank 1 done" in r.stderr and "[rank 0] not the json line" in r.stderr
    # a rank that fails makes the parent fail with its code
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       env=dict(env, TSP_BENCH_SELFTEST_FAIL_RANK="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 3


def test_bench_parent_never_touches_the_gpu_or_execs():
    """The self-launching parent must not initialise HIP (a later exec / fork from such a process takes the box down) and
    must not exec: checked on the source -- no os.exec*, and nothing but the standard library is imported before the
    children are started."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src
    body = src[src.index("def launch_ranks"):src.index("def launcher_selftest_child")]
    assert "import torch" not in body and "tsp_optimization_amd" not in body and "ctypes" not in body
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(") < main.index("from tsp_optimization_amd import engine")
    assert main.index("launch_ranks(") < main.index("import torch")
