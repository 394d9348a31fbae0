"""Where a round of HEU_VNS (src/vns.c:102-166) goes on resident tours: kick, alg_2opt, bookkeeping.
usage: vns_time.py [instance ...]   (through gpurun)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance

ctx = E.Context(0)
for name in (sys.argv[1:] or ["rand10000", "pr1002"]):
    xy, wt = load_instance(name)
    n = len(xy)
    inst = E.Instance(ctx, xy, wt, 1)
    succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
    tours = E.Tours(inst, 1)
    tours.upload(succ[0], obj[0])
    tours.two_opt(E.FIRST)
    tours.snapshot()
    rng = np.random.default_rng(1)
    tk = t2 = ts = 0.0
    steps = moves = sweeps = 0
    rounds = 200
    _, _, st0 = tours.download()
    for r in range(rounds):
        p = np.sort(rng.choice(n - 1, size=3, replace=False)) + 0
        t0 = time.perf_counter()
        tours.vns_kick(int(p[0]), int(p[1]), int(p[2]))
        t1 = time.perf_counter()
        rc, o = tours.two_opt(E.FIRST)
        t2_ = time.perf_counter()
        tours.snapshot() if r % 2 else tours.restore()
        t3 = time.perf_counter()
        tk += t1 - t0; t2 += t2_ - t1; ts += t3 - t2_
    _, _, st1 = tours.download()
    d = {k: st1[0][k] - st0[0][k] for k in ("steps", "moves", "sweeps", "evals")}
    print("%-10s per round: kick %.1f us, alg_2opt %.1f us, snapshot/restore %.1f us; per round %.1f steps %.1f moves %.2f sweeps %.3g evals"
          % (name, 1e6 * tk / rounds, 1e6 * t2 / rounds, 1e6 * ts / rounds, d["steps"] / rounds, d["moves"] / rounds,
             d["sweeps"] / rounds, d["evals"] / rounds), flush=True)
    for eng, en in ((E.ENGINE_GRID, "GRID"), (E.ENGINE_CLUSTER, "CLUSTER"), (E.ENGINE_LDS, "LDS")):
        try:
            tt = 0.0
            for r in range(50):
                p = np.sort(rng.choice(n - 1, size=3, replace=False))
                tours.restore()
                tours.vns_kick(int(p[0]), int(p[1]), int(p[2]))
                t0 = time.perf_counter()
                tours.two_opt(E.FIRST, engine=eng)
                tt += time.perf_counter() - t0
            print("   engine %-8s alg_2opt %.1f us per round" % (en, 1e6 * tt / 50), flush=True)
        except Exception as ex:
            print("   engine %-8s: %s" % (en, ex))
    tours.close(); inst.close()
