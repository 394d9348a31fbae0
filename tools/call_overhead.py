"""Fixed cost of one alg_2opt_tabu / alg_2opt call on a resident tour that is already at its local optimum (one sweep that finds
nothing): what every iteration of tabu() / HEU_VNS pays besides its sweeps.  usage: call_overhead.py  (through gpurun)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance

ctx = E.Context(0)
for name in (sys.argv[1:] or ["rand10000", "pr1002"]):
    xy, wt = load_instance(name)
    inst = E.Instance(ctx, xy, wt, 1)
    succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
    for eng in ("0", "1"):
        os.environ["TSP_ENGINE"] = eng
        inst.reload_switches()
        tours = E.Tours(inst, 1)
        tours.upload(succ[0], obj[0])
        tb = E.Tabu(inst)
        tours.two_opt_tabu(tb, 1, 10)
        tours.two_opt(E.FIRST)          # a tour that is a local optimum of both rules?  not necessarily: iterate
        for _ in range(20):
            tours.two_opt_tabu(tb, 1, 10); tours.two_opt(E.FIRST)
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps): tours.two_opt_tabu(tb, 2, 10)
        t1 = time.perf_counter()
        for _ in range(reps): tours.two_opt_tabu(None, 2, 10)
        t2 = time.perf_counter()
        for _ in range(reps): tours.two_opt(E.FIRST)
        t3 = time.perf_counter()
        print("%-10s TSP_ENGINE=%s: call at the optimum: with a list %.1f us, without %.1f us, alg_2opt (first improvement, one clean sweep) %.1f us"
              % (name, eng, 1e6 * (t1 - t0) / reps, 1e6 * (t2 - t1) / reps, 1e6 * (t3 - t2) / reps), flush=True)
        tb.close(); tours.close()
    inst.close()
