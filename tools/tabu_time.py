"""Cost of one alg_2opt_tabu sweep with a tabu list (src/tabusearch.c:127-165) on resident tours and stamps.
Full descents of the greedy tour with an empty list and with a list of LIVE random stamps; ms per sweep.
usage: tabu_time.py [instance ...]   (through gpurun)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance

ctx = E.Context(0)
for name in (sys.argv[1:] or ["rand10000", "rand5000", "pr1002"]):
    xy, wt = load_instance(name)
    n = len(xy)
    inst = E.Instance(ctx, xy, wt, 1)
    succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
    tours = E.Tours(inst, 1)
    for live in (0, 200, 2000):
        tb = E.Tabu(inst)
        if live:
            rng = np.random.default_rng(live)
            idx = rng.choice(n * (n - 1) // 2, size=live, replace=False).astype(np.int32)
            tb.set(idx, np.full(live, 5, dtype=np.int32))     # stamped at iteration 5, looked at in iteration 6: live for any tenure >= 1
        for rep in range(2):
            tours.upload(succ[0], obj[0])
            ctx.synchronize()
            t0 = time.perf_counter()
            rc, o = tours.two_opt_tabu(tb, 6, max(2, n // 50))
            dt = time.perf_counter() - t0
        _, _, st = tours.download()
        st0 = st if isinstance(st, dict) else st[0]
        print("%-10s live stamps %5d: %.1f ms, %d sweeps -> %.1f us per sweep, cost %.0f, evals %d"
              % (name, live, 1e3 * dt, st0["sweeps"], 1e6 * dt / max(1, st0["sweeps"]), o, st0["evals"]), flush=True)
        tb.close()
    # the same descent without a list (what the default engine does with it)
    tours.upload(succ[0], obj[0])
    ctx.synchronize()
    t0 = time.perf_counter()
    rc, o = tours.two_opt_tabu(None, 1, 0)
    dt = time.perf_counter() - t0
    print("%-10s no list: %.1f ms, cost %.0f" % (name, 1e3 * dt, o), flush=True)
    tours.close(); inst.close()
