import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance
ctx = E.Context(0)
for name in ("att532", "pr1002", "rand5000", "rand10000"):
    xy, wt = load_instance(name)
    n = len(xy)
    for nn in ("1", "0"):
        os.environ["TSP_CONSTRUCT_NN"] = nn
        inst = E.Instance(ctx, xy, wt, 1)
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32)); ts.append(time.perf_counter() - t0)
        line = "%-9s nn=%s: greedy(0) %.2f ms cost %.0f" % (name, nn, 1e3 * min(ts), obj[0])
        if n <= 1100:
            dt = 1e9
            for rep in range(3):
                t0 = time.perf_counter(); s2, o2, _ = inst.construct(E.GREEDY, np.arange(n, dtype=np.int32)); dt = min(dt, time.perf_counter() - t0)
            line += "; all %d starts %.2f ms best %.0f" % (n, 1e3 * dt, o2.min())
        print(line)
        inst.close()
