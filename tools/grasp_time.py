import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance
ctx = E.Context(0)
for name in ("pr1002", "rand10000"):
    xy, wt = load_instance(name); n = len(xy)
    rng = np.random.default_rng(1)
    urand = rng.random((2, n))
    res = []
    for nn in ("1", "0"):
        os.environ["TSP_CONSTRUCT_NN"] = nn
        inst = E.Instance(ctx, xy, wt, 1)
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); succ, obj, _ = inst.construct(E.GRASP, np.array([3, n - 2], dtype=np.int32), urand); ts.append(time.perf_counter() - t0)
        res.append((succ.copy(), obj.copy()))
        print("%-9s grasp x2 nn=%s: %.2f ms cost %s" % (name, nn, 1e3 * min(ts), obj))
        inst.close()
    print("   same tours:", bool((res[0][0] == res[1][0]).all()), bool((res[0][1] == res[1][1]).all()))
