import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance
ctx = E.Context(0)
for name in ['att532', 'pr1002', 'rand5000']:
    xy, wt = load_instance(name); n = len(xy)
    inst = E.Instance(ctx, xy, wt, 1)
    inst.extramileage()
    t0 = time.perf_counter(); s, o = inst.extramileage(); dt = time.perf_counter() - t0
    print("%-9s extramileage %.1f ms (obj %.0f)" % (name, dt * 1e3, o))
    rng = np.random.default_rng(0)
    perms = np.stack([rng.permutation(n).astype(np.int32) for _ in range(1000)])
    inst.perm_cost(perms[:4])
    t0 = time.perf_counter(); c = inst.perm_cost(perms); dt = time.perf_counter() - t0
    print("%-9s fitness of 1000 permutations %.2f ms" % (name, dt * 1e3))
    t0 = time.perf_counter(); succ, obj, _ = inst.construct(E.GREEDY, np.arange(min(n, 1024), dtype=np.int32)); dt = time.perf_counter() - t0
    print("%-9s greedy from %d starts %.1f ms" % (name, min(n, 1024), dt * 1e3))
    inst.close()
