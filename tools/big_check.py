"""Sanity at large n: sorted and tiled best-improvement sweeps agree for a few steps; timings."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import rand_instance
from oracle import oracle as O
n = int(os.environ.get("N", "100000")); steps = int(os.environ.get("STEPS", "6"))
ctx = E.Context(0)
xy = rand_instance(n)
out = []
for min_n in ("0", "1000000000"):
    os.environ["TSP_SORTED_MIN_N"] = min_n
    inst = E.Instance(ctx, xy, O.EUC_2D, 1)
    t0 = time.perf_counter(); succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32)); tg = time.perf_counter() - t0
    tours = E.Tours(inst, 1)
    tours.upload(succ[0], obj[0])
    t0 = time.perf_counter(); tours.run(E.BEST, max_steps=steps); dt = time.perf_counter() - t0
    s, o, st = tours.download()
    out.append((s.copy(), st[0]["moves"], st[0]["reversed"]))
    print("n %d sorted_min_n %s: greedy %.1f ms (cost %.0f), %d steps in %.2f ms (%.1f us/step)" % (n, min_n, 1e3 * tg, obj[0], steps, 1e3 * dt, 1e6 * dt / steps))
    if min_n == "0":
        t0 = time.perf_counter(); tours.run(E.FIRST, max_steps=50); dtf = time.perf_counter() - t0
        print("   50 first-improvement steps: %.2f ms" % (1e3 * dtf))
    tours.close(); inst.close()
print("agree:", bool((out[0][0] == out[1][0]).all()), out[0][1:], out[1][1:])
