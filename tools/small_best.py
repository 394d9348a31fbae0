"""Best-improvement descents of small instances on the CLUSTER engine: tiles scan (TSP_SORTED_MIN_N above n) against the sorted
scan (TSP_SORTED_MIN_N=0).  usage: small_best.py  (through gpurun)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from helpers import load_instance
for mn in ("1000000000", "0"):
    os.environ["TSP_SORTED_MIN_N"] = mn
    from tsp_optimization_amd import engine as E
    ctx = E.Context(0)
    for name in ("berlin52", "kroA100", "pr299", "att532", "d657", "rand800"):
        xy, wt = load_instance(name)
        inst = E.Instance(ctx, xy, wt, 1)
        succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
        best = None
        for _ in range(4):
            rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=E.BEST, engine=E.ENGINE_CLUSTER)
            best = st["device_ms"] if best is None else min(best, st["device_ms"])
        print("min_n %-10s %-9s n %4d: %.3f ms, %d sweeps -> %.2f us per sweep" % (mn, name, len(xy), best, st["sweeps"], 1e3 * best / st["sweeps"]), flush=True)
        inst.close()
    ctx.close()
