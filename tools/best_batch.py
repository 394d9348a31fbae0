"""Best-improvement descents of a batch of tours of a small instance: the engine the library picks against the GRID engine
(TSP_ENGINE=1).  usage: best_batch.py  (through gpurun)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance, random_tour
ctx = E.Context(0)
for name in ("pr299", "att532", "rand800"):
    xy, wt = load_instance(name)
    n = len(xy)
    inst = E.Instance(ctx, xy, wt, 1)
    rng = np.random.default_rng(1)
    for B in (8, 16, 64, 128, 256):
        succ, obj, _ = inst.construct(E.GREEDY, rng.integers(0, n, size=B).astype(np.int32))
        for force in ("", "1", "3"):
            if force: os.environ["TSP_ENGINE"] = force
            else: os.environ.pop("TSP_ENGINE", None)
            inst.reload_switches()
            best = None
            for _ in range(3):
                rc, s, o, st = inst.two_opt(succ, obj, mode=E.BEST)
                best = st[0]["device_ms"] if best is None else min(best, st[0]["device_ms"])
            print("%-8s B %3d %-5s: %.2f ms" % (name, B, {"": "auto", "1": "GRID", "3": "CLUSTER"}[force], best), flush=True)
    os.environ.pop("TSP_ENGINE", None)
    inst.close()
