"""First-improvement descents of a batch of random tours of an instance beyond the LDS engine (n = 10 000): the engine the
library picks against the CLUSTER and GRID engines.  usage: first_batch_big.py  (through gpurun)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance, random_tour
ctx = E.Context(0)
xy, wt = load_instance("rand10000")
n = len(xy)
inst = E.Instance(ctx, xy, wt, 1)
for B in (8, 32, 64):
    succ, obj, _ = inst.construct(E.GREEDY, np.arange(B, dtype=np.int32) * 37)
    for force in ("", "1", "3"):
        if force: os.environ["TSP_ENGINE"] = force
        else: os.environ.pop("TSP_ENGINE", None)
        inst.reload_switches()
        t0 = time.perf_counter()
        rc, s, o, st = inst.two_opt(succ, obj, mode=E.FIRST)
        dt = time.perf_counter() - t0
        print("rand10000 greedy starts B %3d %-7s: device %.1f ms (wall %.1f)" % (B, {"": "auto", "1": "GRID", "3": "CLUSTER"}[force], st[0]["device_ms"], 1e3 * dt), flush=True)
os.environ.pop("TSP_ENGINE", None)
inst.close()
