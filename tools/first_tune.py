"""first-improvement descent of rand10000 under different k_first geometries (env knobs read at Tours creation)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import build as BLD
if os.environ.get('LIBDIR'): BLD.LIB_DIR = os.path.join(R, 'tsp_optimization_amd', os.environ['LIBDIR'])
from tsp_optimization_amd import engine as E
from helpers import load_instance
ctx = E.Context(0)
xy, wt = load_instance(os.environ.get("INST", "rand10000"))
for v1 in os.environ.get("V1", "0").split(","):
    for gy in os.environ.get("GY", "32").split(","):
        for mn in os.environ.get("MINROWS", "32").split(","):
            os.environ["TSP_FIRST_V1"] = v1; os.environ["TSP_FIRST_GRID_ROWS"] = gy; os.environ["TSP_FIRST_MIN_ROWS"] = mn
            inst = E.Instance(ctx, xy, wt, 1)
            succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
            best = 1e9
            for rep in range(3):
                rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=E.FIRST, engine=1)
                best = min(best, st["device_ms"])
            print("v1=%s gy=%s min_rows=%s: %.2f ms, %d steps, %.2f us/step, cost %.0f" % (v1, gy, mn, best, st["steps"], 1e3 * best / st["steps"], o))
            inst.close()
