"""The exhaustive tiled sweep (every delta expression executed, TSP_NO_FILTER=1: bench.py's roofline.exhaustive) against
rows per block.  usage: exhaustive_time.py   (through gpurun)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
os.environ["TSP_NO_FILTER"] = "1"
from tsp_optimization_amd import engine as E
from helpers import load_instance

ctx = E.Context(0)
xy, wt = load_instance("rand10000")
inst = E.Instance(ctx, xy, wt, 1)
succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
for rpb in sys.argv[1:] or ["16", "32", "64", "128", "256"]:
    os.environ["TSP_BEST_ROWS_PER_BLOCK"] = rpb
    inst.reload_switches()
    tours = E.Tours(inst, 1)
    tours.upload(succ[0], obj[0])
    best = min(tours.time_scan(50)[0] for _ in range(3))
    ev = tours.time_scan(1)[1]
    print("rows per block %4s: %.1f us per sweep, %.3g delta/s, %.3f of the fp64 vector peak (35 ops per delta, 39.3 T/s)"
          % (rpb, 1e3 * best, ev / (1e-3 * best), ev / (1e-3 * best) * 35 / 39.3e12), flush=True)
    tours.close()
