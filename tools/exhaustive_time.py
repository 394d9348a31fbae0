"""The exhaustive best-improvement sweep (every delta expression executed, TSP_NO_FILTER=1: bench.py's timed kernel) on rand10000:
the position-order kernel k_move_pos + k_exh (two_opt_exh.hpp) against its grid shape, and the tiled k_recs + k_step it replaced.
usage: exhaustive_time.py [rj:waves ...]   (through gpurun)"""
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
cases = sys.argv[1:] or ["old", "1:4", "2:2", "2:4", "2:8", "4:2", "4:4", "4:8"]
for c in cases:
    if c == "old":
        os.environ["TSP_EXH_POS"] = "0"
    else:
        os.environ["TSP_EXH_POS"] = "1"
        os.environ["TSP_EXH_RJ"], os.environ["TSP_EXH_WAVES"] = c.split(":")
    inst.reload_switches()
    tours = E.Tours(inst, 1)
    tours.upload(succ[0], obj[0])
    best = min(tours.time_scan(50)[0] for _ in range(3))
    ev = tours.time_scan(1)[1]
    print("%-6s %-90s %.1f us per sweep, %.3g delta/s" % (c, tours.describe(E.BEST), 1e3 * best, ev / (1e-3 * best)), flush=True)
    tours.close()
