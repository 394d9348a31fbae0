"""us per best-improvement step of the non-integer-coordinate kernel variants (general sqrt), sorted vs tiled."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from oracle import oracle as O
ctx = E.Context(0)
rng = np.random.default_rng(4)
for n in (2000, 10000):
    xy = rng.uniform(0, 1e6, size=(n, 2))
    for wt, ic, nm in ((O.EUC_2D, 1, "EUC_2D int cost"), (O.EUC_2D, 0, "EUC_2D fcost"), (O.ATT, 1, "ATT int"), (O.CEIL_2D, 1, "CEIL_2D")):
        row = []
        for min_n in ("0", "1000000000"):
            os.environ["TSP_SORTED_MIN_N"] = min_n
            inst = E.Instance(ctx, xy, wt, ic)
            succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
            tours = E.Tours(inst, 1)
            tours.upload(succ[0], obj[0])
            tours.run(E.BEST, max_steps=20)
            ms, ev = tours.time_scan(reps=60)
            row.append(ms * 1e3)
            tours.close(); inst.close()
        print("n %5d %-15s: sorted %.2f us/step, tiled %.2f us/step" % (n, nm, row[0], row[1]))
