"""us per best-improvement step, sorted sweep vs tiled sweep, over instance sizes (picks TSP_SORTED_MIN_N)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import rand_instance
from oracle import oracle as O
ctx = E.Context(0)
for n in [int(x) for x in os.environ.get('NS', '500,1000,1500,2000,3000,4000,6000,20000,50000').split(',')]:
    xy = rand_instance(n)
    row = []
    for min_n in ("0", "1000000000"):
        os.environ["TSP_SORTED_MIN_N"] = min_n
        inst = E.Instance(ctx, xy, O.EUC_2D, 1)
        succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
        tours = E.Tours(inst, 1)
        tours.upload(succ[0], obj[0])
        tours.run(E.BEST, max_steps=20)
        ms, ev = tours.time_scan(reps=100)
        row.append(ms * 1e3)
        tours.close(); inst.close()
    print("n %6d: sorted %.2f us/step, tiled %.2f us/step" % (n, row[0], row[1]))
