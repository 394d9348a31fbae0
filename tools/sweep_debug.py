"""Timing of the sorted sweep with parts switched off (TSP_SWEEP_DEBUG bits: 1 = no pair loop, 2 = no staging loads)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import build as BLD
if os.environ.get('LIBDIR'): BLD.LIB_DIR = os.path.join(R, 'tsp_optimization_amd', os.environ['LIBDIR'])
from tsp_optimization_amd import engine as E
from helpers import load_instance
ctx = E.Context(0)
xy, wt = load_instance('rand10000')
inst = E.Instance(ctx, xy, wt, 1)
succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
for blocks in os.environ.get("BLOCKS", "512").split(","):
    os.environ["TSP_SWEEP_BLOCKS"] = blocks
    tours = E.Tours(inst, 1)
    for dbg in os.environ.get("DBG", "0,1,2,3").split(","):
        os.environ["TSP_SWEEP_DEBUG"] = dbg
        tours.upload(succ[0], obj[0])
        ms, ev = tours.time_scan(reps=200)
        print("blocks %s debug %s: %.2f us per sweep (recs + sweep)" % (blocks, dbg, ms * 1e3))
    tours.close()
