"""A/B of TSP_CLUSTER_DEBUG bits on the best-improvement descent of rand10000 (run through gpurun).
usage: best_ab.py <bits> [<bits> ...]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance
ctx = E.Context(0)
xy, wt = load_instance("rand10000")
inst = E.Instance(ctx, xy, wt, 1)
succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
for rep in range(2):
    for bits in sys.argv[1:]:
        os.environ["TSP_CLUSTER_DEBUG"] = bits
        inst.reload_switches()
        ms = []
        for _ in range(6):
            rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=E.BEST, engine=E.ENGINE_CLUSTER)
            ms.append(st["device_ms"])
        print("debug %s: device ms min %.3f median %.3f, cost %.0f sweeps %d, tier1 %s exact %s" % (bits, min(ms), sorted(ms)[3], o, st["sweeps"], st.get("tier1_pairs"), st.get("exact_pairs")), flush=True)
