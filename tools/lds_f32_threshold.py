"""LDS engine, first improvement on 128 random tours: the fp32 first tier of the rows x columns scan on (TSP_LDS_F32_MIN_N=0) against
off (a large value), by instance size.  usage: lds_f32_threshold.py   (through gpurun)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance, random_tour
from oracle import oracle as O
ctx = E.Context(0)
for name in ["att532", "pr1002", "rand1500", "rand2000", "rand3000", "rand5000"]:
    xy, wt = load_instance(name)
    n = len(xy)
    rng = np.random.default_rng(1)
    B = 128
    tours = np.stack([random_tour(n, rng) for _ in range(B)])
    res = []
    for v in ("0", "100000"):
        os.environ["TSP_LDS_F32_MIN_N"] = v
        inst = E.Instance(ctx, xy, wt, 1)
        cost = inst.perm_cost(np.stack([np.argsort(np.zeros(1))]*0 + [np.arange(n, dtype=np.int32)]))  # warm
        c0 = np.zeros(B)
        rc, s, o, st = inst.two_opt(tours, c0, mode=E.FIRST, engine=E.ENGINE_LDS)
        rc, s, o, st = inst.two_opt(tours, c0, mode=E.FIRST, engine=E.ENGINE_LDS)
        res.append((st[0]["device_ms"], int(sum(x["moves"] for x in st))))
        inst.close()
    print("%-9s n %5d: fp32 tier on %.2f ms, off %.2f ms (%d moves)" % (name, n, res[0][0], res[1][0], res[0][1]), flush=True)
