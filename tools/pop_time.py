"""BASELINE configs[4] (rand5000, random individuals, alg_2opt each) for a shard of S individuals on the engine TSP_ENGINE picks
(unset = the library's choice): device ms.  usage: pop_time.py [S ...]   (through gpurun)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from tsp_optimization_amd import engine as E, multistart as MS

ctx = E.Context(0)
xy5 = np.random.default_rng(5000).integers(0, 1_000_000, size=(5000, 2)).astype(np.float64)
inst5 = E.Instance(ctx, xy5, E.EUC_2D, 1)
rng = MS.LibcRandom(123)
perms = np.stack([rng.random_perm(5000) for _ in range(128)])
r5 = MS.config5_refiner(E, inst5, perms)
for S in [int(x) for x in sys.argv[1:]] or [128, 64, 32, 16]:
    for rep in range(2):
        t0 = time.perf_counter(); r5(list(range(S))); t5 = time.perf_counter() - t0
    print("engine %s: %3d individuals: 2-opt device %.1f ms (call %.1f ms), steps %d, moves %d, lane pairs %s"
          % (os.environ.get("TSP_ENGINE", "auto"), S, r5.stats[0]["device_ms"], 1e3 * t5, sum(x["steps"] for x in r5.stats),
             sum(x["moves"] for x in r5.stats), r5.stats[0]["exact_pairs"] >= 0 and "counted (CLUSTER)" or "-"), flush=True)
