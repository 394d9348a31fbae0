"""BASELINE configs[3] (att532, 256 GRASP starts) and configs[4] (rand5000, 128 random individuals) on the engine AUTO picks for a
whole-chip batch on one GPU (the LDS engine, one workgroup per tour): device ms of the 2-opt, steps.  Profiled by
tools/profile_r03.sh (kernel trace + SQ counters of k_lds_two_opt).   usage: lds_time.py [reps]   (through gpurun)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from tsp_optimization_amd import engine as E, multistart as MS, tsplib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = E.Context(0)
xy, wt = tsplib.parse(os.path.join(R, "tests", "golden", "instances", "att532.tsp"))
inst4 = E.Instance(ctx, xy, wt, 1)
rng = MS.LibcRandom(123)
starts, stream = MS.grasp_stream(rng.urand, len(xy), 256)
xy5 = np.random.default_rng(5000).integers(0, 1_000_000, size=(5000, 2)).astype(np.float64)
inst5 = E.Instance(ctx, xy5, E.EUC_2D, 1)
rng = MS.LibcRandom(123)
perms = np.stack([rng.random_perm(5000) for _ in range(128)])
r4 = MS.config4_refiner(E, inst4, starts, stream)
r5 = MS.config5_refiner(E, inst5, perms)
for r in range(reps):
    t0 = time.perf_counter(); r4(list(range(256))); t4 = time.perf_counter() - t0
    d4, s4 = r4.stats[0]["device_ms"], sum(x["steps"] for x in r4.stats)
    t0 = time.perf_counter(); r5(list(range(128))); t5 = time.perf_counter() - t0
    d5, s5 = r5.stats[0]["device_ms"], sum(x["steps"] for x in r5.stats)
    print("configs[3] att532 x 256: 2-opt device %.3f ms (call %.1f ms), %d steps in all | configs[4] rand5000 x 128: 2-opt device %.1f ms (call %.1f ms), %d steps in all, %d moves"
          % (d4, 1e3 * t4, s4, d5, 1e3 * t5, s5, sum(x["moves"] for x in r5.stats)), flush=True)
