"""Timing of the CLUSTER engine against GRID / LDS on BASELINE's configs (run through gpurun).
usage: cluster_time.py [quick]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from oracle import oracle as O
from helpers import load_instance

ctx = E.Context(0)


def run(inst, succ, obj, mode, engine, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        rc, s, o, st = inst.two_opt(succ, obj, mode=mode, engine=engine)
        dt = time.perf_counter() - t0
        st0 = st if isinstance(st, dict) else st[0]
        if best is None or st0["device_ms"] < best[0]:
            best = (st0["device_ms"], dt, o, st)
    return best


for name in (["rand10000"] if len(sys.argv) > 1 else ["rand10000", "rand5000", "rand2000", "pr1002", "att532"]):
    xy, wt = load_instance(name)
    inst = E.Instance(ctx, xy, wt, 1)
    succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
    for mode, mn in ((E.FIRST, "FIRST"), (E.BEST, "BEST")):
        for eng, en in ((E.ENGINE_GRID, "GRID"), (E.ENGINE_CLUSTER, "CLUSTER")):
            for C in ([None] if eng == E.ENGINE_GRID else [256, 128, 64]):
                if C: os.environ["TSP_CLUSTER_BLOCKS"] = str(C)
                inst.reload_switches()
                ms, dt, o, st = run(inst, succ[0], obj[0], mode, eng)
                print("%-10s %-5s %-8s C=%-4s device %.2f ms wall %.2f ms cost %.0f steps %d sweeps %d moves %d -> %.2f us/step"
                      % (name, mn, en, C, ms, 1e3 * dt, o, st["steps"], st["sweeps"], st["moves"], 1e3 * ms / max(1, st["steps"])), flush=True)
                os.environ.pop("TSP_CLUSTER_BLOCKS", None)
                inst.reload_switches()
    inst.close()
