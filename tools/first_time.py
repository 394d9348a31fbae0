"""alg_2opt (first improvement, src/heuristics.c:438-502) of the rand10000 greedy tour on resident tours, `reps` times:
device ms (HIP events), steps, counters.  Profiled by tools/profile_r03.sh (kernel trace + SQ counters of the FIRST cluster kernel).
usage: first_time.py [reps] [instance]   (through gpurun)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
name = sys.argv[2] if len(sys.argv) > 2 else "rand10000"
ctx = E.Context(0)
xy, wt = load_instance(name)
inst = E.Instance(ctx, xy, wt, 1)
succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
tours = E.Tours(inst, 1)
tours.upload(succ[0], obj[0])
ms = []
for r in range(reps + 2):
    tours.reset()
    rc, done = tours.run_engine(E.FIRST, engine=E.ENGINE_AUTO)
    assert rc == 0 and done
    _, o, st = tours.download()
    if r >= 2:
        ms.append(st[0]["device_ms"])
st = st[0]
print("%s alg_2opt: device ms mean %.3f min %.3f max %.3f over %d runs; steps %d (%.2f us per step), sweeps %d, evals %d, moves %d, cost %.0f"
      % (name, np.mean(ms), np.min(ms), np.max(ms), reps, st["steps"], 1e3 * np.mean(ms) / st["steps"], st["sweeps"], st["evals"], st["moves"], o[0]))
# the last sweep of every descent finds nothing: a resident call on the local optimum is that sweep alone
t = []
for r in range(reps + 2):
    ctx.synchronize()
    t0 = time.perf_counter()
    rc, o2 = tours.two_opt(E.FIRST)
    t.append(time.perf_counter() - t0)
print("%s sweep that finds nothing (resident call on the local optimum): %.1f us per call (min %.1f)" % (name, 1e6 * np.mean(t[2:]), 1e6 * np.min(t[2:])))
