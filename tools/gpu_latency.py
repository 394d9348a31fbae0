"""Per-call wall time of the host-tour entry points on small instances (what a VNS / GA loop pays)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from oracle import oracle as O
from helpers import load_instance
ctx = E.Context(0)
for name in ['berlin52', 'pr299', 'att532', 'dsj1000', 'rand2000']:
    xy, wt = load_instance(name)
    inst = E.Instance(ctx, xy, wt, 1)
    _, succ, obj = O.greedy(xy, wt)
    for mode, nm in [(E.FIRST, 'first'), (E.BEST, 'best')]:
        for engine in (1, 2):
            if engine == 2 and mode == E.BEST and len(xy) > 1500: continue
            inst.two_opt(succ, obj, mode=mode, engine=engine)
            t0 = time.perf_counter(); reps = 20
            for _ in range(reps): rc, s, o, st = inst.two_opt(succ, obj, mode=mode, engine=engine)
            dt = (time.perf_counter() - t0) / reps
            print("%-9s n=%5d %-5s engine=%d  wall %.3f ms  device %.3f ms  steps %d" % (name, len(xy), nm, engine, dt*1e3, st['device_ms'], st['steps']))
    t0 = time.perf_counter()
    for _ in range(20): inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
    print("%-9s greedy construct wall %.3f ms" % (name, (time.perf_counter()-t0)/20*1e3))
    inst.close()
