"""A/B of switch settings on the two single-tour descents of rand10000 (run through gpurun).
usage: ab.py TSP_X=1[,TSP_Y=2] TSP_X=8 ...   (each argument one setting; '-' = defaults)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance
ctx = E.Context(0)
xy, wt = load_instance(os.environ.get("AB_INSTANCE", "rand10000"))
inst = E.Instance(ctx, xy, wt, 1)
succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
for rep in range(2):
    for setting in sys.argv[1:]:
        kv = [x.split("=") for x in setting.split(",") if "=" in x]
        for k, v in kv: os.environ[k] = v
        inst.reload_switches()
        out = []
        for mode in (E.BEST, E.FIRST):
            ms = []
            for _ in range(6):
                rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=mode, engine=E.ENGINE_CLUSTER)
                ms.append(st["device_ms"])
            out.append("%s min %.3f median %.3f ms (cost %.0f, steps %d)" % ("BEST" if mode == E.BEST else "FIRST", min(ms), sorted(ms)[3], o, st["steps"]))
        print("%-28s %s" % (setting, " | ".join(out)), flush=True)
        for k, v in kv: os.environ.pop(k, None)
