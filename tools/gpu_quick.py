import sys, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from oracle import oracle as O
from helpers import load_instance
ctx = E.Context(0)
xy, wt = load_instance('rand10000')
inst = E.Instance(ctx, xy, wt, 1)
t=time.time(); succ, obj, _ = inst.construct(E.GREEDY, np.array([0],dtype=np.int32)); print('greedy dev', obj[0], time.time()-t)
tours = E.Tours(inst, 1); tours.upload(succ[0], obj[0])
ms, ev = tours.time_scan(20); print('scan ms', ms, 'evals', ev, 'evals/s %.3e'%(ev/ms*1e3))
for mode,name in [(E.FIRST,'first'),(E.BEST,'best')]:
    t=time.time(); rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=mode); dt=time.time()-t
    print(name, rc, o, st, 'wall', dt, 'evals/s %.3e'%(st['evals']/dt))
_, ms = inst.dist_matrix(as_int32=True, fetch=False); print('distmat int32 ms', ms, 'GB/s', 4e8/ms/1e6)
_, ms = inst.dist_matrix(as_int32=False, fetch=False); print('distmat f64 ms', ms, 'GB/s', 8e8/ms/1e6)
