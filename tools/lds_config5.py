"""BASELINE configs[4] alone (rand5000, 128 random individuals refined by alg_2opt, LDS engine) for profiling."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import build as BLD
if os.environ.get('DIAG'): BLD.LIB_DIR = os.path.join(R, 'tsp_optimization_amd', 'lib_diag')
from tsp_optimization_amd import engine as E
from helpers import rand_instance, random_tour
from oracle import oracle as O
n, B = int(os.environ.get("N", "5000")), int(os.environ.get("B", "128"))
ctx = E.Context(0)
xy = rand_instance(n)
inst = E.Instance(ctx, xy, O.EUC_2D, 1)
rng = np.random.default_rng(5)
succ = np.stack([random_tour(n, rng) for _ in range(B)])
perm = np.stack([O.succ_to_perm(s) for s in succ])
obj = inst.perm_cost(perm)
t0 = time.perf_counter()
rc, s2, o2, st = inst.two_opt(succ, obj, mode=E.FIRST)
dt = time.perf_counter() - t0
steps = np.array([x["steps"] for x in st]); moves = np.array([x["moves"] for x in st])
print("n %d B %d: %.3f s; steps per tour mean %.0f max %.0f; moves mean %.0f; us per step of the slowest tour %.2f"
      % (n, B, dt, steps.mean(), steps.max(), moves.mean(), 1e6 * dt / steps.max()))

import ctypes as C
L = E.lib()
if hasattr(L, "tsp_dev_debug_lds"):
    c = (C.c_ulonglong * 8)()
    L.tsp_dev_debug_lds(c)
    stp = max(c[7], 1)
    print("tour 0, thread 0, cycles per step: scan %.0f  arg-min %.0f  counters %.0f  move %.0f  (steps %d)" % (c[0] / stp, c[1] / stp, c[2] / stp, c[3] / stp, c[7]))
