"""Diagnostic build only (lib_diag, -DTSP_STAMPS): where a step of the LDS engine (one workgroup per tour) spends its cycles,
BASELINE configs[4] (16 random individuals of rand5000) and configs[3] (att532 GRASP starts).  Tour 0, thread 0."""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import build as B
B.LIB_DIR = os.path.join(R, 'tsp_optimization_amd', 'lib_diag')
from tsp_optimization_amd import engine as E, multistart as MS, tsplib
ctx = E.Context(0)
L = E.lib()
buf = (C.c_ulonglong * 8)()
L.tsp_dev_debug_lds.argtypes = [C.POINTER(C.c_ulonglong)]
names = ["scan (rows x columns)", "block argmin", "adjacency count", "move", "probe", "", "control block"]
xy5 = np.random.default_rng(5000).integers(0, 1_000_000, size=(5000, 2)).astype(np.float64)
inst = E.Instance(ctx, xy5, E.EUC_2D, 1)
rng = MS.LibcRandom(123)
perms = np.stack([rng.random_perm(5000) for _ in range(16)])
succ = np.stack([MS.perm_to_succ(p) for p in perms])
cost = inst.perm_cost(perms)
sb = (C.c_ulonglong * 8)()
L.tsp_dev_debug_lds_scan.argtypes = [C.POINTER(C.c_ulonglong)]
L.tsp_dev_debug_lds_scan(sb)
L.tsp_dev_debug_lds(buf)
rc, s2, o2, st = inst.two_opt(succ, cost, mode=E.FIRST, engine=E.ENGINE_LDS)
L.tsp_dev_debug_lds(buf)
steps = max(1, buf[7])
print("rand5000 random individual, LDS engine: %d steps, %d moves, device %.1f ms -> %.2f us/step" % (steps, st[0]["moves"], st[0]["device_ms"], 1e3 * st[0]["device_ms"] / steps))
for k in (0, 1, 2, 3, 4, 6):
    print("  %-22s %8.0f cycles/step" % (names[k], buf[k] / steps))
print("  steps decided by the probe: %.1f %%" % (100.0 * buf[5] / steps))
L.tsp_dev_debug_lds_scan(sb)
nbt = max(1, sb[3])
print("  scan, per batch of columns (%.1f batches per step, %.2f rows each): column records %.0f cycles, row loop %.0f, vote %.0f" % (sb[3] / steps, sb[4] / nbt, sb[0] / nbt, sb[1] / nbt, sb[2] / nbt))
print("  probe: records %.0f cycles/step, delta + ballots %.0f, barrier %.0f" % (sb[5] / steps, sb[6] / steps, sb[7] / steps))
print("  tour 0: pairs scanned %d = %.0f per step = %.1f batches of 512 columns; reference evaluations %d = %.0f per move"
      % (st[0]["pairs_scanned"], st[0]["pairs_scanned"] / st[0]["steps"], st[0]["pairs_scanned"] / st[0]["steps"] / 512.0,
         st[0]["evals"], st[0]["evals"] / st[0]["moves"]))
