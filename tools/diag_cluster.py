"""Diagnostic build only (lib_diag, -DTSP_STAMPS; recipe in tools/diag_stamps.py): where a CLUSTER-engine step spends its
time, per phase, mean and max over the workgroups of the cluster.  usage: diag_cluster.py [instance] [C]"""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import build as B
B.LIB_DIR = os.path.join(R, 'tsp_optimization_amd', 'lib_diag')
from tsp_optimization_amd import engine as E
from helpers import load_instance
name = sys.argv[1] if len(sys.argv) > 1 else 'rand10000'
if len(sys.argv) > 2: os.environ["TSP_CLUSTER_BLOCKS"] = sys.argv[2]
names = ["tests+list", "scan (units/tiles)", "block argmin", "exchange", "adjacency count", "move (+bounds)"]
ctx = E.Context(0)
xy, wt = load_instance(name)
inst = E.Instance(ctx, xy, wt, 1)
succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
L = E.lib()
buf = (C.c_ulonglong * (256 * 8))()
L.tsp_dev_debug_cluster.argtypes = [C.POINTER(C.c_ulonglong)]
L.tsp_dev_debug_cluster(buf)
cnt = (C.c_ulonglong * 8)()
L.tsp_dev_debug_cluster_counts.argtypes = [C.POINTER(C.c_ulonglong)]
L.tsp_dev_debug_cluster_counts(cnt)
for mode, nm in [(E.FIRST, 'FIRST'), (E.BEST, 'BEST')]:
    rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=mode, engine=E.ENGINE_CLUSTER)
    L.tsp_dev_debug_cluster(buf)
    a = np.array(buf[:], dtype=np.float64).reshape(256, 8)
    used = a[:, 7] > 0
    steps = st['steps']
    print(nm, name, 'steps', steps, 'device_ms %.2f' % st['device_ms'], 'us/step %.2f' % (1e3 * st['device_ms'] / steps), 'workgroups', int(used.sum()))
    for k in range(6):
        col = a[used, k] / 100.0 / steps
        print('  %-22s mean %6.2f  min %6.2f  max %6.2f us/step' % (names[k], col.mean(), col.min(), col.max()))
    print('  %-22s mean %6.2f us/step' % ('sum', (a[used, :6].sum(1) / 100.0 / steps).mean()))
    L.tsp_dev_debug_cluster_counts(cnt)
    if mode == E.FIRST:
        nb = int(used.sum())
        print('  inside the scan (thread 0, mean over workgroups): before tile %.2f  derive+barriers %.2f  row loop %.2f us/step; tiles per workgroup and step %.2f'
              % (cnt[0] / 100.0 / steps / nb, cnt[1] / 100.0 / steps / nb, cnt[2] / 100.0 / steps / nb, cnt[3] / steps / nb))
    if mode == E.BEST:
        nb = int(used.sum())
        print('  inside the scan (thread 0, mean over workgroups): stage+culling %.2f  (-) %.2f  rows+queues %.2f us/step (the rest: final barrier); live rows per workgroup and step %.1f'
              % (cnt[0] / 100.0 / steps / nb, cnt[1] / 100.0 / steps / nb, cnt[2] / 100.0 / steps / nb, cnt[3] / steps / nb))
        cyc = (C.c_ulonglong * 8)()
        L.tsp_dev_debug_cluster_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
        L.tsp_dev_debug_cluster_cycles(cyc)
        turns = max(1, cyc[1])
        print('  wave 0, rows loop: %.2f turns per step; per turn %.0f cycles fetching items, %.1f units in %.2f trips of %.0f cycles each, %.0f cycles in full tier-1/2 passes'
              % (turns / steps / nb, cyc[0] / turns, cyc[2] / turns, cyc[4] / turns, cyc[3] / max(1, cyc[4]), cyc[5] / turns))
        print('  wave 0: tier-1/2 passes per step %.2f with %.1f pairs each' % (cnt[4] / steps / nb, cnt[5] / max(1, cnt[4])))
