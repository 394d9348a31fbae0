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
    col = a[used, 6] / 100.0 / steps
    print('  own scan time of a workgroup (staging + units, thread 0), mean over the steps: mean %.2f  min %.2f  p10 %.2f  p90 %.2f  max %.2f us/step' % (col.mean(), col.min(), np.percentile(col, 10), np.percentile(col, 90), col.max()))
    L.tsp_dev_debug_cluster_counts(cnt)
    ws = (C.c_ulonglong * 1024)()
    L.tsp_dev_debug_cluster_wstat.argtypes = [C.POINTER(C.c_ulonglong)]
    L.tsp_dev_debug_cluster_wstat(ws)
    w = np.array(ws[:], dtype=np.float64).reshape(256, 4)[used] / steps
    if mode == E.BEST:
        X = np.column_stack([np.ones(len(col)), w[:, 0] / 16.0, w[:, 1], w[:, 3] / 128.0])
        beta, *_ = np.linalg.lstsq(X, col, rcond=None)
        print('  per workgroup and step: units mean %.0f (min %.0f max %.0f), tier-1 pairs mean %.0f (min %.0f max %.0f), staged pairs mean %.2f (min %.2f max %.2f)'
              % (X[:, 1].mean(), X[:, 1].min(), X[:, 1].max(), X[:, 2].mean(), X[:, 2].min(), X[:, 2].max(), X[:, 3].mean(), X[:, 3].min(), X[:, 3].max()))
        print('  own scan time ~ %.2f + %.4f units + %.4f tier-1 pairs + %.3f staged pairs (least squares, residual sd %.2f us; correlations %.2f %.2f %.2f)'
              % (beta[0], beta[1], beta[2], beta[3], (col - X @ beta).std(), np.corrcoef(col, X[:, 1])[0, 1], np.corrcoef(col, X[:, 2])[0, 1], np.corrcoef(col, X[:, 3])[0, 1]))
    if mode == E.FIRST:
        nb = int(used.sum())
        print('  inside the scan (thread 0, mean over workgroups): before tile %.2f  derive+barriers %.2f  row loop %.2f us/step; tiles per workgroup and step %.2f'
              % (cnt[0] / 100.0 / steps / nb, cnt[1] / 100.0 / steps / nb, cnt[2] / 100.0 / steps / nb, cnt[3] / steps / nb))
    if mode == E.BEST:
        nb = int(used.sum())
        print('  inside the scan (thread 0, mean over workgroups): stage+culling %.2f  (-) %.2f  rows+queues %.2f us/step (the rest: final barrier); live rows per workgroup and step %.1f'
              % (cnt[0] / 100.0 / steps / nb, cnt[1] / 100.0 / steps / nb, cnt[2] / 100.0 / steps / nb, cnt[3] / steps / nb))
        b0s = (C.c_ulonglong * 8)()
        L.tsp_dev_debug_cluster_b0.argtypes = [C.POINTER(C.c_ulonglong)]
        L.tsp_dev_debug_cluster_b0(b0s)
        ns = max(1, b0s[0])
        print('  the bound a sweep starts from: mean -b0 %.0f against mean -delta of the winner %.0f; b0 == delta in %.1f %% of the sweeps, within a factor 1.25 in %.1f %%, 2 in %.1f %%'
              % (b0s[1] / ns, b0s[2] / ns, 100.0 * b0s[3] / ns, 100.0 * b0s[5] / ns, 100.0 * b0s[4] / ns))
        print('  wave 0: tier-1/2 passes per step %.2f with %.1f pairs each' % (cnt[4] / steps / nb, cnt[5] / max(1, cnt[4])))
