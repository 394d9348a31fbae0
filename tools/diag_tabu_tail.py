"""Diagnostic build only (lib_diag, -DTSP_STAMPS; recipe in tools/diag_stamps.py, plus `make -C tsp_optimization_amd/host OUT=../lib_diag`):
where the tail of a tabu() iteration inside a CLUSTER launch spends its time, and the phases of the sweeps around it.
usage: diag_tabu_tail.py [instance] [iterations]"""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import build as B
B.LIB_DIR = os.path.join(R, 'tsp_optimization_amd', 'lib_diag')
from helpers import HostInstance, Instance
name = sys.argv[1] if len(sys.argv) > 1 else "rand10000"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
L = C.CDLL(B.lib_path("libtsp_host.so"))
H = C.CDLL(B.lib_path("libtsp_hip.so"))
L.tsp_host_tabu.argtypes = [C.POINTER(Instance), C.c_int, C.c_longlong]
L.tsp_host_last_driver_loop_seconds.restype = C.c_double
w = HostInstance(name); w.c.params.time_limit = 3600
L.tsp_host_tabu(C.byref(w.c), 0, 3)
buf = (C.c_ulonglong * (256 * 8))()
tbuf = (C.c_ulonglong * (256 * 12))()
H.tsp_dev_debug_cluster_tail(tbuf); H.tsp_dev_debug_cluster(buf)
h = HostInstance(name); h.c.params.time_limit = 3600
C.CDLL(None).srandom(123)
os.environ["TSP_HOST_STATS"] = "1"
L.tsp_host_tabu(C.byref(h.c), 0, iters)
t = L.tsp_host_last_driver_loop_seconds()
print("%s: %d iterations, %.1f us each (diagnostic build)" % (name, iters, 1e6 * t / iters))
H.tsp_dev_debug_cluster_tail(tbuf)
a = np.array(tbuf[:], dtype=np.float64).reshape(256, 12)
tail_extra = a[:, 8:12].copy(); a_tail = a.copy()
used = a[:, 7] > 0
tails = a[used, 7].mean()
names = ["control block (+ cost, first workgroup)", "decision + list (first workgroup)", "release fence", "exchange", "acquire fence", "barrier + kick on the replica"]
print("tails per workgroup %.0f" % tails)
for k in range(6):
    col = a[used, k] / 100.0 / tails
    print("  %-42s mean %6.2f  min %6.2f  max %6.2f  first workgroup %6.2f us/tail" % (names[k], col.mean(), col.min(), col.max(), a[0, k] / 100.0 / a[0, 7]))
print("  %-42s mean %6.2f us/tail" % ("sum", (a[used, :6].sum(1) / 100.0 / tails).mean()))
H.tsp_dev_debug_cluster(buf)
a = np.array(buf[:], dtype=np.float64).reshape(256, 8)
used = a[:, 7] > 0
steps = a[used, 7].mean()
pn = ["tests+list", "scan (units/tiles)", "block argmin", "exchange", "adjacency count", "move (+bounds)"]
print("sweeps per workgroup %.0f (%.2f per iteration)" % (steps, steps / iters))
for k in range(6):
    col = a[used, k] / 100.0 / steps
    print("  %-42s mean %6.2f  min %6.2f  max %6.2f us/sweep" % (pn[k], col.mean(), col.min(), col.max()))
print("  %-42s mean %6.2f us/sweep" % ("sum", (a[used, :6].sum(1) / 100.0 / steps).mean()))
col = tail_extra[used, 0] / 100.0 / steps
print("  inside that exchange phase: the list's side effects (thread 64) mean %.2f  max %.2f us/sweep; the exchange proper (thread 0) mean %.2f  min %.2f  max %.2f; the first workgroup's read of the live-edge count %.2f us/sweep"
      % (col.mean(), col.max(), (tail_extra[used, 2] / 100.0 / steps).mean(), (tail_extra[used, 2] / 100.0 / steps).min(), (tail_extra[used, 2] / 100.0 / steps).max(), tail_extra[0, 1] / 100.0 / steps))
ex = tail_extra[used, 2] / 100.0 / steps
order_ = np.argsort(ex)
print("  exchange proper per workgroup: lowest five", [(int(i), round(float(ex[i]), 2)) for i in order_[:5]], "highest three", [(int(i), round(float(ex[i]), 2)) for i in order_[-3:]])
pre = (a[used, 0] + a[used, 1] + a[used, 2] + a[used, 5]) / 100.0 / steps
order_ = np.argsort(-pre)
print("  own work outside the exchange phase (tests + scan + arg-min + move) per workgroup: highest five", [(int(i), round(float(pre[i]), 2)) for i in order_[:5]], "mean %.2f" % pre.mean())
se = tail_extra[used, 0] / 100.0 / steps
order_ = np.argsort(-se)
print("  list side effects per workgroup (thread 64): highest five", [(int(i), round(float(se[i]), 2)) for i in order_[:5]])
cnt = a_tail[used]
print("  thread 64 of the workgroups >= 1, per sweep: entries loaded %.3f, answered from the registers %.3f, skipped as zero %.3f"
      % ((cnt[1:, 6] / steps).mean(), (cnt[1:, 9] / steps).mean(), (cnt[1:, 11] / steps).mean()))
