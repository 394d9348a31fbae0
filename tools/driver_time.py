"""Iterations per second of the tabu and VNS drivers of the host mirror (resident tours and stamps), pr1002 and rand10000.
usage: driver_time.py   (through gpurun)"""
import ctypes as C
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from test_gpu_host_cli import Instance, HostInstance
from tsp_optimization_amd.build import lib_path
L = C.CDLL(lib_path("libtsp_host.so"))
L.tsp_host_vns.argtypes = [C.POINTER(Instance), C.c_longlong]
L.tsp_host_tabu.argtypes = [C.POINTER(Instance), C.c_int, C.c_longlong]
libc = C.CDLL(None)
for name, iters in (("pr1002", 4000), ("rand10000", 1000)):
    for what in ("tabu", "vns"):
        for k in (iters // 10, iters):       # the difference of two runs leaves the initial solution (HEU_2opt_greedy_iter) and
                                             # the first, long descent out
            h = HostInstance(name)
            h.c.params.time_limit = 3600
            libc.srandom(123)
            t0 = time.perf_counter()
            rc = L.tsp_host_tabu(C.byref(h.c), 0, k) if what == "tabu" else L.tsp_host_vns(C.byref(h.c), k)
            dt = time.perf_counter() - t0
            if k != iters:
                base = dt
            else:
                print("%-10s %-5s iterations %d .. %d in %.3f s = %.1f iterations/s (the first %d with the initial solution: %.3f s), cost %.0f"
                      % (name, what, iters // 10, k, dt - base, (k - iters // 10) / max(dt - base, 1e-9), iters // 10, base, h.obj), flush=True)
L.tsp_host_shutdown()
