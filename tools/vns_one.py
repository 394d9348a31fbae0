"""One tsp_host_vns run with a cap on the rounds, for a kernel trace:  rocprofv3 --kernel-trace --stats -- python3 tools/vns_one.py rand10000 300"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import HostInstance, Instance
from tsp_optimization_amd.build import lib_path
name = sys.argv[1] if len(sys.argv) > 1 else "rand10000"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
L = C.CDLL(lib_path("libtsp_host.so"))
L.tsp_host_vns.argtypes = [C.POINTER(Instance), C.c_longlong]
L.tsp_host_last_driver_loop_seconds.restype = C.c_double
w = HostInstance(name); w.c.params.time_limit = 3600
L.tsp_host_vns(C.byref(w.c), 3)
h = HostInstance(name); h.c.params.time_limit = 3600
C.CDLL(None).srandom(123)
L.tsp_host_vns(C.byref(h.c), rounds)
t = L.tsp_host_last_driver_loop_seconds()
print("%s: %d rounds in %.3f s of loop = %.0f rounds/s, %.1f us each; incumbent %.0f" % (name, rounds, t, rounds / t, 1e6 * t / rounds, h.obj))
