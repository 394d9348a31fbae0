"""One tsp_host_tabu run (step policy, seed 123) with a cap on the iterations, for a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 tools/tabu_one.py rand10000 2000     (TSP_HOST_STATS=1: chain statistics on stderr)"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import HostInstance, Instance
from tsp_optimization_amd.build import lib_path
name = sys.argv[1] if len(sys.argv) > 1 else "rand10000"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
policy = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 0 step, 1 linear, 2 random
L = C.CDLL(lib_path("libtsp_host.so"))
L.tsp_host_tabu.argtypes = [C.POINTER(Instance), C.c_int, C.c_longlong]
L.tsp_host_last_driver_loop_seconds.restype = C.c_double
if os.environ.get("TSP_HOST_STATS") == "": os.environ.pop("TSP_HOST_STATS")
w = HostInstance(name)      # warm run: the HIP runtime draws from libc's stream while it initialises (a cold process would not
w.c.params.time_limit = 3600   # see the values srandom(123) promises), and the code objects are loaded
L.tsp_host_tabu(C.byref(w.c), policy, 3)
h = HostInstance(name)
h.c.params.time_limit = 3600
C.CDLL(None).srandom(123)
L.tsp_host_tabu(C.byref(h.c), policy, iters)
t = L.tsp_host_last_driver_loop_seconds()
print("%s policy %d: %d iterations in %.3f s of loop = %.0f iterations/s, %.1f us each; incumbent %.0f" % (name, policy, iters, t, iters / t, 1e6 * t / iters, h.obj))
