"""tabu() on resident state through the C host (tsp_host_tabu with a cap on the iterations), chains of TSP_TABU_CHAIN iterations
per wait for the device against one per wait.  Rate = iterations / the seconds the call spent in its iteration loop
(tsp_host_last_driver_loop_seconds: the initial HEU_2opt_greedy_iter is not in it), best of three.
usage: driver_time2.py [instance] [iterations]   (through gpurun)"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import HostInstance, Instance
from tsp_optimization_amd.build import lib_path
name = sys.argv[1] if len(sys.argv) > 1 else "rand10000"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
L = C.CDLL(lib_path("libtsp_host.so"))
L.tsp_host_tabu.argtypes = [C.POINTER(Instance), C.c_int, C.c_longlong]
L.tsp_host_last_driver_loop_seconds.restype = C.c_double


def run(count):
    best, obj = 1e9, None
    for rep in range(3):
        h = HostInstance(name)
        h.c.params.time_limit = 3600
        C.CDLL(None).srandom(123)
        L.tsp_host_tabu(C.byref(h.c), 0, count)
        best = min(best, L.tsp_host_last_driver_loop_seconds())
        obj = h.obj
    return best, obj


for chain in ("1", "4", "8", "16", "64", "auto"):
    if chain == "auto":
        os.environ.pop("TSP_TABU_CHAIN", None)
    else:
        os.environ["TSP_TABU_CHAIN"] = chain
    t, obj = run(iters)
    print("%s: chain %4s: %.0f iterations/s (%d iterations in %.3f s of loop; incumbent %.0f)" % (name, chain, iters / t, iters, t, obj), flush=True)
