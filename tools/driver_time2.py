"""tabu() on resident state through the C host (tsp_host_tabu with a cap on the iterations), chains of TSP_TABU_CHAIN iterations
per wait for the device against one per wait.  Rate = (N2 - N1) / (t(N2) - t(N1)), each time the best of two runs: the initial
solution and the first iterations' transient cancel out.  usage: driver_time2.py [instance] [N1] [N2]   (through gpurun)"""
import ctypes as C, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import HostInstance, Instance
from tsp_optimization_amd.build import lib_path
name = sys.argv[1] if len(sys.argv) > 1 else "rand10000"
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n2 = int(sys.argv[3]) if len(sys.argv) > 3 else 2200
L = C.CDLL(lib_path("libtsp_host.so"))
L.tsp_host_tabu.argtypes = [C.POINTER(Instance), C.c_int, C.c_longlong]


def run(count):
    best, obj = 1e9, None
    for rep in range(2):
        h = HostInstance(name)
        h.c.params.time_limit = 3600
        C.CDLL(None).srandom(123)
        t0 = time.perf_counter()
        L.tsp_host_tabu(C.byref(h.c), 0, count)
        best = min(best, time.perf_counter() - t0)
        obj = h.obj
    return best, obj


run(n1)
for chain in ("1", "4", "8", "16", "64", "auto"):
    if chain == "auto":
        os.environ.pop("TSP_TABU_CHAIN", None)
    else:
        os.environ["TSP_TABU_CHAIN"] = chain
    ta, _ = run(n1)
    tb, obj = run(n2)
    print("%s: chain %4s: %.0f iterations/s (%d iterations in %.3f s, %d in %.3f s; incumbent %.0f)" % (name, chain, (n2 - n1) / (tb - ta), n1, ta, n2, tb, obj), flush=True)
