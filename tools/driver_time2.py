"""tabu() on resident state through the C host (tsp_host_tabu with a cap on the iterations), chains of TSP_TABU_CHAIN iterations
per wait for the device against one per wait.  usage: driver_time2.py [instance] [iterations]   (through gpurun)"""
import ctypes as C, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import HostInstance, Instance
from tsp_optimization_amd.build import lib_path
name = sys.argv[1] if len(sys.argv) > 1 else "rand10000"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 600
L = C.CDLL(lib_path("libtsp_host.so"))
L.tsp_host_tabu.argtypes = [C.POINTER(Instance), C.c_int, C.c_longlong]
for chain in ("1", "8", "64", "auto"):
    if chain == "auto": os.environ.pop("TSP_TABU_CHAIN", None)
    else: os.environ["TSP_TABU_CHAIN"] = chain
    res = []
    for rep in range(2):
        h = HostInstance(name)
        h.c.params.time_limit = 3600
        C.CDLL(None).srandom(123)
        t0 = time.perf_counter()
        L.tsp_host_tabu(C.byref(h.c), 0, 40)            # the initial solution + warm-up
        t1 = time.perf_counter()
        h2 = HostInstance(name)
        h2.c.params.time_limit = 3600
        C.CDLL(None).srandom(123)
        t2 = time.perf_counter()
        L.tsp_host_tabu(C.byref(h2.c), 0, 40 + iters)
        t3 = time.perf_counter()
        res.append(iters / ((t3 - t2) - (t1 - t0)))
    print("%s: chain %4s: %.0f iterations/s (incumbent %.0f)" % (name, chain, max(res), h2.obj), flush=True)
