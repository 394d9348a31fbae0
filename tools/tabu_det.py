"""Determinism of a capped tabu() run: three runs in one process, incumbent and loop time of each."""
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
L.tsp_host_random_lookahead.restype = C.c_int
for rep in range(3):
    h = HostInstance(name)
    h.c.params.time_limit = 3600
    C.CDLL(None).srandom(123)
    L.tsp_host_tabu(C.byref(h.c), 0, iters)
    print("rep %d: incumbent %.0f, loop %.3f s, window %d, next random %d" % (rep, h.obj, L.tsp_host_last_driver_loop_seconds(), L.tsp_host_random_lookahead(), C.CDLL(None).random()), flush=True)
