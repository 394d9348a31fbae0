"""Does the first use of the device disturb libc's random() stream?  A process that seeds (srandom) and then calls a heuristic
whose first act is device work (tabu(), HEU_VNS: HEU_2opt_greedy_iter before the first draw) must see the values the seed promises.
usage (through gpurun): rng_probe.py"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
libc = C.CDLL(None)
libc.random.restype = C.c_long
libc.srandom(123)
want = [libc.random() for _ in range(4)]
from helpers import HostInstance, Instance
from tsp_optimization_amd.build import lib_path
L = C.CDLL(lib_path("libtsp_host.so"))
h = HostInstance("pr299")
libc.srandom(123)
L.HEU_greedy.argtypes = [C.POINTER(Instance)]
L.HEU_greedy(C.byref(h.c))          # the process's first device work: no draw in it
got = [libc.random() for _ in range(4)]
print("after the first device call:", "stream intact" if got == want else "STREAM DISTURBED", want[:2], got[:2])
libc.srandom(123)
L.HEU_2opt_greedy.argtypes = [C.POINTER(Instance)]
L.HEU_2opt_greedy(C.byref(h.c))     # more first-time work (other kernels' code objects)
got = [libc.random() for _ in range(4)]
print("after a first 2-opt:", "stream intact" if got == want else "STREAM DISTURBED", want[:2], got[:2])
