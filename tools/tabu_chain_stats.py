import ctypes as C, os, sys
R="/root/repo"
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import HostInstance, Instance
from tsp_optimization_amd.build import lib_path
L = C.CDLL(lib_path("libtsp_host.so"))
L.tsp_host_tabu.argtypes = [C.POINTER(Instance), C.c_int, C.c_longlong]
os.environ["TSP_HOST_STATS"]="1"
for chain in ("8","64",None):
    if chain: os.environ["TSP_TABU_CHAIN"]=chain
    else: os.environ.pop("TSP_TABU_CHAIN",None)
    for cnt in (200, 2200):
        h = HostInstance(sys.argv[1]); h.c.params.time_limit = 3600
        C.CDLL(None).srandom(123)
        L.tsp_host_tabu(C.byref(h.c), 0, cnt)
