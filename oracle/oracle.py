"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (tsp_optimization_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

EUC_2D, MAX_2D, MAN_2D, CEIL_2D, GEO, ATT = 0, 1, 2, 3, 4, 5
WTYPE_NAMES = {"EUC_2D": EUC_2D, "MAX_2D": MAX_2D, "MAN_2D": MAN_2D, "CEIL_2D": CEIL_2D,
               "GEO": GEO, "ATT": ATT}


class Stats(C.Structure):
    _fields_ = [("sweeps", C.c_longlong), ("evals", C.c_longlong), ("moves", C.c_longlong),
                ("reversed", C.c_longlong), ("seconds", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Move(C.Structure):
    _fields_ = [("i", C.c_int), ("j", C.c_int), ("delta", C.c_double)]


def build(force=False):
    """Compile liboracle.so with gcc (no-op when it is up to date)."""
    src = os.path.join(_HERE, "tsp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src),
                                              os.path.getmtime(os.path.join(_HERE, "tsp_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.orc_dist.restype = C.c_double
        L.orc_dist.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_dist_matrix.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp]
        L.orc_greedy.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp]
        L.orc_grasp.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int, dp, ip, dp]
        L.orc_greedy_iter.argtypes = [dp, C.c_int, C.c_int, C.c_int, ip, dp]
        L.orc_grasp_iter_prefix.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_longlong, ip, dp, C.POINTER(C.c_longlong)]
        L.orc_extramileage.argtypes = [dp, C.c_int, C.c_int, C.c_int, ip, dp]
        L.orc_two_opt_first.argtypes = [dp, C.c_int, C.c_int, C.c_int, ip, dp, C.c_double, C.c_int,
                                        C.POINTER(Stats), C.POINTER(Move), C.c_longlong]
        L.orc_two_opt_first_moves.argtypes = [dp, C.c_int, C.c_int, C.c_int, ip, dp, C.c_longlong, C.POINTER(Stats)]
        L.orc_two_opt_best.argtypes = [dp, C.c_int, C.c_int, C.c_int, ip, dp, ip, C.c_int, C.c_int,
                                       ip, C.c_double, C.c_longlong, C.POINTER(Stats),
                                       C.POINTER(Move), C.c_longlong]
        L.orc_reverse_path.argtypes = [C.c_int, ip, C.c_int, C.c_int, ip]
        L.orc_perm_cost.restype = C.c_double
        L.orc_perm_cost.argtypes = [dp, C.c_int, C.c_int, C.c_int, ip]
        L.orc_succ_cost.restype = C.c_double
        L.orc_succ_cost.argtypes = [dp, C.c_int, C.c_int, C.c_int, ip]
        L.orc_perm_to_succ.argtypes = [C.c_int, ip, ip]
        L.orc_succ_to_perm.argtypes = [C.c_int, ip, ip]
        L.orc_random_perm.argtypes = [C.c_int, ip]
        L.orc_udir_pos.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_vns_kick.argtypes = [dp, C.c_int, C.c_int, C.c_int, ip, dp]
        L.orc_vns.argtypes = [dp, C.c_int, C.c_int, C.c_int, ip, dp, C.c_longlong, C.POINTER(C.c_longlong)]
        L.orc_tabu.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp, C.c_longlong,
                               C.POINTER(C.c_longlong)]
        L.orc_genetic.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_longlong, ip, dp]
        L.orc_genetic_ex.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_double, ip, dp]
        L.orc_srandom.argtypes = [C.c_uint]
        L.orc_urand.restype = C.c_double
        L.orc_parse_tsplib.argtypes = [C.c_char_p, dp, C.c_int, ip]
        L.orc_fnv1a.restype = C.c_ulonglong
        L.orc_fnv1a.argtypes = [ip, C.c_int]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _xy(xy):
    xy = np.ascontiguousarray(xy, dtype=np.float64)
    assert xy.ndim == 2 and xy.shape[1] == 2
    return xy


def parse_tsplib(path):
    """-> (xy float64 [n,2], wtype)"""
    wt = C.c_int(-1)
    n = lib().orc_parse_tsplib(path.encode(), None, 0, C.byref(wt))
    if n <= 0:
        raise ValueError("cannot parse %s (%d)" % (path, n))
    xy = np.zeros((n, 2), dtype=np.float64)
    lib().orc_parse_tsplib(path.encode(), _d(xy), n, C.byref(wt))
    return xy, wt.value


def dist(xy, i, j, wtype, integer_cost=1):
    xy = _xy(xy)
    return lib().orc_dist(_d(xy), i, j, wtype, integer_cost)


def dist_matrix(xy, wtype, integer_cost=1):
    xy = _xy(xy)
    n = len(xy)
    out = np.empty((n, n), dtype=np.float64)
    lib().orc_dist_matrix(_d(xy), n, wtype, integer_cost, _d(out))
    return out


def greedy(xy, wtype, start=0, integer_cost=1):
    xy = _xy(xy)
    n = len(xy)
    succ = np.zeros(n, dtype=np.int32)
    obj = C.c_double(0)
    st = lib().orc_greedy(_d(xy), n, wtype, integer_cost, start, _i(succ), C.byref(obj))
    return st, succ, obj.value


def grasp(xy, wtype, start=0, integer_cost=1, urand=None):
    """urand=None draws from libc random() (seed it with srandom first)."""
    xy = _xy(xy)
    n = len(xy)
    succ = np.zeros(n, dtype=np.int32)
    obj = C.c_double(0)
    up = None
    if urand is not None:
        urand = np.ascontiguousarray(urand, dtype=np.float64)
        assert len(urand) >= n
        up = _d(urand)
    st = lib().orc_grasp(_d(xy), n, wtype, integer_cost, start, up, _i(succ), C.byref(obj))
    return st, succ, obj.value


def grasp_iter_prefix(xy, wtype, starts, integer_cost=1):
    """HEU_Grasp_iter for exactly `starts` starts of the libc stream (seed it first) -> (succ, obj, index of the best start)."""
    xy = _xy(xy)
    n = len(xy)
    succ = np.zeros(n, dtype=np.int32)
    obj, k = C.c_double(0), C.c_longlong(-1)
    lib().orc_grasp_iter_prefix(_d(xy), n, wtype, integer_cost, C.c_longlong(starts), _i(succ), C.byref(obj), C.byref(k))
    return succ, obj.value, k.value


def greedy_iter(xy, wtype, integer_cost=1):
    xy = _xy(xy)
    n = len(xy)
    succ = np.zeros(n, dtype=np.int32)
    obj = C.c_double(0)
    st = lib().orc_greedy_iter(_d(xy), n, wtype, integer_cost, _i(succ), C.byref(obj))
    return st, succ, obj.value


def extramileage(xy, wtype, integer_cost=1):
    xy = _xy(xy)
    n = len(xy)
    succ = np.zeros(n, dtype=np.int32)
    obj = C.c_double(0)
    st = lib().orc_extramileage(_d(xy), n, wtype, integer_cost, _i(succ), C.byref(obj))
    return st, succ, obj.value


def _trace_out(tr, nmoves, cap):
    k = int(min(nmoves, cap))
    return [(tr[m].i, tr[m].j, tr[m].delta) for m in range(k)]


def two_opt_first(xy, wtype, succ, obj, integer_cost=1, time_limit=-1.0, clock_per_pair=0,
                  trace_cap=0):
    """-> (status, succ', obj', stats dict, trace list)"""
    xy = _xy(xy)
    n = len(xy)
    succ = np.array(succ, dtype=np.int32, copy=True)
    o = C.c_double(obj)
    st = Stats()
    tr = (Move * max(1, trace_cap))()
    status = lib().orc_two_opt_first(_d(xy), n, wtype, integer_cost, _i(succ), C.byref(o),
                                     time_limit, clock_per_pair, C.byref(st), tr, trace_cap)
    return status, succ, o.value, st.as_dict(), _trace_out(tr, st.moves, trace_cap)


def two_opt_first_moves(xy, wtype, succ, obj, max_moves, integer_cost=1):
    """-> (succ', obj', stats) after exactly max_moves moves of the first-improvement trajectory (or fewer at the optimum)"""
    xy = _xy(xy)
    succ = np.array(succ, dtype=np.int32, copy=True)
    o = C.c_double(obj)
    st = Stats()
    lib().orc_two_opt_first_moves(_d(xy), len(xy), wtype, integer_cost, _i(succ), C.byref(o), max_moves, C.byref(st))
    return succ, o.value, st.as_dict()


def two_opt_best(xy, wtype, succ, obj=0.0, integer_cost=1, tabu=None, iter_=1, tenure=0,
                 want_prev=False, time_limit=-1.0, max_sweeps=-1, trace_cap=0):
    """-> (status, succ', obj', stats dict, trace list, prev or None).  tabu is modified in place."""
    xy = _xy(xy)
    n = len(xy)
    succ = np.array(succ, dtype=np.int32, copy=True)
    o = C.c_double(obj)
    st = Stats()
    tr = (Move * max(1, trace_cap))()
    tp = None
    if tabu is not None:
        assert tabu.dtype == np.int32 and tabu.flags.c_contiguous and len(tabu) == n * (n - 1) // 2
        tp = _i(tabu)
    prev = np.zeros(n, dtype=np.int32) if want_prev else None
    status = lib().orc_two_opt_best(_d(xy), n, wtype, integer_cost, _i(succ), C.byref(o), tp, iter_,
                                    tenure, _i(prev) if want_prev else None, time_limit, max_sweeps,
                                    C.byref(st), tr, trace_cap)
    return status, succ, o.value, st.as_dict(), _trace_out(tr, st.moves, trace_cap), prev


def vns_kick(xy, wtype, succ, integer_cost=1):
    """-> (succ', obj') after one kick (draws from libc random())"""
    xy = _xy(xy)
    succ = np.array(succ, dtype=np.int32, copy=True)
    o = C.c_double(0)
    lib().orc_vns_kick(_d(xy), len(xy), wtype, integer_cost, _i(succ), C.byref(o))
    return succ, o.value


def vns(xy, wtype, succ, obj, rounds, integer_cost=1):
    """-> (succ', obj', rounds that improved the incumbent)"""
    xy = _xy(xy)
    succ = np.array(succ, dtype=np.int32, copy=True)
    o = C.c_double(obj)
    imp = C.c_longlong(0)
    lib().orc_vns(_d(xy), len(xy), wtype, integer_cost, _i(succ), C.byref(o), rounds, C.byref(imp))
    return succ, o.value, imp.value


def tabu(xy, wtype, succ, obj, iterations, policy=0, integer_cost=1):
    """-> (succ', obj', total best-improvement moves)"""
    xy = _xy(xy)
    succ = np.array(succ, dtype=np.int32, copy=True)
    o = C.c_double(obj)
    mv = C.c_longlong(0)
    lib().orc_tabu(_d(xy), len(xy), wtype, integer_cost, policy, _i(succ), C.byref(o), iterations, C.byref(mv))
    return succ, o.value, mv.value


def genetic(xy, wtype, generations, integer_cost=1, two_opt_prob=0.0):
    """-> (incumbent succ, incumbent cost) after `generations` generations (draws from libc random());
    two_opt_prob = probability of mutation method 3 (0.00 in the reference, genetic.c:18)"""
    xy = _xy(xy)
    succ = np.zeros(len(xy), dtype=np.int32)
    o = C.c_double(0)
    lib().orc_genetic_ex(_d(xy), len(xy), wtype, integer_cost, generations, two_opt_prob, _i(succ), C.byref(o))
    return succ, o.value


def perm_cost(xy, wtype, perm, integer_cost=1):
    xy = _xy(xy)
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    return lib().orc_perm_cost(_d(xy), len(xy), wtype, integer_cost, _i(perm))


def succ_cost(xy, wtype, succ, integer_cost=1):
    xy = _xy(xy)
    succ = np.ascontiguousarray(succ, dtype=np.int32)
    return lib().orc_succ_cost(_d(xy), len(xy), wtype, integer_cost, _i(succ))


def perm_to_succ(perm):
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    succ = np.zeros(len(perm), dtype=np.int32)
    lib().orc_perm_to_succ(len(perm), _i(perm), _i(succ))
    return succ


def succ_to_perm(succ):
    succ = np.ascontiguousarray(succ, dtype=np.int32)
    perm = np.zeros(len(succ), dtype=np.int32)
    lib().orc_succ_to_perm(len(succ), _i(succ), _i(perm))
    return perm


def random_perm(n):
    perm = np.zeros(n, dtype=np.int32)
    lib().orc_random_perm(n, _i(perm))
    return perm


def srandom(seed):
    lib().orc_srandom(seed)


def urand():
    return lib().orc_urand()


def fnv1a(v):
    v = np.ascontiguousarray(v, dtype=np.int32)
    return int(lib().orc_fnv1a(_i(v), len(v)))


def is_tour(succ):
    """True iff succ is one Hamiltonian cycle."""
    n = len(succ)
    seen = np.zeros(n, dtype=bool)
    v = 0
    for _ in range(n):
        if seen[v]:
            return False
        seen[v] = True
        v = int(succ[v])
    return v == 0 and bool(seen.all())
