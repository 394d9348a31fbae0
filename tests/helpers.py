"""Shared helpers for the test-suite (inputs, golden files).  The oracle is imported here only
because tests are one of the three places allowed to use it."""
import ctypes as C
import json
import os

import numpy as np

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
INSTANCES = os.path.join(GOLDEN, "instances")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def rand_instance(n, seed=None, hi=1_000_000):
    """uniform integer coordinates in [0, 1e6)^2, numpy PCG64 seeded with n."""
    rng = np.random.default_rng(n if seed is None else seed)
    return rng.integers(0, hi, size=(n, 2)).astype(np.float64)


def load_instance(name):
    """-> (xy, wtype) for a TSPLIB fixture name or 'rand<n>'."""
    if name.startswith("rand"):
        return rand_instance(int(name[4:])), O.EUC_2D
    return O.parse_tsplib(os.path.join(INSTANCES, name + ".tsp"))


def random_tour(n, rng):
    perm = rng.permutation(n).astype(np.int32)
    succ = np.empty(n, dtype=np.int32)
    succ[perm] = np.roll(perm, -1)
    return succ


# ---- ctypes view of include/utility.h:113-160 as restated in host/tsp_host.h ---------------------
class SolMethod(C.Structure):
    _fields_ = [("id", C.c_int), ("edge_type", C.c_int), ("name", C.c_char_p), ("use_cplex", C.c_int)]


class Params(C.Structure):
    _fields_ = [("file_path", C.c_char_p), ("num_threads", C.c_int), ("time_limit", C.c_int),
                ("method", SolMethod), ("verbose", C.c_int), ("integer_cost", C.c_int), ("seed", C.c_int),
                ("perf_prof", C.c_int), ("callback_2opt", C.c_int)]


class Edge(C.Structure):
    _fields_ = [("i", C.c_int), ("j", C.c_int)]


class Solution(C.Structure):
    _fields_ = [("obj_best", C.c_double), ("edges", C.POINTER(Edge)), ("time_to_solve", C.c_double),
                ("xbest", C.POINTER(C.c_double))]


class Instance(C.Structure):
    _fields_ = [("params", Params), ("name", C.c_char_p), ("comment", C.c_char_p),
                ("nodes", C.POINTER(C.c_double)), ("num_nodes", C.c_int), ("weight_type", C.c_int),
                ("num_columns", C.c_long), ("ind", C.POINTER(C.c_int)), ("thread_seeds", C.POINTER(C.c_uint)),
                ("solution", Solution)]


class HostInstance:
    """Owns the numpy buffers an `instance` points into."""

    def __init__(self, name, integer_cost=1):
        self.xy, self.wt = load_instance(name)
        self.n = len(self.xy)
        self.edges = np.zeros((self.n, 2), dtype=np.int32)
        self.c = Instance()
        self.c.params.time_limit = -1
        self.c.params.integer_cost = integer_cost
        self.c.params.seed = 123
        self.c.params.verbose = 0
        self.c.params.perf_prof = 1
        self.c.nodes = self.xy.ctypes.data_as(C.POINTER(C.c_double))
        self.c.num_nodes = self.n
        self.c.weight_type = self.wt
        self.c.num_columns = self.n * (self.n - 1) // 2
        self.c.solution.edges = self.edges.ctypes.data_as(C.POINTER(Edge))

    @property
    def succ(self):
        return self.edges[:, 1].copy()

    @property
    def obj(self):
        return self.c.solution.obj_best

    def set_tour(self, succ, obj):
        self.edges[:, 0] = np.arange(self.n)
        self.edges[:, 1] = succ
        self.c.solution.obj_best = obj
