"""ctypes binding of include/tsp_hip.h (libtsp_hip.so).  No fallback of any kind: if the library
is missing or no MI355X is visible the calls raise."""
import ctypes as C
import os

import numpy as np

from .build import lib_path

EUC_2D, MAX_2D, MAN_2D, CEIL_2D, GEO, ATT = 0, 1, 2, 3, 4, 5
FIRST, BEST = 0, 1
GREEDY, GRASP = 0, 1
ENGINE_AUTO, ENGINE_GRID, ENGINE_LDS, ENGINE_CLUSTER = 0, 1, 2, 3
OK, WRONG_STARTING_NODE, TIME_LIMIT_EXCEEDED = 0, 1, 2


class TspDeviceError(RuntimeError):
    pass


class Stats(C.Structure):
    _fields_ = [("sweeps", C.c_int64), ("evals", C.c_int64), ("moves", C.c_int64),
                ("reversed", C.c_int64), ("pairs_scanned", C.c_int64), ("steps", C.c_int64),
                ("seconds", C.c_double), ("device_ms", C.c_double),
                ("lane_pairs", C.c_int64), ("tier1_pairs", C.c_int64), ("exact_pairs", C.c_int64),
                ("staged_recs", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    """Loads libtsp_hip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise TspDeviceError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        L = C.CDLL(path)
        vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
        sp = C.POINTER(Stats)
        L.tsp_dev_last_error.restype = C.c_char_p
        L.tsp_dev_open.argtypes = [C.c_int, C.POINTER(vp)]
        L.tsp_dev_close.argtypes = [vp]
        L.tsp_dev_close.restype = None
        L.tsp_dev_synchronize.argtypes = [vp]
        L.tsp_dev_stream.argtypes = [vp]
        L.tsp_dev_stream.restype = vp
        L.tsp_dev_inst_create.argtypes = [vp, dp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.tsp_dev_inst_destroy.argtypes = [vp]
        L.tsp_dev_inst_destroy.restype = None
        L.tsp_dev_inst_size.argtypes = [vp]
        L.tsp_dev_inst_reload_switches.argtypes = [vp]
        L.tsp_dev_dist_pairs.argtypes = [vp, ip, ip, C.c_int, dp]
        L.tsp_dev_selftest_raw_sqrt.argtypes = [vp, dp, C.c_int, dp]
        L.tsp_dev_dist_matrix.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_float)]
        L.tsp_dev_construct.argtypes = [vp, C.c_int, C.c_int, ip, dp, ip, C.c_int, C.c_int64, dp, ip]
        L.tsp_dev_extramileage.argtypes = [vp, ip, C.c_int, dp]
        L.tsp_dev_two_opt.argtypes = [vp, C.c_int, C.c_int, C.c_int, ip, C.c_int, C.c_int64, dp,
                                      C.c_double, sp]
        L.tsp_dev_tabu_create.argtypes = [vp, C.POINTER(vp)]
        L.tsp_dev_tabu_destroy.argtypes = [vp]
        L.tsp_dev_tabu_destroy.restype = None
        L.tsp_dev_tabu_set.argtypes = [vp, ip, ip, C.c_int]
        L.tsp_dev_tabu_get.argtypes = [vp, ip, ip, C.c_int]
        L.tsp_dev_tabu_upload.argtypes = [vp, ip]
        L.tsp_dev_tabu_download.argtypes = [vp, ip]
        L.tsp_dev_tabu_list_info.argtypes = [vp, ip, ip]
        L.tsp_dev_two_opt_tabu.argtypes = [vp, vp, C.c_int, C.c_int, ip, C.c_int, dp, ip, C.c_double, sp]
        L.tsp_dev_perm_cost.argtypes = [vp, C.c_int, ip, C.c_int64, dp]
        L.tsp_dev_tours_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.tsp_dev_tours_destroy.argtypes = [vp]
        L.tsp_dev_tours_destroy.restype = None
        L.tsp_dev_tours_upload.argtypes = [vp, ip, C.c_int, C.c_int64, dp]
        L.tsp_dev_tours_reset.argtypes = [vp]
        L.tsp_dev_tours_download.argtypes = [vp, ip, C.c_int, C.c_int64, dp, sp]
        L.tsp_dev_tours_run.argtypes = [vp, C.c_int, C.c_int64, C.c_double, C.c_int, ip]
        L.tsp_dev_tours_run_engine.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.c_double, ip]
        L.tsp_dev_tours_two_opt.argtypes = [vp, C.c_int, C.c_int, C.c_double, dp]
        L.tsp_dev_tours_two_opt_tabu.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, dp]
        L.tsp_dev_tours_tabu_kick.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, ip]
        L.tsp_dev_tours_tabu_iteration.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, dp, dp, ip, ip]
        L.tsp_dev_tours_tabu_iterations.argtypes = [vp, vp, C.c_int, C.c_int, ip, ip, C.c_double, dp, dp, ip, ip, ip]
        L.tsp_dev_tours_tabu_iterations_ex.argtypes = [vp, vp, C.c_int, C.c_int, ip, C.c_int, ip, C.c_double, dp, dp, ip, ip, ip, ip]
        L.tsp_dev_tours_vns_kick.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp]
        L.tsp_dev_tours_snapshot.argtypes = [vp]
        L.tsp_dev_tours_restore.argtypes = [vp]
        L.tsp_dev_tours_time_scan.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        L.tsp_dev_tours_describe.argtypes = [vp, C.c_int, C.c_char_p, C.c_int]
        L.tsp_dev_tours_device_ms.argtypes = [vp, C.POINTER(C.c_double)]
        L.tsp_dev_tours_best.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
        L.tsp_dev_comm_last_error.restype = C.c_char_p
        L.tsp_dev_comm_unique_id.argtypes = [C.c_char_p]
        L.tsp_dev_comm_init_rank.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]
        L.tsp_dev_comm_init_all.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp)]
        L.tsp_dev_comm_destroy.argtypes = [vp]
        L.tsp_dev_comm_destroy.restype = None
        L.tsp_dev_comm_info.argtypes = [vp, ip, ip, ip]
        L.tsp_dev_multistart_pack.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_int64)]
        L.tsp_dev_multistart_allreduce.argtypes = [vp, C.c_int64, C.POINTER(C.c_int64)]
        L.tsp_dev_multistart_allreduce_f64.argtypes = [vp, C.c_double, C.POINTER(C.c_double)]
        L.tsp_dev_multistart_allreduce_f64_group.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.tsp_dev_multistart_bcast_tour.argtypes = [vp, C.c_int, ip, C.c_int, C.c_int]
        L.tsp_dev_multistart_allreduce_group.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.tsp_dev_multistart_bcast_tour_group.argtypes = [C.POINTER(vp), C.c_int, C.c_int, ip, C.c_int, C.c_int, C.c_int, ip]
        _lib = L
    return _lib


EXPORTED = [
    "tsp_dev_open", "tsp_dev_close", "tsp_dev_last_error", "tsp_dev_count", "tsp_dev_synchronize",
    "tsp_dev_stream", "tsp_dev_inst_create", "tsp_dev_inst_destroy", "tsp_dev_inst_size", "tsp_dev_inst_reload_switches",
    "tsp_dev_dist_pairs", "tsp_dev_selftest_raw_sqrt", "tsp_dev_dist_matrix", "tsp_dev_construct", "tsp_dev_extramileage", "tsp_dev_two_opt",
    "tsp_dev_tabu_create", "tsp_dev_tabu_destroy", "tsp_dev_tabu_set", "tsp_dev_tabu_get",
    "tsp_dev_tabu_upload", "tsp_dev_tabu_download", "tsp_dev_tabu_list_info", "tsp_dev_two_opt_tabu", "tsp_dev_perm_cost",
    "tsp_dev_tours_create", "tsp_dev_tours_destroy", "tsp_dev_tours_upload", "tsp_dev_tours_reset",
    "tsp_dev_tours_download", "tsp_dev_tours_run", "tsp_dev_tours_run_engine", "tsp_dev_tours_time_scan", "tsp_dev_tours_best", "tsp_dev_tours_describe", "tsp_dev_tours_device_ms",
    "tsp_dev_tours_two_opt", "tsp_dev_tours_two_opt_tabu", "tsp_dev_tours_tabu_kick", "tsp_dev_tours_tabu_iteration", "tsp_dev_tours_tabu_iterations", "tsp_dev_tours_tabu_iterations_ex", "tsp_dev_tours_vns_kick",
    "tsp_dev_tours_snapshot", "tsp_dev_tours_restore", "tsp_dev_host_register", "tsp_dev_host_unregister",
    "tsp_dev_comm_last_error", "tsp_dev_comm_available", "tsp_dev_comm_unique_id", "tsp_dev_comm_init_rank", "tsp_dev_comm_init_all", "tsp_dev_comm_destroy",
    "tsp_dev_comm_info", "tsp_dev_multistart_pack", "tsp_dev_multistart_allreduce", "tsp_dev_multistart_bcast_tour",
    "tsp_dev_multistart_allreduce_group", "tsp_dev_multistart_bcast_tour_group",
    "tsp_dev_multistart_allreduce_f64", "tsp_dev_multistart_allreduce_f64_group",
]

COMM_ID_BYTES = 128
E_COMM = -6


def _check(rc, allow=(OK,)):
    if rc in allow:
        return rc
    msg = lib().tsp_dev_last_error().decode()
    if rc == E_COMM:
        msg = lib().tsp_dev_comm_last_error().decode() + " " + msg
    raise TspDeviceError("tsp_dev call failed with %d %s" % (rc, msg))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def device_count():
    return lib().tsp_dev_count()


class Context:
    """One device + the engine's stream (tsp_dev_open)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().tsp_dev_open(device, C.byref(self._h)))
        self.device = device

    def synchronize(self):
        _check(lib().tsp_dev_synchronize(self._h))

    def raw_sqrt(self, x):
        """The hardware's approximate v_sqrt_f64 (self-test of the integer-root variants' premise)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        _check(lib().tsp_dev_selftest_raw_sqrt(self._h, _d(x), len(x), _d(out)))
        return out

    def close(self):
        if self._h:
            lib().tsp_dev_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Instance:
    """Node coordinates resident in HBM (tsp_dev_inst_create)."""

    def __init__(self, ctx, xy, wtype, integer_cost=1):
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        assert xy.ndim == 2 and xy.shape[1] == 2
        self.ctx, self.n, self.wtype, self.integer_cost = ctx, len(xy), wtype, integer_cost
        self._h = C.c_void_p()
        _check(lib().tsp_dev_inst_create(ctx._h, _d(xy), self.n, wtype, integer_cost, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().tsp_dev_inst_destroy(self._h)
            self._h = C.c_void_p()

    def reload_switches(self):
        """Re-read the TSP_* environment switches for this instance (they are read once, at creation)."""
        _check(lib().tsp_dev_inst_reload_switches(self._h))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- calc_dist ------------------------------------------------------------------------
    def dist_pairs(self, i, j):
        i = np.ascontiguousarray(i, dtype=np.int32)
        j = np.ascontiguousarray(j, dtype=np.int32)
        out = np.empty(len(i), dtype=np.float64)
        _check(lib().tsp_dev_dist_pairs(self._h, _i(i), _i(j), len(i), _d(out)))
        return out

    def dist_matrix(self, as_int32=False, fetch=True):
        """-> (matrix or None, kernel_ms)"""
        n = self.n
        out = np.empty((n, n), dtype=np.int32 if as_int32 else np.float64) if fetch else None
        ms = C.c_float(0)
        _check(lib().tsp_dev_dist_matrix(self._h, out.ctypes.data_as(C.c_void_p) if fetch else None,
                                         1 if as_int32 else 0, C.byref(ms)))
        return out, ms.value

    # -- greedy / grasp -------------------------------------------------------------------
    def construct(self, kind, starts, urand=None):
        """-> (succ [B,n] int32, obj [B], status [B])"""
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        B, n = len(starts), self.n
        succ = np.zeros((B, n), dtype=np.int32)
        obj = np.zeros(B, dtype=np.float64)
        status = np.zeros(B, dtype=np.int32)
        up = None
        if kind == GRASP:
            urand = np.ascontiguousarray(urand, dtype=np.float64)
            assert urand.shape == (B, n)
            up = _d(urand)
        _check(lib().tsp_dev_construct(self._h, kind, B, _i(starts), up, _i(succ), 1, n, _d(obj), _i(status)),
               allow=(OK, WRONG_STARTING_NODE))
        return succ, obj, status

    def extramileage(self):
        """HEU_extramileage -> (succ [n] int32, obj)"""
        succ = np.zeros(self.n, dtype=np.int32)
        obj = C.c_double(0)
        _check(lib().tsp_dev_extramileage(self._h, _i(succ), 1, C.byref(obj)))
        return succ, obj.value

    # -- alg_2opt / alg_2opt_tabu(NULL) ---------------------------------------------------
    def two_opt(self, succ, obj, mode=FIRST, engine=ENGINE_AUTO, time_limit=-1.0):
        """succ [n] or [B,n]; obj scalar or [B].  -> (status, succ', obj', [stats dict])"""
        succ = np.array(succ, dtype=np.int32, copy=True, order="C")
        single = succ.ndim == 1
        succ2 = succ.reshape(1, -1) if single else succ
        B, n = succ2.shape
        assert n == self.n
        o = np.array(np.broadcast_to(np.asarray(obj, dtype=np.float64), (B,)), copy=True)
        st = (Stats * B)()
        rc = lib().tsp_dev_two_opt(self._h, mode, engine, B, _i(succ2), 1, n, _d(o), time_limit, st)
        _check(rc, allow=(OK, TIME_LIMIT_EXCEEDED))
        stats = [s.as_dict() for s in st]
        if single:
            return rc, succ2[0], float(o[0]), stats[0]
        return rc, succ2, o, stats

    def perm_cost(self, perms):
        perms = np.ascontiguousarray(perms, dtype=np.int32)
        if perms.ndim == 1:
            perms = perms.reshape(1, -1)
        B, n = perms.shape
        cost = np.zeros(B, dtype=np.float64)
        _check(lib().tsp_dev_perm_cost(self._h, B, _i(perms), n, _d(cost)))
        return cost


class Tabu:
    """n(n-1)/2 tabu stamps resident in HBM (tabusearch.c:195)."""

    def __init__(self, inst):
        self.inst = inst
        self._h = C.c_void_p()
        _check(lib().tsp_dev_tabu_create(inst._h, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().tsp_dev_tabu_destroy(self._h)
            self._h = C.c_void_p()

    def set(self, idx, value):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        value = np.ascontiguousarray(value, dtype=np.int32)
        _check(lib().tsp_dev_tabu_set(self._h, _i(idx), _i(value), len(idx)))

    def get(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        out = np.zeros(len(idx), dtype=np.int32)
        _check(lib().tsp_dev_tabu_get(self._h, _i(idx), _i(out), len(idx)))
        return out

    def upload(self, stamps):
        stamps = np.ascontiguousarray(stamps, dtype=np.int32)
        n = self.inst.n
        assert len(stamps) == n * (n - 1) // 2
        _check(lib().tsp_dev_tabu_upload(self._h, _i(stamps)))

    def download(self):
        n = self.inst.n
        out = np.zeros(n * (n - 1) // 2, dtype=np.int32)
        _check(lib().tsp_dev_tabu_download(self._h, _i(out)))
        return out

    def list_info(self):
        """-> (entries of the compact list of non-zero stamps or -1 while out of date, last run worked from the list)"""
        e, u = C.c_int(0), C.c_int(0)
        _check(lib().tsp_dev_tabu_list_info(self._h, C.byref(e), C.byref(u)))
        return e.value, bool(u.value)

    def two_opt(self, succ, iter_, tenure, want_prev=False, time_limit=-1.0):
        """alg_2opt_tabu(inst, skip_edge, stored_prev, iter, tenure) -> (status, succ', obj', stats, prev)"""
        succ = np.array(succ, dtype=np.int32, copy=True, order="C")
        o = C.c_double(0.0)
        st = Stats()
        prev = np.zeros(self.inst.n, dtype=np.int32) if want_prev else None
        rc = lib().tsp_dev_two_opt_tabu(self.inst._h, self._h, iter_, tenure, _i(succ), 1, C.byref(o),
                                        _i(prev) if want_prev else None, time_limit, C.byref(st))
        _check(rc, allow=(OK, TIME_LIMIT_EXCEEDED))
        return rc, succ, o.value, st.as_dict(), prev


class Tours:
    """B tours resident in HBM (what bench.py times)."""

    def __init__(self, inst, B=1):
        self.inst, self.B = inst, B
        self._h = C.c_void_p()
        _check(lib().tsp_dev_tours_create(inst._h, B, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().tsp_dev_tours_destroy(self._h)
            self._h = C.c_void_p()

    def upload(self, succ, obj):
        succ = np.ascontiguousarray(succ, dtype=np.int32).reshape(self.B, self.inst.n)
        o = np.array(np.broadcast_to(np.asarray(obj, dtype=np.float64), (self.B,)), copy=True)
        _check(lib().tsp_dev_tours_upload(self._h, _i(succ), 1, self.inst.n, _d(o)))

    def reset(self):
        _check(lib().tsp_dev_tours_reset(self._h))

    def run(self, mode, max_steps=-1, time_limit=-1.0, sync=True):
        """-> (status, all_done).  max_steps < 0 (until done) needs sync=True."""
        if not sync and max_steps < 0:
            raise ValueError("an unbounded run needs sync=True (it is driven by polls of the done flags)")
        done = C.c_int(0)
        rc = lib().tsp_dev_tours_run(self._h, mode, max_steps, time_limit, 1 if sync else 0, C.byref(done))
        _check(rc, allow=(OK, TIME_LIMIT_EXCEEDED))
        return rc, bool(done.value)

    def device_ms(self):
        """Device time of the last run_engine / two_opt on this handle (HIP events on the engine's stream)."""
        ms = C.c_double(0)
        _check(lib().tsp_dev_tours_device_ms(self._h, C.byref(ms)))
        return ms.value

    def describe(self, mode):
        """The kernels one GRID-engine step of `mode` launches for this handle."""
        buf = C.create_string_buffer(256)
        _check(lib().tsp_dev_tours_describe(self._h, mode, buf, 256))
        return buf.value.decode()

    def run_engine(self, mode, engine=ENGINE_AUTO, max_steps=-1, time_limit=-1.0):
        """Resident tours on a chosen engine, waits for completion.  -> (status, all_done)"""
        done = C.c_int(0)
        rc = lib().tsp_dev_tours_run_engine(self._h, mode, engine, max_steps, time_limit, C.byref(done))
        _check(rc, allow=(OK, TIME_LIMIT_EXCEEDED))
        return rc, bool(done.value)

    # -- drivers on resident tours -------------------------------------------------------------
    def two_opt(self, mode, engine=ENGINE_AUTO, time_limit=-1.0):
        """alg_2opt / alg_2opt_tabu(NULL) on the tours as they are -> (status, obj [B])"""
        obj = np.zeros(self.B, dtype=np.float64)
        rc = lib().tsp_dev_tours_two_opt(self._h, mode, engine, time_limit, _d(obj))
        _check(rc, allow=(OK, TIME_LIMIT_EXCEEDED))
        return rc, obj

    def two_opt_tabu(self, tabu, iter_, tenure, time_limit=-1.0):
        obj = C.c_double(0)
        rc = lib().tsp_dev_tours_two_opt_tabu(self._h, tabu._h if tabu is not None else None, iter_, tenure, time_limit, C.byref(obj))
        _check(rc, allow=(OK, TIME_LIMIT_EXCEEDED))
        return rc, obj.value

    def tabu_kick(self, tabu, a, b, iter_, tenure):
        acc = C.c_int(0)
        _check(lib().tsp_dev_tours_tabu_kick(self._h, tabu._h, a, b, iter_, tenure, C.byref(acc)))
        return bool(acc.value)

    def tabu_iterations(self, tabu, iter0, tenures, ab, best_obj, time_limit=-1.0):
        """K = len(tenures) iterations of tabu() in one wait for the device (tsp_dev_tours_tabu_iterations)
        -> (status, completed, last_accepted, best_obj', obj [completed], improved [completed])"""
        K = len(tenures)
        ten = np.ascontiguousarray(tenures, dtype=np.int32)
        abv = np.ascontiguousarray(ab, dtype=np.int32).reshape(2 * K)
        best, obj, imp = C.c_double(best_obj), np.zeros(K), np.zeros(K, dtype=np.int32)
        done, acc = C.c_int(0), C.c_int(0)
        rc = lib().tsp_dev_tours_tabu_iterations(self._h, tabu._h, iter0, K, _i(ten), _i(abv), time_limit, C.byref(best), _d(obj), _i(imp),
                                                 C.byref(done), C.byref(acc))
        _check(rc, allow=(OK, TIME_LIMIT_EXCEEDED))
        return rc, done.value, bool(acc.value), best.value, obj[:done.value].copy(), imp[:done.value].copy()

    def tabu_iterations_ex(self, tabu, iter0, tenures, ab, best_obj, time_limit=-1.0):
        """K = len(tenures) iterations of tabu() in one launch with the kick's trials taken in order from the len(ab) >= K node pairs
        (tsp_dev_tours_tabu_iterations_ex) -> (status, completed, last_accepted, best_obj', obj [completed], improved [completed],
        trials [completed])"""
        K = len(tenures)
        ten = np.ascontiguousarray(tenures, dtype=np.int32)
        abv = np.ascontiguousarray(ab, dtype=np.int32)
        P = abv.size // 2
        abv = abv.reshape(2 * P)
        best, obj, imp, tri = C.c_double(best_obj), np.zeros(K), np.zeros(K, dtype=np.int32), np.zeros(K, dtype=np.int32)
        done, acc = C.c_int(0), C.c_int(0)
        rc = lib().tsp_dev_tours_tabu_iterations_ex(self._h, tabu._h, iter0, K, _i(ten), P, _i(abv), time_limit, C.byref(best), _d(obj), _i(imp),
                                                    _i(tri), C.byref(done), C.byref(acc))
        _check(rc, allow=(OK, TIME_LIMIT_EXCEEDED))
        d = done.value
        return rc, d, bool(acc.value), best.value, obj[:d].copy(), imp[:d].copy(), tri[:d].copy()

    def tabu_iteration(self, tabu, iter_, tenure, a, b, best_obj, time_limit=-1.0):
        """-> (status, obj, best_obj', improved, accepted): alg_2opt_tabu, incumbent, first kick trial (tabusearch.c:238-309)"""
        best, obj, imp, acc = C.c_double(best_obj), C.c_double(0), C.c_int(0), C.c_int(0)
        rc = lib().tsp_dev_tours_tabu_iteration(self._h, tabu._h, iter_, tenure, time_limit, a, b, C.byref(best), C.byref(obj),
                                                C.byref(imp), C.byref(acc))
        _check(rc, allow=(OK, TIME_LIMIT_EXCEEDED))
        return rc, obj.value, best.value, bool(imp.value), bool(acc.value)

    def vns_kick(self, p1, p2, p3):
        obj = C.c_double(0)
        _check(lib().tsp_dev_tours_vns_kick(self._h, p1, p2, p3, C.byref(obj)))
        return obj.value

    def snapshot(self):
        _check(lib().tsp_dev_tours_snapshot(self._h))

    def restore(self):
        _check(lib().tsp_dev_tours_restore(self._h))

    def download(self):
        """-> (succ [B,n], obj [B], [stats])"""
        n = self.inst.n
        succ = np.zeros((self.B, n), dtype=np.int32)
        obj = np.zeros(self.B, dtype=np.float64)
        st = (Stats * self.B)()
        _check(lib().tsp_dev_tours_download(self._h, _i(succ), 1, n, _d(obj), st))
        return succ, obj, [s.as_dict() for s in st]

    def time_scan(self, reps=20):
        """-> (mean kernel ms, evals per launch) for the BEST-mode sweep kernel"""
        ms = C.c_float(0)
        ev = C.c_int64(0)
        _check(lib().tsp_dev_tours_time_scan(self._h, reps, C.byref(ms), C.byref(ev)))
        return ms.value, ev.value

    def best(self, true_cost=True):
        """-> (cost, tour index) minimum over the handle's tours"""
        p = C.c_int64(0)
        _check(lib().tsp_dev_tours_best(self._h, 1 if true_cost else 0, C.byref(p)))
        return p.value >> 24, p.value & 0xFFFFFF, p.value


def multistart_pack(cost, start_id):
    """(cost, start id) -> cost << 24 | start id through the C ABI; raises ValueError for costs the all-reduce cannot carry
    (not a non-negative integer below 2^39: --fcost, GEO)."""
    p = C.c_int64(0)
    if lib().tsp_dev_multistart_pack(float(cost), int(start_id), C.byref(p)) != OK:
        raise ValueError("cost %r / start %r cannot be packed for the all-reduce(min)" % (cost, start_id))
    return p.value


def comm_available():
    """True if librccl can be opened by the C ABI (no communicator is formed)."""
    return bool(lib().tsp_dev_comm_available())


def comm_unique_id():
    """ncclGetUniqueId through the C ABI (rank 0): 128 bytes to hand to every rank."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib().tsp_dev_comm_unique_id(buf))
    return buf.raw


class Comm:
    """One rank of the RCCL communicator the C ABI's multi-start epilogue runs on (tsp_dev_comm_init_rank)."""

    def __init__(self, ctx, world, rank, unique_id):
        assert len(unique_id) == COMM_ID_BYTES
        self.ctx, self.world, self.rank = ctx, world, rank
        self._h = C.c_void_p()
        _check(lib().tsp_dev_comm_init_rank(ctx._h, world, rank, unique_id, C.byref(self._h)))

    def rccl_version(self):
        v = C.c_int(0)
        _check(lib().tsp_dev_comm_info(self._h, None, None, C.byref(v)))
        return v.value

    def allreduce_min(self, packed):
        out = C.c_int64(0)
        _check(lib().tsp_dev_multistart_allreduce(self._h, int(packed), C.byref(out)))
        return out.value

    def allreduce_min_f64(self, cost):
        """all-reduce(min) of one double per rank (costs the packed word cannot carry: --fcost)."""
        out = C.c_double(0)
        _check(lib().tsp_dev_multistart_allreduce_f64(self._h, float(cost), C.byref(out)))
        return out.value

    def bcast_tour(self, root, succ):
        """succ: int32 [n], the winner's tour on `root`, overwritten with it on the other ranks."""
        assert succ.dtype == np.int32 and succ.flags["C_CONTIGUOUS"]
        _check(lib().tsp_dev_multistart_bcast_tour(self._h, root, _i(succ), 1, len(succ)))
        return succ

    def close(self):
        if self._h:
            lib().tsp_dev_comm_destroy(self._h)
            self._h = C.c_void_p()
