"""tsp_optimization_amd -- MI355X-native 2-opt local-search engine behind the entry points of
deno750/TSP_Optimization's heuristics path.

The product is the C-ABI library (include/tsp_hip.h -> lib/libtsp_hip.so, hand-written HIP for
gfx950) plus the C host mirror of the reference's solver.h / heuristics.h functions
(host/ -> lib/libtsp_host.so and the `tsp` CLI).  This Python package is a thin ctypes binding
over the C ABI for tests and bench.py; there is no CPU fallback: importing `engine` without the
built library, or calling it without a GPU, raises.
"""
from .build import build_all, lib_path  # noqa: F401
