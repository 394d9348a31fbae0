"""Diagnostic build only (lib_diag, -DTSP_STAMPS; recipe in tools/diag_stamps.py): the timeline of one exhaustive sweep
(k_exh, two_opt_exh.hpp) on rand10000 -- when the first / last wave starts, leaves its rows, when the last candidate is
published and when the last block's apply is done, in microseconds after the first wave's start.
usage: diag_exh.py [rj:waves]   (through gpurun)"""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
os.environ["TSP_NO_FILTER"] = "1"
from tsp_optimization_amd import build as B
B.LIB_DIR = os.path.join(R, 'tsp_optimization_amd', 'lib_diag')
from tsp_optimization_amd import engine as E
from helpers import load_instance
case = (sys.argv[1:] or ["4:4"])[0]
os.environ["TSP_EXH_RJ"], os.environ["TSP_EXH_WAVES"] = case.split(":")
ctx = E.Context(0)
xy, wt = load_instance('rand10000')
inst = E.Instance(ctx, xy, wt, 1)
succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
L = E.lib()
L.tsp_dev_debug_exh_stamps.argtypes = [C.POINTER(C.c_double)]
buf = (C.c_double * 16)()
t = E.Tours(inst, 1)
t.upload(succ[0], obj[0])
print(t.describe(E.BEST))
t.run(E.BEST, max_steps=5)
L.tsp_dev_debug_exh_stamps(buf)
acc = np.zeros(16)
N = 20
for _ in range(N):
    t.run(E.BEST, max_steps=1)
    L.tsp_dev_debug_exh_stamps(buf)
    acc += np.array(list(buf))
acc /= N
for name, v in zip(["last wave starts", "first wave out of its rows", "mean wave out of its rows", "last wave out of its rows", "last candidate published",
                    "apply done", "shader clock in the rows (MHz)", "waves", "bookkeeping branches per wave", "cycles in them per wave",
                    "cycles in the rows per wave"], acc):
    print("  %-32s %8.2f" % (name, v))

# where the waves ran: HW_ID bits (gfx9): wave_id 3:0, simd_id 5:4, pipe 7:6, cu_id 11:8, sh_id 12, se_id 15:13
dump = os.environ.get("TSP_EXH_DUMP")
if dump and os.path.exists(dump):
    from collections import defaultdict
    per = defaultdict(list)
    for ln in open(dump):
        k, t0, t1, xcc, hw = ln.split()
        hw = int(hw)
        per[(int(xcc), (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)].append(float(t1))
    counts = defaultdict(int)
    for key, v in per.items():
        counts[len(v)] += 1
    print("  SIMDs by number of waves they hosted:", dict(sorted(counts.items())), "(%d SIMDs seen)" % len(per))
    by = defaultdict(list)
    for key, v in per.items():
        by[len(v)].append(max(v))
    for c in sorted(by):
        print("    %d waves on a SIMD: last of them out of its rows at %.1f us (mean over %d SIMDs)" % (c, sum(by[c]) / len(by[c]), len(by[c])))
    cus = defaultdict(int)
    for key, v in per.items():
        cus[key[:4]] += len(v)
    cc = defaultdict(int)
    for v in cus.values():
        cc[v] += 1
    print("  CUs by number of waves:", dict(sorted(cc.items())), "(%d CUs seen)" % len(cus))
    # does a workgroup's age on its CU follow its index?  mean time out of the rows per quarter of the grid, and per wave of the workgroup
    rows = [ln.split() for ln in open(dump)]
    nw = len(rows)
    for part in range(4):
        sel = [float(r[2]) for r in rows if int(r[0]) * 4 // nw == part]
        print("    workgroups %d/4 of the grid: mean out of rows at %.1f us (min %.1f, max %.1f)" % (part + 1, sum(sel) / len(sel), min(sel), max(sel)))
