"""Diagnostic build only (lib_diag, -DTSP_STAMPS): where a FIRST/BEST step's last block spends its time,
and the shader clock during the row loops.  Build it first (never shipped, never timed as the product):

  make -C tsp_optimization_amd/csrc OUT=../lib_diag -j8 \
       HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function -DTSP_STAMPS"
"""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import build as B
B.LIB_DIR = os.path.join(R, 'tsp_optimization_amd', 'lib_diag')
from tsp_optimization_amd import engine as E
from helpers import load_instance
names = ["start->prologue done", "row loop", "block argmin", "publish+drain", "ticket atomic+barrier",
         "read partials+argmin", "pos + adj count + barrier", "swaps / (k_first: control block)", "cost/next-active/state"]
ctx = E.Context(0)
xy, wt = load_instance('rand10000')
inst = E.Instance(ctx, xy, wt, 1)
succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
L = E.lib()
L.tsp_dev_debug_stamps.argtypes = [C.POINTER(C.c_double)]
buf = (C.c_double * 16)()
L.tsp_dev_debug_stamps(buf)
for mode, nm in [(E.FIRST, 'FIRST'), (E.BEST, 'BEST')]:
    rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=mode)
    n = L.tsp_dev_debug_stamps(buf)
    print(nm, 'steps', st['steps'], 'device_ms', st['device_ms'], 'us/step', 1e3 * st['device_ms'] / st['steps'], 'stamped', n)
    tot = 0
    for k in range(1, 10):
        print('  %-28s %7.2f us' % (names[k - 1], buf[k] / 100.0)); tot += buf[k] / 100.0
    print('  %-28s %7.2f us' % ('sum (last block lifetime)', tot))
    print('  shader clock during row loops: %.0f MHz; row-loop block-us total %.0f (=> per step %.1f block-us)' % (buf[15], buf[14], buf[14] / max(n, 1) if False else buf[14] / max(st['steps'], 1)))

