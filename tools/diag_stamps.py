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
         "read partials+argmin", "pos + adj count + barrier", "swaps", "cost/next-active/state"]
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

    if nm == 'BEST' and hasattr(L, 'tsp_dev_debug_sweep'):
        c = (C.c_ulonglong * 8192)()
        L.tsp_dev_debug_sweep(c)
        a = np.array(c[:], dtype=np.float64).reshape(1024, 8)
        a = a[a[:, 0] > 0]
        m = lambda k: ((a[:, k] / a[:, 0]).mean() / 100, (a[:, k] / a[:, 0]).max() / 100)
        print('  k_sweep blocks %d: kept/block mean %.2f' % (len(a), (a[:, 1] / a[:, 0]).mean()))
        for k, nm2 in ((2, 'start->tests done'), (3, 'tests->staged+barrier'), (4, 'pair loop (wave 0)'), (5, 'barrier after loop')):
            print('    %-24s mean %.2f us, slowest block %.2f us' % ((nm2,) + m(k)))
        print('    last launch: block start skew %.2f us, span first start -> last loop end %.2f us'
              % ((a[:, 6].max() - a[:, 6].min()) / 100, (a[:, 7].max() - a[:, 6].min()) / 100))
        c2 = (C.c_ulonglong * 4096)()
        L.tsp_dev_debug_sweep2(c2)
        f2 = np.array(c2[:], dtype=np.float64).reshape(1024, 4)
        full = np.array(c[:], dtype=np.float64).reshape(1024, 8)
        idx = np.argsort(-(full[:, 4] / np.maximum(full[:, 0], 1)))[:12]
        for b in idx:
            print('      block %4d (cluster %3d member %d): kept %.2f  loop us %.2f  barrier us %.2f | wave 0 per launch: quads %.1f hot cycles %.0f tier-1+ cycles %.0f diag units %.2f' % (b, b // 8, b % 8, full[b, 1] / full[b, 0], full[b, 4] / full[b, 0] / 100, full[b, 5] / full[b, 0] / 100, f2[b, 0] / full[b, 0], f2[b, 1] / full[b, 0], f2[b, 2] / full[b, 0], f2[b, 3] / full[b, 0]))
        print('      all blocks, wave 0 per launch: quads %.2f hot cycles %.0f tier-1+ cycles %.0f diag units %.3f' % tuple(f2[:512].sum(0) / full[:512, 0].sum()))
        kk = full[:512, 1] / np.maximum(full[:512, 0], 1)
        print('      kept per block: min %.2f max %.2f; per cluster sums min %.1f max %.1f' % (kk.min(), kk.max(), kk.reshape(64, 8).sum(1).min(), kk.reshape(64, 8).sum(1).max()))
