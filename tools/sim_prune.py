"""Design study (not product, not a test): how many 64 x 64 group pairs of a best-improvement sweep survive a
bounding-box form of the new-edge bound, at several stages of the rand10000 descent.  Groups = 64 consecutive
nodes along a space-filling curve (Hilbert / Morton) or 64 consecutive tour positions.  Runs on the GPU box
(the descent itself is done by the engine); the analysis is numpy.
"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from tsp_optimization_amd import engine as E
from helpers import load_instance

G = int(os.environ.get("G", "64"))


def hilbert_key(x, y, bits=16):
    x = x.astype(np.int64).copy(); y = y.astype(np.int64).copy()
    d = np.zeros_like(x)
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64); ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        # rotate
        m = ry == 0
        fl = m & (rx == 1)
        x = np.where(fl, s - 1 - x, x); y = np.where(fl, s - 1 - y, y)
        x2 = np.where(m, y, x); y2 = np.where(m, x, y)
        x, y = x2, y2
        s >>= 1
    return d


def morton_key(x, y, bits=16):
    def spread(v):
        v = v.astype(np.uint64)
        v = (v | (v << 16)) & np.uint64(0x0000FFFF0000FFFF)
        v = (v | (v << 8)) & np.uint64(0x00FF00FF00FF00FF)
        v = (v | (v << 4)) & np.uint64(0x0F0F0F0F0F0F0F0F)
        v = (v | (v << 2)) & np.uint64(0x3333333333333333)
        v = (v | (v << 1)) & np.uint64(0x5555555555555555)
        return v
    return spread(x) | (spread(y) << np.uint64(1))


def analyse(xy, succ, perm, label, bd):
    n = len(xy)
    d = np.rint(np.sqrt(((xy - xy[succ]) ** 2).sum(1)))
    ng = (n + G - 1) // G
    X = xy[perm]; D = d[perm]
    pad = ng * G - n
    if pad:
        X = np.vstack([X, np.repeat(X[-1:], pad, 0)]); D = np.concatenate([D, np.zeros(pad)])
    X = X.reshape(ng, G, 2); D = D.reshape(ng, G)
    lo = X.min(1); hi = X.max(1); gm = D.max(1)
    gx = np.maximum(0, np.maximum(lo[:, None, 0] - hi[None, :, 0], lo[None, :, 0] - hi[:, None, 0]))
    gy = np.maximum(0, np.maximum(lo[:, None, 1] - hi[None, :, 1], lo[None, :, 1] - hi[:, None, 1]))
    mind2 = gx * gx + gy * gy
    iu = np.triu_indices(ng)
    out = []
    for b in (0.0, bd):
        T = b + gm[:, None] + gm[None, :] + 1.0
        surv = (T > 0) & (mind2 < T * T)
        out.append(surv[iu].mean())
    if label == "hilbert" and G == 64:
        # row culling inside surviving pairs: a row can matter only if it reaches the other group's box
        surv = (mind2 < (gm[:, None] + gm[None, :] + 1.0) ** 2)
        rr, cc = np.nonzero(np.triu(surv))
        tot_a = tot_best = tot_rule = 0
        for r_, c_ in zip(rr, cc):
            def rows_alive(a_, b_):
                P = X[a_]; dsa = D[a_]
                gx_ = np.maximum(0, np.maximum(lo[b_, 0] - P[:, 0], P[:, 0] - hi[b_, 0]))
                gy_ = np.maximum(0, np.maximum(lo[b_, 1] - P[:, 1], P[:, 1] - hi[b_, 1]))
                return int((gx_ * gx_ + gy_ * gy_ < (dsa + gm[b_] + 1.0) ** 2).sum())
            ra, rb = rows_alive(r_, c_), rows_alive(c_, r_)
            tot_a += ra; tot_best += min(ra, rb); tot_rule += (ra if gm[r_] >= gm[c_] else rb)
        print("    surviving pairs %d: rows alive per pair: rows=r %.1f, best orientation %.1f, rows=larger-gmax group %.1f (of %d)"
              % (len(rr), tot_a / len(rr), tot_best / len(rr), tot_rule / len(rr), G))
    print("  %-10s groups %d  survive(bound 0) %.3f  survive(bound = sweep's best %.0f) %.3f   gmax median %.0f p90 %.0f"
          % (label, ng, out[0], bd, out[1], np.median(gm), np.percentile(gm, 90)))


def best_delta(xy, succ):
    n = len(xy)
    d = np.rint(np.sqrt(((xy - xy[succ]) ** 2).sum(1)))
    best = 0.0
    xs = xy[succ]
    for r0 in range(0, n, 500):
        a = xy[r0:r0 + 500]; a1 = xs[r0:r0 + 500]
        dab = np.rint(np.sqrt(((a[:, None, :] - xy[None, :, :]) ** 2).sum(2)))
        da1b1 = np.rint(np.sqrt(((a1[:, None, :] - xs[None, :, :]) ** 2).sum(2)))
        delta = dab + da1b1 - d[r0:r0 + 500, None] - d[None, :]
        idx = np.arange(r0, min(r0 + 500, n))
        delta[idx - r0, idx] = 0
        best = min(best, delta.min())
    return best


ctx = E.Context(0)
xy, wt = load_instance('rand10000')
inst = E.Instance(ctx, xy, wt, 1)
succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
n = len(xy)
q = ((xy - xy.min(0)) / (xy.max(0) - xy.min(0)) * 65535).astype(np.int64)
perm_h = np.argsort(hilbert_key(q[:, 0], q[:, 1]), kind='stable')
perm_m = np.argsort(morton_key(q[:, 0], q[:, 1]), kind='stable')
tours = E.Tours(inst, 1)
tours.upload(succ[0], obj[0])
done = 0
for stage in (0, 50, 200, 420, 800, 1200, 1420):
    if stage > done:
        tours.run(E.BEST, max_steps=stage - done)
        done = stage
    s, o, st = tours.download()
    s = np.asarray(s).reshape(-1)[:n]
    bd = best_delta(xy, s)
    print("stage %d (sweeps done %d): best delta of the next sweep %.0f" % (stage, done, bd))
    analyse(xy, s, perm_h, "hilbert", bd)
    analyse(xy, s, perm_m, "morton", bd)
    # tour-position groups
    order = np.empty(n, dtype=np.int64); v = 0
    for p in range(n):
        order[p] = v; v = s[v]
    analyse(xy, s, order, "tour-pos", bd)
