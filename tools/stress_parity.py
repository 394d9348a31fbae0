"""Randomised parity sweep: random sizes around the tile / group / batch boundaries, random metrics and cost modes, random
tours; both rules, forced sorted sweep / sorted scan, both construction kernels, GRID, LDS and CLUSTER engines with a random
cluster size, a dense random tabu list per small case -- all against the oracle.  `run(seed, cases)` returns the number of
mismatching cases (it stops at the first); tests/test_gpu_stress.py runs a seeded slice of it under `-m gpu`, and as a
script (through gpurun) SEED=<n> CASES=<n> select a longer run."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if R not in sys.path: sys.path.insert(0, R)
if os.path.join(R, 'tests') not in sys.path: sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
SIZES = [3, 4, 5, 6, 7, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 640]   # the CPU oracle bounds the run time


def run(seed, cases, ctx=None, verbose=True, max_n=3000):
    prev = os.environ.get("TSP_SORTED_MIN_N")
    os.environ["TSP_SORTED_MIN_N"] = "0"
    try:
        return _run(seed, cases, ctx, verbose, max_n)
    finally:
        if prev is None: os.environ.pop("TSP_SORTED_MIN_N", None)
        else: os.environ["TSP_SORTED_MIN_N"] = prev
        for k in ("TSP_CLUSTER_BLOCKS", "TSP_LDS_PROBE", "TSP_CLUSTER_FIRST_SORTED", "TSP_CLUSTER_FS_ROWS", "TSP_CLUSTER_COPIES", "TSP_CLUSTER_XCD_LOCAL"): os.environ.pop(k, None)


FAILED_AT = []


def _chk(ok, line, cond):
    if ok and not cond:
        FAILED_AT.append(line)      # the first check of the case that failed (a line of this file)
    return bool(ok and cond)


def _run(seed, cases, ctx, verbose, max_n):
    from tsp_optimization_amd import engine as E
    from helpers import random_tour
    from oracle import oracle as O
    own = ctx is None
    if own: ctx = E.Context(0)
    rng = np.random.default_rng(seed)
    sizes = SIZES
    bad = 0
    for c in range(cases):
        t_case = time.perf_counter()
        n = int(rng.choice(sizes)) if rng.random() < 0.6 else int(rng.integers(3, 600))
        if rng.random() < 0.06 and max_n > 1000: n = int(rng.integers(1000, max_n))   # a few larger ones (first improvement only on the CPU side)
        wt = int(rng.choice([O.EUC_2D, O.ATT, O.CEIL_2D, O.MAN_2D, O.MAX_2D]))
        ic = int(rng.random() < 0.75)
        int_coords = rng.random() < 0.5
        if int_coords: xy = rng.integers(0, int(rng.choice([20, 1000, 1000000])), size=(n, 2)).astype(np.float64)
        else: xy = rng.uniform(-5000, 5000, size=(n, 2))
        # float costs with exactly tied distances (coincident or lattice points) can make the reference's best-improvement
        # loop cycle forever on rounding-noise deltas (seen: ATT, 127 points on a 20 x 20 lattice): integer costs there
        if int_coords: ic = 1
        if wt == O.CEIL_2D: ic = 1
        # float costs on MAN_2D / MAX_2D (dy = |y2 - y2| = 0 in the reference) can cycle forever on rounding noise: the
        # reference relies on its time limit there, and so would this run
        if wt in (O.MAN_2D, O.MAX_2D): ic = 1
        inst = E.Instance(ctx, xy, wt, ic)
        s0 = int(rng.integers(0, n))
        succ, obj, _ = inst.construct(E.GREEDY, np.array([s0], dtype=np.int32))
        _, es, eo = O.greedy(xy, wt, start=s0, integer_cost=ic)
        ok = (succ[0] == es).all() and obj[0] == eo
        tour = random_tour(n, rng) if rng.random() < 0.5 else es
        cost = O.succ_cost(xy, wt, tour, integer_cost=ic)
        if n <= 300:   # the oracle's best-improvement descent from a random tour is O(n^3)
            rc, s, o, st = inst.two_opt(tour, cost, mode=E.BEST, engine=1)
            _, bs, bo, bst, _, _ = O.two_opt_best(xy, wt, tour, integer_cost=ic)
            ok = _chk(ok, 60, (s == bs).all() and o == bo and (st["sweeps"], st["evals"], st["moves"]) == (bst["sweeps"], bst["evals"], bst["moves"]))
        rc, s, o, st = inst.two_opt(tour, cost, mode=E.FIRST, engine=1)
        _, fs, fo, fst, _ = O.two_opt_first(xy, wt, tour, cost, integer_cost=ic)
        ok = _chk(ok, 63, (s == fs).all() and o == fo and (st["sweeps"], st["evals"], st["moves"], st["reversed"]) == (fst["sweeps"], fst["evals"], fst["moves"], fst["reversed"]))
        # the same two descents on the CLUSTER engine with a random number of workgroups per tour
        os.environ["TSP_CLUSTER_BLOCKS"] = str(int(rng.choice([1, 2, 3, 5, 8, 17, 64, 200, 256])))
        # first improvement: the plain replica, or the one in rank order with the box-pruned step never / always / now and then
        fsr = int(rng.choice([-1, 0, 1, 3, 60]))
        os.environ["TSP_CLUSTER_FIRST_SORTED"] = "0" if fsr < 0 else "8"
        os.environ["TSP_CLUSTER_FS_ROWS"] = str(max(fsr, 0))
        os.environ["TSP_CLUSTER_COPIES"] = str(int(rng.choice([0, 1, 2, 3, 8])))   # copies of the exchange area
        inst.reload_switches()
        if n <= 300:
            rc, s, o, st = inst.two_opt(tour, cost, mode=E.BEST, engine=3)
            ok = _chk(ok, 73, (s == bs).all() and o == bo and (st["sweeps"], st["evals"], st["moves"]) == (bst["sweeps"], bst["evals"], bst["moves"]))
        rc, s, o, st = inst.two_opt(tour, cost, mode=E.FIRST, engine=3)
        ok = _chk(ok, 75, (s == fs).all() and o == fo and (st["sweeps"], st["evals"], st["moves"], st["reversed"]) == (fst["sweeps"], fst["evals"], fst["moves"], fst["reversed"]))
        del os.environ["TSP_CLUSTER_BLOCKS"]
        del os.environ["TSP_CLUSTER_COPIES"]
        inst.reload_switches()
        if n <= 8000:   # the LDS engine (one workgroup per tour), first improvement, with and without the probe
            os.environ["TSP_LDS_PROBE"] = str(int(rng.choice([0, 1, 600, 1 << 30])))
            inst.reload_switches()
            rc, s, o, st = inst.two_opt(tour, cost, mode=E.FIRST, engine=2)
            del os.environ["TSP_LDS_PROBE"]
            inst.reload_switches()
            ok = _chk(ok, 84, (s == fs).all() and o == fo and (st["sweeps"], st["evals"], st["moves"], st["reversed"]) == (fst["sweeps"], fst["evals"], fst["moves"], fst["reversed"]))
        if 4 <= n <= 220:
            # alg_2opt_tabu with a dense random tabu list (live and expired stamps, tour edges included): the list path
            it, ten = int(rng.integers(2, 40)), int(rng.integers(0, 15))
            stamps = np.zeros(n * (n - 1) // 2, dtype=np.int32)
            for _ in range(int(rng.choice([1, n // 2, 2 * n]))):
                a, b = int(rng.integers(0, n)), int(rng.integers(0, n))
                if rng.random() < 0.4: b = int(tour[a])
                if a != b: stamps[min(a, b) * n + max(a, b) - (min(a, b) + 1) * (min(a, b) + 2) // 2] = int(rng.integers(1, it + 1))
            exp = stamps.copy()
            _, ts, to, tst, _, _ = O.two_opt_best(xy, wt, tour, integer_cost=ic, tabu=exp, iter_=it, tenure=ten)
            tb = E.Tabu(inst)
            tb.upload(stamps)
            rc, s, o, st, _ = tb.two_opt(tour, it, ten)
            ok = _chk(ok, 98, (s == ts).all() and o == to and (st["sweeps"], st["evals"], st["moves"]) == (tst["sweeps"], tst["evals"], tst["moves"]))
            ok = _chk(ok, 99, (tb.download() == exp).all() and tb.list_info()[1])
            tb.close()
        if 8 <= n <= 300 and wt in (O.EUC_2D, O.ATT, O.CEIL_2D):
            # a chain of tabu() iterations INSIDE one launch (tsp_dev_tours_tabu_iterations_ex) from a dense random tabu list: random
            # tenures (one per iteration), random kick pairs with rejected ones among them (a == b, neighbours, tabu edges), against
            # the oracle's alg_2opt_tabu + a restatement of the kick's trials (tabusearch.c:262-309) iteration by iteration
            it0, K = int(rng.integers(1, 30)), int(rng.integers(2, 9))
            tens = [int(rng.integers(0, 12)) for _ in range(K)]
            P = K + int(rng.integers(0, 6))
            ab = rng.integers(0, n, size=(P, 2)).astype(np.int32)
            for q in range(P):
                r = rng.random()
                if r < 0.15: ab[q, 1] = ab[q, 0]
            stamps = np.zeros(n * (n - 1) // 2, dtype=np.int32)
            up = lambda a, b: min(a, b) * n + max(a, b) - (min(a, b) + 1) * (min(a, b) + 2) // 2
            for _ in range(int(rng.choice([0, n // 2, 2 * n]))):
                a, b = int(rng.integers(0, n)), int(rng.integers(0, n))
                if rng.random() < 0.4: b = int(tour[a])
                if a != b: stamps[up(a, b)] = int(rng.integers(1, it0 + 1))
            exp = stamps.copy()
            cur, curo = np.array(tour, dtype=np.int32), float(cost)
            best, objs, trials, p, stopped = float("inf"), [], [], 0, False

            def is_tabu(a, b, itn, ten):      # check_tenure, :83-92
                v = exp[up(a, b)]
                if v == 0: return False
                if itn - v > ten: exp[up(a, b)] = 0; return False
                return True
            for k in range(K):
                itn, ten = it0 + k, tens[k]
                _, cur, curo, _, _, _ = O.two_opt_best(xy, wt, cur, integer_cost=ic, tabu=exp, iter_=itn, tenure=ten)
                cur = np.array(cur, dtype=np.int32)
                best = min(best, curo)
                objs.append(curo)
                used, acc = 0, False
                while p < P and not acc:
                    a, b = int(ab[p, 0]), int(ab[p, 1]); p += 1; used += 1
                    a1, b1 = int(cur[a]), int(cur[b])
                    if a == b or a1 == b or b1 == a: continue
                    if not is_tabu(a, a1, itn, ten) and not is_tabu(b, b1, itn, ten) and not is_tabu(a, b, itn, ten) and not is_tabu(a1, b1, itn, ten):
                        acc = True
                        path, v = [], a1                       # a1 ... b along the tour, reversed (utility.c:708-717)
                        while True:
                            path.append(v)
                            if v == b: break
                            v = int(cur[v])
                        cur[a] = b
                        for q in range(len(path) - 1, 0, -1): cur[path[q]] = path[q - 1]
                        cur[a1] = b1
                        exp[up(a, a1)] = itn; exp[up(b, b1)] = itn
                trials.append(used)
                if not acc: stopped = True; break
            t, tb = E.Tours(inst, 1), E.Tabu(inst)
            t.upload(np.array(tour, dtype=np.int32), float(cost))
            tb.upload(stamps)
            rc, done, last_acc, b2, o2, _, tr2 = t.tabu_iterations_ex(tb, it0, tens, ab, float("inf"))
            if done > 0:                                        # (0: the chain does not apply to this case -- nothing to compare)
                ok = _chk(ok, 131, rc == 0 and done == len(objs) and list(o2) == objs and list(tr2) == trials and b2 == best and bool(last_acc) == (not stopped))
                sd, od, _ = t.download()
                ok = _chk(ok, 133, (sd[0] == cur).all())
                got = tb.download()
                if verbose and not (got == exp).all():
                    w = np.nonzero(got != exp)[0]
                    print("  stamps differ at %d entries; first: index %d device %d oracle %d (initial %d); chain from iteration %d, tenures %s, trials %s"
                          % (len(w), w[0], got[w[0]], exp[w[0]], stamps[w[0]], it0, tens, trials))
                ok = _chk(ok, 134, (got == exp).all())
            t.close(); tb.close()
        if c % 3 == 0 and n >= 8:
            # a batch of three or eight tours through the engine the library picks, and one GRASP tour on the oracle's URAND stream
            nt = 8 if n <= 200 else 3   # eight tours: the grid divides by XCD (a tour's workgroups from one XCD)
            t3 = np.stack([random_tour(n, rng) for _ in range(nt)])
            c3 = np.array([O.succ_cost(xy, wt, t, integer_cost=ic) for t in t3])
            rc, s3, o3, st3 = inst.two_opt(t3, c3, mode=E.FIRST)
            os.environ["TSP_CLUSTER_BLOCKS"] = str(int(rng.choice([1, 2, 7, 33, 85] if nt == 3 else [1, 2, 7, 16, 32])))   # whole clusters must be resident
            os.environ["TSP_CLUSTER_XCD_LOCAL"] = str(int(rng.integers(0, 2)))
            inst.reload_switches()
            rc, s4, o4, st4 = inst.two_opt(t3, c3, mode=E.FIRST, engine=3)      # clusters side by side
            del os.environ["TSP_CLUSTER_BLOCKS"]
            del os.environ["TSP_CLUSTER_XCD_LOCAL"]
            inst.reload_switches()
            for b in range(nt):
                _, fs3, fo3, fst3, _ = O.two_opt_first(xy, wt, t3[b], c3[b], integer_cost=ic)
                ok = _chk(ok, 113, (s3[b] == fs3).all() and o3[b] == fo3 and st3[b]["evals"] == fst3["evals"])
                ok = _chk(ok, 114, (s4[b] == fs3).all() and o4[b] == fo3 and st4[b]["evals"] == fst3["evals"])
            O.srandom(1000 + c)
            ur = np.array([[O.urand() for _ in range(n)]])
            O.srandom(1000 + c)
            _, gs, go = O.grasp(xy, wt, start=s0, integer_cost=ic)
            sg, og, _ = inst.construct(E.GRASP, np.array([s0], dtype=np.int32), ur)
            ok = _chk(ok, 120, (sg[0] == gs).all() and og[0] == go)
        inst.close()
        if verbose: print("case %d n %d wt %d ic %d %s  %.2f s" % (c, n, wt, ic, "ok" if ok else "MISMATCH", time.perf_counter() - t_case), flush=True)
        if not ok:
            bad += 1
            print("MISMATCH case %d: n %d wt %d ic %d, first failing check at stress_parity.py:%s; switches %s" % (c, n, wt, ic, FAILED_AT[-1:], {k: v for k, v in os.environ.items() if k.startswith("TSP_")}))
            break
    if own: ctx.close()
    return bad


if __name__ == "__main__":
    n_cases = int(os.environ.get("CASES", "60"))
    n_bad = run(int(os.environ.get("SEED", "1")), n_cases)
    print("cases %d, mismatches %d" % (n_cases, n_bad))
    sys.exit(1 if n_bad else 0)
