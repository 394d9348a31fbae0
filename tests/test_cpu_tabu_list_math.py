"""CPU check of the closed forms behind two_opt_tabu_list.hpp (no GPU): what the reference's scan of a tabu list does to the
list (lazy clears, tabusearch.c:83-92) and how many pairs it skips (:137-149), computed from the non-zero stamps alone, against
the literal scan.  The device code is checked against the oracle on the GPU (test_gpu_tabu_list.py); this test pins the
derivation itself, in plain Python, on small random tours with dense random lists."""
import numpy as np
import pytest


def upos(i, j, n):
    i, j = min(i, j), max(i, j)
    return i * n + j - (i + 1) * (i + 2) // 2


def literal_scan(n, succ, stamps, it, tenure):
    """One sweep of tabusearch.c:128-157 without the delta: -> (pairs that reach :150, stamps after the lazy clears)."""
    st = stamps.copy()

    def check(e):                       # check_tenure, :83-92
        if it < 0 or tenure < 0 or st[e] == 0:
            return False
        if it - st[e] > tenure:
            st[e] = 0
            return False
        return True

    evals = 0
    for a in range(n - 1):
        for b in range(a + 1, n):
            a1, b1 = succ[a], succ[b]
            if b == a1 or b1 == a:
                continue
            if check(upos(a, b, n)) or check(upos(a, a1, n)) or check(upos(b, b1, n)) or check(upos(a, b1, n)):
                continue
            evals += 1
    return evals, st


def closed_form(n, succ, stamps, it, tenure):
    """The same two results from the list of non-zero stamps, O(1) per entry (plus early-exit walks for expired stamps on
    tour edges): two_opt_tabu_list.hpp, tabu_side()."""
    st = stamps.copy()
    pred = np.empty(n, dtype=np.int64)
    pred[succ] = np.arange(n)
    nonadj = sum(1 for a in range(n - 1) for b in range(a + 1, n) if b != succ[a] and succ[b] != a)

    def live_v(s):
        return s != 0 and not (it - s > tenure)

    def live(x, y):
        return live_v(stamps[upos(x, y, n)])      # the ORIGINAL stamps: only expired ones are ever cleared

    skipped, nf, f_succ_in_f = 0, 0, 0
    entries = [(u, v) for u in range(n - 1) for v in range(u + 1, n) if stamps[upos(u, v, n)] != 0]
    for u, v in entries:
        s = stamps[upos(u, v, n)]
        su, sv = succ[u], succ[v]
        uv, vu = su == v, sv == u
        if not live_v(s):
            if not uv and not vu:
                st[upos(u, v, n)] = 0
            else:
                x, y = (u, v) if uv else (v, u)
                px = pred[x]
                looked = any(b != y and b != px and not live(x, b) for b in range(x + 1, n)) or \
                    any(a != px and a != y and not live(a, x) and not live(a, succ[a]) for a in range(x)) or \
                    (y < px and px != succ[y] and not live(y, px) and not live(y, succ[y]) and not live(px, x))
                if looked:
                    st[upos(u, v, n)] = 0
            continue
        if not uv and not vu:
            fu, fv = int(live(u, su)), int(live(v, sv))
            skipped += 1 - fu - fv + fu * fv
        else:
            nf += 1
            skipped += n - 3
            y = v if uv else u
            if live(y, succ[y]):
                skipped += 1
        for a, w in ((u, v), (v, u)):
            b, sa = pred[w], succ[a]
            if b != a and a < b and b != sa and not live(a, b) and not live(a, sa) and not live(b, w):
                skipped += 1
    skipped -= nf * (nf - 1) // 2
    return nonadj - skipped, st


@pytest.mark.parametrize("seed", range(40))
def test_closed_form_equals_the_literal_scan(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 40))
    perm = rng.permutation(n)
    succ = np.empty(n, dtype=np.int64)
    succ[perm] = np.roll(perm, -1)
    it, tenure = int(rng.integers(1, 30)), int(rng.integers(0, 12))
    stamps = np.zeros(n * (n - 1) // 2, dtype=np.int64)
    for _ in range(int(rng.choice([1, n, 4 * n]))):
        a, b = int(rng.integers(0, n)), int(rng.integers(0, n))
        if rng.random() < 0.5:
            b = int(succ[a])                      # plenty of stamps on tour edges: the only place the forms are subtle
        if a != b:
            stamps[upos(a, b, n)] = int(rng.integers(1, it + 1))
    e1, s1 = literal_scan(n, succ, stamps, it, tenure)
    e2, s2 = closed_form(n, succ, stamps, it, tenure)
    assert e1 == e2
    assert (s1 == s2).all()
    # and the C oracle's alg_2opt_tabu (the checker of the GPU tests) does the same in its first sweep: a third, independent
    # statement of tabusearch.c:83-92,137-149 (the delta does not enter either number)
    from oracle import oracle as O
    xy = rng.integers(0, 1000, size=(n, 2)).astype(np.float64)
    st32 = stamps.astype(np.int32)
    _, _, _, ost, _, _ = O.two_opt_best(xy, O.EUC_2D, succ.astype(np.int32), tabu=st32, iter_=it, tenure=tenure, max_sweeps=1)
    assert ost["sweeps"] == 1 and ost["evals"] == e1
    assert (st32 == s1).all()
