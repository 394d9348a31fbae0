"""GPU parity of the sorted best-improvement sweep (k_move_recs + k_sweep, two_opt_sweep.hpp).

By default the engine uses it for n >= 1000, where the CPU oracle soon needs minutes per descent; these tests set
TSP_SORTED_MIN_N=0 so that the same kernels run on the small instances the oracle finishes in seconds, and check
tours, costs and counters against alg_2opt_tabu's restatement (tabusearch.c:107-178).  At full size the sorted
and the tiled sweep are run against each other (they must agree move for move)."""
import numpy as np
import pytest

from oracle import oracle as O
from helpers import golden, load_instance, rand_instance, random_tour

pytestmark = pytest.mark.gpu

APB = golden("survey_appendix_b.json")


@pytest.fixture(scope="module")
def eng():
    from tsp_optimization_amd import engine as E
    assert E.device_count() >= 1, "no HIP device visible: the product path has no CPU fallback"
    return E


@pytest.fixture(scope="module")
def ctx(eng):
    c = eng.Context(0)
    yield c
    c.close()


@pytest.fixture()
def sorted_always(monkeypatch):
    monkeypatch.setenv("TSP_SORTED_MIN_N", "0")


def _check_best(eng, inst, xy, wt, succ0, obj0, integer_cost=1):
    rc, s, o, st = inst.two_opt(succ0, obj0, mode=eng.BEST, engine=1)   # 1 = GRID
    _, es, eo, est, _, _ = O.two_opt_best(xy, wt, succ0, integer_cost=integer_cost)
    assert rc == 0 and O.is_tour(s)
    assert (s == es).all(), "final tour differs from the oracle's"
    assert o == eo
    assert (st["sweeps"], st["evals"], st["moves"], st["reversed"]) == \
        (est["sweeps"], est["evals"], est["moves"], est["reversed"])
    return o, st


@pytest.mark.parametrize("name", ["berlin52", "pr299", "att532", "rand1000"])
def test_sorted_sweep_matches_survey_counters(eng, ctx, sorted_always, name):
    xy, wt = load_instance(name)
    inst = eng.Instance(ctx, xy, wt, 1)
    _, succ0, obj0 = O.greedy(xy, wt)
    o, st = _check_best(eng, inst, xy, wt, succ0, obj0)
    inst.close()
    e = APB[name]["best"]
    assert (o, st["sweeps"], st["evals"], st["moves"]) == (e["cost"], e["sw"], e["ev"], e["mv"])


@pytest.mark.parametrize("name,ic", [("d493", 0), ("d493", 1), ("dsj1000", 1), ("eil51", 1), ("kroA100", 0)])
def test_sorted_sweep_other_metrics_and_float_costs(eng, ctx, sorted_always, name, ic):
    """CEIL_2D (dsj1000), non-integer coordinates (d493), --fcost double costs: the order of the sums matters."""
    xy, wt = load_instance(name)
    inst = eng.Instance(ctx, xy, wt, ic)
    _, succ0, obj0 = O.greedy(xy, wt, integer_cost=ic)
    _check_best(eng, inst, xy, wt, succ0, obj0, integer_cost=ic)
    inst.close()


@pytest.mark.parametrize("n", [5, 63, 64, 65, 128, 129, 200, 449])
def test_sorted_sweep_random_tours_and_group_boundaries(eng, ctx, sorted_always, n):
    rng = np.random.default_rng(n)
    xy = rng.integers(0, 3000, size=(n, 2)).astype(np.float64)
    inst = eng.Instance(ctx, xy, O.EUC_2D, 1)
    succ0 = random_tour(n, rng)
    _check_best(eng, inst, xy, O.EUC_2D, succ0, O.succ_cost(xy, O.EUC_2D, succ0))
    inst.close()


def test_sorted_sweep_duplicate_points_and_tied_deltas(eng, ctx, sorted_always):
    rng = np.random.default_rng(11)
    xy = rng.integers(0, 12, size=(300, 2)).astype(np.float64)   # coincident nodes, many equal deltas
    for wt in (O.EUC_2D, O.ATT, O.CEIL_2D):
        inst = eng.Instance(ctx, xy, wt, 1)
        succ0 = random_tour(300, rng)
        _check_best(eng, inst, xy, wt, succ0, O.succ_cost(xy, wt, succ0))
        inst.close()


def test_sorted_sweep_batch_of_tours(eng, ctx, sorted_always):
    xy, wt = load_instance("pr299")
    inst = eng.Instance(ctx, xy, wt, 1)
    n = len(xy)
    rng = np.random.default_rng(3)
    succ0 = np.stack([random_tour(n, rng) for _ in range(4)])
    obj0 = np.array([O.succ_cost(xy, wt, s) for s in succ0])
    rc, s, o, st = inst.two_opt(succ0, obj0, mode=eng.BEST, engine=1)
    for b in range(4):
        _, es, eo, est, _, _ = O.two_opt_best(xy, wt, succ0[b])
        assert (s[b] == es).all() and o[b] == eo
        assert (st[b]["sweeps"], st[b]["evals"], st[b]["moves"]) == (est["sweeps"], est["evals"], est["moves"])
    inst.close()


@pytest.mark.parametrize("name,steps", [("rand10000", 300), ("rand5000", 400), ("att532x", 0)])
def test_sorted_and_tiled_sweeps_agree_move_for_move(eng, ctx, monkeypatch, name, steps):
    """Full size: the same descent through k_sweep and through the tiled k_step (every pair visited)."""
    if name == "att532x":
        xy, wt = load_instance("att532")
        steps = 99
    else:
        xy, wt = load_instance(name)
    out = []
    for min_n in ("0", "1000000000"):
        monkeypatch.setenv("TSP_SORTED_MIN_N", min_n)
        inst = eng.Instance(ctx, xy, wt, 1)
        succ, obj, _ = inst.construct(eng.GREEDY, np.array([0], dtype=np.int32))
        tours = eng.Tours(inst, 1)
        tours.upload(succ[0], obj[0])
        tours.run(eng.BEST, max_steps=steps)
        s, o, st = tours.download()
        out.append((s.copy(), st[0]))
        tours.close()
        inst.close()
    assert (out[0][0] == out[1][0]).all()
    keys = ("sweeps", "evals", "moves", "reversed", "pairs_scanned", "steps")
    assert [out[0][1][k] for k in keys] == [out[1][1][k] for k in keys]
    assert out[0][1]["moves"] >= min(steps, 90)


@pytest.mark.parametrize("name", ["att532", "rand2000", "d493"])
def test_first_improvement_both_kernel_forms_agree(eng, ctx, monkeypatch, name):
    """k_first (fixed grid, out-of-place moves, two control-block slots) against k_step<FIRST> (the first form),
    and both against the oracle's alg_2opt counters."""
    xy, wt = load_instance(name)
    _, succ0, obj0 = O.greedy(xy, wt)
    _, es, eo, est, _ = O.two_opt_first(xy, wt, succ0, obj0)
    for v1 in ("0", "1"):
        for gy, rj in (("8", "2"), ("3", "1"), ("64", "2")):
            monkeypatch.setenv("TSP_FIRST_V1", v1)
            monkeypatch.setenv("TSP_FIRST_GRID_ROWS", gy)
            monkeypatch.setenv("TSP_FIRST_RJ", rj)
            inst = eng.Instance(ctx, xy, wt, 1)
            rc, s, o, st = inst.two_opt(succ0, obj0, mode=eng.FIRST, engine=1)
            inst.close()
            assert (s == es).all() and o == eo, (v1, gy, rj)
            assert (st["sweeps"], st["evals"], st["moves"], st["reversed"]) == \
                (est["sweeps"], est["evals"], est["moves"], est["reversed"]), (v1, gy, rj)


def test_first_then_best_then_first_on_one_handle(eng, ctx):
    """Runs of both rules on the same resident tours: every run leaves the tour in the first copy of order/pos
    and the control block in a slot the next run finds."""
    xy, wt = load_instance("pr1002")
    inst = eng.Instance(ctx, xy, wt, 1)
    _, succ0, obj0 = O.greedy(xy, wt)
    tours = eng.Tours(inst, 1)
    tours.upload(succ0, obj0)
    tours.run(eng.FIRST, max_steps=37)
    tours.run(eng.BEST, max_steps=5)
    tours.run(eng.FIRST, max_steps=11)
    s, o, st = tours.download()
    assert O.is_tour(s[0])
    # replay on the CPU: 37 first-improvement steps cannot be cut out of the oracle's loop, so check invariants:
    # the running cost of a first-improvement run started from the true cost stays the true cost
    tours2 = eng.Tours(inst, 1)
    tours2.upload(s[0], O.succ_cost(xy, wt, s[0]))
    rc, done = tours2.run(eng.FIRST)
    s2, o2, _ = tours2.download()
    assert done and O.is_tour(s2[0]) and o2[0] == O.succ_cost(xy, wt, s2[0])
    _, es, eo, _, _ = O.two_opt_first(xy, wt, s[0], O.succ_cost(xy, wt, s[0]))
    assert (s2[0] == es).all() and o2[0] == eo
    tours.close(); tours2.close(); inst.close()


def test_sorted_sweep_several_tours_at_full_size(eng, ctx, monkeypatch):
    """B = 3 random tours of a 4500-node instance: sorted and tiled sweeps agree move for move on every tour."""
    xy = rand_instance(4500)
    rng = np.random.default_rng(8)
    succ0 = np.stack([random_tour(4500, rng) for _ in range(3)])
    obj0 = np.zeros(3)
    out = []
    for min_n in ("0", "1000000000"):
        monkeypatch.setenv("TSP_SORTED_MIN_N", min_n)
        inst = eng.Instance(ctx, xy, O.EUC_2D, 1)
        tours = eng.Tours(inst, 3)
        tours.upload(succ0, obj0)
        tours.run(eng.BEST, max_steps=60)
        s, o, st = tours.download()
        out.append((s.copy(), [(x["moves"], x["reversed"], x["evals"]) for x in st]))
        tours.close(); inst.close()
    assert (out[0][0] == out[1][0]).all() and out[0][1] == out[1][1]
    assert all(m[0] == 60 for m in out[0][1])


def test_sorted_sweep_time_limit_leaves_a_consistent_tour(eng, ctx):
    """A run cut short by the time limit (tabusearch.c:131 checks it per sweep): status 2, a valid tour in the
    first copy of order/pos, and the recomputed cost the reference reports on every exit path (:168-172)."""
    xy, wt = load_instance("rand10000")
    inst = eng.Instance(ctx, xy, wt, 1)
    succ, obj, _ = inst.construct(eng.GREEDY, np.array([0], dtype=np.int32))
    rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=eng.BEST, time_limit=0.004)
    assert rc == eng.TIME_LIMIT_EXCEEDED and 0 < st["sweeps"] < 1428
    assert O.is_tour(s) and o == O.succ_cost(xy, wt, s)
    rc2, s2, o2, st2 = inst.two_opt(s, o, mode=eng.BEST)          # and the descent can be resumed from it
    big = golden("oracle_vectors_big.json")["rand10000_best"]["final"]   # the oracle's full CPU descent
    assert rc2 == 0 and o2 == big["cost"] and O.fnv1a(s2) == big["hash"]
    assert st["sweeps"] + st2["sweeps"] == big["stats"]["sweeps"]
    inst.close()


@pytest.mark.parametrize("name,ic", [("att532", 1), ("dsj1000", 1), ("d493", 1), ("d493", 0), ("kroA100", 0), ("rand2000", 1)])
def test_greedy_spatial_and_dense_kernels_agree_with_the_oracle(eng, ctx, monkeypatch, name, ic):
    """k_construct_nn (nearest neighbour over the Hilbert groups, one wave per start; packed keys for integer
    costs, the generic reduction for --fcost) and k_construct_lds (every candidate every step) against greedy()."""
    xy, wt = load_instance(name)
    n = len(xy)
    starts = np.array([0, n // 2, n - 1, 7], dtype=np.int32)
    exp = [O.greedy(xy, wt, start=int(s0), integer_cost=ic) for s0 in starts]
    for nn in ("1", "0"):
        monkeypatch.setenv("TSP_CONSTRUCT_NN", nn)
        inst = eng.Instance(ctx, xy, wt, ic)
        succ, obj, _ = inst.construct(eng.GREEDY, starts)
        inst.close()
        for b in range(len(starts)):
            assert (succ[b] == exp[b][1]).all() and obj[b] == exp[b][2], (nn, b)


def test_greedy_spatial_kernel_with_coincident_points(eng, ctx):
    rng = np.random.default_rng(21)
    xy = rng.integers(0, 15, size=(700, 2)).astype(np.float64)     # heavy ties: the lowest index must win
    inst = eng.Instance(ctx, xy, O.EUC_2D, 1)
    starts = np.arange(0, 700, 97, dtype=np.int32)
    succ, obj, _ = inst.construct(eng.GREEDY, starts)
    inst.close()
    for b, s0 in enumerate(starts):
        _, es, eo = O.greedy(xy, O.EUC_2D, start=int(s0))
        assert (succ[b] == es).all() and obj[b] == eo


@pytest.mark.parametrize("ic", [1, 0])
def test_greedy_large_instance_kernel_matches_the_oracle(eng, ctx, ic):
    """n = 17 000 does not fit in LDS: k_construct_nn_big (coordinates in L2, supergroups) against greedy();
    integer costs use the packed keys, --fcost the generic reduction."""
    rng = np.random.default_rng(17)
    xy = rng.integers(0, 400000, size=(17000, 2)).astype(np.float64) if ic else rng.uniform(0, 4e5, size=(17000, 2))
    inst = eng.Instance(ctx, xy, O.EUC_2D, ic)
    succ, obj, _ = inst.construct(eng.GREEDY, np.array([123, 16999], dtype=np.int32))
    inst.close()
    for b, s0 in enumerate((123, 16999)):
        _, es, eo = O.greedy(xy, O.EUC_2D, start=s0, integer_cost=ic)
        assert (succ[b] == es).all() and obj[b] == eo


def _edge_cases():
    rng = np.random.default_rng(9)
    cases = []
    for n in (4, 5, 6, 64, 128, 512):
        cases.append(("identical-%d" % n, np.full((n, 2), 7.0), O.EUC_2D, 1))
        cases.append(("collinear-%d" % n, np.stack([np.arange(n) * 3.0, np.zeros(n)], 1), O.EUC_2D, 1))
    cases.append(("huge-coordinates", rng.integers(10**9, 10**9 + 5 * 10**6, size=(300, 2)).astype(np.float64), O.EUC_2D, 1))
    cases.append(("span-above-the-integer-variant-bound", rng.integers(0, 3 * 10**6, size=(300, 2)).astype(np.float64), O.EUC_2D, 1))
    cases.append(("negative-non-integer", rng.uniform(-1e4, 1e4, size=(257, 2)), O.ATT, 1))
    cases.append(("two-far-clusters", np.vstack([rng.integers(0, 50, (100, 2)), rng.integers(10**6, 10**6 + 50, (100, 2))]).astype(np.float64), O.CEIL_2D, 1))
    cases.append(("fcost-collinear", np.stack([np.arange(130) * 0.5, np.arange(130) * 0.25], 1), O.EUC_2D, 0))
    return cases


@pytest.mark.parametrize("case", _edge_cases(), ids=lambda c: c[0])
def test_edge_inputs_all_kernels(eng, ctx, sorted_always, case):
    """Tiny n, identical and collinear points (every distance tied or zero), huge / negative / non-integer
    coordinates, sizes on the group and tile boundaries: construction, both rules (sorted sweep forced), both engines."""
    name, xy, wt, ic = case
    n = len(xy)
    rng = np.random.default_rng(n)
    inst = eng.Instance(ctx, xy, wt, ic)
    for s0 in (0, n - 1):
        succ, obj, _ = inst.construct(eng.GREEDY, np.array([s0], dtype=np.int32))
        _, es, eo = O.greedy(xy, wt, start=s0, integer_cost=ic)
        assert (succ[0] == es).all() and obj[0] == eo
    tour = random_tour(n, rng)
    cost = O.succ_cost(xy, wt, tour, integer_cost=ic)
    _, fs, fo, fst, _ = O.two_opt_first(xy, wt, tour, cost, integer_cost=ic)
    for engine in ((1, 2) if n >= 8 else (1,)):
        rc, s, o, st = inst.two_opt(tour, cost, mode=eng.FIRST, engine=engine)
        assert (s == fs).all() and o == fo
        assert (st["sweeps"], st["evals"], st["moves"]) == (fst["sweeps"], fst["evals"], fst["moves"])
    if n <= 300:
        rc, s, o, st = inst.two_opt(tour, cost, mode=eng.BEST, engine=1)
        _, bs, bo, bst, _, _ = O.two_opt_best(xy, wt, tour, integer_cost=ic)
        assert (s == bs).all() and o == bo
        assert (st["sweeps"], st["evals"], st["moves"]) == (bst["sweeps"], bst["evals"], bst["moves"])
    inst.close()


@pytest.mark.parametrize("name", ["att532", "rand2000", "d493"])
def test_grasp_spatial_and_dense_kernels_agree_with_the_oracle(eng, ctx, monkeypatch, name):
    """grasp() (heuristics.c:82-156) through k_construct_nn (the runner-up is a second nearest-neighbour query among
    the nodes with a smaller index than the winner) and through k_construct_lds, same URAND stream as the oracle."""
    xy, wt = load_instance(name)
    n = len(xy)
    starts = np.array([5, n - 1], dtype=np.int32)
    exp, urand = [], np.zeros((2, n))
    for b, s0 in enumerate(starts):
        O.srandom(100 + b)
        urand[b] = [O.urand() for _ in range(n)]
        O.srandom(100 + b)
        exp.append(O.grasp(xy, wt, start=int(s0)))
    for nn in ("1", "0"):
        monkeypatch.setenv("TSP_CONSTRUCT_NN", nn)
        inst = eng.Instance(ctx, xy, wt, 1)
        succ, obj, _ = inst.construct(eng.GRASP, starts, urand)
        inst.close()
        for b in range(2):
            assert (succ[b] == exp[b][1]).all() and obj[b] == exp[b][2], (nn, b)
