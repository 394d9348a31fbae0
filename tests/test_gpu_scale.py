"""GPU, BASELINE.json's full sizes: parity with the oracle where it finishes in seconds and
size-independent properties elsewhere."""
import numpy as np
import pytest

from oracle import oracle as O
from helpers import golden, load_instance

pytestmark = pytest.mark.gpu
APB = golden("survey_appendix_b.json")
BIG = golden("oracle_vectors_big.json")   # full CPU descents (oracle, minutes): make_golden_big.py


@pytest.fixture(scope="module")
def eng():
    from tsp_optimization_amd import engine as E
    assert E.device_count() >= 1
    return E


@pytest.fixture(scope="module")
def ctx(eng):
    c = eng.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", ["rand5000", "rand10000"])
def test_first_improvement_full_size_matches_reference_counters(eng, ctx, name):
    xy, wt = load_instance(name)
    inst = eng.Instance(ctx, xy, wt, 1)
    succ, obj, _ = inst.construct(eng.GREEDY, np.array([0], dtype=np.int32))
    e = APB[name]
    assert obj[0] == e["greedy"]
    rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=eng.FIRST)
    inst.close()
    assert O.is_tour(s)
    assert (o, st["sweeps"], st["evals"], st["moves"]) == \
        (e["first"]["cost"], e["first"]["sw"], e["first"]["ev"], e["first"]["mv"])
    assert o == O.succ_cost(xy, wt, s)


def test_best_improvement_rand2000_matches_reference_counters(eng, ctx):
    xy, wt = load_instance("rand2000")
    inst = eng.Instance(ctx, xy, wt, 1)
    succ, obj, _ = inst.construct(eng.GREEDY, np.array([0], dtype=np.int32))
    e = APB["rand2000"]
    assert obj[0] == e["greedy"]
    rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=eng.BEST)
    inst.close()
    assert (o, st["sweeps"], st["evals"], st["moves"]) == \
        (e["best"]["cost"], e["best"]["sw"], e["best"]["ev"], e["best"]["mv"])


def test_best_improvement_rand10000_properties(eng, ctx):
    """The CPU needs ~0.5 h for this descent; check what does not need it: a valid tour, cost equal
    to the recomputed cost, strictly fewer sweeps than n, idempotence (a second run makes no move),
    and the first 3 sweeps equal to the oracle's."""
    xy, wt = load_instance("rand10000")
    inst = eng.Instance(ctx, xy, wt, 1)
    _, succ0, obj0 = O.greedy(xy, wt)
    _, es, eo, est, tr, _ = O.two_opt_best(xy, wt, succ0, max_sweeps=3, trace_cap=3)
    tours = eng.Tours(inst, 1)
    tours.upload(succ0, obj0)
    tours.run(eng.BEST, max_steps=3)
    s3, _, st3 = tours.download()
    tours.close()
    assert (s3[0] == es).all() and st3[0]["moves"] == 3
    rc, s, o, st = inst.two_opt(succ0, obj0, mode=eng.BEST)
    assert O.is_tour(s) and o == O.succ_cost(xy, wt, s) and o < obj0
    assert st["moves"] == st["sweeps"] - 1
    rc2, s2, o2, st2 = inst.two_opt(s, o, mode=eng.BEST)
    assert st2["moves"] == 0 and (s2 == s).all() and o2 == o
    # a best-improvement local optimum is also a first-improvement local optimum
    rc3, s3b, o3, st3b = inst.two_opt(s, o, mode=eng.FIRST)
    assert st3b["moves"] == 0 and st3b["sweeps"] == 1 and st3b["evals"] == 10000 * 9999 // 2 - 10000
    inst.close()


@pytest.mark.parametrize("n,integer_coords", [(20011, True), (12007, False), (11003, True), (6007, False)])
def test_large_odd_sizes_prefix_of_both_trajectories(eng, ctx, n, integer_coords):
    """Sizes beyond the BASELINE configs (odd n, tiles that do not divide, general and integer-coordinate
    variants): the first best-improvement sweeps and the first first-improvement moves equal the oracle's.  11 003 and 6 007 still
    fit the CLUSTER engine's sorted scan, with room for four or five staged group pairs instead of eight (several rounds of
    staging per step)."""
    rng = np.random.default_rng(n)
    xy = rng.integers(0, 700_000, size=(n, 2)).astype(np.float64)
    if not integer_coords:
        xy = xy + rng.integers(0, 4, size=(n, 2)) * 0.25
    wt = O.EUC_2D
    inst = eng.Instance(ctx, xy, wt, 1)
    succ0, obj0, _ = inst.construct(eng.GREEDY, np.array([7], dtype=np.int32))
    _, es0, eo0 = O.greedy(xy, wt, start=7)
    assert obj0[0] == eo0 and (succ0[0] == es0).all()
    tours = eng.Tours(inst, 1)
    # best improvement: 2 sweeps
    tours.upload(es0, eo0)
    tours.run(eng.BEST, max_steps=2)
    s, _, st = tours.download()
    _, eb, _, _, _, _ = O.two_opt_best(xy, wt, es0, max_sweeps=2)
    assert (s[0] == eb).all() and st[0]["moves"] == 2
    # first improvement: step until 25 moves have been applied
    tours.reset()
    moves = 0
    for _ in range(400):
        tours.run(eng.FIRST, max_steps=1)
        s, o, st = tours.download()
        moves = st[0]["moves"]
        if moves >= 25:
            break
    assert moves == 25
    ef, eof, est = O.two_opt_first_moves(xy, wt, es0, eo0, 25)
    assert (s[0] == ef).all() and o[0] == eof and st[0]["reversed"] == est["reversed"]
    tours.close()
    inst.close()


def _grid_instance(n):
    return np.random.default_rng(n).integers(0, 700_000, size=(n, 2)).astype(np.float64)


def test_full_alg_2opt_descent_beyond_the_lds_engines(eng, ctx):
    """n = 20 011: no replica fits a CU's LDS, the whole first-improvement descent runs on the GRID engine (k_first, tour state
    in HBM).  Final tour, cost and every counter against the oracle's committed vector (make_golden_grid.py)."""
    g = golden("oracle_vectors_grid.json")["rand20011_first"]
    xy = _grid_instance(20011)
    inst = eng.Instance(ctx, xy, O.EUC_2D, 1)
    succ, obj, _ = inst.construct(eng.GREEDY, np.array([g["start"]], dtype=np.int32))
    assert obj[0] == g["greedy"]["obj"] and O.fnv1a(succ[0]) == g["greedy"]["hash"]
    rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=eng.FIRST)
    inst.close()
    f = g["final"]
    assert rc == 0 and o == f["cost"] and O.fnv1a(s) == f["hash"]
    assert (st["sweeps"], st["evals"], st["moves"], st["reversed"]) == \
        (f["stats"]["sweeps"], f["stats"]["evals"], f["stats"]["moves"], f["stats"]["reversed"])
    assert st["exact_pairs"] == -1          # not the CLUSTER engine


def test_more_nodes_than_a_uint16_id_holds(eng, ctx):
    """n = 70 001 (the reference ships data/art/stefano_128k.tsp and pla85900): the LDS-resident engines index nodes with 16
    bits and must not be picked; greedy, the first two best-improvement sweeps and the first 60 first-improvement moves on the
    GRID engine equal the oracle's committed vectors.  The tabu stamp array of such an instance is refused (int index)."""
    g = golden("oracle_vectors_grid.json")["rand70001"]
    xy = _grid_instance(70001)
    inst = eng.Instance(ctx, xy, O.EUC_2D, 1)
    succ, obj, _ = inst.construct(eng.GREEDY, np.array([g["start"]], dtype=np.int32))
    assert obj[0] == g["greedy"]["obj"] and O.fnv1a(succ[0]) == g["greedy"]["hash"]
    tours = eng.Tours(inst, 1)
    tours.upload(succ[0], obj[0])
    rc, done = tours.run_engine(eng.BEST, engine=eng.ENGINE_AUTO, max_steps=2)
    s, o, st = tours.download()
    assert O.fnv1a(s[0]) == g["best_2_sweeps"]["hash"] and o[0] == g["best_2_sweeps"]["cost"] and st[0]["moves"] == 2
    assert st[0]["evals"] == g["best_2_sweeps"]["evals"]
    tours.reset()
    moves = 0
    for _ in range(2000):
        tours.run(eng.FIRST, max_steps=1)
        s, o, st = tours.download()
        moves = st[0]["moves"]
        if moves >= 60:
            break
    f = g["first_60_moves"]
    assert moves == 60 and O.fnv1a(s[0]) == f["hash"] and o[0] == f["cost"] and st[0]["reversed"] == f["reversed"]
    tours.close()
    for engine in (eng.ENGINE_LDS, eng.ENGINE_CLUSTER):
        with pytest.raises(eng.TspDeviceError):
            inst.two_opt(succ[0], obj[0], mode=eng.FIRST, engine=engine)
    with pytest.raises(eng.TspDeviceError):
        eng.Tabu(inst)
    inst.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_time_limit_stops_with_a_valid_tour_and_status_2(eng, ctx, mode):
    """TIME_LIMIT_EXCEEDED (include/heuristics.h:7): the descent stops between launch batches with a valid tour;
    alg_2opt's obj_best still equals start + applied deltas, alg_2opt_tabu's is the recomputed cost (:168-172)."""
    xy, wt = load_instance("rand10000")
    inst = eng.Instance(ctx, xy, wt, 1)
    _, succ0, obj0 = O.greedy(xy, wt)
    rc, s, o, st = inst.two_opt(succ0, obj0, mode=mode, engine=eng.ENGINE_GRID, time_limit=0.004)
    inst.close()
    assert rc == eng.TIME_LIMIT_EXCEEDED
    assert O.is_tour(s) and 0 < st["moves"] < (2704 if mode == 0 else 1427)
    assert o == O.succ_cost(xy, wt, s) and o < obj0


# ---- full best-improvement descents and the 128-individual population against the committed oracle vectors -----------
@pytest.mark.parametrize("engine", [1, 3])     # GRID (k_move_recs + k_sweep), CLUSTER (sorted scan, 256 workgroups)
@pytest.mark.parametrize("name", ["rand5000", "rand10000"])
def test_best_improvement_full_descent_equals_golden(eng, ctx, name, engine):
    """alg_2opt_tabu(skip_edge == NULL) from greedy(0) to the local optimum at BASELINE's sizes: final tour, recomputed cost,
    sweeps, evaluations, moves and reversal length equal the oracle's full CPU descent (rand10000: 1428 sweeps, 23 CPU
    minutes; SURVEY.md Appendix B left these cells blank), and so do the tours after 1, 10, 100 and 500 sweeps."""
    g = BIG[name + "_best"]
    xy, wt = load_instance(name)
    inst = eng.Instance(ctx, xy, wt, 1)
    succ, obj, _ = inst.construct(eng.GREEDY, np.array([0], dtype=np.int32))
    assert obj[0] == g["greedy"]["obj"] and O.fnv1a(succ[0]) == g["greedy"]["hash"]
    tours = eng.Tours(inst, 1)
    for cp in g["checkpoints"]:
        tours.upload(succ[0], obj[0])
        tours.run_engine(eng.BEST, engine=engine, max_steps=cp["sweeps"])
        s, o, st = tours.download()
        assert O.fnv1a(s[0]) == cp["hash"] and o[0] == cp["cost"], cp["sweeps"]
        assert (st[0]["sweeps"], st[0]["moves"], st[0]["evals"]) == (cp["sweeps"], cp["moves"], cp["evals"])
    tours.upload(succ[0], obj[0])
    rc, done = tours.run_engine(eng.BEST, engine=engine)
    s, o, st = tours.download()
    tours.close(); inst.close()
    f = g["final"]
    assert rc == 0 and done and O.fnv1a(s[0]) == f["hash"] and o[0] == f["cost"]
    assert {k: st[0][k] for k in ("sweeps", "evals", "moves", "reversed")} == f["stats"]


def test_population_refinement_config5_all_128_individuals_equal_golden(eng, ctx):
    """BASELINE configs[4] at full size: 128 random individuals of rand5000 (src/genetic.c:349-364, libc stream seeded
    with 123), each refined by alg_2opt: per individual the fitness, final cost, tour hash, sweeps, evaluations, moves
    and reversal length of the oracle's table."""
    g = BIG["config5_rand5000_pop128"]
    xy, wt = load_instance("rand5000")
    O.srandom(g["seed"])
    perms = np.stack([O.random_perm(g["n"]) for _ in range(g["population"])])
    succ = np.stack([O.perm_to_succ(p) for p in perms])
    inst = eng.Instance(ctx, xy, wt, 1)
    cost = inst.perm_cost(perms)
    rc, s2, o2, st = inst.two_opt(succ, cost, mode=eng.FIRST)
    inst.close()
    assert rc == 0
    for k, row in enumerate(g["individuals"]):
        assert O.fnv1a(perms[k]) == row["perm_hash"] and cost[k] == row["fitness"], k
        assert o2[k] == row["cost"] and O.fnv1a(s2[k]) == row["hash"], k
        assert (st[k]["sweeps"], st[k]["evals"], st[k]["moves"], st[k]["reversed"]) == \
            (row["sw"], row["ev"], row["mv"], row["reversed"]), k
