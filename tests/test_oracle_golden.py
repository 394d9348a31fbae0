"""CPU: pins the oracle (oracle/tsp_oracle.c) to the reference's published known answers
(results/*.csv cells, committed as tests/golden/reference_results.json) and to the counters
SURVEY.md Appendix B recorded from the unmodified reference."""
import numpy as np
import pytest

from oracle import oracle as O
from helpers import golden, load_instance

REF = golden("reference_results.json")["instances"]
APB = golden("survey_appendix_b.json")
VEC = golden("oracle_vectors.json")

# n^3 multistart is minutes for the GEO / n>=1000 instances: keep GREEDY_ITER to the smaller ones
ITER_OK = [k for k in REF if k in ("att532", "lin318", "pr439", "pcb442", "rd400", "d493", "rat575", "u574")]


@pytest.mark.parametrize("name", sorted(REF))
def test_greedy_and_2opt_greedy_match_reference_csv(name):
    xy, wt = load_instance(name)
    st, succ, obj = O.greedy(xy, wt)
    assert st == 0 and O.is_tour(succ)
    assert obj == REF[name]["GREEDY"]                      # results/constructive_heuristics_new.csv
    st, succ2, obj2, stats, _ = O.two_opt_first(xy, wt, succ, obj)
    assert st == 0 and O.is_tour(succ2)
    assert obj2 == REF[name]["2OPT_GREEDY"]                # results/constructive_heuristics_2opt_new.csv
    assert obj2 == O.succ_cost(xy, wt, succ2)


@pytest.mark.parametrize("name", sorted(REF))
def test_grasp_seed123_matches_reference_csv(name):
    xy, wt = load_instance(name)
    O.srandom(123)                                         # src/solver.c:264-266
    st, succ, obj = O.grasp(xy, wt)
    assert st == 0 and O.is_tour(succ)
    assert obj == REF[name]["GRASP"]
    # the reported value double-counts the closing edge (heuristics.c:135,152)
    last = int(np.where(succ == 0)[0][0])
    assert obj == O.succ_cost(xy, wt, succ) + O.dist(xy, last, 0, wt)


@pytest.mark.parametrize("name", sorted(ITER_OK))
def test_greedy_iter_and_2opt_match_reference_csv(name):
    xy, wt = load_instance(name)
    st, succ, obj = O.greedy_iter(xy, wt)
    assert obj == REF[name]["GREEDY_ITER"]
    st, succ2, obj2, stats, _ = O.two_opt_first(xy, wt, succ, obj)
    assert obj2 == REF[name]["2OPT_GREEDY_ITER"]


@pytest.mark.parametrize("name", sorted(ITER_OK))
def test_extramileage_and_2opt_match_reference_csv(name):
    xy, wt = load_instance(name)
    st, succ, obj = O.extramileage(xy, wt)
    assert st == 0 and O.is_tour(succ)
    assert obj == REF[name]["EXTR_MILE"]                    # results/constructive_heuristics_new.csv
    assert obj == O.succ_cost(xy, wt, succ)
    _, s2, o2, _, _ = O.two_opt_first(xy, wt, succ, obj)
    assert o2 == REF[name]["2OPT_EXTR_MIL"]                 # results/constructive_heuristics_2opt_new.csv


@pytest.mark.parametrize("name", ["berlin52", "pr299", "att532", "rand1000"])
def test_counters_match_survey_appendix_b(name):
    xy, wt = load_instance(name)
    exp = APB[name]
    _, succ, obj = O.greedy(xy, wt)
    assert obj == exp["greedy"]
    _, s1, o1, st1, _ = O.two_opt_first(xy, wt, succ, obj)
    assert (o1, st1["sweeps"], st1["evals"], st1["moves"]) == \
        (exp["first"]["cost"], exp["first"]["sw"], exp["first"]["ev"], exp["first"]["mv"])
    if "reversed" in exp["first"]:
        assert st1["reversed"] == exp["first"]["reversed"]
    _, s2, o2, st2, _, _ = O.two_opt_best(xy, wt, succ)
    assert (o2, st2["sweeps"], st2["evals"], st2["moves"]) == \
        (exp["best"]["cost"], exp["best"]["sw"], exp["best"]["ev"], exp["best"]["mv"])
    if "grasp123" in exp:
        O.srandom(123)
        _, r, ro = O.grasp(xy, wt)
        assert ro == exp["grasp123"]["reported"]
        _, r2, ro2, st3, _ = O.two_opt_first(xy, wt, r, ro)
        assert ro2 == exp["first_from_grasp123"]["reported"]
        if "true" in exp["first_from_grasp123"]:
            assert O.succ_cost(xy, wt, r2) == exp["first_from_grasp123"]["true"]
        if "ev" in exp["first_from_grasp123"]:
            e = exp["first_from_grasp123"]
            assert (st3["sweeps"], st3["evals"], st3["moves"]) == (e["sw"], e["ev"], e["mv"])


def test_rand5000_first_improvement_matches_survey():
    xy, wt = load_instance("rand5000")
    exp = APB["rand5000"]
    _, succ, obj = O.greedy(xy, wt)
    assert obj == exp["greedy"]
    _, s1, o1, st1, _ = O.two_opt_first(xy, wt, succ, obj)
    assert (o1, st1["sweeps"], st1["evals"], st1["moves"]) == \
        (exp["first"]["cost"], exp["first"]["sw"], exp["first"]["ev"], exp["first"]["mv"])


def test_att532_multistart_stream_matches_survey():
    """heuristics.c:519 then :127: one draw for the start node, then n draws inside grasp()."""
    xy, wt = load_instance("att532")
    n = len(xy)
    exp = APB["att532"]["multistart256"]
    O.srandom(123)
    for k in range(2):
        node = int(O.urand() * (n - 1))
        _, succ, obj = O.grasp(xy, wt, start=node)
        _, s2, o2, _, _ = O.two_opt_first(xy, wt, succ, obj)
        e = exp["start%d" % k]
        assert (node, obj, O.succ_cost(xy, wt, succ), o2, O.succ_cost(xy, wt, s2)) == \
            (e["node"], e["grasp_reported"], e["grasp_true"], e["opt_reported"], e["opt_true"])
    table = VEC["att532_multistart256"]
    best = min(table, key=lambda r: (r["opt_true"], r["k"]))
    assert (best["opt_true"], best["k"]) == (exp["best_true"], exp["best_start"])


def test_oracle_vectors_are_reproducible():
    for name in ["berlin52", "pr299"]:
        xy, wt = load_instance(name)
        c = VEC["cases"][name]
        _, succ, obj = O.greedy(xy, wt)
        assert O.fnv1a(succ) == c["greedy"]["hash"]
        _, s1, o1, st1, tr = O.two_opt_first(xy, wt, succ, obj, trace_cap=4096)
        assert O.fnv1a(s1) == c["first"]["hash"]
        assert [list(m) for m in tr[:96]] == [list(m) for m in c["first"]["trace"]]


def test_grasp_urand_array_equals_libc_stream():
    xy, wt = load_instance("berlin52")
    n = len(xy)
    O.srandom(7)
    u = np.array([O.urand() for _ in range(n)])
    O.srandom(7)
    _, a, ao = O.grasp(xy, wt, start=3)
    _, b, bo = O.grasp(xy, wt, start=3, urand=u)
    assert ao == bo and (a == b).all()


def test_metric_quirks():
    xy = np.array([[0.0, 0.0], [3.0, 4.0], [10.4, 7.7]])
    assert O.dist(xy, 0, 1, O.EUC_2D) == 5.0
    assert O.dist(xy, 0, 1, O.CEIL_2D) == 5.0
    assert O.dist(xy, 0, 2, O.CEIL_2D) == np.ceil(np.sqrt(10.4 ** 2 + 7.7 ** 2))
    # MAN_2D / MAX_2D carry the reference's dy = |y2 - y2| = 0 (distutil.c:35,41)
    assert O.dist(xy, 0, 1, O.MAN_2D) == 3.0
    assert O.dist(xy, 0, 1, O.MAX_2D) == 3.0
    # ATT rounds up (distutil.c:26-27)
    r = np.sqrt((3.0 ** 2 + 4.0 ** 2) / 10.0)
    assert O.dist(xy, 0, 1, O.ATT) == np.ceil(r)
    assert O.dist(xy, 0, 1, O.ATT, 0) == r
    assert O.dist(xy, 0, 2, O.EUC_2D, 0) == np.sqrt(10.4 * 10.4 + 7.7 * 7.7)


def test_best_improvement_with_tabu_skips_and_expires():
    xy, wt = load_instance("berlin52")
    n = len(xy)
    _, succ, _ = O.greedy(xy, wt)
    tabu = np.zeros(n * (n - 1) // 2, dtype=np.int32)
    _, s_free, o_free, st_free, tr_free, _ = O.two_opt_best(xy, wt, succ, max_sweeps=1, trace_cap=4)
    i, j, _ = tr_free[0]
    tabu[O.lib().orc_udir_pos(i, j, n)] = 5            # the best move's new edge is tabu at iter 6
    _, s_t, o_t, st_t, tr_t, prev = O.two_opt_best(xy, wt, succ, tabu=tabu, iter_=6, tenure=3,
                                                     max_sweeps=1, trace_cap=4, want_prev=True)
    assert tr_t[0][:2] != (i, j)
    assert (prev[s_t] == np.arange(n)).all()
    # expired at iter 10 (10-5 > 3): cleared lazily and the move is allowed again
    _, s_e, o_e, st_e, tr_e, _ = O.two_opt_best(xy, wt, succ, tabu=tabu, iter_=10, tenure=3,
                                                 max_sweeps=1, trace_cap=4)
    assert tr_e[0][:2] == (i, j) and tabu[O.lib().orc_udir_pos(i, j, n)] == 0


def test_big_vectors_are_consistent_with_the_oracle_prefixes():
    """oracle_vectors_big.json (full descents, minutes of CPU) is not regenerated here; its cheap prefixes are: greedy(0),
    the tour after 1 and after 10 best-improvement sweeps of rand5000, and the first individuals' fitness of config 5."""
    big = golden("oracle_vectors_big.json")
    xy, wt = load_instance("rand5000")
    _, g, gobj = O.greedy(xy, wt)
    e = big["rand5000_best"]
    assert gobj == e["greedy"]["obj"] and O.fnv1a(g) == e["greedy"]["hash"]
    for cp in e["checkpoints"][:2]:
        _, s, o, st, _, _ = O.two_opt_best(xy, wt, g, max_sweeps=cp["sweeps"])
        assert (O.fnv1a(s), o, st["moves"], st["evals"]) == (cp["hash"], cp["cost"], cp["moves"], cp["evals"])
    c5 = big["config5_rand5000_pop128"]
    O.srandom(c5["seed"])
    for row in c5["individuals"][:3]:
        p = O.random_perm(c5["n"])
        assert O.fnv1a(p) == row["perm_hash"] and O.perm_cost(xy, wt, p) == row["fitness"]
