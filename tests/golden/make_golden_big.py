#!/usr/bin/env python3
"""Full-size oracle vectors that take minutes of CPU (run in the build container only):

* rand5000 / rand10000: the COMPLETE best-improvement descent of alg_2opt_tabu(skip_edge == NULL)
  (src/tabusearch.c:107-178) from greedy(0): final tour hash, recomputed cost, sweeps, delta
  evaluations, moves, reversed length, and the tour hash / cost after a few fixed sweep counts so
  that a device run that goes wrong early is caught early.  SURVEY.md Appendix B leaves these
  cells blank ("CPU ~ 0.5 h").
* config5_rand5000_pop128: BASELINE configs[4] -- 128 random permutations (src/genetic.c:349-364,
  libc random() seeded with 123) of rand5000, each refined by alg_2opt (first improvement,
  src/heuristics.c:438-502): per individual the initial fitness, final cost, tour hash, sweeps,
  evaluations and moves.

Provenance: every number in oracle_vectors_big.json is produced by oracle/tsp_oracle.c, whose
restatement is pinned by reference_results.json (cells of the reference's own result CSVs) and by
survey_appendix_b.json (gcov counters of the unmodified reference recorded by the survey).  None
of the numbers written here is reference-held.
"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

CHECKPOINTS = [1, 10, 100, 500]


def rand_instance(n):
    return np.random.default_rng(n).integers(0, 1_000_000, size=(n, 2)).astype(np.float64)


def best_descent(n):
    xy = rand_instance(n)
    _, g, gobj = O.greedy(xy, O.EUC_2D)
    out = {"n": n, "greedy": {"obj": gobj, "hash": O.fnv1a(g)}, "checkpoints": []}
    for k in CHECKPOINTS:   # each from scratch: a capped run recomputes the cost on exit like the full one
        _, s, o, st, _, _ = O.two_opt_best(xy, O.EUC_2D, g, max_sweeps=k)
        out["checkpoints"].append({"sweeps": k, "hash": O.fnv1a(s), "cost": o, "moves": st["moves"],
                                   "evals": st["evals"]})
    t0 = time.time()
    _, s, o, st, tr, _ = O.two_opt_best(xy, O.EUC_2D, g, trace_cap=64)
    st = dict(st)
    secs = st.pop("seconds")
    out["final"] = {"hash": O.fnv1a(s), "cost": o, "stats": st, "first_moves": tr[:64],
                    "oracle_seconds": round(secs, 1)}
    print("rand%d best: cost %.0f sweeps %d evals %d moves %d in %.0f s" %
          (n, o, st["sweeps"], st["evals"], st["moves"], time.time() - t0), flush=True)
    return out


def config5(pop=128, n=5000):
    xy = rand_instance(n)
    O.srandom(123)
    perms = [O.random_perm(n) for _ in range(pop)]

    def one(k):
        succ = O.perm_to_succ(perms[k])
        fit = O.perm_cost(xy, O.EUC_2D, perms[k])
        _, s, o, st, _ = O.two_opt_first(xy, O.EUC_2D, succ, fit)
        return {"k": k, "perm_hash": O.fnv1a(perms[k]), "fitness": fit, "cost": o, "hash": O.fnv1a(s),
                "sw": st["sweeps"], "ev": st["evals"], "mv": st["moves"], "reversed": st["reversed"]}

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=4) as ex:   # ctypes releases the GIL; the descent has no global state
        rows = list(ex.map(one, range(pop)))
    print("config5: %d individuals in %.0f s" % (pop, time.time() - t0), flush=True)
    return {"n": n, "population": pop, "seed": 123, "individuals": rows}


def main():
    out = {"_source": "oracle/tsp_oracle.c via tests/golden/make_golden_big.py (oracle-derived, not reference-held; "
                      "the oracle is pinned by reference_results.json and survey_appendix_b.json)"}
    with ThreadPoolExecutor(max_workers=3) as ex:
        f10 = ex.submit(best_descent, 10000)
        f5 = ex.submit(best_descent, 5000)
        fc = ex.submit(config5)
        out["config5_rand5000_pop128"] = fc.result()
        out["rand5000_best"] = f5.result()
        out["rand10000_best"] = f10.result()
    with open(os.path.join(HERE, "oracle_vectors_big.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote oracle_vectors_big.json")


if __name__ == "__main__":
    main()
