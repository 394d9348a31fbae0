#!/usr/bin/env python3
"""Oracle vectors for tours beyond the LDS-resident engines (GRID engine only), run in the build container only:

* rand20011_first: the COMPLETE alg_2opt descent (first improvement, src/heuristics.c:438-502) of greedy(7) on 20 011 random
  integer points: final tour hash, cost, sweeps, delta evaluations, moves, reversed length.
* rand70001: more nodes than a uint16 id holds (data/art/stefano_128k.tsp, pla85900 are of this kind): greedy(7), the tour
  after the first 2 best-improvement sweeps (src/tabusearch.c:107-178) and after the first 60 first-improvement moves.

Provenance: produced by oracle/tsp_oracle.c (pinned by reference_results.json and survey_appendix_b.json); not reference-held.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def inst(n):
    return np.random.default_rng(n).integers(0, 700_000, size=(n, 2)).astype(np.float64)


def main():
    out = {"_source": "tests/golden/make_golden_grid.py (oracle/tsp_oracle.c)"}
    t0 = time.time()
    xy = inst(20011)
    _, g, gobj = O.greedy(xy, O.EUC_2D, start=7)
    _, s, o, st, _ = O.two_opt_first(xy, O.EUC_2D, g, gobj)
    st = dict(st); st.pop("seconds", None)
    out["rand20011_first"] = {"n": 20011, "start": 7, "greedy": {"obj": gobj, "hash": O.fnv1a(g)},
                              "final": {"hash": O.fnv1a(s), "cost": o, "stats": st}}
    print("rand20011 first: %.0f -> %.0f, %s in %.0f s" % (gobj, o, st, time.time() - t0), flush=True)
    t0 = time.time()
    xy = inst(70001)
    _, g, gobj = O.greedy(xy, O.EUC_2D, start=7)
    _, b2, bo, bst, _, _ = O.two_opt_best(xy, O.EUC_2D, g, max_sweeps=2)
    f60, fo, fst = O.two_opt_first_moves(xy, O.EUC_2D, g, gobj, 60)
    out["rand70001"] = {"n": 70001, "start": 7, "greedy": {"obj": gobj, "hash": O.fnv1a(g)},
                        "best_2_sweeps": {"hash": O.fnv1a(b2), "cost": bo, "moves": bst["moves"], "evals": bst["evals"]},
                        "first_60_moves": {"hash": O.fnv1a(f60), "cost": fo, "reversed": fst["reversed"], "moves": fst["moves"]}}
    print("rand70001: greedy %.0f, best 2 sweeps %.0f, first 60 moves %.0f in %.0f s" % (gobj, bo, fo, time.time() - t0), flush=True)
    with open(os.path.join(HERE, "oracle_vectors_grid.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
