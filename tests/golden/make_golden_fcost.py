"""Golden table for the multi-GPU epilogue with non-integer costs (--fcost, src/utility.c:285): att532, seed 123, 256 GRASP
starts in HEU_Grasp_iter's draw order (src/heuristics.c:519, :127), each refined by alg_2opt with integer_cost = 0; the cost
of a start is the fitness of its refined tour walked from node 0 (src/genetic.c:51-60: the summation order matters for
doubles).  Produced by the pinned oracle (oracle-derived, not reference-held).  Doubles are stored as C99 hex strings.

    python tests/golden/make_golden_fcost.py        (about a minute of CPU)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    xy, wt = O.parse_tsplib(os.path.join(HERE, "instances", "att532.tsp"))
    n = len(xy)
    O.srandom(123)
    rows = []
    for k in range(256):
        node = int(O.urand() * (n - 1))
        u = [O.urand() for _ in range(n)]
        _, succ, obj = O.grasp(xy, wt, node, integer_cost=0, urand=u)
        _, s2, o2, st, _ = O.two_opt_first(xy, wt, succ, obj, integer_cost=0)
        cost = O.perm_cost(xy, wt, O.succ_to_perm(s2), integer_cost=0)
        rows.append({"k": k, "node": node, "cost_hex": float(cost).hex(), "reported_hex": float(o2).hex(),
                     "hash": O.fnv1a(s2), "mv": st["moves"], "sw": st["sweeps"], "ev": st["evals"]})
    best = min(rows, key=lambda r: (float.fromhex(r["cost_hex"]), r["k"]))
    out = {"_source": "oracle/tsp_oracle.c via tests/golden/make_golden_fcost.py (oracle-derived, not reference-held)",
           "instance": "att532", "seed": 123, "integer_cost": 0, "starts": rows,
           "best": {"k": best["k"], "cost_hex": best["cost_hex"], "cost": float.fromhex(best["cost_hex"])}}
    with open(os.path.join(HERE, "oracle_vectors_fcost.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote oracle_vectors_fcost.json; best", out["best"])


if __name__ == "__main__":
    main()
