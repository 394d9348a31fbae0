"""Shared helpers for the test-suite (inputs, golden files).  The oracle is imported here only
because tests are one of the three places allowed to use it."""
import json
import os

import numpy as np

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
INSTANCES = os.path.join(GOLDEN, "instances")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def rand_instance(n, seed=None, hi=1_000_000):
    """uniform integer coordinates in [0, 1e6)^2, numpy PCG64 seeded with n."""
    rng = np.random.default_rng(n if seed is None else seed)
    return rng.integers(0, hi, size=(n, 2)).astype(np.float64)


def load_instance(name):
    """-> (xy, wtype) for a TSPLIB fixture name or 'rand<n>'."""
    if name.startswith("rand"):
        return rand_instance(int(name[4:])), O.EUC_2D
    return O.parse_tsplib(os.path.join(INSTANCES, name + ".tsp"))


def random_tour(n, rng):
    perm = rng.permutation(n).astype(np.int32)
    succ = np.empty(n, dtype=np.int32)
    succ[perm] = np.roll(perm, -1)
    return succ
