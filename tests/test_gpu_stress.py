"""GPU: a seeded slice of the randomised parity sweep (tools/stress_parity.py: 3 500 cases were run by hand in round 2) under
`-m gpu`, so that the randomised evidence is driver-run: 6 seeds x 25 cases = 150 cases -- random sizes around the tile / group /
batch boundaries (n = 3 included), EUC_2D / ATT / CEIL_2D / MAN_2D / MAX_2D, integer and --fcost costs, random and greedy tours,
both rules, GRID / LDS / CLUSTER engines with random cluster sizes and probe settings, a dense random tabu list per small case,
batches, greedy and GRASP construction -- every result (tour, cost, sweeps, evaluations, moves, reversal length, whole stamp
array) against the oracle."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def ctx():
    from tsp_optimization_amd import engine as E
    c = E.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", [3001, 3002, 3003, 3004, 3005, 3006])
def test_randomised_parity_slice(ctx, seed):
    import stress_parity
    assert stress_parity.run(seed, 25, ctx=ctx, verbose=False, max_n=1500) == 0
