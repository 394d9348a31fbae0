#!/bin/bash
# the same capped tabu() run through every path (in-kernel chains, queued chains, one iteration per wait), a fresh process each:
# the incumbents must agree.  usage (through gpurun): bash tools/tabu_paths.sh [instance] [iterations] [policy 0 step / 1 linear / 2 random]
R=${GRAFT_REPO_ROOT:-/root/repo}
I=${1:-rand10000}; N=${2:-2000}; P=${3:-0}
for mode in 1 0; do for chain in auto 1 7 64; do
  if [ $chain = auto ]; then unset TSP_TABU_CHAIN; else export TSP_TABU_CHAIN=$chain; fi
  echo -n "inkernel=$mode chain=$chain: "; TSP_HOST_STATS= TSP_TABU_INKERNEL=$mode timeout 300 python3 $R/tools/tabu_one.py $I $N $P 2>&1 | tail -1
done; done
