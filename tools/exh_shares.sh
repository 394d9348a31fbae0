#!/bin/bash
# k_exh: per-cent shares of the rows for the four age groups of the grid (TSP_EXH_SHARES), two runs each; through gpurun
for sh in "53,26,13,8" "53,27,13,7" "52,28,13,7" "54,27,12,7" "53,28,12,7" "51,28,14,7" "53,27,14,6" "52,27,14,7" "54,26,13,7" "53,26,13,8" "53,27,13,7"; do
  echo -n "shares $sh: "
  for r in 1 2; do TSP_EXH_SHARES=$sh timeout 120 python3 tools/exhaustive_time.py 4:4 2>&1 | tail -1 | sed 's/.* \([0-9.]* us per sweep\).*/\1/' | tr '\n' ' '; done; echo
done
