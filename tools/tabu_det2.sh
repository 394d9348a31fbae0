#!/bin/bash
# first differing iteration between the first (cold) and the second run of one process
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
TSP_HOST_TRACE=1 TSP_TABU_CHAIN=${1:-1} TSP_TABU_INKERNEL=${2:-0} python3 $R/tools/tabu_det.py rand10000 ${3:-300} > $O/det.out 2> $O/det.err
cat $O/det.out
python3 - <<PY
import re
runs=[]; cur=None
for l in open("$O/det.err"):
    if l.startswith("[tabu-trace] start"): cur=[l.strip()]; runs.append(cur)
    elif l.startswith("[tabu-trace]") and cur is not None: cur.append(l.strip())
print([len(r) for r in runs])
a,b=runs[0],runs[1]
for k,(x,y) in enumerate(zip(a,b)):
    if x!=y:
        print("first difference at line",k); print(" cold:",a[max(0,k-2):k+3]); print(" warm:",b[max(0,k-2):k+3]); break
else: print("identical")
PY
