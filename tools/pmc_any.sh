#!/bin/bash
# rocprofv3 --pmc pass over any python tool: tools/pmc_any.sh <tag> "<counters>" <script.py> ; prints the per-kernel means
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; CTR=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 "$@" > $R/gpurun_out/$TAG.out 2> $R/gpurun_out/$TAG.log
python3 $R/tools/summarize_rocprof.py pmc $R/gpurun_out/prof_$TAG $R/gpurun_out/${TAG}.json > /dev/null
rm -rf $R/gpurun_out/prof_$TAG
cat $R/gpurun_out/$TAG.out
python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/${TAG}.json"))
for k,v in d.items():
    if "$KERNEL" in k:
        print(k)
        for c,x in sorted(v.items()): print("  %-24s %16.1f  (%d dispatches)" % (c, x["mean"], x["dispatches"]))
PY
