#!/bin/bash
# kernel-trace summary of one bench.py run (through gpurun): tools/trace_quick.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-tq}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline --no-variants "$@" > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}.log
python3 $R/tools/summarize_rocprof.py stats $R/gpurun_out/prof_$TAG $R/gpurun_out/${TAG}_stats.csv
rm -rf $R/gpurun_out/prof_$TAG
head -9 $R/gpurun_out/${TAG}_stats.csv
