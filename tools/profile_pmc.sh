#!/bin/bash
# SQ counter pass over bench.py's timed sweeps (run through gpurun).  usage: profile_pmc.sh <tag> "<counters>"
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pmc}
CTR=${2:-"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/$TAG.log
python3 $R/tools/summarize_rocprof.py pmc $R/gpurun_out/prof_$TAG $R/gpurun_out/${TAG}.json > /dev/null
rm -rf $R/gpurun_out/prof_$TAG
python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/${TAG}.json"))
for k,v in d.items():
    if "k_step<6, true, 1" in k or "k_sweep" in k or "k_move_recs" in k:
        print(k)
        for c,x in sorted(v.items()): print("%-24s %16.1f" % (c, x["mean"]))
PY
