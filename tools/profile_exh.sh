#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + SQ counters of the exhaustive sweep (k_move_pos + k_exh) on rand10000.
# usage: profile_exh.sh <tag> [rj:waves]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
CASE=${2:-4:4}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_exh -- python3 $R/tools/exhaustive_time.py $CASE > $O/${TAG}_exhaustive.txt 2> $O/trace_exh.log
python3 $R/tools/summarize_rocprof.py stats $O/prof_exh $O/${TAG}_kernel_stats_exhaustive.csv > /dev/null
timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $O/prof_sq_exh -- python3 $R/tools/exhaustive_time.py $CASE > /dev/null 2> $O/pmc_sq_exh.log
python3 $R/tools/summarize_rocprof.py pmc $O/prof_sq_exh $O/${TAG}_pmc_sq_exhaustive.json > /dev/null
rm -rf $O/prof_exh $O/prof_sq_exh
cat $O/${TAG}_exhaustive.txt; head -6 $O/${TAG}_kernel_stats_exhaustive.csv; python3 -c "
import json,sys; d=json.load(open('$O/${TAG}_pmc_sq_exhaustive.json')); [print(k, v) for k,v in d.items() if 'exh' in k or 'move_pos' in k]" | head -40
