#!/bin/bash
# kernel trace of one capped HEU_VNS run through the C host; through gpurun
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_vns
python3 $R/tools/vns_one.py ${1:-rand10000} ${2:-300}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vns -- python3 $R/tools/vns_one.py ${1:-rand10000} ${2:-300} > $O/vns_one.txt 2> $O/vns_one.log
python3 $R/tools/summarize_rocprof.py stats $O/prof_vns $O/vns_kernel_stats.csv > /dev/null
rm -rf $O/prof_vns
cat $O/vns_one.txt; head -14 $O/vns_kernel_stats.csv
