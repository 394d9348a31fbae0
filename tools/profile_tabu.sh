#!/bin/bash
# kernel trace of one capped tabu() run through the C host (iterations inside the launch, then queued chains); through gpurun
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for mode in 1 0; do
  export TSP_TABU_INKERNEL=$mode TSP_HOST_STATS=1
  rm -rf $O/prof_tabu$mode
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tabu$mode -- python3 $R/tools/tabu_one.py rand10000 2000 > $O/tabu_one_$mode.txt 2> $O/tabu_one_$mode.log
  python3 $R/tools/summarize_rocprof.py stats $O/prof_tabu$mode $O/tabu_kernel_stats_$mode.csv > /dev/null
  rm -rf $O/prof_tabu$mode
  cat $O/tabu_one_$mode.txt; grep "\[tabu\]" $O/tabu_one_$mode.log; head -9 $O/tabu_kernel_stats_$mode.csv
done
