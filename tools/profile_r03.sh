#!/bin/bash
# Run on the GPU box (through gpurun): the rocprofv3 passes VERDICT r02 found missing -- the first-improvement CLUSTER kernel on
# rand10000, k_lds_two_opt on configs[3] / [4], k_dist_matrix with rotating buffers -- each as a kernel trace and, separately, a
# counter pass (program directly after `--`; never --pmc together with a trace).  Summaries land in gpurun_out/<tag>_*.
# usage: profile_r03.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
run() {   # name, script, args...: trace -> <tag>_kernel_stats_<name>.csv + <tag>_<name>.txt ; SQ counters -> <tag>_pmc_sq_<name>.json
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 "$@" > $O/${TAG}_${name}.txt 2> $O/trace_$name.log
  python3 $R/tools/summarize_rocprof.py stats $O/prof_$name $O/${TAG}_kernel_stats_${name}.csv > /dev/null
  timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $O/prof_sq_$name -- python3 "$@" > /dev/null 2> $O/pmc_sq_$name.log
  python3 $R/tools/summarize_rocprof.py pmc $O/prof_sq_$name $O/${TAG}_pmc_sq_${name}.json > /dev/null
  rm -rf $O/prof_$name $O/prof_sq_$name
}
run first $R/tools/first_time.py 10
run lds $R/tools/lds_time.py 3
run dist_matrix $R/tools/dist_matrix_time.py 3
# HBM write traffic of the matrix build: WRITE_SIZE in a pass of its own (exact for 16-byte streaming stores, MI355X_MICROARCH.md)
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_dmw -- python3 $R/tools/dist_matrix_time.py 2 > /dev/null 2> $O/pmc_dmw.log
python3 $R/tools/summarize_rocprof.py pmc $O/prof_dmw $O/${TAG}_pmc_WRITE_SIZE_dist_matrix.json > /dev/null
# clock under load: GRBM_GUI_ACTIVE (sum over the 8 XCDs) / 8 / kernel time = effective clock (MI355X_MICROARCH.md, DVFS give-back)
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/prof_clk -- python3 $R/tools/exhaustive_time.py 32 > $O/${TAG}_exhaustive_clock.txt 2> $O/pmc_clk.log
python3 $R/tools/summarize_rocprof.py pmc $O/prof_clk $O/${TAG}_pmc_clock_exhaustive.json > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_clk2 -- python3 $R/tools/exhaustive_time.py 32 > /dev/null 2> $O/trace_clk2.log
python3 $R/tools/summarize_rocprof.py stats $O/prof_clk2 $O/${TAG}_kernel_stats_exhaustive_clock.csv > /dev/null
rm -rf $O/prof_dmw $O/prof_clk $O/prof_clk2
for f in first lds dist_matrix; do echo "== $f"; cat $O/${TAG}_${f}.txt; head -4 $O/${TAG}_kernel_stats_${f}.csv; done
cat $O/${TAG}_exhaustive_clock.txt | tail -3
