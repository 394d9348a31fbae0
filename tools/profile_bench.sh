#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + PMC passes of bench.py; summaries land in gpurun_out/.
# usage: profile_bench.sh <tag>      (every pass runs under its own timeout)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
O=$R/gpurun_out
mkdir -p $O
make -C $R/oracle >/dev/null
cd /tmp && export TMPDIR=/tmp
# (1) the timed region alone: k_cluster_two_opt, one launch per descent
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-extras > $O/${TAG}_bench_under_rocprof.json 2> $O/trace.log
python3 $R/tools/summarize_rocprof.py stats $O/prof_trace $O/${TAG}_kernel_stats.csv > /dev/null
# (2) the exhaustive sweep (every delta expression executed) next to it
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_bench_under_rocprof_exhaustive.json 2> $O/trace2.log
python3 $R/tools/summarize_rocprof.py stats $O/prof_trace2 $O/${TAG}_kernel_stats_exhaustive.csv > /dev/null
# (3) HBM traffic counters, one pass each (FETCH_SIZE and WRITE_SIZE do not fit in one)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/prof_$C -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$C.log
  python3 $R/tools/summarize_rocprof.py pmc $O/prof_$C $O/${TAG}_pmc_$C.json > /dev/null
done
# (4) SQ counters of the cluster kernel
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/prof_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-variants > /dev/null 2> $O/pmc_sq.log
python3 $R/tools/summarize_rocprof.py pmc $O/prof_sq $O/${TAG}_pmc_sq_wave_counters.json > /dev/null
# (5) alg_2opt_tabu with a list: the per-pair scan (TSP_TABU_DENSE=1, k_step<TABU>) and the list path (k_sweep<TABU>), kernel
#     times and HBM fetch traffic of both (tools/tabu_time.py: full descents of rand10000 with 0 / 200 / 2000 live stamps)
for D in 0 1; do
  export TSP_TABU_DENSE=$D
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tabu$D -- python3 $R/tools/tabu_time.py rand10000 > $O/${TAG}_tabu_time_dense$D.txt 2> $O/trace_tabu$D.log
  python3 $R/tools/summarize_rocprof.py stats $O/prof_tabu$D $O/${TAG}_kernel_stats_tabu_dense$D.csv > /dev/null
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_tabuf$D -- python3 $R/tools/tabu_time.py rand10000 > /dev/null 2> $O/pmc_tabuf$D.log
  python3 $R/tools/summarize_rocprof.py pmc $O/prof_tabuf$D $O/${TAG}_pmc_FETCH_SIZE_tabu_dense$D.json > /dev/null
  rm -rf $O/prof_tabu$D $O/prof_tabuf$D
done
unset TSP_TABU_DENSE
rm -rf $O/prof_trace $O/prof_trace2 $O/prof_FETCH_SIZE $O/prof_WRITE_SIZE $O/prof_sq
head -5 $O/${TAG}_kernel_stats.csv; head -6 $O/${TAG}_kernel_stats_exhaustive.csv; tail -2 $O/trace.log
