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
# (5) alg_2opt_tabu with a list (tools/tabu_time.py: full descents of rand10000 with 0 / 200 / 2000 live stamps), kernel times
#     and HBM fetch traffic: cluster = the default (CLUSTER engine from the list of non-zero stamps), grid = TSP_ENGINE=1 (GRID
#     engine from the list, k_sweep<TABU>), dense = TSP_TABU_DENSE=1 (four stamp reads per pair, k_step<TABU>)
for V in cluster grid dense; do
  unset TSP_TABU_DENSE TSP_ENGINE
  [ $V = grid ] && export TSP_ENGINE=1
  [ $V = dense ] && export TSP_TABU_DENSE=1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tabu_$V -- python3 $R/tools/tabu_time.py rand10000 > $O/${TAG}_tabu_time_$V.txt 2> $O/trace_tabu_$V.log
  python3 $R/tools/summarize_rocprof.py stats $O/prof_tabu_$V $O/${TAG}_kernel_stats_tabu_$V.csv > /dev/null
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_tabuf_$V -- python3 $R/tools/tabu_time.py rand10000 > /dev/null 2> $O/pmc_tabuf_$V.log
  python3 $R/tools/summarize_rocprof.py pmc $O/prof_tabuf_$V $O/${TAG}_pmc_FETCH_SIZE_tabu_$V.json > /dev/null
  rm -rf $O/prof_tabu_$V $O/prof_tabuf_$V
done
unset TSP_TABU_DENSE TSP_ENGINE
rm -rf $O/prof_trace $O/prof_trace2 $O/prof_FETCH_SIZE $O/prof_WRITE_SIZE $O/prof_sq
head -5 $O/${TAG}_kernel_stats.csv; head -6 $O/${TAG}_kernel_stats_exhaustive.csv; tail -2 $O/trace.log
