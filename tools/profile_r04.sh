#!/bin/bash
# Run on the GPU box (through gpurun): the rocprofv3 evidence of round 4.  Summaries land in gpurun_out/<tag>_*; copy what is to
# be judged into profiles/.   usage: profile_r04.sh <tag>
#  (1) bench.py's timed region under --kernel-trace --stats: k_exh + k_move_pos (the exhaustive descent), 3 steps
#  (2) HBM traffic of the same kernels: FETCH_SIZE and WRITE_SIZE in a pass of their own each
#  (3) SQ counters of the same kernels
#  (4) tabu() through the C host in chains of iterations, (5) HEU_VNS: kernel traces; (6) tabu()'s paths against each other
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
O=$R/gpurun_out
mkdir -p $O
make -C $R/oracle >/dev/null
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- python3 $B > $O/${TAG}_bench_under_rocprof.json 2> $O/trace.log
python3 $R/tools/summarize_rocprof.py stats $O/prof_trace $O/${TAG}_kernel_stats.csv > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/prof_$C -- python3 $B > /dev/null 2> $O/pmc_$C.log
  python3 $R/tools/summarize_rocprof.py pmc $O/prof_$C $O/${TAG}_pmc_$C.json > /dev/null
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/prof_sq -- python3 $B > /dev/null 2> $O/pmc_sq.log
python3 $R/tools/summarize_rocprof.py pmc $O/prof_sq $O/${TAG}_pmc_sq_wave_counters.json > /dev/null
# (4) tabu() through the C host, 2000 iterations of rand10000: iterations inside the launch (the default), then queued launches
for mode in 1 0; do
  TSP_TABU_INKERNEL=$mode TSP_HOST_STATS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tabu -- python3 $R/tools/tabu_one.py rand10000 2000 > $O/${TAG}_tabu_inkernel$mode.txt 2> $O/trace_tabu$mode.log
  grep "\[tabu\]" $O/trace_tabu$mode.log | tail -1 >> $O/${TAG}_tabu_inkernel$mode.txt
  python3 $R/tools/summarize_rocprof.py stats $O/prof_tabu $O/${TAG}_kernel_stats_tabu_inkernel$mode.csv > /dev/null
  rm -rf $O/prof_tabu
done
# (5) HEU_VNS through the C host, 300 rounds of rand10000
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vns -- python3 $R/tools/vns_one.py rand10000 300 > $O/${TAG}_vns.txt 2> $O/trace_vns.log
python3 $R/tools/summarize_rocprof.py stats $O/prof_vns $O/${TAG}_vns_kernel_stats.csv > /dev/null
# (6) every path of tabu() gives the same incumbent; the tail of an iteration and the phases of a tabu sweep (diagnostic build)
bash $R/tools/tabu_paths.sh rand10000 2000 > $O/${TAG}_tabu_paths.txt 2>&1
bash $R/tools/tabu_paths.sh pr1002 3000 >> $O/${TAG}_tabu_paths.txt 2>&1
[ -f $R/tsp_optimization_amd/lib_diag/libtsp_host.so ] && timeout 300 python3 $R/tools/diag_tabu_tail.py rand10000 1000 > $O/${TAG}_diag_tabu_tail.txt 2>&1
rm -rf $O/prof_trace $O/prof_FETCH_SIZE $O/prof_WRITE_SIZE $O/prof_sq $O/prof_vns
head -5 $O/${TAG}_kernel_stats.csv; cat $O/${TAG}_tabu_inkernel1.txt; head -6 $O/${TAG}_kernel_stats_tabu_inkernel1.csv; cat $O/${TAG}_vns.txt; head -5 $O/${TAG}_vns_kernel_stats.csv; cat $O/${TAG}_tabu_paths.txt
python3 - <<PY
import json
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = json.load(open("$O/${TAG}_pmc_%s.json" % c))
    for k, v in d.items():
        if "k_exh" in k or "k_move_pos" in k: print(c, k[:40], v)
PY
