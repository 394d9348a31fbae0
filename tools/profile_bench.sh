#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + two PMC passes of bench.py; summaries land in gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
mkdir -p $R/gpurun_out
make -C $R/oracle >/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/bench.py --no-cpu-baseline --no-variants > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/trace.log
python3 $R/tools/summarize_rocprof.py stats $R/gpurun_out/prof_trace $R/gpurun_out/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/fetch.log
python3 $R/tools/summarize_rocprof.py pmc $R/gpurun_out/prof_fetch $R/gpurun_out/${TAG}_pmc_fetch.json
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/write.log
python3 $R/tools/summarize_rocprof.py pmc $R/gpurun_out/prof_write $R/gpurun_out/${TAG}_pmc_write.json
rm -rf $R/gpurun_out/prof_trace $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
tail -3 $R/gpurun_out/trace.log
