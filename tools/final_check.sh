#!/bin/bash
# Run on the GPU box (through gpurun): what the driver runs at round end, plus the micro-benchmarks' outputs.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for u in xchg dist2 dist3; do [ -x tools/ubench/$u ] && ./tools/ubench/$u > $O/ubench_${u}.txt 2>&1; done
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $O/full_gpu_tests.log 2>&1
cat $O/full_gpu_tests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > $O/smoke.log 2>&1
cat $O/smoke.log
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err
tail -2 $O/bench_final.err
python - <<PY
import json
d=json.loads(open("$O/bench_final.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","all_checks_ok"): print(k, d.get(k))
print("roofline", d["roofline"]["frac"], d["roofline"]["launch_ms"], d["roofline"]["survey_8d_model"]["frac"], d["roofline"]["traffic"])
t=d["alg_2opt_tabu_with_a_list"]["tabu_iterations_on_resident_state"]; print("tabu", t["iterations_per_s"], t["queued_launches"]["iterations_per_s"], t["one_iteration_per_wait"]["iterations_per_s"], t["same_incumbent_all_ways"])
print("c5", d["other_configs"]["config5_rand5000_population128_2opt"]["wall_s"], "cpu", d["cpu_baseline"]["value"], "ttlo", d["time_to_local_optimum"]["best_improvement_alg_2opt_tabu"]["device_ms"], d["time_to_local_optimum"]["first_improvement_alg_2opt"]["device_ms"])
print("dm", d["distance_matrix_build"]["int32"]["frac"], d["distance_matrix_build"]["f64"]["frac"])
PY
