#!/bin/bash
# the cluster / scale / stress / tabu-list / sorted-sweep parity tests and the two headline descents of bench.py
python -m pytest tests/test_gpu_cluster.py tests/test_gpu_scale.py tests/test_gpu_stress.py tests/test_gpu_tabu_list.py tests/test_gpu_sorted_sweep.py -m gpu -x -q --timeout 300 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 > /tmp/qc.json 2>/tmp/qc.err || tail -5 /tmp/qc.err
python - <<'PY'
import json
d = json.loads(open('/tmp/qc.json').read().strip().splitlines()[-1])
r = d['roofline']; f = r['first']
print('ms_per_step', d['ms_per_step'], d['config'].get('all_checks_ok', d.get('all_checks_ok')), f['device_ms'], r.get('counters'))
PY
