#!/bin/bash
# full GPU check: the whole -m gpu suite, then the default bench line
mkdir -p gpurun_out
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/full_gpu_tests.log
timeout -s KILL 900 python bench.py > gpurun_out/bench_cur.json 2> gpurun_out/bench_cur.err
echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/bench_cur.json') if l.startswith('{')][-1])
print(j['ms_per_step'], j['value'], j['roofline']['kernel'], j['roofline']['frac'], j.get('hbm_path_frac'), (j.get('alt_arithmetic') or {}).get('ms_per_step'))
for k in j['kernel_breakdown'][:14]: print(k)
PY
