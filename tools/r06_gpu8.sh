#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_f16x2_gpu.py -m gpu -q -s -k "batched or kernel" 2>&1 | grep -v amdgpu.ids | grep "batched wgrad\|passed\|failed\|^E " | tail -30
for r in 1 2 3; do for b in 1 2 4; do
  echo -n "BIHOME_WGRAD_BATCH=$b "; BIHOME_WGRAD_BATCH=$b python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-alt 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step %.3f' % d['ms_per_step'], 'p50 %.3f' % d['step_ms_percentiles']['p50'])"
done; done 2>&1 | tee gpurun_out/r06i_step_ab_wgrad_batch.txt
python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -5
