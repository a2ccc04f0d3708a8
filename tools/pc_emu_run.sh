#!/bin/bash
# GPU box: the PC_EMU variants next to each other (two alternating rounds), then the barrier time line of each
mkdir -p gpurun_out
for rnd in 1 2; do for v in 0 1 3 2; do BIHOME_LIB_VARIANT=emu$v python tools/pc_emu_time.py; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pc_emu_time.txt
for v in 0 1 3 2; do echo "== emu$v"; BIHOME_LIB_VARIANT=emu$v python tools/pc_timeline.py 128,32,64,64 fwd | tail -12; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pc_emu_timeline.txt
# in the step (timing only: the variants' results are wrong)
for rnd in 1 2; do for v in 0 1; do
  BIHOME_LIB_VARIANT=emu$v python3 bench.py --no-alt --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('emu$v', 'ms_per_step %.3f' % d['ms_per_step'], 'p50 %.3f' % d['step_ms_percentiles']['p50'])"
done; done 2>&1 | tee gpurun_out/pc_emu_step.txt
