#!/bin/bash
# GPU run of the persistent 3x3 kernel: smoke under a hard timeout, then the bit-identity tests, then the A/B timings and the ablation
mkdir -p gpurun_out
timeout -s KILL 400 python tools/pc_smoke.py > gpurun_out/pc_smoke.log 2>&1
rc=$?; echo "smoke rc=$rc"; tail -4 gpurun_out/pc_smoke.log
if [ $rc -ne 0 ] || ! grep -q SMOKE_DONE gpurun_out/pc_smoke.log; then exit 1; fi
timeout -s KILL 1200 python -m pytest tests/test_conv_pc_gpu.py -q > gpurun_out/pc_test.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/pc_test.log
timeout -s KILL 900 python tests/test_conv_pc_gpu.py > gpurun_out/pc_bench.log 2>&1
echo "bench rc=$?"; tail -30 gpurun_out/pc_bench.log
BIHOME_TUNING=1 timeout -s KILL 900 python tools/pc_ablate.py > gpurun_out/pc_ablate.log 2>&1
echo "ablate rc=$?"; tail -12 gpurun_out/pc_ablate.log
