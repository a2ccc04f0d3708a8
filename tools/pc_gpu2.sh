#!/bin/bash
mkdir -p gpurun_out
timeout -s KILL 400 python tools/pc_smoke.py > gpurun_out/pc_smoke.log 2>&1
rc=$?; echo "smoke rc=$rc"; tail -3 gpurun_out/pc_smoke.log
if [ $rc -ne 0 ] || ! grep -q SMOKE_DONE gpurun_out/pc_smoke.log; then exit 1; fi
timeout -s KILL 1200 python -m pytest tests/test_conv_pc_gpu.py -q > gpurun_out/pc_test.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/pc_test.log
timeout -s KILL 900 python tests/test_conv_pc_gpu.py 128,32,64,64 128,16,128,128 128,8,256,256 > gpurun_out/pc_bench.log 2>&1
echo "bench rc=$?"; tail -14 gpurun_out/pc_bench.log
BIHOME_TUNING=1 timeout -s KILL 900 python tools/pc_ablate.py > gpurun_out/pc_ablate.log 2>&1
echo "ablate rc=$?"; tail -10 gpurun_out/pc_ablate.log
for m in fwd bnr; do BIHOME_TUNING=1 timeout -s KILL 300 python tools/pc_timeline.py 128,32,64,64 $m > gpurun_out/pc_timeline_A_$m.log 2>&1; tail -90 gpurun_out/pc_timeline_A_$m.log; done
BIHOME_TUNING=1 timeout -s KILL 300 python tools/pc_timeline.py 128,8,256,256 fwd > gpurun_out/pc_timeline_C_fwd.log 2>&1; tail -40 gpurun_out/pc_timeline_C_fwd.log
