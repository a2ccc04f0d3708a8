#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu.ids > gpurun_out/r06c_gpu_tests.txt
tail -8 gpurun_out/r06c_gpu_tests.txt; grep "MEASURED fused\|FAILED\|Error" gpurun_out/r06c_gpu_tests.txt | head -30
python tools/stem_warp_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c_stem_warp_ab.txt
tools/ab_env.sh BIHOME_WARP_IN_STEM_DGRAD=0 3 2>&1 | tee gpurun_out/r06c_step_ab_warp_fold.txt
