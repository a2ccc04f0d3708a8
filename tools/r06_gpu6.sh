#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_f16x2_gpu.py tests/test_head_kernels_gpu.py tests/test_conv_pc_gpu.py -m gpu -q -s -x 2>&1 | grep -v amdgpu.ids > gpurun_out/r06g_new_tests.txt
tail -5 gpurun_out/r06g_new_tests.txt; grep "BatchNorm adjoint on load\|^E  " gpurun_out/r06g_new_tests.txt | head -20
tools/ab_env.sh BIHOME_WGRAD_BNADJ=0 3 2>&1 | tee gpurun_out/r06g_step_ab_bnadj.txt
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r06g_tests_tail.txt
