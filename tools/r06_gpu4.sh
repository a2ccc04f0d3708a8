#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_head_kernels_gpu.py tests/test_determinism_gpu.py tests/test_model_gpu.py tests/test_branches_gpu.py tests/test_rgb_gpu.py tests/test_fullsize_gpu.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids > gpurun_out/r06d_gpu_tests.txt
tail -4 gpurun_out/r06d_gpu_tests.txt; grep "MEASURED fused\|FAILED\|^E  " gpurun_out/r06d_gpu_tests.txt | head -30
python tools/stem_warp_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d_stem_warp_ab.txt
BIHOME_TUNING=1 python tools/hbm_path_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d_hbm_path_bench.txt
