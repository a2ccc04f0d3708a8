#!/bin/bash
# round 6, GPU call 2: the whole GPU suite with the measured maxima printed, the graph executor's knobs, the persistent kernel on constant data
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids > gpurun_out/r06b_gpu_tests.txt
grep -c PASSED gpurun_out/r06b_gpu_tests.txt; tail -5 gpurun_out/r06b_gpu_tests.txt; grep MEASURED gpurun_out/r06b_gpu_tests.txt | head -60
run() { "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step %.3f' % d['ms_per_step'], 'p50 %.3f' % d['step_ms_percentiles']['p50'])"; }
B="python3 bench.py --no-alt --steps 40 --warmup 8 --no-cpu-baseline --no-roofline"
{
echo "eager"; run $B
echo "eager one stream"; run $B --no-overlap
echo "graph"; run $B --graph
echo "graph, one stream"; run $B --graph --no-overlap
for q in 1 2 4 8; do echo "graph DEBUG_HIP_FORCE_GRAPH_QUEUES=$q"; DEBUG_HIP_FORCE_GRAPH_QUEUES=$q run $B --graph; done
for b in 1 16 256; do echo "graph DEBUG_HIP_GRAPH_BATCH_SIZE=$b"; DEBUG_HIP_GRAPH_BATCH_SIZE=$b run $B --graph; done
echo "graph DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run $B --graph
echo "eager again"; run $B
} 2>&1 | tee gpurun_out/r06b_graph_knobs.txt
for dta in randn zeros ones randn; do PC_DATA=$dta python tools/pc_emu_time.py; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06b_pc_data.txt
