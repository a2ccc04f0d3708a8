#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rm -rf $O/gaps_kt
rocprofv3 --kernel-trace --output-format csv -d $O/gaps_kt -o g -- python3 bench.py --no-alt --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/gaps_kt.log 2>&1
python3 tools/trace_gaps.py $(find $O/gaps_kt -name "*kernel_trace.csv" | head -1) > $O/trace_gaps.txt 2>&1
cat $O/trace_gaps.txt
grep -o '"ms_per_step": [0-9.]*' $O/gaps_kt.log | head -1
rm -rf $O/gaps_kt
