#!/bin/bash
# Trimmed round profile: GPU tests, default bench, per-layer table, kernel trace.  tools/r04_baseline.sh <tag> [notests]
T=${1:-r04a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
last_json() { python3 -c "import sys; print([l for l in open(sys.argv[1]) if l.startswith('{')][-1].strip())" "$1"; }
if [ "$2" != "notests" ]; then
  timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.txt 2>&1; tail -5 $O/${T}_gpu_tests.txt
fi
python3 bench.py > $O/${T}_bench.log 2>&1; last_json $O/${T}_bench.log > $O/${T}_bench.json
python3 tools/step_detail.py zeng-bihome 64 > $O/${T}_step_detail.txt 2>&1
python3 bench.py --no-cpu-baseline --no-overlap > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_no_overlap.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_kt -o ${T} -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap > $O/${T}_kt.log 2>&1
cp $(find $O/${T}_kt -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv
rm -rf $O/${T}_kt
cat $O/${T}_bench.json
