#!/bin/bash
# end-of-round check on the final code: the whole -m gpu suite, the default bench line, configs[3], and the host profiles of both
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
last_json() { python3 -c "import sys; print([l for l in open(sys.argv[1]) if l.startswith('{')][-1].strip())" "$1"; }
BIHOME_TEST_DDP_TIMEOUT=300 timeout -s KILL 1500 python3 -m pytest tests -m gpu -x -q > $O/r05g_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r05g_gpu_tests.log
timeout -s KILL 900 python3 bench.py > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/r05g_bench.json
timeout -s KILL 600 python3 bench.py --no-alt --no-cpu-baseline --config detone-bihome > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/r05g_bench_detone.json
timeout -s KILL 300 python3 bench.py --no-alt --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --host-profile 2>&1 | grep -v amdgpu.ids | head -50 > $O/r05g_host_profile.txt
timeout -s KILL 300 python3 bench.py --config detone-bihome --no-alt --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --host-profile 2>&1 | grep -v amdgpu.ids | head -50 > $O/r05g_host_profile_detone.txt
for f in $O/r05g_bench*.json; do python3 -c "import json,sys; j=json.load(open(sys.argv[1])); print(sys.argv[1], j['ms_per_step'], round(j['value'],1), j['roofline']['kernel'], round(j['roofline']['frac'],3), j.get('hbm_path_frac'), j['step_ms_percentiles'])" $f; done
head -2 $O/r05g_host_profile.txt $O/r05g_host_profile_detone.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
