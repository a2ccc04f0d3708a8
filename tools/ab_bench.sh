#!/bin/bash
# same-box A/B of the step time: default library against libbihome_hip_ab.so (built from another tree with `make ab`), alternating runs.
# tools/ab_bench.sh [rounds] [extra bench.py flags]
R=${1:-3}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
for i in $(seq $R); do
  for v in default ab; do
    if [ $v = ab ]; then export BIHOME_LIB_VARIANT=ab; else unset BIHOME_LIB_VARIANT; fi
    python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', 'ms_per_step %.3f' % d['ms_per_step'], 'p50 %.3f' % d['step_ms_percentiles']['p50'])"
  done
done
