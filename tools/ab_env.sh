#!/bin/bash
# same-box A/B of the step time with and without an environment setting, alternating runs: tools/ab_env.sh VAR=VALUE [rounds] [bench flags]
KV=$1; R=${2:-3}; shift; shift
cd "$GRAFT_REPO_ROOT" || exit 1
for i in $(seq $R); do
  for v in base "$KV"; do
    if [ "$v" = base ]; then E=""; else E="$KV"; fi
    env $E python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', 'ms_per_step %.3f' % d['ms_per_step'], 'p50 %.3f' % d['step_ms_percentiles']['p50'])"
  done
done
