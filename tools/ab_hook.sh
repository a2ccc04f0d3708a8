#!/bin/bash
# same-box A/B of the step time with and without a tuning hook (bench.py --hook A,B; tuning build), alternating runs:
#   tools/ab_hook.sh "-33,0" [rounds] [bench flags]
H=$1; R=${2:-3}; shift; shift
cd "$GRAFT_REPO_ROOT" || exit 1
for i in $(seq $R); do
  for v in base hook; do
    if [ "$v" = base ]; then E=""; else E="--hook=$H"; fi
    BIHOME_TUNING=1 python3 bench.py --no-alt --steps 40 --warmup 8 --no-cpu-baseline --no-roofline $E "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', 'ms_per_step %.3f' % d['ms_per_step'], 'p50 %.3f' % d['step_ms_percentiles']['p50'])"
  done
done
