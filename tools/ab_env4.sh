#!/bin/bash
# A/B of one environment switch of the engine on the headline bench, interleaved: tools/ab_env4.sh VAR "v1 v2 ..." [rounds] [bench args]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
V=$1; VALS=$2; R=${3:-2}; shift 3
for r in $(seq 1 $R); do
  for v in $VALS; do
    env $V=$v python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['kernel_breakdown_last_step']
print('$V=$v round $r: %.1f ms/step  %.1f M/s  thermal %.1f ms  rpkt %.1f ms' % (d['ms_per_step'], d['value']/1e6, b['thermal_ms'], b['rpkt_ms']))"
  done
done
