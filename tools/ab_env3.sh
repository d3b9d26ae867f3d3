#!/bin/bash
# tools/ab_env3.sh [--options X] "<ENV=.. ENV=..>" ... : one bench run (3 timed steps) per argument, ms/step and the split
cd ${GRAFT_REPO_ROOT:-.}
OPT=""
if [ "$1" = "--options" ]; then OPT="--options $2"; shift 2; fi
for cfg in "$@"; do
  env $cfg python3 bench.py $OPT --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['kernel_breakdown_last_step']
print('%-60s %.1f ms/step thermal %.1f (%d) rpkt %.1f (%d)' % ('$cfg', d['ms_per_step'], b['thermal_ms'], b['thermal_launches'], b['rpkt_ms'], b['rpkt_launches']))"
done
