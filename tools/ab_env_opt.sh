#!/bin/bash
# tools/ab_env_opt.sh <options preset> "<ENV=.. ENV=..>" ... : one bench run per argument on another options build
O=$1; shift
for cfg in "$@"; do
  env $cfg python bench.py --options $O --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['kernel_breakdown_last_step']
print('%-60s %.1f ms/step %.1f M/s thermal %.1f (%d) rpkt+dense %.1f (%d)' % ('$cfg', d['ms_per_step'], d['value']/1e6, b['thermal_ms'], b['thermal_launches'], b['rpkt_ms'], b['rpkt_launches']))"
done
