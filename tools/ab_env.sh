#!/bin/bash
# tools/ab_env.sh "<ENV=.. ENV=..>" ... : one bench run per argument (environment assignments), prints ms/step and the split
for cfg in "$@"; do
  env $cfg ARTIS_BENCH_VERBOSE=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline $AB_ARGS 2>/tmp/_err | python -c "
import json,sys,re
d=json.loads(sys.stdin.read()); b=d['kernel_breakdown_last_step']
err=open('/tmp/_err').read()
m=re.search(r\"'X_MA_JUMPS': (\d+)\", err); j=int(m.group(1)) if m else 0
m=re.search(r\"'X_56': (\d+)\", err); h=int(m.group(1)) if m else 0
k=d.get('kernel_ms_by_kind_last_step',{})
print('%-70s %.1f ms/step thermal %.1f ms rpkt %.1f ms launches %d slow %.1f tail %.1f bb %.1f' % ('$cfg', d['ms_per_step'], b['thermal_ms'], b['rpkt_ms'], b['thermal_launches'], k.get('k_slow',{}).get('ms',0), k.get('k_tail',{}).get('ms',0), k.get('k_blackbody',{}).get('ms',0)))"
done
