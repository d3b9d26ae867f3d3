#!/bin/bash
# tools/refill_stats.sh "<ENV=.. ENV=..>" ... : one bench run per argument; prints ms/step, the kernel split and the lane
# counters of k_thermal_q (stats slots 44..47: service slots, walking lanes, wave-rounds, service passes)
for cfg in "$@"; do
  env $cfg ARTIS_BENCH_VERBOSE=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/tmp/_err | python -c "
import json,sys,re
d=json.loads(sys.stdin.read()); b=d['kernel_breakdown_last_step']
err=open('/tmp/_err').read()
def g(k):
    m=re.search(r\"'%s': (\d+)\" % k, err); return int(m.group(1)) if m else 0
j=g('X_MA_JUMPS'); rounds=g('X_46'); lanes=g('X_45'); sp=g('X_47'); sl=g('X_44')
print('%-60s %.1f ms/step thermal %.1f rpkt %.1f | rounds %.3e lanes/round %.1f | service passes %.3e slots/pass %.1f | jumps %.3e' % ('$cfg', d['ms_per_step'], b['thermal_ms'], b['rpkt_ms'], rounds, lanes/max(rounds,1), sp, sl/max(sp,1), j))"
done
