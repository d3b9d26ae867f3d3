#!/bin/bash
cd $GRAFT_REPO_ROOT
for name in xv1 xv2 xv4; do
  ARTIS_AMD_SO_CI_CLASSIC_VPKT=$PWD/artis_amd/libartis_amd_$name.so python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --options ci_classic_vpkt --t-days 5 --packets 1000000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['kernel_breakdown_last_step']
print('$name: %.1f ms/step  %.1f M/s  thermal(+vpkt) %.1f ms  rpkt(+vpkt) %.1f ms' % (d['ms_per_step'], d['value']/1e6, b['thermal_ms'], b['rpkt_ms']))"
done
