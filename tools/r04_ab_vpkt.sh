#!/bin/bash
# A/B of ci_classic_vpkt engine builds on the virtual-packet bench: tools/r04_ab_vpkt.sh <name> ... ("base" = the preset's own library)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for name in "$@"; do
  so=$PWD/artis_amd/libartis_amd_$name.so
  [ "$name" = base ] && so=$PWD/artis_amd/libartis_amd_ci_classic_vpkt.so
  ARTIS_AMD_SO_CI_CLASSIC_VPKT=$so python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --options ci_classic_vpkt --t-days 5 --packets 1000000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['kernel_breakdown_last_step']
print('$name: %.1f ms/step  %.1f M/s  thermal(+vpkt) %.1f ms  rpkt(+vpkt) %.1f ms' % (d['ms_per_step'], d['value']/1e6, b['thermal_ms'], b['rpkt_ms']))"
done
