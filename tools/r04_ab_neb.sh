#!/bin/bash
# A/B of nltenebular engine builds, interleaved: tools/r04_ab_neb.sh <rounds> <name> ... ("base" = the preset's own library)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
R=$1; shift
for r in $(seq 1 $R); do
  for name in "$@"; do
    so=$PWD/artis_amd/libartis_amd_$name.so
    [ "$name" = base ] && so=$PWD/artis_amd/libartis_amd_nltenebular.so
    ARTIS_AMD_SO_NLTENEBULAR=$so python3 bench.py --options nltenebular --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['kernel_breakdown_last_step']
print('$name round $r: %.1f ms/step  %.1f M/s  thermal %.1f ms  rpkt %.1f ms' % (d['ms_per_step'], d['value']/1e6, b['thermal_ms'], b['rpkt_ms']))"
  done
done
