#!/bin/bash
# round 4: are radial shells the better tiles? A 1D spherical grid's tiles ARE shells (contiguous cell ranges = radial ranges): the same
# population untiled and forced into 4 / 2 tiles, one sweep direction and zigzag
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
run() { python3 bench.py --grid 1d --ncoord 100 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['kernel_breakdown_last_step']
print('$1: %.1f ms/step  thermal %.1f rpkt %.1f' % (d['ms_per_step'], b['thermal_ms'], b['rpkt_ms']), d['config']['cell_cache'])"; }
run untiled
ARTIS_AMD_CACHE_BUDGET_MB=18 run "4 tiles"
ARTIS_AMD_CACHE_BUDGET_MB=18 ARTIS_AMD_TILE_ZIGZAG=1 run "4 tiles zigzag"
ARTIS_AMD_CACHE_BUDGET_MB=35 run "2 tiles"
