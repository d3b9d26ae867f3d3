#!/bin/bash
# round 4: tiled runs with and without parked visit tails (ARTIS_AMD_TILE_PARK), headline data forced into 4 tiles
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for sm in 512 4096 16384; do for v in 1 0; do
  ARTIS_AMD_SPARSE_MAX=$sm ARTIS_AMD_TILE_PARK=$v ARTIS_AMD_CACHE_BUDGET_MB=11600 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['kernel_breakdown_last_step']
print('sparse_max $sm park $v: %.1f ms/step  thermal %.1f rpkt %.1f' % (d['ms_per_step'], b['thermal_ms'], b['rpkt_ms']), d['config']['cell_cache'])"
done; done
