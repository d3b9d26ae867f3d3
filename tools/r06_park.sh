#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r06_park.txt; : > $O
tb() { python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']['cell_cache']
print('$1', round(d['ms_per_step'],1), {k:v['ms'] for k,v in d['kernel_ms_by_kind_last_step'].items() if v['ms']>0}, {k:c[k] for k in c if k in ('tiles','sweeps','tile_fills','fill_ms','listed','sparse_fills','cells_filled','parked')})" >> $O; }
export ARTIS_AMD_CACHE_BUDGET_MB=13000
for pa in 32768 131072 524288 2097152; do ARTIS_AMD_TILE_PARK_AT=$pa tb park_at_$pa; done
ARTIS_AMD_TILE_PARK_AT=524288 ARTIS_AMD_SPARSE_MAX=65536 tb park_524288_sparse65536
cat $O
