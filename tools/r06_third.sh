#!/bin/bash
# round 6: the whole GPU suite on the round's kernels (LDS prefix fills, adaptive tiles, level table of up to 9856 levels in LDS), then tiled and big-data benches
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r06_third; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
grep -E "passed|failed" $O/tests.log | tail -2
tb() { python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']['cell_cache']
print('$1', round(d['ms_per_step'],1), {k:v['ms'] for k,v in d['kernel_ms_by_kind_last_step'].items() if v['ms']>0}, {k:c[k] for k in c if k in ('tiles','sweeps','tile_fills','fill_ms','listed','sparse_fills','cells_filled','parked')})"; }
ARTIS_AMD_CACHE_BUDGET_MB=13000 tb tiles4_adaptive > $O/tiling.txt
ARTIS_AMD_CACHE_BUDGET_MB=13000 ARTIS_AMD_TILE_ADAPT=0 tb tiles4_fixed >> $O/tiling.txt
ARTIS_AMD_CACHE_BUDGET_MB=26000 tb tiles2_adaptive >> $O/tiling.txt
ARTIS_AMD_CACHE_BUDGET_MB=26000 ARTIS_AMD_TILE_ADAPT=0 tb tiles2_fixed >> $O/tiling.txt
cat $O/tiling.txt
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset cd23like > $O/bench_cd23like.json 2>/dev/null; tail -c 420 $O/bench_cd23like.json
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset w7big > $O/bench_w7big.json 2>/dev/null; tail -c 420 $O/bench_w7big.json
