#!/bin/bash
# tiled cell cache (by default a quarter of the headline's cache) under variants of the environment:
#   tools/tiles_ab.sh [budget MB] ["VAR=val VAR2=val" ...]   -> gpurun_out/tiles_ab_<budget>.txt (e.g. 13000 ARTIS_AMD_TILE_PARK_AT=524288 "ARTIS_AMD_MA_HOTFRAC=0.5 ARTIS_AMD_POOL_KEEP=0")
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
MB=${1:-13000}
shift
out=gpurun_out/tiles_ab_$MB.txt
mkdir -p gpurun_out
: > $out
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']['cell_cache']; b=d['kernel_breakdown_last_step']
print('%s: tiles %d x %d cells, %d B/cell, hot %s | %.1f ms/step | %s | %s' % ('$1', c['tiles'], c['cells_per_tile'], c['bytes_per_cell'], c.get('record_tiers', {}).get('hot_fraction', 1), d['ms_per_step'], {k:c[k] for k in c if k in ('sweeps','tile_fills','fill_ms','listed','sparse_fills','cells_filled','parked','pool_resets','pool_units_used','pool_units')}, {k: (round(v['ms'], 1), v['launches']) for k, v in d['kernel_ms_by_kind_last_step'].items()}))"
}
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | line untiled >> $out
for v in "" "$@"; do
  env ARTIS_AMD_CACHE_BUDGET_MB=$MB $v python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>gpurun_out/tiles_ab_err.log | line "budget $MB $v" >> $out
done
cat $out
