#!/bin/bash
# tiled cell cache with on-demand record tiers: tools/tiles_hot_ab.sh [budget MB] [hot fractions ...] -> gpurun_out/tiles_hot_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
MB=${1:-13000}
shift
HOTS=${@:-0.5 0.3 0.2 0.1}
out=gpurun_out/tiles_hot_ab.txt
mkdir -p gpurun_out
: > $out
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']['cell_cache']; b=d['kernel_breakdown_last_step']
print('%s: tiles %d x %d cells, %d B/cell | %.1f ms/step | thermal %.1f rpkt %.1f tail %.1f slow %.1f ms | %s |' % ('$1', c['tiles'], c['cells_per_tile'], c['bytes_per_cell'], d['ms_per_step'], b['thermal_ms'], b['rpkt_ms'], b.get('tail_ms', 0), b.get('slow_ms', 0), {k:c[k] for k in c if k in ('sweeps','tile_fills','fill_ms','listed','sparse_fills','cells_filled','parked','pool_resets','pool_units_used','pool_units')}), {k: (round(v['ms'], 1), v['launches']) for k, v in d['kernel_ms_by_kind_last_step'].items()})"
}
for h in $HOTS; do
  env ARTIS_AMD_CACHE_BUDGET_MB=$MB ARTIS_AMD_MA_HOTFRAC=$h $EXTRA_ENV python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>gpurun_out/tiles_hot_err.log | line "budget $MB hot $h" >> $out
done
cat $out
