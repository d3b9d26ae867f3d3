#!/bin/bash
# ms/step of the headline workload against the number of cell-cache tiles: tools/tiling_bench.sh [extra bench args]
# (ARTIS_AMD_CACHE_BUDGET_MB forces the tiling; the cache is 1.55 MB per cell x 65 752 cells = 102 GB with the w7 data)
for mb in 200000 54000 27000 13500; do
  ARTIS_AMD_CACHE_BUDGET_MB=$mb python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']['cell_cache']; b=d['kernel_breakdown_last_step']
print('budget %6d MB: tiles %d cells/tile %d | %.1f ms/step %.1f M packet-steps/s | thermal %.1f rpkt %.1f ms | %s' % ($mb, c['tiles'], c['cells_per_tile'], d['ms_per_step'], d['value']/1e6, b['thermal_ms'], b['rpkt_ms'], {k:c[k] for k in c if k in ('sweeps','tile_fills','fill_ms','listed','sparse_fills','cells_filled')}))"
done
