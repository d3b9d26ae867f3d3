#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r06_pool.txt; : > $O
tb() { python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset cd23like 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']['cell_cache']
print('$1', round(d['ms_per_step'],1), {k:v['ms'] for k,v in d['kernel_ms_by_kind_last_step'].items() if v['ms']>0}, c)" >> $O; }
ARTIS_AMD_MA_POOLFRAC=0.15 tb pool0.15
ARTIS_AMD_MA_POOLFRAC=0.10 tb pool0.10
cat $O
