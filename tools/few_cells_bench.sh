#!/bin/bash
# ms/step of the 1e7-packet workload on models with few cells (DESIGN.md "Models with few cells"): gpurun_out/few_cells.txt
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/few_cells.txt; : > $O
run() { python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "${@:2}" 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['value']/1e6,1), 'M', round(j['ms_per_step'],1), 'ms', j['config']['nonempty_cells'], 'cells', {k:v['ms'] for k,v in j['kernel_ms_by_kind_last_step'].items() if v['ms']>0})" >> $O; }
run 6cubed --ncoord 6
run 12cubed --ncoord 12
run 20cubed --ncoord 20
run 1d_30 --grid 1d --ncoord 30
run 1d_100 --grid 1d --ncoord 100
run 2d_25x50 --grid 2d --ncoord 25
cat $O
