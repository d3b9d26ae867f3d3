#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r06_sweep2.txt; : > $O
run() { python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],1), {k:v['ms'] for k,v in j['kernel_ms_by_kind_last_step'].items() if v['ms']>0})" >> $O; }
run default
for v in maxilp memclause o2 nounroll; do ARTIS_AMD_SO=$PWD/scratch/ab/libartis_amd_$v.so run $v; done
run default_again
cat $O
