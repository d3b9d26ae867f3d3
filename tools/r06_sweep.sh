#!/bin/bash
# round 6 sweep: phase length of the thermal kernel (builds under scratch/ab), launch budgets, the refill kernel -- after the loop's instruction diet
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r06_sweep.txt; : > $O
run() { python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],1), {k:v['ms'] for k,v in j['kernel_ms_by_kind_last_step'].items() if v['ms']>0}, j['kernel_ms_by_kind_last_step']['k_rpkt']['launches'])" >> $O; }
run default
for v in phase24 phase40 phase48 phase64; do ARTIS_AMD_SO=$PWD/scratch/ab/libartis_amd_$v.so run $v; done
ARTIS_AMD_REFILL=1 run refill
ARTIS_AMD_BUDGET_R=3 run budget_r3
ARTIS_AMD_BUDGET_R=5 run budget_r5
ARTIS_AMD_BUDGET_T=1024 run budget_t1024
ARTIS_AMD_BUDGET_T=4096 run budget_t4096
ARTIS_AMD_DRAIN_T=24 run drain_t24
ARTIS_AMD_DRAIN_T=96 run drain_t96
ARTIS_AMD_TAIL=8192 run tail8192
ARTIS_AMD_TAIL=2048 run tail2048
run default_again
cat $O
