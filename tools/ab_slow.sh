#!/bin/bash
# A/B of the slow-path kernel at 2 waves per SIMD for the builds whose k_slow takes > 168 registers (ARTIS_SLOW_EU=2; libraries under scratch/ab)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/ab_slow.txt; : > $O
run() { python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --options $2 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['value']/1e6,1), round(j['ms_per_step'],1), {k:v['ms'] for k,v in j['kernel_ms_by_kind_last_step'].items() if v['ms']>0})" >> $O; }
run neb_default nltenebular
ARTIS_AMD_SO_NLTENEBULAR=$PWD/scratch/ab/libartis_amd_nltenebular_sloweu2.so run neb_sloweu2 nltenebular
run kl_default kilonova_lte
ARTIS_AMD_SO_KILONOVA_LTE=$PWD/scratch/ab/libartis_amd_kilonova_lte_sloweu2.so run kl_sloweu2 kilonova_lte
ARTIS_AMD_TAIL=8192 run neb_tail8192 nltenebular
ARTIS_AMD_TAIL=32768 run neb_tail32768 nltenebular
ARTIS_AMD_BUDGET_R=5 run neb_budget_r5 nltenebular
cat $O
