#!/bin/bash
# neighbouring launch parameters of the thermal kernel after the round's changes (defaults: BUDGET_T 2048, DRAIN_T 48, DRAIN_MIN 1000000)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_sweep; mkdir -p $O
bash tools/ab_env.sh "X=1" "ARTIS_AMD_BUDGET_T=1024" "ARTIS_AMD_BUDGET_T=1536" "ARTIS_AMD_BUDGET_T=3072" "ARTIS_AMD_BUDGET_T=4096" "ARTIS_AMD_DRAIN_T=24" "ARTIS_AMD_DRAIN_T=96" "ARTIS_AMD_DRAIN_T=192" "ARTIS_AMD_DRAIN_MIN=300000" "ARTIS_AMD_DRAIN_MIN=3000000" "ARTIS_AMD_BUDGET_T_SMALL=512" "ARTIS_AMD_BUDGET_T_SMALL=256 ARTIS_AMD_SMALL_LIST=100000" "X=1" 2>&1 | tee $O/ab.txt
