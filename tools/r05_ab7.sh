#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_ab7; mkdir -p $O
N=ARTIS_AMD_SORT_NUMAJOR=1
bash tools/ab_env.sh "X=0" "$N" "ARTIS_AMD_SO=scratch/lib_nu32.so" "ARTIS_AMD_SO=scratch/lib_nu32.so $N" "ARTIS_AMD_SO=scratch/lib_nu64.so $N" "ARTIS_AMD_SO=scratch/lib_nu128.so $N" "ARTIS_AMD_SO=scratch/lib_nu64.so" "$N ARTIS_AMD_BUDGET_R=8" "ARTIS_AMD_SO=scratch/lib_nu64.so $N ARTIS_AMD_BUDGET_R=8" "X=0" 2>&1 | tee $O/ab.txt
