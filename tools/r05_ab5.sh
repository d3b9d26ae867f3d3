#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_ab5; mkdir -p $O
ARTIS_AMD_TQ_LOW=32 ARTIS_AMD_SO=scratch/libprof.so python3 tools/r05_refill_prof.py 2>&1 | tail -4 | tee $O/prof.txt
A=ARTIS_AMD_REFILL=1
bash tools/ab_env.sh "ARTIS_AMD_REFILL=0" \
 "ARTIS_AMD_SO=scratch/lib_tb1024_80.so $A ARTIS_AMD_TQ_LOW=16" "ARTIS_AMD_SO=scratch/lib_tb1024_80.so $A ARTIS_AMD_TQ_LOW=32" \
 "ARTIS_AMD_SO=scratch/lib_tb1024.so $A ARTIS_AMD_TQ_LOW=32" "ARTIS_AMD_SO=scratch/lib_tb1024.so $A ARTIS_AMD_TQ_LOW=40" \
 "ARTIS_AMD_REFILL=0" 2>&1 | tee $O/ab.txt
