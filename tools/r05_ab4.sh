#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_ab4; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "walker_refill" > $O/test.log 2>&1; tail -2 $O/test.log
A=ARTIS_AMD_REFILL=1
bash tools/ab_env.sh "ARTIS_AMD_REFILL=0" \
 "ARTIS_AMD_SO=scratch/lib_tb1024_128.so $A ARTIS_AMD_TQ_LOW=24" "ARTIS_AMD_SO=scratch/lib_tb1024_128.so $A ARTIS_AMD_TQ_LOW=32" \
 "ARTIS_AMD_SO=scratch/lib_tb1024_128.so $A ARTIS_AMD_TQ_LOW=40" "ARTIS_AMD_SO=scratch/lib_tb1024_128.so $A ARTIS_AMD_TQ_LOW=48" \
 "ARTIS_AMD_SO=scratch/lib_tb1024.so $A ARTIS_AMD_TQ_LOW=24" "ARTIS_AMD_SO=scratch/lib_tb1024.so $A ARTIS_AMD_TQ_LOW=32" \
 "ARTIS_AMD_SO=scratch/lib_tb768_160.so $A ARTIS_AMD_TQ_LOW=32" "ARTIS_AMD_SO=scratch/lib_tb768_160.so $A ARTIS_AMD_TQ_LOW=48" \
 "ARTIS_AMD_REFILL=0" 2>&1 | tee $O/ab.txt
