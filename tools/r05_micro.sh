#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_micro; mkdir -p $O
timeout 300 ./scratch/chase 2>&1 | tee $O/chase.txt
bash tools/final_check.sh 2>&1 | tee $O/final.txt
