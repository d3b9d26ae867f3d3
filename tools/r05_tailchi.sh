#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_tailchi; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nltenebular or tail or tiling" 2>&1 | tail -3 | tee $O/test.log
AB_ARGS="--options nltenebular" bash tools/ab_env.sh "X=1" "ARTIS_AMD_SO_NLTENEBULAR=scratch/lib_neb_chi0.so" "X=1" "ARTIS_AMD_SO_NLTENEBULAR=scratch/lib_neb_chi0.so" 2>&1 | tee $O/ab.txt
bash tools/ab_env.sh "X=1" 2>&1 | tee -a $O/ab.txt
