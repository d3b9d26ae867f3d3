#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_ab6; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "on_demand or walker_refill or (engine_matches_oracle and not large)" > $O/test.log 2>&1; tail -3 $O/test.log
bash tools/ab_env.sh "X=0" "ARTIS_AMD_SO=scratch/lib_noabs.so" "ARTIS_AMD_SORT_NUMAJOR=1" "X=0" "ARTIS_AMD_SO=scratch/lib_noabs.so" "ARTIS_AMD_SORT_NUMAJOR=1" "ARTIS_AMD_MA_HOTFRAC=0.3 ARTIS_AMD_MA_POOLFRAC=0.25" 2>&1 | tee $O/ab.txt
