#!/bin/bash
# round 5: k_thermal_q (ARTIS_AMD_REFILL=1) against k_thermal: parity, then timing
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_refill; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "walker_refill" > $O/test.log 2>&1; tail -5 $O/test.log
bash tools/ab_env.sh "ARTIS_AMD_REFILL=0" "ARTIS_AMD_REFILL=1" "ARTIS_AMD_REFILL=1 ARTIS_AMD_TQ_LOW=32" "ARTIS_AMD_REFILL=1 ARTIS_AMD_TQ_LOW=56" "ARTIS_AMD_REFILL=0" "ARTIS_AMD_REFILL=1" 2>&1 | tee $O/ab.txt
