#!/bin/bash
# the round's end on the GPU box: final check, then the parity stress at 2e6 packets (default records, and the headline data forced onto on-demand records)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_stress; mkdir -p $O
bash tools/final_check.sh 2>&1 | tee $O/final.txt
timeout 1500 python3 tools/stress_parity.py 2000000 12 classic,nltenebular 2>&1 | tail -8 | tee $O/stress.txt
ARTIS_AMD_MA_HOTFRAC=0.3 timeout 900 python3 tools/stress_parity.py 2000000 12 classic 2>&1 | tail -4 | tee $O/stress_ondemand.txt
