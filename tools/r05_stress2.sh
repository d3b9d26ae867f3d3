#!/bin/bash
# parity stress at 2e6 packets for the other options builds (tools/stress_parity.py; the C oracle on the box's host cores)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_stress2; mkdir -p $O
timeout 3000 python3 tools/stress_parity.py 2000000 12 kilonova_lte,kilonova_expopac,classic_expopac_therm,ci_classic_vpkt 2>&1 | tail -14 | tee $O/stress.txt
