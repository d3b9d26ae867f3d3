#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_pool; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "on_demand or tiling or cooling_guides" 2>&1 | tail -5 | tee $O/test.log
ARTIS_AMD_MA_HOTFRAC=0.3 timeout 900 python3 tools/stress_parity.py 2000000 12 classic 2>&1 | tail -4 | tee $O/stress_ondemand.txt
