#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_dbg5; mkdir -p $O
VARIANTS=";;ARTIS_AMD_SORT=0" timeout 900 python3 tools/r05_determinism.py classic 3 1 > $O/det1.txt 2>&1; head -60 $O/det1.txt
