#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_guide; mkdir -p $O
for g in 1 0; do echo "== ARTIS_AMD_COOLGUIDE=$g"; ARTIS_AMD_COOLGUIDE=$g ARTIS_AMD_SO=scratch/libprof.so timeout 600 python3 tools/thermal_counts.py 2>&1 | tail -7; done | tee $O/prof.txt
