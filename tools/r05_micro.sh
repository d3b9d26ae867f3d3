#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_sortevery; mkdir -p $O
A="ARTIS_AMD_SO=scratch/lib_se.so"
bash tools/ab_env.sh "$A" "$A ARTIS_AMD_SORT_EVERY_R=2" "$A ARTIS_AMD_SORT_EVERY_T=2" "$A ARTIS_AMD_SORT_EVERY_R=2 ARTIS_AMD_SORT_EVERY_T=2" "$A ARTIS_AMD_SORT_EVERY_R=3 ARTIS_AMD_SORT_EVERY_T=3" "$A ARTIS_AMD_SORT_EVERY_T=4" "$A" 2>&1 | tee $O/ab.txt
