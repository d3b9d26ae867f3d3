#!/bin/bash
# compiler flag variants of the classic build (scratch/lib_<name>.so), one bench step each
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_flags; mkdir -p $O
bash tools/ab_env.sh "X=1" "ARTIS_AMD_SO=scratch/lib_maxilp.so" "ARTIS_AMD_SO=scratch/lib_maxmem.so" "ARTIS_AMD_SO=scratch/lib_o2.so" "ARTIS_AMD_SO=scratch/lib_nounroll.so" "X=1" 2>&1 | tee $O/ab.txt
