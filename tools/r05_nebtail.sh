#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_nebtail; mkdir -p $O
AB_ARGS="--options nltenebular" bash tools/ab_env.sh "X=1" "ARTIS_AMD_TAIL=8192" "ARTIS_AMD_TAIL=24576" "ARTIS_AMD_TAIL=32768" "ARTIS_AMD_TAIL=49152" "X=1" 2>&1 | tee $O/ab.txt
