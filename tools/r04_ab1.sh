#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
bash tools/ab_bench.sh 1 base xnt xspec
ARTIS_AMD_THERMAL_LDS_TB=768 bash tools/ab_bench.sh 1 base
done
