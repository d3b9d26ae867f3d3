#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_env4.sh ARTIS_AMD_DRAIN_T "48 24 96 0" 1
bash tools/ab_env4.sh ARTIS_AMD_BUDGET_T "2048 1024 4096" 1
bash tools/ab_env4.sh ARTIS_AMD_TAIL "4096 2048 8192" 1
bash tools/ab_env4.sh ARTIS_AMD_DRAIN_MIN "1000000 500000 2000000" 1
