#!/bin/bash
# round 4, after the pipelined k_rpkt: the r-packet launch parameters again (steps per packet and launch, drain, tail threshold)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
bash tools/ab_env4.sh ARTIS_AMD_BUDGET_R "8 6 12 16" 1
bash tools/ab_env4.sh ARTIS_AMD_DRAIN_R "1 0 2 4" 1
bash tools/ab_env4.sh ARTIS_AMD_TAIL "4096 2048 8192" 1
bash tools/ab_env4.sh ARTIS_AMD_BUDGET_T "2048 1024 4096" 1
