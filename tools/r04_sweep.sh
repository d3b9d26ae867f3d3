#!/bin/bash
# round 4, after the pipelined k_rpkt and four r-packet steps per launch: the neighbouring launch parameters again
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
bash tools/ab_env4.sh ARTIS_AMD_DRAIN_R "1 0 2" 2
bash tools/ab_env4.sh ARTIS_AMD_BUDGET_T "2048 1536 3072" 2
bash tools/ab_env4.sh ARTIS_AMD_BUDGET_R_SMALL "0 2 8" 1
bash tools/ab_env4.sh ARTIS_AMD_DRAIN_T "48 32 64" 1
