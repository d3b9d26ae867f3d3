#!/bin/bash
# round 5: the whole GPU suite, then everything profiles/r05 holds
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_suite; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 1500 python3 -m pytest tests -x -q -m gpu -rs > $O/gputest.log 2>&1; tail -4 $O/gputest.log
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
