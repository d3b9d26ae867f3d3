#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r06_last; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 -m pytest tests -x -q -m gpu -rs > $O/tests.log 2>&1; grep -E "passed|failed|SKIPPED" $O/tests.log | tail -3
bash tools/profile_nltenebular.sh r06 > $O/profile_neb.log 2>&1; tail -c 300 gpurun_out/profile_r06/bench_nltenebular.json
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
