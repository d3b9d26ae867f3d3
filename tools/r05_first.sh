#!/bin/bash
# round 5, first GPU call: which (cell, level) records a step visits (w7, w7big), the GPU suite, the default bench line
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_first; mkdir -p $O
python3 tools/visit_sparsity.py --preset w7 > $O/visit_sparsity_w7.md 2> $O/visit_w7.err
python3 tools/visit_sparsity.py --preset w7big > $O/visit_sparsity_w7big.md 2> $O/visit_w7big.err
python3 -m pytest tests -x -q -m gpu -rs > $O/gputest.log 2>&1; tail -3 $O/gputest.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
cat $O/visit_sparsity_w7.md $O/visit_sparsity_w7big.md
