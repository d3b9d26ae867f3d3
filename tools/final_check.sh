#!/bin/bash
# what the driver runs at a round's end, in one go on the GPU box: build + smoke, the GPU test suite, the default bench line
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
python3 -m pytest tests -x -q -m gpu -rs > gpurun_out/gputest_final.log 2>&1; grep -E "passed|failed" gpurun_out/gputest_final.log | tail -2
# a skipped test is reported, never silent (the compiled reference-side binding skips when oracle/_ref/update_packets_amd did not travel)
grep -E "^SKIPPED" gpurun_out/gputest_final.log | sed 's/^/  !! /'
python3 bench.py 2>/dev/null | tail -1 | cut -c1-330
