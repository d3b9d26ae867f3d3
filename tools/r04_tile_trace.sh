#!/bin/bash
# round 4: launch-by-launch trace of a run forced into 4 tiles (where does the 3.5x go?), then the full GPU suite
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/r04t
mkdir -p $O
cd $R
ARTIS_AMD_TRACE=1 ARTIS_AMD_CACHE_BUDGET_MB=11600 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_4tiles.json 2> $O/trace_4tiles.err
grep -c "launch" $O/trace_4tiles.err
python3 bench.py --steps 1 --warmup 1 --preset w7big --no-cpu-baseline > $O/bench_w7big.json 2> $O/bench_w7big.err
tail -c 1500 $O/bench_w7big.json | head -c 600
python3 -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1
tail -3 $O/gputest.log
