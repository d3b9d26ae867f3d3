#!/bin/bash
# rocprofv3 kernel trace of the default bench (3 steps) -> gpurun_out/r06_trace/kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/r06_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats_bench_default.csv 2>/dev/null
rm -rf $O/trace
head -40 $O/kernel_stats_bench_default.csv | cut -c1-160
cd $R && python3 -m pytest tests/test_lightcurve_phases.py -x -q -m gpu -s 2>&1 | tail -6
