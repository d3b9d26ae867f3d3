#!/bin/bash
# rocprofv3 --kernel-trace --stats of the nltenebular bench command, its bench line, and a per-launch trace of one step
# usage: bash tools/profile_nltenebular.sh r03
T=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profile_neb_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --options nltenebular --steps 3 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats_bench_nltenebular.csv 2>/dev/null
rm -rf $O/trace
cd $R
python3 bench.py --options nltenebular --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_nltenebular.json 2> /dev/null
ARTIS_AMD_TRACE=1 python3 bench.py --options nltenebular --steps 1 --warmup 0 --no-cpu-baseline 2> $O/launch_trace.txt > /dev/null
tail -c 300 $O/trace.log
