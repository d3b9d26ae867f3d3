#!/bin/bash
# the classic part of tools/profile_round.sh alone: kernel trace, PMC passes + summary, the default bench line
T=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/profile_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats_bench_default.csv 2>/dev/null
cp $O/trace/*/*domain_stats.csv $O/domain_stats_bench_default.csv 2>/dev/null
rm -rf $O/trace
cd $R && bash tools/pmc_collect.sh 10000000 $T
cd $R && PMC_TRAFFIC_JSON=$O/pmc_traffic.json python3 tools/pmc_summary.py gpurun_out/pmc_$T/pass* > $O/pmc_summary.txt
cp $O/pmc_traffic.json $R/profiles/$T/pmc_traffic.json
rm -rf $R/gpurun_out/pmc_$T
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json
