#!/bin/bash
# The nltenebular part of profiles/<round>/ on the GPU box (outputs under gpurun_out/profile_<round>/):
#   kernel_stats_bench_nltenebular.csv, pmc_summary_nltenebular.txt, pmc_traffic_nltenebular.json, bench_nltenebular.json
# usage: bash tools/profile_nltenebular.sh r03
T=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/profile_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_neb -- python3 $R/bench.py --options nltenebular --steps 3 --warmup 1 --no-cpu-baseline > $O/trace_neb.log 2>&1
cp $O/trace_neb/*/*kernel_stats.csv $O/kernel_stats_bench_nltenebular.csv 2>/dev/null
rm -rf $O/trace_neb
cd $R && bash tools/pmc_collect.sh 10000000 neb_$T nltenebular
cd $R && PMC_TRAFFIC_JSON=$O/pmc_traffic_nltenebular.json python3 tools/pmc_summary.py gpurun_out/pmc_neb_$T/pass* > $O/pmc_summary_nltenebular.txt
mkdir -p $R/profiles/$T && cp $O/pmc_traffic_nltenebular.json $R/profiles/$T/pmc_traffic_nltenebular.json
rm -rf $R/gpurun_out/pmc_neb_$T
python3 bench.py --options nltenebular --no-cpu-baseline > $O/bench_nltenebular.json 2> $O/bench_nltenebular.err
tail -c 300 $O/bench_nltenebular.json
