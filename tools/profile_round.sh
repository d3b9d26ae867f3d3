#!/bin/bash
# Everything profiles/<round>/ holds, in one go on the GPU box (outputs under gpurun_out/profile_<round>/):
#   1. rocprofv3 --kernel-trace --stats of the default bench command           -> kernel_stats / domain_stats
#   2. the PMC passes (tools/pmc_collect.sh) and their summary                   -> pmc_summary.txt, pmc_traffic.json
#   3. bench.py with the fresh pmc_traffic.json in place (its roofline.traffic)  -> bench_default.json
#   4. the same workload on the other two options builds                         -> bench_kilonova_lte.json, bench_nltenebular.json
# usage: bash tools/profile_round.sh r02
T=${1:-r06}
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
python3 bench.py --options kilonova_lte --no-cpu-baseline > $O/bench_kilonova_lte.json 2> /dev/null
bash tools/profile_nltenebular.sh $T
# round 4: the built paths without a headline of their own, and atomic data of realistic size in one tile
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --options kilonova_expopac > $O/bench_kilonova_expopac.json 2> /dev/null
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --options classic_expopac_therm > $O/bench_classic_expopac_therm.json 2> /dev/null
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --options ci_classic_vpkt --t-days 5 --packets 1000000 > $O/bench_ci_classic_vpkt_1e6_t5d.json 2> /dev/null
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset w7big > $O/bench_w7big.json 2> /dev/null
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset cd23like > $O/bench_cd23like.json 2> /dev/null
bash tools/r04_populate_trace.sh w7 > $O/populate_trace.txt 2>&1; cp $R/gpurun_out/populate_trace/kernel_stats_populate_w7.csv $O/ 2>/dev/null
tail -c 400 $O/bench_default.json
