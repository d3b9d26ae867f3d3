#!/bin/bash
# Final measurements of a round (written in round 6) on the GPU box: what profiles/r06/ holds at the round's last source commit.
#   profile_round.sh r06: kernel trace + 7 PMC passes + bench line (with the CPU baseline) of the headline; kilonova_lte bench; nltenebular trace + PMC + bench;
#                         expansion-opacity / virtual-packet benches; w7big and cd23like benches; population trace
#   then the PMC passes of the lines that had none: kilonova_lte, w7big, cd23like -> pmc_traffic_<...>.json, and their bench lines again
#   (now with roofline.frac), and the 2e6-packet stress parity of the round's kernels
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
T=r06
O=$R/gpurun_out/profile_$T
mkdir -p $O $R/profiles/$T
bash tools/profile_round.sh $T > $O/profile_round.log 2>&1
tail -3 $O/profile_round.log | cut -c1-300
# kilonova_lte
bash tools/pmc_collect.sh 10000000 kl_$T kilonova_lte > $O/pmc_kl.log 2>&1
PMC_TRAFFIC_JSON=$O/pmc_traffic_kilonova_lte.json python3 tools/pmc_summary.py gpurun_out/pmc_kl_$T/pass* > $O/pmc_summary_kilonova_lte.txt
cp $O/pmc_traffic_kilonova_lte.json profiles/$T/; rm -rf gpurun_out/pmc_kl_$T
python3 bench.py --options kilonova_lte --no-cpu-baseline > $O/bench_kilonova_lte.json 2> /dev/null
# w7big
PMC_EXTRA="--preset w7big" PMC_TIMEOUT=600 bash tools/pmc_collect.sh 10000000 w7big_$T classic > $O/pmc_w7big.log 2>&1
PMC_TRAFFIC_JSON=$O/pmc_traffic_w7big.json python3 tools/pmc_summary.py gpurun_out/pmc_w7big_$T/pass* > $O/pmc_summary_w7big.txt
cp $O/pmc_traffic_w7big.json profiles/$T/; rm -rf gpurun_out/pmc_w7big_$T
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset w7big > $O/bench_w7big.json 2> /dev/null
# cd23like
PMC_EXTRA="--preset cd23like" PMC_TIMEOUT=900 bash tools/pmc_collect.sh 10000000 cd23_$T classic > $O/pmc_cd23like.log 2>&1
PMC_TRAFFIC_JSON=$O/pmc_traffic_cd23like.json python3 tools/pmc_summary.py gpurun_out/pmc_cd23_$T/pass* > $O/pmc_summary_cd23like.txt
cp $O/pmc_traffic_cd23like.json profiles/$T/; rm -rf gpurun_out/pmc_cd23_$T
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset cd23like > $O/bench_cd23like.json 2> /dev/null
# stress parity of the round's kernels
timeout 1500 python3 tools/stress_parity.py 2000000 10 > $O/stress_parity_2e6.txt 2>&1; tail -4 $O/stress_parity_2e6.txt
rm -f gpurun_out/pmc_*.log
ls -la $O | head -50
