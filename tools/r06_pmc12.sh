#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && bash tools/pmc_collect.sh 10000000 r06q classic "1 2"
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_r06q/pass* > gpurun_out/pmc_r06q_summary.txt
rm -rf gpurun_out/pmc_r06q
grep -A22 "^k_thermal\|^k_rpkt" gpurun_out/pmc_r06q_summary.txt | head -80
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --preset cd23like > gpurun_out/bench_cd23like_slow3.json 2>/dev/null; tail -c 400 gpurun_out/bench_cd23like_slow3.json
