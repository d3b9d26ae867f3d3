#!/bin/bash
# round 5, end: what the driver runs (smoke, the GPU suite, the default bench line) + the nltenebular line at the last kernels
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_final; mkdir -p $O
bash tools/final_check.sh 2>&1 | tee $O/final_check.txt
python3 bench.py --options nltenebular --no-cpu-baseline > $O/bench_nltenebular.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_nltenebular.json')); print('nltenebular', round(d['ms_per_step'],1), round(d['value']/1e6,1), d['kernel_ms_by_kind_last_step'])"
python3 bench.py > $O/bench_default.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print('classic', round(d['ms_per_step'],1), round(d['value']/1e6,1), d['kernel_ms_by_kind_last_step'], d['cpu_baseline']['value'])"
