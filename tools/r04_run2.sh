#!/bin/bash
# round 4: full GPU suite + bench of the three options builds (interleaved estimator records)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8
summ() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); b=d['kernel_breakdown_last_step']
print('$1: %.1f ms/step %.1f M/s thermal %.1f rpkt %.1f' % (d['ms_per_step'], d['value']/1e6, b['thermal_ms'], b['rpkt_ms']), d['config']['cell_cache'])"; }
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; summ $O/bench_default.json
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --options nltenebular > $O/bench_nltenebular.json 2> $O/bench_nltenebular.err; summ $O/bench_nltenebular.json
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --options kilonova_lte > $O/bench_kilonova_lte.json 2> $O/bench_kilonova_lte.err; summ $O/bench_kilonova_lte.json
