#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_cold2; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "on_demand or (tiling and 0.3)" > $O/test.log 2>&1; tail -3 $O/test.log
bash tools/ab_env.sh "X=0" "ARTIS_AMD_MA_HOTFRAC=0.3 ARTIS_AMD_MA_POOLFRAC=0.25" 2>&1 | tee $O/ab_w7.txt
timeout 1500 python3 bench.py --preset cd23like --steps 1 --warmup 1 --no-cpu-baseline 2>$O/cd23.err > $O/bench_cd23like.json; python3 -c "
import json; d=json.load(open('$O/bench_cd23like.json')); print('cd23like auto', round(d['ms_per_step'],1), round(d['value']/1e6,1), d['config']['cell_cache'], d['kernel_ms_by_kind_last_step'], d['kernel_breakdown_last_step'])" | tee $O/big.txt
