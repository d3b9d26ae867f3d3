#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r06_fills; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "on_demand or pool_used or tiling or large_cases or long_directions or cellcache" > $O/tests.log 2>&1; tail -3 $O/tests.log
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset cd23like > $O/bench_cd23like.json 2>/dev/null; tail -c 420 $O/bench_cd23like.json
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2>/dev/null; tail -c 420 $O/bench_default.json
ARTIS_AMD_BUDGET_R=5 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_default_r5.json 2>/dev/null; tail -c 420 $O/bench_default_r5.json
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --options nltenebular > $O/bench_nltenebular.json 2>/dev/null; tail -c 420 $O/bench_nltenebular.json
