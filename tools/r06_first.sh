#!/bin/bash
# round 6, first GPU call: the new parity cases + the suites the record change touches, then bench lines for the three data sets
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r06_first; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_cases or cellcache or long_directions or on_demand or pool_used or tiling or filters_decide or refill or tail_kernel or repeated" > $O/tests.log 2>&1
tail -5 $O/tests.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
ARTIS_AMD_TRACE=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset w7big > $O/bench_w7big.json 2> $O/bench_w7big.err; tail -c 900 $O/bench_w7big.json
ARTIS_AMD_TRACE=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset cd23like > $O/bench_cd23like.json 2> $O/bench_cd23like.err; tail -c 900 $O/bench_cd23like.json
grep -c "kind 1 " $O/bench_cd23like.err; grep -c "launch" $O/bench_cd23like.err
gzip -f $O/*.err
