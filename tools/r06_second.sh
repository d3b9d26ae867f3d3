#!/bin/bash
# round 6: the whole GPU suite on the restructured transition loop (sorted count, one rare-path test), then the bench lines
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r06_second; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
tail -5 $O/tests.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -c 700 $O/bench_default.json
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --options nltenebular > $O/bench_nltenebular.json 2> /dev/null; tail -c 500 $O/bench_nltenebular.json
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset w7big > $O/bench_w7big.json 2> /dev/null; tail -c 500 $O/bench_w7big.json
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset cd23like > $O/bench_cd23like.json 2> /dev/null; tail -c 500 $O/bench_cd23like.json
