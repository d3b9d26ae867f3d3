#!/bin/bash
# the two large atomic data sets on the round's last kernels (one step each)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_big; mkdir -p $O
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset w7big > $O/bench_w7big.json 2> /dev/null; tail -c 420 $O/bench_w7big.json; echo
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset cd23like > $O/bench_cd23like.json 2> /dev/null; tail -c 420 $O/bench_cd23like.json; echo
