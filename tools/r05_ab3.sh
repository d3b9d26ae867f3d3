#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_ab3; mkdir -p $O
bash tools/ab_env.sh "ARTIS_AMD_REFILL=0" "ARTIS_AMD_SO=scratch/lib_fb0.so" "ARTIS_AMD_REFILL=1" "ARTIS_AMD_SO=scratch/lib_blk.so ARTIS_AMD_REFILL=1" "ARTIS_AMD_SO=scratch/lib_blk.so ARTIS_AMD_REFILL=0" "ARTIS_AMD_SO=scratch/lib_tb1024.so ARTIS_AMD_REFILL=1" "ARTIS_AMD_SO=scratch/lib_tb1024.so ARTIS_AMD_REFILL=1 ARTIS_AMD_TQ_LOW=32" "ARTIS_AMD_REFILL=0" 2>&1 | tee $O/ab.txt
ARTIS_BENCH_VERBOSE=1 python3 bench.py --preset w7big --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_w7big.json 2> $O/bench_w7big.err; tail -1 $O/bench_w7big.err | cut -c1-1500
ARTIS_BENCH_VERBOSE=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_w7.json 2> $O/bench_w7.err; tail -1 $O/bench_w7.err | cut -c1-1500
