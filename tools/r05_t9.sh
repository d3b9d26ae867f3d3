#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_t9; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiling or on_demand or tile" > $O/test.log 2>&1; tail -4 $O/test.log
for t in 4096 8192 32768; do ARTIS_AMD_TAIL=$t python3 bench.py --options nltenebular --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('nltenebular tail $t', round(d['ms_per_step'],1), d['kernel_ms_by_kind_last_step'])" | tee -a $O/neb.txt; done
