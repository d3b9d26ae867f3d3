#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_neb; mkdir -p $O
for cfg in "X=0" "ARTIS_AMD_SO_NLTENEBULAR=scratch/lib_neb_fb0.so" "X=0" "ARTIS_AMD_SO_NLTENEBULAR=scratch/lib_neb_fb0.so"; do env $cfg python3 bench.py --options nltenebular --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('nltenebular $cfg', round(d['ms_per_step'],1), d['kernel_ms_by_kind_last_step'])" | tee -a $O/neb.txt; done
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size and vpkt" > $O/test.log 2>&1; tail -2 $O/test.log
ARTIS_AMD_CACHE_BUDGET_MB=11600 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('quarter cache, engine choice', round(d['ms_per_step'],1), d['config']['cell_cache'])" | tee $O/quarter.txt
