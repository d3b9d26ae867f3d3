#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_suite2; mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu -rs > $O/gputest.log 2>&1; tail -6 $O/gputest.log
ARTIS_AMD_CACHE_BUDGET_MB=11600 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('quarter cache, auto tiers', round(d['ms_per_step'],1), d['config']['cell_cache'], d['kernel_ms_by_kind_last_step'])" | tee $O/quarter.txt
ARTIS_AMD_MA_HOTFRAC=1 ARTIS_AMD_CACHE_BUDGET_MB=11600 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('quarter cache, static records', round(d['ms_per_step'],1), d['config']['cell_cache'], d['kernel_ms_by_kind_last_step'])" | tee -a $O/quarter.txt
