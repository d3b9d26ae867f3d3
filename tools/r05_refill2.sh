#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_refill; mkdir -p $O
ARTIS_AMD_SO=scratch/libprof.so python3 tools/r05_refill_prof.py 2>&1 | tail -3 | tee $O/prof.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "engine_matches_oracle and not large" > $O/test_fb.log 2>&1; tail -3 $O/test_fb.log
bash tools/ab_env.sh "ARTIS_AMD_REFILL=0" 2>&1 | tee $O/ab_fbwave.txt
python3 bench.py --options nltenebular --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('nltenebular', d['ms_per_step'], d['kernel_breakdown_last_step'])" | tee -a $O/ab_fbwave.txt
