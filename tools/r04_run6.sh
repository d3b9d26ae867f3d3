#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sparse_fills or tiling" 2>&1 | tail -3
bash tools/ab_bench.sh 2 base xbs xp16 xp32 xp40
for g in "1d 100" "1d 30" "2d 25" "3d 12" "3d 20"; do set -- $g
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --grid $1 --ncoord $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['kernel_breakdown_last_step']
print('grid $1 ncoord $2: %.1f ms/step  %.1f M/s  thermal %.1f ms  rpkt %.1f ms' % (d['ms_per_step'], d['value']/1e6, b['thermal_ms'], b['rpkt_ms']))"
done
