#!/bin/bash
# round 4, first measurements of the filter-only records: kernel trace of the headline bench, w7big in one tile, 4 tiles
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats_bench_default.csv 2>/dev/null
rm -rf $O/trace
cd $R
summ() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); b=d['kernel_breakdown_last_step']
print('$1: %.1f ms/step %.1f M/s thermal %.1f rpkt %.1f' % (d['ms_per_step'], d['value']/1e6, b['thermal_ms'], b['rpkt_ms']), d['config']['cell_cache'])"; }
python3 bench.py --steps 1 --warmup 1 --preset w7big --no-cpu-baseline > $O/bench_w7big.json 2> $O/bench_w7big.err; summ $O/bench_w7big.json
ARTIS_AMD_CACHE_BUDGET_MB=11600 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_4tiles.json 2> $O/bench_4tiles.err; summ $O/bench_4tiles.json
ARTIS_AMD_CACHE_BUDGET_MB=23100 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_2tiles.json 2> $O/bench_2tiles.err; summ $O/bench_2tiles.json
head -30 $O/kernel_stats_bench_default.csv | cut -c1-150
