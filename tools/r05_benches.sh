#!/bin/bash
# the bench lines of profiles/<round>/ alone (no traces, no counters), after a change late in a round: bash tools/r05_benches.sh r05
T=${1:-r05}
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/profile_$T; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --options kilonova_lte --no-cpu-baseline > $O/bench_kilonova_lte.json 2> /dev/null
python3 bench.py --options nltenebular --no-cpu-baseline > $O/bench_nltenebular.json 2> /dev/null
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --options kilonova_expopac > $O/bench_kilonova_expopac.json 2> /dev/null
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --options classic_expopac_therm > $O/bench_classic_expopac_therm.json 2> /dev/null
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --options ci_classic_vpkt --t-days 5 --packets 1000000 > $O/bench_ci_classic_vpkt_1e6_t5d.json 2> /dev/null
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset w7big > $O/bench_w7big.json 2> /dev/null
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset cd23like > $O/bench_cd23like.json 2> /dev/null
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,1), 'M', round(d['ms_per_step'],1), 'ms')"; done
