#!/bin/bash
# round 4: the new GPU tests, bench lines of the expansion-opacity and virtual-packet builds
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
python3 -m pytest tests -x -q -m gpu -k "reciprocity or hubble or long_directions or kilonova_lte_50cubed" 2>&1 | tail -8
summ() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); b=d['kernel_breakdown_last_step']
print('$1: %.1f ms/step %.1f M/s thermal %.1f rpkt %.1f steps %.3g' % (d['ms_per_step'], d['value']/1e6, b['thermal_ms'], b['rpkt_ms'], d['config']['packet_steps_per_step']), d['config']['cell_cache'])"; }
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --options kilonova_expopac > $O/bench_kilonova_expopac.json 2> $O/bench_kilonova_expopac.err; summ $O/bench_kilonova_expopac.json; tail -2 $O/bench_kilonova_expopac.err
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --options classic_expopac_therm > $O/bench_classic_expopac_therm.json 2> $O/bench_classic_expopac_therm.err; summ $O/bench_classic_expopac_therm.json
timeout 900 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --options ci_classic_vpkt --t-days 5 --packets 1000000 > $O/bench_vpkt_1e6.json 2> $O/bench_vpkt_1e6.err; summ $O/bench_vpkt_1e6.json; tail -2 $O/bench_vpkt_1e6.err
