#!/bin/bash
# A/B of engine builds on one GPU box, interleaved rounds in one call: tools/ab_bench.sh <rounds> <name> [<name> ...]
# (name -> artis_amd/libartis_amd_<name>.so; "base" = artis_amd/libartis_amd.so). Prints ms/step and the kernel split.
R=$1; shift
for r in $(seq 1 $R); do
  for name in "$@"; do
    so=$PWD/artis_amd/libartis_amd_$name.so
    [ "$name" = base ] && so=$PWD/artis_amd/libartis_amd.so
    ARTIS_AMD_SO=$so python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['kernel_breakdown_last_step']
print('$name round $r: %.1f ms/step  %.1f M/s  thermal %.1f ms  rpkt %.1f ms' % (d['ms_per_step'], d['value']/1e6, b['thermal_ms'], b['rpkt_ms']))"
  done
done
