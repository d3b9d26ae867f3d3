#!/bin/bash
# per-launch trace of the thermal kernel for each environment given: tools/trace_thermal.sh "<ENV=..>" ...
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg ARTIS_AMD_TRACE=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 >/dev/null | grep "kind 2 " | awk '{printf "%s n=%s %s ms | ", $3, $6, $7} END {print ""}' | sed 's/n=n=/n=/g'
done
