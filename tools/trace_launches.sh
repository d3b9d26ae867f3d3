#!/bin/bash
# per-launch trace (kind, packets, ms) of one step: tools/trace_launches.sh "<ENV=..>" ...
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg ARTIS_AMD_TRACE=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 >/dev/null | grep "\] launch" | awk '{k=$4; if ($3=="tail") k="tail"; else k=$5; printf "%s:%s:%s ", ($3=="tail"?"tail":"k"$5), $6, ($3=="tail"?$5:$7)} END {print ""}'
done
