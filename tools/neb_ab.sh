cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for cfg in "$@"; do
  env $cfg python3 bench.py --options nltenebular --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$cfg', round(j['ms_per_step'],1), round(j['value']/1e6,1), 'M/s rpkt+dense', round(j['kernel_breakdown_last_step']['rpkt_ms'],1), 'thermal', round(j['kernel_breakdown_last_step']['thermal_ms'],1))"
done
