cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('classic', round(j['ms_per_step'],1), round(j['value']/1e6,1), 'M/s', {k:round(v,1) for k,v in j['kernel_breakdown_last_step'].items() if k.endswith('_ms')})"
bash tools/neb_ab.sh A=1
