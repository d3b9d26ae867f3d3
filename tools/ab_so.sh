# A/B of the classic bench with another build of the library: tools/ab_so.sh <path to .so> [rounds]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
run() { python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],1), {k:round(v,1) for k,v in j['kernel_breakdown_last_step'].items() if k.endswith('_ms')})"; }
for r in $(seq 1 ${2:-1}); do
  run default
  ARTIS_AMD_SO=$1 run $(basename $1)
done
