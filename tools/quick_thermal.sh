# bench line + instruction counters of k_thermal (two PMC passes) for the library as built
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); b=j['kernel_breakdown_last_step']; print(round(j['ms_per_step'],1), round(j['value']/1e6,1), 'M/s thermal', round(b['thermal_ms'],1), 'rpkt', round(b['rpkt_ms'],1), 'other', round(j['ms_per_step']-b['thermal_ms']-b['rpkt_ms'],1))"
bash tools/pmc_collect.sh 10000000 x ${1:-classic} "1 2" > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_x/pass* | grep -A22 "^k_thermal" | grep -E "SQ_INSTS_V|SQ_INSTS_SALU|lane"
rm -rf gpurun_out/pmc_x gpurun_out/pmc_x_pass*.log
