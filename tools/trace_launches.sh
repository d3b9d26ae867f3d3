#!/bin/bash
# per-launch list sizes and durations of one headline step (ARTIS_AMD_TRACE): gpurun_out/launch_trace.txt
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
ARTIS_AMD_TRACE=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline ${1:+--options $1} > /dev/null 2> gpurun_out/launch_trace_raw.txt
grep "launch " gpurun_out/launch_trace_raw.txt | tail -n +1 > gpurun_out/launch_trace.txt
python3 - <<'PY'
import re
rows=[]
for l in open("gpurun_out/launch_trace.txt"):
    m=re.search(r"launch (\d+) kind (\d+) n=(\d+) ([\d.]+) ms",l)
    if m: rows.append((int(m.group(1)),int(m.group(2)),int(m.group(3)),float(m.group(4))))
    m=re.search(r"launch (\d+) tail n=(\d+)\+\d+ ([\d.]+) ms",l)
    if m: rows.append((int(m.group(1)),99,int(m.group(2)),float(m.group(3))))
# second half of the file = the timed step
half=len(rows)//2; rows=rows[half:]
import collections
for kind,name in ((1,"k_rpkt"),(2,"k_thermal"),(3,"k_slow"),(99,"k_tail")):
    r=[x for x in rows if x[1]==kind]
    print(name, len(r), "launches", round(sum(x[3] for x in r),1), "ms")
    for lo,hi in ((0,1e5),(1e5,3e5),(3e5,1e6),(1e6,3e6),(3e6,2e7)):
        s=[x for x in r if lo<=x[2]<hi]
        if s: print(f"   n in [{lo:.0e},{hi:.0e}): {len(s)} launches, {sum(x[3] for x in s):.1f} ms, packets {sum(x[2] for x in s):.3e}")
PY
