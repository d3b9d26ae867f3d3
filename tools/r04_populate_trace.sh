#!/bin/bash
# kernel trace of the cell-cache population alone (tools/populate_time.py): gpurun_out/populate_trace/kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/populate_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/populate_time.py ${1:-w7} 4 > $O/trace.log 2>&1
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats_populate_${1:-w7}.csv 2>/dev/null
rm -rf $O/trace
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats_populate_${1:-w7}.csv')))
for r in rows[:16]:
    print('%-60s %5d calls %8.2f ms per population' % (r['Name'][:60], int(r['Calls']), float(r['TotalDurationNs'])/1e6/5))
PY
