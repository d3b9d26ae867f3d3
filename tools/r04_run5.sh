#!/bin/bash
# kernel trace of the virtual-packet bench (1e6 packets, t = 5 d) and of the expansion-opacity bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "ci_classic_vpkt --t-days 5 --packets 1000000" "kilonova_expopac"; do
  name=$(echo $cfg | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --options $cfg > $O/trace_$name.log 2>&1
  cp $O/trace_$name/*/*kernel_stats.csv $O/kernel_stats_bench_$name.csv 2>/dev/null
  rm -rf $O/trace_$name
  head -12 $O/kernel_stats_bench_$name.csv | cut -c1-60,100-200
done
