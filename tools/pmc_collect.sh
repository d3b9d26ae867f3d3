#!/bin/bash
# Collect PMC counters for the propagation kernels: separate rocprofv3 passes (kernel-trace + pmc only).
# (A pass with TA_* counters did not finish within 300 s on this pool and is left out.)
# usage (on the GPU box): bash tools/pmc_collect.sh [packets] [tag] [options preset] [passes, e.g. "1 2 4"]
#   -> gpurun_out/pmc_<tag>/<pass>/..., gpurun_out/pmc_<tag>_<pass>.log
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
P=${1:-10000000}
T=${2:-r02}
O=${3:-classic}
ONLY=${4:-}
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
  "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  tag=pass$i
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $i "; then continue; fi
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$T/$tag -- python3 $R/bench.py --packets $P --steps 1 --warmup 0 --no-cpu-baseline --options $O $PMC_EXTRA > $R/gpurun_out/pmc_${T}_$tag.log 2>&1
  echo "$tag rc=$? : $set"
done
