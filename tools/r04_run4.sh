#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_engine_matches_oracle or tiling or tail_kernel or deterministic or work_list" 2>&1 | tail -3
bash tools/ab_env4.sh ARTIS_AMD_WCACHE "0 1" 2
bash tools/ab_env4.sh ARTIS_AMD_WCACHE "0 1" 1 --options nltenebular
for w in 0 1; do
  ARTIS_AMD_WCACHE=$w bash tools/pmc_collect.sh 10000000 wc$w classic "7" > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_wc$w/pass* | grep -A6 "^k_rpkt" | grep -E "k_rpkt|WRITE_SIZE"
  rm -rf gpurun_out/pmc_wc$w gpurun_out/pmc_wc${w}_pass*.log
done
