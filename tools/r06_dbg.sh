#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r06_dbg; mkdir -p $O
ARTIS_AMD_TRACE=1 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "tiling and classic_expopac_therm" > $O/adapt.log 2>&1; grep -v "launch " $O/adapt.log | tail -30
ARTIS_AMD_TILE_ADAPT=0 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiling and classic_expopac_therm" > $O/noadapt.log 2>&1; tail -3 $O/noadapt.log
