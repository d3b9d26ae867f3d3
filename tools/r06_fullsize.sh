#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r06_fullsize; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size and (expopac or vpkt)" > $O/expopac.log 2>&1; tail -3 $O/expopac.log
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "expansion or expopac" > $O/expopac_small.log 2>&1; tail -3 $O/expopac_small.log
