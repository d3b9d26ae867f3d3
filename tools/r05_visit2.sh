#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r05_visit2; mkdir -p $O; export VISIT_OUT=$O
python3 tools/visit_sparsity.py --preset w7 > $O/visit_sparsity_w7.md 2> $O/visit_w7.err
python3 tools/visit_sparsity.py --preset w7big > $O/visit_sparsity_w7big.md 2> $O/visit_w7big.err
tail -14 $O/visit_sparsity_w7.md $O/visit_sparsity_w7big.md; tail -3 $O/*.err
