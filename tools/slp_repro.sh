#!/bin/bash
# Reproducer for the reason -fno-slp-vectorize is in artis_amd/build.py FLAGS: the nltenebular library built WITH SLP
# vectorisation (everything else equal) against the GPU parity test that caught the wrong store in k_gamma.
#   here (no GPU needed):  bash tools/slp_repro.sh build      -> artis_amd/libartis_amd_nltenebular_slp.so
#   on the GPU box:        bash tools/slp_repro.sh test       -> the parity test with that library, then with the shipped one
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
if [ "$1" = build ]; then
  python3 - <<'PY'
import os, subprocess
from artis_amd import build as B
so = B.so_path("nltenebular").replace(".so", "_slp.so")
cmd = ["/opt/rocm/bin/hipcc", *B.FLAGS, "-fslp-vectorize", "-DARTIS_PRESET_NLTENEBULAR", '-DARTIS_PRESET_NAME="nltenebular"', "-o", so,
       os.path.join(B.CSRC, "artis_engine.hip")]
subprocess.check_call(cmd)
print(so)
PY
else
  echo "== with -fslp-vectorize"
  ARTIS_AMD_SO_NLTENEBULAR=$PWD/artis_amd/libartis_amd_nltenebular_slp.so python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nltenebular_preset" 2>&1 | tail -12
  echo "== shipped build (-fno-slp-vectorize)"
  python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nltenebular_preset" 2>&1 | tail -3
fi
