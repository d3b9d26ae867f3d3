#!/usr/bin/env python3
"""Compile the engine with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs, SGPRs, spills, scratch bytes per lane, occupancy (waves/SIMD), LDS bytes.

    python tools/kernel_resources.py [classic|kilonova_lte|nltenebular] [extra hipcc flags ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artis_amd import build as B  # noqa: E402


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "classic"
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    pflags = [] if preset == "classic" else [f"-DARTIS_PRESET_{preset.upper()}"]
    cmd = ["/opt/rocm/bin/hipcc", *B.FLAGS, *pflags, *extra, "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_kres.so",
           os.path.join(B.CSRC, "artis_engine.hip")]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True, check=True).stderr
    cur = None
    rows = {}
    for line in err.splitlines():
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: (\S+)", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = subprocess.run(["c++filt", val], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = cur.replace("(anonymous namespace)::", "").split("(")[0]
            rows[cur] = {}
        elif cur:
            rows[cur][key] = val
    print(f"{'kernel':22s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>4s} {'LDS':>6s}")
    for k, r in rows.items():
        print(f"{k:22s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} {r.get('VGPRs Spill', '?'):>6s} "
              f"{r.get('SGPRs Spill', '?'):>6s} {r.get('ScratchSize', '?'):>7s} {r.get('Occupancy', '?'):>4s} {r.get('LDS Size', '?'):>6s}")


if __name__ == "__main__":
    main()
