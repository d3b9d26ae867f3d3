"""Build the HIP engine in-tree: artis_amd/libartis_amd.so (gfx950)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libartis_amd.so")
SOURCES = ["artis_engine.hip", "physics.h", "tables.h", "model_build.h"]
# -ffp-contract=off: the operation order of physics.h is part of the parity contract (no FMA contraction).
# -munsafe-fp-atomics: estimator adds become global_atomic_add_f64, not compare-and-swap loops.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-munsafe-fp-atomics"]


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "..", "include", h)
                                                        for h in ("artis_amd.h", "artis_options.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, extra_flags=()) -> str:
    if not force and not needs_build():
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *FLAGS, *extra_flags, "-o", SO, os.path.join(CSRC, "artis_engine.hip")]
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True))
