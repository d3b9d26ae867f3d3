"""Build the HIP engine in-tree: artis_amd/libartis_amd.so (gfx950)."""
from __future__ import annotations

import fcntl
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libartis_amd.so")
SOURCES = ["artis_engine.hip", "physics.h", "tables.h", "model_build.h"]
# -ffp-contract=off: the operation order of physics.h is part of the parity contract (no FMA contraction).
# -munsafe-fp-atomics: estimator adds become global_atomic_add_f64, not compare-and-swap loops.
# -fno-slp-vectorize: with SLP vectorisation on, hipcc 7.2 packs the four int32 of a packet's hot line that follow each other
#   (emissiontype, trueemissiontype, absorptiontype, flags) into one vector in k_gamma and, on the path through the NT_ON
#   branch of do_ntlepton_deposit(), stores the vector it loaded instead of the updated absorptiontype (GPU parity test
#   test_engine_matches_oracle_nltenebular_preset caught it; -O1, noinline or this flag all give the right answer). The
#   kernels compute in f64 and their loads/stores are merged by the separate load/store vectoriser: no measured cost.
#   Round 3: tools/slp_repro.sh builds the nltenebular library WITH SLP vectorisation and runs that test: it passes now
#   (profiles/r03/slp_repro.txt: the code around the store has changed since); the flag stays as a guard, and the classic
#   bench is 1 % faster with it than without.
# -mllvm -disable-machine-licm (round 6): the machine-level loop-invariant code motion hoists the materialisation of every literal a loop body
#   uses (the polynomial coefficients of exp / log / sincos, masks, table strides) out of the propagation kernels' long loops, and the register
#   allocator then spills those long-lived values: k_rpkt 48 spilled VGPRs / 416 B of scratch -> 0 / 256 B (its stack arrays), none left inside the
#   opacity sum or the line walk; k_thermal 59 -> 16; k_tail 1109 -> 0; nltenebular k_thermal 253 -> 47 (tools/kernel_resources.py). A literal
#   re-made inside a loop is one v_mov; a spilled one is a scratch round trip.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-fno-slp-vectorize", "-mllvm", "-disable-machine-licm", "-ldl"]


PRESETS = ("classic", "kilonova_lte", "nltenebular", "christinenonthermal", "nltephotospheric", "nltewithoutnonthermal", "nltenebular_lineest", "kilonova_barnes", "kilonova_wollaeger", "kilonova_gammaproducts",
           "kilonova_gamma_barnes", "kilonova_gamma_wollaeger", "kilonova_gamma_guttman", "kilonova_gamma_grey", "classic_gamma_xcom",
           "kilonova_expopac", "classic_expopac_therm",
           # the option sets of the reference's CI (tests/setup_*.sh)
           "ci_kilonova", "ci_kilonova_barnes", "ci_kilonova_expopac", "ci_kilonova_xcom", "ci_nebular", "ci_nebular_limitbfest",
           "ci_nltephotospheric", "ci_classic_vpkt", "ci_classic_vpkt_expopac")  # options presets of include/artis_options.h (the reference's artisoptions_*.h)


def so_path(preset: str = "classic") -> str:
    return SO if preset == "classic" else os.path.join(HERE, f"libartis_amd_{preset}.so")


def needs_build(preset: str = "classic") -> bool:
    so = so_path(preset)
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "..", "include", h)
                                                        for h in ("artis_amd.h", "artis_options.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, extra_flags=(), preset: str = "classic") -> str:
    """One library per options preset, like one sn3d binary per artisoptions.h in the reference."""
    so = so_path(preset)
    if not force and not needs_build(preset):
        return so
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    pflags = [] if preset == "classic" else [f"-DARTIS_PRESET_{preset.upper()}", f'-DARTIS_PRESET_NAME="{preset}"']
    # one compiler run per library at a time (pytest-xdist workers, ranks of one node): the others wait, find it fresh and return;
    # the library appears under its name only when it is complete
    with open(so + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or needs_build(preset):
            tmp = f"{so}.{os.getpid()}.tmp"
            cmd = [hipcc, *FLAGS, *pflags, *extra_flags, "-o", tmp, os.path.join(CSRC, "artis_engine.hip")]
            subprocess.check_call(cmd)
            os.replace(tmp, so)
    return so


def build_all(force: bool = False, jobs: int = 0):
    """Every preset's library; the hipcc runs are independent, so a few go side by side (each takes ~12 s and ~1 GB)."""
    from concurrent.futures import ThreadPoolExecutor

    jobs = jobs or min(4, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        return list(pool.map(lambda p: build(force=force, preset=p), PRESETS))


if __name__ == "__main__":
    print(build_all(force=True))
