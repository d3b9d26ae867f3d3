"""ctypes binding of the CPU oracle (oracle/libartis_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module. The artis_amd package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from artis_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
PRESETS = ("classic", "kilonova_lte", "nltenebular", "christinenonthermal", "nltephotospheric", "nltewithoutnonthermal", "nltenebular_lineest", "kilonova_barnes", "kilonova_wollaeger", "kilonova_gammaproducts",
           "kilonova_gamma_barnes", "kilonova_gamma_wollaeger", "kilonova_gamma_guttman", "kilonova_gamma_grey", "classic_gamma_xcom",
           "kilonova_expopac", "classic_expopac_therm")  # options presets of include/artis_options.h the oracle is built for


def _soname(preset: str) -> str:
    return "libartis_oracle.so" if preset == "classic" else f"libartis_oracle_{preset}.so"


def build(force: bool = False, preset: str = "classic") -> str:
    so = os.path.join(_HERE, _soname(preset))
    deps = [os.path.join(_HERE, "artis_oracle.c"), os.path.join(_HERE, "..", "include", "artis_options.h"),
            os.path.join(_HERE, "..", "include", "artis_amd.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(["make", "-C", _HERE, _soname(preset)], stdout=subprocess.DEVNULL)
    return so


# bench.py's cpu_baseline leg only: the same restatement compiled the way the reference's Makefile compiles sn3d by default
# (-O3 -march=native -flto, FASTMATH on: /root/reference/Makefile:38, :236-251) instead of with the checker's -O2 -ffp-contract=off.
# -march=native: built on the machine that runs it, into a scratch directory, never shipped. Its results are NOT the checker's
# (contraction and re-association change the last bits): it is timed, never compared.
FAST_CFLAGS = ["-O3", "-march=native", "-flto=auto", "-ffast-math", "-funsafe-math-optimizations", "-fno-finite-math-only",
               "-std=gnu11", "-fPIC", "-w"]


def build_fast(preset: str = "classic") -> str:
    import tempfile

    d = os.path.join(tempfile.gettempdir(), f"artis_oracle_fast_{os.getuid()}")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, _soname(preset))
    src = os.path.join(_HERE, "artis_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        pflags = [] if preset == "classic" else [f"-DARTIS_PRESET_{preset.upper()}"]
        tmp = f"{so}.{os.getpid()}.tmp"
        subprocess.check_call([os.environ.get("CC", "gcc"), *FAST_CFLAGS, *pflags, "-shared", "-o", tmp, src, "-lm"])
        os.replace(tmp, so)
    return so


def lib(preset: str = "classic", fast: bool = False):
    if fast:
        key = ("fast", preset)
        if key not in _LIBS:
            L = C.CDLL(build_fast(preset))
            L.artis_oracle_update_packets.restype = C.c_int
            L.artis_oracle_update_packets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
            L.artis_oracle_last_populate_seconds.restype = C.c_double
            _LIBS[key] = L
        return _LIBS[key]
    if preset not in _LIBS:
        L = C.CDLL(build(preset=preset))
        L.artis_oracle_update_packets.restype = C.c_int
        L.artis_oracle_update_packets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.artis_oracle_cellcache.restype = C.c_int
        L.artis_oracle_rng_next.restype = C.c_uint32
        L.artis_oracle_rng_uniform.restype = C.c_float
        L.artis_oracle_doppler.restype = C.c_double
        L.artis_oracle_get_linedistance.restype = C.c_double
        L.artis_oracle_get_linedistance.argtypes = [C.c_double, C.c_double, C.c_double]
        L.artis_oracle_rad_deexcitation_ratecoeff.restype = C.c_double
        L.artis_oracle_rad_deexcitation_ratecoeff.argtypes = [C.c_double, C.c_float, C.c_double, C.c_double, C.c_double,
                                                              C.c_double, C.c_double]
        L.artis_oracle_phixs_fromtable.restype = C.c_float
        L.artis_oracle_phixs_fromtable.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        L.artis_oracle_planck.restype = C.c_double
        L.artis_oracle_planck.argtypes = [C.c_double, C.c_double]
        L.artis_oracle_closest_transition.restype = C.c_int
        L.artis_oracle_closest_transition.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int]
        L.artis_oracle_sizeof_packet.restype = C.c_size_t
        L.artis_oracle_seed_packets.argtypes = [C.c_void_p, C.c_int64, C.c_uint32]
        _LIBS[preset] = L
    return _LIBS[preset]


def update_packets(model: abi.Model, cells: abi.CellState, ts: abi.Timestep, packets: np.ndarray,
                   est: abi.Estimators, preset: str = "classic", fast: bool = False) -> None:
    rc = lib(preset, fast).artis_oracle_update_packets(C.cast(model.ref(), C.c_void_p), C.cast(cells.ref(), C.c_void_p),
                                           C.cast(ts.ref(), C.c_void_p), abi.packets_ptr(packets), len(packets),
                                           C.cast(est.ref(), C.c_void_p))
    if rc != 0:
        raise RuntimeError("oracle reported an internal assertion failure")


def cellcache(model: abi.Model, cells: abi.CellState, ts: abi.Timestep, nonemptymgi: int) -> dict:
    d = model.d
    out = {
        "levelpops": np.zeros(d["nlevels"]),
        "maprocessrates": np.zeros(d["nlevels"] * 9),
        "matrans": np.zeros(max(d["nmatransblock"], 1)),
        "allcont_nnlevel": np.zeros(max(d["nbfcontinua"], 1)),
        "allcont_departure": np.zeros(max(d["nbfcontinua"], 1)),
        "allcont_edgepart": np.zeros(max(d["nbfcontinua"], 1)),
        "allcont_keepbits": np.zeros((d["nbfcontinua"] + 63) // 64 + 1, dtype=np.uint64),
        "corrphotoioncoeff": np.zeros(max(d["nphixstargets_total"], 1)),
        "cooling_contrib": np.zeros(max(d["ncoolingterms"], 1)),
        "ion_cooling_contribs": np.zeros(d["nions"]),
    }
    chi = C.c_double(0.0)
    args = [C.cast(model.ref(), C.c_void_p), C.cast(cells.ref(), C.c_void_p), C.cast(ts.ref(), C.c_void_p),
            C.c_int(nonemptymgi)] + [v.ctypes.data_as(C.c_void_p) for v in out.values()] + [C.byref(chi)]
    rc = lib().artis_oracle_cellcache(*args)
    if rc != 0:
        raise RuntimeError("oracle cellcache failed")
    out["chi_ff_nnionpart"] = chi.value
    return out
