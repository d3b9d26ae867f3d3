"""ctypes binding of the test-only host emulation of the kernel bodies (tests/hostemu)."""
import ctypes as C
import fcntl
import os
import subprocess

import numpy as np

from artis_amd import abi

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostemu")
_LIBS = {}


def lib(preset: str = "classic"):
    if preset not in _LIBS:
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:  # (pytest-xdist workers: one make at a time)
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(_HERE, "libartis_hostemu.so" if preset == "classic" else f"libartis_hostemu_{preset}.so"))
        L.artis_emu_update_packets.restype = C.c_int
        L.artis_emu_update_packets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        L.artis_emu_cellcache.restype = C.c_int
        _LIBS[preset] = L
    return _LIBS[preset]


def update_packets(model, cells, ts, packets, est, budget=4, preset="classic"):
    rc = lib(preset).artis_emu_update_packets(C.cast(model.ref(), C.c_void_p), C.cast(cells.ref(), C.c_void_p),
                                        C.cast(ts.ref(), C.c_void_p), abi.packets_ptr(packets), len(packets),
                                        C.cast(est.ref(), C.c_void_p), budget)
    if rc != 0:
        raise RuntimeError(f"kernel-body emulation raised error flag {rc}")


def cellcache(model, cells, ts, c):
    d = model.d
    out = {
        "levelpops": np.zeros(d["nlevels"]), "maprocessrates": np.zeros(d["nlevels"] * 9),
        "matrans": np.zeros(max(d["nmatransblock"], 1)), "allcont_nnlevel": np.zeros(max(d["nbfcontinua"], 1)),
        "allcont_departure": np.zeros(max(d["nbfcontinua"], 1)), "allcont_edgepart": np.zeros(max(d["nbfcontinua"], 1)),
        "allcont_keepbits": np.zeros((d["nbfcontinua"] + 63) // 64 + 1, dtype=np.uint64),
        "corrphotoioncoeff": np.zeros(max(d["nphixstargets_total"], 1)), "cooling_contrib": np.zeros(max(d["ncoolingterms"], 1)),
        "ion_cooling_contribs": np.zeros(d["nions"]),
    }
    chi = C.c_double(0.0)
    args = [C.cast(model.ref(), C.c_void_p), C.cast(cells.ref(), C.c_void_p), C.cast(ts.ref(), C.c_void_p), C.c_int(c)] + \
           [v.ctypes.data_as(C.c_void_p) for v in out.values()] + [C.byref(chi)]
    rc = lib().artis_emu_cellcache(*args)
    if rc != 0:
        raise RuntimeError("emulated populate raised an error flag")
    out["chi_ff_nnionpart"] = chi.value
    return out
