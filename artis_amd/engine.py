"""Python host side over the C-ABI of libartis_amd.so (ctypes; no torch types cross the boundary).

Mirrors the reference's call sequence for one timestep:
    update_grid()      -> Engine.set_cellstate(cells, ts)       (cell cache populated on the GPU)
    update_packets()   -> Engine.update_packets(packets, est)   (host buffers)  or the device-resident
                          upload_packets / step / download_packets calls used by bench.py
    reduce_estimators  -> Engine.estimators_devptr() handed to an RCCL all-reduce by the caller
The library has no CPU path: creating an Engine without a HIP device raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi
from .build import SO, build, so_path


class EngineError(RuntimeError):
    pass


_LIBS = {}


def load_library(build_if_missing: bool = False, preset: str = "classic"):
    """Load the engine built for an options preset (include/artis_options.h). One library per preset."""
    if preset in _LIBS:
        return _LIBS[preset]
    so = so_path(preset)
    so = os.environ.get("ARTIS_AMD_SO" if preset == "classic" else f"ARTIS_AMD_SO_{preset.upper()}", so)  # A/B builds (tuning only)
    if not os.path.exists(so):
        if not build_if_missing:
            raise EngineError(f"{so} is missing: run `python -m artis_amd.build` (hipcc, gfx950). There is no CPU fallback.")
        build(preset=preset)
    L = C.CDLL(so)
    L.artis_amd_last_error.restype = C.c_char_p
    L.artis_amd_abi_version.restype = C.c_int
    L.artis_amd_sizeof_packet.restype = C.c_size_t
    L.artis_amd_engine_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.artis_amd_engine_destroy.argtypes = [C.c_void_p]
    L.artis_amd_engine_destroy.restype = None
    L.artis_amd_set_cellstate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.artis_amd_populate_cellcache.argtypes = [C.c_void_p, C.c_void_p]
    L.artis_amd_update_packets.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.artis_amd_packets_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.artis_amd_packets_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.artis_amd_packets_snapshot.argtypes = [C.c_void_p]
    L.artis_amd_packets_restore.argtypes = [C.c_void_p]
    L.artis_amd_update_packets_device.argtypes = [C.c_void_p, C.c_void_p]
    L.artis_amd_estimators_zero.argtypes = [C.c_void_p, C.c_void_p]
    L.artis_amd_estimators_download.argtypes = [C.c_void_p, C.c_void_p]
    L.artis_amd_estimators_devptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.artis_amd_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    if L.artis_amd_abi_version() != abi.ABI_VERSION:  # a stale or foreign build would read these ctypes structs with another layout
        raise EngineError(f"{so}: ABI version {L.artis_amd_abi_version()}, this package describes version {abi.ABI_VERSION}")
    assert L.artis_amd_sizeof_packet() == abi.PACKET_DTYPE.itemsize
    L.artis_amd_options_preset.restype = C.c_char_p
    assert L.artis_amd_options_preset().decode() == preset, (L.artis_amd_options_preset(), preset)
    _LIBS[preset] = L
    return L


EXPORTED_SYMBOLS = [
    "artis_amd_last_error", "artis_amd_abi_version", "artis_amd_sizeof_packet", "artis_amd_engine_create",
    "artis_amd_engine_destroy", "artis_amd_set_cellstate", "artis_amd_update_packets", "artis_amd_packets_upload",
    "artis_amd_packets_download", "artis_amd_packets_snapshot", "artis_amd_packets_restore",
    "artis_amd_update_packets_device", "artis_amd_estimators_zero", "artis_amd_estimators_download",
    "artis_amd_estimators_devptr", "artis_amd_last_kernel_ms", "artis_amd_debug_cellcache", "artis_amd_debug_visit_counts", "artis_amd_last_kernel_ms_by_kind",
    "artis_amd_populate_cellcache",
    "artis_amd_last_kernel_breakdown",
    "artis_amd_last_kernel_launches",
    "artis_amd_last_kernel_table",
    "artis_amd_options_preset",
    "artis_amd_allreduce_estimators", "artis_amd_comm_unique_id", "artis_amd_comm_init", "artis_amd_comm_count",
    "artis_amd_cache_tiles", "artis_amd_last_tiling", "artis_amd_last_tiling_fills", "artis_amd_last_tiling_parked", "artis_amd_last_pool_resets", "artis_amd_record_tiers", "artis_amd_last_thermal_variants", "artis_amd_last_pool_usage",
]


class Engine:
    def __init__(self, model: abi.Model, device: int = 0, preset: str = "classic"):
        self.L = load_library(preset=preset)
        self.model = model
        self.h = C.c_void_p()
        self._check(self.L.artis_amd_engine_create(C.cast(model.ref(), C.c_void_p), device, C.byref(self.h)))

    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(f"artis_amd error {rc}: {self.L.artis_amd_last_error().decode()}")

    def close(self):
        if self.h:
            self.L.artis_amd_engine_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_cellstate(self, cells: abi.CellState, ts: abi.Timestep):
        self._cells, self._ts = cells, ts
        self._check(self.L.artis_amd_set_cellstate(self.h, C.cast(cells.ref(), C.c_void_p), C.cast(ts.ref(), C.c_void_p)))

    def populate_cellcache(self, stream: int = 0):
        self._check(self.L.artis_amd_populate_cellcache(self.h, C.c_void_p(stream)))

    def update_packets(self, packets: np.ndarray, est: abi.Estimators):
        """Host-buffer form of the reference's update_packets() (update_packets.cc:530)."""
        self._check(self.L.artis_amd_update_packets(self.h, abi.packets_ptr(packets), len(packets), C.cast(est.ref(), C.c_void_p)))

    # device-resident form
    def upload_packets(self, packets: np.ndarray):
        self._check(self.L.artis_amd_packets_upload(self.h, abi.packets_ptr(packets), len(packets)))

    def download_packets(self, packets: np.ndarray):
        self._check(self.L.artis_amd_packets_download(self.h, abi.packets_ptr(packets), len(packets)))

    def snapshot(self):
        self._check(self.L.artis_amd_packets_snapshot(self.h))

    def restore(self):
        self._check(self.L.artis_amd_packets_restore(self.h))

    def step(self, stream: int = 0):
        self._check(self.L.artis_amd_update_packets_device(self.h, C.c_void_p(stream)))

    def zero_estimators(self, stream: int = 0):
        self._check(self.L.artis_amd_estimators_zero(self.h, C.c_void_p(stream)))

    def download_estimators(self, est: abi.Estimators):
        self._check(self.L.artis_amd_estimators_download(self.h, C.cast(est.ref(), C.c_void_p)))

    def estimators_devptr(self):
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.L.artis_amd_estimators_devptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def debug_cellcache(self, c: int) -> dict:
        d = self.model.d
        out = {
            "levelpops": np.zeros(d["nlevels"]), "maprocessrates": np.zeros(d["nlevels"] * 9),
            "matrans": np.zeros(max(d["nmatransblock"], 1)), "allcont_nnlevel": np.zeros(max(d["nbfcontinua"], 1)),
            "allcont_departure": np.zeros(max(d["nbfcontinua"], 1)), "allcont_edgepart": np.zeros(max(d["nbfcontinua"], 1)),
            "allcont_keepbits": np.zeros((d["nbfcontinua"] + 63) // 64 + 1, dtype=np.uint64),
            "corrphotoioncoeff": np.zeros(max(d["nphixstargets_total"], 1)),
            "cooling_contrib": np.zeros(max(d["ncoolingterms"], 1)), "ion_cooling_contribs": np.zeros(d["nions"]),
        }
        chi = C.c_double(0.0)
        args = [self.h, C.c_int(c)] + [v.ctypes.data_as(C.c_void_p) for v in out.values()] + [C.byref(chi)]
        self._check(self.L.artis_amd_debug_cellcache(*args))
        out["chi_ff_nnionpart"] = chi.value
        return out

    def cache_tiles(self):
        """(number of cell-cache tiles, cells per tile, cache bytes per cell)"""
        nt, cells, bpc = C.c_int32(), C.c_int64(), C.c_int64()
        self.L.artis_amd_cache_tiles.argtypes = [C.c_void_p] * 4
        self._check(self.L.artis_amd_cache_tiles(self.h, C.byref(nt), C.byref(cells), C.byref(bpc)))
        return nt.value, cells.value, bpc.value

    def record_tiers(self):
        """(share of every ion's levels with a static macro-atom record, cold levels, pool slots per resident cell): what the engine chose
        from its cache budget at creation, or was given (ARTIS_AMD_MA_HOTFRAC / _POOLFRAC)"""
        h, n, p = C.c_double(), C.c_int32(), C.c_int64()
        self.L.artis_amd_record_tiers.argtypes = [C.c_void_p] * 4
        self._check(self.L.artis_amd_record_tiers(self.h, C.byref(h), C.byref(n), C.byref(p)))
        return {"hot_fraction": h.value, "ncold": n.value, "pool_slots": p.value}

    THERMAL_PLAIN, THERMAL_LDS_TABLES, THERMAL_LDS_LEVELPACK, THERMAL_REFILL, THERMAL_COLD, THERMAL_TAIL = 1, 2, 4, 8, 16, 32

    def last_thermal_variants(self) -> int:
        """mask of the thermal-kernel forms the last step() launched (include/artis_amd.h ARTIS_AMD_THERMAL_*)"""
        m = C.c_int32()
        self.L.artis_amd_last_thermal_variants.argtypes = [C.c_void_p] * 2
        self._check(self.L.artis_amd_last_thermal_variants(self.h, C.byref(m)))
        return int(m.value)

    def last_tiling(self):
        """sweeps over the cache tiles, tile fills, their summed ms and the packets listed in the last step()"""
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_double(), C.c_int64()
        self.L.artis_amd_last_tiling.argtypes = [C.c_void_p] * 5
        self._check(self.L.artis_amd_last_tiling(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        e, f = C.c_int64(), C.c_int64()
        self.L.artis_amd_last_tiling_fills.argtypes = [C.c_void_p] * 3
        self._check(self.L.artis_amd_last_tiling_fills(self.h, C.byref(e), C.byref(f)))
        g = C.c_int64()
        self.L.artis_amd_last_tiling_parked.argtypes = [C.c_void_p] * 2
        self._check(self.L.artis_amd_last_tiling_parked(self.h, C.byref(g)))
        r = C.c_int64()
        self.L.artis_amd_last_pool_resets.argtypes = [C.c_void_p] * 2
        self._check(self.L.artis_amd_last_pool_resets(self.h, C.byref(r)))
        pu, pc = C.c_int64(), C.c_int64()
        self.L.artis_amd_last_pool_usage.argtypes = [C.c_void_p] * 3
        self._check(self.L.artis_amd_last_pool_usage(self.h, C.byref(pu), C.byref(pc)))
        return {"sweeps": a.value, "tile_fills": b.value, "fill_ms": c.value, "listed": d.value, "sparse_fills": e.value,
                "cells_filled": f.value, "parked": g.value, "pool_resets": r.value, "pool_units_used": pu.value, "pool_units": pc.value}

    # estimator reduction in the C++ host layer (RCCL)
    COMM_ID_BYTES = 128

    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(self.COMM_ID_BYTES)
        self.L.artis_amd_comm_unique_id.argtypes = [C.c_void_p]
        self._check(self.L.artis_amd_comm_unique_id(buf))
        return buf.raw

    def comm_count(self) -> int:
        """ranks of the engine's communicator as RCCL reports them (ncclCommCount)"""
        n = C.c_int(0)
        self.L.artis_amd_comm_count.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        self._check(self.L.artis_amd_comm_count(self.h, None, C.byref(n)))
        return int(n.value)

    def comm_init(self, nranks: int, rank: int, id_bytes: bytes):
        assert len(id_bytes) == self.COMM_ID_BYTES
        self.L.artis_amd_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        self._check(self.L.artis_amd_comm_init(self.h, nranks, rank, id_bytes))

    def allreduce_estimators(self, stream: int = 0, comm: int = 0):
        self.L.artis_amd_allreduce_estimators.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._check(self.L.artis_amd_allreduce_estimators(self.h, C.c_void_p(comm), C.c_void_p(stream)))

    def last_kernel_breakdown(self):
        a, b, c, d = C.c_double(), C.c_int64(), C.c_double(), C.c_int64()
        self.L.artis_amd_last_kernel_breakdown.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        self._check(self.L.artis_amd_last_kernel_breakdown(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        e, f = C.c_int64(), C.c_int64()
        self.L.artis_amd_last_kernel_launches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._check(self.L.artis_amd_last_kernel_launches(self.h, C.byref(e), C.byref(f)))
        return {"rpkt_ms": a.value, "rpkt_threads": b.value, "rpkt_launches": e.value, "thermal_ms": c.value,
                "thermal_threads": d.value, "thermal_launches": f.value}

    def last_kernel_table(self):
        ms, nl, npk = (C.c_double * 4)(), (C.c_int64 * 4)(), (C.c_int64 * 4)()
        self.L.artis_amd_last_kernel_table.argtypes = [C.c_void_p] * 4
        self._check(self.L.artis_amd_last_kernel_table(self.h, ms, nl, npk))
        names = ["k_rpkt", "k_ma", "k_kpkt", "k_slow"]
        return {n: {"ms": ms[i], "launches": nl[i], "packets": npk[i]} for i, n in enumerate(names)}

    def last_kernel_ms_by_kind(self):
        ms, nl = (C.c_double * 8)(), (C.c_int64 * 8)()
        self.L.artis_amd_last_kernel_ms_by_kind.argtypes = [C.c_void_p] * 3
        self._check(self.L.artis_amd_last_kernel_ms_by_kind(self.h, ms, nl))
        names = ["k_rpkt", "k_thermal", "k_slow", "k_gamma", "k_blackbody", "k_tail", "tile_fills"]
        return {n: {"ms": round(ms[i], 3), "launches": nl[i]} for i, n in enumerate(names)}

    def last_kernel_ms(self):
        ms, n = C.c_double(), C.c_int64()
        self._check(self.L.artis_amd_last_kernel_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value
