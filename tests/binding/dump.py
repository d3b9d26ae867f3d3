"""Writes the flat arrays the reference-side binding (tests/binding/update_packets_amd.cc) reads in place of the reference's
globals:: / grid:: storage: one record per array or scalar, 64-byte name | int64 count | int64 bytes | data."""
import numpy as np

from artis_amd import abi


def _rec(f, name: str, arr: np.ndarray, count: int | None = None):
    b = np.ascontiguousarray(arr).tobytes()
    f.write(name.encode().ljust(64, b"\0"))
    f.write(np.array([len(arr) if count is None else count, len(b)], dtype=np.int64).tobytes())
    f.write(b)


def write_dump(path: str, model: abi.Model, cs: abi.CellState, ts: abi.Timestep, packets: np.ndarray) -> None:
    with open(path, "wb") as f:
        for name, ctype, npdt in abi._MODEL_FIELDS:
            if name not in model.d:
                continue  # optional and absent: the binding leaves it NULL / 0
            v = model.d[name]
            if npdt is None:
                _rec(f, name, np.array([v], dtype=np.float64 if ctype is abi.C.c_double else np.int32))
            elif npdt == "i3":
                _rec(f, "ncoordgrid", np.asarray(v, dtype=np.int32))
            elif npdt == "p3":
                for k in range(3):
                    _rec(f, f"coord_pos_min_tmin{k}", np.asarray(v[k], dtype=np.float64))
            else:
                _rec(f, name, v)
        for name, ctype, npdt in abi._CELL_FIELDS:
            if name not in cs.d:
                continue
            v = cs.d[name]
            _rec(f, "cell." + name, np.array([v], dtype=np.int32) if npdt is None else v)
        for name in ("nts", "start", "width", "mid", "max_path_step"):
            v = getattr(ts.c, name)
            _rec(f, "ts." + name, np.array([v], dtype=np.int32 if name == "nts" else np.float64))
        _rec(f, "packets", packets.view(np.uint8).reshape(-1), count=len(packets))


def read_output(path: str, model: abi.Model, npackets: int):
    n, g = model["npts_nonempty"], max(model["nbfcontinua_ground"], 1)
    with open(path, "rb") as f:
        pk = np.frombuffer(f.read(abi.PACKET_DTYPE.itemsize * npackets), dtype=abi.PACKET_DTYPE).copy()
        sizes = [("J", n), ("nuJ", n), ("ffheatingestimator", n), ("colheatingestimator", n), ("gammaestimator", n * g),
                 ("bfheatingestimator", n * g), ("dep_estimator_gamma", n), ("scalars", abi.NSCALARS), ("dep_estimator_electron", n),
                 ("dep_estimator_positron", n), ("dep_estimator_alpha", n)]
        est = {k: np.frombuffer(f.read(8 * c), dtype=np.float64).copy() for k, c in sizes}
        stats = np.frombuffer(f.read(8 * abi.NSTATS), dtype=np.int64).copy()
    return pk, est, stats
