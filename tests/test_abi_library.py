"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol
that include/artis_amd.h declares, and refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from artis_amd import abi, synth
from artis_amd.build import SO, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build()
    return C.CDLL(SO)


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "artis_amd.h")).read()
    declared = set(re.findall(r"\b(artis_amd_[a-z_0-9]+)\s*\(", hdr))
    declared.discard("artis_amd_engine")  # the opaque struct name
    assert len(declared) >= 17
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/artis_amd.h but not exported"
    from artis_amd.engine import EXPORTED_SYMBOLS
    assert set(EXPORTED_SYMBOLS) == declared


def test_packet_layout_matches_header(lib):
    lib.artis_amd_sizeof_packet.restype = C.c_size_t
    assert lib.artis_amd_sizeof_packet() == abi.PACKET_DTYPE.itemsize == 256
    assert lib.artis_amd_abi_version() == 1


def test_no_cpu_fallback(lib):
    """On a machine without a GPU the engine must fail loudly instead of computing on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    model, cs, ts, aux = synth.build("tiny", ncoord=4)
    h = C.c_void_p()
    lib.artis_amd_engine_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    rc = lib.artis_amd_engine_create(C.cast(model.ref(), C.c_void_p), 0, C.byref(h))
    assert rc != 0 and not h.value
    lib.artis_amd_last_error.restype = C.c_char_p
    assert b"no" in lib.artis_amd_last_error().lower()


def test_product_never_imports_oracle():
    """The package may not import, link or execute anything under oracle/ or tests/hostemu."""
    pkg = os.path.join(ROOT, "artis_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cc")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_py" not in txt and "libartis_oracle" not in txt and "hostemu" not in txt.replace(
                    "host-emulation", "").replace("tests/hostemu", ""), f
