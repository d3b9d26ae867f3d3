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
    assert lib.artis_amd_abi_version() == abi.ABI_VERSION == 6


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


def test_packet_struct_matches_reference_layout():
    """artis_packet (include/artis_amd.h, mirrored by abi.PACKET_DTYPE) against offsetof/sizeof of the reference's own
    struct Packet compiled with -DGPU_ON (tests/golden/packet_layout_reference.json, made by
    tests/golden/make_packet_layout_golden.py from /root/reference/packet.h): a reference build can hand its
    std::span<Packet> to artis_amd_update_packets() as it is."""
    import json
    from artis_amd import abi
    with open(os.path.join(os.path.dirname(__file__), "golden", "packet_layout_reference.json")) as f:
        gold = json.load(f)
    dt = abi.PACKET_DTYPE
    assert dt.itemsize == gold["sizeof"] == 256
    assert set(dt.names) == set(gold["fields"])
    for name, ref in gold["fields"].items():
        field_dtype, offset = dt.fields[name][:2]
        assert offset == ref["offset"], name
        assert field_dtype.itemsize == ref["size"], name
    # the C header itself (not only the numpy mirror): compile-time layout of struct artis_packet
    import subprocess, tempfile
    fields = list(gold["fields"])
    src = '#include <stddef.h>\n#include <stdio.h>\n#include "artis_amd.h"\nint main(void){' + "".join(
        f'printf("{n} %zu\\n", offsetof(artis_packet, {n}));' for n in fields) + 'printf("sizeof %zu\\n", sizeof(artis_packet));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "t.c"), "w") as f:
            f.write(src)
        inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
        subprocess.check_call(["gcc", "-I", inc, "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        got = dict(line.split() for line in subprocess.check_output([os.path.join(d, "t")], text=True).strip().splitlines())
    assert int(got["sizeof"]) == gold["sizeof"]
    for name, ref in gold["fields"].items():
        assert int(got[name]) == ref["offset"], name
    e = gold["enums"]
    assert (abi.TYPE_RPKT, abi.TYPE_KPKT, abi.TYPE_PRE_KPKT, abi.TYPE_ESCAPE) == (e["TYPE_RPKT"], e["TYPE_KPKT"], e["TYPE_PRE_KPKT"], e["TYPE_ESCAPE"])
    assert (abi.EMTYPE_NOTSET, abi.EMTYPE_FREEFREE) == (e["EMTYPE_NOTSET"], e["EMTYPE_FREEFREE"])


def test_event_counters_match_reference_enum():
    """ARTIS_STAT_* (include/artis_amd.h) and abi.STAT_NAMES against stats::Counter of the reference's stats.h compiled in
    place (tests/golden/packet_layout_reference.json): artis_estimators.stats is indexed exactly like the reference's
    event counters."""
    import json
    from artis_amd import abi
    with open(os.path.join(os.path.dirname(__file__), "golden", "packet_layout_reference.json")) as f:
        gold = json.load(f)["stats_counters"]
    assert gold["COUNT"] == abi.STAT_COUNT == 34
    for name, val in gold.items():
        if name != "COUNT":
            assert abi.STAT_NAMES[val] == name, (name, val)
    hdr = open(os.path.join(ROOT, "include", "artis_amd.h")).read()
    for name, val in gold.items():
        cname = name.replace("_STAT_", "_", 1)  # MA_STAT_ACTIVATION_BB -> ARTIS_STAT_MA_ACTIVATION_BB
        m = re.search(r"ARTIS_STAT_%s = (\d+)" % re.escape(cname), hdr)
        assert m and int(m.group(1)) == val, name


def test_one_library_per_options_preset():
    """Like the reference (one sn3d per artisoptions.h), every options preset of include/artis_options.h is its own
    library; each reports the preset it was compiled with and exports the same C-ABI."""
    from artis_amd.build import PRESETS, build as build_preset, so_path
    from artis_amd.engine import EXPORTED_SYMBOLS
    assert set(PRESETS) == {"classic", "kilonova_lte", "nltenebular", "christinenonthermal", "nltephotospheric", "nltewithoutnonthermal",
                            "nltenebular_lineest", "kilonova_barnes", "kilonova_wollaeger", "kilonova_gammaproducts", "kilonova_gamma_barnes",
                            "kilonova_gamma_wollaeger", "kilonova_gamma_guttman", "kilonova_gamma_grey", "classic_gamma_xcom", "kilonova_expopac",
                            "classic_expopac_therm", *abi.CI_PRESETS}
    for preset in PRESETS:
        L = C.CDLL(build_preset(preset=preset))
        assert os.path.samefile(build_preset(preset=preset), so_path(preset))
        L.artis_amd_options_preset.restype = C.c_char_p
        assert L.artis_amd_options_preset().decode() == preset
        for name in EXPORTED_SYMBOLS:
            assert hasattr(L, name), (preset, name)
