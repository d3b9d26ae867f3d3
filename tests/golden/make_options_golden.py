"""Regenerate tests/golden/options_reference.json from the REFERENCE's own artisoptions_*.h.

Needs /root/reference (this container only). `make -C oracle ref` compiles oracle/ref_harness/ref_options_main.cc once
per options file (all six artisoptions_*.h; included where they lie, never copied) into
oracle/_ref/ref_options_<name>; each prints the compile-time options the packet path reads. The test compares them with
what include/artis_options.h gives for the preset of the same name (tests/options_printer.c).
"""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
out = {}
# preset of include/artis_options.h -> options file of the reference
FILES = {"classic": "classic", "kilonova_lte": "kilonova_lte", "nltenebular": "nltenebular",
         "christinenonthermal": "christinenonthermal", "nltephotospheric": "nltephotospheric_dynamic_ion_range",
         "nltewithoutnonthermal": "nltewithoutnonthermal"}
for name, fname in FILES.items():
    txt = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", f"ref_options_{fname}")], text=True)
    out[name] = dict(line.split() for line in txt.strip().splitlines())
# the option sets of the reference's CI: the options file each tests/setup_<script>.sh makes (its own sed lines applied to
# a scratch copy under oracle/_ref/ci/<script>/, oracle/Makefile) against the ci_* presets of include/artis_options.h
import sys
sys.path.insert(0, ROOT)
from artis_amd import abi  # noqa: E402
for name, (script, _, _) in abi.CI_PRESETS.items():
    sname = script[len("setup_"):-len(".sh")]
    txt = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", f"ref_options_ci_{sname}")], text=True)
    out[name] = dict(line.split() for line in txt.strip().splitlines())
# setup_kilonova_1d.sh makes the same packet-path options as setup_kilonova_2d.sh (one preset for both)
k1 = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_options_ci_kilonova_1d")], text=True)
assert dict(line.split() for line in k1.strip().splitlines()) == out["ci_kilonova"]
with open(os.path.join(HERE, "options_reference.json"), "w") as f:
    json.dump({"source": "reference artisoptions_<preset>.h via oracle/ref_harness/ref_options_main.cc", "presets": out}, f, indent=1)
print("wrote options_reference.json:", {k: len(v) for k, v in out.items()})
