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
with open(os.path.join(HERE, "options_reference.json"), "w") as f:
    json.dump({"source": "reference artisoptions_<preset>.h via oracle/ref_harness/ref_options_main.cc", "presets": out}, f, indent=1)
print("wrote options_reference.json:", {k: len(v) for k, v in out.items()})
