"""Regenerate tests/golden/rng_reference.json from the REFERENCE's own random.h.

Needs /root/reference (this container only). `make -C oracle ref` compiles
oracle/ref_harness/ref_random_main.cc against /root/reference/random.h (the
reference source is included where it lies, never copied) into oracle/_ref/ref_random;
this script runs it for the seeds the reference's unit tests use (unittests.cc:93,120,
177,201,227,437) plus edge seeds, and stores raw generator outputs and rng_uniform()
float bit patterns (GPU_ON form, random.h:185-187).
"""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SEEDS = [20260729, 81102, 5501, 99001, 31415, 777, 0, 1, 4294967295, 1281360349]
COUNT = 48

subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
out = {"generator": "Xoshiro128PP seeded via SplitMix32 (reference random.h)", "count": COUNT, "streams": {}}
for seed in SEEDS:
    txt = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_random"), str(seed), str(COUNT)], text=True)
    raw, zbits = [], []
    for line in txt.strip().splitlines():
        a, b = line.split()
        raw.append(int(a, 16))
        zbits.append(int(b, 16))
    out["streams"][str(seed)] = {"raw_u32": raw, "uniform_float_bits": zbits}
with open(os.path.join(HERE, "rng_reference.json"), "w") as f:
    json.dump(out, f)
print("wrote rng_reference.json")
