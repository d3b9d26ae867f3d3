"""Regenerate tests/golden/packet_layout_reference.json from the REFERENCE's own packet.h (GPU_ON build) and stats.h.

Needs /root/reference (this container only). `make -C oracle ref` compiles oracle/ref_harness/ref_packet_layout_main.cc
against /root/reference/packet.h (included where it lies, never copied); the harness prints offsetof/sizeof of every
member of struct Packet and the enum values the packet path uses.
"""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
txt = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_packet_layout")], text=True)
out = {"struct": "Packet (reference packet.h, -DGPU_ON)", "fields": {}, "enums": {}}
for line in txt.strip().splitlines():
    parts = line.split()
    if parts[0] == "sizeof":
        out["sizeof"] = int(parts[1])
    elif parts[0] == "enum":
        out["enums"][parts[1]] = int(parts[2])
    else:
        out["fields"][parts[0]] = {"offset": int(parts[1]), "size": int(parts[2])}
# event counters: stats::Counter of the reference's stats.h
out["stats_counters"] = {}
for line in subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_stats_enum")], text=True).strip().splitlines():
    name, val = line.split()
    out["stats_counters"][name] = int(val)
# physical constants of the reference's constants.h, as hex floats
out["constants"] = {}
for line in subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_constants")], text=True).strip().splitlines():
    name, val = line.split()
    out["constants"][name] = val
with open(os.path.join(HERE, "packet_layout_reference.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote packet_layout_reference.json:", len(out["fields"]), "fields, sizeof", out["sizeof"])
