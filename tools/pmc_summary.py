#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (sum over dispatches)."""
import collections, csv, glob, sys
dirs = sys.argv[1:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
ndisp = collections.defaultdict(int)
for d in dirs:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            k = k.replace("void ", "").split("<")[0].strip()  # template instantiations (k_thermal<false, 256>) by base name
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Dispatch_Id"] not in seen and row["Counter_Name"] in ("FETCH_SIZE", "SQ_WAVES"):
                seen.add(row["Dispatch_Id"])
                if row["Counter_Name"] == "FETCH_SIZE":
                    dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
                    ndisp[k] += 1
for k, v in sorted(agg.items()):
    if not any(x in k for x in ("k_thermal", "k_rpkt", "k_slow", "k_macroatom", "k_bfest_dense", "k_tail", "k_matrans", "k_mafilter")):
        continue
    print(f"{k}: {ndisp[k]} dispatches, {dur[k]:.4f} s (in the FETCH_SIZE pass)")
    for c, val in sorted(v.items()):
        print(f"   {c:36s} {val:.6g}")
    if "SQ_THREAD_CYCLES_VALU" in v and "SQ_ACTIVE_INST_VALU" in v and v["SQ_ACTIVE_INST_VALU"]:
        print(f"   -> VALU lane utilisation           {v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_ACTIVE_INST_VALU']):.3f}")
    if "TCC_HIT_sum" in v:
        print(f"   -> L2 hit rate                     {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f}")
    if "FETCH_SIZE" in v and dur[k]:
        print(f"   -> FETCH_SIZE KB/s (x2 per guide)  {v['FETCH_SIZE'] / dur[k]:.4g}  WRITE_SIZE KB/s {v.get('WRITE_SIZE', 0) / dur[k]:.4g}")

# per-launch HBM traffic of the propagation kernels for bench.py's roofline.traffic (FETCH_SIZE and WRITE_SIZE are in KB;
# FETCH_SIZE is doubled on gfx950, MI355X_MICROARCH.md "HBM")
import json, os
out = {}
for k, v in agg.items():
    if k in ("k_thermal", "k_rpkt", "k_bfest_dense") and "FETCH_SIZE" in v and "WRITE_SIZE" in v and ndisp[k]:
        out[k] = {"fetch_size_kb": v["FETCH_SIZE"], "write_size_kb": v["WRITE_SIZE"], "dispatches": ndisp[k],
                  "hbm_bytes_per_launch": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / ndisp[k],
                  "hbm_bytes_per_launch_fetch_undoubled": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / ndisp[k],
                  "seconds_in_fetch_pass": dur[k],
                  # the raw counters bench.py derives its limiter figures from (sums over the dispatches of one step)
                  "counters": {c: v[c] for c in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                                                 "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                                                 "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum",
                                                 "TCP_TOTAL_ACCESSES_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum",
                                                 "TCC_EA0_RDREQ_32B_sum", "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum", "TCC_REQ_sum") if c in v}}
if out and os.environ.get("PMC_TRAFFIC_JSON"):
    with open(os.environ["PMC_TRAFFIC_JSON"], "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.environ["PMC_TRAFFIC_JSON"])
