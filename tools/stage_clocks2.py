#!/usr/bin/env python3
"""Wave clocks of the stages of one macro-atom transition (ARTIS_AMD_SO = a -DARTIS_PROFILE -DARTIS_PROFILE_MA build),
for k_thermal (ARTIS_AMD_REFILL=0) or k_thermal_q (=1): python tools/stage_clocks2.py [ncoord] [packets]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artis_amd import abi, synth, engine
model, cs, ts, aux = synth.build("w7", ncoord=int(sys.argv[1]) if len(sys.argv) > 1 else 50)
pk = synth.make_packets(model, aux, int(sys.argv[2]) if len(sys.argv) > 2 else 10000000, kpkt_fraction=0.02)
est = abi.estimators_for(model, "classic")
eng = engine.Engine(model)
eng.set_cellstate(cs, ts)
eng.update_packets(pk, est)
s = np.asarray(est.stats).astype(float)
wr = s[46]
print("REFILL", os.environ.get("ARTIS_AMD_REFILL"), "transitions %.4g wave-rounds %.4g lanes/round %.1f" % (s[abi.STAT_X_MA_JUMPS], wr, s[abi.STAT_X_MA_JUMPS] / wr))
tot = 0.
for name, slot in (("mark cost", 63), ("rates read (stage 1 wait)", 59), ("process drawn", 60), ("direction searched (stage 2)", 61), ("target read (stage 3)", 62)):
    print(f"  {name:32s} {s[slot] / wr:8.1f} clocks per wave-round")
    tot += s[slot] / wr
print("  sum of the marked stages %.0f" % tot)
print("  k_thermal wave clocks x1e9: pull+load %.1f | MA phase %.1f | kpkt phase %.1f | store+append %.1f  (wave slots x kernel clocks = %.1f)" % (16*s[42]/1e9, 16*s[43]/1e9, 16*s[44]/1e9, 16*s[45]/1e9, 4096*2.4e6*eng.last_kernel_breakdown()["thermal_ms"]/1e9))
print(" ", eng.last_kernel_breakdown())
