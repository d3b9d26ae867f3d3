#!/usr/bin/env python3
"""Wave-level iteration counts of k_thermal's phases (ARTIS_AMD_SO = a -DARTIS_PROFILE build): python tools/thermal_counts.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artis_amd import abi, synth, engine
model, cs, ts, aux = synth.build("w7", ncoord=50)
pk = synth.make_packets(model, aux, 10000000, kpkt_fraction=0.02)
est = abi.estimators_for(model, "classic")
eng = engine.Engine(model)
eng.set_cellstate(cs, ts)
eng.update_packets(pk, est)
s = np.asarray(est.stats).astype(float)
print("transitions %.4g  k-packet steps %.4g  thermal visits %.4g" % (s[abi.STAT_X_MA_JUMPS], s[abi.STAT_X_KPKT_STEPS], eng.last_kernel_breakdown()["thermal_threads"]))
print("wave-level: transition rounds %.4g (%.1f lanes)  k-packet phases %.4g (%.1f lanes)" % (s[46], s[abi.STAT_X_MA_JUMPS] / s[46], s[47], s[abi.STAT_X_KPKT_STEPS] / max(s[47], 1)))
print("wave-level rounds of the collisional-excitation scan of the k-packet step (8 reads each): %.4g" % s[58])
print("wave clocks x1e9: pull+load %.1f | MA phase %.1f | kpkt phase %.1f | store+append %.1f" % (16*s[42]/1e9, 16*s[43]/1e9, 16*s[44]/1e9, 16*s[45]/1e9))
print("k-packet step, wave clocks x1e9: up to the ion drawn %.1f | up to the cooling term drawn %.1f | the term's process %.1f" % (16*s[56]/1e9, 16*s[57]/1e9, 16*s[41]/1e9))
print(eng.last_kernel_breakdown())
