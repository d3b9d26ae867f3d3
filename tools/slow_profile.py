#!/usr/bin/env python3
"""The slow-path kernel's visits by kind of pending action, bench workload (ARTIS_AMD_SO = a -DARTIS_PROFILE_SLOW build): python tools/slow_profile.py [preset] [packets]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artis_amd import abi, synth, engine
preset = sys.argv[1] if len(sys.argv) > 1 else "cd23like"
npk = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
model, cs, ts, aux = synth.build(preset, ncoord=50)
pk = synth.make_packets(model, aux, npk, seed_base=1281360349, kpkt_fraction=0.02, seed=99)
est = abi.estimators_for(model, "classic")
eng = engine.Engine(model)
eng.set_cellstate(cs, ts)
eng.update_packets(pk.copy(), est)
s = np.asarray(est.stats).astype(float)
k = eng.last_kernel_ms_by_kind()
t = eng.last_tiling()
eng.close()
names = ["none", "1", "MA_ACTION", "KPKT_FB", "MA_SEARCH", "MA_RADSEARCH", "KPKT_COLLEXC", "MA_FILL", "RPKT_ABSORB", ">= 9"]
print(f"{preset}: k_slow {k.get('k_slow')}, pool resets {t.get('pool_resets')}; k-packet steps {s[abi.STAT_X_KPKT_STEPS]:.4g}, transitions {s[abi.STAT_X_MA_JUMPS]:.4g}")
print("slow-path visits by pending action: " + ", ".join(f"{n} {s[48 + i]:.4g}" for i, n in enumerate(names) if s[48 + i] > 0))
