#!/usr/bin/env python3
"""What k_tail's waves spend their clocks on, bench workload of an options build (ARTIS_AMD_SO = a -DARTIS_PROFILE_TAIL build of it):
python tools/tail_profile.py [options] [packets]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artis_amd import abi, synth, engine
options = sys.argv[1] if len(sys.argv) > 1 else "nltenebular"
npk = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
model, cs, ts, aux = synth.build("w7", ncoord=50, options=options)
pk = synth.make_packets(model, aux, npk, seed_base=1281360349, kpkt_fraction=0.02, seed=99)
est = abi.estimators_for(model, options)
eng = engine.Engine(model, preset=options)
eng.set_cellstate(cs, ts)
eng.update_packets(pk.copy(), est)
s = np.asarray(est.stats).astype(float)
k = eng.last_kernel_ms_by_kind()
eng.close()
print(f"{options}: k_tail {k.get('k_tail')}")
names = ["slow path", "r-packet steps", "thermal (walks + k-packet steps)", "blackbody"]
tot = sum(s[42:46])
print("all waves:   " + ", ".join(f"{n} {16 * s[42 + i]:.3g} clocks ({s[42 + i] / max(tot, 1):.2f})" for i, n in enumerate(names)) + f"; r-packet iterations {s[46]:.4g} ({16 * s[43] / max(s[46], 1):.0f} clocks each), thermal iterations {s[47]:.4g} ({16 * s[44] / max(s[47], 1):.0f} clocks each)")
tl = sum(s[50:54])
print(f"waves > 2^25 clocks ({s[54]:.0f} of them): " + ", ".join(f"{n} {16 * s[50 + i]:.3g} ({s[50 + i] / max(tl, 1):.2f})" for i, n in enumerate(names)) + f"; per wave {16 * tl / max(s[54], 1):.3g} clocks, r-packet iterations {s[55] / max(s[54], 1):.0f}, thermal iterations {s[56] / max(s[54], 1):.0f}")
