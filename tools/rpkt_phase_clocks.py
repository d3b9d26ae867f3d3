#!/usr/bin/env python3
"""Wave clocks of the phases of do_rpkt_step() in k_rpkt (a -DARTIS_PROFILE build given as ARTIS_AMD_SO[_<PRESET>]):
   python tools/rpkt_phase_clocks.py [options preset]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artis_amd import abi, synth, engine
opt = sys.argv[1] if len(sys.argv) > 1 else "classic"
model, cs, ts, aux = synth.build("w7", ncoord=50, options=opt)
pk = synth.make_packets(model, aux, 10000000, kpkt_fraction=0.02)
est = abi.estimators_for(model, opt)
eng = engine.Engine(model, preset=opt)
eng.set_cellstate(cs, ts)
eng.update_packets(pk, est)
s = np.asarray(est.stats).astype(float)
names = {48: "boundary distance", 49: "continuum opacity", 50: "line walk (possible event)", 51: "move + estimators", 52: "event", 53: "outside do_rpkt_step (pull, load, store, append)"}
tot = sum(s[k] for k in names)
print(opt, "rpkt steps %.4g chi evals %.4g continua visited %.4g (%.1f per eval) lines visited %.4g (%.1f per step)" % (
    s[abi.STAT_X_RPKT_STEPS], s[38], s[39], s[39] / max(s[38], 1), s[36], s[36] / max(s[abi.STAT_X_RPKT_STEPS], 1)))
for k, n in names.items():
    print(f"  {n:48s} {100 * s[k] / tot:5.1f} %")
print(" ", eng.last_kernel_breakdown())
