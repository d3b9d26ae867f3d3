import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from artis_amd import abi, synth, engine
model, cs, ts, aux = synth.build("w7", ncoord=50)
pk = synth.make_packets(model, aux, 2000000, kpkt_fraction=0.02)
est = abi.estimators_for(model, "classic")
eng = engine.Engine(model); eng.set_cellstate(cs, ts); eng.update_packets(pk, est)
s = np.asarray(est.stats).astype(float)
j = s[abi.STAT_X_MA_JUMPS]
print("transitions %.4g internal %.4g: nsel==1 %.3f ti<2 %.3f ti<4 %.3f ti<8 %.3f beyond %.3f | down fraction %.3f" % (j, s[53:58].sum(), *(s[53:58]/s[53:58].sum()), s[58]/s[53:58].sum()))
