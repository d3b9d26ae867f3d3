import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from artis_amd import abi, synth, engine
from oracle import oracle_py
import hostemu_binding as emu
options = sys.argv[1] if len(sys.argv) > 1 else "ci_classic_vpkt"
kw = dict(kpkt_fraction=float(sys.argv[2]) if len(sys.argv) > 2 else 0.2)
model, cs, ts, aux = synth.build("small", ncoord=8, options=options, t_days=5.0)
pk0 = synth.make_packets(model, aux, 3000, **kw)
pa, pb = pk0.copy(), pk0.copy()
ea, eb = abi.estimators_for(model, options), abi.estimators_for(model, options)
oracle_py.update_packets(model, cs, ts, pa, ea, preset=options)
eng = engine.Engine(model, preset=options); eng.set_cellstate(cs, ts); eng.update_packets(pb, eb)
print("oracle", ea.stats[48:52], "engine", eb.stats[48:52])
va = ea.vspecpol.reshape(5, 12, 2500, 3); vb = eb.vspecpol.reshape(5, 12, 2500, 3)
print("I per comb oracle", va[..., 0].sum(axis=(0, 2)))
print("I per comb engine", vb[..., 0].sum(axis=(0, 2)))
print("Q per comb oracle", va[..., 1].sum(axis=(0, 2)))
print("Q per comb engine", vb[..., 1].sum(axis=(0, 2)))
d = np.abs(va - vb); i = np.unravel_index(d.argmax(), d.shape); print("worst at", i, va[i], vb[i], "max", va.max())
