import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from artis_amd import abi, synth, engine
from oracle import oracle_py
import parity
model, cs, ts, aux = synth.build("tiny", ncoord=6)
pk0 = synth.make_packets(model, aux, 4000, kpkt_fraction=0.2)
n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
pa, pb = pk0.copy(), pk0.copy()
ea, eb = abi.Estimators(n, g), abi.Estimators(n, g)
oracle_py.update_packets(model, cs, ts, pa, ea)
eng = engine.Engine(model); eng.set_cellstate(cs, ts)
t0 = time.time(); eng.update_packets(pb, eb); print("gpu time", time.time() - t0, eng.last_kernel_table())
try:
    parity.compare_packets(pb, pa, 1e-9, "dbg"); parity.compare_stats(eb, ea, "dbg", same_libm=False); print("PARITY OK")
except AssertionError as e:
    print("MISMATCH", str(e)[:500])
