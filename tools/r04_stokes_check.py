#!/usr/bin/env python3
"""One-off (round 4): the stress run at 2e6 packets failed the 1e-9 bar on stokes_q by 17 % for one packet. Which packet, and how large is
its q? (A Stokes parameter is a difference of O(1) terms: a relative error measured against |q| itself grows as |q| -> 0.)
Usage (GPU box): python tools/r04_stokes_check.py [npackets] [ncoord]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from artis_amd import abi, engine, synth
from oracle import oracle_py
npk = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ncoord = int(sys.argv[2]) if len(sys.argv) > 2 else 12
model, cs, ts, aux = synth.build("w7", ncoord=ncoord, gridtype=abi.GRID_CARTESIAN3D, options="classic", t_days=20.0)
pk0 = synth.make_packets(model, aux, npk, kpkt_fraction=0.1, gamma_fraction=0.1, pellet_fraction=0.2)
pa, pb = pk0.copy(), pk0.copy()
ea, eb = abi.estimators_for(model, "classic"), abi.estimators_for(model, "classic")
oracle_py.update_packets(model, cs, ts, pa, ea, preset="classic")
eng = engine.Engine(model, preset="classic"); eng.set_cellstate(cs, ts); eng.update_packets(pb, eb)
for f in abi.PACKET_INT_FIELDS:
    assert np.array_equal(pa[f], pb[f]), f
assert np.array_equal(pa["rngstate"], pb["rngstate"])
for f in ("stokes_q", "stokes_u") if "stokes_q" in pa.dtype.names else [n for n in pa.dtype.names if "stokes" in n]:
    a, b = np.asarray(pb[f], float), np.asarray(pa[f], float)
    den = np.maximum(np.abs(a), np.abs(b)); den[~(den > 0)] = 1.0
    rel = np.abs(a - b) / den
    worst = np.argsort(rel)[-5:][::-1]
    print(f, "worst relative differences:")
    for i in worst:
        print("   packet %d: engine %.17g oracle %.17g  rel %.3e  abs %.3e" % (i, a[i], b[i], rel[i], abs(a[i] - b[i])))
    print("   largest ABSOLUTE difference %.3e; packets with |q| < 1e-3: %d of %d" % (np.abs(a - b).max(), np.count_nonzero(np.abs(b) < 1e-3), len(b)))
