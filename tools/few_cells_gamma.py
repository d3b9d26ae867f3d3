#!/usr/bin/env python3
"""A 30-shell spherical model with gamma rays and pellets (4e6 packets), host-buffer update_packets: the per-cell estimator
accumulators in LDS on and off ("Models with few cells"). GPU box: python tools/few_cells_gamma.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artis_amd import abi, engine, synth  # noqa: E402

model, cs, ts, aux = synth.build("w7", ncoord=30, gridtype=abi.GRID_SPHERICAL1D)
pk0 = synth.make_packets(model, aux, 4000000, kpkt_fraction=0.02, gamma_fraction=0.6, pellet_fraction=0.1)
for lds in ("1", "0"):
    os.environ["ARTIS_AMD_CELLEST_LDS"] = lds
    eng = engine.Engine(model)
    eng.set_cellstate(cs, ts)
    for rep in range(2):
        pk, est = pk0.copy(), abi.estimators_for(model, "classic")
        t0 = time.time()
        eng.update_packets(pk, est)
        dt = time.time() - t0
    print("ARTIS_AMD_CELLEST_LDS", lds, "update_packets (host buffers)", round(dt * 1e3), "ms; dep_estimator_gamma sum",
          est.arrays()["dep_estimator_gamma"].sum())
    eng.close()
