#!/usr/bin/env python3
"""Wall time of the cell-cache population alone (bench grid): python tools/populate_time.py [preset] [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from artis_amd import synth, engine
preset = sys.argv[1] if len(sys.argv) > 1 else "w7"
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model, cs, ts, aux = synth.build(preset, ncoord=50)
eng = engine.Engine(model)
eng.set_cellstate(cs, ts)
torch.cuda.synchronize()
for r in range(rep):
    t0 = time.perf_counter()
    eng.populate_cellcache()
    torch.cuda.synchronize()
    print(f"{preset}: populate {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
eng.close()
