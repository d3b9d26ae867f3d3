#!/usr/bin/env python3
"""Run the same two timesteps several times on the engine and compare the packets of the runs with each other (GPU).
usage: r05_determinism.py <preset> <nruns> [tiles] -- the environment variants are the KEY=VAL words of $VARIANTS, ';'-separated sets."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artis_amd import abi, engine as engine_mod, synth  # noqa: E402

options = sys.argv[1] if len(sys.argv) > 1 else "classic"
nruns = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 1
model, cs, ts, aux = synth.build("small", ncoord=8, options=options, nts=13)
pk0 = synth.make_packets(model, aux, 30000, kpkt_fraction=0.2, gamma_fraction=0.1, pellet_fraction=0.2)
n = model["npts_nonempty"]
os.environ["ARTIS_AMD_MA_HOTFRAC"] = os.environ.get("ARTIS_AMD_MA_HOTFRAC", "1")
os.environ["ARTIS_AMD_MA_POOLFRAC"] = "1"


def run():
    eng = engine_mod.Engine(model, preset=options)
    p = pk0.copy()
    eng.upload_packets(p)
    t = aux["t"]
    mids = []
    for step in range(2):
        tsn = synth.make_timestep(t, width_frac=0.05, vmax=model["vmax"], nts=12 + step)
        eng.set_cellstate(cs, tsn)
        eng.step()
        t = tsn.c.start + tsn.c.width
        q = pk0.copy()
        eng.download_packets(q)
        mids.append(q)
    ct = eng.cache_tiles()
    eng.close()
    return mids, ct


variants = [v.strip() for v in os.environ.get("VARIANTS", "").split(";")] or [""]
for var in variants:
    saved = {}
    for kv in var.split():
        k, v = kv.split("=")
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    if tiles > 1:
        os.environ.pop("ARTIS_AMD_CACHE_BUDGET_MB", None)
        _, ct = run()
        os.environ["ARTIS_AMD_CACHE_BUDGET_MB"] = str(ct[2] * (n // tiles + 1) / 1048576.0 + 0.01)
    runs = [run() for _ in range(nruns)]
    ref = runs[0][0]
    nbad = 0
    for r, (mids, ct) in enumerate(runs[1:], 1):
        for s in range(2):
            a, b = ref[s], mids[s]
            diff = np.zeros(len(a), dtype=bool)
            for f in a.dtype.names:  # (named fields only: the padding bytes of the record are whatever the device buffer held)
                x, y = np.asarray(a[f]), np.asarray(b[f])
                ne = (x != y) & ~((x != x) & (y != y)) if x.dtype.kind == "f" else (x != y)
                diff |= ne.reshape(len(a), -1).any(axis=1)
            bad = np.nonzero(diff)[0]
            if len(bad):
                nbad += 1
                print(f"  [{var}] run {r} step {s}: {len(bad)} packets differ from run 0: {bad[:4]}")
                for i in bad[:2]:
                    for f in a.dtype.names:
                        if np.asarray(a[f][i]).tobytes() != np.asarray(b[f][i]).tobytes():
                            print(f"      {f}: {a[f][i]} | {b[f][i]}")
                    print(f"      (same) type {a['type'][i]} cell {a['cellindex'][i]} initial type {pk0['type'][i]}; other run type {b['type'][i]} cell {b['cellindex'][i]}")
                break
    print(f"[{var}] {options} tiles={runs[0][1][0]}: {nbad} of {nruns - 1} runs differ from run 0", flush=True)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    os.environ.pop("ARTIS_AMD_CACHE_BUDGET_MB", None)
