#!/usr/bin/env python3
"""Which (cell, level) macro-atom records does one timestep of the bench workload actually visit?

The reference fills a level's rates when a packet first reaches it (macroatom.cc:398-417 calc_rates_if_needed); the engine
fills every record of every resident cell. This builds the classic library with -DARTIS_VISIT_COUNTS (every transition
drawn adds one to its record's counter), runs ONE step of `bench.py`'s workload and reports the fraction of records visited:
overall, per cell (quantiles), by how many visits, and how the visits concentrate.

    python tools/visit_sparsity.py [--preset w7|w7big|cd23like] [--packets N] [--ncoord 50] > profiles/r05/visit_sparsity_<preset>.md
"""
import argparse
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="w7")
    ap.add_argument("--packets", type=int, default=10_000_000)
    ap.add_argument("--ncoord", type=int, default=50)
    args = ap.parse_args()
    from artis_amd import build as B

    so = os.path.join(tempfile.gettempdir(), "libartis_amd_visitcounts.so")
    if not os.path.exists(so):
        import subprocess

        subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, "-DARTIS_VISIT_COUNTS", "-o", so, os.path.join(B.CSRC, "artis_engine.hip")],
                              stderr=subprocess.DEVNULL)
    os.environ["ARTIS_AMD_SO"] = so
    from artis_amd import abi, engine, synth

    model, cs, ts, aux = synth.build(args.preset, ncoord=args.ncoord)
    pk = synth.make_packets(model, aux, args.packets, seed_base=1281360349, kpkt_fraction=0.02, seed=99)
    eng = engine.Engine(model, device=0)
    eng.set_cellstate(cs, ts)
    eng.upload_packets(pk)
    eng.zero_estimators(0)
    eng.populate_cellcache(0)
    eng.step(0)
    ncell, nlev = int(model["npts_nonempty"]), int(model["nlevels"])
    counts = np.zeros(ncell * nlev, dtype=np.uint32)
    eng.L.artis_amd_debug_visit_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    eng._check(eng.L.artis_amd_debug_visit_counts(eng.h, counts.ctypes.data_as(C.c_void_p), counts.size))
    counts = counts.reshape(ncell, nlev)
    total = int(counts.sum(dtype=np.int64))
    touched = counts > 0
    per_cell = touched.sum(axis=1) / nlev
    q = np.quantile(per_cell, [0, 0.05, 0.25, 0.5, 0.75, 0.95, 1.0])
    nt, tcells, bpc = eng.cache_tiles()
    print(f"# Visited macro-atom records, one timestep ({args.preset}: {model['nlines']} lines, {nlev} levels; {args.ncoord}^3 grid, "
          f"{ncell} non-empty cells, {args.packets} packets)\n")
    print(f"* transitions drawn: {total:.4g}; records: {ncell * nlev:.4g}; cache row {bpc / 1e6:.2f} MB/cell in {nt} tile(s)")
    print(f"* **records visited at all: {touched.mean():.3f}** of (cell, level); cells with no visit: {(per_cell == 0).mean():.3f}")
    print("* per cell, fraction of its levels visited: min / 5 % / 25 % / median / 75 % / 95 % / max = " + " / ".join(f"{x:.3f}" for x in q))
    for thr in (1, 2, 4, 16, 64, 256):
        print(f"* records with >= {thr} visits: {(counts >= thr).mean():.3f}")
    flat = np.sort(counts.ravel())[::-1].astype(np.float64)
    cum = np.cumsum(flat) / max(total, 1)
    for frac in (0.5, 0.9, 0.99, 0.999):
        k = int(np.searchsorted(cum, frac)) + 1
        print(f"* {frac:.3f} of the transitions are drawn in the {k / flat.size:.4f} most-visited records")
    lev_any = touched.any(axis=0)
    print(f"* levels visited in at least one cell: {lev_any.mean():.3f}; levels visited in >= half of the visited cells: "
          f"{(touched.sum(axis=0) >= 0.5 * (per_cell > 0).sum()).mean():.3f}")
    print(f"* a cell's visited levels by the cell's packet count: correlation of per-cell fraction with transitions per cell "
          f"{np.corrcoef(per_cell, counts.sum(axis=1))[0, 1]:.2f}")
    # What would a STATIC set of levels (the same in every cell) capture? Levels ranked by their transitions over all cells, and --
    # what the engine could know before the step -- by excitation energy within their ion.
    lev_tot = counts.sum(axis=0, dtype=np.float64)
    order = np.argsort(-lev_tot)
    ion_start = np.asarray(model["ion_uniquelevelindexstart"])
    ion_nlev = np.asarray(model["ion_nlevels"])
    rank_in_ion = np.concatenate([np.arange(n) for n in ion_nlev])  # levels of an ion are in rising energy
    print("\n| static set of levels | share of levels | transitions drawn inside the set | visited (cell, level) records inside | worst cell: transitions inside |")
    print("|---|---|---|---|---|")
    cell_tot = np.maximum(counts.sum(axis=1, dtype=np.float64), 1.0)
    for frac in (0.02, 0.05, 0.1, 0.2, 0.3, 0.5):
        k = max(1, int(frac * nlev))
        sel = np.zeros(nlev, dtype=bool)
        sel[order[:k]] = True
        inside = counts[:, sel].sum(axis=1, dtype=np.float64)
        print(f"| the {k} levels with most transitions | {frac:.2f} | {lev_tot[sel].sum() / total:.5f} | {touched[:, sel].sum() / touched.sum():.3f} | {(inside / cell_tot).min():.4f} |")
    for frac in (0.05, 0.1, 0.2, 0.3, 0.5):
        sel = rank_in_ion < np.repeat(np.maximum(1, (frac * ion_nlev).astype(int)), ion_nlev)
        inside = counts[:, sel].sum(axis=1, dtype=np.float64)
        print(f"| the lowest {frac:.2f} of every ion's levels | {sel.mean():.2f} | {lev_tot[sel].sum() / total:.5f} | {touched[:, sel].sum() / touched.sum():.3f} | {(inside / cell_tot).min():.4f} |")
    np.save(os.path.join(os.environ.get("VISIT_OUT", tempfile.gettempdir()), f"visit_level_totals_{args.preset}.npy"), lev_tot)
    eng.close()


if __name__ == "__main__":
    main()
