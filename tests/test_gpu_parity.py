"""GPU parity tests (run on the MI355X box with `pytest -m gpu`): the HIP engine, called through its C-ABI,
against the CPU oracle on the same seeded inputs.

Bar: every integer field of every packet (type, cell, next line, emission/absorption ids, scatter counts), the
per-packet RNG state and all event counters are IDENTICAL; floating-point fields agree to FLOAT_RTOL. The only
source of float differences is the device math library (exp/log/sin/cos/expm1/atan2/pow are not bit-identical to
glibc); with the same libm the kernel bodies are bit-exact (tests/test_kernel_bodies_vs_oracle.py).
Estimators are atomic float sums, compared to EST_RTOL of each array's maximum.
"""
import os

import numpy as np
import pytest

import parity
from artis_amd import abi, synth

pytestmark = pytest.mark.gpu

FLOAT_RTOL = 1e-9
EST_RTOL = 1e-9


BIG_CASES = {
    # BASELINE.json configs[0] as written: 1D spherical grid, the bench's atomic data ("w7": 7 elements, 33 ions, 1.4e4
    # lines), artisoptions_classic physics, 1e5 packets
    "w7_1d_1e5": dict(build=dict(preset="w7", ncoord=50, gridtype=abi.GRID_SPHERICAL1D), npk=100_000,
                      pkw=dict(kpkt_fraction=0.05), dense_cells=0),
    # the bench grid itself (50^3 Cartesian, w7 data) sampled densely: 2e5 packets started in the 1500 heaviest cells
    # (>= 100 packets per cell as in the 1e7-packet bench, where a cell holds ~150), so that the per-cell paths the
    # bench exercises -- many packets of one cell in a wave, sorted lists, chunked work pulling -- meet the oracle
    "w7_50cubed_dense_2e5": dict(build=dict(preset="w7", ncoord=50), npk=200_000, pkw=dict(kpkt_fraction=0.02),
                                 dense_cells=1500),
    # the same cut for the nltenebular build (BASELINE.json configs[4] on the bench grid): the deferred bound-free estimator
    # records and k_bfest_dense at the bench's packet density, against the oracle (smaller: its per-step work is ~15x)
    "nltenebular_50cubed_dense_6e4": dict(build=dict(preset="w7", ncoord=50, options="nltenebular", nts=13), npk=60_000,
                                          pkw=dict(kpkt_fraction=0.02), dense_cells=500),
    # ... and for the kilonova_lte build (configs[3]'s packet-path options on the bench grid; round 4)
    "kilonova_lte_50cubed_dense_1e5": dict(build=dict(preset="w7", ncoord=50, options="kilonova_lte"), npk=100_000,
                                           pkw=dict(kpkt_fraction=0.02), dense_cells=800),
    # Round 6 -- the realistic-size atomic data under the oracle. `w7big` (110 860 lines, 4 427 levels): 6000 packets of
    # which 4800 start as k-packets, so that the first thermal lists hold >= 4096 entries and the launch takes k_thermal<1024, 2> (LevelPack
    # alone in LDS, the 2-byte target levels in HBM: what the 50^3 / 1e7 run of this data set executes); 3.9e8 transitions
    # (ARTIS_AMD_CELLEST_LDS=0: a model with few cells keeps its per-cell estimators in the workgroup's LDS, which that form of the kernel
    # leaves to the level table since round 6 -- the 50^3 grid has no such accumulators either)
    "w7big_5cubed_6e3": dict(build=dict(preset="w7big", ncoord=5), npk=6000, pkw=dict(kpkt_fraction=0.8), dense_cells=0,
                             expect_variants="LDS_LEVELPACK", env={"ARTIS_AMD_CELLEST_LDS": "0"}),
    # `cd23like` (406 132 lines, 8 457 levels, directions of hundreds of transitions: k_mafilter_long) with the record tiers the ENGINE chooses
    # when its static rows do not fit the cache budget (forced here to 0.7 of them on a 5^3 grid): static records for the lowest levels of every
    # ion, the rest filled on demand in the shared pool by the slow-path kernel's waves, k_thermal<256, 0, COLD>; 3.2e8 transitions
    "cd23like_5cubed_ondemand_5e3": dict(build=dict(preset="cd23like", ncoord=5), npk=5000, pkw=dict(kpkt_fraction=0.5), dense_cells=0,
                                         budget_frac=0.7, expect_variants="PLAIN|COLD"),
}


@pytest.fixture(scope="module")
def oracle_big(oracle):
    """The oracle's answers for the large cases, computed in forked worker processes BEFORE this process touches the GPU."""
    os.environ.setdefault("ARTIS_ORACLE_CACHE_CAP", "2500")
    out = {}
    for name, c in BIG_CASES.items():
        model, cs, ts, aux = synth.build(**c["build"])
        pkw = dict(c["pkw"])
        if c["dense_cells"]:
            w = aux["cellvol_tmin"] * np.exp(-aux["v"] / 4.0e8)
            pkw["cells_only"] = np.argsort(w)[-c["dense_cells"]:]
        pk0 = synth.make_packets(model, aux, c["npk"], **pkw)
        pa = pk0.copy()
        options = c["build"].get("options", "classic")
        ea = abi.estimators_for(model, options)
        parity.oracle_parallel(model, cs, ts, pa, ea, preset=options)
        out[name] = (model, cs, ts, pk0, pa, ea)
    return out


@pytest.fixture(scope="module")
def engine_mod(oracle_big):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from artis_amd import engine

    engine.load_library()
    return engine


def _run_case(engine_mod, oracle, preset, ncoord, gridtype, thick_v, npk, kfrac=0.2, gfrac=0.0, pfrac=0.0, bkw=None, pkw=None,
              options="classic", model_override=None, rho_scale=None):
    model, cs, ts, aux = synth.build(preset, ncoord=ncoord, gridtype=gridtype, thick_below_v=thick_v, options=options, **(bkw or {}))
    if model_override:
        model = abi.Model({**model.d, **model_override})
    if rho_scale:
        cs = abi.CellState({**cs.d, "rho": cs.d["rho"] * rho_scale})
    pk0 = synth.make_packets(model, aux, npk, kpkt_fraction=kfrac, gamma_fraction=gfrac, pellet_fraction=pfrac, **(pkw or {}))
    pa, pb = pk0.copy(), pk0.copy()
    ea, eb = abi.estimators_for(model, options), abi.estimators_for(model, options)
    oracle.update_packets(model, cs, ts, pa, ea, preset=options)
    eng = engine_mod.Engine(model, preset=options)
    eng.set_cellstate(cs, ts)
    eng.update_packets(pb, eb)
    return model, cs, ts, eng, pa, pb, ea, eb


@pytest.mark.parametrize("preset,ncoord,gridtype,thick_v,npk", [
    ("tiny", 6, abi.GRID_CARTESIAN3D, 0.0, 4000),
    ("small", 8, abi.GRID_CARTESIAN3D, 0.0, 20000),
    ("small", 8, abi.GRID_CARTESIAN3D, 6e8, 8000),
    ("small", 24, abi.GRID_SPHERICAL1D, 0.0, 8000),
    ("small", 16, abi.GRID_SPHERICAL1D, 5e8, 4000),
    ("small", 8, abi.GRID_CYLINDRICAL2D, 0.0, 10000),
    ("tiny", 6, abi.GRID_CYLINDRICAL2D, 5e8, 4000),
])
def test_engine_matches_oracle(engine_mod, oracle, preset, ncoord, gridtype, thick_v, npk):
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, preset, ncoord, gridtype, thick_v, npk)
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, "HIP engine vs oracle")
    parity.compare_stats(eb, ea, "HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, "HIP engine vs oracle")
    assert ea.stats[abi.STAT_X_RPKT_STEPS] > npk
    print(f"worst float rel diff {rep['worst_rel']:.3e}; packet-steps {ea.stats[34] + ea.stats[35]}")
    eng.close()


@pytest.mark.parametrize("preset,ncoord,gridtype,npk", [
    ("small", 8, abi.GRID_CARTESIAN3D, 20000),
    ("tiny", 16, abi.GRID_SPHERICAL1D, 8000),
    ("tiny", 6, abi.GRID_CYLINDRICAL2D, 8000),
])
def test_engine_matches_oracle_gamma_packets(engine_mod, oracle, preset, ncoord, gridtype, npk):
    """gamma packets (k_gamma: gammapkt.cc transport, Compton / photoelectric / pair production) mixed with r- and
    k-packets; thermalised gamma packets continue as k-packets in the same call"""
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, preset, ncoord, gridtype, 0.0, npk, gfrac=0.6)
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, "gamma: HIP engine vs oracle")
    parity.compare_stats(eb, ea, "gamma: HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, "gamma: HIP engine vs oracle")
    st = eb.stats_dict()
    assert st["X_GAMMA_STEPS"] > 0.6 * npk and st["NT_STAT_FROM_GAMMA"] > 0.1 * npk
    assert eb.scalars[0] > 0 and eb.dep_estimator_gamma.sum() > 0
    print(f"worst float rel diff {rep['worst_rel']:.3e}; gamma steps {st['X_GAMMA_STEPS']}")
    eng.close()


@pytest.mark.parametrize("gridtype,ncoord,bkw,pkw", [
    (abi.GRID_CARTESIAN3D, 8, {}, {}),
    (abi.GRID_CYLINDRICAL2D, 6, {"nts": 0, "t_days": 2.0, "tmin_days": 2.0}, {"early_pellets": True}),
])
def test_engine_matches_oracle_all_packet_types(engine_mod, oracle, gridtype, ncoord, bkw, pkw):
    """pellets, gamma packets, non-thermal particles and deposits together with r-/k-packets: every type of do_packet()"""
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, "small", ncoord, gridtype, 0.0, 16000, kfrac=0.1, gfrac=0.2,
                                                    pfrac=0.6, bkw=bkw, pkw=pkw)
    parity.compare_packets(pb, pa, FLOAT_RTOL, "all types: HIP engine vs oracle")
    parity.compare_stats(eb, ea, "all types: HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, "all types: HIP engine vs oracle")
    assert eb.scalars[2] > 1000 and eb.dep_estimator_alpha.sum() > 0   # pellet decays, alpha deposition
    eng.close()


@pytest.mark.parametrize("gridtype,ncoord", [(abi.GRID_CARTESIAN3D, 8), (abi.GRID_SPHERICAL1D, 16)])
def test_engine_matches_oracle_kilonova_lte_preset(engine_mod, oracle, gridtype, ncoord):
    """the engine built with the packet-path options of artisoptions_kilonova_lte.h (libartis_amd_kilonova_lte.so)
    against the oracle built with the same options; all packet types"""
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, "small", ncoord, gridtype, 0.0, 16000, kfrac=0.15, gfrac=0.15,
                                                    pfrac=0.4, options="kilonova_lte")
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, "kilonova_lte: HIP engine vs oracle")
    parity.compare_stats(eb, ea, "kilonova_lte: HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, "kilonova_lte: HIP engine vs oracle")
    assert eb.stats[abi.STAT_X_RPKT_STEPS] > 16000 and eb.colheatingestimator.sum() == 0
    print(f"worst float rel diff {rep['worst_rel']:.3e}")
    eng.close()


@pytest.mark.parametrize("name,spec", [("oneion", ([(26, 2, 1)], 8, 0.5, 10)), ("twoel", ([(14, 1, 2), (26, 1, 1)], 5, 0.6, 8))])
def test_degenerate_atomic_data(engine_mod, oracle, name, spec):
    """zero lines / zero bound-free continua / a single cooling term through the C-ABI (zero-length tables are legal)"""
    synth.PRESETS[name] = spec
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, name, 5, abi.GRID_CARTESIAN3D, 0.0, 6000, kfrac=0.3, gfrac=0.1)
    parity.compare_packets(pb, pa, FLOAT_RTOL, name)
    parity.compare_stats(eb, ea, name, same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, name)
    eng.close()


def test_population_sizes_around_wave_and_sort_boundaries(engine_mod, oracle):
    """ragged populations: 1 .. 1025 packets (below / at / above a wavefront, a workgroup and the list-sort threshold), all
    packet types, one engine reused for all sizes (buffers grow and shrink)"""
    model, cs, ts, aux = synth.build("tiny", ncoord=6)
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    eng = engine_mod.Engine(model)
    eng.set_cellstate(cs, ts)
    for npk in (1025, 1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1024):
        pk0 = synth.make_packets(model, aux, npk, kpkt_fraction=0.3, gamma_fraction=0.1, pellet_fraction=0.2, seed=npk)
        pa, pb = pk0.copy(), pk0.copy()
        ea, eb = abi.Estimators(n, g), abi.Estimators(n, g)
        oracle.update_packets(model, cs, ts, pa, ea)
        eng.update_packets(pb, eb)
        parity.compare_packets(pb, pa, FLOAT_RTOL, f"{npk} packets")
        parity.compare_stats(eb, ea, f"{npk} packets", same_libm=False)
        parity.compare_estimators(eb, ea, EST_RTOL, f"{npk} packets")
    eng.close()


def test_consecutive_timesteps_match_oracle(engine_mod, oracle):
    """three consecutive timesteps, with the packets resident on the device in between (upload once, set the next
    timestep, step, ... download once) against the oracle called once per timestep"""
    model, cs, ts, aux = synth.build("small", ncoord=8)
    pk0 = synth.make_packets(model, aux, 20000, kpkt_fraction=0.2, gamma_fraction=0.1, pellet_fraction=0.3)
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    pa, pb = pk0.copy(), pk0.copy()
    ea, eb = abi.Estimators(n, g), abi.Estimators(n, g)
    eng = engine_mod.Engine(model)
    eng.upload_packets(pb)
    t = aux["t"]
    for step in range(3):
        tsn = synth.make_timestep(t, width_frac=0.05, vmax=model["vmax"], nts=10 + step)
        oracle.update_packets(model, cs, tsn, pa, ea)
        eng.set_cellstate(cs, tsn)
        eng.step()
        t = tsn.c.start + tsn.c.width
    eng.download_packets(pb)
    eng.download_estimators(eb)
    parity.compare_packets(pb, pa, FLOAT_RTOL, "3 timesteps: HIP engine vs oracle")
    parity.compare_stats(eb, ea, "3 timesteps: HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, "3 timesteps: HIP engine vs oracle")
    eng.close()


def test_engine_matches_oracle_w7_atomic_data(engine_mod, oracle):
    """The benchmark's atomic data set (7 elements, 33 ions, ~1.4e4 lines) on a small grid."""
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, "w7", 10, abi.GRID_CARTESIAN3D, 0.0, 6000, kfrac=0.05)
    parity.compare_packets(pb, pa, FLOAT_RTOL, "HIP engine vs oracle (w7)")
    parity.compare_stats(eb, ea, "HIP engine vs oracle (w7)", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, "HIP engine vs oracle (w7)")
    eng.close()


@pytest.mark.parametrize("name", list(BIG_CASES))
def test_engine_matches_oracle_large_cases(engine_mod, oracle_big, name, monkeypatch):
    """configs[0] at its stated size, dense cuts of the bench grid, and the atomic data sets of realistic size on the kernels the 50^3 / 1e7 runs
    of those data take (see BIG_CASES), same bars as the small cases"""
    model, cs, ts, pk0, pa, ea = oracle_big[name]
    case = BIG_CASES[name]
    options = case["build"].get("options", "classic")
    pb = pk0.copy()
    eb = abi.estimators_for(model, options)
    if "budget_frac" in case:
        # a cache budget below what the static rows take (without line_dpop, which the engine drops first): the engine chooses the tiers itself
        eng = engine_mod.Engine(model, preset=options)
        ntiles, _, bpc = eng.cache_tiles()
        assert ntiles == 1 and eng.record_tiers()["ncold"] == 0      # (with the whole device free the static rows fit)
        eng.close()
        static_mb = (bpc - 8 * model["nlines"]) * model["npts_nonempty"] / 2**20
        monkeypatch.setenv("ARTIS_AMD_CACHE_BUDGET_MB", f"{case['budget_frac'] * static_mb:.1f}")
        monkeypatch.delenv("ARTIS_AMD_MA_HOTFRAC", raising=False)
    for k, v in case.get("env", {}).items():
        monkeypatch.setenv(k, v)
    eng = engine_mod.Engine(model, preset=options)
    if "budget_frac" in case:
        tiers = eng.record_tiers()
        assert eng.cache_tiles()[0] == 1 and tiers["ncold"] > 0 and 0 < tiers["hot_fraction"] < 1 and tiers["pool_slots"] > 0, tiers
        print(f"{name}: record tiers chosen by the engine: {tiers}")
    eng.set_cellstate(cs, ts)
    eng.update_packets(pb, eb)
    if "expect_variants" in case:
        want = 0
        for v in case["expect_variants"].split("|"):
            want |= getattr(eng, "THERMAL_" + v)
        got = eng.last_thermal_variants()
        assert got & want == want, f"{name}: thermal kernel forms launched {got:#x}, expected {want:#x} among them"
        print(f"{name}: thermal kernel forms {got:#x}, pool resets {eng.last_tiling()['pool_resets']}")
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, f"{name}: HIP engine vs oracle")
    parity.compare_stats(eb, ea, f"{name}: HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, f"{name}: HIP engine vs oracle")
    steps = int(ea.stats[abi.STAT_X_RPKT_STEPS] + ea.stats[abi.STAT_X_KPKT_STEPS])
    assert steps > 20 * len(pk0)
    if BIG_CASES[name]["dense_cells"]:
        counts = np.bincount(pk0["cellindex"])
        assert np.median(counts[counts > 0]) >= 100
    print(f"{name}: {len(pk0)} packets, {steps} packet-steps, {int(ea.stats[abi.STAT_X_MA_JUMPS])} transitions, "
          f"worst float rel diff {rep['worst_rel']:.3e}")
    eng.close()


@pytest.mark.parametrize("options,hotfrac", [("classic", "1"), ("nltenebular", "1"), ("classic_expopac_therm", "1"),
                                             # round 5: tiles AND on-demand macro-atom records (a refill empties the pool: a packet that
                                             # waited for its tile finds its cold level's record gone and has it filled again)
                                             ("classic", "0.3"), ("nltenebular", "0.3")])
def test_cell_cache_tiling_gives_identical_packets(engine_mod, oracle, monkeypatch, options, hotfrac):
    """the cell cache cut into tiles that do not fit together (ARTIS_AMD_CACHE_BUDGET_MB): the engine sweeps over the
    tiles, parking packets that enter a cell of another tile; packet histories must not depend on it. Compared with the
    untiled engine bit for bit, and with the oracle to the usual bars; all packet types, two consecutive timesteps. Also
    for the nltenebular build (a non-thermal deposit may activate a macro-atom in a cell of another tile) and an
    expansion-opacity build (the engine's own opacity tables are made tile by tile)."""
    model, cs, ts, aux = synth.build("small", ncoord=8, options=options, nts=13)
    pk0 = synth.make_packets(model, aux, 30000, kpkt_fraction=0.2, gamma_fraction=0.1, pellet_fraction=0.2)
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    # (given, so that the forced budget does not make the engine choose record tiers of its own; the pool of this small model whole)
    monkeypatch.setenv("ARTIS_AMD_MA_HOTFRAC", hotfrac)
    monkeypatch.setenv("ARTIS_AMD_MA_POOLFRAC", "1")

    def run(budget_mb):
        if budget_mb is None:
            monkeypatch.delenv("ARTIS_AMD_CACHE_BUDGET_MB", raising=False)
        else:
            monkeypatch.setenv("ARTIS_AMD_CACHE_BUDGET_MB", str(budget_mb))
        eng = engine_mod.Engine(model, preset=options)
        tiles = eng.cache_tiles()
        p, est = pk0.copy(), abi.estimators_for(model, options)
        eng.upload_packets(p)
        t = aux["t"]
        for step in range(2):
            tsn = synth.make_timestep(t, width_frac=0.05, vmax=model["vmax"], nts=12 + step)
            eng.set_cellstate(cs, tsn)
            eng.step()
            t = tsn.c.start + tsn.c.width
        eng.download_packets(p)
        eng.download_estimators(est)
        if budget_mb is not None and options == "classic":  # the diagnostics copy of a cell of the LAST tile comes from a refilled tile
            a, b = oracle.cellcache(model, cs, tsn, n - 1), eng.debug_cellcache(n - 1)
            assert np.allclose(a["levelpops"], b["levelpops"], rtol=1e-12) and np.allclose(a["cooling_contrib"], b["cooling_contrib"], rtol=1e-12)
        eng.close()
        return p, est, tiles

    p1, e1, t1 = run(None)
    bytes_per_cell = t1[2]
    p3, e3, t3 = run(bytes_per_cell * (n // 3 + 1) / 1048576.0 + 0.01)
    assert t1[0] == 1 and t3[0] == 3, (t1, t3)
    parity.compare_packets(p3, p1, 0.0, "3 cache tiles vs 1")
    skip = abi.STAT_NAMES.index("UPDATECELL")
    mask = np.arange(abi.NSTATS) != skip
    assert np.array_equal(e3.stats[mask], e1.stats[mask])
    assert e3.stats[skip] > e1.stats[skip]  # tiles were refilled
    parity.compare_estimators(e3, e1, EST_RTOL, "3 cache tiles vs 1")
    pa, ea = pk0.copy(), abi.estimators_for(model, options)
    t = aux["t"]
    for step in range(2):
        tsn = synth.make_timestep(t, width_frac=0.05, vmax=model["vmax"], nts=12 + step)
        oracle.update_packets(model, cs, tsn, pa, ea, preset=options)
        t = tsn.c.start + tsn.c.width
    parity.compare_packets(p3, pa, FLOAT_RTOL, "3 cache tiles vs oracle")


@pytest.mark.parametrize("options", ["classic", "kilonova_expopac"])
def test_line_population_factors_on_the_fly_give_identical_packets(engine_mod, oracle, monkeypatch, options):
    """ARTIS_AMD_DPOP=0: no line_dpop rows in the cell cache (what the engine chooses by itself when the cache does not fit one tile with
    them); the row shrinks by 8 bytes per line, the packets are those of the default run and of the oracle."""
    model, cs, ts, aux = synth.build("small", ncoord=8, options=options)
    pk0 = synth.make_packets(model, aux, 30000, kpkt_fraction=0.2)
    outs = []
    for dpop in ("1", "0"):
        monkeypatch.setenv("ARTIS_AMD_DPOP", dpop)
        eng = engine_mod.Engine(model, preset=options)
        eng.set_cellstate(cs, ts)
        p, e = pk0.copy(), abi.estimators_for(model, options)
        eng.update_packets(p, e)
        outs.append((p, e, eng.cache_tiles()[2]))
        eng.close()
    monkeypatch.delenv("ARTIS_AMD_DPOP", raising=False)
    assert outs[0][2] - outs[1][2] == 8 * model["nlines"], (outs[0][2], outs[1][2])
    parity.compare_packets(outs[1][0], outs[0][0], 0.0, "population factors on the fly vs stored")
    parity.compare_stats(outs[1][1], outs[0][1], "population factors on the fly vs stored")
    pa, ea = pk0[:6000].copy(), abi.estimators_for(model, options)
    oracle.update_packets(model, cs, ts, pa, ea, preset=options)
    parity.compare_packets(outs[1][0][:6000], pa, FLOAT_RTOL, "population factors on the fly vs oracle")


@pytest.mark.parametrize("options", ["classic", "nltenebular"])
def test_cooling_guides_give_the_bisections_packets(engine_mod, monkeypatch, options):
    """ARTIS_AMD_COOLGUIDE=0: the two draws of a k-packet step (kpkt.cc:430-447) by bisection, as in rounds 1-4, instead of by the per-cell
    guide tables (tables.h "COOLING GUIDES"): the row shrinks by the guides, packets, generator states and counters are identical."""
    model, cs, ts, aux = synth.build("small", ncoord=8, options=options)
    pk0 = synth.make_packets(model, aux, 30000, kpkt_fraction=0.5)
    outs = []
    for guide in ("1", "0"):
        monkeypatch.setenv("ARTIS_AMD_COOLGUIDE", guide)
        eng = engine_mod.Engine(model, preset=options)
        eng.set_cellstate(cs, ts)
        p, e = pk0.copy(), abi.estimators_for(model, options)
        eng.update_packets(p, e)
        outs.append((p, e, eng.cache_tiles()[2]))
        eng.close()
    monkeypatch.delenv("ARTIS_AMD_COOLGUIDE", raising=False)
    assert 0 < outs[0][2] - outs[1][2] < 2 * 2 * (model["ncoolingterms"] + 4 * model["nions"] + 70), (outs[0][2], outs[1][2])
    assert outs[0][1].stats[abi.STAT_X_KPKT_STEPS] > 10000
    parity.compare_packets(outs[1][0], outs[0][0], 0.0, "cooling guides vs bisection")
    parity.compare_stats(outs[1][1], outs[0][1], "cooling guides vs bisection")


def test_sparse_fills_do_not_cost_sweeps_and_vpkt_refuses_tiles(engine_mod, monkeypatch):
    """Two findings of the round-3 review of the tiled cache. (1) A sparse fill (a late visit that populates only the cells in which
    packets wait) left its residency bitmap switched on while the next tile's packets were classified, which hid that tile until the
    next sweep: with the fills on, a tiled run must not need more sweeps than with whole-tile fills, and gives the same packets.
    (2) A VPKT_ON build reads every cell's row along a virtual packet's ray: an engine whose cache does not fit one tile is refused."""
    model, cs, ts, aux = synth.build("small", ncoord=12)
    pk0 = synth.make_packets(model, aux, 60000, kpkt_fraction=0.2)
    n = model["npts_nonempty"]
    monkeypatch.setenv("ARTIS_AMD_MA_HOTFRAC", "1")  # (the forced budget is not to make the engine choose on-demand record tiers)
    outs = []
    for sparse in ("0", "1"):
        monkeypatch.delenv("ARTIS_AMD_CACHE_BUDGET_MB", raising=False)
        eng = engine_mod.Engine(model)
        bpc = eng.cache_tiles()[2]
        eng.close()
        monkeypatch.setenv("ARTIS_AMD_CACHE_BUDGET_MB", str(bpc * (n // 4 + 1) / 1048576.0 + 0.01))
        monkeypatch.setenv("ARTIS_AMD_SPARSE_FILL", sparse)
        monkeypatch.setenv("ARTIS_AMD_SPARSE_MAX", "100000")  # every late visit of this small population qualifies
        eng = engine_mod.Engine(model)
        assert eng.cache_tiles()[0] == 4
        eng.set_cellstate(cs, ts)
        p, e = pk0.copy(), abi.estimators_for(model, "classic")
        eng.update_packets(p, e)
        outs.append((p, e, eng.last_tiling()))
        eng.close()
    (p0, e0, t0), (p1, e1, t1) = outs
    assert t1["sparse_fills"] > 0 and t0["sparse_fills"] == 0, (t0, t1)
    assert t1["sweeps"] <= t0["sweeps"], (t0, t1)
    parity.compare_packets(p1, p0, 0.0, "sparse fills vs whole-tile fills")
    # (2)
    vmodel, vcs, vts, vaux = synth.build("small", ncoord=8, options="ci_classic_vpkt", t_days=5.0)
    monkeypatch.setenv("ARTIS_AMD_CACHE_BUDGET_MB", "1")
    with pytest.raises(Exception, match="VPKT_ON"):
        engine_mod.Engine(vmodel, preset="ci_classic_vpkt")
    monkeypatch.delenv("ARTIS_AMD_CACHE_BUDGET_MB", raising=False)


def test_tiles_as_sets_of_cells_give_identical_packets(engine_mod, monkeypatch):
    """Round 6: a cache that does not fit has rows for a SET of cells at a time, addressed through a table (physics.h Env::krow_tab): a cell that is
    resident and still wanted keeps its row, a fill of the new cells leaves the pool of on-demand records alone, and the engine picks the record tiers
    that need the fewest tiles. Packet histories depend on none of it: every variant gives the untiled engine's packets and counters bit for bit --
    the engine's own choice of tiers under the forced budget, a small pool (used up and emptied during the run), sets made of blocks of cells,
    the fixed ranges of rounds 2-5, a pool emptied with every fill."""
    model, cs, ts, aux = synth.build("small", ncoord=10)
    pk0 = synth.make_packets(model, aux, 40000, kpkt_fraction=0.2, pellet_fraction=0.1)
    n = model["npts_nonempty"]
    names = ("ARTIS_AMD_CACHE_BUDGET_MB", "ARTIS_AMD_MA_HOTFRAC", "ARTIS_AMD_MA_POOLFRAC", "ARTIS_AMD_TILE_BLOCK", "ARTIS_AMD_TILE_ADAPT",
             "ARTIS_AMD_POOL_KEEP")

    def run(env):
        for k in names:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = engine_mod.Engine(model)
        tiles, tiers = eng.cache_tiles(), eng.record_tiers()
        eng.set_cellstate(cs, ts)
        p, e = pk0.copy(), abi.estimators_for(model, "classic")
        eng.update_packets(p, e)
        lt = eng.last_tiling()
        eng.close()
        return p, e, tiles, tiers, lt

    p1, e1, t1, r1, _ = run({})
    assert t1[0] == 1 and r1["ncold"] == 0
    budget = str(t1[2] * (n // 3 + 1) / 1048576.0 + 0.01)  # a third of the cells' static rows
    skip = abi.STAT_NAMES.index("UPDATECELL")
    mask = np.arange(abi.NSTATS) != skip
    seen = {}
    for label, env in (("own tiers", {}), ("small pool", {"ARTIS_AMD_MA_HOTFRAC": "0.3", "ARTIS_AMD_MA_POOLFRAC": "0.1"}),
                       ("blocks of cells", {"ARTIS_AMD_MA_HOTFRAC": "1", "ARTIS_AMD_TILE_BLOCK": "16"}),
                       ("fixed ranges", {"ARTIS_AMD_MA_HOTFRAC": "1", "ARTIS_AMD_TILE_ADAPT": "0"}),
                       ("pool emptied per fill", {"ARTIS_AMD_MA_HOTFRAC": "0.3", "ARTIS_AMD_MA_POOLFRAC": "0.1", "ARTIS_AMD_POOL_KEEP": "0"})):
        p, e, tiles, tiers, lt = run({"ARTIS_AMD_CACHE_BUDGET_MB": budget, **env})
        seen[label] = (tiles, tiers, lt)
        assert tiles[0] > 1, (label, tiles)
        parity.compare_packets(p, p1, 0.0, f"{label} vs untiled")
        assert np.array_equal(e.stats[mask], e1.stats[mask]), label
        parity.compare_estimators(e, e1, EST_RTOL, f"{label} vs untiled")
    for k in names:
        monkeypatch.delenv(k, raising=False)
    # the engine's own choice: smaller rows, fewer tiles than static rows need under the same budget
    assert seen["own tiers"][1]["ncold"] > 0 and seen["own tiers"][0][0] < seen["fixed ranges"][0][0], seen
    # rows are kept: the adaptive run with static rows fills fewer cells than its fills x the rows there are
    lt = seen["blocks of cells"][2]
    assert lt["cells_filled"] < lt["tile_fills"] * seen["blocks of cells"][0][1], seen["blocks of cells"]


def test_parked_visit_tails_give_identical_packets(engine_mod, monkeypatch):
    """Round 4: in a tiled run the last packets of a visit that began larger wait in the tile for its next visit (they run with the
    packets that return to it) instead of getting a long launch of their own. Packet histories do not depend on it."""
    model, cs, ts, aux = synth.build("small", ncoord=12)
    pk0 = synth.make_packets(model, aux, 60000, kpkt_fraction=0.2)
    n = model["npts_nonempty"]
    monkeypatch.setenv("ARTIS_AMD_MA_HOTFRAC", "1")
    monkeypatch.delenv("ARTIS_AMD_CACHE_BUDGET_MB", raising=False)
    eng = engine_mod.Engine(model)
    bpc = eng.cache_tiles()[2]
    eng.close()
    monkeypatch.setenv("ARTIS_AMD_CACHE_BUDGET_MB", str(bpc * (n // 4 + 1) / 1048576.0 + 0.01))
    monkeypatch.setenv("ARTIS_AMD_TAIL", "1024")
    outs = []
    for park in ("0", "1"):
        monkeypatch.setenv("ARTIS_AMD_TILE_PARK", park)
        eng = engine_mod.Engine(model)
        assert eng.cache_tiles()[0] == 4
        eng.set_cellstate(cs, ts)
        p, e = pk0.copy(), abi.estimators_for(model, "classic")
        eng.update_packets(p, e)
        outs.append((p, e, eng.last_tiling()))
        eng.close()
    for k in ("ARTIS_AMD_CACHE_BUDGET_MB", "ARTIS_AMD_TAIL", "ARTIS_AMD_TILE_PARK"):
        monkeypatch.delenv(k, raising=False)
    (p0, e0, t0), (p1, e1, t1) = outs
    assert t0["parked"] == 0 and t1["parked"] > 0, (t0, t1)
    parity.compare_packets(p1, p0, 0.0, "parked visit tails vs visits run to their end")
    parity.compare_estimators(e1, e0, EST_RTOL, "parked visit tails vs visits run to their end")


def test_estimator_allreduce_through_the_c_abi(engine_mod):
    """artis_amd_comm_unique_id / artis_amd_comm_init / artis_amd_allreduce_estimators on a one-rank communicator (the GPU
    box has one device): RCCL is found at run time, the communicator comes up, the in-place sum leaves the block as is"""
    model, cs, ts, aux = synth.build("tiny", ncoord=6)
    pk = synth.make_packets(model, aux, 4000, kpkt_fraction=0.3)
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    eng = engine_mod.Engine(model)
    eng.set_cellstate(cs, ts)
    eng.upload_packets(pk)
    eng.zero_estimators()
    eng.step()
    before = abi.Estimators(n, g)
    eng.download_estimators(before)
    ident = eng.comm_unique_id()
    assert len(ident) == 128 and any(ident)
    eng.comm_init(1, 0, ident)
    assert eng.comm_count() == 1  # ncclCommCount: what bench.py reports as rccl_nranks
    eng.allreduce_estimators()
    import torch

    torch.cuda.synchronize()
    after = abi.Estimators(n, g)
    eng.download_estimators(after)
    for k, a in before.arrays().items():
        assert np.array_equal(a, after.arrays()[k]), k
    assert before.J.sum() > 0
    eng.close()


def test_compiled_reference_side_binding(engine_mod, tmp_path):
    """tests/binding/update_packets_amd.cc: a C++20 translation unit that includes the REFERENCE's packet.h / constants.h /
    stats.h (static_asserts: struct Packet == artis_packet member by member, packet types, counters), owns the packets as
    std::span<Packet> and calls the engine through the C-ABI alone -- what a maintainer adds beside update_packets.cc.
    Prebuilt by `make -C oracle ref` (the reference's headers do not travel to the GPU box; the binary does). Every
    member of every packet is bit for bit that of the ctypes path; estimators to summation order."""
    import subprocess
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "binding"))
    import dump as bdump

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "update_packets_amd")
    if not os.path.exists(exe):  # git-ignored, built where the reference's headers are: a fresh checkout on a GPU box has none
        pytest.skip("oracle/_ref/update_packets_amd is missing: run `make -C oracle ref` where /root/reference exists")
    model, cs, ts, aux = synth.build("small", ncoord=8)
    pk0 = synth.make_packets(model, aux, 20000, kpkt_fraction=0.2, gamma_fraction=0.1, pellet_fraction=0.1)
    dfile, ofile = str(tmp_path / "case.dump"), str(tmp_path / "case.out")
    bdump.write_dump(dfile, model, cs, ts, pk0)
    from artis_amd.build import so_path

    out = subprocess.run([exe, so_path("classic"), dfile, ofile], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    pk_bin, est_bin, stats_bin = bdump.read_output(ofile, model, len(pk0))
    eng = engine_mod.Engine(model)
    eng.set_cellstate(cs, ts)
    pk_py, est_py = pk0.copy(), abi.Estimators(model["npts_nonempty"], model["nbfcontinua_ground"])
    eng.update_packets(pk_py, est_py)
    eng.close()
    for f in abi.PACKET_DTYPE.names:  # every member bit for bit (the padding between members is nobody's: numpy's copy does not keep it)
        assert pk_bin[f].tobytes() == pk_py[f].tobytes(), f
    assert np.array_equal(stats_bin[parity.EXACT_STATS], np.asarray(est_py.stats)[parity.EXACT_STATS])
    for k, a in est_bin.items():
        b = est_py.arrays()[k]
        assert np.abs(a - b).max() <= 1e-11 * max(np.abs(b).max(), 1e-300), k
    assert "packet-steps" in out.stdout


def test_cellcache_matches_oracle(engine_mod, oracle):
    model, cs, ts, aux = synth.build("small", ncoord=8, thick_below_v=4e8)
    eng = engine_mod.Engine(model)
    eng.set_cellstate(cs, ts)
    for c in (0, 17, model["npts_nonempty"] - 1):
        a = oracle.cellcache(model, cs, ts, c)
        b = eng.debug_cellcache(c)
        assert np.array_equal(a["allcont_keepbits"], b["allcont_keepbits"])
        for k in a:
            if k == "allcont_keepbits":
                continue
            x, y = np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64)
            denom = np.maximum(np.maximum(np.abs(x), np.abs(y)), 1e-300)
            assert (np.abs(x - y) / denom).max() < 1e-12, f"cell {c}: {k}"
    eng.close()


def test_filter_records_of_long_directions_match_the_sequential_form(engine_mod, oracle):
    """Atomic data with more transitions per direction than a block of k_matrans holds in LDS (256; here one ion with 420 levels: up to
    ~250 upward and ~290 downward transitions of a level): the rates of such a direction are summed 64 terms at a time and its filters
    written by k_mafilter_long (a wave per cell and direction, 63 transitions at a time). debug_cellcache() re-adds every cumulative
    sum of the cell on the device, compares every filter entry and mark of every record with the sequential form's (it fails
    otherwise), and its sums and rates meet the oracle's stored arrays. (The w7big data -- directions of up to 73 transitions,
    blocks of ~10 segments -- are checked the same way.)"""
    synth.PRESETS["longdir"] = ([(26, 1, 2)], 420, 0.7, 20)
    for preset in ("longdir", "w7big"):
        model, cs, ts, aux = synth.build(preset, ncoord=4)
        assert max(model.d["level_ndowntrans"]) > (256 if preset == "longdir" else 64)
        eng = engine_mod.Engine(model)
        eng.set_cellstate(cs, ts)
        c = model["npts_nonempty"] // 2
        a = oracle.cellcache(model, cs, ts, c)
        b = eng.debug_cellcache(c)
        for k in ("maprocessrates", "matrans", "cooling_contrib"):
            x, y = np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64)
            denom = np.maximum(np.maximum(np.abs(x), np.abs(y)), 1e-300)
            assert (np.abs(x - y) / denom).max() < 1e-12, f"{preset} cell {c}: {k}"
        eng.close()
    model, cs, ts, aux = synth.build("longdir", ncoord=4)
    eng = engine_mod.Engine(model)
    eng.set_cellstate(cs, ts)
    for c in (0, model["npts_nonempty"] - 1):
        a = oracle.cellcache(model, cs, ts, c)
        b = eng.debug_cellcache(c)
        for k in ("maprocessrates", "matrans", "cooling_contrib"):
            x, y = np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64)
            denom = np.maximum(np.maximum(np.abs(x), np.abs(y)), 1e-300)
            assert (np.abs(x - y) / denom).max() < 1e-12, f"cell {c}: {k}"
    # ... and the packets through those records
    pk0 = synth.make_packets(model, aux, 3000, kpkt_fraction=0.3)
    pa, pb = pk0.copy(), pk0.copy()
    ea, eb = abi.estimators_for(model, "classic"), abi.estimators_for(model, "classic")
    oracle.update_packets(model, cs, ts, pa, ea)
    eng.update_packets(pb, eb)
    parity.compare_packets(pb, pa, FLOAT_RTOL, "longdir atomic data: HIP engine vs oracle")
    parity.compare_stats(eb, ea, "longdir atomic data: HIP engine vs oracle", same_libm=False)
    eng.close()


def test_device_runs_are_deterministic_and_idempotent(engine_mod):
    """Size-independent properties at a larger size than the oracle is run at:
    (1) two runs from the same snapshot give bit-identical packets whatever the scheduling;
    (2) every packet ends escaped or exactly at the end of the timestep; counters are consistent;
    (3) updating an already finished population changes nothing."""
    model, cs, ts, aux = synth.build("small", ncoord=12)
    npk = 200_000
    pk0 = synth.make_packets(model, aux, npk, kpkt_fraction=0.1)
    eng = engine_mod.Engine(model)
    eng.set_cellstate(cs, ts)
    eng.upload_packets(pk0)
    eng.snapshot()
    outs, stats = [], []
    for _ in range(2):
        eng.restore()
        eng.zero_estimators()
        eng.step()
        out = pk0.copy()
        eng.download_packets(out)
        est = abi.Estimators(model["npts_nonempty"], model["nbfcontinua_ground"])
        eng.download_estimators(est)
        outs.append(out)
        stats.append(est)
    parity.compare_packets(outs[1], outs[0], 0.0, "run-to-run determinism")
    assert np.array_equal(stats[0].stats, stats[1].stats)
    end = ts.c.start + ts.c.width
    out = outs[0]
    esc = out["type"] == abi.TYPE_ESCAPE
    assert np.all(esc | (out["prop_time"] == end))
    assert stats[0].stats[abi.STAT_NAMES.index("PKTESCAPES")] == np.count_nonzero(esc)
    assert np.all(out["escape_time"][esc] > 0) and np.all(out["escape_type"][esc] == abi.TYPE_RPKT)
    assert np.all(np.abs(np.sqrt((out["dir"][~esc] ** 2).sum(axis=1)) - 1.0) < 1e-6)
    # (3) idempotence
    eng.zero_estimators()
    eng.step()
    again = pk0.copy()
    eng.download_packets(again)
    parity.compare_packets(again, out, 0.0, "idempotence")
    est = abi.Estimators(model["npts_nonempty"], model["nbfcontinua_ground"])
    eng.download_estimators(est)
    assert est.stats[abi.STAT_X_RPKT_STEPS] == 0 and est.J.sum() == 0.0
    eng.close()


@pytest.mark.parametrize("gridtype,ncoord", [(abi.GRID_SPHERICAL1D, 12), (abi.GRID_CARTESIAN3D, 6),
                                             (abi.GRID_CARTESIAN3D, 12)])  # 912 cells: k_rpkt's estimators take the LDS of its continuum table
def test_work_list_order_does_not_change_packets(engine_mod, oracle, monkeypatch, gridtype, ncoord):
    """Models with few cells: the kernels accumulate the per-cell estimators (J, nuJ, ffheating, colheating) in LDS and add
    a workgroup's sums to the global arrays once, and the work lists are counting-sorted by cell with the LDS form of the
    sort kernels. Against the same run with every estimator add a global atomic, lists never sorted / always sorted:
    same packets and counters, estimators equal to summation order; and the oracle's packets."""
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype)
    pk0 = synth.make_packets(model, aux, 60000, kpkt_fraction=0.2)
    outs = []
    for maxpc in (None, "1", "1000000000"):
        for v in ("ARTIS_AMD_SORT_MAXPC_R", "ARTIS_AMD_SORT_MAXPC_T", "ARTIS_AMD_CELLEST_LDS"):
            if maxpc is None:
                monkeypatch.delenv(v, raising=False)
            else:
                monkeypatch.setenv(v, "0" if v == "ARTIS_AMD_CELLEST_LDS" else maxpc)
        eng = engine_mod.Engine(model)
        eng.set_cellstate(cs, ts)
        p, e = pk0.copy(), abi.estimators_for(model, "classic")
        eng.update_packets(p, e)
        eng.close()
        outs.append((p, e))
    for p, e in outs[1:]:
        parity.compare_packets(p, outs[0][0], 0.0, "work-list order")
        parity.compare_stats(e, outs[0][1], "work-list order")
        parity.compare_estimators(e, outs[0][1], 1e-11, "work-list order")
    pa, ea = pk0[:6000].copy(), abi.estimators_for(model, "classic")
    oracle.update_packets(model, cs, ts, pa, ea)
    parity.compare_packets(outs[0][0][:6000], pa, FLOAT_RTOL, "default list policy vs oracle")


@pytest.mark.parametrize("options,gridtype,ncoord", [("classic", abi.GRID_CARTESIAN3D, 8), ("classic", abi.GRID_SPHERICAL1D, 16),
                                                     ("nltenebular", abi.GRID_CARTESIAN3D, 8), ("kilonova_expopac", abi.GRID_CARTESIAN3D, 8)])
def test_tail_kernel_gives_the_split_kernels_packets(engine_mod, oracle, monkeypatch, options, gridtype, ncoord):
    """k_tail carries the last r-packets and thermal packets of a population through all their remaining steps in one
    launch. Whole population through it (ARTIS_AMD_TAIL_ALWAYS), never (ARTIS_AMD_TAIL=0) and the default (the end of a
    population that began above the threshold): identical packets and counters, estimators to summation order; and the
    oracle's packets. All packet types, so that packets leave the tail kernel for the other kernels and come back."""
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype, options=options)
    pk0 = synth.make_packets(model, aux, 30000, kpkt_fraction=0.2, gamma_fraction=0.1, pellet_fraction=0.1)
    outs = []
    for env in ({"ARTIS_AMD_TAIL": "0"}, {"ARTIS_AMD_TAIL": "100000000", "ARTIS_AMD_TAIL_ALWAYS": "1"}, {"ARTIS_AMD_TAIL": "2000"}, {}):
        for v in ("ARTIS_AMD_TAIL", "ARTIS_AMD_TAIL_ALWAYS"):
            monkeypatch.delenv(v, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = engine_mod.Engine(model, preset=options)
        eng.set_cellstate(cs, ts)
        p, e = pk0.copy(), abi.estimators_for(model, options)
        eng.update_packets(p, e)
        eng.close()
        outs.append((p, e))
    for p, e in outs[1:]:
        parity.compare_packets(p, outs[0][0], 0.0, "tail kernel vs split kernels")
        parity.compare_stats(e, outs[0][1], "tail kernel vs split kernels")
        parity.compare_estimators(e, outs[0][1], 1e-11, "tail kernel vs split kernels")
    pa, ea = pk0[:6000].copy(), abi.estimators_for(model, options)
    oracle.update_packets(model, cs, ts, pa, ea, preset=options)
    parity.compare_packets(outs[1][0][:6000], pa, FLOAT_RTOL, "tail kernel vs oracle")


@pytest.mark.parametrize("options", ["classic", "nltenebular"])
def test_macroatom_filters_decide_nothing_the_f64_values_would_not(engine_mod, monkeypatch, options):
    """The 15-bit filters of the macro-atom records (tables.h "FILTERS") against the f64 comparisons they stand for, on the
    bench grid with the bench's atomic data. Round 4: the records hold nothing but the filters, and a draw they cannot decide
    re-adds the cumulative sums from the transitions' rate coefficients (physics.h ma_exact_search / kpkt_collexc_exact).
    With ARTIS_AMD_MAFILTERS=0 EVERY transition, radiative de-excitation and collisional-excitation cooling draw is decided
    that way (4e8 transitions, each re-adding its sums: what round 3 read from 1 MB of sums per cell): every field of every
    packet, the generator states and the event counters are identical to the run on the filters; estimators to summation order."""
    model, cs, ts, aux = synth.build("w7", ncoord=50, options=options)
    pk0 = synth.make_packets(model, aux, 150_000 if options == "classic" else 100_000, seed_base=1281360349, kpkt_fraction=0.02)
    outs = []
    for off in (False, True):
        monkeypatch.delenv("ARTIS_AMD_MAFILTERS", raising=False)
        if off:
            monkeypatch.setenv("ARTIS_AMD_MAFILTERS", "0")
        eng = engine_mod.Engine(model, preset=options)
        eng.set_cellstate(cs, ts)
        p, e = pk0.copy(), abi.estimators_for(model, options)
        eng.update_packets(p, e)
        eng.close()
        outs.append((p, e))
    assert outs[0][1].stats[abi.STAT_X_MA_JUMPS] > (3e8 if options == "classic" else 4e7)
    parity.compare_packets(outs[1][0], outs[0][0], 0.0, "f64 decisions vs filters")
    parity.compare_stats(outs[1][1], outs[0][1], "f64 decisions vs filters")
    parity.compare_estimators(outs[1][1], outs[0][1], 1e-10, "f64 decisions vs filters")


def test_walker_refill_kernel_gives_the_phase_kernels_packets(engine_mod, monkeypatch):
    """k_thermal_q (ARTIS_AMD_REFILL=1: walk contexts in per-wave LDS slots, the transition loop's lanes refilled inside the loop,
    exits and k-packet steps in full-wave service passes; round 5) against k_thermal on the bench grid with the bench's atomic data:
    every field of every packet, the generator states and the event counters identical (the same functions in the same order per
    packet, on the packet's own generator); estimators to summation order. With a low and a high low-water mark."""
    model, cs, ts, aux = synth.build("w7", ncoord=50)
    pk0 = synth.make_packets(model, aux, 400_000, seed_base=1281360349, kpkt_fraction=0.02)
    outs = []
    for cfg in ({}, {"ARTIS_AMD_REFILL": "1"}, {"ARTIS_AMD_REFILL": "1", "ARTIS_AMD_TQ_LOW": "16"}):
        for k in ("ARTIS_AMD_REFILL", "ARTIS_AMD_TQ_LOW"):
            monkeypatch.delenv(k, raising=False)
        for k, v in cfg.items():
            monkeypatch.setenv(k, v)
        eng = engine_mod.Engine(model)
        eng.set_cellstate(cs, ts)
        p, e = pk0.copy(), abi.estimators_for(model, "classic")
        eng.update_packets(p, e)
        eng.close()
        outs.append((p, e))
    assert outs[0][1].stats[abi.STAT_X_MA_JUMPS] > 5e8
    for o in outs[1:]:
        parity.compare_packets(o[0], outs[0][0], 0.0, "k_thermal_q vs k_thermal")
        parity.compare_stats(o[1], outs[0][1], "k_thermal_q vs k_thermal")
        parity.compare_estimators(o[1], outs[0][1], 1e-10, "k_thermal_q vs k_thermal")


def test_on_demand_records_give_the_static_records_packets(engine_mod, monkeypatch):
    """On-demand macro-atom records (tables.h "ON-DEMAND RECORDS"; round 5): static records for the lowest 30 % of every ion's levels,
    the others filled in their cell's pool by the slow-path kernel when a packet first reaches them -- against the run with a static record
    for every level, on the bench grid with the bench's atomic data: every field of every packet, the generator states and the event
    counters identical; estimators to summation order. Also through k_thermal_q, and with a pool too small (emptied whenever it is used up: the same packets)."""
    model, cs, ts, aux = synth.build("w7", ncoord=50)
    pk0 = synth.make_packets(model, aux, 400_000, seed_base=1281360349, kpkt_fraction=0.02)
    outs = []
    for cfg in ({}, {"ARTIS_AMD_MA_HOTFRAC": "0.3", "ARTIS_AMD_MA_POOLFRAC": "0.5"},
                {"ARTIS_AMD_MA_HOTFRAC": "0.1", "ARTIS_AMD_MA_POOLFRAC": "0.5", "ARTIS_AMD_REFILL": "1"}):
        for k in ("ARTIS_AMD_MA_HOTFRAC", "ARTIS_AMD_MA_POOLFRAC", "ARTIS_AMD_REFILL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in cfg.items():
            monkeypatch.setenv(k, v)
        eng = engine_mod.Engine(model)
        eng.set_cellstate(cs, ts)
        p, e = pk0.copy(), abi.estimators_for(model, "classic")
        eng.update_packets(p, e)
        outs.append((p, e, eng.cache_tiles()[2]))
        if cfg:
            # the records the run left in the pool (filled by waves of the slow-path kernel, ma_fill_record_wave) against the sequential
            # form: artis_amd_debug_cellcache() fails if any filter entry or mark of any record of the cell differs from populate_dirfilter_seq()
            nfilled = 0
            for c in (0, 777, 20000, 33000, 65000):
                d = eng.debug_cellcache(c)
                nfilled += int(np.count_nonzero(d["maprocessrates"].reshape(-1, 9).sum(axis=1) > 0))
            assert nfilled > 100  # (records with rates: the static ones and the cold ones packets reached in these cells)
        eng.close()
    assert outs[1][2] < 0.9 * outs[0][2] and outs[2][2] < outs[1][2]  # bytes per cell of the cache row
    for o in outs[1:]:
        parity.compare_packets(o[0], outs[0][0], 0.0, "on-demand records vs static records")
        parity.compare_stats(o[1], outs[0][1], "on-demand records vs static records")
        parity.compare_estimators(o[1], outs[0][1], 1e-10, "on-demand records vs static records")
    # a pool too small for the cold records the packets reach is emptied whenever it is used up (the records are filled again when next
    # needed, as after a tile's refill): the same packets, at the price of fills
    monkeypatch.setenv("ARTIS_AMD_MA_HOTFRAC", "0.1")
    monkeypatch.setenv("ARTIS_AMD_MA_POOLFRAC", "0.001")
    monkeypatch.delenv("ARTIS_AMD_REFILL", raising=False)
    eng = engine_mod.Engine(model)
    eng.set_cellstate(cs, ts)
    p, e = pk0.copy(), abi.estimators_for(model, "classic")
    eng.update_packets(p, e)
    resets = eng.last_tiling()["pool_resets"]
    eng.close()
    assert resets >= 2, resets
    parity.compare_packets(p, outs[0][0], 0.0, "on-demand records in a pool used up again and again vs static records")
    parity.compare_stats(e, outs[0][1], "on-demand records in a pool used up again and again vs static records")
    # ... and a pool that cannot hold one record is an error
    monkeypatch.setenv("ARTIS_AMD_MA_POOLFRAC", "0")
    eng = engine_mod.Engine(model)
    eng.set_cellstate(cs, ts)
    with pytest.raises(Exception, match="pool"):
        eng.update_packets(pk0.copy(), abi.estimators_for(model, "classic"))
    eng.close()


@pytest.mark.parametrize("options", ["classic", "nltenebular"])
def test_pool_used_up_under_the_tail_kernel_gives_the_static_records_packets(engine_mod, monkeypatch, options):
    """The pool of on-demand records used up while the TAIL kernel holds the packets (ARTIS_AMD_TAIL above the population: every packet is a
    tail packet from the start): a packet that waits for a record leaves the kernel for the slow-path list -- into the list's ALTERNATE buffer,
    not into the one other waves still read their packets from (round 5: found by tools/r05_determinism.py as packets that never finished their
    timestep) -- the host empties the pool, the tail takes the packets up again. Two timesteps, against static records: identical."""
    model, cs, ts, aux = synth.build("small", ncoord=8, options=options, nts=13)
    pk0 = synth.make_packets(model, aux, 30000, kpkt_fraction=0.2, gamma_fraction=0.1, pellet_fraction=0.2)
    outs = []
    for cfg in ({"ARTIS_AMD_MA_HOTFRAC": "1"}, {"ARTIS_AMD_MA_HOTFRAC": "0.3", "ARTIS_AMD_MA_POOLFRAC": "0.02", "ARTIS_AMD_TAIL": "40000"},
                {"ARTIS_AMD_MA_HOTFRAC": "0.3", "ARTIS_AMD_MA_POOLFRAC": "0.02"}):
        for k in ("ARTIS_AMD_MA_HOTFRAC", "ARTIS_AMD_MA_POOLFRAC", "ARTIS_AMD_TAIL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in cfg.items():
            monkeypatch.setenv(k, v)
        eng = engine_mod.Engine(model, preset=options)
        p, est = pk0.copy(), abi.estimators_for(model, options)
        eng.upload_packets(p)
        t, resets = aux["t"], 0
        for step in range(2):
            tsn = synth.make_timestep(t, width_frac=0.05, vmax=model["vmax"], nts=12 + step)
            eng.set_cellstate(cs, tsn)
            eng.step()
            resets += eng.last_tiling()["pool_resets"]
            t = tsn.c.start + tsn.c.width
        eng.download_packets(p)
        eng.download_estimators(est)
        eng.close()
        outs.append((p, est, resets))
    assert outs[0][2] == 0 and outs[1][2] >= 2 and outs[2][2] >= 2, [o[2] for o in outs]
    for o in outs[1:]:
        parity.compare_packets(o[0], outs[0][0], 0.0, "pool used up (tail kernel / split kernels) vs static records")
        parity.compare_stats(o[1], outs[0][1], "pool used up (tail kernel / split kernels) vs static records")


def test_repeated_runs_give_identical_packets(engine_mod):
    """The same two timesteps three times over on fresh engines: every named field of every packet identical between the runs. Work-pulling,
    atomics and list order make the ORDER of the work different from run to run; a packet's history must not notice (its own generator, its own
    record). Both defects of round 5 that lost a packet now and then -- a sort key beyond the histogram, an entry appended to a list other waves
    still read -- showed up here (tools/r05_determinism.py) before anywhere else."""
    model, cs, ts, aux = synth.build("small", ncoord=8, nts=13)
    pk0 = synth.make_packets(model, aux, 30000, kpkt_fraction=0.2, gamma_fraction=0.1, pellet_fraction=0.2)
    runs = []
    for _ in range(3):
        eng = engine_mod.Engine(model)
        p = pk0.copy()
        eng.upload_packets(p)
        t = aux["t"]
        for step in range(2):
            tsn = synth.make_timestep(t, width_frac=0.05, vmax=model["vmax"], nts=12 + step)
            eng.set_cellstate(cs, tsn)
            eng.step()
            t = tsn.c.start + tsn.c.width
        eng.download_packets(p)
        eng.close()
        runs.append(p)
    for r in runs[1:]:
        parity.compare_packets(r, runs[0], 0.0, "a repeated run vs the first")
    assert np.count_nonzero(runs[0]["prop_time"] < tsn.c.start + tsn.c.width * (1 - 1e-12)) == np.count_nonzero(
        (runs[0]["prop_time"] < tsn.c.start + tsn.c.width * (1 - 1e-12)) & (runs[0]["type"] != abi.TYPE_RPKT)), "an r-packet stopped before the end of the timestep"


def test_budget_independence_on_device(engine_mod, monkeypatch):
    model, cs, ts, aux = synth.build("tiny", ncoord=6)
    pk0 = synth.make_packets(model, aux, 20000, kpkt_fraction=0.3)
    outs = []
    for budget in ("1", "3", "1000000"):
        monkeypatch.setenv("ARTIS_AMD_BUDGET", budget)
        eng = engine_mod.Engine(model)
        eng.set_cellstate(cs, ts)
        p = pk0.copy()
        eng.update_packets(p, abi.Estimators(model["npts_nonempty"], model["nbfcontinua_ground"]))
        outs.append(p)
        eng.close()
    # ... and the hand-over to the next launch once a launch's list is used up (ARTIS_AMD_DRAIN_T / _R), forced on for every
    # launch however short its list, and switched off
    monkeypatch.delenv("ARTIS_AMD_BUDGET", raising=False)
    for env in ({"ARTIS_AMD_DRAIN_T": "1", "ARTIS_AMD_DRAIN_R": "1", "ARTIS_AMD_DRAIN_MIN": "0"}, {"ARTIS_AMD_DRAIN_T": "0", "ARTIS_AMD_DRAIN_R": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = engine_mod.Engine(model)
        eng.set_cellstate(cs, ts)
        p = pk0.copy()
        eng.update_packets(p, abi.Estimators(model["npts_nonempty"], model["nbfcontinua_ground"]))
        outs.append(p)
        eng.close()
    for p in outs[1:]:
        parity.compare_packets(p, outs[0], 0.0, "launch budget independence")


def test_edge_cases(engine_mod):
    model, cs, ts, aux = synth.build("tiny", ncoord=6)
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    eng = engine_mod.Engine(model)
    eng.set_cellstate(cs, ts)
    eng.update_packets(np.zeros(0, dtype=abi.PACKET_DTYPE), abi.Estimators(n, g))  # empty population
    pk = synth.make_packets(model, aux, 64)
    pk["type"][:16] = 0   # TYPE_NONE
    pk["type"][16:32] = 13  # TYPE_MA: types outside do_packet()'s switch are returned untouched
    pk["prop_time"][32:48] = ts.c.start + ts.c.width
    ref = pk.copy()
    eng.update_packets(pk, abi.Estimators(n, g))
    for f in abi.PACKET_DTYPE.names:
        assert pk[f][:48].tobytes() == ref[f][:48].tobytes(), f
    for f in ("tdecay", "number", "pellet_decaytype", "pellet_nucindex", "originated_from_particlenotgamma"):
        assert np.array_equal(pk[f], ref[f])  # fields the path never touches survive the round trip
    eng.close()


@pytest.mark.parametrize("options,gridtype,ncoord,host_tables", [("kilonova_expopac", abi.GRID_CARTESIAN3D, 8, False),
                                                                 ("kilonova_expopac", abi.GRID_SPHERICAL1D, 16, True),
                                                                 ("classic_expopac_therm", abi.GRID_CARTESIAN3D, 8, False),
                                                                 ("classic_expopac_therm", abi.GRID_CARTESIAN3D, 8, True)])
def test_engine_matches_oracle_expansion_opacities(engine_mod, oracle, options, gridtype, ncoord, host_tables):
    """the expansion-opacity builds (rpkt.cc:221-330; with the bin re-trace, and with the thermalisation probability of
    rpkt.cc:624-648 and kpkt.cc:402) against the oracle built alike, with the tables of calculate_expansion_opacities()
    (rpkt.cc:1071) made by the engine's own kernels or handed over by the host; half a set of tables is refused"""
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, "small", ncoord, gridtype, 4e8 if "therm" in options else 0.0,
                                                    16000, kfrac=0.15, gfrac=0.1, pfrac=0.2, options=options,
                                                    bkw=dict(host_expopac=host_tables))
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, options + ": HIP engine vs oracle")
    parity.compare_stats(eb, ea, options + ": HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, options + ": HIP engine vs oracle")
    st = eb.stats_dict()
    assert st["X_RPKT_STEPS"] > 16000 and ((st["MA_STAT_ACTIVATION_BB"] == 0) == ("therm" in options))
    if host_tables and "therm" in options:
        bare = abi.CellState({k: v for k, v in cs.d.items() if k != "expansionopacity_planck_cumulative"})
        with pytest.raises(engine_mod.EngineError):
            eng.set_cellstate(bare, ts)
    print(f"worst float rel diff {rep['worst_rel']:.3e}")
    eng.close()


@pytest.mark.parametrize("options", ["kilonova_gamma_barnes", "kilonova_gamma_wollaeger", "kilonova_gamma_guttman"])
def test_engine_matches_oracle_parameterised_gamma_thermalisation(engine_mod, oracle, options):
    """the Barnes / Wollaeger / Guttman gamma-ray thermalisation builds (gammapkt.cc:775-866) against the oracle built alike"""
    mtot = 0.5 * 1.98855e33   # Barnes: t_ineff = 14 d, f_gamma(20 d) ~ 0.4
    base = synth.build("small", ncoord=8, options=options)[0]
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, "small", 8, abi.GRID_CARTESIAN3D, 0.0, 12000, kfrac=0.1, gfrac=0.7,
                                                    pfrac=0.1, options=options,
                                                    model_override={"mtot_input": mtot, "ejecta_kinetic_energy": 0.5 * mtot * (0.2 * 2.99792458e10) ** 2,
                                                                    "rho_tmin": base.d["rho_tmin"] * 0.5})
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, options + ": HIP engine vs oracle")
    parity.compare_stats(eb, ea, options + ": HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, options + ": HIP engine vs oracle")
    esc = pb[pb["type"] == abi.TYPE_ESCAPE]
    assert np.count_nonzero(esc["escape_type"] == abi.TYPE_GAMMA) > 200 and eb.dep_estimator_gamma.sum() > 0
    print(f"worst float rel diff {rep['worst_rel']:.3e}")
    eng.close()


@pytest.mark.parametrize("options", ["kilonova_gamma_grey", "classic_gamma_xcom"])
def test_engine_matches_oracle_gamma_opacity_options(engine_mod, oracle, options):
    """the grey-opacity and XCOM gamma-ray builds (gammapkt.cc:266-553) against the oracle built alike"""
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, "small", 8, abi.GRID_CARTESIAN3D, 0.0, 16000, kfrac=0.1, gfrac=0.7,
                                                    pfrac=0.1, options=options)
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, options + ": HIP engine vs oracle")
    parity.compare_stats(eb, ea, options + ": HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, options + ": HIP engine vs oracle")
    assert eb.stats_dict()["NT_STAT_FROM_GAMMA"] > 1000 and eb.dep_estimator_gamma.sum() > 0
    print(f"worst float rel diff {rep['worst_rel']:.3e}")
    eng.close()


def test_engine_matches_oracle_gamma_products(engine_mod, oracle):
    """the TIMEDEPENDENTWITHGAMMAPRODUCTS build (gammapkt.cc:404, :572, :630, :734, :925) against the oracle built alike"""
    P = "kilonova_gammaproducts"
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, "small", 8, abi.GRID_CARTESIAN3D, 0.0, 16000, kfrac=0.1, gfrac=0.7,
                                                    pfrac=0.1, options=P)
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, P + ": HIP engine vs oracle")
    parity.compare_stats(eb, ea, P + ": HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, P + ": HIP engine vs oracle")
    assert eb.stats_dict()["NT_STAT_FROM_GAMMA"] > 1000 and eb.dep_estimator_gamma.sum() > 0
    print(f"worst float rel diff {rep['worst_rel']:.3e}")
    eng.close()


@pytest.mark.parametrize("options", ["kilonova_barnes", "kilonova_wollaeger"])
def test_engine_matches_oracle_analytic_thermalisation(engine_mod, oracle, options):
    """the Barnes / Wollaeger particle thermalisation builds (update_packets.cc:53-88) against the oracle built alike"""
    mtot = 5.0e-3 * 1.98855e33  # a 0.005 Msun, 0.2 c kilonova: Barnes' f_p(20 d) ~ 0.2
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, "small", 8, abi.GRID_CARTESIAN3D, 0.0, 16000, kfrac=0.1, gfrac=0.1,
                                                    pfrac=0.7, options=options, rho_scale=1e-4,
                                                    model_override={"mtot_input": mtot, "ejecta_kinetic_energy": 0.5 * mtot * (0.2 * 2.99792458e10) ** 2})
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, options + ": HIP engine vs oracle")
    parity.compare_stats(eb, ea, options + ": HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, options + ": HIP engine vs oracle")
    esc = pb[pb["type"] == abi.TYPE_ESCAPE]
    assert np.count_nonzero(np.isin(esc["escape_type"], [21, 22, 23])) > 100 and eb.stats_dict()["NT_STAT_TO_KPKT"] > 100
    print(f"worst float rel diff {rep['worst_rel']:.3e}")
    eng.close()


@pytest.mark.parametrize("gridtype,ncoord,nts", [(abi.GRID_CARTESIAN3D, 8, 13), (abi.GRID_SPHERICAL1D, 16, 13),
                                                  (abi.GRID_CARTESIAN3D, 8, 10)])
def test_engine_matches_oracle_nltenebular_preset(engine_mod, oracle, gridtype, ncoord, nts):
    """the engine built with the packet-path options of artisoptions_nltenebular.h (libartis_amd_nltenebular.so):
    host level populations and photoionisation coefficients, the binned radiation field (read past
    FIRST_NLTE_RADFIELD_TIMESTEP, accumulated always), the detailed bound-free estimators and the NT_ON channels
    (Spencer-Fano fractions from the host), against the oracle built with the same options"""
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, "small", ncoord, gridtype, 0.0, 16000, kfrac=0.15, gfrac=0.15,
                                                    pfrac=0.3, options="nltenebular", bkw=dict(nts=nts))
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, "nltenebular: HIP engine vs oracle")
    parity.compare_stats(eb, ea, "nltenebular: HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, "nltenebular: HIP engine vs oracle")
    assert eb.stats[abi.STAT_X_RPKT_STEPS] > 16000 and eb.gammaestimator.sum() == 0
    assert np.count_nonzero(eb.bfrate_raw) > 100 and eb.radfieldbin_J.sum() > 0.5 * eb.J.sum()
    st = eb.stats_dict()
    assert st["NT_STAT_TO_IONISATION"] > 100 and st["NT_STAT_TO_EXCITATION"] > 30 and st["MA_STAT_INTERNALUPHIGHERNT"] > 50
    # a cell state without the solver's arrays is refused, not silently replaced by LTE values
    for missing in (("levelpops", "corrphotoioncoeff"), ("nt_frac_ionisation",), ("nt_exc_alltransindex",)):
        bare = abi.CellState({k: v for k, v in cs.d.items() if k not in missing})
        with pytest.raises(engine_mod.EngineError):
            eng.set_cellstate(bare, ts)
    print(f"worst float rel diff {rep['worst_rel']:.3e}")
    eng.close()


@pytest.mark.parametrize("options", ["christinenonthermal", "nltephotospheric", "nltewithoutnonthermal"])
def test_engine_matches_oracle_remaining_option_files(engine_mod, oracle, options):
    """the builds for the reference's other three options files (64 / 256 / 512 radiation-field bins, estimators for a
    subset of the continua, polarisation with the non-thermal channels, level-population bound-free cooling) against the
    oracle built alike; all packet types"""
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, "small", 8, abi.GRID_CARTESIAN3D, 0.0, 16000, kfrac=0.15, gfrac=0.15,
                                                    pfrac=0.3, options=options, bkw=dict(nts=13))
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, options + ": HIP engine vs oracle")
    parity.compare_stats(eb, ea, options + ": HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, options + ": HIP engine vs oracle")
    assert eb.stats[abi.STAT_X_RPKT_STEPS] > 16000 and np.count_nonzero(eb.bfrate_raw) > 100
    if options == "nltephotospheric":   # a model without the estimator map is refused by this build
        with pytest.raises(engine_mod.EngineError):
            engine_mod.Engine(abi.Model({k: v for k, v in model.d.items() if k not in ("allcont_bfestimindex", "nbfestim")}), preset=options)
    print(f"worst float rel diff {rep['worst_rel']:.3e}")
    eng.close()


def test_engine_matches_oracle_detailed_line_estimators(engine_mod, oracle):
    """the DETAILED_LINE_ESTIMATORS_ON build (radfield.cc:773, rpkt.cc:173-207, macroatom.cc:628) against the oracle built
    alike: contribution counts identical, intensities to the estimator tolerance"""
    P = "nltenebular_lineest"
    model, cs, ts, eng, pa, pb, ea, eb = _run_case(engine_mod, oracle, "small", 8, abi.GRID_CARTESIAN3D, 0.0, 16000, kfrac=0.15, gfrac=0.1,
                                                    pfrac=0.2, options=P, bkw=dict(nts=13))
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, P + ": HIP engine vs oracle")
    parity.compare_stats(eb, ea, P + ": HIP engine vs oracle", same_libm=False)
    assert np.array_equal(ea.Jb_lu_contribcount, eb.Jb_lu_contribcount) and eb.Jb_lu_contribcount.sum() > 5000
    parity.compare_estimators(eb, ea, EST_RTOL, P + ": HIP engine vs oracle")
    print(f"worst float rel diff {rep['worst_rel']:.3e}")
    eng.close()


def test_deferred_bound_free_estimators_equal_in_place(engine_mod, oracle, monkeypatch):
    """nltenebular build: the detailed bound-free estimator updates recorded by k_rpkt and added by k_bfest_dense (a wave
    per update) against the same build adding them in place (ARTIS_AMD_BFDEFER=0): identical packets and counters, the
    estimators to the accuracy of float summation order. w7 atomic data (1851 continua: windows longer than a wave); and
    both against the oracle on the bench's atomic data."""
    P = "nltenebular"
    model, cs, ts, aux = synth.build("w7", ncoord=10, options=P, nts=13)
    pk0 = synth.make_packets(model, aux, 20000, kpkt_fraction=0.05)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ARTIS_AMD_BFDEFER", mode)
        eng = engine_mod.Engine(model, preset=P)
        eng.set_cellstate(cs, ts)
        p, est = pk0.copy(), abi.estimators_for(model, P)
        eng.update_packets(p, est)
        eng.close()
        out[mode] = (p, est)
    (p1, e1), (p0, e0) = out["1"], out["0"]
    parity.compare_packets(p1, p0, 0.0, "deferred vs in-place bound-free estimators")
    assert np.array_equal(e1.stats, e0.stats)
    assert np.count_nonzero(e0.bfrate_raw) > 10000 and np.array_equal(e1.bfrate_raw != 0, e0.bfrate_raw != 0)
    parity.compare_estimators(e1, e0, 1e-11, "deferred vs in-place bound-free estimators")
    pa, ea = pk0.copy(), abi.estimators_for(model, P)
    oracle.update_packets(model, cs, ts, pa, ea, preset=P)
    parity.compare_packets(p1, pa, FLOAT_RTOL, "nltenebular, w7 atomic data: HIP engine vs oracle")
    parity.compare_stats(e1, ea, "nltenebular, w7 atomic data: HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(e1, ea, EST_RTOL, "nltenebular, w7 atomic data: HIP engine vs oracle")


@pytest.mark.parametrize("options", sorted(abi.CI_PRESETS))
def test_engine_matches_oracle_reference_ci_option_sets(engine_mod, oracle, options):
    """the engine built for each option COMBINATION the reference's CI runs (tests/setup_*.sh -> presets ci_*; see
    tests/test_kernel_bodies_vs_oracle.py::test_reference_ci_option_sets_bit_exact), on the grid type that set-up uses,
    all packet types, against the oracle built with the same options"""
    from test_kernel_bodies_vs_oracle import ci_case

    model, cs, ts, aux = ci_case(options)
    pk0 = synth.make_packets(model, aux, 16000, kpkt_fraction=0.15, gamma_fraction=0.2, pellet_fraction=0.3)
    pa, pb = pk0.copy(), pk0.copy()
    ea, eb = abi.estimators_for(model, options), abi.estimators_for(model, options)
    oracle.update_packets(model, cs, ts, pa, ea, preset=options)
    eng = engine_mod.Engine(model, preset=options)
    eng.set_cellstate(cs, ts)
    eng.update_packets(pb, eb)
    rep = parity.compare_packets(pb, pa, FLOAT_RTOL, options + ": HIP engine vs oracle")
    parity.compare_stats(eb, ea, options + ": HIP engine vs oracle", same_libm=False)
    parity.compare_estimators(eb, ea, EST_RTOL, options + ": HIP engine vs oracle")
    assert eb.stats[abi.STAT_X_RPKT_STEPS] > 10000
    if options == "ci_kilonova_xcom":  # a cell state without the per-cell mean atomic weights is refused
        with pytest.raises(engine_mod.EngineError):
            eng.set_cellstate(abi.CellState({k: v for k, v in cs.d.items() if k != "elem_meanweight"}), ts)
    print(f"worst float rel diff {rep['worst_rel']:.3e}")
    eng.close()


@pytest.mark.parametrize("options,npk,t_days", [("classic", 10_000_000, 20.0), ("kilonova_lte", 10_000_000, 20.0), ("nltenebular", 10_000_000, 20.0),
                                                # round 5: the expansion-opacity build at full size; the virtual-packet build inside its spectra
                                                # window (1e6 packets send 3.9e6 virtual packets to the observers: what its bench line runs)
                                                ("kilonova_expopac", 10_000_000, 20.0), ("ci_classic_vpkt", 1_000_000, 5.0)])
def test_full_size_properties_50cubed_1e7_packets(engine_mod, options, npk, t_days):
    """BASELINE.json's bench configuration itself (50^3 cells, w7 atomic data, 1e7 packets; configs[1] with the classic
    options, configs[3]'s packet-path options with libartis_amd_kilonova_lte.so, configs[4]'s with
    libartis_amd_nltenebular.so) through properties that do not
    need the oracle: (1) two runs from the same device snapshot are bit-identical (packets and event counters);
    (2) every packet ends escaped or exactly at the end of the timestep, with finite positive state; (3) packets are
    independent, so the event counters of the whole population equal the sum over its two halves run separately --
    exactly -- and the estimators agree to the accuracy of float summation (a checksum of checksums)."""
    model, cs, ts, aux = synth.build("w7", ncoord=50, options=options, t_days=t_days)
    pk0 = synth.make_packets(model, aux, npk, seed_base=1281360349, kpkt_fraction=0.02)
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    t_end = ts.c.start + ts.c.width
    eng = engine_mod.Engine(model, preset=options)
    eng.set_cellstate(cs, ts)
    eng.upload_packets(pk0)
    eng.snapshot()

    def run():
        eng.restore()
        eng.zero_estimators()
        eng.step()
        est = abi.estimators_for(model, options)
        eng.download_estimators(est)
        out = np.empty_like(pk0)
        out[:] = pk0
        eng.download_packets(out)
        return out, est

    p1, e1 = run()
    p2, e2 = run()
    assert np.array_equal(e1.stats, e2.stats)
    for f in abi.PACKET_DTYPE.names:
        assert np.array_equal(p1[f], p2[f], equal_nan=True), f      # (1) determinism, field by field
    del p2
    esc = p1["type"] == abi.TYPE_ESCAPE
    assert 1000 < esc.sum() < npk
    assert np.all(p1["prop_time"][~esc] == t_end)                     # (2)
    assert np.all(np.isin(p1["type"][~esc], [abi.TYPE_RPKT, abi.TYPE_KPKT]))
    for f in ("pos", "dir", "nu_cmf", "nu_rf", "e_cmf", "e_rf"):
        assert np.all(np.isfinite(p1[f])), f
    assert np.all(p1["e_rf"] > 0) and np.all(p1["nu_rf"][p1["type"] == abi.TYPE_RPKT] > 0)
    flying = p1["type"] != abi.TYPE_KPKT  # (a packet that began and ended the timestep as a k-packet has no direction yet)
    assert np.all(np.abs(np.sqrt((p1["dir"][flying] ** 2).sum(axis=1)) - 1) < 1e-9)
    r_esc = np.sqrt((p1["pos"][esc] ** 2).sum(axis=1)) * (model["tmin"] / p1["prop_time"][esc])
    assert np.all(r_esc > 0.7 * model["rmax"])
    steps = int(e1.stats[abi.STAT_X_RPKT_STEPS] + e1.stats[abi.STAT_X_KPKT_STEPS])
    assert steps > 3e8 and e1.stats_dict()["PKTESCAPES"] == int(esc.sum())
    if "vpkt" in options:  # virtual packets were traced and reached the observers' spectra
        assert e1.stats[abi.STAT_X_VPKT_CREATED] > npk and e1.vspecpol.sum() > 0
    assert np.all(e1.J >= 0) and e1.J.sum() > 0 and np.all(np.isfinite(e1.gammaestimator))
    keep = p1 if options == "classic" else None  # (for the fresh-engine comparison below)
    del p1

    half = npk // 2                                                    # (3) additivity over a split of the population
    ea, eb = abi.estimators_for(model, options), abi.estimators_for(model, options)
    pa, pb = pk0[:half].copy(), pk0[half:].copy()
    eng.update_packets(pa, ea)
    eng.update_packets(pb, eb)
    skip = abi.STAT_NAMES.index("UPDATECELL")
    mask = np.arange(abi.NSTATS) != skip
    assert np.array_equal((ea.stats + eb.stats)[mask], e1.stats[mask])
    for k, whole in e1.arrays().items():
        parts = ea.arrays()[k] + eb.arrays()[k]
        scale = max(np.abs(whole).max(), 1e-300)
        assert np.abs(parts - whole).max() / scale < 1e-9, k
    bytes_per_cell = eng.cache_tiles()[2]
    eng.close()
    if keep is not None:
        # (4) round 6: a second, FRESH engine (its own allocations, list buffers, pool and launch history) gives the same packets field by field and
        # the same counters: the two races of round 5 (a sort key beyond the histogram; a tail packet appended to a list other waves still read)
        # showed only across engines and only now and then, at this size
        del pa, pb
        eng2 = engine_mod.Engine(model, preset=options)
        eng2.set_cellstate(cs, ts)
        p3 = pk0.copy()
        e3 = abi.estimators_for(model, options)
        eng2.update_packets(p3, e3)
        eng2.close()
        assert np.array_equal(e3.stats[mask], e1.stats[mask])
        for f in abi.PACKET_DTYPE.names:
            assert np.array_equal(p3[f], keep[f], equal_nan=True), f"fresh engine: {f}"
        esc3 = p3["type"] == abi.TYPE_ESCAPE
        assert np.all(p3["prop_time"][~esc3] == t_end)   # no packet ends before the timestep does there either
        # (5) round 6: a third engine that may use a QUARTER of the memory the cell cache takes -- rows for a set of cells at a time, addressed through
        # a table, the record tiers of its own choice, the pool of on-demand records kept across fills -- gives the same packets and counters again
        del p3
        os.environ["ARTIS_AMD_CACHE_BUDGET_MB"] = str(bytes_per_cell * (n // 4 + 1) / 1048576.0)
        try:
            eng3 = engine_mod.Engine(model, preset=options)
            tiles3, tiers3 = eng3.cache_tiles(), eng3.record_tiers()
            eng3.set_cellstate(cs, ts)
            p4 = pk0.copy()
            e4 = abi.estimators_for(model, options)
            eng3.update_packets(p4, e4)
            lt = eng3.last_tiling()
            eng3.close()
        finally:
            del os.environ["ARTIS_AMD_CACHE_BUDGET_MB"]
        assert tiles3[0] > 1 and tiers3["ncold"] > 0 and lt["tile_fills"] > 1, (tiles3, tiers3, lt)
        assert np.array_equal(e4.stats[mask], e1.stats[mask])
        for f in abi.PACKET_DTYPE.names:
            assert np.array_equal(p4[f], keep[f], equal_nan=True), f"tiled engine: {f}"
