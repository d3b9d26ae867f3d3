"""CPU differential tests: the kernel bodies (artis_amd/csrc/physics.h, compiled for x86 by tests/hostemu)
against the CPU oracle. Both use glibc libm here, so packet histories must be bit-identical; on the GPU the
same comparison runs through the C-ABI with a tolerance for the device math library (tests/test_gpu_parity.py).
"""
import numpy as np
import pytest

import hostemu_binding as emu
import parity
from artis_amd import abi, synth


def _run_both(oracle, model, cs, ts, pk0, budget, options="classic"):
    pa, pb = pk0.copy(), pk0.copy()
    ea, eb = abi.estimators_for(model, options), abi.estimators_for(model, options)
    oracle.update_packets(model, cs, ts, pa, ea, preset=options)
    emu.update_packets(model, cs, ts, pb, eb, budget=budget, preset=options)
    return pa, pb, ea, eb


@pytest.mark.parametrize("preset,ncoord,gridtype,thick_v,npk,budget", [
    ("tiny", 6, abi.GRID_CARTESIAN3D, 0.0, 1500, 1),
    ("small", 8, abi.GRID_CARTESIAN3D, 0.0, 3000, 4),
    ("small", 8, abi.GRID_CARTESIAN3D, 6e8, 2000, 1000000),   # optically thick core: grey path + do_kpkt_blackbody
    ("small", 24, abi.GRID_SPHERICAL1D, 0.0, 2000, 2),        # configs[0]-like: 1D spherical shells
    ("small", 16, abi.GRID_SPHERICAL1D, 5e8, 1000, 3),
    ("small", 8, abi.GRID_CYLINDRICAL2D, 0.0, 2500, 2),       # 2D cylindrical (r_cyl, z) grid, grid.cc:2602
    ("tiny", 6, abi.GRID_CYLINDRICAL2D, 5e8, 1500, 5),
])
def test_packets_bit_exact(oracle, preset, ncoord, gridtype, thick_v, npk, budget):
    model, cs, ts, aux = synth.build(preset, ncoord=ncoord, gridtype=gridtype, thick_below_v=thick_v)
    pk0 = synth.make_packets(model, aux, npk, kpkt_fraction=0.2)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, budget)
    parity.compare_packets(pb, pa, 0.0, "kernel bodies vs oracle")
    parity.compare_stats(eb, ea, "kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, "kernel bodies vs oracle")
    assert ea.stats[abi.STAT_X_RPKT_STEPS] > npk  # the run really propagated packets
    assert np.count_nonzero(pa["type"] == abi.TYPE_ESCAPE) > 0


@pytest.mark.parametrize("preset,ncoord,gridtype,npk,budget", [
    ("small", 8, abi.GRID_CARTESIAN3D, 3000, 3),
    ("tiny", 16, abi.GRID_SPHERICAL1D, 2000, 1),
    ("tiny", 6, abi.GRID_CYLINDRICAL2D, 2000, 1000000),
])
def test_gamma_packets_bit_exact(oracle, preset, ncoord, gridtype, npk, budget):
    """TYPE_GAMMA packets (transport_gamma gammapkt.cc:655: Compton scattering incl. the Thomson limit, photoelectric
    absorption, pair production) and their hand-over to the thermal pool (do_ntlepton_deposit -> k-packet) in one
    population with r- and k-packets."""
    model, cs, ts, aux = synth.build(preset, ncoord=ncoord, gridtype=gridtype)
    pk0 = synth.make_packets(model, aux, npk, kpkt_fraction=0.2, gamma_fraction=0.6)
    ngamma = np.count_nonzero(pk0["type"] == abi.TYPE_GAMMA)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, budget)
    parity.compare_packets(pb, pa, 0.0, "gamma: kernel bodies vs oracle")
    parity.compare_stats(eb, ea, "gamma: kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, "gamma: kernel bodies vs oracle")
    st = ea.stats_dict()
    assert st["X_GAMMA_STEPS"] > ngamma and st["NT_STAT_FROM_GAMMA"] > 0.2 * ngamma and st["NT_STAT_TO_KPKT"] == st["NT_STAT_FROM_GAMMA"]
    assert ea.scalars[0] > 0 and ea.scalars[0] == ea.scalars[1]          # every deposit became a k-packet at once
    assert abs(ea.dep_estimator_gamma.sum() / ea.scalars[0] - 1) < 0.2   # path estimator ~ discrete deposition
    assert np.count_nonzero(pa["type"] == abi.TYPE_NTLEPTON_DEPOSITED) == 0
    assert np.count_nonzero((pa["type"] == abi.TYPE_ESCAPE) & (pa["escape_type"] == abi.TYPE_GAMMA)) > 0


@pytest.mark.parametrize("gridtype,ncoord,kw,pkw", [
    (abi.GRID_CARTESIAN3D, 8, {}, {}),
    (abi.GRID_SPHERICAL1D, 16, {}, {}),
    (abi.GRID_CYLINDRICAL2D, 6, {"nts": 0, "t_days": 2.0, "tmin_days": 2.0}, {"early_pellets": True}),  # first timestep
])
def test_all_packet_types_bit_exact(oracle, gridtype, ncoord, kw, pkw):
    """Every packet type of do_packet() (update_packets.cc:257) in one population: pellets (update_pellet: carried with
    the flow, decaying to gamma packets / k-packets / non-thermal particles, or -- in timestep 0 -- already decayed),
    gamma packets, non-thermal pre-deposits and deposits, r-, k- and pre-k-packets."""
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype, **kw)
    pk0 = synth.make_packets(model, aux, 4000, kpkt_fraction=0.1, gamma_fraction=0.2, pellet_fraction=0.6, **pkw)
    npellets = np.count_nonzero(pk0["type"] == abi.TYPE_RADIOACTIVE_PELLET)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3)
    parity.compare_packets(pb, pa, 0.0, "all types: kernel bodies vs oracle")
    parity.compare_stats(eb, ea, "all types: kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, "all types: kernel bodies vs oracle")
    sc = dict(zip(abi.SCALAR_NAMES, ea.scalars))
    e_pkt = pk0["e_cmf"][0]
    ndecayed = npellets - np.count_nonzero(pa["type"] == abi.TYPE_RADIOACTIVE_PELLET) - ea.stats_dict()["K_STAT_FROM_EARLIERDECAY"]
    assert sc["pellet_decays"] == ndecayed > 100
    emitted = sc["gamma_emission"] + sc["positron_emission"] + sc["electron_emission"] + sc["alpha_emission"] + sc["spfission_dep_discrete"]
    assert abs(emitted / (ndecayed * e_pkt) - 1) < 1e-12                      # every decay is counted in exactly one channel
    assert sc["electron_dep_discrete"] == sc["electron_emission"] and sc["alpha_dep_discrete"] == sc["alpha_emission"]
    assert ea.dep_estimator_electron.sum() > 0 and ea.dep_estimator_positron.sum() > 0 and ea.dep_estimator_alpha.sum() > 0
    if kw.get("nts") == 0:
        assert ea.stats_dict()["K_STAT_FROM_EARLIERDECAY"] > 100
    still = pa[pa["type"] == abi.TYPE_RADIOACTIVE_PELLET]
    assert np.all(still["tdecay"] > ts.c.start + ts.c.width) and np.all(still["prop_time"] == ts.c.start + ts.c.width)
    assert not np.any(np.isin(pa["type"], [20, 21, 22, 23, 24]))               # no deposit type survives a call


@pytest.mark.parametrize("gridtype,ncoord,thick_v", [
    (abi.GRID_CARTESIAN3D, 8, 0.0),
    (abi.GRID_SPHERICAL1D, 16, 5e8),
    (abi.GRID_CYLINDRICAL2D, 6, 0.0),
])
def test_kilonova_lte_options_preset_bit_exact(oracle, gridtype, ncoord, thick_v):
    """The packet-path options of artisoptions_kilonova_lte.h (BASELINE.json configs[3]) as a second build of the same
    sources (-DARTIS_PRESET_KILONOVA_LTE, include/artis_options.h): isotropic electron scattering without polarisation,
    DIRECT_COL_HEAT, interpolated photoionisation cross sections, relativistic Doppler factor with the linear frequency
    approximation in the line walk (rpkt.cc:188), 200-point rate-coefficient tables from 500 K, and the time-dependent
    thermalisation of non-thermal particles (update_packets.cc:90). All packet types in one population."""
    P = "kilonova_lte"
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype, thick_below_v=thick_v, options=P)
    pk0 = synth.make_packets(model, aux, 4000, kpkt_fraction=0.15, gamma_fraction=0.15, pellet_fraction=0.4)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=P)
    parity.compare_packets(pb, pa, 0.0, "kilonova_lte: kernel bodies vs oracle")
    parity.compare_stats(eb, ea, "kilonova_lte: kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, "kilonova_lte: kernel bodies vs oracle")
    st = ea.stats_dict()
    assert st["X_RPKT_STEPS"] > 4000 and st["ELECTRON_SCATTERINGS"] > 100 and st["X_MA_JUMPS"] > 10000
    assert np.all(pa["stokes_q"] == 0) and np.all(pa["stokes_u"] == 0)       # POL_ON off: never touched
    assert ea.colheatingestimator.sum() == 0                                   # DIRECT_COL_HEAT: estimator unused
    assert np.count_nonzero(np.isin(pa["type"], [21, 22, 23])) > 0             # particles still slowing down at t_end
    # the options matter: the classic build gives another history for the same input
    pc, ec = pk0.copy(), abi.Estimators(model["npts_nonempty"], model["nbfcontinua_ground"])
    model_c, cs_c, ts_c, aux_c = synth.build("small", ncoord=ncoord, gridtype=gridtype, thick_below_v=thick_v)
    oracle.update_packets(model_c, cs_c, ts_c, pc, ec)
    assert not np.array_equal(pc["nu_cmf"], pa["nu_cmf"])


def test_consecutive_timesteps_bit_exact(oracle):
    """Three consecutive timesteps on the same population (the reference's timestep loop calls update_packets() once per
    timestep): packets that end a timestep in flight, as k-packets or as undecayed pellets continue in the next one."""
    model, cs, ts, aux = synth.build("small", ncoord=8)
    pk0 = synth.make_packets(model, aux, 2500, kpkt_fraction=0.2, gamma_fraction=0.1, pellet_fraction=0.3)
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    pa, pb = pk0.copy(), pk0.copy()
    t = aux["t"]
    for step in range(3):
        tsn = synth.make_timestep(t, width_frac=0.05, vmax=model["vmax"], nts=10 + step)
        ea, eb = abi.Estimators(n, g), abi.Estimators(n, g)
        oracle.update_packets(model, cs, tsn, pa, ea)
        emu.update_packets(model, cs, tsn, pb, eb, budget=3)
        parity.compare_packets(pb, pa, 0.0, f"timestep {step}")
        parity.compare_stats(eb, ea, f"timestep {step}")
        parity.compare_estimators(eb, ea, 1e-11, f"timestep {step}")
        t = tsn.c.start + tsn.c.width
        alive = pa["type"] != abi.TYPE_ESCAPE
        assert np.all(pa["prop_time"][alive] == t)
    assert np.count_nonzero(pa["type"] == abi.TYPE_RADIOACTIVE_PELLET) < np.count_nonzero(pk0["type"] == abi.TYPE_RADIOACTIVE_PELLET)


DEGENERATE = {"oneion": ([(26, 2, 1)], 8, 0.5, 10),            # one ion, one level: no lines, no continua at all
              "twoel": ([(14, 1, 2), (26, 1, 1)], 5, 0.6, 8)}    # a handful of levels, one ground continuum


@pytest.mark.parametrize("name", sorted(DEGENERATE))
def test_degenerate_atomic_data_bit_exact(oracle, name):
    """Empty and nearly empty tables (zero lines, zero bound-free continua, a single cooling term): the paths that loop
    over them must simply do nothing."""
    synth.PRESETS[name] = DEGENERATE[name]
    model, cs, ts, aux = synth.build(name, ncoord=5)
    assert model["nlines"] == (0 if name == "oneion" else model["nlines"])
    pk0 = synth.make_packets(model, aux, 1500, kpkt_fraction=0.3, gamma_fraction=0.1)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3)
    parity.compare_packets(pb, pa, 0.0, name)
    parity.compare_stats(eb, ea, name)
    parity.compare_estimators(eb, ea, 1e-11, name)
    assert ea.stats[abi.STAT_X_RPKT_STEPS] > 1500


def test_budget_independence(oracle):
    """A launch boundary may fall between any two do_packet() calls without changing a packet's history."""
    model, cs, ts, aux = synth.build("tiny", ncoord=6)
    pk0 = synth.make_packets(model, aux, 800, kpkt_fraction=0.3)
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    outs = []
    for budget in (1, 2, 7, 10**6):
        p = pk0.copy()
        emu.update_packets(model, cs, ts, p, abi.Estimators(n, g), budget=budget)
        outs.append(p)
    for p in outs[1:]:
        parity.compare_packets(p, outs[0], 0.0, "budget independence")


def test_cellcache_bit_exact(oracle):
    model, cs, ts, aux = synth.build("small", ncoord=8, thick_below_v=4e8)
    for c in (0, 17, model["npts_nonempty"] - 1):
        a = oracle.cellcache(model, cs, ts, c)
        b = emu.cellcache(model, cs, ts, c)
        for k in a:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), f"cell {c}: {k} differs"


def test_empty_and_untouched_packets(oracle):
    """Edge cases: no packets; packets of types outside do_packet()'s switch are returned untouched;
    packets already at the end of the timestep are not moved."""
    model, cs, ts, aux = synth.build("tiny", ncoord=6)
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    empty = np.zeros(0, dtype=abi.PACKET_DTYPE)
    emu.update_packets(model, cs, ts, empty, abi.Estimators(n, g))
    pk = synth.make_packets(model, aux, 64)
    pk["type"][:16] = 0  # TYPE_NONE
    pk["type"][16:32] = 13  # TYPE_MA: not a state a packet is handed over in
    pk["prop_time"][32:48] = ts.c.start + ts.c.width
    ref = pk.copy()
    emu.update_packets(model, cs, ts, pk, abi.Estimators(n, g))
    for f in abi.PACKET_DTYPE.names:  # (padding bytes are not compared)
        assert pk[f][:48].tobytes() == ref[f][:48].tobytes(), f
    assert np.all((pk["prop_time"][48:] >= ts.c.start + ts.c.width) | (pk["type"][48:] == abi.TYPE_ESCAPE))


def test_search_helpers_match_numpy():
    """partition_point8 / upper_bound_wide (physics.h) against numpy.searchsorted on arrays with ties and at the edges"""
    import ctypes as C
    L = emu.lib()
    for f in (L.artis_emu_upper_bound, L.artis_emu_lower_bound, L.artis_emu_upper_bound_wide, L.artis_emu_upper_bound_blocked6,
              L.artis_emu_upper_bound_blocked16):
        f.argtypes = [C.c_void_p, C.c_int, C.c_double]
        f.restype = C.c_int
    L.artis_emu_closest_transition.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int]
    L.artis_emu_closest_transition.restype = C.c_int
    rng = np.random.default_rng(5)
    for n in [1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 100, 511, 512, 513, 1851, 13619]:
        a = np.sort(rng.integers(0, max(2, n // 2), n).astype(np.float64))  # many ties
        probes = np.concatenate([a[:: max(1, n // 50)], a[:: max(1, n // 50)] + 0.5, [-1.0, a[-1] + 1.0]])
        for v in probes:
            assert L.artis_emu_upper_bound(a.ctypes.data, n, v) == np.searchsorted(a, v, side="right")
            assert L.artis_emu_lower_bound(a.ctypes.data, n, v) == np.searchsorted(a, v, side="left")
            assert L.artis_emu_upper_bound_wide(a.ctypes.data, n, v) == np.searchsorted(a, v, side="right")
            assert L.artis_emu_upper_bound_blocked6(a.ctypes.data, n, v) == np.searchsorted(a, v, side="right")
            assert L.artis_emu_upper_bound_blocked16(a.ctypes.data, n, v) == np.searchsorted(a, v, side="right")
        d = np.ascontiguousarray(a[::-1])  # line list: falling frequencies (rpkt.h:155)
        for v in probes:
            got = L.artis_emu_closest_transition(d.ctypes.data, n, v, 0)
            if v < d[-1]:
                assert got == -1
            else:
                assert got == int(np.sum(d > v))


@pytest.mark.parametrize("gridtype,ncoord,nts", [
    (abi.GRID_CARTESIAN3D, 8, 13),      # past FIRST_NLTE_RADFIELD_TIMESTEP: radfield() reads the fitted bins
    (abi.GRID_SPHERICAL1D, 16, 13),
    (abi.GRID_CARTESIAN3D, 8, 10),      # before it: radfield() is the dilute blackbody (radfield.cc:788)
])
def test_nltenebular_options_preset_bit_exact(oracle, gridtype, ncoord, nts):
    """The packet-path options of artisoptions_nltenebular.h (BASELINE.json configs[2]) as a third build of the same
    sources (-DARTIS_PRESET_NLTENEBULAR): level populations and photoionisation coefficients handed over by the host's
    NLTE solver instead of computed from (T_e, T_R, W) (USE_LUT_PHOTOION off, atomic.h / ltepop.cc:get_levelpop),
    the 256-bin radiation field in radfield() and its J / nuJ bin estimators (radfield.cc:update_estimators), the
    detailed bound-free rate estimators (radfield.cc:update_bfestimators), and the NT_ON channels read from the host's
    Spencer-Fano solution: ionisation / excitation branches of do_ntlepton_deposit() (nonthermal.cc:2529), non-thermal
    excitation and ionisation rates of the macro-atom (macroatom.cc:133, :181, :562) with Auger multi-ionisation."""
    P = "nltenebular"
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype, options=P, nts=nts)
    pk0 = synth.make_packets(model, aux, 4000, kpkt_fraction=0.15, gamma_fraction=0.15, pellet_fraction=0.3)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=P)
    parity.compare_packets(pb, pa, 0.0, "nltenebular: kernel bodies vs oracle")
    parity.compare_stats(eb, ea, "nltenebular: kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, "nltenebular: kernel bodies vs oracle")
    st = ea.stats_dict()
    assert st["X_RPKT_STEPS"] > 4000 and st["X_MA_JUMPS"] > 10000
    assert st["NT_STAT_TO_IONISATION"] > 30 and st["NT_STAT_TO_EXCITATION"] > 10 and st["NT_STAT_TO_KPKT"] > 100
    assert st["MA_STAT_ACTIVATION_NTCOLLION"] == st["NT_STAT_TO_IONISATION"] and st["MA_STAT_INTERNALUPHIGHERNT"] > 20
    assert ea.gammaestimator.sum() == 0 and ea.bfheatingestimator.sum() == 0   # LUT estimators are not kept
    nb = abi.RADFIELDBINCOUNT
    binJ = ea.radfieldbin_J.reshape(-1, nb)
    assert np.all(binJ.sum(axis=1) <= ea.J * (1 + 1e-12))                       # bins cover part of the spectrum
    assert binJ.sum() > 0.5 * ea.J.sum()
    assert np.count_nonzero(ea.bfrate_raw) > 100                                # kept from the first timestep on (radfield.cc:759)
    # the host populations matter: the same input with levelpops at pure Boltzmann gives another history
    cs2_cells = {k: np.array(v) for k, v in cs.d.items()}
    cs2_cells["levelpops"] = cs2_cells["levelpops"] * 1.5
    pc, ec = pk0.copy(), abi.estimators_for(model, P)
    oracle.update_packets(model, abi.CellState(cs2_cells), ts, pc, ec, preset=P)
    assert not np.array_equal(pc["nu_cmf"], pa["nu_cmf"])


def kilonova_like_ejecta(model):
    """the same model with the whole-ejecta scalars of a 0.005 Msun, 0.2 c kilonova: Barnes' inefficiency time scale is
    then 7.4 d and f_p(20 d) ~ 0.2 (with the 1.4 Msun of the synthetic supernova every particle would thermalise)"""
    mtot = 5.0e-3 * 1.98855e33
    return abi.Model({**model.d, "mtot_input": mtot, "ejecta_kinetic_energy": 0.5 * mtot * (0.2 * 2.99792458e10) ** 2})


@pytest.mark.parametrize("options", ["kilonova_barnes", "kilonova_wollaeger"])
def test_analytic_thermalisation_schemes_bit_exact(oracle, options):
    """PARTICLE_THERMALISATION_SCHEME BARNES and WOLLAEGER (update_packets.cc:53-88): a non-thermal particle deposits with
    the analytic efficiency f_p(t) (Barnes et al. 2016: from the ejecta's mass and kinetic energy; Wollaeger et al. 2018:
    from the local density) or leaves the grid with its particle type recorded as escape_type. Built on the kilonova_lte
    options; no options file of the reference selects these schemes."""
    model, cs, ts, aux = synth.build("small", ncoord=8, options=options)
    model = kilonova_like_ejecta(model)
    cs = abi.CellState({**cs.d, "rho": cs.d["rho"] * 1e-4})   # kilonova-like densities: Wollaeger's f_p(rho t) well below 1
    pk0 = synth.make_packets(model, aux, 6000, kpkt_fraction=0.1, gamma_fraction=0.1, pellet_fraction=0.7)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, options + ": kernel bodies vs oracle")
    parity.compare_stats(eb, ea, options + ": kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, options + ": kernel bodies vs oracle")
    esc = pa[pa["type"] == abi.TYPE_ESCAPE]
    nescaped_particles = np.count_nonzero(np.isin(esc["escape_type"], [21, 22, 23]))   # TYPE_NONTHERMAL_PREDEPOSIT_*
    ndeposited = ea.stats_dict()["NT_STAT_TO_KPKT"]
    assert nescaped_particles > 50 and ndeposited > 50
    sc = dict(zip(abi.SCALAR_NAMES, ea.scalars))
    # the escaped particles deposit nothing: the discrete deposition is below the emission by their share
    emitted = sc["electron_emission"] + sc["positron_emission"] + sc["alpha_emission"]
    deposited = sc["electron_dep_discrete"] + sc["positron_dep_discrete"] + sc["alpha_dep_discrete"]
    assert 0.05 < deposited / emitted < 0.98
    # the scheme matters: the time-dependent kilonova_lte build gives another history
    pc, ec = pk0.copy(), abi.estimators_for(model, "kilonova_lte")
    oracle.update_packets(model, cs, ts, pc, ec, preset="kilonova_lte")
    assert not np.array_equal(pc["type"], pa["type"])


@pytest.mark.parametrize("options,gridtype,ncoord,host_tables", [
    ("kilonova_expopac", abi.GRID_CARTESIAN3D, 8, False),
    ("kilonova_expopac", abi.GRID_SPHERICAL1D, 16, True),
    ("classic_expopac_therm", abi.GRID_CARTESIAN3D, 8, False),
    ("classic_expopac_therm", abi.GRID_CYLINDRICAL2D, 6, True),
])
def test_expansion_opacity_builds_bit_exact(oracle, options, gridtype, ncoord, host_tables):
    """RPKT_USE_EXPANSION_OPACITIES (rpkt.cc:221): r-packets walk the 20 A bins of the cell's expansion opacity instead of
    the line list. kilonova_expopac: the bin of the event is re-traced line by line and a bound-bound event activates a
    macro-atom (relativistic Doppler branch). classic_expopac_therm: with RPKT_BOUNDBOUND_THERMALISATION_PROBABILITY = 0.9
    a bound-bound event redistributes the frequency over kappa * B_nu (sample_planck_times_expansion_opacity rpkt.cc:964)
    or scatters, and pre-k-packets emit from the same distribution (kpkt.cc:402). The per-cell tables are those of
    calculate_expansion_opacities() (rpkt.cc:1071) evaluated by oracle and kernel bodies from the cell's level populations,
    or (host_tables) arbitrary ones handed over with the cell state."""
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype, options=options, thick_below_v=4e8 if "therm" in options else 0.0,
                                     host_expopac=host_tables)
    pk0 = synth.make_packets(model, aux, 4000, kpkt_fraction=0.15, gamma_fraction=0.1, pellet_fraction=0.2)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, options + ": kernel bodies vs oracle")
    parity.compare_stats(eb, ea, options + ": kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, options + ": kernel bodies vs oracle")
    st = ea.stats_dict()
    assert st["X_RPKT_STEPS"] > 4000
    if "therm" in options:
        assert st["MA_STAT_ACTIVATION_BB"] == 0                        # a bound-bound event never activates a macro-atom
        rp = pa[pa["type"] == abi.TYPE_RPKT]
        assert np.count_nonzero(rp["trueem_time"] == -1.0) > 100       # thermal redistributions (rpkt.cc:641)
        assert st["ELECTRON_SCATTERINGS"] > 100
    else:
        assert st["MA_STAT_ACTIVATION_BB"] > 100 and st["X_LINES_VISITED"] > 1000   # the re-trace walked lines
    # the options matter: the same input through the line-by-line build gives another history
    base = "kilonova_lte" if options.startswith("kilonova") else "classic"
    pc, ec = pk0.copy(), abi.estimators_for(model, base)
    oracle.update_packets(model, cs, ts, pc, ec, preset=base)
    assert not np.array_equal(pc["nu_cmf"], pa["nu_cmf"])


def test_gamma_products_thermalisation_bit_exact(oracle):
    """TIMEDEPENDENTWITHGAMMAPRODUCTS (constants.h:86): Compton scattering, photoelectric absorption and pair production
    hand the gamma ray's energy to an electron or positron (gammapkt.cc:404, :630, :734) that slows down with the local
    time-dependent scheme; its deposition counts as gamma deposition (update_packets.cc:174) and the path estimator of the
    gamma rays themselves is off (gammapkt.cc:572)."""
    P = "kilonova_gammaproducts"
    model, cs, ts, aux = synth.build("small", ncoord=8, options=P)
    pk0 = synth.make_packets(model, aux, 5000, kpkt_fraction=0.1, gamma_fraction=0.7, pellet_fraction=0.1)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=P)
    parity.compare_packets(pb, pa, 0.0, P + ": kernel bodies vs oracle")
    parity.compare_stats(eb, ea, P + ": kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, P + ": kernel bodies vs oracle")
    st, sc = ea.stats_dict(), dict(zip(abi.SCALAR_NAMES, ea.scalars))
    assert st["NT_STAT_FROM_GAMMA"] > 300
    assert ea.dep_estimator_gamma.sum() > 0 and sc["gamma_dep_discrete"] > 0    # deposited by the products, not by the gamma rays
    # with the plain time-dependent scheme the same gamma rays deposit at once
    pc, ec = pk0.copy(), abi.estimators_for(model, "kilonova_lte")
    oracle.update_packets(model, cs, ts, pc, ec, preset="kilonova_lte")
    assert ec.scalars[abi.SCALAR_NAMES.index("gamma_dep_discrete")] > sc["gamma_dep_discrete"]


@pytest.mark.parametrize("options", ["christinenonthermal", "nltephotospheric", "nltewithoutnonthermal"])
def test_remaining_reference_option_files_bit_exact(oracle, options):
    """The reference's other three options files (artisoptions_christinenonthermal.h,
    artisoptions_nltephotospheric_dynamic_ion_range.h, artisoptions_nltewithoutnonthermal.h): the nltenebular packet path
    with other table grids and frequency limits, 64 / 256 / 512 radiation-field bins, NT_EXCITATION_ON off, bound-free
    estimators for a subset of the continua (LEVEL_HAS_BFEST: estimator index != continuum index, input.cc:932-947),
    dipole scattering with polarisation together with the non-thermal channels, and bound-free cooling weighted by level
    populations (BFCOOLING_USELEVELPOPNOTIONPOP, kpkt.cc:191). All packet types."""
    model, cs, ts, aux = synth.build("small", ncoord=8, options=options, nts=13)
    pk0 = synth.make_packets(model, aux, 4000, kpkt_fraction=0.15, gamma_fraction=0.15, pellet_fraction=0.3)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, options + ": kernel bodies vs oracle")
    parity.compare_stats(eb, ea, options + ": kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, options + ": kernel bodies vs oracle")
    st = ea.stats_dict()
    nb = abi.NEBULAR_FAMILY[options]
    assert ea.radfieldbin_J.size == model["npts_nonempty"] * nb and ea.radfieldbin_J.sum() > 0
    assert st["X_RPKT_STEPS"] > 4000 and st["NT_STAT_TO_IONISATION"] > 30
    assert (st["NT_STAT_TO_EXCITATION"] > 0) == (options == "nltephotospheric")          # NT_EXCITATION_ON
    if options == "nltephotospheric":
        nest = model["nbfestim"]
        assert 0 < nest < model["nbfcontinua"] and ea.bfrate_raw.size == model["npts_nonempty"] * nest
        assert np.count_nonzero(ea.bfrate_raw) > 100
    if options == "nltewithoutnonthermal":
        assert np.any(pa["stokes_q"] != 0)                                                  # POL_ON


@pytest.mark.parametrize("options,gridtype,ncoord", [
    ("kilonova_gamma_barnes", abi.GRID_CARTESIAN3D, 8),
    ("kilonova_gamma_wollaeger", abi.GRID_CARTESIAN3D, 8),
    ("kilonova_gamma_wollaeger", abi.GRID_SPHERICAL1D, 16),
    ("kilonova_gamma_guttman", abi.GRID_CARTESIAN3D, 8),
    ("kilonova_gamma_guttman", abi.GRID_CYLINDRICAL2D, 6),
])
def test_parameterised_gamma_thermalisation_bit_exact(oracle, options, gridtype, ncoord):
    """GAMMA_THERMALISATION_SCHEME BARNES, WOLLAEGER, GUTTMAN (gammapkt.cc:775-866): no gamma-ray transport; a gamma packet
    is absorbed where it is born with the scheme's deposition probability -- from the ejecta's mass and kinetic energy
    (Barnes), from the column density of a radial ray through the grid (Wollaeger), or averaged over 100 random rays
    (Guttman) -- or leaves the grid as a gamma packet. Built on the kilonova_lte options with kilonova-like ejecta."""
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype, options=options)
    mtot = 0.5 * 1.98855e33   # Barnes: t_ineff = 1.4 d sqrt(M / 0.005 Msun) (0.2 c / v_ej) = 14 d, f_gamma(20 d) ~ 0.4
    model = abi.Model({**model.d, "mtot_input": mtot, "ejecta_kinetic_energy": 0.5 * mtot * (0.2 * 2.99792458e10) ** 2,
                       "rho_tmin": model.d["rho_tmin"] * 0.5})                                   # rays of optical depth ~1
    pk0 = synth.make_packets(model, aux, 3000, kpkt_fraction=0.1, gamma_fraction=0.7, pellet_fraction=0.1)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, options + ": kernel bodies vs oracle")
    parity.compare_stats(eb, ea, options + ": kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, options + ": kernel bodies vs oracle")
    esc = pa[pa["type"] == abi.TYPE_ESCAPE]
    nesc_gamma = np.count_nonzero(esc["escape_type"] == abi.TYPE_GAMMA)
    ngamma = np.count_nonzero(pk0["type"] == abi.TYPE_GAMMA)
    assert 0.05 * ngamma < nesc_gamma < 0.95 * ngamma                      # both outcomes occur
    assert np.count_nonzero(pa["type"] == abi.TYPE_GAMMA) == 0              # no gamma packet is left to transport
    sc = dict(zip(abi.SCALAR_NAMES, ea.scalars))
    assert sc["gamma_dep_discrete"] > 0 and abs(ea.dep_estimator_gamma.sum() / sc["gamma_dep_discrete"] - 1) < 1e-12


@pytest.mark.parametrize("options,gridtype,ncoord", [("kilonova_gamma_grey", abi.GRID_CARTESIAN3D, 8),
                                                     ("classic_gamma_xcom", abi.GRID_CARTESIAN3D, 8),
                                                     ("classic_gamma_xcom", abi.GRID_SPHERICAL1D, 16)])
def test_gamma_opacity_options_bit_exact(oracle, options, gridtype, ncoord):
    """GAMMA_USE_KAPPA_GREY (gammapkt.cc:266, :420, :517, :553: one grey absorption opacity, every interaction deposits the
    packet) and USE_XCOM_GAMMAPHOTOION (:443-495: the photoelectric opacity summed over the elements from tabulated cross
    sections, log-log interpolated, clamped beyond the table, elements without data skipped)."""
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype, options=options)
    pk0 = synth.make_packets(model, aux, 4000, kpkt_fraction=0.1, gamma_fraction=0.7, pellet_fraction=0.1)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, options + ": kernel bodies vs oracle")
    parity.compare_stats(eb, ea, options + ": kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, options + ": kernel bodies vs oracle")
    st = ea.stats_dict()
    assert st["X_GAMMA_STEPS"] > 3000 and st["NT_STAT_FROM_GAMMA"] > 200 and ea.dep_estimator_gamma.sum() > 0
    absorbed = pa[np.isin(pa["absorptiontype"], [-3, -4, -5])]
    if "grey" in options:
        assert np.all(absorbed["absorptiontype"] == -4)       # only the grey "photoelectric" channel exists
    base = "kilonova_lte" if options.startswith("kilonova") else "classic"
    pc, ec = pk0.copy(), abi.estimators_for(model, base)
    oracle.update_packets(model, cs, ts, pc, ec, preset=base)
    assert not np.array_equal(pc["type"], pa["type"]) or not np.array_equal(pc["prop_time"], pa["prop_time"])


@pytest.mark.parametrize("gridtype,ncoord", [(abi.GRID_CARTESIAN3D, 8), (abi.GRID_SPHERICAL1D, 16)])
def test_detailed_line_estimators_bit_exact(oracle, gridtype, ncoord):
    """DETAILED_LINE_ESTIMATORS_ON on top of the nltenebular options: every packet that redshifts through a line with its
    own estimator adds prop_time * c * e_cmf / nu_cmf to it and counts itself (radfield.cc:773; rpkt.cc:173-207: also at
    the line it is absorbed in), and the radiative excitation rate of such a line uses the host's normalised intensity
    instead of the binned field (macroatom.cc:628)."""
    P = "nltenebular_lineest"
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype, options=P, nts=13)
    pk0 = synth.make_packets(model, aux, 4000, kpkt_fraction=0.15, gamma_fraction=0.1, pellet_fraction=0.2)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=P)
    parity.compare_packets(pb, pa, 0.0, P + ": kernel bodies vs oracle")
    parity.compare_stats(eb, ea, P + ": kernel bodies vs oracle")
    assert np.array_equal(ea.Jb_lu_contribcount, eb.Jb_lu_contribcount)
    parity.compare_estimators(eb, ea, 1e-11, P + ": kernel bodies vs oracle")
    nl = model["detailed_linecount"]
    assert ea.Jb_lu_raw.size == model["npts_nonempty"] * nl and ea.Jb_lu_contribcount.sum() > 1000
    assert np.array_equal(ea.Jb_lu_raw > 0, ea.Jb_lu_contribcount > 0)
    # a seventh of the lines have an estimator; a line counts when the packet reaches it (not when the walk ends before it)
    frac = ea.Jb_lu_contribcount.sum() / ea.stats_dict()["X_LINES_VISITED"]
    assert 0.01 < frac < 0.3
    # the host's line intensities matter: the same input through the plain nltenebular build gives another history
    pc, ec = pk0.copy(), abi.estimators_for(model, "nltenebular")
    oracle.update_packets(model, cs, ts, pc, ec, preset="nltenebular")
    assert not np.array_equal(pc["nu_cmf"], pa["nu_cmf"])


# grid type each of the reference's CI set-ups runs on (tests/setup_<script>.sh: the model it links and GRID_TYPE_OVERRIDE)
CI_GRIDS = {"ci_kilonova": (abi.GRID_CYLINDRICAL2D, 6), "ci_kilonova_barnes": (abi.GRID_CYLINDRICAL2D, 6),
            "ci_kilonova_expopac": (abi.GRID_CYLINDRICAL2D, 6), "ci_kilonova_xcom": (abi.GRID_CYLINDRICAL2D, 6),
            "ci_nebular": (abi.GRID_CARTESIAN3D, 8), "ci_nebular_limitbfest": (abi.GRID_CARTESIAN3D, 8),
            "ci_nltephotospheric": (abi.GRID_SPHERICAL1D, 16),
            "ci_classic_vpkt": (abi.GRID_CARTESIAN3D, 8), "ci_classic_vpkt_expopac": (abi.GRID_CARTESIAN3D, 8)}


def ci_case(options):
    """model, cell state, timestep and a population of every packet type for one of the reference's CI option sets, with
    what the set's physics needs to be exercised (a light, fast ejecta for the Barnes efficiencies)"""
    gridtype, ncoord = CI_GRIDS[options]
    # (virtual packets: a timestep inside the spectra's own window of 3-8 days, vpkt.h:30)
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype, options=options, nts=13,
                                     **({"t_days": 5.0, "thick_below_v": 3e8} if "vpkt" in options else {}))
    if options == "ci_kilonova_barnes":
        mtot = 5.0e-3 * 1.98855e33
        model = abi.Model({**model.d, "mtot_input": mtot, "ejecta_kinetic_energy": 0.5 * mtot * (0.2 * 2.99792458e10) ** 2})
        cs = abi.CellState({**cs.d, "rho": cs.d["rho"] * 1e-4})
    return model, cs, ts, aux


@pytest.mark.parametrize("options", sorted(abi.CI_PRESETS))
def test_reference_ci_option_sets_bit_exact(oracle, options):
    """The option COMBINATIONS the reference's own CI runs (tests/setup_*.sh; presets ci_* of include/artis_options.h, each
    pinned against the options file its script makes): kilonova_lte on the 20-point 1000-20000 K tables, with particle AND
    gamma thermalisation after Barnes, with expansion opacities and thermalisation probability 1 (no random number drawn
    at rpkt.cc:626), with XCOM photoelectric opacities over per-cell mean atomic weights; nltenebular on its CI tables with
    FIRST_NLTE_RADFIELD_TIMESTEP 7, and with bound-free estimators for a subset of the levels; nltephotospheric with 24
    radiation-field bins. On the grid type the CI set-up uses; all packet types."""
    model, cs, ts, aux = ci_case(options)
    pk0 = synth.make_packets(model, aux, 4000, kpkt_fraction=0.15, gamma_fraction=0.2, pellet_fraction=0.3)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, options + ": kernel bodies vs oracle")
    parity.compare_stats(eb, ea, options + ": kernel bodies vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, options + ": kernel bodies vs oracle")
    st = ea.stats_dict()
    assert st["X_RPKT_STEPS"] > 3000
    if options == "ci_kilonova_barnes":
        esc = pa[pa["type"] == abi.TYPE_ESCAPE]
        assert np.count_nonzero(np.isin(esc["escape_type"], [21, 22, 23])) > 30      # particles that left without thermalising
        gam = pk0["type"] == abi.TYPE_GAMMA
        assert st["X_GAMMA_STEPS"] <= np.count_nonzero(gam) + np.count_nonzero(pk0["type"] == abi.TYPE_RADIOACTIVE_PELLET)  # one call each: deposited or gone, no transport
    if options == "ci_kilonova_expopac":
        assert st["MA_STAT_ACTIVATION_BB"] == 0 and st["X_LINES_VISITED"] == 0        # thermalisation, never a macro-atom; no line walk
    if options == "ci_kilonova_xcom":
        assert st["X_GAMMA_STEPS"] > 1000
    if options in abi.NEBULAR_FAMILY:
        assert ea.radfieldbin_J.size == model["npts_nonempty"] * abi.NEBULAR_FAMILY[options] and ea.radfieldbin_J.sum() > 0
    if options in ("ci_nebular_limitbfest", "ci_nltephotospheric"):
        assert 0 < model["nbfestim"] < model["nbfcontinua"] and np.count_nonzero(ea.bfrate_raw) > 50
    if "vpkt" in options:
        # virtual packets (vpkt.cc): created at every emission / electron scattering in a thin cell, escapes of all three
        # origins, spectra of all observers and opacity choices filled, Q/U only from scattered packets
        created, esc_r, esc_k, esc_ma = (int(ea.stats[i]) for i in (abi.STAT_X_VPKT_CREATED, abi.STAT_X_VPKT_CREATED + 1,
                                                                     abi.STAT_X_VPKT_CREATED + 2, abi.STAT_X_VPKT_CREATED + 3))
        assert created > 20000 and esc_r > 1000 and esc_k > 20 and esc_ma > 200 and created > esc_r + esc_k + esc_ma
        v = ea.vspecpol.reshape(abi.VSPEC_TIMEBINS, 3 * 4, abi.VSPEC_NUBINS, 3)
        per_comb = v[..., 0].sum(axis=(0, 2))
        assert np.all(per_comb > 0) and np.all(v[..., 0] >= 0) and np.abs(v[..., 1]).sum() > 0
        full, nolines = per_comb[0::4], per_comb[1::4]
        assert np.all(nolines > full)                       # switching the line opacity off lets more through
        if options == "ci_classic_vpkt":
            assert ea.vgrid_flux.sum() > 0                  # the velocity-grid map
            assert np.all(per_comb[3::4] >= full)           # without the lines of one element


@pytest.mark.parametrize("filters", ["1", "0"])
@pytest.mark.parametrize("options,gridtype,ncoord", [("classic", abi.GRID_CARTESIAN3D, 8), ("nltenebular", abi.GRID_SPHERICAL1D, 16),
                                                     ("ci_classic_vpkt", abi.GRID_CARTESIAN3D, 6)])
def test_undecided_searches_through_the_slow_path_bit_exact(oracle, monkeypatch, options, gridtype, ncoord, filters):
    """What k_thermal does with a search its filters cannot decide (ARTIS_EMU_SPLIT=1: ma_jump<true>, do_kpkt<true>): the packet
    leaves with the draw in its pend fields (PEND_MA_SEARCH / _RADSEARCH / PEND_KPKT_COLLEXC) and the slow path re-adds the sums and
    carries on. With the filters as they are (3e-4 of the transitions take that way) and with ARTIS_EMU_MAFILTERS=0 (EVERY transition,
    radiative de-excitation and cooling draw does): bit-exact against the oracle."""
    monkeypatch.setenv("ARTIS_EMU_SPLIT", "1")
    monkeypatch.setenv("ARTIS_EMU_MAFILTERS", filters)
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype, options=options, t_days=5.0 if "vpkt" in options else 20.0)
    pk0 = synth.make_packets(model, aux, 2500 if filters == "1" else 600, kpkt_fraction=0.3)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, "searches through the slow path vs oracle")
    parity.compare_stats(eb, ea, "searches through the slow path vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, "searches through the slow path vs oracle")


@pytest.mark.parametrize("options,preset,ncoord,gridtype,npk", [
    ("classic", "small", 8, abi.GRID_CARTESIAN3D, 3000),
    ("classic", "w7", 5, abi.GRID_CARTESIAN3D, 600),
    ("kilonova_lte", "small", 16, abi.GRID_SPHERICAL1D, 1500),
    ("nltenebular", "small", 6, abi.GRID_CARTESIAN3D, 1500),
])
def test_undecided_draws_readd_the_sums_bit_exact(oracle, monkeypatch, options, preset, ncoord, gridtype, npk):
    """Round 4: the records hold filters only; a draw a filter cannot decide re-adds the cumulative sums from the transitions'
    terms (physics.h ma_exact_search, kpkt_collexc_exact). With ARTIS_EMU_MAFILTERS=0 EVERY macro-atom transition,
    radiative de-excitation and collisional-excitation cooling draw takes that path: bit-exact against the oracle, whose
    sums are the reference's stored arrays."""
    monkeypatch.setenv("ARTIS_EMU_MAFILTERS", "0")
    model, cs, ts, aux = synth.build(preset, ncoord=ncoord, gridtype=gridtype, options=options)
    pk0 = synth.make_packets(model, aux, npk, kpkt_fraction=0.3)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, "re-added sums vs oracle")
    parity.compare_stats(eb, ea, "re-added sums vs oracle")
    assert ea.stats[abi.STAT_X_MA_JUMPS] > 10 * npk and ea.stats_dict()["K_STAT_TO_MA_COLLEXC"] > 0


@pytest.mark.parametrize("options,gridtype,ncoord,t_days", [
    ("classic", abi.GRID_CARTESIAN3D, 8, 20.0),
    ("kilonova_expopac", abi.GRID_CYLINDRICAL2D, 6, 20.0),     # k_expopac reads the population factors too
    ("ci_classic_vpkt", abi.GRID_CARTESIAN3D, 6, 5.0),         # ... and the virtual packets' line walk
])
def test_line_population_factors_formed_on_the_fly_bit_exact(oracle, monkeypatch, options, gridtype, ncoord, t_days):
    """The engine drops the cell cache's line_dpop rows (8 bytes per line and cell) when the cache would not fit one tile with them:
    the line walk then forms B_lu n_l - B_ul n_u from the line record and the two level populations where it reads it (physics.h
    line_dpop_at). ARTIS_EMU_DPOP=0 runs the kernel bodies that way: bit-exact against the oracle."""
    monkeypatch.setenv("ARTIS_EMU_DPOP", "0")
    model, cs, ts, aux = synth.build("small", ncoord=ncoord, gridtype=gridtype, options=options, t_days=t_days)
    pk0 = synth.make_packets(model, aux, 2000, kpkt_fraction=0.2)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, "population factors on the fly vs oracle")
    parity.compare_stats(eb, ea, "population factors on the fly vs oracle")
    parity.compare_estimators(eb, ea, 1e-11, "population factors on the fly vs oracle")
    assert ea.stats[abi.STAT_X_LINES_VISITED] > (10 * len(pk0) if "expopac" not in options else 1000)


def test_macroatom_filters_decide_what_the_f64_comparison_decides():
    """tables.h "FILTERS": 4e6 random cumulative lists and 24-bit draws, a quarter of them with a value within two ulp of
    z * whole, another quarter with a value within three ulp of the lower edge of the draw's filter cell and the draw 0 ... 4 units of
    2^-24 above that edge (the bound of round 5's rule: an entry one unit below the draw's is counted where the draw's nine lower bits
    are >= 2); whenever the 15-bit filter does not hand the draw to the f64 path, its count is the f64 comparison's"""
    import ctypes as C

    L = emu.lib()
    L.artis_emu_mafilter_selftest.restype = C.c_int64
    L.artis_emu_mafilter_selftest.argtypes = [C.c_int64, C.c_uint64, C.POINTER(C.c_int64)]
    namb = C.c_int64(0)
    n = 4_000_000
    assert L.artis_emu_mafilter_selftest(n, 12345, C.byref(namb)) == 0
    # ambiguous: the first planted quarter, ~3/4 of the second (its value lies in the draw's own cell, or one below with the lower bits < 2),
    # plus ~8 entries x 1 of 32768 values of zi for the rest
    assert 0.42 * n <= namb.value < 0.46 * n, namb.value


def test_macroatom_fine_bytes_decide_what_the_f64_comparison_decides():
    """tables.h "FINE BYTES" (round 6): 6e6 random lines of 7 cumulative values and 24-bit draws on the kernels' own mafilt_quant23() /
    mafilt_count_fine(); a third with a value within three ulp of z * whole, a third with a value within three ulp of an edge of the draw's
    23-bit cell or its neighbours (the bounds of "u >= 2 q + 3 counts, u <= 2 q - 1 does not"); half of the lists reach the whole at their end.
    Whenever the 23-bit count does not leave the draw to the re-added sums it equals the f64 comparison's; the line's 15-bit entry is q23 >> 8;
    the 15-bit count, where it decides, agrees."""
    import ctypes as C

    L = emu.lib()
    L.artis_emu_mafilter_fine_selftest.restype = C.c_int64
    L.artis_emu_mafilter_fine_selftest.argtypes = [C.c_int64, C.c_uint64, C.POINTER(C.c_int64)]
    namb = C.c_int64(0)
    n = 6_000_000
    assert L.artis_emu_mafilter_fine_selftest(n, 4242, C.byref(namb)) == 0
    # undecided: the first planted third (the value IS the draw's product), ~3/5 of the second (k = -1 ... 1 lands in {2q, 2q+1, 2q+2}: roughly), and
    # ~7 entries x 3 of 2^24 draws of the rest: nothing
    assert 0.33 * n <= namb.value < 0.60 * n, namb.value


@pytest.mark.parametrize("options,preset,ncoord,npk", [("classic", "small", 8, 3000), ("nltenebular", "small", 6, 1500)])
def test_kpkt_draws_by_bisection_bit_exact(oracle, monkeypatch, options, preset, ncoord, npk):
    """ARTIS_AMD_COOLGUIDE=0: the two draws of a k-packet step by std::upper_bound's bisection (kpkt.cc:430-447, rounds 1-4) instead of by the
    guide tables every other test of this file runs with: bit-exact against the oracle as well."""
    monkeypatch.setenv("ARTIS_AMD_COOLGUIDE", "0")
    model, cs, ts, aux = synth.build(preset, ncoord=ncoord, options=options)
    pk0 = synth.make_packets(model, aux, npk, kpkt_fraction=0.5)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, "k-packet draws by bisection vs oracle")
    parity.compare_stats(eb, ea, "k-packet draws by bisection vs oracle")
    assert ea.stats[abi.STAT_X_KPKT_STEPS] > npk


def test_cooling_guides_give_the_bisections_index():
    """tables.h "COOLING GUIDES": 2e5 random cumulative lists (equal neighbours, entries that add nothing, a dominant term in most) with guides
    of 1 ... 512 ranges; for random 24-bit draws and for the first and last draw of the ranges the guided look-up of do_kpkt() returns the
    index upper_bound_d() returns (kpkt.cc:430-447's std::upper_bound); most draws are decided by the two guide entries alone"""
    import ctypes as C

    L = emu.lib()
    L.artis_emu_coolguide_selftest.restype = C.c_int64
    L.artis_emu_coolguide_selftest.argtypes = [C.c_int64, C.c_uint64, C.POINTER(C.c_int64)]
    noread = C.c_int64(0)
    assert L.artis_emu_coolguide_selftest(200_000, 20261004, C.byref(noread)) == 0
    assert noread.value > 0.5 * 200_000 * 64


@pytest.mark.parametrize("options,preset,ncoord,gridtype,npk", [
    ("classic", "w7", 5, abi.GRID_CARTESIAN3D, 600),
    ("classic", "small", 8, abi.GRID_CYLINDRICAL2D, 2500),
    ("kilonova_lte", "small", 16, abi.GRID_SPHERICAL1D, 1500),
    ("nltenebular", "small", 6, abi.GRID_CARTESIAN3D, 1500),
    ("ci_classic_vpkt", "small", 6, abi.GRID_CARTESIAN3D, 1500),
])
def test_on_demand_macroatom_records_bit_exact(oracle, monkeypatch, options, preset, ncoord, gridtype, npk):
    """Round 5 (tables.h "ON-DEMAND RECORDS"; the reference: calc_rates_if_needed macroatom.cc:398-417): static macro-atom records for the
    lowest third of every ion's levels only, a cold level's record filled in its cell's pool when a packet first reaches it there (the
    slow path's ma_slow_fill with the sequential forms of the population), the k-packet step deciding on the re-added sums where a cold
    level has no record yet. Bit-exact against the oracle, also with every decision on the f64 sums; and test_cellcache_bit_exact's view of a
    cell -- every cold record filled the on-demand way -- holds against the oracle's stored arrays."""
    monkeypatch.setenv("ARTIS_AMD_MA_HOTFRAC", "0.3")
    monkeypatch.setenv("ARTIS_AMD_MA_POOLFRAC", "1")
    model, cs, ts, aux = synth.build(preset, ncoord=ncoord, gridtype=gridtype, options=options, **({"t_days": 5.0} if "vpkt" in options else {}))
    pk0 = synth.make_packets(model, aux, npk, kpkt_fraction=0.3)
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, "on-demand records vs oracle")
    parity.compare_stats(eb, ea, "on-demand records vs oracle")
    assert ea.stats[abi.STAT_X_MA_JUMPS] > 10 * npk
    monkeypatch.setenv("ARTIS_EMU_MAFILTERS", "0")
    pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
    parity.compare_packets(pb, pa, 0.0, "on-demand records, f64 decisions vs oracle")
    if options == "classic" and preset == "small":
        monkeypatch.delenv("ARTIS_EMU_MAFILTERS")
        c = int(model["npts_nonempty"]) // 2
        got, want = emu.cellcache(model, cs, ts, c), oracle.cellcache(model, cs, ts, c)
        for k in ("maprocessrates", "matrans"):
            assert np.array_equal(got[k], want[k]), k
        # a pool that cannot hold the cold records the packets reach is emptied when it is used up and filled again on demand: it costs fills,
        # never an answer
        monkeypatch.setenv("ARTIS_AMD_MA_POOLFRAC", "0.02")
        pa, pb, ea, eb = _run_both(oracle, model, cs, ts, pk0, 3, options=options)
        parity.compare_packets(pb, pa, 0.0, "on-demand records in a pool that is used up again and again vs oracle")
        parity.compare_stats(eb, ea, "on-demand records in a pool that is used up again and again vs oracle")
        L = emu.lib()
        L.artis_emu_last_pool_resets.restype = __import__("ctypes").c_int
        assert L.artis_emu_last_pool_resets() > 3
        # ... and one that cannot hold a single record of the level a packet reaches is an error
        monkeypatch.setenv("ARTIS_AMD_MA_POOLFRAC", "0")
        with pytest.raises(RuntimeError, match="46"):
            _run_both(oracle, model, cs, ts, pk0, 3, options=options)
