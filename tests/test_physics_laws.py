"""Physics checks that do not use the oracle: laws any correct implementation of the reference's packet path has to obey,
evaluated on the packets the engine returns (GPU, `-m gpu`) and on the kernel bodies compiled for x86 (CPU suite).

1. Sobolev escape probability (rpkt.cc:75 get_tau_sobolev, rpkt.cc:106 get_possible_event): an r-packet that redshifts
   through ONE line of Sobolev optical depth tau is absorbed with probability 1 - exp(-tau).
2. Grey scattering (rpkt.cc:575-580): in a uniform optically thick medium the number of scatterings a packet makes in a
   time dt is Poisson distributed with mean kappa * rho * c * dt.
3. Adiabatic losses (vectors.h:119 move_pkt_withtime + rpkt.cc:331 scattering in the comoving frame): radiation trapped
   in homologously expanding matter loses energy as 1/t: <e_cmf(t1)/e_cmf(t0)> = t0/t1.
4. Thomson scattering (rpkt.cc:331 electron_scatter_rpkt with DIPOLE, vectors.h:325): unpolarised radiation scatters with
   the phase function 3/(16 pi) (1 + mu^2): <mu> = 0, <mu^2> = 2/5 (isotropic would give 1/3).
5. Free-free emission (kpkt.cc:495-515): a k-packet that cools by free-free emission emits at h nu / k T_e drawn from
   exp(-x): mean 1, P(x > 1) = 1/e, P(x > 2) = 1/e^2.
6. Cooling channels (kpkt.cc:57-260, :430-470): the channels k-packets of one cell leave through (free-free, free-bound,
   collisional excitation, collisional ionisation) occur in the ratio of the cell's summed cooling rates of each kind.
7. Free-bound emission (ratecoeff.cc:563 select_continuum_nu): the photons of one recombination continuum are distributed
   as sigma_bf(nu) nu^3 exp(-h (nu - nu_edge) / k T_e) (the emissivity of Milne's relation): mean frequency and the
   fraction above the median of that law.
8. Compton scattering (gammapkt.cc:266-420): the energy ratio f = E'/E of the gamma packets that survive a scattering
   follows f * dsigma/df of Klein and Nishina, dsigma/df ~ f + 1/f - 1 + cos^2(theta(f)): mean and quartile fractions.
9. Lorentz transformation of an emission (vectors.h:70 angle_ab, :91 calculate_doppler_nucmf_on_nurf; kpkt.cc:495-560 and
   macroatom.cc:283-330 emit isotropically in the comoving frame): matter moving with beta = v/c radiates, seen from the
   rest frame, the momentum beta * E / c along v and the energy E (1 + O(beta^2)): sum(e_rf dir.beta) = sum(e_cmf beta^2).
10. Reciprocity of the macro-atom in thermodynamic equilibrium (macroatom.cc:64-190 rate coefficients, :385-577 the walk;
   kpkt.cc:57-260 the thermal pool). With LTE populations, T_R = T_e = T_J and W = 1 every process balances its inverse, the
   macro-atom's internal "conductances" n_i R_ij eps are symmetric, and energy absorbed in line a and re-emitted in line e
   flows at the same rate as from e to a (Lucy 2002): N(a -> e) = N(e -> a) for Planck-distributed packets -- for every pair
   of lines, whatever lies between (internal jumps up and down, collisional de-excitation into the thermal pool and back,
   ionisation and recombination). Would catch a wrong Boltzmann or statistical-weight factor in any rate coefficient, a
   missing stimulated term, a channel drawn with the wrong weight: none of the macro-atom's rates is pinned otherwise.
11. The photon number of a packet and the energy in the Hubble flow: e_cmf / nu_cmf = e_rf / nu_rf for every r-packet (both
   transform with one Doppler factor, vectors.h:91), and e_cmf * t is conserved by every step of the path -- free flight in
   homologous expansion, scattering, the macro-atom and the thermal pool (kpkt.cc:441 e_cmf *= t / t_new) -- to the first
   order in v/c: sum(e_cmf t) after a timestep equals the sum before within (v/c)(dt/t). Would catch an energy leak, a missing
   adiabatic loss, an emission in the wrong frame.
12. Detailed balance channel by channel (round 6). In the same equilibrium medium every process runs as often as its inverse -- counted
   by the reference's own event counters (stats.h), which see EVERY event, not only a packet's last one, so the law holds at any optical
   depth: bound-bound absorptions = emissions; photoionisations (macro-atom activations + k-packets made by bound-free absorption) =
   radiative recombinations (macro-atom deactivations + k-packets cooling free-bound), and separately the macro-atom and the thermal halves
   of it (bound-free heating = free-bound cooling); free-free absorptions = emissions; collisional excitations out of the thermal pool =
   collisional de-excitations into it; ionisations inside the macro-atom (internal up to the higher ion) = recombinations (internal down to
   the lower ion). Would catch a wrong Milne relation or Saha factor in any bound-free rate (rpkt.cc:721 opacity with its stimulated
   correction, macroatom.cc:141-190, kpkt.cc:123-190 cooling, :519 emission), a wrong split of an absorbed photon between ionisation and
   heating (rpkt.cc:459-480), a cooling channel with the wrong weight.
13. Gamma-ray energy deposition counted twice (round 6). The path-length estimator of the deposited energy (gammapkt.cc:568
   update_gamma_dep: chi_cmf * e_rf * ds * doppler^2 with the mean Compton loss, the photoelectric opacity and the kinetic share of a
   pair production) and the energy of the packets that actually thermalise in the same span (indivisible packets: Compton scattering
   deposits a packet with probability 1 - f, gammapkt.cc:266-420, :742-764) estimate the same number: their sums over the grid agree
   within the Monte Carlo noise of the second. And every gamma packet is escaped, deposited or still in flight, with
   e_rf(escaped) + e_cmf(deposited) + e_rf(in flight) = the emitted energy to first order in v/c.
Each would catch a misreading of the transport loop that oracle and kernels share (same hand, same reading): a wrong
phase function or frame, a wrong sampling law, a channel that is drawn with the wrong weight.
"""
import numpy as np
import pytest

import hostemu_binding as emu
from artis_amd import abi, synth

CLIGHT = 2.99792458e10
HCLIGHTOVERFOURPI = 6.6260755e-27 * CLIGHT / (4 * np.pi)


def _backend_emu(model, cs, ts, pk):
    est = abi.Estimators(model["npts_nonempty"], model["nbfcontinua_ground"])
    emu.update_packets(model, cs, ts, pk, est)
    return est


def _backend_gpu(model, cs, ts, pk):
    import torch

    assert torch.cuda.is_available()
    from artis_amd import engine

    est = abi.Estimators(model["npts_nonempty"], model["nbfcontinua_ground"])
    eng = engine.Engine(model)
    eng.set_cellstate(cs, ts)
    eng.update_packets(pk, est)
    eng.close()
    return est


def _rpackets(model, aux, cellindex, n, nu_cmf, rng, margin=0.05):
    """r-packets at random places well inside one Cartesian cell, isotropic directions, given comoving frequency"""
    d = model.d
    t, tmin = aux["t"], d["tmin"]
    nc = int(d["ncoordgrid"][0])
    cmin = d["coord_pos_min_tmin"][0]
    dx = 2 * d["rmax"] / nc
    ix, iy, iz = cellindex % nc, (cellindex // nc) % nc, cellindex // (nc * nc)
    u = margin + (1 - 2 * margin) * rng.random((n, 3))
    pos = np.stack([cmin[ix] + u[:, 0] * dx, cmin[iy] + u[:, 1] * dx, cmin[iz] + u[:, 2] * dx], axis=1) * (t / tmin)
    mu = 2 * rng.random(n) - 1
    ph = 2 * np.pi * rng.random(n)
    s = np.sqrt(1 - mu**2)
    dirs = np.stack([s * np.cos(ph), s * np.sin(ph), mu], axis=1)
    pk = np.zeros(n, dtype=abi.PACKET_DTYPE)
    pk["pos"], pk["dir"], pk["prop_time"] = pos, dirs, t
    dop = 1. - (dirs * pos / t).sum(axis=1) / CLIGHT  # calculate_doppler_nucmf_on_nurf vectors.h:92 (classic options)
    pk["nu_cmf"], pk["e_cmf"] = nu_cmf, 1e40
    pk["nu_rf"], pk["e_rf"] = nu_cmf / dop, 1e40 / dop
    pk["type"], pk["cellindex"] = abi.TYPE_RPKT, cellindex
    pk["next_trans"] = 0
    pk["emissiontype"] = pk["trueemissiontype"] = abi.EMTYPE_NOTSET
    pk["absorptiontype"] = -77
    pk["em_pos"], pk["trueem_pos"] = np.nan, np.nan
    pk["escape_time"], pk["tdecay"], pk["pellet_decaytype"], pk["pellet_nucindex"] = -1., -1., -1, -1
    pk["number"] = np.arange(n)
    abi.seed_packet_rng(pk, 424242)
    return pk


def _sobolev_case(oracle):
    """a (cell, line) of the synthetic model with 0.3 < tau_Sobolev < 3 and no other line within 0.1 % redwards"""
    model, cs, ts, aux = synth.build("tiny", ncoord=6, width_frac=1e-4)
    d = model.d
    nu = np.asarray(d["line_nu"])
    lo, up = np.asarray(d["line_uniquelevelindex_lower"]), np.asarray(d["line_uniquelevelindex_upper"])
    B_lu, B_ul = np.asarray(d["line_B_lu"], dtype=np.float64), np.asarray(d["line_B_ul"], dtype=np.float64)
    gap_ok = np.ones(len(nu), dtype=bool)
    gap_ok[:-1] = (nu[:-1] - nu[1:]) / nu[:-1] > 1e-3
    gap_ok[1:] &= (nu[:-1] - nu[1:]) / nu[1:] > 1e-5   # and none just bluewards, between the packets and the line
    for c in range(model["npts_nonempty"] // 2, model["npts_nonempty"]):
        pops = oracle.cellcache(model, cs, ts, c)["levelpops"]  # LTE level populations (ltepop.cc:412), pinned elsewhere
        tau = (B_lu * pops[lo] - B_ul * pops[up]) * HCLIGHTOVERFOURPI * aux["t"]
        cand = np.nonzero(gap_ok & (tau > 0.3) & (tau < 3.) & (nu > 2e14) & (nu < 4e15))[0]
        if len(cand):
            return model, cs, ts, aux, c, int(cand[0]), float(tau[cand[0]])
    raise AssertionError("no suitable line in the synthetic model")


def _check_sobolev(oracle, backend, n):
    model, cs, ts, aux, c, L, tau = _sobolev_case(oracle)
    cellindex = int(aux["nonempty_cellindex"][c])
    rng = np.random.default_rng(5)
    pk = _rpackets(model, aux, cellindex, n, model.d["line_nu"][L] * (1 + 1e-7), rng)
    backend(model, cs, ts, pk)
    p_abs = np.mean(pk["absorptiontype"] == L)
    expect = 1. - np.exp(-tau)
    sigma = np.sqrt(expect * (1 - expect) / n)
    assert abs(p_abs - expect) < 4 * sigma + 2e-3, (p_abs, expect, tau, L, c)
    # the packets that were not absorbed by the line have seen nothing else in this short a timestep
    other = (pk["absorptiontype"] != L) & (pk["absorptiontype"] != -77)
    assert np.mean(other) < 2e-3


def _uniform_grey_model(ncoord=9):
    """every cell optically thick (grey), uniform density: a box of trapped radiation in homologous expansion"""
    atomic = synth.make_atomic(seed=1, elements=synth.PRESETS["tiny"][0], nlevels_per_ion=synth.PRESETS["tiny"][1],
                               line_fraction=synth.PRESETS["tiny"][2], nphixspoints=synth.PRESETS["tiny"][3])
    grid, cells, aux = synth.make_grid_and_cells(atomic, ncoord=ncoord, thick_below_v=1e30)
    cells["rho"] = np.full_like(cells["rho"], 2.0e-12)
    md = {k: v for k, v in atomic.items() if not k.startswith("_")}
    md.update(grid)
    model, cs = abi.Model(md), abi.CellState(cells)
    ts = synth.make_timestep(aux["t"], width_frac=0.05, vmax=grid["vmax"])
    return model, cs, ts, aux, 0.1 * 2.0e-12  # chi_grey = kappagrey * rho


def _check_grey(backend, n):
    model, cs, ts, aux, chi = _uniform_grey_model()
    nc = 9
    centre = (nc // 2) * (1 + nc + nc * nc)  # the central cell: v < 0.1 vmax, (v/c)^2 corrections < 1e-4
    rng = np.random.default_rng(11)
    pk = _rpackets(model, aux, centre, n, 1e15, rng, margin=0.3)
    e0 = pk["e_cmf"].copy()
    backend(model, cs, ts, pk)
    assert np.all(pk["type"] == abi.TYPE_RPKT) and np.all(pk["prop_time"] == ts.c.start + ts.c.width)
    lam = chi * CLIGHT * ts.c.width
    ns = pk["nscatterings"].astype(np.float64)
    assert lam > 50
    assert abs(ns.mean() - lam) < 4 * np.sqrt(lam / n) + 3e-3 * lam, (ns.mean(), lam)      # law 2: mean
    assert abs(ns.var() / ns.mean() - 1.) < 6 * np.sqrt(2. / n) + 0.02, ns.var() / ns.mean()  # ... and Poisson variance
    ratio = (pk["e_cmf"] / e0).mean()                                                          # law 3: 1/t
    assert abs(ratio - ts.c.start / (ts.c.start + ts.c.width)) < 1.5e-3, ratio


def _centre_cell(nc):
    return (nc // 2) * (1 + nc + nc * nc)


def _check_thomson(backend, n):
    """r-packets of one frequency in the central cell (v << c: comoving and rest frame directions agree to 1e-3); those
    that scattered off an electron exactly once and met nothing else"""
    model, cs, ts, aux = synth.build("tiny", ncoord=7, width_frac=2e-4)  # ~0.2 scatterings per packet in this cell
    cell = _centre_cell(7)
    rng = np.random.default_rng(21)
    pk = _rpackets(model, aux, cell, n, 6e14, rng, margin=0.3)
    d0 = pk["dir"].copy()
    backend(model, cs, ts, pk)
    once = (pk["nscatterings"] == 1) & (pk["absorptiontype"] == -77) & (pk["type"] == abi.TYPE_RPKT) & (pk["emissiontype"] == abi.EMTYPE_NOTSET)
    m = int(once.sum())
    assert m > n // 50, m
    mu = (d0[once] * pk["dir"][once]).sum(axis=1)
    # var(mu) = 2/5, var(mu^2) = <mu^4> - (2/5)^2 = (2/5 + 2/7) / (8/3) - 4/25
    assert abs(mu.mean()) < 4 * np.sqrt(0.4 / m) + 2e-3, mu.mean()
    var_mu2 = (2 / 5 + 2 / 7) / (8 / 3) - 0.16
    assert abs((mu**2).mean() - 0.4) < 4 * np.sqrt(var_mu2 / m) + 2e-3, ((mu**2).mean(), m)
    assert abs((mu**2).mean() - 1 / 3) > 0.03  # ... and is told from isotropic scattering


def _kpkt_case(n, width_frac=2e-5, preset="small", ncoord=6):
    """k-packets, one very short timestep: each samples its cooling channel where it sits; what is emitted barely moves"""
    model, cs, ts, aux = synth.build(preset, ncoord=ncoord, width_frac=width_frac)
    pk = synth.make_packets(model, aux, n, kpkt_fraction=1.0, seed=7)
    return model, cs, ts, aux, pk


def _check_freefree_spectrum(backend, n):
    model, cs, ts, aux, pk = _kpkt_case(n)
    est = backend(model, cs, ts, pk)
    ff = (pk["type"] == abi.TYPE_RPKT) & (pk["emissiontype"] == abi.EMTYPE_FREEFREE) & (pk["nscatterings"] == 0)
    m = int(ff.sum())
    assert m > n // 200 and est.stats_dict()["K_STAT_TO_R_FF"] >= m, m
    c = model.d["propcell_nonemptymgi"][pk["cellindex"][ff]]
    x = 6.6260755e-27 * pk["nu_cmf"][ff] / (1.38064852e-16 * cs.d["Te"][c].astype(np.float64))
    assert abs(x.mean() - 1.) < 4 / np.sqrt(m) + 3e-3, (x.mean(), m)
    for cut in (1., 2.):
        pexp = np.exp(-cut)
        assert abs(np.mean(x > cut) - pexp) < 4 * np.sqrt(pexp * (1 - pexp) / m) + 3e-3, (cut, np.mean(x > cut))


def _check_cooling_channels(oracle, backend, n):
    """all k-packets in ONE cell; expected channel fractions from the cell's cooling list (cumulative per ion, kpkt.cc:57)"""
    model, cs, ts, aux = synth.build("small", ncoord=6, width_frac=2e-5)
    c = model["npts_nonempty"] // 2
    pk = synth.make_packets(model, aux, n, kpkt_fraction=1.0, seed=8, cells_only=[c])
    est = backend(model, cs, ts, pk)
    d = model.d
    contrib = oracle.cellcache(model, cs, ts, c)["cooling_contrib"]
    want = np.zeros(4)
    for ui in range(d["nions"]):
        o, nt = int(d["ion_coolingoffset"][ui]), int(d["ion_ncoolingterms"][ui])
        terms = np.diff(np.concatenate([[0.], contrib[o:o + nt]]))
        for k in range(4):
            want[k] += terms[np.asarray(d["coolinglist_type"][o:o + nt]) == k].sum()
    want /= want.sum()
    st = est.stats_dict()
    got = np.array([st["K_STAT_TO_R_FF"], st["K_STAT_TO_R_FB"], st["K_STAT_TO_MA_COLLEXC"], st["K_STAT_TO_MA_COLLION"]], dtype=np.float64)
    tot = got.sum()
    assert tot >= n
    for k in range(4):
        sig = np.sqrt(max(want[k] * (1 - want[k]), 1e-12) / tot)
        assert abs(got[k] / tot - want[k]) < 5 * sig + 2e-3, (k, got / tot, want)


def _check_freebound_spectrum(backend, n):
    """every free-bound photon of the run (k-packet cooling and macro-atom recombination, any continuum, any cell) through
    the probability integral transform of ITS continuum's law at ITS cell's temperature: the values are uniform on [0, 1].
    (The masses of the law over the reference's pieces are what is tested; inside a piece the reference draws uniformly.)"""
    model, cs, ts, aux, pk = _kpkt_case(n)
    backend(model, cs, ts, pk)
    d = model.d
    fb = (pk["type"] == abi.TYPE_RPKT) & (pk["emissiontype"] < 0) & (pk["emissiontype"] > abi.EMTYPE_NOTSET) & (pk["nscatterings"] == 0) & \
         (pk["emissiontype"] == pk["trueemissiontype"])
    m = int(fb.sum())
    assert m > n // 400, m
    bfl = np.asarray(d["level_bflist_start"])
    npts, inc = int(d["NPHIXSPOINTS"]), float(d["NPHIXSNUINCREMENT"])
    allxs = np.asarray(d["allphixs"], dtype=np.float64)
    cont_ul, cont_t = np.asarray(d["allcont_uniquelevelindex"]), np.asarray(d["allcont_phixstargetindex"])
    cells = d["propcell_nonemptymgi"][pk["cellindex"]]
    grid = np.linspace(0, inc * (npts - 1), 20001)
    u = []
    for code, c in set(zip(pk["emissiontype"][fb].tolist(), cells[fb].tolist())):
        sel = fb & (pk["emissiontype"] == code) & (cells == c)
        key = -1 - code   # emtype = -1 - level_bflist_start[ul] - t (atomic.h:508)
        ul = int(np.searchsorted(bfl, key, side="right") - 1)
        t = key - int(bfl[ul])
        i = int(np.nonzero((cont_ul == ul) & (cont_t == t))[0][0])
        nu_edge = float(d["allcont_nu_edge"][i])
        xs = allxs[int(d["level_phixsstart"][ul]) * npts:][:npts]
        nu = nu_edge * (1 + grid)
        idx = np.minimum((grid / inc).astype(int), npts - 1)   # photoionisation_crosssection_fromtable, classic: no interpolation
        w = xs[idx] * nu**3 * np.exp(-6.6260755e-27 * (nu - nu_edge) / (1.38064852e-16 * float(cs.d["Te"][c])))
        cdf = np.concatenate([[0.], np.cumsum(0.5 * (w[1:] + w[:-1]))])
        cdf /= cdf[-1]
        got = pk["nu_cmf"][sel]
        assert got.min() >= nu_edge * (1 - 1e-4) and got.max() <= nu[-1] * (1 + 1e-4), (code, got.min(), got.max(), nu_edge)
        # select_continuum_nu() inverts the law piece by piece: NPHIXSPOINTS pieces of equal width, each drawn with its exact
        # mass, the place inside a piece by linear interpolation (ratecoeff.cc:590-620): the law's CDF at the piece boundaries
        bounds = nu_edge * (1 + np.linspace(0, inc * (npts - 1), npts + 1))
        u.append(np.interp(got, bounds, np.interp(bounds, nu, cdf)))
    u = np.concatenate(u)
    assert abs(u.mean() - 0.5) < 4 * np.sqrt(1 / 12 / m) + 4e-3, (u.mean(), m)
    assert abs(u.var() - 1 / 12) < 4 * np.sqrt(1 / 180 / m) + 4e-3, u.var()
    for q in (0.25, 0.75):
        assert abs(np.mean(u < q) - q) < 4 * 0.5 / np.sqrt(m) + 6e-3, (q, np.mean(u < q))


def _check_compton(backend, n):
    """gamma packets of 0.8 MeV (below the pair-production threshold, far above the photoelectric range), a timestep short
    enough that few scatter twice; f = nu_cmf' / nu_cmf of the packets that are still gamma packets at another frequency"""
    model, cs, ts, aux = synth.build("tiny", ncoord=6, width_frac=4e-4)
    pk = synth.make_packets(model, aux, n, kpkt_fraction=0.0, gamma_fraction=1.0, seed=10)
    isg = pk["type"] == abi.TYPE_GAMMA
    nu0 = 0.8e6 * 1.6021772e-12 / 6.6260755e-27
    dop = pk["nu_cmf"] / pk["nu_rf"]
    pk["nu_cmf"] = np.where(isg, nu0, pk["nu_cmf"])
    pk["nu_rf"] = np.where(isg, nu0 / dop, pk["nu_rf"])
    backend(model, cs, ts, pk)
    x = 0.8 / 0.51099891
    fmin = 1 / (1 + 2 * x)
    fcut = 0.99  # (the comoving frequency of a packet in flight drifts by ~1e-3 in this timestep: scattered = below the cut)
    f_all = pk["nu_cmf"] / nu0
    sc = isg & (pk["type"] == abi.TYPE_GAMMA) & (f_all < fcut) & (f_all > fmin * (1 - 3e-3))   # (below fmin: scattered twice)
    m = int(sc.sum())
    assert m > n // 200, m
    assert np.count_nonzero(isg & (pk["type"] == abi.TYPE_GAMMA) & (f_all <= fmin * (1 - 3e-3))) < 0.05 * m
    f = f_all[sc]
    ff = np.linspace(fmin, fcut, 400001)
    cos_t = 1 - (1 / ff - 1) / x
    w = ff * (ff + 1 / ff - 1 + cos_t**2)          # survivors: f * dsigma/df (Klein-Nishina)
    cdf = np.cumsum(w) / w.sum()
    mean_want = (ff * w).sum() / w.sum()
    sd = np.sqrt(((ff - mean_want) ** 2 * w).sum() / w.sum())
    # (a few per cent of the packets scattered twice: their f is a product of two draws, lower than one)
    assert abs(f.mean() - mean_want) < 4 * sd / np.sqrt(m) + 0.012, (f.mean(), mean_want, m)
    for q in (0.25, 0.5, 0.75):
        cut = ff[np.searchsorted(cdf, q)]
        assert abs(np.mean(f < cut) - q) < 4 * 0.5 / np.sqrt(m) + 0.02, (q, np.mean(f < cut))


def _check_emission_aberration(backend, n):
    """k-packets in the outer cells (beta = 0.06 ... 0.1), one very short timestep; every r-packet they turned into (free-free,
    free-bound or a macro-atom line) that has not scattered since: its rest-frame direction and energy against v = r / t"""
    nc = 6
    model, cs, ts, aux = synth.build("small", ncoord=nc, width_frac=2e-5)
    d = model.d
    idx = np.arange(nc**3)
    ijk = np.stack([idx % nc, (idx // nc) % nc, idx // (nc * nc)], axis=1)
    radius = np.linalg.norm((ijk + 0.5) * (2 * d["rmax"] / nc) - d["rmax"], axis=1)
    mgi = np.asarray(d["propcell_nonemptymgi"])
    outer = np.unique(mgi[(mgi >= 0) & (radius > 0.7 * d["rmax"])])
    pk = synth.make_packets(model, aux, n, kpkt_fraction=1.0, seed=12, cells_only=outer)
    backend(model, cs, ts, pk)
    em = (pk["type"] == abi.TYPE_RPKT) & (pk["nscatterings"] == 0) & (pk["emissiontype"] != abi.EMTYPE_NOTSET)
    m = int(em.sum())
    assert m > n // 2, m
    beta = pk["pos"][em] / pk["prop_time"][em][:, None] / CLIGHT
    beta2 = (beta * beta).sum(axis=1)
    assert 3e-3 < beta2.mean() < 1e-2
    momentum = ((pk["dir"][em] * beta).sum(axis=1) * pk["e_rf"][em]).sum() / (beta2 * pk["e_cmf"][em]).sum()
    sigma = 1 / np.sqrt(3 * beta2.sum())  # e * beta * mu' of an isotropic mu' has the variance e^2 beta^2 / 3
    assert abs(momentum - 1.) < 4 * sigma + 3 * beta2.mean(), (momentum, sigma, m)  # (no aberration: 0; the wrong sign: -1)
    energy = pk["e_rf"][em].sum() / pk["e_cmf"][em].sum() - 1.
    assert abs(energy) < 4 * np.sqrt(beta2.mean() / 3 / m) + 2 * beta2.mean(), (energy, beta2.mean())


def _te_medium(T=11000., rho=3e-14, ncoord=4, width_frac=5e-4, lut_nsub=6, nphixspoints=None, phixsnuincrement=0.1):
    """every cell in strict thermodynamic equilibrium (synth.make_grid_and_cells uniform_te). A SHORT timestep: only a packet's
    last absorption and emission are recorded, and with dt/t = 2e-3 a tenth of the re-emitted packets meet another line before
    the step ends -- which ones depends on the emitting line, a bias of several per cent per pair (measured: chi2 246 over 118
    pairs); with 5e-4 the pairs are clean (53 over 70 at 8e6 packets)"""
    p = synth.PRESETS["small"]
    atomic = synth.make_atomic(seed=3, elements=p[0], nlevels_per_ion=p[1], line_fraction=0.6, nphixspoints=nphixspoints or p[3],
                               phixsnuincrement=phixsnuincrement, lut_nsub=lut_nsub)
    grid, cells, aux = synth.make_grid_and_cells(atomic, ncoord=ncoord, uniform_te=dict(T=T, rho=rho))
    md = {k: v for k, v in atomic.items() if not k.startswith("_")}
    md.update(grid)
    model, cs = abi.Model(md), abi.CellState(cells)
    ts = synth.make_timestep(aux["t"], width_frac=width_frac, vmax=grid["vmax"])
    return model, cs, ts, aux, T


def _planck_packets(model, aux, n, T, rng):
    """equal-energy r-packets of a Planck field: nu drawn from B_nu(T) (x^3 / (e^x - 1): the sum of four exponential
    variates over a geometric index), in the inner cells, isotropic"""
    nc = int(model.d["ncoordgrid"][0])
    inner = [ix + nc * (iy + nc * iz) for ix in (nc // 2 - 1, nc // 2) for iy in (nc // 2 - 1, nc // 2) for iz in (nc // 2 - 1, nc // 2)]
    # x^3/(e^x-1) = sum_k x^3 e^{-kx}: pick k with weight 1/k^4, then x = Gamma(4)/k
    k = np.arange(1, 60)
    w = 1.0 / k**4
    kk = rng.choice(k, size=n, p=w / w.sum())
    x = rng.gamma(4.0, 1.0, size=n) / kk
    nu = x * 1.380658e-16 * T / 6.6260755e-27
    parts = [_rpackets(model, aux, c, n // len(inner), 1.0, rng, margin=0.02) for c in inner]
    pk = np.concatenate(parts)
    nu = nu[:len(pk)]
    f = nu / pk["nu_cmf"]
    pk["nu_cmf"] *= f
    pk["nu_rf"] *= f
    pk["number"] = np.arange(len(pk))
    abi.seed_packet_rng(pk, 777)
    return pk


def _check_te_reciprocity(backend, n):
    model, cs, ts, aux, T = _te_medium()
    rng = np.random.default_rng(21)
    pk = _planck_packets(model, aux, n, T, rng)
    backend(model, cs, ts, pk)
    nlines = model["nlines"]
    nu_line = np.asarray(model.d["line_nu"])
    sel = (pk["type"] == abi.TYPE_RPKT) & (pk["absorptiontype"] >= 0) & (pk["absorptiontype"] < nlines) & (pk["emissiontype"] >= 0)
    a, e = pk["absorptiontype"][sel].astype(np.int64), pk["emissiontype"][sel].astype(np.int64)
    assert len(a) > n // 1000, len(a)
    N = np.zeros((nlines, nlines))
    np.add.at(N, (a, e), 1.0)
    off = N.copy()
    np.fill_diagonal(off, 0.)
    assert off.sum() > 0.2 * len(a)  # not resonance scattering alone: the walks do redistribute
    up = off[nu_line[None, :] > nu_line[:, None]].sum()    # re-emitted bluewards of the absorbing line ...
    down = off[nu_line[None, :] < nu_line[:, None]].sum()  # ... and redwards: equal in equilibrium
    assert abs(up - down) < 4.5 * np.sqrt(up + down) + 0.01 * (up + down), (up, down)
    S, D = off + off.T, off - off.T
    iu = np.triu_indices(nlines, 1)
    big = S[iu] >= 60
    z = D[iu][big] / np.sqrt(S[iu][big])
    K = int(big.sum())
    assert K >= 8, K
    chi2 = float((z * z).sum())
    assert chi2 < K + 5 * np.sqrt(2. * K) + 0.15 * K, (chi2, K)  # pair by pair: N(a -> e) = N(e -> a)
    assert np.abs(z).max() < 5.5, np.abs(z).max()
    # (the test has teeth: with T_R = 0.8 T_e the same packets give up : down = 8774 : 10093)
    return up, down, chi2, K


def _check_hubble_flow_energy(backend, n):
    model, cs, ts, aux = synth.build("small", ncoord=6, width_frac=0.05)
    pk = synth.make_packets(model, aux, n, kpkt_fraction=0.3, seed=31)
    before = (pk["e_cmf"] * pk["prop_time"]).sum()
    backend(model, cs, ts, pk)
    r = (pk["type"] == abi.TYPE_RPKT) | (pk["type"] == abi.TYPE_ESCAPE)
    assert r.sum() > n // 2
    photons_cmf, photons_rf = pk["e_cmf"][r] / pk["nu_cmf"][r], pk["e_rf"][r] / pk["nu_rf"][r]
    assert np.abs(photons_cmf / photons_rf - 1.).max() < 1e-12, np.abs(photons_cmf / photons_rf - 1.).max()
    # an escaped packet keeps the time and energy of its escape; everything else has been advanced to the end of the step
    after = (pk["e_cmf"] * np.where(pk["type"] == abi.TYPE_ESCAPE, pk["escape_time"], pk["prop_time"])).sum()
    beta_max = model.d["vmax"] / CLIGHT
    assert abs(after / before - 1.) < beta_max * 0.05 + 2e-3, after / before
    assert abs(after / before - 1.) > 0. or n < 10


def _check_te_channel_balance(backend, n, nphixspoints=None, phixsnuincrement=0.1, only_thermal_pool=False):
    """law 12: the event counters of a run in the equilibrium medium, thick enough for plenty of events of every kind (the counters see them
    all: a photon emitted just above the dominant ion's edge is absorbed again at once, and counted again on both sides)"""
    model, cs, ts, aux, T = _te_medium(T=25000., rho=1e-11, width_frac=1e-4, nphixspoints=nphixspoints, phixsnuincrement=phixsnuincrement)
    rng = np.random.default_rng(33)
    pk = _planck_packets(model, aux, n, T, rng)
    est = backend(model, cs, ts, pk)
    st = est.stats_dict()
    if only_thermal_pool:
        return st["K_STAT_FROM_BF"], st["K_STAT_TO_R_FB"]
    pending = int(np.count_nonzero(pk["type"] != abi.TYPE_RPKT))  # absorbed, not emitted yet when the step ended
    pairs = {
        "bound-bound": (st["MA_STAT_ACTIVATION_BB"], st["MA_STAT_DEACTIVATION_BB"] + st["K_STAT_TO_R_BB"]),
        "bound-free": (st["MA_STAT_ACTIVATION_BF"] + st["K_STAT_FROM_BF"], st["MA_STAT_DEACTIVATION_FB"] + st["K_STAT_TO_R_FB"]),
        "bound-free, macro-atom": (st["MA_STAT_ACTIVATION_BF"], st["MA_STAT_DEACTIVATION_FB"]),
        "bound-free, thermal pool": (st["K_STAT_FROM_BF"], st["K_STAT_TO_R_FB"]),
        "free-free": (st["K_STAT_FROM_FF"], st["K_STAT_TO_R_FF"]),
        "collisional (bound-bound)": (st["K_STAT_TO_MA_COLLEXC"], st["MA_STAT_DEACTIVATION_COLLDEEXC"]),
        "collisional (bound-free)": (st["K_STAT_TO_MA_COLLION"], st["MA_STAT_DEACTIVATION_COLLRECOMB"]),
        "ionisation inside the macro-atom": (st["MA_STAT_INTERNALUPHIGHER"], st["MA_STAT_INTERNALDOWNLOWER"]),
    }
    assert st["K_STAT_TO_MA_COLLEXC"] == st["MA_STAT_ACTIVATION_COLLEXC"] and st["K_STAT_TO_MA_COLLION"] == st["MA_STAT_ACTIVATION_COLLION"]
    total_in = st["MA_STAT_ACTIVATION_BB"] + st["MA_STAT_ACTIVATION_BF"] + st["K_STAT_FROM_BF"] + st["K_STAT_FROM_FF"]
    total_out = st["MA_STAT_DEACTIVATION_BB"] + st["MA_STAT_DEACTIVATION_FB"] + st["K_STAT_TO_R_FB"] + st["K_STAT_TO_R_FF"] + st["K_STAT_TO_R_BB"]
    assert total_in - total_out == pending, (total_in, total_out, pending)  # bookkeeping: what went in and is not out is still inside
    assert total_in > n // 20
    worst = 0.
    for name, (fwd, bwd) in pairs.items():
        # the difference of the two counts is a sum of +1 / -1 over the events that change channel: its variance is at most their sum
        sigma = np.sqrt(max(fwd + bwd, 1))
        z = (fwd - bwd) / sigma
        if name == "bound-free, thermal pool":
            # FINDING (round 6, engine and host emulation alike, i.e. the algorithm as restated): bound-free absorption makes 5-8 % more k-packets
            # than free-bound cooling removes (8e6 packets on the GPU: 39 445 against 36 789, 9.6 sigma; 1.6e7 on the CPU: 78 948 against 72 960),
            # made up by ~1 % more recombination emissions than photoionisations in the macro-atom half -- the two halves together balance.
            # What it is NOT: the RATES -- the continuous expectation of the heating, integrated over the cell's own populations and cross-
            # sections, is 0.996 of the cooling list's free-bound total; not the tables' quadrature (the same with lut_nsub = 200) nor the nu^-3
            # tail beyond the tables (same with tables ten times as long).
            # What it IS (found at the round's end): the reference draws the frequency of a free-bound photon UNIFORMLY within each of the
            # NPHIXSPOINTS pieces of the continuum (select_continuum_nu, ratecoeff.cc:563-637: the tail integrals at the pieces' boundaries,
            # interpolated linearly in between), and a piece is 0.1 nu_edge wide -- across which the emissivity sigma nu^3 exp(-h nu / kT) of a
            # thick edge (h nu_edge = 14 kT here) falls by a factor of four. The photon is therefore emitted too blue on average, and when it is
            # absorbed again (at once: optical depth ~1e3) its thermal share 1 - nu_edge / nu is 15 % larger than the share of the thermal
            # pool in the emission (0.072 against 0.062 for the two thickest continua, which carry 10 % of the free-bound cooling); the excess
            # heat leaves through collisional excitation and ionisation and comes back as recombination photons of the macro-atom half, and the
            # path-length estimators see 15 % more radiation above those edges than Planck's (the photons sit where the opacity is lower).
            # The imbalance is quadratic in the pieces' width: NPHIXSNUINCREMENT 0.1 / 0.05 / 0.025 / 0.0125 with as many more points (the same
            # tables, 1.6e7 packets each) give 1.082 / 1.033 / 1.020 / 1.021 +- 0.005 -- test_thermal_pool_imbalance_is_the_free_bound_sampling
            # below holds that. A floor of ~2 % (3.7 sigma), the macro-atom half's 0.7 % and 1.5 % more free-free absorptions than emissions do not
            # depend on the width. They need the bound-free processes (with the cross-sections scaled to nothing every pair balances: bound-bound 0.9,
            # free-free 0.6, collisional 0.3 sigma at 1.6e7 packets) and come partly from the SYNTHETIC tables, which are not exactly in equilibrium
            # with their own cross-sections: synth.make_atomic integrates its rate-coefficient tables over a cross-section that is constant between
            # two table points while the opacity interpolates it linearly (lut_linear_sigma=True: free-free +1.2 % -> +0.4 %, macro-atom half
            # -0.54 % -> -0.39 %), and the photoionisation table, steep in T for a thick edge, is interpolated linearly in T like the reference's
            # (T on a grid point: -0.66 % -> -0.54 %). What is left after both (2-3 sigma per pair) was not followed further.
            # The pair is therefore held to 12 % here, not to its noise.
            assert abs(fwd - bwd) <= 0.12 * max(fwd, bwd) + 4.5 * sigma, (name, fwd, bwd)
            continue
        worst = max(worst, abs(z))
        assert abs(fwd - bwd) <= 4.5 * sigma + pending, (name, fwd, bwd)
    for name in ("bound-free", "free-free", "collisional (bound-bound)", "ionisation inside the macro-atom"):
        assert min(pairs[name]) > n // 400, (name, pairs[name])  # the channels the law is about do occur
    return pairs, worst


def _check_gamma_deposition(backend, n):
    """law 13: gamma packets born all over the synthetic ejecta at t = 20 d, one timestep of 5 %"""
    model, cs, ts, aux = synth.build("small", ncoord=8, width_frac=0.05)
    pk = synth.make_packets(model, aux, n, kpkt_fraction=0.0, gamma_fraction=1.0, seed=41)
    assert np.all(pk["type"] == abi.TYPE_GAMMA)
    e0_rf, e0_cmf = pk["e_rf"].copy(), pk["e_cmf"].copy()
    est = backend(model, cs, ts, pk)
    ty = pk["type"]
    esc = (ty == abi.TYPE_ESCAPE) & (pk["escape_type"] == abi.TYPE_GAMMA)  # left the grid as a gamma ray
    fly = ty == abi.TYPE_GAMMA
    # thermalised: a non-thermal lepton's deposit, by now a k-packet, an r-packet (perhaps escaped as one) or an active macro-atom
    dep = ~esc & ~fly
    assert esc.sum() > n // 50 and dep.sum() > n // 10, (esc.sum(), dep.sum(), fly.sum())
    assert est.stats_dict()["NT_STAT_FROM_GAMMA"] == dep.sum()
    # (a) the two estimates of the deposited energy. A deposited packet carries on as a k-packet / r-packet, whose e_cmf follows e_cmf * t =
    # const from the deposit to the end of the step (law 11): its energy AT the deposit lies between e_cmf now and e_cmf * t_end / t_start
    analog_lo = pk["e_cmf"][dep].sum()
    analog_hi = analog_lo * (1. + 0.05)
    est_dep = est.dep_estimator_gamma.sum()
    noise = 4.5 * e0_cmf.mean() * np.sqrt(dep.sum())  # the analog count's Poisson noise (the estimator's is much smaller)
    assert analog_lo - noise <= est_dep <= analog_hi + noise, (analog_lo, est_dep, analog_hi, noise)
    assert abs(est_dep / (0.5 * (analog_lo + analog_hi)) - 1.) < 0.06
    # (b) energy bookkeeping to first order in v/c
    total = pk["e_rf"][esc].sum() + pk["e_cmf"][dep].sum() + pk["e_rf"][fly].sum()
    beta = model.d["vmax"] / CLIGHT
    assert abs(total / e0_rf.sum() - 1.) < 2. * beta + 0.05, total / e0_rf.sum()
    return est_dep / (0.5 * (analog_lo + analog_hi)), int(dep.sum()), int(esc.sum())


def test_detailed_balance_channel_by_channel_kernel_bodies():
    _check_te_channel_balance(_backend_emu, 400_000)


def test_thermal_pool_imbalance_is_the_free_bound_sampling():
    """the open finding of law 12, explained (see _check_te_channel_balance): with the photoionisation tables four times as fine (the same cross-
    sections, the same range) the excess of bound-free heating events over free-bound cooling events falls from ~6 % to ~1.5 % (4e6 packets:
    1.064 and 1.014, +- 0.011 each) -- it is the uniform draw within a piece of select_continuum_nu(), not the kernels and not the rates"""
    coarse = _check_te_channel_balance(_backend_emu, 4_000_000, only_thermal_pool=True)
    fine = _check_te_channel_balance(_backend_emu, 4_000_000, nphixspoints=157, phixsnuincrement=0.025, only_thermal_pool=True)
    r_coarse, r_fine = coarse[0] / coarse[1], fine[0] / fine[1]
    assert min(coarse + fine) > 10_000, (coarse, fine)
    assert r_coarse > 1.035 and r_fine < 1.035 and r_coarse - r_fine > 0.025, (coarse, fine)


def test_gamma_deposition_estimator_and_analog_kernel_bodies():
    _check_gamma_deposition(_backend_emu, 200_000)


@pytest.mark.gpu
def test_detailed_balance_channel_by_channel_engine():
    pairs, worst = _check_te_channel_balance(_backend_gpu, 8_000_000)
    print("TE channel balance:", {k: v for k, v in pairs.items()}, f"worst {worst:.2f} sigma")


@pytest.mark.gpu
def test_gamma_deposition_estimator_and_analog_engine():
    ratio, ndep, nesc = _check_gamma_deposition(_backend_gpu, 2_000_000)
    print(f"gamma deposition: estimator / analog = {ratio:.4f} ({ndep} deposited, {nesc} escaped)")


def test_macroatom_reciprocity_in_equilibrium_kernel_bodies():
    _check_te_reciprocity(_backend_emu, 1_000_000)


def test_photon_number_and_hubble_flow_energy_kernel_bodies():
    _check_hubble_flow_energy(_backend_emu, 20000)


@pytest.mark.gpu
def test_macroatom_reciprocity_in_equilibrium_engine():
    up, down, chi2, K = _check_te_reciprocity(_backend_gpu, 16_000_000)
    print(f"TE reciprocity: {up:.0f} up, {down:.0f} down, chi2 {chi2:.1f} over {K} pairs")


@pytest.mark.gpu
def test_photon_number_and_hubble_flow_energy_engine():
    _check_hubble_flow_energy(_backend_gpu, 400000)


def test_thomson_phase_function_kernel_bodies():
    _check_thomson(_backend_emu, 120000)


def test_freefree_emission_spectrum_kernel_bodies():
    _check_freefree_spectrum(_backend_emu, 20000)


def test_cooling_channel_fractions_kernel_bodies(oracle):
    _check_cooling_channels(oracle, _backend_emu, 10000)


def test_freebound_emission_spectrum_kernel_bodies():
    _check_freebound_spectrum(_backend_emu, 20000)


def test_compton_klein_nishina_kernel_bodies():
    _check_compton(_backend_emu, 60000)


def test_emission_momentum_and_energy_in_the_rest_frame_kernel_bodies():
    _check_emission_aberration(_backend_emu, 200000)


@pytest.mark.gpu
def test_emission_momentum_and_energy_in_the_rest_frame_engine():
    _check_emission_aberration(_backend_gpu, 3000000)


@pytest.mark.gpu
def test_thomson_phase_function_engine():
    _check_thomson(_backend_gpu, 2000000)


@pytest.mark.gpu
def test_freefree_emission_spectrum_engine():
    _check_freefree_spectrum(_backend_gpu, 600000)


@pytest.mark.gpu
def test_cooling_channel_fractions_engine(oracle):
    _check_cooling_channels(oracle, _backend_gpu, 400000)


@pytest.mark.gpu
def test_freebound_emission_spectrum_engine():
    _check_freebound_spectrum(_backend_gpu, 600000)


@pytest.mark.gpu
def test_compton_klein_nishina_engine():
    _check_compton(_backend_gpu, 2000000)


def test_sobolev_escape_probability_kernel_bodies(oracle):
    _check_sobolev(oracle, _backend_emu, 40000)


def test_grey_scattering_and_adiabatic_losses_kernel_bodies():
    _check_grey(_backend_emu, 4000)


@pytest.mark.gpu
def test_sobolev_escape_probability_engine(oracle):
    _check_sobolev(oracle, _backend_gpu, 400000)


@pytest.mark.gpu
def test_grey_scattering_and_adiabatic_losses_engine():
    _check_grey(_backend_gpu, 100000)
