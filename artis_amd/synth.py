"""Synthetic atomic data and W7-like ejecta for tests and bench.py.

There is no network in the build or GPU environment, so the reference's atomic
data release (tests/setup_classicmode_3d.sh downloads atomicdata_classic.tar.xz)
is not available. This module builds a self-consistent stand-in with the same
TABLE STRUCTURE the reference's input.cc produces (line list sorted by falling
frequency, alltrans blocks per level, continua sorted by rising edge frequency,
ground-continuum estimator indices, cooling list, temperature LUTs) and a
W7-like exponential-density ejecta mapped on a Cartesian (or 1D spherical) grid.

It is host-side set-up code (the reference does this in input.cc / grid.cc /
ratecoeff.cc / update_grid.cc, all outside the packet path). Physical fidelity is
"plausible", not a reproduction of real atomic data; structural rules are cited.
"""
from __future__ import annotations

import numpy as np

from . import abi

CLIGHT = 2.99792458e10
H = 6.6260755e-27
KB = 1.38064852e-16
EV = 1.6021772e-12
ME = 9.1093897e-28
QE = 4.80325e-10
MH = 1.67352e-24
MSUN = 1.98855e33
DAY = 86400.0
SAHACONST = 2.0706659e-16
CLIGHTSQUAREDOVERTWOH = CLIGHT**2 / (2 * H)
TABLESIZE = 100
MINTEMP = 3500.0
MAXTEMP = 140000.0
MINPOP = 1e-30

# ionisation potentials [eV] of stages I.. (NIST values, rounded); fallback formula otherwise
_IONPOT = {
    8: [13.62, 35.12, 54.94, 77.41, 113.9],
    14: [8.15, 16.35, 33.49, 45.14, 166.8],
    16: [10.36, 23.34, 34.79, 47.22, 72.59],
    20: [6.11, 11.87, 50.91, 67.27, 84.5],
    26: [7.90, 16.20, 30.65, 54.91, 75.0],
    27: [7.88, 17.08, 33.50, 51.27, 79.5],
    28: [7.64, 18.17, 35.19, 54.92, 76.06],
}
_AMASS = {8: 16.0, 14: 28.09, 16: 32.06, 20: 40.08, 26: 55.85, 27: 58.93, 28: 58.69}


def _ionpot_ev(Z: int, stage: int) -> float:
    tab = _IONPOT.get(Z)
    if tab is not None and stage - 1 < len(tab):
        return tab[stage - 1]
    return 7.5 * stage**1.6


# rate-coefficient table grids of the options presets (include/artis_options.h: TABLESIZE, MINTEMP, MAXTEMP)
OPTION_TABLES = {"classic": (100, 3500.0, 140000.0), "kilonova_lte": (200, 500.0, 150000.0),
                 "nltenebular": (100, 1000.0, 30000.0), "christinenonthermal": (100, 3000.0, 140000.0),
                 "nltephotospheric": (100, 3500.0, 140000.0), "nltewithoutnonthermal": (200, 4000.0, 140000.0),
                 "nltenebular_lineest": (100, 1000.0, 30000.0),
                 "kilonova_barnes": (200, 500.0, 150000.0),
                 "kilonova_wollaeger": (200, 500.0, 150000.0), "kilonova_expopac": (200, 500.0, 150000.0),
                 "kilonova_gammaproducts": (200, 500.0, 150000.0), "kilonova_gamma_barnes": (200, 500.0, 150000.0),
                 "kilonova_gamma_wollaeger": (200, 500.0, 150000.0), "kilonova_gamma_guttman": (200, 500.0, 150000.0),
                 "kilonova_gamma_grey": (200, 500.0, 150000.0), "classic_gamma_xcom": (100, 3500.0, 140000.0),
                 "classic_expopac_therm": (100, 3500.0, 140000.0)}


def make_atomic(seed: int = 1, elements=None, nlevels_per_ion: int = 12, line_fraction: float = 0.4,
                nphixspoints: int = 40, phixsnuincrement: float = 0.1, lut_linear_sigma: bool = False, two_target_fraction: float = 0.25,
                forbidden_fraction: float = 0.2, collstr_fraction: float = 0.1, options: str = "classic", lut_nsub: int = 6) -> dict:
    """Build the atomic part of struct artis_model. `elements` = list of (Z, lowest_ionstage, nions).
    lut_nsub: sub-intervals per cross-section table cell of the quadrature behind the rate-coefficient tables (6: 1e-3 accurate, what every
    model of the tests and the bench is made with; the equilibrium media of tests/test_physics_laws.py ask for 200: in an optically thick
    continuum a table that is 0.15 % off its own cross-sections is amplified by the hundreds of absorption / re-emission cycles of a photon)."""
    rng = np.random.default_rng(seed)
    if elements is None:
        elements = [(14, 1, 4), (26, 1, 5), (27, 1, 5)]
    nelements = len(elements)

    elem_nions, elem_uiis, elem_anumber, elem_lowest = [], [], [], []
    ion_element, ion_nlevels, ion_uls = [], [], []
    level_eps, level_g, level_ion = [], [], []
    ion_ionpot = []
    nions_total = 0
    for e, (Z, lowest, nions) in enumerate(elements):
        elem_nions.append(nions)
        elem_uiis.append(nions_total)
        elem_anumber.append(Z)
        elem_lowest.append(lowest)
        eps_ground = 0.0
        for ion in range(nions):
            stage = lowest + ion
            ionpot = _ionpot_ev(Z, stage) * EV
            top = ion == nions - 1
            nl = 1 if top else nlevels_per_ion  # SINGLE_LEVEL_TOP_ION (artisoptions_classic.h:28)
            frac = np.sort(rng.random(nl - 1)) if nl > 1 else np.zeros(0)
            E = np.concatenate([[0.0], 0.9 * ionpot * (0.04 + 0.96 * frac**0.8)])
            E = np.sort(E)
            ion_element.append(e)
            ion_nlevels.append(nl)
            ion_uls.append(len(level_eps))
            ion_ionpot.append(ionpot)
            for k in range(nl):
                level_eps.append(eps_ground + E[k])
                level_g.append(float(rng.integers(1, 11)))
                level_ion.append(nions_total)
            eps_ground += ionpot
            nions_total += 1
    nlevels = len(level_eps)
    level_eps = np.array(level_eps)
    level_g = np.array(level_g, dtype=np.float32)
    ion_uls = np.array(ion_uls, dtype=np.int32)
    ion_nlevels = np.array(ion_nlevels, dtype=np.int32)
    ion_element = np.array(ion_element, dtype=np.int32)

    # ---- bound-bound transitions
    lines = []  # (nu, ion, lower(level in ion), upper, A, collstr, forbidden)
    for ui in range(nions_total):
        nl = ion_nlevels[ui]
        s = ion_uls[ui]
        for lo in range(nl):
            for up in range(lo + 1, nl):
                if rng.random() >= line_fraction:
                    continue
                dE = level_eps[s + up] - level_eps[s + lo]
                nu = dE / H
                if nu < 2e13:
                    continue
                forb = rng.random() < forbidden_fraction
                A = 10 ** rng.uniform(-2, 2) if forb else 10 ** rng.uniform(5.0, 8.5) * min(1.0, (dE / (5 * EV)) ** 2)
                cs = rng.uniform(0.1, 5.0) if rng.random() < collstr_fraction else -1.0
                lines.append((nu, ui, lo, up, A, cs, forb))
    # the line list is sorted by falling frequency (input.cc: linelist sort; rpkt.h:179)
    lines.sort(key=lambda t: -t[0])
    nlines = len(lines)
    line_nu = np.array([t[0] for t in lines])
    # recompute nu exactly as the path does (macroatom.cc:229: nu = epsilon_trans / H)
    line_ui = np.array([t[1] for t in lines], dtype=np.int32)
    line_lo = np.array([t[2] for t in lines], dtype=np.int32)
    line_up = np.array([t[3] for t in lines], dtype=np.int32)
    line_A = np.array([t[4] for t in lines], dtype=np.float32)
    line_cs = np.array([t[5] for t in lines], dtype=np.float32)
    line_forb = np.array([t[6] for t in lines], dtype=np.uint8)
    line_ul_lower = ion_uls[line_ui] + line_lo
    line_ul_upper = ion_uls[line_ui] + line_up
    line_nu = (level_eps[line_ul_upper] - level_eps[line_ul_lower]) / H
    order = np.argsort(-line_nu, kind="stable")
    (line_nu, line_ui, line_lo, line_up, line_A, line_cs, line_forb, line_ul_lower, line_ul_upper) = (
        a[order] for a in (line_nu, line_ui, line_lo, line_up, line_A, line_cs, line_forb, line_ul_lower, line_ul_upper))
    g_u = level_g[line_ul_upper].astype(np.float64)
    g_l = level_g[line_ul_lower].astype(np.float64)
    B_ul = (CLIGHTSQUAREDOVERTWOH / line_nu**3 * line_A.astype(np.float64)).astype(np.float32)
    B_lu = (g_u / g_l * B_ul.astype(np.float64)).astype(np.float32)
    f_lu = (g_u / g_l * line_A.astype(np.float64) * ME * CLIGHT**3 / (8 * np.pi**2 * QE**2 * line_nu**2)).astype(np.float32)
    line_elementindex = ion_element[line_ui]
    line_ionindex = (line_ui - np.array(elem_uiis, dtype=np.int32)[line_elementindex]).astype(np.int32)

    # alltrans: per level its down transitions, then its up transitions (atomic.h:462-471)
    downs = [[] for _ in range(nlevels)]
    ups = [[] for _ in range(nlevels)]
    for li in range(nlines):
        downs[line_ul_upper[li]].append((int(line_lo[li]), li))
        ups[line_ul_lower[li]].append((int(line_up[li]), li))
    at_line, at_target, at_A, at_cs, at_f, at_forb = [], [], [], [], [], []
    level_startdown = np.zeros(nlevels, dtype=np.int32)
    level_ndown = np.zeros(nlevels, dtype=np.int32)
    level_nup = np.zeros(nlevels, dtype=np.int32)
    level_mablock = np.zeros(nlevels, dtype=np.int32)
    mab = 0
    for ul in range(nlevels):
        level_startdown[ul] = len(at_line)
        for lst in (sorted(downs[ul]), sorted(ups[ul])):
            for target, li in lst:
                at_line.append(li)
                at_target.append(target)
                at_A.append(line_A[li])
                at_cs.append(line_cs[li])
                at_f.append(f_lu[li])
                at_forb.append(line_forb[li])
        level_ndown[ul] = len(downs[ul])
        level_nup[ul] = len(ups[ul])
        level_mablock[ul] = mab  # input.cc:1542
        mab += 2 * level_ndown[ul] + level_nup[ul]

    # ---- photoionisation: every level of a non-top ion has a table and 1 or 2 targets
    NP = nphixspoints
    inc = phixsnuincrement
    ion_nlevels_ionising = np.zeros(nions_total, dtype=np.int32)
    ion_maxrecomb = np.full(nions_total, -1, dtype=np.int32)
    level_phixsstart = np.full(nlevels, -1, dtype=np.int32)
    level_nphixstargets = np.zeros(nlevels, dtype=np.int32)
    level_phixstargetstart = np.full(nlevels, -1, dtype=np.int32)
    level_bflist_start = np.full(nlevels, -1, dtype=np.int32)
    allphixs = []
    pt_level, pt_prob = [], []
    nphixslevels = 0
    bfl = 0
    for e, (Z, lowest, nions) in enumerate(elements):
        for ion in range(nions - 1):
            ui = elem_uiis[e] + ion
            nl = ion_nlevels[ui]
            ion_nlevels_ionising[ui] = nl
            nl_upper = ion_nlevels[ui + 1]
            for lev in range(nl):
                ul = ion_uls[ui] + lev
                sigma0 = 10 ** rng.uniform(-18.5, -17.3) / (1 + 0.3 * lev)
                xs = sigma0 * (1.0 + inc * np.arange(NP)) ** -3.0
                level_phixsstart[ul] = nphixslevels
                allphixs.append(xs.astype(np.float32))
                nphixslevels += 1
                level_phixstargetstart[ul] = len(pt_level)
                level_bflist_start[ul] = bfl
                if nl_upper >= 2 and rng.random() < two_target_fraction:
                    gu0 = float(level_g[ion_uls[ui + 1]])
                    gu1 = float(level_g[ion_uls[ui + 1] + 1])
                    p0 = gu0 / (gu0 + gu1)
                    pt_level += [0, 1]
                    pt_prob += [p0, 1.0 - p0]
                    level_nphixstargets[ul] = 2
                    ion_maxrecomb[ui + 1] = max(ion_maxrecomb[ui + 1], 1)
                    bfl += 2
                else:
                    pt_level += [0]
                    pt_prob += [1.0]
                    level_nphixstargets[ul] = 1
                    ion_maxrecomb[ui + 1] = max(ion_maxrecomb[ui + 1], 0)
                    bfl += 1
    nbfcontinua = bfl
    allphixs = np.concatenate(allphixs) if allphixs else np.zeros(0, dtype=np.float32)
    pt_level = np.array(pt_level, dtype=np.int32)
    pt_prob = np.array(pt_prob, dtype=np.float64)

    def threshold(e, ion, lev, t):
        ul = ion_uls[elem_uiis[e] + ion] + lev
        upper = pt_level[level_phixstargetstart[ul] + t]
        return level_eps[ion_uls[elem_uiis[e] + ion + 1] + upper] - level_eps[ul]

    # ground continua, sorted by rising edge (input.cc:785-806)
    ground = []
    for e, (Z, lowest, nions) in enumerate(elements):
        for ion in range(nions - 1):
            ground.append((threshold(e, ion, 0, 0) / H, e, ion))
    ground.sort(key=lambda t: t[0])
    groundcont_nu_edge = np.array([g[0] for g in ground])
    nbfg = len(ground)

    def search_groundphixslist(nu_edge, e_in, ion_in, lev_in):  # input.cc:703
        if nbfg <= 0 or nu_edge < groundcont_nu_edge[0]:
            return -1
        i = 1
        while i < nbfg and not (nu_edge < groundcont_nu_edge[i]):
            i += 1
        if i == nbfg:
            return i - 1
        left = nu_edge - groundcont_nu_edge[i - 1]
        right = groundcont_nu_edge[i] - nu_edge
        return i - 1 if left <= right else i

    level_closest = np.full(nlevels, -1, dtype=np.int32)
    conts = []
    for e, (Z, lowest, nions) in enumerate(elements):
        for ion in range(nions - 1):
            ui = elem_uiis[e] + ion
            for lev in range(ion_nlevels_ionising[ui]):
                ul = ion_uls[ui] + lev
                level_closest[ul] = search_groundphixslist(threshold(e, ion, lev, 0) / H, e, ion, lev)
                for t in range(level_nphixstargets[ul]):
                    gci = level_closest[ul] if (lev == 0 and t == 0) else -1
                    conts.append((threshold(e, ion, lev, t) / H, e, ion, lev, t, int(pt_level[level_phixstargetstart[ul] + t]),
                                  ul, float(pt_prob[level_phixstargetstart[ul] + t]), gci))
    conts.sort(key=lambda t: t[0])  # stable sort by nu_edge (input.cc:893)
    assert len(conts) == nbfcontinua

    # ---- cooling list (kpkt.cc:233 set_ncoolingterms, kpkt.cc:311 setup_coolinglist)
    ion_coolingoffset = np.zeros(nions_total, dtype=np.int32)
    ion_ncoolingterms = np.zeros(nions_total, dtype=np.int32)
    cl_type, cl_level, cl_t = [], [], []
    for e, (Z, lowest, nions) in enumerate(elements):
        for ion in range(nions):
            ui = elem_uiis[e] + ion
            ion_coolingoffset[ui] = len(cl_type)
            if lowest + ion - 1 > 0:
                cl_type.append(0); cl_level.append(-99); cl_t.append(-99)
            for lev in range(ion_nlevels[ui]):
                if level_nup[ion_uls[ui] + lev] > 0:
                    cl_type.append(2); cl_level.append(lev); cl_t.append(-1)
            if ion < nions - 1:
                for ctype in (3, 1):  # COLLION block, then FREEBOUND block
                    for lev in range(ion_nlevels_ionising[ui]):
                        for t in range(level_nphixstargets[ion_uls[ui] + lev]):
                            cl_type.append(ctype); cl_level.append(lev); cl_t.append(t)
            ion_ncoolingterms[ui] = len(cl_type) - ion_coolingoffset[ui]

    # ---- temperature LUTs (ratecoeff.cc:143 precalculate_rate_coefficient_integrals), simple quadrature
    TABLESIZE, MINTEMP, MAXTEMP = abi.CI_PRESETS[options][2] if options in abi.CI_PRESETS else OPTION_TABLES[options]
    T_step_log = (np.log(MAXTEMP) - np.log(MINTEMP)) / (TABLESIZE - 1.0)
    Tgrid = (MINTEMP * np.exp(np.arange(TABLESIZE) * T_step_log)).astype(np.float32).astype(np.float64)
    spont = np.zeros((nbfcontinua, TABLESIZE))
    gammac = np.zeros((nbfcontinua, TABLESIZE))
    bfcool = np.zeros((nbfcontinua, TABLESIZE))
    nsub = int(lut_nsub)
    for e, (Z, lowest, nions) in enumerate(elements):
        for ion in range(nions - 1):
            ui = elem_uiis[e] + ion
            for lev in range(ion_nlevels_ionising[ui]):
                ul = ion_uls[ui] + lev
                xs = allphixs[level_phixsstart[ul] * NP:(level_phixsstart[ul] + 1) * NP].astype(np.float64)
                for t in range(level_nphixstargets[ul]):
                    ci = level_bflist_start[ul] + t
                    nu_thr = threshold(e, ion, lev, t) / H
                    prob = pt_prob[level_phixstargetstart[ul] + t]
                    g_low = float(level_g[ul])
                    g_up = float(level_g[ion_uls[ui + 1] + pt_level[level_phixstargetstart[ul] + t]])
                    # sample the NP-1 table cells between nu_thr and nu_thr*last_phixs_nuovernuedge
                    k = np.arange((NP - 1) * nsub)
                    x = (k + 0.5) / nsub * inc * nu_thr  # nu - nu_edge
                    dx = inc * nu_thr / nsub
                    sig = xs[np.minimum((k // nsub), NP - 1)]
                    if lut_linear_sigma:  # the cross-section between two table points as the opacity interpolates it (rpkt.cc: linear in nu)
                        cell = np.minimum(k // nsub, NP - 2)
                        sig = xs[cell] + (xs[cell + 1] - xs[cell]) * (((k % nsub) + 0.5) / nsub)
                    nu = nu_thr + x
                    ex = np.exp(-H * x[None, :] / (KB * Tgrid[:, None]))
                    saha = SAHACONST * g_low / g_up * Tgrid**-1.5
                    spont[ci] = 4 * np.pi * saha * prob * np.sum((2 / CLIGHT**2) * sig * nu**2 * ex, axis=1) * dx
                    bfcool[ci] = 4 * np.pi * saha * prob * np.sum(sig * x * (2 * H / CLIGHT**2) * nu**2 * ex, axis=1) * dx
                    hnkt = H * nu[None, :] / (KB * Tgrid[:, None])
                    with np.errstate(over="ignore"):
                        planck = 2 * H * nu**3 / CLIGHT**2 / np.expm1(hnkt)
                    gammac[ci] = 4 * np.pi * prob * np.sum(sig / (H * nu) * planck * (1 - np.exp(-hnkt)), axis=1) * dx

    d = dict(
        nelements=nelements, nions=nions_total, nlevels=nlevels, nlines=nlines, nalltrans=len(at_line),
        nphixstargets_total=len(pt_level), nphixslevels=nphixslevels, nbfcontinua=nbfcontinua, nbfcontinua_ground=nbfg,
        ncoolingterms=len(cl_type), nmatransblock=int(mab), NPHIXSPOINTS=NP, NPHIXSNUINCREMENT=float(inc),
        elem_nions=elem_nions, elem_uniqueionindexstart=elem_uiis, elem_anumber=elem_anumber,
        elem_lowest_ionstage=elem_lowest,
        ion_element=ion_element, ion_nlevels=ion_nlevels, ion_nlevels_ionising=ion_nlevels_ionising,
        ion_maxrecombininglevel=ion_maxrecomb, ion_uniquelevelindexstart=ion_uls,
        ion_coolingoffset=ion_coolingoffset, ion_ncoolingterms=ion_ncoolingterms,
        level_epsilon=level_eps, level_statweight=level_g, level_alltrans_startdown=level_startdown,
        level_ndowntrans=level_ndown, level_nuptrans=level_nup, level_closestgroundlevelcont=level_closest,
        level_phixsstart=level_phixsstart, level_nphixstargets=level_nphixstargets,
        level_phixstargetstart=level_phixstargetstart, level_bflist_start=level_bflist_start,
        level_matransblock_start=level_mablock,
        alltrans_lineindex=at_line, alltrans_targetlevelindex=at_target, alltrans_einstein_A=at_A,
        alltrans_coll_str=at_cs, alltrans_osc_strength=at_f, alltrans_forbidden=at_forb,
        line_nu=line_nu, line_elementindex=line_elementindex, line_ionindex=line_ionindex,
        line_uniquelevelindex_lower=line_ul_lower, line_uniquelevelindex_upper=line_ul_upper,
        line_B_ul=B_ul, line_B_lu=B_lu,
        allphixs=allphixs, allphixstargets_levelindex=pt_level, allphixstargets_probability=pt_prob,
        allcont_nu_edge=[c[0] for c in conts], allcont_element=[c[1] for c in conts], allcont_ion=[c[2] for c in conts],
        allcont_level=[c[3] for c in conts], allcont_phixstargetindex=[c[4] for c in conts],
        allcont_upperlevel=[c[5] for c in conts], allcont_uniquelevelindex=[c[6] for c in conts],
        allcont_probability=[c[7] for c in conts], allcont_groundcontestimindex=[c[8] for c in conts],
        groundcont_nu_edge=groundcont_nu_edge,
        spontrecombcoeffs=spont.ravel(), corrphotoioncoeffs=gammac.ravel(), bfcooling_coeffs=bfcool.ravel(),
        coolinglist_type=cl_type, coolinglist_level=cl_level, coolinglist_phixstargetindex=cl_t,
    )
    d["_elements"] = elements
    d["_ion_ionpot"] = np.array(ion_ionpot)
    d["_level_ion"] = np.array(level_ion, dtype=np.int32)
    return d


def _zone_massfracs(elements, v, vmax):
    """W7-like abundance stratification by velocity: Fe-group core, Co/Ni (decayed 56Ni) zone,
    Si/S/Ca layer, O-rich outer layer. Elements absent from `elements` are dropped and the rest renormalised."""
    zones = [
        (0.12, {26: 0.55, 28: 0.30, 27: 0.10, 14: 0.03, 16: 0.02}),
        (0.42, {27: 0.55, 26: 0.22, 28: 0.13, 14: 0.05, 16: 0.03, 20: 0.02}),
        (0.62, {14: 0.55, 16: 0.28, 20: 0.05, 26: 0.06, 27: 0.02, 8: 0.04}),
        (2.00, {8: 0.70, 14: 0.20, 16: 0.06, 20: 0.01, 26: 0.03, 27: 0.0, 28: 0.0}),
    ]
    Zs = [e[0] for e in elements]
    X = np.zeros((len(v), len(Zs)), dtype=np.float64)
    x = v / vmax
    lo = 0.0
    for hi, comp in zones:
        sel = (x >= lo) & (x < hi)
        for j, Z in enumerate(Zs):
            X[sel, j] = comp.get(Z, 0.0)
        lo = hi
    s = X.sum(axis=1, keepdims=True)
    empty = s[:, 0] <= 0
    X[empty, :] = 1.0 / len(Zs)
    s = X.sum(axis=1, keepdims=True)
    return X / s


def make_grid_and_cells(atomic: dict, ncoord: int = 8, gridtype: int = abi.GRID_CARTESIAN3D, t_days: float = 20.0,
                        tmin_days: float = 2.0, vmax: float = 2.4e9, mass_msun: float = 1.38, ve: float = 2.7e8,
                        thick_below_v: float = 0.0, seed: int = 7, uniform_te: dict | None = None):
    """W7-like ejecta on an ncoord^3 Cartesian grid (or ncoord radial shells): exponential density profile,
    stratified composition, Saha ion balance at (T_e, nne). Returns (grid dict, cellstate dict, aux).
    uniform_te = dict(T=..., rho=...): every cell in strict thermodynamic equilibrium instead -- one density, T_e = T_R = T_J = T,
    W = 1, Saha-Boltzmann populations at T, unit photoionisation renormalisation (tests/test_physics_laws.py)."""
    rng = np.random.default_rng(seed)
    elements = atomic["_elements"]
    nelements = len(elements)
    nions = atomic["nions"]
    tmin = tmin_days * DAY
    t = t_days * DAY
    rmax = vmax * tmin

    if gridtype == abi.GRID_CARTESIAN3D:
        n = ncoord
        # setup_grid_cartesian_3d grid.cc:1269: coord_pos_min_tmin[axis][i] = -rmax + 2*i*rmax/ncoordgrid
        cmin = np.array([-rmax + (2 * i * rmax / n) for i in range(n)])
        centers_v = (cmin + rmax / n) / tmin
        iz, iy, ix = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
        # cellindex = ix + n*iy + n*n*iz (get_coordcellindexstride grid.cc:200)
        vx, vy, vz = centers_v[ix.ravel()], centers_v[iy.ravel()], centers_v[iz.ravel()]
        vr = np.sqrt(vx**2 + vy**2 + vz**2)
        ngrid = n**3
        nonempty_mask = vr < vmax
        ncoordgrid = [n, n, n]
        coord = [cmin, cmin, cmin]
        cellvol_tmin = np.full(ngrid, (2 * rmax / n) ** 3)
    elif gridtype == abi.GRID_CYLINDRICAL2D:
        # setup_grid_cylindrical_2d grid.cc:1290: ncoord cells in r_cyl over [0, rmax], 2*ncoord cells in z over [-rmax, rmax]
        nr, nz = ncoord, 2 * ncoord
        rmin = np.array([i * rmax / nr for i in range(nr)])
        zmin = np.array([rmax * (-1 + (i * 2.0 / nz)) for i in range(nz)])
        rout = np.concatenate([rmin[1:], [rmax]])
        zout = np.concatenate([zmin[1:], [rmax]])
        iz, ir = np.meshgrid(np.arange(nz), np.arange(nr), indexing="ij")  # cellindex = ir + nr*iz
        ir, iz = ir.ravel(), iz.ravel()
        vrc = 0.5 * (rmin[ir] + rout[ir]) / tmin
        vz = 0.5 * (zmin[iz] + zout[iz]) / tmin
        vr = np.sqrt(vrc**2 + vz**2)
        ngrid = nr * nz
        nonempty_mask = vr < vmax
        ncoordgrid = [nr, nz, 1]
        coord = [rmin, zmin, np.zeros(1)]
        cellvol_tmin = np.pi * (rout[ir] ** 2 - rmin[ir] ** 2) * (zout[iz] - zmin[iz])
    else:
        n = ncoord
        vout = vmax * (np.arange(1, n + 1) / n)
        vin = np.concatenate([[0.0], vout[:-1]])
        cmin = vin * tmin  # setup_grid_spherical_1d grid.cc:1285
        vr = 0.5 * (vin + vout)
        ngrid = n
        nonempty_mask = np.ones(n, dtype=bool)
        ncoordgrid = [n, 1, 1]
        coord = [cmin, np.zeros(1), np.zeros(1)]
        cellvol_tmin = 4.0 / 3.0 * np.pi * ((vout * tmin) ** 3 - (vin * tmin) ** 3)

    propcell_nonemptymgi = np.full(ngrid, -1, dtype=np.int32)
    nne_cells = int(nonempty_mask.sum())
    propcell_nonemptymgi[nonempty_mask] = np.arange(nne_cells, dtype=np.int32)
    v = vr[nonempty_mask]

    # exponential density: rho(v,t) = M / (8 pi ve^3 t^3) exp(-v/ve)
    rho = mass_msun * MSUN / (8 * np.pi * ve**3 * t**3) * np.exp(-v / ve)
    Te = 4000.0 + 12000.0 * np.exp(-v / 1.0e9)
    TR = Te * (1.0 - 0.05 * rng.random(nne_cells))
    TJ = Te * (0.92 + 0.05 * rng.random(nne_cells))
    vph = 0.35 * vmax
    W = np.where(v < vph, 0.5, 0.5 * (1.0 - np.sqrt(np.clip(1.0 - (vph / np.maximum(v, 1.0)) ** 2, 0.0, 1.0))))
    W = np.clip(W, 1e-3, 1.0)
    thick = (v < thick_below_v).astype(np.int32)
    W = np.where(thick == 1, 1.0, W)
    if uniform_te is not None:
        rho = np.full(nne_cells, float(uniform_te["rho"]))
        Te = np.full(nne_cells, float(uniform_te["T"]))
        TR, TJ, W = Te.copy(), Te.copy(), np.ones(nne_cells)

    X = _zone_massfracs(elements, v, vmax)  # [cells, elements]
    amass = np.array([_AMASS.get(Z, 2.0 * Z) for Z, _, _ in elements])
    nelem = rho[:, None] * X / (amass[None, :] * MH)  # number density of each element

    # Saha ion balance with partition functions U = sum g exp(-E/kT_J)
    level_ion = atomic["_level_ion"]
    eps = np.asarray(atomic["level_epsilon"])
    g = np.asarray(atomic["level_statweight"], dtype=np.float64)
    ion_uls = np.asarray(atomic["ion_uniquelevelindexstart"])
    U = np.zeros((nne_cells, nions))
    for ul in range(len(eps)):
        ui = level_ion[ul]
        U[:, ui] += g[ul] * np.exp(-(eps[ul] - eps[ion_uls[ui]]) / (KB * TJ))
    ionpot = atomic["_ion_ionpot"]
    elem_uiis = np.asarray(atomic["elem_uniqueionindexstart"])
    elem_nions = np.asarray(atomic["elem_nions"])
    lowest = np.asarray(atomic["elem_lowest_ionstage"])
    nne = 0.5 * nelem.sum(axis=1) + 1.0
    nion = np.zeros((nne_cells, nions))
    for _ in range(60):
        nne_new = np.zeros(nne_cells)
        for e in range(nelements):
            ni = elem_nions[e]
            s0 = elem_uiis[e]
            logr = np.zeros((nne_cells, ni))
            for k in range(1, ni):
                # n_{k}/n_{k-1} = (2 U_k / U_{k-1}) / (nne * SAHACONST-like) ... Saha
                phi = (2.0 * U[:, s0 + k] / U[:, s0 + k - 1]) * (2 * np.pi * ME * KB * Te / H**2) ** 1.5 * np.exp(
                    -ionpot[s0 + k - 1] / (KB * Te))
                logr[:, k] = logr[:, k - 1] + np.log(np.maximum(phi / nne, 1e-300))
            logr -= logr.max(axis=1, keepdims=True)
            f = np.exp(logr)
            f /= f.sum(axis=1, keepdims=True)
            nion[:, s0:s0 + ni] = nelem[:, e:e + 1] * f
            charge = (lowest[e] - 1) + np.arange(ni)
            nne_new += (nion[:, s0:s0 + ni] * charge[None, :]).sum(axis=1)
        nne = np.sqrt(nne * np.maximum(nne_new, 1e-3))
    Zs = np.array([Z for Z, _, _ in elements], dtype=np.float64)
    nnetot = (nelem * Zs[None, :]).sum(axis=1)
    g0 = g[ion_uls]
    groundpops = nion * g0[None, :] / U

    nbfg = atomic["nbfcontinua_ground"]
    renorm = 0.8 + 0.4 * rng.random((nne_cells, max(nbfg, 1)))
    if uniform_te is not None:
        renorm = np.ones((nne_cells, max(nbfg, 1)))

    grid = dict(gridtype=gridtype, ncoordgrid=ncoordgrid, ngrid=int(ngrid), npts_nonempty=nne_cells, tmin=float(tmin),
                vmax=float(vmax), rmax=float(rmax), coord_pos_min_tmin=coord, propcell_nonemptymgi=propcell_nonemptymgi)
    cells = dict(rho=rho, Te=Te, TJ=TJ, TR=TR, W=W, nne=nne, nnetot=nnetot, kappagrey=np.full(nne_cells, 0.1),
                 thick=thick, clumpfactor=np.ones(nne_cells), ion_groundlevelpops=groundpops.ravel(),
                 ion_partfuncts=U.ravel(), elem_massfracs=X.ravel(), corrphotoionrenorm=renorm.ravel(),
                 ffegrp=X[:, [k for k, (Z, _, _) in enumerate(elements) if 21 <= Z <= 28]].sum(axis=1) if any(
                     21 <= Z <= 28 for Z, _, _ in elements) else np.zeros(nne_cells))
    aux = dict(t=t, v=v, X=X, cellvol_tmin=cellvol_tmin[nonempty_mask], nonempty_cellindex=np.nonzero(nonempty_mask)[0])
    return grid, cells, aux


def make_timestep(t: float, width_frac: float = 0.05, vmax: float = 2.4e9, nts: int = 10) -> abi.Timestep:
    width = t * width_frac
    mid = np.sqrt(t * (t + width))  # logarithmic mid time
    return abi.Timestep(nts, t, width, mid, min(1e35, vmax * mid / 10.0))  # update_grid.cc:753


def make_packets(model: abi.Model, aux: dict, npackets: int, seed_base: int = 12345, kpkt_fraction: float = 0.1,
                 e_total: float = 1e45, seed: int = 99, gamma_fraction: float = 0.0, pellet_fraction: float = 0.0,
                 ts_width_frac: float = 0.05, early_pellets: bool = False, cells_only=None) -> np.ndarray:
    """Packets at the start of the timestep: thermal energy waiting to be emitted (TYPE_PRE_KPKT -> blackbody
    r-packet, kpkt.cc:399) or k-packets (kpkt.cc:425), placed in non-empty cells with probability ~ rho * X(Co,Ni,Fe)."""
    rng = np.random.default_rng(seed)
    t = aux["t"]
    d = model.d
    tmin = d["tmin"]
    pk = np.zeros(npackets, dtype=abi.PACKET_DTYPE)
    cells = aux["nonempty_cellindex"]
    w = aux["cellvol_tmin"] * np.exp(-aux["v"] / 4.0e8)
    if cells_only is not None:  # start every packet in this subset of the non-empty cells (dense sampling of a few cells)
        mask = np.zeros(len(w), dtype=bool)
        mask[np.asarray(cells_only)] = True
        w = np.where(mask, w, 0.0)
    w = w / w.sum()
    which = rng.choice(len(cells), size=npackets, p=w)
    cellindex = cells[which]
    u = 0.02 + 0.96 * rng.random((npackets, 3))
    if d["gridtype"] == abi.GRID_CARTESIAN3D:
        n = int(d["ncoordgrid"][0])
        cmin = d["coord_pos_min_tmin"][0]
        dx = 2 * d["rmax"] / n
        ix = cellindex % n
        iy = (cellindex // n) % n
        iz = cellindex // (n * n)
        pos = np.stack([cmin[ix] + u[:, 0] * dx, cmin[iy] + u[:, 1] * dx, cmin[iz] + u[:, 2] * dx], axis=1) * (t / tmin)
    elif d["gridtype"] == abi.GRID_CYLINDRICAL2D:
        nr = int(d["ncoordgrid"][0])
        rmin, zmin = d["coord_pos_min_tmin"][0], d["coord_pos_min_tmin"][1]
        rout = np.concatenate([rmin[1:], [d["rmax"]]])
        zout = np.concatenate([zmin[1:], [d["rmax"]]])
        ir, iz = cellindex % nr, cellindex // nr
        rc = np.sqrt(rmin[ir] ** 2 + u[:, 0] * (rout[ir] ** 2 - rmin[ir] ** 2)) * (t / tmin)
        phi = 2 * np.pi * u[:, 1]
        z = (zmin[iz] + u[:, 2] * (zout[iz] - zmin[iz])) * (t / tmin)
        pos = np.stack([rc * np.cos(phi), rc * np.sin(phi), z], axis=1)
    else:
        n = int(d["ncoordgrid"][0])
        cmin = d["coord_pos_min_tmin"][0]
        cmax = np.concatenate([cmin[1:], [d["rmax"]]])
        r = (cmin[cellindex] + u[:, 0] * (cmax[cellindex] - cmin[cellindex])) * (t / tmin)
        mu = 2 * u[:, 1] - 1
        phi = 2 * np.pi * u[:, 2]
        s = np.sqrt(1 - mu**2)
        pos = np.stack([r * s * np.cos(phi), r * s * np.sin(phi), r * mu], axis=1)
    pk["pos"] = pos
    pk["prop_time"] = t
    pk["e_cmf"] = e_total / npackets
    pk["e_rf"] = e_total / npackets
    pk["next_trans"] = -1
    pk["emissiontype"] = abi.EMTYPE_NOTSET
    pk["em_pos"] = np.nan
    pk["em_time"] = -1.0
    pk["trueemissiontype"] = abi.EMTYPE_NOTSET
    pk["trueem_pos"] = np.nan
    pk["trueem_time"] = -1.0
    pk["type"] = np.where(rng.random(npackets) < kpkt_fraction, abi.TYPE_KPKT, abi.TYPE_PRE_KPKT)
    pk["cellindex"] = cellindex
    if gamma_fraction > 0:
        # gamma packets as pellet_gamma_decay() leaves them (gammapkt.cc:894): a line of the 56Ni/56Co decay spectra (a few
        # below the Thomson limit of the Compton treatment and above the pair-production threshold included), isotropic
        # direction; rest-frame quantities from the Doppler factor of the homologous flow
        isg = rng.random(npackets) < gamma_fraction
        lines_mev = np.array([0.004, 0.158, 0.27, 0.48, 0.75, 0.812, 0.847, 1.238, 1.771, 2.598, 3.253])
        e_mev = lines_mev[rng.integers(0, len(lines_mev), npackets)]
        mu = 2 * rng.random(npackets) - 1
        ph = 2 * np.pi * rng.random(npackets)
        sth = np.sqrt(1 - mu**2)
        gdir = np.stack([sth * np.cos(ph), sth * np.sin(ph), mu], axis=1)
        clight = 2.99792458e10
        vel = pos / t
        ndotv = (gdir * vel).sum(axis=1)
        gamma_rel = 1.0 / np.sqrt(1.0 - (vel**2).sum(axis=1) / clight**2)
        doppler = gamma_rel * (1.0 - ndotv / clight)  # nu_cmf / nu_rf
        nu_cmf = e_mev * 1.0e6 * 1.6021772e-12 / 6.6260755e-27
        pk["type"] = np.where(isg, abi.TYPE_GAMMA, pk["type"])
        pk["dir"] = np.where(isg[:, None], gdir, pk["dir"])
        pk["nu_cmf"] = np.where(isg, nu_cmf, pk["nu_cmf"])
        pk["nu_rf"] = np.where(isg, nu_cmf / doppler, pk["nu_rf"])
        pk["e_rf"] = np.where(isg, pk["e_cmf"] / doppler, pk["e_rf"])
    pk["escape_time"] = -1.0
    pk["tdecay"] = -1.0
    pk["pellet_decaytype"] = -1
    if pellet_fraction > 0:
        # pellets as packet_init() leaves them (packet.cc): a decay time, the gamma line already chosen (nu_cmf, or -1 for
        # a nuclide without a gamma spectrum), or marked as a particle-emitting decay with its decay type
        isp = (rng.random(npackets) < pellet_fraction) & (pk["type"] != abi.TYPE_GAMMA)
        width = t * ts_width_frac
        tdec = t + rng.random(npackets) * 3.0 * width          # about a third decays within the timestep
        if early_pellets:                                      # nts == 0 only: some decayed before the simulation start
            tdec = np.where(rng.random(npackets) < 0.3, t * (0.2 + 0.7 * rng.random(npackets)), tdec)
        particle = rng.random(npackets) < 0.35
        dtype = rng.choice([0, 2, 3, 5], size=npackets)        # alpha, beta+, beta-, spontaneous fission (decay.h:21)
        lines_mev = np.array([0.158, 0.48, 0.75, 0.812, 0.847, 1.238, 1.771, 2.598])
        nu_line = lines_mev[rng.integers(0, len(lines_mev), npackets)] * 1.0e6 * 1.6021772e-12 / 6.6260755e-27
        nu_line = np.where(rng.random(npackets) < 0.05, -1.0, nu_line)
        # a particle-emitting pellet carries the kinetic energy of its particles instead (0.3 .. 5 MeV), which the
        # time-dependent thermalisation schemes read (update_packets.cc:93)
        nu_line = np.where(particle, (0.3 + 4.7 * rng.random(npackets)) * 1.0e6 * 1.6021772e-12 / 6.6260755e-27, nu_line)
        pk["type"] = np.where(isp, abi.TYPE_RADIOACTIVE_PELLET, pk["type"])
        pk["tdecay"] = np.where(isp, tdec, pk["tdecay"])
        pk["originated_from_particlenotgamma"] = np.where(isp & particle, 1, 0)
        pk["pellet_decaytype"] = np.where(isp, np.where(particle, dtype, 1), -1)
        pk["nu_cmf"] = np.where(isp, nu_line, pk["nu_cmf"])
    pk["number"] = np.arange(npackets, dtype=np.int32)
    pk["pellet_nucindex"] = -1
    abi.seed_packet_rng(pk, seed_base)
    return pk


PRESETS = {
    # name: (elements, nlevels_per_ion, line_fraction, nphixspoints)
    "tiny": ([(14, 1, 3), (26, 1, 4)], 6, 0.5, 12),
    "small": ([(14, 1, 4), (26, 1, 5), (27, 1, 5)], 12, 0.4, 40),
    "w7": ([(8, 1, 4), (14, 1, 5), (16, 1, 5), (20, 1, 4), (26, 1, 5), (27, 1, 5), (28, 1, 5)], 60, 0.3, 100),
    # the same ions with 170 levels each: 5.6e3 levels, ~1.4e5 lines -- the size at which the cell cache of a 50^3 grid no
    # longer fits one tile (profiles/r03/tiling.md)
    "w7big": ([(8, 1, 4), (14, 1, 5), (16, 1, 5), (20, 1, 4), (26, 1, 5), (27, 1, 5), (28, 1, 5)], 170, 0.3, 100),
    # the size of the small real classic data set (CD23-like: >= 4e5 lines): 325 levels per ion, 1.07e4 levels. With a static macro-atom
    # record for every level its cell-cache row is ~11 MB (710 GB for the 50^3 grid: four tiles); with on-demand records (tables.h) ~3.5 MB
    "cd23like": ([(8, 1, 4), (14, 1, 5), (16, 1, 5), (20, 1, 4), (26, 1, 5), (27, 1, 5), (28, 1, 5)], 325, 0.3, 100),
}


def build(preset: str = "small", ncoord: int = 8, gridtype: int = abi.GRID_CARTESIAN3D, seed: int = 1,
          t_days: float = 20.0, thick_below_v: float = 0.0, width_frac: float = 0.05, nts: int = 10, tmin_days: float = 2.0,
          options: str = "classic", host_expopac: bool = False):
    """One-call construction of (Model, CellState, Timestep, aux). host_expopac: hand synthetic expansion-opacity tables
    over with the cell state (expansion-opacity builds; otherwise the engine calculates them from the line list)."""
    elements, nl, lf, npx = PRESETS[preset]
    atomic = make_atomic(seed=seed, elements=elements, nlevels_per_ion=nl, line_fraction=lf, nphixspoints=npx, options=options)
    grid, cells, aux = make_grid_and_cells(atomic, ncoord=ncoord, gridtype=gridtype, t_days=t_days, tmin_days=tmin_days,
                                           thick_below_v=thick_below_v, seed=seed + 100)
    md = {k: v for k, v in atomic.items() if not k.startswith("_")}
    md.update(grid)
    # whole-ejecta scalars of the Barnes thermalisation scheme (grid.h:139 get_ejecta_kinetic_energy, grid.h:40 mtot_input)
    m_cell = np.asarray(cells["rho"], dtype=np.float64) * (aux["t"] / grid["tmin"]) ** 3 * aux["cellvol_tmin"]
    md["rho_tmin"] = (np.asarray(cells["rho"], dtype=np.float64) * (aux["t"] / grid["tmin"]) ** 3).astype(np.float32)  # grid::get_rho_tmin
    md["mtot_input"] = float(m_cell.sum())
    md["ejecta_kinetic_energy"] = float((0.5 * m_cell * aux["v"] ** 2).sum())
    like = abi.inputs_like(options)
    if "vpkt" in options:
        md.update(vpkt_config(expopac="expopac" in options))
    if like == "classic_gamma_xcom":
        md.update(nonthermal_model_inputs(atomic))     # element masses for the number densities
        md.update(xcom_tables(atomic))
    if options == "ci_kilonova_xcom":
        # USE_CALCULATED_MEANATOMICWEIGHT (artisoptions_kilonova_lte.h:120): grid::elem_meanweight_allcells, the mean weight
        # of each element's isotope mix in each cell (here: the stable mean scattered by a few per cent from cell to cell)
        rng = np.random.default_rng(seed + 600)
        amass = np.asarray(md["elem_meannucmass"], dtype=np.float64)
        cells["elem_meanweight"] = (amass[None, :] * rng.uniform(0.97, 1.05, (grid["npts_nonempty"], len(amass)))).astype(np.float32).ravel()
    model = abi.Model(md)
    if "expopac" in options and host_expopac:
        cells.update(expansion_opacity_cellstate(cells, grid["npts_nonempty"], seed=seed + 400))
    if options in abi.NEBULAR_FAMILY:
        cells.update(nebular_cellstate(atomic, cells, grid["npts_nonempty"], seed=seed + 200, nbins=abi.NEBULAR_FAMILY[options]))
        cells.update(nonthermal_cellstate(atomic, cells, grid["npts_nonempty"], seed=seed + 300))
        md.update(nonthermal_model_inputs(atomic))
        if options == "nltenebular_lineest":
            # radfield.cc detailed_lineindices: every seventh line has its own intensity estimator; the normalised
            # intensities of the previous timestep scatter around the dilute blackbody
            rng = np.random.default_rng(seed + 500)
            idx = np.arange(3, atomic["nlines"], 7, dtype=np.int32)
            nu = np.asarray(atomic["line_nu"], dtype=np.float64)[idx]
            TR = np.asarray(cells["TR"], dtype=np.float64)[:, None]
            Wd = np.asarray(cells["W"], dtype=np.float64)[:, None]
            jb = Wd * 2 * H * nu[None, :] ** 3 / CLIGHT ** 2 / np.expm1(np.minimum(H * nu[None, :] / (KB * TR), 700.0))
            jb *= rng.uniform(0.3, 3.0, jb.shape)
            md.update(detailed_lineindices=idx, detailed_linecount=len(idx))
            cells["Jb_lu_normed"] = jb.ravel()
        if like == "nltephotospheric":
            # LEVEL_HAS_BFEST (artisoptions_nltephotospheric_dynamic_ion_range.h:80): estimators for the lowest levels only;
            # globals::allcont.bfestimindex is the running count over the continua that have one (input.cc:932-947)
            has = np.asarray(atomic["allcont_level"]) <= 3
            idx = np.where(has, np.cumsum(has) - 1, -1).astype(np.int32)
            md.update(allcont_bfestimindex=idx, nbfestim=int(has.sum()))
        model = abi.Model(md)
    cs = abi.CellState(cells)
    ts = make_timestep(aux["t"], width_frac=width_frac, vmax=grid["vmax"], nts=nts)
    return model, cs, ts, aux


def evolve_cellstate(cs: abi.CellState, t0: float, t1: float, tfloor: float = 2500.0) -> abi.CellState:
    """The cell state of build() carried from time t0 to t1 by a DETERMINISTIC host rule that stands in for update_grid() (which is not part of
    the packet path) in multi-timestep tests: homologous expansion thins every density as (t0/t1)^3 -- rho, nne, nnetot, the ions' ground-level
    populations, the NLTE level populations and the non-thermal deposition rate density where the state has them -- and the temperatures
    (T_e, T_J, T_R, the radiation-field bins' T_R) fall as t0/t1, not below `tfloor`; everything else (W, partition functions, mass
    fractions, photoionisation renormalisation / coefficients, Spencer-Fano fractions) stays. Same arrays for oracle and engine."""
    f3, f1 = (t0 / t1) ** 3, t0 / t1
    d = dict(cs.d)
    for k in ("rho", "nne", "nnetot", "ion_groundlevelpops", "levelpops", "nt_deposition_rate_density"):
        if d.get(k) is not None:
            d[k] = (np.asarray(d[k], dtype=np.float64) * f3).astype(np.asarray(d[k]).dtype)
    for k in ("Te", "TJ", "TR", "radfieldbin_T_R"):
        if d.get(k) is not None:
            a = np.asarray(d[k], dtype=np.float64)
            d[k] = np.maximum(a * f1, np.minimum(a, tfloor)).astype(np.asarray(d[k]).dtype)
    return abi.CellState(d)


def vpkt_config(expopac: bool = False) -> dict:
    """What read_vpktparameterfile() (vpkt.cc:673) leaves of a vpkt.txt like tests/classicmode_3d_inputfiles/vpkt.txt: three
    observers near +z, on the equator and towards -z (not ON the axis as in that file: there the meridian frame of the
    Stokes rotation, vectors.h meridian(), is singular and the sign of Q / U hangs on the last bit of sin / cos / acos --
    measured: glibc and the GPU's libm give opposite signs for single packets, equal I); four spectra per observer with different opacity choices (with the binned
    expansion opacities no element can be excluded, vpkt.cc:395); packets arriving inside the spectra's own window; one
    frequency range; a low optical-depth cut so that it is exercised; the velocity-grid map on for the line-by-line form."""
    day = 86400.0
    numin, numax = CLIGHT / 10000 * 1e8, CLIGHT / 3500 * 1e8
    return dict(vpkt_nobsdirections=3, vpkt_obsdirs_costheta=np.array([0.9, 0.0, -0.6]), vpkt_obsdirs_phi=np.array([0.0, 0.7, 0.0]),
                vpkt_nspectraperobsdir=4, vpkt_opacityexclusions=np.array([0, -1, -3, -4] if expopac else [0, -1, -2, 26], dtype=np.int32),
                vpkt_timemin_input=3 * day, vpkt_timemax_input=8 * day, vpkt_nwavelengthranges=1,
                vpkt_numin_input=np.array([numin]), vpkt_numax_input=np.array([numax]), vpkt_tau_max=8.0,
                vpkt_vgrid_on=0 if expopac else 1, vpkt_tmin_grid=3 * day, vpkt_tmax_grid=8 * day,
                vpkt_grid_nwavelengthranges=2, vpkt_nu_grid_min=np.array([numin, CLIGHT / 6000 * 1e8]),
                vpkt_nu_grid_max=np.array([CLIGHT / 6000 * 1e8, numax]), vpkt_nprocs=1)


def expansion_opacity_cellstate(cells: dict, ncell: int, seed: int = 401) -> dict:
    """What calculate_expansion_opacities() (rpkt.cc:1071) leaves per cell: a binned line opacity kappa(lambda) [cm^2/g]
    on the 20 A grid from 60 A to 40000 A (a forest that thins out to the red, with bins and whole stretches without
    lines), and the running integral over the bins of (kappa + a continuum share) * B_nu(T_e) * delta_nu."""
    rng = np.random.default_rng(seed)
    nb = abi.EXPOPAC_NBINS
    lam_lo = 60.0 + 20.0 * np.arange(nb)
    nu_upper = 1e8 * CLIGHT / lam_lo
    nu_lower = 1e8 * CLIGHT / (lam_lo + 20.0)
    nu_mid = 0.5 * (nu_upper + nu_lower)
    envelope = 10.0 ** (1.0 - 2.5 * (lam_lo / 40000.0))                       # ~10 cm^2/g in the UV, ~0.03 in the IR
    kappa = envelope[None, :] * 10.0 ** rng.normal(0.0, 0.6, (ncell, nb))
    kappa[rng.random((ncell, nb)) < 0.15] = 0.0
    kappa[:, lam_lo > 30000.0] = 0.0
    kappa = kappa.astype(np.float32)
    Te = np.asarray(cells["Te"], dtype=np.float64)
    x = H * nu_mid[None, :] / (KB * Te[:, None])
    planck = 2 * H * nu_mid[None, :] ** 3 / CLIGHT ** 2 / np.expm1(np.minimum(x, 700.0))
    cum = np.cumsum((kappa.astype(np.float64) + 1e-3) * planck * (nu_upper - nu_lower)[None, :], axis=1)
    return dict(expansionopacities=kappa.ravel(), expansionopacity_planck_cumulative=cum.ravel())


def xcom_tables(atomic: dict) -> dict:
    """XCOM-like photoionisation cross sections per element of the model (gammapkt.cc:244: energy [MeV] rising, sigma [cm^2]):
    sigma ~ Z^4.5 E^-3 with K-edge-like steps, 30-60 points from 1 keV to 100 MeV; the last element has no data."""
    rng = np.random.default_rng(77)
    start, en, sg = [0], [], []
    elements = atomic["_elements"]
    for k, (Z, _, _) in enumerate(elements):
        if k == len(elements) - 1 and len(elements) > 1:
            start.append(start[-1])
            continue
        n = int(rng.integers(30, 61))
        E = np.sort(10 ** rng.uniform(-3.0, 2.0, n))
        edge = 1e-5 * Z ** 2.2                                  # MeV
        sigma = 3e-29 * Z ** 4.5 * E ** -3.0 * np.where(E > edge, 1.0, 0.12) + 1e-30
        en.extend(E)
        sg.extend(sigma)
        start.append(start[-1] + n)
    return dict(xcom_elem_start=np.array(start, dtype=np.int32), xcom_energy=np.array(en), xcom_sigma=np.array(sg))


def nonthermal_model_inputs(atomic: dict) -> dict:
    """Static inputs of the non-thermal channels (include/artis_amd.h artis_model): mean nuclear mass of each element and
    the sum over shells of occupancy / binding energy of each ion (nonthermal.cc:553; here one valence-like term)."""
    elements = atomic["_elements"]
    amass = np.array([_AMASS.get(Z, 2.0 * Z) for Z, _, _ in elements]) * MH
    ionpot = np.asarray(atomic["_ion_ionpot"], dtype=np.float64)
    elem_uiis = np.asarray(atomic["elem_uniqueionindexstart"])
    lowest = np.asarray(atomic["elem_lowest_ionstage"])
    binding = np.zeros(atomic["nions"])
    for e, (Z, _, _) in enumerate(elements):
        for k in range(int(atomic["elem_nions"][e])):
            nbound = Z - (int(lowest[e]) - 1 + k)
            binding[elem_uiis[e] + k] = max(nbound, 0) / max(ionpot[elem_uiis[e] + k], 1e-12)
    return dict(elem_meannucmass=amass, ion_nt_sum_q_over_binding=binding)


def nonthermal_cellstate(atomic: dict, cells: dict, ncell: int, seed: int = 301, stored: int = 48) -> dict:
    """A Spencer-Fano solution per cell as the host's solver stores it (nonthermal.cc:215-245): deposition fractions,
    effective ionisation potentials (a few zero: no cross-section data, the Axelrod work-function fallback), Auger
    probabilities that sum to one in float, and a truncated list of excitation transitions sorted by alltransindex.
    Some cells have no deposition at all (rate density 0: select_nt_ionisation() finds no ion)."""
    rng = np.random.default_rng(seed)
    nions = atomic["nions"]
    ionpot = np.asarray(atomic["_ion_ionpot"], dtype=np.float64)
    rho = np.asarray(cells["rho"], dtype=np.float64)
    dep = rho * 1e8 * rng.uniform(0.5, 1.5, ncell)
    dep[rng.random(ncell) < 0.08] = 0.0
    frac_ion = rng.uniform(0.02, 0.3, ncell).astype(np.float32)
    frac_exc = rng.uniform(0.02, 0.2, ncell).astype(np.float32)
    eff = (ionpot[None, :] * rng.uniform(1.5, 4.0, (ncell, nions))).astype(np.float32)
    eff[rng.random((ncell, nions)) < 0.05] = 0.0

    # number of ions above each ion: an ionisation cannot eject more Auger electrons than there are stages left
    elem_nions = np.asarray(atomic["elem_nions"])
    elem_uiis = np.asarray(atomic["elem_uniqueionindexstart"])
    nabove = np.concatenate([np.arange(n - 1, -1, -1) for n in elem_nions])
    assert len(nabove) == nions and elem_uiis[-1] + elem_nions[-1] == nions

    def auger():
        p = np.zeros((ncell, nions, abi.NT_NAUGER), dtype=np.float32)
        p[:, :, 1] = rng.uniform(0.0, 0.2, (ncell, nions)) * (nabove >= 2)[None, :]
        p[:, :, 2] = rng.uniform(0.0, 0.05, (ncell, nions)) * (nabove >= 3)[None, :]
        p[:, :, 0] = np.float32(1.0) - p[:, :, 1] - p[:, :, 2]
        return p

    # candidate excitations: the upward transitions of the lowest levels of every ion
    level_ion = np.asarray(atomic["_level_ion"])
    ion_uls = np.asarray(atomic["ion_uniquelevelindexstart"])
    startdown = np.asarray(atomic["level_alltrans_startdown"])
    ndown = np.asarray(atomic["level_ndowntrans"])
    nup = np.asarray(atomic["level_nuptrans"])
    cand = []
    for ul in range(len(level_ion)):
        if ul - ion_uls[level_ion[ul]] < 5:
            cand.extend(range(int(startdown[ul] + ndown[ul]), int(startdown[ul] + ndown[ul] + nup[ul])))
    cand = np.array(cand, dtype=np.int32)
    count = np.zeros(ncell, dtype=np.int32)
    fdep = np.zeros((ncell, stored))
    rcoeff = np.zeros((ncell, stored))
    ati = np.zeros((ncell, stored), dtype=np.int32)
    for c in range(ncell):
        n = int(min(len(cand), rng.integers(0, stored + 1)))
        count[c] = n
        if n == 0:
            continue
        ati[c, :n] = np.sort(rng.choice(cand, size=n, replace=False))
        w = rng.random(n) + 0.05
        fdep[c, :n] = w / w.sum() * 0.8 * float(frac_exc[c])
        rcoeff[c, :n] = 10 ** rng.uniform(-1.0, 3.0, n)
    return dict(nt_frac_ionisation=frac_ion, nt_frac_excitation=frac_exc, nt_deposition_rate_density=dep,
                nt_eff_ionpot=eff.ravel(), nt_prob_num_auger=auger().ravel(), nt_ionenfrac_num_auger=auger().ravel(),
                nt_exc_count=count, nt_exc_frac_deposition=fdep.ravel(), nt_exc_ratecoeffperdeposition=rcoeff.ravel(),
                nt_exc_alltransindex=ati.ravel(), nt_excitations_stored=stored)


def nebular_cellstate(atomic: dict, cells: dict, ncell: int, seed: int = 201, nbins: int = 256) -> dict:
    """What the host's NLTE / radiation-field solvers hand to the packet path under artisoptions_nltenebular.h
    (include/artis_amd.h artis_cellstate): level populations with departures from LTE, photoionisation coefficients of
    every bound-free pair, and W, T_R of the 256 radiation-field bins (a few bins without a solution: W = -1)."""
    rng = np.random.default_rng(seed)
    eps = np.asarray(atomic["level_epsilon"])
    g = np.asarray(atomic["level_statweight"], dtype=np.float64)
    level_ion = atomic["_level_ion"]
    ion_uls = np.asarray(atomic["ion_uniquelevelindexstart"])
    nlevels, nions = len(eps), atomic["nions"]
    Te = np.asarray(cells["Te"], dtype=np.float64)
    ground = np.asarray(cells["ion_groundlevelpops"], dtype=np.float64).reshape(ncell, nions)
    start = ion_uls[level_ion]
    # Boltzmann at T_e times a departure coefficient (ground levels keep the ion's ground population)
    boltz = g[None, :] / g[start][None, :] * np.exp(-(eps - eps[start])[None, :] / (KB * Te[:, None]))
    dep = np.exp(rng.normal(0.0, 0.5, size=(ncell, nlevels)))
    dep[:, ion_uls] = 1.0
    pops = np.maximum(ground[:, level_ion] * boltz * dep, 1e-40)
    npt = atomic["nphixstargets_total"]
    corr = 10 ** rng.uniform(-4.0, 1.0, size=(ncell, max(npt, 1))) * np.asarray(cells["W"], dtype=np.float64)[:, None]
    nb = nbins
    W = np.asarray(cells["W"], dtype=np.float64)[:, None] * rng.uniform(0.3, 1.7, size=(ncell, nb))
    W[rng.random((ncell, nb)) < 0.05] = -1.0
    TR = np.asarray(cells["TR"], dtype=np.float64)[:, None] * rng.uniform(0.7, 1.3, size=(ncell, nb))
    return dict(levelpops=pops.ravel(), corrphotoioncoeff=corr.ravel(), radfieldbin_W=W.ravel(), radfieldbin_T_R=TR.ravel())
