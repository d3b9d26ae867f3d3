#!/usr/bin/env python3
"""Emergent spectrum and light curve of a packet array, binned by the reference's rules.

Host-side post-processing (the reference's exspec / write_partial_lightcurve_spectra, spectrum_lightcurve.cc), NOT part
of the packet path and not used by the engine: it turns the engine's (or the oracle's) escaped packets into the
artefact the reference compares runs by -- spec.out / light_curve.out -- so that a reference run of the same model can
be laid next to an engine run as soon as one exists. Angle-averaged (dirbin = -1) or one of the MABINS = 100 equal-solid-angle
direction bins of the reference (dirbin >= 0), one rank (nprocs_exspec = 1).

Rules restated (file:line of the reference):
  * a packet counts if type == TYPE_ESCAPE and escape_type == TYPE_RPKT              spectrum_lightcurve.cc:254-257
  * arrival time  t_arrive = escape_time - dot(pos, dir) / c                          :555 / :695
  * time bin = the timestep [start, next start) containing t_arrive                   :209 get_timestep
  * MNUBINS = 1000 log-spaced frequency bins over (nu_min, nu_max); index
    clamp(floor((ln nu - ln nu_min) / dlognu), 0, MNUBINS-1); edges stored as float32 exspec.h:8, sn3d.h:134-144, :487-504
  * flux += e_rf / width[nts] / delta_freq[nnu] / 4e12 / pi / PARSEC^2                :563  [erg/s/cm^2/Hz at 1 Mpc]
  * luminosity light curve  L[nts] += e_rf / width[nts]                               :698
  * comoving light curve  t_cmf = escape_time * sqrt(1 - vmax^2/c^2):
    Lcmf[nts] += e_cmf / width[nts] / sqrt(1 - vmax^2/c^2)                            :702-711
  * direction-resolved (dirbin >= 0): only packets with get_escapedirectionbin(dir) == dirbin, every contribution times
    MABINS (a bin sees 1/MABINS of the sphere)                                        :545, :562, :689-691; vectors.h:147
"""
from __future__ import annotations

import numpy as np

CLIGHT = 2.99792458e10
PARSEC = 3.0857e18  # constants.h
MNUBINS = 1000
TYPE_ESCAPE, TYPE_RPKT = 32, 11
NPHIBINS, NCOSTHETABINS = 10, 10  # exspec.h:10-11
MABINS = NPHIBINS * NCOSTHETABINS  # exspec.h:12


def escapedirectionbin(dirs: np.ndarray) -> np.ndarray:
    """get_escapedirectionbin (vectors.h:147) for an [n, 3] array of directions: costheta bin (about syn_dir = z, constants.h:94) *
    NPHIBINS + phi bin, the phi bins in decreasing phi order."""
    d = np.asarray(dirs, dtype=np.float64)
    d = d / np.sqrt((d * d).sum(axis=1))[:, None]
    syn = np.array([0., 0., 1.])
    xhat = np.array([1., 0., 0.])
    costheta = d @ syn
    costhetabin = np.clip(((costheta + 1.0) * NCOSTHETABINS / 2.0).astype(np.int64), 0, NCOSTHETABINS - 1)
    vec1 = np.cross(d, syn)
    vec2 = np.cross(xhat, syn)
    vec1_len = np.sqrt((vec1 * vec1).sum(axis=1))
    safe = np.where(vec1_len > 1e-12, vec1_len, 1.0)
    cosphi = np.where(vec1_len > 1e-12, np.clip((vec1 @ vec2) / safe, -1.0, 1.0), 1.0)
    vec3 = np.cross(vec2, syn)
    testphi = vec1 @ vec3
    phi = np.where(testphi > 0, np.arccos(cosphi), np.arccos(cosphi) + np.pi)
    phibin = np.clip((phi / 2. / np.pi * NPHIBINS).astype(np.int64), 0, NPHIBINS - 1)
    return costhetabin * NPHIBINS + phibin


def timestep_index(t: np.ndarray, starts: np.ndarray, tmax: float) -> np.ndarray:
    """get_timestep(): index of the timestep [starts[i], starts[i+1]) (the last one ends at tmax) containing t, or -1"""
    idx = np.searchsorted(starts, t, side="right") - 1
    ok = (t >= starts[0]) & (t < tmax)
    return np.where(ok, idx, -1)


def spectrum_and_lightcurve(packets: np.ndarray, ts_starts, ts_widths, tmin: float, tmax: float, vmax: float,
                            nu_min: float = 1e14, nu_max: float = 5e15, dirbin: int = -1):
    """Returns dict(flux[MNUBINS, nts], lower_freq, delta_freq, lum[nts], lumcmf[nts]). dirbin >= 0: the spectrum and light curves
    seen from that direction bin (add_to_spec_res / add_to_lc_res with dirbin, spectrum_lightcurve.cc:545, :689)."""
    starts = np.asarray(ts_starts, dtype=np.float64)
    widths = np.asarray(ts_widths, dtype=np.float64)
    nts_all = len(starts)
    sel = (packets["type"] == TYPE_ESCAPE) & (packets["escape_type"] == TYPE_RPKT)
    p = packets[sel]
    solidanglefactor = 1.0
    if dirbin >= 0:
        p = p[escapedirectionbin(p["dir"]) == dirbin]
        solidanglefactor = float(MABINS)
    dlognu = (np.log(nu_max) - np.log(nu_min)) / MNUBINS
    edges = np.exp(np.log(nu_min) + np.arange(MNUBINS + 1) * dlognu)
    lower = edges[:-1].astype(np.float32)
    delta = (edges[1:] - lower.astype(np.float64)).astype(np.float32)
    t_arrive = p["escape_time"].astype(np.float64) - (p["pos"] * p["dir"]).sum(axis=1) / CLIGHT
    flux = np.zeros((MNUBINS, nts_all))
    lum = np.zeros(nts_all)
    lumcmf = np.zeros(nts_all)
    ok_t = (t_arrive > tmin) & (t_arrive < tmax)
    nts = timestep_index(t_arrive, starts, tmax)
    ok_t &= nts >= 0
    np.add.at(lum, nts[ok_t], p["e_rf"][ok_t] / widths[nts[ok_t]] * solidanglefactor)
    ok = ok_t & (p["nu_rf"] > nu_min) & (p["nu_rf"] < nu_max)
    nnu = np.clip(np.floor((np.log(p["nu_rf"][ok]) - np.log(nu_min)) / dlognu).astype(np.int64), 0, MNUBINS - 1)
    dE = p["e_rf"][ok] / widths[nts[ok]] / delta[nnu].astype(np.float64) / 4.e12 / np.pi / PARSEC / PARSEC * solidanglefactor
    np.add.at(flux, (nnu, nts[ok]), dE)
    inv_gamma = np.sqrt(1. - (vmax * vmax / CLIGHT**2))
    t_cmf = p["escape_time"].astype(np.float64) * inv_gamma
    ok_c = (t_cmf > tmin) & (t_cmf < tmax)
    ntc = timestep_index(t_cmf, starts, tmax)
    ok_c &= ntc >= 0
    np.add.at(lumcmf, ntc[ok_c], p["e_cmf"][ok_c] / widths[ntc[ok_c]] * solidanglefactor / inv_gamma)
    return dict(flux=flux, lower_freq=lower, delta_freq=delta, lum=lum, lumcmf=lumcmf, nescaped=int(len(p)))
