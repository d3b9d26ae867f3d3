#!/usr/bin/env python3
"""Emergent spectrum and light curve of a packet array, binned by the reference's rules.

Host-side post-processing (the reference's exspec / write_partial_lightcurve_spectra, spectrum_lightcurve.cc), NOT part
of the packet path and not used by the engine: it turns the engine's (or the oracle's) escaped packets into the
artefact the reference compares runs by -- spec.out / light_curve.out -- so that a reference run of the same model can
be laid next to an engine run as soon as one exists. Angle-averaged (dirbin = -1), one rank (nprocs_exspec = 1).

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
"""
from __future__ import annotations

import numpy as np

CLIGHT = 2.99792458e10
PARSEC = 3.0857e18  # constants.h
MNUBINS = 1000
TYPE_ESCAPE, TYPE_RPKT = 32, 11


def timestep_index(t: np.ndarray, starts: np.ndarray, tmax: float) -> np.ndarray:
    """get_timestep(): index of the timestep [starts[i], starts[i+1]) (the last one ends at tmax) containing t, or -1"""
    idx = np.searchsorted(starts, t, side="right") - 1
    ok = (t >= starts[0]) & (t < tmax)
    return np.where(ok, idx, -1)


def spectrum_and_lightcurve(packets: np.ndarray, ts_starts, ts_widths, tmin: float, tmax: float, vmax: float,
                            nu_min: float = 1e14, nu_max: float = 5e15):
    """Returns dict(flux[MNUBINS, nts], lower_freq, delta_freq, lum[nts], lumcmf[nts])."""
    starts = np.asarray(ts_starts, dtype=np.float64)
    widths = np.asarray(ts_widths, dtype=np.float64)
    nts_all = len(starts)
    sel = (packets["type"] == TYPE_ESCAPE) & (packets["escape_type"] == TYPE_RPKT)
    p = packets[sel]
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
    np.add.at(lum, nts[ok_t], p["e_rf"][ok_t] / widths[nts[ok_t]])
    ok = ok_t & (p["nu_rf"] > nu_min) & (p["nu_rf"] < nu_max)
    nnu = np.clip(np.floor((np.log(p["nu_rf"][ok]) - np.log(nu_min)) / dlognu).astype(np.int64), 0, MNUBINS - 1)
    dE = p["e_rf"][ok] / widths[nts[ok]] / delta[nnu].astype(np.float64) / 4.e12 / np.pi / PARSEC / PARSEC
    np.add.at(flux, (nnu, nts[ok]), dE)
    inv_gamma = np.sqrt(1. - (vmax * vmax / CLIGHT**2))
    t_cmf = p["escape_time"].astype(np.float64) * inv_gamma
    ok_c = (t_cmf > tmin) & (t_cmf < tmax)
    ntc = timestep_index(t_cmf, starts, tmax)
    ok_c &= ntc >= 0
    np.add.at(lumcmf, ntc[ok_c], p["e_cmf"][ok_c] / widths[ntc[ok_c]] / inv_gamma)
    return dict(flux=flux, lower_freq=lower, delta_freq=delta, lum=lum, lumcmf=lumcmf, nescaped=int(sel.sum()))
