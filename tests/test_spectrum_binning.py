"""Emergent spectrum / light curve of the escaped packets (tools/exspec.py: the reference's binning rules of
spectrum_lightcurve.cc restated) -- the artefact north_star's acceptance is phrased in. No reference run of this model
exists in this environment (the reference is not buildable here), so the tests pin what can be pinned: the binning
conserves the escaped energy, and engine / kernel bodies / oracle give the same spectrum for the same input."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import exspec  # noqa: E402
import hostemu_binding as emu  # noqa: E402
from artis_amd import abi, synth  # noqa: E402


def _timegrid(model, ts):
    tmin, tmax = 0.8 * ts.c.start, 1.3 * ts.c.start
    edges = np.geomspace(tmin, tmax, 21)
    return edges[:-1], np.diff(edges), tmin, tmax


def _case(npk=20000):
    model, cs, ts, aux = synth.build("small", ncoord=8)
    pk0 = synth.make_packets(model, aux, npk, kpkt_fraction=0.2)
    return model, cs, ts, pk0


def test_binning_conserves_escaped_energy_and_matches_between_restatements(oracle):
    model, cs, ts, pk0 = _case()
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    pa, pb = pk0.copy(), pk0.copy()
    oracle.update_packets(model, cs, ts, pa, abi.Estimators(n, g))
    emu.update_packets(model, cs, ts, pb, abi.Estimators(n, g))
    starts, widths, tmin, tmax = _timegrid(model, ts)
    sa = exspec.spectrum_and_lightcurve(pa, starts, widths, tmin, tmax, model["vmax"])
    sb = exspec.spectrum_and_lightcurve(pb, starts, widths, tmin, tmax, model["vmax"])
    assert sa["nescaped"] > 100
    for k in ("flux", "lum", "lumcmf"):
        assert np.array_equal(sa[k], sb[k]), k  # bit-identical packets -> identical spectra
    esc = (pa["type"] == abi.TYPE_ESCAPE) & (pa["escape_type"] == abi.TYPE_RPKT)
    e_all = pa["e_rf"][esc].sum()
    # every escaped packet arrives inside the time grid here, so the light curve integrates to the escaped energy ...
    assert abs((sa["lum"] * widths).sum() - e_all) <= 1e-12 * e_all
    # ... and the spectrum to the energy of the packets inside the frequency window
    inwin = esc & (pa["nu_rf"] > 1e14) & (pa["nu_rf"] < 5e15)
    e_spec = (sa["flux"] * sa["delta_freq"][:, None].astype(np.float64) * widths[None, :]).sum() * 4.e12 * np.pi * exspec.PARSEC**2
    assert abs(e_spec - pa["e_rf"][inwin].sum()) <= 1e-9 * e_all
    # arrival-time rule: t_arrive = escape_time - pos.dir/c: never later than the escape time by more than the light
    # crossing time of the grid, and earlier for most packets (they leave a cube face heading outwards)
    t_arr = pa["escape_time"][esc] - (pa["pos"][esc] * pa["dir"][esc]).sum(axis=1) / exspec.CLIGHT
    assert np.all(t_arr > tmin) and np.mean(t_arr < pa["escape_time"][esc]) > 0.9


def test_timestep_index_rule():
    starts = np.array([1., 2., 4.])
    assert list(exspec.timestep_index(np.array([0.5, 1., 1.99, 2., 3.9, 7.99, 8.]), starts, 8.)) == [-1, 0, 0, 1, 1, 2, -1]


@pytest.mark.gpu
def test_engine_spectrum_matches_oracle_spectrum(oracle):
    import torch

    assert torch.cuda.is_available()
    from artis_amd import engine

    model, cs, ts, pk0 = _case(60000)
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    pa, pb = pk0.copy(), pk0.copy()
    oracle.update_packets(model, cs, ts, pa, abi.Estimators(n, g))
    eng = engine.Engine(model)
    eng.set_cellstate(cs, ts)
    eng.update_packets(pb, abi.Estimators(n, g))
    eng.close()
    starts, widths, tmin, tmax = _timegrid(model, ts)
    sa = exspec.spectrum_and_lightcurve(pa, starts, widths, tmin, tmax, model["vmax"])
    sb = exspec.spectrum_and_lightcurve(pb, starts, widths, tmin, tmax, model["vmax"])
    assert sa["nescaped"] == sb["nescaped"] > 300
    assert np.allclose(sa["lum"], sb["lum"], rtol=1e-9, atol=0) and np.allclose(sa["lumcmf"], sb["lumcmf"], rtol=1e-9, atol=0)
    # frequencies agree to ~1e-12, so a packet can change bin only if it sits on an edge: compare 20-bin groups
    ca = sa["flux"].reshape(50, 20, -1).sum(axis=1)
    cb = sb["flux"].reshape(50, 20, -1).sum(axis=1)
    assert np.abs(ca - cb).max() <= 1e-6 * ca.max()


def test_direction_resolved_spectra_add_up_to_the_angle_average(oracle):
    """Direction-resolved spectra and light curves (add_to_spec_res / add_to_lc_res with dirbin >= 0, spectrum_lightcurve.cc:545,
    :562, :689-691): every escaped packet falls in exactly one of the MABINS direction bins and counts MABINS-fold there, so the
    mean over the bins is the angle-averaged result."""
    model, cs, ts, pk0 = _case()
    n, g = model["npts_nonempty"], model["nbfcontinua_ground"]
    pa = pk0.copy()
    oracle.update_packets(model, cs, ts, pa, abi.Estimators(n, g))
    starts, widths, tmin, tmax = _timegrid(model, ts)
    avg = exspec.spectrum_and_lightcurve(pa, starts, widths, tmin, tmax, model["vmax"])
    acc = {k: np.zeros_like(avg[k]) for k in ("flux", "lum", "lumcmf")}
    nesc = 0
    for b in range(exspec.MABINS):
        r = exspec.spectrum_and_lightcurve(pa, starts, widths, tmin, tmax, model["vmax"], dirbin=b)
        nesc += r["nescaped"]
        for k in acc:
            acc[k] += r[k] / exspec.MABINS
    assert nesc == avg["nescaped"]
    for k in acc:
        assert np.allclose(acc[k], avg[k], rtol=1e-12, atol=0), k
