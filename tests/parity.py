"""Shared comparison helpers for the parity tests (engine or emulated kernel bodies vs the CPU oracle)."""
from __future__ import annotations

import numpy as np

from artis_amd import abi

# Reference-defined counters must match exactly; so must the step counters this project adds.
EXACT_STATS = list(range(abi.STAT_COUNT)) + [abi.STAT_X_RPKT_STEPS, abi.STAT_X_KPKT_STEPS, abi.STAT_X_LINES_VISITED,
                                            abi.STAT_X_MA_JUMPS,
                                            # virtual packets created / escaped after an r-packet, k-packet, macro-atom (VPKT_ON builds)
                                            *range(abi.STAT_X_VPKT_CREATED, abi.STAT_X_VPKT_CREATED + 4)]


STOKES_ABS_FLOOR = 1e-12  # above the largest absolute q / u difference ever measured against the oracle (1.6e-13)


def compare_packets(got: np.ndarray, want: np.ndarray, rtol: float, what: str = "") -> dict:
    """Integer fields (type, cell, line indices, emission/absorption types, scatter counts) and the RNG state
    must be identical; floating-point fields within rtol (0.0 = bit-exact), vectors relative to their length."""
    assert len(got) == len(want)
    rep = {}
    for f in abi.PACKET_INT_FIELDS:
        bad = np.nonzero(got[f] != want[f])[0]
        assert len(bad) == 0, f"{what}: integer field {f} differs for {len(bad)} packets, first {bad[:5]}: {got[f][bad[:5]]} vs {want[f][bad[:5]]}"
    assert np.array_equal(got["rngstate"], want["rngstate"]), f"{what}: RNG state differs (a different number of draws was made)"
    worst = 0.0
    for f in abi.PACKET_FLOAT_FIELDS:
        a = np.asarray(got[f], dtype=np.float64)
        b = np.asarray(want[f], dtype=np.float64)
        both_nan = np.isnan(a) & np.isnan(b)
        if rtol == 0.0:
            same = (a == b) | both_nan
            assert same.all(), f"{what}: float field {f} not bit-identical for {np.count_nonzero(~same)} values"
        else:
            if a.ndim == 2:  # 3-vectors (pos, dir, em_pos, trueem_pos): error relative to the length of the vector,
                # not to each component (a component that happens to be ~0 carries the absolute rounding of the others)
                denom = np.broadcast_to(np.sqrt(np.nansum(b * b, axis=1))[:, None], a.shape).copy()
            else:
                denom = np.maximum(np.abs(a), np.abs(b))
            denom[~(denom > 0)] = 1.0
            rel = np.abs(a - b) / denom
            rel[both_nan] = 0.0
            if f in ("stokes_q", "stokes_u"):
                # q, u are differences of O(1) terms of the normalised Stokes vector (vectors.h:266-370), so a value near zero carries the
                # ABSOLUTE rounding of those terms: every value has to meet the relative bar OR lie within STOKES_ABS_FLOOR of the
                # oracle's. (Round 4, 2e6 packets: one packet with q = -3.2e-7 differed by 3.7e-16 = 1.2e-9 of |q|; the largest absolute
                # difference of all packets was 1.6e-13: profiles/r04/stress_parity_2e6.txt. Round 4 measured q, u against |(1, q, u)|
                # instead, which turned the bar into an absolute 1e-9 for every packet)
                rel[np.abs(a - b) <= STOKES_ABS_FLOOR] = 0.0
            assert not np.isnan(rel).any(), f"{what}: NaN mismatch in {f}"
            worst = max(worst, float(rel.max()))
            assert rel.max() <= rtol, f"{what}: float field {f} rel diff {rel.max():.3e} > {rtol}"
    rep["worst_rel"] = worst
    return rep


def compare_estimators(got: abi.Estimators, want: abi.Estimators, rtol: float, what: str = "") -> None:
    """Estimators are sums of many terms; the GPU adds them with atomics in arbitrary order, so they are compared to a
    relative tolerance scaled by the largest entry of each array (float sums are not associative)."""
    for k, a in got.arrays().items():
        b = want.arrays()[k]
        scale = max(np.abs(b).max(), 1e-300)
        err = np.abs(a - b).max() / scale
        assert err <= rtol, f"{what}: estimator {k} differs by {err:.3e} (rel. to max) > {rtol}"


UPDATECELL = abi.STAT_NAMES.index("UPDATECELL")
UPSCATTER = abi.STAT_NAMES.index("UPSCATTER")
DOWNSCATTER = abi.STAT_NAMES.index("DOWNSCATTER")


def compare_stats(got: abi.Estimators, want: abi.Estimators, what: str = "", same_libm: bool = True) -> None:
    """Event counters (stats.h:13 of the reference, plus this project's step counters) must be identical.
    UPDATECELL is skipped: the engine fills the cache of every cell up front, the oracle on first use.
    With different math libraries (GPU vs glibc) UPSCATTER and DOWNSCATTER are compared as a sum: the
    reference classifies a line scattering by `oldnucmf < nu_cmf` (macroatom.cc:232), and for a resonance
    scattering the two frequencies are equal to the last bit or two, so the split is not a property of the
    algorithm but of the exp/log rounding of the platform."""
    for i in EXACT_STATS:
        if i == UPDATECELL:
            continue
        if not same_libm and i in (UPSCATTER, DOWNSCATTER):
            continue
        assert got.stats[i] == want.stats[i], f"{what}: event counter {abi.STAT_NAMES[i]}: {got.stats[i]} vs {want.stats[i]}"
    assert got.stats[UPSCATTER] + got.stats[DOWNSCATTER] == want.stats[UPSCATTER] + want.stats[DOWNSCATTER], what


def _oracle_slice(args):
    lo, hi = args
    from oracle import oracle_py

    model, cs, ts, pk, preset = _oracle_slice.shared
    sub = pk[lo:hi].copy()
    est = abi.estimators_for(model, preset)
    oracle_py.update_packets(model, cs, ts, sub, est, preset=preset)
    return sub, est.arrays(), np.array(est.stats)


def oracle_parallel(model, cs, ts, pk, est: abi.Estimators, preset: str = "classic", nproc: int = 0) -> None:
    """The CPU oracle on slices of the population in forked worker processes (packets are independent: per-packet
    generator, per-packet opacity cache), results written back into pk and added into est. Event counters add exactly;
    estimators are float sums whose order differs from a single call, which the comparison tolerance covers.
    Call before anything in the process has touched the GPU (fork)."""
    import multiprocessing as mp
    import os

    from oracle import oracle_py

    oracle_py.lib(preset)  # built (if its sources are newer) and loaded once here, not by every forked worker at once
    nproc = nproc or min(os.cpu_count() or 1, 16)
    n = len(pk)
    bounds = [(n * i // nproc, n * (i + 1) // nproc) for i in range(nproc)]
    _oracle_slice.shared = (model, cs, ts, pk, preset)
    with mp.get_context("fork").Pool(nproc) as pool:
        res = pool.map(_oracle_slice, bounds)
    for (lo, hi), (sub, arrs, stats) in zip(bounds, res):
        pk[lo:hi] = sub
        for k, a in est.arrays().items():
            a += arrs[k]
        est.stats[:] = np.asarray(est.stats) + stats
