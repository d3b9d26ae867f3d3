"""Pin the CPU oracle against what the reference itself holds for this path.

 1. tests/golden/rng_reference.json: generator streams produced by the REFERENCE's
    own random.h, compiled where it lies (tests/golden/make_rng_golden.py).
 2. The known-answer / property checks of the reference's unittests.cc that touch the
    packet path, restated on the oracle's functions with the same seeds and tolerances:
    test_vector_geometry (unittests.cc:118), test_frame_transform (:199),
    test_random_sampling (:225), test_rad_deexcitation (:356),
    test_phixs_table_lookup (:382, classic branch), test_closest_transition_randomised (:435)
    and the static_asserts next to closest_transition / get_linedistance (rpkt.h:147-186).
The reference's end-to-end md5 fixtures (tests/*_inputfiles/results_md5_*.txt) need the
downloaded atomic data release and an MPI build; they cannot be reproduced here.
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from artis_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
CLIGHT = 2.99792458e10
H = 6.6260755e-27
DAY = 86400.0
EV = 1.6021772e-12
PI = math.pi


class Rng:
    def __init__(self, L, seed):
        self.L = L
        self.s = (C.c_uint32 * 4)()
        L.artis_oracle_rng_seed(self.s, C.c_uint32(seed))

    def uniform(self):
        return float(self.L.artis_oracle_rng_uniform(self.s))

    def raw(self):
        return int(self.L.artis_oracle_rng_next(self.s))

    def isotropic(self):
        out = (C.c_double * 3)()
        self.L.artis_oracle_rand_isotropic_unitvec(self.s, out)
        return np.array(out[:])


def _vec(a):
    return (C.c_double * 3)(*[float(x) for x in a])


def angle_ab(L, d, v):
    out = (C.c_double * 3)()
    L.artis_oracle_angle_ab(_vec(d), _vec(v), out)
    return np.array(out[:])


def test_packet_struct_size(oracle):
    assert oracle.lib().artis_oracle_sizeof_packet() == abi.PACKET_DTYPE.itemsize


def test_rng_streams_match_reference_random_h(oracle):
    L = oracle.lib()
    with open(os.path.join(HERE, "golden", "rng_reference.json")) as f:
        gold = json.load(f)
    for seed, st in gold["streams"].items():
        r = Rng(L, int(seed))
        assert [r.raw() for _ in range(gold["count"])] == st["raw_u32"], f"raw stream differs for seed {seed}"
        r = Rng(L, int(seed))
        s = (C.c_uint32 * 4)(*r.s)
        out = np.zeros(gold["count"], dtype=np.float32)
        L.artis_oracle_rng_fill_uniform(s, C.c_int64(gold["count"]), out.ctypes.data_as(C.c_void_p))
        assert out.view(np.uint32).tolist() == st["uniform_float_bits"], f"rng_uniform differs for seed {seed}"


def test_numpy_seeding_matches_oracle(oracle):
    pk = np.zeros(1000, dtype=abi.PACKET_DTYPE)
    abi.seed_packet_rng(pk, 4294967000)  # wraps through 2^32 like the reference's uint32 arithmetic
    pk2 = pk.copy()
    oracle.lib().artis_oracle_seed_packets(abi.packets_ptr(pk2), len(pk2), C.c_uint32(4294967000))
    assert np.array_equal(pk["rngstate"], pk2["rngstate"])


def test_vector_geometry_unittests_cc_118(oracle):
    L = oracle.lib()
    L.artis_oracle_doppler.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
    r = Rng(L, 81102)
    # first block of the reference test consumes 100 x 2 isotropic vectors
    for _ in range(100):
        a = r.isotropic()
        b = r.isotropic() * 2.5
        c = np.cross(a, b)
        assert abs(c @ a) < 1e-12 and abs(c @ b) < 1e-12
    for _ in range(100):
        d1 = r.isotropic()
        vel = r.isotropic() * (r.uniform() * 0.3 * CLIGHT)
        d2 = angle_ab(L, d1, vel)
        back = angle_ab(L, d2, -vel)
        assert np.all(np.abs(back - d1) < 1e-12)
    pos = np.array([1.1e14, -2.4e14, 0.8e14])
    dirv = np.array([0.3, -0.1, 0.9])
    dirv = dirv / np.sqrt((dirv**2).sum())
    t = 10.0 * DAY
    expected = 1.0 - (dirv @ (pos / t)) / CLIGHT
    got = L.artis_oracle_doppler(_vec(pos), _vec(dirv), C.c_double(t))
    assert abs(got - expected) <= 1e-14 * max(abs(got), abs(expected))
    # move_pkt_withtime preserves e_cmf/nu_cmf = e_rf/nu_rf
    p = _vec(dirv * 2.0e14)
    tcur = C.c_double(10.0 * DAY)
    nu_rf, e_rf = 3e15, 4e-12
    dop = L.artis_oracle_doppler(p, _vec(dirv), tcur)
    nu_cmf = C.c_double(nu_rf * dop)
    e_cmf = C.c_double(e_rf * dop)
    L.artis_oracle_move_pkt_withtime(p, _vec(dirv), C.byref(tcur), C.c_double(nu_rf), C.byref(nu_cmf), C.c_double(e_rf),
                                     C.byref(e_cmf), C.c_double(3.0e13))
    a, b = e_cmf.value / nu_cmf.value, e_rf / nu_rf
    assert abs(a - b) <= 1e-12 * max(abs(a), abs(b))


def test_frame_transform_unittests_cc_199(oracle):
    L = oracle.lib()
    r = Rng(L, 99001)

    def ft(n, q, u, v):
        out = (C.c_double * 3)()
        qq, uu = C.c_double(), C.c_double()
        L.artis_oracle_frame_transform(_vec(n), C.c_double(q), C.c_double(u), _vec(v), out, C.byref(qq), C.byref(uu))
        return np.array(out[:]), qq.value, uu.value

    for _ in range(100):
        n_rf = r.isotropic()
        pol_p = r.uniform() * 0.9
        ang = r.uniform() * 2.0 * PI
        q0, u0 = pol_p * math.cos(ang), pol_p * math.sin(ang)
        vel = r.isotropic() * (r.uniform() * 0.2 * CLIGHT)
        n_cmf, q_cmf, u_cmf = ft(n_rf, q0, u0, vel)
        assert abs(math.sqrt(q_cmf**2 + u_cmf**2) - pol_p) < 1e-10
        n2, q2, u2 = ft(n_cmf, q_cmf, u_cmf, -vel)
        assert np.all(np.abs(n2 - n_rf) < 1e-10)
        assert abs(q2 - q0) < 1e-8 and abs(u2 - u0) < 1e-8


def test_random_sampling_unittests_cc_225(oracle):
    L = oracle.lib()
    n = 1_000_000
    r = Rng(L, 31415)
    z = np.zeros(n, dtype=np.float32)
    L.artis_oracle_rng_fill_uniform(r.s, C.c_int64(n), z.ctypes.data_as(C.c_void_p))
    assert z.min() >= 0.0 and z.max() < 1.0
    assert abs(z.astype(np.float64).mean() - 0.5) < 6.0 / math.sqrt(12.0 * n)
    d = np.zeros(3 * n)
    L.artis_oracle_fill_isotropic(r.s, C.c_int64(n), d.ctypes.data_as(C.c_void_p))
    d = d.reshape(n, 3)
    assert np.all(np.abs(np.sqrt((d**2).sum(axis=1)) - 1.0) < 1e-6)
    assert abs(d[:, 2].mean()) < 6.0 / math.sqrt(3.0 * n)
    assert abs((d[:, 2] ** 2).mean() - 1.0 / 3.0) < 1e-3


def test_rad_deexcitation_unittests_cc_356(oracle):
    L = oracle.lib()
    eps_trans, A = 2.0 * EV, 1e7
    gu, gl, t = 3.0, 1.0, 20.0 * DAY
    assert L.artis_oracle_rad_deexcitation_ratecoeff(eps_trans, A, gu, gl, 0.0, 0.0, t) == np.float32(A)
    nnu, nnl = 1e5, 1e8
    nu_trans = eps_trans / H
    b_ul = CLIGHT**2 / (2 * H) / nu_trans**3 * float(np.float32(A))
    b_lu = gu / gl * b_ul
    tau = ((b_lu * nnl) - (b_ul * nnu)) * (H * CLIGHT / (4 * PI)) * t
    assert tau > 1.0
    got = L.artis_oracle_rad_deexcitation_ratecoeff(eps_trans, A, gu, gl, nnu, nnl, t)
    want = float(np.float32(A)) * (-math.expm1(-tau)) / tau
    assert abs(got - want) <= 1e-12 * max(abs(got), abs(want))


def test_phixs_table_lookup_classic_unittests_cc_382(oracle):
    L = oracle.lib()
    npts, inc = 10, 0.1
    xs = np.array([(i + 1) * 1e-18 for i in range(npts)], dtype=np.float32)
    ptr = xs.ctypes.data_as(C.c_void_p)
    nu_edge = 3e15

    def f(nu):
        return np.float32(L.artis_oracle_phixs_fromtable(ptr, npts, inc, nu_edge, nu))

    assert f(nu_edge * 0.99) == 0.0
    assert f(nu_edge) == xs[0]
    assert f(nu_edge * 1.25) == xs[2]
    nu_out = nu_edge * (1 + inc * npts)
    nu = nu_out * (1.0 - 1e-13)
    while nu < nu_out:
        assert f(nu) in xs
        nu = np.nextafter(nu, nu_out)
    above = 4.0 * nu_edge
    want = xs[npts - 1] * (nu_out / above) ** 3
    got = f(above)
    assert abs(got - want) <= 1e-6 * max(abs(got), abs(want))


def test_closest_transition_randomised_unittests_cc_435(oracle):
    L = oracle.lib()
    r = Rng(L, 777)
    nu = np.array([1e14 + r.uniform() * 1e15 for _ in range(500)])
    nu = np.sort(nu)[::-1].copy()
    ptr = nu.ctypes.data_as(C.c_void_p)
    for _ in range(1000):
        nu_cmf = 0.5e14 + r.uniform() * 1.2e15
        expected = -1
        idx = np.nonzero(nu <= nu_cmf)[0]
        if len(idx):
            expected = int(idx[0])
        assert L.artis_oracle_closest_transition(ptr, 500, nu_cmf, -1) == expected


def test_closest_transition_and_linedistance_static_asserts_rpkt_h(oracle):
    L = oracle.lib()
    lst = np.array([9.0, 7.0, 5.0, 3.0])
    p = lst.ctypes.data_as(C.c_void_p)
    assert L.artis_oracle_closest_transition(p, 4, 10.0, -1) == 0
    assert L.artis_oracle_closest_transition(p, 4, 8.0, -1) == 1
    assert L.artis_oracle_closest_transition(p, 4, 5.0, -1) == 2
    assert L.artis_oracle_closest_transition(p, 4, 2.0, -1) == -1
    assert L.artis_oracle_closest_transition(p, 4, 8.0, 2) == 2
    assert L.artis_oracle_closest_transition(p, 4, 8.0, 4) == -1
    assert L.artis_oracle_get_linedistance(100.0, 1.0, 2.0) == 0.0
    assert L.artis_oracle_get_linedistance(2.0, 4.0, 2.0) == CLIGHT * 2.0 * 2.0 / 2.0


def test_planck_matches_closed_form(oracle):
    L = oracle.lib()
    for nu, T in [(1e14, 5000.0), (8e14, 12000.0), (4e15, 3500.0)]:
        want = 2 * H * nu**3 / CLIGHT**2 / math.expm1(H / 1.38064852e-16 * nu / T)
        assert abs(L.artis_oracle_planck(nu, T) - want) <= 1e-14 * want


def test_gauss_kronrod_matches_reference_golden(oracle):
    """The oracle's restatement of gauss_kronrod_integrate<31> (gausskronrod.h:244: Boost tables, summation order,
    adaptive bisection with inherited tolerances) against results of the reference's own header, bit for bit.
    Golden file: tests/golden/gk31_reference.json (tests/golden/make_gk31_golden.py)."""
    import ctypes as C
    L = oracle.lib()
    L.artis_oracle_gk31_test.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
    L.artis_oracle_gk31_test.restype = C.c_double
    with open(os.path.join(os.path.dirname(__file__), "golden", "gk31_reference.json")) as f:
        gold = json.load(f)
    assert len(gold["cases"]) >= 20
    ndeep = 0
    for case in gold["cases"]:
        err = C.c_double(0.)
        res = L.artis_oracle_gk31_test(case["mode"], case["p0"], case["p1"], case["a"], case["b"], case["tol"], C.byref(err))
        assert float(res).hex() == float.fromhex(case["result"]).hex(), case
        if case["a"] != case["b"]:
            assert float(err.value).hex() == float.fromhex(case["error"]).hex(), case
        ndeep += 1
    assert ndeep == len(gold["cases"])


def test_rad_deexcitation_matches_reference_golden(oracle):
    """rad_deexcitation_ratecoeff() of the oracle against the reference's macroatom.h:61 compiled in place, bit for bit
    (tests/golden/macroatom_reference.json, tests/golden/make_macroatom_golden.py)."""
    L = oracle.lib()
    with open(os.path.join(os.path.dirname(__file__), "golden", "macroatom_reference.json")) as f:
        gold = json.load(f)
    assert len(gold["cases"]) >= 200
    kinds = set()
    for c in gold["cases"]:
        got = L.artis_oracle_rad_deexcitation_ratecoeff(c["epsilon_trans"], c["A_ul"], c["g_upper"], c["g_lower"], c["nn_upper"],
                                                        c["nn_lower"], c["t_current"])
        want = float.fromhex(c["result"])
        assert float(got).hex() == want.hex(), c
        kinds.add("thin" if want == float(np.float32(c["A_ul"])) else "escape")
    assert kinds == {"thin", "escape"}


@pytest.mark.parametrize("gridtype,ncoord", [(abi.GRID_SPHERICAL1D, 16), (abi.GRID_CYLINDRICAL2D, 8), (abi.GRID_CARTESIAN3D, 8)])
def test_packets_stay_inside_their_cells(oracle, gridtype, ncoord):
    """Geometric invariant of boundary_distance()/change_cell_or_escape() (grid.cc:2480, grid.h:118) on every grid type:
    after a timestep each packet that has not escaped lies inside the homologously expanded bounds of the cell it says
    it is in (to the reference's boundary tolerance, grid.cc cellbound_tolerance = 1e-7 relative), escaped packets
    left through the outer surface, and energy only changes through the Doppler factor (e_rf stays positive)."""
    from artis_amd import synth
    model, cs, ts, aux = synth.build("tiny", ncoord=ncoord, gridtype=gridtype)
    pk = synth.make_packets(model, aux, 3000, kpkt_fraction=0.1)
    est = abi.Estimators(model["npts_nonempty"], model["nbfcontinua_ground"])
    oracle.update_packets(model, cs, ts, pk, est)
    d = model.d
    tmin, rmax = d["tmin"], d["rmax"]
    alive = pk["type"] != abi.TYPE_ESCAPE
    assert alive.sum() > 1000 and (~alive).sum() > 10
    pos = pk["pos"][alive] * (tmin / pk["prop_time"][alive])[:, None]   # scaled back to tmin
    ci = pk["cellindex"][alive]
    if gridtype == abi.GRID_SPHERICAL1D:
        coords = [np.sqrt((pos ** 2).sum(axis=1))]
    elif gridtype == abi.GRID_CYLINDRICAL2D:
        coords = [np.sqrt(pos[:, 0] ** 2 + pos[:, 1] ** 2), pos[:, 2]]
    else:
        coords = [pos[:, 0], pos[:, 1], pos[:, 2]]
    stride = 1
    for axis, x in enumerate(coords):
        n = int(d["ncoordgrid"][axis])
        lo_edges = np.asarray(d["coord_pos_min_tmin"][axis], dtype=np.float64)
        hi_edges = np.concatenate([lo_edges[1:], [rmax]])
        idx = (ci // stride) % n
        stride *= n
        tol = 1e-6 * rmax
        assert np.all(x >= lo_edges[idx] - tol), (axis, float((lo_edges[idx] - x).max()))
        assert np.all(x <= hi_edges[idx] + tol), (axis, float((x - hi_edges[idx]).max()))
    esc = pk[~alive]
    r_esc = np.sqrt((esc["pos"] ** 2).sum(axis=1)) * (tmin / esc["prop_time"])
    assert np.all(r_esc >= 0.7 * rmax)   # left through the outer surface (a face of the cube / cylinder / the sphere)
    assert np.all(pk["e_rf"] > 0) and np.all(np.isfinite(pk["pos"]))


def test_physical_constants_match_reference_constants_h(oracle):
    """The constants restated in oracle/artis_oracle.c and in artis_amd/csrc/physics.h against the reference's constants.h
    compiled in place (tests/golden/packet_layout_reference.json), bit for bit."""
    import hostemu_binding
    with open(os.path.join(os.path.dirname(__file__), "golden", "packet_layout_reference.json")) as f:
        gold = json.load(f)["constants"]
    for L, fn in ((oracle.lib(), "artis_oracle_constants"), (hostemu_binding.lib(), "artis_emu_constants")):
        names = (C.c_char_p * 64)()
        vals = (C.c_double * 64)()
        f = getattr(L, fn)
        f.restype = C.c_int
        n = f(names, vals, 64)
        assert n == len(gold) == 19
        for i in range(n):
            name = names[i].decode()
            assert float(vals[i]).hex() == float.fromhex(gold[name]).hex(), (fn, name)


def _compton_functions(oracle):
    """(name, sigma_compton_partial, choose_f, meanf_sigma) of the oracle and of the kernel bodies (physics.h on x86)"""
    import ctypes as C

    import hostemu_binding

    out = []
    for name, L, prefix in (("oracle", oracle.lib(), "artis_oracle_"), ("kernel bodies", hostemu_binding.lib(), "artis_emu_")):
        fs = []
        for fn, nargs in (("sigma_compton_partial", 2), ("choose_f", 2), ("meanf_sigma", 1)):
            f = getattr(L, prefix + fn)
            f.restype = C.c_double
            f.argtypes = [C.c_double] * nargs
            fs.append(f)
        out.append((name, *fs))
    return out


def test_compton_cross_sections_unittests_cc_323(oracle):
    """unittests.cc:323 test_compton, on the oracle's and the kernels' restatements of gammapkt.h:28-90: the partial
    Compton cross section over the full energy-loss range is the Klein-Nishina total; choose_f() inverts it to the
    solver tolerance; meanf_sigma() is continuous across the Taylor-series / closed-form crossover."""
    SIGMA_T, THOMSON_LIMIT = 6.6524e-25, 1e-2

    def kn_total(x):
        return 0.75 * SIGMA_T * ((((1. + x) / x**3) * (((2. * x * (1. + x)) / (1. + (2. * x))) - np.log1p(2. * x))) +
                                 (np.log1p(2. * x) / (2. * x)) - ((1. + (3. * x)) / (1. + (2. * x))**2))

    for name, partial, choose_f, meanf in _compton_functions(oracle):
        for x in (0.05, 0.2, 1., 5.):
            assert abs(partial(x, 1. + 2. * x) - kn_total(x)) <= 1e-10 * kn_total(x), (name, x)
        for x in (0.05, 1., 5.):
            for z in (0.1, 0.5, 0.9):
                f = choose_f(x, z)
                assert abs(partial(x, f) / partial(x, 1. + 2. * x) - z) < 2e-4, (name, x, z)
        a, b = meanf(THOMSON_LIMIT * (1. - 1e-6)), meanf(THOMSON_LIMIT * (1. + 1e-6))
        assert abs(a - b) <= 1e-5 * max(abs(a), abs(b)), name
    # and the two restatements agree bit for bit with each other
    (_, p0, c0, m0), (_, p1, c1, m1) = _compton_functions(oracle)
    for x in np.geomspace(1e-3, 30., 40):
        assert p0(x, 1. + 2. * x) == p1(x, 1. + 2. * x) and m0(x) == m1(x)
        for z in (0.03, 0.4, 0.97):
            assert c0(x, z) == c1(x, z)


def test_planck_function_integrals_unittests_cc_254(oracle):
    """unittests.cc:254 test_planck checks radfield::calculate_planck_integral (a host-side routine outside the packet
    path) against the Stefan-Boltzmann law and the mean frequency 4 zeta(5)/zeta(4) kT/h. The packet path only has the
    Planck function itself (radfield.h:50 dbb(), used by sample_planck_montecarlo kpkt.cc:266 and radfield()): the same
    two laws are checked on it by numerical quadrature, for the oracle and for the kernel bodies."""
    import ctypes as C

    import hostemu_binding

    H, KB, CLIGHT, STEBO = 6.6260755e-27, 1.38064852e-16, 2.99792458e10, 5.670400e-5
    T = 6000.
    x = np.geomspace(1e-6, 80., 400001)  # h nu / k T
    nu = x * KB * T / H
    for name, L, fn in (("oracle", oracle.lib(), "artis_oracle_planck"), ("kernel bodies", hostemu_binding.lib(), "artis_emu_planck")):
        f = getattr(L, fn)
        f.restype = C.c_double
        f.argtypes = [C.c_double, C.c_double]
        B = np.array([f(v, T) for v in nu[::40]])
        n = nu[::40]
        total = np.trapezoid(B, n)
        assert abs(total - STEBO * T**4 / np.pi) <= 1e-3 * total, name
        nubar = np.trapezoid(B * n, n) / total
        assert abs(nubar - KB * T / H * 4. * 1.03692775514337 / 1.08232323371114) <= 1e-4 * nubar, name
        # Wien peak of B_nu: x = 2.821439...
        assert abs(n[np.argmax(B)] * H / (KB * T) - 2.8214393721) < 2e-3, name


@pytest.mark.parametrize("preset", ["classic", "kilonova_lte", "nltenebular", "christinenonthermal", "nltephotospheric",
                                    "nltewithoutnonthermal", *abi.CI_PRESETS])  # ci_*: the option sets of the reference's CI scripts
def test_options_presets_match_reference_option_files(preset, tmp_path):
    """include/artis_options.h against the reference's own artisoptions_<preset>.h, all six files: every compile-time
    option the packet path reads (39 of them: grids of the rate-coefficient tables, frequency limits, scattering and polarisation switches,
    estimator and radiation-field model switches, non-thermal switches, thermalisation schemes with the reference's enum
    numbering, expansion opacities ...) has the value the reference's file gives it. Golden file:
    tests/golden/options_reference.json, printed by the reference's headers compiled where they lie
    (tests/golden/make_options_golden.py)."""
    import subprocess
    root = os.path.dirname(HERE)
    with open(os.path.join(HERE, "golden", "options_reference.json")) as f:
        want = json.load(f)["presets"][preset]
    exe = str(tmp_path / "optprint")
    flags = [] if preset == "classic" else [f"-DARTIS_PRESET_{preset.upper()}"]
    subprocess.check_call(["gcc", *flags, "-o", exe, os.path.join(HERE, "options_printer.c")], cwd=root)
    got = dict(line.split() for line in subprocess.check_output([exe], text=True).strip().splitlines())
    assert len(want) >= 41 and set(got) == set(want)
    for name, val in want.items():
        assert got[name] == val, f"{preset}: {name} = {got[name]}, reference {val}"


def test_expansion_opacity_bin_helpers_static_asserts(oracle):
    """The compile-time checks the reference keeps beside its bin helpers, restated on oracle and kernel bodies (the
    expansion-opacity builds): get_linearbinindex floors, is left-closed and signed (sn3d.h:124-128); the 20 A wavelength
    bins are contiguous and ordered (rpkt.h:42-44); there are (40000 - 60) / 20 of them (rpkt.h:26)."""
    import ctypes as C

    import hostemu_binding

    P = "kilonova_expopac"
    for L, pre in ((oracle.lib(P), "artis_oracle_"), (hostemu_binding.lib(P), "artis_emu_")):
        idx = getattr(L, pre + "linearbinindex")
        idx.restype, idx.argtypes = C.c_longlong, [C.c_double] * 3
        nu = getattr(L, pre + "expopac_bin_nu")
        nu.restype, nu.argtypes = C.c_double, [C.c_longlong, C.c_int]
        assert idx(1.5, 1., 1.) == 0 and idx(3., 1., 1.) == 2 and idx(1., 1., 1.) == 0
        assert idx(0.5, 1., 1.) == -1 and idx(-5., 1., 2.) == -3
        assert nu(0, 0) == nu(1, 1) and nu(0, 0) < nu(0, 1)
        assert nu(abi.EXPOPAC_NBINS - 1, 1) > nu(abi.EXPOPAC_NBINS - 1, 0)
        assert nu(0, 1) == 1e8 * 2.99792458e10 / 60. and abs(nu(abi.EXPOPAC_NBINS - 1, 0) / (1e8 * 2.99792458e10 / 40000.) - 1) < 1e-15
        # a packet at 5000 A sits in the bin whose edges bracket it
        b = idx(5000., 60., 20.)
        assert nu(b, 0) < 1e8 * 2.99792458e10 / 5000. <= nu(b, 1)
    assert abi.EXPOPAC_NBINS == int((40000. - 60.) / 20.)


def test_binindex_helpers_unittests_cc_68(oracle):
    """unittests.cc:68 test_binindex_helpers on the oracle's get_logbinindex / get_loggrid_edge (sn3d.h:134, :142) and on the
    kernel bodies' logbinindex (physics.h add_to_vspecpol, vpkt.cc:124-125) with the host's loggrid_edge (model_build.h
    make_vpkt_config, vpkt.cc:500-501): increasing edges, the geometric midpoint of every bin maps back to it, both clamps."""
    import hostemu_binding

    P = "ci_classic_vpkt"
    minvalue, dlog, nbins = 1e14, 0.05, 100
    for L, pre in ((oracle.lib(P), "artis_oracle_"), (hostemu_binding.lib(P), "artis_emu_")):
        idx = getattr(L, pre + "logbinindex")
        idx.restype, idx.argtypes = C.c_longlong, [C.c_double, C.c_double, C.c_double, C.c_longlong]
        edge = getattr(L, pre + "loggrid_edge")
        edge.restype, edge.argtypes = C.c_double, [C.c_double] * 3
        for i in range(nbins):
            lo, hi = edge(minvalue, dlog, float(i)), edge(minvalue, dlog, float(i + 1))
            assert hi > lo, "get_loggrid_edge produces increasing bin edges"
            assert idx(math.sqrt(lo * hi), minvalue, dlog, nbins) == i, "get_logbinindex returns the bin containing its geometric midpoint"
        assert idx(minvalue / 10., minvalue, dlog, nbins) == 0, "clamps below-range to bin 0"
        assert idx(minvalue * 1e10, minvalue, dlog, nbins) == nbins - 1, "clamps above-range to the last bin"
    # the two restatements give the same edges bit for bit (both are exp(log(min) + i * dlog) through glibc)
    ea = [oracle.lib(P).artis_oracle_loggrid_edge(minvalue, dlog, float(i)) for i in range(nbins + 1)]
    eb = [hostemu_binding.lib(P).artis_emu_loggrid_edge(minvalue, dlog, float(i)) for i in range(nbins + 1)]
    assert ea == eb


def test_range_chunks_unittests_cc_91(oracle):
    """unittests.cc:91 test_range_chunks on artis_amd.dist.packet_shard (get_range_chunk mpi_logging.h:158: how the packets of one
    population are shared out over the ranks, bench.py --packets-total): 200 random (size, nchunks) drawn with the reference's
    generator and seed -- contiguous, complete, chunk sizes within one of each other; and the static_asserts of mpi_logging.h:175-177."""
    from artis_amd import dist as adist

    rng = Rng(oracle.lib(), 20260729)
    for _ in range(200):
        size = int(rng.uniform() * 10000)
        nchunks = 1 + int(rng.uniform() * 32)
        expected_next_start, sizes = 0, []
        for nchunk in range(nchunks):
            nstart, nsize = adist.packet_shard(size, nchunks, nchunk)
            assert nstart == expected_next_start and nsize >= 0
            expected_next_start = nstart + nsize
            sizes.append(nsize)
        assert expected_next_start == size and max(sizes) - min(sizes) <= 1
    assert [adist.packet_shard(10, 3, r) for r in range(3)] == [(0, 4), (4, 3), (7, 3)]
    # get_chunk_count (mpi_logging.h:179; globals.h:401 counts the keep-bitmap words with it): abi.keepwordcount's rule
    chunk_count = lambda size, mx: size // mx + (1 if size % mx else 0)  # noqa: E731
    assert chunk_count(0, 5) == 0 and chunk_count(10, 5) == 2 and chunk_count(11, 5) == 3


def test_escapedirectionbin_unittests_cc_175(oracle):
    """unittests.cc:175 test_escapedirectionbin on the oracle's restatement of get_escapedirectionbin (vectors.h:147) and on
    tools/exspec.py's numpy form: 200 000 isotropic directions from the reference's generator (seed 5501) fall in [0, MABINS) and
    fill the equal-solid-angle bins to within six sigma; the two forms agree on every direction."""
    import sys

    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    import exspec

    L = oracle.lib()
    ndirs = 200000
    s = (C.c_uint32 * 4)()
    L.artis_oracle_rng_seed(s, C.c_uint32(5501))
    dirs = np.zeros((ndirs, 3))
    L.artis_oracle_fill_isotropic(s, C.c_int64(ndirs), dirs.ctypes.data_as(C.c_void_p))
    L.artis_oracle_escapedirectionbin.restype = C.c_int
    L.artis_oracle_escapedirectionbin.argtypes = [C.c_void_p]
    bins_np = exspec.escapedirectionbin(dirs)
    sample = np.array([L.artis_oracle_escapedirectionbin(dirs[i].ctypes.data_as(C.c_void_p)) for i in range(0, ndirs, 7)])
    assert np.array_equal(sample, bins_np[::7])
    assert bins_np.min() >= 0 and bins_np.max() < exspec.MABINS
    counts = np.bincount(bins_np, minlength=exspec.MABINS)
    expected = ndirs / exspec.MABINS
    sixsigma = 6. * math.sqrt(expected * (1. - 1. / exspec.MABINS))
    assert abs(counts.min() - expected) < sixsigma and abs(counts.max() - expected) < sixsigma
    # the axis itself and unnormalised directions (vectors.h:151: "sometimes dir vectors aren't accurately normalised")
    for d, want in (((0., 0., 1.), 90), ((0., 0., -1.), 0), ((0., 0., 3.), 90)):
        v = np.array(d)
        assert L.artis_oracle_escapedirectionbin(v.ctypes.data_as(C.c_void_p)) // 10 * 10 == want
        assert exspec.escapedirectionbin(v[None, :])[0] // 10 * 10 == want
