"""ctypes mirror of include/artis_amd.h (the C-ABI structs), backed by numpy arrays.

This is plumbing for tests and bench.py: it only describes memory that is handed
to the C-ABI (the HIP engine, or -- from tests -- the CPU oracle). It contains no
physics.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

STAT_X_VPKT_CREATED = 48  # then _ESC_RPKT, _ESC_KPKT, _ESC_MA (include/artis_amd.h ARTIS_STAT_X_VPKT_*)
ABI_VERSION = 6  # artis_amd_abi_version() of the library these ctypes structs describe (include/artis_amd.h)
NSTATS = 64
NSCALARS = 11  # ARTIS_SCALAR_* of include/artis_amd.h
SCALAR_NAMES = ["gamma_dep_discrete", "nt_energy_deposited", "pellet_decays", "gamma_emission", "positron_emission",
                "electron_emission", "alpha_emission", "spfission_dep_discrete", "electron_dep_discrete",
                "positron_dep_discrete", "alpha_dep_discrete"]
STAT_COUNT = 34
STAT_X_RPKT_STEPS = 34
STAT_X_KPKT_STEPS = 35
STAT_X_LINES_VISITED = 36
STAT_X_MA_JUMPS = 37
STAT_X_GAMMA_STEPS = 40

TYPE_GAMMA = 10
TYPE_RPKT = 11
TYPE_KPKT = 12
TYPE_NTLEPTON_DEPOSITED = 20
TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS = 21
TYPE_NONTHERMAL_PREDEPOSIT_BETAPLUS = 22
TYPE_NONTHERMAL_PREDEPOSIT_ALPHA = 23
TYPE_NTALPHA_FISPROD_DEPOSITED = 24
TYPE_RADIOACTIVE_PELLET = 100
TYPE_ESCAPE = 32
TYPE_PRE_KPKT = 120
EMTYPE_NOTSET = -9999000
EMTYPE_FREEFREE = -9999999

GRID_SPHERICAL1D = 0
GRID_CYLINDRICAL2D = 1
GRID_CARTESIAN3D = 2

STAT_NAMES = [
    "MA_STAT_ACTIVATION_COLLEXC", "MA_STAT_ACTIVATION_COLLION", "MA_STAT_ACTIVATION_NTCOLLEXC",
    "MA_STAT_ACTIVATION_NTCOLLION", "MA_STAT_ACTIVATION_BB", "MA_STAT_ACTIVATION_BF", "MA_STAT_ACTIVATION_FB",
    "MA_STAT_DEACTIVATION_COLLDEEXC", "MA_STAT_DEACTIVATION_COLLRECOMB", "MA_STAT_DEACTIVATION_BB",
    "MA_STAT_DEACTIVATION_FB", "MA_STAT_INTERNALUPHIGHER", "MA_STAT_INTERNALUPHIGHERNT",
    "MA_STAT_INTERNALDOWNLOWER", "K_STAT_TO_MA_COLLEXC", "K_STAT_TO_MA_COLLION", "K_STAT_TO_R_FF", "K_STAT_TO_R_FB",
    "K_STAT_TO_R_BB", "K_STAT_FROM_FF", "K_STAT_FROM_BF", "NT_STAT_FROM_GAMMA", "NT_STAT_TO_IONISATION",
    "NT_STAT_TO_EXCITATION", "NT_STAT_TO_KPKT", "K_STAT_FROM_EARLIERDECAY", "INTERACTIONS", "ELECTRON_SCATTERINGS",
    "RESONANCESCATTERINGS", "CELLCROSSINGS", "UPSCATTER", "DOWNSCATTER", "UPDATECELL", "PKTESCAPES",
    "X_RPKT_STEPS", "X_KPKT_STEPS", "X_LINES_VISITED", "X_MA_JUMPS", "X_CHI_EVALS", "X_CONT_VISITED",
    "X_GAMMA_STEPS",
] + [f"X_{i}" for i in range(41, 64)]

# struct artis_packet (include/artis_amd.h), natural C alignment == numpy align=True
PACKET_DTYPE = np.dtype(
    [
        ("rngstate", np.uint32, (4,)),
        ("prop_time", np.float64),
        ("pos", np.float64, (3,)),
        ("dir", np.float64, (3,)),
        ("nu_cmf", np.float64),
        ("e_cmf", np.float64),
        ("nu_rf", np.float64),
        ("e_rf", np.float64),
        ("next_trans", np.int32),
        ("nscatterings", np.int32),
        ("emissiontype", np.int32),
        ("em_pos", np.float64, (3,)),
        ("em_time", np.float32),
        ("absorptiontype", np.int32),
        ("absorptionfreq", np.float64),
        ("stokes_q", np.float64),
        ("stokes_u", np.float64),
        ("trueemissiontype", np.int32),
        ("trueem_pos", np.float64, (3,)),
        ("trueem_time", np.float32),
        ("type", np.int32),
        ("cellindex", np.int32),
        ("escape_type", np.int32),
        ("escape_time", np.float32),
        ("tdecay", np.float64),
        ("number", np.int32),
        ("originated_from_particlenotgamma", np.uint8),
        ("pellet_decaytype", np.int32),
        ("pellet_nucindex", np.int32),
    ],
    align=True,
)

PACKET_INT_FIELDS = ["next_trans", "nscatterings", "emissiontype", "absorptiontype", "trueemissiontype", "type",
                     "cellindex", "escape_type", "number"]
PACKET_FLOAT_FIELDS = ["prop_time", "pos", "dir", "nu_cmf", "e_cmf", "nu_rf", "e_rf", "em_pos", "em_time",
                       "absorptionfreq", "stokes_q", "stokes_u", "trueem_pos", "trueem_time", "escape_time"]

_I32P = C.POINTER(C.c_int32)
_F32P = C.POINTER(C.c_float)
_F64P = C.POINTER(C.c_double)
_U8P = C.POINTER(C.c_uint8)
_I64P = C.POINTER(C.c_int64)

# (name, ctype-or-pointer, numpy dtype or None for scalars) in the exact order of struct artis_model
_MODEL_FIELDS = [
    ("nelements", C.c_int32, None), ("nions", C.c_int32, None), ("nlevels", C.c_int32, None),
    ("nlines", C.c_int32, None), ("nalltrans", C.c_int32, None), ("nphixstargets_total", C.c_int32, None),
    ("nphixslevels", C.c_int32, None), ("nbfcontinua", C.c_int32, None), ("nbfcontinua_ground", C.c_int32, None),
    ("ncoolingterms", C.c_int32, None), ("nmatransblock", C.c_int32, None), ("NPHIXSPOINTS", C.c_int32, None),
    ("NPHIXSNUINCREMENT", C.c_double, None),
    ("elem_nions", _I32P, np.int32), ("elem_uniqueionindexstart", _I32P, np.int32), ("elem_anumber", _I32P, np.int32),
    ("elem_lowest_ionstage", _I32P, np.int32),
    ("ion_element", _I32P, np.int32), ("ion_nlevels", _I32P, np.int32), ("ion_nlevels_ionising", _I32P, np.int32),
    ("ion_maxrecombininglevel", _I32P, np.int32), ("ion_uniquelevelindexstart", _I32P, np.int32),
    ("ion_coolingoffset", _I32P, np.int32), ("ion_ncoolingterms", _I32P, np.int32),
    ("level_epsilon", _F64P, np.float64), ("level_statweight", _F32P, np.float32),
    ("level_alltrans_startdown", _I32P, np.int32), ("level_ndowntrans", _I32P, np.int32),
    ("level_nuptrans", _I32P, np.int32), ("level_closestgroundlevelcont", _I32P, np.int32),
    ("level_phixsstart", _I32P, np.int32), ("level_nphixstargets", _I32P, np.int32),
    ("level_phixstargetstart", _I32P, np.int32), ("level_bflist_start", _I32P, np.int32),
    ("level_matransblock_start", _I32P, np.int32),
    ("alltrans_lineindex", _I32P, np.int32), ("alltrans_targetlevelindex", _I32P, np.int32),
    ("alltrans_einstein_A", _F32P, np.float32), ("alltrans_coll_str", _F32P, np.float32),
    ("alltrans_osc_strength", _F32P, np.float32), ("alltrans_forbidden", _U8P, np.uint8),
    ("line_nu", _F64P, np.float64), ("line_elementindex", _I32P, np.int32), ("line_ionindex", _I32P, np.int32),
    ("line_uniquelevelindex_lower", _I32P, np.int32), ("line_uniquelevelindex_upper", _I32P, np.int32),
    ("line_B_ul", _F32P, np.float32), ("line_B_lu", _F32P, np.float32),
    ("allphixs", _F32P, np.float32), ("allphixstargets_levelindex", _I32P, np.int32),
    ("allphixstargets_probability", _F64P, np.float64),
    ("allcont_nu_edge", _F64P, np.float64), ("allcont_element", _I32P, np.int32), ("allcont_ion", _I32P, np.int32),
    ("allcont_level", _I32P, np.int32), ("allcont_phixstargetindex", _I32P, np.int32),
    ("allcont_upperlevel", _I32P, np.int32), ("allcont_uniquelevelindex", _I32P, np.int32),
    ("allcont_probability", _F64P, np.float64), ("allcont_groundcontestimindex", _I32P, np.int32),
    ("groundcont_nu_edge", _F64P, np.float64),
    ("spontrecombcoeffs", _F64P, np.float64), ("corrphotoioncoeffs", _F64P, np.float64),
    ("bfcooling_coeffs", _F64P, np.float64),
    ("coolinglist_type", _U8P, np.uint8), ("coolinglist_level", _I32P, np.int32),
    ("coolinglist_phixstargetindex", _I32P, np.int32),
    ("gridtype", C.c_int32, None), ("ncoordgrid", C.c_int32 * 3, "i3"), ("ngrid", C.c_int32, None),
    ("npts_nonempty", C.c_int32, None), ("tmin", C.c_double, None), ("vmax", C.c_double, None),
    ("rmax", C.c_double, None), ("coord_pos_min_tmin", _F64P * 3, "p3"), ("propcell_nonemptymgi", _I32P, np.int32),
    # optional (NULL when absent): static inputs of the non-thermal channels
    ("elem_meannucmass", _F32P, np.float32), ("ion_nt_sum_q_over_binding", _F64P, np.float64),
    ("ejecta_kinetic_energy", C.c_double, None), ("mtot_input", C.c_double, None),
    ("allcont_bfestimindex", _I32P, np.int32), ("nbfestim", C.c_int32, None),
    ("rho_tmin", _F32P, np.float32),
    ("xcom_elem_start", _I32P, np.int32), ("xcom_energy", _F64P, np.float64), ("xcom_sigma", _F64P, np.float64),
    ("detailed_lineindices", _I32P, np.int32), ("detailed_linecount", C.c_int32, None),
    # optional: the virtual-packet configuration (vpkt.txt as read_vpktparameterfile() leaves it; builds with VPKT_ON)
    ("vpkt_nobsdirections", C.c_int32, None), ("vpkt_obsdirs_costheta", _F64P, np.float64), ("vpkt_obsdirs_phi", _F64P, np.float64),
    ("vpkt_nspectraperobsdir", C.c_int32, None), ("vpkt_opacityexclusions", _I32P, np.int32),
    ("vpkt_timemin_input", C.c_double, None), ("vpkt_timemax_input", C.c_double, None),
    ("vpkt_nwavelengthranges", C.c_int32, None), ("vpkt_numin_input", _F64P, np.float64), ("vpkt_numax_input", _F64P, np.float64),
    ("vpkt_tau_max", C.c_double, None), ("vpkt_vgrid_on", C.c_int32, None),
    ("vpkt_tmin_grid", C.c_double, None), ("vpkt_tmax_grid", C.c_double, None),
    ("vpkt_grid_nwavelengthranges", C.c_int32, None), ("vpkt_nu_grid_min", _F64P, np.float64), ("vpkt_nu_grid_max", _F64P, np.float64),
    ("vpkt_nprocs", C.c_int32, None),
]
VSPEC_NUBINS, VSPEC_TIMEBINS, VGRID_NY, VGRID_NZ = 2500, 5, 50, 50  # vpkt.h:21-32
_MODEL_OPTIONAL = ("elem_meannucmass", "ion_nt_sum_q_over_binding", "ejecta_kinetic_energy", "mtot_input",
                   "allcont_bfestimindex", "nbfestim", "rho_tmin", "xcom_elem_start", "xcom_energy", "xcom_sigma",
                   "detailed_lineindices", "detailed_linecount",
                   "vpkt_nobsdirections", "vpkt_obsdirs_costheta", "vpkt_obsdirs_phi", "vpkt_nspectraperobsdir",
                   "vpkt_opacityexclusions", "vpkt_timemin_input", "vpkt_timemax_input", "vpkt_nwavelengthranges",
                   "vpkt_numin_input", "vpkt_numax_input", "vpkt_tau_max", "vpkt_vgrid_on", "vpkt_tmin_grid", "vpkt_tmax_grid",
                   "vpkt_grid_nwavelengthranges", "vpkt_nu_grid_min", "vpkt_nu_grid_max", "vpkt_nprocs")

_CELL_FIELDS = [
    ("rho", _F32P, np.float32), ("Te", _F32P, np.float32), ("TJ", _F32P, np.float32), ("TR", _F32P, np.float32),
    ("W", _F32P, np.float32), ("nne", _F32P, np.float32), ("nnetot", _F32P, np.float32),
    ("kappagrey", _F32P, np.float32), ("thick", _I32P, np.int32), ("clumpfactor", _F32P, np.float32),
    ("ion_groundlevelpops", _F32P, np.float32), ("ion_partfuncts", _F32P, np.float32),
    ("elem_massfracs", _F32P, np.float32), ("corrphotoionrenorm", _F64P, np.float64),
    ("ffegrp", _F32P, np.float32),
    # optional (NULL when absent from the dict): host level populations, host photoionisation coefficients, multibin field
    ("levelpops", _F64P, np.float64), ("corrphotoioncoeff", _F64P, np.float64),
    ("radfieldbin_W", _F32P, np.float32), ("radfieldbin_T_R", _F32P, np.float32),
    # optional: the Spencer-Fano solution (builds with NT_ON)
    ("nt_frac_ionisation", _F32P, np.float32), ("nt_frac_excitation", _F32P, np.float32),
    ("nt_deposition_rate_density", _F64P, np.float64), ("nt_eff_ionpot", _F32P, np.float32),
    ("nt_prob_num_auger", _F32P, np.float32), ("nt_ionenfrac_num_auger", _F32P, np.float32),
    ("nt_exc_count", _I32P, np.int32), ("nt_exc_frac_deposition", _F64P, np.float64),
    ("nt_exc_ratecoeffperdeposition", _F64P, np.float64), ("nt_exc_alltransindex", _I32P, np.int32),
    ("nt_excitations_stored", C.c_int32, None),
    ("expansionopacities", _F32P, np.float32), ("expansionopacity_planck_cumulative", _F64P, np.float64),
    ("Jb_lu_normed", _F64P, np.float64),
    ("elem_meanweight", _F32P, np.float32),
]
EXPOPAC_NBINS = 1997
_CELL_OPTIONAL = ("levelpops", "corrphotoioncoeff", "radfieldbin_W", "radfieldbin_T_R", "nt_frac_ionisation",
                  "nt_frac_excitation", "nt_deposition_rate_density", "nt_eff_ionpot", "nt_prob_num_auger",
                  "nt_ionenfrac_num_auger", "nt_exc_count", "nt_exc_frac_deposition", "nt_exc_ratecoeffperdeposition",
                  "nt_exc_alltransindex", "nt_excitations_stored", "expansionopacities", "expansionopacity_planck_cumulative", "Jb_lu_normed",
                  "elem_meanweight")
NT_NAUGER = 3  # NT_MAX_AUGER_ELECTRONS + 1
RADFIELDBINCOUNT = 256
# options presets with the multibin radiation field and the detailed bound-free estimators (artisoptions_nltenebular.h and
# the three files that differ from it in constants): RADFIELDBINCOUNT of each
NEBULAR_FAMILY = {"nltenebular": 256, "christinenonthermal": 64, "nltephotospheric": 256, "nltewithoutnonthermal": 512,
                  "nltenebular_lineest": 256, "ci_nebular": 256, "ci_nebular_limitbfest": 256, "ci_nltephotospheric": 24}
# The option sets of the reference's CI (tests/setup_*.sh; include/artis_options.h ARTIS_PRESET_CI_*): the setup script, the
# preset whose host inputs the build needs (what synth.build hands over), and TABLESIZE / MINTEMP / MAXTEMP
CI_PRESETS = {
    "ci_kilonova": ("setup_kilonova_2d.sh", "kilonova_lte", (20, 1000.0, 20000.0)),
    "ci_kilonova_barnes": ("setup_kilonova_2d_barnesthermalisation.sh", "kilonova_lte", (20, 1000.0, 20000.0)),
    "ci_kilonova_expopac": ("setup_kilonova_2d_expansionopac.sh", "kilonova_expopac", (20, 1000.0, 20000.0)),
    "ci_kilonova_xcom": ("setup_kilonova_2d_xcomgammaphotoion.sh", "classic_gamma_xcom", (20, 1000.0, 20000.0)),
    "ci_nebular": ("setup_nebular_1d_3dgrid.sh", "nltenebular", (20, 2000.0, 10000.0)),
    "ci_nebular_limitbfest": ("setup_nebular_1d_3dgrid_limitbfest.sh", "nltephotospheric", (20, 2000.0, 10000.0)),
    "ci_nltephotospheric": ("setup_nltephotospheric_dynamic_ion_range_1d_1dgrid.sh", "nltephotospheric", (40, 3500.0, 140000.0)),
    # classic + virtual packets (VPKT_ON): line by line, and with the binned expansion opacities beyond the first bin
    "ci_classic_vpkt": ("setup_classicmode_1d_3dgrid.sh", "classic", (100, 3500.0, 140000.0)),
    "ci_classic_vpkt_expopac": ("setup_classicmode_3d.sh", "classic", (100, 3500.0, 140000.0)),
}


def inputs_like(options: str) -> str:
    """the preset whose kind of host inputs (synth.build) an options preset needs"""
    return CI_PRESETS[options][1] if options in CI_PRESETS else options


class CModel(C.Structure):
    _fields_ = [(n, t) for n, t, _ in _MODEL_FIELDS]


class CCellState(C.Structure):
    _fields_ = [(n, t) for n, t, _ in _CELL_FIELDS]


class CTimestep(C.Structure):
    _fields_ = [("nts", C.c_int32), ("start", C.c_double), ("width", C.c_double), ("mid", C.c_double),
                ("max_path_step", C.c_double)]


class CEstimators(C.Structure):
    _fields_ = [("J", _F64P), ("nuJ", _F64P), ("ffheatingestimator", _F64P), ("colheatingestimator", _F64P),
                ("gammaestimator", _F64P), ("bfheatingestimator", _F64P), ("stats", _I64P),
                ("dep_estimator_gamma", _F64P), ("scalars", _F64P), ("dep_estimator_electron", _F64P),
                ("dep_estimator_positron", _F64P), ("dep_estimator_alpha", _F64P),
                ("radfieldbin_J", _F64P), ("radfieldbin_nuJ", _F64P), ("bfrate_raw", _F64P),
                ("Jb_lu_raw", _F64P), ("Jb_lu_contribcount", _I64P), ("vspecpol", _F64P), ("vgrid_flux", _F64P)]


def _as_ptr(arr: np.ndarray, ptype):
    return arr.ctypes.data_as(ptype)


class Model:
    """Static model: dict of numpy arrays + scalars -> struct artis_model."""

    def __init__(self, d: dict):
        self.d = {}
        self.c = CModel()
        for name, ctype, npdt in _MODEL_FIELDS:
            if name in _MODEL_OPTIONAL and d.get(name) is None:
                continue  # stays a NULL pointer
            v = d[name]
            if npdt is None:
                setattr(self.c, name, v)
                self.d[name] = v
            elif npdt == "i3":
                arr = np.ascontiguousarray(v, dtype=np.int32)
                self.d[name] = arr
                for k in range(3):
                    self.c.ncoordgrid[k] = int(arr[k])
            elif npdt == "p3":
                arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in v]
                self.d[name] = arrs
                for k in range(3):
                    self.c.coord_pos_min_tmin[k] = _as_ptr(arrs[k], _F64P)
            else:
                arr = np.ascontiguousarray(v, dtype=npdt)
                if arr.size == 0:  # keep a valid pointer for empty tables
                    arr = np.zeros(1, dtype=npdt)
                self.d[name] = arr
                setattr(self.c, name, _as_ptr(arr, ctype))

    def __getitem__(self, k):
        return self.d[k]

    def ref(self):
        return C.byref(self.c)


class CellState:
    def __init__(self, d: dict):
        self.d = {}
        self.c = CCellState()
        for name, ctype, npdt in _CELL_FIELDS:
            if name in _CELL_OPTIONAL and d.get(name) is None:
                continue  # stays a NULL pointer
            if npdt is None:
                setattr(self.c, name, int(d[name]))
                self.d[name] = int(d[name])
                continue
            arr = np.ascontiguousarray(d[name], dtype=npdt)
            if arr.size == 0:
                arr = np.zeros(1, dtype=npdt)
            self.d[name] = arr
            setattr(self.c, name, _as_ptr(arr, ctype))

    def __getitem__(self, k):
        return self.d[k]

    def ref(self):
        return C.byref(self.c)


class Timestep:
    def __init__(self, nts: int, start: float, width: float, mid: float, max_path_step: float):
        self.c = CTimestep(nts, start, width, mid, max_path_step)

    def ref(self):
        return C.byref(self.c)


class Estimators:
    """Host estimator arrays (accumulated into by update_packets)."""

    def __init__(self, npts_nonempty: int, nbfcontinua_ground: int, nbfcontinua: int = 0, nbins: int = RADFIELDBINCOUNT,
                 ndetailedlines: int = 0, vpkt_shape=None):
        """vpkt_shape = (nobsdirections * nspectraperobsdir, nobsdirections * grid_nwavelengthranges or 0): also the
        virtual-packet spectra and velocity-grid map (builds with VPKT_ON). nbfcontinua > 0: also the estimators of the nltenebular options (radiation-field bins, detailed bound-free;
        nbfcontinua = the number of bound-free estimators, nbins = RADFIELDBINCOUNT of the options preset)"""
        n, g = npts_nonempty, max(nbfcontinua_ground, 1)
        self.radfieldbin_J = np.zeros(n * nbins if nbfcontinua > 0 else 1)
        self.radfieldbin_nuJ = np.zeros(n * nbins if nbfcontinua > 0 else 1)
        self.bfrate_raw = np.zeros(n * nbfcontinua if nbfcontinua > 0 else 1)
        self.Jb_lu_raw = np.zeros(max(n * ndetailedlines, 1))
        self.Jb_lu_contribcount = np.zeros(max(n * ndetailedlines, 1), dtype=np.int64)
        self.lineest = ndetailedlines > 0
        self.extended = nbfcontinua > 0
        self.vpkt = vpkt_shape is not None
        ncomb, ngridcomb = vpkt_shape if self.vpkt else (0, 0)
        self.vspecpol = np.zeros(max(VSPEC_TIMEBINS * ncomb * VSPEC_NUBINS * 3, 1))
        self.vgrid_flux = np.zeros(max(VGRID_NY * VGRID_NZ * ngridcomb * 3, 1))
        self.J = np.zeros(n)
        self.nuJ = np.zeros(n)
        self.ffheatingestimator = np.zeros(n)
        self.colheatingestimator = np.zeros(n)
        self.gammaestimator = np.zeros(n * g)
        self.bfheatingestimator = np.zeros(n * g)
        self.stats = np.zeros(NSTATS, dtype=np.int64)
        self.dep_estimator_gamma = np.zeros(n)
        self.scalars = np.zeros(NSCALARS)
        self.dep_estimator_electron = np.zeros(n)
        self.dep_estimator_positron = np.zeros(n)
        self.dep_estimator_alpha = np.zeros(n)
        self.c = CEstimators(
            _as_ptr(self.J, _F64P), _as_ptr(self.nuJ, _F64P), _as_ptr(self.ffheatingestimator, _F64P),
            _as_ptr(self.colheatingestimator, _F64P), _as_ptr(self.gammaestimator, _F64P),
            _as_ptr(self.bfheatingestimator, _F64P), _as_ptr(self.stats, _I64P),
            _as_ptr(self.dep_estimator_gamma, _F64P), _as_ptr(self.scalars, _F64P),
            _as_ptr(self.dep_estimator_electron, _F64P), _as_ptr(self.dep_estimator_positron, _F64P),
            _as_ptr(self.dep_estimator_alpha, _F64P),
            _as_ptr(self.radfieldbin_J, _F64P) if self.extended else None,
            _as_ptr(self.radfieldbin_nuJ, _F64P) if self.extended else None,
            _as_ptr(self.bfrate_raw, _F64P) if self.extended else None,
            _as_ptr(self.Jb_lu_raw, _F64P) if self.lineest else None,
            _as_ptr(self.Jb_lu_contribcount, _I64P) if self.lineest else None,
            _as_ptr(self.vspecpol, _F64P) if self.vpkt else None,
            _as_ptr(self.vgrid_flux, _F64P) if (self.vpkt and ngridcomb > 0) else None)

    def ref(self):
        return C.byref(self.c)

    def arrays(self):
        ext = {"radfieldbin_J": self.radfieldbin_J, "radfieldbin_nuJ": self.radfieldbin_nuJ,
               "bfrate_raw": self.bfrate_raw} if self.extended else {}
        if self.lineest:
            ext = {**ext, "Jb_lu_raw": self.Jb_lu_raw, "Jb_lu_contribcount": self.Jb_lu_contribcount}
        if self.vpkt:
            ext = {**ext, "vspecpol": self.vspecpol, "vgrid_flux": self.vgrid_flux}
        return {**ext, "J": self.J, "nuJ": self.nuJ, "ffheatingestimator": self.ffheatingestimator,
                "colheatingestimator": self.colheatingestimator, "gammaestimator": self.gammaestimator,
                "bfheatingestimator": self.bfheatingestimator, "dep_estimator_gamma": self.dep_estimator_gamma,
                "scalars": self.scalars, "dep_estimator_electron": self.dep_estimator_electron,
                "dep_estimator_positron": self.dep_estimator_positron, "dep_estimator_alpha": self.dep_estimator_alpha}

    def stats_dict(self):
        return {STAT_NAMES[i]: int(self.stats[i]) for i in range(NSTATS)}


def estimators_for(model, options: str = "classic") -> Estimators:
    """Estimator arrays sized for a model under an options preset (nltenebular adds the bin and bound-free arrays)"""
    if "vpkt" in options:
        nobs = model.d.get("vpkt_nobsdirections") or 0
        ngrid = (model.d.get("vpkt_grid_nwavelengthranges") or 0) if model.d.get("vpkt_vgrid_on") else 0
        return Estimators(model["npts_nonempty"], model["nbfcontinua_ground"],
                          vpkt_shape=(nobs * (model.d.get("vpkt_nspectraperobsdir") or 0), nobs * ngrid))
    if options not in NEBULAR_FAMILY:
        return Estimators(model["npts_nonempty"], model["nbfcontinua_ground"])
    nest = model.d.get("nbfestim") or model["nbfcontinua"]
    nlines = (model.d.get("detailed_linecount") or 0) if options.endswith("_lineest") else 0  # (no CI option set has them)
    return Estimators(model["npts_nonempty"], model["nbfcontinua_ground"], nest, NEBULAR_FAMILY[options], nlines)


def packets_ptr(packets: np.ndarray):
    assert packets.dtype == PACKET_DTYPE and packets.flags["C_CONTIGUOUS"]
    return packets.ctypes.data_as(C.c_void_p)


def seed_packet_rng(packets: np.ndarray, seed_base: int) -> None:
    """Per-packet Xoshiro128PP seeding: seed = seed_base + n through SplitMix32
    (reference random.h:32-40, 78-93, 117-123 and input.cc:1912-1916)."""
    n = len(packets)
    seed = (np.uint64(seed_base) + np.arange(n, dtype=np.uint64)) & np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        st = seed + np.uint64(0x9E3779B97F4A7C15)
        st = (st ^ (st >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        st = (st ^ (st >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        sm = ((st ^ (st >> np.uint64(31))) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        out = np.empty((n, 4), dtype=np.uint32)
        for i in range(4):
            sm = sm + np.uint32(0x9E3779B9)
            r = sm.copy()
            r = (r ^ (r >> np.uint32(16))) * np.uint32(0x21F0AAAD)
            r = (r ^ (r >> np.uint32(15))) * np.uint32(0x735A2D97)
            out[:, i] = r ^ (r >> np.uint32(15))
    packets["rngstate"] = out
