/*
 * artis_amd.h -- C-ABI of the MI355X packet-propagation engine for ARTIS.
 *
 * Drop-in boundary: the per-timestep packet update of the reference,
 *   update_packets(nts, packets)                     update_packets.cc:530
 *     -> update_packet_cellcache_group()             update_packets.cc:468
 *        -> do_packet() for TYPE_RPKT / TYPE_KPKT / TYPE_PRE_KPKT
 *                                                    update_packets.cc:257
 *           -> do_rpkt() / do_rpkt_step()            rpkt.cc:983 / rpkt.cc:542
 *           -> kpkt::do_kpkt(), do_kpkt_blackbody()  kpkt.cc:425 / kpkt.cc:399
 *           -> do_macroatom()                        macroatom.cc:360
 *     preceded by cellcacheslot_populate()           update_packets.cc:397
 * and the estimator reduction radfield::reduce_estimators() radfield.cc:988.
 *
 * Every struct below is a flat view (plain pointers and sizes) of arrays the
 * reference already owns; field names follow the reference's globals.
 * Nothing here depends on torch, HIP types or C++.
 */
#ifndef ARTIS_AMD_H
#define ARTIS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums: values are the reference's, they are written to packets*.out -- */
/* packet.h:38 enum packet_type */
enum {
  ARTIS_TYPE_NONE = 0,
  ARTIS_TYPE_GAMMA = 10,
  ARTIS_TYPE_RPKT = 11,
  ARTIS_TYPE_KPKT = 12,
  ARTIS_TYPE_MA = 13,
  ARTIS_TYPE_NTLEPTON_DEPOSITED = 20,
  ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS = 21,
  ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAPLUS = 22,
  ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA = 23,
  ARTIS_TYPE_NTALPHA_FISPROD_DEPOSITED = 24,
  ARTIS_TYPE_ESCAPE = 32,
  ARTIS_TYPE_RADIOACTIVE_PELLET = 100,
  ARTIS_TYPE_PRE_KPKT = 120
};

/* packet.h:85 */
#define ARTIS_EMTYPE_NOTSET (-9999000)
#define ARTIS_EMTYPE_FREEFREE (-9999999)
/* packet.h:89 enum absorption_type */
#define ARTIS_ABSTYPE_FREEFREE (-1)
#define ARTIS_ABSTYPE_BOUNDFREE (-2)
#define ARTIS_ABSTYPE_GAMMA_COMPTON (-3)
#define ARTIS_ABSTYPE_GAMMA_PHOTOELECTRIC (-4)
#define ARTIS_ABSTYPE_GAMMA_PAIRPRODUCTION (-5)
#define ARTIS_ABSTYPE_PELLET_NOGAMMASPEC (-6)
#define ARTIS_ABSTYPE_PELLET_BEFORESIMSTART (-7)
#define ARTIS_ABSTYPE_PELLET_PARTICLEDECAY (-10)

/* globals.h:21 enum ma_action */
enum {
  ARTIS_MA_ACTION_RADDEEXC = 0,
  ARTIS_MA_ACTION_COLDEEXC = 1,
  ARTIS_MA_ACTION_RADRECOMB = 2,
  ARTIS_MA_ACTION_COLRECOMB = 3,
  ARTIS_MA_ACTION_INTERNALDOWNSAME = 4,
  ARTIS_MA_ACTION_INTERNALDOWNLOWER = 5,
  ARTIS_MA_ACTION_INTERNALUPSAME = 6,
  ARTIS_MA_ACTION_INTERNALUPHIGHER = 7,
  ARTIS_MA_ACTION_INTERNALUPHIGHERNT = 8,
  ARTIS_MA_ACTION_COUNT = 9
};

/* stats.h:13 enum class Counter (same numbering) */
enum {
  ARTIS_STAT_MA_ACTIVATION_COLLEXC = 0,
  ARTIS_STAT_MA_ACTIVATION_COLLION = 1,
  ARTIS_STAT_MA_ACTIVATION_NTCOLLEXC = 2,
  ARTIS_STAT_MA_ACTIVATION_NTCOLLION = 3,
  ARTIS_STAT_MA_ACTIVATION_BB = 4,
  ARTIS_STAT_MA_ACTIVATION_BF = 5,
  ARTIS_STAT_MA_ACTIVATION_FB = 6,
  ARTIS_STAT_MA_DEACTIVATION_COLLDEEXC = 7,
  ARTIS_STAT_MA_DEACTIVATION_COLLRECOMB = 8,
  ARTIS_STAT_MA_DEACTIVATION_BB = 9,
  ARTIS_STAT_MA_DEACTIVATION_FB = 10,
  ARTIS_STAT_MA_INTERNALUPHIGHER = 11,
  ARTIS_STAT_MA_INTERNALUPHIGHERNT = 12,
  ARTIS_STAT_MA_INTERNALDOWNLOWER = 13,
  ARTIS_STAT_K_TO_MA_COLLEXC = 14,
  ARTIS_STAT_K_TO_MA_COLLION = 15,
  ARTIS_STAT_K_TO_R_FF = 16,
  ARTIS_STAT_K_TO_R_FB = 17,
  ARTIS_STAT_K_TO_R_BB = 18,
  ARTIS_STAT_K_FROM_FF = 19,
  ARTIS_STAT_K_FROM_BF = 20,
  ARTIS_STAT_NT_FROM_GAMMA = 21,
  ARTIS_STAT_NT_TO_IONISATION = 22,
  ARTIS_STAT_NT_TO_EXCITATION = 23,
  ARTIS_STAT_NT_TO_KPKT = 24,
  ARTIS_STAT_K_FROM_EARLIERDECAY = 25,
  ARTIS_STAT_INTERACTIONS = 26,
  ARTIS_STAT_ELECTRON_SCATTERINGS = 27,
  ARTIS_STAT_RESONANCESCATTERINGS = 28,
  ARTIS_STAT_CELLCROSSINGS = 29,
  ARTIS_STAT_UPSCATTER = 30,
  ARTIS_STAT_DOWNSCATTER = 31,
  ARTIS_STAT_UPDATECELL = 32,
  ARTIS_STAT_PKTESCAPES = 33,
  ARTIS_STAT_COUNT = 34,
  /* extra slots, not in the reference: the unit of the headline metric */
  ARTIS_STAT_X_RPKT_STEPS = 34, /* calls of do_rpkt_step() rpkt.cc:542 */
  ARTIS_STAT_X_KPKT_STEPS = 35, /* calls of do_kpkt()/do_kpkt_blackbody() */
  ARTIS_STAT_X_LINES_VISITED = 36, /* lines walked in get_possible_event() rpkt.cc:121 */
  ARTIS_STAT_X_MA_JUMPS = 37, /* iterations of the do_macroatom() loop macroatom.cc:385 */
  ARTIS_STAT_X_CHI_EVALS = 38, /* continuum opacity evaluations that missed the packet's cache, rpkt.cc:1029 */
  ARTIS_STAT_X_CONT_VISITED = 39, /* bound-free continua summed in calculate_chi_bf_gammacontr() rpkt.cc:808 */
  ARTIS_STAT_X_GAMMA_STEPS = 40, /* calls of gammapkt::do_gamma() gammapkt.cc:911 */
  /* virtual packets (vpkt.h:34-37): nvpkt_created, nvpkt_esc_from_rpkt / _kpkt / _macroatom */
  ARTIS_STAT_X_VPKT_CREATED = 48, ARTIS_STAT_X_VPKT_ESC_RPKT = 49, ARTIS_STAT_X_VPKT_ESC_KPKT = 50, ARTIS_STAT_X_VPKT_ESC_MA = 51,
  /* 41 (do_kpkt's third phase clock), 42..47 and 52..63: free for profiling builds (-DARTIS_PROFILE / -DARTIS_PROFILE_MA: wave-cycle accounting; the
   * do_rpkt_step() phase clocks also take 48..52, so a profiling build of a VPKT_ON preset is refused at compile time) */
  ARTIS_NSTATS = 64
};

/* constants.h:73 GridType */
enum { ARTIS_GRID_SPHERICAL1D = 0, ARTIS_GRID_CYLINDRICAL2D = 1, ARTIS_GRID_CARTESIAN3D = 2 };
/* grid.h:29 CellThickness */
enum { ARTIS_CELL_THIN = 0, ARTIS_CELL_THICK = 1, ARTIS_CELL_THICK_VPKT_ONLY = 2 };
/* kpkt.cc:42 CoolingType */
enum { ARTIS_COOLING_FREEFREE = 0, ARTIS_COOLING_FREEBOUND = 1, ARTIS_COOLING_COLLEXC = 2, ARTIS_COOLING_COLLION = 3 };

/* ---- packet: layout of the reference's struct Packet in its GPU_ON form --- */
/* packet.h:117-169 (rngstate first: packet.h:118-122). Field order and types
 * are the reference's so a reference-side Packet* can be passed unchanged. */
typedef struct artis_packet {
  uint32_t rngstate[4]; /* Xoshiro128PP state, random.h:101 */
  double prop_time;
  double pos[3];
  double dir[3];
  double nu_cmf;
  double e_cmf;
  double nu_rf;
  double e_rf;
  int32_t next_trans;
  int32_t nscatterings;
  int32_t emissiontype;
  double em_pos[3];
  float em_time;
  int32_t absorptiontype;
  double absorptionfreq;
  double stokes_q;
  double stokes_u;
  int32_t trueemissiontype;
  double trueem_pos[3];
  float trueem_time;
  int32_t type;
  int32_t cellindex;
  int32_t escape_type;
  float escape_time;
  double tdecay;
  int32_t number;
  uint8_t originated_from_particlenotgamma;
  int32_t pellet_decaytype;
  int32_t pellet_nucindex;
} artis_packet;

/* ---- static model: atomic data + propagation grid (set once per run) ------ */
typedef struct artis_model {
  /* sizes */
  int32_t nelements;
  int32_t nions;      /* get_includedions() atomic.h:391 */
  int32_t nlevels;    /* get_includedlevels() atomic.h:37 */
  int32_t nlines;     /* globals::nlines */
  int32_t nalltrans;  /* size of globals::alltrans arrays */
  int32_t nphixstargets_total; /* size of globals::allphixstargets_* */
  int32_t nphixslevels;        /* number of levels with a phixs table */
  int32_t nbfcontinua;         /* globals::nbfcontinua */
  int32_t nbfcontinua_ground;  /* globals::nbfcontinua_ground */
  int32_t ncoolingterms;       /* kpkt ncoolingterms kpkt.cc:233 */
  int32_t nmatransblock;       /* sum over levels of 2*ndowntrans+nuptrans input.cc:1542 */
  int32_t NPHIXSPOINTS;        /* globals::NPHIXSPOINTS */
  double NPHIXSNUINCREMENT;    /* globals::NPHIXSNUINCREMENT */

  /* per element (globals::elements, globals.h:60) */
  const int32_t *elem_nions;
  const int32_t *elem_uniqueionindexstart;
  const int32_t *elem_anumber;
  const int32_t *elem_lowest_ionstage;

  /* per ion, indexed by uniqueionindex (struct Ion, globals.h:45) */
  const int32_t *ion_element;
  const int32_t *ion_nlevels;
  const int32_t *ion_nlevels_ionising;
  const int32_t *ion_maxrecombininglevel;
  const int32_t *ion_uniquelevelindexstart;
  const int32_t *ion_coolingoffset;
  const int32_t *ion_ncoolingterms;

  /* per level, indexed by uniquelevelindex (struct AllLevels, globals.h:181) */
  const double *level_epsilon;
  const float *level_statweight;
  const int32_t *level_alltrans_startdown;
  const int32_t *level_ndowntrans;
  const int32_t *level_nuptrans;
  const int32_t *level_closestgroundlevelcont;
  const int32_t *level_phixsstart;
  const int32_t *level_nphixstargets;
  const int32_t *level_phixstargetstart;
  const int32_t *level_bflist_start;
  const int32_t *level_matransblock_start;

  /* globals::alltrans (struct AllTransitions, globals.h:151) */
  const int32_t *alltrans_lineindex;
  const int32_t *alltrans_targetlevelindex;
  const float *alltrans_einstein_A;
  const float *alltrans_coll_str;
  const float *alltrans_osc_strength;
  const uint8_t *alltrans_forbidden;

  /* globals::linelist (struct TransitionLines, globals.h:229), descending nu */
  const double *line_nu;
  const int32_t *line_elementindex;
  const int32_t *line_ionindex;
  const int32_t *line_uniquelevelindex_lower;
  const int32_t *line_uniquelevelindex_upper;
  const float *line_B_ul;
  const float *line_B_lu;

  /* photoionisation */
  const float *allphixs;                       /* [nphixslevels*NPHIXSPOINTS] globals.h:149 */
  const int32_t *allphixstargets_levelindex;   /* globals.h:173 */
  const double *allphixstargets_probability;   /* globals.h:174 */

  /* globals::allcont (struct AllCont, globals.h:253), ascending nu_edge */
  const double *allcont_nu_edge;
  const int32_t *allcont_element;
  const int32_t *allcont_ion;
  const int32_t *allcont_level;
  const int32_t *allcont_phixstargetindex;
  const int32_t *allcont_upperlevel;
  const int32_t *allcont_uniquelevelindex;
  const double *allcont_probability;
  const int32_t *allcont_groundcontestimindex;

  /* ground-level continua, ascending nu_edge (globals.h:272-274) */
  const double *groundcont_nu_edge;

  /* temperature LUTs, [nbfcontinua*TABLESIZE], index get_bflutindex() ratecoeff.cc:123 */
  const double *spontrecombcoeffs;
  const double *corrphotoioncoeffs;
  const double *bfcooling_coeffs;

  /* cooling list (kpkt.cc:44-46) */
  /* (the entries of an ion after its collisional excitations must be in the order calculate_cooling_rates_ion() writes them -- kpkt.cc:122-190,
   * i.e. as setup_coolinglist() lays them out: artis_amd_engine_create() checks it and refuses a list that is not, ARTIS_ERR_ARG) */
  const uint8_t *coolinglist_type;
  const int32_t *coolinglist_level;
  const int32_t *coolinglist_phixstargetindex;

  /* propagation grid (grid.cc) */
  int32_t gridtype;       /* ARTIS_GRID_* */
  int32_t ncoordgrid[3];  /* grid.cc ncoordgrid */
  int32_t ngrid;          /* grid::ngrid */
  int32_t npts_nonempty;  /* grid::get_nonempty_npts_model() */
  double tmin;            /* globals::tmin */
  double vmax;            /* globals::vmax */
  double rmax;            /* globals::rmax */
  const double *coord_pos_min_tmin[3]; /* grid.cc coord_pos_min_tmin */
  const int32_t *propcell_nonemptymgi; /* [ngrid] grid::get_propcell_nonemptymgi(), -1 = empty */

  /* static inputs of the non-thermal channels, required by builds with ARTIS_OPT_NT_ON (else may be NULL):
   * [nelements] globals::elements[].initstablemeannucmass (grid.cc:1515, USE_CALCULATED_MEANATOMICWEIGHT off) and
   * [nions] nonthermal get_sum_q_over_binding_energy(element, ion) (nonthermal.cc:553; from the host's binding-energy data) */
  const float *elem_meannucmass;
  const double *ion_nt_sum_q_over_binding;

  /* whole-ejecta scalars of the Barnes thermalisation efficiency (update_packets.cc:69-77): grid::get_ejecta_kinetic_energy()
   * (grid.h:139) and grid::mtot_input (grid.h:40). Read only by builds with that scheme; 0 elsewhere. */
  double ejecta_kinetic_energy;
  double mtot_input;

  /* [nbfcontinua] globals::allcont.bfestimindex (input.cc:932-947): index of the continuum's detailed bound-free estimator,
   * -1 where LEVEL_HAS_BFEST() is false; nbfestim = number of estimators (globals::bfestim_nu_edge.size()). NULL / 0: every
   * continuum has one (artisoptions_nltenebular.h:82), the estimator index is the continuum index. */
  const int32_t *allcont_bfestimindex;
  int32_t nbfestim;

  /* [npts_nonempty] grid::get_rho_tmin(mgi): the density at tmin, read by the Wollaeger and Guttman gamma-ray
   * thermalisation schemes (gammapkt.cc:819, :853) for the column density along a ray; NULL elsewhere */
  const float *rho_tmin;

  /* XCOM photoionisation cross sections (gammapkt.cc:244-261 photoion_data, read from xcom_photoion_data.txt), per element
   * of the model: element e has the points [xcom_elem_start[e], xcom_elem_start[e+1]) of xcom_energy [MeV, rising] and
   * xcom_sigma [cm^2]; an element beyond the table or without data has none. Builds with USE_XCOM_GAMMAPHOTOION (which also
   * read elem_meannucmass for the element number densities); NULL elsewhere. */
  const int32_t *xcom_elem_start; /* [nelements + 1] */
  const double *xcom_energy;
  const double *xcom_sigma;

  /* [detailed_linecount] radfield.cc detailed_lineindices: the lines with their own intensity estimator, rising line index
   * (builds with DETAILED_LINE_ESTIMATORS_ON; NULL / 0 elsewhere) */
  const int32_t *detailed_lineindices;
  int32_t detailed_linecount;

  /* ---- virtual packets (builds with VPKT_ON): what read_vpktparameterfile() (vpkt.cc:673) leaves of vpkt.txt; 0 / NULL
   * elsewhere (ABI 5). Observer directions (costheta, phi); the opacity choices of every spectrum (vpkt.cc:79: 0 full, -1 no
   * lines, -2 no bound-free, -3 no free-free, -4 no electron scattering, Z > 0 without that element's lines); the arrival
   * time window and the frequency ranges in which a virtual packet is traced at all; the optical depth at which it is
   * dropped; the optional velocity-grid map (time window, frequency ranges); globals::nprocs (vpkt.cc:130). */
  int32_t vpkt_nobsdirections;
  const double *vpkt_obsdirs_costheta; /* [vpkt_nobsdirections] */
  const double *vpkt_obsdirs_phi;
  int32_t vpkt_nspectraperobsdir;
  const int32_t *vpkt_opacityexclusions; /* [vpkt_nspectraperobsdir] */
  double vpkt_timemin_input, vpkt_timemax_input;
  int32_t vpkt_nwavelengthranges;
  const double *vpkt_numin_input; /* [vpkt_nwavelengthranges] */
  const double *vpkt_numax_input;
  double vpkt_tau_max;
  int32_t vpkt_vgrid_on;
  double vpkt_tmin_grid, vpkt_tmax_grid;
  int32_t vpkt_grid_nwavelengthranges;
  const double *vpkt_nu_grid_min; /* [vpkt_grid_nwavelengthranges] */
  const double *vpkt_nu_grid_max;
  int32_t vpkt_nprocs;
} artis_model;
/* the fixed grids of the virtual-packet spectra, vpkt.h:21-32 */
#define ARTIS_VGRID_NY 50
#define ARTIS_VGRID_NZ 50
#define ARTIS_VSPEC_NUBINS 2500
#define ARTIS_VSPEC_TIMEBINS 5
#define ARTIS_VSPEC_NUMIN (2.99792458e+10 / 10000 * 1e8)
#define ARTIS_VSPEC_NUMAX (2.99792458e+10 / 3500 * 1e8)
#define ARTIS_VSPEC_TIMEMIN (3 * 86400.)
#define ARTIS_VSPEC_TIMEMAX (8 * 86400.)

/* ---- per-timestep cell state written by the reference's update_grid() ----- */
typedef struct artis_cellstate {
  /* [npts_nonempty], grid.h:19-37 */
  const float *rho;
  const float *Te;
  const float *TJ;
  const float *TR;
  const float *W;
  const float *nne;
  const float *nnetot;
  const float *kappagrey;
  const int32_t *thick;
  const float *clumpfactor;
  /* [npts_nonempty*nions] grid.h:44-45 */
  const float *ion_groundlevelpops;
  const float *ion_partfuncts;
  /* [npts_nonempty*nelements] grid.h:42 */
  const float *elem_massfracs;
  /* [npts_nonempty*nbfcontinua_ground] globals.h:126 */
  const double *corrphotoionrenorm;
  /* [npts_nonempty] grid::get_ffegrp(mgi): iron-group mass fraction, read by the gamma-ray opacities
   * (gammapkt.cc:416, :516). May be NULL when no TYPE_GAMMA packet is handed over (treated as 0). */
  const float *ffegrp;
  /* [npts_nonempty*nlevels] level populations from the host's NLTE / LTE solution (ltepop.cc:169-180 get_levelpop); NULL:
   * the engine evaluates the LTE populations itself (calculate_levelpop_lte ltepop.cc:395) */
  const double *levelpops;
  /* [npts_nonempty*nphixstargets_total] corrected photoionisation coefficient of every bound-free pair, what
   * get_corrphotoioncoeff() (ratecoeff.cc:840) returns when USE_LUT_PHOTOION is off (the normalised bound-free estimators of
   * the previous timestep, or the radiation-field integral): required by builds with ARTIS_OPT_USE_LUT_PHOTOION == 0 */
  const double *corrphotoioncoeff;
  /* [npts_nonempty*RADFIELDBINCOUNT] W and T_R of the multibin radiation field (radfield.cc:208-214; W < 0: bin without a
   * solution): required by builds with ARTIS_OPT_MULTIBIN_RADFIELD_MODEL_ON */
  const float *radfieldbin_W;
  const float *radfieldbin_T_R;
  /* Solution of the host's Spencer-Fano solver, read by the non-thermal channels of do_ntlepton_deposit() and of the
   * macro-atom (nonthermal.cc:215-250): required by builds with ARTIS_OPT_NT_ON.
   *   nt_frac_ionisation / nt_frac_excitation [npts_nonempty]  NonThermalCellSolution::frac_ionisation / frac_excitation
   *   nt_deposition_rate_density [npts_nonempty]               ntlepton_deposition_rate_density_all_cells
   *   nt_eff_ionpot [npts_nonempty*nions]                      NonThermalSolutionIon::eff_ionpot
   *   nt_prob_num_auger, nt_ionenfrac_num_auger [npts_nonempty*nions*(NT_MAX_AUGER_ELECTRONS+1)]
   *   nt_exc_count [npts_nonempty]                             NonThermalCellSolution::frac_excitations_list_size
   *   nt_exc_* [npts_nonempty*nt_excitations_stored]           the cell's NonThermalExcitation list, ascending alltransindex */
  const float *nt_frac_ionisation;
  const float *nt_frac_excitation;
  const double *nt_deposition_rate_density;
  const float *nt_eff_ionpot;
  const float *nt_prob_num_auger;
  const float *nt_ionenfrac_num_auger;
  const int32_t *nt_exc_count;
  const double *nt_exc_frac_deposition;
  const double *nt_exc_ratecoeffperdeposition;
  const int32_t *nt_exc_alltransindex;
  int32_t nt_excitations_stored; /* nonthermal.cc nt_excitations_stored: stride of the nt_exc_* lists */
  /* [npts_nonempty*ARTIS_EXPOPAC_NBINS] what calculate_expansion_opacities() (rpkt.cc:1071, called from update_grid.cc:655)
   * leaves per cell: the binned line opacity kappa [cm^2/g] (read by builds with RPKT_USE_EXPANSION_OPACITIES) and the
   * running integral of kappa * B_nu(T_e) over the bins (read with RPKT_BOUNDBOUND_THERMALISATION_PROBABILITY).
   * Both NULL: the engine calculates them itself when it populates the cell cache. */
  const float *expansionopacities;
  const double *expansionopacity_planck_cumulative;
  /* [npts_nonempty*detailed_linecount] radfield.cc prev_Jb_lu_normed[].value: the normalised line intensities of the previous
   * timestep, what get_Jb_lu() returns (builds with DETAILED_LINE_ESTIMATORS_ON) */
  const double *Jb_lu_normed;
  /* [npts_nonempty*nelements] grid::elem_meanweight_allcells (grid.cc:1509): the mean atomic weight [g] of each element in
   * each cell; read for the element number densities (XCOM photoelectric opacities, non-thermal channels) by builds with
   * USE_CALCULATED_MEANATOMICWEIGHT (kilonova_lte and its variants); NULL elsewhere (ABI 4) */
  const float *elem_meanweight;
} artis_cellstate;

typedef struct artis_timestep {
  int32_t nts;          /* timestep number */
  double start;         /* globals::timesteps[nts].start */
  double width;         /* globals::timesteps[nts].width */
  double mid;           /* globals::timesteps[nts].mid */
  double max_path_step; /* globals::max_path_step update_grid.cc:753 */
} artis_timestep;

/* ---- estimators: accumulated into, never zeroed, by update_packets -------- */
typedef struct artis_estimators {
  double *J;                   /* [npts_nonempty] radfield.cc J */
  double *nuJ;                 /* [npts_nonempty] radfield.cc nuJ */
  double *ffheatingestimator;  /* [npts_nonempty] globals.h:133 */
  double *colheatingestimator; /* [npts_nonempty] globals.h:134 */
  double *gammaestimator;      /* [npts_nonempty*nbfcontinua_ground] globals.h:128 */
  double *bfheatingestimator;  /* [npts_nonempty*nbfcontinua_ground] globals.h:131 */
  int64_t *stats;              /* [ARTIS_NSTATS] stats.cc event counters */
  double *dep_estimator_gamma; /* [npts_nonempty] globals::dep_estimator_gamma (gammapkt.cc:568); may be NULL */
  double *scalars;             /* [ARTIS_NSCALARS] per-timestep sums, see ARTIS_SCALAR_*; may be NULL */
  /* [npts_nonempty] globals::dep_estimator_electron / _positron / _alpha (update_packets.cc:160-173); may be NULL */
  double *dep_estimator_electron;
  double *dep_estimator_positron;
  double *dep_estimator_alpha;
  /* multibin radiation field estimators radfieldbins.J_raw / nuJ_raw [npts_nonempty*RADFIELDBINCOUNT] (radfield.cc:745-790)
   * and detailed bound-free estimators bfrate_raw [npts_nonempty*nbfestim] (radfield.cc:215; nbfestim = nbfcontinua unless
   * artis_model.allcont_bfestimindex selects a subset). Written by builds with the corresponding options; may be NULL. */
  double *radfieldbin_J;
  double *radfieldbin_nuJ;
  double *bfrate_raw;
  /* detailed line estimators Jb_lu_raw[][].value and .contribcount [npts_nonempty*detailed_linecount] (radfield.cc:773
   * update_lineestimator). Builds with DETAILED_LINE_ESTIMATORS_ON; may be NULL. */
  double *Jb_lu_raw;
  int64_t *Jb_lu_contribcount;
  /* virtual-packet spectra vspecpol[timebin][obsdir * nspectraperobsdir + opacity choice].flux[nubin].{I, Q, U}
   * (vpkt.cc:56, add_to_vspecpol :116): [ARTIS_VSPEC_TIMEBINS][nobsdirections*nspectraperobsdir][ARTIS_VSPEC_NUBINS][3], and the
   * velocity-grid map vgrid[ny][nz].flux[wlbin][obsdir].{I, Q, U} (:138): [ARTIS_VGRID_NY][ARTIS_VGRID_NZ][grid_nwavelengthranges]
   * [nobsdirections][3]. Builds with VPKT_ON; may be NULL (ABI 5). The counters nvpkt_created / nvpkt_esc_from_* are
   * stats[ARTIS_STAT_X_VPKT_*]. */
  double *vspecpol;
  double *vgrid_flux;
} artis_estimators;
enum {
  ARTIS_SCALAR_GAMMA_DEP_DISCRETE = 0,   /* globals::timesteps[nts].gamma_dep_discrete gammapkt.cc:926 */
  ARTIS_SCALAR_NT_ENERGY_DEPOSITED = 1,  /* nonthermal.cc nt_energy_deposited (:2524, :2530) */
  /* globals::timesteps[nts].* written by update_pellet() (update_packets.cc:199-231) ... */
  ARTIS_SCALAR_PELLET_DECAYS = 2,
  ARTIS_SCALAR_GAMMA_EMISSION = 3,
  ARTIS_SCALAR_POSITRON_EMISSION = 4,
  ARTIS_SCALAR_ELECTRON_EMISSION = 5,
  ARTIS_SCALAR_ALPHA_EMISSION = 6,
  ARTIS_SCALAR_SPFISSION_DEP_DISCRETE = 7,
  /* ... and by do_nonthermal_predeposit() (update_packets.cc:162-172) */
  ARTIS_SCALAR_ELECTRON_DEP_DISCRETE = 8,
  ARTIS_SCALAR_POSITRON_DEP_DISCRETE = 9,
  ARTIS_SCALAR_ALPHA_DEP_DISCRETE = 10,
  ARTIS_NSCALARS = 11
};
/* decay::DecayType values read from Packet::pellet_decaytype (decay.h:21-26) */
enum { ARTIS_DECAYTYPE_ALPHA = 0, ARTIS_DECAYTYPE_ELECTRONCAPTURE = 1, ARTIS_DECAYTYPE_BETAPLUS = 2, ARTIS_DECAYTYPE_BETAMINUS = 3,
       ARTIS_DECAYTYPE_NONE = 4, ARTIS_DECAYTYPE_SPONTFISSION = 5 };

/* ---- engine ---------------------------------------------------------------- */
typedef struct artis_amd_engine artis_amd_engine;

/* All functions return 0 on success, a negative ARTIS_ERR_* otherwise;
 * artis_amd_last_error() gives the message. No function falls back to a CPU
 * path: without a usable HIP device artis_amd_engine_create() fails. */
#define ARTIS_OK 0
#define ARTIS_ERR_NODEVICE (-1)
#define ARTIS_ERR_HIP (-2)
#define ARTIS_ERR_ARG (-3)
#define ARTIS_ERR_UNSUPPORTED (-4)
#define ARTIS_ERR_NOTCONVERGED (-5)
#define ARTIS_ERR_RCCL (-6)

const char *artis_amd_last_error(void);
int artis_amd_abi_version(void); /* 6: artis_amd_record_tiers() added (no struct changed); 5: virtual-packet configuration and spectra appended; 4: artis_cellstate.elem_meanweight appended (3: cell state and estimators of the nltenebular options) */
/* Name of the options preset the library was compiled with (include/artis_options.h): "classic" or "kilonova_lte".
 * Like the reference, one binary per artisoptions.h. */
const char *artis_amd_options_preset(void);
size_t artis_amd_sizeof_packet(void);

/* Create an engine on HIP device `device` and upload the static model.
 * Replaces the table set-up the reference's GPU build leaves in unified memory
 * (MPI_shared_array, mpi_logging.h). */
int artis_amd_engine_create(const artis_model *model, int device, artis_amd_engine **out);
void artis_amd_engine_destroy(artis_amd_engine *eng);

/* Upload the cell state of the coming timestep and fill every cell's cache:
 * cellcacheslot_populate() for all non-empty cells (update_packets.cc:397,
 * multi-slot form update_packets.cc:551-560) and the per-ion cumulative
 * cooling of kpkt::calculate_cooling_rates() (kpkt.cc:281). */
int artis_amd_set_cellstate(artis_amd_engine *eng, const artis_cellstate *cells, const artis_timestep *ts);

/* Re-run the cell-cache population for the cell state already uploaded (what update_packets() of the
 * reference does at its start, update_packets.cc:551-560). hip_stream is a hipStream_t (NULL = default). */
int artis_amd_populate_cellcache(artis_amd_engine *eng, void *hip_stream);

/* Cell-cache tiling. The cache of every non-empty cell is resident when it fits the budget (60 % of the free HBM at
 * engine creation, or ARTIS_AMD_CACHE_BUDGET_MB); otherwise there are rows for `*cells_per_tile` cells at a time
 * (`*ntiles` = how many such sets cover the model) and artis_amd_update_packets*() visits sets of cells (make the cells in
 * which most packets wait resident -- cells that are resident already keep their rows --, advance the packets that sit in
 * them until they leave the set or are done, next set, ... until no packet is left): the reference's single-slot cell cache
 * (update_packets.cc:397-460, :551-621) with a set of cells instead of one cell. With a cache that does not fit the engine also
 * chooses macro-atom record tiers that need few tiles (artis_amd_record_tiers). Packet histories do not depend on any of it. */
int artis_amd_cache_tiles(artis_amd_engine *eng, int32_t *ntiles, int64_t *cells_per_tile, int64_t *bytes_per_cell);

/* Host-buffer form of update_packets() (update_packets.cc:530): every packet
 * whose type is in do_packet()'s switch (update_packets.cc:257: pellets, gamma
 * packets, non-thermal pre-deposits and deposits, r-, k- and pre-k-packets) is
 * advanced to the end of the timestep with the physics of the classic preset
 * (include/artis_options.h); packets of any other type are returned untouched.
 * Estimators are ADDED to est (host arrays; NULL members are skipped). */
int artis_amd_update_packets(artis_amd_engine *eng, artis_packet *packets, int64_t npackets,
                             artis_estimators *est);

/* Device-resident form, for callers that keep packets in HBM across timesteps:
 * upload once, step many times, download when needed. */
int artis_amd_packets_upload(artis_amd_engine *eng, const artis_packet *packets, int64_t npackets);
int artis_amd_packets_download(artis_amd_engine *eng, artis_packet *packets, int64_t npackets);
int artis_amd_packets_snapshot(artis_amd_engine *eng); /* save current device packets */
int artis_amd_packets_restore(artis_amd_engine *eng);  /* restore the snapshot */
/* Propagate the resident packets through the timestep set by
 * artis_amd_set_cellstate(). hip_stream is a hipStream_t (NULL = default). */
int artis_amd_update_packets_device(artis_amd_engine *eng, void *hip_stream);
int artis_amd_estimators_zero(artis_amd_engine *eng, void *hip_stream);
int artis_amd_estimators_download(artis_amd_engine *eng, artis_estimators *est_add_into);

/* Device pointers of the estimator block for an in-place RCCL all-reduce
 * (the reference's radfield::reduce_estimators(), radfield.cc:988, and the
 * MPI_Allreduce of the heating/gamma estimators in sn3d.cc). The block is one
 * contiguous array of `*ndoubles` doubles. */
int artis_amd_estimators_devptr(artis_amd_engine *eng, void **dptr, int64_t *ndoubles);

/* The estimator reduction at the end of a timestep in the host layer: ONE in-place RCCL all-reduce (sum, f64) of the
 * whole estimator block over the ranks' GPUs, enqueued on hip_stream. Replaces radfield::reduce_estimators()
 * (radfield.cc:988) and the MPI_Allreduce calls on the heating / photoionisation estimators in the timestep loop
 * (sn3d.cc:565-590). nccl_comm is the caller's ncclComm_t, or NULL to use the communicator made by
 * artis_amd_comm_init(). RCCL is bound at run time (the librccl already in the process, else ROCm's). */
#define ARTIS_AMD_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
int artis_amd_allreduce_estimators(artis_amd_engine *eng, void *nccl_comm, void *hip_stream);
/* Communicator set-up for hosts that do not hold an ncclComm_t yet: rank 0 draws an id (ncclGetUniqueId) and hands its
 * 128 bytes to the other ranks by whatever the host already has (MPI_Bcast in the reference, globals::my_rank /
 * nprocs); every rank then calls artis_amd_comm_init (ncclCommInitRank on the engine's device; collective). The
 * engine owns the communicator and destroys it with itself. */
int artis_amd_comm_unique_id(void *id_out /* ARTIS_AMD_COMM_ID_BYTES */);
int artis_amd_comm_init(artis_amd_engine *eng, int nranks, int rank, const void *id_bytes);
/* Number of ranks RCCL itself reports for the communicator (ncclCommCount): what the reduce really spans. */
int artis_amd_comm_count(artis_amd_engine *eng, void *nccl_comm, int *nranks);

/* Timing of the dominant kernel inside the last artis_amd_update_packets_device
 * call, measured with HIP events on the launch stream. */
int artis_amd_last_kernel_ms(artis_amd_engine *eng, double *propagate_ms, int64_t *nlaunches);
/* Tiled cell cache (artis_amd_cache_tiles): sweeps over the tiles made by the last artis_amd_update_packets_device call, tile
 * fills inside it with their summed duration [ms], and the packets listed over all (sweep, tile) visits. */
int artis_amd_last_tiling(artis_amd_engine *eng, int64_t *sweeps, int64_t *tile_fills, double *fill_ms, int64_t *listed);
/* ... of which sparse fills (only the cells of the tile in which packets waited), and the cells populated over all fills */
int artis_amd_last_tiling_fills(artis_amd_engine *eng, int64_t *sparse_fills, int64_t *cells_filled);
/* ... and the packets that a visit of a tile left waiting in it for the tile's next visit instead of running them to their end in a
 * launch of their own (round 4: the last <= ARTIS_AMD_TAIL packets of a visit that began larger; ARTIS_AMD_TILE_PARK=0 switches it off) */
int artis_amd_last_tiling_parked(artis_amd_engine *eng, int64_t *parked);
/* On-demand macro-atom records (rows that hold static records for the lowest levels of every ion only: DESIGN.md section 2): how often the last
 * artis_amd_update_packets_device call found the shared pool of the other levels' records used up and emptied it (the records are filled again
 * when next needed: it costs fills, never an answer; the reference, which fills a level's rates on first use too, keeps them all: macroatom.cc:398-417) */
int artis_amd_last_pool_resets(artis_amd_engine *eng, int64_t *resets);
/* Which record tiers the engine keeps for this model (chosen at artis_amd_engine_create from the cache budget -- free device memory at that
 * moment or ARTIS_AMD_CACHE_BUDGET_MB -- unless ARTIS_AMD_MA_HOTFRAC gives them): the share of every ion's levels with a static record in every
 * cell's row (1 = all of them: no on-demand records), the number of cold levels, and the 16-byte slots of the shared pool per resident cell.
 * Two runs are comparable in time and in artis_amd_last_pool_resets() only if these agree. Any pointer may be NULL. (ABI 6) */
int artis_amd_record_tiers(artis_amd_engine *eng, double *hot_fraction, int32_t *ncold_levels, int64_t *pool_slots);
/* ... and how much of that pool the last artis_amd_update_packets_device call left in use (since the pool was last emptied), in units of 128
 * bytes, beside the pool's size: a pool that runs near its size thrashes (artis_amd_last_pool_resets), one far below it could give its memory
 * to static records (a larger hot fraction). (ABI 6) */
int artis_amd_last_pool_usage(artis_amd_engine *eng, int64_t *units_used, int64_t *units_cap);
/* Which forms of the thermal kernel (macro-atom walks + k-packet steps) the last artis_amd_update_packets_device call
 * launched, as a mask: the form is chosen per launch from the atomic data's size and the list's length, and a parity test has to know that the
 * form it means to check is the one that ran. (ABI 6) */
#define ARTIS_AMD_THERMAL_PLAIN 1         /* k_thermal<256, 0>: target tables in HBM (any size; lists below 4096 entries) */
#define ARTIS_AMD_THERMAL_LDS_TABLES 2    /* k_thermal<1024, 1>: target levels + LevelPack in LDS (<= 2048 levels, <= 32768 transitions) */
#define ARTIS_AMD_THERMAL_LDS_LEVELPACK 4 /* k_thermal<1024, 2>: LevelPack alone in LDS (<= 6144 levels) */
#define ARTIS_AMD_THERMAL_REFILL 8        /* k_thermal_q (ARTIS_AMD_REFILL=1) */
#define ARTIS_AMD_THERMAL_COLD 16         /* ... instantiated with the on-demand records' look-ups (the model has cold levels) */
#define ARTIS_AMD_THERMAL_TAIL 32         /* k_tail took the population's last packets */
int artis_amd_last_thermal_variants(artis_amd_engine *eng, int32_t *mask);

/* Per-kernel split of the last artis_amd_update_packets_device call: summed launch durations [ms] and summed
 * packet counts of the r-packet kernel (k_rpkt) and of the thermal kernels (k_ma + k_kpkt). */
int artis_amd_last_kernel_breakdown(artis_amd_engine *eng, double *rpkt_ms, int64_t *rpkt_threads, double *thermal_ms,
                                    int64_t *thermal_threads);

int artis_amd_last_kernel_launches(artis_amd_engine *eng, int64_t *rpkt_launches, int64_t *thermal_launches);
/* The same for each kernel: index 0 k_rpkt, 1 k_ma, 2 k_kpkt, 3 k_slow. */
int artis_amd_last_kernel_table(artis_amd_engine *eng, double ms[4], int64_t launches[4], int64_t packets[4]);

/* Diagnostics: the cell cache of one non-empty cell in the reference's layout (globals::cellcache[nonemptymgi] spans,
 * globals.h:283-311). Any pointer may be NULL. The engine's rows hold the macro-atom's cumulative sums as 15-bit filters only
 * (DESIGN.md section 2): `matrans` (allmacroatomictransitions, globals.h:287) is re-added on the device from the transitions'
 * rate coefficients -- the values a draw gets that the filters cannot decide -- and the call fails with
 * ARTIS_ERR_NOTCONVERGED if a filter entry of the cell's records differs from the sequential form's. An engine that keeps
 * on-demand records (atomic data whose static records do not fit; ARTIS_AMD_MA_HOTFRAC) shows the records
 * that exist: a cold level that no packet has reached in this cell since the cache was filled reads as zeros. */
int artis_amd_debug_cellcache(artis_amd_engine *eng, int nonemptymgi, double *levelpops, double *maprocessrates,
                              double *matrans, double *allcont_nnlevel, double *allcont_departure,
                              double *allcont_edgepart, uint64_t *allcont_keepbits, double *corrphotoioncoeff,
                              double *cooling_contrib, double *ion_cooling_contribs, double *chi_ff_nnionpart);

/* Summed launch durations (HIP events on the launch stream) of the last artis_amd_update_packets_device() call by kind of kernel:
 * 0 k_rpkt (+ k_bfest_dense), 1 k_thermal, 2 k_slow, 3 k_gamma, 4 k_blackbody, 5 k_tail, 6 tile fills inside the call; 7 unused.
 * launches may be NULL. */
int artis_amd_last_kernel_ms_by_kind(artis_amd_engine *eng, double ms[8], int64_t launches[8]);

/* Measurement builds only (-DARTIS_VISIT_COUNTS, tools/visit_sparsity.py): how many macro-atom transitions the last
 * propagation call drew in the record of every (non-empty cell, level), counts[nonemptymgi * nlevels + level]; the reference fills a
 * level's rates when a packet first reaches it (macroatom.cc:398-417 calc_rates_if_needed). Any other build returns ARTIS_ERR_ARG. */
int artis_amd_debug_visit_counts(artis_amd_engine *eng, uint32_t *counts, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* ARTIS_AMD_H */
