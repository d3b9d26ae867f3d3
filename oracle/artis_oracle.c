/*
 * artis_oracle.c -- TEST INFRASTRUCTURE, NOT A PRODUCT PATH.
 *
 * A sequential CPU restatement, in plain C, of the reference's packet path
 * (update_packets -> do_rpkt_step / do_kpkt / do_macroatom) with the
 * artisoptions_classic.h preset. Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it. The HIP engine never does.
 *
 * Every function cites the reference file:line it follows. The statement is
 * deliberately close to the reference's own order of floating-point operations
 * (double/float mixing included) so that results can be compared field by
 * field with the HIP engine.
 *
 * Where the reference's result depends on the order packets are processed in,
 * the oracle (and the engine) fix one order-independent choice:
 *  (1) RNG: the reference's per-packet generator of its GPU_ON build
 *      (packet.h:118-122, input.cc:1912-1916, random.h:185-187).
 *  (2) ContinuumOpacity cache: one per packet, reset when do_rpkt() is entered
 *      (GPU_ON analogue: chi_rpkt_cont_vec[index_in_group],
 *      update_packets.cc:478-501; CPU form is a thread_local shared by packets).
 *  (3) The lazily filled per-cell caches of calculate_chi_bf_gammacontr()
 *      (allcont_stimfactor_edgepart, rpkt.cc:840-889) and
 *      get_corrphotoioncoeff() (ratecoeff.cc:840) are filled when the cell
 *      cache is populated; every later evaluation in the reference reads
 *      exactly these values ("every writer stores identical values").
 *  (4) The cell cache is the multi-slot form (cellcache_singleslot == false,
 *      constants.h:84-88): all rates pre-calculated (update_packets.cc:442-458).
 *
 * Parity pinning (tests/test_oracle_reference_props.py): PARTIAL.
 *  - pinned bit for bit against reference code compiled where it lies (oracle/Makefile target `ref`, binaries in
 *    oracle/_ref/): random.h (generator streams and rng_uniform, tests/golden/rng_reference.json), gausskronrod.h
 *    (the adaptive integrator behind select_continuum_nu, tests/golden/gk31_reference.json) and macroatom.h
 *    (rad_deexcitation_ratecoeff, tests/golden/macroatom_reference.json);
 *  - pinned against the known answers and properties of the reference's unittests.cc for the pieces of this path
 *    (vector/Doppler/frame transforms, move_pkt_withtime, closest_transition, line distance, rate-coefficient helpers);
 *  - the transport loop as a whole is UNPINNED: the reference binary cannot be built in this image (it needs <print>
 *    and <mdspan>, MPI and a downloaded atomic data set), so no end-to-end output of it exists to compare with.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/artis_amd.h"
#include "../include/artis_options.h"

/* ------------------------------------------------------------------ constants.h */
#define CLIGHT 2.99792458e+10
#define CLIGHT_PROP CLIGHT
#define H_PLANCK 6.6260755e-27
#define ME 9.1093897e-28
#define MH 1.67352e-24
#define MEV 1.6021772e-6
#define MSUN 1.98855e+33 /* constants.h:27 */
#define DAY 86400.      /* constants.h:35 */
#define THOMSON_LIMIT 1e-2 /* constants.h:38 */
#define NU_100KEV 2.41326e+19 /* gammapkt.cc:64-67 */
#define NU_1MEV 2.41326e+20
#define NU_1P022MEV 2.46636e+20
#define NU_1P5MEV 3.61990e+20
#define PI 3.14159265358979323846
#define EV 1.6021772e-12
#define SIGMA_T 6.6524e-25
#define KB 1.38064852e-16
#define SAHACONST 2.0706659e-16
#define EULERGAMMA 0.577215664901532860606512090082402431
#define CLIGHTSQUARED (CLIGHT * CLIGHT)
#define CLIGHTSQUAREDOVERTWOH (CLIGHT * CLIGHT / (2 * H_PLANCK))
#define HOVERKB (H_PLANCK / KB)
#define HCLIGHTOVERFOURPI (H_PLANCK * CLIGHT / (4 * PI))
#define H_ionpot (13.5979996 * EV)
#define C_0 5.465e-11
#define DBL_MAXV 1.7976931348623157e308
#define DBL_MINV 2.2250738585072014e-308

static inline double pow2(double x) { return x * x; }
static inline double pow3(double x) { return x * x * x; }
static inline double dmin(double a, double b) { return (b < a) ? b : a; } /* std::min */
static inline double dmax(double a, double b) { return (a < b) ? b : a; } /* std::max */
static inline double dclamp(double v, double lo, double hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }

/* ------------------------------------------------------------------ state */
typedef struct {
  int populated;
  float *expansionopacities;                  /* [ARTIS_EXPOPAC_NBINS] when the host did not hand the tables over */
  double *expansionopacity_planck_cumulative; /* [ARTIS_EXPOPAC_NBINS] */
  int have_ion_cooling;         /* ion_cooling_contribs filled (kpkt.cc:281 is part of update_grid in the reference) */
  double chi_ff_nnionpart;
  double *levelpops;            /* [nlevels] alllevels_pops */
  double *maprocessrates;       /* [nlevels*9] alllevels_maprocessrates */
  double *matrans;              /* [nmatransblock] allmacroatomictransitions */
  double *allcont_nnlevel;      /* [nbfcontinua] */
  double *allcont_departure;    /* [nbfcontinua] allcont_modified_departureratios */
  double *allcont_edgepart;     /* [nbfcontinua] allcont_stimfactor_edgepart */
  uint64_t *allcont_keepbits;   /* [ceil(nbfcontinua/64)] */
  double *corrphotoioncoeff;    /* [nphixstargets_total] */
  double *cooling_contrib;      /* [ncoolingterms] */
  double *ion_cooling_contribs; /* [nions] cumulative, kpkt.cc:281 */
} CellCache;

typedef struct {
  const artis_model *m;
  const artis_cellstate *cs;
  artis_timestep ts;
  artis_estimators est;
  CellCache *cache; /* [npts_nonempty] */
  double temperature_grid[ARTIS_OPT_TABLESIZE + 1];
  double T_step_log;
  double last_phixs_nuovernuedge;
  int error;
  int npopulated;     /* cells currently holding a cache */
  int cache_cap;      /* evict everything when this many cells are cached (bounds host memory) */
  /* detailed bound-free estimators (input.cc:932-955): the estimator of every continuum (-1: none), their number and
   * their edge frequencies in rising order (globals::allcont.bfestimindex, globals::bfestim_nu_edge) */
  int32_t *allcont_bfestimindex;
  double *bfestim_nu_edge;
  int nbfestim;
  double t_populate;  /* seconds spent filling caches */
} Oracle;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static double g_last_populate_seconds = 0.;
static int64_t *g_visit_hist = NULL; /* diagnostics: macro-atom visits per (cell, level), set by artis_oracle_set_visit_hist */
static int g_visit_nlevels = 0;

/* ContinuumOpacity (rpkt.h:70) with its Phixslist (rpkt.h:48) */
typedef struct {
  double nu;
  double chi_escatter;
  double chi_freefree_heat;
  double chi_boundfree;
  int nonemptymgi;
  double *groundcont_gamma_contr; /* [nbfcontinua_ground] */
  /* Phixslist rpkt.h:48: with DETAILED_BF_ESTIMATORS_ON the contribution of every continuum of the window walked at `nu`
   * (every continuum has an estimator, LEVEL_HAS_BFEST true: the estimator index is the continuum index) */
  double *gamma_contr; /* [nbfcontinua] */
  int bfestimbegin, bfestimend;
} ContOpacity;

typedef struct { int element, ion, level, activatingline; } MacroAtomState;

#define ORACLE_FAIL(o, msg)                                              \
  do {                                                                   \
    if (!(o)->error) fprintf(stderr, "[oracle] assert failed: %s (%s:%d)\n", msg, __FILE__, __LINE__); \
    (o)->error = 1;                                                      \
  } while (0)

static inline void stat_inc(Oracle *o, int i) { o->est.stats[i]++; }

/* ------------------------------------------------------------------ random.h */
/* SplitMix32 seeding of Xoshiro128PP: random.h:32-40 (_mix_seed), random.h:78-93, random.h:117-123 */
static void rng_seed(uint32_t s[4], uint32_t seed) {
  uint64_t st = (uint64_t)seed + 0x9E3779B97f4A7C15ULL;
  st = (st ^ (st >> 30U)) * 0xBF58476D1CE4E5B9ULL;
  st = (st ^ (st >> 27U)) * 0x94D049BB133111EBULL;
  uint32_t sm = (uint32_t)(st ^ (st >> 31U));
  for (int i = 0; i < 4; i++) {
    uint32_t r = (sm += 0x9e3779b9U);
    r = (r ^ (r >> 16U)) * 0x21f0aaadU;
    r = (r ^ (r >> 15U)) * 0x735a2d97U;
    s[i] = r ^ (r >> 15U);
  }
}
static inline uint32_t rotl32(uint32_t x, unsigned k) { return (x << k) | (x >> (32U - k)); }
/* Xoshiro128PP::operator() random.h:125-136 */
static inline uint32_t rng_next(uint32_t s[4]) {
  const uint32_t result = rotl32(s[0] + s[3], 7U) + s[0];
  const uint32_t t = s[1] << 9U;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl32(s[3], 11U);
  return result;
}
/* rng_uniform, GPU_ON branch: random.h:178-193 with generate_canonical_float random.h:141-164 */
static inline float rng_uniform(uint32_t s[4]) {
  while (1) {
    const float zrand = (float)(rng_next(s) >> 8U) * 0x1.0p-24F;
    if (zrand != 1.) return zrand;
  }
}
/* random.h:195-202 */
static inline float rng_uniform_pos(uint32_t s[4]) {
  while (1) {
    const float zrand = rng_uniform(s);
    if (zrand > 0) return zrand;
  }
}

/* ------------------------------------------------------------------ vectors.h */
static inline double vec_len3(const double v[3]) { /* vectors.h:21 */
  double sq = 0.;
  for (int i = 0; i < 3; i++) sq += pow2(v[i]);
  return sqrt(sq);
}
static inline double dot3(const double x[3], const double y[3]) { /* vectors.h:40 */
  double sum = 0.;
  for (int i = 0; i < 3; i++) sum += x[i] * y[i];
  return sum;
}
static inline void vec_norm3(const double in[3], double out[3]) { /* vectors.h:31 */
  const double mag = vec_len3(in);
  out[0] = in[0] / mag;
  out[1] = in[1] / mag;
  out[2] = in[2] / mag;
}
static inline void get_velocity(const double x[3], double t, double v[3]) { /* vectors.h:50 */
  v[0] = x[0] / t;
  v[1] = x[1] / t;
  v[2] = x[2] / t;
}
static inline void cross_prod(const double a[3], const double b[3], double c[3]) { /* vectors.h:54 */
  c[0] = (a[1] * b[2]) - (b[1] * a[2]);
  c[1] = (a[2] * b[0]) - (b[2] * a[0]);
  c[2] = (a[0] * b[1]) - (b[0] * a[1]);
}
/* angle_ab vectors.h:70 */
static void angle_ab(const double dir1[3], const double vel[3], double dir2[3]) {
  const double vsqr = dot3(vel, vel) / CLIGHTSQUARED;
  const double gamma_rel = 1. / sqrt(1 - vsqr);
  const double ndotv = dot3(dir1, vel);
  const double fact1 = gamma_rel * (1 - (ndotv / CLIGHT));
  const double fact2 = (gamma_rel - (pow2(gamma_rel) * ndotv / (gamma_rel + 1) / CLIGHT)) / CLIGHT;
  const double tmp[3] = {(dir1[0] - (vel[0] * fact2)) / fact1, (dir1[1] - (vel[1] * fact2)) / fact1,
                         (dir1[2] - (vel[2] * fact2)) / fact1};
  vec_norm3(tmp, dir2);
}
/* calculate_doppler_nucmf_on_nurf vectors.h:92 */
static inline double doppler_nucmf_on_nurf(const double pos[3], const double dir[3], double prop_time) {
  double vel[3];
  get_velocity(pos, prop_time, vel);
  const double ndotv = dot3(dir, vel);
  double dopplerfactor = 1. - (ndotv / CLIGHT);
#if ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT
  const double betasq = dot3(vel, vel) / CLIGHTSQUARED;
  dopplerfactor = dopplerfactor / sqrt(1 - betasq);
#endif
  return dopplerfactor;
}
/* move_pkt_withtime vectors.h:119 */
static void move_pkt_withtime_raw(double pos[3], const double dir[3], double *prop_time, double nu_rf, double *nu_cmf,
                                  double e_rf, double *e_cmf, double distance) {
  const double nu_cmf_old = *nu_cmf;
  *prop_time += distance / CLIGHT_PROP;
  pos[0] = pos[0] + (dir[0] * distance);
  pos[1] = pos[1] + (dir[1] * distance);
  pos[2] = pos[2] + (dir[2] * distance);
  const double dopplerfactor = doppler_nucmf_on_nurf(pos, dir, *prop_time);
  *nu_cmf = dmin(nu_rf * dopplerfactor, nu_cmf_old);
  *e_cmf = e_rf * dopplerfactor;
}
static void move_pkt_withtime(artis_packet *p, double distance) { /* vectors.h:139 */
  move_pkt_withtime_raw(p->pos, p->dir, &p->prop_time, p->nu_rf, &p->nu_cmf, p->e_rf, &p->e_cmf, distance);
}
/* set_pkt_restframe_from_cmf vectors.h:145 */
static void set_pkt_restframe_from_cmf(artis_packet *p) {
  const double d = doppler_nucmf_on_nurf(p->pos, p->dir, p->prop_time);
  p->nu_rf = p->nu_cmf / d;
  p->e_rf = p->e_cmf / d;
}
/* get_rand_isotropic_unitvec vectors.h:185 */
static void get_rand_isotropic_unitvec(uint32_t s[4], double out[3]) {
  const double u = rng_uniform(s);
  const double costheta = (2. * u) - 1.;
  const double sintheta = 2. * sqrt(u * (1. - u));
  const double phi = rng_uniform(s) * 2 * PI;
  out[0] = sintheta * cos(phi);
  out[1] = sintheta * sin(phi);
  out[2] = costheta;
}
/* get_rot_angle vectors.h:196 */
static double get_rot_angle(const double n1[3], const double n2[3], const double ref1[3], const double ref2[3]) {
  const double n1_dot_n2 = dot3(n1, n2);
  const double u[3] = {(n1[0] * n1_dot_n2) - n2[0], (n1[1] * n1_dot_n2) - n2[1], (n1[2] * n1_dot_n2) - n2[2]};
  const double len = vec_len3(u);
  if (len < 1e-12) return 0.0;
  const double ref1_sc[3] = {u[0] / len, u[1] / len, u[2] / len};
  const double cos_stokes_rot_1 = dclamp(dot3(ref1_sc, ref1), -1., 1.);
  const double cos_stokes_rot_2 = dot3(ref1_sc, ref2);
  const double rot_angle = atan2(cos_stokes_rot_2, cos_stokes_rot_1);
  return rot_angle < 0 ? rot_angle + (2 * PI) : rot_angle;
}
/* meridian vectors.h:219 */
static void meridian(const double dir[3], double ref1[3], double ref2[3]) {
  const double n_xylen = sqrt(pow2(dir[0]) + pow2(dir[1]));
  if (n_xylen == 0.) {
    ref1[0] = 1.; ref1[1] = 0.; ref1[2] = 0.;
    ref2[0] = 0.; ref2[1] = 1.; ref2[2] = 0.;
    return;
  }
  ref1[0] = -dir[0] * dir[2] / n_xylen;
  ref1[1] = -dir[1] * dir[2] / n_xylen;
  ref1[2] = (1 - pow2(dir[2])) / n_xylen;
  cross_prod(ref1, dir, ref2);
}
/* lorentz vectors.h:233 */
static void lorentz(const double elec_rf[3], const double n_rf[3], const double v[3], double elec_cmf[3]) {
  const double beta[3] = {v[0] / CLIGHT, v[1] / CLIGHT, v[2] / CLIGHT};
  const double betasquared = dot3(beta, beta);
  if (betasquared == 0.) {
    elec_cmf[0] = elec_rf[0]; elec_cmf[1] = elec_rf[1]; elec_cmf[2] = elec_rf[2];
    return;
  }
  const double gamma_rel = 1. / sqrt(1 - betasquared);
  const double edb = dot3(elec_rf, beta);
  const double elec_par[3] = {edb * beta[0] / betasquared, edb * beta[1] / betasquared, edb * beta[2] / betasquared};
  const double elec_perp[3] = {elec_rf[0] - elec_par[0], elec_rf[1] - elec_par[1], elec_rf[2] - elec_par[2]};
  double b_rf[3], v_cross_b[3];
  cross_prod(n_rf, elec_rf, b_rf);
  cross_prod(beta, b_rf, v_cross_b);
  const double tmp[3] = {elec_par[0] + (gamma_rel * (elec_perp[0] + v_cross_b[0])),
                         elec_par[1] + (gamma_rel * (elec_perp[1] + v_cross_b[1])),
                         elec_par[2] + (gamma_rel * (elec_perp[2] + v_cross_b[2]))};
  vec_norm3(tmp, elec_cmf);
}
/* frame_transform vectors.h:266 */
static void frame_transform(const double n_rf[3], double q0, double u0, const double v[3], double n_cmf[3],
                            double *q_cmf, double *u_cmf) {
  double ref1_rf[3], ref2_rf[3];
  meridian(n_rf, ref1_rf, ref2_rf);
  const double p = sqrt(pow2(q0) + pow2(u0));
  double rot_angle = 0;
  if (p > 0) {
    const double pol_angle = atan2(u0, q0);
    rot_angle = (pol_angle < 0 ? pol_angle + (2. * PI) : pol_angle) / 2.;
  }
  const double cos_rot = cos(rot_angle);
  const double sin_rot = sin(rot_angle);
  const double elec_rf[3] = {(cos_rot * ref1_rf[0]) - (sin_rot * ref2_rf[0]), (cos_rot * ref1_rf[1]) - (sin_rot * ref2_rf[1]),
                             (cos_rot * ref1_rf[2]) - (sin_rot * ref2_rf[2])};
  angle_ab(n_rf, v, n_cmf);
  double elec_cmf[3];
  lorentz(elec_rf, n_rf, v, elec_cmf);
  double ref1_cmf[3], ref2_cmf[3];
  meridian(n_cmf, ref1_cmf, ref2_cmf);
  const double c1 = dot3(elec_cmf, ref1_cmf);
  const double c2 = dot3(elec_cmf, ref2_cmf);
  double theta_rot = atan2(-c2, c1);
  if (theta_rot < 0) theta_rot += 2 * PI;
  *q_cmf = cos(2 * theta_rot) * p;
  *u_cmf = sin(2 * theta_rot) * p;
}
/* scatter_polarisation_to_rf vectors.h:325 */
static void scatter_polarisation_to_rf(const double old_dir_cmf[3], const double new_dir_cmf[3], double q_i_cmf,
                                       double u_i_cmf, const double vel_vec[3], double new_dir_rf[3], double *q_rf,
                                       double *u_rf) {
  double ref1_old[3], ref2_old[3];
  meridian(old_dir_cmf, ref1_old, ref2_old);
  const double i1 = get_rot_angle(old_dir_cmf, new_dir_cmf, ref1_old, ref2_old);
  const double cos2i1 = cos(2 * i1);
  const double sin2i1 = sin(2 * i1);
  const double q_old = (q_i_cmf * cos2i1) - (u_i_cmf * sin2i1);
  const double u_old = (q_i_cmf * sin2i1) + (u_i_cmf * cos2i1);
  const double mu = dot3(old_dir_cmf, new_dir_cmf);
  const double musquared = pow2(mu);
  const double I_new = 0.75 * ((musquared + 1.) + (q_old * (musquared - 1.)));
  const double q_new = (0.75 * ((musquared - 1.) + (q_old * (musquared + 1.)))) / I_new;
  const double u_new = (1.5 * mu * u_old) / I_new;
  double ref1[3], ref2[3];
  meridian(new_dir_cmf, ref1, ref2);
  const double i2 = PI + get_rot_angle(new_dir_cmf, old_dir_cmf, ref1, ref2);
  const double cos2i2 = cos(2 * i2);
  const double sin2i2 = sin(2 * i2);
  const double q_cmf = (q_new * cos2i2) + (u_new * sin2i2);
  const double u_cmf = (-q_new * sin2i2) + (u_new * cos2i2);
  const double negvel[3] = {-vel_vec[0], -vel_vec[1], -vel_vec[2]};
  frame_transform(new_dir_cmf, q_cmf, u_cmf, negvel, new_dir_rf, q_rf, u_rf);
}

/* ------------------------------------------------------------------ atomic.h accessors */
static inline double stat_weight(const Oracle *o, int ul) { return o->m->level_statweight[ul]; } /* atomic.h:270 */
static inline double epsilon(const Oracle *o, int ul) { return o->m->level_epsilon[ul]; }        /* atomic.h:281 */
static inline int uniqueion(const Oracle *o, int element, int ion) { return o->m->elem_uniqueionindexstart[element] + ion; }
static inline int ionlevelstart(const Oracle *o, int element, int ion) { /* atomic.h:88 */
  return o->m->ion_uniquelevelindexstart[uniqueion(o, element, ion)];
}
static inline int get_ionstage(const Oracle *o, int element, int ion) { return o->m->elem_lowest_ionstage[element] + ion; }
static inline int get_nions(const Oracle *o, int element) { return o->m->elem_nions[element]; }
static inline int get_nlevels(const Oracle *o, int element, int ion) { return o->m->ion_nlevels[uniqueion(o, element, ion)]; }
static inline int get_nlevels_ionising(const Oracle *o, int element, int ion) {
  return o->m->ion_nlevels_ionising[uniqueion(o, element, ion)];
}
static inline int get_phixsupperlevel(const Oracle *o, int ul, int t) { /* atomic.h:155 */
  return o->m->allphixstargets_levelindex[o->m->level_phixstargetstart[ul] + t];
}
static inline double get_phixsprobability(const Oracle *o, int ul, int t) { /* atomic.h:168 */
  return o->m->allphixstargets_probability[o->m->level_phixstargetstart[ul] + t];
}
static inline const float *get_phixs_table(const Oracle *o, int ul) { /* atomic.h:188 */
  return o->m->allphixs + ((ptrdiff_t)o->m->level_phixsstart[ul] * o->m->NPHIXSPOINTS);
}
static inline int find_phixstargetindex(const Oracle *o, int ul, int upperionlevel) { /* atomic.h:493 */
  const int n = o->m->level_nphixstargets[ul];
  for (int t = 0; t < n; t++)
    if (upperionlevel == get_phixsupperlevel(o, ul, t)) return t;
  return -1;
}
/* get_phixs_threshold atomic.h:534 */
static inline double get_phixs_threshold(const Oracle *o, int element, int ion, int level, int t) {
  const int ul = ionlevelstart(o, element, ion) + level;
  const int upperlevel = get_phixsupperlevel(o, ul, t);
  return epsilon(o, ionlevelstart(o, element, ion + 1) + upperlevel) - epsilon(o, ul);
}
static inline int get_emtype_continuum(const Oracle *o, int ul, int t) { /* atomic.h:508 */
  return -1 - o->m->level_bflist_start[ul] - t;
}
/* photoionisation_crosssection_fromtable atomic.h:201 */
static float photoionisation_crosssection_fromtable(const Oracle *o, const float *xs, double nu_edge, double nu) {
  const int NP = o->m->NPHIXSPOINTS;
  const double INC = o->m->NPHIXSNUINCREMENT;
  float sigma_bf = 0.;
#if ARTIS_OPT_PHIXS_CLASSIC_NO_INTERPOLATION
  if (nu < nu_edge) {
    sigma_bf = 0.;
  } else if (nu == nu_edge) {
    sigma_bf = xs[0];
  } else if (nu < nu_edge * (1 + (INC * NP))) {
    int i = (int)((nu - nu_edge) / (INC * nu_edge));
    if (NP - 1 < i) i = NP - 1;
    sigma_bf = xs[i];
  } else {
    sigma_bf = (float)(xs[NP - 1] * pow(nu_edge * (1 + (INC * NP)) / nu, 3));
  }
  return sigma_bf;
#else
  const double ireal = ((nu / nu_edge) - 1.0) / INC;
  const int i = (int)floor(ireal);
  if (i < 0) {
    sigma_bf = 0.;
  } else if (i < NP - 1) {
    const double a = xs[i];
    const double b = xs[i + 1];
    const double fb = ireal - i;
    sigma_bf = (float)(((1. - fb) * a) + (fb * b));
  } else {
    const double nu_max_phixs = nu_edge * o->last_phixs_nuovernuedge;
    sigma_bf = (float)(xs[NP - 1] * pow3(nu_max_phixs / nu));
  }
  return sigma_bf;
#endif
}

/* std::ranges::upper_bound / lower_bound on ascending double arrays */
static int upper_bound_d(const double *a, int n, double v) { /* first a[i] > v */
  int lo = 0, len = n;
  while (len > 0) {
    int half = len / 2;
    if (!(v < a[lo + half])) { lo += half + 1; len -= half + 1; } else { len = half; }
  }
  return lo;
}
static int lower_bound_d(const double *a, int n, double v) { /* first a[i] >= v */
  int lo = 0, len = n;
  while (len > 0) {
    int half = len / 2;
    if (a[lo + half] < v) { lo += half + 1; len -= half + 1; } else { len = half; }
  }
  return lo;
}

/* ------------------------------------------------------------------ grid accessors */
static inline int propcell_nonemptymgi(const Oracle *o, int cellindex) { return o->m->propcell_nonemptymgi[cellindex]; }
static inline int coordstride(const Oracle *o, int axis) { /* grid.cc:200 */
  int stride = 1;
  for (int a = 0; a < axis; ++a) stride *= o->m->ncoordgrid[a];
  return stride;
}
static inline int cellcoordindex(const Oracle *o, int cellindex, int axis) { /* grid.cc:209 */
  return (cellindex / coordstride(o, axis)) % o->m->ncoordgrid[axis];
}
static inline double cellcoordmin(const Oracle *o, int cellindex, int axis) { /* grid.cc:215 */
  return o->m->coord_pos_min_tmin[axis][cellcoordindex(o, cellindex, axis)];
}
static inline double cellcoordmax(const Oracle *o, int cellindex, int axis) { /* grid.cc:221 */
  const int idx = cellcoordindex(o, cellindex, axis);
  return idx < o->m->ncoordgrid[axis] - 1 ? o->m->coord_pos_min_tmin[axis][idx + 1] : o->m->rmax;
}
static inline double cellbound_tolerance(double boundarypos) { return dmax(10., fabs(boundarypos) * 1e-12); } /* grid.cc:1530 */
/* is_boundary_overshoot_within_tolerance grid.cc:1542 */
static int overshoot_within_tol(const Oracle *o, int upper, double pktpos, double pktvel, double boundarypos_tmin, double tstart) {
  const double boundaryvel = boundarypos_tmin / o->m->tmin;
  const double boundarypos = boundaryvel * tstart;
  const double overshoot = upper ? (pktpos - boundarypos) : (boundarypos - pktpos);
  const int movingtowards = upper ? (pktvel > boundaryvel) : (pktvel < boundaryvel);
  return movingtowards && (overshoot >= 0.) && (overshoot <= cellbound_tolerance(boundarypos));
}
/* distance_cartesian_boundary grid.cc:1518 */
static double distance_cartesian_boundary(const Oracle *o, double pktpos, double pktvel, double cellboundarypos, double tstart) {
  return CLIGHT_PROP * (pktpos - (cellboundarypos / o->m->tmin * tstart)) / ((cellboundarypos / o->m->tmin) - pktvel);
}
/* expanding_shell_intersection grid.cc:1413, S1 == 3 */
static double expanding_shell_intersection(int lower, const double pos[3], const double dir[3], double speed,
                                           double shellradiuststart, double tstart) {
  const double a = dot3(dir, dir) - pow2(shellradiuststart / tstart / speed);
  const double b = 2 * (dot3(dir, pos) - (pow2(shellradiuststart) / tstart / speed));
  const double c = dot3(pos, pos) - pow2(shellradiuststart);
  const double discriminant = pow2(b) - (4 * a * c);
  if (discriminant < 0) return -1;
  if (discriminant > 0) {
    double dist1 = (-b + sqrt(discriminant)) / 2 / a;
    double dist2 = (-b - sqrt(discriminant)) / 2 / a;
    double posf1[3], posf2[3];
    for (int d = 0; d < 3; d++) {
      posf1[d] = pos[d] + (dist1 * dir[d]);
      posf2[d] = pos[d] + (dist2 * dir[d]);
    }
    const double v_rad_shell = shellradiuststart / tstart;
    const double v_rad_final1 = dot3(dir, posf1) * speed / vec_len3(posf1);
    const double v_rad_final2 = dot3(dir, posf2) * speed / vec_len3(posf2);
    if (lower) {
      if (v_rad_final1 > v_rad_shell) dist1 = -1;
      if (v_rad_final2 > v_rad_shell) dist2 = -1;
    } else {
      if (v_rad_final1 < v_rad_shell) dist1 = -1;
      if (v_rad_final2 < v_rad_shell) dist2 = -1;
    }
    if (dist1 < 0 && dist2 < 0) return -1;
    if (dist2 < 0) return dist1;
    if (dist1 < 0) return dist2;
    return dmin(dist1, dist2);
  }
  return -1.;
}

/* boundary_distance grid.cc:2480 (TESTMODE and FORCE_SPHERICAL_ESCAPE_SURFACE off) */
static double boundary_distance(Oracle *o, const double dir[3], const double pos[3], double tstart, int cellindex,
                                int *next_cellindex_out) {
  const artis_model *m = o->m;
  const double tmin = m->tmin;
  double distance = DBL_MAXV;
  int next_cellindex = -1;

  if (m->gridtype == ARTIS_GRID_SPHERICAL1D) {
    const double pktpos0 = vec_len3(pos);                           /* get_gridcoords_from_xyz grid.cc:1374 */
    const double pktvel0 = dot3(pos, dir) / pktpos0 * CLIGHT_PROP;  /* grid.cc:1400 */
    const int idx0 = cellcoordindex(o, cellindex, 0);
    const double cmin = cellcoordmin(o, cellindex, 0);
    const double cmax = cellcoordmax(o, cellindex, 0);
    const double speed = vec_len3(dir) * CLIGHT_PROP;
    const double r_outer = cmax * tstart / tmin;
    const double d_max = overshoot_within_tol(o, 1, pktpos0, pktvel0, cmax, tstart)
                             ? 0.
                             : expanding_shell_intersection(0, pos, dir, speed, r_outer, tstart);
    if ((d_max >= 0.) && (d_max < distance)) {
      distance = d_max;
      next_cellindex = (idx0 == (m->ncoordgrid[0] - 1)) ? -99 : cellindex + coordstride(o, 0);
    }
    const double r_inner = cmin * tstart / tmin;
    if (r_inner > 0.) {
      const double d_min = overshoot_within_tol(o, 0, pktpos0, pktvel0, cmin, tstart)
                               ? 0.
                               : expanding_shell_intersection(1, pos, dir, speed, r_inner, tstart);
      if ((d_min >= 0.) && (d_min < distance)) {
        distance = d_min;
        next_cellindex = (idx0 == 0) ? -99 : cellindex - coordstride(o, 0);
      }
    }
  } else if (m->gridtype == ARTIS_GRID_CARTESIAN3D) {
    for (int d = 0; d < 3; d++) {
      const double pktpos = pos[d];
      const double pktvel = dir[d] * CLIGHT_PROP; /* grid.cc:1391 */
      const int idx = cellcoordindex(o, cellindex, d);
      const double cmin = cellcoordmin(o, cellindex, d);
      const double cmax = cellcoordmax(o, cellindex, d);
      if (pktvel > (cmax / tmin)) {
        const double dd = overshoot_within_tol(o, 1, pktpos, pktvel, cmax, tstart)
                              ? 0.
                              : distance_cartesian_boundary(o, pktpos, pktvel, cmax, tstart);
        if ((dd >= 0.) && (dd < distance)) {
          distance = dd;
          next_cellindex = (idx == (m->ncoordgrid[d] - 1)) ? -99 : cellindex + coordstride(o, d);
        }
      } else if (pktvel < (cmin / tmin)) {
        const double dd = overshoot_within_tol(o, 0, pktpos, pktvel, cmin, tstart)
                              ? 0.
                              : distance_cartesian_boundary(o, pktpos, pktvel, cmin, tstart);
        if ((dd >= 0.) && (dd < distance)) {
          distance = dd;
          next_cellindex = (idx == 0) ? -99 : cellindex - coordstride(o, d);
        }
      }
    }
  } else if (m->gridtype == ARTIS_GRID_CYLINDRICAL2D) {
    /* grid.cc:2602-2695: coordinate 0 is the cylindrical radius, coordinate 1 is z */
    const double posnoz[3] = {pos[0], pos[1], 0.}; /* the reference's 2-vectors; a zero third component changes no sum */
    const double pktpos0 = sqrt(pow2(pos[0]) + pow2(pos[1]));                                   /* grid.cc:1377 */
    const double pktvel0 = ((pos[0] * dir[0]) + (pos[1] * dir[1])) / pktpos0 * CLIGHT_PROP;     /* grid.cc:1394 */
    const double pktpos1 = pos[2];
    const double pktvel1 = dir[2] * CLIGHT_PROP;
    const int idx0 = cellcoordindex(o, cellindex, 0);
    const double cmin0 = cellcoordmin(o, cellindex, 0);
    const double cmax0 = cellcoordmax(o, cellindex, 0);
    const double dirxylen = sqrt(pow2(dir[0]) + pow2(dir[1]));
    const double xyspeed = dirxylen * CLIGHT_PROP;
    if (dirxylen > 0.) {
      const double dirnoz[3] = {dir[0] / dirxylen, dir[1] / dirxylen, 0.};
      const double r_outer = cmax0 * tstart / tmin;
      const double d_rcyl_max = overshoot_within_tol(o, 1, pktpos0, pktvel0, cmax0, tstart)
                                    ? 0.
                                    : expanding_shell_intersection(0, posnoz, dirnoz, xyspeed, r_outer, tstart);
      if (d_rcyl_max >= 0.) {
        const double d_z = d_rcyl_max / xyspeed * dir[2] * CLIGHT_PROP;
        const double dd = sqrt(pow2(d_rcyl_max) + pow2(d_z));
        if ((dd >= 0.) && (dd < distance)) {
          distance = dd;
          next_cellindex = (idx0 == (m->ncoordgrid[0] - 1)) ? -99 : cellindex + coordstride(o, 0);
        }
      }
      const double r_inner = cmin0 * tstart / tmin;
      if (r_inner > 0) {
        const double d_rcyl_min = overshoot_within_tol(o, 0, pktpos0, pktvel0, cmin0, tstart)
                                      ? 0.
                                      : expanding_shell_intersection(1, posnoz, dirnoz, xyspeed, r_inner, tstart);
        if (d_rcyl_min >= 0.) {
          const double d_z = d_rcyl_min / xyspeed * dir[2] * CLIGHT_PROP;
          const double dd = sqrt(pow2(d_rcyl_min) + pow2(d_z));
          if ((dd >= 0.) && (dd < distance)) {
            distance = dd;
            next_cellindex = (idx0 == 0) ? -99 : cellindex - coordstride(o, 0);
          }
        }
      }
    } else {
      /* moving exactly along z: only the expanding inner r_cyl boundary can catch up with the packet (grid.cc:2654) */
      const double rcyl_inner_tmin = cmin0;
      if (rcyl_inner_tmin > 0.) {
        const double dd = overshoot_within_tol(o, 0, pktpos0, pktvel0, cmin0, tstart)
                              ? 0.
                              : ((pktpos0 * tmin / rcyl_inner_tmin) - tstart) * CLIGHT_PROP;
        if ((dd >= 0.) && (dd < distance)) {
          distance = dd;
          next_cellindex = (idx0 == 0) ? -99 : cellindex - coordstride(o, 0);
        }
      }
    }
    { /* z boundaries are Cartesian (grid.cc:2671) */
      const int d = 1;
      const int idx = cellcoordindex(o, cellindex, d);
      const double cmin = cellcoordmin(o, cellindex, d);
      const double cmax = cellcoordmax(o, cellindex, d);
      if (pktvel1 > (cmax / tmin)) {
        const double dd = overshoot_within_tol(o, 1, pktpos1, pktvel1, cmax, tstart)
                              ? 0.
                              : distance_cartesian_boundary(o, pktpos1, pktvel1, cmax, tstart);
        if ((dd >= 0.) && (dd < distance)) {
          distance = dd;
          next_cellindex = (idx == (m->ncoordgrid[d] - 1)) ? -99 : cellindex + coordstride(o, d);
        }
      } else if (pktvel1 < (cmin / tmin)) {
        const double dd = overshoot_within_tol(o, 0, pktpos1, pktvel1, cmin, tstart)
                              ? 0.
                              : distance_cartesian_boundary(o, pktpos1, pktvel1, cmin, tstart);
        if ((dd >= 0.) && (dd < distance)) {
          distance = dd;
          next_cellindex = (idx == 0) ? -99 : cellindex - coordstride(o, d);
        }
      }
    }
  } else {
    ORACLE_FAIL(o, "gridtype not supported by the oracle");
  }

  if (!((next_cellindex == -99) || ((next_cellindex >= 0) && (next_cellindex < m->ngrid))))
    ORACLE_FAIL(o, "boundary_distance: bad next_cellindex");
  if (!(distance >= 0.)) ORACLE_FAIL(o, "boundary_distance: negative distance");

  if (distance > o->ts.max_path_step) { /* grid.cc:2750 */
    *next_cellindex_out = cellindex;
    return o->ts.max_path_step;
  }
  *next_cellindex_out = next_cellindex;
  return distance;
}

/* snap_pos_to_cell grid.cc:2460 */
static void snap_pos_to_cell(const Oracle *o, double pos[3], double time, int cellindex) {
  if (o->m->gridtype != ARTIS_GRID_CARTESIAN3D) return;
  for (int d = 0; d < 3; d++) {
    const int idx = cellcoordindex(o, cellindex, d);
    const double cellposmin = o->m->coord_pos_min_tmin[d][idx] / o->m->tmin * time;
    const double cellposmax = (idx < (o->m->ncoordgrid[d] - 1)) ? o->m->coord_pos_min_tmin[d][idx + 1] / o->m->tmin * time
                                                                 : cellcoordmax(o, cellindex, d) / o->m->tmin * time;
    pos[d] = dclamp(pos[d], cellposmin, cellposmax);
  }
}
/* change_cell_or_escape grid.h:118 */
static void change_cell_or_escape(Oracle *o, artis_packet *p, int next_cellindex) {
  if (next_cellindex >= 0) {
    if (next_cellindex != p->cellindex) snap_pos_to_cell(o, p->pos, p->prop_time, next_cellindex);
    p->cellindex = next_cellindex;
    stat_inc(o, ARTIS_STAT_CELLCROSSINGS);
  } else {
    p->escape_type = p->type;
    p->escape_time = (float)p->prop_time;
    p->type = ARTIS_TYPE_ESCAPE;
    stat_inc(o, ARTIS_STAT_PKTESCAPES);
  }
}

/* ------------------------------------------------------------------ cell state accessors */
static inline float cell_nne(const Oracle *o, int c) { return o->cs->nne[c]; }
static inline float cell_clumpednne(const Oracle *o, int c) { return o->cs->clumpfactor[c] * o->cs->nne[c]; } /* float product */
/* get_groundlevelpop ltepop.h:74 */
static double get_groundlevelpop(const Oracle *o, int c, int element, int ion) {
  const double nn = o->cs->ion_groundlevelpops[((ptrdiff_t)c * o->m->nions) + uniqueion(o, element, ion)];
  if (nn < ARTIS_OPT_MINPOP) {
    if (o->cs->elem_massfracs[((ptrdiff_t)c * o->m->nelements) + element] > 0) return ARTIS_OPT_MINPOP;
    return 0.;
  }
  return nn;
}
/* get_nnion ltepop.h:106 */
static double get_nnion(const Oracle *o, int c, int element, int ion) {
  return get_groundlevelpop(o, c, element, ion) *
         o->cs->ion_partfuncts[((ptrdiff_t)c * o->m->nions) + uniqueion(o, element, ion)] /
         stat_weight(o, ionlevelstart(o, element, ion));
}
#if ARTIS_OPT_NT_ON || ARTIS_OPT_USE_XCOM_GAMMAPHOTOION
/* grid::get_elem_numberdens grid.cc:1693 (float mass fraction / double(float mean weight) * float rho) */
static double get_elem_numberdens(const Oracle *o, int c, int element) {
#if ARTIS_OPT_USE_CALCULATED_MEANATOMICWEIGHT /* grid::get_element_meanweight grid.cc:1509-1515 */
  const float mu = o->cs->elem_meanweight[((ptrdiff_t)c * o->m->nelements) + element];
#else
  const float mu = o->m->elem_meannucmass[element];
#endif
  return o->cs->elem_massfracs[((ptrdiff_t)c * o->m->nelements) + element] / (double)mu * o->cs->rho[c];
}
#endif
#if ARTIS_OPT_NT_ON
/* ---- non-thermal channels (nonthermal.cc), read from the Spencer-Fano solution the host hands over ---- */
#define QE 4.80325E-10 /* constants.h:31 */
#define NT_NAUGER (ARTIS_OPT_NT_MAX_AUGER_ELECTRONS + 1)
/* get_nnion_tot atomic.h:51 */
static double get_nnion_tot(const Oracle *o, int c) {
  double nntot = 0.;
  for (int element = 0; element < o->m->nelements; element++) nntot += get_elem_numberdens(o, c, element);
  return nntot;
}
/* nt_ionisation_maxupperion nonthermal.cc:2435 (NT_SOLVE_SPENCERFANO) */
static int nt_ionisation_maxupperion(const Oracle *o, int element, int lowerion) {
  const int nions = get_nions(o, element);
  int maxupper = lowerion + 1 + ARTIS_OPT_NT_MAX_AUGER_ELECTRONS;
  if (nions - 1 < maxupper) maxupper = nions - 1;
  return maxupper;
}
/* nt_ionisation_upperion_probability nonthermal.cc:2398 */
static double nt_ionisation_upperion_probability(Oracle *o, int c, int element, int lowerion, int upperion, int energyweighted) {
  const int numaugerelec = upperion - lowerion - 1;
  const float *prob = (energyweighted ? o->cs->nt_ionenfrac_num_auger : o->cs->nt_prob_num_auger) +
                      ((((ptrdiff_t)c * o->m->nions) + uniqueion(o, element, lowerion)) * NT_NAUGER);
  if (numaugerelec < ARTIS_OPT_NT_MAX_AUGER_ELECTRONS) return prob[numaugerelec];
  if (numaugerelec == ARTIS_OPT_NT_MAX_AUGER_ELECTRONS) {
    double prob_remaining = 1.;
    for (int a = 0; a < ARTIS_OPT_NT_MAX_AUGER_ELECTRONS; a++) prob_remaining -= prob[a];
    if (!(fabs(prob_remaining - prob[numaugerelec]) < 0.001)) ORACLE_FAIL(o, "Auger probabilities do not sum to one");
    return prob_remaining;
  }
  return 0.;
}
/* nt_random_upperion nonthermal.cc:2450 */
static int nt_random_upperion(Oracle *o, int c, int element, int lowerion, int energyweighted, uint32_t *rngstate) {
  const double zrand = rng_uniform(rngstate);
  double prob_sum = 0.;
  const int maxupper = nt_ionisation_maxupperion(o, element, lowerion);
  for (int upperion = lowerion + 1; upperion <= maxupper; upperion++) {
    prob_sum += nt_ionisation_upperion_probability(o, c, element, lowerion, upperion, energyweighted);
    if (zrand < prob_sum) return upperion;
  }
  if (!(prob_sum > 0.99)) ORACLE_FAIL(o, "nt_random_upperion: probabilities sum below 0.99");
  return maxupper;
}
/* get_oneoverw_approx_axelrod nonthermal.cc:1207 */
static double get_oneoverw_approx_axelrod(const Oracle *o, int element, int ion, int c) {
  double nntot = 0., Zbar = 0.;
  for (int ielement = 0; ielement < o->m->nelements; ielement++) {
    const double nnelement = get_elem_numberdens(o, c, ielement);
    Zbar += nnelement * o->m->elem_anumber[ielement];
    nntot += nnelement;
  }
  if (nntot > 0) Zbar /= nntot;
  const double binding = o->m->ion_nt_sum_q_over_binding[uniqueion(o, element, ion)];
  const double Aconst = 1.33e-14 * EV * EV;
  return Aconst * binding / Zbar / (2 * PI * (pow2(QE) * pow2(QE)));
}
/* nt_ionisation_ratecoeff nonthermal.cc:2478 with nt_ionisation_ratecoeff_sf :1420 and _wfapprox :1251 */
static double nt_ionisation_ratecoeff(const Oracle *o, int c, int element, int ion) {
  const double deposition_rate_density = o->cs->nt_deposition_rate_density[c];
  double Y_nt = 0.;
  if (deposition_rate_density > 0.)
    Y_nt = deposition_rate_density / get_nnion_tot(o, c) / o->cs->nt_eff_ionpot[((ptrdiff_t)c * o->m->nions) + uniqueion(o, element, ion)];
  if (!isfinite(Y_nt)) return deposition_rate_density / get_nnion_tot(o, c) * get_oneoverw_approx_axelrod(o, element, ion, c);
  return Y_nt;
}
/* nt_excitation_ratecoeff nonthermal.cc:2496 */
static double nt_excitation_ratecoeff(const Oracle *o, int c, int lowerlevel, int upperlevel, int alltransindex) {
  if (!ARTIS_OPT_NT_EXCITATION_ON) return 0.;
  if (lowerlevel >= ARTIS_OPT_NTEXCITATION_MAXNLEVELS_LOWER) return 0.;
  if (upperlevel >= ARTIS_OPT_NTEXCITATION_MAXNLEVELS_UPPER) return 0.;
  const ptrdiff_t base = (ptrdiff_t)c * o->cs->nt_excitations_stored;
  const int32_t *ati = o->cs->nt_exc_alltransindex + base;
  int lo = 0, hi = o->cs->nt_exc_count[c]; /* std::ranges::lower_bound */
  const int n = hi;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) / 2);
    if (ati[mid] < alltransindex) lo = mid + 1; else hi = mid;
  }
  if (lo == n || ati[lo] != alltransindex) return 0.;
  return o->cs->nt_exc_ratecoeffperdeposition[base + lo] * o->cs->nt_deposition_rate_density[c];
}
static double get_nnion(const Oracle *o, int c, int element, int ion);
/* ion_ntion_energyrate nonthermal.cc:1509 */
static double ion_ntion_energyrate(Oracle *o, int c, int element, int lowerion) {
  const double nnlowerion = get_nnion(o, c, element, lowerion);
  double enrate = 0.;
  const int maxupperion = nt_ionisation_maxupperion(o, element, lowerion);
  for (int upperion = lowerion + 1; upperion <= maxupperion; upperion++) {
    const double upperionprobfrac = nt_ionisation_upperion_probability(o, c, element, lowerion, upperion, 0);
    const double epsilon_trans = epsilon(o, ionlevelstart(o, element, upperion)) - epsilon(o, ionlevelstart(o, element, lowerion));
    enrate += nnlowerion * upperionprobfrac * epsilon_trans;
  }
  return nt_ionisation_ratecoeff(o, c, element, lowerion) * enrate;
}
/* select_nt_ionisation nonthermal.cc:1537 with get_ntion_energyrate :1524; returns 0 when no ion can be selected */
static int select_nt_ionisation(Oracle *o, int c, uint32_t *rngstate, int *element_out, int *lowerion_out) {
  double ratetotal = 0.;
  for (int ielement = 0; ielement < o->m->nelements; ielement++)
    for (int ilowerion = 0; ilowerion < get_nions(o, ielement) - 1; ilowerion++) ratetotal += ion_ntion_energyrate(o, c, ielement, ilowerion);
  if (!(ratetotal > 0.)) return 0;
  const double zrand = rng_uniform(rngstate);
  double ratesum = 0.;
  for (int ielement = 0; ielement < o->m->nelements; ielement++) {
    for (int ilowerion = 0; ilowerion < get_nions(o, ielement) - 1; ilowerion++) {
      ratesum += ion_ntion_energyrate(o, c, ielement, ilowerion);
      if (ratesum > zrand * ratetotal) {
        *element_out = ielement;
        *lowerion_out = ilowerion;
        return 1;
      }
    }
  }
  ORACLE_FAIL(o, "select_nt_ionisation: nothing selected");
  return 0;
}
#endif

/* calculate_levelpop ltepop.cc:412 via calculate_levelpop_nominpop ltepop.cc:170 (no NLTE levels in classic)
 * and calculate_levelpop_boltzmann ltepop.cc:395 */
static double calculate_levelpop(const Oracle *o, int c, int element, int ion, int level) {
  if (o->cs->levelpops) /* the host's NLTE / LTE solution, get_levelpop ltepop.cc:169 */
    return o->cs->levelpops[((ptrdiff_t)c * o->m->nlevels) + ionlevelstart(o, element, ion) + level];
  double nn;
  const double nnground = get_groundlevelpop(o, c, element, ion);
  if (level == 0) {
    nn = nnground;
  } else {
    const float T_exc = ARTIS_OPT_LTEPOP_EXCITATION_USE_TJ ? o->cs->TJ[c] : o->cs->Te[c];
    const int start = ionlevelstart(o, element, ion);
    const double E_aboveground = epsilon(o, start + level) - epsilon(o, start);
    nn = (nnground * stat_weight(o, start + level) / stat_weight(o, start) * exp(-E_aboveground / KB / T_exc));
  }
  if (nn < ARTIS_OPT_MINPOP) {
    if (o->cs->elem_massfracs[((ptrdiff_t)c * o->m->nelements) + element] > 0) return ARTIS_OPT_MINPOP;
    return 0.;
  }
  return nn;
}

/* radfield::planck radfield.h:50 */
static inline double planck(double nu, double T) { return 2 * H_PLANCK * pow3(nu) / pow2(CLIGHT) / expm1(HOVERKB * nu / T); }
/* multibin radiation field: get_bin_nu_upper radfield.cc:118, select_bin :138 (get_linearbinindex sn3d.h:115) */
#define RADBIN_DELTA_NU ((ARTIS_OPT_RADFIELDBINS_NU_MAX - ARTIS_OPT_RADFIELDBINS_NU_MIN) / (ARTIS_OPT_RADFIELDBINCOUNT - 1))
static double radbin_nu_upper(int binindex) {
  if (binindex == ARTIS_OPT_RADFIELDBINCOUNT - 1) return ARTIS_OPT_RADFIELDBINS_T_E_SUPERBIN_NU_MAX;
  return ARTIS_OPT_RADFIELDBINS_NU_MIN + ((binindex + 1) * RADBIN_DELTA_NU);
}
static int radbin_select(double nu) {
  if (nu < ARTIS_OPT_RADFIELDBINS_NU_MIN) return -2;
  if (nu >= ARTIS_OPT_RADFIELDBINS_T_E_SUPERBIN_NU_MAX) return -1;
  if (nu >= ARTIS_OPT_RADFIELDBINS_NU_MAX) return ARTIS_OPT_RADFIELDBINCOUNT - 1;
  const double fracindex = (nu - ARTIS_OPT_RADFIELDBINS_NU_MIN) / RADBIN_DELTA_NU;
  const ptrdiff_t truncated = (ptrdiff_t)fracindex;
  const int binindex = (int)((fracindex < (double)truncated) ? truncated - 1 : truncated);
  if (nu == radbin_nu_upper(binindex)) return binindex + 1;
  return binindex;
}
/* radfield::radfield radfield.cc:786 */
static inline double radfield(const Oracle *o, double nu, int c) {
#if ARTIS_OPT_MULTIBIN_RADFIELD_MODEL_ON
  if (o->ts.nts >= ARTIS_OPT_FIRST_NLTE_RADFIELD_TIMESTEP) {
    const int binindex = radbin_select(nu);
    if (binindex >= 0) {
      const float W = o->cs->radfieldbin_W[((ptrdiff_t)c * ARTIS_OPT_RADFIELDBINCOUNT) + binindex];
      if (W >= 0.) return W * planck(nu, o->cs->radfieldbin_T_R[((ptrdiff_t)c * ARTIS_OPT_RADFIELDBINCOUNT) + binindex]);
    }
    return 0.;
  }
#endif
  return o->cs->W[c] * planck(nu, o->cs->TR[c]);
}

/* ------------------------------------------------------------------ ratecoeff.cc lookups */
/* get_temperature_gridupperindex ratecoeff.cc:54 */
static int get_temperature_gridupperindex(const Oracle *o, double temperature) {
  const int gridsize = ARTIS_OPT_TABLESIZE + 1;
  int index = (int)(log(temperature / ARTIS_OPT_MINTEMP) / o->T_step_log) + 1;
  if (index < 0) index = 0;
  if (index > gridsize) index = gridsize;
  while (index > 0 && o->temperature_grid[index - 1] > temperature) index--;
  while (index < gridsize && o->temperature_grid[index] <= temperature) index++;
  return index;
}
/* lerp_or_last ratecoeff.cc:524; temperature arrives as the reference's float */
static double lerp_or_last(const Oracle *o, const double *table, int ul, int t, float temperature) {
  const int contindex = o->m->level_bflist_start[ul] + t; /* get_bflutindex ratecoeff.cc:123 */
  const double *row = table + ((ptrdiff_t)contindex * ARTIS_OPT_TABLESIZE);
  const int upperindex = get_temperature_gridupperindex(o, temperature);
  if (upperindex == 0) return row[0];
  if (upperindex < ARTIS_OPT_TABLESIZE) {
    const double T_lower = o->temperature_grid[upperindex - 1];
    const double T_upper = o->temperature_grid[upperindex];
    const double f_lower = row[upperindex - 1];
    const double f_upper = row[upperindex];
    return (f_lower + ((f_upper - f_lower) / (T_upper - T_lower) * (temperature - T_lower)));
  }
  return row[ARTIS_OPT_TABLESIZE - 1];
}
static inline double get_spontrecombcoeff(const Oracle *o, int ul, int t, float T_e) { /* ratecoeff.cc:679 */
  return lerp_or_last(o, o->m->spontrecombcoeffs, ul, t, T_e);
}
static inline double get_bfcoolingcoeff(const Oracle *o, int ul, int t, float T_e) { /* ratecoeff.cc:833 */
  return lerp_or_last(o, o->m->bfcooling_coeffs, ul, t, T_e);
}
/* get_corrphotoioncoeff ratecoeff.cc:840, USE_LUT_PHOTOION branch, uncached value */
static double calc_corrphotoioncoeff(const Oracle *o, int c, int ul, int t) {
#if !ARTIS_OPT_USE_LUT_PHOTOION
  /* USE_LUT_PHOTOION off: the estimator-based / integrated coefficient is host data (include/artis_amd.h) */
  return o->cs->corrphotoioncoeff[((ptrdiff_t)c * o->m->nphixstargets_total) + o->m->level_phixstargetstart[ul] + t];
#endif
  const double W = o->cs->W[c];
  const double T_R = o->cs->TR[c];
  double gammacorr = W * lerp_or_last(o, o->m->corrphotoioncoeffs, ul, t, (float)T_R) /* T_R passed as double; table lerp in double */;
  const int ig = o->m->level_closestgroundlevelcont[ul];
  if (ig >= 0) gammacorr *= o->cs->corrphotoionrenorm[((ptrdiff_t)c * o->m->nbfcontinua_ground) + ig];
  return gammacorr;
}

/* ------------------------------------------------------------------ macroatom.cc rate coefficients */
static inline double gaunt_factor(int ionstage) { return ionstage == 1 ? 0.1 : (ionstage == 2 ? 0.2 : 0.3); } /* macroatom.cc:327 */
/* rad_deexcitation_ratecoeff macroatom.h:61 */
static double rad_deexcitation_ratecoeff(double epsilon_trans, float A_ul, double upperstatweight, double lowerstatweight,
                                         double nnlevelupper, double nnlevellower, double t_current) {
  const double nu_trans = epsilon_trans / H_PLANCK;
  const double B_ul = CLIGHTSQUAREDOVERTWOH / pow3(nu_trans) * A_ul;
  const double B_lu = upperstatweight / lowerstatweight * B_ul;
  const double tau_sobolev = ((B_lu * nnlevellower) - (B_ul * nnlevelupper)) * HCLIGHTOVERFOURPI * t_current;
  if (tau_sobolev > 1e-100) {
    const double beta = 1.0 / tau_sobolev * (-expm1(-tau_sobolev));
    return A_ul * beta;
  }
  return A_ul;
}
/* rad_excitation_ratecoeff macroatom.cc:611 */
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
/* radfield::get_Jblueindex radfield.cc:695 */
static int get_Jblueindex(const Oracle *o, int lineindex) {
  int low = 0, high = o->m->detailed_linecount - 1;
  while (low <= high) {
    const int mid = low + ((high - low) / 2);
    if (o->m->detailed_lineindices[mid] < lineindex) low = mid + 1;
    else if (o->m->detailed_lineindices[mid] > lineindex) high = mid - 1;
    else return mid;
  }
  return -1;
}
/* radfield::update_lineestimator radfield.cc:773 */
static void update_lineestimator(Oracle *o, int c, int lineindex, double increment) {
  const int jblueindex = get_Jblueindex(o, lineindex);
  if (jblueindex >= 0 && o->est.Jb_lu_raw) {
    const ptrdiff_t k = ((ptrdiff_t)c * o->m->detailed_linecount) + jblueindex;
    o->est.Jb_lu_raw[k] += increment;
    if (o->est.Jb_lu_contribcount) o->est.Jb_lu_contribcount[k] += 1;
  }
}
#endif
static double rad_excitation_ratecoeff(const Oracle *o, int c, double upper_statweight, double einstein_A, double epsilon_trans,
                                       double nnlevel_lower, double nnlevel_upper, double statweight_lower, double t_current,
                                       int alltransindex) {
  const double nu_trans = epsilon_trans / H_PLANCK;
  const double B_ul = CLIGHTSQUAREDOVERTWOH / pow3(nu_trans) * einstein_A;
  const double B_lu = upper_statweight / statweight_lower * B_ul;
  const double tau_sobolev = ((B_lu * nnlevel_lower) - (B_ul * nnlevel_upper)) * HCLIGHTOVERFOURPI * t_current;
  if (tau_sobolev > 1e-100) {
    const double beta = 1.0 / tau_sobolev * (-expm1(-tau_sobolev));
    const double R_over_J_nu = nnlevel_lower > 0. ? (B_lu - (B_ul * nnlevel_upper / nnlevel_lower)) * beta : B_lu * beta;
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
    { /* macroatom.cc:628 (globals::lte_iteration is false while packets propagate) */
      const int jblueindex = get_Jblueindex(o, o->m->alltrans_lineindex[alltransindex]);
      if (jblueindex >= 0) return R_over_J_nu * o->cs->Jb_lu_normed[((ptrdiff_t)c * o->m->detailed_linecount) + jblueindex]; /* get_Jb_lu */
    }
#else
    (void)alltransindex;
#endif
    return R_over_J_nu * radfield(o, nu_trans, c);
  }
  return 0.;
}
/* rad_recombination_ratecoeff macroatom.cc:646 */
static double rad_recombination_ratecoeff(const Oracle *o, float T_e, float clumpednne, int element, int upperion,
                                          int lowerionlevel, int t) {
  const int ul = ionlevelstart(o, element, upperion - 1) + lowerionlevel;
  return clumpednne * get_spontrecombcoeff(o, ul, t, T_e);
}
/* col_recombination_ratecoeff macroatom.cc:660 */
static double col_recombination_ratecoeff(const Oracle *o, float T_e, float clumpednne, int element, int upperion, int lower,
                                          int t, double epsilon_trans) {
  const int ul = ionlevelstart(o, element, upperion - 1) + lower;
  const double statw_lower = stat_weight(o, ul);
  const double g = gaunt_factor(get_ionstage(o, element, upperion - 1));
  const double sigma_bf = (get_phixs_table(o, ul)[0] * get_phixsprobability(o, ul, t));
  const double statw_upper = stat_weight(o, ionlevelstart(o, element, upperion) + get_phixsupperlevel(o, ul, t));
  return clumpednne * clumpednne * SAHACONST * statw_lower / statw_upper * 1.55e13 * g * sigma_bf * KB / T_e / epsilon_trans;
}
/* col_ionisation_ratecoeff macroatom.cc:686 */
static double col_ionisation_ratecoeff(const Oracle *o, float T_e, float clumpednne, int element, int ion, int lower, int t,
                                       double epsilon_trans) {
  const int ul = ionlevelstart(o, element, ion) + lower;
  const double g = gaunt_factor(get_ionstage(o, element, ion));
  const double fac1 = epsilon_trans / KB / T_e;
  const double sigma_bf = get_phixs_table(o, ul)[0] * get_phixsprobability(o, ul, t);
  return clumpednne * 1.55e13 * pow(T_e, -0.5) * g * sigma_bf * exp(-fac1) / fac1;
}
/* col_deexcitation_ratecoeff macroatom.cc:708; std::sqrt(T_e) is the float overload */
static double col_deexcitation_ratecoeff(const Oracle *o, float T_e, float clumpednne, double epsilon_trans,
                                         double upperstatweight, double lowerstatweight, int alltransindex) {
  const float coll_strength = o->m->alltrans_coll_str[alltransindex];
  if (coll_strength < 0) {
    if (!o->m->alltrans_forbidden[alltransindex]) {
      const double trans_osc_strength = o->m->alltrans_osc_strength[alltransindex];
      const double eoverkt = epsilon_trans / (KB * T_e);
      const double g_bar = 0.2;
      const double gauntfac = (eoverkt > 0.33421) ? g_bar : 0.276 * exp(eoverkt) * (-EULERGAMMA - log(eoverkt));
      const double g_ratio = lowerstatweight / upperstatweight;
      return C_0 * 14.51039491 * clumpednne * sqrtf(T_e) * trans_osc_strength * pow2(H_ionpot / epsilon_trans) * eoverkt *
             g_ratio * gauntfac;
    }
    return clumpednne * 8.629e-6 * 0.01 * lowerstatweight / sqrtf(T_e);
  }
  return clumpednne * 8.629e-6 * (double)coll_strength / upperstatweight / sqrtf(T_e);
}
/* col_excitation_ratecoeff macroatom.cc:750 */
static double col_excitation_ratecoeff(const Oracle *o, float T_e, float clumpednne, double epsilon_trans,
                                       double upperstatweight, double lowerstatweight, int alltransindex) {
  const float coll_strength = o->m->alltrans_coll_str[alltransindex];
  const double eoverkt = epsilon_trans / (KB * T_e);
  if (coll_strength < 0) {
    if (!o->m->alltrans_forbidden[alltransindex]) {
      const double trans_osc_strength = o->m->alltrans_osc_strength[alltransindex];
      const double g_bar = 0.2;
      const double exp_eoverkt = exp(eoverkt);
      const double Gamma = dmax(g_bar, 0.276 * exp_eoverkt * (-EULERGAMMA - log(eoverkt)));
      return C_0 * clumpednne * sqrtf(T_e) * 14.51039491 * trans_osc_strength * pow2(H_ionpot / epsilon_trans) * eoverkt /
             exp_eoverkt * Gamma;
    }
    return clumpednne * 8.629e-6 * 0.01 * exp(-eoverkt) * upperstatweight / sqrtf(T_e);
  }
  return clumpednne * 8.629e-6 * (double)coll_strength * exp(-eoverkt) / lowerstatweight / sqrtf(T_e);
}

/* ------------------------------------------------------------------ cell cache */
/* calculate_chi_ffheat_nnionpart rpkt.cc:932 */
static double calculate_chi_ffheat_nnionpart(const Oracle *o, int c) {
  const double g_ff = 1;
  double s = 0.;
  for (int element = 0; element < o->m->nelements; element++) {
    const int nions = get_nions(o, element);
    for (int ion = 0; ion < nions; ion++) {
      const double nnion = get_nnion(o, c, element, ion);
      const int ioncharge = get_ionstage(o, element, ion) - 1;
      s += pow2(ioncharge) * g_ff * nnion;
    }
  }
  const float T_e = o->cs->Te[c];
  return s * 3.69255e8 / sqrt(T_e);
}

/* calculate_macroatom_transitionrates macroatom.cc:64 */
static void calculate_macroatom_transitionrates(Oracle *o, CellCache *cc, int c, int element, int ion, int level, double t_mid) {
  const artis_model *m = o->m;
  const int start = ionlevelstart(o, element, ion);
  const int ul = start + level;
  double *levelrates = cc->maprocessrates + ((ptrdiff_t)ul * ARTIS_MA_ACTION_COUNT);
  double *transblock = cc->matrans;
  const int blockstart = m->level_matransblock_start[ul];
  const float T_e = o->cs->Te[c];
  const float clumpednne = cell_clumpednne(o, c);
  const double epsilon_current = epsilon(o, ul);
  const double statweight = stat_weight(o, ul);
  const double nnlevel = cc->levelpops[ul];

  double sum_internal_down_same = 0., sum_raddeexc = 0., sum_coldeexc = 0.;
  const int startdown = m->level_alltrans_startdown[ul];
  const int ndowntrans = m->level_ndowntrans[ul];
  for (int i = 0; i < ndowntrans; i++) {
    const int ati = startdown + i;
    const int lower = m->alltrans_targetlevelindex[ati];
    const float A_ul = m->alltrans_einstein_A[ati];
    const int lul = start + lower;
    const double epsilon_target = epsilon(o, lul);
    const double epsilon_trans = epsilon_current - epsilon_target;
    const double lower_statweight = stat_weight(o, lul);
    const double R = rad_deexcitation_ratecoeff(epsilon_trans, A_ul, statweight, lower_statweight, nnlevel, cc->levelpops[lul], t_mid);
    const double C = col_deexcitation_ratecoeff(o, T_e, clumpednne, epsilon_trans, statweight, lower_statweight, ati);
    sum_raddeexc += R * epsilon_trans;
    sum_coldeexc += C * epsilon_trans;
    sum_internal_down_same += (R + C) * epsilon_target;
    transblock[blockstart + i] = sum_raddeexc;
    transblock[blockstart + ndowntrans + i] = sum_internal_down_same;
  }
  levelrates[ARTIS_MA_ACTION_RADDEEXC] = sum_raddeexc;
  levelrates[ARTIS_MA_ACTION_COLDEEXC] = sum_coldeexc;
  levelrates[ARTIS_MA_ACTION_INTERNALDOWNSAME] = sum_internal_down_same;

  double sum_internal_up_same = 0.;
  const int nuptrans = m->level_nuptrans[ul];
  const int startup = startdown + ndowntrans;
  for (int ii = 0; ii < nuptrans; ii++) {
    const int ati = startup + ii;
    const int upper = m->alltrans_targetlevelindex[ati];
    const int uul = start + upper;
    const double epsilon_trans = epsilon(o, uul) - epsilon_current;
    const double upper_statweight = stat_weight(o, uul);
    const double R = rad_excitation_ratecoeff(o, c, upper_statweight, m->alltrans_einstein_A[ati], epsilon_trans, nnlevel,
                                              cc->levelpops[uul], statweight, t_mid, ati);
    const double C = col_excitation_ratecoeff(o, T_e, clumpednne, epsilon_trans, upper_statweight, statweight, ati);
#if ARTIS_OPT_NT_ON
    const double NT = nt_excitation_ratecoeff(o, c, level, upper, ati);
#else
    const double NT = 0.; /* nonthermal::nt_excitation_ratecoeff with NT_ON false */
#endif
    sum_internal_up_same += (R + C + NT) * epsilon_current;
    transblock[blockstart + (2 * ndowntrans) + ii] = sum_internal_up_same;
  }
  levelrates[ARTIS_MA_ACTION_INTERNALUPSAME] = sum_internal_up_same;

  double sum_internal_down_lower = 0., sum_radrecomb = 0., sum_colrecomb = 0.;
  if (ion > 0 && level <= m->ion_maxrecombininglevel[uniqueion(o, element, ion)]) {
    const int nlevels = get_nlevels_ionising(o, element, ion - 1);
    const int lstart = ionlevelstart(o, element, ion - 1);
    for (int lower = 0; lower < nlevels; lower++) {
      const int t = find_phixstargetindex(o, lstart + lower, level);
      if (t < 0) continue;
      const double epsilon_target = epsilon(o, lstart + lower);
      const double epsilon_trans = epsilon_current - epsilon_target;
      const double R = rad_recombination_ratecoeff(o, T_e, clumpednne, element, ion, lower, t);
      const double C = col_recombination_ratecoeff(o, T_e, clumpednne, element, ion, lower, t, epsilon_trans);
      sum_internal_down_lower += (R + C) * epsilon_target;
      sum_radrecomb += R * epsilon_trans;
      sum_colrecomb += C * epsilon_trans;
    }
  }
  levelrates[ARTIS_MA_ACTION_INTERNALDOWNLOWER] = sum_internal_down_lower;
  levelrates[ARTIS_MA_ACTION_RADRECOMB] = sum_radrecomb;
  levelrates[ARTIS_MA_ACTION_COLRECOMB] = sum_colrecomb;

  double sum_up_highernt = 0., sum_up_higher = 0.;
  const int ionisinglevels = get_nlevels_ionising(o, element, ion);
  if (ion < get_nions(o, element) - 1 && level < ionisinglevels) {
#if ARTIS_OPT_NT_ON
    sum_up_highernt = nt_ionisation_ratecoeff(o, c, element, ion) * epsilon_current; /* macroatom.cc:181 */
#endif
    const int nt = m->level_nphixstargets[ul];
    for (int t = 0; t < nt; t++) {
      const double epsilon_trans = get_phixs_threshold(o, element, ion, level, t);
      const double R = cc->corrphotoioncoeff[m->level_phixstargetstart[ul] + t];
      const double C = col_ionisation_ratecoeff(o, T_e, clumpednne, element, ion, level, t, epsilon_trans);
      sum_up_higher += (R + C) * epsilon_current;
    }
  }
  levelrates[ARTIS_MA_ACTION_INTERNALUPHIGHERNT] = sum_up_highernt;
  levelrates[ARTIS_MA_ACTION_INTERNALUPHIGHER] = sum_up_higher;
}

/* calculate_cooling_rates_ion kpkt.cc:57. contribs != NULL is the
 * update_cellcache_contribs == true instantiation. Returns C_ion. */
static double calculate_cooling_rates_ion(Oracle *o, const CellCache *cc, int c, int element, int ion, double *ion_contribs) {
  const artis_model *m = o->m;
  const float clumpednne = cell_clumpednne(o, c);
  const float T_e = o->cs->Te[c];
  double C_ion = 0.;
  int k = 0;
  const int nionisinglevels = get_nlevels_ionising(o, element, ion);
  const double nncurrention = get_nnion(o, c, element, ion);
  const int ioncharge = get_ionstage(o, element, ion) - 1;
  if (ioncharge > 0) {
    const double C_ff_ion = 1.426e-27 * sqrt(T_e) * pow2(ioncharge) * nncurrention * clumpednne;
    C_ion += C_ff_ion;
    if (ion_contribs) ion_contribs[k++] = C_ion;
  }
  const int start = ionlevelstart(o, element, ion);
  const int nlevels = get_nlevels(o, element, ion);
  for (int level = 0; level < nlevels; level++) {
    const int ul = start + level;
    const double nnlevel = cc->levelpops[ul]; /* == calculate_levelpop() */
    const double epsilon_current = epsilon(o, ul);
    const double statweight = stat_weight(o, ul);
    const int startup = m->level_alltrans_startdown[ul] + m->level_ndowntrans[ul];
    const int nuptrans = m->level_nuptrans[ul];
    for (int ati = startup; ati < (startup + nuptrans); ati++) {
      const int upper = m->alltrans_targetlevelindex[ati];
      const double epsilon_trans = epsilon(o, start + upper) - epsilon_current;
      const double upper_statweight = stat_weight(o, start + upper);
      const double C = nnlevel * col_excitation_ratecoeff(o, T_e, clumpednne, epsilon_trans, upper_statweight, statweight, ati) *
                       epsilon_trans;
      C_ion += C;
    }
    if (ion_contribs && nuptrans > 0) ion_contribs[k++] = C_ion;
  }
  if (ion < (get_nions(o, element) - 1) && m->nbfcontinua > 0) {
    const double nnupperion = get_nnion(o, c, element, ion + 1);
    const int ustart = ionlevelstart(o, element, ion + 1);
    for (int level = 0; level < nionisinglevels; level++) {
      const int ul = start + level;
      const double epsilon_current = epsilon(o, ul);
      const double nnlevel = cc->levelpops[ul];
      const int nt = m->level_nphixstargets[ul];
      for (int t = 0; t < nt; t++) {
        const int upper = get_phixsupperlevel(o, ul, t);
        const double epsilon_upper = epsilon(o, ustart + upper);
        const double epsilon_trans = epsilon_upper - epsilon_current;
        const double C = nnlevel * col_ionisation_ratecoeff(o, T_e, clumpednne, element, ion, level, t, epsilon_trans) * epsilon_trans;
        C_ion += C;
        if (ion_contribs) ion_contribs[k++] = C_ion;
      }
    }
    for (int level = 0; level < nionisinglevels; level++) {
      const int ul = start + level;
      const int nt = m->level_nphixstargets[ul];
      double targetweight_sum = 0.;
      double E_target_min = 0.;
      (void)targetweight_sum; (void)E_target_min; (void)nnupperion;
#if !ARTIS_OPT_BFCOOLING_USELEVELPOPNOTIONPOP
      if (nt > 1) {
        E_target_min = DBL_MAXV;
        for (int t = 0; t < nt; t++) E_target_min = dmin(E_target_min, epsilon(o, ustart + get_phixsupperlevel(o, ul, t)));
        for (int t = 0; t < nt; t++) {
          const int upperlevel = get_phixsupperlevel(o, ul, t);
          targetweight_sum += stat_weight(o, ustart + upperlevel) * exp(-(epsilon(o, ustart + upperlevel) - E_target_min) / KB / T_e);
        }
      }
#endif
      for (int t = 0; t < nt; t++) {
        double pop;
#if ARTIS_OPT_BFCOOLING_USELEVELPOPNOTIONPOP
        pop = cc->levelpops[ustart + get_phixsupperlevel(o, ul, t)];
#else
        if (nt == 1) {
          pop = nnupperion;
        } else {
          const int upperlevel = get_phixsupperlevel(o, ul, t);
          const double targetweight = stat_weight(o, ustart + upperlevel) * exp(-(epsilon(o, ustart + upperlevel) - E_target_min) / KB / T_e);
          pop = nnupperion * targetweight / targetweight_sum;
        }
#endif
        const double C = get_bfcoolingcoeff(o, ul, t, T_e) * pop * clumpednne;
        C_ion += C;
        if (ion_contribs) ion_contribs[k++] = C_ion;
      }
    }
  }
  if (ion_contribs && k != m->ion_ncoolingterms[uniqueion(o, element, ion)]) ORACLE_FAIL(o, "coolinglist size mismatch");
  return C_ion;
}

static void cellcache_free_one(CellCache *cc) {
  free(cc->levelpops); free(cc->maprocessrates); free(cc->matrans); free(cc->allcont_nnlevel);
  free(cc->allcont_departure); free(cc->allcont_edgepart); free(cc->allcont_keepbits);
  free(cc->corrphotoioncoeff); free(cc->cooling_contrib); free(cc->ion_cooling_contribs);
  free(cc->expansionopacities); free(cc->expansionopacity_planck_cumulative);
  memset(cc, 0, sizeof(*cc));
}

#if ARTIS_EXPOPAC_TABLES
static void calculate_expansion_opacities(Oracle *o, CellCache *cc, int c);
#endif
/* cellcacheslot_populate update_packets.cc:397. As in the reference's CPU build (cellcache_singleslot,
 * update_packets.cc:459-463 and :408) the macro-atom rates of a level and the cooling terms of an ion are
 * left at a negative sentinel here and calculated on first use (macroatom.cc:403, kpkt.cc:459); the values
 * are the ones the multi-slot form pre-calculates, so results do not depend on which form runs. */
static void cellcache_populate(Oracle *o, int c) {
  const artis_model *m = o->m;
  CellCache *cc = &o->cache[c];
  if (cc->populated) return;
  const double t0 = now_s();
  if (o->cache_cap > 0 && o->npopulated >= o->cache_cap) {
    for (int k = 0; k < m->npts_nonempty; k++)
      if (o->cache[k].populated) cellcache_free_one(&o->cache[k]);
    o->npopulated = 0;
  }
  cc->levelpops = (double *)malloc(sizeof(double) * (size_t)m->nlevels);
  cc->maprocessrates = (double *)malloc(sizeof(double) * (size_t)m->nlevels * ARTIS_MA_ACTION_COUNT);
  cc->matrans = (double *)calloc((size_t)(m->nmatransblock > 0 ? m->nmatransblock : 1), sizeof(double));
  cc->allcont_nnlevel = (double *)malloc(sizeof(double) * (size_t)(m->nbfcontinua + 1));
  cc->allcont_departure = (double *)malloc(sizeof(double) * (size_t)(m->nbfcontinua + 1));
  cc->allcont_edgepart = (double *)malloc(sizeof(double) * (size_t)(m->nbfcontinua + 1));
  const int nwords = (m->nbfcontinua + 63) / 64;
  cc->allcont_keepbits = (uint64_t *)calloc((size_t)(nwords + 1), sizeof(uint64_t));
  cc->corrphotoioncoeff = (double *)malloc(sizeof(double) * (size_t)(m->nphixstargets_total + 1));
  cc->cooling_contrib = (double *)calloc((size_t)(m->ncoolingterms + 1), sizeof(double));
  cc->ion_cooling_contribs = (double *)malloc(sizeof(double) * (size_t)m->nions);
  stat_inc(o, ARTIS_STAT_UPDATECELL);

  cc->chi_ff_nnionpart = calculate_chi_ffheat_nnionpart(o, c);
  for (int element = 0; element < m->nelements; element++) {
    const int nions = get_nions(o, element);
    for (int ion = 0; ion < nions; ion++) {
      const int nlevels = get_nlevels(o, element, ion);
      const int start = ionlevelstart(o, element, ion);
      for (int level = 0; level < nlevels; level++) cc->levelpops[start + level] = calculate_levelpop(o, c, element, ion, level);
      cc->cooling_contrib[m->ion_coolingoffset[uniqueion(o, element, ion)]] = -99.; /* update_packets.cc:408 */
    }
  }
  const float T_e = o->cs->Te[c];
  const float nnetot = o->cs->nnetot[c];
  const float clumpednne = cell_nne(o, c) * o->cs->clumpfactor[c];
  const double modified_sahafact_statweightpart = SAHACONST * pow(T_e, -1.5); /* rpkt.cc:738 */
  for (int i = 0; i < m->nbfcontinua; i++) {
    const double nnlevel = cc->levelpops[m->allcont_uniquelevelindex[i]];
    const int element = m->allcont_element[i];
    const int ion = m->allcont_ion[i];
    const int level = m->allcont_level[i];
    /* keep_this_cont rpkt.h:189 (DETAILED_BF_ESTIMATORS_ON false) */
    const int keep = nnlevel > 0 && ((get_nnion(o, c, element, ion) / nnetot > 1.e-6) || (level == 0));
    cc->allcont_nnlevel[i] = nnlevel;
    cc->allcont_departure[i] = -1.;
    cc->allcont_edgepart[i] = -1.;
    if (keep) {
      cc->allcont_keepbits[i / 64] |= UINT64_C(1) << (unsigned)(i % 64);
      /* slow path of calculate_chi_bf_gammacontr, rpkt.cc:853-889 (header note 3) */
      const int upper = m->allcont_upperlevel[i];
      const double nnupperionlevel = cc->levelpops[ionlevelstart(o, element, ion + 1) + upper];
      const double modified_sahafact = modified_sahafact_statweightpart * stat_weight(o, ionlevelstart(o, element, ion) + level) /
                                       stat_weight(o, ionlevelstart(o, element, ion + 1) + upper);
      const double ratio = nnupperionlevel / nnlevel * clumpednne * modified_sahafact;
      cc->allcont_departure[i] = ratio;
      const double edge_exponent = HOVERKB * m->allcont_nu_edge[i] / T_e;
      if (edge_exponent < 690.) {
        const double edgepart = ratio * exp(edge_exponent);
        if (isfinite(edgepart)) cc->allcont_edgepart[i] = edgepart;
      }
    }
  }
  for (int ul = 0; ul < m->nlevels; ul++) {
    const int nt = m->level_nphixstargets[ul];
    for (int t = 0; t < nt; t++) cc->corrphotoioncoeff[m->level_phixstargetstart[ul] + t] = calc_corrphotoioncoeff(o, c, ul, t);
    cc->maprocessrates[(ptrdiff_t)ul * ARTIS_MA_ACTION_COUNT] = -99.; /* update_packets.cc:461 */
  }
#if ARTIS_EXPOPAC_TABLES
  if (!o->cs->expansionopacities && o->cs->thick[c] != ARTIS_CELL_THICK) calculate_expansion_opacities(o, cc, c); /* update_grid.cc:655 */
#endif
  cc->have_ion_cooling = 0;
  cc->populated = 1;
  o->npopulated++;
  o->t_populate += now_s() - t0;
}

/* macroatom.cc:403 calc_rates_if_needed */
static const double *macroatom_levelrates(Oracle *o, int c, int element, int ion, int level) {
  CellCache *cc = &o->cache[c];
  const int ul = ionlevelstart(o, element, ion) + level;
  double *levelrates = cc->maprocessrates + ((ptrdiff_t)ul * ARTIS_MA_ACTION_COUNT);
  if (levelrates[0] < 0.) calculate_macroatom_transitionrates(o, cc, c, element, ion, level, o->ts.mid);
  return levelrates;
}
/* kpkt.cc:459 calc_cooling_if_needed */
static const double *cooling_ion_contribs(Oracle *o, int c, int element, int ion) {
  CellCache *cc = &o->cache[c];
  const int ui = uniqueion(o, element, ion);
  double *ion_contribs = cc->cooling_contrib + o->m->ion_coolingoffset[ui];
  if (ion_contribs[0] < 0.) calculate_cooling_rates_ion(o, cc, c, element, ion, ion_contribs);
  return ion_contribs;
}
/* kpkt::calculate_cooling_rates kpkt.cc:281 (done by update_grid in the reference; here on the first k-packet of a cell) */
static const double *cell_ion_cooling_contribs(Oracle *o, int c) {
  CellCache *cc = &o->cache[c];
  if (!cc->have_ion_cooling) {
    double cumulative_cooling = 0.;
    for (int element = 0; element < o->m->nelements; element++) {
      const int nions = get_nions(o, element);
      for (int ion = 0; ion < nions; ion++) {
        cumulative_cooling += calculate_cooling_rates_ion(o, cc, c, element, ion, NULL);
        cc->ion_cooling_contribs[uniqueion(o, element, ion)] = cumulative_cooling;
      }
    }
    cc->have_ion_cooling = 1;
  }
  return cc->ion_cooling_contribs;
}

/* ------------------------------------------------------------------ rpkt.cc opacities */
/* calculate_chi_ffheating rpkt.cc:697 */
static double calculate_chi_ffheating(const Oracle *o, const CellCache *cc, int c, double nu) {
  const float clumpednne = cell_nne(o, c) * o->cs->clumpfactor[c];
  const float T_e = o->cs->Te[c];
  return cc->chi_ff_nnionpart / pow3(nu) * clumpednne * (1 - exp(-HOVERKB * nu / T_e));
}

/* calculate_chi_bf_gammacontr<true, SELECTCONTINUUM> rpkt.cc:721 */
static double calculate_chi_bf_gammacontr(Oracle *o, const CellCache *cc, int c, double nu, double *groundcont_gamma_contr,
                                          int selectcontinuum, double threshold, int *selected, ContOpacity *phixslist) {
  const artis_model *m = o->m;
  double chi_bf_sum = 0.;
  if (!selectcontinuum && (ARTIS_OPT_USE_LUT_PHOTOION || ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS)) {
    for (int i = 0; i < m->nbfcontinua_ground; i++) groundcont_gamma_contr[i] = 0.;
  }
  const float T_e = o->cs->Te[c];
  const double exp_minus_hnu_over_kte = exp(-HOVERKB * nu / T_e);
  const int stimfactor_split_usable = (exp_minus_hnu_over_kte >= DBL_MINV);
  const int allcontend = upper_bound_d(m->allcont_nu_edge, m->nbfcontinua, nu);
  const int allcontbegin = lower_bound_d(m->allcont_nu_edge, allcontend, nu / o->last_phixs_nuovernuedge);
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
  if (!selectcontinuum && phixslist) {
    /* rpkt.cc:762-776: the window in estimator indices, cleared (only contributing continua write below) */
    phixslist->bfestimend = upper_bound_d(o->bfestim_nu_edge, o->nbfestim, nu);
    phixslist->bfestimbegin = lower_bound_d(o->bfestim_nu_edge, phixslist->bfestimend, nu / o->last_phixs_nuovernuedge);
    for (int i = phixslist->bfestimbegin; i < phixslist->bfestimend; i++) phixslist->gamma_contr[i] = 0.;
  }
#endif

  for (int word = allcontbegin / 64; word * 64 < allcontend; word++) {
    uint64_t bits = cc->allcont_keepbits[word];
    if (word == (allcontbegin / 64)) bits &= ~UINT64_C(0) << (unsigned)(allcontbegin % 64);
    if (((word + 1) * 64) > allcontend) bits &= ~UINT64_C(0) >> (unsigned)(64 - (allcontend % 64));
    while (bits != 0) {
      const int i = (word * 64) + __builtin_ctzll(bits);
      bits &= bits - 1;
      const double nnlevel = cc->allcont_nnlevel[i];
      const double nu_edge = m->allcont_nu_edge[i];
      const double sigma_bf = photoionisation_crosssection_fromtable(o, get_phixs_table(o, m->allcont_uniquelevelindex[i]), nu_edge, nu);
      const double stimfactor_edgepart = cc->allcont_edgepart[i];
      double stimfactor;
      if (stimfactor_edgepart >= 0. && stimfactor_split_usable) {
        stimfactor = stimfactor_edgepart * exp_minus_hnu_over_kte;
      } else {
        stimfactor = cc->allcont_departure[i] * exp(-HOVERKB * (nu - nu_edge) / T_e);
      }
      const double corrfactor = dmax(0., 1 - stimfactor);
      const double sigma_contr = sigma_bf * m->allcont_probability[i] * corrfactor;
      if (!selectcontinuum && (ARTIS_OPT_USE_LUT_PHOTOION || ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS)) {
        if (m->allcont_groundcontestimindex[i] >= 0) groundcont_gamma_contr[m->allcont_groundcontestimindex[i]] = sigma_contr;
      }
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
      if (!selectcontinuum && phixslist && o->allcont_bfestimindex[i] >= 0)
        phixslist->gamma_contr[o->allcont_bfestimindex[i]] = sigma_contr; /* rpkt.cc:905 */
#endif
      chi_bf_sum += nnlevel * sigma_contr;
      if (selectcontinuum && chi_bf_sum > threshold) {
        *selected = i;
        return chi_bf_sum;
      }
    }
  }
  if (selectcontinuum) {
    *selected = allcontend - 1;
    return chi_bf_sum;
  }
  if (!isfinite(chi_bf_sum)) ORACLE_FAIL(o, "chi_bf_sum not finite");
  return chi_bf_sum;
}

/* calculate_chi_rpkt_cont<true> rpkt.cc:1021 */
static void calculate_chi_rpkt_cont(Oracle *o, const CellCache *cc, double nu_cmf, ContOpacity *chi, int c) {
  if ((c == chi->nonemptymgi) && (fabs((chi->nu / nu_cmf) - 1.0) < 1e-4)) return;
  const float nne = cell_nne(o, c);
  chi->chi_freefree_heat = calculate_chi_ffheating(o, cc, c, nu_cmf);
  chi->chi_escatter = SIGMA_T * nne;
  chi->chi_boundfree = calculate_chi_bf_gammacontr(o, cc, c, nu_cmf, chi->groundcont_gamma_contr, 0, 0., NULL, chi);
  chi->nonemptymgi = c;
  chi->nu = nu_cmf;
}
static inline double chi_total(const ContOpacity *chi) { /* rpkt.h:100 */
  return chi->chi_escatter + chi->chi_boundfree + chi->chi_freefree_heat;
}

/* closest_transition rpkt.h:155 */
static int closest_transition(const double *linelistnu, int nlines, double nu_cmf, int next_trans) {
  if (nlines <= 0) return -1; /* an empty line list: the reference would read linelist.nu.back() of an empty span */
  if (next_trans > (nlines - 1)) return -1;
  if (nu_cmf < linelistnu[nlines - 1]) return -1;
  if (next_trans > 0) return next_trans;
  if (nu_cmf >= linelistnu[0]) return 0;
  /* lower_bound with greater{}: first index where !(nu[i] > nu_cmf) */
  int lo = 0, len = nlines;
  while (len > 0) {
    int half = len / 2;
    if (linelistnu[lo + half] > nu_cmf) { lo += half + 1; len -= half + 1; } else { len = half; }
  }
  return lo;
}
/* get_linedistance rpkt.h:125 (non-relativistic) */
static inline double get_linedistance(double prop_time, double nu_cmf, double nu_trans, double dnu_on_dl) {
  if (nu_cmf <= nu_trans) return 0.;
  const double delta_nu = nu_cmf - nu_trans;
#if ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT
  (void)prop_time;
  return -delta_nu / dnu_on_dl; /* linear interpolation of the frequency along the path, rpkt.h:126-132 */
#else
  (void)dnu_on_dl;
  return CLIGHT * prop_time * delta_nu / nu_trans;
#endif
}
/* get_tau_sobolev<true> rpkt.cc:75 */
static inline double get_tau_sobolev(const Oracle *o, const CellCache *cc, int lineindex, double t_current) {
  const double n_l = cc->levelpops[o->m->line_uniquelevelindex_lower[lineindex]];
  const double n_u = cc->levelpops[o->m->line_uniquelevelindex_upper[lineindex]];
  const double B_ul = o->m->line_B_ul[lineindex];
  const double B_lu = o->m->line_B_lu[lineindex];
  return dmax(((B_lu * n_l) - (B_ul * n_u)) * HCLIGHTOVERFOURPI * t_current, 0.);
}
/* get_nu_cmf_abort rpkt.cc:54 */
static double get_nu_cmf_abort(const double pos[3], const double dir[3], double prop_time, double nu_rf, double abort_dist) {
  const double half = abort_dist / 2.;
  const double abort_time = prop_time + (half / CLIGHT_PROP) + (half / CLIGHT_PROP);
  const double abort_pos[3] = {pos[0] + (dir[0] * half) + (dir[0] * half), pos[1] + (dir[1] * half) + (dir[1] * half),
                               pos[2] + (dir[2] * half) + (dir[2] * half)};
  return nu_rf * doppler_nucmf_on_nurf(abort_pos, dir, abort_time);
}

/* get_possible_event rpkt.cc:106. Returns edist; *next_trans_out, *is_bb set. */
static double get_possible_event(Oracle *o, const CellCache *cc, const artis_packet *pkt, const ContOpacity *chi,
                                 MacroAtomState *mastate, double tau_rnd, double abort_dist, double nu_cmf_abort,
                                 double dnu_on_dl, double doppler, int *next_trans_out, int *is_bb) {
  const artis_model *m = o->m;
  double pos[3] = {pkt->pos[0], pkt->pos[1], pkt->pos[2]};
  double nu_cmf = pkt->nu_cmf;
  double e_cmf = pkt->e_cmf;
  double prop_time = pkt->prop_time;
  int next_trans = pkt->next_trans;
  const double chi_cont = chi_total(chi) * doppler;
  double tau = 0.;
  double dist = 0.;
  while (1) {
    const int lineindex = closest_transition(m->line_nu, m->nlines, nu_cmf, next_trans);
    if (lineindex < 0) {
      const double tau_cont = chi_cont * (abort_dist - dist);
      if (tau_rnd - tau > tau_cont) {
        *next_trans_out = next_trans;
        *is_bb = 0;
        return DBL_MAXV;
      }
      *next_trans_out = m->nlines + 1;
      *is_bb = 0;
      return dist + ((tau_rnd - tau) / chi_cont);
    }
    o->est.stats[ARTIS_STAT_X_LINES_VISITED]++;
    const double nu_trans = m->line_nu[lineindex];
    next_trans = lineindex + 1;
    const double ldist = get_linedistance(prop_time, nu_cmf, nu_trans, dnu_on_dl);
    const double tau_cont = chi_cont * ldist;
    if (tau_rnd - tau > tau_cont) {
      if (nu_trans < nu_cmf_abort) {
        *next_trans_out = next_trans - 1;
        *is_bb = 0;
        return DBL_MAXV;
      }
      const double tau_line = get_tau_sobolev(o, cc, lineindex, prop_time);
      if ((tau_rnd - tau) <= (tau_cont + tau_line)) {
        const int element = m->line_elementindex[lineindex];
        const int ion = m->line_ionindex[lineindex];
        const int upper = m->line_uniquelevelindex_upper[lineindex] - ionlevelstart(o, element, ion);
        mastate->element = element;
        mastate->ion = ion;
        mastate->level = upper;
        mastate->activatingline = lineindex;
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
        move_pkt_withtime_raw(pos, pkt->dir, &prop_time, pkt->nu_rf, &nu_cmf, pkt->e_rf, &e_cmf, ldist); /* rpkt.cc:173-176 */
        update_lineestimator(o, (int)(cc - o->cache), lineindex, prop_time * CLIGHT * e_cmf / nu_cmf);
#endif
        *next_trans_out = next_trans;
        *is_bb = 1;
        return dist + ldist;
      }
      dist += ldist;
      tau += tau_cont + tau_line;
#if !ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT
      move_pkt_withtime_raw(pos, pkt->dir, &prop_time, pkt->nu_rf, &nu_cmf, pkt->e_rf, &e_cmf, ldist);
#else
      /* rpkt.cc:190-196: the linear approximation instead of the Doppler formula */
      pos[0] += (pkt->dir[0] * ldist);
      pos[1] += (pkt->dir[1] * ldist);
      pos[2] += (pkt->dir[2] * ldist);
      prop_time += ldist / CLIGHT_PROP;
      nu_cmf = pkt->nu_cmf + (dnu_on_dl * dist);
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
      e_cmf = nu_cmf * pkt->e_rf / pkt->nu_rf; /* rpkt.cc:199-203 */
#else
      (void)e_cmf;
#endif
#endif
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
      update_lineestimator(o, (int)(cc - o->cache), lineindex, prop_time * CLIGHT * e_cmf / nu_cmf); /* rpkt.cc:206 */
#endif
    } else {
      *next_trans_out = next_trans - 1;
      *is_bb = 0;
      return dist + ((tau_rnd - tau) / chi_cont);
    }
  }
}

/* emit_rpkt rpkt.cc:991 */
#if ARTIS_EXPOPAC_TABLES
/* wavelength bins in ascending wavelength (descending frequency), rpkt.h:30-40 */
static inline double get_expopac_bin_nu_upper(ptrdiff_t binindex) {
  return 1e8 * CLIGHT / (ARTIS_EXPOPAC_LAMBDAMIN + ((double)binindex * ARTIS_EXPOPAC_DELTALAMBDA));
}
static inline double get_expopac_bin_nu_lower(ptrdiff_t binindex) {
  return 1e8 * CLIGHT / (ARTIS_EXPOPAC_LAMBDAMIN + ((double)(binindex + 1) * ARTIS_EXPOPAC_DELTALAMBDA));
}
/* get_linearbinindex sn3d.h:115 */
static inline ptrdiff_t get_linearbinindex(double value, double minvalue, double binwidth) {
  const double fracindex = (value - minvalue) / binwidth;
  const ptrdiff_t truncated = (ptrdiff_t)fracindex;
  return (fracindex < (double)truncated) ? truncated - 1 : truncated;
}
/* calculate_expansion_opacities rpkt.cc:1071 (get_tau_sobolev<false> evaluates the populations the cell cache holds) */
static void calculate_expansion_opacities(Oracle *o, CellCache *cc, int c) {
  const artis_model *m = o->m;
  const float rho = o->cs->rho[c];
  const float temperature = o->cs->Te[c];
  const double t_mid = o->ts.mid;
  if (!cc->expansionopacities) cc->expansionopacities = (float *)calloc(ARTIS_EXPOPAC_NBINS, sizeof(float));
  if (!cc->expansionopacity_planck_cumulative) cc->expansionopacity_planck_cumulative = (double *)calloc(ARTIS_EXPOPAC_NBINS, sizeof(double));
  /* the first line with nu below the upper limit of the first bin: lower_bound(linelist.nu, nu_upper(0), greater) */
  int lineindex = 0;
  {
    const double nu0 = get_expopac_bin_nu_upper(0);
    int lo = 0, len = m->nlines;
    while (len > 0) {
      const int half = len / 2;
      if (m->line_nu[lo + half] > nu0) { lo += half + 1; len -= half + 1; } else { len = half; }
    }
    lineindex = lo;
  }
  double kappa_planck_cumulative = 0.;
  for (ptrdiff_t binindex = 0; binindex < ARTIS_EXPOPAC_NBINS; binindex++) {
    double bin_linesum = 0.;
    const double nu_lower = get_expopac_bin_nu_lower(binindex);
    while (lineindex < m->nlines && m->line_nu[lineindex] >= nu_lower) {
      const double tau_line = get_tau_sobolev(o, cc, lineindex, t_mid);
      const double linelambda = 1e8 * CLIGHT / m->line_nu[lineindex];
      bin_linesum += (linelambda / ARTIS_EXPOPAC_DELTALAMBDA) * -expm1(-tau_line);
      lineindex++;
    }
    const float bin_kappa_bb = (float)(1. / (CLIGHT * t_mid * rho) * bin_linesum);
    if (!isfinite(bin_kappa_bb)) ORACLE_FAIL(o, "calculate_expansion_opacities: kappa not finite");
    cc->expansionopacities[binindex] = bin_kappa_bb;
#if ARTIS_OPT_RPKT_BB_THERMALISATION
    const double nu_upper = get_expopac_bin_nu_upper(binindex);
    const double nu_mid = (nu_upper + nu_lower) / 2.;
    const double bin_kappa_cont = calculate_chi_ffheating(o, cc, c, nu_mid) / rho;
    const double planck_val = 2 * H_PLANCK * pow3(nu_mid) / pow2(CLIGHT) / expm1(HOVERKB * nu_mid / temperature); /* radfield.h:50 */
    const double kappa_planck = (bin_kappa_bb + bin_kappa_cont) * planck_val;
    const double delta_nu = nu_upper - nu_lower;
    kappa_planck_cumulative += kappa_planck * delta_nu;
    cc->expansionopacity_planck_cumulative[binindex] = kappa_planck_cumulative;
#else
    (void)temperature; (void)kappa_planck_cumulative;
#endif
  }
}
#endif
#if ARTIS_EXPOPAC_TABLES
/* the cell's tables: the host's (artis_cellstate) or the ones calculate_expansion_opacities() below made at population */
static const float *cell_expansionopacities(const Oracle *o, int c) {
  return o->cs->expansionopacities ? o->cs->expansionopacities + ((ptrdiff_t)c * ARTIS_EXPOPAC_NBINS) : o->cache[c].expansionopacities;
}
static const double *cell_expopac_planck_cumulative(const Oracle *o, int c) {
  return o->cs->expansionopacity_planck_cumulative ? o->cs->expansionopacity_planck_cumulative + ((ptrdiff_t)c * ARTIS_EXPOPAC_NBINS)
                                                   : o->cache[c].expansionopacity_planck_cumulative;
}
#endif
#if ARTIS_OPT_RPKT_BB_THERMALISATION
/* sample_planck_times_expansion_opacity rpkt.cc:964 */
static double sample_planck_times_expansion_opacity(Oracle *o, int c, uint32_t *rngstate) {
  const double *kappa_planck_bins = cell_expopac_planck_cumulative(o, c);
  if (!(kappa_planck_bins[ARTIS_EXPOPAC_NBINS - 1] > 0)) ORACLE_FAIL(o, "sample_planck_times_expansion_opacity: empty integral");
  const double rnd_integral = rng_uniform(rngstate) * kappa_planck_bins[ARTIS_EXPOPAC_NBINS - 1];
  int binindex = upper_bound_d(kappa_planck_bins, ARTIS_EXPOPAC_NBINS, rnd_integral); /* index_upperbound sn3d.h:85 */
  if (binindex > ARTIS_EXPOPAC_NBINS - 1) binindex = ARTIS_EXPOPAC_NBINS - 1;
  const double bin_nu_lower = get_expopac_bin_nu_lower(binindex);
  const double delta_nu = get_expopac_bin_nu_upper(binindex) - bin_nu_lower;
  const double nuoffset = rng_uniform(rngstate) * delta_nu;
  return bin_nu_lower + nuoffset;
}
#endif
#if ARTIS_OPT_RPKT_USE_EXPANSION_OPACITIES
/* get_possible_event_expansion_opacity rpkt.cc:221 */
static double get_possible_event_expansion_opacity(Oracle *o, const CellCache *cc, int c, artis_packet *pkt, const ContOpacity *chi,
                                                   MacroAtomState *mastate, double tau_rnd, double nu_cmf_abort, double dnu_on_dl,
                                                   double doppler, int *is_bb) {
  double pos[3] = {pkt->pos[0], pkt->pos[1], pkt->pos[2]};
  const double nu_rf = pkt->nu_rf;
  double nu_cmf = pkt->nu_cmf;
  const double e_rf = pkt->e_rf;
  double e_cmf = pkt->e_cmf;
  double prop_time = pkt->prop_time;
  (void)nu_rf; (void)e_rf; (void)e_cmf;
  double dist = 0.;
  double tau = 0.;
  ptrdiff_t binindex_start = get_linearbinindex(1e8 * CLIGHT / nu_cmf, ARTIS_EXPOPAC_LAMBDAMIN, ARTIS_EXPOPAC_DELTALAMBDA);
  if (binindex_start < -1) binindex_start = -1;
  for (ptrdiff_t binindex = binindex_start; binindex < ARTIS_EXPOPAC_NBINS; binindex++) {
    const double next_bin_edge_nu = (binindex < 0) ? get_expopac_bin_nu_upper(0) : get_expopac_bin_nu_lower(binindex);
    const double binedgedist = get_linedistance(prop_time, nu_cmf, next_bin_edge_nu, dnu_on_dl);
    const double chi_cont = chi_total(chi) * doppler;
    double chi_bb_expansionopac = 0.;
    if (binindex >= 0) {
      const float kappa = cell_expansionopacities(o, c)[binindex];
      chi_bb_expansionopac = kappa * o->cs->rho[c]; /* float product: kappa and get_rho() are floats */
    }
    const double chi_tot = chi_cont + chi_bb_expansionopac;
    if (chi_tot * binedgedist > tau_rnd - tau) {
#if ARTIS_OPT_RPKT_BB_THERMALISATION
      (void)cc; (void)mastate;
      const double edist = dmax(dist + ((tau_rnd - tau) / chi_tot), 0.);
      *is_bb = rng_uniform(pkt->rngstate) < chi_bb_expansionopac / chi_tot;
      return edist;
#else
      /* re-trace this bin line by line */
      artis_packet pkt_bin_start = *pkt;
      pkt_bin_start.pos[0] = pos[0]; pkt_bin_start.pos[1] = pos[1]; pkt_bin_start.pos[2] = pos[2];
      pkt_bin_start.nu_rf = nu_rf;
      pkt_bin_start.nu_cmf = nu_cmf;
      pkt_bin_start.e_rf = e_rf;
      pkt_bin_start.e_cmf = e_cmf;
      pkt_bin_start.prop_time = o->ts.mid; /* expansion opacity was calculated at t_mid, so match it */
      pkt_bin_start.next_trans = -1;
      int next_trans = -1;
      const double edist_after_bin = get_possible_event(o, cc, &pkt_bin_start, chi, mastate, tau_rnd - tau, DBL_MAXV, 0., dnu_on_dl,
                                                        doppler, &next_trans, is_bb);
      dist = dist + edist_after_bin;
      return dist;
#endif
    }
    tau += chi_tot * binedgedist;
    dist += binedgedist;
#if !ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT
    move_pkt_withtime_raw(pos, pkt->dir, &prop_time, nu_rf, &nu_cmf, e_rf, &e_cmf, binedgedist);
#else
    pos[0] += (pkt->dir[0] * binedgedist);
    pos[1] += (pkt->dir[1] * binedgedist);
    pos[2] += (pkt->dir[2] * binedgedist);
    prop_time += binedgedist / CLIGHT_PROP;
    nu_cmf = pkt->nu_cmf + (dnu_on_dl * dist);
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
    e_cmf = nu_cmf * e_rf / nu_rf; /* rpkt.cc:303-307 */
#endif
#endif
    if (nu_cmf <= nu_cmf_abort) {
      *is_bb = 0;
      return DBL_MAXV;
    }
  }
  const double chi_cont = chi_total(chi) * doppler;
  *is_bb = 0;
  if (chi_cont > 0.) return dist + ((tau_rnd - tau) / chi_cont);
  return DBL_MAXV;
}
#endif

static void emit_rpkt(artis_packet *p) {
  p->type = ARTIS_TYPE_RPKT;
  double dir_cmf[3];
  get_rand_isotropic_unitvec(p->rngstate, dir_cmf);
  double vel_vec[3];
  get_velocity(p->pos, -p->prop_time, vel_vec);
  angle_ab(dir_cmf, vel_vec, p->dir);
  set_pkt_restframe_from_cmf(p);
#if ARTIS_OPT_POL_ON
  p->stokes_u = 0.;
  p->stokes_q = 0.;
#endif
  p->em_pos[0] = p->pos[0];
  p->em_pos[1] = p->pos[1];
  p->em_pos[2] = p->pos[2];
  p->em_time = (float)p->prop_time;
}

/* electron_scatter_rpkt rpkt.cc:331 */
static void electron_scatter_rpkt(artis_packet *p) {
  p->type = ARTIS_TYPE_RPKT;
  double vel_vec[3];
  get_velocity(p->pos, p->prop_time, vel_vec);
  double old_dir_cmf[3], q_i_cmf = 0., u_i_cmf = 0.;
#if ARTIS_OPT_POL_ON
  frame_transform(p->dir, p->stokes_q, p->stokes_u, vel_vec, old_dir_cmf, &q_i_cmf, &u_i_cmf);
#else
  angle_ab(p->dir, vel_vec, old_dir_cmf);
  (void)q_i_cmf;
  (void)u_i_cmf;
#endif
  double M = 0., phisc = 0.;
#if ARTIS_OPT_DIPOLE
  {
    double pfn = 0., x = 1.;
    while (x > pfn) {
      M = (2. * rng_uniform_pos(p->rngstate)) - 1.;
      const double musquared = pow2(M);
      phisc = 2 * PI * rng_uniform(p->rngstate);
      pfn = (musquared + 1) + ((musquared - 1) * ((cos(2 * phisc) * q_i_cmf) + (sin(2 * phisc) * u_i_cmf)));
      x = 2. * rng_uniform(p->rngstate);
    }
  }
#else
  M = (2. * rng_uniform(p->rngstate)) - 1.;
  phisc = 2 * PI * rng_uniform(p->rngstate);
#endif
  double new_dir_cmf[3];
  const double cos_tsc = M;
  const double sin_tsc = sqrt(1. - pow2(M));
  if (fabs(old_dir_cmf[2]) < 0.99999) {
    const double sin_polar = sqrt(1. - pow2(old_dir_cmf[2]));
    const double common_factor = sin_tsc / sin_polar;
    const double cos_phisc = cos(phisc);
    const double sin_phisc = sin(phisc);
    new_dir_cmf[0] = (common_factor * ((old_dir_cmf[1] * sin_phisc) - (old_dir_cmf[0] * old_dir_cmf[2] * cos_phisc))) + (old_dir_cmf[0] * cos_tsc);
    new_dir_cmf[1] = (common_factor * ((-old_dir_cmf[0] * sin_phisc) - (old_dir_cmf[1] * old_dir_cmf[2] * cos_phisc))) + (old_dir_cmf[1] * cos_tsc);
    new_dir_cmf[2] = (sin_tsc * cos_phisc * sin_polar) + (old_dir_cmf[2] * cos_tsc);
  } else {
    new_dir_cmf[0] = sin_tsc * cos(phisc);
    new_dir_cmf[1] = sin_tsc * sin(phisc);
    new_dir_cmf[2] = (old_dir_cmf[2] > 0) ? cos_tsc : -cos_tsc;
  }
#if ARTIS_OPT_POL_ON
  {
    double nd[3], q, u;
    scatter_polarisation_to_rf(old_dir_cmf, new_dir_cmf, q_i_cmf, u_i_cmf, vel_vec, nd, &q, &u);
    p->dir[0] = nd[0]; p->dir[1] = nd[1]; p->dir[2] = nd[2];
    p->stokes_q = q;
    p->stokes_u = u;
  }
#else
  {
    const double negvel[3] = {-vel_vec[0], -vel_vec[1], -vel_vec[2]};
    angle_ab(new_dir_cmf, negvel, p->dir);
  }
#endif
  set_pkt_restframe_from_cmf(p);
}

/* ------------------------------------------------------------------ Gauss-Kronrod, gausskronrod.h */
/* 31-point Kronrod rule (QUADPACK dqk31 / Boost.Math tables as the reference stores them, gausskronrod.h:38-90).
 * The literals are long double in the reference and converted to double; 20+ digits give the same double. */
static const double gk31_abscissa[16] = {
    0.00000000000000000000000000000000000e+00, 1.01142066918717499027074231447392339e-01,
    2.01194093997434522300628303394596208e-01, 2.99180007153168812166780024266388963e-01,
    3.94151347077563369897207370981045468e-01, 4.85081863640239680693655740232350613e-01,
    5.70972172608538847537226737253910641e-01, 6.50996741297416970533735895313274693e-01,
    7.24417731360170047416186054613938010e-01, 7.90418501442465932967649294817947347e-01,
    8.48206583410427216200648320774216851e-01, 8.97264532344081900882509656454495883e-01,
    9.37273392400705904307758947710209471e-01, 9.67739075679139134257347978784337225e-01,
    9.87992518020485428489565718586612581e-01, 9.98002298693397060285172840152271209e-01};
static const double gk31_weights[16] = {
    1.01330007014791549017374792767492547e-01, 1.00769845523875595044946662617569722e-01,
    9.91735987217919593323931734846031311e-02, 9.66427269836236785051799076275893351e-02,
    9.31265981708253212254868727473457186e-02, 8.85644430562117706472754436937743032e-02,
    8.30805028231330210382892472861037896e-02, 7.68496807577203788944327774826590067e-02,
    6.98541213187282587095200770991474758e-02, 6.20095678006706402851392309608029322e-02,
    5.34815246909280872653431472394302968e-02, 4.45897513247648766082272993732796902e-02,
    3.53463607913758462220379484783600481e-02, 2.54608473267153201868740010196533594e-02,
    1.50079473293161225383747630758072681e-02, 5.37747987292334898779205143012764982e-03};
static const double gk31_gauss_weights[8] = {
    2.02578241925561272880620199967519315e-01, 1.98431485327111576456118326443839325e-01,
    1.86161000015562211026800561866422825e-01, 1.66269205816993933553200860481208811e-01,
    1.39570677926154314447804794511028323e-01, 1.07159220467171935011869546685869303e-01,
    7.03660474881081247092674164506673385e-02, 3.07532419961172683546283935772044177e-02};

typedef struct {
  const Oracle *o;
  const float *xs;
  double nu_edge;
  float T_e;
  int testmode;   /* != 0: one of the analytic integrands of artis_oracle_gk31_test() (pins the integrator only) */
  double p0, p1;
} FbIntegrand;

/* Analytic integrands used only to pin this file's Gauss-Kronrod restatement against the reference's own
 * gausskronrod.h (oracle/ref_harness/ref_gk31_main.cc holds the same three formulas). */
static double gk31_test_integrand(int mode, double p0, double p1, double x) {
  if (mode == 1) return exp(-p0 * x) * (1. + floor(x * p1)); /* steps: deep recursion */
  if (mode == 2) return x * x * exp(-p0 * x) / (1. + (p1 * x * x * x));
  return sqrt(fabs(x - p0)) + p1; /* kink */
}
/* alpha_sp_E_integrand ratecoeff.cc:84 */
static double alpha_sp_E_integrand(const FbIntegrand *f, double nu_minus_nu_edge) {
  if (f->testmode != 0) return gk31_test_integrand(f->testmode, f->p0, f->p1, nu_minus_nu_edge);
  const double nu = f->nu_edge + nu_minus_nu_edge;
  const float sigma_bf = photoionisation_crosssection_fromtable(f->o, f->xs, f->nu_edge, nu);
  return (2 / CLIGHTSQUARED) * sigma_bf * pow3(nu) / f->nu_edge * exp(-HOVERKB * nu_minus_nu_edge / f->T_e);
}
/* integrate_non_adaptive_m1_1<31> gausskronrod.h:173 on ff(x) = f(scale*x + mean) */
static double gk31_m1_1(const FbIntegrand *f, double scale, double mean, double *error) {
  /* gauss_order 15 is odd: centre is a Gauss node; gauss_start 2, kronrod_start 1 */
  const double f_centre = alpha_sp_E_integrand(f, (scale * 0.) + mean);
  double kronrod_result = f_centre * gk31_weights[0];
  double gauss_result = 0.;
  gauss_result += f_centre * gk31_gauss_weights[0];
  for (unsigned i = 2; i < 16; i += 2) {
    const double fp = alpha_sp_E_integrand(f, (scale * gk31_abscissa[i]) + mean);
    const double fm = alpha_sp_E_integrand(f, (scale * -gk31_abscissa[i]) + mean);
    kronrod_result += (fp + fm) * gk31_weights[i];
    gauss_result += (fp + fm) * gk31_gauss_weights[i / 2];
  }
  for (unsigned i = 1; i < 16; i += 2) {
    const double fp = alpha_sp_E_integrand(f, (scale * gk31_abscissa[i]) + mean);
    const double fm = alpha_sp_E_integrand(f, (scale * -gk31_abscissa[i]) + mean);
    kronrod_result += (fp + fm) * gk31_weights[i];
  }
  *error = dmax(fabs(kronrod_result - gauss_result), fabs(kronrod_result * 2.220446049250313e-16 * 2));
  return kronrod_result;
}
/* recursive_adaptive_integrate<31> gausskronrod.h:208 */
static double gk31_recursive(const FbIntegrand *f, double tol, double a, double b, unsigned max_levels, double abs_tol,
                             double *error) {
  double error_local = 0.;
  const double mean = (b + a) / 2;
  const double scale = (b - a) / 2;
  const double r1 = gk31_m1_1(f, scale, mean, &error_local);
  double estimate = scale * r1;
  const double abs_tol1 = fabs(estimate * tol);
  if (abs_tol == 0) abs_tol = abs_tol1;
  if ((max_levels != 0) && (abs_tol1 < error_local) && (abs_tol < error_local)) {
    const double mid = (a + b) / 2;
    estimate = gk31_recursive(f, tol, a, mid, max_levels - 1, abs_tol / 2, error);
    estimate += gk31_recursive(f, tol, mid, b, max_levels - 1, abs_tol / 2, &error_local);
    *error += error_local;
    return estimate;
  }
  *error = error_local;
  return estimate;
}
/* integrator<31> integrator.h:48 -> gauss_kronrod_integrate gausskronrod.h:244 (max_depth 15) */
static double integrator31(const FbIntegrand *f, double a, double b, double epsrel, double *abserr) {
  if (a == b) return 0.;
  if (b < a) return -gk31_recursive(f, epsrel, b, a, 15, 0., abserr);
  return gk31_recursive(f, epsrel, a, b, 15, 0., abserr);
}

/* gauss_kronrod_integrate<31>(f, a, b, 15, tol, &error) of this restatement on an analytic integrand */
double artis_oracle_gk31_test(int mode, double p0, double p1, double a, double b, double tol, double *error) {
  FbIntegrand f = {NULL, NULL, 0., 0.f, mode, p0, p1};
  double err = 0.;
  const double r = integrator31(&f, a, b, tol, &err);
  if (error) *error = err;
  return r;
}

/* select_continuum_nu ratecoeff.cc:563 */
static double select_continuum_nu(Oracle *o, int element, int lowerion, int lower, int t, float T_e, uint32_t rng[4]) {
  const int ul = ionlevelstart(o, element, lowerion) + lower;
  const double E_threshold = get_phixs_threshold(o, element, lowerion, lower, t);
  const double nu_threshold = (1. / H_PLANCK) * E_threshold;
  const double nu_max_phixs = nu_threshold * o->last_phixs_nuovernuedge;
  const int npieces = o->m->NPHIXSPOINTS;
  FbIntegrand f = {o, get_phixs_table(o, ul), nu_threshold, T_e, 0, 0., 0.};
  const double zrand = 1. - rng_uniform(rng);
  const double nu_range = nu_max_phixs - nu_threshold;
  const double deltanu = nu_range / npieces;
  double error = NAN;
  const double RATECOEFF_INTEGRAL_ACCURACY = 1e-3; /* ratecoeff.cc:37 */
  const double total = integrator31(&f, 0., nu_range, RATECOEFF_INTEGRAL_ACCURACY, &error);
  if (!(total > 0.) || !isfinite(total)) return nu_threshold;
  double tail_prev = total;
  double tail = total;
  int i = 1;
  for (; i < npieces; i++) {
    tail_prev = tail;
    const double low = i * deltanu;
    tail = integrator31(&f, low, nu_range, RATECOEFF_INTEGRAL_ACCURACY, &error);
    if (zrand >= tail / total) break;
  }
  double nuoffset = 0.;
  if (i < npieces) {
    nuoffset = (tail != tail_prev) ? ((total * zrand) - tail_prev) / (tail - tail_prev) * deltanu : 0.;
  } else if (tail > 0.) {
    nuoffset = (tail - (total * zrand)) / tail * deltanu;
  }
  return nu_threshold + ((i - 1) * deltanu) + nuoffset;
}

/* ------------------------------------------------------------------ macroatom.cc */

#if ARTIS_OPT_VPKT_ON
/* ------------------------------------------------------------------ vpkt.cc: virtual packets
 * Restated from vpkt.cc:116-490 and :948-1010. The per-call continuum opacity of a virtual packet is a fresh
 * ContinuumOpacity per traced direction (THREADLOCALONHOST is empty in the reference's GPU build, constants.h:110). */
#define PARSEC 3.0857e+18 /* constants.h:39 */
static inline double get_loggrid_edge(double minvalue, double dlog, double index) { return exp(log(minvalue) + (index * dlog)); } /* sn3d.h:142 */
static inline ptrdiff_t get_logbinindex(double value, double minvalue, double dlog, ptrdiff_t nbins) { /* sn3d.h:134 */
  ptrdiff_t i = (ptrdiff_t)floor((log(value) - log(minvalue)) / dlog);
  return i < 0 ? 0 : (i > nbins - 1 ? nbins - 1 : i);
}
static inline double vspec_dlogt(void) { return (log(ARTIS_VSPEC_TIMEMAX) - log(ARTIS_VSPEC_TIMEMIN)) / ARTIS_VSPEC_TIMEBINS; } /* vpkt.cc:106 */
static inline double vspec_dlognu(void) { return (log(ARTIS_VSPEC_NUMAX) - log(ARTIS_VSPEC_NUMIN)) / ARTIS_VSPEC_NUBINS; }   /* vpkt.cc:107 */
/* add_to_vspecpol vpkt.cc:116 (delta_t and delta_freq_vspec are floats there: init_vspecpol vpkt.cc:491-512) */
static void add_to_vspecpol(Oracle *o, double nu_rf, double e_rf, double prob, double q_rf, double u_rf, int obsdirindex,
                            int opachoiceindex, double t_arrive) {
  const artis_model *m = o->m;
  if (t_arrive <= ARTIS_VSPEC_TIMEMIN || t_arrive >= ARTIS_VSPEC_TIMEMAX || nu_rf <= ARTIS_VSPEC_NUMIN || nu_rf >= ARTIS_VSPEC_NUMAX) return;
  const int nt = (int)get_logbinindex(t_arrive, ARTIS_VSPEC_TIMEMIN, vspec_dlogt(), ARTIS_VSPEC_TIMEBINS);
  const int nnu = (int)get_logbinindex(nu_rf, ARTIS_VSPEC_NUMIN, vspec_dlognu(), ARTIS_VSPEC_NUBINS);
  const float lower_time = (float)get_loggrid_edge(ARTIS_VSPEC_TIMEMIN, vspec_dlogt(), nt);
  const float delta_t = (float)(get_loggrid_edge(ARTIS_VSPEC_TIMEMIN, vspec_dlogt(), nt + 1) - lower_time);
  const float lower_freq = (float)get_loggrid_edge(ARTIS_VSPEC_NUMIN, vspec_dlognu(), nnu);
  const float delta_freq = (float)(get_loggrid_edge(ARTIS_VSPEC_NUMIN, vspec_dlognu(), nnu + 1) - lower_freq);
  const int ind_comb = (m->vpkt_nspectraperobsdir * obsdirindex) + opachoiceindex;
  const double pktcontrib = e_rf / delta_t / delta_freq / 4.e12 / PI / PARSEC / PARSEC / m->vpkt_nprocs * 4 * PI;
  double *flux = o->est.vspecpol + ((((ptrdiff_t)nt * (m->vpkt_nobsdirections * m->vpkt_nspectraperobsdir) + ind_comb) * ARTIS_VSPEC_NUBINS + nnu) * 3);
  flux[0] += prob * pktcontrib;
  flux[1] += prob * q_rf * pktcontrib;
  flux[2] += prob * u_rf * pktcontrib;
}
/* add_to_vpkt_grid vpkt.cc:138 */
static void add_to_vpkt_grid(Oracle *o, double nu_rf, double e_rf, double prob, double stokes_q, double stokes_u, const double vel[3],
                             int wlbin, int obsdirindex, const double obsdir[3]) {
  const artis_model *m = o->m;
  double vref1, vref2;
  if (obsdir[0] == 1) {
    vref1 = vel[1];
    vref2 = vel[2];
  } else if (obsdir[0] == -1) {
    vref1 = -vel[1];
    vref2 = -vel[2];
  } else {
    const double crossterm = obsdir[1] * obsdir[2] / (1 + obsdir[0]);
    vref1 = (-obsdir[1] * vel[0]) + ((obsdir[0] + (pow2(obsdir[2]) / (1 + obsdir[0]))) * vel[1]) - (crossterm * vel[2]);
    vref2 = (-obsdir[2] * vel[0]) - (crossterm * vel[1]) + ((obsdir[0] + (pow2(obsdir[1]) / (1 + obsdir[0]))) * vel[2]);
  }
  if (fabs(vref1) >= m->vmax || fabs(vref2) >= m->vmax) return;
  const int ny = (int)((m->vmax - vref1) / (2 * m->vmax / ARTIS_VGRID_NY));
  const int nz = (int)((m->vmax - vref2) / (2 * m->vmax / ARTIS_VGRID_NZ));
  if (nu_rf > m->vpkt_nu_grid_min[wlbin] && nu_rf < m->vpkt_nu_grid_max[wlbin]) {
    double *flux = o->est.vgrid_flux + (((((ptrdiff_t)ny * ARTIS_VGRID_NZ + nz) * m->vpkt_grid_nwavelengthranges + wlbin) * m->vpkt_nobsdirections + obsdirindex) * 3);
    flux[0] += prob * e_rf;
    flux[1] += prob * stokes_q * e_rf;
    flux[2] += prob * stokes_u * e_rf;
  }
}
static int all_taus_past_taumax(const double *tau, int n, double tau_max) { /* vpkt.cc:111 */
  for (int i = 0; i < n; i++)
    if (!(tau[i] > tau_max)) return 0;
  return 1;
}
#define VPKT_MAXSPEC 16
/* the loop of trace_lines_to_dist (vpkt.cc:298-358): 0 when every opacity choice is past tau_max */
static int vpkt_trace_lines_to_dist(Oracle *o, const CellCache *cc, double dist_limit, double t_future, double nu_cmf, double dnu_on_dl,
                                    int *next_trans, double *tau_vpkt) {
  const artis_model *m = o->m;
  const double t_gridstate = o->ts.mid;
  while (1) {
    const int lineindex = closest_transition(m->line_nu, m->nlines, nu_cmf, *next_trans);
    if (lineindex < 0) {
      *next_trans = m->nlines + 1;
      break;
    }
    const double nutrans = m->line_nu[lineindex];
    *next_trans = lineindex + 1;
    const double ldist = get_linedistance(t_future, nu_cmf, nutrans, dnu_on_dl);
    if (ldist > dist_limit) {
      (*next_trans)--;
      break;
    }
    const double t_line = t_future + (ldist / CLIGHT_PROP);
    const double B_ul = m->line_B_ul[lineindex];
    const double B_lu = m->line_B_lu[lineindex];
    const double n_u = cc->levelpops[m->line_uniquelevelindex_upper[lineindex]]; /* calculate_levelpop(): what the cache holds */
    const double n_l = cc->levelpops[m->line_uniquelevelindex_lower[lineindex]];
    const double popscalefactor = pow3(t_gridstate / t_line);
    const double tau_line = dmax(0., ((B_lu * n_l) - (B_ul * n_u)) * popscalefactor * HCLIGHTOVERFOURPI * t_line);
    const int Z = m->elem_anumber[m->line_elementindex[lineindex]];
    for (int i = 0; i < m->vpkt_nspectraperobsdir; i++)
      if (m->vpkt_opacityexclusions[i] != -1 && m->vpkt_opacityexclusions[i] != Z) tau_vpkt[i] += tau_line;
    if (all_taus_past_taumax(tau_vpkt, m->vpkt_nspectraperobsdir, m->vpkt_tau_max)) return 0;
  }
  return 1;
}
/* trace_vpkt_direction vpkt.cc:183 */
static int trace_vpkt_direction(Oracle *o, const artis_packet *rpkt, double t_arrive, double nu_rf, double e_rf, double rpkt_doppler,
                                int obsdirindex, const double obsdir[3], int type_before_rpkt) {
  const artis_model *m = o->m;
  const int nspec = m->vpkt_nspectraperobsdir;
  int cellindex = rpkt->cellindex;
  int next_trans = rpkt->next_trans;
  double e_cmf = rpkt->e_cmf;
  double nu_cmf = rpkt->nu_cmf;
  double vpktpos[3] = {rpkt->pos[0], rpkt->pos[1], rpkt->pos[2]};
  int end_packet = 0;
  const double t_start = rpkt->prop_time;
  double t_future = t_start;
  const double t_gridstate = o->ts.mid;
  double tau_vpkt[VPKT_MAXSPEC];
  for (int i = 0; i < nspec; i++) tau_vpkt[i] = 0.;
  stat_inc(o, ARTIS_STAT_X_VPKT_CREATED);
  double vel_vec[3];
  get_velocity(rpkt->pos, t_start, vel_vec);
  double pn = 1 / (4 * PI);
  double q_rf = 0., u_rf = 0.;
  if (type_before_rpkt == ARTIS_TYPE_RPKT) {
    double old_dir_cmf[3], q_i_cmf = 0., u_i_cmf = 0.;
#if ARTIS_OPT_POL_ON
    frame_transform(rpkt->dir, rpkt->stokes_q, rpkt->stokes_u, vel_vec, old_dir_cmf, &q_i_cmf, &u_i_cmf);
#else
    angle_ab(rpkt->dir, vel_vec, old_dir_cmf);
#endif
    double obs_cmf[3];
    angle_ab(obsdir, vel_vec, obs_cmf);
    double new_dir_rf[3];
    scatter_polarisation_to_rf(old_dir_cmf, obs_cmf, q_i_cmf, u_i_cmf, vel_vec, new_dir_rf, &q_rf, &u_rf);
    /* pn of scatter_polarisation_to_rf (vectors.h:353): the phase function of the scattering */
    {
      double ref1[3], ref2[3];
      meridian(old_dir_cmf, ref1, ref2);
      const double i1 = get_rot_angle(old_dir_cmf, obs_cmf, ref1, ref2);
      const double q_old = (q_i_cmf * cos(2 * i1)) - (u_i_cmf * sin(2 * i1));
      const double musquared = pow2(dot3(old_dir_cmf, obs_cmf));
      pn = 3. / (16. * PI) * (1. + musquared + ((musquared - 1.) * q_old));
    }
  }
  pn /= pow2(rpkt_doppler);

  int mgi = propcell_nonemptymgi(o, cellindex); /* model cells are the non-empty cells here */
  ContOpacity chi_vpkt_cont;
  memset(&chi_vpkt_cont, 0, sizeof(chi_vpkt_cont));
  chi_vpkt_cont.nonemptymgi = -1;
  chi_vpkt_cont.nu = NAN;
  while (!end_packet) {
    int next_cellindex = -1;
    const double boundarydist = boundary_distance(o, obsdir, vpktpos, t_future, cellindex, &next_cellindex);
    if (mgi < 0) {
      next_trans = -1;
    } else if (boundarydist > 0) {
      const int c = mgi;
      cellcache_populate(o, c);
      const CellCache *cc = &o->cache[c];
      const double doppler = doppler_nucmf_on_nurf(vpktpos, obsdir, t_future);
      /* calculate_chi_rpkt_cont<false> rpkt.cc:1021: the same sums without the packet's phixslist */
      if (!((c == chi_vpkt_cont.nonemptymgi) && (fabs((chi_vpkt_cont.nu / nu_cmf) - 1.0) < 1e-4))) {
        chi_vpkt_cont.chi_freefree_heat = calculate_chi_ffheating(o, cc, c, nu_cmf);
        chi_vpkt_cont.chi_escatter = SIGMA_T * cell_nne(o, c);
        int dummy = -1;
        chi_vpkt_cont.chi_boundfree = calculate_chi_bf_gammacontr(o, cc, c, nu_cmf, NULL, 1, DBL_MAXV, &dummy, NULL);
        chi_vpkt_cont.nonemptymgi = c;
        chi_vpkt_cont.nu = nu_cmf;
      }
      const double densityscalefactor = pow3(t_gridstate / t_future);
      const double chi_escatter = chi_vpkt_cont.chi_escatter * densityscalefactor;
      const double chi_bf = chi_vpkt_cont.chi_boundfree * densityscalefactor;
      const double chi_ff = chi_vpkt_cont.chi_freefree_heat * pow2(densityscalefactor);
      const double chi_cont = chi_escatter + chi_bf + chi_ff;
      for (int i = 0; i < nspec; i++) {
        double chi_cont_thischoice = chi_cont;
        if (m->vpkt_opacityexclusions[i] == -2) {
          chi_cont_thischoice -= chi_bf;
        } else if (m->vpkt_opacityexclusions[i] == -3) {
          chi_cont_thischoice -= chi_ff;
        } else if (m->vpkt_opacityexclusions[i] == -4) {
          chi_cont_thischoice -= chi_escatter;
        }
        tau_vpkt[i] += chi_cont_thischoice * boundarydist * doppler;
      }
      if (all_taus_past_taumax(tau_vpkt, nspec, m->vpkt_tau_max)) return 0;
      const double pos_boundary[3] = {vpktpos[0] + (obsdir[0] * boundarydist), vpktpos[1] + (obsdir[1] * boundarydist),
                                      vpktpos[2] + (obsdir[2] * boundarydist)};
      const double nu_cmf_boundary =
          dmin(nu_rf * doppler_nucmf_on_nurf(pos_boundary, obsdir, t_future + (boundarydist / CLIGHT_PROP)), nu_cmf);
      const double dnu_on_dl = (nu_cmf_boundary - nu_cmf) / boundarydist;
#if ARTIS_OPT_VPKT_USE_EXPANSION_OPACITIES
      ptrdiff_t binindex_start = get_linearbinindex(1e8 * CLIGHT / nu_cmf, ARTIS_EXPOPAC_LAMBDAMIN, ARTIS_EXPOPAC_DELTALAMBDA);
      if (binindex_start < -1) binindex_start = -1;
      if (binindex_start < ARTIS_EXPOPAC_NBINS) {
        const double first_bin_edge_nu = (binindex_start < 0) ? get_expopac_bin_nu_upper(0) : get_expopac_bin_nu_lower(binindex_start);
        const double first_bin_edge_dist = get_linedistance(t_future, nu_cmf, first_bin_edge_nu, dnu_on_dl);
        const double line_by_line_limit = dmin(first_bin_edge_dist, boundarydist);
        next_trans = -1;
        if (!vpkt_trace_lines_to_dist(o, cc, line_by_line_limit, t_future, nu_cmf, dnu_on_dl, &next_trans, tau_vpkt)) return 0;
        double dist = line_by_line_limit;
        if (dist < boundarydist) {
          const float *kappa_bins = cell_expansionopacities(o, c);
          for (ptrdiff_t binindex = binindex_start + 1; binindex < ARTIS_EXPOPAC_NBINS; binindex++) {
            const double next_bin_edge_nu = get_expopac_bin_nu_lower(binindex);
            const double binedgedist = get_linedistance(t_future, nu_cmf, next_bin_edge_nu, dnu_on_dl);
            const float kappa = kappa_bins[binindex];
            const double chi_bb_expansionopac = kappa * o->cs->rho[c] * densityscalefactor; /* float product first */
            const double tau_bin = chi_bb_expansionopac * (dmin(binedgedist, boundarydist) - dist);
            dist = dmin(binedgedist, boundarydist);
            for (int i = 0; i < nspec; i++)
              if (m->vpkt_opacityexclusions[i] != -1) tau_vpkt[i] += tau_bin;
            if (all_taus_past_taumax(tau_vpkt, nspec, m->vpkt_tau_max)) return 0;
            if (dist >= boundarydist) break;
          }
        }
      }
#else
      if (!vpkt_trace_lines_to_dist(o, cc, boundarydist, t_future, nu_cmf, dnu_on_dl, &next_trans, tau_vpkt)) return 0;
#endif
    }
    move_pkt_withtime_raw(vpktpos, obsdir, &t_future, nu_rf, &nu_cmf, e_rf, &e_cmf, boundarydist);
    if (next_cellindex >= 0) {
      if (next_cellindex != cellindex) snap_pos_to_cell(o, vpktpos, t_future, next_cellindex);
      cellindex = next_cellindex;
      mgi = propcell_nonemptymgi(o, cellindex);
      if (mgi >= 0 && o->cs->thick[mgi] != ARTIS_CELL_THIN) return 0;
    } else {
      end_packet = 1;
    }
  }
  if (type_before_rpkt == ARTIS_TYPE_RPKT) {
    stat_inc(o, ARTIS_STAT_X_VPKT_ESC_RPKT);
  } else if (type_before_rpkt == ARTIS_TYPE_KPKT) {
    stat_inc(o, ARTIS_STAT_X_VPKT_ESC_KPKT);
  } else if (type_before_rpkt == ARTIS_TYPE_MA) {
    stat_inc(o, ARTIS_STAT_X_VPKT_ESC_MA);
  }
  for (int i = 0; i < nspec; i++) {
    const double prob = pn * exp(-tau_vpkt[i]);
    if (!isfinite(prob)) ORACLE_FAIL(o, "vpkt: prob not finite");
    add_to_vspecpol(o, nu_rf, e_rf, prob, q_rf, u_rf, obsdirindex, i, t_arrive);
  }
  if (m->vpkt_vgrid_on) {
    const double prob = pn * exp(-tau_vpkt[0]);
    for (int wlbin = 0; wlbin < m->vpkt_grid_nwavelengthranges; wlbin++)
      if ((nu_rf > m->vpkt_nu_grid_min[wlbin] && nu_rf < m->vpkt_nu_grid_max[wlbin]) && (t_arrive > m->vpkt_tmin_grid && t_arrive < m->vpkt_tmax_grid))
        add_to_vpkt_grid(o, nu_rf, e_rf, prob, q_rf, u_rf, vel_vec, wlbin, obsdirindex, obsdir);
  }
  return 1;
}
/* trace_vpkts vpkt.cc:948 */
static void trace_vpkts(Oracle *o, const artis_packet *pkt, int type_before_rpkt) {
  const artis_model *m = o->m;
  const int c = propcell_nonemptymgi(o, pkt->cellindex);
  if (o->cs->thick[c] != ARTIS_CELL_THIN) return;
  for (int obsdirindex = 0; obsdirindex < m->vpkt_nobsdirections; obsdirindex++) {
    const double ct = m->vpkt_obsdirs_costheta[obsdirindex], ph = m->vpkt_obsdirs_phi[obsdirindex];
    const double obsdir[3] = {sqrt(1 - (ct * ct)) * cos(ph), sqrt(1 - (ct * ct)) * sin(ph), ct};
    const double t_arrive = pkt->prop_time - (dot3(pkt->pos, obsdir) / CLIGHT_PROP);
    if (t_arrive >= m->vpkt_timemin_input && t_arrive <= m->vpkt_timemax_input) {
      const double doppler = doppler_nucmf_on_nurf(pkt->pos, obsdir, pkt->prop_time);
      const double nu_rf = pkt->nu_cmf / doppler;
      const double e_rf = pkt->e_cmf / doppler;
      for (int i = 0; i < m->vpkt_nwavelengthranges; i++) {
        if ((nu_rf > m->vpkt_numin_input[i] && nu_rf < m->vpkt_numax_input[i]) ||
            (pkt->absorptionfreq > m->vpkt_numin_input[i] && pkt->absorptionfreq < m->vpkt_numax_input[i])) {
          (void)trace_vpkt_direction(o, pkt, t_arrive, nu_rf, e_rf, doppler, obsdirindex, obsdir, type_before_rpkt);
          break;
        }
      }
    }
  }
}
#endif

static int index_upperbound(const double *a, int n, double target) { return upper_bound_d(a, n, target); } /* sn3d.h:85 */

/* do_macroatom_raddeexcitation macroatom.cc:204 */
static void do_macroatom_raddeexcitation(Oracle *o, const CellCache *cc, artis_packet *p, int start, int ul, int activatingline,
                                         double epsilon_current, double totalrate) {
  const artis_model *m = o->m;
  const double targetval = rng_uniform(p->rngstate) * totalrate;
  const int ndowntrans = m->level_ndowntrans[ul];
  const double *sums = cc->matrans + m->level_matransblock_start[ul]; /* get_sum_epstrans_rad_deexc_exceptlast macroatom.cc:58 */
  const int downtransindex = index_upperbound(sums, ndowntrans - 1, targetval);
  const int startdown = m->level_alltrans_startdown[ul];
  const int lineindex = m->alltrans_lineindex[startdown + downtransindex];
  if (lineindex == activatingline) stat_inc(o, ARTIS_STAT_RESONANCESCATTERINGS);
  const int lul = start + m->alltrans_targetlevelindex[startdown + downtransindex];
  const double epsilon_trans = epsilon_current - epsilon(o, lul);
  const double oldnucmf = p->nu_cmf;
  p->nu_cmf = epsilon_trans / H_PLANCK;
  if (activatingline >= 0) stat_inc(o, (oldnucmf < p->nu_cmf) ? ARTIS_STAT_UPSCATTER : ARTIS_STAT_DOWNSCATTER);
  stat_inc(o, ARTIS_STAT_MA_DEACTIVATION_BB);
  emit_rpkt(p);
  p->next_trans = lineindex + 1;
  p->emissiontype = lineindex;
  p->nscatterings = 0;
}

/* do_macroatom_radrecomb macroatom.cc:248 */
static int do_macroatom_radrecomb(Oracle *o, artis_packet *p, int c, int element, int upperion, int upperionlevel, double rad_recomb) {
  const float T_e = o->cs->Te[c];
  const float clumpednne = cell_clumpednne(o, c);
  const double epsilon_current = epsilon(o, ionlevelstart(o, element, upperion) + upperionlevel);
  const double targetval = rng_uniform(p->rngstate) * rad_recomb;
  double rate = 0;
  const int nlevels = get_nlevels_ionising(o, element, upperion - 1);
  const int lstart = ionlevelstart(o, element, upperion - 1);
  int lowerionlevel = -1;
  int selected_t = -1;
  for (int l = 0; l < nlevels; l++) {
    const int t = find_phixstargetindex(o, lstart + l, upperionlevel);
    if (t < 0) continue;
    const double epsilon_trans = epsilon_current - epsilon(o, lstart + l);
    const double R = rad_recombination_ratecoeff(o, T_e, clumpednne, element, upperion, l, t);
    rate += R * epsilon_trans;
    if (targetval < rate) {
      lowerionlevel = l;
      selected_t = t;
      break;
    }
  }
  if (lowerionlevel < 0) {
    ORACLE_FAIL(o, "do_macroatom_radrecomb: no level selected");
    return 0;
  }
  const int lowerion = upperion - 1;
  p->nu_cmf = select_continuum_nu(o, element, lowerion, lowerionlevel, selected_t, T_e, p->rngstate);
  stat_inc(o, ARTIS_STAT_MA_DEACTIVATION_FB);
  emit_rpkt(p);
  p->next_trans = -1;
  p->emissiontype = get_emtype_continuum(o, lstart + lowerionlevel, selected_t);
  p->nscatterings = 0;
  return lowerionlevel;
}

/* do_macroatom_ionisation macroatom.cc:298 */
static int do_macroatom_ionisation(Oracle *o, const CellCache *cc, int c, int element, int ion, int level, double epsilon_current,
                                   double internal_up_higher, uint32_t rng[4]) {
  const float T_e = o->cs->Te[c];
  const float clumpednne = cell_clumpednne(o, c);
  const double targetrate = rng_uniform(rng) * internal_up_higher;
  double rate = 0.;
  const int ul = ionlevelstart(o, element, ion) + level;
  const int nt = o->m->level_nphixstargets[ul];
  for (int t = 0; t < nt; t++) {
    const double epsilon_trans = get_phixs_threshold(o, element, ion, level, t);
    const double R = cc->corrphotoioncoeff[o->m->level_phixstargetstart[ul] + t];
    const double C = col_ionisation_ratecoeff(o, T_e, clumpednne, element, ion, level, t, epsilon_trans);
    rate += (R + C) * epsilon_current;
    if (rate > targetrate) return get_phixsupperlevel(o, ul, t);
  }
  ORACLE_FAIL(o, "do_macroatom_ionisation: no target selected");
  return 0;
}

/* do_macroatom macroatom.cc:360 */
static void do_macroatom(Oracle *o, artis_packet *p, const MacroAtomState *ma) {
  const artis_model *m = o->m;
  const int c = propcell_nonemptymgi(o, p->cellindex);
  const CellCache *cc = &o->cache[c];
  const float T_e = o->cs->Te[c];
  const float clumpednne = cell_clumpednne(o, c);
  const int element = ma->element;
  int ion = ma->ion;
  int level = ma->level;
  const int activatingline = ma->activatingline;
  int end_packet = 0;
  while (!end_packet && !o->error) {
    o->est.stats[ARTIS_STAT_X_MA_JUMPS]++;
    const int start = ionlevelstart(o, element, ion);
    const int ul = start + level;
    if (g_visit_hist) g_visit_hist[(int64_t)c * g_visit_nlevels + ul]++;
    const double epsilon_current = epsilon(o, ul);
    const double *levelrates = macroatom_levelrates(o, c, element, ion, level);
    double cumulative[ARTIS_MA_ACTION_COUNT];
    cumulative[0] = levelrates[0];
    for (int i = 1; i < ARTIS_MA_ACTION_COUNT; i++) cumulative[i] = cumulative[i - 1] + levelrates[i]; /* std::partial_sum */
    const double total_rate = cumulative[ARTIS_MA_ACTION_COUNT - 1];
    if (!(total_rate > 0.)) {
      ORACLE_FAIL(o, "do_macroatom: total_rate <= 0");
      return;
    }
    const double randomrate = rng_uniform(p->rngstate) * total_rate;
    int selected_action = index_upperbound(cumulative, ARTIS_MA_ACTION_COUNT, randomrate);
    if (selected_action > ARTIS_MA_ACTION_COUNT - 1) selected_action = ARTIS_MA_ACTION_COUNT - 1;
    stat_inc(o, ARTIS_STAT_INTERACTIONS);
    switch (selected_action) {
      case ARTIS_MA_ACTION_RADDEEXC:
        do_macroatom_raddeexcitation(o, cc, p, start, ul, activatingline, epsilon_current, levelrates[ARTIS_MA_ACTION_RADDEEXC]);
        end_packet = 1;
        break;
      case ARTIS_MA_ACTION_COLDEEXC:
        stat_inc(o, ARTIS_STAT_MA_DEACTIVATION_COLLDEEXC);
        p->type = ARTIS_TYPE_KPKT;
        end_packet = 1;
#if !ARTIS_OPT_DIRECT_COL_HEAT
        o->est.colheatingestimator[c] += p->e_cmf;
#endif
        break;
      case ARTIS_MA_ACTION_INTERNALDOWNSAME: {
        const double targetval = rng_uniform(p->rngstate) * levelrates[ARTIS_MA_ACTION_INTERNALDOWNSAME];
        const int ndowntrans = m->level_ndowntrans[ul];
        const double *sums = cc->matrans + m->level_matransblock_start[ul] + ndowntrans; /* macroatom.cc:44 */
        const int downtransindex = index_upperbound(sums, ndowntrans - 1, targetval);
        level = m->alltrans_targetlevelindex[m->level_alltrans_startdown[ul] + downtransindex];
        break;
      }
      case ARTIS_MA_ACTION_RADRECOMB:
        level = do_macroatom_radrecomb(o, p, c, element, ion, level, levelrates[ARTIS_MA_ACTION_RADRECOMB]);
        ion -= 1;
        end_packet = 1;
        break;
      case ARTIS_MA_ACTION_COLRECOMB:
        stat_inc(o, ARTIS_STAT_MA_DEACTIVATION_COLLRECOMB);
        p->type = ARTIS_TYPE_KPKT;
        end_packet = 1;
#if !ARTIS_OPT_DIRECT_COL_HEAT
        o->est.colheatingestimator[c] += p->e_cmf;
#endif
        break;
      case ARTIS_MA_ACTION_INTERNALDOWNLOWER: {
        stat_inc(o, ARTIS_STAT_MA_INTERNALDOWNLOWER);
        const double targetrate = rng_uniform(p->rngstate) * levelrates[ARTIS_MA_ACTION_INTERNALDOWNLOWER];
        double rate = 0.;
        const int nlevels = get_nlevels_ionising(o, element, ion - 1);
        int lower = -1;
        const int lstart = ionlevelstart(o, element, ion - 1);
        for (int l = 0; l < nlevels; l++) {
          const int t = find_phixstargetindex(o, lstart + l, level);
          if (t < 0) continue;
          const double epsilon_target = epsilon(o, lstart + l);
          const double epsilon_trans = epsilon_current - epsilon_target;
          const double R = rad_recombination_ratecoeff(o, T_e, clumpednne, element, ion, l, t);
          const double C = col_recombination_ratecoeff(o, T_e, clumpednne, element, ion, l, t, epsilon_trans);
          rate += (R + C) * epsilon_target;
          if (rate > targetrate) {
            lower = l;
            break;
          }
        }
        if (lower < 0) {
          ORACLE_FAIL(o, "do_macroatom: internal down lower found no level");
          return;
        }
        ion--;
        level = lower;
        break;
      }
      case ARTIS_MA_ACTION_INTERNALUPSAME: {
        const int ndowntrans = m->level_ndowntrans[ul];
        const int nuptrans = m->level_nuptrans[ul];
        const double *sums = cc->matrans + m->level_matransblock_start[ul] + (2 * ndowntrans); /* macroatom.cc:51 */
        const double targetval = rng_uniform(p->rngstate) * levelrates[ARTIS_MA_ACTION_INTERNALUPSAME];
        const int uptransindex = index_upperbound(sums, nuptrans - 1, targetval);
        const int startup = m->level_alltrans_startdown[ul] + ndowntrans;
        level = m->alltrans_targetlevelindex[startup + uptransindex];
        break;
      }
      case ARTIS_MA_ACTION_INTERNALUPHIGHER:
        stat_inc(o, ARTIS_STAT_MA_INTERNALUPHIGHER);
        level = do_macroatom_ionisation(o, cc, c, element, ion, level, epsilon_current, levelrates[ARTIS_MA_ACTION_INTERNALUPHIGHER],
                                        p->rngstate);
        ion += 1;
        break;
#if ARTIS_OPT_NT_ON
      case ARTIS_MA_ACTION_INTERNALUPHIGHERNT: /* macroatom.cc:562 */
        ion = nt_random_upperion(o, c, element, ion, 0, p->rngstate);
        level = 0;
        stat_inc(o, ARTIS_STAT_MA_INTERNALUPHIGHERNT);
        break;
#endif
      default: /* MA_ACTION_INTERNALUPHIGHERNT needs NT_ON */
        ORACLE_FAIL(o, "do_macroatom: non-thermal action selected with NT_ON false");
        return;
    }
  }
  if (p->type == ARTIS_TYPE_RPKT) {
    if (p->trueemissiontype == ARTIS_EMTYPE_NOTSET) {
      p->trueemissiontype = p->emissiontype;
      p->trueem_pos[0] = p->em_pos[0];
      p->trueem_pos[1] = p->em_pos[1];
      p->trueem_pos[2] = p->em_pos[2];
      p->trueem_time = p->em_time;
    }
#if ARTIS_OPT_VPKT_ON
    trace_vpkts(o, p, ARTIS_TYPE_MA); /* macroatom.cc:588 */
#endif
  } else {
    p->trueemissiontype = ARTIS_EMTYPE_NOTSET;
  }
}

/* ------------------------------------------------------------------ rpkt events */
/* rpkt_event_continuum rpkt.cc:422 */
static void rpkt_event_continuum(Oracle *o, const CellCache *cc, artis_packet *p, ContOpacity *chi) {
  const artis_model *m = o->m;
  const double nu = p->nu_cmf;
  const double dopplerfactor = doppler_nucmf_on_nurf(p->pos, p->dir, p->prop_time);
  const double chi_cont = chi_total(chi) * dopplerfactor;
  const double chi_escatter = chi->chi_escatter * dopplerfactor;
  const double chi_ff = chi->chi_freefree_heat * dopplerfactor;
  const double chi_bf = chi->chi_boundfree * dopplerfactor;
  const double chi_rnd = rng_uniform(p->rngstate) * chi_cont;
  if (chi_rnd < chi_escatter) {
    p->nscatterings++;
    stat_inc(o, ARTIS_STAT_ELECTRON_SCATTERINGS);
#if ARTIS_OPT_VPKT_ON
    trace_vpkts(o, p, ARTIS_TYPE_RPKT); /* rpkt.cc:441 */
#endif
    electron_scatter_rpkt(p);
    p->em_pos[0] = p->pos[0];
    p->em_pos[1] = p->pos[1];
    p->em_pos[2] = p->pos[2];
    p->em_time = (float)p->prop_time;
  } else if (chi_rnd < chi_escatter + chi_ff) {
    stat_inc(o, ARTIS_STAT_K_FROM_FF);
    p->type = ARTIS_TYPE_KPKT;
    p->absorptiontype = ARTIS_ABSTYPE_FREEFREE;
  } else if (chi_rnd < chi_escatter + chi_ff + chi_bf) {
    p->absorptiontype = ARTIS_ABSTYPE_BOUNDFREE;
    const double chi_bf_rand = rng_uniform(p->rngstate) * chi->chi_boundfree;
    int allcontindex = -1;
    calculate_chi_bf_gammacontr(o, cc, chi->nonemptymgi, chi->nu, NULL, 1, chi_bf_rand, &allcontindex, NULL);
    const double nu_edge = m->allcont_nu_edge[allcontindex];
    const int element = m->allcont_element[allcontindex];
    const int ion = m->allcont_ion[allcontindex];
    const int level = m->allcont_level[allcontindex];
    const int t = m->allcont_phixstargetindex[allcontindex];
    if (rng_uniform(p->rngstate) < nu_edge / nu) {
      stat_inc(o, ARTIS_STAT_MA_ACTIVATION_BF);
      MacroAtomState ma = {element, ion + 1, get_phixsupperlevel(o, ionlevelstart(o, element, ion) + level, t), -99};
      do_macroatom(o, p, &ma);
    } else {
      stat_inc(o, ARTIS_STAT_K_FROM_BF);
      p->type = ARTIS_TYPE_KPKT;
    }
  } else {
    ORACLE_FAIL(o, "rpkt_event_continuum: no process selected");
  }
}

/* update_estimators rpkt.cc:502 + radfield::update_estimators radfield.cc:745 */
static void update_estimators(Oracle *o, double e_cmf, double nu_cmf, double distance, int c, const ContOpacity *chi, int thickcell) {
  const double distance_e_cmf = distance * e_cmf;
  if (distance_e_cmf != 0) {
    o->est.J[c] += distance_e_cmf;
    o->est.nuJ[c] += distance_e_cmf * nu_cmf;
  }
  if (thickcell) return;
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
  if (distance_e_cmf != 0 && o->est.bfrate_raw) { /* radfield::update_bfestimators radfield.cc:215 */
    const artis_model *m = o->m;
    const double distance_e_cmf_over_nu = distance_e_cmf / nu_cmf;
    (void)m;
    const int bfestimend = upper_bound_d(o->bfestim_nu_edge, chi->bfestimend, nu_cmf);
    const int bfestimbegin_stored = chi->bfestimbegin < bfestimend ? chi->bfestimbegin : bfestimend;
    const int bfestimbegin = bfestimbegin_stored + lower_bound_d(o->bfestim_nu_edge + bfestimbegin_stored, bfestimend - bfestimbegin_stored,
                                                                 nu_cmf / o->last_phixs_nuovernuedge);
    for (int i = bfestimbegin; i < bfestimend; i++)
      o->est.bfrate_raw[((ptrdiff_t)c * o->nbfestim) + i] += chi->gamma_contr[i] * distance_e_cmf_over_nu;
  }
#endif
#if ARTIS_OPT_MULTIBIN_RADFIELD_MODEL_ON
  if (distance_e_cmf != 0 && o->est.radfieldbin_J) { /* radfield.cc:762-770 */
    const int binindex = radbin_select(nu_cmf);
    if (binindex >= 0) {
      const ptrdiff_t mgibinindex = ((ptrdiff_t)c * ARTIS_OPT_RADFIELDBINCOUNT) + binindex;
      o->est.radfieldbin_J[mgibinindex] += distance_e_cmf;
      o->est.radfieldbin_nuJ[mgibinindex] += distance_e_cmf * nu_cmf;
    }
  }
#endif
  o->est.ffheatingestimator[c] += distance_e_cmf * chi->chi_freefree_heat;
#if ARTIS_OPT_USE_LUT_PHOTOION || ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS
  const int nbfg = o->m->nbfcontinua_ground;
  for (int i = 0; i < nbfg; i++) {
    const double nu_edge = o->m->groundcont_nu_edge[i];
    if (nu_cmf <= nu_edge) return;
    const ptrdiff_t ionestimindex = ((ptrdiff_t)c * nbfg) + i;
#if ARTIS_OPT_USE_LUT_PHOTOION
    o->est.gammaestimator[ionestimindex] += chi->groundcont_gamma_contr[i] * (distance_e_cmf / nu_cmf);
#endif
#if ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS
    o->est.bfheatingestimator[ionestimindex] += chi->groundcont_gamma_contr[i] * distance_e_cmf * (1. - (nu_edge / nu_cmf));
#endif
  }
#endif
}

/* do_rpkt_step rpkt.cc:542 */
static int do_rpkt_step(Oracle *o, artis_packet *p, double t2, ContOpacity *chi) {
  o->est.stats[ARTIS_STAT_X_RPKT_STEPS]++;
  const int c = propcell_nonemptymgi(o, p->cellindex);
  MacroAtomState pktmastate = {-1, -1, -1, -99};
  const double tau_rnd = -log((double)rng_uniform_pos(p->rngstate));
  int next_cellindex = -1;
  const double boundarydist = boundary_distance(o, p->dir, p->pos, p->prop_time, p->cellindex, &next_cellindex);
  if (boundarydist == 0) {
    change_cell_or_escape(o, p, next_cellindex);
    const int new_c = propcell_nonemptymgi(o, p->cellindex);
    return (p->type == ARTIS_TYPE_RPKT && (new_c < 0 || new_c == c));
  }
  const double tdist = (t2 - p->prop_time) * CLIGHT_PROP;
  if (!(tdist >= 0)) ORACLE_FAIL(o, "tdist < 0");
  const double abort_dist = dmin(tdist, boundarydist);
  double edist = -1;
  int event_is_boundbound = 1;
  const int thickcell = (c >= 0) && (o->cs->thick[c] == ARTIS_CELL_THICK);
  const CellCache *cc = NULL;
  if (c < 0) {
    edist = DBL_MAXV;
    p->next_trans = -1;
  } else if (thickcell) {
    const double chi_grey = o->cs->kappagrey[c] * o->cs->rho[c] * doppler_nucmf_on_nurf(p->pos, p->dir, p->prop_time);
    edist = tau_rnd / chi_grey;
    p->next_trans = -1;
  } else {
    cellcache_populate(o, c);
    cc = &o->cache[c];
    calculate_chi_rpkt_cont(o, cc, p->nu_cmf, chi, c);
    const double nu_cmf_abort = get_nu_cmf_abort(p->pos, p->dir, p->prop_time, p->nu_rf, abort_dist);
    const double doppler = doppler_nucmf_on_nurf(p->pos, p->dir, p->prop_time);
    const double dnu_on_dl = (nu_cmf_abort - p->nu_cmf) / abort_dist; /* rpkt.cc:591 */
#if ARTIS_OPT_RPKT_USE_EXPANSION_OPACITIES
    edist = get_possible_event_expansion_opacity(o, cc, c, p, chi, &pktmastate, tau_rnd, nu_cmf_abort, dnu_on_dl, doppler,
                                                 &event_is_boundbound); /* rpkt.cc:594 */
#else
    int nt = p->next_trans;
    edist = get_possible_event(o, cc, p, chi, &pktmastate, tau_rnd, abort_dist, nu_cmf_abort, dnu_on_dl, doppler, &nt, &event_is_boundbound);
    p->next_trans = nt;
#endif
  }
  if (!(edist >= 0)) ORACLE_FAIL(o, "edist < 0");

  if ((edist < boundarydist) && (edist <= tdist)) {
    move_pkt_withtime(p, edist / 2.);
    update_estimators(o, p->e_cmf, p->nu_cmf, edist, c, chi, thickcell);
    move_pkt_withtime(p, edist / 2.);
    stat_inc(o, ARTIS_STAT_INTERACTIONS);
    if (thickcell) {
      p->nscatterings++;
      stat_inc(o, ARTIS_STAT_ELECTRON_SCATTERINGS);
      emit_rpkt(p);
    } else if (!event_is_boundbound) {
      rpkt_event_continuum(o, cc, p, chi);
    } else {
#if !ARTIS_OPT_RPKT_BB_THERMALISATION
      stat_inc(o, ARTIS_STAT_MA_ACTIVATION_BB);
      p->absorptiontype = pktmastate.activatingline;
      p->absorptionfreq = p->nu_rf;
      do_macroatom(o, p, &pktmastate);
#else
      /* probability based thermalisation (redistribution of the packet frequency) or scattering, rpkt.cc:624-648 */
      if (ARTIS_OPT_RPKT_BB_THERMALISATION_PROBABILITY >= 1. || rng_uniform(p->rngstate) < ARTIS_OPT_RPKT_BB_THERMALISATION_PROBABILITY) {
        p->absorptiontype = pktmastate.activatingline;
        p->absorptionfreq = p->nu_rf;
        p->nu_cmf = sample_planck_times_expansion_opacity(o, c, p->rngstate);
        p->next_trans = -1;
        p->emissiontype = ARTIS_EMTYPE_NOTSET;
        p->trueemissiontype = ARTIS_EMTYPE_NOTSET;
        p->trueem_pos[0] = p->trueem_pos[1] = p->trueem_pos[2] = NAN;
        p->trueem_time = -1.;
        p->nscatterings = 0;
      } else {
        p->nscatterings++;
        stat_inc(o, ARTIS_STAT_ELECTRON_SCATTERINGS);
      }
      emit_rpkt(p);
#endif
    }
    return (p->type == ARTIS_TYPE_RPKT);
  }
  if ((boundarydist <= tdist) && (boundarydist <= edist)) {
    move_pkt_withtime(p, boundarydist / 2.);
    if (c >= 0) update_estimators(o, p->e_cmf, p->nu_cmf, boundarydist, c, chi, thickcell);
    move_pkt_withtime(p, boundarydist / 2.);
    if (next_cellindex != p->cellindex) {
      change_cell_or_escape(o, p, next_cellindex);
      if (next_cellindex < 0) return 0;
      const int new_c = propcell_nonemptymgi(o, p->cellindex);
      return ((new_c < 0) || (new_c == c));
    }
    return 1;
  }
  if ((tdist < boundarydist) && (tdist <= edist)) {
    move_pkt_withtime(p, tdist / 2.);
    if (c >= 0) update_estimators(o, p->e_cmf, p->nu_cmf, tdist, c, chi, thickcell);
    move_pkt_withtime(p, tdist / 2.);
    p->prop_time = t2;
    return 0;
  }
  ORACLE_FAIL(o, "do_rpkt_step: no branch taken");
  return 0;
}

/* ------------------------------------------------------------------ kpkt.cc */
/* sample_planck_montecarlo kpkt.cc:266 */
static double sample_planck_montecarlo(double T, uint32_t rng[4]) {
  const double nu_peak = 5.879e10 * T;
  const double B_peak = planck(nu_peak, T);
  while (1) {
    const double nu = ARTIS_OPT_NU_MIN_R + (rng_uniform(rng) * (ARTIS_OPT_NU_MAX_R - ARTIS_OPT_NU_MIN_R));
    if (rng_uniform(rng) * B_peak <= planck(nu, T)) return nu;
  }
}
/* do_kpkt_blackbody kpkt.cc:399 */
static void do_kpkt_blackbody(Oracle *o, artis_packet *p) {
  o->est.stats[ARTIS_STAT_X_KPKT_STEPS]++;
  const int c = propcell_nonemptymgi(o, p->cellindex);
#if ARTIS_OPT_RPKT_BB_THERMALISATION
  if (o->cs->thick[c] != ARTIS_CELL_THICK) { /* kpkt.cc:402 */
    /* (the reference's tables exist for every cell after update_grid(); this oracle makes a cell's tables with its cache) */
    if (!o->cs->expansionopacity_planck_cumulative) cellcache_populate(o, c);
    p->nu_cmf = sample_planck_times_expansion_opacity(o, c, p->rngstate);
  } else
#endif
  p->nu_cmf = sample_planck_montecarlo(o->cs->Te[c], p->rngstate);
  emit_rpkt(p);
  p->next_trans = -1;
  stat_inc(o, ARTIS_STAT_K_TO_R_BB);
  stat_inc(o, ARTIS_STAT_INTERACTIONS);
  p->emissiontype = ARTIS_EMTYPE_FREEFREE;
  p->trueemissiontype = p->emissiontype;
  p->trueem_pos[0] = p->em_pos[0];
  p->trueem_pos[1] = p->em_pos[1];
  p->trueem_pos[2] = p->em_pos[2];
  p->trueem_time = p->em_time;
  p->nscatterings = 0;
}
/* get_ionfromuniqueionindex atomic.h:413 */
static void ion_from_unique(const Oracle *o, int ui, int *element, int *ion) {
  *element = o->m->ion_element[ui];
  *ion = ui - o->m->elem_uniqueionindexstart[*element];
}
/* do_kpkt kpkt.cc:425 */
static void do_kpkt(Oracle *o, artis_packet *p, double t2) {
  o->est.stats[ARTIS_STAT_X_KPKT_STEPS]++;
  const artis_model *m = o->m;
  const double deltat = ARTIS_KPKTDIFFUSION_TIMESTEP_FRACTION * o->ts.width; /* float * double */
  const double t_current = dmin(p->prop_time + deltat, t2);
  const double sf = t_current / p->prop_time;
  p->pos[0] = p->pos[0] * sf; /* vec_scale vectors.h:62 */
  p->pos[1] = p->pos[1] * sf;
  p->pos[2] = p->pos[2] * sf;
  p->e_cmf *= p->prop_time / t_current;
  p->prop_time = t_current;
  if (t_current >= t2) return;
  stat_inc(o, ARTIS_STAT_INTERACTIONS);
  const int c = propcell_nonemptymgi(o, p->cellindex);
  cellcache_populate(o, c);
  const CellCache *cc = &o->cache[c];
  const double *cell_ion_contribs = cell_ion_cooling_contribs(o, c);
  const double rndcool_ion = rng_uniform(p->rngstate) * cell_ion_contribs[m->nions - 1];
  const int ui = index_upperbound(cell_ion_contribs, m->nions, rndcool_ion);
  if (!(ui < m->nions)) {
    ORACLE_FAIL(o, "do_kpkt: uniqueionindex out of range");
    return;
  }
  int element, ion;
  ion_from_unique(o, ui, &element, &ion);
  const int ionstart = m->ion_coolingoffset[ui];
  const int nterms = m->ion_ncoolingterms[ui];
  const double *ion_contribs = cooling_ion_contribs(o, c, element, ion);
  const double C_ion_procsum = ion_contribs[nterms - 1];
  const double rndcool_ion_process = rng_uniform(p->rngstate) * C_ion_procsum;
  const int ionoffset = index_upperbound(ion_contribs, nterms, rndcool_ion_process);
  if (!(ionoffset < nterms)) {
    ORACLE_FAIL(o, "do_kpkt: ionoffset out of range");
    return;
  }
  const int i = ionstart + ionoffset;
  const int rndcoolingtype = m->coolinglist_type[i];
  const float T_e = o->cs->Te[c];
  if (rndcoolingtype == ARTIS_COOLING_FREEFREE) {
    p->nu_cmf = -KB * T_e / H_PLANCK * log((double)rng_uniform_pos(p->rngstate));
    emit_rpkt(p);
    p->next_trans = -1;
    stat_inc(o, ARTIS_STAT_K_TO_R_FF);
    p->emissiontype = ARTIS_EMTYPE_FREEFREE;
    p->trueemissiontype = p->emissiontype;
    p->trueem_pos[0] = p->em_pos[0]; p->trueem_pos[1] = p->em_pos[1]; p->trueem_pos[2] = p->em_pos[2];
    p->trueem_time = p->em_time;
    p->nscatterings = 0;
#if ARTIS_OPT_VPKT_ON
    trace_vpkts(o, p, ARTIS_TYPE_KPKT); /* kpkt.cc:515 */
#endif
  } else if (rndcoolingtype == ARTIS_COOLING_FREEBOUND) {
    const int lowerion = ion;
    const int lowerlevel = m->coolinglist_level[i];
    const int t = m->coolinglist_phixstargetindex[i];
    p->nu_cmf = select_continuum_nu(o, element, lowerion, lowerlevel, t, T_e, p->rngstate);
    emit_rpkt(p);
    p->next_trans = -1;
    stat_inc(o, ARTIS_STAT_K_TO_R_FB);
    p->emissiontype = get_emtype_continuum(o, ionlevelstart(o, element, lowerion) + lowerlevel, t);
    p->trueemissiontype = p->emissiontype;
    p->trueem_pos[0] = p->em_pos[0]; p->trueem_pos[1] = p->em_pos[1]; p->trueem_pos[2] = p->em_pos[2];
    p->trueem_time = p->em_time;
    p->nscatterings = 0;
#if ARTIS_OPT_VPKT_ON
    trace_vpkts(o, p, ARTIS_TYPE_KPKT); /* kpkt.cc:541 */
#endif
  } else if (rndcoolingtype == ARTIS_COOLING_COLLEXC) {
    const float clumpednne = cell_clumpednne(o, c);
    const double contrib_low = (i > ionstart) ? cc->cooling_contrib[i - 1] : 0.;
    double contrib = contrib_low;
    const int start = ionlevelstart(o, element, ion);
    const int ul = start + m->coolinglist_level[i];
    const double epsilon_current = epsilon(o, ul);
    const double nnlevel = cc->levelpops[ul];
    const double statweight = stat_weight(o, ul);
    int upper = -1;
    const int startup = m->level_alltrans_startdown[ul] + m->level_ndowntrans[ul];
    const int nuptrans = m->level_nuptrans[ul];
    for (int ati = startup; ati < (startup + nuptrans); ati++) {
      const int tmpupper = m->alltrans_targetlevelindex[ati];
      const int uul = start + tmpupper;
      const double epsilon_trans = epsilon(o, uul) - epsilon_current;
      const double upper_statweight = stat_weight(o, uul);
      const double C = nnlevel * col_excitation_ratecoeff(o, T_e, clumpednne, epsilon_trans, upper_statweight, statweight, ati) *
                       epsilon_trans;
      contrib += C;
      if (contrib > rndcool_ion_process) {
        upper = tmpupper;
        break;
      }
    }
    if (!(contrib > rndcool_ion_process)) {
      ORACLE_FAIL(o, "do_kpkt: collexc found no transition");
      return;
    }
    stat_inc(o, ARTIS_STAT_MA_ACTIVATION_COLLEXC);
    stat_inc(o, ARTIS_STAT_K_TO_MA_COLLEXC);
    p->trueemissiontype = ARTIS_EMTYPE_NOTSET;
    p->trueem_pos[0] = NAN; p->trueem_pos[1] = NAN; p->trueem_pos[2] = NAN;
    MacroAtomState ma = {element, ion, upper, -99};
    do_macroatom(o, p, &ma);
  } else if (rndcoolingtype == ARTIS_COOLING_COLLION) {
    const int upperion = ion + 1;
    const int upper = get_phixsupperlevel(o, ionlevelstart(o, element, ion) + m->coolinglist_level[i], m->coolinglist_phixstargetindex[i]);
    stat_inc(o, ARTIS_STAT_MA_ACTIVATION_COLLION);
    stat_inc(o, ARTIS_STAT_K_TO_MA_COLLION);
    p->trueemissiontype = ARTIS_EMTYPE_NOTSET;
    p->trueem_pos[0] = NAN; p->trueem_pos[1] = NAN; p->trueem_pos[2] = NAN;
    MacroAtomState ma = {element, upperion, upper, -99};
    do_macroatom(o, p, &ma);
  } else {
    ORACLE_FAIL(o, "do_kpkt: bad cooling type");
  }
}

/* ------------------------------------------------------------------ gammapkt.cc / gammapkt.h (classic preset:
 * GAMMA_THERMALISATION_SCHEME FREQUENCYDEPENDENT, no grey opacity, USE_XCOM_GAMMAPHOTOION off,
 * PARTICLE_THERMALISATION_SCHEME INSTANTFULLDEPOSITION; artisoptions_classic.h:144-150) */
/* sigma_compton_partial gammapkt.h:28 */
static double sigma_compton_partial(double x, double f_max) {
  const double term1 = ((x * x) - (2 * x) - 2) * log(f_max) / x / x;
  const double term2 = (((f_max * f_max) - 1) / (f_max * f_max)) / 2;
  const double term3 = ((f_max - 1) / x) * ((1 / x) + (2 / f_max) + (1 / (x * f_max)));
  return (3 * SIGMA_T * (term1 + term2 + term3) / (8 * x));
}
/* choose_f gammapkt.h:38 */
static double choose_f(double xx, double zrand) {
  double f_max = 1 + (2 * xx);
  double f_min = 1;
  const double norm = zrand * sigma_compton_partial(xx, f_max);
  int count = 0;
  double err = 1e20;
  double ftry = (f_max + f_min) / 2;
  while ((err > 1.e-4) && (count < 1000)) {
    ftry = (f_max + f_min) / 2;
    const double sigma_try = sigma_compton_partial(xx, ftry);
    if (sigma_try > norm) {
      f_max = ftry;
      err = (sigma_try - norm) / norm;
    } else {
      f_min = ftry;
      err = (norm - sigma_try) / norm;
    }
    count++;
  }
  return ftry;
}
/* meanf_sigma gammapkt.h:68 */
static double meanf_sigma(double x) {
  if (x < THOMSON_LIMIT) {
    static const double taylor_coeffs[8] = {1., -21. / 5., 147. / 10., -1616. / 35., 940. / 7., -2584. / 7., 14588. / 15., -409088. / 165.};
    double series = taylor_coeffs[7];
    for (int i = 6; i >= 0; i--) series = taylor_coeffs[i] + (x * series);
    return SIGMA_T * x * series;
  }
  const double f = 1 + (2 * x);
  const double term0 = 2 / x;
  const double term1 = (1 - (2 / x) - (3 / (x * x))) * log(f);
  const double term2 = ((4 / x) + (3 / (x * x)) - 1) * 2 * x / f;
  const double term3 = (1 - (2 / x) - (1 / (x * x))) * 2 * x * (1 + x) / f / f;
  const double term4 = -2. * x * ((4 * x * x) + (6 * x) + 3) / 3 / f / f / f;
  return 3 * SIGMA_T * (term0 + term1 + term2 + term3 + term4) / (8 * x);
}
/* get_chi_compton_cmf gammapkt.cc:265 */
static double get_chi_compton_cmf(const Oracle *o, int c, double nu_cmf) {
  if (ARTIS_OPT_GAMMA_USE_KAPPA_GREY) return 0.; /* gammapkt.cc:266 */
  const double xx = H_PLANCK * nu_cmf / ME / CLIGHT / CLIGHT;
  const double sigma_cmf = (xx < THOMSON_LIMIT) ? SIGMA_T : sigma_compton_partial(xx, 1 + (2 * xx));
  return sigma_cmf * o->cs->nnetot[c];
}
/* get_chi_photo_electric_cmf gammapkt.cc:416 (Veigele fit, no XCOM tables) */
static double get_chi_photo_electric_cmf(const Oracle *o, int c, double ffegrp, double nu_cmf) {
  const double rho = o->cs->rho[c];
  if (ARTIS_OPT_GAMMA_USE_KAPPA_GREY) return ARTIS_OPT_GAMMA_KAPPA_GREY * rho; /* gammapkt.cc:420 */
#if ARTIS_OPT_USE_XCOM_GAMMAPHOTOION
  { /* gammapkt.cc:443-495 */
    (void)ffegrp;
    const artis_model *m = o->m;
    const double hnu_over_1MeV = nu_cmf / NU_1MEV;
    const double log10_hnu_over_1MeV = log10(hnu_over_1MeV);
    double chi_cmf = 0.;
    for (int i = 0; i < m->nelements; i++) {
      const int s0 = m->xcom_elem_start[i];
      const int numb_energies = m->xcom_elem_start[i + 1] - s0; /* none beyond the table (Z > xcom_max_atomic_number) or without data */
      if (numb_energies == 0) continue;
      const double n_i = get_elem_numberdens(o, c, i);
      if (n_i == 0) continue;
      const double *energy = m->xcom_energy + s0;
      const double *sigma_xcom = m->xcom_sigma + s0;
      int idx_above = -1;
      for (int j = 0; j < numb_energies; j++) {
        if (energy[j] > hnu_over_1MeV) {
          idx_above = j;
          break;
        }
      }
      if (idx_above == 0) {
        chi_cmf += sigma_xcom[0] * n_i;
        continue;
      }
      if (idx_above == -1) {
        chi_cmf += sigma_xcom[numb_energies - 1] * n_i;
        continue;
      }
      const int idx_below = idx_above - 1;
      const double log10_E = log10_hnu_over_1MeV;
      const double log10_E_above = log10(energy[idx_above]);
      const double log10_E_below = log10(energy[idx_below]);
      const double log10_sigma_below = log10(sigma_xcom[idx_below]);
      const double log10_sigma_above = log10(sigma_xcom[idx_above]);
      const double log10_sigma_interp =
          log10_sigma_below + ((log10_sigma_above - log10_sigma_below) / (log10_E_above - log10_E_below) * (log10_E - log10_E_below));
      const double sigma_interp = pow(10., log10_sigma_interp);
      if (!(sigma_interp >= 0.)) ORACLE_FAIL((Oracle *)o, "XCOM: negative cross section");
      chi_cmf += sigma_interp * n_i;
    }
    return chi_cmf;
  }
#endif
  const double hnu_over_100kev = nu_cmf / NU_100KEV;
  const double sigma_cmf_si = 1.16e-24 * pow(hnu_over_100kev, -3.13);
  const double sigma_cmf_fe = 25.7e-24 * pow(hnu_over_100kev, -3.0);
  const double chi_cmf_si = sigma_cmf_si * (rho / MH / 28);
  const double chi_cmf_fe = sigma_cmf_fe * (rho / MH / 56);
  return (chi_cmf_fe * ffegrp) + (chi_cmf_si * (1. - ffegrp));
}
/* get_sigma_pair_prod_factor gammapkt.cc:501 */
static double get_sigma_pair_prod_factor(double nu_cmf) {
  const double hnu_over_1MeV = nu_cmf / NU_1MEV;
  if (nu_cmf > NU_1P5MEV) return 0.0481 + (0.301 * (hnu_over_1MeV - 1.5));
  return 0.10063 * (hnu_over_1MeV - 1.022);
}
/* get_chi_pair_prod_cmf gammapkt.cc:516 */
static double get_chi_pair_prod_cmf(const Oracle *o, int c, double ffegrp, double nu_cmf) {
  if (ARTIS_OPT_GAMMA_USE_KAPPA_GREY) return 0.; /* gammapkt.cc:517 */
  const double rho = o->cs->rho[c];
  if (nu_cmf <= NU_1P022MEV) return 0.;
  const double sigma_factor = get_sigma_pair_prod_factor(nu_cmf);
  const double sigma_cmf_si = sigma_factor * 196.e-27;
  const double sigma_cmf_fe = sigma_factor * 784.e-27;
  const double chi_cmf_si = sigma_cmf_si * (rho / MH / 28);
  const double chi_cmf_fe = sigma_cmf_fe * (rho / MH / 56);
  const double chi_cmf = (chi_cmf_fe * ffegrp) + (chi_cmf_si * (1. - ffegrp));
  return dmax(chi_cmf, 0.);
}
static inline double cell_ffegrp(const Oracle *o, int c) { return o->cs->ffegrp ? o->cs->ffegrp[c] : 0.; } /* grid::get_ffegrp */
/* get_chi_cmf_loss_weighted gammapkt.cc:548 */
static double get_chi_cmf_loss_weighted(const Oracle *o, int c, double nu_cmf) {
  const double ffegrp = cell_ffegrp(o, c);
  const double chi_photo_electric_cmf = get_chi_photo_electric_cmf(o, c, ffegrp, nu_cmf);
  if (ARTIS_OPT_GAMMA_USE_KAPPA_GREY) return chi_photo_electric_cmf; /* gammapkt.cc:553 */
  const double xx = H_PLANCK * nu_cmf / ME / CLIGHT / CLIGHT;
  const double chi_pair_prod_cmf = get_chi_pair_prod_cmf(o, c, ffegrp, nu_cmf);
  return ((meanf_sigma(xx) * o->cs->nnetot[c]) + chi_photo_electric_cmf + (chi_pair_prod_cmf * (1. - (NU_1P022MEV / nu_cmf))));
}
/* update_gamma_dep gammapkt.cc:568 */
static void update_gamma_dep(Oracle *o, const artis_packet *p, int c, double dist) {
  if (!(dist > 0)) return;
  if (ARTIS_GAMMAPRODUCTS) return; /* the particles the gamma rays produce deposit instead, gammapkt.cc:572 */
  if (c < 0) return;
  const double doppler_sq = pow2(doppler_nucmf_on_nurf(p->pos, p->dir, p->prop_time));
  const double heating_cont = get_chi_cmf_loss_weighted(o, c, p->nu_cmf) * p->e_rf * dist * doppler_sq;
  if (o->est.dep_estimator_gamma) o->est.dep_estimator_gamma[c] += heating_cont;
}
/* thomson_angle gammapkt.cc:284 */
static double thomson_angle(uint32_t rng[4]) {
  const double B_coeff = (8. * rng_uniform(rng)) - 4.;
  const double t_coeff = cbrt((sqrt(pow2(B_coeff) + 4) - B_coeff) / 2);
  return (1 / t_coeff) - t_coeff;
}
/* scatter_dir gammapkt.cc:297 */
static void scatter_dir(const double dir_in[3], double cos_theta, uint32_t rng[4], double dir_out[3]) {
  const double phi = rng_uniform(rng) * 2 * PI;
  const double sin_theta_sq = 1. - pow2(cos_theta);
  const double sin_theta = sqrt(sin_theta_sq);
  const double zprime = cos_theta;
  const double xprime = sin_theta * cos(phi);
  const double yprime = sin_theta * sin(phi);
  if (fabs(dir_in[2]) > 0.999999999) {
    dir_out[0] = xprime;
    dir_out[1] = yprime;
    dir_out[2] = (dir_in[2] > 0) ? zprime : -zprime;
    return;
  }
  const double norm1 = 1. / sqrt(pow2(dir_in[0]) + pow2(dir_in[1]));
  const double norm2 = 1. / vec_len3(dir_in);
  const double r11 = dir_in[1] * norm1;
  const double r12 = -dir_in[0] * norm1;
  const double r13 = 0.;
  const double r21 = dir_in[0] * dir_in[2] * norm1 * norm2;
  const double r22 = dir_in[1] * dir_in[2] * norm1 * norm2;
  const double r23 = -norm2 / norm1;
  const double r31 = dir_in[0] * norm2;
  const double r32 = dir_in[1] * norm2;
  const double r33 = dir_in[2] * norm2;
  dir_out[0] = (r11 * xprime) + (r21 * yprime) + (r31 * zprime);
  dir_out[1] = (r12 * xprime) + (r22 * yprime) + (r32 * zprime);
  dir_out[2] = (r13 * xprime) + (r23 * yprime) + (r33 * zprime);
}
/* compton_scatter gammapkt.cc:346 */
static void compton_scatter(Oracle *o, artis_packet *p) {
  const double xx = H_PLANCK * p->nu_cmf / ME / CLIGHT / CLIGHT;
  double f = 1.;
  int stay_gamma = 1;
  if (xx >= THOMSON_LIMIT) {
    f = choose_f(xx, rng_uniform(p->rngstate));
    const double prob_gamma = 1. / f;
    stay_gamma = (rng_uniform(p->rngstate) < prob_gamma);
  }
  if (stay_gamma) {
    p->nu_cmf = p->nu_cmf / f;
    double vel_vec[3], cmf_dir[3], new_dir[3];
    get_velocity(p->pos, p->prop_time, vel_vec);
    angle_ab(p->dir, vel_vec, cmf_dir);
    const double cos_theta = (xx < THOMSON_LIMIT) ? thomson_angle(p->rngstate) : 1. - ((f - 1) / xx);
    scatter_dir(cmf_dir, cos_theta, p->rngstate, new_dir);
    const double negvel[3] = {vel_vec[0] * -1., vel_vec[1] * -1., vel_vec[2] * -1.}; /* vec_scale(vel_vec, -1.) */
    angle_ab(new_dir, negvel, p->dir);
    set_pkt_restframe_from_cmf(p);
  } else {
#if ARTIS_GAMMAPRODUCTS
    p->nu_cmf = p->nu_cmf * (1 - (1 / f)); /* the gamma's energy loss is the electron's energy, gammapkt.cc:404 */
    p->type = ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS;
#else
    p->type = ARTIS_TYPE_NTLEPTON_DEPOSITED;
#endif
    p->absorptiontype = ARTIS_ABSTYPE_GAMMA_COMPTON;
    stat_inc(o, ARTIS_STAT_NT_FROM_GAMMA);
  }
}
/* emit_gamma_isotropic gammapkt.cc:603 */
static void emit_gamma_isotropic(artis_packet *p) {
  double dir_cmf[3], vel_vec[3];
  get_rand_isotropic_unitvec(p->rngstate, dir_cmf);
  get_velocity(p->pos, -p->prop_time, vel_vec);
  angle_ab(dir_cmf, vel_vec, p->dir);
  set_pkt_restframe_from_cmf(p);
  p->type = ARTIS_TYPE_GAMMA;
}
/* pair_production gammapkt.cc:618 */
static void pair_production(Oracle *o, artis_packet *p) {
  const double pair_rest_mass_energy = 1.022 * MEV;
  const double gamma_energy = H_PLANCK * p->nu_cmf;
  const double prob_gamma = pair_rest_mass_energy / gamma_energy;
  if (rng_uniform(p->rngstate) > prob_gamma) {
#if ARTIS_GAMMAPRODUCTS
    const double particle_kinetic_energy = (gamma_energy - pair_rest_mass_energy) / 2; /* gammapkt.cc:630 */
    p->nu_cmf = particle_kinetic_energy / H_PLANCK;
    p->type = (rng_uniform(p->rngstate) > 0.5) ? ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS : ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAPLUS;
#else
    p->type = ARTIS_TYPE_NTLEPTON_DEPOSITED;
#endif
    p->absorptiontype = ARTIS_ABSTYPE_GAMMA_PAIRPRODUCTION;
    stat_inc(o, ARTIS_STAT_NT_FROM_GAMMA);
  } else {
    p->nu_cmf = 0.511 * MEV / H_PLANCK;
    emit_gamma_isotropic(p);
  }
}
/* transport_gamma gammapkt.cc:655 */
static void transport_gamma(Oracle *o, artis_packet *p, double t2) {
  const double tau_next = -log((double)rng_uniform_pos(p->rngstate));
  int next_cellindex = -1;
  const double boundarydist = boundary_distance(o, p->dir, p->pos, p->prop_time, p->cellindex, &next_cellindex);
  const int c = propcell_nonemptymgi(o, p->cellindex);
  const double doppler = doppler_nucmf_on_nurf(p->pos, p->dir, p->prop_time);
  const double ffegrp = (c >= 0) ? cell_ffegrp(o, c) : 0.;
  const double chi_compton = (c >= 0) ? get_chi_compton_cmf(o, c, p->nu_cmf) * doppler : 0.;
  const double chi_photo_electric = (c >= 0) ? get_chi_photo_electric_cmf(o, c, ffegrp, p->nu_cmf) * doppler : 0.;
  const double chi_pair_prod = (c >= 0) ? get_chi_pair_prod_cmf(o, c, ffegrp, p->nu_cmf) * doppler : 0.;
  const double chi_tot = chi_compton + chi_photo_electric + chi_pair_prod;
  const double edist = chi_tot > 0. ? tau_next / chi_tot : DBL_MAXV;
  if (!(edist >= 0)) ORACLE_FAIL(o, "transport_gamma: negative edist");
  const double tdist = (t2 - p->prop_time) * CLIGHT_PROP;
  if (!(tdist >= 0)) ORACLE_FAIL(o, "transport_gamma: negative tdist");
  if ((boundarydist <= tdist) && (boundarydist <= edist)) {
    move_pkt_withtime(p, boundarydist / 2.);
    if (chi_tot > 0) update_gamma_dep(o, p, c, boundarydist);
    move_pkt_withtime(p, boundarydist / 2.);
    if (next_cellindex != p->cellindex) change_cell_or_escape(o, p, next_cellindex);
  } else if ((tdist < boundarydist) && (tdist <= edist)) {
    move_pkt_withtime(p, tdist / 2.);
    if (chi_tot > 0) update_gamma_dep(o, p, c, tdist);
    move_pkt_withtime(p, tdist / 2.);
    p->prop_time = t2;
  } else if ((edist < boundarydist) && (edist <= tdist)) {
    move_pkt_withtime(p, edist / 2.);
    if (chi_tot > 0) update_gamma_dep(o, p, c, edist);
    move_pkt_withtime(p, edist / 2.);
    const double chi_rnd = rng_uniform(p->rngstate) * chi_tot;
    if (chi_compton > chi_rnd) {
      compton_scatter(o, p);
    } else if ((chi_compton + chi_photo_electric) > chi_rnd) {
      p->type = ARTIS_GAMMAPRODUCTS ? ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS : ARTIS_TYPE_NTLEPTON_DEPOSITED; /* gammapkt.cc:734 */
      p->absorptiontype = ARTIS_ABSTYPE_GAMMA_PHOTOELECTRIC;
      stat_inc(o, ARTIS_STAT_NT_FROM_GAMMA);
    } else {
      pair_production(o, p);
    }
  } else {
    ORACLE_FAIL(o, "transport_gamma: no branch");
  }
}
#if ARTIS_OPT_GAMMA_THERMALISATION_SCHEME != ARTIS_GAMMA_FREQUENCYDEPENDENT
/* absorb_or_escape_gamma gammapkt.cc:754 */
static void absorb_or_escape_gamma(Oracle *o, artis_packet *p, double f_gamma) {
  if (!(f_gamma >= 0.) || !(f_gamma <= 1.)) ORACLE_FAIL(o, "gamma thermalisation: f_gamma outside [0, 1]");
  if (rng_uniform(p->rngstate) < f_gamma) {
    p->type = ARTIS_TYPE_NTLEPTON_DEPOSITED;
    p->absorptiontype = ARTIS_ABSTYPE_GAMMA_PHOTOELECTRIC;
  } else {
    change_cell_or_escape(o, p, -99);
  }
}
/* the ray integration shared by wollaeger_thermalisation() and guttman_thermalisation() (gammapkt.cc:805-828, :846-858):
 * a copy of the packet is moved cell by cell out of the grid, change_cell_or_escape(.., tally_stats = false) */
static double gamma_ray_tau(Oracle *o, artis_packet pkt_copy, double mean_gamma_opac) {
  double tau = 0.;
  while (pkt_copy.type != ARTIS_TYPE_ESCAPE) {
    int next_cellindex = -1;
    const double boundarydist = boundary_distance(o, pkt_copy.dir, pkt_copy.pos, pkt_copy.prop_time, pkt_copy.cellindex, &next_cellindex);
    const int c = propcell_nonemptymgi(o, pkt_copy.cellindex);
    if (c >= 0) {
      const double rho = o->m->rho_tmin[c] * pow3(o->m->tmin / pkt_copy.prop_time);
      tau += mean_gamma_opac * rho * boundarydist;
    }
    move_pkt_withtime(&pkt_copy, boundarydist);
    if (next_cellindex >= 0) {
      if (next_cellindex != pkt_copy.cellindex) snap_pos_to_cell(o, pkt_copy.pos, pkt_copy.prop_time, next_cellindex);
      pkt_copy.cellindex = next_cellindex;
    } else {
      pkt_copy.type = ARTIS_TYPE_ESCAPE;
    }
    if (o->error) break;
  }
  return tau;
}
#endif
/* do_gamma gammapkt.cc:911 */
static void do_gamma(Oracle *o, artis_packet *p, double t2) {
  stat_inc(o, ARTIS_STAT_X_GAMMA_STEPS);
#if ARTIS_OPT_GAMMA_THERMALISATION_SCHEME == ARTIS_GAMMA_FREQUENCYDEPENDENT
  transport_gamma(o, p, t2);
#elif ARTIS_OPT_GAMMA_THERMALISATION_SCHEME == ARTIS_GAMMA_BARNES
  { /* barnes_thermalisation gammapkt.cc:779 */
    (void)t2;
    const double E_kin = o->m->ejecta_kinetic_energy;
    const double v_ej = sqrt(E_kin * 2 / o->m->mtot_input);
    const double t_ineff = 1.4 * DAY * sqrt(o->m->mtot_input / (5.e-3 * MSUN)) * ((0.2 * CLIGHT) / v_ej);
    const double tau = pow2(t_ineff / p->prop_time);
    absorb_or_escape_gamma(o, p, 1. - exp(-tau));
  }
#elif ARTIS_OPT_GAMMA_THERMALISATION_SCHEME == ARTIS_GAMMA_WOLLAEGER
  { /* wollaeger_thermalisation gammapkt.cc:797 */
    (void)t2;
    artis_packet pkt_copy = *p;
    vec_norm3(p->pos, pkt_copy.dir); /* integrate the optical depth radially outwards */
    const double tau = gamma_ray_tau(o, pkt_copy, 0.1);
    absorb_or_escape_gamma(o, p, 1. - exp(-tau));
  }
#else
  { /* guttman_thermalisation gammapkt.cc:831 */
    (void)t2;
    double deposition_probability_sum = 0.;
    for (int i = 0; i < 100; i++) {
      artis_packet pkt_copy = *p;
      get_rand_isotropic_unitvec(p->rngstate, pkt_copy.dir);
      deposition_probability_sum -= expm1(-gamma_ray_tau(o, pkt_copy, 0.03));
    }
    absorb_or_escape_gamma(o, p, deposition_probability_sum / 100);
  }
#endif
  if (p->type != ARTIS_TYPE_GAMMA && p->type != ARTIS_TYPE_ESCAPE) { /* gammapkt.cc:924-936 */
    if (!ARTIS_GAMMAPRODUCTS && o->est.scalars) o->est.scalars[ARTIS_SCALAR_GAMMA_DEP_DISCRETE] += p->e_cmf;
#if ARTIS_OPT_GAMMA_THERMALISATION_SCHEME != ARTIS_GAMMA_FREQUENCYDEPENDENT
    const int c = propcell_nonemptymgi(o, p->cellindex); /* no transport: the path-based estimator is fed here */
    if (c >= 0 && o->est.dep_estimator_gamma) o->est.dep_estimator_gamma[c] += p->e_cmf;
#endif
  }
}
/* nonthermal::do_ntlepton_deposit nonthermal.cc:2529 (NT_ON false in artisoptions_classic.h:95; NT_ON with
 * NT_SOLVE_SPENCERFANO in artisoptions_nltenebular.h:102-104) */
static void do_ntlepton_deposit(Oracle *o, artis_packet *p) {
  if (o->est.scalars) o->est.scalars[ARTIS_SCALAR_NT_ENERGY_DEPOSITED] += p->e_cmf;
#if ARTIS_OPT_NT_ON
  const int c = propcell_nonemptymgi(o, p->cellindex);
  if (o->cs->thick[c] != ARTIS_CELL_THICK) { /* macroatom should not be activated in thick cells */
    cellcache_populate(o, c); /* TYPE_NTLEPTON_DEPOSITED is a cell-cache type: get_packet_cellcachegroupid update_packets.cc:333 */
    double zrand = rng_uniform(p->rngstate);
    const double frac_ionisation = o->cs->nt_frac_ionisation[c];
    if (zrand < frac_ionisation) {
      int element = -1, lowerion = -1;
      if (select_nt_ionisation(o, c, p->rngstate, &element, &lowerion)) {
        const int upperion = nt_random_upperion(o, c, element, lowerion, 1, p->rngstate);
        stat_inc(o, ARTIS_STAT_MA_ACTIVATION_NTCOLLION);
        stat_inc(o, ARTIS_STAT_INTERACTIONS);
        p->trueemissiontype = ARTIS_EMTYPE_NOTSET;
        p->trueem_pos[0] = p->trueem_pos[1] = p->trueem_pos[2] = NAN;
        stat_inc(o, ARTIS_STAT_NT_TO_IONISATION);
        const MacroAtomState ma = {element, upperion, 0, -99};
        do_macroatom(o, p, &ma);
        return;
      }
      p->type = ARTIS_TYPE_KPKT;
      stat_inc(o, ARTIS_STAT_NT_TO_KPKT);
      return;
    }
    const double frac_excitation = ARTIS_OPT_NT_EXCITATION_ON ? o->cs->nt_frac_excitation[c] : 0.;
    if (zrand < (frac_ionisation + frac_excitation)) {
      zrand -= frac_ionisation;
      const ptrdiff_t base = (ptrdiff_t)c * o->cs->nt_excitations_stored;
      for (int i = 0; i < o->cs->nt_exc_count[c]; i++) {
        const double frac_deposition_exc = o->cs->nt_exc_frac_deposition[base + i];
        if (zrand < frac_deposition_exc) {
          const int lineindex = o->m->alltrans_lineindex[o->cs->nt_exc_alltransindex[base + i]];
          const int element = o->m->line_elementindex[lineindex];
          const int ion = o->m->line_ionindex[lineindex];
          const int upper = o->m->line_uniquelevelindex_upper[lineindex] - ionlevelstart(o, element, ion); /* get_levelfromuniquelevelindex */
          stat_inc(o, ARTIS_STAT_MA_ACTIVATION_NTCOLLEXC);
          stat_inc(o, ARTIS_STAT_INTERACTIONS);
          p->trueemissiontype = ARTIS_EMTYPE_NOTSET;
          p->trueem_pos[0] = p->trueem_pos[1] = p->trueem_pos[2] = NAN;
          stat_inc(o, ARTIS_STAT_NT_TO_EXCITATION);
          const MacroAtomState ma = {element, ion, upper, -99};
          do_macroatom(o, p, &ma);
          return;
        }
        zrand -= frac_deposition_exc;
      }
    }
  }
#endif
  p->type = ARTIS_TYPE_KPKT;
  stat_inc(o, ARTIS_STAT_NT_TO_KPKT);
}

/* nonthermal::do_ntalpha_fisprod_deposit nonthermal.cc:2520 */
static void do_ntalpha_fisprod_deposit(Oracle *o, artis_packet *p) {
  if (o->est.scalars) o->est.scalars[ARTIS_SCALAR_NT_ENERGY_DEPOSITED] += p->e_cmf;
  p->type = ARTIS_TYPE_KPKT;
  stat_inc(o, ARTIS_STAT_NT_TO_KPKT);
}
static inline void scalar_add(Oracle *o, int i, double v) {
  if (o->est.scalars) o->est.scalars[i] += v;
}
/* do_nonthermal_predeposit update_packets.cc:42 (INSTANTFULLDEPOSITION, TIMEDEPENDENT, TIMEDEPENDENT_WITH_ADIABATIC_LOSS) */
static void do_nonthermal_predeposit(Oracle *o, artis_packet *p, double ts_end) {
  double e_cmf_deposited = p->e_cmf;
  const int c = propcell_nonemptymgi(o, p->cellindex);
  const int priortype = p->type;
  const double ts = p->prop_time;
  const int deposit_type = (p->type == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA) ? ARTIS_TYPE_NTALPHA_FISPROD_DEPOSITED : ARTIS_TYPE_NTLEPTON_DEPOSITED;
#if ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_INSTANTFULLDEPOSITION
  (void)ts; (void)ts_end;
  p->type = deposit_type; /* absorption happens */
#elif ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_BARNES || ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_WOLLAEGER
  { /* update_packets.cc:53-88: deposit with probability f_p, otherwise discard the deposited energy and let the particle escape */
    (void)ts_end;
    double f_p;
    if (ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_BARNES) { /* :69 */
      const double E_kin = o->m->ejecta_kinetic_energy;
      const double v_ej = sqrt(E_kin * 2 / o->m->mtot_input);
      const double prefactor = (p->type == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA) ? 7.74 : 7.4;
      const double tau_ineff = prefactor * DAY * sqrt(o->m->mtot_input / (5.e-3 * MSUN)) * pow((0.2 * CLIGHT) / v_ej, 3. / 2.);
      f_p = log1p(2. * ts * ts / tau_ineff / tau_ineff) / (2. * ts * ts / tau_ineff / tau_ineff);
    } else { /* :78 Wollaeger et al. 2018 */
      const double A = (p->type == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA) ? 1.2 * 1.e-11 : 1.3 * 1.e-11;
      const double aux_term = 2 * A / (ts * o->cs->rho[c]);
      f_p = log1p(aux_term) / aux_term;
    }
    if (!(f_p >= 0.) || !(f_p <= 1.)) ORACLE_FAIL(o, "thermalisation efficiency outside [0, 1]");
    if (rng_uniform(p->rngstate) < f_p) {
      p->type = deposit_type;
    } else {
      e_cmf_deposited = 0.;
      change_cell_or_escape(o, p, -99);
    }
  }
#else
  { /* local time-dependent absorption, update_packets.cc:90-150 */
    const double rho = o->cs->rho[c];
    const double particle_en = H_PLANCK * p->nu_cmf;
    const double endot_collisional = (p->type == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA) ? 5.e11 * MEV * rho : 4.e10 * MEV * rho;
    const double endot_adiabatic =
        (ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_TIMEDEPENDENT_WITH_ADIABATIC_LOSS) ? particle_en / ts : 0.;
    const double endot = endot_collisional + endot_adiabatic;
    e_cmf_deposited = p->e_cmf * endot_collisional * dmin(ts_end - ts, particle_en / endot) / particle_en;
    const double rnd_en_absorb = rng_uniform(p->rngstate) * particle_en;
    const double t_absorb = ts + (rnd_en_absorb / endot);
    const double t_new = dmin(t_absorb, ts_end);
    const int absorbed = (t_absorb <= ts_end);
    if (absorbed) {
      p->type = deposit_type;
    } else {
      p->nu_cmf -= (endot * (ts_end - ts)) / H_PLANCK;
    }
    const double scale = t_new / ts;
    p->pos[0] = p->pos[0] * scale; p->pos[1] = p->pos[1] * scale; p->pos[2] = p->pos[2] * scale;
    p->prop_time = t_new;
    if (ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_TIMEDEPENDENT_WITH_ADIABATIC_LOSS && absorbed)
      p->e_cmf *= endot_collisional / endot;
  }
#endif
  if (p->originated_from_particlenotgamma) {
    if (priortype == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS) {
      if (o->est.dep_estimator_electron) o->est.dep_estimator_electron[c] += e_cmf_deposited;
      if (p->type == deposit_type) scalar_add(o, ARTIS_SCALAR_ELECTRON_DEP_DISCRETE, p->e_cmf);
    } else if (priortype == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAPLUS) {
      if (o->est.dep_estimator_positron) o->est.dep_estimator_positron[c] += e_cmf_deposited;
      if (p->type == deposit_type) scalar_add(o, ARTIS_SCALAR_POSITRON_DEP_DISCRETE, p->e_cmf);
    } else if (priortype == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA) {
      if (o->est.dep_estimator_alpha) o->est.dep_estimator_alpha[c] += e_cmf_deposited;
      if (p->type == deposit_type) scalar_add(o, ARTIS_SCALAR_ALPHA_DEP_DISCRETE, p->e_cmf);
    }
  } else if (ARTIS_GAMMAPRODUCTS) { /* update_packets.cc:174: products of gamma rays count as gamma deposition */
    if (o->est.dep_estimator_gamma) o->est.dep_estimator_gamma[c] += e_cmf_deposited;
    if (p->type == ARTIS_TYPE_NTLEPTON_DEPOSITED) scalar_add(o, ARTIS_SCALAR_GAMMA_DEP_DISCRETE, p->e_cmf);
  }
}
/* pellet_gamma_decay gammapkt.cc:894 */
static void pellet_gamma_decay(artis_packet *p) {
  if (p->nu_cmf < 0) {
    p->type = ARTIS_TYPE_KPKT;
    p->absorptiontype = ARTIS_ABSTYPE_PELLET_NOGAMMASPEC;
    return;
  }
  emit_gamma_isotropic(p);
}
/* update_pellet update_packets.cc:185 */
static void update_pellet(Oracle *o, artis_packet *p, double t2) {
  const double ts = p->prop_time;
  const double tdecay = p->tdecay;
  if (tdecay > t2) {
    const double scale = t2 / ts; /* vec_scale(pkt.pos, t2 / ts) */
    p->pos[0] = p->pos[0] * scale; p->pos[1] = p->pos[1] * scale; p->pos[2] = p->pos[2] * scale;
    p->prop_time = t2;
  } else if (tdecay > ts) {
    scalar_add(o, ARTIS_SCALAR_PELLET_DECAYS, 1.);
    p->prop_time = tdecay;
    const double scale = tdecay / ts;
    p->pos[0] = p->pos[0] * scale; p->pos[1] = p->pos[1] * scale; p->pos[2] = p->pos[2] * scale;
    if (p->originated_from_particlenotgamma) {
      if (p->pellet_decaytype == ARTIS_DECAYTYPE_BETAPLUS) {
        p->type = ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAPLUS;
        scalar_add(o, ARTIS_SCALAR_POSITRON_EMISSION, p->e_cmf);
      } else if (p->pellet_decaytype == ARTIS_DECAYTYPE_BETAMINUS) {
        p->type = ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS;
        scalar_add(o, ARTIS_SCALAR_ELECTRON_EMISSION, p->e_cmf);
      } else if (p->pellet_decaytype == ARTIS_DECAYTYPE_ALPHA) {
        scalar_add(o, ARTIS_SCALAR_ALPHA_EMISSION, p->e_cmf);
        p->type = ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA;
      } else if (p->pellet_decaytype == ARTIS_DECAYTYPE_SPONTFISSION) {
        scalar_add(o, ARTIS_SCALAR_SPFISSION_DEP_DISCRETE, p->e_cmf);
        p->type = ARTIS_TYPE_NTALPHA_FISPROD_DEPOSITED;
      } else {
        ORACLE_FAIL(o, "update_pellet: particle pellet with a decay type that emits no particle");
        return;
      }
      p->em_time = (float)p->prop_time;
      p->absorptiontype = ARTIS_ABSTYPE_PELLET_PARTICLEDECAY;
    } else {
      scalar_add(o, ARTIS_SCALAR_GAMMA_EMISSION, p->e_cmf);
      pellet_gamma_decay(p);
    }
  } else if ((tdecay > 0) && (o->ts.nts == 0)) {
    p->e_cmf *= tdecay / o->m->tmin;
    p->type = ARTIS_TYPE_PRE_KPKT;
    p->absorptiontype = ARTIS_ABSTYPE_PELLET_BEFORESIMSTART;
    stat_inc(o, ARTIS_STAT_K_FROM_EARLIERDECAY);
    p->prop_time = o->m->tmin;
  } else {
    ORACLE_FAIL(o, "update_pellet: decay time before the start of the timestep");
  }
}

/* ------------------------------------------------------------------ driver */
/* packetprop_update_required update_packets.cc:321, restricted to the types this path owns */
static int handled_type(int type) {
  /* every type of do_packet()'s switch (update_packets.cc:258-300) */
  return type == ARTIS_TYPE_RPKT || type == ARTIS_TYPE_KPKT || type == ARTIS_TYPE_PRE_KPKT || type == ARTIS_TYPE_GAMMA ||
         type == ARTIS_TYPE_NTLEPTON_DEPOSITED || type == ARTIS_TYPE_NTALPHA_FISPROD_DEPOSITED ||
         type == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS || type == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAPLUS ||
         type == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA || type == ARTIS_TYPE_RADIOACTIVE_PELLET;
}

static void oracle_init(Oracle *o, const artis_model *m, const artis_cellstate *cs, const artis_timestep *ts, artis_estimators *est) {
  memset(o, 0, sizeof(*o));
  o->m = m;
  o->cs = cs;
  o->ts = *ts;
  o->est = *est;
  o->cache = (CellCache *)calloc((size_t)m->npts_nonempty, sizeof(CellCache));
  o->T_step_log = (log(ARTIS_OPT_MAXTEMP) - log(ARTIS_OPT_MINTEMP)) / (ARTIS_OPT_TABLESIZE - 1.); /* ratecoeff.cc:39 */
  for (int i = 0; i < ARTIS_OPT_TABLESIZE + 1; i++) o->temperature_grid[i] = ARTIS_OPT_MINTEMP * exp(i * o->T_step_log); /* ratecoeff.cc:41 */
  o->last_phixs_nuovernuedge = (1.0 + (m->NPHIXSNUINCREMENT * (m->NPHIXSPOINTS - 1))); /* input.cc:310 */
  const char *cap = getenv("ARTIS_ORACLE_CACHE_CAP");
  o->cache_cap = cap ? atoi(cap) : 0;
#if ARTIS_OPT_VPKT_ON
  o->cache_cap = 0; /* a virtual packet fills the caches of the cells it crosses while its caller still reads one: no eviction */
  if (!m->vpkt_obsdirs_costheta || m->vpkt_nspectraperobsdir > VPKT_MAXSPEC || !est->vspecpol) ORACLE_FAIL(o, "VPKT_ON: vpkt configuration / vspecpol missing");
#endif
  /* input.cc:932-955: estimator index of every continuum and the estimators' edge frequencies */
  o->allcont_bfestimindex = (int32_t *)calloc((size_t)m->nbfcontinua + 1, sizeof(int32_t));
  o->bfestim_nu_edge = (double *)calloc((size_t)m->nbfcontinua + 1, sizeof(double));
  o->nbfestim = 0;
  for (int i = 0; i < m->nbfcontinua; i++) {
    const int has = m->allcont_bfestimindex ? (m->allcont_bfestimindex[i] >= 0) : 1;
    if (has) {
      o->allcont_bfestimindex[i] = o->nbfestim;
      o->bfestim_nu_edge[o->nbfestim++] = m->allcont_nu_edge[i];
    } else {
      o->allcont_bfestimindex[i] = -1;
    }
  }
  if (m->allcont_bfestimindex) {
    if (o->nbfestim != m->nbfestim) ORACLE_FAIL(o, "artis_model.nbfestim does not match allcont_bfestimindex");
    for (int i = 0; i < m->nbfcontinua; i++)
      if (m->allcont_bfestimindex[i] != o->allcont_bfestimindex[i]) ORACLE_FAIL(o, "allcont_bfestimindex is not the running count of input.cc:940");
  }
}
double artis_oracle_last_populate_seconds(void) { return g_last_populate_seconds; }
/* the constants of constants.h as restated at the top of this file, for tests/test_oracle_reference_props.py */
int artis_oracle_constants(const char **names, double *values, int maxn) {
  static const char *N[] = {"CLIGHT", "CLIGHT_PROP", "H", "MH", "ME", "PI", "EV", "MEV", "SIGMA_T", "THOMSON_LIMIT", "KB", "SAHACONST",
                            "EULERGAMMA", "CLIGHTSQUARED", "CLIGHTSQUAREDOVERTWOH", "HOVERKB", "HCLIGHTOVERFOURPI", "H_ionpot", "C_0"};
  const double V[] = {CLIGHT, CLIGHT_PROP, H_PLANCK, MH, ME, PI, EV, MEV, SIGMA_T, THOMSON_LIMIT, KB, SAHACONST,
                      EULERGAMMA, CLIGHTSQUARED, CLIGHTSQUAREDOVERTWOH, HOVERKB, HCLIGHTOVERFOURPI, H_ionpot, C_0};
  const int n = (int)(sizeof(V) / sizeof(V[0]));
  for (int i = 0; i < n && i < maxn; i++) {
    names[i] = N[i];
    values[i] = V[i];
  }
  return n;
}
void artis_oracle_set_visit_hist(int64_t *hist, int nlevels) { g_visit_hist = hist; g_visit_nlevels = nlevels; }
static void oracle_free(Oracle *o) {
  for (int c = 0; c < o->m->npts_nonempty; c++) {
    CellCache *cc = &o->cache[c];
    if (cc->populated) cellcache_free_one(cc);
  }
  free(o->cache);
  free(o->allcont_bfestimindex);
  free(o->bfestim_nu_edge);
  g_last_populate_seconds = o->t_populate;
}

/* update_packets update_packets.cc:530 / do_packet update_packets.cc:257 for the r/k-packet types.
 * Returns 0 on success. */
int artis_oracle_update_packets(const artis_model *m, const artis_cellstate *cs, const artis_timestep *ts,
                                artis_packet *packets, int64_t npackets, artis_estimators *est) {
  Oracle o;
  oracle_init(&o, m, cs, ts, est);
  const double ts_end = ts->start + ts->width;
  ContOpacity chi;
  chi.groundcont_gamma_contr = (double *)calloc((size_t)(m->nbfcontinua_ground + 1), sizeof(double));
  chi.gamma_contr = (double *)calloc((size_t)(m->nbfcontinua + 1), sizeof(double));
  chi.bfestimbegin = chi.bfestimend = 0;
  for (int64_t n = 0; n < npackets && !o.error; n++) {
    artis_packet *p = &packets[n];
    while (handled_type(p->type) && p->prop_time < ts_end && !o.error) {
      switch (p->type) {
        case ARTIS_TYPE_RPKT:
          /* do_rpkt rpkt.cc:983, with the per-packet ContinuumOpacity reset on entry (header note 2) */
          chi.nu = -1.; chi.chi_escatter = 0.; chi.chi_freefree_heat = 0.; chi.chi_boundfree = 0.; chi.nonemptymgi = -1;
          while (do_rpkt_step(&o, p, ts_end, &chi) && !o.error) {
          }
          break;
        case ARTIS_TYPE_GAMMA:
          do_gamma(&o, p, ts_end);
          break;
        case ARTIS_TYPE_NTLEPTON_DEPOSITED:
          do_ntlepton_deposit(&o, p);
          break;
        case ARTIS_TYPE_NTALPHA_FISPROD_DEPOSITED:
          do_ntalpha_fisprod_deposit(&o, p);
          break;
        case ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA:
        case ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS:
        case ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAPLUS:
          do_nonthermal_predeposit(&o, p, ts_end);
          break;
        case ARTIS_TYPE_RADIOACTIVE_PELLET:
          update_pellet(&o, p, ts_end);
          break;
        case ARTIS_TYPE_PRE_KPKT:
          do_kpkt_blackbody(&o, p);
          break;
        case ARTIS_TYPE_KPKT: {
          const int c = propcell_nonemptymgi(&o, p->cellindex);
          if (o.cs->thick[c] == ARTIS_CELL_THICK) {
            do_kpkt_blackbody(&o, p);
          } else {
            do_kpkt(&o, p, ts_end);
          }
          break;
        }
        default:
          break;
      }
    }
  }
  free(chi.groundcont_gamma_contr);
  free(chi.gamma_contr);
  const int err = o.error;
  oracle_free(&o);
  return err ? -1 : 0;
}

/* Cell cache of one cell, for populate-kernel parity tests. Arrays sized as in CellCache. */
int artis_oracle_cellcache(const artis_model *m, const artis_cellstate *cs, const artis_timestep *ts, int nonemptymgi,
                           double *levelpops, double *maprocessrates, double *matrans, double *allcont_nnlevel,
                           double *allcont_departure, double *allcont_edgepart, uint64_t *allcont_keepbits,
                           double *corrphotoioncoeff, double *cooling_contrib, double *ion_cooling_contribs,
                           double *chi_ff_nnionpart) {
  Oracle o;
  int64_t stats[ARTIS_NSTATS] = {0};
  artis_estimators est;
  memset(&est, 0, sizeof(est));
  est.stats = stats;
  oracle_init(&o, m, cs, ts, &est);
  cellcache_populate(&o, nonemptymgi);
  for (int element = 0; element < m->nelements; element++)
    for (int ion = 0; ion < get_nions(&o, element); ion++) {
      for (int level = 0; level < get_nlevels(&o, element, ion); level++) (void)macroatom_levelrates(&o, nonemptymgi, element, ion, level);
      (void)cooling_ion_contribs(&o, nonemptymgi, element, ion);
    }
  (void)cell_ion_cooling_contribs(&o, nonemptymgi);
  const CellCache *cc = &o.cache[nonemptymgi];
  memcpy(levelpops, cc->levelpops, sizeof(double) * (size_t)m->nlevels);
  memcpy(maprocessrates, cc->maprocessrates, sizeof(double) * (size_t)m->nlevels * ARTIS_MA_ACTION_COUNT);
  memcpy(matrans, cc->matrans, sizeof(double) * (size_t)m->nmatransblock);
  memcpy(allcont_nnlevel, cc->allcont_nnlevel, sizeof(double) * (size_t)m->nbfcontinua);
  memcpy(allcont_departure, cc->allcont_departure, sizeof(double) * (size_t)m->nbfcontinua);
  memcpy(allcont_edgepart, cc->allcont_edgepart, sizeof(double) * (size_t)m->nbfcontinua);
  memcpy(allcont_keepbits, cc->allcont_keepbits, sizeof(uint64_t) * (size_t)((m->nbfcontinua + 63) / 64));
  memcpy(corrphotoioncoeff, cc->corrphotoioncoeff, sizeof(double) * (size_t)m->nphixstargets_total);
  memcpy(cooling_contrib, cc->cooling_contrib, sizeof(double) * (size_t)m->ncoolingterms);
  memcpy(ion_cooling_contribs, cc->ion_cooling_contribs, sizeof(double) * (size_t)m->nions);
  *chi_ff_nnionpart = cc->chi_ff_nnionpart;
  const int err = o.error;
  oracle_free(&o);
  return err ? -1 : 0;
}

/* ---- small entry points for the known-answer tests of the reference's unit tests (unittests.cc) ---- */
void artis_oracle_rng_seed(uint32_t s[4], uint32_t seed) { rng_seed(s, seed); }
uint32_t artis_oracle_rng_next(uint32_t s[4]) { return rng_next(s); }
float artis_oracle_rng_uniform(uint32_t s[4]) { return rng_uniform(s); }
void artis_oracle_rand_isotropic_unitvec(uint32_t s[4], double out[3]) { get_rand_isotropic_unitvec(s, out); }
void artis_oracle_angle_ab(const double dir1[3], const double vel[3], double out[3]) { angle_ab(dir1, vel, out); }
double artis_oracle_doppler(const double pos[3], const double dir[3], double t) { return doppler_nucmf_on_nurf(pos, dir, t); }
void artis_oracle_move_pkt_withtime(double pos[3], const double dir[3], double *prop_time, double nu_rf, double *nu_cmf, double e_rf,
                                    double *e_cmf, double distance) {
  move_pkt_withtime_raw(pos, dir, prop_time, nu_rf, nu_cmf, e_rf, e_cmf, distance);
}
void artis_oracle_frame_transform(const double n_rf[3], double q0, double u0, const double v[3], double n_cmf[3], double *q, double *u) {
  frame_transform(n_rf, q0, u0, v, n_cmf, q, u);
}
int artis_oracle_closest_transition(const double *linelistnu, int nlines, double nu_cmf, int next_trans) {
  return closest_transition(linelistnu, nlines, nu_cmf, next_trans);
}
double artis_oracle_get_linedistance(double prop_time, double nu_cmf, double nu_trans) { return get_linedistance(prop_time, nu_cmf, nu_trans, -1.); }
double artis_oracle_rad_deexcitation_ratecoeff(double epsilon_trans, float A_ul, double gu, double gl, double nu_, double nl_, double t) {
  return rad_deexcitation_ratecoeff(epsilon_trans, A_ul, gu, gl, nu_, nl_, t);
}
float artis_oracle_phixs_fromtable(const float *xs, int npoints, double nuincrement, double nu_edge, double nu) {
  artis_model m;
  memset(&m, 0, sizeof(m));
  m.NPHIXSPOINTS = npoints;
  m.NPHIXSNUINCREMENT = nuincrement;
  Oracle o;
  memset(&o, 0, sizeof(o));
  o.m = &m;
  o.last_phixs_nuovernuedge = 1.0 + (nuincrement * (npoints - 1));
  return photoionisation_crosssection_fromtable(&o, xs, nu_edge, nu);
}
double artis_oracle_planck(double nu, double T) { return planck(nu, T); }
#if ARTIS_OPT_VPKT_ON
/* sn3d.h:134, :142 -- for the restatement of unittests.cc:68 test_binindex_helpers (the virtual-packet spectra bin with them, vpkt.cc:124-125) */
long long artis_oracle_logbinindex(double value, double minvalue, double dlog, long long nbins) { return (long long)get_logbinindex(value, minvalue, dlog, (ptrdiff_t)nbins); }
double artis_oracle_loggrid_edge(double minvalue, double dlog, double index) { return get_loggrid_edge(minvalue, dlog, index); }
#endif
/* get_escapedirectionbin vectors.h:147 (syn_dir = z constants.h:94; NPHIBINS = NCOSTHETABINS = 10 exspec.h:10-12): the direction bin of
 * an escaping packet, by which the reference's exspec resolves spectra and light curves (spectrum_lightcurve.cc:545, :689). Not on
 * the packet path; restated for tools/exspec.py's direction-resolved spectra and for unittests.cc:175. */
int artis_oracle_escapedirectionbin(const double dir_in[3]) {
  enum { NPHIBINS = 10, NCOSTHETABINS = 10 };
  const double syn_dir[3] = {0., 0., 1.};
  const double xhat[3] = {1.0, 0.0, 0.0};
  const double dirmag = vec_len3(dir_in);
  const double dir[3] = {dir_in[0] / dirmag, dir_in[1] / dirmag, dir_in[2] / dirmag};
  const double costheta = dot3(dir, syn_dir);
  int costhetabin = (int)((costheta + 1.0) * NCOSTHETABINS / 2.0);
  costhetabin = costhetabin < 0 ? 0 : (costhetabin > NCOSTHETABINS - 1 ? NCOSTHETABINS - 1 : costhetabin);
  double vec1[3], vec2[3], vec3[3];
  cross_prod(dir, syn_dir, vec1);
  cross_prod(xhat, syn_dir, vec2);
  const double vec1_len = vec_len3(vec1);
  double cosphi = 1.0;
  if (vec1_len > 1e-12) {
    cosphi = dot3(vec1, vec2) / vec1_len;
    cosphi = cosphi < -1.0 ? -1.0 : (cosphi > 1.0 ? 1.0 : cosphi);
  }
  cross_prod(vec2, syn_dir, vec3);
  const double testphi = dot3(vec1, vec3);
  int phibin = (int)((testphi > 0 ? acos(cosphi) : acos(cosphi) + PI) / 2. / PI * NPHIBINS);
  phibin = phibin < 0 ? 0 : (phibin > NPHIBINS - 1 ? NPHIBINS - 1 : phibin);
  return (costhetabin * NPHIBINS) + phibin;
}
#if ARTIS_EXPOPAC_TABLES
long long artis_oracle_linearbinindex(double value, double minvalue, double binwidth) { return get_linearbinindex(value, minvalue, binwidth); }
double artis_oracle_expopac_bin_nu(long long b, int upper) { return upper ? get_expopac_bin_nu_upper(b) : get_expopac_bin_nu_lower(b); }
#endif
/* gammapkt.h:28, :38, :68 -- for the restatement of unittests.cc:323 test_compton */
double artis_oracle_sigma_compton_partial(double x, double f_max) { return sigma_compton_partial(x, f_max); }
double artis_oracle_choose_f(double xx, double zrand) { return choose_f(xx, zrand); }
double artis_oracle_meanf_sigma(double x) { return meanf_sigma(x); }
void artis_oracle_seed_packets(artis_packet *packets, int64_t npackets, uint32_t seed_base) {
  /* input.cc:1912-1916: packet n gets seed rank_seed_base + n */
  for (int64_t n = 0; n < npackets; n++) rng_seed(packets[n].rngstate, seed_base + (uint32_t)n);
}
size_t artis_oracle_sizeof_packet(void) { return sizeof(artis_packet); }
void artis_oracle_rng_fill_uniform(uint32_t s[4], int64_t n, float *out) {
  for (int64_t i = 0; i < n; i++) out[i] = rng_uniform(s);
}
void artis_oracle_fill_isotropic(uint32_t s[4], int64_t n, double *out3n) {
  for (int64_t i = 0; i < n; i++) get_rand_isotropic_unitvec(s, out3n + (3 * i));
}
