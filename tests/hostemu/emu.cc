// emu.cc -- TEST HARNESS ONLY (never shipped, never loaded by the artis_amd package).
//
// Compiles the kernel bodies of artis_amd/csrc/physics.h for x86 with g++ and runs them in the same
// order the HIP kernels are launched (populate kernels, then advance_packet() with a launch budget and
// an active list), so that the engine's logic can be diffed against the CPU oracle on machines without
// a GPU (the CI container). It is not a CPU fallback: the product library has no such entry point and
// fails without a HIP device.
#define ARTIS_HOST_EMU 1
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#include "../../artis_amd/csrc/model_build.h"
#include "../../artis_amd/csrc/physics.h"

using namespace artis;

namespace {
struct Emu {
  ModelOwned own;
  Env env;
  std::vector<std::vector<uint8_t>> cachebuf;
  std::vector<stat_t> stats;
  std::vector<double> ws;
  std::vector<int32_t> ws_gi, ws_n;
  std::vector<double> nt_ionratecoeff, nt_ionenrate_cum;
  std::vector<double> lineest_count;  // Jb_lu contribution counts as f64
  std::vector<float> expopac_kappa;
  std::vector<double> expopac_planck;
  bool expopac_own = false;
  int32_t pool_resets = 0;  // times the pool of on-demand records was used up and emptied (Env::ma_pool_full)
  VpktConfig vpkt_config;
  int32_t err = 0;
};

void setup(Emu &e, const artis_model *m, const artis_cellstate *cs, const artis_timestep *ts, artis_estimators *est, int64_t nslots) {
  std::memset(&e.env, 0, sizeof(e.env));
  {  // macro-atom record tiers (tables.h "ON-DEMAND RECORDS"): ARTIS_AMD_MA_HOTFRAC / _POOLFRAC as in the engine (default: every level static)
    double hot = 1., pool = 0.25;
    ma_tiers_from_env(&hot, &pool);
    e.env.M = make_host_model_view(*m, e.own, hot, pool);
  }
  e.env.C = make_host_cells_view(*cs);
  e.env.S = make_step(*ts);
  if (est) {
    e.env.E.J = est->J; e.env.E.nuJ = est->nuJ; e.env.E.ffheatingestimator = est->ffheatingestimator;
    e.env.E.colheatingestimator = est->colheatingestimator; e.env.E.gammaestimator = est->gammaestimator;
    e.env.E.bfheatingestimator = est->bfheatingestimator;
    e.env.E.dep_estimator_gamma = est->dep_estimator_gamma;
    e.env.E.dep_estimator_electron = est->dep_estimator_electron;
    e.env.E.dep_estimator_positron = est->dep_estimator_positron;
    e.env.E.dep_estimator_alpha = est->dep_estimator_alpha;
    e.env.E.scalars = est->scalars;
    e.env.E.radfieldbin_J = est->radfieldbin_J;
    e.env.E.radfieldbin_nuJ = est->radfieldbin_nuJ;
    e.env.E.bfrate_raw = est->bfrate_raw;
    e.env.E.vspecpol = est->vspecpol;
    e.env.E.vgrid_flux = est->vgrid_flux;
    e.env.E.Jb_lu_raw = est->Jb_lu_raw;
    if (est->Jb_lu_raw) {  // counted as f64 here (like the engine's block) and handed back as integers at the end
      e.lineest_count.assign((size_t)((int64_t)m->npts_nonempty * m->detailed_linecount) + 1, 0.);
      e.env.E.Jb_lu_contribcount = e.lineest_count.data();
    }
  }
#if ARTIS_OPT_VPKT_ON
  if (!make_vpkt_config(*m, e.vpkt_config) || !est || !est->vspecpol) e.err = 93;
  e.env.M.vpkt = &e.vpkt_config;
  e.env.vpkt_queue = nullptr;  // traced in place
#endif
  // ARTIS_EMU_DPOP=0: no line_dpop rows, the population factor of a line formed where it is read (physics.h line_dpop_at)
  if (const char *b = std::getenv("ARTIS_EMU_DPOP"))
    if (std::atoi(b) == 0) e.env.M.ndpop = 0;
  const DevModel &M = e.env.M;
  const int64_t ncell = M.npts_nonempty;
#define ALLOC(f, T, per) { e.cachebuf.emplace_back((size_t)(ncell * (int64_t)(per) + MAREC_SLACK) * sizeof(T)); e.env.K.f = (T *)e.cachebuf.back().data(); }
  ARTIS_CACHE_ARRAYS(ALLOC, M)
#undef ALLOC
  if (M.ndpop == 0) e.env.K.line_dpop = nullptr;
  for (int64_t i = 0; i < ncell * (int64_t)M.ncold; i++) e.env.K.ma_rowtab[i] = -1;  // k_ma_reset: no cold level has a record yet
  if (M.ncold > 0) *e.env.K.ma_pool_used = 0;
  e.env.ma_pool_cap = (uint32_t)((ncell * (int64_t)M.ma_pool_slots) / MAPOOL_UNIT);
  e.pool_resets = 0;
  e.env.ma_pool_full = &e.pool_resets;  // (the emulation empties a used-up pool on the spot and counts it here: physics.h ma_ensure_record)
  e.stats.assign(ARTIS_NSTATS, 0);
  e.env.stats = e.stats.data();
  e.ws.assign((size_t)((M.nbfcontinua_ground + 1) * nslots), 0.);
  e.env.gamma_ws = e.ws.data();
  e.ws_gi.assign((size_t)((M.nbfcontinua_ground + 1) * nslots), 0);
  e.ws_n.assign((size_t)nslots + 1, 0);
  e.env.gamma_gi = e.ws_gi.data();
  e.env.gamma_n = e.ws_n.data();
  e.env.errflag = &e.err;
  e.env.est_stride = 1;  // the caller's separate arrays
  e.env.pair_stride = 1;
  // ARTIS_EMU_MAFILTERS=0: every macro-atom and cooling decision on the re-added f64 sums (the path of an undecided draw)
  if (const char *b = std::getenv("ARTIS_EMU_MAFILTERS")) e.env.ma_filters_off = (std::atoi(b) == 0) ? 1 : 0;
  e.env.tile_lo = 0;
  e.env.tile_hi = M.npts_nonempty;
  e.env.tile_all = 1;
#if ARTIS_EXPOPAC_TABLES
  if (!e.env.C.expansionopacities) {  // the engine's own tables (artis_amd_set_cellstate), filled in populate_all()
    e.expopac_kappa.assign((size_t)(ncell * ARTIS_EXPOPAC_NBINS) + 1, 0.f);
    e.expopac_planck.assign((size_t)(ncell * ARTIS_EXPOPAC_NBINS) + 1, 0.);
    e.env.C.expansionopacities = e.expopac_kappa.data();
    e.env.C.expansionopacity_planck_cumulative = e.expopac_planck.data();
    e.expopac_own = true;
  }
#endif
#if ARTIS_OPT_NT_ON
  // k_nt_cells (artis_amd_set_cellstate): derived non-thermal arrays of every cell
  e.nt_ionratecoeff.assign((size_t)(ncell * M.nions) + 1, 0.);
  e.nt_ionenrate_cum.assign((size_t)(ncell * M.nions) + 1, 0.);
  e.env.C.nt_ionratecoeff = e.nt_ionratecoeff.data();
  e.env.C.nt_ionenrate_cum = e.nt_ionenrate_cum.data();
  for (int c = 0; c < ncell; c++)
    if (!populate_nt_cell(e.env, c)) e.err = 90;
#endif
}

// the populate kernels, in launch order (artis_engine.hip: k_levelpops, k_line_dpop, k_cell_scalars, k_allcont, k_corrphotoion,
// k_matrans, k_macroatom, k_cooling_head / _chain, k_collexc_filter, k_cooling_tail, k_cooling_prefix)
void populate_all(Emu &e) {
  const DevModel &M = e.env.M;
  std::vector<double> upterms((size_t)M.nupcum + 1, 0.);  // the population's scratch row of cooling terms (Env::collexc_terms)
  for (int c = 0; c < M.npts_nonempty; c++) {
    for (int ul = 0; ul < M.nlevels; ul++) populate_levelpop(e.env, c, ul);
    if (M.ndpop > 0)
      for (int li = 0; li < M.nlines; li++) populate_line_dpop(e.env, c, li);
    populate_chi_ff(e.env, c);
    uint64_t *kb = e.env.K.allcont_keepbits + ((int64_t)c * M.nkeepwords);
    for (int w = 0; w < M.nkeepwords; w++) kb[w] = 0;
    for (int i = 0; i < M.nbfcontinua; i++)
      if (populate_allcont(e.env, c, i)) kb[i / 64] |= UINT64_C(1) << (unsigned)(i % 64);
    if (M.nbfcontinua > 0) populate_keptlist(e.env, c);
    for (int ul = 0; ul < M.nlevels; ul++)
      for (int t = 0; t < M.level_nphixstargets[ul]; t++) populate_corrphotoion(e.env, c, ul, t);
    for (int ul = 0; ul < M.nlevels; ul++) populate_mainit(e.env, c, ul);  // k_mainit (once per engine there)
    for (int ul = 0; ul < M.nlevels; ul++) populate_level_bb(e.env, c, ul, upterms.data());  // k_matrans (+ k_mafilter_long)
    for (int ul = 0; ul < M.nlevels; ul++)
      if (M.level_pack[ul].rec_off >= 0) populate_macroatom(e.env, c, ul);  // (a cold level: when a packet reaches it, ma_slow_fill)
#if ARTIS_EXPOPAC_TABLES
    if (e.expopac_own && e.env.C.thick[c] != ARTIS_CELL_THICK) {  // k_expopac, k_expopac_planck
      for (int b = 0; b < ARTIS_EXPOPAC_NBINS; b++) populate_expopac_bin(e.env, c, b);
      if (ARTIS_OPT_RPKT_BB_THERMALISATION) populate_expopac_planck(e.env, c);
    }
#endif
    for (int ui = 0; ui < M.nions; ui++) populate_cooling_ion(e.env, c, ui, upterms.data());
    for (int li = 0; li < M.ncoollines; li++) populate_coolfilter_line(e.env, c, li, upterms.data());  // k_collexc_filter
    populate_cooling_prefix(e.env, c);
    for (int g = 0; g < M.nguide; g++) populate_cool_guide(e.env, c, g);  // k_cool_guide
  }
}
}  // namespace

extern "C" {

// the constants of constants.h as restated in physics.h, for tests/test_oracle_reference_props.py
// Property check of the macro-atom filters (tables.h "FILTERS") on the functions the kernels use: for cumulative lists of 8
// values and 24-bit draws u, whenever mafilt_count() does not call the draw ambiguous its count equals the number of values
// <= (double)(u * 2^-24f) * whole, the comparison the f64 path makes. Every fourth trial puts a value within a few ulp of
// z * whole (the cases the margins exist for); another fourth a value within a few ulp of the lower edge of the draw's filter cell, with
// a draw 0 ... 4 units of 2^-24 above that edge (the bound of the rule that counts an entry one unit below the draw's). Returns the number of mismatches; *n_ambiguous: draws left to the f64 path.
int64_t artis_emu_mafilter_selftest(int64_t ntrials, uint64_t seed, int64_t *n_ambiguous) {
  using namespace artis;
  uint64_t s = seed ? seed : 1;
  auto next = [&s]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  auto unif = [&next]() { return (double)(next() >> 11) * 0x1.0p-53; };
  int64_t mism = 0, namb = 0;
  for (int64_t t = 0; t < ntrials; t++) {
    double v[8];
    const double scale = std::exp((unif() - 0.5) * 200.);  // wholes from 1e-43 to 1e43
    double run = 0.;
    for (int j = 0; j < 8; j++) {
      if (unif() < 0.7) run += unif() * unif();  // (equal neighbours now and then, like actions without a rate)
      v[j] = run;
    }
    const double whole_raw = (v[7] > 0.) ? v[7] : 1.;
    uint32_t u = (uint32_t)(next() >> 40);  // 24 bits
    // every fourth trial (t & 3 == 1): a draw whose nine bits below the filter's resolution are 0 ... 4 -- the cases around the bound of
    // mafilt_count()'s round-5 rule (an entry one unit below zi is counted where those bits are >= 2) ...
    if ((t & 3) == 1) u = (u & ~0x1FFu) | (uint32_t)(next() % 5);
    const double z = (double)rng_u24_value(u);
    double whole = whole_raw * scale;
    for (int j = 0; j < 8; j++) v[j] *= scale;
    if ((t & 3) == 0) {  // a value right at the decision: z * whole and its neighbours in f64
      const int j = (int)(next() % 7);
      double x = z * whole;
      const int k = (int)(next() % 5) - 2;
      for (int i = 0; i < (k < 0 ? -k : k); i++) x = std::nextafter(x, k < 0 ? 0. : 2. * whole);
      v[j] = x;  // ... kept non-decreasing by moving the neighbours it passes
      for (int i = 0; i < j; i++) v[i] = (v[i] < x) ? v[i] : x;
      for (int i = j + 1; i < 8; i++) v[i] = (v[i] > x) ? v[i] : x;  // (all eight: the entries of a filter are sorted, mafilt_count() relies on it)
    }
    if ((t & 3) == 1 && (u >> 9) > 0) {  // ... with a value within a few ulp of the edge of the filter cell just below the draw's: zi / 32768 * whole
      const int j = (int)(next() % 7);
      double x = ((double)(u >> 9) / MAFILT_SCALE) * whole;
      const int k = (int)(next() % 7) - 3;
      for (int i = 0; i < (k < 0 ? -k : k); i++) x = std::nextafter(x, k < 0 ? 0. : 2. * whole);
      v[j] = x;
      for (int i = 0; i < j; i++) v[i] = (v[i] < x) ? v[i] : x;
      for (int i = j + 1; i < 8; i++) v[i] = (v[i] > x) ? v[i] : x;  // (all eight: the entries of a filter are sorted, mafilt_count() relies on it)
    }
    bool ok = true;
    uint32_t q[8];
    for (int j = 0; j < 8; j++) q[j] = mafilt_quant(v[j], whole, &ok);
    if (!ok) continue;
    U4 f;
    for (int j = 0; j < 4; j++) f.w[j] = q[2 * j] | (q[2 * j + 1] << 16);
    bool amb = false;
    const int cnt = mafilt_count(f, u, &amb);
    if (amb) {
      namb++;
      continue;
    }
    const double target = z * whole;
    int exact = 0;
    for (int j = 0; j < 8; j++) exact += (v[j] <= target) ? 1 : 0;
    if (cnt != exact) mism++;
  }
  if (n_ambiguous) *n_ambiguous = namb;
  return mism;
}

// The same property for the lines' FINE BYTES (tables.h "FINE BYTES"; round 6) on mafilt_quant23() / mafilt_count_fine(): cumulative lists of 7 values,
// 24-bit draws; a third of the trials put a value within a few ulp of z * whole, another third within a few ulp of an edge of the draw's 23-bit
// cell or the cells beside it ((u / 2 + k) * 2^-23 * whole, k = -2 ... 2: the bounds of the rule "u >= 2 q + 3 counts, u <= 2 q - 1 does not"), and some
// lists end in values equal to the whole (never counted). Whenever the fine count does not call the draw undecided it must equal the number of
// values <= (double)(u * 2^-24f) * whole; q23 >> 8 must be mafilt_quant()'s entry; and whenever the 15-bit count decides, the fine count agrees.
// Returns the number of mismatches; *n_ambiguous: draws left to the re-added sums.
int64_t artis_emu_mafilter_fine_selftest(int64_t ntrials, uint64_t seed, int64_t *n_ambiguous) {
  using namespace artis;
  uint64_t s = seed ? seed : 1;
  auto next = [&s]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  auto unif = [&next]() { return (double)(next() >> 11) * 0x1.0p-53; };
  int64_t mism = 0, namb = 0;
  for (int64_t t = 0; t < ntrials; t++) {
    double v[MAREC_PER];
    const double scale = std::exp((unif() - 0.5) * 200.);
    double run = 0.;
    for (int j = 0; j < MAREC_PER; j++) {
      if (unif() < 0.7) run += unif() * unif();
      v[j] = run;
    }
    double whole = ((run > 0.) ? run : 1.) * (1. + ((next() & 1) ? unif() : 0.)) * scale;  // (half of the lists reach the whole at their end)
    for (int j = 0; j < MAREC_PER; j++) v[j] *= scale;
    for (int j = 0; j < MAREC_PER; j++) v[j] = (v[j] < whole) ? v[j] : whole;
    const uint32_t u = (uint32_t)(next() >> 40);
    const double z = (double)rng_u24_value(u);
    const int mode = (int)(t % 3);
    if (mode != 2) {
      const int j = (int)(next() % MAREC_PER);
      double x = z * whole;
      if (mode == 1) x = (((double)(u >> 1) + (double)((int)(next() % 5) - 2)) / 8388608.) * whole;
      if (x < 0.) x = 0.;
      const int k = (int)(next() % 7) - 3;
      for (int i = 0; i < (k < 0 ? -k : k); i++) x = std::nextafter(x, k < 0 ? 0. : 2. * whole);
      if (x > whole) x = whole;
      v[j] = x;
      for (int i = 0; i < j; i++) v[i] = (v[i] < x) ? v[i] : x;
      for (int i = j + 1; i < MAREC_PER; i++) v[i] = (v[i] > x) ? v[i] : x;
    }
    bool ok = true;
    uint32_t q23[MAREC_PER], q15[8];
    for (int j = 0; j < MAREC_PER; j++) {
      q23[j] = mafilt_quant23(v[j], whole, &ok);
      bool ok15 = true;
      q15[j] = mafilt_quant(v[j], whole, &ok15);
      if (ok && ok15 && (q23[j] >> 8) != q15[j]) mism++;
    }
    if (!ok) continue;
    q15[7] = MAFILT_NONE;
    U4 f;
    for (int j = 0; j < 4; j++) f.w[j] = q15[2 * j] | (q15[2 * j + 1] << 16);
    uint64_t fine = 0;
    for (int j = 0; j < MAREC_PER; j++) fine |= (uint64_t)(q23[j] & 0xFFu) << (8 * j);
    const double target = z * whole;
    int exact = 0;
    for (int j = 0; j < MAREC_PER; j++) exact += (v[j] <= target) ? 1 : 0;
    bool amb15 = false, amb = false;
    const int cnt15 = mafilt_count(f, u, &amb15);
    const int cnt = mafilt_count_fine(f, fine, u, &amb);
    if (!amb15 && cnt15 != exact) mism++;
    if (amb) {
      namb++;
      continue;
    }
    if (cnt != exact) mism++;
  }
  if (n_ambiguous) *n_ambiguous = namb;
  return mism;
}

// Property check of the cooling guides (tables.h "COOLING GUIDES") on the functions the kernels use: random cumulative lists (1..400 sums,
// equal neighbours and leading zeros now and then, one dominant term in most), guides of 2^lg ranges; for random 24-bit draws and for the
// first and last draw of every range, guided_upper_bound() must return what upper_bound_d() returns. Returns the number of mismatches;
// *n_noread: draws decided by the two guide entries alone.
int64_t artis_emu_coolguide_selftest(int64_t ntrials, uint64_t seed, int64_t *n_noread) {
  using namespace artis;
  uint64_t s = seed ? seed : 1;
  auto next = [&s]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  auto unif = [&next]() { return (double)(next() >> 11) * 0x1.0p-53; };
  int64_t mism = 0, noread = 0;
  std::vector<double> list;
  std::vector<uint16_t> g;
  for (int64_t t = 0; t < ntrials; t++) {
    const int n = 1 + (int)(next() % 400);
    list.assign((size_t)n, 0.);
    const double scale = std::exp((unif() - 0.5) * 150.);
    double run = 0.;
    const int dominant = (int)(next() % (uint64_t)n);
    for (int j = 0; j < n; j++) {
      const double r = unif();
      if (r > 0.15) run += unif() * unif() * ((j == dominant && unif() < 0.8) ? 1000. * n : 1.);  // (else: an entry that adds nothing)
      list[(size_t)j] = run * scale;
    }
    if (!(list[(size_t)n - 1] > 0.)) list[(size_t)n - 1] = scale;
    const int lg = (int)(next() % 10);  // 1 .. 512 ranges
    const int shift = 24 - lg, nranges = 1 << lg;
    g.assign((size_t)nranges + 2, 0);
    for (int k = 0; k <= nranges; k++) g[(size_t)k] = cool_guide_entry(list.data(), n, shift, k);
    auto check = [&](uint32_t u) {
      const double v = rng_u24_value(u) * list[(size_t)n - 1];
      const int want = upper_bound_d(list.data(), n, v);
      const int got = guided_upper_bound(list.data(), n, v, g.data(), shift, u);
      if (want != got) mism++;
      const int k = (int)(u >> shift);
      if (g[(size_t)k] == g[(size_t)k + 1]) noread++;
    };
    for (int i = 0; i < 64; i++) check((uint32_t)(next() >> 40));
    for (int k = 0; k < nranges; k += (nranges > 16 ? nranges / 16 : 1)) {
      check((uint32_t)k << shift);
      check((((uint32_t)k + 1u) << shift) - 1u);
    }
    check(0u);
    check(0xFFFFFFu);
  }
  if (n_noread) *n_noread = noread;
  return mism;
}

int artis_emu_constants(const char **names, double *values, int maxn) {
  using namespace artis;
  static const char *N[] = {"CLIGHT", "CLIGHT_PROP", "H", "MH", "ME", "PI", "EV", "MEV", "SIGMA_T", "THOMSON_LIMIT", "KB", "SAHACONST",
                            "EULERGAMMA", "CLIGHTSQUARED", "CLIGHTSQUAREDOVERTWOH", "HOVERKB", "HCLIGHTOVERFOURPI", "H_ionpot", "C_0"};
  const double V[] = {CLIGHT, CLIGHT_PROP, HPLANCK, MH, ME, PI, EV, MEV, SIGMA_T, THOMSON_LIMIT, KB, SAHACONST,
                      EULERGAMMA, CLIGHTSQUARED, CLIGHTSQUAREDOVERTWOH, HOVERKB, HCLIGHTOVERFOURPI, H_ionpot, C_0};
  const int n = (int)(sizeof(V) / sizeof(V[0]));
  for (int i = 0; i < n && i < maxn; i++) {
    names[i] = N[i];
    values[i] = V[i];
  }
  return n;
}
// search helpers of physics.h, for tests/test_kernel_bodies_vs_oracle.py
int artis_emu_upper_bound(const double *a, int n, double v) { return artis::upper_bound_d(a, n, v); }
int artis_emu_lower_bound(const double *a, int n, double v) { return artis::lower_bound_d(a, n, v); }
int artis_emu_upper_bound_wide(const double *a, int n, double v) { return artis::upper_bound_wide(a, n, v); }
int artis_emu_upper_bound_blocked6(const double *a, int n, double v) { return artis::upper_bound_blocked<6>(a, n, v); }
int artis_emu_upper_bound_blocked16(const double *a, int n, double v) { return artis::upper_bound_blocked<16>(a, n, v); }
// Compton cross-section helpers of physics.h (gammapkt.h:28, :38, :68), for the restatement of unittests.cc:323
double artis_emu_sigma_compton_partial(double x, double f_max) { return artis::sigma_compton_partial(x, f_max); }
double artis_emu_choose_f(double xx, double zrand) { return artis::choose_f(xx, zrand); }
double artis_emu_meanf_sigma(double x) { return artis::meanf_sigma(x); }
double artis_emu_planck(double nu, double T) { return artis::planck(nu, T); }
#if ARTIS_OPT_VPKT_ON
// the bin helpers of the virtual-packet spectra (sn3d.h:134, :142): the kernels' index, the host's edges; for unittests.cc:68
long long artis_emu_logbinindex(double value, double minvalue, double dlog, long long nbins) { return (long long)artis::logbinindex(value, minvalue, dlog, nbins); }
double artis_emu_loggrid_edge(double minvalue, double dlog, double index) { return artis::loggrid_edge(minvalue, dlog, index); }
#endif
#if ARTIS_EXPOPAC_TABLES
// the wavelength-bin helpers of the expansion opacities (sn3d.h:115, rpkt.h:30-40), for tests/test_oracle_reference_props.py
long long artis_emu_linearbinindex(double value, double minvalue, double binwidth) { return artis::linearbinindex(value, minvalue, binwidth); }
double artis_emu_expopac_bin_nu(long long b, int upper) { return upper ? artis::expopac_bin_nu_upper(b) : artis::expopac_bin_nu_lower(b); }
#endif
int artis_emu_closest_transition(const double *nu, int nlines, double nu_cmf, int next_trans) {
  return artis::closest_transition(nu, nlines, nu_cmf, next_trans);
}

static int g_last_pool_resets = 0;
int artis_emu_update_packets(const artis_model *m, const artis_cellstate *cs, const artis_timestep *ts, artis_packet *packets,
                             int64_t npackets, artis_estimators *est, int budget) {
  Emu e;
  setup(e, m, cs, ts, est, npackets > 0 ? npackets : 1);
  populate_all(e);
  std::vector<uint8_t> recbuf(pkt_store_bytes(npackets) + 128);
  void *recbase = (void *)(((uintptr_t)recbuf.data() + 127) & ~(uintptr_t)127);
  e.env.P = carve_pkt_store(recbase, npackets);
  for (int64_t i = 0; i < npackets; i++) aos_to_rec(packets[i], e.env.P, i);
  // work lists + budgeted launches, as in artis_amd_update_packets_device(): one list per kind of pending work;
  // a launch consumes the whole current list of its kind and appends to the lists of the other kinds
  std::vector<int64_t> lists[NEXT_NKINDS], self;
  for (int64_t i = 0; i < npackets; i++) {
    Pkt p;
    pkt_load(e.env.P, i, p);
    const int kind = classify(e.env, p, e.env.S.ts_end);
    e.env.P.hot[i].chi_mgi = -1;  // k_classify: a ContinuumOpacity never survives into another update_packets() call
    if (kind != NEXT_DONE) lists[kind].push_back(i);
  }
  auto any = [&] {
    for (int k = 1; k < NEXT_NKINDS; k++)
      if (!lists[k].empty()) return true;
    return false;
  };
  const int order[6] = {NEXT_SLOW, NEXT_GAMMA, NEXT_BB, NEXT_KPKT, NEXT_MA, NEXT_RPKT};
  const bool split = std::getenv("ARTIS_EMU_SPLIT") != nullptr && std::atoi(std::getenv("ARTIS_EMU_SPLIT")) != 0;
  while (any() && !e.err) {
    for (int kind : order) {
      if (lists[kind].empty()) continue;
      std::vector<int64_t> cur;
      cur.swap(lists[kind]);
      for (int64_t pi : cur) {
        Pkt p;
        int next = NEXT_DONE;
        if (kind == NEXT_MA || kind == NEXT_KPKT || kind == NEXT_BB) {
          // as k_thermal / k_blackbody do: only the hot line is loaded, the flight line is written when (and only when) an
          // r-packet was emitted
          pkt_load_thermal(e.env.P, pi, p);
          // ARTIS_EMU_SPLIT=1: the form k_thermal runs -- undecided searches become pending slow-path actions
          if (split)
            next = (kind == NEXT_MA) ? advance_ma<true>(e.env, p, pi, budget * 8)
                                     : ((kind == NEXT_KPKT) ? advance_kpkt<true>(e.env, p, pi) : advance_blackbody(e.env, p, pi));
          else
          next = (kind == NEXT_MA) ? advance_ma(e.env, p, pi, budget * 8)
                                   : ((kind == NEXT_KPKT) ? advance_kpkt(e.env, p, pi) : advance_blackbody(e.env, p, pi));
          pkt_store_thermal(e.env.P, pi, p);
        } else {
          pkt_load(e.env.P, pi, p);
          if (kind == NEXT_RPKT) {
            Chi x;
            chi_load(e.env.P, pi, p, x);
            next = advance_rpkt(e.env, p, pi, x, budget);
            chi_store(e.env.P, pi, p, x);
          } else if (kind == NEXT_GAMMA) {
            next = advance_gamma(e.env, p, pi, budget);
          } else {
            next = advance_slow(e.env, p, pi);
          }
          pkt_store(e.env.P, pi, p);
        }
        if (next != NEXT_DONE) lists[next].push_back(pi);
      }
    }
  }
  for (int64_t i = 0; i < npackets; i++) rec_to_aos(e.env.P, i, packets[i]);
  if (est && est->stats)
    for (int i = 0; i < ARTIS_NSTATS; i++) est->stats[i] += (int64_t)e.stats[i];
  if (est && est->stats) est->stats[ARTIS_STAT_UPDATECELL] += e.env.M.npts_nonempty;
  if (est && est->Jb_lu_contribcount)
    for (size_t i = 0; i + 1 < e.lineest_count.size(); i++) est->Jb_lu_contribcount[i] += (int64_t)e.lineest_count[i];
  g_last_pool_resets = e.pool_resets;
  return e.err;
}
// times the last artis_emu_update_packets() call found the pool of on-demand records used up and emptied it
int artis_emu_last_pool_resets() { return g_last_pool_resets; }

int artis_emu_cellcache(const artis_model *m, const artis_cellstate *cs, const artis_timestep *ts, int c, double *levelpops,
                        double *maprocessrates, double *matrans, double *allcont_nnlevel, double *allcont_departure,
                        double *allcont_edgepart, uint64_t *allcont_keepbits, double *corrphotoioncoeff, double *cooling_contrib,
                        double *ion_cooling_contribs, double *chi_ff_nnionpart) {
  Emu e;
  setup(e, m, cs, ts, nullptr, 1);
  populate_all(e);
  const DevModel &M = e.env.M;
  const DevCache &K = e.env.K;
  std::memcpy(levelpops, K.levelpops + (int64_t)c * M.nlevels, sizeof(double) * M.nlevels);
  if (M.ncold > 0) {
    // on-demand records (ARTIS_AMD_MA_HOTFRAC < 1): the view shows every level, so the cold ones of this cell are filled the way a packet's
    // first visit fills them (physics.h ma_slow_fill) -- the test then holds the on-demand form against the oracle like the population's
    Pkt p;
    std::memset(&p, 0, sizeof(p));
    int cellindex = 0;
    while (M.propcell_nonemptymgi[cellindex] != c) cellindex++;
    p.cellindex = cellindex;
    for (int ui = 0; ui < M.nions; ui++)
      for (int l = 0; l < M.ion_nlevels[ui]; l++) {
        if (M.level_pack[M.ion_uniquelevelindexstart[ui] + l].rec_off >= 0) continue;
        p.ma_element = M.ion_element[ui];
        p.ma_ion = ui - M.elem_uniqueionindexstart[p.ma_element];
        p.ma_level = l;
        p.pend = PEND_MA_FILL;
        ma_slow_fill(e.env, p);
      }
  }
  for (int ul = 0; ul < M.nlevels; ul++)
    if (debug_level_record(e.env, c, ul, maprocessrates, matrans) != 0) e.err = 94;
  std::memcpy(allcont_nnlevel, K.allcont_nnlevel + (int64_t)c * M.nbfcontinua, sizeof(double) * M.nbfcontinua);
  std::memcpy(allcont_departure, K.allcont_departure + (int64_t)c * M.nbfcontinua, sizeof(double) * M.nbfcontinua);
  std::memcpy(allcont_edgepart, K.allcont_edgepart + (int64_t)c * M.nbfcontinua, sizeof(double) * M.nbfcontinua);
  std::memcpy(allcont_keepbits, K.allcont_keepbits + (int64_t)c * M.nkeepwords, sizeof(uint64_t) * ((M.nbfcontinua + 63) / 64));
  std::memcpy(corrphotoioncoeff, K.corrphotoioncoeff + (int64_t)c * M.nphixstargets_total, sizeof(double) * M.nphixstargets_total);
  std::memcpy(cooling_contrib, K.cooling_contrib + (int64_t)c * M.ncoolingterms, sizeof(double) * M.ncoolingterms);
  std::memcpy(ion_cooling_contribs, K.ion_cooling_contribs + (int64_t)c * M.nions, sizeof(double) * M.nions);
  *chi_ff_nnionpart = K.chi_ff_nnionpart[c];
  return e.err;
}
}
