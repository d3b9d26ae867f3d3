// model_build.h -- host-side construction of the kernel views (tables.h) from the C-ABI structs.
// Shared by the HIP engine (which then mirrors every array into HBM) and by the host-emulation
// test build. No physics here: only pointer plumbing and the three derived tables
// (temperature grid ratecoeff.cc:39-46, level -> ion map, packed line records).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tables.h"

namespace artis {

struct ModelOwned {
  std::vector<double> temperature_grid;
  std::vector<int32_t> level_ion;
  std::vector<LinePack> line_pack;
  std::vector<LevelPack> level_pack;
  std::vector<int32_t> level_upcum_start;
  std::vector<int32_t> alltrans_owner;
  std::vector<int32_t> scanblk_seg0;
  std::vector<uint8_t> scanperm;
  std::vector<MaLongSeg> scansegs, malongsegs;
  std::vector<MaTarget> alltrans_target;
  std::vector<uint16_t> alltrans_tlevel16;
  std::vector<CoolLineRef> coollines;
  std::vector<ContPack> cont_pack;
  std::vector<int32_t> expopac_linestart;
  std::vector<int32_t> upcum_coolslot;
  std::vector<int32_t> level_recomb_start, recomb_lower, recomb_target, recomb_levels, ion_cooltail_start, level_coolhi, ion_guideoff, ion_guideshift;
};

// X(field, element type, element count) for every array pointer of DevModel
#define ARTIS_MODEL_ARRAYS(X, m)                                                   \
  X(temperature_grid, double, (ARTIS_OPT_TABLESIZE + 1))                           \
  X(elem_nions, int32_t, (m).nelements)                                            \
  X(elem_uniqueionindexstart, int32_t, (m).nelements)                              \
  X(elem_lowest_ionstage, int32_t, (m).nelements)                                  \
  X(elem_anumber, int32_t, (m).nelements)                                          \
  X(ion_element, int32_t, (m).nions)                                               \
  X(ion_nlevels, int32_t, (m).nions)                                               \
  X(ion_nlevels_ionising, int32_t, (m).nions)                                      \
  X(ion_maxrecombininglevel, int32_t, (m).nions)                                   \
  X(ion_uniquelevelindexstart, int32_t, (m).nions)                                 \
  X(ion_coolingoffset, int32_t, (m).nions)                                         \
  X(ion_ncoolingterms, int32_t, (m).nions)                                         \
  X(level_epsilon, double, (m).nlevels)                                            \
  X(level_statweight, float, (m).nlevels)                                          \
  X(level_alltrans_startdown, int32_t, (m).nlevels)                                \
  X(level_ndowntrans, int32_t, (m).nlevels)                                        \
  X(level_nuptrans, int32_t, (m).nlevels)                                          \
  X(level_closestgroundlevelcont, int32_t, (m).nlevels)                            \
  X(level_phixsstart, int32_t, (m).nlevels)                                        \
  X(level_nphixstargets, int32_t, (m).nlevels)                                     \
  X(level_phixstargetstart, int32_t, (m).nlevels)                                  \
  X(level_bflist_start, int32_t, (m).nlevels)                                      \
  X(level_matransblock_start, int32_t, (m).nlevels)                                \
  X(level_ion, int32_t, (m).nlevels)                                               \
  X(level_pack, LevelPack, (m).nlevels)                                            \
  X(level_coolhi, int32_t, (m).nlevels)                                            \
  X(ion_guideoff, int32_t, (m).nions)                                              \
  X(ion_guideshift, int32_t, (m).nions)                                            \
  X(level_upcum_start, int32_t, (m).nlevels)                                       \
  X(alltrans_lineindex, int32_t, (m).nalltrans)                                    \
  X(alltrans_targetlevelindex, int32_t, (m).nalltrans)                             \
  X(alltrans_owner, int32_t, (m).nalltrans)                                        \
  X(scansegs, MaLongSeg, ((m).nscansegs > 0 ? (m).nscansegs : 1))                  \
  X(scanblk_seg0, int32_t, ((m).nscanblk + 1))                                     \
  X(scanperm, uint8_t, ((m).nalltrans > 0 ? (m).nalltrans : 1))                    \
  X(malongsegs, MaLongSeg, ((m).nmalongsegs > 0 ? (m).nmalongsegs : 1))            \
  X(alltrans_target, MaTarget, (m).nalltrans)                                      \
  X(alltrans_tlevel16, uint16_t, (((m).nalltrans + 1) & ~1)) /* whole 32-bit words: k_thermal copies it as words */ \
  X(coollines, CoolLineRef, ((m).ncoollines > 0 ? (m).ncoollines : 1))             \
  X(alltrans_einstein_A, float, (m).nalltrans)                                     \
  X(alltrans_coll_str, float, (m).nalltrans)                                       \
  X(alltrans_osc_strength, float, (m).nalltrans)                                   \
  X(alltrans_forbidden, uint8_t, (m).nalltrans)                                    \
  X(line_nu, double, (m).nlines)                                                   \
  X(line_pack, LinePack, (m).nlines)                                               \
  X(cont_pack, ContPack, (m).nbfcontinua)                                          \
  X(line_elementindex, int32_t, (m).nlines)                                        \
  X(line_ionindex, int32_t, (m).nlines)                                            \
  X(allphixs, float, ((int64_t)(m).nphixslevels * (m).NPHIXSPOINTS))               \
  X(allphixstargets_levelindex, int32_t, (m).nphixstargets_total)                  \
  X(allphixstargets_probability, double, (m).nphixstargets_total)                  \
  X(allcont_nu_edge, double, (m).nbfcontinua)                                      \
  X(allcont_element, int32_t, (m).nbfcontinua)                                     \
  X(allcont_ion, int32_t, (m).nbfcontinua)                                         \
  X(allcont_level, int32_t, (m).nbfcontinua)                                       \
  X(allcont_phixstargetindex, int32_t, (m).nbfcontinua)                            \
  X(allcont_upperlevel, int32_t, (m).nbfcontinua)                                  \
  X(allcont_uniquelevelindex, int32_t, (m).nbfcontinua)                            \
  X(allcont_groundcontestimindex, int32_t, (m).nbfcontinua)                        \
  X(allcont_probability, double, (m).nbfcontinua)                                  \
  X(groundcont_nu_edge, double, (m).nbfcontinua_ground)                            \
  X(spontrecombcoeffs, double, ((int64_t)(m).nbfcontinua * ARTIS_OPT_TABLESIZE))   \
  X(corrphotoioncoeffs, double, ((int64_t)(m).nbfcontinua * ARTIS_OPT_TABLESIZE))  \
  X(bfcooling_coeffs, double, ((int64_t)(m).nbfcontinua * ARTIS_OPT_TABLESIZE))    \
  X(coolinglist_type, uint8_t, (m).ncoolingterms)                                  \
  X(coolinglist_level, int32_t, (m).ncoolingterms)                                 \
  X(coolinglist_phixstargetindex, int32_t, (m).ncoolingterms)                      \
  X(expopac_linestart, int32_t, (ARTIS_EXPOPAC_NBINS + 1))                         \
  X(upcum_coolslot, int32_t, (m).nupcum)                                           \
  X(level_recomb_start, int32_t, ((m).nlevels + 1))                                \
  X(recomb_lower, int32_t, (m).nrecomb)                                            \
  X(recomb_target, int32_t, (m).nrecomb)                                           \
  X(recomb_levels, int32_t, (m).nrecomblevels)                                     \
  X(ion_cooltail_start, int32_t, (m).nions)                                        \
  X(propcell_nonemptymgi, int32_t, (m).ngrid)

// arrays of DevModel that may be absent (null) on the host
#define ARTIS_MODEL_OPTIONAL_ARRAYS(X, m)      \
  X(elem_meannucmass, float, (m).nelements)    \
  X(ion_nt_sum_q_over_binding, double, (m).nions) \
  X(allcont_bfestimindex, int32_t, (m).nbfcontinua) \
  X(rho_tmin, float, (m).npts_nonempty) \
  X(xcom_elem_start, int32_t, ((m).nelements + 1)) \
  X(xcom_energy, double, (m).nxcom) \
  X(xcom_sigma, double, (m).nxcom) \
  X(detailed_lineindices, int32_t, (m).detailed_linecount)

// X(field, element type, element count) for every array pointer of DevCells
#define ARTIS_CELL_ARRAYS(X, m)                                          \
  X(rho, float, (m).npts_nonempty)                                       \
  X(Te, float, (m).npts_nonempty)                                        \
  X(TJ, float, (m).npts_nonempty)                                        \
  X(TR, float, (m).npts_nonempty)                                        \
  X(W, float, (m).npts_nonempty)                                         \
  X(nne, float, (m).npts_nonempty)                                       \
  X(nnetot, float, (m).npts_nonempty)                                    \
  X(kappagrey, float, (m).npts_nonempty)                                 \
  X(clumpfactor, float, (m).npts_nonempty)                               \
  X(thick, int32_t, (m).npts_nonempty)                                   \
  X(ion_groundlevelpops, float, ((int64_t)(m).npts_nonempty * (m).nions)) \
  X(ion_partfuncts, float, ((int64_t)(m).npts_nonempty * (m).nions))     \
  X(elem_massfracs, float, ((int64_t)(m).npts_nonempty * (m).nelements)) \
  X(corrphotoionrenorm, double, ((int64_t)(m).npts_nonempty * ((m).nbfcontinua_ground > 0 ? (m).nbfcontinua_ground : 1))) \
  X(ffegrp, float, (m).npts_nonempty)

// the optional arrays of DevCells (uploaded only when the caller hands them over)
#define ARTIS_CELL_OPTIONAL_ARRAYS(X, m)                                                            \
  X(levelpops, double, ((int64_t)(m).npts_nonempty * (m).nlevels))                                  \
  X(corrphotoioncoeff, double, ((int64_t)(m).npts_nonempty * (m).nphixstargets_total))              \
  X(radfieldbin_W, float, ((int64_t)(m).npts_nonempty * ARTIS_OPT_RADFIELDBINCOUNT))                \
  X(radfieldbin_T_R, float, ((int64_t)(m).npts_nonempty * ARTIS_OPT_RADFIELDBINCOUNT))                \
  X(nt_frac_ionisation, float, (m).npts_nonempty)                                                   \
  X(nt_frac_excitation, float, (m).npts_nonempty)                                                   \
  X(nt_deposition_rate_density, double, (m).npts_nonempty)                                          \
  X(nt_eff_ionpot, float, ((int64_t)(m).npts_nonempty * (m).nions))                                 \
  X(nt_prob_num_auger, float, ((int64_t)(m).npts_nonempty * (m).nions * (ARTIS_OPT_NT_MAX_AUGER_ELECTRONS + 1)))      \
  X(nt_ionenfrac_num_auger, float, ((int64_t)(m).npts_nonempty * (m).nions * (ARTIS_OPT_NT_MAX_AUGER_ELECTRONS + 1))) \
  X(nt_exc_count, int32_t, (m).npts_nonempty)                                                       \
  X(nt_exc_frac_deposition, double, ((int64_t)(m).npts_nonempty * (nt_stored)))                     \
  X(nt_exc_ratecoeffperdeposition, double, ((int64_t)(m).npts_nonempty * (nt_stored)))              \
  X(nt_exc_alltransindex, int32_t, ((int64_t)(m).npts_nonempty * (nt_stored)))                      \
  X(expansionopacities, float, ((int64_t)(m).npts_nonempty * ARTIS_EXPOPAC_NBINS))                  \
  X(expansionopacity_planck_cumulative, double, ((int64_t)(m).npts_nonempty * ARTIS_EXPOPAC_NBINS)) \
  X(Jb_lu_normed, double, ((int64_t)(m).npts_nonempty * (m).detailed_linecount))            \
  X(elem_meanweight, float, ((int64_t)(m).npts_nonempty * (m).nelements))

// X(field, element type, elements per cell) for every array of DevCache
#define ARTIS_CACHE_ARRAYS(X, m)                                \
  X(levelpops, double, (m).nlevels)                             \
  X(macache, U4, (m).nmacache)                                  \
  X(ma_rowtab, int32_t, (m).ncold)                              \
  X(ma_pool, U4, (m).ma_pool_slots)                             \
  X(ma_pool_used, uint32_t, ((m).ncold > 0 ? 1 : 0))            \
  X(allcont_nnlevel, double, (m).nbfcontinua)                   \
  X(allcont_departure, double, (m).nbfcontinua)                 \
  X(allcont_edgepart, double, (m).nbfcontinua)                  \
  X(allcont_pair, D2, (m).nbfcontinua)                          \
  X(allcont_keepbits, uint64_t, (m).nkeepwords)                 \
  X(allcont_keptlist, int32_t, (m).nbfcontinua)                 \
  X(allcont_keepprefix, int32_t, (m).nkeepwords)                \
  X(allcont_keptpair, D2, (m).nbfcontinua)                      \
  X(corrphotoioncoeff, double, (m).nphixstargets_total)         \
  X(bf_radrecomb, double, (m).nphixstargets_total)              \
  X(bf_colrecomb, double, (m).nphixstargets_total)              \
  X(bf_colion, double, (m).nphixstargets_total)                 \
  X(bf_cooling, double, (m).nphixstargets_total)                \
  X(cooling_contrib, double, (m).ncoolingterms)                 \
  X(line_dpop, double, (m).ndpop)                               \
  X(ion_cooling_contribs, double, (m).nions)                    \
  X(ion_cooling_C, double, (m).nions)                           \
  X(cool_guide, uint16_t, (m).nguide)                           \
  X(chi_ff_nnionpart, double, 1)

// Host view of the model: pointers into the caller's arrays plus the derived tables in `own`.
// The macro-atom record tiers (tables.h "ON-DEMAND RECORDS"): ARTIS_AMD_MA_HOTFRAC = share of every ion's levels (the lowest ones) with a
// static record in every cell's row, ARTIS_AMD_MA_POOLFRAC = the pool (shared by the resident cells) as a share of what the cold levels'
// records of all of them would take.
// Unset: hot 1 (everything static); the engine sets them itself when the whole cache does not fit one tile.
inline void ma_tiers_from_env(double *hotfrac, double *poolfrac) {
  if (const char *b = std::getenv("ARTIS_AMD_MA_HOTFRAC")) *hotfrac = std::min(1., std::max(0., std::atof(b)));
  if (const char *b = std::getenv("ARTIS_AMD_MA_POOLFRAC")) *poolfrac = std::min(1., std::max(0., std::atof(b)));
}
inline DevModel make_host_model_view(const artis_model &m, ModelOwned &own, double hotfrac = 1., double poolfrac = 0.25) {
  DevModel v;
  std::memset(&v, 0, sizeof(v));
  v.nelements = m.nelements; v.nions = m.nions; v.nlevels = m.nlevels; v.nlines = m.nlines; v.nalltrans = m.nalltrans;
  v.nphixstargets_total = m.nphixstargets_total; v.nphixslevels = m.nphixslevels; v.nbfcontinua = m.nbfcontinua;
  v.nbfcontinua_ground = m.nbfcontinua_ground; v.ncoolingterms = m.ncoolingterms; v.nmatransblock = m.nmatransblock;
  v.NPHIXSPOINTS = m.NPHIXSPOINTS;
  v.ndpop = m.nlines;  // (the engine drops the rows when the cell cache would not fit one tile with them)
  v.nkeepwords = (((m.nbfcontinua + 63) / 64) + 3) & ~3;  // rows padded to whole 4-word chunks (physics.h KeepIter)
  v.NPHIXSNUINCREMENT = m.NPHIXSNUINCREMENT;
  v.last_phixs_nuovernuedge = (1.0 + (m.NPHIXSNUINCREMENT * (m.NPHIXSPOINTS - 1)));                    // input.cc:310
  v.T_step_log = (std::log(ARTIS_OPT_MAXTEMP) - std::log(ARTIS_OPT_MINTEMP)) / (ARTIS_OPT_TABLESIZE - 1.);  // ratecoeff.cc:39
  own.temperature_grid.resize(ARTIS_OPT_TABLESIZE + 1);
  for (int i = 0; i < ARTIS_OPT_TABLESIZE + 1; i++) own.temperature_grid[i] = ARTIS_OPT_MINTEMP * std::exp(i * v.T_step_log);
  own.level_ion.assign(m.nlevels, 0);
  for (int ui = 0; ui < m.nions; ui++)
    for (int l = 0; l < m.ion_nlevels[ui]; l++) own.level_ion[m.ion_uniquelevelindexstart[ui] + l] = ui;
  own.line_pack.resize(m.nlines);
  for (int i = 0; i < m.nlines; i++)
    own.line_pack[i] = LinePack{m.line_uniquelevelindex_lower[i], m.line_uniquelevelindex_upper[i], m.line_B_ul[i], m.line_B_lu[i]};
  v.temperature_grid = own.temperature_grid.data();
  v.level_ion = own.level_ion.data();
  v.line_pack = own.line_pack.data();
  own.level_pack.resize(m.nlevels);
  int32_t rec = 0;  // in 16-byte slots
  int32_t ncold = 0;
  int64_t cold_slots = 0;
  for (int ui = 0; ui < m.nions; ui++) {
    const int nl = m.ion_nlevels[ui];
    const int nhot = (hotfrac >= 1.) ? nl : std::min(nl, std::max(1, (int)std::ceil(hotfrac * nl)));  // (an ion's ground level is always hot)
    for (int l = 0; l < nl; l++) {
      const int i = m.ion_uniquelevelindexstart[ui] + l;
      const int sz = ((marec_slots(m.level_ndowntrans[i], m.level_nuptrans[i]) + MAREC_ALIGN - 1) / MAREC_ALIGN) * MAREC_ALIGN;
      if (l < nhot) {
        own.level_pack[i] = LevelPack{rec, m.level_alltrans_startdown[i], m.level_ndowntrans[i], m.level_nuptrans[i]};
        rec += sz;
      } else {
        own.level_pack[i] = LevelPack{-(ncold++) - 1, m.level_alltrans_startdown[i], m.level_ndowntrans[i], m.level_nuptrans[i]};
        cold_slots += ((sz + MAPOOL_UNIT - 1) / MAPOOL_UNIT) * MAPOOL_UNIT;  // (the pool's records are whole units)
      }
    }
  }
  v.ncold = ncold;
  v.ma_pool_slots = (ncold > 0) ? (int32_t)(((int64_t)std::ceil(poolfrac * (double)cold_slots) + MAPOOL_UNIT - 1) / MAPOOL_UNIT) * MAPOOL_UNIT : 0;
  v.nmacache = rec;
  v.level_pack = own.level_pack.data();
  // what a transition needs to know of the level it leads to (tables.h MaTarget): static, one table for all cells
  own.alltrans_target.assign((size_t)(m.nalltrans > 0 ? m.nalltrans : 1), MaTarget{0, 0, 0, 0});
  for (int ui = 0; ui < m.nions; ui++) {
    const int start = m.ion_uniquelevelindexstart[ui];
    for (int l = 0; l < m.ion_nlevels[ui]; l++) {
      const LevelPack &lp = own.level_pack[start + l];
      for (int t = 0; t < lp.ndown + lp.nup; t++) {
        const int tl = m.alltrans_targetlevelindex[lp.alltrans_startdown + t];
        const LevelPack &tp = own.level_pack[start + tl];
        own.alltrans_target[lp.alltrans_startdown + t] = MaTarget{tp.rec_off, tp.alltrans_startdown, tl, (uint32_t)tp.ndown | ((uint32_t)tp.nup << 16)};
      }
    }
  }
  v.alltrans_target = own.alltrans_target.data();
  own.alltrans_tlevel16.assign((size_t)(m.nalltrans > 0 ? ((m.nalltrans + 1) & ~1) : 2), 0);  // padded to whole 32-bit words
  for (int i = 0; i < m.nalltrans; i++) own.alltrans_tlevel16[i] = (uint16_t)m.alltrans_targetlevelindex[i];
  v.alltrans_tlevel16 = own.alltrans_tlevel16.data();
  own.level_upcum_start.resize(m.nlevels);
  int32_t nupcum = 0;
  for (int i = 0; i < m.nlevels; i++) {
    own.level_upcum_start[i] = nupcum;
    nupcum += m.level_nuptrans[i];
  }
  v.nupcum = nupcum;
  own.alltrans_owner.assign((size_t)(m.nalltrans > 0 ? m.nalltrans : 1), 0);
  for (int i = 0; i < m.nlevels; i++)
    for (int t = 0; t < m.level_ndowntrans[i] + m.level_nuptrans[i]; t++) own.alltrans_owner[m.level_alltrans_startdown[i] + t] = i;
  v.alltrans_owner = own.alltrans_owner.data();
  {
    // the work of k_matrans (tables.h DevModel::scansegs): the (level, direction) segments in alltrans order, and blocks of whole
    // segments with at most MATRANS_BLOCK entries together; a longer segment is a block of its own
    own.scansegs.clear();
    own.scanblk_seg0.assign(1, 0);
    own.malongsegs.clear();
    int blk_fill = 0;
    for (int i = 0; i < m.nlevels; i++) {
      const LevelPack &lp = own.level_pack[i];
      for (int dir = 0; dir < 2; dir++) {
        const int n = dir == 0 ? lp.ndown : lp.nup;
        if (n <= 0) continue;
        const int sa = lp.alltrans_startdown + (dir == 0 ? 0 : lp.ndown);
        if (blk_fill > 0 && (blk_fill + n > MATRANS_BLOCK)) {
          own.scanblk_seg0.push_back((int32_t)own.scansegs.size());
          blk_fill = 0;
        }
        own.scansegs.push_back(MaLongSeg{sa, n, i, dir});
        blk_fill += n;
        if (n > MATRANS_BLOCK) {
          if (lp.rec_off >= 0) own.malongsegs.push_back(MaLongSeg{sa, n, i, dir});  // (k_mafilter_long writes filters of static records only)
          own.scanblk_seg0.push_back((int32_t)own.scansegs.size());
          blk_fill = 0;
        }
      }
    }
    if (blk_fill > 0) own.scanblk_seg0.push_back((int32_t)own.scansegs.size());
    v.nscansegs = (int32_t)own.scansegs.size();
    v.nscanblk = (int32_t)own.scanblk_seg0.size() - 1;
    if (own.scansegs.empty()) own.scansegs.push_back(MaLongSeg{0, 0, 0, 0});
    v.scansegs = own.scansegs.data();
    v.scanblk_seg0 = own.scanblk_seg0.data();
    // the order in which a wave of k_matrans evaluates the entries of its block
    own.scanperm.assign((size_t)(m.nalltrans > 0 ? m.nalltrans : 1), 0);
    for (int b = 0; b < v.nscanblk; b++) {
      const MaLongSeg &s0 = own.scansegs[own.scanblk_seg0[b]], &s1 = own.scansegs[own.scanblk_seg0[b + 1] - 1];
      const int a0 = s0.ats0, nent = (s1.ats0 + s1.n) - a0;
      if (nent > MATRANS_BLOCK) continue;  // (a long segment: evaluated 64 consecutive entries at a time)
      std::vector<int> idx((size_t)nent);
      for (int e = 0; e < nent; e++) idx[e] = e;
      auto kind = [&](int e) {
        const int ati = a0 + e, ul = own.alltrans_owner[ati];
        const bool down = (ati - m.level_alltrans_startdown[ul]) < m.level_ndowntrans[ul];
        const int branch = (m.alltrans_coll_str[ati] < 0) ? (m.alltrans_forbidden[ati] ? 2 : 1) : 0;
        return (down ? 0 : 3) + branch;
      };
      std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return kind(x) < kind(y); });
      for (int e = 0; e < nent; e++) own.scanperm[a0 + e] = (uint8_t)idx[e];
    }
    v.scanperm = own.scanperm.data();
    v.nmalongsegs = (int32_t)own.malongsegs.size();
    if (own.malongsegs.empty()) own.malongsegs.push_back(MaLongSeg{0, 0, 0, 0});
    v.malongsegs = own.malongsegs.data();
  }
  v.level_upcum_start = own.level_upcum_start.data();
  own.cont_pack.resize(m.nbfcontinua);
  for (int i = 0; i < m.nbfcontinua; i++)
    own.cont_pack[i] = ContPack{m.allcont_nu_edge[i], m.allcont_probability[i],
                                m.level_phixsstart[m.allcont_uniquelevelindex[i]] * m.NPHIXSPOINTS,
                                m.allcont_groundcontestimindex[i], {0, 0}};
  v.cont_pack = own.cont_pack.data();
  // which lines calculate_expansion_opacities() (rpkt.cc:1083-1097) adds into which wavelength bin: the walk starts at the
  // first line with nu <= nu_upper(0) and bin b takes lines while nu >= nu_lower(b)
  own.expopac_linestart.assign(ARTIS_EXPOPAC_NBINS + 1, m.nlines);
  {
    int li = 0;
    const double nu0 = 1e8 * 2.99792458e+10 / ARTIS_EXPOPAC_LAMBDAMIN;
    while (li < m.nlines && m.line_nu[li] > nu0) li++;
    for (int b = 0; b < ARTIS_EXPOPAC_NBINS; b++) {
      own.expopac_linestart[b] = li;
      const double nu_lower = 1e8 * 2.99792458e+10 / (ARTIS_EXPOPAC_LAMBDAMIN + ((double)(b + 1) * ARTIS_EXPOPAC_DELTALAMBDA));
      while (li < m.nlines && m.line_nu[li] >= nu_lower) li++;
    }
    own.expopac_linestart[ARTIS_EXPOPAC_NBINS] = li;
  }
  v.expopac_linestart = own.expopac_linestart.data();
  own.upcum_coolslot.assign((size_t)(nupcum > 0 ? nupcum : 1), -1);
  for (int e = 0; e < m.nelements; e++)
    for (int ion = 0; ion < m.elem_nions[e]; ion++) {
      const int ui = m.elem_uniqueionindexstart[e] + ion;
      int k = ((m.elem_lowest_ionstage[e] + ion - 1) > 0) ? 1 : 0;  // the free-free entry comes first (kpkt.cc:75)
      for (int l = 0; l < m.ion_nlevels[ui]; l++) {
        const int ul = m.ion_uniquelevelindexstart[ui] + l;
        if (m.level_nuptrans[ul] > 0) own.upcum_coolslot[own.level_upcum_start[ul] + m.level_nuptrans[ul] - 1] = m.ion_coolingoffset[ui] + k++;
      }
    }
  v.upcum_coolslot = own.upcum_coolslot.data();
  // the lines of every level's collisional-excitation cooling filter (tables.h CoolLineRef): the level's entry of the cooling
  // list holds the running sum after its last upward transition, the entry before it (the free-free term or the previous level
  // with upward transitions; none for the first entry of an ion: the sum starts at 0) the sum before its first
  own.coollines.clear();
  own.level_coolhi.assign((size_t)(m.nlevels > 0 ? m.nlevels : 1), -1);
  for (int e = 0; e < m.nelements; e++)
    for (int ion = 0; ion < m.elem_nions[e]; ion++) {
      const int ui = m.elem_uniqueionindexstart[e] + ion;
      int k = ((m.elem_lowest_ionstage[e] + ion - 1) > 0) ? 1 : 0;
      for (int l = 0; l < m.ion_nlevels[ui]; l++) {
        const int ul = m.ion_uniquelevelindexstart[ui] + l;
        const int nup = m.level_nuptrans[ul];
        if (nup <= 0) continue;
        const LevelPack &lp = own.level_pack[ul];
        const int hi = m.ion_coolingoffset[ui] + k, lo = (k > 0) ? hi - 1 : -1;  // (lo = hi - 1 unless hi is the ion's first entry)
        own.level_coolhi[ul] = hi;
        if (lp.rec_off >= 0)  // (a cold level's filter is written with its record, on demand: physics.h ma_fill_record)
          for (int line = 0; line < marec_lines(nup); line++)
            own.coollines.push_back(CoolLineRef{lp.rec_off + marec_slot(MADIR_COOL, line, lp.ndown, lp.nup), own.level_upcum_start[ul],
                                                line * MAREC_PER, nup, hi, lo});
        k++;
      }
    }
  v.level_coolhi = own.level_coolhi.data();
  {
    // the cooling guides (tables.h "COOLING GUIDES"): as many ranges of the draw as the list has entries / 2, rounded up to a power of two
    auto shift_for = [](int n) {
      int lg = 0;
      while ((1 << lg) < n && lg < 16) lg++;
      return 24 - lg;
    };
    own.ion_guideoff.assign((size_t)(m.nions > 0 ? m.nions : 1), 0);
    own.ion_guideshift.assign((size_t)(m.nions > 0 ? m.nions : 1), 24);
    bool ok = m.nions > 0 && m.nions <= 65535;
    if (const char *b = std::getenv("ARTIS_AMD_COOLGUIDE")) ok = ok && std::atoi(b) != 0;
    v.guide_ion_shift = shift_for(2 * m.nions);
    int64_t n = ((int64_t)1 << (24 - v.guide_ion_shift)) + 1;
    for (int ui = 0; ui < m.nions && ok; ui++) {
      if (m.ion_ncoolingterms[ui] > 65535) ok = false;
      own.ion_guideoff[ui] = (int32_t)n;
      own.ion_guideshift[ui] = shift_for((m.ion_ncoolingterms[ui] + 1) / 2);
      n += ((int64_t)1 << (24 - own.ion_guideshift[ui])) + 1;
    }
    v.nguide = ok ? (int32_t)((n + 3) & ~(int64_t)3) : 0;  // (rows of whole 8 bytes)
    v.ion_guideoff = own.ion_guideoff.data();
    v.ion_guideshift = own.ion_guideshift.data();
  }
  v.ncoollines = (int32_t)own.coollines.size();
  if (own.coollines.empty()) own.coollines.push_back(CoolLineRef{0, 0, 0, 0, 0, -1});
  v.coollines = own.coollines.data();
  // recombination lists: for every level of ion i > 0, the levels of ion i-1 with a photoionisation target equal to it
  // (the first such target, like find_phixstargetindex atomic.h:493), lower level rising
  own.level_recomb_start.assign((size_t)m.nlevels + 1, 0);
  own.recomb_lower.clear();
  own.recomb_target.clear();
  for (int e = 0; e < m.nelements; e++)
    for (int ion = 0; ion < m.elem_nions[e]; ion++) {
      const int ui = m.elem_uniqueionindexstart[e] + ion;
      for (int l = 0; l < m.ion_nlevels[ui]; l++) {
        const int ul = m.ion_uniquelevelindexstart[ui] + l;
        own.level_recomb_start[ul] = (int32_t)own.recomb_lower.size();
        if (ion == 0) continue;
        const int ls = m.ion_uniquelevelindexstart[ui - 1];
        for (int lower = 0; lower < m.ion_nlevels_ionising[ui - 1]; lower++)
          for (int t = 0; t < m.level_nphixstargets[ls + lower]; t++)
            if (m.allphixstargets_levelindex[m.level_phixstargetstart[ls + lower] + t] == l) {
              own.recomb_lower.push_back(lower);
              own.recomb_target.push_back(t);
              break;
            }
      }
    }
  own.level_recomb_start[m.nlevels] = (int32_t)own.recomb_lower.size();
  v.nrecomb = (int32_t)own.recomb_lower.size();
  if (own.recomb_lower.empty()) { own.recomb_lower.push_back(0); own.recomb_target.push_back(0); }
  v.level_recomb_start = own.level_recomb_start.data();
  v.recomb_lower = own.recomb_lower.data();
  v.recomb_target = own.recomb_target.data();
  own.recomb_levels.clear();
  for (int ui = 0; ui < m.nions; ui++)
    for (int l = 0; l < m.ion_nlevels[ui]; l++) {
      const int ul = m.ion_uniquelevelindexstart[ui] + l;
      if (own.level_recomb_start[ul + 1] > own.level_recomb_start[ul] && l <= m.ion_maxrecombininglevel[ui] && own.level_pack[ul].rec_off >= 0)
        own.recomb_levels.push_back(ul);  // (k_macroatom_recomb: the levels with a static record; a cold level's sums are formed when it is filled)
    }
  v.nrecomblevels = (int32_t)own.recomb_levels.size();
  if (own.recomb_levels.empty()) own.recomb_levels.push_back(0);
  v.recomb_levels = own.recomb_levels.data();
  own.ion_cooltail_start.assign((size_t)(m.nions > 0 ? m.nions : 1), 0);
  for (int e = 0; e < m.nelements; e++)
    for (int ion = 0; ion < m.elem_nions[e]; ion++) {
      const int ui = m.elem_uniqueionindexstart[e] + ion;
      int k = ((m.elem_lowest_ionstage[e] + ion - 1) > 0) ? 1 : 0;  // the free-free entry (kpkt.cc:75)
      for (int l = 0; l < m.ion_nlevels[ui]; l++) k += (m.level_nuptrans[m.ion_uniquelevelindexstart[ui] + l] > 0) ? 1 : 0;
      own.ion_cooltail_start[ui] = k;
    }
  v.ion_cooltail_start = own.ion_cooltail_start.data();
#define ARTIS_COPY_PTR(f) v.f = m.f;
  ARTIS_COPY_PTR(elem_nions) ARTIS_COPY_PTR(elem_uniqueionindexstart) ARTIS_COPY_PTR(elem_lowest_ionstage)
  ARTIS_COPY_PTR(elem_anumber) ARTIS_COPY_PTR(elem_meannucmass) ARTIS_COPY_PTR(ion_nt_sum_q_over_binding)
  ARTIS_COPY_PTR(ion_element) ARTIS_COPY_PTR(ion_nlevels) ARTIS_COPY_PTR(ion_nlevels_ionising)
  ARTIS_COPY_PTR(ion_maxrecombininglevel) ARTIS_COPY_PTR(ion_uniquelevelindexstart) ARTIS_COPY_PTR(ion_coolingoffset)
  ARTIS_COPY_PTR(ion_ncoolingterms) ARTIS_COPY_PTR(level_epsilon) ARTIS_COPY_PTR(level_statweight)
  ARTIS_COPY_PTR(level_alltrans_startdown) ARTIS_COPY_PTR(level_ndowntrans) ARTIS_COPY_PTR(level_nuptrans)
  ARTIS_COPY_PTR(level_closestgroundlevelcont) ARTIS_COPY_PTR(level_phixsstart) ARTIS_COPY_PTR(level_nphixstargets)
  ARTIS_COPY_PTR(level_phixstargetstart) ARTIS_COPY_PTR(level_bflist_start) ARTIS_COPY_PTR(level_matransblock_start)
  ARTIS_COPY_PTR(alltrans_lineindex) ARTIS_COPY_PTR(alltrans_targetlevelindex) ARTIS_COPY_PTR(alltrans_einstein_A)
  ARTIS_COPY_PTR(alltrans_coll_str) ARTIS_COPY_PTR(alltrans_osc_strength) ARTIS_COPY_PTR(alltrans_forbidden)
  ARTIS_COPY_PTR(line_nu) ARTIS_COPY_PTR(line_elementindex) ARTIS_COPY_PTR(line_ionindex) ARTIS_COPY_PTR(allphixs)
  ARTIS_COPY_PTR(allphixstargets_levelindex) ARTIS_COPY_PTR(allphixstargets_probability) ARTIS_COPY_PTR(allcont_nu_edge)
  ARTIS_COPY_PTR(allcont_element) ARTIS_COPY_PTR(allcont_ion) ARTIS_COPY_PTR(allcont_level)
  ARTIS_COPY_PTR(allcont_phixstargetindex) ARTIS_COPY_PTR(allcont_upperlevel) ARTIS_COPY_PTR(allcont_uniquelevelindex)
  ARTIS_COPY_PTR(allcont_groundcontestimindex) ARTIS_COPY_PTR(allcont_probability) ARTIS_COPY_PTR(groundcont_nu_edge)
  ARTIS_COPY_PTR(spontrecombcoeffs) ARTIS_COPY_PTR(corrphotoioncoeffs) ARTIS_COPY_PTR(bfcooling_coeffs)
  ARTIS_COPY_PTR(coolinglist_type) ARTIS_COPY_PTR(coolinglist_level) ARTIS_COPY_PTR(coolinglist_phixstargetindex)
  ARTIS_COPY_PTR(propcell_nonemptymgi)
#undef ARTIS_COPY_PTR
  v.gridtype = m.gridtype;
  int stride = 1;
  for (int a = 0; a < 3; a++) {
    v.ncoordgrid[a] = m.ncoordgrid[a];
    v.coordstride[a] = stride;  // get_coordcellindexstride grid.cc:200
    stride *= m.ncoordgrid[a];
    v.coord_pos_min_tmin[a] = m.coord_pos_min_tmin[a];
  }
  v.ngrid = m.ngrid;
  v.npts_nonempty = m.npts_nonempty;
  v.tmin = m.tmin; v.vmax = m.vmax; v.rmax = m.rmax;
  v.ejecta_kinetic_energy = m.ejecta_kinetic_energy; v.mtot_input = m.mtot_input;
  v.allcont_bfestimindex = m.allcont_bfestimindex;
  v.rho_tmin = m.rho_tmin;
  v.xcom_elem_start = m.xcom_elem_start; v.xcom_energy = m.xcom_energy; v.xcom_sigma = m.xcom_sigma;
  v.nxcom = m.xcom_elem_start ? m.xcom_elem_start[m.nelements] : 0;
  v.detailed_lineindices = m.detailed_lineindices;
  v.detailed_linecount = m.detailed_lineindices ? m.detailed_linecount : 0;
  v.nbfestim = (m.allcont_bfestimindex && m.nbfestim > 0) ? m.nbfestim : m.nbfcontinua;
  v.vpkt = nullptr;  // the owner of the view places the block (make_vpkt_config)
  return v;
}

// The virtual-packet configuration of a VPKT_ON build as one block (tables.h VpktConfig): the observer unit vectors as
// trace_vpkts() forms them (vpkt.cc:967-971), the bin widths of init_vspecpol() (vpkt.cc:491-512; floats there). False when
// the model carries no usable configuration.
// get_loggrid_edge sn3d.h:142: lower edge of bin `index` of a grid spaced uniformly in the log of the value
inline double loggrid_edge(double minvalue, double dlog, double index) { return std::exp(std::log(minvalue) + (index * dlog)); }
inline bool make_vpkt_config(const artis_model &m, VpktConfig &V) {
  std::memset(&V, 0, sizeof(V));
  if (m.vpkt_nobsdirections < 1 || m.vpkt_nobsdirections > VPKT_MAXOBS || !m.vpkt_obsdirs_costheta || !m.vpkt_obsdirs_phi ||
      m.vpkt_nspectraperobsdir < 1 || m.vpkt_nspectraperobsdir > VPKT_MAXSPEC || !m.vpkt_opacityexclusions ||
      m.vpkt_nwavelengthranges < 1 || m.vpkt_nwavelengthranges > VPKT_MAXRANGES || !m.vpkt_numin_input || !m.vpkt_numax_input ||
      m.vpkt_nprocs < 1)
    return false;
  if (m.vpkt_vgrid_on && (m.vpkt_grid_nwavelengthranges < 1 || m.vpkt_grid_nwavelengthranges > VPKT_MAXRANGES || !m.vpkt_nu_grid_min ||
                          !m.vpkt_nu_grid_max))
    return false;
  V.nobsdirections = m.vpkt_nobsdirections;
  V.nspectraperobsdir = m.vpkt_nspectraperobsdir;
  V.nwavelengthranges = m.vpkt_nwavelengthranges;
  V.vgrid_on = m.vpkt_vgrid_on ? 1 : 0;
  V.grid_nwavelengthranges = m.vpkt_vgrid_on ? m.vpkt_grid_nwavelengthranges : 0;
  V.nprocs = m.vpkt_nprocs;
  for (int i = 0; i < V.nobsdirections; i++) {
    const double ct = m.vpkt_obsdirs_costheta[i], ph = m.vpkt_obsdirs_phi[i];
    V.obsdir[i][0] = std::sqrt(1 - (ct * ct)) * std::cos(ph);
    V.obsdir[i][1] = std::sqrt(1 - (ct * ct)) * std::sin(ph);
    V.obsdir[i][2] = ct;
  }
  for (int i = 0; i < V.nspectraperobsdir; i++) V.opacityexclusions[i] = m.vpkt_opacityexclusions[i];
  V.timemin_input = m.vpkt_timemin_input;
  V.timemax_input = m.vpkt_timemax_input;
  V.tau_max = m.vpkt_tau_max;
  V.tmin_grid = m.vpkt_tmin_grid;
  V.tmax_grid = m.vpkt_tmax_grid;
  for (int i = 0; i < V.nwavelengthranges; i++) {
    V.numin_input[i] = m.vpkt_numin_input[i];
    V.numax_input[i] = m.vpkt_numax_input[i];
  }
  for (int i = 0; i < V.grid_nwavelengthranges; i++) {
    V.nu_grid_min[i] = m.vpkt_nu_grid_min[i];
    V.nu_grid_max[i] = m.vpkt_nu_grid_max[i];
  }
  const double dlogt = (std::log(ARTIS_VSPEC_TIMEMAX) - std::log(ARTIS_VSPEC_TIMEMIN)) / ARTIS_VSPEC_TIMEBINS;
  const double dlognu = (std::log(ARTIS_VSPEC_NUMAX) - std::log(ARTIS_VSPEC_NUMIN)) / ARTIS_VSPEC_NUBINS;
  const auto edge = loggrid_edge;
  for (int n = 0; n < ARTIS_VSPEC_TIMEBINS; n++) {
    const float lower = (float)edge(ARTIS_VSPEC_TIMEMIN, dlogt, n);
    V.delta_t[n] = (float)(edge(ARTIS_VSPEC_TIMEMIN, dlogt, n + 1) - lower);
  }
  for (int n = 0; n < ARTIS_VSPEC_NUBINS; n++) {
    const float lower = (float)edge(ARTIS_VSPEC_NUMIN, dlognu, n);
    V.delta_freq[n] = (float)(edge(ARTIS_VSPEC_NUMIN, dlognu, n + 1) - lower);
  }
  return true;
}

inline DevCells make_host_cells_view(const artis_cellstate &c) {
  DevCells v;
  v.rho = c.rho; v.Te = c.Te; v.TJ = c.TJ; v.TR = c.TR; v.W = c.W; v.nne = c.nne; v.nnetot = c.nnetot;
  v.kappagrey = c.kappagrey; v.clumpfactor = c.clumpfactor; v.thick = c.thick;
  v.ion_groundlevelpops = c.ion_groundlevelpops; v.ion_partfuncts = c.ion_partfuncts; v.elem_massfracs = c.elem_massfracs;
  v.corrphotoionrenorm = c.corrphotoionrenorm;
  v.ffegrp = c.ffegrp;
  v.levelpops = c.levelpops;
  v.corrphotoioncoeff = c.corrphotoioncoeff;
  v.radfieldbin_W = c.radfieldbin_W;
  v.radfieldbin_T_R = c.radfieldbin_T_R;
  v.nt_frac_ionisation = c.nt_frac_ionisation; v.nt_frac_excitation = c.nt_frac_excitation;
  v.nt_deposition_rate_density = c.nt_deposition_rate_density; v.nt_eff_ionpot = c.nt_eff_ionpot;
  v.nt_prob_num_auger = c.nt_prob_num_auger; v.nt_ionenfrac_num_auger = c.nt_ionenfrac_num_auger;
  v.nt_exc_count = c.nt_exc_count; v.nt_exc_frac_deposition = c.nt_exc_frac_deposition;
  v.nt_exc_ratecoeffperdeposition = c.nt_exc_ratecoeffperdeposition; v.nt_exc_alltransindex = c.nt_exc_alltransindex;
  v.nt_excitations_stored = c.nt_excitations_stored;
  v.nt_ionratecoeff = nullptr; v.nt_ionenrate_cum = nullptr;
  v.expansionopacities = c.expansionopacities;
  v.expansionopacity_planck_cumulative = c.expansionopacity_planck_cumulative;
  v.Jb_lu_normed = c.Jb_lu_normed;
  v.elem_meanweight = c.elem_meanweight;
  return v;
}

inline DevStep make_step(const artis_timestep &t) {
  DevStep s;
  s.nts = t.nts; s.start = t.start; s.width = t.width; s.mid = t.mid; s.max_path_step = t.max_path_step;
  s.ts_end = t.start + t.width;  // update_packets.cc:536
  return s;
}

// Carve the three record arrays of n packets out of one allocation of pkt_store_bytes(n) bytes (base aligned to 128 B).
inline size_t pkt_store_bytes(int64_t n) { return PKT_BYTES_PER_PACKET * (size_t)(n > 0 ? n : 1); }
inline PktStore carve_pkt_store(void *base, int64_t n) {
  const size_t m = (size_t)(n > 0 ? n : 1);
  PktStore P;
  P.hot = (PktHot *)base;
  P.flight = (PktFlight *)((char *)base + sizeof(PktHot) * m);
  P.cold = (PktCold *)((char *)base + (sizeof(PktHot) + sizeof(PktFlight)) * m);
  P.n = n;
  return P;
}

}  // namespace artis
