// tables.h -- plain-old-data views of everything the packet kernels read or write.
//
// One struct per lifetime:
//   DevModel  static atomic data + grid (uploaded once per run)
//   DevCells  per-timestep cell state (the reference's update_grid() output)
//   DevCache  per-timestep cell cache (the reference's globals::cellcache, multi-slot form,
//             globals.h:283 / update_packets.cc:397), one row per non-empty cell
//   PktStore  the packet population in HBM (three arrays of cache-line records)
//   DevEst    estimator accumulators
// The same structs are used by the host-emulation test build (tests/hostemu), where the
// pointers are host pointers.
#pragma once
#include <stdint.h>

#include "../../include/artis_amd.h"
#include "../../include/artis_options.h"

namespace artis {

// 16-byte record walked by get_possible_event(): the two level indices and the two Einstein B
// coefficients of a line (globals::linelist uniquelevelindex_lower/upper, B_ul, B_lu globals.h:234-237)
struct alignas(16) LinePack {
  int32_t lower;
  int32_t upper;
  float B_ul;
  float B_lu;
};

// 16-byte record of the static per-level indices a macro-atom transition needs (AllLevels, globals.h:181):
// alltrans_startdown, ndowntrans, nuptrans, and the offset (in 16-byte slots) of the level's macro-atom record inside a
// cell's macache row (see DevCache::macache)
struct alignas(16) LevelPack {
  int32_t rec_off;
  int32_t alltrans_startdown;
  int32_t ndown;
  int32_t nup;
};
// 32-byte record of the static data of one bound-free continuum that calculate_chi_bf_gammacontr() needs
// (globals::allcont nu_edge, probability, groundcontestimindex globals.h:253-263; xs_off = offset of the lower level's
// photoionisation table in allphixs, i.e. level_phixsstart * NPHIXSPOINTS)
struct alignas(32) ContPack {
  double nu_edge;
  double probability;
  int32_t xs_off;
  int32_t gi;
  int32_t pad[2];
};
// Macro-atom record of one (cell, level): FILTERS ONLY (round 4). The decisions of a macro-atom transition (macroatom.cc:385-577)
// are "how many cumulative values are <= z * whole" with z a 24-bit draw; a record holds the cumulative values as 15-bit
// fractions of their whole (below, "FILTERS") and the nine process rates as doubles, nothing else: no cumulative sums, no
// transition targets. Rounds 2-3 kept both per cell (7 sums + 7 targets per 128-byte line: 1.08 of the 1.55 MB of a row with
// the 13 619-line atomic data, 6.3 of 8.75 MB with 110 860 lines); now a draw that the filter cannot decide (5e-4 per decision)
// RECOMPUTES the sums it needs from the rate coefficients, term by term in the reference's order (physics.h ma_exact_search:
// the same matrans_terms() that fills the records, so the same bits), and the targets -- static data -- sit in one table for all
// cells (DevModel::alltrans_target). A level's record is 12 slots of 16 bytes with ~9 transitions per direction (192 B instead
// of 690), 18 with 25 (288 B instead of 1415). The unit is the 16-byte SLOT (one load instruction):
//   slot 0            action filter: 8 x uint16, the cumulative rates of actions 0..7 as fractions of the total of all nine
//   slot 1, 2, 3      the FIRST line of the internal-down-same, internal-up-same and radiative-deexcitation filters: 7 x uint16
//                     fractions of the direction's whole rate + the line's "usable" mark (87.7 % of the searches end in the first
//                     two entries: profiles/r03/ti_hist.txt) -- with slot 0 one aligned 64-byte block: a transition reads ONE
//                     cell-specific sector
//   then              the further lines of the three directions (7 transitions per slot), the lines of the level's
//                     collisional-excitation cooling filter (kpkt.cc:461-476: fractions of the level's span of the ion's cooling
//                     list), and 5 slots with the 9 process rates (alllevels_maprocessrates, globals.h:286) as doubles, read only
//                     when a filter cannot decide and by the slow-path actions
// Records are aligned to 4 slots (64 B). Entries of a line beyond its direction's searched sums hold 0x7FFF (never counted);
// those slots are initialised once (k_mainit), the population writes the entries of real transitions and each line's mark.
//
// FILTERS. k_thermal is bound by the NUMBER of vector-memory instructions it issues (two more 8-byte reads of a line it
// has already read, per transition: 617 -> 786 ms). Both decisions are "how many cumulative values are
// <= z * whole" with z uniform in [0, 1): the same as "how many fractions value / whole are <= z" unless z lies within rounding
// of a fraction. The fractions are kept as 15-bit integers q = floor(fraction * 32768) (clamped to 32767) in uint16, so
// q <= fraction * 32768 <= q + 1: with zi = floor(z * 32768) (the top 15 bits of the 24-bit draw), zi >= q + 2 proves
// value <= z * whole and zi <= q - 1 proves the opposite, by margins of 3e-5 and 6e-8 of the whole against f64 rounding errors
// of 1e-16 (physics.h mafilt_count). Anything in between (q == zi or zi - 1: 5e-4 of the draws per decision) is decided on
// f64 values -- same random numbers, same result. (Round 5: the draw's nine bits below zi also settle q == zi - 1 unless they are 0 or 1:
// half as many undecided draws, physics.h mafilt_count.) 15 bits, so that two entries are compared by ONE 32-bit subtraction.
// What a transition needs to know about the level it leads to (static, one 16-byte load from DevModel::alltrans_target): where
// the target's record is in a cell's row (slots), its first entry in alltrans, its index within the ion, its transition counts.
struct alignas(16) MaTarget {
  int32_t rec, ats, level;
  uint32_t ndnu;  // ndown | nup << 16
};
constexpr int MATGT_MAX_NTRANS = 1 << 16;
static_assert(sizeof(MaTarget) == 16, "transition target size");
struct alignas(16) U4 {  // one slot: 8 x uint16 of a macro-atom filter
  uint32_t w[4];
};
constexpr int MAREC_ALIGN = 4;   // slots: records start on 64-byte boundaries
// The pool of on-demand records is handed out in units of 128 bytes = one line of a compute unit's vector L1: a record filled by one wave
// while the same kernel's waves on other compute units read their own records (the tail kernel) must not share a line with them -- the L1s
// are not coherent within a kernel, and a line fetched for a neighbouring record before this one was written would be read stale (found
// as one packet in 30 000 of a tiled run taking another history, round 5).
constexpr int MAPOOL_UNIT = 8;   // slots
constexpr int MAREC_PER = 7;     // transitions per filter line
constexpr int MAREC_QUAD = 4;    // slots 0..3: action filter, first lines of the down / up / rad filters
constexpr int MAREC_RATE_SLOTS = 5;  // 9 doubles (+ one spare)
enum { MADIR_DOWN = 0, MADIR_UP = 1, MADIR_RAD = 2, MADIR_COOL = 3 };
constexpr int marec_lines(int n) { return (n + MAREC_PER - 1) / MAREC_PER; }
constexpr int marec_rest(int n) { return n > MAREC_PER ? marec_lines(n) - 1 : 0; }  // a direction's lines after its first
// slot of line l of direction d in the record of a level with nd downward and nu upward transitions
constexpr int marec_slot(int d, int l, int nd, int nu) {
  return (d < MADIR_COOL && l == 0)
             ? 1 + d
             : (d == MADIR_DOWN ? MAREC_QUAD + (l - 1)
                                : (d == MADIR_UP ? MAREC_QUAD + marec_rest(nd) + (l - 1)
                                                 : (d == MADIR_RAD ? MAREC_QUAD + marec_rest(nd) + marec_rest(nu) + (l - 1)
                                                                   : MAREC_QUAD + (2 * marec_rest(nd)) + marec_rest(nu) + l)));
}
constexpr int marec_rates_slot(int nd, int nu) { return MAREC_QUAD + (2 * marec_rest(nd)) + marec_rest(nu) + marec_lines(nu); }
// FINE BYTES (round 6). The 15-bit entries leave ~n / 32768 of the draws of a direction of n transitions undecided, and each such draw re-adds
// the direction's sums from its first transition (n rate coefficients: the cost grows with n^2) in the slow-path kernel, a launch later: with
// 4e5 lines 83 % of the slow path's 3.7e8 visits per step (profiles/r05/slow_path_cd23like.txt). Behind the rates every line of the internal-down
// and internal-up filters therefore has 8 more bytes: entry j's byte holds the next 8 bits of its fraction, q23 = floor(fraction * 2^23) =
// (q15 << 8) | byte (clamped to 2^23 - 1; an entry that is never counted: 0x7FFF | 0xFF). They are read only when the 15-bit entries cannot
// decide (physics.h mafilt_count_fine): with the 24-bit draw u = z * 2^24, u >= 2 q23 + 3 proves fraction < z by 2^-24 of the whole and
// u <= 2 q23 - 1 proves the opposite by the same -- the margins (6e-8) the 15-bit rule already relies on, against f64 rounding of 1e-16 -- so three
// draws of 2^24 per entry are left to the re-added sums instead of ~1.5 of 32768: 170x fewer. Two lines' bytes per slot, downward lines first.
constexpr int marec_fine_dirslots(int n) { return (marec_lines(n) + 1) / 2; }
constexpr int marec_fine_slot0(int nd, int nu) { return marec_rates_slot(nd, nu) + MAREC_RATE_SLOTS; }
// byte offset (from the record's start) of the 8 fine bytes of line l of direction d (MADIR_DOWN / MADIR_UP only)
constexpr int marec_fine_byte0(int d, int l, int nd, int nu) {
  return ((marec_fine_slot0(nd, nu) + (d == MADIR_UP ? marec_fine_dirslots(nd) : 0)) * 16) + (l * 8);
}
constexpr int marec_slots(int nd, int nu) { return marec_fine_slot0(nd, nu) + marec_fine_dirslots(nd) + marec_fine_dirslots(nu); }
constexpr double MAFILT_SCALE = 32768.;
constexpr uint32_t MAFILT_NONE = 0x7FFFu;  // an entry that is never counted
constexpr int MATRANS_BLOCK = 256;  // entries of alltrans a wave of k_matrans holds in LDS at a time
constexpr int MAREC_SLACK = 16;  // elements past the last row of every cache array (padded reads stay inside the allocation)

struct alignas(16) D2 {
  double x, y;
};
// a (level, direction) with transitions (static): the segments k_matrans sums (DevModel::scansegs); those longer than one of its
// blocks (real atomic data: levels with hundreds of transitions) are listed again in DevModel::malongsegs, for k_mafilter_long
struct alignas(16) MaLongSeg {
  int32_t ats0, n, ul, dir;  // first entry in alltrans, transitions, level, 0 = downward / 1 = upward
};
// one line of a level's collisional-excitation cooling filter (static): its slot in a cell's row, the level's first entry of
// the cell's upward-transition terms, the line's first transition, the level's nup, and the entries of the cooling list
// that hold the running sum after (hi) and before (lo; -1: the sum starts at 0) the level (k_collexc_filter)
struct alignas(8) CoolLineRef {
  int32_t slot, up0, first, n, cool_hi, cool_lo;
};

struct VpktConfig;  // virtual packets, below
struct DevModel {
  int32_t nelements, nions, nlevels, nlines, nalltrans, nphixstargets_total, nphixslevels, nbfcontinua, nbfcontinua_ground,
      ncoolingterms, nmatransblock, NPHIXSPOINTS;
  int32_t nmacache;    // 16-byte slots per cell in DevCache::macache
  // ON-DEMAND RECORDS (round 5; the reference fills a level's rates when a packet first reaches it, macroatom.cc:398-417). A step visits
  // 13-15 % of the (cell, level) records (profiles/r05/visit_sparsity_*.md), and 98-99 % of its transitions are drawn in the lowest third of
  // every ion's levels. With ncold > 0 a cell's row holds static records for those HOT levels only (LevelPack::rec_off >= 0); a COLD level
  // (rec_off = -(cold index) - 1) gets a record in the POOL -- one for all resident cells: DevCache::ma_pool, ma_pool_slots slots per resident
  // cell on average, handed out in units of 128 bytes (MAPOOL_UNIT) -- when a packet first reaches it in a cell: the slow-path kernel fills it with the
  // sequential forms of the population (the same terms added in the same order: the same bits), and DevCache::ma_rowtab[cell][cold index]
  // says where it is. A pool that is used up is emptied by the host before the slow-path list's next launch (Env::ma_pool_full; every cold level
  // is then without a record again and is filled when next needed, as after a tile's refill): it costs fills, never an answer.
  // ncold == 0: every level has a static record (the default whenever the whole cache fits one tile).
  int32_t ncold, ma_pool_slots;
  // [nlevels] the entry of the cooling list that holds the running sum after the level's collisional excitations (-1: no upward transitions)
  const int32_t *level_coolhi;
  // COOLING GUIDES (round 5). A k-packet step draws an ion from the cell's cumulative list of the ions' cooling and then a term from that ion's
  // cumulative list (kpkt.cc:430-447): two bisections, 6 + 8 DEPENDENT reads with the bench's data, each waiting for the one before. Both
  // draws are 24-bit integers u; DevCache::cool_guide holds, for the 2^(24 - shift) equal ranges of u, the answer for the range's first draw:
  // g[k] = upper_bound(list, value of the draw k << shift). Rounding is monotonic, so the answer for any u of range k lies in [g[k], g[k + 1]]:
  // one read of two neighbouring guide entries -- issued beside the read of the list's total, not after it -- and, only where the two differ
  // (5 % of the draws: most ranges lie inside one dominant term), the comparison with those few sums. The same index as the bisection gives.
  // Row of a cell: [the ions' guide: 2^(24 - guide_ion_shift) + 1 entries | ion 0's guide | ...]; ion ui's guide starts at ion_guideoff[ui]
  // and has 2^(24 - ion_guideshift[ui]) + 1 entries (as many ranges as half the ion's terms, rounded up to a power of two); the last entry
  // of a guide is the list's length. nguide == 0: no guides (a list longer than 65535 entries, or ARTIS_AMD_COOLGUIDE=0): bisection.
  int32_t nguide, guide_ion_shift;
  const int32_t *ion_guideoff, *ion_guideshift;
  int32_t nupcum;      // upward transitions (= nlines): doubles per cell of the population's scratch of cooling terms (Env::collexc_terms)
  int32_t ndpop;       // doubles per cell in DevCache::line_dpop: nlines, or 0 when the population factors are formed on the fly
  int32_t nkeepwords;  // ceil(nbfcontinua/64) (get_allcont_keepwordcount globals.h:401) rounded up to a multiple of 4
  double NPHIXSNUINCREMENT;
  double last_phixs_nuovernuedge;  // input.cc:310
  double T_step_log;               // ratecoeff.cc:39
  const double *temperature_grid;  // [TABLESIZE+1] ratecoeff.cc:41

  const int32_t *elem_nions, *elem_uniqueionindexstart, *elem_lowest_ionstage, *elem_anumber;
  const float *elem_meannucmass;             // optional (NT_ON builds)
  const double *ion_nt_sum_q_over_binding;   // optional (NT_ON builds)
  double ejecta_kinetic_energy, mtot_input;  // Barnes thermalisation scheme only
  // [ARTIS_EXPOPAC_NBINS + 1] first line of each wavelength bin of the expansion opacities (lines in falling frequency:
  // bin b holds the lines [start[b], start[b+1])); derived on the host, model_build.h
  const int32_t *expopac_linestart;
  // [nupcum] for the LAST upward transition of a level: the entry of the cell's cooling list that holds the running
  // sum after that level (calculate_cooling_rates_ion kpkt.cc:108-121); -1 for every other transition. Static.
  const int32_t *upcum_coolslot;
  // recombination list of every level (static): the (level of the ion below, photoionisation target) pairs that ionise
  // INTO this level, in rising lower level -- what the loops of macroatom.cc:147-168 find with find_phixstargetindex()
  // for every level of the lower ion. Entries [level_recomb_start[ul], level_recomb_start[ul+1]).
  const int32_t *level_recomb_start;   // [nlevels + 1]
  const int32_t *recomb_lower;         // [nrecomb] level index within the lower ion
  const int32_t *recomb_target;        // [nrecomb] its phixstargetindex
  int32_t nrecomb;
  // the levels whose recombination list is summed at all (non-empty list, level <= ion_maxrecombininglevel): k_macroatom_recomb
  const int32_t *recomb_levels;        // [nrecomblevels] unique level index
  int32_t nrecomblevels;
  // first entry of an ion's part of the cooling list that calculate_cooling_rates_ion() writes after the collisional excitations
  // (the free-free entry and one entry per level with upward transitions come before it; kpkt.cc:122-190). Static.
  const int32_t *ion_cooltail_start;   // [nions]
  // detailed bound-free estimators: estimator index of every continuum (-1: none) or null = identity; their number
  const int32_t *allcont_bfestimindex;
  int32_t nbfestim;
  const float *rho_tmin;  // [npts_nonempty] optional: column densities of the Wollaeger / Guttman gamma-ray schemes
  // optional (USE_XCOM_GAMMAPHOTOION builds): XCOM photoionisation points of every element of the model
  const int32_t *xcom_elem_start;  // [nelements + 1]
  const double *xcom_energy, *xcom_sigma;
  int32_t nxcom;
  // optional (DETAILED_LINE_ESTIMATORS_ON builds): the lines with their own intensity estimator, rising line index
  const int32_t *detailed_lineindices;
  int32_t detailed_linecount;
  const VpktConfig *vpkt;  // builds with VPKT_ON (else null)
  const int32_t *ion_element, *ion_nlevels, *ion_nlevels_ionising, *ion_maxrecombininglevel, *ion_uniquelevelindexstart,
      *ion_coolingoffset, *ion_ncoolingterms;
  const double *level_epsilon;
  const float *level_statweight;
  const int32_t *level_alltrans_startdown, *level_ndowntrans, *level_nuptrans, *level_closestgroundlevelcont, *level_phixsstart,
      *level_nphixstargets, *level_phixstargetstart, *level_bflist_start, *level_matransblock_start;
  const int32_t *level_ion;  // derived: uniqueionindex of each level
  const LevelPack *level_pack;  // derived
  const int32_t *level_upcum_start;  // derived: offset of the level's upward transitions in a cell's row of Env::collexc_terms
  const int32_t *alltrans_lineindex, *alltrans_targetlevelindex;
  const int32_t *alltrans_owner;  // derived: the level whose block of alltrans an entry belongs to
  // derived: the work of k_matrans. scansegs: every (level, direction) with transitions, in alltrans order; scanblk_seg0: [nscanblk + 1]
  // first segment of each BLOCK, a run of whole segments of at most MATRANS_BLOCK entries together (a longer segment is a block of its
  // own): one wave per (cell, block). malongsegs: the segments longer than a block, whose filters k_mafilter_long writes.
  const MaLongSeg *scansegs;
  const int32_t *scanblk_seg0;
  int32_t nscansegs, nscanblk;
  // [nalltrans] per block of k_matrans: the block's entries (offsets from its first) ordered by kind -- downward / upward, and the
  // branch of the collisional rate coefficient they take (macroatom.cc:708-792) -- so that the 64 entries a wave evaluates together run
  // the same code (round 4: with the entries in alltrans order every wave held both directions and ran both)
  const uint8_t *scanperm;
  const MaLongSeg *malongsegs;
  int32_t nmalongsegs;
  const MaTarget *alltrans_target;  // derived: [nalltrans] what a transition needs to know of the level it leads to
  // derived: [nalltrans] the target level alone (index within the ion), 2 bytes: with level_pack the same information in two
  // small static tables that fit the LDS of a compute unit for atomic data of the bench's size (k_thermal<.., true>)
  const uint16_t *alltrans_tlevel16;
  const CoolLineRef *coollines;     // derived: every line of every level's cooling filter, [ncoollines]
  int32_t ncoollines;
  const float *alltrans_einstein_A, *alltrans_coll_str, *alltrans_osc_strength;
  const uint8_t *alltrans_forbidden;
  const double *line_nu;
  const LinePack *line_pack;
  const ContPack *cont_pack;  // derived
  const int32_t *line_elementindex, *line_ionindex;
  const float *allphixs;
  const int32_t *allphixstargets_levelindex;
  const double *allphixstargets_probability;
  const double *allcont_nu_edge;
  const int32_t *allcont_element, *allcont_ion, *allcont_level, *allcont_phixstargetindex, *allcont_upperlevel,
      *allcont_uniquelevelindex, *allcont_groundcontestimindex;
  const double *allcont_probability;
  const double *groundcont_nu_edge;
  const double *spontrecombcoeffs, *corrphotoioncoeffs, *bfcooling_coeffs;
  const uint8_t *coolinglist_type;
  const int32_t *coolinglist_level, *coolinglist_phixstargetindex;

  int32_t gridtype, ncoordgrid[3], coordstride[3], ngrid, npts_nonempty;
  double tmin, vmax, rmax;
  const double *coord_pos_min_tmin[3];
  const int32_t *propcell_nonemptymgi;
};

struct DevCells {
  const float *rho, *Te, *TJ, *TR, *W, *nne, *nnetot, *kappagrey, *clumpfactor;
  const int32_t *thick;
  const float *ion_groundlevelpops, *ion_partfuncts, *elem_massfracs;
  const double *corrphotoionrenorm;
  const float *ffegrp;  // may be null on the host view (then the engine uploads zeros)
  // optional / option-dependent (include/artis_amd.h artis_cellstate): null when not handed over
  const double *levelpops;           // [cell][nlevels] host level populations
  const double *corrphotoioncoeff;   // [cell][nphixstargets_total] host photoionisation coefficients (USE_LUT_PHOTOION off)
  const float *radfieldbin_W, *radfieldbin_T_R;  // [cell][RADFIELDBINCOUNT] multibin radiation field
  // Spencer-Fano solution of the host (NT_ON builds; include/artis_amd.h artis_cellstate)
  const float *nt_frac_ionisation, *nt_frac_excitation;          // [cell]
  const double *nt_deposition_rate_density;                      // [cell]
  const float *nt_eff_ionpot;                                    // [cell][nions]
  const float *nt_prob_num_auger, *nt_ionenfrac_num_auger;       // [cell][nions][NT_MAX_AUGER_ELECTRONS + 1]
  const int32_t *nt_exc_count;                                   // [cell]
  const double *nt_exc_frac_deposition, *nt_exc_ratecoeffperdeposition;  // [cell][nt_excitations_stored]
  const int32_t *nt_exc_alltransindex;                           // [cell][nt_excitations_stored], ascending
  int32_t nt_excitations_stored;
  // derived once per cell state (populate_nt_cell): nt_ionisation_ratecoeff() of every ion and the running sum of
  // ion_ntion_energyrate() in select_nt_ionisation()'s order. Not part of the tiled cell cache: a deposit reads them in
  // whichever cell the particle stops.
  double *nt_ionratecoeff, *nt_ionenrate_cum;                    // [cell][nions]
  // [cell][ARTIS_EXPOPAC_NBINS] binned line opacity and its Planck-weighted running integral (expansion-opacity builds)
  const float *expansionopacities;
  const double *expansionopacity_planck_cumulative;
  const double *Jb_lu_normed;  // [cell][detailed_linecount] normalised line intensities of the previous timestep
  const float *elem_meanweight;  // [cell][nelements] mean atomic weights (USE_CALCULATED_MEANATOMICWEIGHT builds), else null
};

// (every array [row][...]: the row of a cell is the cell while the whole cache is resident, else Env::krow_tab[cell] -- physics.h krow())
struct DevCache {
  double *levelpops;             // [cell][nlevels]
  U4 *macache;                   // [cell][nmacache]: one record of filters + process rates per level (above; LevelPack::rec_off)
  // [cell][ncold] where a cold level's record is in the pool (units of MAPOOL_UNIT slots): -1 none yet, <= -3 being filled at unit -(v + 3), >= 0 ready;
  // the pool ([resident cells x ma_pool_slots] slots, shared by all of them: not indexed by row) and the units handed out
  int32_t *ma_rowtab;
  U4 *ma_pool;
  uint32_t *ma_pool_used;
  double *allcont_nnlevel;       // [cell][nbfcontinua]
  double *allcont_departure;     // [cell][nbfcontinua]
  double *allcont_edgepart;      // [cell][nbfcontinua]
  D2 *allcont_pair;              // [cell][nbfcontinua] {nnlevel, edgepart} as one 16-byte read for calculate_chi_bf_gammacontr
  uint64_t *allcont_keepbits;    // [cell][nkeepwords]
  // the kept continua of the cell as a list (rising index), per bitmap word the number of kept continua below it, and
  // allcont_pair in the order of the list: the kept continua of a window [begin, end) are the places [r0, r1) of the list
  // (two words, two counts), and what the opacity sum and k_bfest_dense read of them is contiguous
  int32_t *allcont_keptlist;     // [cell][nbfcontinua]
  int32_t *allcont_keepprefix;   // [cell][nkeepwords]
  D2 *allcont_keptpair;          // [cell][nbfcontinua]
  double *line_dpop;             // [cell][ndpop]: B_lu n_l - B_ul n_u of every line, the population factor of get_tau_sobolev() (rpkt.cc:75); null when ndpop == 0
  double *corrphotoioncoeff;     // [cell][nphixstargets_total]
  // [cell][nphixstargets_total] the other coefficients of each bound-free pair (populate_corrphotoion): radiative and
  // collisional recombination, collisional ionisation, bound-free cooling
  double *bf_radrecomb, *bf_colrecomb, *bf_colion, *bf_cooling;
  double *cooling_contrib;       // [cell][ncoolingterms]
  double *ion_cooling_contribs;  // [cell][nions]
  uint16_t *cool_guide;          // [cell][nguide] (DevModel "COOLING GUIDES")
  double *ion_cooling_C;         // [cell][nions] per-ion totals before the prefix sum
  double *chi_ff_nnionpart;      // [cell]
};

// One deferred update of the detailed bound-free estimators (physics.h update_bfestimators): the opacity's frequency, the
// weight distance * e_cmf / nu_cmf, the cell and the window of continua [begin, end) still in range at the packet's
// frequency. k_rpkt records them; k_bfest_dense adds the contributions with a whole wave per record.
struct alignas(16) BfEvent {
  double nu, w;
  int32_t c, begin, end, pad;
};

struct DevStep {
  int32_t nts;
  double start, width, mid, max_path_step, ts_end;
};

// Virtual packets (builds with VPKT_ON). What read_vpktparameterfile() (vpkt.cc:673) leaves, as one small block in HBM;
// the observer directions as the unit vectors trace_vpkts() forms from (costheta, phi) (vpkt.cc:967), made once on the host.
constexpr int VPKT_MAXOBS = 16, VPKT_MAXSPEC = 16, VPKT_MAXRANGES = 16;
struct VpktConfig {
  int32_t nobsdirections, nspectraperobsdir, nwavelengthranges, vgrid_on, grid_nwavelengthranges, nprocs;
  double obsdir[VPKT_MAXOBS][3];
  int32_t opacityexclusions[VPKT_MAXSPEC];
  double timemin_input, timemax_input, tau_max, tmin_grid, tmax_grid;
  double numin_input[VPKT_MAXRANGES], numax_input[VPKT_MAXRANGES], nu_grid_min[VPKT_MAXRANGES], nu_grid_max[VPKT_MAXRANGES];
  // init_vspecpol() vpkt.cc:491: widths of the time and frequency bins of the spectra (floats there), made on the host
  float delta_t[ARTIS_VSPEC_TIMEBINS];
  float delta_freq[ARTIS_VSPEC_NUBINS];
};
// The real packet as trace_vpkts() (vpkt.cc:948) sees it at an emission or an electron scattering. A virtual packet never
// changes the real one and draws no random number, so the kernels only RECORD the event; k_vpkt traces every recorded
// event towards every observer afterwards (one lane per event and direction) and adds to the spectra.
struct alignas(16) VpktSeed {
  double pos[3], dir[3];
  double nu_cmf, e_cmf, prop_time, stokes_q, stokes_u, absorptionfreq;
  int32_t cellindex, next_trans, type_before, pad;
};

struct DevEst {
  double *J, *nuJ, *ffheatingestimator, *colheatingestimator, *gammaestimator, *bfheatingestimator;
  double *dep_estimator_gamma;  // [cell] gammapkt.cc:568
  double *dep_estimator_electron, *dep_estimator_positron, *dep_estimator_alpha;  // [cell] update_packets.cc:160-173
  double *scalars;              // [ARTIS_NSCALARS]
  // builds with the multibin radiation field / detailed bound-free estimators (else null)
  double *radfieldbin_J, *radfieldbin_nuJ;  // [cell][RADFIELDBINCOUNT] radfield.cc:745-790
  double *bfrate_raw;                       // [cell][nbfestim] radfield.cc:215
  double *Jb_lu_raw, *Jb_lu_contribcount;   // [cell][detailed_linecount] radfield.cc:773 (the count kept as f64: one block, one all-reduce)
  // builds with VPKT_ON (else null): the observers' spectra and the velocity-grid map (include/artis_amd.h artis_estimators)
  double *vspecpol, *vgrid_flux;
};

// Packet population in HBM: three arrays of cache-line records, slot-major ("structure of lines").
// The propagation kernels are persistent and work-pulling: a lane retires its packet and takes the next one on its own,
// so a packet is loaded and stored by ONE lane at a time. With one array per field that lane would touch ~40 cache
// lines and dirty a few bytes of each; here it reads and writes whole lines:
//   PktHot    everything a thermal packet (k-packet, walking macro-atom) needs and changes, and the dispatch fields
//   PktFlight what only an r-packet / gamma packet in flight needs: direction, rest-frame quantities, its ContinuumOpacity
//   PktCold   written at rare events, read at download
// Fields are those of the reference's struct Packet (packet.h:117-169; number and pellet_nucindex never leave the
// caller's array) plus engine-private state that makes any kernel boundary legal.
constexpr int32_t PKT_FLAG_TRUEEM_NAN = 1;  // trueem_pos is NaN (kpkt.cc:483): materialised at download, not stored per event
constexpr int32_t PKT_FLAG_EMITTED = 2;     // transient, inside a kernel: emit_rpkt() ran, the flight line has to be written

struct alignas(128) PktHot {
  uint32_t rng[4];                        // Xoshiro128PP words
  double prop_time, pos_x;
  double pos_y, pos_z;
  double e_cmf, nu_cmf;
  int32_t type, cellindex, next_trans, nscatterings;
  // an activated macro-atom that has not deactivated yet (MacroAtomState packet.h:103; ma_level < 0 = none).
  // ma_origin: 1 = activated inside do_rpkt_step() (the packet continues its do_rpkt() loop afterwards), 0 = by a k-packet.
  int32_t ma_element, ma_ion, ma_level, ma_line;
  // pend != 0: a rare, register-hungry action that was sampled but is executed by the slow-path kernel
  //   PEND_MA_ACTION: the bound-free macro-atom transition `pend_arg` (an ARTIS_MA_ACTION_*) of the active macro-atom;
  //   PEND_KPKT_FB: free-bound emission of a k-packet into continuum (ma_element, ma_ion, pend_arg = lower level, ma_line = target).
  // chi_mgi: cell of the packet's ContinuumOpacity (rpkt.h:70; < 0 = not valid)
  int32_t ma_origin, pend, pend_arg, chi_mgi;
  int32_t emissiontype, trueemissiontype, absorptiontype, flags;
};
struct alignas(128) PktFlight {
  double dir_x, dir_y;
  double dir_z, nu_rf;
  double e_rf, stokes_q;
  double stokes_u, absorptionfreq;
  double chi_nu, chi_es;                  // ContinuumOpacity: nu, chi_escatter, chi_freefree_heat, chi_boundfree
  double chi_ff, chi_bf;
  double em_pos_x, em_pos_y;
  double em_pos_z;
  float em_time, pad0;
};
struct alignas(64) PktCold {
  double trueem_pos_x, trueem_pos_y;
  double trueem_pos_z, tdecay;            // tdecay, pellet_decaytype, originated_particle: read-only here (update_pellet)
  float escape_time, trueem_time;
  int32_t escape_type, pellet_decaytype;
  int32_t originated_particle, pad[3];
};
static_assert(sizeof(PktHot) == 128 && sizeof(PktFlight) == 128 && sizeof(PktCold) == 64, "packet records are whole cache lines");

struct PktStore {
  PktHot *hot;
  PktFlight *flight;
  PktCold *cold;
  int64_t n;
};
constexpr size_t PKT_BYTES_PER_PACKET = sizeof(PktHot) + sizeof(PktFlight) + sizeof(PktCold);

}  // namespace artis
