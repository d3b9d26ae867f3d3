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
// alltrans_startdown, ndowntrans, nuptrans, and the offset (in doubles) of the level's macro-atom record inside a
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
// Macro-atom record of one (cell, level), 128-byte aligned and SELF-CONTAINED: one transition of the walk
// (macroatom.cc:385-577) reads this record and nothing else -- no level table, no transition table -- so the chain of
// dependent reads per transition is record -> cumulative sums -> target, all inside a few adjacent cache lines.
// The unit of the layout is the 128-byte line: a transition touches line 0 and one line of its direction. In doubles:
//   line 0   [0..1]        MaHeader (16 B): ndown, nup, unique level index, alltrans_startdown, place in the hot block
//            [2..10]       the 9 process rates              (alllevels_maprocessrates, globals.h:286)
//            [12], [13]    copies of the MaTarget of the first downward and of the first upward transition (a direction
//                          with one transition, a third of all searches, reads no other line)
//            [14..15]      FILTER: 8 x uint16, the cumulative rates of actions 0..7 as fractions of the total (below)
//   then one line per 7 transitions of a direction, downward lines first (marec_down), then upward (marec_up(ndown)):
//            [0..1]        FILTER: 7 x uint16, the line's cumulative sums as fractions of the direction's whole rate
//                          (0x7FFF for a sum that is not searched); the 8th uint16: 0x7FFF, anything else = "do not use
//                          this line's filter" (a sum that is not a finite fraction)
//            [2..9)        cumulative internal-down-same / internal-up-same (allmacroatomictransitions blocks 2 and 3,
//                          macroatom.cc:44, :51) of transitions 7b .. 7b+6
//            [9..16)       their MaTarget (8 B each): the target level and the offset of ITS record in the cell's row
//                          (static data, repeated per cell so that it sits in the line of the sums that select it)
//   [marec_rad(..) ..)     cumulative radiative deexc.      (block 1, macroatom.cc:58), contiguous (read once per walk)
// A search reads whole lines of sums (entries beyond the count are never used); the rad block may be read up to 7 doubles
// past its end, which stays inside the row (+ MAREC_SLACK at the end of the allocation).
//
// FILTERS. k_thermal is bound by the NUMBER of vector-memory instructions it issues (two more 8-byte reads of a line it
// has already read, per transition: 617 -> 786 ms), and a transition decided on the f64 values reads
// 64 B of rates + 56 B of sums + a target = 9 instructions. Both decisions are "how many cumulative values are <= z * whole"
// with z uniform in [0, 1): the same as "how many fractions value / whole are <= z" unless z lies within rounding of a
// fraction. The fractions are kept as 15-bit integers q = floor(fraction * 32768) (clamped to 32767) in uint16, so
// q <= fraction * 32768 <= q + 1: with zi = floor(z * 32768) (the top 15 bits of the 24-bit draw), zi >= q + 2 proves
// value <= z * whole and zi <= q - 1 proves the opposite, by margins of 3e-5 and 6e-8 of the whole against f64 rounding errors
// of 1e-16 (physics.h mafilt_count). Anything in between (q == zi or zi - 1: 5e-4 of the draws per
// decision) is decided on the f64 values as before -- same random numbers, same result. 15 bits, so that two entries are
// compared by ONE 32-bit subtraction (physics.h mafilt_count). A transition then reads 16 B + 16 B + a target.
struct alignas(16) MaHeader {
  int16_t ndown, nup;
  int32_t ul, alltrans_startdown;
  int16_t hot, pad;
};
// What a transition needs to know about the level it leads to, in 8 bytes, so that the NEXT transition reads neither a
// level table nor the target record's header: where the target's record is, which level it is, and how many downward
// and upward transitions it has (= where the blocks of its record begin).
struct MaTarget {
  uint64_t bits;  // [0..20) record offset / MAREC_ALIGN   [20..36) level within its ion   [36..50) ndown   [50..64) nup
};
constexpr int64_t MATGT_MAX_RECUNITS = 1 << 20;  // rows up to 128 MB
constexpr int MATGT_MAX_LEVEL = 1 << 16, MATGT_MAX_NTRANS = 1 << 14;
static_assert(sizeof(MaHeader) == 16 && sizeof(MaTarget) == 8, "record header and target sizes");
#ifndef ARTIS_HOT_DOUBLES
#define ARTIS_HOT_DOUBLES 512  // 4 KB per cell
#endif
constexpr int HOT_DOUBLES = ARTIS_HOT_DOUBLES;
constexpr int MAREC_ALIGN = 16;  // doubles
constexpr int marec_even(int n) { return (n + 1) & ~1; }
constexpr int marec_rates = 2;
constexpr int marec_tgt0 = 12;   // [12] first downward target, [13] first upward target
constexpr int marec_filt0 = 14;  // [14..15] the action filter of line 0
constexpr int MAREC_LINE = 16;   // doubles per line of a direction: filter (2), 7 sums, 7 targets
constexpr int MAREC_PER = 7;     // transitions per line
constexpr int marec_lines(int n) { return (n + MAREC_PER - 1) / MAREC_PER; }
constexpr int marec_down = 16;
constexpr int marec_up(int ndown) { return marec_down + (marec_lines(ndown) * MAREC_LINE); }
constexpr int marec_rad(int ndown, int nup) { return marec_up(ndown) + (marec_lines(nup) * MAREC_LINE); }
constexpr int marec_size(int ndown, int nup) { return marec_rad(ndown, nup) + marec_even(ndown); }
// entry i of a direction whose lines begin at `base`: its cumulative sum, its target
constexpr int marec_sum(int base, int i) { return base + ((i / MAREC_PER) * MAREC_LINE) + 2 + (i % MAREC_PER); }
constexpr int marec_tgt(int base, int i) { return marec_sum(base, i) + MAREC_PER; }
constexpr double MAFILT_SCALE = 32768.;
constexpr uint32_t MAFILT_NONE = 0x7FFFu;  // an entry that is never counted
constexpr int MAREC_SLACK = 16;  // doubles past the last row that a padded search may touch

struct alignas(16) D2 {
  double x, y;
};
struct alignas(16) U4 {  // 8 x uint16 of a macro-atom filter
  uint32_t w[4];
};
// one line of sums of one direction of one level's record (static): where it is in a cell's row, where the direction's
// whole rate is, which transitions of the direction it holds (k_mafilter: a thread per cell and line)
struct alignas(16) MaLineRef {
  int32_t line_off, rate_off, first, n;
};

struct VpktConfig;  // virtual packets, below
struct DevModel {
  int32_t nelements, nions, nlevels, nlines, nalltrans, nphixstargets_total, nphixslevels, nbfcontinua, nbfcontinua_ground,
      ncoolingterms, nmatransblock, NPHIXSPOINTS;
  int32_t nmacache;    // doubles per cell in DevCache::macache
  int32_t nupcum;      // doubles per cell in DevCache::collexc_cum (= number of upward transitions = nlines)
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
  const int32_t *level_upcum_start;  // derived: offset of the level's upward transitions in DevCache::collexc_cum
  const int32_t *alltrans_lineindex, *alltrans_targetlevelindex;
  const int32_t *alltrans_owner;  // derived: the level whose block of alltrans an entry belongs to
  // derived: alltrans cut at level boundaries into runs of at most ~256 entries, [nscanblk + 1] start indices: one wave
  // of k_matrans forms the running sums of one run
  const int32_t *scanblk_start;
  int32_t nscanblk;
  const MaLineRef *malines;  // derived: every line of sums of every record, [nmalines]
  int32_t nmalines;
  // ... and those whose direction's transitions are not all inside one 64-transition chunk of k_matrans' scan: k_matrans
  // writes the filter of a line when it has the direction's whole rate at hand, k_mafilter the filters of these lines
  const MaLineRef *malines_fix;
  int32_t nmalines_fix;
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

struct DevCache {
  double *levelpops;             // [cell][nlevels]
  double *macache;               // [cell][nmacache]: one record per level, see LevelPack
  double *hotblk;                // [cell][HOT_DOUBLES]: copies of the records of the cell's hottest levels (MaHeader::hot)
  float *hotness;                // [cell][nlevels]: level population x total macro-atom rate (populate_macroatom)
  int16_t *hotoff;               // [cell][nlevels]: place of each level's copy in the hot block, or -1
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
  double *line_dpop;             // [cell][nlines]: B_lu n_l - B_ul n_u of every line, the population factor of get_tau_sobolev() (rpkt.cc:75)
  double *collexc_cum;           // [cell][nupcum]: running cooling sum after each upward transition of each level (kpkt.cc:461-476)
  double *corrphotoioncoeff;     // [cell][nphixstargets_total]
  // [cell][nphixstargets_total] the other coefficients of each bound-free pair (populate_corrphotoion): radiative and
  // collisional recombination, collisional ionisation, bound-free cooling
  double *bf_radrecomb, *bf_colrecomb, *bf_colion, *bf_cooling;
  double *cooling_contrib;       // [cell][ncoolingterms]
  double *ion_cooling_contribs;  // [cell][nions]
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
