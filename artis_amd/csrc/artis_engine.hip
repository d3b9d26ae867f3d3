// artis_engine.hip -- the MI355X packet-propagation engine behind include/artis_amd.h.
//
// Layout of the work on the GPU (gfx950, wave64):
//   * model tables, cell state and the per-cell cache live in HBM for the whole timestep;
//   * packets are three arrays of cache-line records (tables.h PktStore: hot / flight / cold), one lane per packet;
//   * one timestep = populate kernels (cell cache) + repeated k_propagate launches over a compacted
//     list of packets that still need updating. Each thread advances its packet by at most `budget`
//     do_packet() calls, then survivors are appended to the next list with a wave ballot;
//   * estimators are accumulated with hardware f64 atomics, event counters in LDS and flushed per block.
// There is no CPU path in this library: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is bound at run time (rccl_api below)

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "model_build.h"
#include "physics.h"

using namespace artis;

namespace {

thread_local std::string g_last_error;

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                            \
      return ARTIS_ERR_HIP;                                                                        \
    }                                                                                              \
  } while (0)

constexpr int BLOCK = 256;

// ------------------------------------------------------------------ populate kernels
// The cells a fill works on: every cell (the whole cache at once, or a batch of it), or -- a tiled cache (Env::fill_cells) -- the listed ones: the cells
// make_resident() gave rows to.
__device__ inline int64_t fill_count(const Env &env) { return env.fill_cells ? env.nfill : (env.tile_hi - env.tile_lo); }
__device__ inline int fill_cell(const Env &env, int64_t k) { return env.fill_cells ? env.fill_cells[k] : env.tile_lo + (int)k; }
#ifndef ARTIS_MA_WAVE_FILL
#define ARTIS_MA_WAVE_FILL 1  // a cold level's record filled by the wave (ma_fill_record_wave); 0: by the lane (physics.h ma_fill_record)
#endif
#ifndef ARTIS_COLD_COOLING_ONLY
#define ARTIS_COLD_COOLING_ONLY 1  // the population evaluates of a cold level's transitions only the cooling terms (physics.h matrans_terms<true>)
#endif
__global__ void __launch_bounds__(BLOCK) k_levelpops(Env env) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t total = fill_count(env) * env.M.nlevels;
  if (i >= total) return;
  populate_levelpop(env, fill_cell(env, i / env.M.nlevels), (int)(i % env.M.nlevels));
}
__global__ void __launch_bounds__(BLOCK) k_line_dpop(Env env) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t total = fill_count(env) * env.M.nlines;
  if (i >= total) return;
  populate_line_dpop(env, fill_cell(env, i / env.M.nlines), (int)(i % env.M.nlines));
}
#if ARTIS_EXPOPAC_TABLES
// calculate_expansion_opacities() for the cells of the resident tile, when the host did not hand the tables over
__global__ void __launch_bounds__(BLOCK) k_expopac(Env env) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t total = fill_count(env) * ARTIS_EXPOPAC_NBINS;
  if (i >= total) return;
  const int c = fill_cell(env, i / ARTIS_EXPOPAC_NBINS);
  if (env.C.thick[c] == ARTIS_CELL_THICK) return;  // update_grid.cc:657
  populate_expopac_bin(env, c, (int)(i % ARTIS_EXPOPAC_NBINS));
}
__global__ void __launch_bounds__(BLOCK) k_expopac_planck(Env env) {
  const int64_t kf = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (kf >= fill_count(env)) return;
  const int c = fill_cell(env, kf);
  if (env.C.thick[c] == ARTIS_CELL_THICK) return;
  populate_expopac_planck(env, c);
}
#endif
#if ARTIS_OPT_NT_ON
// every cell of the model (not a tile): the non-thermal ionisation rate coefficients and energy-rate sums
__global__ void __launch_bounds__(BLOCK) k_nt_cells(Env env, int ncell) {
  const int c = blockIdx.x * BLOCK + threadIdx.x;
  if (c >= ncell) return;
  if (!populate_nt_cell(env, c)) fail(env, 90);
}
#endif
__global__ void __launch_bounds__(BLOCK) k_cell_scalars(Env env) {
  const int64_t kf = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (kf >= fill_count(env)) return;
  const int c = fill_cell(env, kf);
  populate_chi_ff(env, c);
}
// one wave = one 64-bit word of a cell's keep bitmap: the ballot IS the word (globals.h:296-305)
__global__ void __launch_bounds__(BLOCK) k_allcont(Env env) {
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = fill_count(env) * env.M.nkeepwords;
  if (wave >= nwaves) return;
  const int c = fill_cell(env, wave / env.M.nkeepwords);
  const int word = (int)(wave % env.M.nkeepwords);
  const int i = word * 64 + lane;
  bool keep = false;
  if (i < env.M.nbfcontinua) keep = populate_allcont(env, c, i);
  const unsigned long long bits = __ballot(keep);
  if (lane == 0) env.K.allcont_keepbits[(krow(env, c) * env.M.nkeepwords) + word] = bits;
}
// one wave = one cell: the kept continua as a list, the count of kept continua below each bitmap word and the pairs in list
// order (physics.h populate_keptlist; after k_allcont)
__global__ void __launch_bounds__(BLOCK) k_keptlist(Env env) {
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= fill_count(env)) return;
  const int c = fill_cell(env, wave);
  const int nw = env.M.nkeepwords;
  const uint64_t *keep = env.K.allcont_keepbits + (krow(env, c) * nw);
  int32_t *list = env.K.allcont_keptlist + (krow(env, c) * env.M.nbfcontinua);
  int32_t *prefix = env.K.allcont_keepprefix + (krow(env, c) * nw);
  const D2 *pair = env.K.allcont_pair + (krow(env, c) * env.M.nbfcontinua);
  D2 *keptpair = env.K.allcont_keptpair + (krow(env, c) * env.M.nbfcontinua);
  int base = 0;
  for (int w0 = 0; w0 < nw; w0 += 64) {
    const int j = w0 + lane;
    unsigned long long word = (j < nw) ? keep[j] : 0ull;
    const int pc = __popcll(word);
    int incl = pc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    int at = base + incl - pc;
    if (j < nw) prefix[j] = at;
    while (word != 0ull) {
      const int i = (j * 64) + __builtin_ctzll(word);
      list[at] = i;
      keptpair[at] = pair[i];
      at++;
      word &= word - 1ull;
    }
    base += __shfl(incl, 63);
  }
}
__global__ void __launch_bounds__(BLOCK) k_corrphotoion(Env env, const int32_t *target_level) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t total = fill_count(env) * env.M.nphixstargets_total;
  if (i >= total) return;
  const int c = fill_cell(env, i / env.M.nphixstargets_total);
  const int k = (int)(i % env.M.nphixstargets_total);
  const int ul = target_level[k];
  populate_corrphotoion(env, c, ul, k - env.M.level_phixstargetstart[ul]);
}
// Bound-bound part of a cell's macro-atom records: one WAVE per (cell, block of alltrans entries), three passes over the
// block's entries in LDS (round 4; rounds 2-3: a wave scan in registers whose serial steps -- one per position in the longest
// segment of a 64-entry chunk, 18 instructions each for all 64 lanes -- were 11 % of the population with ~9 transitions per
// direction and 26 % with ~25):
//   1. every lane evaluates the rate coefficients of its transitions (every lane busy, whatever the levels' transition counts)
//      and leaves the three terms each adds to its level's running sums (macroatom.cc:64-140) in LDS;
//   2. ONE LANE PER SEGMENT (a level's downward or upward transitions) adds its segment's terms up in the reference's order --
//      the additions of the sequential form (physics.h populate_level_bb / populate_dirfilter_seq), so the same bits -- 10-30
//      segments side by side, and writes the level's rates;
//   3. every lane turns its transitions' sums into filter entries (tables.h "FILTERS": fractions of the segment's last sum); a
//      line is usable when every one of its entries is a finite fraction (a flag per line in LDS).
// A segment longer than a block (DevModel::malongsegs) is a block of its own: its rates are summed here 64 terms at a time, its
// filters written by k_mafilter_long.
// the value of the lane below (lane 0: 0), as two DPP wave-shift moves
__device__ inline double wave_shr1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// the lanes of the wave that hold the entries of this lane's filter line (7 transitions of one direction: consecutive lanes)
__device__ inline unsigned long long line_lanes(int lane, int ti, int seglen) {
  const int first = lane - (ti % MAREC_PER);
  const int left = seglen - (ti - (ti % MAREC_PER));
  const int cnt = left < MAREC_PER ? left : MAREC_PER;
  return ((1ull << cnt) - 1ull) << first;
}
// where entry e of a block (alltrans index a0 + e) sits in its segment
struct MaEntryPos {
  LevelPack lpk;
  int ti, seglen, e_seg0;  // index within the direction, the direction's transitions, the block entry of the direction's first
  bool isdown;
};
__device__ inline MaEntryPos ma_entry_pos(const DevModel &M, int a0, int e) {
  MaEntryPos r;
  r.lpk = M.level_pack[M.alltrans_owner[a0 + e]];
  const int i = (a0 + e) - r.lpk.alltrans_startdown;
  r.isdown = i < r.lpk.ndown;
  r.ti = r.isdown ? i : i - r.lpk.ndown;
  r.seglen = r.isdown ? r.lpk.ndown : r.lpk.nup;
  r.e_seg0 = e - r.ti;
  return r;
}
__global__ void __launch_bounds__(BLOCK, 4) k_matrans(Env env) {
  __shared__ double lds_v[BLOCK / 64][3][MATRANS_BLOCK];          // the terms, then the running sums
  __shared__ uint16_t lds_q[BLOCK / 64][2][MATRANS_BLOCK];        // filter entries: internal, radiative
  __shared__ uint8_t lds_qf[BLOCK / 64][MATRANS_BLOCK];           // ... the internal entries' fine bytes (tables.h "FINE BYTES")
  __shared__ uint8_t lds_ok[BLOCK / 64][2][MATRANS_BLOCK];        // per filter line (at its first entry): every entry a finite fraction
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const DevModel &M = env.M;
  const int nblk = M.nscanblk;
  if (wave >= fill_count(env) * nblk) return;
  const int64_t kf = wave / nblk;
  const int c = fill_cell(env, kf);
  const int blk = (int)(wave % nblk);
  const int seg0 = M.scanblk_seg0[blk], seg1 = M.scanblk_seg0[blk + 1];
  const MaLongSeg sfirst = M.scansegs[seg0], slast = M.scansegs[seg1 - 1];
  const int a0 = sfirst.ats0, nent = (slast.ats0 + slast.n) - a0;
  U4 *row = env.K.macache + (krow(env, c) * M.nmacache);
  double *upterms = env.collexc_terms + (kf * M.nupcum);
  double(*v)[MATRANS_BLOCK] = lds_v[w];
  if (nent > MATRANS_BLOCK) {
    // one long segment: its rates, 64 terms at a time (lane k adds its term to the finished sum of lane k - 1; lane 0 takes the carry)
    const LevelPack lpk = M.level_pack[sfirst.ul];
    double c0 = 0., c1 = 0., c2 = 0.;
    for (int base = 0; base < nent; base += 64) {
      const bool valid = base + lane < nent;
      MaTransTerms t;
      t.v0 = t.v1 = t.v2 = t.kterm = 0.;
      if (valid) {
        t = matrans_terms<ARTIS_COLD_COOLING_ONLY != 0>(env, c, a0 + base + lane);
        if (!t.isdown) upterms[M.level_upcum_start[t.ul] + t.i] = t.kterm;
      }
      double s0 = (lane == 0 ? c0 : 0.) + t.v0, s1 = (lane == 0 ? c1 : 0.) + t.v1, s2 = (lane == 0 ? c2 : 0.) + t.v2;
      for (int k = 1; k < 64; k++) {
        const double p0 = wave_shr1(s0), p1 = wave_shr1(s1), p2 = wave_shr1(s2);
        if (lane == k) {
          s0 = p0 + t.v0;
          s1 = p1 + t.v1;
          s2 = p2 + t.v2;
        }
      }
      const int last = (nent - base < 64 ? nent - base : 64) - 1;
      c0 = __shfl(s0, last);
      c1 = __shfl(s1, last);
      c2 = __shfl(s2, last);
    }
    if (lane == 0 && lpk.rec_off >= 0) {  // (a cold level has no static record: only its cooling terms above are kept)
      double *rates = ma_rates_of(row + lpk.rec_off, lpk.ndown, lpk.nup);
      if (sfirst.dir == 0) {
        rates[ARTIS_MA_ACTION_RADDEEXC] = c0;
        rates[ARTIS_MA_ACTION_COLDEEXC] = c1;
        rates[ARTIS_MA_ACTION_INTERNALDOWNSAME] = c2;
      } else {
        rates[ARTIS_MA_ACTION_INTERNALUPSAME] = c0;
      }
    }
    return;
  }
  // 1. the terms (the block's entries in the order of DevModel::scanperm: a wave's 64 entries are of one kind wherever the block has 64 of it)
  for (int k = lane; k < nent; k += 64) {
    const int e = M.scanperm[a0 + k];
    const MaTransTerms t = matrans_terms<ARTIS_COLD_COOLING_ONLY != 0>(env, c, a0 + e);
    v[0][e] = t.v0;
    v[1][e] = t.v1;
    v[2][e] = t.v2;
    lds_ok[w][0][e] = 1;
    lds_ok[w][1][e] = 1;
    if (!t.isdown) upterms[M.level_upcum_start[t.ul] + t.i] = t.kterm;
  }
  __threadfence_block();  // (the passes exchange data between the lanes of this wave through LDS: in order, but the compiler must not move them)
  // 2. the running sums, a lane per segment
  for (int si = seg0 + lane; si < seg1; si += 64) {
    const MaLongSeg sg = M.scansegs[si];
    const int o = sg.ats0 - a0;
    double s0 = 0., s1 = 0., s2 = 0.;
    for (int i = 0; i < sg.n; i++) {
      s0 += v[0][o + i];
      s1 += v[1][o + i];
      s2 += v[2][o + i];
      v[0][o + i] = s0;
      v[2][o + i] = s2;
    }
    const LevelPack lpk = M.level_pack[sg.ul];
    if (lpk.rec_off < 0) continue;
    double *rates = ma_rates_of(row + lpk.rec_off, lpk.ndown, lpk.nup);
    if (sg.dir == 0) {
      rates[ARTIS_MA_ACTION_RADDEEXC] = s0;
      rates[ARTIS_MA_ACTION_COLDEEXC] = s1;
      rates[ARTIS_MA_ACTION_INTERNALDOWNSAME] = s2;
    } else {
      rates[ARTIS_MA_ACTION_INTERNALUPSAME] = s0;
    }
  }
  __threadfence_block();
  // 3a. the filter entries of every transition: its running sums as fractions of the direction's last ones
  for (int e = lane; e < nent; e += 64) {
    const MaEntryPos ps = ma_entry_pos(M, a0, e);
    const int e_last = ps.e_seg0 + ps.seglen - 1;
    const double whole_int = v[ps.isdown ? 2 : 0][e_last], whole_rad = v[0][e_last];
    bool ok_int = (whole_int > 0.) && (whole_int <= DBLMAX), ok_rad = (whole_rad > 0.) && (whole_rad <= DBLMAX);
    uint32_t q_int = MAFILT_NONE23, q_rad = MAFILT_NONE;
    if (ps.ti < ps.seglen - 1) {
      if (ok_int) q_int = mafilt_quant23(v[ps.isdown ? 2 : 0][e], whole_int, &ok_int);
      if (ok_rad && ps.isdown) q_rad = mafilt_quant(v[0][e], whole_rad, &ok_rad);
    }
    lds_q[w][0][e] = (uint16_t)(q_int >> 8);
    lds_qf[w][e] = (uint8_t)(q_int & 0xFFu);
    lds_q[w][1][e] = (uint16_t)q_rad;
    const int e_line = e - (ps.ti % MAREC_PER);
    if (!ok_int) lds_ok[w][0][e_line] = 0;
    if (ps.isdown && !ok_rad) lds_ok[w][1][e_line] = 0;
  }
  __threadfence_block();
  // 3b. ... into the records: the entries of a line that is not usable are 0, its mark too
  for (int e = lane; e < nent; e += 64) {
    const MaEntryPos ps = ma_entry_pos(M, a0, e);
    if (ps.lpk.rec_off < 0) continue;
    const int e_line = e - (ps.ti % MAREC_PER);
    U4 *rec = row + ps.lpk.rec_off;
    const bool lok_int = lds_ok[w][0][e_line] != 0;
    U4 *line = rec + marec_slot(ps.isdown ? MADIR_DOWN : MADIR_UP, ps.ti / MAREC_PER, ps.lpk.ndown, ps.lpk.nup);
    mafilt_put(line, ps.ti % MAREC_PER, lok_int ? lds_q[w][0][e] : 0u);
    ((uint8_t *)rec)[marec_fine_byte0(ps.isdown ? MADIR_DOWN : MADIR_UP, ps.ti / MAREC_PER, ps.lpk.ndown, ps.lpk.nup) + (ps.ti % MAREC_PER)] =
        lok_int ? lds_qf[w][e] : (uint8_t)0u;
    if (ps.ti % MAREC_PER == 0) mafilt_put(line, 7, lok_int ? MAFILT_NONE : 0u);  // the line's "usable" mark
    if (ps.isdown) {
      const bool lok_rad = lds_ok[w][1][e_line] != 0;
      U4 *rline = rec + marec_slot(MADIR_RAD, ps.ti / MAREC_PER, ps.lpk.ndown, ps.lpk.nup);
      mafilt_put(rline, ps.ti % MAREC_PER, lok_rad ? lds_q[w][1][e] : 0u);
      if (ps.ti % MAREC_PER == 0) mafilt_put(rline, 7, lok_rad ? MAFILT_NONE : 0u);
    }
  }
}
__device__ inline double row_shr1(double x);
// The recombination sums of a level (macroatom.cc:147-168: over the levels of the ion below that ionise into it) -- three running sums
// over up to ~60 channels for the 45 levels of the bench's data that have any, none for the other 1522. As a loop of populate_macroatom()
// the lane of such a level walked its list with three dependent gathers per channel while the other 63 lanes of its wave waited
// (k_macroatom at 0.21 lane utilisation). Here a ROW OF 16 LANES per (cell, listed level) reads 16 channels at a time and forms the
// sums with the loop's additions in the loop's order (lane k adds its term to lane k-1's finished sum, as k_cooling_chain does).
__global__ void __launch_bounds__(BLOCK) k_macroatom_recomb(Env env) {
  const int64_t row_id = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 4;
  const int r = threadIdx.x & 15;
  const DevModel &M = env.M;
  const int64_t nrows = fill_count(env) * M.nrecomblevels;
  const bool valid = row_id < nrows;
  const int c = fill_cell(env, (valid ? row_id : 0) / M.nrecomblevels);
  const int ul = M.recomb_levels[(valid ? row_id : 0) % M.nrecomblevels];
  const int ui = M.level_ion[ul];
  const int ls = M.ion_uniquelevelindexstart[ui > 0 ? ui - 1 : 0];
  const int64_t cb = krow(env, c) * M.nphixstargets_total;
  const double e_cur = eps(M, ul);
  const int j0 = valid ? M.level_recomb_start[ul] : 0, j1 = valid ? M.level_recomb_start[ul + 1] : 0;
  double carry0 = 0., carry1 = 0., carry2 = 0.;
  for (int j = j0; __any(j < j1); j += 16) {
    const bool in = (j + r) < j1;
    double x0 = 0., x1 = 0., x2 = 0.;
    if (in) {
      const int lower = M.recomb_lower[j + r];
      const int t = M.recomb_target[j + r];
      const double e_target = eps(M, ls + lower);
      const double e_trans = e_cur - e_target;
      const int64_t o = cb + M.level_phixstargetstart[ls + lower] + t;
      const double R = env.K.bf_radrecomb[o];
      const double Cc = env.K.bf_colrecomb[o];
      x0 = (R + Cc) * e_target;
      x1 = R * e_trans;
      x2 = Cc * e_trans;
    }
    double a0 = (r == 0) ? carry0 + x0 : x0, a1 = (r == 0) ? carry1 + x1 : x1, a2 = (r == 0) ? carry2 + x2 : x2;
#pragma unroll
    for (int s = 1; s < 16; s++) {
      const double p0 = row_shr1(a0), p1 = row_shr1(a1), p2 = row_shr1(a2);
      if (r == s) {
        a0 = p0 + x0;
        a1 = p1 + x1;
        a2 = p2 + x2;
      }
    }
    // the sums after the row's last channel, broadcast to the row (lanes past the list's end added 0. to them)
    const int src = ((threadIdx.x & 63) | 15) << 2;
    carry0 = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(a0)), __builtin_amdgcn_ds_bpermute(src, __double2loint(a0)));
    carry1 = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(a1)), __builtin_amdgcn_ds_bpermute(src, __double2loint(a1)));
    carry2 = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(a2)), __builtin_amdgcn_ds_bpermute(src, __double2loint(a2)));
  }
  if (valid && r == 0) {
    const LevelPack lpk = M.level_pack[ul];
    double *rates = ma_rates_of(ma_rec_of(env, c, lpk), lpk.ndown, lpk.nup);
    rates[ARTIS_MA_ACTION_INTERNALDOWNLOWER] = carry0;
    rates[ARTIS_MA_ACTION_RADRECOMB] = carry1;
    rates[ARTIS_MA_ACTION_COLRECOMB] = carry2;
  }
}
__global__ void __launch_bounds__(BLOCK) k_macroatom(Env env) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t total = fill_count(env) * env.M.nlevels;
  if (i >= total) return;
  if (env.M.level_pack[i % env.M.nlevels].rec_off < 0) return;  // (a cold level: filled when a packet reaches it, physics.h ma_slow_fill)
  populate_macroatom<true>(env, fill_cell(env, i / env.M.nlevels), (int)(i % env.M.nlevels));
}
// The filters of the directions with more transitions than a block of k_matrans holds (DevModel::malongsegs: > MATRANS_BLOCK):
// a wave per (cell, segment) re-forms the running sums 63 transitions (nine filter lines) at a time -- the same terms
// added in the same order, lane k to the finished sum of lane k-1 -- and quantises them with the direction's whole rates,
// which k_matrans has left in the record. (physics.h populate_dirfilter_seq is the sequential form.)
__global__ void __launch_bounds__(BLOCK) k_mafilter_long(Env env) {
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int nseg = env.M.nmalongsegs;
  if (wave >= fill_count(env) * nseg) return;
  const int c = fill_cell(env, wave / nseg);
  const MaLongSeg seg = env.M.malongsegs[wave % nseg];
  const LevelPack lpk = env.M.level_pack[seg.ul];
  U4 *rec = env.K.macache + (krow(env, c) * env.M.nmacache) + lpk.rec_off;
  const double *rates = ma_rates_of(rec, lpk.ndown, lpk.nup);
  const bool down = seg.dir == 0;
  const double whole_int = rates[down ? ARTIS_MA_ACTION_INTERNALDOWNSAME : ARTIS_MA_ACTION_INTERNALUPSAME];
  const double whole_rad = rates[ARTIS_MA_ACTION_RADDEEXC];
  double ca = 0., cb = 0.;
  for (int base = 0; base < seg.n; base += 63) {
    const int ti = base + lane;
    const bool valid = lane < 63 && ti < seg.n;
    double a = 0., b = 0.;
    if (valid) {
      const MaTransTerms t = matrans_terms(env, c, seg.ats0 + ti);
      a = down ? t.v2 : t.v0;
      b = down ? t.v0 : 0.;
    }
    double sa = (lane == 0 ? ca : 0.) + a, sb = (lane == 0 ? cb : 0.) + b;
    for (int k = 1; k < 63; k++) {
      const double pa = wave_shr1(sa), pb = wave_shr1(sb);
      if (lane == k) {
        sa = pa + a;
        sb = pb + b;
      }
    }
    bool ok_int = (whole_int > 0.) && (whole_int <= DBLMAX), ok_rad = (whole_rad > 0.) && (whole_rad <= DBLMAX);
    uint32_t q_int = MAFILT_NONE23, q_rad = MAFILT_NONE;
    if (valid && ti < seg.n - 1) {
      if (ok_int) q_int = mafilt_quant23(sa, whole_int, &ok_int);
      if (ok_rad && down) q_rad = mafilt_quant(sb, whole_rad, &ok_rad);
    }
    const unsigned long long bad_int = __ballot(valid && !ok_int), bad_rad = __ballot(valid && down && !ok_rad);
    if (valid) {
      const unsigned long long mine = line_lanes(lane, ti, seg.n);  // (63 lanes = nine whole lines: lane % 7 == ti % 7)
      const bool lok_int = (bad_int & mine) == 0ull, lok_rad = (bad_rad & mine) == 0ull;
      U4 *line = rec + marec_slot(down ? MADIR_DOWN : MADIR_UP, ti / MAREC_PER, lpk.ndown, lpk.nup);
      mafilt_put23(rec, down ? MADIR_DOWN : MADIR_UP, ti, lpk.ndown, lpk.nup, lok_int ? q_int : 0u);
      if (ti % MAREC_PER == 0) mafilt_put(line, 7, lok_int ? MAFILT_NONE : 0u);
      if (down) {
        U4 *rline = rec + marec_slot(MADIR_RAD, ti / MAREC_PER, lpk.ndown, lpk.nup);
        mafilt_put(rline, ti % MAREC_PER, lok_rad ? q_rad : 0u);
        if (ti % MAREC_PER == 0) mafilt_put(rline, 7, lok_rad ? MAFILT_NONE : 0u);
      }
    }
    const int last = (seg.n - base < 63 ? seg.n - base : 63) - 1;
    ca = __shfl(sa, last);
    cb = __shfl(sb, last);
  }
}
// ---- A cold level's record filled by a WAVE (tables.h "ON-DEMAND RECORDS"; physics.h ma_fill_record is the sequential form, which one lane of
// the slow-path kernel ran in the first version: 585 ms of a 6.5 s step with the w7big data, most of the 3.3 s of k_slow with cd23like). The
// transitions' terms -- the expensive part: rate coefficients with exp() and divisions -- are evaluated by 63 lanes side by side; the running
// sums are formed in the sequential loop's order (lane j ends with carry + t_0 + ... + t_j, added left to right); each lane quantises its own
// entry, a filter line's "usable" mark is a vote of its lanes (as in k_mafilter_long); lane 0 writes the rates and runs the bound-free part
// and the action filter (populate_macroatom). Same terms, same additions in the same order: the same bits (GPU test: the records a run
// left in the pool against populate_dirfilter_seq, artis_amd_debug_cellcache()).
__device__ inline double wave_prefix_inorder(double t, double carry, int n, int lane) {
  double acc = carry;
  for (int k = 0; k < n; k++) {
    const double x = wave_bcast(t, k);
    if (lane >= k) acc += x;
  }
  return acc;
}
// ... NS running sums at once, through LDS (round 6): the lanes leave their terms in the wave's own rows, lane j < NS adds row j up from left to
// right -- the same additions in the same order, so the same bits -- and every lane reads its sums back. As broadcasts and masked additions the
// three sums of a block of 63 transitions were ~950 wave instructions; this is 63 dependent additions by three lanes side by side (the fills
// of the 4e5-line set: 2.1 s of its 17 s step).
constexpr int PREFIX_ROWS = 3;
// the wave's rows (one set for all instantiations: 12 KB per workgroup of BLOCK threads)
__device__ inline double (*prefix_rows(bool out))[64] {
  __shared__ double rows_in[BLOCK / 64][PREFIX_ROWS][64];
  __shared__ double rows_out[BLOCK / 64][PREFIX_ROWS][64];
  const int w = (int)(threadIdx.x >> 6) % (BLOCK / 64);
  return out ? rows_out[w] : rows_in[w];
}
template <int NS>
__device__ inline void wave_prefix_inorder_n(const double (&t)[NS], const double (&carry)[NS], int n, int lane, double (&out)[NS]) {
  static_assert(NS <= PREFIX_ROWS, "rows");
  double(*rin)[64] = prefix_rows(false);
  double(*rout)[64] = prefix_rows(true);
#pragma unroll
  for (int j = 0; j < NS; j++) rin[j][lane] = t[j];
  __threadfence_block();  // (lanes of one wave exchange data through LDS: in order, but the compiler must not move the accesses)
  if (lane < NS) {
    double acc = 0.;
#pragma unroll
    for (int j = 0; j < NS; j++) acc = (lane == j) ? carry[j] : acc;
    const double *in = rin[lane];
    double *o = rout[lane];
    for (int k = 0; k < n; k++) {
      acc += in[k];
      o[k] = acc;
    }
  }
  __threadfence_block();
  const int k = (lane < n) ? lane : n - 1;  // (a lane past the block's last transition holds the block's total, as wave_prefix_inorder() gives it)
#pragma unroll
  for (int j = 0; j < NS; j++) out[j] = rout[j][k];
  __threadfence_block();  // (the next call writes the rows again)
}
// the entries of one block (lanes 0..62 = transitions base .. base + 62 of a direction of n) into the record: internal filter from s_int,
// radiative filter (downward only) from s_rad
__device__ inline void wave_put_dirfilters(U4 *rec, const LevelPack &lpk, bool down, int n, int ti, bool valid, double s_int, double s_rad,
                                           double whole_int, double whole_rad, int lane) {
  bool ok_int = (whole_int > 0.) && (whole_int <= DBLMAX), ok_rad = (whole_rad > 0.) && (whole_rad <= DBLMAX);
  uint32_t q_int = MAFILT_NONE23, q_rad = MAFILT_NONE;
  if (valid && ti < n - 1) {
    if (ok_int) q_int = mafilt_quant23(s_int, whole_int, &ok_int);
    if (ok_rad && down) q_rad = mafilt_quant(s_rad, whole_rad, &ok_rad);
  }
  const unsigned long long bad_int = __ballot(valid && !ok_int), bad_rad = __ballot(valid && down && !ok_rad);
  if (valid) {
    const unsigned long long mine = line_lanes(lane, ti, n);
    const bool lok_int = (bad_int & mine) == 0ull, lok_rad = (bad_rad & mine) == 0ull;
    U4 *line = rec + marec_slot(down ? MADIR_DOWN : MADIR_UP, ti / MAREC_PER, lpk.ndown, lpk.nup);
    mafilt_put23(rec, down ? MADIR_DOWN : MADIR_UP, ti, lpk.ndown, lpk.nup, lok_int ? q_int : 0u);
    if (ti % MAREC_PER == 0) mafilt_put(line, 7, lok_int ? MAFILT_NONE : 0u);
    if (down) {
      U4 *rline = rec + marec_slot(MADIR_RAD, ti / MAREC_PER, lpk.ndown, lpk.nup);
      mafilt_put(rline, ti % MAREC_PER, lok_rad ? q_rad : 0u);
      if (ti % MAREC_PER == 0) mafilt_put(rline, 7, lok_rad ? MAFILT_NONE : 0u);
    }
  }
}
__device__ inline void ma_fill_record_wave(const Env &env, int c, int ul) {
  const DevModel &M = env.M;
  const int lane = (int)(threadIdx.x & 63);
  const LevelPack lpk = M.level_pack[ul];
  U4 *rec = ma_rec_of(env, c, lpk);
  const int nd = lpk.ndown, nu = lpk.nup;
  {  // populate_mainit_at(): every filter entry "never counted", the rates zero
    const int nfilt = marec_rates_slot(nd, nu), nfine0 = marec_fine_slot0(nd, nu), nrec = marec_slots(nd, nu);
    const int ntot = ((nrec + MAREC_ALIGN - 1) / MAREC_ALIGN) * MAREC_ALIGN;
    const uint32_t none2 = MAFILT_NONE | (MAFILT_NONE << 16);
    for (int i = lane; i < ntot; i += 64)
      rec[i] = (i < nfilt) ? U4{{none2, none2, none2, none2}} : ((i >= nfine0 && i < nrec) ? U4{{~0u, ~0u, ~0u, ~0u}} : U4{{0u, 0u, 0u, 0u}});
  }
  __threadfence_block();  // (other lanes write into these slots below)
  // ---- downward: rates (radiative, collisional de-excitation, internal down), then the internal-down and radiative filters
  double w_rad = 0., w_col = 0., w_down = 0.;
  double b0_rad = 0., b0_down = 0.;  // block 0's running sums, kept for a direction of one block (no second evaluation of its terms)
  for (int base = 0; base < nd; base += 63) {
    const int nb = (nd - base < 63) ? nd - base : 63;
    const bool valid = lane < nb;
    double v0 = 0., v1 = 0., v2 = 0.;
    if (valid) {
      const MaTransTerms t = matrans_terms(env, c, lpk.alltrans_startdown + base + lane);
      v0 = t.v0; v1 = t.v1; v2 = t.v2;
    }
    double ss[3];
    wave_prefix_inorder_n<3>({v0, v1, v2}, {w_rad, w_col, w_down}, nb, lane, ss);
    const double s0 = ss[0], s1 = ss[1], s2 = ss[2];
    if (base == 0) { b0_rad = s0; b0_down = s2; }
    w_rad = wave_bcast(s0, nb - 1);
    w_col = wave_bcast(s1, nb - 1);
    w_down = wave_bcast(s2, nb - 1);
  }
  if (nd > 0 && nd <= 63) {
    wave_put_dirfilters(rec, lpk, true, nd, lane, lane < nd, b0_down, b0_rad, w_down, w_rad, lane);
  } else {
    double c_rad = 0., c_down = 0.;
    for (int base = 0; base < nd; base += 63) {
      const int nb = (nd - base < 63) ? nd - base : 63;
      const bool valid = lane < nb;
      double v0 = 0., v2 = 0.;
      if (valid) {
        const MaTransTerms t = matrans_terms(env, c, lpk.alltrans_startdown + base + lane);
        v0 = t.v0; v2 = t.v2;
      }
      double ss[2];
      wave_prefix_inorder_n<2>({v0, v2}, {c_rad, c_down}, nb, lane, ss);
      const double s0 = ss[0], s2 = ss[1];
      wave_put_dirfilters(rec, lpk, true, nd, base + lane, valid, s2, s0, w_down, w_rad, lane);
      c_rad = wave_bcast(s0, nb - 1);
      c_down = wave_bcast(s2, nb - 1);
    }
  }
  // ---- upward: the internal-up rate and filter, and the level's collisional-excitation cooling filter (running sums from the
  // cooling list's value before the level; populate_coolfilter_level_seq)
  const int hi_i = M.level_coolhi[ul];
  const double *cool = env.K.cooling_contrib + (krow(env, c) * M.ncoolingterms);
  const double c_hi = (nu > 0 && hi_i >= 0) ? cool[hi_i] : 0.;
  const double c_lo = (nu > 0 && hi_i > M.ion_coolingoffset[M.level_ion[ul]]) ? cool[hi_i - 1] : 0.;
  const double span = c_hi - c_lo;
  double w_up = 0.;
  double b0_up = 0., b0_kt = 0.;
  for (int base = 0; base < nu; base += 63) {
    const int nb = (nu - base < 63) ? nu - base : 63;
    double v0 = 0., kt = 0.;
    if (lane < nb) {
      const MaTransTerms t = matrans_terms(env, c, lpk.alltrans_startdown + nd + base + lane);
      v0 = t.v0; kt = t.kterm;
    }
    double ss[1];
    wave_prefix_inorder_n<1>({v0}, {w_up}, nb, lane, ss);
    const double s0 = ss[0];
    if (base == 0) { b0_up = s0; b0_kt = kt; }
    w_up = wave_bcast(s0, nb - 1);
  }
  {
    double c_up = 0., c_cool = c_lo;
    for (int base = 0; base < nu; base += 63) {
      const int nb = (nu - base < 63) ? nu - base : 63;
      const bool valid = lane < nb;
      const int ti = base + lane;
      double v0 = 0., kt = b0_kt;
      if (valid && nu > 63) {  // (a direction of one block keeps its terms from the pass above)
        const MaTransTerms t = matrans_terms(env, c, lpk.alltrans_startdown + nd + base + lane);
        v0 = t.v0; kt = t.kterm;
      }
      // (the internal-up sums again -- a direction of one block keeps them from the pass above -- and the cooling terms' sums from the list's
      // value before the level, side by side)
      double ss[2];
      wave_prefix_inorder_n<2>({v0, kt}, {c_up, c_cool}, nb, lane, ss);
      const double s0 = (nu <= 63) ? b0_up : ss[0];
      wave_put_dirfilters(rec, lpk, false, nu, ti, valid, s0, 0., w_up, 0., lane);
      c_up = wave_bcast(s0, nb - 1);
      if (hi_i >= 0) {
        const double sk = ss[1];
        bool ok = (span > 0.) && (span <= DBLMAX);
        uint32_t q = MAFILT_NONE;
        if (valid && ti < nu - 1 && ok) q = mafilt_quant(sk - c_lo, span, &ok);
        const unsigned long long bad = __ballot(valid && !ok);
        if (valid) {
          const bool lok = (bad & line_lanes(lane, ti, nu)) == 0ull;
          U4 *line = rec + marec_slot(MADIR_COOL, ti / MAREC_PER, nd, nu);
          if (lok) {
            mafilt_put(line, ti % MAREC_PER, q);
          } else if (ti % MAREC_PER == 0) {
            *line = U4{{0u, 0u, 0u, 0u}};  // (populate_coolfilter_line: a line that is not usable is all zero, its mark too)
          }
        }
        c_cool = wave_bcast(sk, nb - 1);
      }
    }
  }
  __threadfence_block();
  if (lane == 0) {
    double *rates = ma_rates_of(rec, nd, nu);
    rates[ARTIS_MA_ACTION_RADDEEXC] = w_rad;
    rates[ARTIS_MA_ACTION_COLDEEXC] = w_col;
    rates[ARTIS_MA_ACTION_INTERNALDOWNSAME] = w_down;
    rates[ARTIS_MA_ACTION_INTERNALUPSAME] = w_up;
    populate_macroatom<false>(env, c, ul);  // the bound-free channels and the action filter
  }
}
// every lane of the wave whose packet waits for a record it has claimed (physics.h ma_slow_fill_claim) gets it filled, lane by lane
__device__ inline void ma_fill_wave(const Env &env, bool mine, int c, int ul) {
  unsigned long long m = __ballot(mine);
  while (m != 0) {
    const int src = __ffsll((long long)m) - 1;
    ma_fill_record_wave(env, __builtin_amdgcn_readlane(c, src), __builtin_amdgcn_readlane(ul, src));
    m &= m - 1;
  }
}
// test / debug view of one cell's records (artis_amd_debug_cellcache): a thread per level
__global__ void __launch_bounds__(BLOCK) k_debug_macache(Env env, int c, double *maprocessrates, double *matrans, int32_t *bad) {
  const int ul = blockIdx.x * BLOCK + threadIdx.x;
  if (ul >= env.M.nlevels) return;
  if (ma_resolve(env, c, env.M.level_pack[ul].rec_off) == MA_REC_NONE) return;  // (a cold level no packet has reached in this cell: no record)
  const int b = debug_level_record(env, c, ul, maprocessrates, matrans);
  if (b != 0) atomicAdd(bad, b);
}
// before a fill: no cold level of the cells to be filled has a record (tables.h "ON-DEMAND RECORDS"; the pool itself is emptied by populate_tile())
__global__ void __launch_bounds__(BLOCK) k_ma_reset(Env env) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int ncold = env.M.ncold;
  if (i >= fill_count(env) * ncold) return;
  const int c = fill_cell(env, i / ncold);
  env.K.ma_rowtab[(krow(env, c) * ncold) + (i % ncold)] = -1;
}
// the static part of every record of every resident row, once per engine: filter entries "never counted", lines usable
__global__ void __launch_bounds__(BLOCK) k_mainit(Env env, int64_t nrows) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= nrows * env.M.nlevels) return;
  populate_mainit(env, i / env.M.nlevels, (int)(i % env.M.nlevels));
}
// The cooling list of a (cell, ion) is one running sum over hundreds of terms (kpkt.cc:57-190): a free-free term, the
// collisional-excitation terms k_matrans left in the population's scratch rows (most of them), and the bound-free tail. Three kernels:
// head and tail with a lane per (cell, ion) -- 64 chains side by side, each short -- and the long middle with a ROW OF
// 16 LANES per (cell, ion): it reads 16 terms (one cache line) at a time, forms the running sum with the sequential
// additions of the reference's loop (lane k adds its term to lane k-1's finished sum: DPP row shifts, same order, same
// bits), writes it back and drops the level totals into the cooling list through the static slot table. (A lane per
// chain walked the terms with strided 8-byte accesses: 35 ms per step for the three parts, now 15.) The running sum
// travels between the kernels in ion_cooling_C.
__global__ void __launch_bounds__(BLOCK) k_cooling_head(Env env) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t total = fill_count(env) * env.M.nions;
  if (i >= total) return;
  const int c = fill_cell(env, i / env.M.nions);
  const int ui = (int)(i % env.M.nions);
  int k = 0;
  env.K.ion_cooling_C[(krow(env, c) * env.M.nions) + ui] = cooling_ion_head(env, c, ui, &k);
}
// the value of the lane below within a row of 16 lanes (the row's first lane: 0)
__device__ inline double row_shr1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x111 /* row_shr:1 */, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// Four chains per wave, one per row of 16 lanes: a row reads 16 terms (one 128-byte line) at a time and needs 15 serial
// steps for them; four independent chains keep the wave's issue slots four times as busy as one chain of 64 would.
__global__ void __launch_bounds__(BLOCK) k_cooling_chain(Env env) {
  const int64_t row_id = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 4;
  const int r = threadIdx.x & 15;
  const DevModel &M = env.M;
  const int64_t nchains = fill_count(env) * M.nions;
  const bool valid = row_id < nchains;
  const int c = fill_cell(env, (valid ? row_id : 0) / M.nions);
  const int ui = (int)((valid ? row_id : 0) % M.nions);
  const int start = M.ion_uniquelevelindexstart[ui];
  const int nlevels = M.ion_nlevels[ui];
  const int j0 = (valid && nlevels > 0) ? M.level_upcum_start[start] : 0;
  const int j1 = (valid && nlevels > 0) ? M.level_upcum_start[start + nlevels - 1] + M.level_nuptrans[start + nlevels - 1] : 0;
  double carry = valid ? env.K.ion_cooling_C[(krow(env, c) * M.nions) + ui] : 0.;
  double *upcum = env.collexc_terms + (((valid ? row_id : 0) / M.nions) * M.nupcum);  // the cell's row of the population's scratch
  double *cool = env.K.cooling_contrib + (krow(env, c) * M.ncoolingterms);
  for (int j = j0; __any(j < j1); j += 16) {  // rows with shorter chains idle through the longer ones' chunks
    const bool in = (j + r) < j1;
    const double x = in ? upcum[j + r] : 0.;
    double acc = (r == 0) ? carry + x : x;
#pragma unroll
    for (int s = 1; s < 16; s++) {
      const double prev = row_shr1(acc);  // lane s-1 of the row holds its finished sum
      if (r == s) acc = prev + x;
    }
    if (in) {
      upcum[j + r] = acc;
      const int slot = M.upcum_coolslot[j + r];
      if (slot >= 0) cool[slot] = acc;
    }
    // the sum after the row's last term, broadcast to the row (lanes past the end added 0. to it: x + 0. == x exactly)
    const int src = ((threadIdx.x & 63) | 15) << 2;
    carry = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(acc)), __builtin_amdgcn_ds_bpermute(src, __double2loint(acc)));
  }
  if (valid && r == 0 && j1 > j0) env.K.ion_cooling_C[(krow(env, c) * M.nions) + ui] = carry;
}
// the cooling filters of the levels' records from the running sums the chain left in the scratch rows: a thread per (cell, line)
__global__ void __launch_bounds__(BLOCK) k_collexc_filter(Env env) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t per = env.M.ncoollines;
  if (i >= fill_count(env) * per) return;
  const int64_t kf = i / per;
  populate_coolfilter_line(env, fill_cell(env, kf), (int)(i % per), env.collexc_terms + (kf * env.M.nupcum));
}
// The bound-free tail of an ion's cooling list (kpkt.cc:122-190: a collisional-ionisation term and then a bound-free term per (ionising
// level, target), ~120 entries, each behind three or four dependent gathers): a lane per (cell, ion) walked it serially (11 ms per step).
// Round 4: a ROW OF 16 LANES per (cell, ion), like k_cooling_chain -- every lane forms the term of one entry (what the entry is it reads
// from the cooling list's own static tables, checked against the loop order when the engine is created), the running sum is formed with the
// loop's additions in the loop's order (DPP row shifts) and written back 16 entries = one 128-byte line at a time.
// physics.h cooling_ion_tail() is the sequential form (test emulation).
__global__ void __launch_bounds__(BLOCK) k_cooling_tail(Env env) {
  const int64_t row_id = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 4;
  const int r = threadIdx.x & 15;
  const DevModel &M = env.M;
  const int64_t nrows = fill_count(env) * M.nions;
  const bool valid = row_id < nrows;
  const int c = fill_cell(env, (valid ? row_id : 0) / M.nions);
  const int ui = (int)((valid ? row_id : 0) % M.nions);
  const int element = M.ion_element[ui];
  const int ion = ui - M.elem_uniqueionindexstart[element];
  const bool has = valid && ion < (M.elem_nions[element] - 1) && M.nbfcontinua > 0;
  const double *pops = env.K.levelpops + (krow(env, c) * M.nlevels);
  const int ionstart = M.ion_coolingoffset[ui];
  double *contribs = env.K.cooling_contrib + (krow(env, c) * M.ncoolingterms) + ionstart;
  const float cnne = clumpednne(env.C, c);
  const float T_e = env.C.Te[c];
  const int start = M.ion_uniquelevelindexstart[ui];
  const int ustart = has ? M.ion_uniquelevelindexstart[ui + 1] : 0;
  const double nnupperion = has ? nnion(env, c, element, ion + 1) : 0.;
  const int k0 = has ? M.ion_cooltail_start[ui] : 0, k1 = has ? M.ion_ncoolingterms[ui] : 0;
  double carry = valid ? env.K.ion_cooling_C[(krow(env, c) * M.nions) + ui] : 0.;
  for (int j = k0; __any(j < k1); j += 16) {
    const bool in = (j + r) < k1;
    double x = 0.;
    if (in) {
      const int i = ionstart + j + r;
      const int level = M.coolinglist_level[i];
      const int t = M.coolinglist_phixstargetindex[i];
      const int ul = start + level;
      const int64_t o = (krow(env, c) * M.nphixstargets_total) + M.level_phixstargetstart[ul];
      if (M.coolinglist_type[i] == ARTIS_COOLING_COLLION) {
        const double e_trans = eps(M, ustart + phixs_upperlevel(M, ul, t)) - eps(M, ul);
        x = pops[ul] * env.K.bf_colion[o + t] * e_trans;
      } else {
        double pop;
#if ARTIS_OPT_BFCOOLING_USELEVELPOPNOTIONPOP
        pop = pops[ustart + phixs_upperlevel(M, ul, t)];
#else
        const int nt = M.level_nphixstargets[ul];
        if (nt == 1) {
          pop = nnupperion;
        } else {
          double E_min = DBLMAX;
          for (int tt = 0; tt < nt; tt++) E_min = dmin(E_min, eps(M, ustart + phixs_upperlevel(M, ul, tt)));
          double wsum = 0.;
          for (int tt = 0; tt < nt; tt++) {
            const int up = phixs_upperlevel(M, ul, tt);
            wsum += statw(M, ustart + up) * exp(-(eps(M, ustart + up) - E_min) / KB / T_e);
          }
          const int up = phixs_upperlevel(M, ul, t);
          const double w = statw(M, ustart + up) * exp(-(eps(M, ustart + up) - E_min) / KB / T_e);
          pop = nnupperion * w / wsum;
        }
#endif
        x = env.K.bf_cooling[o + t] * pop * cnne;
      }
    }
    double acc = (r == 0) ? carry + x : x;
#pragma unroll
    for (int s = 1; s < 16; s++) {
      const double prev = row_shr1(acc);
      if (r == s) acc = prev + x;
    }
    if (in) contribs[j + r] = acc;
    const int src = ((threadIdx.x & 63) | 15) << 2;
    carry = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(acc)), __builtin_amdgcn_ds_bpermute(src, __double2loint(acc)));
  }
  if (has && r == 0) env.K.ion_cooling_C[(krow(env, c) * M.nions) + ui] = carry;
}
__global__ void __launch_bounds__(BLOCK) k_cooling_prefix(Env env) {
  const int64_t kf = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (kf >= fill_count(env)) return;
  const int c = fill_cell(env, kf);
  populate_cooling_prefix(env, c);
}
// the guides of the two cumulative lists a k-packet step draws from (tables.h "COOLING GUIDES"): a thread per entry
__global__ void __launch_bounds__(BLOCK) k_cool_guide(Env env) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t total = fill_count(env) * env.M.nguide;
  if (i >= total) return;
  populate_cool_guide(env, fill_cell(env, i / env.M.nguide), (int)(i % env.M.nguide));
}

// ------------------------------------------------------------------ packet layout kernels
// The slot of a packet is not its index in the caller's array: slots are handed out in the order of the packets'
// propagation cells at upload (perm[slot] = index in the caller's array), so that the lanes of a wave -- whose work list
// is sorted by cell -- read and write neighbouring records. Thermal packets never leave their cell, r-packets drift away
// from this order only gradually. perm == nullptr: identity.
__global__ void __launch_bounds__(BLOCK) k_aos_to_rec(const artis_packet *aos, PktStore P, const int32_t *perm) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < P.n) aos_to_rec(aos[perm ? perm[i] : i], P, i);
}
__global__ void __launch_bounds__(BLOCK) k_rec_to_aos(PktStore P, artis_packet *aos, const int32_t *perm) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < P.n) rec_to_aos(P, i, aos[perm ? perm[i] : i]);
}
__global__ void __launch_bounds__(BLOCK) k_aos_cellkeys(const artis_packet *aos, int64_t n, int32_t ngrid, int32_t *ident, int32_t *keys) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  ident[i] = (int32_t)i;
  const int32_t c = aos[i].cellindex;
  keys[i] = (c >= 0 && c < ngrid) ? c : 0;  // a packet of a type this path does not own may carry any cell index
}

// append (pi, key) of every lane with flag set to list[] / keys[], one atomic per wave (wave-ballot compaction)
// (Round 6, measured and removed: the list's key histogram kept HERE -- one fire-and-forget atomic per appended entry -- instead of by
// k_sort_hist before the sort. Headline step 759.2 ms against 759.9: the 2.7e8 extra atomics per step cost inside the propagation kernels what
// the 14 ms of histogram kernels cost outside them; and one run of the kilonova_expopac build at 50^3 / 1e7 gave different counters from the
// same snapshot with it. profiles/r06/fused_histogram.md)
__device__ inline void wave_append(bool flag, int32_t pi, int32_t key, int32_t *list, int32_t *keys, int32_t *count) {
  const unsigned long long mask = __ballot(flag);
  if (mask == 0) return;
  const int lane = threadIdx.x & 63;
  const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
  int base = 0;
  const int leader = __ffsll((long long)mask) - 1;
  if (lane == leader) base = atomicAdd(count, __popcll(mask));
  base = __shfl(base, leader);
  if (flag) {
    list[base + prefix] = pi;
    keys[base + prefix] = key;
  }
}

// Blocks are dealt round-robin over the 8 XCDs (each with its own L2). The work lists are sorted by cell, so give
// every XCD one contiguous eighth of the list: the per-cell tables its waves read then stay in that XCD's L2.
// (Bijective remap of blockIdx -> list chunk; placement only changes speed, never results.)
__device__ inline int64_t xcd_chunk(int64_t b, int64_t nb) {
  const int64_t q = nb / 8, r = nb % 8, xcd = b % 8;
  return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
}

// Work lists, one per kind of pending work (physics.h NEXT_*). A kernel consumes the whole current list of ITS kind and
// appends packets to the current lists of the other kinds; packets that stay with the kernel's own kind (launch budget
// used up) go to that kind's alternate list, which becomes its current list afterwards. Every entry carries its sort
// key (physics.h list_sort_key), written by the lane that still has the packet in registers, so that sorting a list
// never touches the packet records.
struct Lists {
  int32_t *lst[NEXT_NKINDS];  // current list of each kind
  int32_t *key[NEXT_NKINDS];  // ... and the sort keys of its entries
  int32_t *counts;            // [NEXT_NKINDS] fill counts of the current lists
  int32_t self_kind;          // kind of the running kernel
  int32_t *self_list;         // its alternate list
  int32_t *self_key;
  int32_t *self_count;
  int32_t kpkt_slot;          // list that takes k-packets: NEXT_KPKT, or NEXT_MA when k-packets and macro-atoms share the
                              // fused thermal kernel
  int32_t nubins;             // frequency bins of the r-packet list's keys (1 = sort by cell only)
  int32_t mabins;             // sub-keys of the thermal list's keys (1 = sort by cell only)
  int32_t numajor;            // > 0 = the cell groups of the grid: r-packet keys with the frequency bin as the MAJOR part (ARTIS_AMD_SORT_NUMAJOR=0: cell-major)
  int32_t cellshift;          // ... groups of 2^cellshift cells with neighbouring indices share a key (ARTIS_AMD_SORT_CELLSHIFT)
};
// ma_sub: (tuning, ARTIS_AMD_MABINS=16) a sub-key 0..15 of a thermal-list entry below its cell, -1: none
__device__ inline void append_by_kind(int kind, int32_t pi, int32_t cellindex, double nu_cmf, const Lists &L, int ma_sub = -1) {
  const int slot = (kind == NEXT_KPKT) ? L.kpkt_slot : kind;
  int32_t key = list_sort_key(cellindex, nu_cmf, (slot == NEXT_RPKT) ? L.nubins : 1);
  if (slot == NEXT_RPKT && L.numajor > 0 && L.nubins > 1) key = ((key % SORT_NUBINS) * L.numajor) + ((key / SORT_NUBINS) >> L.cellshift);
  if (slot == NEXT_MA && L.mabins > 1)
    key = (cellindex * SORT_MABINS) + ((kind == NEXT_KPKT || ma_sub < 0) ? SORT_MABINS - 1 : (ma_sub & (SORT_MABINS - 1)));
#pragma unroll
  for (int k = 1; k < NEXT_NKINDS; k++) {
    int32_t *dst = (k == L.self_kind) ? L.self_list : L.lst[k];
    int32_t *dkey = (k == L.self_kind) ? L.self_key : L.key[k];
    int32_t *cnt = (k == L.self_kind) ? L.self_count : (L.counts + k);
    wave_append(slot == k, pi, key, dst, dkey, cnt);
  }
}
// start of update_packets(): every resident packet is put on the list of its kind. A ContinuumOpacity never survives
// into another call (rpkt.cc:1023 compares globals::timestep; the cell state may have changed in between).
// (Called once per cell-cache tile: only the packets whose cell is in the resident tile are listed, physics.h classify().)
__global__ void __launch_bounds__(BLOCK) k_classify(Env env, Lists L, int reset_chi) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  int kind = NEXT_DONE;
  int32_t cellindex = 0;
  double nu_cmf = 0.;
  if (i < env.P.n) {
    PktHot &h = env.P.hot[i];
    const int type = h.type;
    cellindex = h.cellindex;
    nu_cmf = h.nu_cmf;
    if (reset_chi && h.chi_mgi >= 0) h.chi_mgi = -1;
    const bool active = type_handled(type) && h.prop_time < env.S.ts_end;
    const bool waiting = h.pend != PEND_NONE || h.ma_level >= 0;
    if (active && type_gamma(type) && !waiting) {
      kind = NEXT_GAMMA;
    } else if ((waiting || active) && in_tile(env, cellindex)) {
      if (h.pend != PEND_NONE) {
        kind = NEXT_SLOW;
      } else if (h.ma_level >= 0) {
        kind = NEXT_MA;
      } else if (type == ARTIS_TYPE_RPKT) {
        kind = NEXT_RPKT;
      } else {
        kind = kpkt_blackbody_case(env, type, cellindex) ? NEXT_BB : NEXT_KPKT;
      }
    }
  }
  append_by_kind(kind, (int32_t)i, cellindex, nu_cmf, L);
}

// Adaptive tiles (round 6): how many packets wait in every non-empty cell (the predicate of k_classify, whatever tile is resident), so
// that the next tile can be put where most of them are; *other: packets that need no cache row (gamma-ray kinds, packets in empty cells)
__global__ void __launch_bounds__(BLOCK) k_count_waiting(Env env, int32_t *counts, int32_t *other) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= env.P.n) return;
  const PktHot &h = env.P.hot[i];
  const int type = h.type;
  const bool active = type_handled(type) && h.prop_time < env.S.ts_end;
  const bool waiting = h.pend != PEND_NONE || h.ma_level >= 0;
  if (!(waiting || active)) return;
  const int c = env.M.propcell_nonemptymgi[h.cellindex];
  if ((active && type_gamma(type) && !waiting) || c < 0)
    atomicAdd(other, 1);
  else
    atomicAdd(&counts[c], 1);
}
// ---- counting sort of a work list by its entries' keys (propagation cell, frequency bin): three tiny kernels.
// Within a cell, r-packets are ordered by comoving frequency like the reference's own packet sort
// (compare_packet_order, update_packets.cc:363): neighbouring lanes then walk the same part of the line list and
// the same window of bound-free continua.
__global__ void __launch_bounds__(BLOCK) k_sort_hist(const int32_t *keys, int32_t n, int32_t *hist) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) atomicAdd(&hist[keys[i]], 1);
}
// exclusive scan of the key histogram in three steps: per-block scan of SCAN_TILE keys, scan of the block totals
// (one block), add the block offsets
constexpr int SCAN_TILE = 8192;  // keys per block of 1024 threads (8 per thread)
__global__ void __launch_bounds__(1024) k_scan_tiles(int32_t *hist, int32_t nkeys, int32_t *tile_totals) {
  __shared__ int32_t part[1024];
  const int t = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)t * 8;
  int32_t v[8];
  int32_t sum = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    v[k] = (base + k < nkeys) ? hist[base + k] : 0;
    sum += v[k];
  }
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan of the per-thread sums
    const int32_t add = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  int32_t run = part[t] - sum;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (base + k < nkeys) hist[base + k] = run;
    run += v[k];
  }
  if (t == 1023) tile_totals[blockIdx.x] = part[1023];
}
__global__ void __launch_bounds__(1024) k_scan_totals(int32_t *tile_totals, int32_t ntiles) {
  __shared__ int32_t part[1024];
  const int t = threadIdx.x;
  int32_t carry = 0;
  for (int start = 0; start < ntiles; start += 1024) {
    const int i = start + t;
    const int32_t v = (i < ntiles) ? tile_totals[i] : 0;
    part[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int32_t add = (t >= off) ? part[t - off] : 0;
      __syncthreads();
      part[t] += add;
      __syncthreads();
    }
    if (i < ntiles) tile_totals[i] = carry + part[t] - v;
    carry += part[1023];
    __syncthreads();
  }
}
__global__ void __launch_bounds__(1024) k_scan_add(int32_t *hist, int32_t nkeys, const int32_t *tile_totals) {
  const int32_t off = tile_totals[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 8;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (base + k < nkeys) hist[base + k] += off;
}
__global__ void __launch_bounds__(BLOCK) k_sort_scatter(const int32_t *list, const int32_t *keys, int32_t n, int32_t *offsets, int32_t *out) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) out[atomicAdd(&offsets[keys[i]], 1)] = list[i];
}
// The same two steps for FEW keys (1D and 2D models, small 3D grids: a few hundred cells). With one global atomic per
// entry the entries of a cell serialise on its counter (measured on a 6^3 grid with 1e7 packets: 7 ms per sort kernel,
// 36 % of the GPU time of a step, against 0.2 ms on the 50^3 grid). Here a workgroup counts its contiguous chunk of the
// list in LDS and touches each global counter once.
constexpr int SORT_LDS_KEYS = 8192;
constexpr int SORT_LDS_GRID = 2048;
__global__ void __launch_bounds__(BLOCK) k_sort_hist_lds(const int32_t *keys, int32_t n, int32_t *hist, int32_t nkeys) {
  __shared__ int32_t h[SORT_LDS_KEYS];
  for (int k = threadIdx.x; k < nkeys; k += BLOCK) h[k] = 0;
  __syncthreads();
  const int64_t chunk = ((int64_t)n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = chunk * blockIdx.x, hi = (lo + chunk < n) ? lo + chunk : n;
  for (int64_t i = lo + threadIdx.x; i < hi; i += BLOCK) atomicAdd(&h[keys[i]], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < nkeys; k += BLOCK)
    if (h[k] != 0) atomicAdd(&hist[k], h[k]);
}
__global__ void __launch_bounds__(BLOCK) k_sort_scatter_lds(const int32_t *list, const int32_t *keys, int32_t n, int32_t *offsets, int32_t *out,
                                                            int32_t nkeys) {
  __shared__ int32_t h[SORT_LDS_KEYS];
  for (int k = threadIdx.x; k < nkeys; k += BLOCK) h[k] = 0;
  __syncthreads();
  const int64_t chunk = ((int64_t)n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = chunk * blockIdx.x, hi = (lo + chunk < n) ? lo + chunk : n;
  for (int64_t i = lo + threadIdx.x; i < hi; i += BLOCK) atomicAdd(&h[keys[i]], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < nkeys; k += BLOCK)  // the workgroup's range of each key's output
    if (h[k] != 0) h[k] = atomicAdd(&offsets[k], h[k]);
  __syncthreads();
  for (int64_t i = lo + threadIdx.x; i < hi; i += BLOCK) out[atomicAdd(&h[keys[i]], 1)] = list[i];
}
inline int sort_lds_grid(int64_t n) {
  const int64_t b = (n + BLOCK - 1) / BLOCK;
  return (int)(b < SORT_LDS_GRID ? b : SORT_LDS_GRID);
}

// ------------------------------------------------------------------ the propagation kernels
// They are PERSISTENT, work-pulling kernels: a lane that has finished with its packet (budget used up, packet handed
// to another list, escaped, end of timestep) immediately takes the next packet of the work list instead of idling
// until the slowest lane of its wave is done. The list is sorted by cell and cut into `nchunks` contiguous chunks with
// one cursor each -- one chunk per WAVE when the list is long enough: a wave then walks its own run of cells, so at any
// time its lanes sit in one or two cells and read the same cell-cache lines. The chunks of the waves of one XCD are
// adjacent (blocks b and b+8 share an XCD, so each XCD's L2 sees one contiguous range of cells). A wave whose chunk is
// used up steals from the other chunks, its own XCD's first. Placement only affects speed, never results.
#ifndef ARTIS_RPKT_WAVES
#define ARTIS_RPKT_WAVES 2
#endif
#ifndef ARTIS_THERMAL_WAVES
#define ARTIS_THERMAL_WAVES 4
#endif
constexpr int MAX_CHUNKS = 8192;  // >= 256 CUs x 4 waves/SIMD x 4 SIMDs, a multiple of 8
// before a launch of the kernel of one kind: its own list's counter, the alternate counter and the chunk cursors start at zero
__global__ void __launch_bounds__(BLOCK) k_launch_reset(int32_t *count, int kind, int32_t *cursors) {
  for (int i = threadIdx.x; i < MAX_CHUNKS + 1; i += BLOCK) cursors[i] = 0;
  if (threadIdx.x == 0) {
    count[kind] = 0;
    count[NEXT_NKINDS] = 0;
  }
}


struct Puller {
  int chunk;           // current chunk
  int32_t cbeg, cend;  // ... and its range of the list (kept: the bounds cost two 64-bit divisions)
  int home, cx;        // first position of the search order: (XCD, local index); chunks per XCD
  int nchunks;
  bool exhausted;
};
__device__ inline int64_t chunk_begin(int32_t n, int c, int nchunks) { return ((int64_t)n * c) / nchunks; }
// chunk_mode: 0 = one chunk per wave (or fewer, shared in order), 1 = per workgroup, 2 = per COMPUTE UNIT (nchunks == 256):
// the workgroups that the hardware placed on one CU walk one run of cells together, so the CU's L1 sees a handful of
// cells. The CU is read from HW_REG_HW_ID (MI355X: 8 XCDs x 4 shader engines x 8 CUs, CU_ID 0..7 or 1..8).
__device__ inline void puller_init(Puller &q, int32_t n, int nchunks, int chunk_mode = 0) {
  int xcd = blockIdx.x & 7;
  // wave / workgroup / CU index within its XCD
  int local = (chunk_mode == 1) ? (int)(blockIdx.x >> 3) : (int)(blockIdx.x >> 3) * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
  if (chunk_mode == 2) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcd = (int)(xcc & 7u);
    local = (int)(((hwid >> 13) & 3u) * 8u + ((hwid >> 8) & 7u));
  }
  q.nchunks = nchunks;
  q.cx = nchunks >> 3;
  q.home = xcd * q.cx + (local % q.cx);
  q.chunk = q.home;
  q.cbeg = (int32_t)chunk_begin(n, q.chunk, nchunks);
  q.cend = (int32_t)chunk_begin(n, q.chunk + 1, nchunks);
  q.exhausted = false;
}
// position v of a wave's search order -> chunk: its own XCD's chunks first (from its home chunk, wrapping), then the rest
__device__ inline int chunk_at(const Puller &q, int v) {
  const int x0 = (q.home / q.cx) * q.cx;
  if (v < q.cx) return x0 + ((q.home - x0 + v) % q.cx);
  return (x0 + v) % q.nchunks;
}
// Hands list indices to the lanes with need==true. Returns the index for this lane or -1. Wave-uniform control flow.
__device__ inline int32_t pull(Puller &q, bool need, int32_t n, int32_t *cursors) {
  int32_t idx = -1;
  need = need && !q.exhausted;
  const unsigned long long mask = __ballot(need);
  if (mask == 0) return -1;
  const int lane = threadIdx.x & 63;
  const int cnt = __popcll(mask);
  const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
  const int leader = __ffsll((long long)mask) - 1;
  const int64_t cbeg = q.cbeg, cend = q.cend;
  int base = 0;
  if (lane == leader) base = atomicAdd(&cursors[q.chunk], cnt);
  base = __shfl(base, leader);
  const int64_t mine = cbeg + base + prefix;
  if (need && mine < cend) idx = (int32_t)mine;
  if (cbeg + base + cnt > cend) {
    // this chunk is used up: look for one that still has entries, 64 candidates at a time (the unserved lanes ask again).
    // A cursor only grows, so a stale read can only make a chunk look fuller than it is: the next pull finds out.
    bool found = false;
    for (int v0 = 1; v0 < q.nchunks && !found; v0 += 64) {
      const int v = v0 + lane;
      bool has = false;
      int c = 0;
      if (v < q.nchunks) {
        c = chunk_at(q, v);
        const int32_t used = __hip_atomic_load(&cursors[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        has = chunk_begin(n, c, q.nchunks) + used < chunk_begin(n, c + 1, q.nchunks);
      }
      const unsigned long long hm = __ballot(has);
      if (hm != 0) {
        q.chunk = __shfl(c, __ffsll((long long)hm) - 1);
        q.cbeg = (int32_t)chunk_begin(n, q.chunk, q.nchunks);
        q.cend = (int32_t)chunk_begin(n, q.chunk + 1, q.nchunks);
        found = true;
      }
    }
    if (!found) q.exhausted = true;
  }
  return idx;
}
// chunks for a list of n entries consumed by `nwaves` waves: one per wave, fewer when the list is short
inline int chunks_for(int64_t n, int nwaves) {
  int64_t c = std::min<int64_t>(nwaves, n / 128);
  c = std::max<int64_t>(8, (c / 8) * 8);
  return (int)std::min<int64_t>(c, MAX_CHUNKS);
}

// r-packets in flight: boundary distance, continuum opacity, line-by-line Sobolev walk, estimators, events.
// CONT_LDS: the static table of bound-free continua (ContPack: edge frequency, target probability, cross-section table,
// ground-continuum index; 32 B per continuum) is copied into LDS once per workgroup and every read of it in the opacity
// sum (rpkt.cc:721, two of the ~five reads per continuum visited) is a ds_read that does not touch the vector L1.
constexpr int CONT_LDS_MAX = 2048;  // continua (64 KB per workgroup, two workgroups per CU)
// LDS accumulators of per-cell estimators for models with few cells (physics.h Env::cellest_lds)
constexpr int RPKT_CELLEST_CAP = 512;      // cells: 3 estimators x 8 B x 512 = 12 KB per workgroup (next to the 64 KB above)
constexpr int RPKT_CELLEST_CAP_NOCONT = 3072;  // ... 72 KB in the kernel form without the continuum table in LDS
constexpr int GAMMA_CELLEST_CAP = 2048;    // cells: 16 KB per workgroup
constexpr int THERMAL_CELLEST_CAP = 4096;  // cells: 32 KB per workgroup, four workgroups per CU
__device__ inline void cellest_begin(Env &env, double *lds, int n, int nthreads, const double *a0, const double *a1 = nullptr,
                                     const double *a2 = nullptr) {
  const int nkinds = a2 ? 3 : (a1 ? 2 : 1);
  for (int i = threadIdx.x; i < n * nkinds; i += nthreads) lds[i] = 0.;
  env.cellest_lds = lds;
  env.cellest_n = n;
  env.cellest_owner[0] = a0;
  env.cellest_owner[1] = a1;
  env.cellest_owner[2] = a2;
}
__device__ inline void scalars_begin(Env &env, double *lds) {
  if (threadIdx.x < ARTIS_NSCALARS) lds[threadIdx.x] = 0.;
  env.scalars_lds = (env.scalars_in_lds && env.E.scalars != nullptr) ? lds : nullptr;
}
__device__ inline void scalars_flush(const Env &env) {  // after a __syncthreads()
  if (env.scalars_lds != nullptr && threadIdx.x < ARTIS_NSCALARS && env.scalars_lds[threadIdx.x] != 0.)
    unsafeAtomicAdd(&env.E.scalars[threadIdx.x], env.scalars_lds[threadIdx.x]);
}
// call after a __syncthreads(): the workgroup's sums of one estimator go to the global array
__device__ inline void cellest_flush(const Env &env, int kind, double *global_array, int nthreads) {
  for (int c = threadIdx.x; c < env.cellest_n; c += nthreads) {
    const double v = env.cellest_lds[(kind * env.cellest_n) + c];
    if (v != 0.) unsafeAtomicAdd(&global_array[(int64_t)c * env.est_stride], v);
  }
}
// One workgroup of 768 threads per CU: 3 waves/SIMD at 168 VGPRs (the LDS tables allow two workgroups per CU, so 256-thread
// workgroups stop at 2 waves/SIMD whatever their registers). Measured against 256 x 2 (2 waves/SIMD, 256 VGPRs): k_rpkt
// 352 -> 339 ms (classic), 398 -> 379 (kilonova_lte), k_rpkt + k_bfest_dense 825 -> 781 (nltenebular); 1024 x 1 (4
// waves/SIMD, 128 VGPRs, 768 B of scratch): 457 ms.
#ifndef ARTIS_RPKT_TB
#define ARTIS_RPKT_TB 768  // threads per workgroup of k_rpkt ...
#define ARTIS_RPKT_WGS 1   // ... and workgroups per CU
#endif
// LINE_LDS (ARTIS_AMD_LINELDS=1; instead of the continuum table, which it leaves no room for): the line list's frequencies
// (globals::linelist nu, rpkt.cc:106-207 get_possible_event: one of the two reads per line visited, the other being the cell's
// population factor) in LDS, the whole list when it has at most LINE_LDS_MAX lines. What a per-cell, per-wave WINDOW of
// {frequency, population factor} pairs would need -- many lanes of a wave in one cell and one stretch of the list -- the
// cell-sorted work list does not give: a wave's 64 packets sit in ~21 cells (2.6 lanes per cell) and
// anywhere in the spectrum. Measured slower than the continuum table in LDS: profiles/r04/line_window.md.
constexpr int LINE_LDS_MAX = 14336;  // lines (112 KB)
#ifndef ARTIS_RPKT_SPLIT_ABSORB
#define ARTIS_RPKT_SPLIT_ABSORB 0  // (1: measured round 5: k_rpkt 265 -> 258 ms, k_slow 16 -> 32 ms, step unchanged, scratch unchanged: off) free-free / bound-free absorptions of r-packets carried out by the slow-path kernel (physics.h PEND_RPKT_ABSORB)
#endif
template <bool CONT_LDS, int TB, bool LINE_LDS = false>
__global__ void __launch_bounds__(TB, ARTIS_RPKT_WGS) k_rpkt(Env env, const int32_t *list, int32_t n, Lists next,
                                                                   unsigned long long *gstats, int budget, int32_t *cursors, int nchunks,
                                                                   int drain_budget) {
  __shared__ stat_t lstats[ARTIS_NSTATS];
  __shared__ ContPack lds_cont[CONT_LDS ? CONT_LDS_MAX : 1];
  __shared__ double lds_line_nu[LINE_LDS ? LINE_LDS_MAX : 1];
  __shared__ double lds_cellest[3 * ((CONT_LDS || LINE_LDS) ? RPKT_CELLEST_CAP : RPKT_CELLEST_CAP_NOCONT)];
  constexpr int NEC = LINE_LDS ? 1 : ESTCACHE_SLOTS;       // (the line list in LDS leaves no room for the caches)
  __shared__ double lds_estcache[(TB / 64) * NEC * 3];   // per wave: the accumulators of ESTCACHE_SLOTS cells (physics.h Env::estcache) ...
  __shared__ int32_t lds_esttag[(TB / 64) * NEC * 2];    // ... whose cells they are, and the claims of an eviction
  if (threadIdx.x < ARTIS_NSTATS) lstats[threadIdx.x] = 0;
  cellest_begin(env, lds_cellest, env.cellest_n_r, TB, env.E.J, env.E.nuJ, env.E.ffheatingestimator);
  env.estcache = nullptr;
  env.estcache_nv = 0;
  if (!LINE_LDS && env.estcache_on && env.cellest_n_r == 0) {  // many cells: the wave's own cache instead of the workgroup's array
    const int w = threadIdx.x >> 6;
    env.estcache_nv = 3;
    env.estcache = lds_estcache + (w * NEC * 3);
    env.estcache_tag = lds_esttag + (w * NEC * 2);
    for (int l = threadIdx.x & 63; l < NEC; l += 64) {
      env.estcache_tag[l] = -1;
      env.estcache_tag[NEC + l] = -1;
      env.estcache[(l * 3) + 0] = 0.;
      env.estcache[(l * 3) + 1] = 0.;
      env.estcache[(l * 3) + 2] = 0.;
    }
  }
  if (LINE_LDS) {
    for (int i = threadIdx.x; i < env.M.nlines; i += TB) lds_line_nu[i] = env.M.line_nu[i];
    env.M.line_nu = lds_line_nu;
  }
  if (CONT_LDS) {
    const D2 *src = (const D2 *)env.M.cont_pack;
    D2 *dst = (D2 *)lds_cont;
    for (int i = threadIdx.x; i < env.M.nbfcontinua * 2; i += TB) dst[i] = src[i];
    env.M.cont_pack = lds_cont;
    env.cont_in_lds = 1;
  }
  __syncthreads();
  env.stats = lstats;
  const double ts_end = env.S.ts_end;
  Puller q;
  puller_init(q, n, nchunks);
  bool have = false;
  bool drained = false;  // the launch's work list is used up (k_thermal: same hand-over to the next launch)
  int32_t pi = 0;
  int steps = 0;
  Pkt p;
  Chi x;
#ifdef ARTIS_PROFILE
  long long tprev = clock64();
#endif
  while (true) {
    const int32_t idx = pull(q, !have, n, cursors);
    if (idx >= 0) {
      pi = list[idx];
      pkt_load(env.P, pi, p);
      chi_load(env.P, pi, p, x);
      steps = 0;
      have = true;
    }
    if (!drained && q.exhausted) {
      drained = true;
      if ((threadIdx.x & 63) == 0) __hip_atomic_store(&cursors[MAX_CHUNKS], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!drained && drain_budget < budget && (steps & 1) == 0)
      drained = __hip_atomic_load(&cursors[MAX_CHUNKS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (!__any(have)) {
      if (q.exhausted) break;
      continue;
    }
    int kind = NEXT_DONE;
    int32_t out_pi = 0;
#ifdef ARTIS_PROFILE
    {  // slot 53: everything outside do_rpkt_step (pull, load, store, append), 54: inside; 55: wave iterations
      const long long now = clock64();
      if ((threadIdx.x & 63) == 0) {
        atomicAdd(&lstats[53], (stat_t)((now - tprev) >> 4));
        atomicAdd(&lstats[55], (stat_t)1);
      }
      tprev = now;
    }
#endif
    if (have) {
      bool go = rpkt_can_continue(p, ts_end);
      if (go) {
        go = rpkt_iter<ARTIS_RPKT_SPLIT_ABSORB != 0>(env, p, pi, x);
        steps++;
      }
      if (!go || steps >= budget || (drained && steps >= drain_budget)) {
        chi_store(env.P, pi, p, x);
        pkt_store(env.P, pi, p);
        kind = classify(env, p, ts_end);
        out_pi = pi;
        have = false;
      }
    }
#ifdef ARTIS_PROFILE
    {
      const long long now = clock64();
      if ((threadIdx.x & 63) == 0) atomicAdd(&lstats[54], (stat_t)((now - tprev) >> 4));
      tprev = now;
    }
#endif
    append_by_kind(kind, out_pi, p.cellindex, p.nu_cmf, next, (p.ma_element * 5 + p.ma_ion));
  }
  __syncthreads();
  if (env.estcache != nullptr) {  // the wave's accumulators go to the cells' records: a lane per slot
    for (int l = threadIdx.x & 63; l < NEC; l += 64) {
      const int cell = env.estcache_tag[l];
      if (cell < 0) continue;
      const double s0 = env.estcache[(l * 3) + 0], s1 = env.estcache[(l * 3) + 1], s2 = env.estcache[(l * 3) + 2];
      if (s0 != 0.) unsafeAtomicAdd(&env.E.J[(int64_t)cell * env.est_stride], s0);
      if (s1 != 0.) unsafeAtomicAdd(&env.E.nuJ[(int64_t)cell * env.est_stride], s1);
      if (s2 != 0.) unsafeAtomicAdd(&env.E.ffheatingestimator[(int64_t)cell * env.est_stride], s2);
    }
  }
  cellest_flush(env, CELLEST_J, env.E.J, TB);
  cellest_flush(env, CELLEST_NUJ, env.E.nuJ, TB);
  cellest_flush(env, CELLEST_FFHEAT, env.E.ffheatingestimator, TB);
  if (threadIdx.x < ARTIS_NSTATS && lstats[threadIdx.x] != 0) atomicAdd(&gstats[threadIdx.x], lstats[threadIdx.x]);
}

#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
// The deferred updates of the detailed bound-free estimators (radfield.cc:215). A record's continua are the kept ones of
// its window [begin, end); LPR lanes share the record (cell, frequency, weight) and take one kept continuum each, so the
// lanes are busy however long the window is (a lane-per-packet loop ran at 17 % of the lanes). Which continuum is the
// k-th kept one of the window comes from two small per-cell tables made with the keep bitmap (k_keptlist): the kept
// continua as a list, and the number of kept continua below each bitmap word -- two reads and two bit counts give the
// window's places [r0, r1) in the list. (Before: prefix sums of the words' bit counts over the lanes, a bisection with six
// ds_bpermute and a select-k-th-bit per lane and round: 2.5x the VALU instructions.) The contributions are the ones
// update_bfestimators() would have added in place (same arithmetic); only the order of the f64 atomic additions differs.
// (Keeping a cell's sums in LDS across a run of records of the same cell -- per wave, or per workgroup with ds_add_f64 --
// was measured and lost, 651 / 427 ms against 364 ms; so did sorting the records by cell first and summing a cell's whole
// run in an LDS row, 1101 vs 989 ms for k_rpkt + this kernel, profiles/r03/bfest_sorted_lds_rows.patch. Compiled out, the
// additions are worth 98 of the 989 ms.)
// CONT_LDS: the static continuum table (ContPack, two of a contribution's ~six 16-byte reads) is staged in LDS by a
// workgroup of DENSE_TB threads, one per CU.
constexpr int DENSE_TB = 1024;
// LPR lanes per record: a window holds ~30 kept continua (ARTIS_AMD_DENSE_LPR = 64, 32 or 16)
#ifndef ARTIS_DENSE_WGS
#define ARTIS_DENSE_WGS 2  // workgroups of DENSE_TB threads per CU: 8 waves/SIMD at 60 VGPRs (1: 4 waves/SIMD; k_rpkt + this kernel 781 -> 774 ms)
#endif
template <bool CONT_LDS, int TB, int LPR>
__global__ void __launch_bounds__(TB, (TB == DENSE_TB ? ARTIS_DENSE_WGS : 1)) k_bfest_dense(Env env) {
  __shared__ ContPack lds_cont[CONT_LDS ? CONT_LDS_MAX : 1];
  if (CONT_LDS) {
    const D2 *src = (const D2 *)env.M.cont_pack;
    D2 *dst = (D2 *)lds_cont;
    for (int i = threadIdx.x; i < env.M.nbfcontinua * 2; i += TB) dst[i] = src[i];
    env.M.cont_pack = lds_cont;
    env.cont_in_lds = 1;
    __syncthreads();
  }
  static_assert(LPR == 64 || LPR == 32 || LPR == 16, "lanes per record");
  const DevModel &M = env.M;
  const int n = min(*env.bfev_count, env.bfev_cap);
  const int lane = threadIdx.x & (LPR - 1);
  const int unit = (blockIdx.x * TB + threadIdx.x) / LPR;
  const int nunits = gridDim.x * (TB / LPR);
  for (int ei = unit; ei < n; ei += nunits) {
    const BfEvent ev = env.bfev[ei];
    const int c = ev.c;
    const double nu = ev.nu;
    const float T_e = env.C.Te[c];
    // the record's continua are the kept ones of [begin, end): places [r0, r1) of the cell's list of kept continua
    int r0, r1;
    kept_range(env, c, ev.begin, ev.end, r0, r1);
    const double ex = exp(-HOVERKB * nu / T_e);
    const bool split_usable = (ex >= DBLMIN);
    const int32_t *list = env.K.allcont_keptlist + (krow(env, c) * M.nbfcontinua);
    const D2 *keptpair = env.K.allcont_keptpair + (krow(env, c) * M.nbfcontinua);
    // the sums go to the continuum's place in the list when the call keeps them there (all cells resident), else to its
    // estimator: ~30 additions of a record then fall into 4-5 neighbouring 64-byte sectors instead of ~16 scattered ones
    // (k_rpkt + this kernel 978 -> 858 ms per step; with the additions compiled out: 891)
    double *dst = env.bfrate_kept ? env.bfrate_kept + ((int64_t)c * M.nbfcontinua) : env.E.bfrate_raw + ((int64_t)c * M.nbfestim);
    for (int r = r0 + lane; r < r1; r += LPR) {
      const int i = list[r];
      const double ep = keptpair[r].y;
      const int bi = bfestimindex(M, i);
      if (bi >= 0) ARTIS_EST_ADD(&dst[env.bfrate_kept ? r : bi], bf_sigma_contr_ep(env, c, i, nu, T_e, ex, split_usable, ep) * ev.w);
    }
  }
}
// the sums kept by place in the list go to their estimators (one wave per cell; the places are left zero for the next call)
__global__ void __launch_bounds__(BLOCK) k_bfrate_expand(Env env) {
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const DevModel &M = env.M;
  if (wave >= M.npts_nonempty) return;
  const int c = (int)wave;
  const int nw = M.nkeepwords;
  const int nkept = env.K.allcont_keepprefix[(krow(env, c) * nw) + nw - 1] + __popcll(env.K.allcont_keepbits[(krow(env, c) * nw) + nw - 1]);
  const int32_t *list = env.K.allcont_keptlist + (krow(env, c) * M.nbfcontinua);
  double *kept = env.bfrate_kept + ((int64_t)c * M.nbfcontinua);
  double *dst = env.E.bfrate_raw + ((int64_t)c * M.nbfestim);
  for (int r = lane; r < nkept; r += 64) {
    const double v = kept[r];
    if (v != 0.) {
      dst[bfestimindex(M, list[r])] += v;  // (a continuum without an estimator was never added to: v == 0)
      kept[r] = 0.;
    }
  }
}
#endif

// gamma packets (and the non-thermal deposits they end in): one do_gamma() call per iteration, same persistent
// work-pulling form as k_rpkt; a packet that has thermalised leaves as a k-packet for the thermal list
#ifndef ARTIS_GAMMA_WAVES
#define ARTIS_GAMMA_WAVES 2
#endif
__global__ void __launch_bounds__(BLOCK, ARTIS_GAMMA_WAVES) k_gamma(Env env, const int32_t *list, int32_t n, Lists next,
                                                                     unsigned long long *gstats, int budget, int32_t *cursors, int nchunks) {
  __shared__ stat_t lstats[ARTIS_NSTATS];
  __shared__ double lds_cellest[GAMMA_CELLEST_CAP];
  __shared__ double lds_scalars[ARTIS_NSCALARS];
  if (threadIdx.x < ARTIS_NSTATS) lstats[threadIdx.x] = 0;
  scalars_begin(env, lds_scalars);
  cellest_begin(env, lds_cellest, env.cellest_n_g, BLOCK, env.E.dep_estimator_gamma);
  __syncthreads();
  env.stats = lstats;
  const double ts_end = env.S.ts_end;
  Puller q;
  puller_init(q, n, nchunks);
  bool have = false;
  int32_t pi = 0;
  int steps = 0;
  Pkt p;
  while (true) {
    const int32_t idx = pull(q, !have, n, cursors);
    if (idx >= 0) {
      pi = list[idx];
      pkt_load(env.P, pi, p);
      steps = 0;
      have = true;
    }
    if (!__any(have)) {
      if (q.exhausted) break;
      continue;
    }
    int kind = NEXT_DONE;
    int32_t out_pi = 0;
    if (have) {
      bool go = gamma_can_continue(p, ts_end);
      if (go) {
        go = gamma_iter(env, p, pi);
        steps++;
      }
      if (!go || steps >= budget) {
        pkt_store(env.P, pi, p);
        kind = classify(env, p, ts_end);
        out_pi = pi;
        have = false;
      }
    }
    append_by_kind(kind, out_pi, p.cellindex, p.nu_cmf, next, (p.ma_element * 5 + p.ma_ion));
  }
  __syncthreads();
  cellest_flush(env, CELLEST_DEPGAMMA, env.E.dep_estimator_gamma, BLOCK);
  scalars_flush(env);
  if (threadIdx.x < ARTIS_NSTATS && lstats[threadIdx.x] != 0) atomicAdd(&gstats[threadIdx.x], lstats[threadIdx.x]);
}

// Thermal packets (k-packets and walking macro-atoms) are advanced by ONE persistent kernel, so that the k-packet ->
// macro-atom -> k-packet cycle (tens of times per packet and timestep, kpkt.cc:51) needs no kernel boundary.
//
// Staging macro-atom records in LDS was built and measured in rounds 2 and 3 (per-workgroup slots holding the hot levels'
// records of the cells in flight, k_thermal<true>; and k_thermal_q, walk contexts in per-wave LDS slots with lanes refilled
// inside the transition loop): both bit-identical, both slower (profiles/r02/lds_staging.md,
// profiles/r03/k_thermal_lane_compaction.md). Round 4 took the cumulative sums and targets those variants staged out of
// the records altogether (tables.h), and the variants with them.
#ifndef ARTIS_THERMAL_WAVES
#define ARTIS_THERMAL_WAVES 4
#endif
// Fused thermal kernel, phase form (physics.h thermal_iter): up to ARTIS_MA_PHASE macro-atom transitions, then one
// k-packet step, per iteration; lanes take a new packet between iterations. Workgroups of TB threads on the XCD chunks.
#ifndef ARTIS_THERMAL_TB
#define ARTIS_THERMAL_TB BLOCK                 // threads per workgroup of k_thermal,
#define ARTIS_THERMAL_EU ARTIS_THERMAL_WAVES   // the waves per SIMD its registers are to allow (512 / EU VGPRs) ...
#define ARTIS_THERMAL_WGS ARTIS_THERMAL_WAVES  // ... and its workgroups per CU
#endif
// 1: k_thermal holds no rate-coefficient code: a search its filters cannot decide (3e-4 per transition) is handed to the slow-path
// kernel with the draw in the packet's pend fields (physics.h PEND_MA_SEARCH / _RADSEARCH / PEND_KPKT_COLLEXC), which re-adds the sums
// and carries on; the packet is back on the thermal list for the next launch. With the code inlined here the kernel spilled 122
// instead of 21 registers, reloaded around every phase: 200 GB of scratch traffic per step.
// (Not in the builds with detailed bound-free estimators -- the nltenebular family: their steps are made of twice as many launch rounds,
// and every hand-over costs a packet the rest of its launch: measured 1417 ms with, 1403 without.)
#ifndef ARTIS_THERMAL_SPLIT_EXACT
#define ARTIS_THERMAL_SPLIT_EXACT (ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON ? 0 : 1)
#endif
#ifndef ARTIS_MA_DEFER_EXACT
#define ARTIS_MA_DEFER_EXACT 1  // the re-adding of a search's sums outside the transition loop (physics.h ma_jump_internal<true>)
#endif
// TABLES_LDS: the two static tables a transition's target is read from -- the target level of every entry of alltrans
// (2 bytes) and the LevelPack of every level (16 bytes) -- are copied into LDS once per workgroup (one workgroup of 1024
// threads per CU: the same 4 waves/SIMD at 128 VGPRs), and a transition reads its target with two ds_reads instead of a
// 16-byte gather from the static target table in L2: of a transition's three dependent reads (action filter; the
// direction's filter, in the sector the first one brought; the target) one long wait is left. For atomic data whose
// tables fit (MA_LDS_LEVELS / MA_LDS_TRANS: the bench's 1567 levels and 27 238 entries take 80 KB); larger data keep the
// target table in HBM. Static tables, the same for every cell: nothing to stage per cell (what lost in rounds 2-3).
// TABLES_LDS = 2: atomic data whose alltrans table does not fit (the 110 860-line set: 221 720 entries) keep the 2-byte target
// levels in HBM (443 KB instead of the 3.5 MB of 16-byte targets: L2-resident beside the cells' records) and the LevelPack
// table alone in LDS (up to MA_LDS_LEVELS2 levels).
constexpr int MA_LDS_LEVELS = 2048;   // 32 KB
constexpr int MA_LDS_TRANS = 32768;   // 64 KB
constexpr int MA_LDS_LEVELS2 = 9856;  // 154 KB (round 6: 6144 = 96 KB until the workgroup's per-cell estimator accumulators -- 32 KB that only models
                                      // with few cells use -- were left out of this form: the 4e5-line set's 8 457 levels fit now; a model with few cells AND
                                      // more than MA_LDS_LEVELS levels takes the form without LDS tables)
template <int TB, int TABLES_LDS, bool COLD = false>
__global__ void __launch_bounds__(TB, (TABLES_LDS ? 1 : ARTIS_THERMAL_EU)) k_thermal(Env env, const int32_t *list, int32_t n, Lists next,
                                                                     unsigned long long *gstats, int budget, int32_t *cursors,
                                                                     int nchunks, int chunk_mode, int drain_budget) {
  __shared__ stat_t lstats[ARTIS_NSTATS];
  __shared__ double lds_cellest[TABLES_LDS == 2 ? 1 : THERMAL_CELLEST_CAP];
  __shared__ LevelPack lds_levelpack[TABLES_LDS == 1 ? MA_LDS_LEVELS : (TABLES_LDS == 2 ? MA_LDS_LEVELS2 : 1)];
  __shared__ uint16_t lds_tlevel[TABLES_LDS == 1 ? MA_LDS_TRANS : 8];
  if (threadIdx.x < ARTIS_NSTATS) lstats[threadIdx.x] = 0;
  cellest_begin(env, lds_cellest, env.cellest_n_t, TB, env.E.colheatingestimator);
  env.estcache = nullptr;
  env.estcache_nv = 0;
  // many cells: the workgroup's array is not in use; its LDS holds the waves' caches of per-cell sums instead (physics.h Env::estcache): per wave
  // ESTCACHE_SLOTS doubles, then per wave 2 x ESTCACHE_SLOTS int32 (the slots' cells, the claims)
  static_assert(TABLES_LDS == 2 || (TB / 64) * ESTCACHE_SLOTS * 2 <= THERMAL_CELLEST_CAP, "the waves' estimator caches take the few-cells array's LDS");
  if (TABLES_LDS != 2 && env.estcache_on && env.cellest_n_t == 0) {
    const int w = threadIdx.x >> 6;
    env.estcache = lds_cellest + (w * ESTCACHE_SLOTS);
    env.estcache_tag = (int32_t *)(lds_cellest + ((TB / 64) * ESTCACHE_SLOTS)) + (w * ESTCACHE_SLOTS * 2);
    env.estcache_nv = 1;
    for (int l = threadIdx.x & 63; l < ESTCACHE_SLOTS; l += 64) {
      env.estcache[l] = 0.;
      env.estcache_tag[l] = -1;
      env.estcache_tag[ESTCACHE_SLOTS + l] = -1;
    }
  }
  if (TABLES_LDS) {
    for (int i = threadIdx.x; i < env.M.nlevels; i += TB) lds_levelpack[i] = env.M.level_pack[i];
    env.M.level_pack = lds_levelpack;
    if (TABLES_LDS == 1) {
      const uint32_t *src = (const uint32_t *)env.M.alltrans_tlevel16;  // (the allocation is padded to whole words)
      uint32_t *dst = (uint32_t *)lds_tlevel;
      for (int i = threadIdx.x; i < (env.M.nalltrans + 1) / 2; i += TB) dst[i] = src[i];
      env.M.alltrans_tlevel16 = lds_tlevel;
    }
    env.ma_tables_in_lds = 1;
  }
  __syncthreads();
  env.stats = lstats;
  const double ts_end = env.S.ts_end;
  Puller q;
  puller_init(q, n, nchunks, chunk_mode);
  bool have = false;
  int32_t pi = 0;
  int units = 0;
  bool drained = false;  // the launch's work list is used up (any wave found out)
  Pkt p;
  MACtx k;
#ifdef ARTIS_PROFILE
#define PROF_ADD(slot, dt) \
  do {                     \
    if ((threadIdx.x & 63) == 0) atomicAdd(&lstats[slot], (stat_t)((dt) >> 4)); \
  } while (0)
  long long tprev = clock64();
#endif
  while (true) {
    const int32_t idx = pull(q, !have, n, cursors);
    if (idx >= 0) {
      pi = list[idx];
      pkt_load_thermal(env.P, pi, p);  // the hot line only
      k = ma_ctx(env, p);
      units = 0;
      have = true;
    }
    if (!__any(have)) {
      if (q.exhausted) break;
      continue;
    }
    int kind = NEXT_DONE;
    int32_t out_pi = 0;
    // the two phases of thermal_iter() (physics.h), spelled out so that the wave reconverges between them
#ifdef ARTIS_PROFILE
    // wave-cycle accounting (units of 16 clocks) in the spare stats slots 42..47: pull+load | macro-atom phase |
    // k-packet phase | store+append, and the wave-level iteration counts of the two phases
    const long long t0 = clock64();
    PROF_ADD(42, t0 - tprev);
#endif
    bool go = have && thermal_can_continue(p, ts_end);
    if (go) {
      // the loop makes the internal transitions; the process that ends a walk is carried out after it, once per phase
      int j = 0;
      int exit_action = -1;
      const U4 *rec = nullptr;
      if (ma_pending(p) && p.pend == PEND_NONE) ma_prepare<COLD>(env, p, k);  // the record of the current level; the walk carries it on
      while (j < ARTIS_MA_PHASE && exit_action < 0 && ma_pending(p) && p.pend == PEND_NONE) {  // [census: transition loop]
#ifdef ARTIS_PROFILE
        if ((threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) ARTIS_STAT(env, 46);
#endif
        rec = ma_record<COLD>(env, k);
        exit_action = ma_jump_internal<ARTIS_MA_DEFER_EXACT != 0, COLD>(env, p, k, rec);
        j++;
      }
      ma_flush_stats(env, k);
      // (a transition whose search the filters could not decide is finished here, outside the loop: the walk goes on in the
      // next phase)
      if (exit_action == MA_EXIT_FILL) {
        p.pend = PEND_MA_FILL;  // a cold level without a record in this cell: the slow-path kernel fills it (physics.h ma_slow_fill)
      } else if (exit_action == MA_EXIT_DEFER) {
#if ARTIS_THERMAL_SPLIT_EXACT
        // the lines' fine bytes decide all but 1e-6 of these (round 6; tables.h "FINE BYTES"): the walk goes on in the next phase. What they leave:
        if (!ma_jump_deferred_fine<COLD>(env, p, k, rec)) {
          p.pend = PEND_MA_SEARCH;  // the slow-path kernel re-adds the sums and makes the transition (physics.h ma_slow_search)
          p.pend_arg = k.defer;
        }
#else
        ma_jump_deferred(env, p, k, rec);
#endif
      } else if (exit_action >= 0) {
        ma_jump_exit<ARTIS_THERMAL_SPLIT_EXACT != 0>(env, p, pi, k, rec, exit_action);
      }
      if (j > 0) chi_after_ma(p);
      units += j;
    }
#ifdef ARTIS_PROFILE
    const long long t1 = clock64();
    PROF_ADD(43, t1 - t0);
#endif
    if (go) {
      // a pre-k-packet, or a k-packet in a grey cell, leaves for the blackbody kernel (classify() below)
      const bool blackbody = (p.type == ARTIS_TYPE_PRE_KPKT) || k.thick;
      if (kpkt_eligible(p, ts_end) && !blackbody) {
#ifdef ARTIS_PROFILE
        if ((threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) ARTIS_STAT(env, 47);
#endif
        do_kpkt<ARTIS_THERMAL_SPLIT_EXACT != 0>(env, p, pi);
        p.chi_mgi = -1;
        units++;
      }
      go = thermal_can_continue(p, ts_end) && !(blackbody && kpkt_eligible(p, ts_end));
    }
#ifdef ARTIS_PROFILE
    const long long t2 = clock64();
    PROF_ADD(44, t2 - t1);
#endif
    // Once the work list is used up, the launch lasts as long as its slowest packets keep their lanes (up to `budget`
    // units each) while the rest of the GPU idles. In a launch whose successor is large anyway (drain_budget set by the
    // host), a packet then leaves after drain_budget units for that next launch, where its work runs beside a full list.
    // (Where a packet is handed from launch to launch never changes it: budgets are placement, tested.)
    if (!drained && q.exhausted) {
      drained = true;
      if ((threadIdx.x & 63) == 0) __hip_atomic_store(&cursors[MAX_CHUNKS], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!drained && drain_budget < budget)
      drained = __hip_atomic_load(&cursors[MAX_CHUNKS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (have && (!go || units >= budget || (drained && units >= drain_budget))) {
      pkt_store_thermal(env.P, pi, p);  // the hot line; the flight line only if an r-packet was emitted
      kind = classify(env, p, ts_end);
      out_pi = pi;
      have = false;
    }
    append_by_kind(kind, out_pi, p.cellindex, p.nu_cmf, next, (p.ma_element * 5 + p.ma_ion));
    pkt_clear_flight(p);  // a thermal packet never reads them: no live range across iterations
#ifdef ARTIS_PROFILE
    tprev = clock64();
    PROF_ADD(45, tprev - t2);
#endif
  }
  __syncthreads();
  if (env.estcache_nv == 1) {  // the wave's sums go to the cells' records
    for (int l = threadIdx.x & 63; l < ESTCACHE_SLOTS; l += 64) {
      const int cell = env.estcache_tag[l];
      const double s0 = env.estcache[l];
      if (cell >= 0 && s0 != 0.) unsafeAtomicAdd(&env.E.colheatingestimator[(int64_t)cell * env.est_stride], s0);
    }
  }
  cellest_flush(env, CELLEST_COLHEAT, env.E.colheatingestimator, TB);
  if (threadIdx.x < ARTIS_NSTATS && lstats[threadIdx.x] != 0) atomicAdd(&gstats[threadIdx.x], lstats[threadIdx.x]);
}

// ---- k_thermal_q (round 5; ARTIS_AMD_REFILL=1): the thermal kernel with the macro-atom walk DECOUPLED from the packet, so that the
// transition loop's lanes are refilled inside the loop. Round 3 built this once (profiles/r03/k_thermal_lane_compaction.md: lanes 35 -> 50,
// and slower, because a transition was then three dependent L2 gathers whose wait grew with the lanes in flight). A transition is now one
// 64-byte sector of the cell's record and two reads of static tables in LDS, and k_thermal runs at 37 of 64 lanes in a loop that is bound
// by instruction issue: re-measured here under today's read pattern.
// A wave owns TQ_V (> 64) packets. What a walk needs of a packet -- generator state, cell, ion's first level, level, counters: 36 B -- is its
// WALK CONTEXT, kept in the wave's LDS slots; the packet's hot line rests in memory meanwhile. Two phases alternate per wave:
//   walk:    every lane holds one context in registers and makes one transition per round; a lane whose walk ends (any process but an
//            internal transition, an undecided search, the launch budget) writes the context back, pushes the slot on the wave's SERVICE
//            stack and pops the next READY slot in the same round (ballot + popcount);
//   service: once 64 slots wait (or the walkers run short), ONE full-wave pass reloads those packets' hot lines, carries out the process
//            that ended each walk (ma_jump_exit), makes the k-packet step that follows, and either prepares the next walk (slot READY, hot
//            line stored) or retires the packet (stored, appended to the list of its next kind) and pulls a new one into the slot.
// Per packet the same functions run in the same order on the packet's own generator as in k_thermal: identical packets, generator states
// and counters (GPU test); estimator sums differ by the order of their additions only.
#ifndef ARTIS_TQ_SLOTS
#define ARTIS_TQ_SLOTS 128
#endif
constexpr int TQ_V = ARTIS_TQ_SLOTS;  // slots per wave: 64 walking + a buffer that lets a full service pass fall due before the walkers starve
static_assert(TQ_V >= 64 && TQ_V <= 256, "slot indices are kept in bytes");
enum { TQ_EMPTY = -2, TQ_BUDGET = -3 };  // action of a slot on the service stack: no packet | walk interrupted by the launch budget
struct TQWave {  // SoA: a lane reads field[its slot]
  uint32_t s0[TQ_V], s1[TQ_V], s2[TQ_V], s3[TQ_V];
  // lv = level within the ion | the ion's first level << 16; cnt = units | njumps << 16; act = (action + 3) | the draw of an undecided search << 7
  int32_t pi[TQ_V], c[TQ_V], lv[TQ_V], cnt[TQ_V];
  uint32_t act[TQ_V];
  uint8_t ready[TQ_V];    // stack of the slots whose walk can go on
  uint8_t service[TQ_V];  // stack of the slots that wait for the service pass
};
template <int TB, bool COLD = false>
__global__ void __launch_bounds__(TB, 1) k_thermal_q(Env env, const int32_t *list, int32_t n, Lists next, unsigned long long *gstats, int budget,
                                                     int32_t *cursors, int nchunks, int drain_budget, int low_water) {
  // dynamic LDS: [TQWave x waves | LevelPack x nlevels | uint16 x nalltrans (padded to words)]
  extern __shared__ __attribute__((aligned(16))) unsigned char tq_lds[];
  __shared__ stat_t lstats[ARTIS_NSTATS];
  TQWave *tq = (TQWave *)tq_lds;
  LevelPack *lds_levelpack = (LevelPack *)(tq_lds + (((sizeof(TQWave) * (TB / 64)) + 15) & ~(size_t)15));
  uint16_t *lds_tlevel = (uint16_t *)(lds_levelpack + env.M.nlevels);
  if (threadIdx.x < ARTIS_NSTATS) lstats[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < env.M.nlevels; i += TB) lds_levelpack[i] = env.M.level_pack[i];
  {
    const uint32_t *src = (const uint32_t *)env.M.alltrans_tlevel16;  // (the allocation is padded to whole words)
    uint32_t *dst = (uint32_t *)lds_tlevel;
    for (int i = threadIdx.x; i < (env.M.nalltrans + 1) / 2; i += TB) dst[i] = src[i];
  }
  env.M.level_pack = lds_levelpack;
  env.M.alltrans_tlevel16 = lds_tlevel;
  env.ma_tables_in_lds = 1;
  env.cellest_lds = nullptr;
  env.cellest_n = 0;
  __syncthreads();
  env.stats = lstats;
  const double ts_end = env.S.ts_end;
  const int lane = threadIdx.x & 63;
  const unsigned long long lanebit = 1ull << lane;
  TQWave &Q = tq[threadIdx.x >> 6];
  Puller q;
  puller_init(q, n, nchunks, 0);
  // wave-uniform stack heights; every slot starts empty and waits for a packet
  int nready = 0, nservice = TQ_V, ndead = 0;
  for (int i = lane; i < TQ_V; i += 64) {
    Q.service[i] = (uint8_t)i;
    Q.act[i] = (uint32_t)(TQ_EMPTY + 3);
  }
  __builtin_amdgcn_wave_barrier();
  bool drained = false;
#ifdef ARTIS_PROFILE
  long long tq_t = clock64();  // wave clocks / 16 of the two phases in the spare stats slots 42 (service) and 43 (walk)
#define TQ_PROF(slot)                                                                   \
  do {                                                                                  \
    const long long now = clock64();                                                    \
    if (lane == 0) atomicAdd(&lstats[slot], (stat_t)((now - tq_t) >> 4));               \
    tq_t = now;                                                                         \
  } while (0)
#else
#define TQ_PROF(slot) ((void)0)
#endif
#if defined(ARTIS_PROFILE) && !defined(ARTIS_PROFILE_MA)
  long long tq_s = clock64();
#define TQ_SUB(slot)                                                                    \
  do {                                                                                  \
    const long long now = clock64();                                                    \
    if (lane == 0) atomicAdd(&lstats[slot], (stat_t)((now - tq_s) >> 4));               \
    tq_s = now;                                                                         \
  } while (0)
#else
#define TQ_SUB(slot) ((void)0)
#endif
  while (true) {
    // ---------------- service passes: while a full wave of slots waits, or the walkers would run short
    while (nservice >= 64 || (nservice > 0 && nready < low_water)) {
#if defined(ARTIS_PROFILE) && !defined(ARTIS_PROFILE_MA)
      tq_s = clock64();
#endif
      const int take = min(64, nservice);
      const bool has = lane < take;
      const int s = has ? (int)Q.service[nservice - 1 - lane] : 0;
      nservice -= take;
      const uint32_t actw = has ? Q.act[s] : (uint32_t)(TQ_EMPTY + 3);
      const int act = (int)(actw & 127u) - 3;
      const bool isdone = has && act != TQ_EMPTY;
      const int32_t idx = pull(q, has && !isdone, n, cursors);
      if (!drained && q.exhausted) {
        drained = true;
        if (lane == 0) __hip_atomic_store(&cursors[MAX_CHUNKS], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!drained && drain_budget < budget) drained = __hip_atomic_load(&cursors[MAX_CHUNKS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      const int budget_now = drained ? drain_budget : budget;
      const bool have = isdone || idx >= 0;
      int32_t pi = 0;
      int units = 0;
      Pkt p;
      MACtx k;
      if (have) {
        pi = isdone ? Q.pi[s] : list[idx];
        pkt_load_thermal(env.P, pi, p);  // the hot line only
        k = ma_ctx(env, p);
        if (isdone) {  // the walk's own state is the slot's; the level's record comes from the static table again
          p.s0 = Q.s0[s]; p.s1 = Q.s1[s]; p.s2 = Q.s2[s]; p.s3 = Q.s3[s];
          const int lv = Q.lv[s];
          p.ma_level = lv & 0xFFFF;
          k.start = lv >> 16;
          k.start_key = (p.ma_element << 8) | p.ma_ion;
          const LevelPack lp = env.M.level_pack[k.start + p.ma_level];
          k.rec = ma_resolve<COLD>(env, k.c, lp.rec_off); k.nd = lp.ndown; k.nu = lp.nup; k.ats = lp.alltrans_startdown;
          const int cnt = Q.cnt[s];
          units = cnt & 0xFFFF;
          k.njumps = cnt >> 16;
          k.defer = (int)(actw >> 7);
        }
      }
#ifdef ARTIS_PROFILE
      if (lane == 0) {
        ARTIS_STAT(env, 47);            // service passes
        ARTIS_STAT_ADD(env, 44, take);  // ... and the slots they served
      }
#endif
      TQ_SUB(59);  // (-DARTIS_PROFILE: wave clocks / 16 of the parts of a service pass: 59 pull + hot line + context | 60 the process that ended
                   //  the walk | 61 the k-packet step | 62 prepare + store + context | 63 classify + append + stacks)
      int kind = NEXT_DONE;
      int32_t out_pi = 0;
      bool walking = false;
      if (have) {  // (one block: split into parts -- as the profile marks would like it -- the compiler spills 93 instead of 13 registers)
        bool go = thermal_can_continue(p, ts_end);
        if (isdone) {
          ma_flush_stats(env, k);
          const U4 *rec = ma_record<COLD>(env, k);
          if (act == MA_EXIT_FILL) {
            p.pend = PEND_MA_FILL;
          } else if (act == MA_EXIT_DEFER) {
            if (!ma_jump_deferred_fine<COLD>(env, p, k, rec)) {
              p.pend = PEND_MA_SEARCH;  // the slow-path kernel re-adds the sums and makes the transition (physics.h ma_slow_search)
              p.pend_arg = k.defer;
            }
          } else if (act >= 0) {
            ma_jump_exit<true>(env, p, pi, k, rec, act);
          }
          chi_after_ma(p);
          go = thermal_can_continue(p, ts_end);
        }
        TQ_SUB(60);
        if (go) {
          // a pre-k-packet, or a k-packet in a grey cell, leaves for the blackbody kernel (classify() below)
          const bool blackbody = (p.type == ARTIS_TYPE_PRE_KPKT) || k.thick;
          if (kpkt_eligible(p, ts_end) && !blackbody) {
            do_kpkt<true>(env, p, pi);
            p.chi_mgi = -1;
            units++;
          }
          go = thermal_can_continue(p, ts_end) && !(blackbody && kpkt_eligible(p, ts_end));
        }
        TQ_SUB(61);
        walking = go && units < budget_now && ma_pending(p) && p.pend == PEND_NONE;
        if (walking) ma_prepare<COLD>(env, p, k);  // (only k.start is kept: the walk phase reads the level's record shape from LDS)
        pkt_store_thermal(env.P, pi, p);  // the hot line; the flight line only if an r-packet was emitted
        if (walking) {
          Q.s0[s] = p.s0; Q.s1[s] = p.s1; Q.s2[s] = p.s2; Q.s3[s] = p.s3;
          Q.pi[s] = pi;
          Q.c[s] = k.c;
          Q.lv[s] = p.ma_level | (k.start << 16);
          Q.cnt[s] = units;
        } else {
          kind = classify(env, p, ts_end);
          out_pi = pi;
        }
      }
      TQ_SUB(62);
      append_by_kind(kind, out_pi, p.cellindex, p.nu_cmf, next, (p.ma_element * 5 + p.ma_ion));
      // where the slots go: READY, or back on the service stack as empty (a packet is pulled into it by the next pass), or --
      // once the work list is used up -- out of use
      {
        const unsigned long long rm = __ballot(walking);
        if (walking) Q.ready[nready + __popcll(rm & (lanebit - 1ull))] = (uint8_t)s;
        nready += __popcll(rm);
        const bool again = has && !walking && !q.exhausted;
        const unsigned long long em = __ballot(again);
        if (again) {
          Q.act[s] = (uint32_t)(TQ_EMPTY + 3);
          Q.service[nservice + __popcll(em & (lanebit - 1ull))] = (uint8_t)s;
        }
        nservice += __popcll(em);
        ndead += take - __popcll(rm) - __popcll(em);
      }
      __builtin_amdgcn_wave_barrier();
      TQ_SUB(63);
    }
    TQ_PROF(42);
    if (nready == 0) break;  // (then nothing waits for service either: every slot is out of use)
    // ---------------- walk phase
    {
      // in the drain (work list used up) a service pass is worth its cost only for a reasonable share of the live slots
      const int drain_min = max(1, min(16, (TQ_V - ndead) / 4));
      const int budget_now = drained ? drain_budget : budget;
      int myslot = -1;
      int units = 0;
      Pkt w;  // only the generator state and ma_level are live
      MACtx k;
      w.ma_level = -1;
      k.c = 0; k.cellma = nullptr; k.rec = 0; k.nd = k.nu = 0; k.ats = 0; k.njumps = 0; k.start = 0; k.start_key = -1; k.defer = 0; k.thick = false;
#ifdef ARTIS_PROFILE
      int prof_rounds = 0, prof_lanes = 0;  // wave-uniform: rounds of this phase and the lanes that made a transition in them
#endif
      while (true) {
        {  // lanes without a context pop READY slots
          const bool need = myslot < 0;
          const unsigned long long m = __ballot(need);
          if (m != 0 && nready > 0) {
            const int takeN = min(__popcll(m), nready);
            const int prefix = __popcll(m & (lanebit - 1ull));
            if (need && prefix < takeN) {
              const int s = (int)Q.ready[nready - 1 - prefix];
              myslot = s;
              w.s0 = Q.s0[s]; w.s1 = Q.s1[s]; w.s2 = Q.s2[s]; w.s3 = Q.s3[s];
              const int lv = Q.lv[s];
              w.ma_level = lv & 0xFFFF;
              k.c = Q.c[s];
              k.cellma = env.K.macache + (krow(env, k.c) * env.M.nmacache);
              k.start = lv >> 16;
              const LevelPack lp = env.M.level_pack[k.start + w.ma_level];
              k.rec = ma_resolve<COLD>(env, k.c, lp.rec_off); k.nd = lp.ndown; k.nu = lp.nup; k.ats = lp.alltrans_startdown;
              const int cnt = Q.cnt[s];
              units = cnt & 0xFFFF;
              k.njumps = cnt >> 16;
            }
            nready -= takeN;
          }
        }
        const int nactive = __popcll(__ballot(myslot >= 0));
        if (nactive == 0 || nservice >= 64 || (nactive < low_water && nservice >= drain_min)) break;
        bool ended = false;
        int end_action = 0;
        {
          const bool go = myslot >= 0;
#ifdef ARTIS_PROFILE
          prof_lanes += __popcll(__ballot(go));
          prof_rounds++;
#endif
          if (go) {
            const int action = ma_jump_internal<true, COLD>(env, w, k, ma_record<COLD>(env, k));
            units++;
            if (action >= 0 || units >= budget_now) {
              ended = true;
              end_action = action >= 0 ? action : TQ_BUDGET;
            }
          }
        }
        const unsigned long long em = __ballot(ended);
        if (em != 0) {
          if (ended) {
            const int s = myslot;
            Q.s0[s] = w.s0; Q.s1[s] = w.s1; Q.s2[s] = w.s2; Q.s3[s] = w.s3;
            Q.lv[s] = w.ma_level | (k.start << 16);
            Q.cnt[s] = units | (k.njumps << 16);
            Q.act[s] = (uint32_t)(end_action + 3) | ((uint32_t)k.defer << 7);
            Q.service[nservice + __popcll(em & (lanebit - 1ull))] = (uint8_t)s;
            myslot = -1;
          }
          nservice += __popcll(em);
        }
      }
      // park the walks in progress: their slots are READY again
      const bool parked = myslot >= 0;
      const unsigned long long pm = __ballot(parked);
      if (parked) {
        const int s = myslot;
        Q.s0[s] = w.s0; Q.s1[s] = w.s1; Q.s2[s] = w.s2; Q.s3[s] = w.s3;
        Q.lv[s] = w.ma_level | (k.start << 16);
        Q.cnt[s] = units | (k.njumps << 16);
        Q.ready[nready + __popcll(pm & (lanebit - 1ull))] = (uint8_t)s;
      }
      nready += __popcll(pm);
#ifdef ARTIS_PROFILE
      if (lane == 0) {
        ARTIS_STAT_ADD(env, 46, prof_rounds);  // wave-rounds of the transition loop
        ARTIS_STAT_ADD(env, 45, prof_lanes);   // ... and the lanes that made a transition in them
      }
#endif
      __builtin_amdgcn_wave_barrier();
      TQ_PROF(43);
    }
  }
  __syncthreads();
  if (threadIdx.x < ARTIS_NSTATS && lstats[threadIdx.x] != 0) atomicAdd(&gstats[threadIdx.x], lstats[threadIdx.x]);
}
#ifndef ARTIS_TQ_TB
#define ARTIS_TQ_TB 768
#endif
constexpr int TQ_TB = ARTIS_TQ_TB;  // 12 waves per CU = 3 waves/SIMD at 168 VGPRs (round 3's optimum for this form; the slots of 16 waves and the tables do not fit the LDS)
inline size_t tq_lds_bytes(int tb, int nlevels, int nalltrans) {
  return (((sizeof(TQWave) * (size_t)(tb / 64)) + 15) & ~(size_t)15) + (sizeof(LevelPack) * (size_t)nlevels) + (sizeof(uint32_t) * (size_t)((nalltrans + 1) / 2));
}

// Tail kernel: the LAST few thousand r-packets and thermal packets of a timestep, one per lane, each carried through
// r-packet steps, macro-atom walks and k-packet steps until it leaves these kinds (end of the timestep, escape, a
// slow-path action, a gamma-ray type, a blackbody step, another cache tile). The event-split kernels need one launch
// pair per r-packet <-> thermal alternation of the longest-lived packets, and a launch lasts as long as its slowest
// packet: with the nltenebular options 530 of 720 launches of a step hold fewer than 10^4 packets and cost 490 ms
// (the last 150 hold 1-5 packets). Here the packets do not wait for each other. Same functions in the same order per
// packet as the split kernels (state goes through the packet record at every change of kind, as it does between
// launches), so the same results; a fat kernel (both bodies: scratch, low occupancy), which is why it is the tail only.
struct TailLists {
  const int32_t *list[4];  // the current r-packet, thermal, slow-path and blackbody lists
  int32_t n[4];
};
#ifndef ARTIS_TAIL_WAVES
#define ARTIS_TAIL_WAVES 2
#endif
#ifndef ARTIS_SLOW_WAVE_FB
#define ARTIS_SLOW_WAVE_FB 1  // free-bound emission frequencies selected by the wave (physics.h FbSel) in k_slow and k_tail
#endif
#ifndef ARTIS_TAIL_WAVE_CHI
#define ARTIS_TAIL_WAVE_CHI 1  // k_tail: the continuum opacity of a step evaluated by the wave (physics.h chi_rpkt_cont_wave); 0: by the packet's lane
#endif
#ifndef ARTIS_SLOW_WAVE_FB_MAX
#define ARTIS_SLOW_WAVE_FB_MAX 6  // ... in k_slow for waves with at most this many emissions
#endif
// (Round 5: with the static tables of a transition's target in LDS as in k_thermal -- workgroups of 8 packets, one per compute unit -- the
// classic tail took 27.6 / 26.7 ms against 29.0 / 25.3, the nltenebular tail 123 against 108: profiles/r05/tail_profile.txt. Not kept.)
__global__ void __launch_bounds__(BLOCK, ARTIS_TAIL_WAVES) k_tail(Env env, TailLists in, Lists next, unsigned long long *gstats) {
  __shared__ stat_t lstats[ARTIS_NSTATS];
  if (threadIdx.x < ARTIS_NSTATS) lstats[threadIdx.x] = 0;
  __syncthreads();
  env.stats = lstats;
  env.ma_concurrent_fill = 1;  // this kernel's waves fill pool records while others read them: published places are read with acquire (physics.h ma_rowtab_acquire)
  const double ts_end = env.S.ts_end;
  // ONE PACKET PER WAVE (lane 0): packets in different kinds of step would otherwise serialise inside a wave (measured
  // with a packet per lane: slower than the split kernels); the tail has fewer packets than the GPU has waves
  const int64_t tid = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  int kind = NEXT_DONE;
  int32_t pi = 0, cellindex = 0;
  double nu_cmf = 0.;
  // (every lane runs the loop over the packet's kinds of step with lane 0's kind; only lane 0 holds the packet and carries the steps out,
  // the other 63 join it where the wave works together: the frequency of a free-bound emission, physics.h FbSel)
  const bool owner = (threadIdx.x & 63) == 0 && tid < (int64_t)in.n[0] + in.n[1] + in.n[2] + in.n[3];
  Pkt p;
  if (owner) {
    int32_t j = (int32_t)tid;
    int which = 0;
    while (j >= in.n[which]) j -= in.n[which++];
    pi = in.list[which][j];
    pkt_load(env.P, pi, p);
  }
#ifdef ARTIS_PROFILE_TAIL
  // (-DARTIS_PROFILE_TAIL, tools/tail_profile.py) clocks / 16 of the owner lane by kind of step: slots 42 slow, 43 r-packet, 44 thermal, 45 blackbody,
  // 46 / 47 the r-packet and thermal iterations; 50..53 the same clocks of the waves that ran longer than 2^25 clocks (~15 ms) only
  long long tp[4] = {0, 0, 0, 0};
  long long tp_n[2] = {0, 0};
#define TAILP(i, code)                   \
  {                                      \
    const long long tp0_ = clock64();    \
    code;                                \
    tp[i] += clock64() - tp0_;           \
  }
#else
#define TAILP(i, code) code
#endif
  while (true) {
    if (owner) kind = classify(env, p, ts_end);
    const int kind_w = __builtin_amdgcn_readfirstlane(__shfl(kind, 0));
    if (kind_w == NEXT_SLOW) {
#ifdef ARTIS_PROFILE_TAIL
      const long long tps0 = clock64();
#endif
#if ARTIS_SLOW_WAVE_FB
      FbSel sel;
      sel.mode = 1;
      sel.valid = false;
      sel.element = sel.lowerion = sel.lower = sel.t = 0;
      sel.T_e = 0.f;
      sel.zrand = sel.nu = 0.;
      if (owner && slow_selects_continuum_nu(p)) {
        Pkt t = p;
        (void)advance_slow(env, t, pi, &sel);
      }
      fbsel_wave(env, sel);
#endif
      {
        int fc = 0, ful = 0;
        int32_t funit = 0;
        const bool fill = ARTIS_MA_WAVE_FILL && owner && p.pend == PEND_MA_FILL && ma_slow_fill_claim(env, p, &fc, &ful, &funit);
        ma_fill_wave(env, fill, fc, ful);
        if (fill) ma_slow_fill_publish(env, fc, ful, funit);
      }
#if ARTIS_SLOW_WAVE_FB
      if (owner) (void)advance_slow(env, p, pi, sel.valid ? &sel : nullptr);
#else
      if (owner) (void)advance_slow(env, p, pi);
#endif
      {
        // the pool of on-demand records is used up and the packet still waits for a record: it leaves for the slow-path list (the host empties
        // the pool before that list's next launch); nothing in this kernel could end the wait
        // (only a packet whose level has no record and none being filled: one whose record another wave is filling goes on -- ADVICE r05)
        const bool waits = owner && env.M.ncold > 0 && p.pend != PEND_NONE &&
                           __hip_atomic_load(env.ma_pool_full, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 && ma_record_absent(env, p);
        if (__builtin_amdgcn_readfirstlane(__shfl((int)waits, 0)) != 0) {
          if (owner) kind = NEXT_SLOW;
          break;
        }
      }
#ifdef ARTIS_PROFILE_TAIL
      tp[0] += clock64() - tps0;
#endif
    } else if (kind_w == NEXT_RPKT) {
      // every lane runs the loop over the packet's steps; before a step that evaluates the continuum opacity (do_rpkt_step()'s own conditions:
      // rpkt_step_evaluates_chi) the WAVE evaluates it -- the window's kept continua side by side, their terms added by the packet's lane in
      // the sequential loop's order (physics.h chi_bf_gammacontr_wave) -- and the step finds it in the packet's cache
      Chi x;
      if (owner) chi_load(env.P, pi, p, x);
      bool go = owner && rpkt_can_continue(p, ts_end);
#ifdef ARTIS_PROFILE_TAIL
      const long long tpr0 = clock64();
#endif
      while (__builtin_amdgcn_readfirstlane(__shfl((int)go, 0)) != 0) {
#if ARTIS_TAIL_WAVE_CHI
        int cc = 0;
        const bool need = go && rpkt_step_evaluates_chi(env, p, x, &cc);
        if (__builtin_amdgcn_readfirstlane(__shfl((int)need, 0)) != 0) {
          const int cw = __builtin_amdgcn_readfirstlane(__shfl(cc, 0));
          const double nuw = wave_bcast(p.nu_cmf, 0);
          const int piw = __builtin_amdgcn_readfirstlane(__shfl(pi, 0));
          chi_rpkt_cont_wave(env, nuw, x, cw, (int64_t)piw, owner);
        }
#endif
        if (go) {
          go = rpkt_iter(env, p, pi, x);
#ifdef ARTIS_PROFILE_TAIL
          tp_n[0]++;
#endif
        }
      }
#ifdef ARTIS_PROFILE_TAIL
      tp[1] += clock64() - tpr0;
#endif
      if (owner) chi_store(env.P, pi, p, x);
    } else if (kind_w == NEXT_MA || kind_w == NEXT_KPKT) {
      if (owner) {
        MACtx k = ma_ctx(env, p);
        bool go = thermal_can_continue(p, ts_end);
#ifdef ARTIS_PROFILE_TAIL
        const long long tpt0 = clock64();
        while (go) {
          (void)thermal_iter(env, p, pi, k, &go);
          tp_n[1]++;
        }
        tp[2] += clock64() - tpt0;
#else
        while (go) (void)thermal_iter(env, p, pi, k, &go);
#endif
      }
    } else if (kind_w == NEXT_BB) {
      TAILP(3, if (owner) (void)advance_blackbody(env, p, pi));
    } else {
      break;
    }
  }
#ifdef ARTIS_PROFILE_TAIL
  if (owner) {
    const bool longwave = (tp[0] + tp[1] + tp[2] + tp[3]) > (1LL << 25);
    for (int i = 0; i < 4; i++) {
      atomicAdd(&lstats[42 + i], (stat_t)(tp[i] >> 4));
      if (longwave) atomicAdd(&lstats[50 + i], (stat_t)(tp[i] >> 4));
    }
    atomicAdd(&lstats[46], (stat_t)tp_n[0]);
    atomicAdd(&lstats[47], (stat_t)tp_n[1]);
    if (longwave) {
      atomicAdd(&lstats[54], (stat_t)1);
      atomicAdd(&lstats[55], (stat_t)tp_n[0]);
      atomicAdd(&lstats[56], (stat_t)tp_n[1]);
    }
  }
#endif
  if (owner) {
    pkt_store(env.P, pi, p);
    cellindex = p.cellindex;
    nu_cmf = p.nu_cmf;
  }
  append_by_kind(kind, pi, cellindex, nu_cmf, next);
  __syncthreads();
  if (threadIdx.x < ARTIS_NSTATS && lstats[threadIdx.x] != 0) atomicAdd(&gstats[threadIdx.x], lstats[threadIdx.x]);
}

// slow path: the rare bound-free actions (rate coefficients with exp(), adaptive Gauss-Kronrod frequency sampling)
// (round 6, with the build's machine LICM off: the classic options' slow path fits 167 VGPRs with 2 spilled, i.e. 3 waves per SIMD instead of 2;
// the builds with interpolated cross-sections / the nebular family need 170-256 and keep what the compiler chooses)
#ifndef ARTIS_SLOW_EU
// (the nebular family's -- NLTE populations, the non-thermal channels -- takes 256 VGPRs + 22 AGPRs alone at one wave per SIMD; held to 256 it
// spills 48 and runs two: k_slow 70.6 -> 51.4 ms, nltenebular step 1018 -> 1001 ms (profiles/r06/ab_slow.txt); kilonova_lte: no difference)
#define ARTIS_SLOW_EU (ARTIS_OPT_PHIXS_CLASSIC_NO_INTERPOLATION ? 3 : (ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON ? 2 : 1))
#endif
__global__ void __launch_bounds__(BLOCK, ARTIS_SLOW_EU) k_slow(Env env, const int32_t *list, int32_t n, Lists next, unsigned long long *gstats) {
  __shared__ stat_t lstats[ARTIS_NSTATS];
  __shared__ double lds_scalars[ARTIS_NSCALARS];
  if (threadIdx.x < ARTIS_NSTATS) lstats[threadIdx.x] = 0;
  scalars_begin(env, lds_scalars);
  __syncthreads();
  env.stats = lstats;
  env.ma_concurrent_fill = 1;  // this kernel's waves fill pool records while others read them: published places are read with acquire (physics.h ma_rowtab_acquire)
  const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  int kind = NEXT_DONE;
  int32_t pi = 0, cellindex = 0;
  double nu_cmf = 0.;
  // A free-bound emission's frequency (select_continuum_nu ratecoeff.cc:563: up to a hundred adaptive 31-point quadratures) is selected by
  // the WAVE (physics.h FbSel): the action is first run on a copy of the packet up to the selection, whose arguments are recorded; the
  // wave's lanes evaluate the abscissae of each rule side by side for one recorded selection after the other; then the action runs for real.
  // (Round 4: one lane per emission while the other lanes of its wave waited -- lane utilisation 0.11; ARTIS_SLOW_WAVE_FB=0 builds that form.)
  const bool mine = tid < n;
  Pkt p;
  FbSel sel;
  sel.mode = 1;
  sel.valid = false;
  sel.element = sel.lowerion = sel.lower = sel.t = 0;
  sel.T_e = 0.f;
  sel.zrand = sel.nu = 0.;
  if (mine) {
    pi = list[tid];
    pkt_load(env.P, pi, p);
#ifdef ARTIS_PROFILE_SLOW
    atomicAdd(&lstats[48 + (p.pend < 9 ? p.pend : 9)], (stat_t)1);  // (-DARTIS_PROFILE_SLOW: the slow-path visits by kind of pending action, stats slots 48..57)
#endif
  }
#if ARTIS_SLOW_WAVE_FB
  {
    // (only where a few lanes of the wave have an emission: the wave takes them one after the other, each ~9x faster than a lane; with many --
    // the nltenebular family, where a bound-free action is pending in nearly every round -- the lanes' own selections side by side are faster:
    // k_slow 83 ms with the wave's form throughout against 68 ms)
    const bool isfb = mine && slow_selects_continuum_nu(p);
    const int nfb = __popcll(__ballot(isfb));
    if (nfb > 0 && nfb <= ARTIS_SLOW_WAVE_FB_MAX) {
      if (isfb) {
        Pkt t = p;
        (void)advance_slow(env, t, pi, &sel);
      }
      fbsel_wave(env, sel);
    }
  }
#endif
  {  // a cold level's record (PEND_MA_FILL): claimed by the lane, filled by the wave, published by the lane
    int fc = 0, ful = 0;
    int32_t funit = 0;
    const bool fill = ARTIS_MA_WAVE_FILL && mine && p.pend == PEND_MA_FILL && ma_slow_fill_claim(env, p, &fc, &ful, &funit);
    ma_fill_wave(env, fill, fc, ful);
    if (fill) ma_slow_fill_publish(env, fc, ful, funit);
  }
  if (mine) {
    kind = advance_slow(env, p, pi, sel.valid ? &sel : nullptr);
    pkt_store(env.P, pi, p);
    cellindex = p.cellindex;
    nu_cmf = p.nu_cmf;
  }
  append_by_kind(kind, pi, cellindex, nu_cmf, next);
  __syncthreads();
  scalars_flush(env);
  if (threadIdx.x < ARTIS_NSTATS && lstats[threadIdx.x] != 0) atomicAdd(&gstats[threadIdx.x], lstats[threadIdx.x]);
}

// blackbody emission of pre-k-packets and of k-packets in grey cells: one do_kpkt_blackbody() per packet
__global__ void __launch_bounds__(BLOCK) k_blackbody(Env env, const int32_t *list, int32_t n, Lists next, unsigned long long *gstats) {
  __shared__ stat_t lstats[ARTIS_NSTATS];
  if (threadIdx.x < ARTIS_NSTATS) lstats[threadIdx.x] = 0;
  __syncthreads();
  env.stats = lstats;
  const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  int kind = NEXT_DONE;
  int32_t pi = 0, cellindex = 0;
  double nu_cmf = 0.;
  if (tid < n) {
    pi = list[tid];
    Pkt p;
    pkt_load_thermal(env.P, pi, p);
    kind = advance_blackbody(env, p, pi);
    pkt_store_thermal(env.P, pi, p);
    cellindex = p.cellindex;
    nu_cmf = p.nu_cmf;
  }
  append_by_kind(kind, pi, cellindex, nu_cmf, next);
  __syncthreads();
  if (threadIdx.x < ARTIS_NSTATS && lstats[threadIdx.x] != 0) atomicAdd(&gstats[threadIdx.x], lstats[threadIdx.x]);
}

#if ARTIS_OPT_VPKT_ON
// Virtual packets (vpkt.cc): the emissions and electron scatterings the launch before recorded (physics.h trace_vpkts), one
// lane per (event, observer direction): the optical depths of every opacity choice along the ray to the grid's edge, then
// the attenuated energy into that observer's spectrum (f64 atomics). Nothing of the real packets is read or written.
// Persistent and work-pulling like the propagation kernels: a lane whose ray has ended (left the grid, absorbed beyond tau_max, met a
// thick cell) takes the next (event, observer) pair at once -- rays cross 1 ... 50 cells, and with one ray per lane from start to end a
// wave lasted as long as its longest.
#ifndef ARTIS_VPKT_WGS
#define ARTIS_VPKT_WGS 4
#endif
#ifndef ARTIS_VPKT_TB
#define ARTIS_VPKT_TB 768  // threads of the one workgroup per CU of the form with the continuum table in LDS: 3 waves/SIMD at 168 VGPRs.
                          // Virtual-packet bench (1e6 packets, t = 5 d): 1470 / 1284 / 1290 ms per step at 1024 / 768 / 512 (MI355X, round 4)
#endif
constexpr int VPKT_CHUNKS = 2048;
// CONT_LDS (round 4): the static continuum table in LDS as in k_rpkt -- every cell a ray crosses sums the bound-free opacity
// (chi_bf_gammacontr) -- with ONE workgroup of 1024 threads per CU instead of four of 256 (the same 4 waves/SIMD).
template <bool CONT_LDS, int TB>
__global__ void __launch_bounds__(TB, (CONT_LDS ? 1 : ARTIS_VPKT_WGS)) k_vpkt(Env env, unsigned long long *gstats, int32_t *cursors) {
  __shared__ stat_t lstats[ARTIS_NSTATS];
  __shared__ ContPack lds_cont[CONT_LDS ? CONT_LDS_MAX : 1];
  if (threadIdx.x < ARTIS_NSTATS) lstats[threadIdx.x] = 0;
  if (CONT_LDS) {
    const D2 *src = (const D2 *)env.M.cont_pack;
    D2 *dst = (D2 *)lds_cont;
    for (int i = threadIdx.x; i < env.M.nbfcontinua * 2; i += TB) dst[i] = src[i];
    env.M.cont_pack = lds_cont;
    env.cont_in_lds = 1;
  }
  __syncthreads();
  env.stats = lstats;
  const int nobs = env.M.vpkt->nobsdirections;
  const int64_t n64 = (int64_t)min(*env.vpkt_count, env.vpkt_cap) * nobs;
  const VpktSeed *queue = env.vpkt_queue;
  if (n64 < 0x7FFFFFF0LL) {
    const int32_t n = (int32_t)n64;
    Puller q;
    puller_init(q, n, VPKT_CHUNKS);
    bool have = false;
    VRay ray;
    while (true) {
      const int32_t idx = pull(q, !have, n, cursors);
      if (idx >= 0) have = vray_begin_seed(env, queue[idx / nobs], idx % nobs, ray);
      if (!__any(have)) {
        if (q.exhausted) break;
        continue;
      }
      if (have) have = vray_step(env, ray);
    }
  } else {  // (more pairs than a list index holds: one ray per lane from start to end)
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n64; i += (int64_t)gridDim.x * TB) {
      const VpktSeed seed = queue[i / nobs];
      vpkt_trace_seed_direction(env, seed, (int)(i % nobs));
    }
  }
  __syncthreads();
  if (threadIdx.x < ARTIS_NSTATS && lstats[threadIdx.x] != 0) atomicAdd(&gstats[threadIdx.x], lstats[threadIdx.x]);
}
#endif

inline int nblocks(int64_t n) { return (int)((n + BLOCK - 1) / BLOCK); }

}  // namespace

// ------------------------------------------------------------------ engine object
struct artis_amd_engine {
  int device = 0;
  ModelOwned own;
  std::vector<void *> model_allocs;
  std::vector<void *> cell_allocs;
  std::vector<void *> cache_allocs;
  DevModel M{};  // device pointers
  DevModel Mh{};  // host view (counts)
  artis_model model_copy{};                        // sizes + level_matransblock_start kept for diagnostics
  std::vector<int32_t> own_matransblock_start;
  DevCells C{};
  DevCache K{};
  DevStep S{};
  DevEst E{};
  bool have_cells = false;
  bool expopac_own = false;  // the expansion-opacity tables are the engine's (calculated at cell-cache population)
  BfEvent *d_bfev = nullptr;     // deferred bound-free estimator updates of one k_rpkt launch (DETAILED_BF builds)
  int32_t *d_bfev_count = nullptr;
  // the sums of the deferred updates, [cell][place in the cell's list of kept continua]: the additions of a record go to
  // neighbouring doubles instead of ~30 separate 64-byte sectors of bfrate_raw; k_bfrate_expand moves them when a call ends
  double *d_bfrate_kept = nullptr;
  bool bfrate_kept_dirty = false;  // a call ended in an error before its sums were moved
  int32_t bfev_cap = 0;
  bool bf_defer = true;          // ARTIS_AMD_BFDEFER=0: add in place inside k_rpkt
  // virtual packets (VPKT_ON builds): the configuration block, and the events one launch records for k_vpkt (at most
  // budget_r per packet on the launch's list: an electron scattering per do_rpkt_step(), one emission per thermal visit)
  VpktConfig *d_vpkt_config = nullptr;
  VpktSeed *d_vpkt_queue = nullptr;
  int32_t *d_vpkt_count = nullptr;
  int32_t vpkt_cap = 0;
  // Cell-cache tiling: the cache rows of `tile_cells` non-empty cells are resident at a time (all of them when they fit
  // the budget: ntiles == 1). With more tiles, update_packets sweeps over them -- populate a tile, advance every packet
  // that sits in one of its cells until it leaves the tile or is done -- until no packet is left (what the reference's
  // single-slot cell cache does cell by cell, update_packets.cc:397-460, 551-621).
  int64_t tile_cells = 0;
  int ntiles = 1;
  int tile_lo = 0, tile_hi = 0;   // one tile: all cells (the range a whole-cache fill works on)
  int tile_valid_lo = -1;         // one tile: 0 once the cache is populated for the current cell state (-1: not)
  // several tiles: which cell's cache each row holds, populated for the current cell state (physics.h Env::krow_tab; make_resident())
  std::vector<int32_t> h_krow;     // [npts_nonempty] row of the cell, -1: not resident
  std::vector<int32_t> h_rowcell;  // [tile_cells] cell of the row, -1: free
  std::vector<int32_t> h_wanted;   // [npts_nonempty] stamp of the last make_resident() that asked for the cell
  int32_t want_stamp = 0;
  int32_t *d_krow = nullptr;
  std::vector<int32_t> h_cell_grid;  // [npts_nonempty] the grid cell of a non-empty cell (the inverse of propcell_nonemptymgi)
  size_t cache_bytes_per_cell = 0;
  int32_t *d_target_level = nullptr;
  double *d_est = nullptr;
  int64_t est_ndoubles = 0;
  int64_t nvspec = 0, nvgrid = 0;  // doubles of the virtual-packet spectra / velocity-grid map at the end of the block
  unsigned long long *d_stats = nullptr;
  int32_t *d_err = nullptr;      // (the slot after the list counters: one copy brings both to the host)
  int32_t *h_counts = nullptr;   // pinned: the host's copy of the list counters and the error flag after every launch
  // packets
  int64_t npackets = -1;           // -1: no resident population
  void *d_pkt = nullptr;           // the three record arrays of the resident population (tables.h PktStore)
  void *d_pkt_snapshot = nullptr;
  size_t pkt_bytes = 0;
  PktStore P{};
  artis_packet *d_aos = nullptr;
  int64_t aos_capacity = 0;
  bool aos_valid = false;  // d_aos still holds the caller's structs of the resident population (set by the upload)
  int32_t *d_lists[NEXT_NKINDS][2] = {};      // per kind: current and alternate work list
  int32_t *d_keys[NEXT_NKINDS][2] = {};       // ... and the sort keys of their entries
  int32_t *d_sorted = nullptr;                // counting-sort output
  int32_t *d_perm = nullptr;                  // record slot -> index in the caller's packet array (k_aos_to_rec)
  bool use_perm = false;                      // d_perm describes the resident population
  bool slot_order_by_cell = true;             // ARTIS_AMD_SLOTSORT=0: slots in the caller's order
  int32_t *d_hist = nullptr;                  // [ngrid * SORT_NUBINS + 1]
  int32_t *d_tiles = nullptr;                 // scan tile totals
  int32_t *d_count = nullptr;                 // [NEXT_NKINDS] current-list counts, [NEXT_NKINDS] alternate-list count
  int32_t *d_cursors = nullptr;               // [MAX_CHUNKS] chunk cursors of the running pull kernel
  int ncu = 256;
  double *d_gamma_ws = nullptr;   // per-packet groundcont_gamma_contr lists (physics.h Env)
  int32_t *d_gamma_gi = nullptr;
  int32_t *d_gamma_n = nullptr;
  int64_t ws_capacity = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  double last_propagate_ms = 0.;
  double kms[NEXT_NKINDS] = {};       // summed launch durations per kind of the last update_packets_device call
  double kms_tail = 0.;               // ... and of the tail kernel
  // (builds with the detailed bound-free estimators -- the nltenebular family -- alternate twice as often between the
  // kernels and their r-packet steps are ~3x as heavy: measured optimum 16384 / 4 / 1024 instead of 4096 / 8 / 2048,
  // nltenebular step 1801 -> 1690 ms, profiles/r03/neb_sweep*.txt)
  int tail_max = ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON ? 16384 : 4096;  // r-packets + thermal packets left at which k_tail takes over (ARTIS_AMD_TAIL; 0 = never)
  bool tail_always = false;           // ... also for a population that starts below it (ARTIS_AMD_TAIL_ALWAYS=1)
  int64_t klaunches[NEXT_NKINDS] = {};
  int64_t kthreads[NEXT_NKINDS] = {};
  int64_t last_nlaunches = 0;
  // tiled runs: sweeps over the tiles, tile fills and their summed time, packets listed per (sweep, tile) of the last call
  // the cells a fill of a tiled cache works on (make_resident()). Sparse fills: a visit for which few packets wait makes the cells in which they
  // wait (and the cells around those) resident instead of a whole window. ARTIS_AMD_SPARSE_FILL=0: whole windows.
  int32_t *d_fill_cells = nullptr;
  bool sparse_fill = true;
  int64_t sparse_max_listed = 16384;  // ... for visits that list at most this many packets (ARTIS_AMD_SPARSE_MAX; 512 in round 3:
                                      // 4 tiles 3327 / 3205 / 3188 ms at 512 / 4096 / 16384, with parked tails 3222 / 3123 / 2982)
  int64_t last_sparse_fills = 0, last_cells_filled = 0;
  bool park_tails = true;     // ARTIS_AMD_TILE_PARK=0: every visit of a tile runs its packets to their end (rounds 2-3)
  // ARTIS_AMD_TILE_PARK_AT: packets left of a larger visit at which it parks them (0 / <= tail_max: round 4's rule, at the tail kernel's
  // threshold). Measured on the headline at a quarter of its cache (4 tiles, adaptive windows; profiles/r06/tiling.md): 4096 / 32768 / 131072 /
  // 524288 / 2097152 -> 2395 / 2147 / 1971 / 1899 / 1868 ms per step (untiled 760); with this round's rows for sets of cells and record tiers (two tiles
  // of rows with a quarter of the levels hot): 131072 / 524288 / 2097152 / 3145728 / 4194304 -> 1550 / 1532 / 1504 (1485 on the box of the last two) / 1448 / 1452 ms;
  // three tiles at 11600 MB: 2097152 / 3145728 / 4194304 -> 1590 / 1573 / 1670 ms
  int64_t park_at = 3145728;
  int64_t last_parked = 0;
  int64_t last_pool_resets = 0;  // times the pool of on-demand records was emptied because it was used up (this call)
  int64_t last_pool_used = 0, last_pool_cap = 0;  // units (128 B) of the pool in use at the end of the last call / the pool's size
  double ma_hotfrac = 1.;        // the share of every ion's levels that has a static record (given, or chosen by engine_fill from the cache budget)
  bool vpkt_cont_lds = true;  // ARTIS_AMD_VPKT_CONTLDS=0: k_vpkt reads the continuum table from memory (four workgroups of 256 per CU)
  bool tile_zigzag = false;  // ARTIS_AMD_TILE_ZIGZAG=1: sweeps alternate their direction (measured slower: profiles/r03/tiling.md)
  // Adaptive tiles (round 6; ARTIS_AMD_TILE_ADAPT=0: the fixed ranges of rounds 2-5). WHICH cells are resident is chosen before every visit from
  // the number of packets that wait in every cell (k_count_waiting): the window of tile_cells consecutive cells in which most packets wait
  // (ARTIS_AMD_TILE_BLOCK=0), or the blocks of tile_block consecutive cells in which most wait, wherever they lie (a tile = a SET of cells: rows are
  // addressed through Env::krow_tab). A cell that is resident and still wanted keeps its row -- only the cells that are new to the set are filled.
  // The first visits take the densest parts of the ejecta; afterwards the waiting packets sit on both sides of the earlier sets' edges, and a set laid
  // across such an edge lets them cross it freely instead of waiting once per crossing and fixed tile.
  bool tile_adapt = true;
  int64_t tile_block = 0;
  bool pool_keep = true;
  int32_t *d_waiting = nullptr;  // [npts_nonempty + 1] packets waiting per cell; the last entry: packets that need no row
  std::vector<int32_t> h_waiting;
  int64_t last_visits = 0;
  int64_t last_sweeps = 0, last_tile_fills = 0, last_listed = 0;
  double last_fill_ms = 0.;
  // do_rpkt_step() calls per packet per launch. 8 in rounds 2-3; with the r-packet kernel's reads requested ahead (round 4) the list's order
  // -- sorted by cell and frequency before every launch -- is worth more than the launches saved: 857 / 855 / 851 / 846 / 860 / 881 ms per step
  // at 8 / 6 / 5 / 4 / 3 / 2 (classic; kilonova_lte 859 / 844 / 858 at 8 / 4 / 3; nltenebular, 4 since round 3: 1199 / 1204 / 1235 at 4 / 3 / 2)
  // (8 stays where a step is cheap and launches are not: the expansion-opacity build with thermalising bound-bound events, 520 vs 567 ms,
  // and the virtual-packet builds, whose every launch is followed by k_vpkt: 1250 vs 1268 ms)
  int budget_r = (ARTIS_OPT_RPKT_BB_THERMALISATION || ARTIS_OPT_VPKT_ON) ? 8 : 4;
  int budget_g = ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON ? 32 : 64;      // ... of a gamma packet (k_gamma)
  int budget_t = ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON ? 1024 : 2048;  // macro-atom transitions / k-packet steps per packet per launch
  // A launch lasts as long as its slowest packet, and a list that does not fill the GPU any more (the last tenth of a
  // timestep's rounds) is bound by that alone. Smaller budgets for such lists (ARTIS_AMD_SMALL_LIST, ARTIS_AMD_BUDGET_T_SMALL /
  // _R_SMALL; 0 = off, the default) were measured: classic 1160 -> 1182..1258 ms per step (the extra rounds cost more than
  // the shorter launches save), nltenebular 1500 -> 1481 (profiles/r03/small_list_budgets.txt).
  int small_list = 300000;
  int budget_t_small = 0;
  int budget_r_small = 0;
  // ... and after the launch's list is used up (ARTIS_AMD_DRAIN_T; 0 = off), in launches of at least drain_min_list packets
  int drain_t = 48;
  int drain_r = 1;       // ... do_rpkt_step() calls after the r-packet list is used up (ARTIS_AMD_DRAIN_R; 0 = off)
  int64_t drain_min_list = 1000000;
  bool sort_lists = true;
  bool sort_nu = true;
  // The r-packet list sorted by (frequency bin, cell) instead of (cell, frequency bin) (round 5; ARTIS_AMD_SORT_NUMAJOR=0: rounds 2-4): the
  // 64 packets a wave holds then lie in one of 32 frequency bins (four per octave) -- their opacity sums run over windows of similar length
  // and their line walks through the same stretch of the line list -- and in ~7 neighbouring cells of it. k_rpkt 263 -> 251 ms with 16 bins,
  // 244 with 32, 245 with 64, 250 with 128 (MI355X, headline workload; profiles/r05/sort_numajor.txt).
  bool sort_numajor = true;
  int sort_cellshift = 0;  // ARTIS_AMD_SORT_CELLSHIFT (with the frequency bin as the major part only)
  bool sort_ma = true;
  // lists with more entries per cell of the tile than this are not sorted (sort_by_key) unless the kernel accumulates its
  // per-cell estimators in LDS; measured crossover of the thermal lists with 1e7 packets: between 20^3 and 30^3 cells
  // (1 250 / 370 per cell). ARTIS_AMD_SORT_MAXPC_R / _T.
  // 512 < cells <= 3072: k_rpkt keeps J / nuJ / ffheating in LDS instead of the continuum table (12^3 grid, 912 cells:
  // k_rpkt 292 -> 215 ms; 14^3, 1472 cells: 249 -> 225 ms). ARTIS_AMD_RPKT_EST_OVER_CONT=0: the table wins the LDS.
  bool rpkt_est_over_cont = true;
  int ma_bins = SORT_MABINS;    // the thermal list sorted by (cell, a hash of the macro-atom's ion) instead of by cell alone: the lanes of a wave
                                // start in the records of one or two ions (ARTIS_AMD_MABINS=1: by cell; step 1060 -> ~1050 ms)
  bool ma_filters = true;       // macro-atom transitions decided on the records' 15-bit filters (ARTIS_AMD_MAFILTERS=0: on the f64 values)
  int dense_lpr = 32;           // k_bfest_dense: lanes per record (64 = a wave per record; ARTIS_AMD_DENSE_LPR)
  bool dense_cont_lds = true;   // k_bfest_dense reads the continuum table from LDS (nltenebular step 1917 -> 1882 ms); ARTIS_AMD_DENSE_CONTLDS=0
  bool cellest_in_lds = true;  // ARTIS_AMD_CELLEST_LDS=0: every estimator add is a global atomic
  bool estcache = true;        // ARTIS_AMD_ESTCACHE=0: k_rpkt without its waves' caches of per-cell accumulators (physics.h Env::estcache)
  int sort_maxpc_r = 20000;
  int sort_maxpc_t = 600;
  // one list chunk per wave instead of one per XCD (artis_engine.hip pull): measured on MI355X, 1e7 packets: k_rpkt -4 %
  // (a wave's lanes share continuum windows and line ranges), k_thermal +30 % (every wave then has its own cells in
  // flight and the L2 working set of macro-atom records triples)
  bool wave_chunks_r = true, wave_chunks_t = false;
  bool cu_chunks_t = false;  // ARTIS_AMD_CUCHUNKS_T=1: k_thermal takes one list chunk per compute unit (HW_REG_HW_ID);
                             // measured +7 %: like every finer chunking it puts more cells in flight per XCD
  bool cont_lds = true;      // k_rpkt keeps the static continuum table (ContPack) in LDS when it fits (ARTIS_AMD_CONTLDS=0: HBM)
  bool line_lds = false;     // ARTIS_AMD_LINELDS=1: the line list's frequencies in LDS instead (k_rpkt<false, .., true>; measured slower)
  int thermal_blocks_per_cu = ARTIS_THERMAL_WAVES;  // tuning: resident k_thermal blocks per CU
  bool thermal_refill = false;  // ARTIS_AMD_REFILL=1: k_thermal_q (walk contexts in per-wave LDS slots, lanes refilled inside the transition loop)
  int tq_low = 48;              // ... its low-water mark of walking lanes (ARTIS_AMD_TQ_LOW)
  bool tq_attr_set = false;     // ... its dynamic-LDS attribute has been set on this engine's device
  int32_t thermal_variants = 0; // which instantiations of the thermal kernel the last call launched (artis_amd_last_thermal_variants)
  bool ma_tables_lds = true;  // k_thermal<1024, true>: the static target tables in LDS when they fit (ARTIS_AMD_MATABLES_LDS=0: in HBM)
  // the population's scratch: the collisional-excitation cooling terms of `pop_batch` cells at a time (k_matrans writes them,
  // k_cooling_chain turns them into running sums, k_collexc_filter into the records' cooling filters; nothing of it is kept)
  double *d_collexc_terms = nullptr;
  int64_t pop_batch = 0;
  bool trace = false;
  uint32_t *d_visit_counts = nullptr;  // -DARTIS_VISIT_COUNTS builds: [cell][level] transitions drawn per record in the last call
  ncclComm_t comm = nullptr;  // created by artis_amd_comm_init(), owned by the engine
};

namespace {

template <typename T>
int upload_array(std::vector<void *> &allocs, const T *host, int64_t count, const T **dev_out) {
  T *d = nullptr;
  const size_t bytes = sizeof(T) * (size_t)(count > 0 ? count : 1);
  HIP_TRY(hipMalloc((void **)&d, bytes));
  allocs.push_back(d);
  if (count > 0) HIP_TRY(hipMemcpy(d, host, sizeof(T) * (size_t)count, hipMemcpyHostToDevice));
  *dev_out = d;
  return ARTIS_OK;
}

int free_all(std::vector<void *> &v) {
  for (void *p : v) (void)hipFree(p);
  v.clear();
  return ARTIS_OK;
}

Env make_env(const artis_amd_engine *e) {
  Env env;
  std::memset(&env, 0, sizeof(env));
  env.M = e->M;
  env.C = e->C;
  env.K = e->K;
  {
    const DevModel &h = e->Mh;
    if (h.ndpop == 0) env.K.line_dpop = nullptr;  // formed on the fly (physics.h line_dpop_at)
    // the pool of on-demand records is one for all resident cells (tables.h "ON-DEMAND RECORDS"): not indexed by cell
    env.ma_pool_cap = (uint32_t)std::min<int64_t>(((int64_t)e->tile_cells * h.ma_pool_slots) / MAPOOL_UNIT, 0x7FFFFFF0LL);
    env.ma_pool_full = e->d_count + (2 * NEXT_NKINDS - 1);  // (a spare slot of the counts the host reads after every launch)
  }
  // rows: one per cell (row = cell), or -- a cache that does not fit -- the rows of the cells that are resident now (make_resident())
  env.krow_tab = e->ntiles > 1 ? e->d_krow : nullptr;
  env.tile_all = e->ntiles > 1 ? 0 : 1;
  env.tile_lo = 0;
  env.tile_hi = e->Mh.npts_nonempty;
  env.fill_cells = nullptr;  // (set by populate_tile() of a tiled cache for its own launches)
  env.nfill = 0;
  env.collexc_terms = e->d_collexc_terms;
  {  // few cells: per-cell estimators accumulate in LDS (physics.h Env::cellest_lds)
    const int nc = e->Mh.npts_nonempty;
    env.cellest_n_t = (e->cellest_in_lds && nc <= THERMAL_CELLEST_CAP) ? nc : 0;
    // between the two caps k_rpkt runs without the continuum table in LDS and keeps the estimators there instead
    env.cellest_n_r = (e->cellest_in_lds && nc <= (e->rpkt_est_over_cont ? RPKT_CELLEST_CAP_NOCONT : RPKT_CELLEST_CAP)) ? nc : 0;
    env.cellest_n_g = (e->cellest_in_lds && nc <= GAMMA_CELLEST_CAP) ? nc : 0;
    env.scalars_in_lds = e->cellest_in_lds ? 1 : 0;
    env.estcache_on = e->estcache ? 1 : 0;
    env.ma_filters_off = e->ma_filters ? 0 : 1;
  }
  env.S = e->S;
  env.E = e->E;
  env.est_stride = 8;
  env.pair_stride = 2;
  env.P = e->P;
  env.stats = nullptr;
  env.gamma_ws = e->d_gamma_ws;
  env.bfev = e->bf_defer ? e->d_bfev : nullptr;
  env.bfev_count = e->d_bfev_count;
  env.bfev_cap = e->bfev_cap;
  env.bfrate_kept = (e->bf_defer && e->ntiles == 1) ? e->d_bfrate_kept : nullptr;
  env.gamma_gi = e->d_gamma_gi;
  env.gamma_n = e->d_gamma_n;
  env.errflag = e->d_err;
  env.vpkt_queue = e->d_vpkt_queue;
  env.vpkt_count = e->d_vpkt_count;
  env.vpkt_cap = e->vpkt_cap;
#ifdef ARTIS_VISIT_COUNTS
  env.visit_counts = e->d_visit_counts;
#endif
  return env;
}

void free_packet_buffers(artis_amd_engine *e) {
  void **singles[] = {&e->d_pkt, &e->d_pkt_snapshot, (void **)&e->d_sorted, (void **)&e->d_perm, (void **)&e->d_gamma_ws,
                      (void **)&e->d_gamma_gi, (void **)&e->d_gamma_n, (void **)&e->d_bfev, (void **)&e->d_bfev_count,
                      (void **)&e->d_vpkt_queue, (void **)&e->d_vpkt_count};
  e->bfev_cap = 0;
  e->vpkt_cap = 0;
  for (void **q : singles) {
    if (*q) (void)hipFree(*q);
    *q = nullptr;
  }
  for (int kind = 0; kind < NEXT_NKINDS; kind++)
    for (int k = 0; k < 2; k++) {
      if (e->d_lists[kind][k]) (void)hipFree(e->d_lists[kind][k]);
      if (e->d_keys[kind][k]) (void)hipFree(e->d_keys[kind][k]);
      e->d_lists[kind][k] = nullptr;
      e->d_keys[kind][k] = nullptr;
    }
  e->npackets = -1;  // nothing resident
  e->use_perm = false;
}

int ensure_packet_buffers(artis_amd_engine *e, int64_t n) {
  if (n == e->npackets && e->d_pkt) return ARTIS_OK;
  free_packet_buffers(e);
  e->pkt_bytes = pkt_store_bytes(n);
  HIP_TRY(hipMalloc(&e->d_pkt, e->pkt_bytes));
  e->P = carve_pkt_store(e->d_pkt, n);
  const size_t listbytes = sizeof(int32_t) * (size_t)(n > 0 ? n : 1);
  for (int kind = 1; kind < NEXT_NKINDS; kind++)
    for (int k = 0; k < 2; k++) {
      HIP_TRY(hipMalloc((void **)&e->d_lists[kind][k], listbytes));
      HIP_TRY(hipMalloc((void **)&e->d_keys[kind][k], listbytes));
    }
  HIP_TRY(hipMalloc((void **)&e->d_sorted, listbytes));
  HIP_TRY(hipMalloc((void **)&e->d_perm, listbytes));
  e->ws_capacity = n > 0 ? n : 1;
  const size_t wsbytes = sizeof(double) * (size_t)(e->Mh.nbfcontinua_ground + 1) * (size_t)e->ws_capacity;
  HIP_TRY(hipMalloc((void **)&e->d_gamma_ws, wsbytes));
  HIP_TRY(hipMalloc((void **)&e->d_gamma_gi, wsbytes / 2));
  HIP_TRY(hipMalloc((void **)&e->d_gamma_n, sizeof(int32_t) * (size_t)e->ws_capacity));
  HIP_TRY(hipMemset(e->d_gamma_n, 0, sizeof(int32_t) * (size_t)e->ws_capacity));
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
  if (e->Mh.nbfcontinua > 0) {  // one k_rpkt launch records at most budget_r updates per packet on its list
    const int64_t cap = std::min<int64_t>((int64_t)(n > 0 ? n : 1) * e->budget_r, 0x7FFFFFF0LL);
    HIP_TRY(hipMalloc((void **)&e->d_bfev, sizeof(BfEvent) * (size_t)cap));
    HIP_TRY(hipMalloc((void **)&e->d_bfev_count, sizeof(int32_t)));
    HIP_TRY(hipMemset(e->d_bfev_count, 0, sizeof(int32_t)));
    e->bfev_cap = (int32_t)cap;
  }
#endif
#if ARTIS_OPT_VPKT_ON
  {
    const int64_t cap = std::min<int64_t>((int64_t)(n > 0 ? n : 1) * std::max(e->budget_r, 1), 0x7FFFFFF0LL);
    HIP_TRY(hipMalloc((void **)&e->d_vpkt_queue, sizeof(VpktSeed) * (size_t)cap));
    HIP_TRY(hipMalloc((void **)&e->d_vpkt_count, sizeof(int32_t)));
    HIP_TRY(hipMemset(e->d_vpkt_count, 0, sizeof(int32_t)));
    e->vpkt_cap = (int32_t)cap;
  }
#endif
  e->npackets = n;  // committed only now: a failed allocation above leaves "nothing resident" (npackets == -1)
  return ARTIS_OK;
}

int ensure_aos(artis_amd_engine *e, int64_t n) {
  if (n <= e->aos_capacity && e->d_aos) return ARTIS_OK;
  if (e->d_aos) (void)hipFree(e->d_aos);
  e->d_aos = nullptr;
  e->aos_valid = false;
  HIP_TRY(hipMalloc((void **)&e->d_aos, sizeof(artis_packet) * (size_t)(n > 0 ? n : 1)));
  e->aos_capacity = n;
  return ARTIS_OK;
}

}  // namespace

namespace {
int engine_fill(artis_amd_engine *e, const artis_model *model);
// scratch of the cell-cache population (allocated after the cache rows): ~2 GB, ARTIS_AMD_POP_SCRATCH_MB
double pop_scratch_mb() {
  double mb = 2048.;
  if (const char *b = std::getenv("ARTIS_AMD_POP_SCRATCH_MB")) mb = std::max(1., std::atof(b));
  return mb;
}
// Bytes the cell-cache rows may take: 80 % of what is free once the population's scratch and a head-room for everything that is
// allocated later (ARTIS_AMD_CACHE_HEADROOM_MB, default 0: the packets at ~1 KB each with their work lists, the caller's structs
// and a snapshot -- 10 GB at 1e7 packets -- fit the remaining fifth of a 288 GB card; a smaller GPU, or two engines on one
// device, set it) are taken off; ARTIS_AMD_CACHE_BUDGET_MB overrides the lot. One rule for the tile count and for the decision
// to drop line_dpop.
double cache_budget_bytes(size_t free_b) {
  if (const char *b = std::getenv("ARTIS_AMD_CACHE_BUDGET_MB")) return std::atof(b) * 1048576.0;
  double headroom_mb = 0.;
  if (const char *b = std::getenv("ARTIS_AMD_CACHE_HEADROOM_MB")) headroom_mb = std::max(0., std::atof(b));
  return std::max(0., 0.8 * ((double)free_b - (pop_scratch_mb() + headroom_mb) * 1048576.0));
}

// RCCL entry points, resolved at run time from the librccl the process already has (a host that links RCCL itself, or
// torch's own copy in bench.py) or else from the ROCm installation: the engine library itself carries no link-time
// dependency on a particular librccl, and a communicator handed in by the caller is used with the library it came from.
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;  // optional: artis_amd_comm_count()
  bool ok = false;
};
const RcclApi &rccl_api() {
  static RcclApi api = [] {
    RcclApi a;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (h) break;
    }
    if (!h)
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
    if (!h) return a;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString;
    return a;
  }();
  return api;
}
#define RCCL_TRY(expr)                                                                             \
  do {                                                                                             \
    ncclResult_t _r = (expr);                                                                      \
    if (_r != ncclSuccess) {                                                                       \
      g_last_error = std::string(#expr) + ": " + rccl_api().GetErrorString(_r);                    \
      return ARTIS_ERR_RCCL;                                                                       \
    }                                                                                              \
  } while (0)
}  // namespace

extern "C" {

const char *artis_amd_last_error(void) { return g_last_error.c_str(); }
int artis_amd_abi_version(void) { return 6; }
const char *artis_amd_options_preset(void) {
#if defined(ARTIS_PRESET_NAME)  // given by the build (artis_amd/build.py): the presets of the reference's CI option sets
  return ARTIS_PRESET_NAME;
#elif defined(ARTIS_PRESET_NLTENEBULAR_LINEEST)
  return "nltenebular_lineest";
#elif defined(ARTIS_PRESET_CHRISTINENONTHERMAL)
  return "christinenonthermal";
#elif defined(ARTIS_PRESET_NLTEPHOTOSPHERIC)
  return "nltephotospheric";
#elif defined(ARTIS_PRESET_NLTEWITHOUTNONTHERMAL)
  return "nltewithoutnonthermal";
#elif defined(ARTIS_PRESET_NLTENEBULAR)
  return "nltenebular";
#elif defined(ARTIS_PRESET_KILONOVA_EXPOPAC)
  return "kilonova_expopac";
#elif defined(ARTIS_PRESET_CLASSIC_EXPOPAC_THERM)
  return "classic_expopac_therm";
#elif defined(ARTIS_PRESET_KILONOVA_GAMMA_GREY)
  return "kilonova_gamma_grey";
#elif defined(ARTIS_PRESET_CLASSIC_GAMMA_XCOM)
  return "classic_gamma_xcom";
#elif defined(ARTIS_PRESET_KILONOVA_GAMMA_BARNES)
  return "kilonova_gamma_barnes";
#elif defined(ARTIS_PRESET_KILONOVA_GAMMA_WOLLAEGER)
  return "kilonova_gamma_wollaeger";
#elif defined(ARTIS_PRESET_KILONOVA_GAMMA_GUTTMAN)
  return "kilonova_gamma_guttman";
#elif defined(ARTIS_PRESET_KILONOVA_GAMMAPRODUCTS)
  return "kilonova_gammaproducts";
#elif defined(ARTIS_PRESET_KILONOVA_BARNES)
  return "kilonova_barnes";
#elif defined(ARTIS_PRESET_KILONOVA_WOLLAEGER)
  return "kilonova_wollaeger";
#elif defined(ARTIS_PRESET_KILONOVA_LTE)
  return "kilonova_lte";
#else
  return "classic";
#endif
}
size_t artis_amd_sizeof_packet(void) { return sizeof(artis_packet); }

int artis_amd_engine_create(const artis_model *model, int device, artis_amd_engine **out) {
  if (!model || !out) {
    g_last_error = "null argument";
    return ARTIS_ERR_ARG;
  }
  if (model->gridtype != ARTIS_GRID_CARTESIAN3D && model->gridtype != ARTIS_GRID_SPHERICAL1D &&
      model->gridtype != ARTIS_GRID_CYLINDRICAL2D) {
    g_last_error = "unknown grid type";
    return ARTIS_ERR_UNSUPPORTED;
  }
  // the per-packet list of ground-continuum contributions (physics.h chi_bf_gammacontr) relies on the order the
  // reference builds these tables in: groundcont_nu_edge rising (input.cc:802) and, with allcont sorted by nu_edge
  // (input.cc:892), a non-decreasing nearest-edge index (input.cc:703)
  for (int i = 1; i < model->nbfcontinua_ground; i++) {
    if (model->groundcont_nu_edge[i] < model->groundcont_nu_edge[i - 1]) {
      g_last_error = "groundcont_nu_edge is not in ascending order";
      return ARTIS_ERR_ARG;
    }
  }
  for (int i = 0, last = -1; i < model->nbfcontinua; i++) {
    const int gi = model->allcont_groundcontestimindex[i];
    if (gi < 0) continue;
    if (gi < last || gi >= model->nbfcontinua_ground) {
      g_last_error = "allcont_groundcontestimindex does not rise with nu_edge";
      return ARTIS_ERR_ARG;
    }
    last = gi;
  }
  for (int i = 0; i < model->nlevels; i++) {  // what a packed macro-atom transition target can hold (tables.h MaTarget)
    if (model->level_ndowntrans[i] >= MATGT_MAX_NTRANS || model->level_nuptrans[i] >= MATGT_MAX_NTRANS) {
      g_last_error = "a level has more transitions than a packed transition target can describe";
      return ARTIS_ERR_UNSUPPORTED;
    }
  }
  // k_cooling_tail reads what the entries of an ion's cooling list after the collisional excitations are from the list itself
  // (coolinglist_type / _level / _phixstargetindex): they must be in the order calculate_cooling_rates_ion() writes them
  // (kpkt.cc:122-190: the collisional ionisations level by level and target by target, then the bound-free terms likewise)
  for (int el = 0; el < model->nelements && model->ncoolingterms > 0; el++)
    for (int ion = 0; ion < model->elem_nions[el]; ion++) {
      const int ui = model->elem_uniqueionindexstart[el] + ion;
      int k = ((model->elem_lowest_ionstage[el] + ion - 1) > 0) ? 1 : 0;
      const int start = model->ion_uniquelevelindexstart[ui];
      for (int l = 0; l < model->ion_nlevels[ui]; l++) k += (model->level_nuptrans[start + l] > 0) ? 1 : 0;
      bool ok = true;
      if (ion < model->elem_nions[el] - 1 && model->nbfcontinua > 0) {
        const int off = model->ion_coolingoffset[ui];
        for (int pass = 0; pass < 2 && ok; pass++)
          for (int l = 0; l < model->ion_nlevels_ionising[ui] && ok; l++)
            for (int t = 0; t < model->level_nphixstargets[start + l] && ok; t++, k++)
              ok = k < model->ion_ncoolingterms[ui] && model->coolinglist_type[off + k] == (pass == 0 ? ARTIS_COOLING_COLLION : ARTIS_COOLING_FREEBOUND) &&
                   model->coolinglist_level[off + k] == l && model->coolinglist_phixstargetindex[off + k] == t;
      }
      if (!ok || k != model->ion_ncoolingterms[ui]) {
        g_last_error = "the cooling list of an ion is not in the order of calculate_cooling_rates_ion()";
        return ARTIS_ERR_ARG;
      }
    }
  {
    // alltrans is laid out level by level, downward then upward transitions, without gaps (input.cc:565): k_matrans takes a block's
    // extent from its first and last segment, ma_entry_pos() an entry's place from its offset, and the records' filter lines are
    // counted from a level's first entry
    int64_t next = 0;
    for (int ul = 0; ul < model->nlevels; ul++) {
      if (model->level_alltrans_startdown[ul] != next || model->level_ndowntrans[ul] < 0 || model->level_nuptrans[ul] < 0) {
        g_last_error = "alltrans is not contiguous in level order (level_alltrans_startdown[i+1] == startdown[i] + ndown[i] + nup[i])";
        return ARTIS_ERR_ARG;
      }
      next += (int64_t)model->level_ndowntrans[ul] + model->level_nuptrans[ul];
    }
    if (next != model->nalltrans) {
      g_last_error = "the last level's transitions do not end at nalltrans";
      return ARTIS_ERR_ARG;
    }
  }
  for (int i = 0; i < model->nions; i++) {  // DevModel::alltrans_tlevel16: a target level within its ion in 16 bits
    if (model->ion_nlevels[i] > 65535) {
      g_last_error = "an ion has more levels than a 16-bit target level can describe";
      return ARTIS_ERR_UNSUPPORTED;
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_last_error = "no HIP device: the artis_amd engine has no CPU path";
    return ARTIS_ERR_NODEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  artis_amd_engine *e = new artis_amd_engine();
  e->device = device;
  const int rc = engine_fill(e, model);
  if (rc != ARTIS_OK) {  // release whatever was allocated before the failing call (g_last_error is already set)
    artis_amd_engine_destroy(e);
    return rc;
  }
  *out = e;
  return ARTIS_OK;
}

}  // extern "C"

namespace {
int engine_fill(artis_amd_engine *e, const artis_model *model) {
  const int device = e->device;
  {
    // Macro-atom record tiers (tables.h "ON-DEMAND RECORDS"): every level a static record while the whole cell cache fits one tile; when it
    // does not (even without line_dpop), static records for the lowest ARTIS_AMD_MA_HOTFRAC (0.3) of every ion's levels and a pool of
    // ARTIS_AMD_MA_POOLFRAC (0.15) of the rest for the cold levels packets reach. Either variable set: taken as given.
    // (pool share: 0.25 in round 5. Measured on the 4e5-line set, 50^3 / 1e7, round 6 -- artis_amd_last_pool_usage(): a step leaves 48 % of a quarter-share
    // pool in use; with 0.15 the tier search affords hot 0.25 instead of 0.15 and the step takes 15.9 s instead of 16.6 (pool 64 % used); with 0.10 hot 0.30,
    // 15.7 s, 87 % used -- too close to a pool that is used up and emptied. profiles/r06/pool_share.txt)
    double hot = 1., pool = 0.15;
    const bool given = std::getenv("ARTIS_AMD_MA_HOTFRAC") != nullptr;
    ma_tiers_from_env(&hot, &pool);
    e->Mh = make_host_model_view(*model, e->own, hot, pool);
    e->ma_hotfrac = hot;
    if (!given) {
      size_t free_b = 0, total_b = 0;
      HIP_TRY(hipMemGetInfo(&free_b, &total_b));
      auto tiles_needed = [&]() -> int64_t {
        size_t per_cell = 0;
#define SZ(f, T, per) per_cell += sizeof(T) * (size_t)(per);
        ARTIS_CACHE_ARRAYS(SZ, e->Mh)
#undef SZ
        per_cell -= sizeof(double) * (size_t)e->Mh.ndpop;  // (dropped first when the cache does not fit: below)
        const int64_t fit = std::max<int64_t>(1, (int64_t)(cache_budget_bytes(free_b) / (double)(per_cell > 0 ? per_cell : 1)));
        return (model->npts_nonempty + fit - 1) / fit;
      };
      // The largest hot share that lets the whole cache be resident (the fewer cold levels, the fewer first visits pay a fill). If none does, the
      // cache is tiled, and the share is the largest one that needs the FEWEST tiles: smaller rows, more cells resident at a time (round 6; the headline
      // forced to a quarter of its cache: 1875 ms with four tiles of static rows, 1532 ms with two tiles of rows with a quarter of the levels hot --
      // profiles/r06/tiling.md. Round 5 kept the rows static in that case: every refill of a tile emptied the pool then, and a tiled run on on-demand
      // records paid its fills again and again, 3.7 s against 3.0 s; now a fill leaves the pool and the rows of the cells that stay alone.)
      // (round 6: in steps of 0.05 below one half -- with the round-5 steps 0.5 / 0.3 / 0.2 / 0.1 a record that grew by a quarter, the fine bytes, sent
      // the 4e5-line set from 0.2 to 0.1 and doubled its fills)
      // (the fewest tiles among the shares >= 0.2: a tile saved by going lower costs more in cold levels filled on demand than the tile did -- at 11600 MB
      // two tiles at hot 0.10 take 1750 ms, three at 0.5 1620; a cache that fits ONE tile at a lower share takes that: one tile beats any tiling)
      int64_t best_nt = tiles_needed(), tiled_nt = best_nt;
      double best_h = 1., cur_h = 1., tiled_h = 1.;
      for (const double h : {0.9, 0.8, 0.7, 0.6, 0.5, 0.45, 0.4, 0.35, 0.3, 0.25, 0.2, 0.15, 0.1, 0.05}) {
        if (best_nt <= 1) break;
        e->Mh = make_host_model_view(*model, e->own, h, pool);
        cur_h = h;
        const int64_t nt = tiles_needed();
        if (nt < best_nt) {
          best_nt = nt;
          best_h = h;
        }
        if (h > 0.199 && nt < tiled_nt) {
          tiled_nt = nt;
          tiled_h = h;
        }
      }
      if (best_nt > 1) best_h = tiled_h;
      if (cur_h != best_h) e->Mh = make_host_model_view(*model, e->own, best_h, pool);
      e->ma_hotfrac = best_h;
    }
  }
  e->model_copy = *model;
  e->own_matransblock_start.assign(model->level_matransblock_start, model->level_matransblock_start + model->nlevels);
  e->model_copy.level_matransblock_start = e->own_matransblock_start.data();
  e->M = e->Mh;
  const DevModel &h = e->Mh;
#define UP(f, T, count)                                                                     \
  {                                                                                         \
    int rc = upload_array<T>(e->model_allocs, h.f, (int64_t)(count), (const T **)&e->M.f);  \
    if (rc != ARTIS_OK) return rc;                                                          \
  }
  ARTIS_MODEL_ARRAYS(UP, h)
#undef UP
#define UPMO(f, T, count)                                                                     \
  {                                                                                           \
    e->M.f = nullptr;                                                                         \
    if (h.f) {                                                                                \
      int rc = upload_array<T>(e->model_allocs, h.f, (int64_t)(count), (const T **)&e->M.f);  \
      if (rc != ARTIS_OK) return rc;                                                          \
    }                                                                                         \
  }
  ARTIS_MODEL_OPTIONAL_ARRAYS(UPMO, h)
#undef UPMO
  if ((ARTIS_OPT_GAMMA_THERMALISATION_SCHEME == ARTIS_GAMMA_WOLLAEGER || ARTIS_OPT_GAMMA_THERMALISATION_SCHEME == ARTIS_GAMMA_GUTTMAN) &&
      !e->M.rho_tmin) {
    g_last_error = "this build integrates gamma-ray column densities: artis_model.rho_tmin is required";
    return ARTIS_ERR_ARG;
  }
  if ((ARTIS_OPT_GAMMA_THERMALISATION_SCHEME == ARTIS_GAMMA_BARNES || ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_BARNES) &&
      !(h.mtot_input > 0. && h.ejecta_kinetic_energy > 0.)) {
    g_last_error = "this build uses a Barnes thermalisation efficiency: artis_model.mtot_input and ejecta_kinetic_energy are required";
    return ARTIS_ERR_ARG;
  }
  if (ARTIS_OPT_USE_XCOM_GAMMAPHOTOION && (!e->M.xcom_elem_start || !e->M.xcom_energy || !e->M.xcom_sigma || !e->M.elem_meannucmass)) {
    g_last_error = "this build has USE_XCOM_GAMMAPHOTOION: artis_model.xcom_elem_start / xcom_energy / xcom_sigma and elem_meannucmass are required";
    return ARTIS_ERR_ARG;
  }
  if (ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON && (!e->M.detailed_lineindices || h.detailed_linecount <= 0)) {
    g_last_error = "this build has DETAILED_LINE_ESTIMATORS_ON: artis_model.detailed_lineindices / detailed_linecount are required";
    return ARTIS_ERR_ARG;
  }
  if (ARTIS_OPT_BFEST_SUBSET && !e->M.allcont_bfestimindex) {
    g_last_error = "this build keeps bound-free estimators for a subset of the continua: artis_model.allcont_bfestimindex / nbfestim are required";
    return ARTIS_ERR_ARG;
  }
  if (ARTIS_OPT_NT_ON && (!e->M.elem_meannucmass || !e->M.ion_nt_sum_q_over_binding)) {
    g_last_error = "this build has NT_ON: artis_model.elem_meannucmass and ion_nt_sum_q_over_binding are required";
    return ARTIS_ERR_ARG;
  }
  for (int a = 0; a < 3; a++) {
    const int ndim = (h.gridtype == ARTIS_GRID_SPHERICAL1D) ? 1 : ((h.gridtype == ARTIS_GRID_CYLINDRICAL2D) ? 2 : 3);  // get_ndim grid.cc:120
    const int64_t cnt = (a >= ndim) ? 1 : h.ncoordgrid[a];
    int rc = upload_array<double>(e->model_allocs, h.coord_pos_min_tmin[a], cnt, &e->M.coord_pos_min_tmin[a]);
    if (rc != ARTIS_OK) return rc;
  }
  // allphixstarget index -> level
  {
    std::vector<int32_t> tl((size_t)(h.nphixstargets_total > 0 ? h.nphixstargets_total : 1), 0);
    for (int ul = 0; ul < h.nlevels; ul++)
      for (int t = 0; t < h.level_nphixstargets[ul]; t++) tl[h.level_phixstargetstart[ul] + t] = ul;
    const int32_t *d = nullptr;
    int rc = upload_array<int32_t>(e->model_allocs, tl.data(), h.nphixstargets_total, &d);
    if (rc != ARTIS_OK) return rc;
    e->d_target_level = (int32_t *)d;
  }
  // per-cell cache: all cells if that fits the budget, else one tile of cells at a time
  const int64_t ncell_all = h.npts_nonempty;
  {
    size_t per_cell = 0;
#define SZ(f, T, per) per_cell += sizeof(T) * (size_t)(per);
    ARTIS_CACHE_ARRAYS(SZ, h)
#undef SZ
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    {
      // The rows of line_dpop (8 bytes per line and cell: 0.9 of the 3.2 MB of a row with 110 860 lines) are dropped when the cell
      // cache does not fit one tile with them and needs fewer tiles without: the line walk then forms a line's population factor from
      // its record and the two level populations (physics.h line_dpop_at: the same expression, the same bits; two more reads per
      // line visited). ARTIS_AMD_DPOP=0 / 1 forces either.
      double budget0 = cache_budget_bytes(free_b);
      const size_t per_without = per_cell - (sizeof(double) * (size_t)h.ndpop);
      auto tiles_of = [&](size_t per) {
        const int64_t fit = std::max<int64_t>(1, (int64_t)(budget0 / (double)(per > 0 ? per : 1)));
        return (ncell_all + fit - 1) / fit;
      };
      bool drop = tiles_of(per_cell) > 1 && tiles_of(per_without) < tiles_of(per_cell);
      if (const char *b = std::getenv("ARTIS_AMD_DPOP")) drop = std::atoi(b) == 0;
      if (drop && h.ndpop > 0) {
        e->Mh.ndpop = 0;
        e->M.ndpop = 0;
        per_cell = per_without;
      }
    }
    e->cache_bytes_per_cell = per_cell;
    double budget = cache_budget_bytes(free_b);
    int64_t fit = (int64_t)(budget / (double)(per_cell > 0 ? per_cell : 1));
    e->tile_cells = std::max<int64_t>(1, std::min<int64_t>(ncell_all > 0 ? ncell_all : 1, fit));
    e->ntiles = (int)((ncell_all + e->tile_cells - 1) / e->tile_cells);
    if (e->ntiles < 1) e->ntiles = 1;
    e->tile_lo = 0;
    e->tile_hi = (int)std::min<int64_t>(ncell_all, e->tile_cells);
    if (ARTIS_OPT_VPKT_ON && e->ntiles > 1) {
      // a virtual packet's ray reads the cache rows of every cell up to the grid's edge (vpkt.cc:183): all of them have to be resident
      g_last_error = "this build has VPKT_ON and the cell cache does not fit one tile (" + std::to_string(per_cell) + " B per cell x " +
                     std::to_string((long long)ncell_all) + " cells): virtual packets need every cell's row resident";
      return ARTIS_ERR_UNSUPPORTED;
    }
  }
  const int64_t nrows = e->tile_cells;  // rows allocated
#define CA(f, T, per)                                                                       \
  {                                                                                         \
    T *d = nullptr;                                                                         \
    HIP_TRY(hipMalloc((void **)&d, sizeof(T) * (size_t)(nrows * (int64_t)(per) + MAREC_SLACK)));    \
    e->cache_allocs.push_back(d);                                                           \
    e->K.f = d;                                                                             \
  }
  ARTIS_CACHE_ARRAYS(CA, h)
#undef CA
  // estimators: one contiguous block [cell][8]{J, nuJ, ff, col, dep_gamma, dep_electron, dep_positron, dep_alpha} | [cell][ground continuum][2]{gamma, bfheat} | scalars
  const int64_t ncell = ncell_all;  // estimators cover every cell
  const int64_t g = h.nbfcontinua_ground > 0 ? h.nbfcontinua_ground : 1;
  // ... | scalars | ([cell][bin][2]{radfieldbin_J, radfieldbin_nuJ}) | (bfrate_raw)]
  const int64_t nbinest = ARTIS_OPT_MULTIBIN_RADFIELD_MODEL_ON ? ncell * ARTIS_OPT_RADFIELDBINCOUNT : 0;
  const int64_t nbfest = ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON ? ncell * (int64_t)h.nbfestim : 0;
  const int64_t nlineest = ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON ? ncell * (int64_t)h.detailed_linecount : 0;
  // ... | (vspecpol | vgrid_flux)]: the observers' spectra of a VPKT_ON build, inside the block the all-reduce covers
  int64_t nvspec = 0, nvgrid = 0;
#if ARTIS_OPT_VPKT_ON
  {
    VpktConfig V;
    if (!make_vpkt_config(*model, V)) {
      g_last_error = "this build has VPKT_ON: artis_model.vpkt_* (the configuration read from vpkt.txt) is required";
      return ARTIS_ERR_ARG;
    }
    HIP_TRY(hipMalloc((void **)&e->d_vpkt_config, sizeof(VpktConfig)));
    e->model_allocs.push_back(e->d_vpkt_config);
    HIP_TRY(hipMemcpy(e->d_vpkt_config, &V, sizeof(V), hipMemcpyHostToDevice));
    e->M.vpkt = e->d_vpkt_config;
    nvspec = (int64_t)ARTIS_VSPEC_TIMEBINS * V.nobsdirections * V.nspectraperobsdir * ARTIS_VSPEC_NUBINS * 3;
    nvgrid = V.vgrid_on ? (int64_t)ARTIS_VGRID_NY * ARTIS_VGRID_NZ * V.grid_nwavelengthranges * V.nobsdirections * 3 : 0;
    e->tail_max = 0;  // k_tail takes a packet through any number of steps in one launch: more events than the queue is sized for
  }
#endif
  e->nvspec = nvspec;
  e->nvgrid = nvgrid;
  e->est_ndoubles = ncell * 8 + 2 * ncell * g + ARTIS_NSCALARS + 2 * nbinest + nbfest + 2 * nlineest + nvspec + nvgrid;
  HIP_TRY(hipMalloc((void **)&e->d_est, sizeof(double) * (size_t)e->est_ndoubles));
  HIP_TRY(hipMemset(e->d_est, 0, sizeof(double) * (size_t)e->est_ndoubles));
  // [cell][8] {J, nuJ, ffheating, colheating, dep_gamma, dep_electron, dep_positron, dep_alpha}: one 64-byte record per cell
  // (physics.h Env::est_stride), then [cell][ground continuum][2] {gamma, bfheating}
  e->E.J = e->d_est;
  e->E.nuJ = e->d_est + 1;
  e->E.ffheatingestimator = e->d_est + 2;
  e->E.colheatingestimator = e->d_est + 3;
  e->E.dep_estimator_gamma = e->d_est + 4;
  e->E.dep_estimator_electron = e->d_est + 5;
  e->E.dep_estimator_positron = e->d_est + 6;
  e->E.dep_estimator_alpha = e->d_est + 7;
  e->E.gammaestimator = e->d_est + 8 * ncell;
  e->E.bfheatingestimator = e->d_est + 8 * ncell + 1;
  e->E.scalars = e->d_est + 8 * ncell + 2 * ncell * g;
  e->E.radfieldbin_J = nbinest ? e->E.scalars + ARTIS_NSCALARS : nullptr;
  e->E.radfieldbin_nuJ = nbinest ? e->E.radfieldbin_J + 1 : nullptr;  // [cell][bin][2] {J, nuJ}
  e->E.bfrate_raw = nbfest ? e->E.scalars + ARTIS_NSCALARS + 2 * nbinest : nullptr;
  e->E.Jb_lu_raw = nlineest ? e->E.scalars + ARTIS_NSCALARS + 2 * nbinest + nbfest : nullptr;
  e->E.Jb_lu_contribcount = nlineest ? e->E.Jb_lu_raw + nlineest : nullptr;
  e->E.vspecpol = nvspec ? e->E.scalars + ARTIS_NSCALARS + 2 * nbinest + nbfest + 2 * nlineest : nullptr;
  e->E.vgrid_flux = nvgrid ? e->E.vspecpol + nvspec : nullptr;
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
  if (nbfest > 0 && h.nbfcontinua > 0 && e->ntiles == 1) {
    const size_t bytes = sizeof(double) * (size_t)ncell * (size_t)h.nbfcontinua;
    HIP_TRY(hipMalloc((void **)&e->d_bfrate_kept, bytes));
    HIP_TRY(hipMemset(e->d_bfrate_kept, 0, bytes));
  }
#endif
  HIP_TRY(hipMalloc((void **)&e->d_stats, sizeof(unsigned long long) * ARTIS_NSTATS));
  HIP_TRY(hipMemset(e->d_stats, 0, sizeof(unsigned long long) * ARTIS_NSTATS));
  HIP_TRY(hipMalloc((void **)&e->d_count, sizeof(int32_t) * (2 * NEXT_NKINDS + 1)));
  HIP_TRY(hipMemset(e->d_count, 0, sizeof(int32_t) * (2 * NEXT_NKINDS + 1)));
  e->d_err = e->d_count + (2 * NEXT_NKINDS);
  HIP_TRY(hipHostMalloc((void **)&e->h_counts, sizeof(int32_t) * (2 * NEXT_NKINDS + 1), hipHostMallocDefault));
  HIP_TRY(hipMalloc((void **)&e->d_cursors, sizeof(int32_t) * (MAX_CHUNKS + 1)));  // + the launch's "list used up" flag
  HIP_TRY(hipMalloc((void **)&e->d_fill_cells, sizeof(int32_t) * (size_t)(ncell_all > 0 ? ncell_all : 1)));
  if (e->ntiles > 1) {  // rows for a set of cells at a time: the table of rows, no cell resident yet
    HIP_TRY(hipMalloc((void **)&e->d_krow, sizeof(int32_t) * (size_t)ncell_all));
    e->h_krow.assign((size_t)ncell_all, -1);
    e->h_wanted.assign((size_t)ncell_all, 0);
    e->h_rowcell.assign((size_t)e->tile_cells, -1);
    HIP_TRY(hipMemset(e->d_krow, 0xFF, sizeof(int32_t) * (size_t)ncell_all));
    e->h_cell_grid.assign((size_t)ncell_all, 0);
    for (int g = 0; g < h.ngrid; g++)
      if (h.propcell_nonemptymgi[g] >= 0) e->h_cell_grid[(size_t)h.propcell_nonemptymgi[g]] = g;
  }
  {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    e->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  HIP_TRY(hipMalloc((void **)&e->d_hist, sizeof(int32_t) * ((size_t)h.ngrid * SORT_NUBINS + 1)));
  HIP_TRY(hipMalloc((void **)&e->d_tiles, sizeof(int32_t) * (((size_t)h.ngrid * SORT_NUBINS) / SCAN_TILE + 2)));
  HIP_TRY(hipEventCreate(&e->ev0));
  HIP_TRY(hipEventCreate(&e->ev1));
  HIP_TRY(hipEventCreate(&e->ev2));
  HIP_TRY(hipEventCreate(&e->ev3));
  if (const char *b = std::getenv("ARTIS_AMD_BUDGET")) {  // tuning / tests: launch budgets never change results
    e->budget_r = std::max(1, std::atoi(b));
    e->budget_t = std::max(1, std::atoi(b));
    e->budget_g = e->budget_r * 8;
  }
  if (const char *b = std::getenv("ARTIS_AMD_BUDGET_R")) e->budget_r = std::max(1, std::atoi(b));
  if (const char *b = std::getenv("ARTIS_AMD_BUDGET_T")) e->budget_t = std::max(1, std::atoi(b));
  if (const char *b = std::getenv("ARTIS_AMD_SMALL_LIST")) e->small_list = std::max(0, std::atoi(b));
  if (const char *b = std::getenv("ARTIS_AMD_BUDGET_T_SMALL")) e->budget_t_small = std::max(0, std::atoi(b));
  if (const char *b = std::getenv("ARTIS_AMD_BUDGET_R_SMALL")) e->budget_r_small = std::max(0, std::atoi(b));
  if (const char *b = std::getenv("ARTIS_AMD_DRAIN_T")) e->drain_t = std::max(0, std::atoi(b));
  if (const char *b = std::getenv("ARTIS_AMD_DRAIN_R")) e->drain_r = std::max(0, std::atoi(b));
  if (const char *b = std::getenv("ARTIS_AMD_DRAIN_MIN")) e->drain_min_list = std::max(0, std::atoi(b));
  if (const char *b = std::getenv("ARTIS_AMD_SLOTSORT")) e->slot_order_by_cell = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_BFDEFER")) e->bf_defer = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_SORT")) e->sort_lists = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_SORT_NU")) e->sort_nu = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_SORT_MA")) e->sort_ma = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_TAIL")) e->tail_max = std::max(0, std::atoi(b));
  if (const char *b = std::getenv("ARTIS_AMD_TAIL_ALWAYS")) e->tail_always = std::atoi(b) != 0;
  if (ARTIS_OPT_VPKT_ON) e->tail_max = 0;  // (see the estimator block: the event queue is sized per split launch)
  if (const char *b = std::getenv("ARTIS_AMD_RPKT_EST_OVER_CONT")) e->rpkt_est_over_cont = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_DENSE_CONTLDS")) e->dense_cont_lds = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_MABINS")) e->ma_bins = (std::atoi(b) > 1) ? SORT_MABINS : 1;
  if (const char *b = std::getenv("ARTIS_AMD_MAFILTERS")) e->ma_filters = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_DENSE_LPR")) e->dense_lpr = (std::atoi(b) == 64) ? 64 : (std::atoi(b) == 16 ? 16 : 32);
  if (const char *b = std::getenv("ARTIS_AMD_CELLEST_LDS")) e->cellest_in_lds = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_ESTCACHE")) e->estcache = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_SORT_MAXPC_R")) e->sort_maxpc_r = std::max(1, std::atoi(b));
  if (const char *b = std::getenv("ARTIS_AMD_SORT_MAXPC_T")) e->sort_maxpc_t = std::max(1, std::atoi(b));
  if (const char *b = std::getenv("ARTIS_AMD_WAVECHUNKS_R")) e->wave_chunks_r = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_WAVECHUNKS_T")) e->wave_chunks_t = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_CUCHUNKS_T")) e->cu_chunks_t = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_CONTLDS")) e->cont_lds = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_LINELDS")) e->line_lds = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_TILE_ZIGZAG")) e->tile_zigzag = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_TILE_ADAPT")) e->tile_adapt = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_TILE_BLOCK")) e->tile_block = std::max<int64_t>(0, std::atoll(b));
  if (const char *b = std::getenv("ARTIS_AMD_POOL_KEEP")) e->pool_keep = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_MATABLES_LDS")) e->ma_tables_lds = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_REFILL")) e->thermal_refill = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_SORT_NUMAJOR")) e->sort_numajor = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_SORT_CELLSHIFT")) e->sort_cellshift = e->sort_numajor ? std::max(0, std::min(20, std::atoi(b))) : 0;
  if (const char *b = std::getenv("ARTIS_AMD_TQ_LOW")) e->tq_low = std::max(1, std::min(64, std::atoi(b)));
  if (const char *b = std::getenv("ARTIS_AMD_SPARSE_FILL")) e->sparse_fill = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_SPARSE_MAX")) e->sparse_max_listed = std::max(0, std::atoi(b));
  {
    // the static part of every macro-atom record (tables.h: filter entries that are never counted) is written once
    const int64_t nrows_ = e->tile_cells;  // every resident row; the static parts do not depend on the cell
    Env env0 = make_env(e);
    env0.K = e->K;
    hipLaunchKernelGGL(k_mainit, dim3(nblocks(nrows_ * e->Mh.nlevels)), dim3(BLOCK), 0, nullptr, env0, nrows_);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
  }
  {
    // the population works through the cells in batches whose cooling terms fit a scratch of ~2 GB (ARTIS_AMD_POP_SCRATCH_MB)
    const double mb = pop_scratch_mb();
    const int64_t per = std::max<int64_t>(1, (int64_t)e->Mh.nupcum) * (int64_t)sizeof(double);
    e->pop_batch = std::max<int64_t>(1, std::min<int64_t>(e->tile_cells, (int64_t)(mb * 1048576.) / per));
    HIP_TRY(hipMalloc((void **)&e->d_collexc_terms, (size_t)(e->pop_batch * per) + 64));
  }
  if (const char *b = std::getenv("ARTIS_AMD_THERMAL_BLOCKS")) e->thermal_blocks_per_cu = std::max(1, std::min(ARTIS_THERMAL_WAVES, std::atoi(b)));
  e->trace = std::getenv("ARTIS_AMD_TRACE") != nullptr;
  if (e->trace)
    fprintf(stderr, "[artis_amd] record tiers: static records for %.2f of every ion's levels, %d cold levels, pool of %d slots per resident cell; %d tile(s) of %lld cells, %zu B per cell\n",
            e->ma_hotfrac, e->Mh.ncold, e->Mh.ma_pool_slots, e->ntiles, (long long)e->tile_cells, e->cache_bytes_per_cell);
  if (const char *b = std::getenv("ARTIS_AMD_VPKT_CONTLDS")) e->vpkt_cont_lds = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_TILE_PARK")) e->park_tails = std::atoi(b) != 0;
  if (const char *b = std::getenv("ARTIS_AMD_TILE_PARK_AT")) e->park_at = std::max<int64_t>(0, std::atoll(b));
  return ARTIS_OK;
}
}  // namespace

extern "C" {

void artis_amd_engine_destroy(artis_amd_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->comm && rccl_api().ok) (void)rccl_api().CommDestroy(e->comm);
  free_all(e->model_allocs);
  free_all(e->cell_allocs);
  free_all(e->cache_allocs);
  free_packet_buffers(e);
  if (e->h_counts) (void)hipHostFree(e->h_counts);
  void *ptrs[] = {e->d_est, e->d_stats, e->d_aos, e->d_hist, e->d_tiles, e->d_count, e->d_cursors, e->d_krow, e->d_fill_cells, e->d_waiting,
                  e->d_bfrate_kept, e->d_collexc_terms, e->d_visit_counts};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  for (hipEvent_t ev : {e->ev0, e->ev1, e->ev2, e->ev3})
    if (ev) (void)hipEventDestroy(ev);
  delete e;
}

int artis_amd_set_cellstate(artis_amd_engine *e, const artis_cellstate *cells, const artis_timestep *ts) {
  if (!e || !cells || !ts) {
    g_last_error = "null argument";
    return ARTIS_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  e->have_cells = false;  // until every array of the new state is resident
  free_all(e->cell_allocs);
  const DevModel &h = e->Mh;
  DevCells hc = make_host_cells_view(*cells);
  std::vector<float> zero_ffegrp;
  if (!hc.ffegrp) {  // no gamma packets will be handed over: the opacities of gammapkt.cc are never evaluated
    zero_ffegrp.assign((size_t)(h.npts_nonempty > 0 ? h.npts_nonempty : 1), 0.f);
    hc.ffegrp = zero_ffegrp.data();
  }
#define UPC(f, T, count)                                                                   \
  {                                                                                        \
    int rc = upload_array<T>(e->cell_allocs, hc.f, (int64_t)(count), (const T **)&e->C.f); \
    if (rc != ARTIS_OK) return rc;                                                         \
  }
  ARTIS_CELL_ARRAYS(UPC, h)
#undef UPC
#define UPO(f, T, count)                                                                     \
  {                                                                                          \
    e->C.f = nullptr;                                                                        \
    if (hc.f) {                                                                              \
      int rc = upload_array<T>(e->cell_allocs, hc.f, (int64_t)(count), (const T **)&e->C.f); \
      if (rc != ARTIS_OK) return rc;                                                         \
    }                                                                                        \
  }
  const int64_t nt_stored = (hc.nt_exc_count && hc.nt_excitations_stored > 0) ? hc.nt_excitations_stored : 0;
  ARTIS_CELL_OPTIONAL_ARRAYS(UPO, h)
#undef UPO
  e->C.nt_excitations_stored = (int32_t)nt_stored;
  e->C.nt_ionratecoeff = nullptr;
  e->C.nt_ionenrate_cum = nullptr;
  // what the options this library was built with need from the host
  if (!ARTIS_OPT_USE_LUT_PHOTOION && h.nphixstargets_total > 0 && !e->C.corrphotoioncoeff) {
    g_last_error = "this build has USE_LUT_PHOTOION off: artis_cellstate.corrphotoioncoeff is required";
    return ARTIS_ERR_ARG;
  }
  if (ARTIS_OPT_MULTIBIN_RADFIELD_MODEL_ON && (!e->C.radfieldbin_W || !e->C.radfieldbin_T_R)) {
    g_last_error = "this build has the multibin radiation field on: artis_cellstate.radfieldbin_W / _T_R are required";
    return ARTIS_ERR_ARG;
  }
  e->expopac_own = false;
  if (ARTIS_OPT_USE_CALCULATED_MEANATOMICWEIGHT && (ARTIS_OPT_USE_XCOM_GAMMAPHOTOION || ARTIS_OPT_NT_ON) && !e->C.elem_meanweight) {
    g_last_error = "this build has USE_CALCULATED_MEANATOMICWEIGHT and reads element number densities: artis_cellstate.elem_meanweight is required";
    return ARTIS_ERR_ARG;
  }
  if (ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON && !e->C.Jb_lu_normed) {
    g_last_error = "this build has DETAILED_LINE_ESTIMATORS_ON: artis_cellstate.Jb_lu_normed is required";
    return ARTIS_ERR_ARG;
  }
  if (ARTIS_EXPOPAC_TABLES) {
    // the tables of calculate_expansion_opacities(): the host's, or (both NULL) made by the engine at cell-cache population
    const bool need_planck = ARTIS_OPT_RPKT_BB_THERMALISATION;
    if (!e->C.expansionopacities && !e->C.expansionopacity_planck_cumulative) {
      const size_t cnt = (size_t)(h.npts_nonempty > 0 ? h.npts_nonempty : 1) * ARTIS_EXPOPAC_NBINS;
      float *dk = nullptr;
      double *dp = nullptr;
      HIP_TRY(hipMalloc((void **)&dk, sizeof(float) * cnt));
      e->cell_allocs.push_back(dk);
      HIP_TRY(hipMemset(dk, 0, sizeof(float) * cnt));
      HIP_TRY(hipMalloc((void **)&dp, sizeof(double) * cnt));
      e->cell_allocs.push_back(dp);
      HIP_TRY(hipMemset(dp, 0, sizeof(double) * cnt));
      e->C.expansionopacities = dk;
      e->C.expansionopacity_planck_cumulative = dp;
      e->expopac_own = true;
    } else if (!e->C.expansionopacities || (need_planck && !e->C.expansionopacity_planck_cumulative)) {
      g_last_error = "expansion-opacity build: hand over both artis_cellstate.expansionopacities and (with a thermalisation "
                     "probability) .expansionopacity_planck_cumulative, or neither (the engine then calculates them)";
      return ARTIS_ERR_ARG;
    }
  }
  e->S = make_step(*ts);
#if ARTIS_OPT_NT_ON
  if (!e->C.nt_frac_ionisation || !e->C.nt_frac_excitation || !e->C.nt_deposition_rate_density || !e->C.nt_eff_ionpot ||
      !e->C.nt_prob_num_auger || !e->C.nt_ionenfrac_num_auger || !e->C.nt_exc_count ||
      (nt_stored > 0 && (!e->C.nt_exc_frac_deposition || !e->C.nt_exc_ratecoeffperdeposition || !e->C.nt_exc_alltransindex))) {
    g_last_error = "this build has NT_ON: the Spencer-Fano solution (artis_cellstate.nt_*) is required";
    return ARTIS_ERR_ARG;
  }
  {
    const size_t bytes = sizeof(double) * (size_t)(h.npts_nonempty > 0 ? h.npts_nonempty : 1) * (size_t)h.nions;
    HIP_TRY(hipMalloc((void **)&e->C.nt_ionratecoeff, bytes));
    e->cell_allocs.push_back(e->C.nt_ionratecoeff);
    HIP_TRY(hipMalloc((void **)&e->C.nt_ionenrate_cum, bytes));
    e->cell_allocs.push_back(e->C.nt_ionenrate_cum);
    if (h.npts_nonempty > 0) {
      const Env env = make_env(e);
      hipLaunchKernelGGL(k_nt_cells, dim3((h.npts_nonempty + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, nullptr, env, h.npts_nonempty);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipDeviceSynchronize());
      int32_t err = 0;
      HIP_TRY(hipMemcpy(&err, e->d_err, sizeof(err), hipMemcpyDeviceToHost));
      if (err != 0) {
        (void)hipMemset(e->d_err, 0, sizeof(int32_t));
        g_last_error = "the Spencer-Fano solution is inconsistent (Auger probabilities do not sum to one): error flag " + std::to_string(err);
        return ARTIS_ERR_ARG;
      }
    }
  }
#endif
  e->have_cells = true;
  return artis_amd_populate_cellcache(e, nullptr);
}

}  // extern "C"

namespace {
// populate the cell cache: of every cell (nfill < 0: one tile, row = cell), or of the cells e->d_fill_cells[0..nfill) (a tiled cache: their rows are
// in e->d_krow, make_resident())
int populate_tile(artis_amd_engine *e, hipStream_t s, int64_t nfill = -1) {
  const DevModel &h = e->Mh;
  const int lo = 0, hi = h.npts_nonempty;
  e->tile_lo = lo;
  e->tile_hi = hi;
  e->tile_valid_lo = -1;
  Env env = make_env(e);
  if (nfill >= 0) {
    env.fill_cells = e->d_fill_cells;
    env.nfill = (int32_t)nfill;
  }
  const int64_t ncell_fill = nfill >= 0 ? nfill : hi - lo;
  if (ncell_fill <= 0) return ARTIS_OK;
  // The pool of on-demand records: a fill of the whole cache empties it. A fill of some cells of a tiled cache leaves it alone -- the cells that stay
  // resident keep their cold levels' records; the rows that are filled start without any (k_ma_reset below), and the units the cells before them held
  // stay handed out until the pool is used up and emptied (reset_pool_if_due() of the propagation loop), which costs fills, never an answer.
  // ARTIS_AMD_POOL_KEEP=0: emptied with every fill (round 5). (Emptying it only with the fills that replace half of the rows or more -- when most of what
  // it holds belongs to cells that leave -- was measured too: no better, 1520 against 1498 ms at a quarter of the cache; profiles/r06/tiling.md.)
  if (h.ncold > 0 && (nfill < 0 || !e->pool_keep)) {
    HIP_TRY(hipMemsetAsync(e->K.ma_pool_used, 0, sizeof(uint32_t), s));
    if (nfill >= 0) HIP_TRY(hipMemsetAsync(e->K.ma_rowtab, 0xFF, sizeof(int32_t) * (size_t)(e->tile_cells * (int64_t)h.ncold), s));
  }
  // in batches of pop_batch cells (the scratch of cooling terms holds that many rows); the kernels of a batch see it as
  // their whole fill: a sub-range of the tile, or a stretch of the list of a sparse fill
  const Env env_tile = env;
  for (int64_t b0 = 0; b0 < ncell_fill; b0 += e->pop_batch) {
  const int64_t ncell = std::min<int64_t>(e->pop_batch, ncell_fill - b0);
  env = env_tile;
  if (nfill >= 0) {
    env.fill_cells = e->d_fill_cells + b0;
    env.nfill = (int32_t)ncell;
  } else {
    env.tile_lo = lo + (int)b0;
    env.tile_hi = lo + (int)(b0 + ncell);
  }
  if (h.ncold > 0) hipLaunchKernelGGL(k_ma_reset, dim3(nblocks(ncell * h.ncold)), dim3(BLOCK), 0, s, env);
  hipLaunchKernelGGL(k_levelpops, dim3(nblocks(ncell * h.nlevels)), dim3(BLOCK), 0, s, env);
  if (h.ndpop > 0) hipLaunchKernelGGL(k_line_dpop, dim3(nblocks(ncell * h.nlines)), dim3(BLOCK), 0, s, env);
  hipLaunchKernelGGL(k_cell_scalars, dim3(nblocks(ncell)), dim3(BLOCK), 0, s, env);
  if (h.nbfcontinua > 0) hipLaunchKernelGGL(k_allcont, dim3(nblocks(ncell * h.nkeepwords * 64)), dim3(BLOCK), 0, s, env);
  if (h.nbfcontinua > 0) hipLaunchKernelGGL(k_keptlist, dim3(nblocks(ncell * 64)), dim3(BLOCK), 0, s, env);
  if (h.nphixstargets_total > 0)
    hipLaunchKernelGGL(k_corrphotoion, dim3(nblocks(ncell * h.nphixstargets_total)), dim3(BLOCK), 0, s, env, e->d_target_level);
  if (h.nalltrans > 0) hipLaunchKernelGGL(k_matrans, dim3(nblocks(ncell * h.nscanblk * 64)), dim3(BLOCK), 0, s, env);
  if (h.nrecomblevels > 0) hipLaunchKernelGGL(k_macroatom_recomb, dim3(nblocks(ncell * (int64_t)h.nrecomblevels * 16)), dim3(BLOCK), 0, s, env);
  hipLaunchKernelGGL(k_macroatom, dim3(nblocks(ncell * h.nlevels)), dim3(BLOCK), 0, s, env);
  if (h.nmalongsegs > 0) hipLaunchKernelGGL(k_mafilter_long, dim3(nblocks(ncell * (int64_t)h.nmalongsegs * 64)), dim3(BLOCK), 0, s, env);
#if ARTIS_EXPOPAC_TABLES
  if (e->expopac_own) {  // needs line_dpop and chi_ff_nnionpart of the tile's cells (k_line_dpop, k_cell_scalars above)
    hipLaunchKernelGGL(k_expopac, dim3(nblocks((int64_t)ncell * ARTIS_EXPOPAC_NBINS)), dim3(BLOCK), 0, s, env);
    if (ARTIS_OPT_RPKT_BB_THERMALISATION) hipLaunchKernelGGL(k_expopac_planck, dim3(nblocks(ncell)), dim3(BLOCK), 0, s, env);
  }
#endif
  hipLaunchKernelGGL(k_cooling_head, dim3(nblocks((int64_t)ncell * h.nions)), dim3(BLOCK), 0, s, env);
  hipLaunchKernelGGL(k_cooling_chain, dim3(nblocks((int64_t)ncell * h.nions * 16)), dim3(BLOCK), 0, s, env);
  if (h.ncoollines > 0) hipLaunchKernelGGL(k_collexc_filter, dim3(nblocks(ncell * (int64_t)h.ncoollines)), dim3(BLOCK), 0, s, env);
  hipLaunchKernelGGL(k_cooling_tail, dim3(nblocks((int64_t)ncell * h.nions * 16)), dim3(BLOCK), 0, s, env);
  hipLaunchKernelGGL(k_cooling_prefix, dim3(nblocks(ncell)), dim3(BLOCK), 0, s, env);
  if (h.nguide > 0) hipLaunchKernelGGL(k_cool_guide, dim3(nblocks(ncell * h.nguide)), dim3(BLOCK), 0, s, env);
  }  // batches
  const int64_t ncell = ncell_fill;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  // UPDATECELL (stats.h:47): one populate per cell
  unsigned long long add = (unsigned long long)ncell, cur = 0;
  HIP_TRY(hipMemcpy(&cur, e->d_stats + ARTIS_STAT_UPDATECELL, sizeof(cur), hipMemcpyDeviceToHost));
  cur += add;
  HIP_TRY(hipMemcpy(e->d_stats + ARTIS_STAT_UPDATECELL, &cur, sizeof(cur), hipMemcpyHostToDevice));
  int32_t err = 0;
  HIP_TRY(hipMemcpy(&err, e->d_err, sizeof(err), hipMemcpyDeviceToHost));
  if (err != 0) {
    (void)hipMemset(e->d_err, 0, sizeof(int32_t));  // the flag is reported once; the next call starts clean
    g_last_error = "cell cache population raised error flag " + std::to_string(err);
    return ARTIS_ERR_NOTCONVERGED;
  }
  e->tile_valid_lo = nfill >= 0 ? -1 : 0;
  return ARTIS_OK;
}

// a tiled cache: no cell is resident (a new cell state: what the rows hold belongs to the old one)
int forget_rows(artis_amd_engine *e, hipStream_t s) {
  if (e->ntiles <= 1) return ARTIS_OK;
  std::fill(e->h_krow.begin(), e->h_krow.end(), -1);
  std::fill(e->h_rowcell.begin(), e->h_rowcell.end(), -1);
  HIP_TRY(hipMemsetAsync(e->d_krow, 0xFF, sizeof(int32_t) * e->h_krow.size(), s));
  if (e->Mh.ncold > 0) {  // ... and no cold level has a record
    HIP_TRY(hipMemsetAsync(e->K.ma_pool_used, 0, sizeof(uint32_t), s));
    HIP_TRY(hipMemsetAsync(e->K.ma_rowtab, 0xFF, sizeof(int32_t) * (size_t)(e->tile_cells * (int64_t)e->Mh.ncold), s));
  }
  return ARTIS_OK;
}

// a tiled cache: make the cells want[0..) resident (at most tile_cells of them). A wanted cell that is resident keeps its row; the others take the
// free rows, then the rows of cells that are not wanted (which stay resident as long as nobody needs their rows); the cells that got a row are filled.
// *nfilled: how many that were.
int make_resident(artis_amd_engine *e, const std::vector<int32_t> &want, hipStream_t s, int64_t *nfilled) {
  *nfilled = 0;
  if (e->ntiles <= 1 || want.empty()) return ARTIS_OK;
  if ((int64_t)want.size() > e->tile_cells) {
    g_last_error = "make_resident: more cells than rows";
    return ARTIS_ERR_ARG;
  }
  const int32_t stamp = ++e->want_stamp;
  std::vector<int32_t> fresh;
  for (const int32_t c : want) {
    e->h_wanted[(size_t)c] = stamp;
    if (e->h_krow[(size_t)c] < 0) fresh.push_back(c);
  }
  if (fresh.empty()) return ARTIS_OK;
  size_t k = 0;
  for (int pass = 0; pass < 2 && k < fresh.size(); pass++)
    for (size_t r = 0; r < e->h_rowcell.size() && k < fresh.size(); r++) {
      const int32_t held = e->h_rowcell[r];
      if (pass == 0 ? held >= 0 : (held < 0 || e->h_wanted[(size_t)held] == stamp)) continue;
      if (held >= 0) e->h_krow[(size_t)held] = -1;
      e->h_rowcell[r] = fresh[k];
      e->h_krow[(size_t)fresh[k]] = (int32_t)r;
      k++;
    }
  if (k < fresh.size()) {
    g_last_error = "make_resident: no row left";
    return ARTIS_ERR_ARG;
  }
  HIP_TRY(hipMemcpyAsync(e->d_krow, e->h_krow.data(), sizeof(int32_t) * e->h_krow.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->d_fill_cells, fresh.data(), sizeof(int32_t) * fresh.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // (fresh is a local)
  const int rc = populate_tile(e, s, (int64_t)fresh.size());
  if (rc != ARTIS_OK) {  // (what the rows of the new cells hold is not to be used)
    for (const int32_t c : fresh) {
      e->h_rowcell[(size_t)e->h_krow[(size_t)c]] = -1;
      e->h_krow[(size_t)c] = -1;
    }
    (void)hipMemcpy(e->d_krow, e->h_krow.data(), sizeof(int32_t) * e->h_krow.size(), hipMemcpyHostToDevice);
    return rc;
  }
  *nfilled = (int64_t)fresh.size();
  return ARTIS_OK;
}

// Which cells should be resident for the next visit of a tiled run? waiting[c]: packets that wait in non-empty cell c (k_count_waiting).
//  - few packets wait (at most sparse_max of them, in less than half as many cells as there are rows): the cells in which they wait and the cells around
//    those, one step in every grid direction (an r-packet crosses a few cells per visit) -- a "sparse" visit
//  - tile_block == 0: the window of tile_cells consecutive cells in which most packets wait
//  - else: the blocks of tile_block consecutive cells in which most packets wait, wherever they lie, as many as there are rows for
// *holds: the packets that wait in the chosen cells.
void choose_cells(artis_amd_engine *e, const std::vector<int32_t> &waiting, std::vector<int32_t> &want, int64_t *holds, bool *sparse,
                  int64_t win_lo = -1) {
  const DevModel &h = e->Mh;
  const int64_t ncell_all = h.npts_nonempty, nrows = e->tile_cells;
  want.clear();
  *sparse = false;
  *holds = 0;
  int64_t lo = win_lo;
  if (win_lo < 0 && e->tile_block > 0 && e->tile_block < nrows) {
    const int64_t B = e->tile_block, nblk = (ncell_all + B - 1) / B;
    std::vector<std::pair<int64_t, int64_t>> score((size_t)nblk);
    for (int64_t b = 0; b < nblk; b++) {
      int64_t sum = 0;
      for (int64_t c = b * B; c < std::min(ncell_all, (b + 1) * B); c++) sum += waiting[(size_t)c];
      score[(size_t)b] = {-sum, b};
    }
    std::sort(score.begin(), score.end());
    for (const auto &sb : score) {
      const int64_t c0 = sb.second * B, c1 = std::min(ncell_all, c0 + B);
      if (sb.first == 0 || (int64_t)want.size() + (c1 - c0) > nrows) break;
      for (int64_t c = c0; c < c1; c++) want.push_back((int32_t)c);
      *holds -= sb.first;
    }
    std::sort(want.begin(), want.end());
  } else {
    if (win_lo < 0) {
      int64_t sum = 0, best = -1;
      lo = 0;
      for (int64_t c = 0; c < ncell_all; c++) {
        sum += waiting[(size_t)c];
        if (c >= nrows) sum -= waiting[(size_t)(c - nrows)];
        if ((c >= nrows - 1 || c == ncell_all - 1) && sum > best) {
          best = sum;
          lo = std::max<int64_t>(0, c - nrows + 1);
        }
      }
      lo = std::min<int64_t>(lo, std::max<int64_t>(0, ncell_all - nrows));
    }
    for (int64_t c = lo; c < std::min(ncell_all, lo + nrows); c++) {
      want.push_back((int32_t)c);
      *holds += waiting[(size_t)c];
    }
  }
  if (!e->sparse_fill || *holds > e->sparse_max_listed) return;
  // a sparse visit: of the chosen cells, those in which packets wait, and the cells around them (whichever cells those are)
  std::vector<int32_t> few;
  const int32_t stamp = -(++e->want_stamp);  // (marks of this choice: negative, never those of make_resident())
  std::vector<int32_t> &mark = e->h_wanted;
  const int ndim = (h.gridtype == ARTIS_GRID_SPHERICAL1D) ? 1 : ((h.gridtype == ARTIS_GRID_CYLINDRICAL2D) ? 2 : 3);
  for (const int32_t c : want) {
    if (waiting[(size_t)c] <= 0) continue;
    const int g = e->h_cell_grid[(size_t)c];
    int lo3[3] = {0, 0, 0}, hi3[3] = {0, 0, 0};
    for (int d = 0; d < ndim; d++) {
      const int idx = (g / h.coordstride[d]) % h.ncoordgrid[d];
      lo3[d] = idx > 0 ? -1 : 0;
      hi3[d] = idx < h.ncoordgrid[d] - 1 ? 1 : 0;
    }
    for (int dz = lo3[2]; dz <= hi3[2]; dz++)
      for (int dy = lo3[1]; dy <= hi3[1]; dy++)
        for (int dx = lo3[0]; dx <= hi3[0]; dx++) {
          const int cn = h.propcell_nonemptymgi[g + (dx * h.coordstride[0]) + (dy * h.coordstride[1]) + (dz * h.coordstride[2])];
          if (cn < 0 || mark[(size_t)cn] == stamp) continue;
          mark[(size_t)cn] = stamp;
          few.push_back(cn);
        }
  }
  if (few.empty() || (int64_t)few.size() * 2 >= nrows) return;
  want.swap(few);
  *sparse = true;
}
}  // namespace

extern "C" {

int artis_amd_populate_cellcache(artis_amd_engine *e, void *hip_stream) {
  if (!e || !e->have_cells) {
    g_last_error = "no cell state uploaded";
    return ARTIS_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  e->tile_valid_lo = -1;
  // with one tile the whole cache is filled now; with several, artis_amd_update_packets_device() makes cells resident as packets need them
  if (e->ntiles == 1) return populate_tile(e, (hipStream_t)hip_stream);
  return forget_rows(e, (hipStream_t)hip_stream);
}

int artis_amd_cache_tiles(artis_amd_engine *e, int32_t *ntiles, int64_t *cells_per_tile, int64_t *bytes_per_cell) {
  if (!e) return ARTIS_ERR_ARG;
  if (ntiles) *ntiles = e->ntiles;
  if (cells_per_tile) *cells_per_tile = e->tile_cells;
  if (bytes_per_cell) *bytes_per_cell = (int64_t)e->cache_bytes_per_cell;
  return ARTIS_OK;
}

int artis_amd_packets_upload(artis_amd_engine *e, const artis_packet *packets, int64_t npackets) {
  if (!e || (!packets && npackets > 0) || npackets < 0 || npackets > 2147483000LL) {
    g_last_error = "bad packet buffer";
    return ARTIS_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  int rc = ensure_packet_buffers(e, npackets);
  if (rc != ARTIS_OK) return rc;
  rc = ensure_aos(e, npackets);
  if (rc != ARTIS_OK) return rc;
  // a snapshot belongs to the population (and slot permutation) it was taken of
  if (e->d_pkt_snapshot) (void)hipFree(e->d_pkt_snapshot);
  e->d_pkt_snapshot = nullptr;
  if (npackets > 0) {
    HIP_TRY(hipMemcpy(e->d_aos, packets, sizeof(artis_packet) * (size_t)npackets, hipMemcpyHostToDevice));
    e->aos_valid = true;
    e->use_perm = false;
    if (e->slot_order_by_cell && e->sort_lists && npackets >= 2 * BLOCK) {
      // counting sort of the packet indices by propagation cell (the work-list sort kernels; the lists are free now)
      int32_t *ident = e->d_lists[NEXT_RPKT][0], *keys = e->d_lists[NEXT_RPKT][1];
      const int32_t n32 = (int32_t)npackets;
      const int32_t nkeys = e->Mh.ngrid;
      hipStream_t s = nullptr;
      hipLaunchKernelGGL(k_aos_cellkeys, dim3(nblocks(npackets)), dim3(BLOCK), 0, s, e->d_aos, npackets, nkeys, ident, keys);
      HIP_TRY(hipMemsetAsync(e->d_hist, 0, sizeof(int32_t) * (size_t)(nkeys + 1), s));
      if (nkeys <= SORT_LDS_KEYS)
        hipLaunchKernelGGL(k_sort_hist_lds, dim3(sort_lds_grid(n32)), dim3(BLOCK), 0, s, keys, n32, e->d_hist, nkeys);
      else
        hipLaunchKernelGGL(k_sort_hist, dim3(nblocks(n32)), dim3(BLOCK), 0, s, keys, n32, e->d_hist);
      const int ntiles = (nkeys + SCAN_TILE - 1) / SCAN_TILE;
      hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(1024), 0, s, e->d_hist, nkeys, e->d_tiles);
      hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, s, e->d_tiles, ntiles);
      hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(1024), 0, s, e->d_hist, nkeys, e->d_tiles);
      if (nkeys <= SORT_LDS_KEYS)
        hipLaunchKernelGGL(k_sort_scatter_lds, dim3(sort_lds_grid(n32)), dim3(BLOCK), 0, s, ident, keys, n32, e->d_hist, e->d_perm, nkeys);
      else
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblocks(n32)), dim3(BLOCK), 0, s, ident, keys, n32, e->d_hist, e->d_perm);
      e->use_perm = true;
    }
    hipLaunchKernelGGL(k_aos_to_rec, dim3(nblocks(npackets)), dim3(BLOCK), 0, nullptr, e->d_aos, e->P, e->use_perm ? e->d_perm : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
  }
  return ARTIS_OK;
}

int artis_amd_packets_download(artis_amd_engine *e, artis_packet *packets, int64_t npackets) {
  if (!e || npackets != e->npackets || (!packets && npackets > 0)) {
    g_last_error = "packet count does not match the resident population";
    return ARTIS_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  if (npackets == 0) return ARTIS_OK;
  int rc = ensure_aos(e, npackets);
  if (rc != ARTIS_OK) return rc;
  // the fields this path never touches keep the values the caller uploaded: the structs are still on the device from
  // artis_amd_packets_upload() (no second trip over PCIe: 256 B per packet); copied again only if that buffer was replaced
  if (!e->aos_valid) HIP_TRY(hipMemcpy(e->d_aos, packets, sizeof(artis_packet) * (size_t)npackets, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_rec_to_aos, dim3(nblocks(npackets)), dim3(BLOCK), 0, nullptr, e->P, e->d_aos, e->use_perm ? e->d_perm : nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(packets, e->d_aos, sizeof(artis_packet) * (size_t)npackets, hipMemcpyDeviceToHost));
  return ARTIS_OK;
}

int artis_amd_packets_snapshot(artis_amd_engine *e) {
  if (!e || !e->d_pkt) {
    g_last_error = "no resident packets";
    return ARTIS_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  if (!e->d_pkt_snapshot) HIP_TRY(hipMalloc(&e->d_pkt_snapshot, e->pkt_bytes));
  HIP_TRY(hipMemcpy(e->d_pkt_snapshot, e->d_pkt, e->pkt_bytes, hipMemcpyDeviceToDevice));
  return ARTIS_OK;
}

int artis_amd_packets_restore(artis_amd_engine *e) {
  if (!e || !e->d_pkt || !e->d_pkt_snapshot) {
    g_last_error = "no snapshot";
    return ARTIS_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpy(e->d_pkt, e->d_pkt_snapshot, e->pkt_bytes, hipMemcpyDeviceToDevice));
  return ARTIS_OK;
}

namespace {
// counting sort of list[0..n) by its entries' keys into e->d_sorted; *out = the list to launch on
// max_per_cell: a list with more entries per cell than this stays in the order it was appended in. A cell-sorted list puts
// every lane that is running on an XCD into the same few cells when the cells are few and full, and their estimator
// atomics then hit the same few addresses at the same time (device-wide atomics on one address are serialised in memory:
// 20^3 cells, 1e7 packets: k_thermal 907 ms sorted, 725 ms unsorted), while the locality the sort buys matters less
// because fewer cells' tables compete for the caches. Models with so few cells that the kernels accumulate their per-cell
// estimators in LDS (Env::cellest_lds) have no such atomics and are always sorted (6^3 cells: 494 ms sorted, 593 unsorted).
int sort_by_key(artis_amd_engine *e, hipStream_t s, const int32_t *list, const int32_t *keys, int32_t n, const int32_t **out, int nbins,
                int64_t ncells, int max_per_cell, int32_t nkeys_given = 0) {
  *out = list;
  if (!e->sort_lists || n < 2 * BLOCK) return ARTIS_OK;
  if ((int64_t)n > (int64_t)max_per_cell * (ncells > 0 ? ncells : 1)) return ARTIS_OK;
  const int32_t nkeys = nkeys_given > 0 ? nkeys_given : e->Mh.ngrid * nbins;
  HIP_TRY(hipMemsetAsync(e->d_hist, 0, sizeof(int32_t) * (size_t)(nkeys + 1), s));
  if (nkeys <= SORT_LDS_KEYS)
    hipLaunchKernelGGL(k_sort_hist_lds, dim3(sort_lds_grid(n)), dim3(BLOCK), 0, s, keys, n, e->d_hist, nkeys);
  else
    hipLaunchKernelGGL(k_sort_hist, dim3(nblocks(n)), dim3(BLOCK), 0, s, keys, n, e->d_hist);
  const int ntiles = (nkeys + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(1024), 0, s, e->d_hist, nkeys, e->d_tiles);
  hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, s, e->d_tiles, ntiles);
  hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(1024), 0, s, e->d_hist, nkeys, e->d_tiles);
  if (nkeys <= SORT_LDS_KEYS)
    hipLaunchKernelGGL(k_sort_scatter_lds, dim3(sort_lds_grid(n)), dim3(BLOCK), 0, s, list, keys, n, e->d_hist, e->d_sorted, nkeys);
  else
    hipLaunchKernelGGL(k_sort_scatter, dim3(nblocks(n)), dim3(BLOCK), 0, s, list, keys, n, e->d_hist, e->d_sorted);
  *out = e->d_sorted;
  return ARTIS_OK;
}
}  // namespace

#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
// the estimator updates the propagation launch before it recorded (the cells' cache rows are still resident)
static int launch_bfest_dense(artis_amd_engine *e, const Env &env, hipStream_t s) {
  const bool lds = e->dense_cont_lds && e->Mh.nbfcontinua <= CONT_LDS_MAX;
#define DENSE_LAUNCH(LPR)                                                                                        \
  if (lds)                                                                                                       \
    hipLaunchKernelGGL((k_bfest_dense<true, DENSE_TB, LPR>), dim3(e->ncu * ARTIS_DENSE_WGS), dim3(DENSE_TB), 0, s, env); \
  else                                                                                                           \
    hipLaunchKernelGGL((k_bfest_dense<false, BLOCK, LPR>), dim3(e->ncu * 8), dim3(BLOCK), 0, s, env);
  if (e->dense_lpr == 64) {
    DENSE_LAUNCH(64)
  } else if (e->dense_lpr == 16) {
    DENSE_LAUNCH(16)
  } else {
    DENSE_LAUNCH(32)
  }
#undef DENSE_LAUNCH
  return ARTIS_OK;
}
#endif

int artis_amd_update_packets_device(artis_amd_engine *e, void *hip_stream) {
  if (!e || !e->have_cells || !e->d_pkt) {
    g_last_error = "engine needs artis_amd_set_cellstate() and resident packets first";
    return ARTIS_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = (hipStream_t)hip_stream;
  e->last_propagate_ms = 0.;
  e->last_nlaunches = 0;
  e->last_sweeps = e->last_tile_fills = e->last_listed = 0;
  e->last_sparse_fills = e->last_cells_filled = 0;
  e->last_parked = 0;
  e->last_pool_resets = 0;
  e->thermal_variants = 0;
  e->last_fill_ms = 0.;
  for (int k = 0; k < NEXT_NKINDS; k++) {
    e->kms[k] = 0.;
    e->kms_tail = 0.;
    e->klaunches[k] = 0;
    e->kthreads[k] = 0;
  }
  const int64_t n = e->npackets;
  if (n == 0) return ARTIS_OK;
#ifdef ARTIS_VISIT_COUNTS
  {
    const size_t vb = sizeof(uint32_t) * (size_t)e->Mh.npts_nonempty * (size_t)e->Mh.nlevels;
    if (e->d_visit_counts == nullptr) HIP_TRY(hipMalloc((void **)&e->d_visit_counts, vb));
    HIP_TRY(hipMemsetAsync(e->d_visit_counts, 0, vb, s));
  }
#endif
  Env env = make_env(e);
  if (e->d_bfrate_kept != nullptr) {
    if (e->bfrate_kept_dirty)
      HIP_TRY(hipMemsetAsync(e->d_bfrate_kept, 0, sizeof(double) * (size_t)e->Mh.npts_nonempty * (size_t)e->Mh.nbfcontinua, s));
    e->bfrate_kept_dirty = true;
  }
  int cur[NEXT_NKINDS] = {};             // which of the two buffers is the current list of each kind
  const int r_nubins = e->sort_nu ? SORT_NUBINS : 1;  // frequency bins in the keys of the r-packet list
  // cell groups of the frequency-major keys: the ONE number both the keys (Lists::numajor) and the sort's key count are made of
  const int32_t r_ngroups = (e->sort_cellshift > 0) ? ((e->Mh.ngrid >> e->sort_cellshift) + 1) : e->Mh.ngrid;
  int32_t cnt[2 * NEXT_NKINDS];                        // host copy of the device counters
  bool pool_reset_due = false;
  auto lists_for = [&](int self_kind) {
    Lists L;
    for (int k = 0; k < NEXT_NKINDS; k++) {
      L.lst[k] = e->d_lists[k][cur[k]];
      L.key[k] = e->d_keys[k][cur[k]];
    }
    L.counts = e->d_count;
    L.self_kind = self_kind;
    L.self_list = self_kind > 0 ? e->d_lists[self_kind][1 - cur[self_kind]] : nullptr;
    L.self_key = self_kind > 0 ? e->d_keys[self_kind][1 - cur[self_kind]] : nullptr;
    L.self_count = e->d_count + NEXT_NKINDS;  // one alternate counter: only one kernel runs at a time
    L.kpkt_slot = NEXT_MA;  // k-packets travel in the thermal list
    L.nubins = r_nubins;
    L.numajor = e->sort_numajor ? r_ngroups : 0;
    L.cellshift = e->sort_cellshift;
    L.mabins = e->ma_bins;
    return L;
  };
  int32_t errflag = 0;
  // (ARTIS_AMD_TRACE: where the host's time of the call goes -- waiting for the stream, submitting sorts, submitting launches)
  using clk = std::chrono::steady_clock;
  double wall_sync = 0., wall_sort = 0., wall_launch = 0.;
  const clk::time_point wall_t0 = clk::now();
  auto since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
  auto read_counts = [&]() -> int {
    // (one copy into pinned memory: counters and error flag are neighbours. Two copies into the stack -- pageable, staged by the runtime -- were a
    // measurable share of the ~90 ms a headline step spends outside its kernels)
    const clk::time_point t_sync = clk::now();
    HIP_TRY(hipMemcpyAsync(e->h_counts, e->d_count, sizeof(int32_t) * (2 * NEXT_NKINDS + 1), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    wall_sync += since(t_sync);
    std::memcpy(cnt, e->h_counts, sizeof(int32_t) * 2 * NEXT_NKINDS);
    errflag = e->h_counts[2 * NEXT_NKINDS];
    HIP_TRY(hipGetLastError());
    if (errflag != 0) {
      g_last_error = "a kernel raised error flag " + std::to_string(errflag) + " (an assert_always of the reference would have fired)";
      if (errflag == 46) g_last_error = "the pool of on-demand macro-atom records cannot hold a single record of this atomic data (error flag 46): raise ARTIS_AMD_MA_POOLFRAC (or ARTIS_AMD_MA_HOTFRAC)";
      (void)hipMemsetAsync(e->d_err, 0, sizeof(int32_t), s);
      return ARTIS_ERR_NOTCONVERGED;
    }
    if (cnt[2 * NEXT_NKINDS - 1] != 0) {  // a lane found the pool of on-demand records used up (Env::ma_pool_full)
      pool_reset_due = true;
      HIP_TRY(hipMemsetAsync(e->d_count + (2 * NEXT_NKINDS - 1), 0, sizeof(int32_t), s));
    }
    return ARTIS_OK;
  };
  // The pool of on-demand records used up: the packets that wait for a record sit on the slow-path list (PEND_MA_FILL). Before that list's next
  // launch -- after the thermal kernel has walked on with the records the last one filled -- the pool is emptied: every cold level of the resident
  // cells is without a record again and is filled when next needed, exactly as after a tile's refill. Costs fills, never an answer.
  auto reset_pool_if_due = [&](const Env &env_now) -> int {
    if (!pool_reset_due || e->Mh.ncold <= 0) return ARTIS_OK;
    pool_reset_due = false;
    (void)env_now;
    HIP_TRY(hipMemsetAsync(e->K.ma_rowtab, 0xFF, sizeof(int32_t) * (size_t)(e->tile_cells * (int64_t)e->Mh.ncold), s));  // (every row: k_ma_reset)
    HIP_TRY(hipMemsetAsync(e->K.ma_pool_used, 0, sizeof(uint32_t), s));
    e->last_pool_resets++;
    if (e->trace) fprintf(stderr, "[artis_amd] the pool of on-demand records was used up: emptied (%lld)\n", (long long)e->last_pool_resets);
    return ARTIS_OK;
  };
  // the launches of the two propagation kernels (on a given stream, with a given set of chunk cursors): one after the other on the call's stream,
  // or side by side on two streams where both lists are short ("duet", below)
  auto launch_rpkt = [&](hipStream_t st, const int32_t *lst, int32_t nk, const Lists &next, int32_t *cursors) -> int {
        const int grid = (int)std::min<int64_t>(((int64_t)nk + ARTIS_RPKT_TB - 1) / ARTIS_RPKT_TB, (int64_t)e->ncu * ARTIS_RPKT_WGS);  // persistent: every block resident
        const int bud_r = (e->budget_r_small > 0 && nk < e->small_list) ? std::min(e->budget_r_small, e->budget_r) : e->budget_r;
        const int nch = e->wave_chunks_r ? chunks_for(nk, grid * (ARTIS_RPKT_TB / 64)) : 8;
        if (e->line_lds && e->Mh.nlines <= LINE_LDS_MAX && e->Mh.nlines > 0 && !(env.cellest_n_r > RPKT_CELLEST_CAP))
          hipLaunchKernelGGL((k_rpkt<false, ARTIS_RPKT_TB, true>), dim3(grid), dim3(ARTIS_RPKT_TB), 0, st, env, lst, nk, next, e->d_stats, bud_r, cursors, nch,
                             (e->drain_r > 0 && nk >= e->drain_min_list) ? e->drain_r : bud_r);
        else if (e->cont_lds && e->Mh.nbfcontinua <= CONT_LDS_MAX && e->Mh.nbfcontinua > 0 &&
            !(env.cellest_n_r > RPKT_CELLEST_CAP))
          hipLaunchKernelGGL((k_rpkt<true, ARTIS_RPKT_TB>), dim3(grid), dim3(ARTIS_RPKT_TB), 0, st, env, lst, nk, next, e->d_stats, bud_r, cursors, nch,
                             (e->drain_r > 0 && nk >= e->drain_min_list) ? e->drain_r : bud_r);
        else
          hipLaunchKernelGGL((k_rpkt<false, ARTIS_RPKT_TB>), dim3(grid), dim3(ARTIS_RPKT_TB), 0, st, env, lst, nk, next, e->d_stats, bud_r, cursors, nch,
                             (e->drain_r > 0 && nk >= e->drain_min_list) ? e->drain_r : bud_r);
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
        if (env.bfev != nullptr) {  // the estimator updates the launch recorded (the cells' cache rows are still resident)
          {
            const int rcd = launch_bfest_dense(e, env, st);
            if (rcd != ARTIS_OK) return rcd;
          }
          HIP_TRY(hipMemsetAsync(e->d_bfev_count, 0, sizeof(int32_t), st));
        }
#endif
    return ARTIS_OK;
  };
  auto launch_thermal = [&](hipStream_t st, const int32_t *lst, int32_t nk, const Lists &next, int32_t *cursors) -> int {
        // persistent: every workgroup resident (ARTIS_THERMAL_WAVES waves per SIMD)
        {
          const int grid = (int)std::min<int64_t>(((int64_t)nk + ARTIS_THERMAL_TB - 1) / ARTIS_THERMAL_TB,
                                                  (int64_t)e->ncu * std::min(e->thermal_blocks_per_cu, ARTIS_THERMAL_WGS));
          const bool per_cu = e->cu_chunks_t && nk >= 256 * 1024;
          // (drain: only where the next thermal launch will be large too, so that what is handed on runs beside a full list)
          const int bud_t = (e->budget_t_small > 0 && nk < e->small_list) ? std::min(e->budget_t_small, e->budget_t) : e->budget_t;
          const int drain = (e->drain_t > 0 && nk >= e->drain_min_list) ? e->drain_t : bud_t;
          const size_t tq_bytes = tq_lds_bytes(TQ_TB, e->Mh.nlevels, e->Mh.nalltrans);
          const bool cold = e->Mh.ncold > 0;  // (kernels built with the on-demand records' look-ups only where the model has cold levels)
          // the COLD = false / true instantiation of a thermal kernel (K<A, B, COLD>)
#define LAUNCH_T2(K, A, B, GRID, TBS, LDSB, ...)                                                                   \
  do {                                                                                                             \
    if (cold)                                                                                                      \
      hipLaunchKernelGGL((K<A, B, true>), dim3(GRID), dim3(TBS), LDSB, st, __VA_ARGS__);                            \
    else                                                                                                           \
      hipLaunchKernelGGL((K<A, B, false>), dim3(GRID), dim3(TBS), LDSB, st, __VA_ARGS__);                           \
  } while (0)
          if (cold) e->thermal_variants |= ARTIS_AMD_THERMAL_COLD;
          if (e->thermal_refill && ARTIS_THERMAL_SPLIT_EXACT && env.cellest_n_t == 0 && tq_bytes <= 160 * 1024 - 1024 && nk >= 4096 && e->Mh.nlevels < 32768) {
            e->thermal_variants |= ARTIS_AMD_THERMAL_REFILL;
            if (!e->tq_attr_set) {  // (per engine, i.e. per device: the attribute is the device'st, not the process's -- ADVICE r05)
              HIP_TRY(hipFuncSetAttribute((const void *)k_thermal_q<TQ_TB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
              HIP_TRY(hipFuncSetAttribute((const void *)k_thermal_q<TQ_TB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
              e->tq_attr_set = true;
            }
            const int grid1 = (int)std::min<int64_t>(((int64_t)nk + TQ_TB - 1) / TQ_TB, (int64_t)e->ncu);
            if (cold)
              hipLaunchKernelGGL((k_thermal_q<TQ_TB, true>), dim3(grid1), dim3(TQ_TB), tq_bytes, st, env, lst, nk, next, e->d_stats, bud_t, cursors,
                                 e->wave_chunks_t ? chunks_for(nk, grid1 * (TQ_TB / 64)) : 8, drain, e->tq_low);
            else
              hipLaunchKernelGGL((k_thermal_q<TQ_TB, false>), dim3(grid1), dim3(TQ_TB), tq_bytes, st, env, lst, nk, next, e->d_stats, bud_t, cursors,
                                 e->wave_chunks_t ? chunks_for(nk, grid1 * (TQ_TB / 64)) : 8, drain, e->tq_low);
          } else if (e->ma_tables_lds && e->Mh.nlevels <= MA_LDS_LEVELS && e->Mh.nalltrans <= MA_LDS_TRANS && nk >= 4096) {
            const int grid1 = (int)std::min<int64_t>(((int64_t)nk + 1023) / 1024, (int64_t)e->ncu);
            e->thermal_variants |= ARTIS_AMD_THERMAL_LDS_TABLES;
            LAUNCH_T2(k_thermal, 1024, 1, grid1, 1024, 0, env, lst, nk, next, e->d_stats, bud_t, cursors, e->wave_chunks_t ? chunks_for(nk, grid1 * 16) : 8, 0, drain);
          } else if (e->ma_tables_lds && e->Mh.nlevels <= MA_LDS_LEVELS2 && nk >= 4096 && env.cellest_n_t == 0) {
            const int grid1 = (int)std::min<int64_t>(((int64_t)nk + 1023) / 1024, (int64_t)e->ncu);
            e->thermal_variants |= ARTIS_AMD_THERMAL_LDS_LEVELPACK;
            LAUNCH_T2(k_thermal, 1024, 2, grid1, 1024, 0, env, lst, nk, next, e->d_stats, bud_t, cursors, e->wave_chunks_t ? chunks_for(nk, grid1 * 16) : 8, 0, drain);
          } else {
            e->thermal_variants |= ARTIS_AMD_THERMAL_PLAIN;
            LAUNCH_T2(k_thermal, ARTIS_THERMAL_TB, 0, grid, ARTIS_THERMAL_TB, 0, env, lst, nk, next, e->d_stats, bud_t, cursors,
                      per_cu ? 256 : (e->wave_chunks_t ? chunks_for(nk, grid * (ARTIS_THERMAL_TB / 64)) : 8), per_cu ? 2 : 0, drain);
          }
#undef LAUNCH_T2
        }
    return ARTIS_OK;
  };
  int rc = ARTIS_OK;
  const int order[6] = {NEXT_SLOW, NEXT_GAMMA, NEXT_BB, NEXT_KPKT, NEXT_MA, NEXT_RPKT};
  int64_t guard = 0;
  const int64_t ncell_all = e->Mh.npts_nonempty;
  bool first_pass = true;
  // Sweeps over the cell-cache tiles (one tile, one sweep when the whole cache is resident): list the packets that sit
  // in the tile, fill the tile's cache if any do, advance them until they leave the tile or are done; repeat until a
  // sweep finds no packet left to advance.
  const bool adaptive = e->tile_adapt && e->ntiles > 1;
  bool all_done = false;
  std::vector<int32_t> want;
  if (e->ntiles > 1 && e->d_waiting == nullptr) {
    HIP_TRY(hipMalloc((void **)&e->d_waiting, sizeof(int32_t) * (size_t)(ncell_all + 1)));
    e->h_waiting.assign((size_t)ncell_all + 1, 0);
  }
  e->last_visits = 0;
  for (int sweep = 0;; sweep++) {
  bool any_active = false;
  for (int tstep = 0; tstep < e->ntiles; tstep++) {
  // sweeps alternate their direction: a packet that left its tile against the direction of one sweep is met by the next
  // one on its way back (with one direction it waits a whole sweep per backward crossing)
  const int tile = (e->tile_zigzag && (sweep & 1)) ? e->ntiles - 1 - tstep : tstep;
  if (e->ntiles > 1) {
    // where do the packets wait? The cells in which most of them do are made resident (choose_cells(): a window or a set of blocks of cells, or -- few
    // packets -- the very cells); cells that are resident already keep their rows, the others are filled
    env = make_env(e);
    HIP_TRY(hipMemsetAsync(e->d_waiting, 0, sizeof(int32_t) * (size_t)(ncell_all + 1), s));
    hipLaunchKernelGGL(k_count_waiting, dim3(nblocks(n)), dim3(BLOCK), 0, s, env, e->d_waiting, e->d_waiting + ncell_all);
    HIP_TRY(hipMemcpyAsync(e->h_waiting.data(), e->d_waiting, sizeof(int32_t) * (size_t)(ncell_all + 1), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    int64_t total = 0;
    for (int64_t c = 0; c < ncell_all; c++) total += e->h_waiting[(size_t)c];
    if (total == 0 && e->h_waiting[(size_t)ncell_all] == 0) {  // nothing left anywhere
      all_done = true;
      break;
    }
    int64_t holds = 0, nfilled = 0;
    bool sparse = false;
    choose_cells(e, e->h_waiting, want, &holds, &sparse, adaptive ? -1 : (int64_t)tile * e->tile_cells);
    if (holds == 0) want.clear();  // (no packet waits for a row of these cells: the visit is for the packets that need none, if any)
    if (e->trace)
      fprintf(stderr, "[artis_amd] visit %lld: %lld packets wait in cells, %d need no row; %zu cells chosen%s hold %lld\n", (long long)e->last_visits,
              (long long)total, e->h_waiting[(size_t)ncell_all], want.size(), sparse ? " (sparse)" : "", (long long)holds);
    HIP_TRY(hipEventRecord(e->ev2, s));
    rc = make_resident(e, want, s, &nfilled);
    if (rc != ARTIS_OK) return rc;
    if (nfilled > 0) {
      HIP_TRY(hipEventRecord(e->ev3, s));
      HIP_TRY(hipEventSynchronize(e->ev3));
      float fms = 0.f;
      HIP_TRY(hipEventElapsedTime(&fms, e->ev2, e->ev3));
      e->last_fill_ms += fms;
      e->last_tile_fills++;
      e->last_cells_filled += nfilled;
      if (sparse) e->last_sparse_fills++;
    }
  } else if (e->tile_valid_lo != 0) {
    rc = populate_tile(e, s);
    if (rc != ARTIS_OK) return rc;
  }
  env = make_env(e);
  for (int k = 0; k < NEXT_NKINDS; k++) cur[k] = 0;
  HIP_TRY(hipMemsetAsync(e->d_count, 0, sizeof(int32_t) * 2 * NEXT_NKINDS, s));
  hipLaunchKernelGGL(k_classify, dim3(nblocks(n)), dim3(BLOCK), 0, s, env, lists_for(0), first_pass ? 1 : 0);
  first_pass = false;
  rc = read_counts();
  if (rc != ARTIS_OK) return rc;
  bool tile_active = false;
  for (int k = 1; k < NEXT_NKINDS; k++) tile_active = tile_active || cnt[k] > 0;
  if (!tile_active) continue;
  any_active = true;
  e->last_visits++;
  for (int k = 1; k < NEXT_NKINDS; k++) e->last_listed += cnt[k];
  if (e->trace) fprintf(stderr, "[artis_amd] sweep %d tile %d of %d\n", sweep, tile, e->ntiles);

  // one launch = the whole current list of one kind. Order: slow path, k-packets, macro-atoms, r-packets, so that a
  // k-packet -> macro-atom -> k-packet cycle costs two launches.
  // the tail kernel takes the end of a population that began larger (a population that begins below the threshold runs on
  // the split kernels throughout, unless ARTIS_AMD_TAIL_ALWAYS=1)
  int64_t listed = 0;
  for (int k = 1; k < NEXT_NKINDS; k++) listed += cnt[k];
  // (tiled runs: the later sweeps bring a tile a few stragglers at a time; each such visit is a tail from its first launch)
  const bool tail_ok = e->tail_max > 0 && (e->tail_always || listed > e->tail_max || sweep > 0 || (adaptive && e->last_visits > e->ntiles));
  int64_t visit_launches = 0;  // split-kernel launches of this visit (a visit parks its tail only after it has advanced its packets)
  while (cnt[NEXT_RPKT] > 0 || cnt[NEXT_MA] > 0 || cnt[NEXT_SLOW] > 0 || cnt[NEXT_KPKT] > 0 || cnt[NEXT_GAMMA] > 0 || cnt[NEXT_BB] > 0) {
    const int tail_kinds[4] = {NEXT_RPKT, NEXT_MA, NEXT_SLOW, NEXT_BB};
    int64_t tail_n = 0;
    for (int k : tail_kinds) tail_n += cnt[k];
    // (round 6) tiled run: a visit that began larger parks what is left of it once that has fallen to park_at packets -- BEFORE the long run
    // of small, latency-bound launches that its last packets would otherwise cost every visit: they wait in their cells and are listed
    // again, merged with the other windows' stragglers, by a later visit (a visit that BEGINS with that few runs them to their end)
    if (e->park_tails && e->ntiles > 1 && e->park_at > e->tail_max && listed > e->park_at && visit_launches > 0 && tail_n > 0 &&
        tail_n + cnt[NEXT_KPKT] <= e->park_at) {
      e->last_parked += tail_n + cnt[NEXT_KPKT] + cnt[NEXT_GAMMA];
      if (e->trace) fprintf(stderr, "[artis_amd] sweep %d tile %d: %lld packets parked (park_at)\n", sweep, tile, (long long)(tail_n + cnt[NEXT_KPKT]));
      break;
    }
    if (tail_ok && tail_n > 0 && tail_n <= e->tail_max && cnt[NEXT_KPKT] == 0) {
      if (e->park_tails && e->ntiles > 1 && listed > e->tail_max && visit_launches > 0) {
        // tiled run, a visit that began larger: its last packets wait in the tile (their state is in the packet store; the next
        // classify pass lists them again) and run with the packets that return to it in the next sweep, instead of one long
        // k_tail launch per visit. A visit that BEGINS with a tail's worth of packets runs them to their end (below): no
        // packet waits more than once without the tile's population having shrunk to that.
        e->last_parked += tail_n + cnt[NEXT_GAMMA];
        if (e->trace) fprintf(stderr, "[artis_amd] sweep %d tile %d: %lld packets parked\n", sweep, tile, (long long)tail_n);
        break;
      }
      // the last packets of these kinds: one launch carries each through all its remaining alternations (k_tail)
      const int32_t nr = (int32_t)tail_n, nt = 0;
      // the four current lists are consumed whole, and no packet comes back to them -- but for one case: a packet that waits for a record of a
      // pool that is used up leaves for the slow-path list (k_tail "waits"). That entry must not land in a buffer other waves still read their
      // packets from: the slow-path kind is the launch's own kind, its entries go to the ALTERNATE slow-path list, which becomes the current one.
      const Lists next = lists_for(NEXT_SLOW);
      TailLists in;
      for (int i = 0; i < 4; i++) {
        in.list[i] = e->d_lists[tail_kinds[i]][cur[tail_kinds[i]]];
        in.n[i] = cnt[tail_kinds[i]];
        HIP_TRY(hipMemsetAsync(e->d_count + tail_kinds[i], 0, sizeof(int32_t), s));
      }
      HIP_TRY(hipMemsetAsync(e->d_count + NEXT_NKINDS, 0, sizeof(int32_t), s));
      rc = reset_pool_if_due(env);
      if (rc != ARTIS_OK) return rc;
      HIP_TRY(hipEventRecord(e->ev0, s));
      e->thermal_variants |= ARTIS_AMD_THERMAL_TAIL;
      hipLaunchKernelGGL(k_tail, dim3(nblocks(tail_n * 64)), dim3(BLOCK), 0, s, env, in, next, e->d_stats);
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
      if (env.bfev != nullptr) {
        rc = launch_bfest_dense(e, env, s);
        if (rc != ARTIS_OK) return rc;
        HIP_TRY(hipMemsetAsync(e->d_bfev_count, 0, sizeof(int32_t), s));
      }
#endif
      HIP_TRY(hipEventRecord(e->ev1, s));
      rc = read_counts();
      if (rc != ARTIS_OK) return rc;
      // (the alternate slow-path list becomes the current one, as after a launch of the slow-path kernel)
      cur[NEXT_SLOW] = 1 - cur[NEXT_SLOW];
      cnt[NEXT_SLOW] = cnt[NEXT_NKINDS];
      HIP_TRY(hipMemcpyAsync(e->d_count + NEXT_SLOW, e->d_count + NEXT_NKINDS, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
      e->kms_tail += ms;
      e->last_nlaunches++;
      if (e->trace)
        fprintf(stderr, "[artis_amd] launch %lld tail n=%d+%d %.3f ms -> r %d ma %d slow %d gamma %d bb %d\n", (long long)e->last_nlaunches, nr, nt,
                ms, cnt[NEXT_RPKT], cnt[NEXT_MA], cnt[NEXT_SLOW], cnt[NEXT_GAMMA], cnt[NEXT_BB]);
      if (++guard > 2000000LL) {
        g_last_error = "packet loop did not terminate";
        return ARTIS_ERR_NOTCONVERGED;
      }
      continue;
    }
    for (int kind : order) {
      const int32_t nk = cnt[kind];
      if (nk <= 0) continue;
      const Lists next = lists_for(kind);
      const int32_t *lst = e->d_lists[kind][cur[kind]];
      const clk::time_point t_sort = clk::now();
      if (kind == NEXT_RPKT || kind == NEXT_GAMMA || (kind == NEXT_MA && e->sort_ma)) {
        rc = sort_by_key(e, s, e->d_lists[kind][cur[kind]], e->d_keys[kind][cur[kind]], nk, &lst, kind == NEXT_RPKT ? r_nubins : (kind == NEXT_MA ? e->ma_bins : 1),
                         e->tile_cells, kind == NEXT_MA ? (env.cellest_n_t > 0 ? INT32_MAX : e->sort_maxpc_t)
                                                  : (env.cellest_n_r > 0 ? INT32_MAX : e->sort_maxpc_r),
                         (kind == NEXT_RPKT && r_nubins > 1 && e->sort_numajor) ? r_ngroups * r_nubins : 0);
        if (rc != ARTIS_OK) return rc;
      }
      wall_sort += since(t_sort);
      const clk::time_point t_launch = clk::now();
      if (kind == NEXT_SLOW) {
        rc = reset_pool_if_due(env);
        if (rc != ARTIS_OK) return rc;
      }
      // the kernel starts with an empty current list of its own kind: everything it keeps goes to the alternate list
      hipLaunchKernelGGL(k_launch_reset, dim3(1), dim3(BLOCK), 0, s, e->d_count, kind, e->d_cursors);  // (one command instead of three memsets)
      HIP_TRY(hipEventRecord(e->ev0, s));
      if (kind == NEXT_RPKT) {
        rc = launch_rpkt(s, lst, nk, next, e->d_cursors);
        if (rc != ARTIS_OK) return rc;
      } else if (kind == NEXT_GAMMA) {
        const int grid = std::min(nblocks(nk), e->ncu * ARTIS_GAMMA_WAVES);
        hipLaunchKernelGGL(k_gamma, dim3(grid), dim3(BLOCK), 0, s, env, lst, nk, next, e->d_stats, e->budget_g, e->d_cursors,
                           e->wave_chunks_r ? chunks_for(nk, grid * (BLOCK / 64)) : 8);
      } else if (kind == NEXT_MA) {
        rc = launch_thermal(s, lst, nk, next, e->d_cursors);
        if (rc != ARTIS_OK) return rc;
      } else if (kind == NEXT_BB) {
        hipLaunchKernelGGL(k_blackbody, dim3(nblocks(nk)), dim3(BLOCK), 0, s, env, lst, nk, next, e->d_stats);
      } else {
        hipLaunchKernelGGL(k_slow, dim3(nblocks(nk)), dim3(BLOCK), 0, s, env, lst, nk, next, e->d_stats);
      }
#if ARTIS_OPT_VPKT_ON
      if (kind != NEXT_GAMMA && kind != NEXT_BB) {  // the virtual packets of the events the launch recorded
        HIP_TRY(hipMemsetAsync(e->d_cursors, 0, sizeof(int32_t) * (MAX_CHUNKS + 1), s));
        if (e->vpkt_cont_lds && e->Mh.nbfcontinua <= CONT_LDS_MAX && e->Mh.nbfcontinua > 0)
          hipLaunchKernelGGL((k_vpkt<true, ARTIS_VPKT_TB>), dim3(e->ncu), dim3(ARTIS_VPKT_TB), 0, s, env, e->d_stats, e->d_cursors);
        else
          hipLaunchKernelGGL((k_vpkt<false, BLOCK>), dim3(e->ncu * ARTIS_VPKT_WGS), dim3(BLOCK), 0, s, env, e->d_stats, e->d_cursors);
        HIP_TRY(hipMemsetAsync(e->d_vpkt_count, 0, sizeof(int32_t), s));
      }
#endif
      HIP_TRY(hipEventRecord(e->ev1, s));
      wall_launch += since(t_launch);
      rc = read_counts();
      if (rc != ARTIS_OK) return rc;
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
      e->kms[kind] += ms;
      e->klaunches[kind]++;
      visit_launches++;
      e->kthreads[kind] += nk;
      e->last_nlaunches++;
      if (e->trace)
        fprintf(stderr, "[artis_amd] launch %lld kind %d n=%d %.3f ms -> r %d ma %d slow %d k %d self %d\n", (long long)e->last_nlaunches,
                kind, nk, ms, cnt[NEXT_RPKT], cnt[NEXT_MA], cnt[NEXT_SLOW], cnt[NEXT_KPKT], cnt[NEXT_NKINDS]);
      // the alternate list of this kind becomes its current list (its count moves on the device: no second sync)
      cur[kind] = 1 - cur[kind];
      cnt[kind] = cnt[NEXT_NKINDS];
      HIP_TRY(hipMemcpyAsync(e->d_count + kind, e->d_count + NEXT_NKINDS, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
      if (++guard > 2000000LL) {
        g_last_error = "packet loop did not terminate";
        return ARTIS_ERR_NOTCONVERGED;
      }
    }
  }
  }  // tiles
  if (any_active) e->last_sweeps++;
  if (e->ntiles == 1 || !any_active || all_done) break;
  }  // sweeps
  for (int k = 1; k < NEXT_NKINDS; k++) e->last_propagate_ms += e->kms[k];
  e->last_propagate_ms += e->kms_tail;
  if (e->trace)
    fprintf(stderr, "[artis_amd] host time of the call: %.1f ms = %.1f waiting for the stream + %.1f submitting sorts + %.1f submitting launches + the rest; kernels by their events %.1f ms\n",
            since(wall_t0), wall_sync, wall_sort, wall_launch, e->last_propagate_ms);
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
  if (env.bfrate_kept != nullptr && env.bfev != nullptr)
    hipLaunchKernelGGL(k_bfrate_expand, dim3(nblocks((int64_t)e->Mh.npts_nonempty * 64)), dim3(BLOCK), 0, s, env);
  e->bfrate_kept_dirty = false;
#endif
  if (e->Mh.ncold > 0) {  // what the pool of on-demand records holds at the call's end (since its last emptying): artis_amd_last_pool_usage()
    uint32_t used = 0;
    HIP_TRY(hipMemcpy(&used, e->K.ma_pool_used, sizeof(used), hipMemcpyDeviceToHost));
    e->last_pool_used = std::min<int64_t>(used, env.ma_pool_cap);
    e->last_pool_cap = env.ma_pool_cap;
  }
  int32_t err = 0;
  HIP_TRY(hipMemcpy(&err, e->d_err, sizeof(err), hipMemcpyDeviceToHost));
  if (err != 0) {
    g_last_error = "a kernel raised error flag " + std::to_string(err) + " (an assert_always of the reference would have fired)";
    if (err == 46) g_last_error = "a cell's pool of on-demand macro-atom records is used up (error flag 46): raise ARTIS_AMD_MA_POOLFRAC (or ARTIS_AMD_MA_HOTFRAC)";
    return ARTIS_ERR_NOTCONVERGED;
  }
  return ARTIS_OK;
}

int artis_amd_estimators_zero(artis_amd_engine *e, void *hip_stream) {
  if (!e) return ARTIS_ERR_ARG;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemsetAsync(e->d_est, 0, sizeof(double) * (size_t)e->est_ndoubles, (hipStream_t)hip_stream));
  HIP_TRY(hipMemsetAsync(e->d_stats, 0, sizeof(unsigned long long) * ARTIS_NSTATS, (hipStream_t)hip_stream));
  // (d_bfrate_kept is zero between calls: k_bfrate_expand leaves it so)
  return ARTIS_OK;
}

int artis_amd_estimators_download(artis_amd_engine *e, artis_estimators *est) {
  if (!e || !est) return ARTIS_ERR_ARG;
  HIP_TRY(hipSetDevice(e->device));
  std::vector<double> h((size_t)e->est_ndoubles);
  HIP_TRY(hipMemcpy(h.data(), e->d_est, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
  const int64_t ncell = e->Mh.npts_nonempty;
  const int64_t g = e->Mh.nbfcontinua_ground > 0 ? e->Mh.nbfcontinua_ground : 1;
  const double *src = h.data();
  auto add = [](double *dst, const double *s, int64_t cnt) {
    if (!dst) return;
    for (int64_t i = 0; i < cnt; i++) dst[i] += s[i];
  };
  auto add_strided = [](double *dst, const double *s, int64_t cnt, int64_t stride) {
    if (!dst) return;
    for (int64_t i = 0; i < cnt; i++) dst[i] += s[i * stride];
  };
  add_strided(est->J, src, ncell, 8);
  add_strided(est->nuJ, src + 1, ncell, 8);
  add_strided(est->ffheatingestimator, src + 2, ncell, 8);
  add_strided(est->colheatingestimator, src + 3, ncell, 8);
  add_strided(est->dep_estimator_gamma, src + 4, ncell, 8);
  add_strided(est->dep_estimator_electron, src + 5, ncell, 8);
  add_strided(est->dep_estimator_positron, src + 6, ncell, 8);
  add_strided(est->dep_estimator_alpha, src + 7, ncell, 8);
  add_strided(est->gammaestimator, src + 8 * ncell, ncell * g, 2);
  add_strided(est->bfheatingestimator, src + 8 * ncell + 1, ncell * g, 2);
  add(est->scalars, src + 8 * ncell + 2 * ncell * g, ARTIS_NSCALARS);
  {
    const int64_t nbinest = ARTIS_OPT_MULTIBIN_RADFIELD_MODEL_ON ? ncell * ARTIS_OPT_RADFIELDBINCOUNT : 0;
    const int64_t nbfest = ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON ? ncell * (int64_t)e->Mh.nbfestim : 0;
    const double *ext = src + 8 * ncell + 2 * ncell * g + ARTIS_NSCALARS;
    if (nbinest) {
      add_strided(est->radfieldbin_J, ext, nbinest, 2);
      add_strided(est->radfieldbin_nuJ, ext + 1, nbinest, 2);
    }
    if (nbfest) add(est->bfrate_raw, ext + 2 * nbinest, nbfest);
    const int64_t nlineest = ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON ? ncell * (int64_t)e->Mh.detailed_linecount : 0;
    if (nlineest) {
      add(est->Jb_lu_raw, ext + 2 * nbinest + nbfest, nlineest);
      if (est->Jb_lu_contribcount)
        for (int64_t i = 0; i < nlineest; i++) est->Jb_lu_contribcount[i] += (int64_t)ext[2 * nbinest + nbfest + nlineest + i];
    }
    if (e->nvspec) add(est->vspecpol, ext + 2 * nbinest + nbfest + 2 * nlineest, e->nvspec);
    if (e->nvgrid) add(est->vgrid_flux, ext + 2 * nbinest + nbfest + 2 * nlineest + e->nvspec, e->nvgrid);
  }
  if (est->stats) {
    unsigned long long st[ARTIS_NSTATS];
    HIP_TRY(hipMemcpy(st, e->d_stats, sizeof(st), hipMemcpyDeviceToHost));
    for (int i = 0; i < ARTIS_NSTATS; i++) est->stats[i] += (int64_t)st[i];
  }
  return ARTIS_OK;
}

int artis_amd_estimators_devptr(artis_amd_engine *e, void **dptr, int64_t *ndoubles) {
  if (!e || !dptr || !ndoubles) return ARTIS_ERR_ARG;
  *dptr = e->d_est;
  *ndoubles = e->est_ndoubles;
  return ARTIS_OK;
}

int artis_amd_comm_unique_id(void *id_out) {
  if (!id_out) return ARTIS_ERR_ARG;
  if (!rccl_api().ok) {
    g_last_error = "librccl could not be loaded";
    return ARTIS_ERR_RCCL;
  }
  ncclUniqueId id;
  RCCL_TRY(rccl_api().GetUniqueId(&id));
  std::memcpy(id_out, id.internal, ARTIS_AMD_COMM_ID_BYTES);
  return ARTIS_OK;
}

int artis_amd_comm_init(artis_amd_engine *e, int nranks, int rank, const void *id_bytes) {
  if (!e || !id_bytes || nranks < 1 || rank < 0 || rank >= nranks) {
    g_last_error = "bad communicator arguments";
    return ARTIS_ERR_ARG;
  }
  if (!rccl_api().ok) {
    g_last_error = "librccl could not be loaded";
    return ARTIS_ERR_RCCL;
  }
  HIP_TRY(hipSetDevice(e->device));
  if (e->comm) {
    (void)rccl_api().CommDestroy(e->comm);
    e->comm = nullptr;
  }
  ncclUniqueId id;
  std::memcpy(id.internal, id_bytes, ARTIS_AMD_COMM_ID_BYTES);
  RCCL_TRY(rccl_api().CommInitRank(&e->comm, nranks, id, rank));
  return ARTIS_OK;
}

int artis_amd_comm_count(artis_amd_engine *e, void *nccl_comm, int *nranks) {
  if (!e || !nranks) return ARTIS_ERR_ARG;
  ncclComm_t comm = nccl_comm ? (ncclComm_t)nccl_comm : e->comm;
  if (!comm || !rccl_api().ok || !rccl_api().CommCount) {
    g_last_error = "no communicator (artis_amd_comm_init), or librccl without ncclCommCount";
    return ARTIS_ERR_RCCL;
  }
  RCCL_TRY(rccl_api().CommCount(comm, nranks));
  return ARTIS_OK;
}

int artis_amd_allreduce_estimators(artis_amd_engine *e, void *nccl_comm, void *hip_stream) {
  if (!e) return ARTIS_ERR_ARG;
  ncclComm_t comm = nccl_comm ? (ncclComm_t)nccl_comm : e->comm;
  if (!comm) {
    g_last_error = "no communicator: pass an ncclComm_t or call artis_amd_comm_init() first";
    return ARTIS_ERR_ARG;
  }
  if (!rccl_api().ok) {
    g_last_error = "librccl could not be loaded";
    return ARTIS_ERR_RCCL;
  }
  HIP_TRY(hipSetDevice(e->device));
  // one in-place sum over the whole block [J | nuJ | ffheat | colheat | gamma | bfheat | dep_* | scalars]
  RCCL_TRY(rccl_api().AllReduce(e->d_est, e->d_est, (size_t)e->est_ndoubles, ncclDouble, ncclSum, comm, (hipStream_t)hip_stream));
  return ARTIS_OK;
}

int artis_amd_update_packets(artis_amd_engine *e, artis_packet *packets, int64_t npackets, artis_estimators *est) {
  int rc = artis_amd_packets_upload(e, packets, npackets);
  if (rc != ARTIS_OK) return rc;
  rc = artis_amd_estimators_zero(e, nullptr);
  if (rc != ARTIS_OK) return rc;
  {
    // keep UPDATECELL of the populate that belongs to this timestep (with several cache tiles the populates happen
    // inside the update and count themselves)
    unsigned long long ncell = (e->ntiles == 1) ? (unsigned long long)e->Mh.npts_nonempty : 0ull;
    HIP_TRY(hipMemcpy(e->d_stats + ARTIS_STAT_UPDATECELL, &ncell, sizeof(ncell), hipMemcpyHostToDevice));
  }
  rc = artis_amd_update_packets_device(e, nullptr);
  if (rc != ARTIS_OK) return rc;
  rc = artis_amd_packets_download(e, packets, npackets);
  if (rc != ARTIS_OK) return rc;
  if (est) rc = artis_amd_estimators_download(e, est);
  return rc;
}

int artis_amd_last_kernel_ms(artis_amd_engine *e, double *propagate_ms, int64_t *nlaunches) {
  if (!e) return ARTIS_ERR_ARG;
  if (propagate_ms) *propagate_ms = e->last_propagate_ms;
  if (nlaunches) *nlaunches = e->last_nlaunches;
  return ARTIS_OK;
}

int artis_amd_last_kernel_launches(artis_amd_engine *e, int64_t *rpkt_launches, int64_t *thermal_launches) {
  if (!e) return ARTIS_ERR_ARG;
  if (rpkt_launches) *rpkt_launches = e->klaunches[NEXT_RPKT];
  if (thermal_launches) *thermal_launches = e->klaunches[NEXT_MA] + e->klaunches[NEXT_KPKT];
  return ARTIS_OK;
}

int artis_amd_last_kernel_breakdown(artis_amd_engine *e, double *rpkt_ms, int64_t *rpkt_threads, double *thermal_ms,
                                    int64_t *thermal_threads) {
  if (!e) return ARTIS_ERR_ARG;
  if (rpkt_ms) *rpkt_ms = e->kms[NEXT_RPKT];
  if (rpkt_threads) *rpkt_threads = e->kthreads[NEXT_RPKT];
  if (thermal_ms) *thermal_ms = e->kms[NEXT_MA] + e->kms[NEXT_KPKT];
  if (thermal_threads) *thermal_threads = e->kthreads[NEXT_MA] + e->kthreads[NEXT_KPKT];
  return ARTIS_OK;
}

int artis_amd_last_tiling_fills(artis_amd_engine *e, int64_t *sparse_fills, int64_t *cells_filled) {
  if (!e) return ARTIS_ERR_ARG;
  if (sparse_fills) *sparse_fills = e->last_sparse_fills;
  if (cells_filled) *cells_filled = e->last_cells_filled;
  return ARTIS_OK;
}

int artis_amd_last_pool_resets(artis_amd_engine *e, int64_t *resets) {
  if (!e) return ARTIS_ERR_ARG;
  if (resets) *resets = e->last_pool_resets;
  return ARTIS_OK;
}

int artis_amd_record_tiers(artis_amd_engine *e, double *hot_fraction, int32_t *ncold_levels, int64_t *pool_slots) {
  if (!e) return ARTIS_ERR_ARG;
  if (hot_fraction) *hot_fraction = e->ma_hotfrac;
  if (ncold_levels) *ncold_levels = e->Mh.ncold;
  if (pool_slots) *pool_slots = e->Mh.ma_pool_slots;
  return ARTIS_OK;
}
int artis_amd_last_pool_usage(artis_amd_engine *e, int64_t *units_used, int64_t *units_cap) {
  if (!e) return ARTIS_ERR_ARG;
  if (units_used) *units_used = e->last_pool_used;
  if (units_cap) *units_cap = e->last_pool_cap;
  return ARTIS_OK;
}
int artis_amd_last_thermal_variants(artis_amd_engine *e, int32_t *mask) {
  if (!e || !mask) return ARTIS_ERR_ARG;
  *mask = e->thermal_variants;
  return ARTIS_OK;
}
int artis_amd_last_tiling_parked(artis_amd_engine *e, int64_t *parked) {
  if (!e) return ARTIS_ERR_ARG;
  if (parked) *parked = e->last_parked;
  return ARTIS_OK;
}

int artis_amd_last_tiling(artis_amd_engine *e, int64_t *sweeps, int64_t *tile_fills, double *fill_ms, int64_t *listed) {
  if (!e) return ARTIS_ERR_ARG;
  if (sweeps) *sweeps = e->last_sweeps;
  if (tile_fills) *tile_fills = e->last_tile_fills;
  if (fill_ms) *fill_ms = e->last_fill_ms;
  if (listed) *listed = e->last_listed;
  return ARTIS_OK;
}

int artis_amd_last_kernel_table(artis_amd_engine *e, double ms[4], int64_t launches[4], int64_t packets[4]) {
  if (!e) return ARTIS_ERR_ARG;
  const int kinds[4] = {NEXT_RPKT, NEXT_MA, NEXT_KPKT, NEXT_SLOW};
  for (int i = 0; i < 4; i++) {
    if (ms) ms[i] = e->kms[kinds[i]];
    if (launches) launches[i] = e->klaunches[kinds[i]];
    if (packets) packets[i] = e->kthreads[kinds[i]];
  }
  return ARTIS_OK;
}

int artis_amd_last_kernel_ms_by_kind(artis_amd_engine *e, double ms[8], int64_t launches[8]) {
  if (!e || !ms) return ARTIS_ERR_ARG;
  const int kinds[5] = {NEXT_RPKT, NEXT_MA, NEXT_SLOW, NEXT_GAMMA, NEXT_BB};
  for (int i = 0; i < 8; i++) {
    ms[i] = 0.;
    if (launches) launches[i] = 0;
  }
  for (int i = 0; i < 5; i++) {
    ms[i] = e->kms[kinds[i]];
    if (launches) launches[i] = e->klaunches[kinds[i]];
  }
  ms[5] = e->kms_tail;
  ms[6] = e->last_fill_ms;
  return ARTIS_OK;
}

int artis_amd_debug_visit_counts(artis_amd_engine *e, uint32_t *counts, int64_t n) {
  if (!e || !counts || n != (int64_t)e->Mh.npts_nonempty * e->Mh.nlevels) {
    g_last_error = "visit counts: [npts_nonempty][nlevels] uint32 expected";
    return ARTIS_ERR_ARG;
  }
  if (e->d_visit_counts == nullptr) {
    g_last_error = "this library was not built with -DARTIS_VISIT_COUNTS (or no propagation call has run yet)";
    return ARTIS_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpy(counts, e->d_visit_counts, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
  return ARTIS_OK;
}

int artis_amd_debug_cellcache(artis_amd_engine *e, int c, double *levelpops, double *maprocessrates, double *matrans,
                              double *allcont_nnlevel, double *allcont_departure, double *allcont_edgepart,
                              uint64_t *allcont_keepbits, double *corrphotoioncoeff, double *cooling_contrib,
                              double *ion_cooling_contribs, double *chi_ff_nnionpart) {
  if (!e || !e->have_cells || c < 0 || c >= e->Mh.npts_nonempty) {
    g_last_error = "bad cell index or no cell state";
    return ARTIS_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  const DevModel &h = e->Mh;
  const int cell = c;
  if (e->ntiles == 1) {  // the cache has to be filled
    if (e->tile_valid_lo != 0) {
      int rc = populate_tile(e, nullptr);
      if (rc != ARTIS_OK) return rc;
    }
  } else {  // the cell has to be resident
    int64_t nfilled = 0;
    int rc = make_resident(e, std::vector<int32_t>{(int32_t)c}, nullptr, &nfilled);
    if (rc != ARTIS_OK) return rc;
    c = e->h_krow[(size_t)c];  // its row
  }
#define DL(dst, f, T, per)                                                                                            \
  if (dst && (per) > 0) HIP_TRY(hipMemcpy(dst, e->K.f + (int64_t)c * (per), sizeof(T) * (size_t)(per), hipMemcpyDeviceToHost));
  DL(levelpops, levelpops, double, h.nlevels)
  if (maprocessrates || matrans) {
    // the rates from the records; the cumulative sums -- which the records hold as filters only -- re-added on the device from
    // the transitions' terms, the way an undecided draw gets them. A record whose filters are not those of the sequential
    // form fails the call.
    double *d_rates = nullptr, *d_trans = nullptr;
    int32_t *d_bad = nullptr;
    HIP_TRY(hipMalloc((void **)&d_rates, sizeof(double) * (size_t)(h.nlevels * 9 + 1)));
    HIP_TRY(hipMalloc((void **)&d_trans, sizeof(double) * (size_t)(h.nmatransblock + 1)));
    HIP_TRY(hipMalloc((void **)&d_bad, sizeof(int32_t)));
    HIP_TRY(hipMemset(d_bad, 0, sizeof(int32_t)));
    const Env env = make_env(e);
    hipLaunchKernelGGL(k_debug_macache, dim3(nblocks(h.nlevels)), dim3(BLOCK), 0, nullptr, env, cell, d_rates, d_trans, d_bad);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    int32_t bad = 0;
    HIP_TRY(hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost));
    if (maprocessrates) HIP_TRY(hipMemcpy(maprocessrates, d_rates, sizeof(double) * (size_t)h.nlevels * 9, hipMemcpyDeviceToHost));
    if (matrans) HIP_TRY(hipMemcpy(matrans, d_trans, sizeof(double) * (size_t)h.nmatransblock, hipMemcpyDeviceToHost));
    (void)hipFree(d_rates);
    (void)hipFree(d_trans);
    (void)hipFree(d_bad);
    if (bad != 0) {
      g_last_error = std::to_string(bad) + " filter entries of the cell's macro-atom records differ from the sequential form";
      return ARTIS_ERR_NOTCONVERGED;
    }
  }
  DL(allcont_nnlevel, allcont_nnlevel, double, h.nbfcontinua)
  DL(allcont_departure, allcont_departure, double, h.nbfcontinua)
  DL(allcont_edgepart, allcont_edgepart, double, h.nbfcontinua)
  if (allcont_keepbits && h.nbfcontinua > 0)  // rows are padded to nkeepwords; the caller's buffer holds ceil(nbfcontinua/64) words
    HIP_TRY(hipMemcpy(allcont_keepbits, e->K.allcont_keepbits + (int64_t)c * h.nkeepwords, sizeof(uint64_t) * (size_t)((h.nbfcontinua + 63) / 64),
                      hipMemcpyDeviceToHost));
  DL(corrphotoioncoeff, corrphotoioncoeff, double, h.nphixstargets_total)
  DL(cooling_contrib, cooling_contrib, double, h.ncoolingterms)
  DL(ion_cooling_contribs, ion_cooling_contribs, double, h.nions)
  DL(chi_ff_nnionpart, chi_ff_nnionpart, double, 1)
#undef DL
  return ARTIS_OK;
}

}  // extern "C"
