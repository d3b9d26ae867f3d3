// physics.h -- bodies of the packet kernels.
//
// Everything a GPU thread does to one packet (or one cell-cache entry) is written here as inline
// functions over the POD views of tables.h. The HIP engine (artis_engine.hip) wraps them in
// __global__ kernels; the host-emulation test build (tests/hostemu/emu.cc) compiles the very same
// bodies with g++ to run differential tests against the CPU oracle on machines without a GPU.
//
// Floating-point discipline: build with -ffp-contract=off. The order of operations (including the
// reference's float/double mixing) follows the reference functions cited at each body, so that a
// packet history is reproducible field by field.
#pragma once
#include <math.h>
#include <stdint.h>

#include "tables.h"

#if defined(__HIPCC__) && !defined(ARTIS_HOST_EMU)
#define AHD __host__ __device__ inline
#define ANOINLINE static __host__ __device__ __attribute__((noinline))
#else
#define AHD inline
#define ANOINLINE static __attribute__((noinline))
#endif

namespace artis {

// ---- constants.h of the reference
constexpr double CLIGHT = 2.99792458e+10;
constexpr double CLIGHT_PROP = CLIGHT;
constexpr double HPLANCK = 6.6260755e-27;
constexpr double PI = 3.14159265358979323846;
constexpr double EV = 1.6021772e-12;
constexpr double SIGMA_T = 6.6524e-25;
constexpr double ME = 9.1093897e-28;
constexpr double MH = 1.67352e-24;
constexpr double MEV = 1.6021772e-6;
constexpr double MSUN = 1.98855e+33;  // constants.h:27
constexpr double DAY = 86400.;        // constants.h:35
constexpr double THOMSON_LIMIT = 1e-2;  // constants.h:38
constexpr double NU_100KEV = 2.41326e+19, NU_1MEV = 2.41326e+20, NU_1P022MEV = 2.46636e+20, NU_1P5MEV = 3.61990e+20;  // gammapkt.cc:64-67
constexpr double KB = 1.38064852e-16;
constexpr double SAHACONST = 2.0706659e-16;
constexpr double EULERGAMMA = 0.577215664901532860606512090082402431;
constexpr double CLIGHTSQUARED = CLIGHT * CLIGHT;
constexpr double CLIGHTSQUAREDOVERTWOH = CLIGHT * CLIGHT / (2 * HPLANCK);
constexpr double HOVERKB = HPLANCK / KB;
constexpr double HCLIGHTOVERFOURPI = HPLANCK * CLIGHT / (4 * PI);
constexpr double H_ionpot = 13.5979996 * EV;
constexpr double C_0 = 5.465e-11;
constexpr double DBLMAX = 1.7976931348623157e308;
constexpr double DBLMIN = 2.2250738585072014e-308;
constexpr int MA_N = ARTIS_MA_ACTION_COUNT;

#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned long long stat_t;  // per-block LDS counters (ds_add_u64): X_MA_JUMPS alone passes 2^32 at realistic sizes
#define ARTIS_STAT_ADD(env, i, v) atomicAdd(&(env).stats[(i)], (stat_t)(v))
#ifdef ARTIS_EST_NOADD  // (measurement only: the kernels without their estimator additions)
#define ARTIS_EST_ADD(ptr, v)                                 \
  do {                                                        \
    const double est_v = (v);                                 \
    if (est_v == 1.2345e-300) unsafeAtomicAdd((ptr), est_v);  \
  } while (0)
#else
#define ARTIS_EST_ADD(ptr, v) unsafeAtomicAdd((ptr), (v))
#endif
#else
typedef unsigned long long stat_t;
#define ARTIS_STAT_ADD(env, i, v) ((env).stats[(i)] += (stat_t)(v))
#define ARTIS_EST_ADD(ptr, v) (*(ptr) += (v))
#endif
#define ARTIS_STAT(env, i) ARTIS_STAT_ADD(env, i, 1)
// -DARTIS_PROFILE (device only): wave-cycle accounting into the spare stats slots, units of 16 clocks, charged by
// the first active lane of the wave for the code between two marks
#if defined(ARTIS_PROFILE) && ARTIS_OPT_VPKT_ON
#error "-DARTIS_PROFILE keeps its phase clocks in the stats slots 48..52, which a VPKT_ON build uses for nvpkt_created / nvpkt_esc_*"
#endif
#if defined(ARTIS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define PROF_BEGIN() long long prof_t = clock64()
#define PROF_MARK(env, slot)                                                                   \
  do {                                                                                         \
    const long long prof_now = clock64();                                                      \
    if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1)                         \
      atomicAdd(&(env).stats[(slot)], (stat_t)((prof_now - prof_t) >> 4));                     \
    prof_t = prof_now;                                                                         \
  } while (0)
#else
#define PROF_BEGIN() ((void)0)
#define PROF_MARK(env, slot) ((void)0)
#endif

// -DARTIS_PROFILE_MA (device only): wave clocks of the stages of one macro-atom transition (ma_jump_internal) into the
// stats slots 59..62, wave-rounds into 63: rates read | process drawn | direction searched | target read. Each read
// stage ends in an explicit wait, so the stage's clocks are its memory latency as the wave sees it.
#if defined(ARTIS_PROFILE_MA) && defined(__HIP_DEVICE_COMPILE__)
#define MA_PROF_BEGIN() long long ma_prof_t = clock64()
#define MA_PROF_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MA_PROF_MARK(env, slot)                                                                \
  do {                                                                                         \
    const long long prof_now = clock64();                                                      \
    if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1)                         \
      atomicAdd(&(env).stats[(slot)], (stat_t)(prof_now - ma_prof_t));                         \
    ma_prof_t = prof_now;                                                                      \
  } while (0)
#else
#define MA_PROF_BEGIN() ((void)0)
#define MA_PROF_WAIT() ((void)0)
#define MA_PROF_MARK(env, slot) ((void)0)
#endif

// add to a per-cell estimator: the workgroup's LDS accumulator when the kernel keeps one for this cell (Env::cellest_lds)
enum { CELLEST_COLHEAT = 0, CELLEST_DEPGAMMA = 0, CELLEST_J = 0, CELLEST_NUJ = 1, CELLEST_FFHEAT = 2 };  // per kernel
template <typename EnvT>
AHD void cellest_add(const EnvT &env, double *global_array, int kind, int c, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (c < env.cellest_n && env.cellest_owner[kind] == global_array) {
    __hip_atomic_fetch_add((__attribute__((address_space(3))) double *)(env.cellest_lds + (kind * env.cellest_n) + c), v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }
#endif
  ARTIS_EST_ADD(&global_array[(int64_t)c * env.est_stride], v);
}

#ifndef ESTCACHE_SLOTS
#define ESTCACHE_SLOTS 128  // cells per wave in its cache of accumulators (a power of two; 28 bytes of LDS each)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// the wave's cache of per-cell accumulators (Env::estcache): J += a, nuJ += b, ffheating += f of non-empty cell c
template <typename EnvT>
__device__ inline void est_cache_add(const EnvT &env, int c, double a, double b, double f) {
  typedef __attribute__((address_space(3))) int32_t lds_i32;
  typedef __attribute__((address_space(3))) double lds_f64;
  volatile lds_i32 *tags = (volatile lds_i32 *)env.estcache_tag;  // [64] the slots' cells, [64] claims
  volatile lds_i32 *claim = tags + ESTCACHE_SLOTS;
  volatile lds_f64 *vsums = (volatile lds_f64 *)env.estcache;
  const int slot = c & (ESTCACHE_SLOTS - 1);
  if (tags[slot] != c) {
    // the lanes of this instruction that miss on one slot agree on one of them (the last store wins); it takes the slot for its cell
    const int lane = (int)(threadIdx.x & 63);
    claim[slot] = lane;
    if (claim[slot] == lane) {
      const int old = tags[slot];
      if (old >= 0) {  // what the slot held goes to the cell's record in memory
        const double s0 = vsums[(slot * 3) + 0], s1 = vsums[(slot * 3) + 1], s2 = vsums[(slot * 3) + 2];
        if (s0 != 0.) ARTIS_EST_ADD(&env.E.J[(int64_t)old * env.est_stride], s0);
        if (s1 != 0.) ARTIS_EST_ADD(&env.E.nuJ[(int64_t)old * env.est_stride], s1);
        if (s2 != 0.) ARTIS_EST_ADD(&env.E.ffheatingestimator[(int64_t)old * env.est_stride], s2);
      }
      vsums[(slot * 3) + 0] = 0.;
      vsums[(slot * 3) + 1] = 0.;
      vsums[(slot * 3) + 2] = 0.;
      tags[slot] = c;
    }
  }
  // (read again by every lane: a lane whose cell held the slot a moment ago may have lost it to a lane of the same instruction)
  if (tags[slot] == c) {
    __attribute__((address_space(3))) double *sl = (__attribute__((address_space(3))) double *)(env.estcache + (slot * 3));
    __hip_atomic_fetch_add(sl + 0, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(sl + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (f != 0.) __hip_atomic_fetch_add(sl + 2, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    ARTIS_EST_ADD(&env.E.J[(int64_t)c * env.est_stride], a);
    ARTIS_EST_ADD(&env.E.nuJ[(int64_t)c * env.est_stride], b);
    if (f != 0.) ARTIS_EST_ADD(&env.E.ffheatingestimator[(int64_t)c * env.est_stride], f);
  }
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// ... with one sum per cell (k_thermal: global_array = colheatingestimator); the same protocol
template <typename EnvT>
__device__ inline void est_cache_add_one(const EnvT &env, double *global_array, int c, double v) {
  typedef __attribute__((address_space(3))) int32_t lds_i32;
  typedef __attribute__((address_space(3))) double lds_f64;
  volatile lds_i32 *tags = (volatile lds_i32 *)env.estcache_tag;
  volatile lds_i32 *claim = tags + ESTCACHE_SLOTS;
  volatile lds_f64 *vsums = (volatile lds_f64 *)env.estcache;
  const int slot = c & (ESTCACHE_SLOTS - 1);
  if (tags[slot] != c) {
    const int lane = (int)(threadIdx.x & 63);
    claim[slot] = lane;
    if (claim[slot] == lane) {
      const int old = tags[slot];
      if (old >= 0) {
        const double s0 = vsums[slot];
        if (s0 != 0.) ARTIS_EST_ADD(&global_array[(int64_t)old * env.est_stride], s0);
      }
      vsums[slot] = 0.;
      tags[slot] = c;
    }
  }
  if (tags[slot] == c)
    __hip_atomic_fetch_add((__attribute__((address_space(3))) double *)(env.estcache + slot), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    ARTIS_EST_ADD(&global_array[(int64_t)c * env.est_stride], v);
}
#endif

template <typename EnvT>
AHD void scalar_add(const EnvT &env, int idx, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (env.scalars_lds != nullptr) {
    __hip_atomic_fetch_add((__attribute__((address_space(3))) double *)(env.scalars_lds + idx), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }
#endif
  ARTIS_EST_ADD(&env.E.scalars[idx], v);
}

AHD double pow2(double x) { return x * x; }
AHD double pow3(double x) { return x * x * x; }
AHD double dmin(double a, double b) { return (b < a) ? b : a; }
AHD double dmax(double a, double b) { return (a < b) ? b : a; }
AHD double dclamp(double v, double lo, double hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }

// Everything a thread needs besides its packet.
struct Env {
  DevModel M;
  DevCells C;
  DevCache K;
  DevStep S;
  DevEst E;
  PktStore P;
  stat_t *stats;       // [ARTIS_NSTATS] LDS on the GPU, plain memory in the emulation
  // groundcont_gamma_contr (rpkt.h:63) of every packet as a compact list of its non-zero entries, packet-major:
  // gamma_n[slot] entries (gamma_gi[slot*nbfg + j], gamma_ws[slot*nbfg + j]), j ascending in ground-continuum index
  double *gamma_ws;
  int32_t *gamma_gi;
  int32_t *gamma_n;
  int32_t *errflag;    // set non-zero when an assert_always of the reference would fire
  // Cell-cache rows. Every cell's row is resident (the usual case): krow_tab == nullptr, the row of non-empty cell c is c. A cache that does not
  // fit in HBM has rows for a SET of cells at a time (a "tile"): krow_tab[c] is the row that holds cell c's cache, or -1 -- a packet that needs the
  // cache of such a cell waits until the host has made the cell resident (classify()). Every access of a cache array goes through krow().
  const int32_t *krow_tab;
  int32_t tile_all;  // every cell is resident: in_tile() needs no look-up
  // A FILL works on the cells fill_cells[0..nfill) (a tiled cache: the cells that became resident), or -- fill_cells == nullptr -- on the
  // cells [tile_lo, tile_hi) (the whole cache at once, or a batch of it).
  int32_t tile_lo, tile_hi;
  const int32_t *fill_cells;
  int32_t nfill;
  // scratch of the cell-cache population: [cell of the fill][M.nupcum] the collisional-excitation cooling terms of the levels'
  // upward transitions (k_matrans), turned into the ions' running sums in place (k_cooling_chain) and into the records' cooling
  // filters (k_collexc_filter); not kept
  double *collexc_terms;
  int32_t cont_in_lds;  // M.cont_pack points into LDS (k_rpkt<true>)
  int32_t ma_tables_in_lds;  // M.level_pack and M.alltrans_tlevel16 point into LDS (k_thermal<.., true>)
  // Per-cell estimators of a model with FEW cells (1D / 2D models, small grids): every packet of the launch adds to one of
  // a few addresses, and device-wide atomics on one address are serialised in memory (measured, 30 shells, 1e7 packets:
  // k_thermal 2330 ms with its one atomic per walk on colheatingestimator[cell], 590 ms without it). A workgroup then
  // accumulates in LDS, cellest_lds[kind * cellest_n + cell] for cell < cellest_n, and adds its sums to the global
  // arrays once, when the kernel ends. cellest_n_t / _r / _g: set by the host for k_thermal / k_rpkt / k_gamma (0 = off).
  // ... and of a model with MANY cells (round 6): the three additions of an r-packet step (J, nuJ, ffheating of the cell the packet crossed) cost k_rpkt 16 of
  // its 224 ms as device-wide atomics -- 1.1e9 requests per step, carried out beyond the XCDs' L2s. A wave's packets sit in ~21 cells that change slowly
  // (the work list is sorted by cell), so every WAVE keeps a small direct-mapped cache of accumulators in LDS: estcache_tag[slot] = the cell whose three
  // sums estcache[slot * 3 ...] holds (-1: none), slot = cell & (ESTCACHE_SLOTS - 1). An addition whose cell holds the slot is an LDS addition; one that does not evicts the
  // slot's cell (its sums go to the global record: three atomics for however many additions they stand for) or, where two cells of one instruction
  // want one slot, goes to memory itself. The wave empties its cache when the kernel ends. Null: every addition is a global atomic (est_cache_add()).
  // k_thermal keeps the one sum it adds per cell (colheatingestimator: one addition per collisional de-excitation / recombination of a macro-atom) the
  // same way (estcache_nv = 1, est_cache_add_one(); in the LDS its workgroup array of a few-cells model would take): with 4e3 ... 2e4 cells -- above the
  // few-cells caps, thousands of packets per cell -- its additions otherwise queue on a few addresses.
  double *estcache;
  int32_t *estcache_tag;
  int32_t estcache_nv;  // sums per slot: 3 (k_rpkt) / 1 (k_thermal); 0: no cache in this kernel
  int32_t estcache_on;  // host switch (ARTIS_AMD_ESTCACHE=0: off)
  double *cellest_lds;
  const double *cellest_owner[3];  // the global array each kind of the running kernel stands for (anything else: global add)
  int32_t cellest_n, cellest_n_t, cellest_n_r, cellest_n_g;
  // the per-timestep scalar sums (ARTIS_SCALAR_*: ONE address each for every packet of the launch) of the running
  // kernel's workgroup in LDS, added to E.scalars when the kernel ends; null: global atomics. scalars_in_lds: host switch
  double *scalars_lds;
  int32_t scalars_in_lds;
  // Layout of the per-cell estimators in HBM. On the device the eight per-cell sums {J, nuJ, ffheating, colheating,
  // dep_gamma, dep_electron, dep_positron, dep_alpha} of a cell are ONE 64-byte record (est_stride = 8: E.J[c * 8] etc.), so that
  // the three additions of an r-packet step (rpkt.cc:502-541) land in one sector instead of three arrays' sectors; and
  // gammaestimator / bfheatingestimator of a (cell, ground continuum) are neighbours (pair_stride = 2). The host emulation adds
  // into the caller's separate arrays (strides 1).
  int32_t est_stride, pair_stride;
  // 1: macro-atom transitions are decided on the f64 rates and sums only (ARTIS_AMD_MAFILTERS=0: the filters of tables.h
  // switched off, for the test that finds the same packets either way)
  int32_t ma_filters_off;
  // deferred detailed bound-free estimator updates (DETAILED_BF builds on the GPU; null: added in place)
  BfEvent *bfev;
  int32_t *bfev_count;
  int32_t bfev_cap;
  // ... and where the deferred updates are summed: [cell][place in the cell's list of kept continua] (null: in bfrate_raw)
  double *bfrate_kept;
  // recorded virtual-packet events (VPKT_ON builds on the GPU; null: traced in place)
  VpktSeed *vpkt_queue;
  int32_t *vpkt_count;
  int32_t vpkt_cap;
  uint32_t ma_pool_cap;  // units (MAPOOL_UNIT slots) in DevCache::ma_pool (tables.h "ON-DEMAND RECORDS")
  int32_t *ma_pool_full;  // set by a lane that found the pool used up: the host empties the pool before the next slow-path launch (the records are filled again on demand)
  // set by a kernel whose waves FILL pool records while others read them (k_tail, k_slow): a cold level's published place is then read with
  // acquire semantics (ma_rowtab_acquire), pairing with the filler's release store. The thermal kernels read records of earlier launches only.
  int32_t ma_concurrent_fill;
#ifdef ARTIS_VISIT_COUNTS
  // (measurement build, tools/visit_sparsity.py) [cell][level] macro-atom transitions drawn in that level's record this call
  uint32_t *visit_counts;
#endif
};
// the row of the cell cache that holds non-empty cell c (which has to be resident: in_tile())
AHD int64_t krow(const Env &env, int c) { return env.krow_tab != nullptr ? (int64_t)env.krow_tab[c] : (int64_t)c; }
// the packet's cell is empty (no cache needed) or its cache row is resident
AHD bool in_tile(const Env &env, int cellindex) {
  if (env.tile_all) return true;
  const int c = env.M.propcell_nonemptymgi[cellindex];
  if (c < 0) return true;
  return env.krow_tab[c] >= 0;
}

// Hot packet state, kept in registers.
struct Pkt {
  uint32_t s0, s1, s2, s3;
  double prop_time, px, py, pz, dx, dy, dz, nu_cmf, e_cmf, nu_rf, e_rf, stokes_q, stokes_u;
  int32_t next_trans, nscatterings, type, cellindex;
  // an activated macro-atom that has not deactivated yet (ma_level < 0: none); see tables.h PktHot
  int32_t ma_element, ma_ion, ma_level, ma_line, ma_origin;
  int32_t pend, pend_arg;  // see tables.h PktHot
  int32_t chi_mgi;         // cell of the packet's ContinuumOpacity, < 0 = not valid (the r-packet kernel keeps it in Chi)
  int32_t emissiontype, trueemissiontype, absorptiontype;
  int32_t flags;           // PKT_FLAG_*
};
// PEND_MA_SEARCH / _RADSEARCH / PEND_KPKT_COLLEXC (round 4): a search of the active macro-atom's internal transition / of its radiative
// de-excitation / of a k-packet's collisional-excitation cooling term that the records' filters could not decide; the draw is in
// pend_arg (24 bits; bit 24 of PEND_MA_SEARCH: downward), and the slow-path kernel re-adds the sums (physics.h ma_search_exact,
// kpkt_collexc_exact) and carries on. k_thermal hands these over so that the rate-coefficient code is not in it at all.
// PEND_RPKT_ABSORB (round 5): a continuum event of an r-packet that is not an electron scattering -- a free-free or bound-free absorption,
// 5e-4 of the r-packet steps -- with the event's draw in pend_arg; the slow-path kernel carries it out (rpkt_slow_absorption), so that the
// r-packet kernel holds no second copy of the opacity sum (the selection of the absorbing continuum, rpkt.cc:455-470).
// PEND_MA_FILL (round 5, tables.h "ON-DEMAND RECORDS"): the macro-atom stands in a cold level whose record does not exist in this cell yet; the
// slow-path kernel fills it (ma_slow_fill) and the walk goes on.
enum { PEND_NONE = 0, PEND_MA_ACTION = 2, PEND_KPKT_FB = 3, PEND_MA_SEARCH = 4, PEND_MA_RADSEARCH = 5, PEND_KPKT_COLLEXC = 6, PEND_MA_FILL = 7, PEND_RPKT_ABSORB = 8 };

// ContinuumOpacity (rpkt.h:70); groundcont_gamma_contr lives in env.gamma_ws
struct Chi {
  double nu, chi_escatter, chi_freefree_heat, chi_boundfree;
  int32_t nonemptymgi;
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
  // the window of continua [bf_begin, bf_end) that calculate_chi_bf_gammacontr() walked at nu (Phixslist::allcontbegin /
  // allcontend, rpkt.cc:762-768), kept while the opacity stays in registers; bf_end < 0: not known (loaded with the packet)
  int32_t bf_begin, bf_end;
#endif
};

struct MAState {
  int32_t element, ion, level, activatingline;
};

AHD void fail(const Env &env, int code) {
  if (env.errflag) *env.errflag = code;
}

// ---------------------------------------------------------------- RNG: random.h:101-136, 141-164, 178-202
AHD uint32_t rotl32(uint32_t x, unsigned k) { return (x << k) | (x >> (32U - k)); }
AHD uint32_t rng_next(Pkt &p) {
  const uint32_t result = rotl32(p.s0 + p.s3, 7U) + p.s0;
  const uint32_t t = p.s1 << 9U;
  p.s2 ^= p.s0;
  p.s3 ^= p.s1;
  p.s1 ^= p.s2;
  p.s0 ^= p.s3;
  p.s2 ^= t;
  p.s3 = rotl32(p.s3, 11U);
  return result;
}
// random.h:49 rng_uniform() redraws while the value is 1. It never is: the 24-bit integer is at most 2^24 - 1, exact as a
// float, and the product with 2^-24 is exact, so the largest value is 1 - 2^-24. Without the (dead) loop a draw is
// straight-line code that the compiler can schedule under the latency of loads in flight.
AHD float rng_uniform(Pkt &p) { return (float)(rng_next(p) >> 8U) * 0x1.0p-24F; }
// the same draw as its 24-bit integer u (rng_uniform() = u * 2^-24 exactly) and back
AHD uint32_t rng_u24(Pkt &p) { return rng_next(p) >> 8U; }
AHD float rng_u24_value(uint32_t u) { return (float)u * 0x1.0p-24F; }
AHD float rng_uniform_pos(Pkt &p) {  // random.h:59: redraw while the value is 0 (one draw in 2^24)
  float z = rng_uniform(p);
  while (!(z > 0)) z = rng_uniform(p);
  return z;
}

// ---------------------------------------------------------------- vectors.h
AHD double vlen(const double v[3]) {
  double sq = 0.;
  for (int i = 0; i < 3; i++) sq += pow2(v[i]);
  return sqrt(sq);
}
AHD double vdot(const double x[3], const double y[3]) {
  double s = 0.;
  for (int i = 0; i < 3; i++) s += x[i] * y[i];
  return s;
}
AHD void vnorm(const double in[3], double out[3]) {
  const double m = vlen(in);
  out[0] = in[0] / m;
  out[1] = in[1] / m;
  out[2] = in[2] / m;
}
AHD void vcross(const double a[3], const double b[3], double c[3]) {
  c[0] = (a[1] * b[2]) - (b[1] * a[2]);
  c[1] = (a[2] * b[0]) - (b[2] * a[0]);
  c[2] = (a[0] * b[1]) - (b[0] * a[1]);
}
// angle_ab vectors.h:70
AHD void angle_ab(const double dir1[3], const double vel[3], double dir2[3]) {
  const double vsqr = vdot(vel, vel) / CLIGHTSQUARED;
  const double gamma_rel = 1. / sqrt(1 - vsqr);
  const double ndotv = vdot(dir1, vel);
  const double fact1 = gamma_rel * (1 - (ndotv / CLIGHT));
  const double fact2 = (gamma_rel - (pow2(gamma_rel) * ndotv / (gamma_rel + 1) / CLIGHT)) / CLIGHT;
  const double t[3] = {(dir1[0] - (vel[0] * fact2)) / fact1, (dir1[1] - (vel[1] * fact2)) / fact1,
                       (dir1[2] - (vel[2] * fact2)) / fact1};
  vnorm(t, dir2);
}
// calculate_doppler_nucmf_on_nurf vectors.h:92
AHD double doppler_at(double px, double py, double pz, double dx, double dy, double dz, double t) {
  const double vel[3] = {px / t, py / t, pz / t};
  const double dir[3] = {dx, dy, dz};
  const double ndotv = vdot(dir, vel);
  double dopplerfactor = 1. - (ndotv / CLIGHT);
#if ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT
  const double betasq = vdot(vel, vel) / CLIGHTSQUARED;
  dopplerfactor = dopplerfactor / sqrt(1 - betasq);
#endif
  return dopplerfactor;
}
AHD double doppler(const Pkt &p) { return doppler_at(p.px, p.py, p.pz, p.dx, p.dy, p.dz, p.prop_time); }
// move_pkt_withtime vectors.h:119
AHD void move_raw(double &px, double &py, double &pz, double dx, double dy, double dz, double &prop_time, double nu_rf,
                  double &nu_cmf, double e_rf, double &e_cmf, double distance) {
  const double nu_cmf_old = nu_cmf;
  prop_time += distance / CLIGHT_PROP;
  px = px + (dx * distance);
  py = py + (dy * distance);
  pz = pz + (dz * distance);
  const double d = doppler_at(px, py, pz, dx, dy, dz, prop_time);
  nu_cmf = dmin(nu_rf * d, nu_cmf_old);
  e_cmf = e_rf * d;
}
AHD void move_pkt(Pkt &p, double distance) {
  move_raw(p.px, p.py, p.pz, p.dx, p.dy, p.dz, p.prop_time, p.nu_rf, p.nu_cmf, p.e_rf, p.e_cmf, distance);
}
// set_pkt_restframe_from_cmf vectors.h:145
AHD void set_restframe_from_cmf(Pkt &p) {
  const double d = doppler(p);
  p.nu_rf = p.nu_cmf / d;
  p.e_rf = p.e_cmf / d;
}
// sin and cos of one angle. On the device ONE call with one argument reduction (round 6: every site of the packet path wants both; as two
// calls they were 1400 of k_rpkt's 13 900 instructions, run by a wave whenever one of its lanes scatters or emits). The host emulation keeps
// the two calls of the reference (vectors.h), whose bits the oracle has.
AHD void sin_cos(double x, double *s, double *c) {
#if defined(__HIP_DEVICE_COMPILE__)
  sincos(x, s, c);
#else
  *s = sin(x);
  *c = cos(x);
#endif
}
// get_rand_isotropic_unitvec vectors.h:185
AHD void rand_isotropic(Pkt &p, double out[3]) {
  const double u = rng_uniform(p);
  const double costheta = (2. * u) - 1.;
  const double sintheta = 2. * sqrt(u * (1. - u));
  const double phi = rng_uniform(p) * 2 * PI;
  double sphi, cphi;
  sin_cos(phi, &sphi, &cphi);
  out[0] = sintheta * cphi;
  out[1] = sintheta * sphi;
  out[2] = costheta;
}
// get_rot_angle vectors.h:196
AHD double rot_angle(const double n1[3], const double n2[3], const double ref1[3], const double ref2[3]) {
  const double c = vdot(n1, n2);
  const double u[3] = {(n1[0] * c) - n2[0], (n1[1] * c) - n2[1], (n1[2] * c) - n2[2]};
  const double len = vlen(u);
  if (len < 1e-12) return 0.0;
  const double r[3] = {u[0] / len, u[1] / len, u[2] / len};
  const double c1 = dclamp(vdot(r, ref1), -1., 1.);
  const double c2 = vdot(r, ref2);
  const double a = atan2(c2, c1);
  return a < 0 ? a + (2 * PI) : a;
}
// meridian vectors.h:219
AHD void meridian(const double dir[3], double ref1[3], double ref2[3]) {
  const double n_xylen = sqrt(pow2(dir[0]) + pow2(dir[1]));
  if (n_xylen == 0.) {
    ref1[0] = 1.; ref1[1] = 0.; ref1[2] = 0.;
    ref2[0] = 0.; ref2[1] = 1.; ref2[2] = 0.;
    return;
  }
  ref1[0] = -dir[0] * dir[2] / n_xylen;
  ref1[1] = -dir[1] * dir[2] / n_xylen;
  ref1[2] = (1 - pow2(dir[2])) / n_xylen;
  vcross(ref1, dir, ref2);
}
// lorentz vectors.h:233
AHD void lorentz(const double elec_rf[3], const double n_rf[3], const double v[3], double elec_cmf[3]) {
  const double beta[3] = {v[0] / CLIGHT, v[1] / CLIGHT, v[2] / CLIGHT};
  const double b2 = vdot(beta, beta);
  if (b2 == 0.) {
    elec_cmf[0] = elec_rf[0]; elec_cmf[1] = elec_rf[1]; elec_cmf[2] = elec_rf[2];
    return;
  }
  const double gamma_rel = 1. / sqrt(1 - b2);
  const double edb = vdot(elec_rf, beta);
  const double par[3] = {edb * beta[0] / b2, edb * beta[1] / b2, edb * beta[2] / b2};
  const double perp[3] = {elec_rf[0] - par[0], elec_rf[1] - par[1], elec_rf[2] - par[2]};
  double b_rf[3], vxb[3];
  vcross(n_rf, elec_rf, b_rf);
  vcross(beta, b_rf, vxb);
  const double t[3] = {par[0] + (gamma_rel * (perp[0] + vxb[0])), par[1] + (gamma_rel * (perp[1] + vxb[1])),
                       par[2] + (gamma_rel * (perp[2] + vxb[2]))};
  vnorm(t, elec_cmf);
}
// frame_transform vectors.h:266
ANOINLINE void frame_transform(const double n_rf[3], double q0, double u0, const double v[3], double n_cmf[3], double *q_cmf,
                               double *u_cmf) {
  double ref1_rf[3], ref2_rf[3];
  meridian(n_rf, ref1_rf, ref2_rf);
  const double pdeg = sqrt(pow2(q0) + pow2(u0));
  double ra = 0;
  if (pdeg > 0) {
    const double pol_angle = atan2(u0, q0);
    ra = (pol_angle < 0 ? pol_angle + (2. * PI) : pol_angle) / 2.;
  }
  double sr, cr;
  sin_cos(ra, &sr, &cr);
  const double elec_rf[3] = {(cr * ref1_rf[0]) - (sr * ref2_rf[0]), (cr * ref1_rf[1]) - (sr * ref2_rf[1]),
                             (cr * ref1_rf[2]) - (sr * ref2_rf[2])};
  angle_ab(n_rf, v, n_cmf);
  double elec_cmf[3];
  lorentz(elec_rf, n_rf, v, elec_cmf);
  double ref1_cmf[3], ref2_cmf[3];
  meridian(n_cmf, ref1_cmf, ref2_cmf);
  const double c1 = vdot(elec_cmf, ref1_cmf);
  const double c2 = vdot(elec_cmf, ref2_cmf);
  double theta = atan2(-c2, c1);
  if (theta < 0) theta += 2 * PI;
  double s2t, c2t;
  sin_cos(2 * theta, &s2t, &c2t);
  *q_cmf = c2t * pdeg;
  *u_cmf = s2t * pdeg;
}
// scatter_polarisation_to_rf vectors.h:325
ANOINLINE void scatter_polarisation_to_rf(const double old_dir_cmf[3], const double new_dir_cmf[3], double q_i, double u_i,
                                          const double vel[3], double new_dir_rf[3], double *q_rf, double *u_rf) {
  double r1o[3], r2o[3];
  meridian(old_dir_cmf, r1o, r2o);
  const double i1 = rot_angle(old_dir_cmf, new_dir_cmf, r1o, r2o);
  double sin2i1, cos2i1;
  sin_cos(2 * i1, &sin2i1, &cos2i1);
  const double q_old = (q_i * cos2i1) - (u_i * sin2i1);
  const double u_old = (q_i * sin2i1) + (u_i * cos2i1);
  const double mu = vdot(old_dir_cmf, new_dir_cmf);
  const double mu2 = pow2(mu);
  const double I_new = 0.75 * ((mu2 + 1.) + (q_old * (mu2 - 1.)));
  const double q_new = (0.75 * ((mu2 - 1.) + (q_old * (mu2 + 1.)))) / I_new;
  const double u_new = (1.5 * mu * u_old) / I_new;
  double r1[3], r2[3];
  meridian(new_dir_cmf, r1, r2);
  const double i2 = PI + rot_angle(new_dir_cmf, old_dir_cmf, r1, r2);
  double sin2i2, cos2i2;
  sin_cos(2 * i2, &sin2i2, &cos2i2);
  const double q_cmf = (q_new * cos2i2) + (u_new * sin2i2);
  const double u_cmf = (-q_new * sin2i2) + (u_new * cos2i2);
  const double nv[3] = {-vel[0], -vel[1], -vel[2]};
  frame_transform(new_dir_cmf, q_cmf, u_cmf, nv, new_dir_rf, q_rf, u_rf);
}

// ---------------------------------------------------------------- atomic.h accessors
AHD double statw(const DevModel &M, int ul) { return M.level_statweight[ul]; }
AHD double eps(const DevModel &M, int ul) { return M.level_epsilon[ul]; }
AHD int uion(const DevModel &M, int element, int ion) { return M.elem_uniqueionindexstart[element] + ion; }
AHD int lstart(const DevModel &M, int element, int ion) { return M.ion_uniquelevelindexstart[uion(M, element, ion)]; }
AHD int ionstage(const DevModel &M, int element, int ion) { return M.elem_lowest_ionstage[element] + ion; }
AHD int phixs_upperlevel(const DevModel &M, int ul, int t) { return M.allphixstargets_levelindex[M.level_phixstargetstart[ul] + t]; }
AHD double phixs_probability(const DevModel &M, int ul, int t) { return M.allphixstargets_probability[M.level_phixstargetstart[ul] + t]; }
AHD const float *phixs_table(const DevModel &M, int ul) { return M.allphixs + ((int64_t)M.level_phixsstart[ul] * M.NPHIXSPOINTS); }
AHD int find_phixstarget(const DevModel &M, int ul, int upperionlevel) {  // atomic.h:493
  const int n = M.level_nphixstargets[ul];
  for (int t = 0; t < n; t++)
    if (upperionlevel == phixs_upperlevel(M, ul, t)) return t;
  return -1;
}
AHD double phixs_threshold(const DevModel &M, int element, int ion, int level, int t) {  // atomic.h:534
  const int ul = lstart(M, element, ion) + level;
  return eps(M, lstart(M, element, ion + 1) + phixs_upperlevel(M, ul, t)) - eps(M, ul);
}
AHD int emtype_continuum(const DevModel &M, int ul, int t) { return -1 - M.level_bflist_start[ul] - t; }  // atomic.h:508
// photoionisation_crosssection_fromtable atomic.h:201
AHD float phixs_fromtable(const DevModel &M, const float *xs, double nu_edge, double nu) {
  const int NP = M.NPHIXSPOINTS;
  const double INC = M.NPHIXSNUINCREMENT;
#if ARTIS_OPT_PHIXS_CLASSIC_NO_INTERPOLATION
  if (nu < nu_edge) return 0.f;
  if (nu == nu_edge) return xs[0];
  if (nu < nu_edge * (1 + (INC * NP))) {
    int i = (int)((nu - nu_edge) / (INC * nu_edge));
    if (NP - 1 < i) i = NP - 1;
    return xs[i];
  }
  return (float)(xs[NP - 1] * pow(nu_edge * (1 + (INC * NP)) / nu, 3));
#else
  const double ireal = ((nu / nu_edge) - 1.0) / INC;
  const int i = (int)floor(ireal);
  if (i < 0) return 0.f;
  if (i < NP - 1) {
    const double a = xs[i];
    const double b = xs[i + 1];
    const double fb = ireal - i;
    return (float)(((1. - fb) * a) + (fb * b));
  }
  const double nu_max_phixs = nu_edge * M.last_phixs_nuovernuedge;
  return (float)(xs[NP - 1] * pow3(nu_max_phixs / nu));
#endif
}
// Partition point of a[0..n) for a predicate that is true on a prefix and false on the rest: the index of the first
// element for which it is false (n if none), i.e. what std::ranges::upper_bound / lower_bound / partition_point return.
// (An 8-ary variant with 7 independent probes per round was measured slower on MI355X: the kernels are bound by the
// number of memory instructions, not by their latency.)
template <class Pred>
AHD int partition_point_d(const double *a, int n, Pred pred) {
  int lo = 0, len = n;
  while (len > 0) {
    const int half = len / 2;
    if (pred(a[lo + half])) { lo += half + 1; len -= half + 1; } else { len = half; }
  }
  return lo;
}
// the same over an index range [0, n) with the predicate given the index
template <class Pred>
AHD int partition_point_f(int n, Pred pred) {
  int lo = 0, len = n;
  while (len > 0) {
    const int half = len / 2;
    if (pred(lo + half)) { lo += half + 1; len -= half + 1; } else { len = half; }
  }
  return lo;
}
// std::ranges::upper_bound / lower_bound on a rising double array
AHD int upper_bound_d(const double *a, int n, double v) {
  if (n <= 0) return 0;
  return partition_point_d(a, n, [v](double x) { return !(v < x); });
}
AHD int lower_bound_d(const double *a, int n, double v) {
  if (n <= 0) return 0;
  return partition_point_d(a, n, [v](double x) { return x < v; });
}
// upper_bound on a non-decreasing array as "count the elements <= v", 8 independent loads per round trip instead of
// one dependent load per bisection step (same result as upper_bound_d for any non-decreasing input)
AHD int upper_bound_wide(const double *a, int n, double v) {
  int idx = 0;
  for (int base = 0; base < n; base += 8) {
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = base + k;
      const double x = (i < n) ? a[i] : DBLMAX;
      cnt += (i < n && x <= v) ? 1 : 0;
    }
    idx += cnt;
    if (cnt < 8) break;
  }
  return idx;
}

// upper_bound on a non-decreasing array in two levels of independent reads: the last element of every block of S
// entries first (how many blocks lie entirely at or below v), then the S entries of the block the answer is in. Two
// (for long arrays a few) dependent read stages instead of log2(n): the k-packet step is a chain of such searches and
// runs at the latency of its dependent reads. Same result as upper_bound_d for any non-decreasing input.
template <int S>
AHD int upper_bound_blocked(const double *a, int n, double v) {
  if (n <= 0) return 0;
  const int nblocks = (n + S - 1) / S;
  int kb = 0;  // blocks whose last element is <= v: all their entries are <= v
  for (int b0 = 0; b0 < nblocks; b0 += 8) {
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int b = b0 + j;
      if (b < nblocks) {
        const int last = (b + 1) * S - 1;
        cnt += (a[last < n ? last : n - 1] <= v) ? 1 : 0;
      }
    }
    kb += cnt;
    if (cnt < 8) break;
  }
  if (kb >= nblocks) return n;
  const int start = kb * S;
  int idx = start;
#pragma unroll
  for (int j = 0; j < S; j++) {
    const int i = start + j;
    idx += (i < n && a[i] <= v) ? 1 : 0;
  }
  return idx;
}

// ---------------------------------------------------------------- grid.cc
AHD int coordidx(const DevModel &M, int cellindex, int axis) { return (cellindex / M.coordstride[axis]) % M.ncoordgrid[axis]; }
AHD double coordmin(const DevModel &M, int cellindex, int axis) { return M.coord_pos_min_tmin[axis][coordidx(M, cellindex, axis)]; }
AHD double coordmax(const DevModel &M, int cellindex, int axis) {
  const int idx = coordidx(M, cellindex, axis);
  return idx < M.ncoordgrid[axis] - 1 ? M.coord_pos_min_tmin[axis][idx + 1] : M.rmax;
}
AHD double cellbound_tol(double b) { return dmax(10., fabs(b) * 1e-12); }  // grid.cc:1530
AHD bool overshoot_in_tol(const DevModel &M, bool upper, double pos, double vel, double bpos_tmin, double tstart) {  // grid.cc:1542
  const double bvel = bpos_tmin / M.tmin;
  const double bpos = bvel * tstart;
  const double overshoot = upper ? (pos - bpos) : (bpos - pos);
  const bool towards = upper ? (vel > bvel) : (vel < bvel);
  return towards && (overshoot >= 0.) && (overshoot <= cellbound_tol(bpos));
}
AHD double dist_cart_boundary(const DevModel &M, double pos, double vel, double bpos, double tstart) {  // grid.cc:1518
  return CLIGHT_PROP * (pos - (bpos / M.tmin * tstart)) / ((bpos / M.tmin) - vel);
}
// expanding_shell_intersection grid.cc:1413 (3-vectors)
AHD double shell_intersection(bool lower, const double pos[3], const double dir[3], double speed, double rshell, double tstart) {
  const double a = vdot(dir, dir) - pow2(rshell / tstart / speed);
  const double b = 2 * (vdot(dir, pos) - (pow2(rshell) / tstart / speed));
  const double c = vdot(pos, pos) - pow2(rshell);
  const double disc = pow2(b) - (4 * a * c);
  if (disc < 0) return -1;
  if (disc > 0) {
    double d1 = (-b + sqrt(disc)) / 2 / a;
    double d2 = (-b - sqrt(disc)) / 2 / a;
    double pf1[3], pf2[3];
    for (int d = 0; d < 3; d++) {
      pf1[d] = pos[d] + (d1 * dir[d]);
      pf2[d] = pos[d] + (d2 * dir[d]);
    }
    const double v_shell = rshell / tstart;
    const double v1 = vdot(dir, pf1) * speed / vlen(pf1);
    const double v2 = vdot(dir, pf2) * speed / vlen(pf2);
    if (lower) {
      if (v1 > v_shell) d1 = -1;
      if (v2 > v_shell) d2 = -1;
    } else {
      if (v1 < v_shell) d1 = -1;
      if (v2 < v_shell) d2 = -1;
    }
    if (d1 < 0 && d2 < 0) return -1;
    if (d2 < 0) return d1;
    if (d1 < 0) return d2;
    return dmin(d1, d2);
  }
  return -1.;
}
// boundary_distance grid.cc:2480
AHD double boundary_distance(const Env &env, const Pkt &p, int *next_cell) {
  const DevModel &M = env.M;
  const double tstart = p.prop_time;
  const int cellindex = p.cellindex;
  double distance = DBLMAX;
  int next = -1;
  const double pos[3] = {p.px, p.py, p.pz};
  const double dir[3] = {p.dx, p.dy, p.dz};
  if (M.gridtype == ARTIS_GRID_CARTESIAN3D) {
    for (int d = 0; d < 3; d++) {
      const double vel = dir[d] * CLIGHT_PROP;
      const int idx = coordidx(M, cellindex, d);
      const double cmin = M.coord_pos_min_tmin[d][idx];
      const double cmax = idx < M.ncoordgrid[d] - 1 ? M.coord_pos_min_tmin[d][idx + 1] : M.rmax;
      if (vel > (cmax / M.tmin)) {
        const double dd = overshoot_in_tol(M, true, pos[d], vel, cmax, tstart) ? 0. : dist_cart_boundary(M, pos[d], vel, cmax, tstart);
        if ((dd >= 0.) && (dd < distance)) {
          distance = dd;
          next = (idx == (M.ncoordgrid[d] - 1)) ? -99 : cellindex + M.coordstride[d];
        }
      } else if (vel < (cmin / M.tmin)) {
        const double dd = overshoot_in_tol(M, false, pos[d], vel, cmin, tstart) ? 0. : dist_cart_boundary(M, pos[d], vel, cmin, tstart);
        if ((dd >= 0.) && (dd < distance)) {
          distance = dd;
          next = (idx == 0) ? -99 : cellindex - M.coordstride[d];
        }
      }
    }
  } else if (M.gridtype == ARTIS_GRID_SPHERICAL1D) {
    const double r = vlen(pos);
    const double vr = vdot(pos, dir) / r * CLIGHT_PROP;
    const int idx = coordidx(M, cellindex, 0);
    const double cmin = coordmin(M, cellindex, 0);
    const double cmax = coordmax(M, cellindex, 0);
    const double speed = vlen(dir) * CLIGHT_PROP;
    const double r_outer = cmax * tstart / M.tmin;
    const double dmaxb = overshoot_in_tol(M, true, r, vr, cmax, tstart) ? 0. : shell_intersection(false, pos, dir, speed, r_outer, tstart);
    if ((dmaxb >= 0.) && (dmaxb < distance)) {
      distance = dmaxb;
      next = (idx == (M.ncoordgrid[0] - 1)) ? -99 : cellindex + M.coordstride[0];
    }
    const double r_inner = cmin * tstart / M.tmin;
    if (r_inner > 0.) {
      const double dminb = overshoot_in_tol(M, false, r, vr, cmin, tstart) ? 0. : shell_intersection(true, pos, dir, speed, r_inner, tstart);
      if ((dminb >= 0.) && (dminb < distance)) {
        distance = dminb;
        next = (idx == 0) ? -99 : cellindex - M.coordstride[0];
      }
    }
  } else if (M.gridtype == ARTIS_GRID_CYLINDRICAL2D) {
    // grid.cc:2602-2695: coordinate 0 is the cylindrical radius, coordinate 1 is z. The reference's 2-vectors are
    // passed as 3-vectors with a zero third component, which changes no sum of shell_intersection().
    const double posnoz[3] = {pos[0], pos[1], 0.};
    const double rc = sqrt(pow2(pos[0]) + pow2(pos[1]));                                    // grid.cc:1377
    const double vrc = ((pos[0] * dir[0]) + (pos[1] * dir[1])) / rc * CLIGHT_PROP;          // grid.cc:1394
    const int idx0 = coordidx(M, cellindex, 0);
    const double cmin0 = coordmin(M, cellindex, 0);
    const double cmax0 = coordmax(M, cellindex, 0);
    const double dirxylen = sqrt(pow2(dir[0]) + pow2(dir[1]));
    const double xyspeed = dirxylen * CLIGHT_PROP;
    if (dirxylen > 0.) {
      const double dirnoz[3] = {dir[0] / dirxylen, dir[1] / dirxylen, 0.};
      const double r_outer = cmax0 * tstart / M.tmin;
      const double d_rcyl_max =
          overshoot_in_tol(M, true, rc, vrc, cmax0, tstart) ? 0. : shell_intersection(false, posnoz, dirnoz, xyspeed, r_outer, tstart);
      if (d_rcyl_max >= 0.) {
        const double d_z = d_rcyl_max / xyspeed * dir[2] * CLIGHT_PROP;
        const double dd = sqrt(pow2(d_rcyl_max) + pow2(d_z));
        if ((dd >= 0.) && (dd < distance)) {
          distance = dd;
          next = (idx0 == (M.ncoordgrid[0] - 1)) ? -99 : cellindex + M.coordstride[0];
        }
      }
      const double r_inner = cmin0 * tstart / M.tmin;
      if (r_inner > 0) {
        const double d_rcyl_min =
            overshoot_in_tol(M, false, rc, vrc, cmin0, tstart) ? 0. : shell_intersection(true, posnoz, dirnoz, xyspeed, r_inner, tstart);
        if (d_rcyl_min >= 0.) {
          const double d_z = d_rcyl_min / xyspeed * dir[2] * CLIGHT_PROP;
          const double dd = sqrt(pow2(d_rcyl_min) + pow2(d_z));
          if ((dd >= 0.) && (dd < distance)) {
            distance = dd;
            next = (idx0 == 0) ? -99 : cellindex - M.coordstride[0];
          }
        }
      }
    } else if (cmin0 > 0.) {
      // moving exactly along z: only the expanding inner r_cyl boundary can catch up with the packet (grid.cc:2654)
      const double dd = overshoot_in_tol(M, false, rc, vrc, cmin0, tstart) ? 0. : ((rc * M.tmin / cmin0) - tstart) * CLIGHT_PROP;
      if ((dd >= 0.) && (dd < distance)) {
        distance = dd;
        next = (idx0 == 0) ? -99 : cellindex - M.coordstride[0];
      }
    }
    {  // z boundaries are Cartesian (grid.cc:2671)
      const int d = 1;
      const double vel = dir[2] * CLIGHT_PROP;
      const int idx = coordidx(M, cellindex, d);
      const double cmin = coordmin(M, cellindex, d);
      const double cmax = coordmax(M, cellindex, d);
      if (vel > (cmax / M.tmin)) {
        const double dd = overshoot_in_tol(M, true, pos[2], vel, cmax, tstart) ? 0. : dist_cart_boundary(M, pos[2], vel, cmax, tstart);
        if ((dd >= 0.) && (dd < distance)) {
          distance = dd;
          next = (idx == (M.ncoordgrid[d] - 1)) ? -99 : cellindex + M.coordstride[d];
        }
      } else if (vel < (cmin / M.tmin)) {
        const double dd = overshoot_in_tol(M, false, pos[2], vel, cmin, tstart) ? 0. : dist_cart_boundary(M, pos[2], vel, cmin, tstart);
        if ((dd >= 0.) && (dd < distance)) {
          distance = dd;
          next = (idx == 0) ? -99 : cellindex - M.coordstride[d];
        }
      }
    }
  } else {
    fail(env, 10);
  }
  if (!((next == -99) || ((next >= 0) && (next < M.ngrid))) || !(distance >= 0.)) fail(env, 11);
  if (distance > env.S.max_path_step) {
    *next_cell = cellindex;
    return env.S.max_path_step;
  }
  *next_cell = next;
  return distance;
}
// change_cell_or_escape grid.h:118 with snap_pos_to_cell grid.cc:2460
AHD void change_cell_or_escape(const Env &env, Pkt &p, int64_t pi, int next_cell) {
  const DevModel &M = env.M;
  if (next_cell >= 0) {
    if (next_cell != p.cellindex && M.gridtype == ARTIS_GRID_CARTESIAN3D) {
      double *pos[3] = {&p.px, &p.py, &p.pz};
      for (int d = 0; d < 3; d++) {
        const int idx = coordidx(M, next_cell, d);
        const double lo = M.coord_pos_min_tmin[d][idx] / M.tmin * p.prop_time;
        const double hi = (idx < (M.ncoordgrid[d] - 1)) ? M.coord_pos_min_tmin[d][idx + 1] / M.tmin * p.prop_time
                                                         : M.rmax / M.tmin * p.prop_time;
        *pos[d] = dclamp(*pos[d], lo, hi);
      }
    }
    p.cellindex = next_cell;
    ARTIS_STAT(env, ARTIS_STAT_CELLCROSSINGS);
  } else {
    env.P.cold[pi].escape_type = p.type;
    env.P.cold[pi].escape_time = (float)p.prop_time;
    p.type = ARTIS_TYPE_ESCAPE;
    ARTIS_STAT(env, ARTIS_STAT_PKTESCAPES);
  }
}

// ---------------------------------------------------------------- cell state
AHD float clumpednne(const DevCells &C, int c) { return C.clumpfactor[c] * C.nne[c]; }
AHD double groundlevelpop(const Env &env, int c, int element, int ion) {  // ltepop.h:74
  const double nn = env.C.ion_groundlevelpops[((int64_t)c * env.M.nions) + uion(env.M, element, ion)];
  if (nn < ARTIS_OPT_MINPOP) {
    if (env.C.elem_massfracs[((int64_t)c * env.M.nelements) + element] > 0) return ARTIS_OPT_MINPOP;
    return 0.;
  }
  return nn;
}
AHD double nnion(const Env &env, int c, int element, int ion) {  // ltepop.h:106
  return groundlevelpop(env, c, element, ion) * env.C.ion_partfuncts[((int64_t)c * env.M.nions) + uion(env.M, element, ion)] /
         statw(env.M, lstart(env.M, element, ion));
}
AHD double planck(double nu, double T) { return 2 * HPLANCK * pow3(nu) / pow2(CLIGHT) / expm1(HOVERKB * nu / T); }  // radfield.h:50
// multibin radiation field model: bins radfield.cc:118-161 (RADFIELDBINCOUNT - 1 bins of equal width from NU_MIN to
// NU_MAX, then the "T_e superbin" up to SUPERBIN_NU_MAX)
constexpr double RADBIN_DELTA_NU = (ARTIS_OPT_RADFIELDBINS_NU_MAX - ARTIS_OPT_RADFIELDBINS_NU_MIN) / (ARTIS_OPT_RADFIELDBINCOUNT - 1);  // radfield.cc:73
AHD double radbin_nu_upper(int b) {  // get_bin_nu_upper radfield.cc:118
  if (b == ARTIS_OPT_RADFIELDBINCOUNT - 1) return ARTIS_OPT_RADFIELDBINS_T_E_SUPERBIN_NU_MAX;
  return ARTIS_OPT_RADFIELDBINS_NU_MIN + ((b + 1) * RADBIN_DELTA_NU);
}
AHD int radbin_select(double nu) {  // select_bin radfield.cc:138 (get_linearbinindex sn3d.h:115)
  if (nu < ARTIS_OPT_RADFIELDBINS_NU_MIN) return -2;
  if (nu >= ARTIS_OPT_RADFIELDBINS_T_E_SUPERBIN_NU_MAX) return -1;
  if (nu >= ARTIS_OPT_RADFIELDBINS_NU_MAX) return ARTIS_OPT_RADFIELDBINCOUNT - 1;
  const double fracindex = (nu - ARTIS_OPT_RADFIELDBINS_NU_MIN) / RADBIN_DELTA_NU;
  const int64_t truncated = (int64_t)fracindex;
  const int b = (int)((fracindex < (double)truncated) ? truncated - 1 : truncated);
  if (nu == radbin_nu_upper(b)) return b + 1;
  return b;
}
AHD double radfield(const Env &env, double nu, int c) {  // radfield.cc:786
#if ARTIS_OPT_MULTIBIN_RADFIELD_MODEL_ON
  if (env.S.nts >= ARTIS_OPT_FIRST_NLTE_RADFIELD_TIMESTEP) {
    const int b = radbin_select(nu);
    if (b >= 0) {
      const float W = env.C.radfieldbin_W[((int64_t)c * ARTIS_OPT_RADFIELDBINCOUNT) + b];
      if (W >= 0.) return W * planck(nu, env.C.radfieldbin_T_R[((int64_t)c * ARTIS_OPT_RADFIELDBINCOUNT) + b]);
    }
    return 0.;
  }
#endif
  return env.C.W[c] * planck(nu, env.C.TR[c]);
}

// ---------------------------------------------------------------- ratecoeff.cc LUTs
AHD int temperature_upperindex(const DevModel &M, double T) {  // ratecoeff.cc:54
  const int gridsize = ARTIS_OPT_TABLESIZE + 1;
  int index = (int)(log(T / ARTIS_OPT_MINTEMP) / M.T_step_log) + 1;
  if (index < 0) index = 0;
  if (index > gridsize) index = gridsize;
  while (index > 0 && M.temperature_grid[index - 1] > T) index--;
  while (index < gridsize && M.temperature_grid[index] <= T) index++;
  return index;
}
AHD double lerp_or_last(const DevModel &M, const double *table, int ul, int t, float T) {  // ratecoeff.cc:524
  const double *row = table + ((int64_t)(M.level_bflist_start[ul] + t) * ARTIS_OPT_TABLESIZE);
  const int up = temperature_upperindex(M, T);
  if (up == 0) return row[0];
  if (up < ARTIS_OPT_TABLESIZE) {
    const double T_lower = M.temperature_grid[up - 1];
    const double T_upper = M.temperature_grid[up];
    const double f_lower = row[up - 1];
    const double f_upper = row[up];
    return (f_lower + ((f_upper - f_lower) / (T_upper - T_lower) * (T - T_lower)));
  }
  return row[ARTIS_OPT_TABLESIZE - 1];
}

// ---------------------------------------------------------------- macroatom.cc rate coefficients
AHD double gaunt_factor(int stage) { return stage == 1 ? 0.1 : (stage == 2 ? 0.2 : 0.3); }  // macroatom.cc:327
AHD double rad_deexc(double epsilon_trans, float A_ul, double gu, double gl, double n_u, double n_l, double t) {  // macroatom.h:61
  const double nu_trans = epsilon_trans / HPLANCK;
  const double B_ul = CLIGHTSQUAREDOVERTWOH / pow3(nu_trans) * A_ul;
  const double B_lu = gu / gl * B_ul;
  const double tau = ((B_lu * n_l) - (B_ul * n_u)) * HCLIGHTOVERFOURPI * t;
  if (tau > 1e-100) {
    const double beta = 1.0 / tau * (-expm1(-tau));
    return A_ul * beta;
  }
  return A_ul;
}
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
// radfield::get_Jblueindex radfield.cc:695: place of the line among the detailed lines, -1 if it has no estimator
AHD int jblueindex(const DevModel &M, int lineindex) {
  int low = 0, high = M.detailed_linecount - 1;
  while (low <= high) {
    const int mid = low + ((high - low) / 2);
    if (M.detailed_lineindices[mid] < lineindex) low = mid + 1;
    else if (M.detailed_lineindices[mid] > lineindex) high = mid - 1;
    else return mid;
  }
  return -1;
}
// radfield::update_lineestimator radfield.cc:773
AHD void update_lineestimator(const Env &env, int c, int lineindex, double increment) {
  const int jb = jblueindex(env.M, lineindex);
  if (jb >= 0) {
    const int64_t o = ((int64_t)c * env.M.detailed_linecount) + jb;
    ARTIS_EST_ADD(&env.E.Jb_lu_raw[o], increment);
    ARTIS_EST_ADD(&env.E.Jb_lu_contribcount[o], 1.);
  }
}
#endif
// alltransindex: only read by builds with detailed line estimators (macroatom.cc:628; globals::lte_iteration is false while
// packets propagate)
AHD double rad_exc(const Env &env, int c, double gu, double A, double epsilon_trans, double n_l, double n_u, double gl, double t,
                   int alltransindex = -1) {  // macroatom.cc:611
  const double nu_trans = epsilon_trans / HPLANCK;
  const double B_ul = CLIGHTSQUAREDOVERTWOH / pow3(nu_trans) * A;
  const double B_lu = gu / gl * B_ul;
  const double tau = ((B_lu * n_l) - (B_ul * n_u)) * HCLIGHTOVERFOURPI * t;
  if (tau > 1e-100) {
    const double beta = 1.0 / tau * (-expm1(-tau));
    const double R_over_J = n_l > 0. ? (B_lu - (B_ul * n_u / n_l)) * beta : B_lu * beta;
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
    if (alltransindex >= 0) {
      const int jb = jblueindex(env.M, env.M.alltrans_lineindex[alltransindex]);
      if (jb >= 0) return R_over_J * env.C.Jb_lu_normed[((int64_t)c * env.M.detailed_linecount) + jb];  // get_Jb_lu radfield.cc:718
    }
#endif
    return R_over_J * radfield(env, nu_trans, c);
  }
  return 0.;
}
AHD double rad_recomb(const DevModel &M, float T_e, float cnne, int element, int upperion, int lowerlevel, int t) {  // macroatom.cc:646
  return cnne * lerp_or_last(M, M.spontrecombcoeffs, lstart(M, element, upperion - 1) + lowerlevel, t, T_e);
}
AHD double col_recomb(const DevModel &M, float T_e, float cnne, int element, int upperion, int lower, int t, double epsilon_trans) {  // macroatom.cc:660
  const int ul = lstart(M, element, upperion - 1) + lower;
  const double gl = statw(M, ul);
  const double g = gaunt_factor(ionstage(M, element, upperion - 1));
  const double sigma_bf = (phixs_table(M, ul)[0] * phixs_probability(M, ul, t));
  const double gu = statw(M, lstart(M, element, upperion) + phixs_upperlevel(M, ul, t));
  return cnne * cnne * SAHACONST * gl / gu * 1.55e13 * g * sigma_bf * KB / T_e / epsilon_trans;
}
AHD double col_ion(const DevModel &M, float T_e, float cnne, int element, int ion, int lower, int t, double epsilon_trans) {  // macroatom.cc:686
  const int ul = lstart(M, element, ion) + lower;
  const double g = gaunt_factor(ionstage(M, element, ion));
  const double fac1 = epsilon_trans / KB / T_e;
  const double sigma_bf = phixs_table(M, ul)[0] * phixs_probability(M, ul, t);
  return cnne * 1.55e13 * pow((double)T_e, -0.5) * g * sigma_bf * exp(-fac1) / fac1;
}
AHD double col_deexc(const DevModel &M, float T_e, float cnne, double epsilon_trans, double gu, double gl, int ati) {  // macroatom.cc:708
  const float cs = M.alltrans_coll_str[ati];
  if (cs < 0) {
    if (!M.alltrans_forbidden[ati]) {
      const double f = M.alltrans_osc_strength[ati];
      const double eoverkt = epsilon_trans / (KB * T_e);
      const double g_bar = 0.2;
      const double gauntfac = (eoverkt > 0.33421) ? g_bar : 0.276 * exp(eoverkt) * (-EULERGAMMA - log(eoverkt));
      const double g_ratio = gl / gu;
      return C_0 * 14.51039491 * cnne * sqrtf(T_e) * f * pow2(H_ionpot / epsilon_trans) * eoverkt * g_ratio * gauntfac;
    }
    return cnne * 8.629e-6 * 0.01 * gl / sqrtf(T_e);
  }
  return cnne * 8.629e-6 * (double)cs / gu / sqrtf(T_e);
}
AHD double col_exc(const DevModel &M, float T_e, float cnne, double epsilon_trans, double gu, double gl, int ati) {  // macroatom.cc:750
  const float cs = M.alltrans_coll_str[ati];
  const double eoverkt = epsilon_trans / (KB * T_e);
  if (cs < 0) {
    if (!M.alltrans_forbidden[ati]) {
      const double f = M.alltrans_osc_strength[ati];
      const double g_bar = 0.2;
      const double ex = exp(eoverkt);
      const double Gamma = dmax(g_bar, 0.276 * ex * (-EULERGAMMA - log(eoverkt)));
      return C_0 * cnne * sqrtf(T_e) * 14.51039491 * f * pow2(H_ionpot / epsilon_trans) * eoverkt / ex * Gamma;
    }
    return cnne * 8.629e-6 * 0.01 * exp(-eoverkt) * gu / sqrtf(T_e);
  }
  return cnne * 8.629e-6 * (double)cs * exp(-eoverkt) / gl / sqrtf(T_e);
}

// ================================================================ cell-cache population
// (cellcacheslot_populate update_packets.cc:397, multi-slot form; one function per kernel)

// one (cell, level): calculate_levelpop ltepop.cc:412 / calculate_levelpop_boltzmann ltepop.cc:395
AHD void populate_levelpop(const Env &env, int c, int ul) {
  const DevModel &M = env.M;
  if (env.C.levelpops) {  // the host's NLTE / LTE solution (get_levelpop ltepop.cc:169)
    env.K.levelpops[(krow(env, c) * M.nlevels) + ul] = env.C.levelpops[((int64_t)c * M.nlevels) + ul];
    return;
  }
  const int ui = M.level_ion[ul];
  const int element = M.ion_element[ui];
  const int ion = ui - M.elem_uniqueionindexstart[element];
  const int start = M.ion_uniquelevelindexstart[ui];
  const double nnground = groundlevelpop(env, c, element, ion);
  double nn;
  if (ul == start) {
    nn = nnground;
  } else {
    const float T_exc = ARTIS_OPT_LTEPOP_EXCITATION_USE_TJ ? env.C.TJ[c] : env.C.Te[c];
    const double E_aboveground = eps(M, ul) - eps(M, start);
    nn = (nnground * statw(M, ul) / statw(M, start) * exp(-E_aboveground / KB / T_exc));
  }
  if (nn < ARTIS_OPT_MINPOP) nn = (env.C.elem_massfracs[((int64_t)c * M.nelements) + element] > 0) ? ARTIS_OPT_MINPOP : 0.;
  env.K.levelpops[(krow(env, c) * M.nlevels) + ul] = nn;
}
// one (cell, line): the level-population factor of get_tau_sobolev<true>() (rpkt.cc:75), evaluated once per timestep so
// that the line walk reads one value per line instead of the line record and two level populations
AHD void populate_line_dpop(const Env &env, int c, int li) {
  const DevModel &M = env.M;
  const double *pops = env.K.levelpops + (krow(env, c) * M.nlevels);
  const LinePack lp = M.line_pack[li];
  const double n_l = pops[lp.lower];
  const double n_u = pops[lp.upper];
  const double B_ul = lp.B_ul;
  const double B_lu = lp.B_lu;
  env.K.line_dpop[(krow(env, c) * M.nlines) + li] = (B_lu * n_l) - (B_ul * n_u);
}
// ... and the reader: the stored value, or -- when the engine keeps no line_dpop rows (atomic data too large for them: 8 bytes per line
// and cell are then a quarter of the cell cache; artis_engine.hip engine_fill) -- the same expression from the line record and the two
// level populations, the same bits
struct LineDpop {
  const double *dpop;  // the cell's row of line_dpop, or null
  const double *pops;  // the cell's level populations
};
AHD LineDpop line_dpop_of(const Env &env, int c) {
  LineDpop r;
  r.dpop = env.K.line_dpop ? env.K.line_dpop + (krow(env, c) * env.M.nlines) : nullptr;
  r.pops = env.K.levelpops + (krow(env, c) * env.M.nlevels);
  return r;
}
AHD double line_dpop_at(const DevModel &M, const LineDpop &d, int li) {
  if (d.dpop) return d.dpop[li];
  const LinePack lp = M.line_pack[li];
  const double B_ul = lp.B_ul, B_lu = lp.B_lu;
  return (B_lu * d.pops[lp.lower]) - (B_ul * d.pops[lp.upper]);
}
// one cell: calculate_chi_ffheat_nnionpart rpkt.cc:932
AHD void populate_chi_ff(const Env &env, int c) {
  const DevModel &M = env.M;
  double s = 0.;
  for (int element = 0; element < M.nelements; element++) {
    const int nions = M.elem_nions[element];
    for (int ion = 0; ion < nions; ion++) {
      const double n = nnion(env, c, element, ion);
      const int ioncharge = ionstage(M, element, ion) - 1;
      s += pow2(ioncharge) * 1. * n;
    }
  }
  const float T_e = env.C.Te[c];
  env.K.chi_ff_nnionpart[krow(env, c)] = s * 3.69255e8 / sqrt((double)T_e);
}
// one (cell, continuum): update_packets.cc:430-440 + the slow path of rpkt.cc:853-889. Returns the keep bit.
AHD bool populate_allcont(const Env &env, int c, int i) {
  const DevModel &M = env.M;
  const double *pops = env.K.levelpops + (krow(env, c) * M.nlevels);
  const int64_t o = (krow(env, c) * M.nbfcontinua) + i;  // (cache row)
  const double nnlevel = pops[M.allcont_uniquelevelindex[i]];
  const int element = M.allcont_element[i];
  const int ion = M.allcont_ion[i];
  const int level = M.allcont_level[i];
  const float nnetot = env.C.nnetot[c];
  const bool keep = nnlevel > 0 && ((nnion(env, c, element, ion) / nnetot > 1.e-6) || (level == 0));  // keep_this_cont rpkt.h:189
  env.K.allcont_nnlevel[o] = nnlevel;
  double dep = -1., edge = -1.;
  if (keep) {
    const float T_e = env.C.Te[c];
    const float cnne = env.C.nne[c] * env.C.clumpfactor[c];
    const double sahapart = SAHACONST * pow((double)T_e, -1.5);
    const int upper = M.allcont_upperlevel[i];
    const double nnupper = pops[lstart(M, element, ion + 1) + upper];
    const double sahafact = sahapart * statw(M, lstart(M, element, ion) + level) / statw(M, lstart(M, element, ion + 1) + upper);
    dep = nnupper / nnlevel * cnne * sahafact;
    const double edge_exponent = HOVERKB * M.allcont_nu_edge[i] / T_e;
    if (edge_exponent < 690.) {
      const double e = dep * exp(edge_exponent);
      if (isfinite(e)) edge = e;
    }
  }
  env.K.allcont_departure[o] = dep;
  env.K.allcont_edgepart[o] = edge;
  env.K.allcont_pair[o] = D2{nnlevel, edge};
  return keep;
}
// after every populate_allcont() of the cell and its keep bitmap: the kept continua as a list, the count of kept continua
// below each bitmap word, and their {nnlevel, edge part} pairs in list order (on the GPU: k_keptlist, a wave per cell)
AHD void populate_keptlist(const Env &env, int c) {
  const DevModel &M = env.M;
  const int nw = M.nkeepwords;
  const uint64_t *keep = env.K.allcont_keepbits + (krow(env, c) * nw);
  int32_t *list = env.K.allcont_keptlist + (krow(env, c) * M.nbfcontinua);
  int32_t *prefix = env.K.allcont_keepprefix + (krow(env, c) * nw);
  const D2 *pair = env.K.allcont_pair + (krow(env, c) * M.nbfcontinua);
  D2 *keptpair = env.K.allcont_keptpair + (krow(env, c) * M.nbfcontinua);
  int at = 0;
  for (int j = 0; j < nw; j++) {
    prefix[j] = at;
    uint64_t word = keep[j];
    while (word != 0) {
      const int i = (j * 64) + __builtin_ctzll(word);
      list[at] = i;
      keptpair[at] = pair[i];
      at++;
      word &= word - 1;
    }
  }
}
// the places [r0, r1) of the continua [begin, end) in the cell's list of kept continua (begin < end)
AHD void kept_range(const Env &env, int c, int begin, int end, int &r0, int &r1) {
  const int nw = env.M.nkeepwords;
  const uint64_t *keep = env.K.allcont_keepbits + (krow(env, c) * nw);
  const int32_t *prefix = env.K.allcont_keepprefix + (krow(env, c) * nw);
  const int wfirst = begin / 64, wlast = (end - 1) / 64;
  r0 = prefix[wfirst] + __builtin_popcountll(keep[wfirst] & ~(~UINT64_C(0) << (unsigned)(begin % 64)));
  r1 = prefix[wlast] + __builtin_popcountll(keep[wlast] & (~UINT64_C(0) >> (unsigned)(63 - ((end - 1) % 64))));
}
// one (cell, phixs target): get_corrphotoioncoeff ratecoeff.cc:840 (USE_LUT_PHOTOION)
// ... and the other rate coefficients of the same bound-free pair (level ul of an ion -> target t in the next ion), which
// the per-level and per-ion stages below only have to combine: rad_recomb / col_recomb (macroatom.cc:646, :660),
// col_ion (macroatom.cc:686) and the bound-free cooling coefficient (kpkt.cc:165). Each is the value the
// reference's loop computes for this pair, evaluated once here instead of inside a loop over levels that only a few
// lanes of a wave would run.
AHD void populate_corrphotoion(const Env &env, int c, int ul, int t) {
  const DevModel &M = env.M;
  const double W = env.C.W[c];
  const double T_R = env.C.TR[c];
  const int64_t o = (krow(env, c) * M.nphixstargets_total) + M.level_phixstargetstart[ul] + t;  // (cache row; the host's arrays are indexed by cell)
#if ARTIS_OPT_USE_LUT_PHOTOION
  double g = W * lerp_or_last(M, M.corrphotoioncoeffs, ul, t, (float)T_R);
  const int ig = M.level_closestgroundlevelcont[ul];
  if (ig >= 0) g *= env.C.corrphotoionrenorm[((int64_t)c * M.nbfcontinua_ground) + ig];
#else
  // the estimator-based / integrated coefficient of get_corrphotoioncoeff() (ratecoeff.cc:840) comes from the host
  (void)W;
  (void)T_R;
  const double g = env.C.corrphotoioncoeff[((int64_t)c * M.nphixstargets_total) + M.level_phixstargetstart[ul] + t];
#endif
  env.K.corrphotoioncoeff[o] = g;
  const int ui = M.level_ion[ul];
  const int element = M.ion_element[ui];
  const int ion = ui - M.elem_uniqueionindexstart[element];
  const int level = ul - M.ion_uniquelevelindexstart[ui];
  const float T_e = env.C.Te[c];
  const float cnne = clumpednne(env.C, c);
  const double e_trans = phixs_threshold(M, element, ion, level, t);  // epsilon(upper) - epsilon(ul)
  env.K.bf_radrecomb[o] = rad_recomb(M, T_e, cnne, element, ion + 1, level, t);
  env.K.bf_colrecomb[o] = col_recomb(M, T_e, cnne, element, ion + 1, level, t, e_trans);
  env.K.bf_colion[o] = col_ion(M, T_e, cnne, element, ion, level, t, e_trans);
  env.K.bf_cooling[o] = lerp_or_last(M, M.bfcooling_coeffs, ul, t, T_e);
}
#if ARTIS_OPT_NT_ON || ARTIS_OPT_USE_XCOM_GAMMAPHOTOION
AHD double elem_numberdens(const DevModel &M, const DevCells &C, int c, int element) {  // grid.cc:1693
#if ARTIS_OPT_USE_CALCULATED_MEANATOMICWEIGHT  // grid::get_element_meanweight grid.cc:1509: the cell's own mean weight of the element
  const float mu = C.elem_meanweight[((int64_t)c * M.nelements) + element];
#else
  const float mu = M.elem_meannucmass[element];
#endif
  return C.elem_massfracs[((int64_t)c * M.nelements) + element] / (double)mu * C.rho[c];
}
#endif
#if ARTIS_OPT_NT_ON
// ---------------------------------------------------------------- non-thermal channels (nonthermal.cc)
// The Spencer-Fano solution comes from the host (DevCells nt_*); the packet path only reads it.
constexpr int NT_NAUGER = ARTIS_OPT_NT_MAX_AUGER_ELECTRONS + 1;
constexpr double QE = 4.80325E-10;  // constants.h:31
AHD int nt_maxupperion(const DevModel &M, int element, int lowerion) {  // nt_ionisation_maxupperion nonthermal.cc:2435
  const int nions = M.elem_nions[element];
  const int maxupper = lowerion + 1 + ARTIS_OPT_NT_MAX_AUGER_ELECTRONS;
  return (nions - 1 < maxupper) ? nions - 1 : maxupper;
}
// nt_ionisation_upperion_probability nonthermal.cc:2398; *bad is set where the reference asserts
AHD double nt_upperion_probability(const DevModel &M, const DevCells &C, int c, int element, int lowerion, int upperion,
                                   bool energyweighted, bool *bad) {
  const int numaugerelec = upperion - lowerion - 1;
  const float *prob = (energyweighted ? C.nt_ionenfrac_num_auger : C.nt_prob_num_auger) +
                      ((((int64_t)c * M.nions) + uion(M, element, lowerion)) * NT_NAUGER);
  if (numaugerelec < ARTIS_OPT_NT_MAX_AUGER_ELECTRONS) return prob[numaugerelec];
  if (numaugerelec == ARTIS_OPT_NT_MAX_AUGER_ELECTRONS) {
    double prob_remaining = 1.;
    for (int a = 0; a < ARTIS_OPT_NT_MAX_AUGER_ELECTRONS; a++) prob_remaining -= prob[a];
    if (!(fabs(prob_remaining - prob[numaugerelec]) < 0.001)) *bad = true;
    return prob_remaining;
  }
  return 0.;
}
AHD int nt_random_upperion(const Env &env, Pkt &p, int c, int element, int lowerion, bool energyweighted) {  // nonthermal.cc:2450
  const double zrand = rng_uniform(p);
  double prob_sum = 0.;
  bool bad = false;
  const int maxupper = nt_maxupperion(env.M, element, lowerion);
  for (int upperion = lowerion + 1; upperion <= maxupper; upperion++) {
    prob_sum += nt_upperion_probability(env.M, env.C, c, element, lowerion, upperion, energyweighted, &bad);
    if (bad) fail(env, 91);
    if (zrand < prob_sum) return upperion;
  }
  if (!(prob_sum > 0.99)) fail(env, 92);
  return maxupper;
}
// one cell: nt_ionisation_ratecoeff() (nonthermal.cc:2478, with _sf :1420 and the Axelrod fallback _wfapprox :1251,
// get_oneoverw_approx_axelrod :1207) of every ion that has a higher stage, and the running sum of ion_ntion_energyrate()
// (:1509) in the order select_nt_ionisation() (:1537) adds them (an ion without a higher stage repeats the sum so far).
// Returns false where the reference would assert (Auger probabilities that do not sum to one).
AHD bool populate_nt_cell(const Env &env, int c) {
  const DevModel &M = env.M;
  const DevCells &C = env.C;
  double nntot = 0., Zbar = 0.;
  for (int e = 0; e < M.nelements; e++) {  // get_nnion_tot atomic.h:51 and the mean atomic number of :1214
    const double nnelement = elem_numberdens(M, C, c, e);
    Zbar += nnelement * M.elem_anumber[e];
    nntot += nnelement;
  }
  if (nntot > 0) Zbar /= nntot;
  const double dep = C.nt_deposition_rate_density[c];
  double ratesum = 0.;
  bool bad = false;
  for (int e = 0; e < M.nelements; e++) {
    const int nions = M.elem_nions[e];
    for (int ion = 0; ion < nions; ion++) {
      const int64_t o = ((int64_t)c * M.nions) + uion(M, e, ion);
      double Y_nt = 0.;
      if (ion < nions - 1) {
        if (dep > 0.) Y_nt = dep / nntot / C.nt_eff_ionpot[o];
        if (!isfinite(Y_nt)) {
          constexpr double Aconst = 1.33e-14 * EV * EV;
          const double oneoverw = Aconst * M.ion_nt_sum_q_over_binding[uion(M, e, ion)] / Zbar / (2 * PI * (pow2(QE) * pow2(QE)));
          Y_nt = dep / nntot * oneoverw;
        }
        const double nnlowerion = nnion(env, c, e, ion);
        double enrate = 0.;
        const int maxupper = nt_maxupperion(M, e, ion);
        for (int upperion = ion + 1; upperion <= maxupper; upperion++) {
          const double frac = nt_upperion_probability(M, C, c, e, ion, upperion, false, &bad);
          const double e_trans = eps(M, lstart(M, e, upperion)) - eps(M, lstart(M, e, ion));
          enrate += nnlowerion * frac * e_trans;
        }
        ratesum += Y_nt * enrate;
      }
      C.nt_ionratecoeff[o] = Y_nt;
      C.nt_ionenrate_cum[o] = ratesum;
    }
  }
  return !bad;
}
// nt_excitation_ratecoeff nonthermal.cc:2496 (lowerlevel, upperlevel: indices within the ion)
AHD double nt_excitation_ratecoeff(const DevCells &C, int c, int lowerlevel, int upperlevel, int alltransindex) {
  if (!ARTIS_OPT_NT_EXCITATION_ON) return 0.;
  if (lowerlevel >= ARTIS_OPT_NTEXCITATION_MAXNLEVELS_LOWER) return 0.;
  if (upperlevel >= ARTIS_OPT_NTEXCITATION_MAXNLEVELS_UPPER) return 0.;
  const int64_t base = (int64_t)c * C.nt_excitations_stored;
  const int32_t *ati = C.nt_exc_alltransindex + base;
  const int n = C.nt_exc_count[c];
  int lo = 0, hi = n;  // std::ranges::lower_bound
  while (lo < hi) {
    const int mid = lo + ((hi - lo) / 2);
    if (ati[mid] < alltransindex) lo = mid + 1; else hi = mid;
  }
  if (lo == n || ati[lo] != alltransindex) return 0.;
  return C.nt_exc_ratecoeffperdeposition[base + lo] * C.nt_deposition_rate_density[c];
}
#endif

// one (cell, entry of alltrans): the rate coefficients of ONE bound-bound transition of
// calculate_macroatom_transitionrates() (macroatom.cc:64-140) and its term of calculate_cooling_rates_ion()
// (kpkt.cc:108-121), written where the per-level / per-ion stages below turn them into running sums. Splitting the work
// this way keeps every lane busy (levels have 0..60 transitions each) and evaluates each collisional-excitation
// coefficient once instead of twice. The later stages form the same products and add them in the same order: same bits.
// the terms one entry of alltrans adds to the running sums of its level: for a downward transition {R e_trans, C e_trans,
// (R + C) e_target}, for an upward one {(R + C + NT) e_cur, -, -} plus its k-packet cooling term n C e_trans
struct MaTransTerms {
  int ul, i;        // owner level; index within its down (isdown) or up block
  bool isdown;
  LevelPack lpk;
  double v0, v1, v2;
  double kterm;
};
// COOLING_ONLY_IF_COLD: the population's pass over a transition of a COLD level (tables.h "ON-DEMAND RECORDS": no static record to fill) needs
// nothing of a downward transition and of an upward one only its collisional-excitation cooling term (the same expression: the same bits).
template <bool COOLING_ONLY_IF_COLD = false>
AHD MaTransTerms matrans_terms(const Env &env, int c, int ati) {
  const DevModel &M = env.M;
  MaTransTerms r;
  r.ul = M.alltrans_owner[ati];
  r.lpk = M.level_pack[r.ul];
  const int ul = r.ul;
  if (COOLING_ONLY_IF_COLD && r.lpk.rec_off < 0) {
    const int i = ati - r.lpk.alltrans_startdown;
    r.isdown = i < r.lpk.ndown;
    r.i = r.isdown ? i : i - r.lpk.ndown;
    r.v0 = r.v1 = r.v2 = r.kterm = 0.;
    if (!r.isdown) {
      const int tul = M.ion_uniquelevelindexstart[M.level_ion[ul]] + M.alltrans_targetlevelindex[ati];
      const double e_trans = eps(M, tul) - eps(M, ul);
      const double Cc = col_exc(M, env.C.Te[c], clumpednne(env.C, c), e_trans, statw(M, tul), statw(M, ul), ati);
      r.kterm = env.K.levelpops[(krow(env, c) * M.nlevels) + ul] * Cc * e_trans;
    }
    return r;
  }
  const int start = M.ion_uniquelevelindexstart[M.level_ion[ul]];
  const double *pops = env.K.levelpops + (krow(env, c) * M.nlevels);
  const float T_e = env.C.Te[c];
  const float cnne = clumpednne(env.C, c);
  const double e_cur = eps(M, ul);
  const double g_cur = statw(M, ul);
  const double nnlevel = pops[ul];
  const int i = ati - r.lpk.alltrans_startdown;
  const int tul = start + M.alltrans_targetlevelindex[ati];
  r.isdown = i < r.lpk.ndown;
  r.kterm = 0.;
  if (r.isdown) {
    r.i = i;
    const float A_ul = M.alltrans_einstein_A[ati];
    const double e_target = eps(M, tul);
    const double e_trans = e_cur - e_target;
    const double g_low = statw(M, tul);
    const double R = rad_deexc(e_trans, A_ul, g_cur, g_low, nnlevel, pops[tul], env.S.mid);
    const double Cc = col_deexc(M, T_e, cnne, e_trans, g_cur, g_low, ati);
    r.v0 = R * e_trans;
    r.v1 = Cc * e_trans;
    r.v2 = (R + Cc) * e_target;
  } else {
    r.i = i - r.lpk.ndown;
    const double e_trans = eps(M, tul) - e_cur;
    const double g_up = statw(M, tul);
    const double R = rad_exc(env, c, g_up, M.alltrans_einstein_A[ati], e_trans, nnlevel, pops[tul], g_cur, env.S.mid, ati);
    const double Cc = col_exc(M, T_e, cnne, e_trans, g_up, g_cur, ati);
#if ARTIS_OPT_NT_ON
    const double NT = nt_excitation_ratecoeff(env.C, c, ul - start, M.alltrans_targetlevelindex[ati], ati);  // macroatom.cc:133
#else
    const double NT = 0.;
#endif
    r.v0 = (R + Cc + NT) * e_cur;
    r.v1 = 0.;
    r.v2 = 0.;
    r.kterm = nnlevel * Cc * e_trans;
  }
  return r;
}
// where things are in a level's record (tables.h): the record of level ul in cell c, its nine process rates
// (tables.h "ON-DEMAND RECORDS") rec_off >= 0: the level's static slot in every row. rec_off < 0: a cold level, whose record -- if a packet
// has reached the level in this cell -- lies in the cell's pool where ma_rowtab says. ma_resolve(): the slot for a WALKER (-1 unless the
// record is complete); ma_rec_of(): the record for the code that fills or reads it knowingly (also while it is being filled).
AHD int32_t ma_rowtab_load(const int32_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (k_tail: another wave may be filling the record right now)
#else
  return *p;
#endif
}
// A READY place (v >= 0) seen in a kernel whose waves fill records (Env::ma_concurrent_fill): the record's plain loads that follow must not be
// served from lines this compute unit or this XCD's L2 fetched before the filler's release (ma_slow_fill_publish / ma_ensure_record) -- an
// acquire fence at agent scope after the relaxed load, i.e. the acquire half of the pair. The pool's 128-byte units (tables.h
// MAPOOL_UNIT: no record shares a vector-L1 line with another) stay as what keeps such stale lines from existing in the first place; this
// makes the ordering hold whatever the line sizes are. The fence invalidates the CU's L1, so only the kernels that need it pay it (cold levels
// only: the headline's records are all static).
AHD void ma_rowtab_acquire(const Env &env, int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (env.ma_concurrent_fill != 0 && v >= 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
  (void)env;
  (void)v;
#endif
}
// MACtx::rec: >= 0 the record's slot in the cell's row; MA_REC_NONE a cold level that has no record in its cell yet; <= -2 a cold level's
// record in the pool, at unit -(rec + 2). COLD = false (a kernel built for models without cold levels, the default): rec_off as it is.
constexpr int MA_REC_NONE = -1;
template <bool COLD = true>
AHD int ma_resolve(const Env &env, int c, int rec_off) {
  if (!COLD) return rec_off;
  if (__builtin_expect(rec_off >= 0, 1)) return rec_off;
  const int32_t v = ma_rowtab_load(env.K.ma_rowtab + (krow(env, c) * env.M.ncold) + (-rec_off - 1));
  ma_rowtab_acquire(env, v);
  return v >= 0 ? -(v + 2) : MA_REC_NONE;
}
AHD U4 *ma_rec_of(const Env &env, int c, const LevelPack &lpk) {
  if (lpk.rec_off >= 0) return env.K.macache + (krow(env, c) * env.M.nmacache) + lpk.rec_off;
  const int32_t v = ma_rowtab_load(env.K.ma_rowtab + (krow(env, c) * env.M.ncold) + (-lpk.rec_off - 1));
  ma_rowtab_acquire(env, v);
  return env.K.ma_pool + ((int64_t)((v >= 0) ? v : -(v + 3)) * MAPOOL_UNIT);  // (ready, or being filled by the caller; never asked for a level without a record)
}
AHD double *ma_rates_of(U4 *rec, int nd, int nu) { return (double *)(rec + marec_rates_slot(nd, nu)); }
AHD const double *ma_rates_of(const U4 *rec, int nd, int nu) { return (const double *)(rec + marec_rates_slot(nd, nu)); }
AHD uint32_t mafilt_quant(double value, double whole, bool *ok);
AHD uint32_t mafilt_quant23(double value, double whole, bool *ok);
// entry i of a filter line (0..6; 7 = the line's "usable" mark)
AHD void mafilt_put(U4 *line, int i, uint32_t q) { ((uint16_t *)line)[i] = (uint16_t)q; }
// ... with its fine byte (tables.h "FINE BYTES"; internal-down / internal-up filters only): q23 = the entry's 23-bit fraction
AHD void mafilt_put23(U4 *rec, int dir, int ti, int nd, int nu, uint32_t q23) {
  mafilt_put(rec + marec_slot(dir, ti / MAREC_PER, nd, nu), ti % MAREC_PER, q23 >> 8);
  ((uint8_t *)rec)[marec_fine_byte0(dir, ti / MAREC_PER, nd, nu) + (ti % MAREC_PER)] = (uint8_t)(q23 & 0xFFu);
}
constexpr uint32_t MAFILT_NONE23 = 0x7FFFFFu;  // never counted: 0x7FFF in the line, 0xFF in its fine byte
// The term of one transition that a direction's cumulative sums add up (macroatom.cc:64-140):
// MADIR_DOWN (R + C) e_target, MADIR_RAD R e_trans (both of a downward transition), MADIR_UP (R + C + NT) e_cur
AHD double matrans_term_of(const MaTransTerms &t, int dir) { return dir == MADIR_DOWN ? t.v2 : t.v0; }
// The filter entries of one direction of one level from its terms, sequential form: the running sums of the reference's loop as
// 15-bit fractions of their last value. `whole` = that last value (the direction's rate in the record). The GPU forms the same
// sums in k_matrans (a wave scan in this order: same bits) and quantises them with the same mafilt_quant().
// check != nullptr: nothing is written; *check counts the entries and marks of the record that differ from these.
AHD void populate_dirfilter_seq(const Env &env, int c, const LevelPack &lpk, int dir, double whole, int *check = nullptr) {
  const bool down = dir != MADIR_UP;
  const int n = down ? lpk.ndown : lpk.nup;
  if (n <= 0) return;
  U4 *rec = ma_rec_of(env, c, lpk);
  const int ats0 = lpk.alltrans_startdown + (down ? 0 : lpk.ndown);
  bool whole_ok = (whole > 0.) && (whole <= DBLMAX);
  double s = 0.;
  for (int l = 0; l < marec_lines(n); l++) {
    U4 *line = rec + marec_slot(dir, l, lpk.ndown, lpk.nup);
    bool ok = whole_ok;
    uint32_t q[MAREC_PER];  // 23-bit fractions: the line's entry is q >> 8, its fine byte q & 0xFF (tables.h "FINE BYTES")
    const int cnt = (n - (l * MAREC_PER) < MAREC_PER) ? n - (l * MAREC_PER) : MAREC_PER;
    for (int j = 0; j < cnt; j++) {
      const int i = (l * MAREC_PER) + j;
      s += matrans_term_of(matrans_terms(env, c, ats0 + i), dir);
      q[j] = (ok && i < n - 1) ? mafilt_quant23(s, whole, &ok) : MAFILT_NONE23;  // (the last sum is the whole: never searched)
    }
    // the line's mark: anything but 0x7FFF = decide this line on the f64 sums
    const bool fine = dir != MADIR_RAD;
    uint8_t *fb = (uint8_t *)rec + marec_fine_byte0(fine ? dir : MADIR_DOWN, l, lpk.ndown, lpk.nup);
    if (check) {
      const uint16_t *have = (const uint16_t *)line;
      for (int j = 0; j < cnt; j++) *check += (have[j] != (uint16_t)(ok ? q[j] >> 8 : 0u)) ? 1 : 0;
      for (int j = cnt; j < MAREC_PER; j++) *check += (have[j] != (uint16_t)MAFILT_NONE) ? 1 : 0;
      *check += (have[7] != (uint16_t)(ok ? MAFILT_NONE : 0u)) ? 1 : 0;
      if (fine) {
        for (int j = 0; j < cnt; j++) *check += (fb[j] != (uint8_t)(ok ? q[j] & 0xFFu : 0u)) ? 1 : 0;
        for (int j = cnt; j < MAREC_PER; j++) *check += (fb[j] != (uint8_t)0xFFu) ? 1 : 0;
      }
    } else {
      for (int j = 0; j < cnt; j++) mafilt_put(line, j, ok ? q[j] >> 8 : 0u);
      mafilt_put(line, 7, ok ? MAFILT_NONE : 0u);
      if (fine)
        for (int j = 0; j < cnt; j++) fb[j] = (uint8_t)(ok ? q[j] & 0xFFu : 0u);
    }
  }
}
// one (cell, level), sequential form (test emulation; k_mafilter_long's reference): the four bound-bound rates of the record
// (macroatom.cc:64-140), the filters of its three directions, and the level's collisional-excitation cooling terms
// n C e_trans (kpkt.cc:108-121) into the cell's row of upward-transition terms. The GPU: k_matrans.
// upterms == nullptr: the record alone (a cold level filled on demand; its cooling terms went into the cell's list at population).
// ONDEMAND false: the population's pass over every level -- a cold level (no static record) contributes its cooling terms only.
template <bool ONDEMAND = false>
AHD void populate_level_bb(const Env &env, int c, int ul, double *upterms) {
  const DevModel &M = env.M;
  const LevelPack lpk = M.level_pack[ul];
  if (!ONDEMAND && lpk.rec_off < 0) {
    for (int i = 0; i < lpk.nup; i++) upterms[M.level_upcum_start[ul] + i] = matrans_terms(env, c, lpk.alltrans_startdown + lpk.ndown + i).kterm;
    return;
  }
  U4 *rec = ma_rec_of(env, c, lpk);
  double *rates = ma_rates_of(rec, lpk.ndown, lpk.nup);
  double s_raddeexc = 0., s_coldeexc = 0., s_down_same = 0., s_up_same = 0.;
  for (int i = 0; i < lpk.ndown; i++) {
    const MaTransTerms t = matrans_terms(env, c, lpk.alltrans_startdown + i);
    s_raddeexc += t.v0;
    s_coldeexc += t.v1;
    s_down_same += t.v2;
  }
  for (int i = 0; i < lpk.nup; i++) {
    const MaTransTerms t = matrans_terms(env, c, lpk.alltrans_startdown + lpk.ndown + i);
    s_up_same += t.v0;
    if (upterms != nullptr) upterms[M.level_upcum_start[ul] + i] = t.kterm;
  }
  rates[ARTIS_MA_ACTION_RADDEEXC] = s_raddeexc;
  rates[ARTIS_MA_ACTION_COLDEEXC] = s_coldeexc;
  rates[ARTIS_MA_ACTION_INTERNALDOWNSAME] = s_down_same;
  rates[ARTIS_MA_ACTION_INTERNALUPSAME] = s_up_same;
  populate_dirfilter_seq(env, c, lpk, MADIR_DOWN, s_down_same);
  populate_dirfilter_seq(env, c, lpk, MADIR_RAD, s_raddeexc);
  populate_dirfilter_seq(env, c, lpk, MADIR_UP, s_up_same);
}
// the static part of a level's record, written once per resident row (k_mainit): every filter entry "never counted", every
// line usable; the population then writes the entries of the transitions and the marks
AHD void populate_mainit_at(U4 *rec, const LevelPack &lpk);
AHD void populate_mainit(const Env &env, int64_t row, int ul) {
  const LevelPack lpk = env.M.level_pack[ul];
  if (lpk.rec_off < 0) return;  // (a cold level: initialised with the rest of its record, on demand)
  populate_mainit_at(env.K.macache + (row * env.M.nmacache) + lpk.rec_off, lpk);
}
AHD void populate_mainit_at(U4 *rec, const LevelPack &lpk) {
  const int nfilt = marec_rates_slot(lpk.ndown, lpk.nup);
  const uint32_t none2 = MAFILT_NONE | (MAFILT_NONE << 16);
  for (int i = 0; i < nfilt; i++) rec[i] = U4{{none2, none2, none2, none2}};
  const int nfine0 = marec_fine_slot0(lpk.ndown, lpk.nup), nrec = marec_slots(lpk.ndown, lpk.nup);
  const int ntot = ((nrec + MAREC_ALIGN - 1) / MAREC_ALIGN) * MAREC_ALIGN;
  for (int i = nfilt; i < ntot; i++) rec[i] = (i >= nfine0 && i < nrec) ? U4{{~0u, ~0u, ~0u, ~0u}} : U4{{0u, 0u, 0u, 0u}};  // (fine bytes: "never counted")
}
// Test / debug view of one level's record (artis_amd_debug_cellcache): the nine process rates as the record holds them, and
// the level's block of allmacroatomictransitions (globals.h:287; block order of input.cc:1542: radiative de-excitation sums,
// internal-down-same sums, internal-up-same sums) RE-ADDED from the transitions' terms -- the values a draw gets that the
// filters cannot decide (ma_exact_search). Returns the number of filter entries and marks of the record that differ from the
// sequential form's (populate_dirfilter_seq): 0 unless the population's scans and the sequential loop disagree.
AHD int debug_level_record(const Env &env, int c, int ul, double *maprocessrates, double *matrans) {
  const DevModel &M = env.M;
  const LevelPack lpk = M.level_pack[ul];
  const double *rates = ma_rates_of(ma_rec_of(env, c, lpk), lpk.ndown, lpk.nup);
  if (maprocessrates)
    for (int a = 0; a < MA_N; a++) maprocessrates[((int64_t)ul * MA_N) + a] = rates[a];
  if (matrans) {
    double *blk = matrans + M.level_matransblock_start[ul];
    double s_rad = 0., s_down = 0., s_up = 0.;
    for (int i = 0; i < lpk.ndown; i++) {
      const MaTransTerms t = matrans_terms(env, c, lpk.alltrans_startdown + i);
      s_rad += t.v0;
      s_down += t.v2;
      blk[i] = s_rad;
      blk[lpk.ndown + i] = s_down;
    }
    for (int i = 0; i < lpk.nup; i++) {
      s_up += matrans_terms(env, c, lpk.alltrans_startdown + lpk.ndown + i).v0;
      blk[(2 * lpk.ndown) + i] = s_up;
    }
  }
  int bad = 0;
  populate_dirfilter_seq(env, c, lpk, MADIR_DOWN, rates[ARTIS_MA_ACTION_INTERNALDOWNSAME], &bad);
  populate_dirfilter_seq(env, c, lpk, MADIR_RAD, rates[ARTIS_MA_ACTION_RADDEEXC], &bad);
  populate_dirfilter_seq(env, c, lpk, MADIR_UP, rates[ARTIS_MA_ACTION_INTERNALUPSAME], &bad);
  return bad;
}
AHD void populate_mafilter_level(const Env &env, int c, int ul);  // (below, with the filters)
// one (cell, level): the bound-free channels of calculate_macroatom_transitionrates macroatom.cc:141-190 (the four
// bound-bound rates are already in the record: populate_level_bb() / k_matrans), then the record's action filter
// RECOMB_DONE: the three sums over the level's recombination list are in the record already (k_macroatom_recomb formed them with a
// row of lanes, the additions in this loop's order)
template <bool RECOMB_DONE = false>
AHD void populate_macroatom(const Env &env, int c, int ul) {
  const DevModel &M = env.M;
  const int ui = M.level_ion[ul];
  const int element = M.ion_element[ui];
  const int ion = ui - M.elem_uniqueionindexstart[element];
  const int start = M.ion_uniquelevelindexstart[ui];
  const int level = ul - start;
  const double *pops = env.K.levelpops + (krow(env, c) * M.nlevels);
  const LevelPack lpk = M.level_pack[ul];
  double *rates = ma_rates_of(ma_rec_of(env, c, lpk), lpk.ndown, lpk.nup);
  const double t_mid = env.S.mid;
  const float T_e = env.C.Te[c];
  const float cnne = clumpednne(env.C, c);
  const double e_cur = eps(M, ul);
  const double g_cur = statw(M, ul);
  const double nnlevel = pops[ul];

  // a level without downward (upward) transitions has no segment that would have written its bound-bound rates
  if (lpk.ndown == 0) rates[ARTIS_MA_ACTION_RADDEEXC] = rates[ARTIS_MA_ACTION_COLDEEXC] = rates[ARTIS_MA_ACTION_INTERNALDOWNSAME] = 0.;
  if (lpk.nup == 0) rates[ARTIS_MA_ACTION_INTERNALUPSAME] = 0.;
  double s_down_lower = 0., s_radrecomb = 0., s_colrecomb = 0.;
  if (RECOMB_DONE) {
    if (ion > 0 && level <= M.ion_maxrecombininglevel[ui] && M.level_recomb_start[ul + 1] > M.level_recomb_start[ul]) {
      s_down_lower = rates[ARTIS_MA_ACTION_INTERNALDOWNLOWER];
      s_radrecomb = rates[ARTIS_MA_ACTION_RADRECOMB];
      s_colrecomb = rates[ARTIS_MA_ACTION_COLRECOMB];
    }
  } else if (ion > 0 && level <= M.ion_maxrecombininglevel[ui]) {
    const int ls = M.ion_uniquelevelindexstart[ui - 1];
    const int64_t cb = krow(env, c) * M.nphixstargets_total;
    // the levels of the ion below that ionise into this one (find_phixstargetindex() >= 0), from the static list
    for (int r = M.level_recomb_start[ul]; r < M.level_recomb_start[ul + 1]; r++) {
      const int lower = M.recomb_lower[r];
      const int t = M.recomb_target[r];
      const double e_target = eps(M, ls + lower);
      const double e_trans = e_cur - e_target;
      const int64_t o = cb + M.level_phixstargetstart[ls + lower] + t;
      const double R = env.K.bf_radrecomb[o];   // rad_recomb_ratecoeff / col_recomb_ratecoeff of the pair
      const double Cc = env.K.bf_colrecomb[o];  // (populate_corrphotoion)
      s_down_lower += (R + Cc) * e_target;
      s_radrecomb += R * e_trans;
      s_colrecomb += Cc * e_trans;
    }
  }
  rates[ARTIS_MA_ACTION_INTERNALDOWNLOWER] = s_down_lower;
  rates[ARTIS_MA_ACTION_RADRECOMB] = s_radrecomb;
  rates[ARTIS_MA_ACTION_COLRECOMB] = s_colrecomb;

  double s_up_higher = 0.;
  if (ion < M.elem_nions[element] - 1 && level < M.ion_nlevels_ionising[ui]) {
    const int nt = M.level_nphixstargets[ul];
    const int64_t o = (krow(env, c) * M.nphixstargets_total) + M.level_phixstargetstart[ul];
    for (int t = 0; t < nt; t++) {
      const double R = env.K.corrphotoioncoeff[o + t];
      const double Cc = env.K.bf_colion[o + t];  // col_ionization_ratecoeff of the pair (populate_corrphotoion)
      s_up_higher += (R + Cc) * e_cur;
    }
  }
  double s_up_highernt = 0.;
#if ARTIS_OPT_NT_ON
  if (ion < M.elem_nions[element] - 1 && level < M.ion_nlevels_ionising[ui])
    s_up_highernt = env.C.nt_ionratecoeff[((int64_t)c * M.nions) + ui] * e_cur;  // macroatom.cc:181
#endif
  rates[ARTIS_MA_ACTION_INTERNALUPHIGHERNT] = s_up_highernt;
  rates[ARTIS_MA_ACTION_INTERNALUPHIGHER] = s_up_higher;
  populate_mafilter_level(env, c, ul);  // every rate of the record is final now: the action filter of its line 0
}
// one (cell, ion): calculate_cooling_rates_ion<true> kpkt.cc:57 in three parts, so that the GPU can form the long middle
// part -- the running sum over the ion's collisional-excitation terms -- with rows of 16 lanes (k_cooling_chain) while the
// test emulation runs it as the plain loop; same additions in the same order either way.
// head: the free-free term (kpkt.cc:75-86). Returns the running sum and the number of list entries written so far.
AHD double cooling_ion_head(const Env &env, int c, int ui, int *k_out) {
  const DevModel &M = env.M;
  const int element = M.ion_element[ui];
  const int ion = ui - M.elem_uniqueionindexstart[element];
  double *contribs = env.K.cooling_contrib + (krow(env, c) * M.ncoolingterms) + M.ion_coolingoffset[ui];
  const float cnne = clumpednne(env.C, c);
  const float T_e = env.C.Te[c];
  double C_ion = 0.;
  int k = 0;
  const double nncurrention = nnion(env, c, element, ion);
  const int ioncharge = ionstage(M, element, ion) - 1;
  if (ioncharge > 0) {
    const double C_ff_ion = 1.426e-27 * sqrt((double)T_e) * pow2(ioncharge) * nncurrention * cnne;
    C_ion += C_ff_ion;
    contribs[k++] = C_ion;
  }
  *k_out = k;
  return C_ion;
}
// middle, sequential form: the running sum over the collisional-excitation terms of the ion's levels (the terms
// nnlevel * C * e_trans were left in `upterms`, the cell's row of upward-transition terms, by populate_level_bb() / k_matrans),
// turned into running sums in place; one list entry per level that has upward transitions (kpkt.cc:108-121). do_kpkt()
// (kpkt.cc:461-476) re-adds exactly these terms, in this order and from the same starting value, to pick the transition: the
// running sums themselves are not kept (round 4) -- populate_coolfilter_line() turns them into the level's 15-bit filter and a
// draw the filter cannot decide re-adds the terms (kpkt_collexc_exact)
AHD double cooling_ion_collexc_chain(const Env &env, int c, int ui, double C_ion, int *k_inout, double *upterms) {
  const DevModel &M = env.M;
  double *contribs = env.K.cooling_contrib + (krow(env, c) * M.ncoolingterms) + M.ion_coolingoffset[ui];
  int k = *k_inout;
  const int start = M.ion_uniquelevelindexstart[ui];
  const int nlevels = M.ion_nlevels[ui];
  for (int level = 0; level < nlevels; level++) {
    const int ul = start + level;
    const int nup = M.level_nuptrans[ul];
    double *upcum = upterms + M.level_upcum_start[ul];
    for (int i = 0; i < nup; i++) {
      C_ion += upcum[i];
      upcum[i] = C_ion;
    }
    if (nup > 0) contribs[k++] = C_ion;
  }
  *k_inout = k;
  return C_ion;
}
// One line of a level's cooling filter (M.coollines[li]) from the running sums in `upcum` (the cell's row, after the chain):
// entry j = the running sum after the level's transition j as a 15-bit fraction of the level's span [lo, hi] of the ion's
// cooling list, F = (sum - lo) / (hi - lo). do_kpkt() compares the sums with rnd_process, lo <= rnd_process < hi: "sum <= rnd"
// is "F <= y", y = (rnd - lo) / (hi - lo) formed the same way, each quotient within 3e-16 of its exact value (the
// differences of doubles are exact to one rounding of the RESULT). The last sum is hi itself: never counted.
AHD void populate_coolfilter_line(const Env &env, int c, int li, const double *upcum) {
  const DevModel &M = env.M;
  const CoolLineRef lr = M.coollines[li];
  if (lr.n <= 0) return;
  const double *cool = env.K.cooling_contrib + (krow(env, c) * M.ncoolingterms);
  const double hi = cool[lr.cool_hi], lo = (lr.cool_lo >= 0) ? cool[lr.cool_lo] : 0.;
  const double span = hi - lo;
  bool ok = (span > 0.) && (span <= DBLMAX);
  uint32_t q[8];
  for (int j = 0; j < 8; j++) q[j] = MAFILT_NONE;
  for (int j = 0; j < MAREC_PER; j++)
    if (lr.first + j < lr.n - 1 && ok) q[j] = mafilt_quant(upcum[lr.up0 + lr.first + j] - lo, span, &ok);
  if (!ok)
    for (int j = 0; j < 8; j++) q[j] = 0u;
  U4 f;
  for (int j = 0; j < 4; j++) f.w[j] = q[2 * j] | (q[2 * j + 1] << 16);
  env.K.macache[(krow(env, c) * M.nmacache) + lr.slot] = f;
}
// ... all lines of ONE level's cooling filter, the running sums re-added from the level's own terms starting at the list's value before
// the level (cooling_ion_collexc_chain()'s additions in its order: the bits the chain left in the population's scratch). For a cold level's
// record filled on demand (the scratch is gone by then).
AHD void populate_coolfilter_level_seq(const Env &env, int c, int ul) {
  const DevModel &M = env.M;
  const LevelPack lpk = M.level_pack[ul];
  const int hi_i = M.level_coolhi[ul];
  if (lpk.nup <= 0 || hi_i < 0) return;
  const int ui = M.level_ion[ul];
  const double *cool = env.K.cooling_contrib + (krow(env, c) * M.ncoolingterms);
  const double hi = cool[hi_i], lo = (hi_i > M.ion_coolingoffset[ui]) ? cool[hi_i - 1] : 0.;
  const double span = hi - lo;
  U4 *rec = ma_rec_of(env, c, lpk);
  double s = lo;
  for (int l = 0; l < marec_lines(lpk.nup); l++) {
    bool ok = (span > 0.) && (span <= DBLMAX);
    uint32_t q[8];
    for (int j = 0; j < 8; j++) q[j] = MAFILT_NONE;
    for (int j = 0; j < MAREC_PER; j++) {
      const int i = (l * MAREC_PER) + j;
      if (i >= lpk.nup) break;
      s += matrans_terms(env, c, lpk.alltrans_startdown + lpk.ndown + i).kterm;
      if (i < lpk.nup - 1 && ok) q[j] = mafilt_quant(s - lo, span, &ok);
    }
    if (!ok)
      for (int j = 0; j < 8; j++) q[j] = 0u;
    U4 f;
    for (int j = 0; j < 4; j++) f.w[j] = q[2 * j] | (q[2 * j + 1] << 16);
    rec[marec_slot(MADIR_COOL, l, lpk.ndown, lpk.nup)] = f;
  }
}
// tail: collisional ionisation and bound-free cooling (kpkt.cc:123-190), then the ion total for the prefix sum of kpkt.cc:281
AHD void cooling_ion_tail(const Env &env, int c, int ui, double C_ion, int k) {
  const DevModel &M = env.M;
  const int element = M.ion_element[ui];
  const int ion = ui - M.elem_uniqueionindexstart[element];
  const double *pops = env.K.levelpops + (krow(env, c) * M.nlevels);
  double *contribs = env.K.cooling_contrib + (krow(env, c) * M.ncoolingterms) + M.ion_coolingoffset[ui];
  const float cnne = clumpednne(env.C, c);
  const float T_e = env.C.Te[c];
  const int nionising = M.ion_nlevels_ionising[ui];
  const int start = M.ion_uniquelevelindexstart[ui];
  if (ion < (M.elem_nions[element] - 1) && M.nbfcontinua > 0) {
    const double nnupperion = nnion(env, c, element, ion + 1);
    const int ustart = M.ion_uniquelevelindexstart[ui + 1];
    for (int level = 0; level < nionising; level++) {
      const int ul = start + level;
      const double e_cur = eps(M, ul);
      const double nnlevel = pops[ul];
      const int nt = M.level_nphixstargets[ul];
      const int64_t o = (krow(env, c) * M.nphixstargets_total) + M.level_phixstargetstart[ul];
      for (int t = 0; t < nt; t++) {
        const double e_trans = eps(M, ustart + phixs_upperlevel(M, ul, t)) - e_cur;
        const double Cc = nnlevel * env.K.bf_colion[o + t] * e_trans;  // col_ionization_ratecoeff: populate_corrphotoion
        C_ion += Cc;
        contribs[k++] = C_ion;
      }
    }
    for (int level = 0; level < nionising; level++) {
      const int ul = start + level;
      const int nt = M.level_nphixstargets[ul];
      double wsum = 0.;
      double E_min = 0.;
#if !ARTIS_OPT_BFCOOLING_USELEVELPOPNOTIONPOP
      if (nt > 1) {
        E_min = DBLMAX;
        for (int t = 0; t < nt; t++) E_min = dmin(E_min, eps(M, ustart + phixs_upperlevel(M, ul, t)));
        for (int t = 0; t < nt; t++) {
          const int up = phixs_upperlevel(M, ul, t);
          wsum += statw(M, ustart + up) * exp(-(eps(M, ustart + up) - E_min) / KB / T_e);
        }
      }
#endif
      for (int t = 0; t < nt; t++) {
        double pop;
#if ARTIS_OPT_BFCOOLING_USELEVELPOPNOTIONPOP
        pop = pops[ustart + phixs_upperlevel(M, ul, t)];
#else
        if (nt == 1) {
          pop = nnupperion;
        } else {
          const int up = phixs_upperlevel(M, ul, t);
          const double w = statw(M, ustart + up) * exp(-(eps(M, ustart + up) - E_min) / KB / T_e);
          pop = nnupperion * w / wsum;
        }
#endif
        const double Cc = env.K.bf_cooling[(krow(env, c) * M.nphixstargets_total) + M.level_phixstargetstart[ul] + t] * pop * cnne;
        C_ion += Cc;
        contribs[k++] = C_ion;
      }
    }
  }
  if (k != M.ion_ncoolingterms[ui]) fail(env, 20);
  env.K.ion_cooling_C[(krow(env, c) * M.nions) + ui] = C_ion;
}
AHD void populate_cooling_ion(const Env &env, int c, int ui, double *upterms) {
  int k = 0;
  double C_ion = cooling_ion_head(env, c, ui, &k);
  C_ion = cooling_ion_collexc_chain(env, c, ui, C_ion, &k, upterms);
  cooling_ion_tail(env, c, ui, C_ion, k);
}
// one cell: cumulative cooling over ions, kpkt.cc:288-294
AHD void populate_cooling_prefix(const Env &env, int c) {
  double cum = 0.;
  for (int ui = 0; ui < env.M.nions; ui++) {
    cum += env.K.ion_cooling_C[(krow(env, c) * env.M.nions) + ui];
    env.K.ion_cooling_contribs[(krow(env, c) * env.M.nions) + ui] = cum;
  }
}

// The cooling guides (tables.h "COOLING GUIDES"). Entry k of the guide of a cumulative list of n sums whose draws fall into 2^(24 - shift)
// ranges: the bisection's answer for the range's first draw; the list's length after the last range; 0 in a row's padding
AHD uint16_t cool_guide_entry(const double *list, int n, int shift, int k) {
  const int nranges = 1 << (24 - shift);
  if (k > nranges || n <= 0) return 0;
  if (k == nranges) return (uint16_t)n;
  // the value the draw u = k << shift is compared with in do_kpkt(): the same expression
  const double b = rng_u24_value((uint32_t)k << shift) * list[n - 1];
  return (uint16_t)upper_bound_d(list, n, b);
}
// Entry e of a cell's row: the ions' guide first, then every ion's.
AHD void populate_cool_guide(const Env &env, int c, int e) {
  const DevModel &M = env.M;
  uint16_t *g = env.K.cool_guide + (krow(env, c) * M.nguide);
  const double *list = env.K.ion_cooling_contribs + (krow(env, c) * M.nions);
  int n = M.nions, shift = M.guide_ion_shift, k = e;
  if (e >= M.ion_guideoff[0]) {
    int ui = 0;
    while (ui + 1 < M.nions && M.ion_guideoff[ui + 1] <= e) ui++;
    list = env.K.cooling_contrib + (krow(env, c) * M.ncoolingterms) + M.ion_coolingoffset[ui];
    n = M.ion_ncoolingterms[ui];
    shift = M.ion_guideshift[ui];
    k = e - M.ion_guideoff[ui];
  }
  g[e] = cool_guide_entry(list, n, shift, k);
}
// upper_bound_d(list, n, v) for v = the value of the 24-bit draw u, by the guide g of the list: the same index
AHD int guided_upper_bound(const double *list, int n, double v, const uint16_t *g, int shift, uint32_t u) {
  const int k = (int)(u >> shift);
  const int lo = g[k], hi = g[k + 1];
  // (the sums of the range, few and rarely any: most ranges lie inside one term; non-decreasing, so the first greater one ends the count)
  int r = lo;
  while (r < hi && !(v < list[r])) r++;
  return r;
}

// ================================================================ r-packet path
AHD double chi_total(const Chi &x) { return x.chi_escatter + x.chi_boundfree + x.chi_freefree_heat; }  // rpkt.h:100

// photoionisation_crosssection_fromtable (atomic.h:201) in two halves, so that the table reads of several continua can
// be in flight together: phixs_lookup() decides which table entries are needed and reads them, phixs_finish() is the
// arithmetic. phixs_finish(phixs_lookup()) == phixs_fromtable() bit for bit.
struct PhixsRead {
  float a, b;
  int i;
};
AHD PhixsRead phixs_lookup(const DevModel &M, const float *xs, double nu_edge, double nu) {
  const int NP = M.NPHIXSPOINTS;
  const double INC = M.NPHIXSNUINCREMENT;
  PhixsRead r = {0.f, 0.f, -1};
#if ARTIS_OPT_PHIXS_CLASSIC_NO_INTERPOLATION
  if (nu < nu_edge) return r;
  int i = 0;
  if (nu == nu_edge) {
    i = 0;
  } else if (nu < nu_edge * (1 + (INC * NP))) {
    i = (int)((nu - nu_edge) / (INC * nu_edge));
    if (NP - 1 < i) i = NP - 1;
  } else {
    i = NP;  // beyond the table: extrapolate from the last point
  }
  r.i = i;
  r.a = xs[(i < NP) ? i : NP - 1];
#else
  const double ireal = ((nu / nu_edge) - 1.0) / INC;
  const int i = (int)floor(ireal);
  if (i < 0) return r;
  r.i = i;
  if (i < NP - 1) {
    r.a = xs[i];
    r.b = xs[i + 1];
  } else {
    r.a = xs[NP - 1];
  }
#endif
  return r;
}
// ... with the table read at a clamped place whatever the branch (one or two unconditional loads; an entry that phixs_finish()
// does not use is read for nothing): phixs_finish(phixs_lookup_u()) == phixs_finish(phixs_lookup())
AHD PhixsRead phixs_lookup_u(const DevModel &M, const float *xs, double nu_edge, double nu) {
  const int NP = M.NPHIXSPOINTS;
  const double INC = M.NPHIXSNUINCREMENT;
  PhixsRead r = {0.f, 0.f, -1};
#if ARTIS_OPT_PHIXS_CLASSIC_NO_INTERPOLATION
  int i = -1;
  if (!(nu < nu_edge)) {
    if (nu == nu_edge) {
      i = 0;
    } else if (nu < nu_edge * (1 + (INC * NP))) {
      i = (int)((nu - nu_edge) / (INC * nu_edge));
      if (NP - 1 < i) i = NP - 1;
    } else {
      i = NP;
    }
  }
  r.i = i;
  r.a = xs[(i < 0) ? 0 : ((i < NP) ? i : NP - 1)];
#else
  const double ireal = ((nu / nu_edge) - 1.0) / INC;
  const int i = (int)floor(ireal);
  r.i = (i < 0) ? -1 : i;
  const int ia = (i < 0) ? 0 : ((i < NP - 1) ? i : NP - 1);
  const int ib = (ia + 1 < NP) ? ia + 1 : NP - 1;
  r.a = xs[ia];
  r.b = xs[ib];
#endif
  return r;
}
AHD float phixs_finish(const DevModel &M, const PhixsRead r, double nu_edge, double nu) {
  const int NP = M.NPHIXSPOINTS;
  const double INC = M.NPHIXSNUINCREMENT;
  if (r.i < 0) return 0.f;
#if ARTIS_OPT_PHIXS_CLASSIC_NO_INTERPOLATION
  if (r.i < NP) return r.a;
  // beyond the table: the last point extrapolated as nu^-3 (atomic.h:222 pow(nu_max_phixs / nu, 3)). On the device the cube is two
  // multiplications (round 6): within 2 ulp of the correctly rounded power, i.e. 2e-16 of a value that is rounded to float next -- the device's
  // pow() was no closer to glibc's, but cost a call with its constants in scratch inside the opacity sum (26 scratch instructions and ~150
  // others per continuum beyond its table: the continua of excited levels seen by an ultraviolet packet, a third of all visits). The host
  // emulation keeps pow(): it is compared with the oracle bit for bit.
#if defined(__HIP_DEVICE_COMPILE__)
  const double x = nu_edge * (1 + (INC * NP)) / nu;
  return (float)(r.a * (x * x * x));
#else
  return (float)(r.a * pow(nu_edge * (1 + (INC * NP)) / nu, 3));
#endif
#else
  if (r.i < NP - 1) {
    const double a = r.a;
    const double b = r.b;
    const double ireal = ((nu / nu_edge) - 1.0) / INC;
    const double fb = ireal - r.i;
    return (float)(((1. - fb) * a) + (fb * b));
  }
  const double nu_max_phixs = nu_edge * M.last_phixs_nuovernuedge;
  return (float)(r.a * pow3(nu_max_phixs / nu));
#endif
}

// calculate_chi_bf_gammacontr<true, SELECT> rpkt.cc:721. The continua that contribute (keep bitmap of the cell) are
// taken CHI_BATCH at a time: all reads of a batch are issued before any of its arithmetic, the sum is accumulated in
// the reference's order.
// Round 4: index -> table row -> cross section is a chain of two memory latencies per continuum, and the loop waited for both.
// ARTIS_CHI_PREFETCH 2 = three continua in flight per lane (indices two ahead, pairs and cross sections one ahead; below), with
// batches of ONE: k_rpkt 345 -> 287 ms (classic), 745 -> 612 ms (nltenebular, with k_bfest_dense). Measured beside it: 1 = only the
// next batch's index and pair requested ahead (323 ms at batches of 2, 397 at 1, 375 / 395 at 3 / 4); 2 with batches of 2: 413 ms
// (its operands do not fit 168 VGPRs: values still on their way are spilled, i.e. waited for); cross sections two / three ahead
// (ARTIS_CHI_DEPTH 2 / 3): 299 / 406 ms. 0 = rounds 2-3 (batches of 2, 4 with detailed bound-free estimators: 345 ms).
#ifndef ARTIS_LINE_AHEAD
#define ARTIS_LINE_AHEAD 3  // lines whose (frequency, population factor) pairs are in flight in the line walk (possible_event()):
                           // k_rpkt 287 (1: the next line's pair only, rounds 2-3) / 283 / 278 / 288 ms at 1 / 2 / 3 / 4 (MI355X, round 4)
#endif
#ifndef ARTIS_CHI_BATCH
#define ARTIS_CHI_BATCH 1
#endif
#ifndef ARTIS_CHI_PREFETCH
#define ARTIS_CHI_PREFETCH 2
#endif
#ifndef ARTIS_CHI_DEPTH
#define ARTIS_CHI_DEPTH 1
#endif
// iterator over the set bits of a cell's keep bitmap inside [cbegin, cend), one 64-bit word per read
struct KeepIter {
  const uint64_t *keep;
  int word, cbegin, cend;
  uint64_t bits;
};
AHD uint64_t keep_masked(KeepIter &it) {
  uint64_t bits = it.keep[it.word];
  if (it.word == (it.cbegin / 64)) bits &= ~UINT64_C(0) << (unsigned)(it.cbegin % 64);
  if (((it.word + 1) * 64) > it.cend) bits &= ~UINT64_C(0) >> (unsigned)(64 - (it.cend % 64));
  return bits;
}
AHD int keep_next(KeepIter &it) {
  while (it.bits == 0) {
    it.word++;
    if (it.word * 64 >= it.cend) return -1;
    it.bits = keep_masked(it);
  }
  const int i = (it.word * 64) + __builtin_ctzll(it.bits);
  it.bits &= it.bits - 1;
  return i;
}
template <bool SELECT>
AHD double chi_bf_gammacontr(const Env &env, int c, double nu, int64_t slot, double threshold, int *selected, Chi *keep = nullptr) {
  const DevModel &M = env.M;
  double sum = 0.;
  // std::ranges::fill(groundcont_gamma_contr, 0.) rpkt.cc:728: the list starts empty. The ground-continuum index of
  // the continua rises with nu_edge (nearest-edge map of a sorted list, input.cc:703), so equal indices are adjacent.
  int ng = 0;
#if ARTIS_OPT_USE_LUT_PHOTOION || ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS
  int lastgi = -1;
  double *wsv = nullptr;
  int32_t *wsi = nullptr;
  if (!SELECT) {
    wsv = env.gamma_ws + (slot * M.nbfcontinua_ground);
    wsi = env.gamma_gi + (slot * M.nbfcontinua_ground);
  }
#endif
  const float T_e = env.C.Te[c];
  const double ex = exp(-HOVERKB * nu / T_e);
  const bool split_usable = (ex >= DBLMIN);
  // the window of continua whose edge lies in (nu / last_phixs_nuovernuedge, nu] (rpkt.cc:745-760). With the continuum
  // table in LDS (k_rpkt) the two bisections read the edges from it (stride 32 B) instead of from the dense array in HBM.
  int cend, cbegin;
  if (env.cont_in_lds) {
    const double nu_lo = nu / M.last_phixs_nuovernuedge;
    const ContPack *cpk = M.cont_pack;
    cend = partition_point_f(M.nbfcontinua, [cpk, nu](int i) { return !(nu < cpk[i].nu_edge); });
    cbegin = partition_point_f(cend, [cpk, nu_lo](int i) { return cpk[i].nu_edge < nu_lo; });
  } else {
    cend = upper_bound_d(M.allcont_nu_edge, M.nbfcontinua, nu);
    cbegin = lower_bound_d(M.allcont_nu_edge, cend, nu / M.last_phixs_nuovernuedge);
  }
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
  if (!SELECT && keep != nullptr) {
    keep->bf_begin = cbegin;
    keep->bf_end = cend;
  }
#endif
  int nvisited = 0;
  const double *departure = env.K.allcont_departure + (krow(env, c) * M.nbfcontinua);
  // the kept continua of the window: places [r0, r1) of the cell's list; their indices and {nnlevel, edgepart} pairs lie
  // next to each other there (rising index, so the sum runs in the reference's order)
  int r0 = 0, r1 = 0;
  if (cbegin < cend) kept_range(env, c, cbegin, cend, r0, r1);
  const int32_t *keptlist = env.K.allcont_keptlist + (krow(env, c) * M.nbfcontinua);
  const D2 *keptpair = env.K.allcont_keptpair + (krow(env, c) * M.nbfcontinua);
  // one continuum's term (rpkt.cc:770-798); true: SELECT has found its continuum
  auto add_term = [&](int i, const ContPack &cpk, const PhixsRead &xrk, double nnk, double epk) -> bool {
    nvisited++;
    const double nu_edge = cpk.nu_edge;
    const double sigma_bf = phixs_finish(M, xrk, nu_edge, nu);
    double stim;
    if (epk >= 0. && split_usable) {
      stim = epk * ex;
    } else {
      stim = departure[i] * exp(-HOVERKB * (nu - nu_edge) / T_e);
    }
    const double corr = dmax(0., 1 - stim);
    const double sigma_contr = sigma_bf * cpk.probability * corr;
#if ARTIS_OPT_USE_LUT_PHOTOION || ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS
    if (!SELECT) {
      const int gi = cpk.gi;
      if (gi >= 0) {
        if (gi == lastgi) {
          wsv[ng - 1] = sigma_contr;
        } else {
          wsv[ng] = sigma_contr;
          wsi[ng] = gi;
          ng++;
          lastgi = gi;
        }
      }
    }
#endif
    sum += nnk * sigma_contr;
    if (SELECT && sum > threshold) {
      *selected = i;
      return true;
    }
    return false;
  };
#if ARTIS_CHI_PREFETCH == 2
  // Three batches in flight (round 4): index -> table row -> cross section is a chain of two memory latencies and one LDS read per
  // continuum. The indices are requested two batches ahead, the pairs and the cross sections one batch ahead (every read
  // unconditional and from a clamped place, so that the wait for this batch's operands leaves exactly the younger requests in
  // flight), and the arithmetic of a batch runs on operands requested a whole iteration earlier. Same terms, same order.
  // The sets of operands rotate by NAME (the loop is unrolled ARTIS_CHI_DEPTH + 2 batches deep): a register move of a value still
  // on its way would wait for it. ARTIS_CHI_DEPTH = how many batches ahead the cross sections are requested (the indices one more,
  // the pairs -- contiguous, four to a 64-byte line -- one).
  {
    constexpr int B = ARTIS_CHI_BATCH, DX = ARTIS_CHI_DEPTH, NS = DX + 2;
    int idx[NS][B];
    D2 pr[NS][B];
    PhixsRead xr[NS][B];
    if (r0 < r1) {
#pragma unroll
      for (int d = 0; d <= DX; d++) {
#pragma unroll
        for (int k = 0; k < B; k++) idx[d][k] = keptlist[(r0 + (d * B) + k < r1) ? r0 + (d * B) + k : r1 - 1];
      }
#pragma unroll
      for (int k = 0; k < B; k++) pr[0][k] = keptpair[(r0 + k < r1) ? r0 + k : r1 - 1];
#pragma unroll
      for (int d = 0; d < DX; d++) {
#pragma unroll
        for (int k = 0; k < B; k++) {
          const ContPack cpk = M.cont_pack[idx[d][k]];
          xr[d][k] = phixs_lookup_u(M, M.allphixs + cpk.xs_off, cpk.nu_edge, nu);
        }
      }
    }
    for (int rs = r0; rs < r1; rs += NS * B) {  // [census: opacity sum] (tools/isa_census.py finds the loop by this tag)
#pragma unroll
      for (int u = 0; u < NS; u++) {
        const int r = rs + (u * B);
        if (r >= r1) break;
        const int sc = u, sp = (u + 1) % NS, sx = (u + DX) % NS, si = (u + DX + 1) % NS;
#pragma unroll
        for (int k = 0; k < B; k++) pr[sp][k] = keptpair[(r + B + k < r1) ? r + B + k : r1 - 1];
#pragma unroll
        for (int k = 0; k < B; k++) idx[si][k] = keptlist[(r + ((DX + 1) * B) + k < r1) ? r + ((DX + 1) * B) + k : r1 - 1];
#pragma unroll
        for (int k = 0; k < B; k++) {
          const ContPack cpk = M.cont_pack[idx[sx][k]];
          xr[sx][k] = phixs_lookup_u(M, M.allphixs + cpk.xs_off, cpk.nu_edge, nu);
        }
#pragma unroll
        for (int k = 0; k < B; k++) {
          if (r + k < r1) {
            const ContPack cpk = M.cont_pack[idx[sc][k]];
            if (add_term(idx[sc][k], cpk, xr[sc][k], pr[sc][k].x, pr[sc][k].y)) return sum;
          }
        }
      }
    }
  }
#else
#if ARTIS_CHI_PREFETCH
  // (the next batch's indices and pairs are requested while this batch's cross sections are on their way: one memory latency
  // per batch instead of two -- index -> table row -> cross section is a chain)
  int idx_n[ARTIS_CHI_BATCH];
  D2 pr_n[ARTIS_CHI_BATCH];
  if (r0 < r1) {
#pragma unroll
    for (int k = 0; k < ARTIS_CHI_BATCH; k++) {
      const int rk = (r0 + k < r1) ? r0 + k : r1 - 1;  // (a place past the end reads the last one: unconditional loads)
      idx_n[k] = keptlist[rk];
      pr_n[k] = keptpair[rk];
    }
  }
#endif
  for (int r = r0; r < r1; r += ARTIS_CHI_BATCH) {
    int idx[ARTIS_CHI_BATCH];
    ContPack cp[ARTIS_CHI_BATCH];
    double nn[ARTIS_CHI_BATCH], ep[ARTIS_CHI_BATCH];
#if ARTIS_CHI_PREFETCH
#pragma unroll
    for (int k = 0; k < ARTIS_CHI_BATCH; k++) {
      idx[k] = (r + k < r1) ? idx_n[k] : -1;
      nn[k] = pr_n[k].x;
      ep[k] = pr_n[k].y;
    }
#else
#pragma unroll
    for (int k = 0; k < ARTIS_CHI_BATCH; k++) {
      const int rk = (r + k < r1) ? r + k : r;
      idx[k] = (r + k < r1) ? keptlist[rk] : -1;
      const D2 pr = keptpair[rk];
      nn[k] = pr.x;
      ep[k] = pr.y;
    }
#endif
#if defined(ARTIS_RPKT_EXTRA_LOADS) && defined(__HIP_DEVICE_COMPILE__)
    {  // (measurement only: one more 8-byte read per batch of continua, of a pair just read)
      const double qx = *(const volatile double *)(&keptpair[r].x);
      asm volatile("" ::"v"(qx));
    }
#endif
#pragma unroll
    for (int k = 0; k < ARTIS_CHI_BATCH; k++) cp[k] = M.cont_pack[(idx[k] >= 0) ? idx[k] : idx[0]];
    PhixsRead xr[ARTIS_CHI_BATCH];
#pragma unroll
    for (int k = 0; k < ARTIS_CHI_BATCH; k++) xr[k] = phixs_lookup(M, M.allphixs + cp[k].xs_off, cp[k].nu_edge, nu);
#if ARTIS_CHI_PREFETCH
    {  // (unconditional, from clamped places: the wait for this batch's cross sections can then leave exactly these in flight)
      const int rn = r + ARTIS_CHI_BATCH;
#pragma unroll
      for (int k = 0; k < ARTIS_CHI_BATCH; k++) {
        const int rk = (rn + k < r1) ? rn + k : r1 - 1;
        idx_n[k] = keptlist[rk];
        pr_n[k] = keptpair[rk];
      }
    }
#endif
#pragma unroll
    for (int k = 0; k < ARTIS_CHI_BATCH; k++) {
      if (idx[k] >= 0) {
        if (add_term(idx[k], cp[k], xr[k], nn[k], ep[k])) return sum;
      }
    }
  }
#endif
  if (!SELECT) {
    env.gamma_n[slot] = ng;
    ARTIS_STAT(env, ARTIS_STAT_X_CHI_EVALS);
    ARTIS_STAT_ADD(env, ARTIS_STAT_X_CONT_VISITED, nvisited);
  }
  if (SELECT) {
    *selected = cend - 1;
    return sum;
  }
  if (!isfinite(sum)) fail(env, 30);
  return sum;
}
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
// globals::allcont.bfestimindex (input.cc:932): the estimator of continuum i, -1 if LEVEL_HAS_BFEST() is false for its level
AHD int bfestimindex(const DevModel &M, int i) { return M.allcont_bfestimindex ? M.allcont_bfestimindex[i] : i; }
// the contribution of continuum i at frequency nu in cell c: sigma_bf * probability * stimulated-emission correction, the
// arithmetic of calculate_chi_bf_gammacontr() (rpkt.cc:770-798) for one continuum
AHD double bf_sigma_contr_ep(const Env &env, int c, int i, double nu, float T_e, double ex, bool split_usable, double ep) {
  const DevModel &M = env.M;
  const ContPack cp = M.cont_pack[i];
  const double sigma_bf = phixs_fromtable(M, M.allphixs + cp.xs_off, cp.nu_edge, nu);
  double stim;
  if (ep >= 0. && split_usable) {
    stim = ep * ex;
  } else {
    stim = env.K.allcont_departure[(krow(env, c) * M.nbfcontinua) + i] * exp(-HOVERKB * (nu - cp.nu_edge) / T_e);
  }
  const double corr = dmax(0., 1 - stim);
  return sigma_bf * cp.probability * corr;
}
AHD double bf_sigma_contr(const Env &env, int c, int i, double nu, float T_e, double ex, bool split_usable) {
  return bf_sigma_contr_ep(env, c, i, nu, T_e, ex, split_usable, env.K.allcont_pair[(krow(env, c) * env.M.nbfcontinua) + i].y);
}
// radfield::update_bfestimators radfield.cc:215. The reference keeps, per packet, the contribution sigma_contr of every
// continuum of the window that calculate_chi_bf_gammacontr() walked at the frequency x.nu (Phixslist::gamma_contr), and
// adds it to bfrate_raw for the continua that are still in the window at the packet's present frequency. Here the
// contributions are not kept (nbfcontinua doubles per packet) but recomputed: same cell, same x.nu, same arithmetic as
// in chi_bf_gammacontr(), so the same bits. Every continuum has an estimator (LEVEL_HAS_BFEST true), so the estimator
// index is the continuum index.
AHD void update_bfestimators(const Env &env, int c, double de, double nu_cmf, const Chi &x) {
  const DevModel &M = env.M;
  const double de_over_nu = de / nu_cmf;
  const double nu = x.nu;  // the frequency the contributions belong to
  // the window stored with the opacity (rpkt.cc:762-768) ...
  int begin_n, end_n;
  if (x.bf_end >= 0) {
    begin_n = x.bf_begin;
    end_n = x.bf_end;
  } else {
    end_n = upper_bound_d(M.allcont_nu_edge, M.nbfcontinua, nu);
    begin_n = lower_bound_d(M.allcont_nu_edge, end_n, nu / M.last_phixs_nuovernuedge);
  }
  // ... narrowed to the packet's present frequency (radfield.cc:229-244). The estimators are taken at the middle of
  // the step, where the comoving frequency is a little below the one the opacity was evaluated at: the lower end of the
  // window stays (its bound only moves down) and the upper end drops by the few edges in between, found by stepping
  // down instead of bisecting the whole table. (upper_bound: first edge > nu_cmf.)
  if (nu_cmf < nu) {
    while (end_n > begin_n && M.allcont_nu_edge[end_n - 1] > nu_cmf) end_n--;
  } else if (nu_cmf > nu) {
    end_n = upper_bound_d(M.allcont_nu_edge, end_n, nu_cmf);
    const int b0 = begin_n < end_n ? begin_n : end_n;
    begin_n = b0 + lower_bound_d(M.allcont_nu_edge + b0, end_n - b0, nu_cmf / M.last_phixs_nuovernuedge);
  }
  if (begin_n >= end_n) return;
#if defined(__HIP_DEVICE_COMPILE__)
  // On the GPU the additions themselves are deferred: lanes of a wave are at windows of very different lengths (0 to
  // more than a hundred continua), so a lane-per-packet loop runs at a fraction of the lanes (17 % measured). The update
  // is recorded (one atomic per wave for the space) and k_bfest_dense gives every record a whole wave.
  if (env.bfev != nullptr) {
    const unsigned long long active = __ballot(1);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __ffsll((long long)active) - 1) base = atomicAdd(env.bfev_count, __popcll(active));
    base = __builtin_amdgcn_readfirstlane(base);  // the first active lane is the one that reserved the space
    const int idx = base + __popcll(active & ((1ull << lane) - 1ull));
    if (idx < env.bfev_cap) {
      env.bfev[idx] = BfEvent{nu, de_over_nu, c, begin_n, end_n, 0};
      return;
    }
  }
#endif
  const float T_e = env.C.Te[c];
  const double ex = exp(-HOVERKB * nu / T_e);
  const bool split_usable = (ex >= DBLMIN);
  const D2 *pairs = env.K.allcont_pair + (krow(env, c) * M.nbfcontinua);
  const double *departure = env.K.allcont_departure + (krow(env, c) * M.nbfcontinua);
  // the same batched walk as chi_bf_gammacontr(): the reads of a few continua are in flight together
  KeepIter it;
  it.keep = env.K.allcont_keepbits + (krow(env, c) * M.nkeepwords);
  it.cbegin = begin_n;
  it.cend = end_n;
  it.word = begin_n / 64;
  it.bits = keep_masked(it);
  double *dst = env.E.bfrate_raw + ((int64_t)c * M.nbfestim);
  bool more = true;
  while (more) {
    int idx[ARTIS_CHI_BATCH];
#pragma unroll
    for (int k = 0; k < ARTIS_CHI_BATCH; k++) {
      idx[k] = more ? keep_next(it) : -1;
      if (idx[k] < 0) more = false;
    }
    if (idx[0] < 0) break;
    ContPack cp[ARTIS_CHI_BATCH];
    double ep[ARTIS_CHI_BATCH];
#pragma unroll
    for (int k = 0; k < ARTIS_CHI_BATCH; k++) {
      const int i = (idx[k] >= 0) ? idx[k] : idx[0];
      cp[k] = M.cont_pack[i];
      ep[k] = pairs[i].y;
    }
    PhixsRead xr[ARTIS_CHI_BATCH];
#pragma unroll
    for (int k = 0; k < ARTIS_CHI_BATCH; k++) xr[k] = phixs_lookup(M, M.allphixs + cp[k].xs_off, cp[k].nu_edge, nu);
#pragma unroll
    for (int k = 0; k < ARTIS_CHI_BATCH; k++) {
      if (idx[k] >= 0) {
        const int i = idx[k];
        const double sigma_bf = phixs_finish(M, xr[k], cp[k].nu_edge, nu);
        double stim;
        if (ep[k] >= 0. && split_usable) {
          stim = ep[k] * ex;
        } else {
          stim = departure[i] * exp(-HOVERKB * (nu - cp[k].nu_edge) / T_e);
        }
        const double corr = dmax(0., 1 - stim);
        const double sigma_contr = sigma_bf * cp[k].probability * corr;
        const int bi = bfestimindex(M, i);  // Phixslist::gamma_contr is indexed by estimator (rpkt.cc:905)
        if (bi >= 0) ARTIS_EST_ADD(&dst[bi], sigma_contr * de_over_nu);
      }
    }
  }
}
#endif
// calculate_chi_rpkt_cont<true> rpkt.cc:1021 with calculate_chi_ffheating rpkt.cc:697
AHD void chi_rpkt_cont(const Env &env, double nu_cmf, Chi &x, int c, int64_t slot) {
  if ((c == x.nonemptymgi) && (fabs((x.nu / nu_cmf) - 1.0) < 1e-4)) return;
  const float nne = env.C.nne[c];
  const float cnne = env.C.nne[c] * env.C.clumpfactor[c];
  const float T_e = env.C.Te[c];
  x.chi_freefree_heat = env.K.chi_ff_nnionpart[krow(env, c)] / pow3(nu_cmf) * cnne * (1 - exp(-HOVERKB * nu_cmf / T_e));
  x.chi_escatter = SIGMA_T * nne;
  x.chi_boundfree = chi_bf_gammacontr<false>(env, c, nu_cmf, slot, 0., nullptr, &x);
  x.nonemptymgi = c;
  x.nu = nu_cmf;
}
// closest_transition rpkt.h:155
AHD int closest_transition(const double *nu, int nlines, double nu_cmf, int next_trans) {
  if (nlines <= 0) return -1;  // an empty line list (the reference would read linelist.nu.back() of an empty span)
  if (next_trans > (nlines - 1)) return -1;
  if (nu_cmf < nu[nlines - 1]) return -1;
  if (next_trans > 0) return next_trans;
  if (nu_cmf >= nu[0]) return 0;
  return partition_point_d(nu, nlines, [nu_cmf](double x) { return x > nu_cmf; });
}
AHD double linedistance(double prop_time, double nu_cmf, double nu_trans, double dnu_on_dl) {  // rpkt.h:117
  if (nu_cmf <= nu_trans) return 0.;
  const double dnu = nu_cmf - nu_trans;
#if ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT
  (void)prop_time;
  return -dnu / dnu_on_dl;  // linear interpolation of the frequency along the path, rpkt.h:126-132
#else
  (void)dnu_on_dl;
  return CLIGHT * prop_time * dnu / nu_trans;
#endif
}
// get_possible_event rpkt.cc:106
AHD double possible_event(const Env &env, int c, const Pkt &p, const Chi &x, MAState &ma, double tau_rnd, double abort_dist,
                          double nu_cmf_abort, double dnu_on_dl, double dop, int *next_trans_out, bool *is_bb) {
  const DevModel &M = env.M;
  const LineDpop dpop = line_dpop_of(env, c);
  double px = p.px, py = p.py, pz = p.pz;
  double nu_cmf = p.nu_cmf;
  double e_cmf = p.e_cmf;
  double prop_time = p.prop_time;
  int next_trans = p.next_trans;
  const double chi_cont = chi_total(x) * dop;
  double tau = 0.;
  double dist = 0.;
  int nvisited = 0;
  double result;
#if ARTIS_LINE_AHEAD > 1
  // The walk's chain of dependent reads is one (line frequency, population factor) pair per line visited, and after its first
  // line the walk takes the lines of the list one after the other (closest_transition() with next_trans > 0). The pairs of the
  // next ARTIS_LINE_AHEAD lines are in flight while a line is worked on: a ring of slots that rotates by NAME (the loop is
  // unrolled once round the ring; a register move of a value still on its way would wait for it); a slot is refilled as soon as
  // its line has been taken. Reads past the walk's last line are wasted, reads past the list's end go to its last line.
  {
    constexpr int NL = ARTIS_LINE_AHEAD;
    double nu_s[NL], dp_s[NL];
    int li = closest_transition(M.line_nu, M.nlines, nu_cmf, next_trans);
    if (li >= 0) {
#pragma unroll
      for (int d = 0; d < NL; d++) {
        const int l = (li + d < M.nlines) ? li + d : M.nlines - 1;
        nu_s[d] = M.line_nu[l];
        dp_s[d] = line_dpop_at(M, dpop, l);
      }
    }
    bool stop = false;
    while (!stop) {  // [census: line walk]
#pragma unroll
      for (int u = 0; u < NL; u++) {
    if (li < 0) {
      const double tau_cont = chi_cont * (abort_dist - dist);
      if (tau_rnd - tau > tau_cont) {
        *next_trans_out = next_trans;
        *is_bb = false;
        result = DBLMAX;
      } else {
        *next_trans_out = M.nlines + 1;
        *is_bb = false;
        result = dist + ((tau_rnd - tau) / chi_cont);
      }
      stop = true;
      break;
    }
    nvisited++;
    const double nu_trans = nu_s[u];
    const double dpop_li = dp_s[u];
    {
      const int l = (li + NL < M.nlines) ? li + NL : M.nlines - 1;
      nu_s[u] = M.line_nu[l];
      dp_s[u] = line_dpop_at(M, dpop, l);
    }
    next_trans = li + 1;
    const double ldist = linedistance(prop_time, nu_cmf, nu_trans, dnu_on_dl);
    const double tau_cont = chi_cont * ldist;
    if (tau_rnd - tau > tau_cont) {
      if (nu_trans < nu_cmf_abort) {
        *next_trans_out = next_trans - 1;
        *is_bb = false;
        result = DBLMAX;
        stop = true;
        break;
      }
      // get_tau_sobolev<true> rpkt.cc:75 with the cell cache's (B_lu n_l - B_ul n_u)
      const double tau_line = dmax(dpop_li * HCLIGHTOVERFOURPI * prop_time, 0.);
      if ((tau_rnd - tau) <= (tau_cont + tau_line)) {
        const LinePack lp = M.line_pack[li];
        const int element = M.line_elementindex[li];
        const int ion = M.line_ionindex[li];
        ma.element = element;
        ma.ion = ion;
        ma.level = lp.upper - lstart(M, element, ion);
        ma.activatingline = li;
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
        move_raw(px, py, pz, p.dx, p.dy, p.dz, prop_time, p.nu_rf, nu_cmf, p.e_rf, e_cmf, ldist);  // rpkt.cc:173-176
        update_lineestimator(env, c, li, prop_time * CLIGHT * e_cmf / nu_cmf);
#endif
        *next_trans_out = next_trans;
        *is_bb = true;
        result = dist + ldist;
        stop = true;
        break;
      }
      dist += ldist;
      tau += tau_cont + tau_line;
#if !ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT
      move_raw(px, py, pz, p.dx, p.dy, p.dz, prop_time, p.nu_rf, nu_cmf, p.e_rf, e_cmf, ldist);
#else
      // rpkt.cc:190-196: the linear approximation instead of the Doppler formula
      px += (p.dx * ldist);
      py += (p.dy * ldist);
      pz += (p.dz * ldist);
      prop_time += ldist / CLIGHT_PROP;
      nu_cmf = p.nu_cmf + (dnu_on_dl * dist);
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
      e_cmf = nu_cmf * p.e_rf / p.nu_rf;  // consistent with the linearly approximated nu_cmf, rpkt.cc:199-203
#else
      (void)e_cmf;
#endif
#endif
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
      update_lineestimator(env, c, li, prop_time * CLIGHT * e_cmf / nu_cmf);  // rpkt.cc:206
#endif
    } else {
      *next_trans_out = next_trans - 1;
      *is_bb = false;
      result = dist + ((tau_rnd - tau) / chi_cont);
      stop = true;
      break;
    }
    li = closest_transition(M.line_nu, M.nlines, nu_cmf, next_trans);
      }
    }
  }
#else
  // The walk's chain of dependent reads is one (line frequency, population factor) pair per line visited. The pair of
  // line li + 1 is requested while line li is worked on (the next line of the list is the next line of the walk,
  // closest_transition() with next_trans > 0): ahead_li says which line nu_ahead / dpop_ahead belong to.
  int ahead_li = -1;
  double nu_ahead = 0., dpop_ahead = 0.;
  while (true) {
    const int li = closest_transition(M.line_nu, M.nlines, nu_cmf, next_trans);
    if (li < 0) {
      const double tau_cont = chi_cont * (abort_dist - dist);
      if (tau_rnd - tau > tau_cont) {
        *next_trans_out = next_trans;
        *is_bb = false;
        result = DBLMAX;
        break;
      }
      *next_trans_out = M.nlines + 1;
      *is_bb = false;
      result = dist + ((tau_rnd - tau) / chi_cont);
      break;
    }
    nvisited++;
#if defined(ARTIS_RPKT_EXTRA_LOADS) && defined(__HIP_DEVICE_COMPILE__)
    if (ARTIS_RPKT_EXTRA_LOADS > 1) {  // (measurement only: one more 8-byte read per line visited, of the frequency just read)
      const double qx = *(const volatile double *)(M.line_nu + li);
      asm volatile("" ::"v"(qx));
    }
#endif
    const bool was_ahead = (li == ahead_li);
    const double nu_trans = was_ahead ? nu_ahead : M.line_nu[li];
    const double dpop_li = was_ahead ? dpop_ahead : line_dpop_at(M, dpop, li);
    if (li + 1 < M.nlines) {
      ahead_li = li + 1;
      nu_ahead = M.line_nu[li + 1];
      dpop_ahead = line_dpop_at(M, dpop, li + 1);
    }
    next_trans = li + 1;
    const double ldist = linedistance(prop_time, nu_cmf, nu_trans, dnu_on_dl);
    const double tau_cont = chi_cont * ldist;
    if (tau_rnd - tau > tau_cont) {
      if (nu_trans < nu_cmf_abort) {
        *next_trans_out = next_trans - 1;
        *is_bb = false;
        result = DBLMAX;
        break;
      }
      // get_tau_sobolev<true> rpkt.cc:75 with the cell cache's (B_lu n_l - B_ul n_u)
      const double tau_line = dmax(dpop_li * HCLIGHTOVERFOURPI * prop_time, 0.);
      if ((tau_rnd - tau) <= (tau_cont + tau_line)) {
        const LinePack lp = M.line_pack[li];
        const int element = M.line_elementindex[li];
        const int ion = M.line_ionindex[li];
        ma.element = element;
        ma.ion = ion;
        ma.level = lp.upper - lstart(M, element, ion);
        ma.activatingline = li;
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
        move_raw(px, py, pz, p.dx, p.dy, p.dz, prop_time, p.nu_rf, nu_cmf, p.e_rf, e_cmf, ldist);  // rpkt.cc:173-176
        update_lineestimator(env, c, li, prop_time * CLIGHT * e_cmf / nu_cmf);
#endif
        *next_trans_out = next_trans;
        *is_bb = true;
        result = dist + ldist;
        break;
      }
      dist += ldist;
      tau += tau_cont + tau_line;
#if !ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT
      move_raw(px, py, pz, p.dx, p.dy, p.dz, prop_time, p.nu_rf, nu_cmf, p.e_rf, e_cmf, ldist);
#else
      // rpkt.cc:190-196: the linear approximation instead of the Doppler formula
      px += (p.dx * ldist);
      py += (p.dy * ldist);
      pz += (p.dz * ldist);
      prop_time += ldist / CLIGHT_PROP;
      nu_cmf = p.nu_cmf + (dnu_on_dl * dist);
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
      e_cmf = nu_cmf * p.e_rf / p.nu_rf;  // consistent with the linearly approximated nu_cmf, rpkt.cc:199-203
#else
      (void)e_cmf;
#endif
#endif
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
      update_lineestimator(env, c, li, prop_time * CLIGHT * e_cmf / nu_cmf);  // rpkt.cc:206
#endif
    } else {
      *next_trans_out = next_trans - 1;
      *is_bb = false;
      result = dist + ((tau_rnd - tau) / chi_cont);
      break;
    }
  }
#endif
  ARTIS_STAT_ADD(env, ARTIS_STAT_X_LINES_VISITED, nvisited);
  return result;
}

#if ARTIS_EXPOPAC_TABLES
// wavelength bins in ascending wavelength (descending frequency), rpkt.h:30-40
AHD double expopac_bin_nu_upper(int64_t b) { return 1e8 * CLIGHT / (ARTIS_EXPOPAC_LAMBDAMIN + ((double)b * ARTIS_EXPOPAC_DELTALAMBDA)); }
AHD double expopac_bin_nu_lower(int64_t b) { return 1e8 * CLIGHT / (ARTIS_EXPOPAC_LAMBDAMIN + ((double)(b + 1) * ARTIS_EXPOPAC_DELTALAMBDA)); }
AHD int64_t linearbinindex(double value, double minvalue, double binwidth) {  // get_linearbinindex sn3d.h:115
  const double fracindex = (value - minvalue) / binwidth;
  const int64_t truncated = (int64_t)fracindex;
  return (fracindex < (double)truncated) ? truncated - 1 : truncated;
}
#endif
#if ARTIS_EXPOPAC_TABLES
// calculate_expansion_opacities rpkt.cc:1071, run by the engine when the host does not hand the tables over (they are a
// product of update_grid() in the reference, update_grid.cc:655). One (cell, bin): the lines of the bin in list order.
AHD void populate_expopac_bin(const Env &env, int c, int b) {
  const DevModel &M = env.M;
  const LineDpop dpop = line_dpop_of(env, c);
  const double t_mid = env.S.mid;
  double bin_linesum = 0.;
  const int l1 = M.expopac_linestart[b + 1];
  for (int li = M.expopac_linestart[b]; li < l1; li++) {
    const double tau_line = dmax(line_dpop_at(M, dpop, li) * HCLIGHTOVERFOURPI * t_mid, 0.);  // get_tau_sobolev rpkt.cc:75
    const double linelambda = 1e8 * CLIGHT / M.line_nu[li];
    bin_linesum += (linelambda / ARTIS_EXPOPAC_DELTALAMBDA) * -expm1(-tau_line);
  }
  const float kappa = (float)(1. / (CLIGHT * t_mid * env.C.rho[c]) * bin_linesum);
  if (!isfinite(kappa)) fail(env, 96);
  const_cast<float *>(env.C.expansionopacities)[((int64_t)c * ARTIS_EXPOPAC_NBINS) + b] = kappa;
}
// ... and one cell: the running integral of (kappa + free-free) * B_nu(T_e) over the bins (rpkt.cc:1102-1118)
AHD void populate_expopac_planck(const Env &env, int c) {
  const float *kappa = env.C.expansionopacities + ((int64_t)c * ARTIS_EXPOPAC_NBINS);
  double *cum = const_cast<double *>(env.C.expansionopacity_planck_cumulative) + ((int64_t)c * ARTIS_EXPOPAC_NBINS);
  const float rho = env.C.rho[c];
  const float T_e = env.C.Te[c];
  const float cnne = env.C.nne[c] * env.C.clumpfactor[c];
  double kappa_planck_cumulative = 0.;
  for (int b = 0; b < ARTIS_EXPOPAC_NBINS; b++) {
    const double nu_lower = expopac_bin_nu_lower(b);
    const double nu_upper = expopac_bin_nu_upper(b);
    const double nu_mid = (nu_upper + nu_lower) / 2.;
    const double chi_ff = env.K.chi_ff_nnionpart[krow(env, c)] / pow3(nu_mid) * cnne * (1 - exp(-HOVERKB * nu_mid / T_e));  // rpkt.cc:697
    const double bin_kappa_cont = chi_ff / rho;
    const double kappa_planck = (kappa[b] + bin_kappa_cont) * planck(nu_mid, T_e);
    const double delta_nu = nu_upper - nu_lower;
    kappa_planck_cumulative += kappa_planck * delta_nu;
    cum[b] = kappa_planck_cumulative;
  }
}
#endif
#if ARTIS_OPT_RPKT_BB_THERMALISATION
// sample_planck_times_expansion_opacity rpkt.cc:964
AHD double sample_planck_times_expopac(const Env &env, int c, Pkt &p) {
  const double *cum = env.C.expansionopacity_planck_cumulative + ((int64_t)c * ARTIS_EXPOPAC_NBINS);
  const double last = cum[ARTIS_EXPOPAC_NBINS - 1];
  if (!(last > 0)) fail(env, 95);
  const double rnd_integral = rng_uniform(p) * last;
  int b = upper_bound_d(cum, ARTIS_EXPOPAC_NBINS, rnd_integral);  // index_upperbound sn3d.h:85
  if (b > ARTIS_EXPOPAC_NBINS - 1) b = ARTIS_EXPOPAC_NBINS - 1;
  const double bin_nu_lower = expopac_bin_nu_lower(b);
  const double delta_nu = expopac_bin_nu_upper(b) - bin_nu_lower;
  const double nuoffset = rng_uniform(p) * delta_nu;
  return bin_nu_lower + nuoffset;
}
#endif
#if ARTIS_OPT_RPKT_USE_EXPANSION_OPACITIES
// get_possible_event_expansion_opacity rpkt.cc:221: the packet walks the wavelength bins of the cell's expansion
// opacity instead of the line list. With a thermalisation probability the event is placed inside the bin; without, the
// bin is re-traced line by line (possible_event() on a copy of the packet at the bin's start and the timestep's mid time).
AHD double possible_event_expopac(const Env &env, int c, Pkt &p, const Chi &x, MAState &ma, double tau_rnd, double nu_cmf_abort,
                                  double dnu_on_dl, double dop, bool *is_bb) {
  double px = p.px, py = p.py, pz = p.pz;
  double nu_cmf = p.nu_cmf;
  double e_cmf = p.e_cmf;
  double prop_time = p.prop_time;
  double dist = 0.;
  double tau = 0.;
  int64_t b0 = linearbinindex(1e8 * CLIGHT / nu_cmf, ARTIS_EXPOPAC_LAMBDAMIN, ARTIS_EXPOPAC_DELTALAMBDA);
  if (b0 < -1) b0 = -1;
  const float *kappa_bins = env.C.expansionopacities + ((int64_t)c * ARTIS_EXPOPAC_NBINS);
  const float rho = env.C.rho[c];
  const double chi_cont = chi_total(x) * dop;
  for (int64_t b = b0; b < ARTIS_EXPOPAC_NBINS; b++) {
    const double next_bin_edge_nu = (b < 0) ? expopac_bin_nu_upper(0) : expopac_bin_nu_lower(b);
    const double binedgedist = linedistance(prop_time, nu_cmf, next_bin_edge_nu, dnu_on_dl);
    double chi_bb = 0.;
    if (b >= 0) chi_bb = kappa_bins[b] * rho;  // float product (both are floats in the reference)
    const double chi_tot = chi_cont + chi_bb;
    if (chi_tot * binedgedist > tau_rnd - tau) {
#if ARTIS_OPT_RPKT_BB_THERMALISATION
      (void)ma;
      const double edist = dmax(dist + ((tau_rnd - tau) / chi_tot), 0.);
      *is_bb = rng_uniform(p) < chi_bb / chi_tot;
      return edist;
#else
      Pkt q = p;
      q.px = px; q.py = py; q.pz = pz;
      q.nu_cmf = nu_cmf;
      q.e_cmf = e_cmf;
      q.prop_time = env.S.mid;  // the expansion opacity was calculated at t_mid
      q.next_trans = -1;
      int nt = -1;
      const double edist_after_bin = possible_event(env, c, q, x, ma, tau_rnd - tau, DBLMAX, 0., dnu_on_dl, dop, &nt, is_bb);
      return dist + edist_after_bin;
#endif
    }
    tau += chi_tot * binedgedist;
    dist += binedgedist;
#if !ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT
    move_raw(px, py, pz, p.dx, p.dy, p.dz, prop_time, p.nu_rf, nu_cmf, p.e_rf, e_cmf, binedgedist);
#else
    px += (p.dx * binedgedist);
    py += (p.dy * binedgedist);
    pz += (p.dz * binedgedist);
    prop_time += binedgedist / CLIGHT_PROP;
    nu_cmf = p.nu_cmf + (dnu_on_dl * dist);
#if ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
    e_cmf = nu_cmf * p.e_rf / p.nu_rf;  // rpkt.cc:306: it seeds the packet copy of the line-by-line retrace
#endif
#endif
    if (nu_cmf <= nu_cmf_abort) {
      *is_bb = false;
      return DBLMAX;
    }
  }
  *is_bb = false;
  if (chi_cont > 0.) return dist + ((tau_rnd - tau) / chi_cont);
  return DBLMAX;
}
#endif

// em_pos = pos, em_time = prop_time (rpkt.cc:1012, rpkt.cc:449): straight to the flight line
AHD void set_em_here(const Env &env, const Pkt &p, int64_t pi) {
  PktFlight &fl = env.P.flight[pi];
  fl.em_pos_x = p.px;
  fl.em_pos_y = p.py;
  fl.em_pos_z = p.pz;
  fl.em_time = (float)p.prop_time;
}
// emit_rpkt rpkt.cc:991
AHD void emit_rpkt(const Env &env, Pkt &p, int64_t pi) {
  p.type = ARTIS_TYPE_RPKT;
  double dir_cmf[3];
  rand_isotropic(p, dir_cmf);
  const double t = -p.prop_time;
  const double vel[3] = {p.px / t, p.py / t, p.pz / t};
  double nd[3];
  angle_ab(dir_cmf, vel, nd);
  p.dx = nd[0];
  p.dy = nd[1];
  p.dz = nd[2];
  set_restframe_from_cmf(p);
#if ARTIS_OPT_POL_ON
  p.stokes_u = 0.;
  p.stokes_q = 0.;
#endif
  set_em_here(env, p, pi);
  p.flags |= PKT_FLAG_EMITTED;  // the new direction and rest-frame quantities have to reach the packet's flight line
}
// trueem_pos = em_pos, trueem_time = em_time (macroatom.cc:588, kpkt.cc:512). Every caller has just run emit_rpkt() on the
// unchanged packet, so em_pos is the packet's position and em_time its time: no read-back of the flight line.
AHD void set_trueem_from_em(const Env &env, Pkt &p, int64_t pi) {
  PktCold &cold = env.P.cold[pi];
  cold.trueem_pos_x = p.px;
  cold.trueem_pos_y = p.py;
  cold.trueem_pos_z = p.pz;
  cold.trueem_time = (float)p.prop_time;
  p.flags &= ~PKT_FLAG_TRUEEM_NAN;
}

// electron_scatter_rpkt rpkt.cc:331
AHD void electron_scatter(Pkt &p) {
  p.type = ARTIS_TYPE_RPKT;
  const double vel[3] = {p.px / p.prop_time, p.py / p.prop_time, p.pz / p.prop_time};
  const double dir[3] = {p.dx, p.dy, p.dz};
  double old_cmf[3], q_i = 0., u_i = 0.;
#if ARTIS_OPT_POL_ON
  frame_transform(dir, p.stokes_q, p.stokes_u, vel, old_cmf, &q_i, &u_i);
#else
  angle_ab(dir, vel, old_cmf);
#endif
  double Mc = 0., phisc = 0.;
#if ARTIS_OPT_DIPOLE
  {
    double pfn = 0., x = 1.;
    while (x > pfn) {
      Mc = (2. * rng_uniform_pos(p)) - 1.;
      const double mu2 = pow2(Mc);
      phisc = 2 * PI * rng_uniform(p);
      double s2p, c2p;
      sin_cos(2 * phisc, &s2p, &c2p);
      pfn = (mu2 + 1) + ((mu2 - 1) * ((c2p * q_i) + (s2p * u_i)));
      x = 2. * rng_uniform(p);
    }
  }
#else
  Mc = (2. * rng_uniform(p)) - 1.;
  phisc = 2 * PI * rng_uniform(p);
#endif
  double new_cmf[3];
  const double cos_tsc = Mc;
  const double sin_tsc = sqrt(1. - pow2(Mc));
  if (fabs(old_cmf[2]) < 0.99999) {
    const double sin_polar = sqrt(1. - pow2(old_cmf[2]));
    const double cf = sin_tsc / sin_polar;
    double sph, cph;
    sin_cos(phisc, &sph, &cph);
    new_cmf[0] = (cf * ((old_cmf[1] * sph) - (old_cmf[0] * old_cmf[2] * cph))) + (old_cmf[0] * cos_tsc);
    new_cmf[1] = (cf * ((-old_cmf[0] * sph) - (old_cmf[1] * old_cmf[2] * cph))) + (old_cmf[1] * cos_tsc);
    new_cmf[2] = (sin_tsc * cph * sin_polar) + (old_cmf[2] * cos_tsc);
  } else {
    double sph, cph;
    sin_cos(phisc, &sph, &cph);
    new_cmf[0] = sin_tsc * cph;
    new_cmf[1] = sin_tsc * sph;
    new_cmf[2] = (old_cmf[2] > 0) ? cos_tsc : -cos_tsc;
  }
#if ARTIS_OPT_POL_ON
  {
    double nd[3], q, u;
    scatter_polarisation_to_rf(old_cmf, new_cmf, q_i, u_i, vel, nd, &q, &u);
    p.dx = nd[0]; p.dy = nd[1]; p.dz = nd[2];
    p.stokes_q = q;
    p.stokes_u = u;
  }
#else
  {
    const double nv[3] = {-vel[0], -vel[1], -vel[2]};
    double nd[3];
    angle_ab(new_cmf, nv, nd);
    p.dx = nd[0]; p.dy = nd[1]; p.dz = nd[2];
  }
#endif
  set_restframe_from_cmf(p);
}

// ---------------------------------------------------------------- Gauss-Kronrod 31 (gausskronrod.h)
// Abscissae/weights of the 31-point Kronrod rule with its embedded 15-point Gauss rule: the
// QUADPACK dqk31 constants, stored as in gausskronrod.h:38-90.
struct GK31 {
  double x[16];
  double w[16];
  double wg[8];
};
AHD GK31 gk31_tables() {
  return GK31{
      {0.00000000000000000000000000000000000e+00, 1.01142066918717499027074231447392339e-01,
       2.01194093997434522300628303394596208e-01, 2.99180007153168812166780024266388963e-01,
       3.94151347077563369897207370981045468e-01, 4.85081863640239680693655740232350613e-01,
       5.70972172608538847537226737253910641e-01, 6.50996741297416970533735895313274693e-01,
       7.24417731360170047416186054613938010e-01, 7.90418501442465932967649294817947347e-01,
       8.48206583410427216200648320774216851e-01, 8.97264532344081900882509656454495883e-01,
       9.37273392400705904307758947710209471e-01, 9.67739075679139134257347978784337225e-01,
       9.87992518020485428489565718586612581e-01, 9.98002298693397060285172840152271209e-01},
      {1.01330007014791549017374792767492547e-01, 1.00769845523875595044946662617569722e-01,
       9.91735987217919593323931734846031311e-02, 9.66427269836236785051799076275893351e-02,
       9.31265981708253212254868727473457186e-02, 8.85644430562117706472754436937743032e-02,
       8.30805028231330210382892472861037896e-02, 7.68496807577203788944327774826590067e-02,
       6.98541213187282587095200770991474758e-02, 6.20095678006706402851392309608029322e-02,
       5.34815246909280872653431472394302968e-02, 4.45897513247648766082272993732796902e-02,
       3.53463607913758462220379484783600481e-02, 2.54608473267153201868740010196533594e-02,
       1.50079473293161225383747630758072681e-02, 5.37747987292334898779205143012764982e-03},
      {2.02578241925561272880620199967519315e-01, 1.98431485327111576456118326443839325e-01,
       1.86161000015562211026800561866422825e-01, 1.66269205816993933553200860481208811e-01,
       1.39570677926154314447804794511028323e-01, 1.07159220467171935011869546685869303e-01,
       7.03660474881081247092674164506673385e-02, 3.07532419961172683546283935772044177e-02}};
}
struct FbIntegrand {
  const DevModel *M;
  const float *xs;
  double nu_edge;
  float T_e;
};
AHD double fb_integrand(const FbIntegrand &f, double x) {  // alpha_sp_E_integrand ratecoeff.cc:84
  const double nu = f.nu_edge + x;
  const float sigma_bf = phixs_fromtable(*f.M, f.xs, f.nu_edge, nu);
  return (2 / CLIGHTSQUARED) * sigma_bf * pow3(nu) / f.nu_edge * exp(-HOVERKB * x / f.T_e);
}
AHD double gk31_unit(const FbIntegrand &f, const GK31 &g, double scale, double mean, double *error) {  // gausskronrod.h:173
  const double fc = fb_integrand(f, (scale * 0.) + mean);
  double kr = fc * g.w[0];
  double gr = 0.;
  gr += fc * g.wg[0];
  for (unsigned i = 2; i < 16; i += 2) {
    const double fp = fb_integrand(f, (scale * g.x[i]) + mean);
    const double fm = fb_integrand(f, (scale * -g.x[i]) + mean);
    kr += (fp + fm) * g.w[i];
    gr += (fp + fm) * g.wg[i / 2];
  }
  for (unsigned i = 1; i < 16; i += 2) {
    const double fp = fb_integrand(f, (scale * g.x[i]) + mean);
    const double fm = fb_integrand(f, (scale * -g.x[i]) + mean);
    kr += (fp + fm) * g.w[i];
  }
  *error = dmax(fabs(kr - gr), fabs(kr * 2.220446049250313e-16 * 2));
  return kr;
}
// recursive_adaptive_integrate<31> gausskronrod.h:208 as an explicit depth-first walk (no recursion on the GPU).
// The reference's evaluation order is: node, then left subtree, then right subtree, with
// estimate = left + right and error accumulated as error_left += error_right at each split.
AHD double gk31_adaptive(const FbIntegrand &f, const GK31 &g, double tol, double a0, double b0, double *error_out) {
  constexpr int MAXD = 16;
  // per depth: interval of the pending RIGHT child, its abs_tol, and the partial sums of the parent
  double ra[MAXD], rb[MAXD], rtol[MAXD], est_left[MAXD], err_left[MAXD];
  int state[MAXD];  // 0: left child in progress, 1: right child in progress
  int depth = 0;
  double a = a0, b = b0, abs_tol = 0.;
  unsigned levels = 15;
  double ret_est = 0., ret_err = 0.;
  while (true) {
    // evaluate node (a, b, levels, abs_tol)
    double err_local = 0.;
    const double mean = (b + a) / 2;
    const double scale = (b - a) / 2;
    const double r1 = gk31_unit(f, g, scale, mean, &err_local);
    const double estimate = scale * r1;
    const double abs_tol1 = fabs(estimate * tol);
    if (abs_tol == 0) abs_tol = abs_tol1;
    if ((levels != 0) && (abs_tol1 < err_local) && (abs_tol < err_local)) {
      const double mid = (a + b) / 2;
      ra[depth] = mid;
      rb[depth] = b;
      rtol[depth] = abs_tol / 2;
      state[depth] = 0;
      depth++;
      b = mid;
      abs_tol = abs_tol / 2;
      levels--;
      continue;
    }
    ret_est = estimate;
    ret_err = err_local;
    // unwind
    while (true) {
      if (depth == 0) {
        *error_out = ret_err;
        return ret_est;
      }
      const int d = depth - 1;
      if (state[d] == 0) {
        est_left[d] = ret_est;
        err_left[d] = ret_err;
        state[d] = 1;
        a = ra[d];
        b = rb[d];
        abs_tol = rtol[d];
        levels = 15 - depth;
        break;  // evaluate the right child
      }
      ret_est = est_left[d] + ret_est;
      ret_err = err_left[d] + ret_err;
      depth--;
    }
  }
}
AHD double integrator31(const FbIntegrand &f, const GK31 &g, double a, double b, double epsrel, double *abserr) {  // integrator.h:48
  if (a == b) return 0.;
  if (b < a) return -gk31_adaptive(f, g, epsrel, b, a, abserr);
  return gk31_adaptive(f, g, epsrel, a, b, abserr);
}
// select_continuum_nu ratecoeff.cc:563
struct Rng4 {
  uint32_t s0, s1, s2, s3;
};
ANOINLINE double select_continuum_nu_impl(const DevModel &M, int element, int lowerion, int lower, int t, float T_e, Rng4 *rs) {
  Pkt p;  // only the generator words are used
  p.s0 = rs->s0; p.s1 = rs->s1; p.s2 = rs->s2; p.s3 = rs->s3;
  const GK31 g = gk31_tables();
  const int ul = lstart(M, element, lowerion) + lower;
  const double E_threshold = phixs_threshold(M, element, lowerion, lower, t);
  const double nu_threshold = (1. / HPLANCK) * E_threshold;
  const double nu_max_phixs = nu_threshold * M.last_phixs_nuovernuedge;
  const int npieces = M.NPHIXSPOINTS;
  const FbIntegrand f = {&M, phixs_table(M, ul), nu_threshold, T_e};
  const double zrand = 1. - rng_uniform(p);
  const double nu_range = nu_max_phixs - nu_threshold;
  const double deltanu = nu_range / npieces;
  double error = 0.;
  const double total = integrator31(f, g, 0., nu_range, 1e-3, &error);
  if (!(total > 0.) || !isfinite(total)) {
    rs->s0 = p.s0; rs->s1 = p.s1; rs->s2 = p.s2; rs->s3 = p.s3;
    return nu_threshold;
  }
  double tail_prev = total;
  double tail = total;
  int i = 1;
  for (; i < npieces; i++) {
    tail_prev = tail;
    const double low = i * deltanu;
    tail = integrator31(f, g, low, nu_range, 1e-3, &error);
    if (zrand >= tail / total) break;
  }
  double nuoffset = 0.;
  if (i < npieces) {
    nuoffset = (tail != tail_prev) ? ((total * zrand) - tail_prev) / (tail - tail_prev) * deltanu : 0.;
  } else if (tail > 0.) {
    nuoffset = (tail - (total * zrand)) / tail * deltanu;
  }
  rs->s0 = p.s0; rs->s1 = p.s1; rs->s2 = p.s2; rs->s3 = p.s3;
  return nu_threshold + ((i - 1) * deltanu) + nuoffset;
}
AHD double select_continuum_nu(const Env &env, int element, int lowerion, int lower, int t, float T_e, Pkt &p) {
  Rng4 rs = {p.s0, p.s1, p.s2, p.s3};
  const double nu = select_continuum_nu_impl(env.M, element, lowerion, lower, t, T_e, &rs);
  p.s0 = rs.s0; p.s1 = rs.s1; p.s2 = rs.s2; p.s3 = rs.s3;
  return nu;
}
// A free-bound emission whose frequency a WAVE selects (round 5; k_slow, k_tail): the slow-path action is run twice on the lane that owns
// the packet -- first on a copy, up to the point where it would call select_continuum_nu(), whose arguments and random number are
// RECORDED there (mode 1: the action returns at once, nothing but the copy has changed); then the wave evaluates the up to 100 adaptive
// 31-point quadratures together (select_continuum_nu_wave: one abscissa per lane, the weighted sums added in gausskronrod.h's order: the
// same bits), and the action runs for real with the frequency handed in (mode 2: the generator makes the one draw the selection makes).
struct FbSel {
  int mode;  // 1: record and return, 2: replay with `nu`
  bool valid;
  int element, lowerion, lower, t;
  float T_e;
  double zrand, nu;
};
AHD double select_continuum_nu_sel(const Env &env, int element, int lowerion, int lower, int t, float T_e, Pkt &p, FbSel *sel) {
  if (sel == nullptr) return select_continuum_nu(env, element, lowerion, lower, t, T_e, p);
  if (sel->mode == 1) {
    sel->valid = true;
    sel->element = element; sel->lowerion = lowerion; sel->lower = lower; sel->t = t;
    sel->T_e = T_e;
    sel->zrand = 1. - rng_uniform(p);  // ratecoeff.cc:575, the selection's only draw
    return 0.;
  }
  (void)rng_uniform(p);
  return sel->nu;
}
#if defined(__HIPCC__) && !defined(ARTIS_HOST_EMU)
// gk31_unit() by a wave: lane 0 evaluates the integrand at the centre, lanes 1..15 at +x[lane], lanes 16..30 at -x[lane - 15]; every
// lane then adds the weighted sums in gk31_unit()'s order from the values of the other lanes (wave-uniform results)
__device__ inline double wave_bcast(double v, int srclane) {
  union { double d; int32_t i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], srclane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], srclane);
  return u.d;
}
__device__ inline double gk31_unit_wave(const FbIntegrand &f, const GK31 &g, double scale, double mean, double *error) {
  const int lane = (int)(threadIdx.x & 63);
  double xj = 0.;
#pragma unroll
  for (int i = 1; i < 16; i++) xj = (lane == i || lane == 15 + i) ? g.x[i] : xj;  // (selects: no indexed table in memory)
  const double arg = (lane == 0) ? ((scale * 0.) + mean) : ((lane <= 15) ? ((scale * xj) + mean) : ((scale * -xj) + mean));
  const double fv = fb_integrand(f, arg);
  const double fc = wave_bcast(fv, 0);
  double kr = fc * g.w[0];
  double gr = 0.;
  gr += fc * g.wg[0];
#pragma unroll
  for (unsigned i = 2; i < 16; i += 2) {
    const double fp = wave_bcast(fv, (int)i);
    const double fm = wave_bcast(fv, (int)(15 + i));
    kr += (fp + fm) * g.w[i];
    gr += (fp + fm) * g.wg[i / 2];
  }
#pragma unroll
  for (unsigned i = 1; i < 16; i += 2) {
    const double fp = wave_bcast(fv, (int)i);
    const double fm = wave_bcast(fv, (int)(15 + i));
    kr += (fp + fm) * g.w[i];
  }
  *error = dmax(fabs(kr - gr), fabs(kr * 2.220446049250313e-16 * 2));
  return kr;
}
// gk31_adaptive() with the wave's unit: the same depth-first walk; every value is wave-uniform
__device__ inline double gk31_adaptive_wave(const FbIntegrand &f, const GK31 &g, double tol, double a0, double b0, double *error_out) {
  constexpr int MAXD = 16;
  double ra[MAXD], rb[MAXD], rtol[MAXD], est_left[MAXD], err_left[MAXD];
  int state[MAXD];
  int depth = 0;
  double a = a0, b = b0, abs_tol = 0.;
  unsigned levels = 15;
  double ret_est = 0., ret_err = 0.;
  while (true) {
    double err_local = 0.;
    const double mean = (b + a) / 2;
    const double scale = (b - a) / 2;
    const double r1 = gk31_unit_wave(f, g, scale, mean, &err_local);
    const double estimate = scale * r1;
    const double abs_tol1 = fabs(estimate * tol);
    if (abs_tol == 0) abs_tol = abs_tol1;
    if ((levels != 0) && (abs_tol1 < err_local) && (abs_tol < err_local)) {
      const double mid = (a + b) / 2;
      ra[depth] = mid;
      rb[depth] = b;
      rtol[depth] = abs_tol / 2;
      state[depth] = 0;
      depth++;
      b = mid;
      abs_tol = abs_tol / 2;
      levels--;
      continue;
    }
    ret_est = estimate;
    ret_err = err_local;
    while (true) {
      if (depth == 0) {
        *error_out = ret_err;
        return ret_est;
      }
      const int d = depth - 1;
      if (state[d] == 0) {
        est_left[d] = ret_est;
        err_left[d] = ret_err;
        state[d] = 1;
        a = ra[d];
        b = rb[d];
        abs_tol = rtol[d];
        levels = 15 - depth;
        break;
      }
      ret_est = est_left[d] + ret_est;
      ret_err = err_left[d] + ret_err;
      depth--;
    }
  }
}
__device__ inline double integrator31_wave(const FbIntegrand &f, const GK31 &g, double a, double b, double epsrel, double *abserr) {
  if (a == b) return 0.;
  if (b < a) return -gk31_adaptive_wave(f, g, epsrel, b, a, abserr);
  return gk31_adaptive_wave(f, g, epsrel, a, b, abserr);
}
// select_continuum_nu_impl() (ratecoeff.cc:563) by a wave, for the recorded arguments and random number of one packet (all wave-uniform)
__device__ inline double select_continuum_nu_wave(const DevModel &M, int element, int lowerion, int lower, int t, float T_e, double zrand) {
  const GK31 g = gk31_tables();
  const int ul = lstart(M, element, lowerion) + lower;
  const double E_threshold = phixs_threshold(M, element, lowerion, lower, t);
  const double nu_threshold = (1. / HPLANCK) * E_threshold;
  const double nu_max_phixs = nu_threshold * M.last_phixs_nuovernuedge;
  const int npieces = M.NPHIXSPOINTS;
  const FbIntegrand f = {&M, phixs_table(M, ul), nu_threshold, T_e};
  const double nu_range = nu_max_phixs - nu_threshold;
  const double deltanu = nu_range / npieces;
  double error = 0.;
  const double total = integrator31_wave(f, g, 0., nu_range, 1e-3, &error);
  if (!(total > 0.) || !isfinite(total)) return nu_threshold;
  double tail_prev = total;
  double tail = total;
  int i = 1;
  for (; i < npieces; i++) {
    tail_prev = tail;
    const double low = i * deltanu;
    tail = integrator31_wave(f, g, low, nu_range, 1e-3, &error);
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
// the wave's part of a slow-path step: every lane whose recorded selection is valid gets its frequency (lane by lane, all lanes working)
__device__ inline void fbsel_wave(const Env &env, FbSel &sel) {
  unsigned long long m = __ballot(sel.valid);
  const int lane = (int)(threadIdx.x & 63);
  while (m != 0) {
    const int src = __ffsll((long long)m) - 1;
    const int element = __builtin_amdgcn_readlane(sel.element, src), lowerion = __builtin_amdgcn_readlane(sel.lowerion, src);
    const int lower = __builtin_amdgcn_readlane(sel.lower, src), t = __builtin_amdgcn_readlane(sel.t, src);
    const float T_e = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sel.T_e), src));
    const double zrand = wave_bcast(sel.zrand, src);
    const double nu = select_continuum_nu_wave(env.M, element, lowerion, lower, t, T_e, zrand);
    if (lane == src) sel.nu = nu;
    m &= m - 1;
  }
  sel.mode = 2;
}

// The continuum opacity of ONE packet evaluated by its WAVE (the tail kernel: a packet per wave, 63 idle lanes; round 5: an r-packet step of the
// nltenebular tail is 269 000 clocks, most of them this sum by one lane). chi_bf_gammacontr<false>() with the kept continua of the window side by
// side -- lane j the (base + j)-th: the same reads and the same arithmetic as add_term() -- and then what the sequential loop does with a
// continuum's contribution IN ITS ORDER: the packet's lane adds the terms one after the other (rpkt.cc:808: the same additions in the same order,
// the same bits) and keeps the ground continua's list. All lanes call with lane 0's arguments (owner: lane 0 holds the packet).
__device__ inline double chi_bf_gammacontr_wave(const Env &env, int c, double nu, int64_t slot, Chi *keep, bool owner) {
  const DevModel &M = env.M;
  const int lane = (int)(threadIdx.x & 63);
  double sum = 0.;
  int ng = 0;
#if ARTIS_OPT_USE_LUT_PHOTOION || ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS
  int lastgi = -1;
  double *wsv = env.gamma_ws + (slot * M.nbfcontinua_ground);
  int32_t *wsi = env.gamma_gi + (slot * M.nbfcontinua_ground);
#endif
  const float T_e = env.C.Te[c];
  const double ex = exp(-HOVERKB * nu / T_e);
  const bool split_usable = (ex >= DBLMIN);
  const int cend = upper_bound_d(M.allcont_nu_edge, M.nbfcontinua, nu);
  const int cbegin = lower_bound_d(M.allcont_nu_edge, cend, nu / M.last_phixs_nuovernuedge);
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
  if (owner && keep != nullptr) {
    keep->bf_begin = cbegin;
    keep->bf_end = cend;
  }
#else
  (void)keep;
#endif
  const double *departure = env.K.allcont_departure + (krow(env, c) * M.nbfcontinua);
  int r0 = 0, r1 = 0;
  if (cbegin < cend) kept_range(env, c, cbegin, cend, r0, r1);
  const int32_t *keptlist = env.K.allcont_keptlist + (krow(env, c) * M.nbfcontinua);
  const D2 *keptpair = env.K.allcont_keptpair + (krow(env, c) * M.nbfcontinua);
  for (int base = r0; base < r1; base += 64) {
    const int n = (r1 - base < 64) ? r1 - base : 64;
    double sigma_contr = 0., term = 0.;
    int gi = -1;
    if (lane < n) {
      const int i = keptlist[base + lane];
      const D2 pr = keptpair[base + lane];
      const ContPack cpk = M.cont_pack[i];
      const PhixsRead xrk = phixs_lookup_u(M, M.allphixs + cpk.xs_off, cpk.nu_edge, nu);
      const double nu_edge = cpk.nu_edge;
      const double sigma_bf = phixs_finish(M, xrk, nu_edge, nu);
      double stim;
      if (pr.y >= 0. && split_usable) {
        stim = pr.y * ex;
      } else {
        stim = departure[i] * exp(-HOVERKB * (nu - nu_edge) / T_e);
      }
      const double corr = dmax(0., 1 - stim);
      sigma_contr = sigma_bf * cpk.probability * corr;
      term = pr.x * sigma_contr;
      gi = cpk.gi;
    }
    for (int k = 0; k < n; k++) {  // (k is wave-uniform: the lane index of a v_readlane)
      const double tk = wave_bcast(term, k);
#if ARTIS_OPT_USE_LUT_PHOTOION || ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS
      const double sk = wave_bcast(sigma_contr, k);
      const int gk = __builtin_amdgcn_readlane(gi, k);
      if (owner && gk >= 0) {
        if (gk == lastgi) {
          wsv[ng - 1] = sk;
        } else {
          wsv[ng] = sk;
          wsi[ng] = gk;
          ng++;
          lastgi = gk;
        }
      }
#endif
      sum += tk;
    }
  }
  if (owner) {
    env.gamma_n[slot] = ng;
    ARTIS_STAT(env, ARTIS_STAT_X_CHI_EVALS);
    ARTIS_STAT_ADD(env, ARTIS_STAT_X_CONT_VISITED, r1 - r0);
    if (!isfinite(sum)) fail(env, 30);
  }
  return sum;
}
// whether the coming do_rpkt_step() of this r-packet evaluates the continuum opacity (its own conditions, in its order: a step that begins on a
// cell boundary changes cell and returns; an empty or grey cell has no continuum opacity; chi_rpkt_cont()'s cache test) -- and in which cell
AHD bool rpkt_step_evaluates_chi(const Env &env, const Pkt &p, const Chi &x, int *c_out) {
  const int c = env.M.propcell_nonemptymgi[p.cellindex];
  *c_out = c;
  if (c < 0 || env.C.thick[c] == ARTIS_CELL_THICK) return false;
  int next_cell = -1;
  if (boundary_distance(env, p, &next_cell) == 0) return false;
  return !((c == x.nonemptymgi) && (fabs((x.nu / p.nu_cmf) - 1.0) < 1e-4));
}
// chi_rpkt_cont() by the wave, for lane 0's packet; leaves x as chi_rpkt_cont() would (the step's own call then finds it evaluated)
__device__ inline void chi_rpkt_cont_wave(const Env &env, double nu_cmf, Chi &x, int c, int64_t slot, bool owner) {
  const double chi_bf = chi_bf_gammacontr_wave(env, c, nu_cmf, slot, &x, owner);
  if (owner) {
    const float nne = env.C.nne[c];
    const float cnne = env.C.nne[c] * env.C.clumpfactor[c];
    const float T_e = env.C.Te[c];
    x.chi_freefree_heat = env.K.chi_ff_nnionpart[krow(env, c)] / pow3(nu_cmf) * cnne * (1 - exp(-HOVERKB * nu_cmf / T_e));
    x.chi_escatter = SIGMA_T * nne;
    x.chi_boundfree = chi_bf;
    x.nonemptymgi = c;
    x.nu = nu_cmf;
  }
}
#endif


#if ARTIS_OPT_VPKT_ON
// ---------------------------------------------------------------- vpkt.cc: virtual packets
constexpr double PARSEC = 3.0857e+18;  // constants.h:39
AHD int64_t logbinindex(double value, double minvalue, double dlog, int64_t nbins) {  // get_logbinindex sn3d.h:134
  const int64_t i = (int64_t)floor((log(value) - log(minvalue)) / dlog);
  return i < 0 ? 0 : (i > nbins - 1 ? nbins - 1 : i);
}
// add_to_vspecpol vpkt.cc:116
AHD void add_to_vspecpol(const Env &env, const VpktConfig &V, double nu_rf, double e_rf, double prob, double q_rf, double u_rf,
                         int obsdirindex, int opachoiceindex, double t_arrive) {
  if (t_arrive <= ARTIS_VSPEC_TIMEMIN || t_arrive >= ARTIS_VSPEC_TIMEMAX || nu_rf <= ARTIS_VSPEC_NUMIN || nu_rf >= ARTIS_VSPEC_NUMAX) return;
  const double dlogt = (log(ARTIS_VSPEC_TIMEMAX) - log(ARTIS_VSPEC_TIMEMIN)) / ARTIS_VSPEC_TIMEBINS;  // vpkt.cc:106-107
  const double dlognu = (log(ARTIS_VSPEC_NUMAX) - log(ARTIS_VSPEC_NUMIN)) / ARTIS_VSPEC_NUBINS;
  const int nt = (int)logbinindex(t_arrive, ARTIS_VSPEC_TIMEMIN, dlogt, ARTIS_VSPEC_TIMEBINS);
  const int nnu = (int)logbinindex(nu_rf, ARTIS_VSPEC_NUMIN, dlognu, ARTIS_VSPEC_NUBINS);
  const int ind_comb = (V.nspectraperobsdir * obsdirindex) + opachoiceindex;
  const double pktcontrib = e_rf / V.delta_t[nt] / V.delta_freq[nnu] / 4.e12 / PI / PARSEC / PARSEC / V.nprocs * 4 * PI;
  double *flux = env.E.vspecpol + ((((int64_t)nt * (V.nobsdirections * V.nspectraperobsdir) + ind_comb) * ARTIS_VSPEC_NUBINS + nnu) * 3);
  ARTIS_EST_ADD(&flux[0], prob * pktcontrib);
  ARTIS_EST_ADD(&flux[1], prob * q_rf * pktcontrib);
  ARTIS_EST_ADD(&flux[2], prob * u_rf * pktcontrib);
}
// add_to_vpkt_grid vpkt.cc:138
AHD void add_to_vpkt_grid(const Env &env, const VpktConfig &V, double nu_rf, double e_rf, double prob, double stokes_q, double stokes_u,
                          const double vel[3], int wlbin, int obsdirindex, const double obsdir[3]) {
  const double vmax = env.M.vmax;
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
  if (fabs(vref1) >= vmax || fabs(vref2) >= vmax) return;
  const int ny = (int)((vmax - vref1) / (2 * vmax / ARTIS_VGRID_NY));
  const int nz = (int)((vmax - vref2) / (2 * vmax / ARTIS_VGRID_NZ));
  if (nu_rf > V.nu_grid_min[wlbin] && nu_rf < V.nu_grid_max[wlbin]) {
    double *flux = env.E.vgrid_flux + (((((int64_t)ny * ARTIS_VGRID_NZ + nz) * V.grid_nwavelengthranges + wlbin) * V.nobsdirections + obsdirindex) * 3);
    ARTIS_EST_ADD(&flux[0], prob * e_rf);
    ARTIS_EST_ADD(&flux[1], prob * stokes_q * e_rf);
    ARTIS_EST_ADD(&flux[2], prob * stokes_u * e_rf);
  }
}
AHD bool all_taus_past_taumax(const double *tau, int n, double tau_max) {  // vpkt.cc:111
  for (int i = 0; i < n; i++)
    if (!(tau[i] > tau_max)) return false;
  return true;
}
// the loop of trace_lines_to_dist (vpkt.cc:298-358): false when every opacity choice is past tau_max. The populations
// are the cell cache's (B_lu n_l - B_ul n_u) of the grid state, scaled to the time the line is reached.
// NREG > 0: the optical depths and exclusions of at most NREG opacity choices, held in registers by the caller below (every loop over
// them unrolled and predicated); NREG == 0: any number, in memory.
template <int NREG>
AHD bool vpkt_trace_lines_core(const Env &env, int nspec, double tau_max, const int *excl, int c, double dist_limit, double t_future,
                               double nu_cmf, double dnu_on_dl, int &next_trans, double *tau_vpkt) {
  const DevModel &M = env.M;
  const LineDpop dpop = line_dpop_of(env, c);
  const double t_gridstate = env.S.mid;
  auto add_line = [&](int Z, double tau_line) -> bool {  // vpkt.cc:340-356; true: every choice is past tau_max
    if (NREG > 0) {
      bool all = true;
#pragma unroll
      for (int i = 0; i < NREG; i++) {
        if (excl[i] != -1 && excl[i] != Z) tau_vpkt[i] += tau_line;
        if (i < nspec && !(tau_vpkt[i] > tau_max)) all = false;
      }
      return all;
    }
    for (int i = 0; i < nspec; i++)
      if (excl[i] != -1 && excl[i] != Z) tau_vpkt[i] += tau_line;
    return all_taus_past_taumax(tau_vpkt, nspec, tau_max);
  };
#if ARTIS_LINE_AHEAD > 1
  // (the lines' frequencies, population factors and elements ARTIS_LINE_AHEAD lines ahead, as in possible_event(): a ring of slots
  // that rotates by name)
  constexpr int NL = ARTIS_LINE_AHEAD;
  double nu_s[NL], dp_s[NL];
  int el_s[NL];
  int li = closest_transition(M.line_nu, M.nlines, nu_cmf, next_trans);
  if (li >= 0) {
#pragma unroll
    for (int d = 0; d < NL; d++) {
      const int l = (li + d < M.nlines) ? li + d : M.nlines - 1;
      nu_s[d] = M.line_nu[l];
      dp_s[d] = line_dpop_at(M, dpop, l);
      el_s[d] = M.line_elementindex[l];
    }
  }
  bool stop = false;
  while (!stop) {
#pragma unroll
    for (int u = 0; u < NL; u++) {
      if (li < 0) {
        next_trans = M.nlines + 1;
        stop = true;
        break;
      }
      const double nutrans = nu_s[u];
      const double dpop_li = dp_s[u];
      const int el = el_s[u];
      {
        const int l = (li + NL < M.nlines) ? li + NL : M.nlines - 1;
        nu_s[u] = M.line_nu[l];
        dp_s[u] = line_dpop_at(M, dpop, l);
        el_s[u] = M.line_elementindex[l];
      }
      next_trans = li + 1;
      const double ldist = linedistance(t_future, nu_cmf, nutrans, dnu_on_dl);
      if (ldist > dist_limit) {
        next_trans--;
        stop = true;
        break;
      }
      const double t_line = t_future + (ldist / CLIGHT_PROP);
      const double popscalefactor = pow3(t_gridstate / t_line);
      const double tau_line = dmax(0., dpop_li * popscalefactor * HCLIGHTOVERFOURPI * t_line);
      if (add_line(M.elem_anumber[el], tau_line)) return false;
      li = closest_transition(M.line_nu, M.nlines, nu_cmf, next_trans);
    }
  }
#else
  while (true) {
    const int li = closest_transition(M.line_nu, M.nlines, nu_cmf, next_trans);
    if (li < 0) {
      next_trans = M.nlines + 1;
      break;
    }
    const double nutrans = M.line_nu[li];
    next_trans = li + 1;
    const double ldist = linedistance(t_future, nu_cmf, nutrans, dnu_on_dl);
    if (ldist > dist_limit) {
      next_trans--;
      break;
    }
    const double t_line = t_future + (ldist / CLIGHT_PROP);
    const double popscalefactor = pow3(t_gridstate / t_line);
    const double tau_line = dmax(0., line_dpop_at(M, dpop, li) * popscalefactor * HCLIGHTOVERFOURPI * t_line);
    if (add_line(M.elem_anumber[M.line_elementindex[li]], tau_line)) return false;
  }
#endif
  return true;
}
// ... with up to four opacity choices (the reference's example configurations have one to four) the optical depths stay in registers for
// the walk: per line visited the memory form reads and writes every choice's sum and reads its exclusion (round 4)
AHD bool vpkt_trace_lines_to_dist(const Env &env, const VpktConfig &V, int c, double dist_limit, double t_future, double nu_cmf,
                                  double dnu_on_dl, int &next_trans, double *tau_vpkt) {
  constexpr int NREG = 4;
  const int nspec = V.nspectraperobsdir;
  if (nspec > NREG) return vpkt_trace_lines_core<0>(env, nspec, V.tau_max, V.opacityexclusions, c, dist_limit, t_future, nu_cmf, dnu_on_dl, next_trans, tau_vpkt);
  double t[NREG];
  int ex[NREG];
#pragma unroll
  for (int i = 0; i < NREG; i++) {
    t[i] = (i < nspec) ? tau_vpkt[i] : 0.;
    ex[i] = (i < nspec) ? V.opacityexclusions[i] : -1;  // (-1: no line is ever added)
  }
  const bool go = vpkt_trace_lines_core<NREG>(env, nspec, V.tau_max, ex, c, dist_limit, t_future, nu_cmf, dnu_on_dl, next_trans, t);
#pragma unroll
  for (int i = 0; i < NREG; i++)
    if (i < nspec) tau_vpkt[i] = t[i];
  return go;
}
// trace_vpkt_direction vpkt.cc:183, in three parts so that a kernel can give a lane its next ray as soon as its ray has ended (k_vpkt: a
// ray crosses 1 ... 50 cells, and with one ray per lane from start to end a wave lasts as long as its longest): vray_begin() =
// everything before the loop, vray_step() = one turn of the loop (one cell), vray_finish() = the contributions to the spectra.
struct VRay {
  Pkt v;  // position, direction, time and cell of the virtual packet (what boundary_distance() reads)
  double vel_vec[3];  // velocity of the matter at the event (get_velocity vectors.h:50)
  int type_before;    // ... and what the real packet was there
  double obsdir[3];
  double t_arrive, nu_rf, e_rf, e_cmf, nu_cmf, pn, q_rf, u_rf;
  double tau_vpkt[VPKT_MAXSPEC];
  Chi x;  // the virtual packet's own ContinuumOpacity: fresh per traced direction
  int next_trans, mgi, obsdirindex;
};
AHD void vray_begin(const Env &env, const VpktSeed &r, double t_arrive, double nu_rf, double e_rf, double rpkt_doppler, int obsdirindex,
                    const double obsdir_in[3], VRay &y) {
  const DevModel &M = env.M;
  y.type_before = r.type_before;
  y.obsdirindex = obsdirindex;
  y.t_arrive = t_arrive;
  y.nu_rf = nu_rf;
  y.e_rf = e_rf;
  for (int d = 0; d < 3; d++) y.obsdir[d] = obsdir_in[d];
  const double *obsdir = y.obsdir;
  Pkt &v = y.v;
  v.px = r.pos[0]; v.py = r.pos[1]; v.pz = r.pos[2];
  v.dx = obsdir[0]; v.dy = obsdir[1]; v.dz = obsdir[2];
  v.cellindex = r.cellindex;
  v.prop_time = r.prop_time;
  y.next_trans = r.next_trans;
  y.e_cmf = r.e_cmf;
  y.nu_cmf = r.nu_cmf;
  const double t_start = r.prop_time;
  for (int i = 0; i < VPKT_MAXSPEC; i++) y.tau_vpkt[i] = 0.;
  ARTIS_STAT(env, ARTIS_STAT_X_VPKT_CREATED);
  const double vel_vec[3] = {r.pos[0] / t_start, r.pos[1] / t_start, r.pos[2] / t_start};  // get_velocity vectors.h:50
  for (int d = 0; d < 3; d++) y.vel_vec[d] = vel_vec[d];
  double pn = 1 / (4 * PI);
  double q_rf = 0., u_rf = 0.;
  if (r.type_before == ARTIS_TYPE_RPKT) {
    double old_dir_cmf[3], q_i_cmf = 0., u_i_cmf = 0.;
    frame_transform(r.dir, r.stokes_q, r.stokes_u, vel_vec, old_dir_cmf, &q_i_cmf, &u_i_cmf);
    double obs_cmf[3], new_dir_rf[3];
    angle_ab(obsdir, vel_vec, obs_cmf);
    scatter_polarisation_to_rf(old_dir_cmf, obs_cmf, q_i_cmf, u_i_cmf, vel_vec, new_dir_rf, &q_rf, &u_rf);
    // pn of scatter_polarisation_to_rf (vectors.h:353): the phase function of the scattering
    double ref1[3], ref2[3];
    meridian(old_dir_cmf, ref1, ref2);
    const double i1 = rot_angle(old_dir_cmf, obs_cmf, ref1, ref2);
    double s2i, c2i;
    sin_cos(2 * i1, &s2i, &c2i);
    const double q_old = (q_i_cmf * c2i) - (u_i_cmf * s2i);
    const double musquared = pow2(vdot(old_dir_cmf, obs_cmf));
    pn = 3. / (16. * PI) * (1. + musquared + ((musquared - 1.) * q_old));
  }
  pn /= pow2(rpkt_doppler);
  y.pn = pn;
  y.q_rf = q_rf;
  y.u_rf = u_rf;
  y.mgi = M.propcell_nonemptymgi[v.cellindex];
  y.x.nonemptymgi = -1;
  y.x.nu = NAN;
  y.x.chi_escatter = y.x.chi_freefree_heat = y.x.chi_boundfree = 0.;
}
// the contributions of a ray that has left the grid (vpkt.cc:395-420)
AHD void vray_finish(const Env &env, VRay &y) {
  const VpktConfig &V = *env.M.vpkt;
  const int nspec = V.nspectraperobsdir;
  const double *obsdir = y.obsdir;
  const double *tau_vpkt = y.tau_vpkt;
  const double pn = y.pn, q_rf = y.q_rf, u_rf = y.u_rf, nu_rf = y.nu_rf, e_rf = y.e_rf, t_arrive = y.t_arrive;
  const int obsdirindex = y.obsdirindex;
  const double *vel_vec = y.vel_vec;
  ARTIS_STAT(env, y.type_before == ARTIS_TYPE_RPKT ? ARTIS_STAT_X_VPKT_ESC_RPKT
                                                   : (y.type_before == ARTIS_TYPE_KPKT ? ARTIS_STAT_X_VPKT_ESC_KPKT : ARTIS_STAT_X_VPKT_ESC_MA));
  for (int i = 0; i < nspec; i++) {
    const double prob = pn * exp(-tau_vpkt[i]);
    if (!isfinite(prob)) fail(env, 91);
    add_to_vspecpol(env, V, nu_rf, e_rf, prob, q_rf, u_rf, obsdirindex, i, t_arrive);
  }
  if (V.vgrid_on) {
    const double prob = pn * exp(-tau_vpkt[0]);
    for (int wlbin = 0; wlbin < V.grid_nwavelengthranges; wlbin++)
      if ((nu_rf > V.nu_grid_min[wlbin] && nu_rf < V.nu_grid_max[wlbin]) && (t_arrive > V.tmin_grid && t_arrive < V.tmax_grid))
        add_to_vpkt_grid(env, V, nu_rf, e_rf, prob, q_rf, u_rf, vel_vec, wlbin, obsdirindex, obsdir);
  }
}
// one turn of the loop of trace_vpkt_direction() (vpkt.cc:240-393): the cell the ray is in. Returns true while the ray goes on; a ray that
// leaves the grid adds its contributions (vray_finish) before it returns false, one that is absorbed or enters a thick cell just ends.
AHD bool vray_step(const Env &env, VRay &y) {
  const DevModel &M = env.M;
  const VpktConfig &V = *M.vpkt;
  const int nspec = V.nspectraperobsdir;
  Pkt &v = y.v;
  Chi &x = y.x;
  const double *obsdir = y.obsdir;
  double *tau_vpkt = y.tau_vpkt;
  int &next_trans = y.next_trans;
  int &mgi = y.mgi;
  double &e_cmf = y.e_cmf, &nu_cmf = y.nu_cmf;
  const double nu_rf = y.nu_rf, e_rf = y.e_rf;
  const double t_gridstate = env.S.mid;
  bool end_packet = false;
    int next_cellindex = -1;
    const double boundarydist = boundary_distance(env, v, &next_cellindex);
    if (mgi < 0) {
      next_trans = -1;
    } else if (boundarydist > 0) {
      const int c = mgi;
      const double dop = doppler(v);
      // calculate_chi_rpkt_cont<false> rpkt.cc:1021: the same sums without the packet's phixslist
      if (!((c == x.nonemptymgi) && (fabs((x.nu / nu_cmf) - 1.0) < 1e-4))) {
        const float cnne = env.C.nne[c] * env.C.clumpfactor[c];
        const float T_e = env.C.Te[c];
        x.chi_freefree_heat = env.K.chi_ff_nnionpart[krow(env, c)] / pow3(nu_cmf) * cnne * (1 - exp(-HOVERKB * nu_cmf / T_e));
        x.chi_escatter = SIGMA_T * env.C.nne[c];
        int dummy = -1;
        x.chi_boundfree = chi_bf_gammacontr<true>(env, c, nu_cmf, 0, DBLMAX, &dummy);
        x.nonemptymgi = c;
        x.nu = nu_cmf;
      }
      const double densityscalefactor = pow3(t_gridstate / v.prop_time);
      const double chi_escatter = x.chi_escatter * densityscalefactor;
      const double chi_bf = x.chi_boundfree * densityscalefactor;
      const double chi_ff = x.chi_freefree_heat * pow2(densityscalefactor);
      const double chi_cont = chi_escatter + chi_bf + chi_ff;
      for (int i = 0; i < nspec; i++) {
        double chi_cont_thischoice = chi_cont;
        if (V.opacityexclusions[i] == -2) {
          chi_cont_thischoice -= chi_bf;
        } else if (V.opacityexclusions[i] == -3) {
          chi_cont_thischoice -= chi_ff;
        } else if (V.opacityexclusions[i] == -4) {
          chi_cont_thischoice -= chi_escatter;
        }
        tau_vpkt[i] += chi_cont_thischoice * boundarydist * dop;
      }
      if (all_taus_past_taumax(tau_vpkt, nspec, V.tau_max)) return false;
      const double nu_cmf_boundary =
          dmin(nu_rf * doppler_at(v.px + (obsdir[0] * boundarydist), v.py + (obsdir[1] * boundarydist), v.pz + (obsdir[2] * boundarydist),
                                  obsdir[0], obsdir[1], obsdir[2], v.prop_time + (boundarydist / CLIGHT_PROP)),
               nu_cmf);
      const double dnu_on_dl = (nu_cmf_boundary - nu_cmf) / boundarydist;
#if ARTIS_OPT_VPKT_USE_EXPANSION_OPACITIES
      int64_t binindex_start = linearbinindex(1e8 * CLIGHT / nu_cmf, ARTIS_EXPOPAC_LAMBDAMIN, ARTIS_EXPOPAC_DELTALAMBDA);
      if (binindex_start < -1) binindex_start = -1;
      if (binindex_start < ARTIS_EXPOPAC_NBINS) {
        const double first_bin_edge_nu = (binindex_start < 0) ? expopac_bin_nu_upper(0) : expopac_bin_nu_lower(binindex_start);
        const double first_bin_edge_dist = linedistance(v.prop_time, nu_cmf, first_bin_edge_nu, dnu_on_dl);
        const double line_by_line_limit = dmin(first_bin_edge_dist, boundarydist);
        next_trans = -1;
        if (!vpkt_trace_lines_to_dist(env, V, c, line_by_line_limit, v.prop_time, nu_cmf, dnu_on_dl, next_trans, tau_vpkt)) return false;
        double dist = line_by_line_limit;
        if (dist < boundarydist) {
          const float *kappa_bins = env.C.expansionopacities + ((int64_t)c * ARTIS_EXPOPAC_NBINS);
          const float rho = env.C.rho[c];
          for (int64_t binindex = binindex_start + 1; binindex < ARTIS_EXPOPAC_NBINS; binindex++) {
            const double next_bin_edge_nu = expopac_bin_nu_lower(binindex);
            const double binedgedist = linedistance(v.prop_time, nu_cmf, next_bin_edge_nu, dnu_on_dl);
            const double chi_bb_expansionopac = kappa_bins[binindex] * rho * densityscalefactor;  // float product first
            const double tau_bin = chi_bb_expansionopac * (dmin(binedgedist, boundarydist) - dist);
            dist = dmin(binedgedist, boundarydist);
            for (int i = 0; i < nspec; i++)
              if (V.opacityexclusions[i] != -1) tau_vpkt[i] += tau_bin;
            if (all_taus_past_taumax(tau_vpkt, nspec, V.tau_max)) return false;
            if (dist >= boundarydist) break;
          }
        }
      }
#else
      if (!vpkt_trace_lines_to_dist(env, V, c, boundarydist, v.prop_time, nu_cmf, dnu_on_dl, next_trans, tau_vpkt)) return false;
#endif
    }
    move_raw(v.px, v.py, v.pz, obsdir[0], obsdir[1], obsdir[2], v.prop_time, nu_rf, nu_cmf, e_rf, e_cmf, boundarydist);
    if (next_cellindex >= 0) {
      if (next_cellindex != v.cellindex && M.gridtype == ARTIS_GRID_CARTESIAN3D) {  // snap_pos_to_cell grid.cc:2460
        double *pos[3] = {&v.px, &v.py, &v.pz};
        for (int d = 0; d < 3; d++) {
          const int idx = coordidx(M, next_cellindex, d);
          const double lo = M.coord_pos_min_tmin[d][idx] / M.tmin * v.prop_time;
          const double hi = (idx < (M.ncoordgrid[d] - 1)) ? M.coord_pos_min_tmin[d][idx + 1] / M.tmin * v.prop_time
                                                           : M.rmax / M.tmin * v.prop_time;
          *pos[d] = dclamp(*pos[d], lo, hi);
        }
      }
      v.cellindex = next_cellindex;
      mgi = M.propcell_nonemptymgi[v.cellindex];
      if (mgi >= 0 && env.C.thick[mgi] != ARTIS_CELL_THIN) return false;
    } else {
      end_packet = true;
    }
  if (end_packet) {
    vray_finish(env, y);
    return false;
  }
  return true;
}
AHD bool trace_vpkt_direction(const Env &env, const VpktSeed &r, double t_arrive, double nu_rf, double e_rf, double rpkt_doppler,
                              int obsdirindex, const double obsdir[3]) {
  VRay y;
  vray_begin(env, r, t_arrive, nu_rf, e_rf, rpkt_doppler, obsdirindex, obsdir, y);
  while (vray_step(env, y)) {
  }
  return true;
}
// the body of trace_vpkts()'s loop (vpkt.cc:962-991) for one observer direction
AHD void vpkt_trace_seed_direction(const Env &env, const VpktSeed &r, int obsdirindex) {
  const VpktConfig &V = *env.M.vpkt;
  const double obsdir[3] = {V.obsdir[obsdirindex][0], V.obsdir[obsdirindex][1], V.obsdir[obsdirindex][2]};
  const double t_arrive = r.prop_time - (((r.pos[0] * obsdir[0]) + (r.pos[1] * obsdir[1]) + (r.pos[2] * obsdir[2])) / CLIGHT_PROP);
  if (t_arrive >= V.timemin_input && t_arrive <= V.timemax_input) {
    const double dop = doppler_at(r.pos[0], r.pos[1], r.pos[2], obsdir[0], obsdir[1], obsdir[2], r.prop_time);
    const double nu_rf = r.nu_cmf / dop;
    const double e_rf = r.e_cmf / dop;
    for (int i = 0; i < V.nwavelengthranges; i++) {
      if ((nu_rf > V.numin_input[i] && nu_rf < V.numax_input[i]) || (r.absorptionfreq > V.numin_input[i] && r.absorptionfreq < V.numax_input[i])) {
        (void)trace_vpkt_direction(env, r, t_arrive, nu_rf, e_rf, dop, obsdirindex, obsdir);
        break;
      }
    }
  }
}
// ... as the start of a ray for k_vpkt: false when the direction is not traced (outside the time or frequency windows)
AHD bool vray_begin_seed(const Env &env, const VpktSeed &r, int obsdirindex, VRay &y) {
  const VpktConfig &V = *env.M.vpkt;
  const double obsdir[3] = {V.obsdir[obsdirindex][0], V.obsdir[obsdirindex][1], V.obsdir[obsdirindex][2]};
  const double t_arrive = r.prop_time - (((r.pos[0] * obsdir[0]) + (r.pos[1] * obsdir[1]) + (r.pos[2] * obsdir[2])) / CLIGHT_PROP);
  if (!(t_arrive >= V.timemin_input && t_arrive <= V.timemax_input)) return false;
  const double dop = doppler_at(r.pos[0], r.pos[1], r.pos[2], obsdir[0], obsdir[1], obsdir[2], r.prop_time);
  const double nu_rf = r.nu_cmf / dop;
  const double e_rf = r.e_cmf / dop;
  for (int i = 0; i < V.nwavelengthranges; i++) {
    if ((nu_rf > V.numin_input[i] && nu_rf < V.numax_input[i]) || (r.absorptionfreq > V.numin_input[i] && r.absorptionfreq < V.numax_input[i])) {
      vray_begin(env, r, t_arrive, nu_rf, e_rf, dop, obsdirindex, obsdir, y);
      return true;
    }
  }
  return false;
}
// trace_vpkts vpkt.cc:948: called where a real packet is emitted or scattered by an electron. On the GPU the event is
// recorded for k_vpkt; in the test emulation it is traced in place.
AHD void trace_vpkts(const Env &env, const Pkt &p, int64_t pi, int type_before) {
  const int c = env.M.propcell_nonemptymgi[p.cellindex];
  if (env.C.thick[c] != ARTIS_CELL_THIN) return;
  VpktSeed r;
  r.pos[0] = p.px; r.pos[1] = p.py; r.pos[2] = p.pz;
  r.dir[0] = p.dx; r.dir[1] = p.dy; r.dir[2] = p.dz;
  r.nu_cmf = p.nu_cmf; r.e_cmf = p.e_cmf; r.prop_time = p.prop_time;
  r.stokes_q = p.stokes_q; r.stokes_u = p.stokes_u;
  r.absorptionfreq = env.P.flight[pi].absorptionfreq;
  r.cellindex = p.cellindex; r.next_trans = p.next_trans; r.type_before = type_before; r.pad = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  const int idx = atomicAdd(env.vpkt_count, 1);  // (the propagation kernels never carry the trace itself)
  if (idx < env.vpkt_cap) env.vpkt_queue[idx] = r; else fail(env, 92);
#else
  for (int o = 0; o < env.M.vpkt->nobsdirections; o++) vpkt_trace_seed_direction(env, r, o);
#endif
}
#endif

// ---------------------------------------------------------------- do_macroatom macroatom.cc:360
// The reference runs a macro-atom to deactivation inside do_macroatom(). Here the activation only records
// the state in the packet (ma_activate) and the walk is advanced one transition at a time by ma_jump(), so
// that the thermal kernel can interleave packets at any point of their walk. The sequence of operations on a
// packet (and so its random numbers) is the reference's.
AHD void ma_activate(Pkt &p, const MAState &ma, int origin_rpkt) {
  p.ma_element = ma.element;
  p.ma_ion = ma.ion;
  p.ma_level = ma.level;
  p.ma_line = ma.activatingline;
  p.ma_origin = origin_rpkt;
}
AHD bool ma_pending(const Pkt &p) { return p.ma_level >= 0; }

// end of do_macroatom(), macroatom.cc:579-595
AHD void ma_finish(const Env &env, Pkt &p, int64_t pi) {
  p.ma_level = -1;
  if (p.type == ARTIS_TYPE_RPKT) {
    if (p.trueemissiontype == ARTIS_EMTYPE_NOTSET) {
      p.trueemissiontype = p.emissiontype;
      set_trueem_from_em(env, p, pi);
    }
#if ARTIS_OPT_VPKT_ON
    trace_vpkts(env, p, pi, ARTIS_TYPE_MA);  // macroatom.cc:588
#endif
  } else {
    p.trueemissiontype = ARTIS_EMTYPE_NOTSET;
  }
}

// per-launch invariants of a thermal packet (it never changes cell while thermal) and the static indices of the level
// its macro-atom is in
struct MACtx {
  int c;                    // non-empty model cell
  bool thick;               // the cell is optically thick (grey): its k-packets go to do_kpkt_blackbody()
  const U4 *cellma;         // the cell's row of macro-atom records
  int start_key, start;     // cached get_ionuniquelevelindexstart(element, ion)
  int rec;                  // the current level's record (ma_prepare, then carried by the walk): its slot in the cell's row, or -- a cold level -- its place in the pool / MA_REC_NONE (ma_resolve)
  int nd, nu;               // ... its numbers of downward / upward transitions (where the record's lines are)
  int ats;                  // ... and its first entry in alltrans (where its transitions' targets are)
  int njumps;               // transitions made since the last ma_flush_stats()
  int defer;                // ma_jump_internal<true>: the draw of a transition left to ma_jump_deferred() (bit 24: downward)
};
AHD MACtx ma_ctx(const Env &env, const Pkt &p) {
  MACtx k;
  k.c = env.M.propcell_nonemptymgi[p.cellindex];
  k.thick = (k.c >= 0) && (env.C.thick[k.c] == ARTIS_CELL_THICK);
  k.cellma = env.K.macache + (krow(env, k.c) * env.M.nmacache);
  k.start_key = -1;
  k.start = 0;
  k.rec = 0;
  k.nd = k.nu = 0;
  k.ats = 0;
  k.njumps = 0;
  k.defer = 0;
  return k;
}

// The uint16 filters of a record (tables.h "FILTERS"). q = floor(value / whole * 32768), clamped to 32767.
AHD uint32_t mafilt_quant(double value, double whole, bool *ok) {
  const double f = (value / whole) * MAFILT_SCALE;
  if (!(f >= 0. && f <= MAFILT_SCALE)) {
    *ok = false;
    return 0;
  }
  return (f >= MAFILT_SCALE - 1.) ? MAFILT_NONE : (uint32_t)f;
}
// ... with 8 more bits (tables.h "FINE BYTES"): q23 = floor(value / whole * 2^23), clamped to 2^23 - 1 (value == whole: "value <= z * whole" never
// holds, z < 1: never counted). q23 >> 8 is mafilt_quant()'s value always (f * 256 is exact; f >= 32767 gives 0x7FFF there and here).
AHD uint32_t mafilt_quant23(double value, double whole, bool *ok) {
  const double f = (value / whole) * MAFILT_SCALE;
  if (!(f >= 0. && f <= MAFILT_SCALE)) {
    *ok = false;
    return 0;
  }
  const double f23 = f * 256.;
  return (f23 >= 8388607.) ? MAFILT_NONE23 : (uint32_t)f23;
}
// The decision of one line on its 23-bit fractions (the line's entries and its fine bytes) for the 24-bit draw u (z = u * 2^-24):
// q23 <= F * 2^23 < q23 + 1 for the computed fraction F = fl(value / whole) (within 2^-53 of the exact one); z * 2^23 = u / 2. u >= 2 q23 + 3:
// z * 2^23 >= q23 + 1.5 > F * 2^23 + 0.5, so F < z - 2^-24: "value <= fl(z * whole)" holds with a margin of 6e-8 of the whole against roundings of
// 2e-16. u <= 2 q23 - 1: z * 2^23 <= q23 - 0.5 <= F * 2^23 - 0.5: it fails by the same margin. u in {2 q23, 2 q23 + 1, 2 q23 + 2}: undecided (*amb).
// Returns the number of entries certainly <= z (entries are non-decreasing; a padding entry, 0x7FFFFF, is never counted).
AHD int mafilt_count_fine(const U4 &f, uint64_t fine, uint32_t u, bool *amb) {
  int lo = 0, hi = 0;
#pragma unroll
  for (int j = 0; j < MAREC_PER; j++) {
    const uint32_t q15 = (f.w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
    const uint32_t q2 = (((q15 << 8) | (uint32_t)((fine >> (8 * j)) & 0xFFu)) << 1);
    lo += (u >= q2 + 3u) ? 1 : 0;
    hi += (u >= q2) ? 1 : 0;
  }
  *amb = (lo != hi);
  return lo;
}
// how many of the filter's 8 entries have q <= lo, and whether more have q <= hi (*amb): lo >= -1, hi <= 32767. Two entries
// per 32-bit word and subtraction: with h = bound + 32768 in both halves, (h - q) has bit 15 set in a half iff q <= bound
// there, and no half borrows from the other (q <= 32767 <= h).
AHD int mafilt_count_between(const U4 &f, int lo, int hi, bool *amb) {
  const uint32_t hl = (uint32_t)(lo + 32768) * 0x10001u, hh = (uint32_t)(hi + 32768) * 0x10001u;
  int c1 = 0, c2 = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    c1 += __builtin_popcount((hl - f.w[j]) & 0x80008000u);
    c2 += __builtin_popcount((hh - f.w[j]) & 0x80008000u);
  }
  *amb = (c1 != c2);
  return c1;
}
// how many of the filter's 8 entries are certainly <= z. z = u * 2^-24 (a 24-bit draw), zi = u >> 9 = floor(z * 32768), so
// zi <= z * 32768 <= zi + 1 - 2^-9. With q <= fraction * 32768 <= q + 1: zi >= q + 2 proves fraction < z by at least 3e-5,
// and zi <= q - 1 proves fraction > z by at least 2^-9 / 32768 = 6e-8 -- both far beyond the 1e-16 by which the f64
// comparison "value <= z * whole" can differ from the exact one. *amb: some entry has q == zi or q == zi - 1. Two counts
// instead of two tests per entry: the entries with q <= zi - 2 are counted, and the filter is ambiguous iff more entries
// have q <= zi. An entry that is not to be counted at all holds 0x7FFF (it can only turn a draw with zi = 32767 ambiguous).
// Round 5: the draw's nine bits below zi decide half of those. z * 32768 = zi + sub / 512 (sub = u & 511), so an entry with q == zi - 1
// has fraction * 32768 < zi <= z * 32768 - sub / 512: with sub >= 2 the fraction lies below z by at least 2^-23 = 1.2e-7 of the whole -- a
// margin of the size the other side's proof (6e-8) already relies on -- and the entry is counted. The lower bound of the count is therefore
// ((u - 2) >> 9) - 1: zi - 1 where sub >= 2, zi - 2 otherwise. Undecided draws fall from ~2n to ~n of 32768 for a direction of n transitions
// (with 4e5 lines 83 % of the slow path's visits were such searches: profiles/r05/slow_path_cd23like.txt).
// Round 6: the entries of a filter are SORTED (cumulative sums of non-negative terms, quantised by a monotonic rule; padding is the largest
// value; populate_mafilter_level() refuses an action filter that is not), so the count of entries <= zi is a three-step binary search over
// e0 .. e6 (+ one look at e7 where all seven are counted: the action filter's eighth entry; a direction line's mark, 0x7FFF, is counted only by
// zi = 32767) -- and the draw is undecided iff the LARGEST counted entry lies above the lower bound: one count and one comparison instead of two
// counts (27 -> ~17 instructions of the ~200 a wave-round issues). Same result as mafilt_count_between(f, lo, zi) on sorted entries (property
// tests, tests/hostemu artis_emu_mafilter_selftest).
AHD int mafilt_count(const U4 &f, uint32_t u, bool *amb) {
  const uint32_t zi = u >> 9;
  int lo = (((int)u - 2) >> 9) - 1;
  lo = (lo > -1) ? lo : -1;
  const uint32_t e3 = f.w[1] >> 16;
  const bool b2 = e3 <= zi;
  const uint32_t x = b2 ? f.w[2] : f.w[0];
  const uint32_t k2 = x >> 16;  // e5 or e1
  const bool b1 = k2 <= zi;
  const uint32_t ya = b1 ? f.w[1] : f.w[0], yb = b1 ? f.w[3] : f.w[2];
  const uint32_t k3 = (b2 ? yb : ya) & 0xFFFFu;  // e6 / e4 / e2 / e0
  const bool b0 = k3 <= zi;
  int c = (b2 ? 4 : 0) + (b1 ? 2 : 0) + (b0 ? 1 : 0);
  int m = b0 ? (int)k3 : (b1 ? (int)k2 : (b2 ? (int)e3 : -1));  // the largest counted entry
  const uint32_t e7 = f.w[3] >> 16;
  const bool b3 = (c == 7) && (e7 <= zi);
  c += b3 ? 1 : 0;
  m = b3 ? (int)e7 : m;
  *amb = m > lo;
  return c;
}
// After every rate of a cell's records is final (populate_macroatom): the action filter of one level's record
AHD void populate_mafilter_level(const Env &env, int c, int ul) {
  const DevModel &M = env.M;
  const LevelPack lpk = M.level_pack[ul];
  U4 *rec = ma_rec_of(env, c, lpk);
  const double *rates = ma_rates_of(rec, lpk.ndown, lpk.nup);
  // cumulative rates of the actions 0..7 over the total of all 9, summed as ma_load_rates() sums them
  double cum[MA_N];
  cum[0] = rates[0];
  for (int a = 1; a < MA_N; a++) cum[a] = cum[a - 1] + rates[a];
  const double total = cum[MA_N - 1];
  bool ok = (total > 0.) && (total <= DBLMAX);
  uint32_t q[8];
  for (int a = 0; a < 8; a++) q[a] = ok ? mafilt_quant(cum[a], total, &ok) : 0u;
  for (int a = 1; a < 8; a++)
    if (q[a] < q[a - 1]) ok = false;  // (a negative rate: never seen; the marker below needs q[0] <= q[7] otherwise)
  if (!ok) {  // marker: first entry above the last one (never so in a usable filter): decided on the f64 rates
    for (int a = 0; a < 8; a++) q[a] = 0u;
    q[0] = MAFILT_NONE;
  }
  U4 f;
  for (int j = 0; j < 4; j++) f.w[j] = q[2 * j] | (q[2 * j + 1] << 16);
  rec[0] = f;
}

// one iteration of the loop of do_macroatom(), macroatom.cc:385-577. ma_prepare() finds the record of the packet's
// current level (once per phase of the walk: afterwards every transition hands over the record of its target);
// ma_jump_internal() / ma_jump_exit() perform the transition from the record at k.rec.
AHD int ma_locate(const Env &env, const Pkt &p, MACtx &k) {
  const int key = (p.ma_element << 8) | p.ma_ion;
  if (key != k.start_key) {
    k.start = env.M.ion_uniquelevelindexstart[uion(env.M, p.ma_element, p.ma_ion)];
    k.start_key = key;
  }
  return k.start + p.ma_level;
}
template <bool COLD = true>
AHD void ma_prepare(const Env &env, const Pkt &p, MACtx &k) {
  const int ul = ma_locate(env, p, k);
  const LevelPack lp = env.M.level_pack[ul];
  k.rec = ma_resolve<COLD>(env, k.c, lp.rec_off);  // (MA_REC_NONE: a cold level without a record in this cell yet: ma_jump_internal() returns MA_EXIT_FILL)
  k.nd = lp.ndown;
  k.nu = lp.nup;
  k.ats = lp.alltrans_startdown;
}
template <bool COLD = true>
AHD const U4 *ma_record(const Env &env, const MACtx &k) {
  if (COLD && k.rec < 0) return env.K.ma_pool + ((int64_t)(-(k.rec + 2)) * MAPOOL_UNIT);  // (MA_REC_NONE: never read, ma_jump_internal() returns first)
  return k.cellma + k.rec;
}
// First half of a transition: draw the process (macroatom.cc:425-431); an internal transition inside the ion is made
// at once and -1 is returned. Every other process ends the walk in this kernel (deactivation, or a bound-free process
// for the slow path): its index is returned, and ma_jump_exit() carries it out. The split lets a kernel keep the rare,
// long deactivation code out of its transition loop.
constexpr int MA_EXIT_FAILED = 99;
// the 9 process rates of a record and their running sums (std::partial_sum macroatom.cc:425), kept in registers
AHD void ma_load_rates(const double *rates, double *r, double *cum) {
  const D2 q0 = *(const D2 *)(rates), q1 = *(const D2 *)(rates + 2), q2 = *(const D2 *)(rates + 4), q3 = *(const D2 *)(rates + 6);
  r[0] = q0.x; r[1] = q0.y; r[2] = q1.x; r[3] = q1.y; r[4] = q2.x; r[5] = q2.y; r[6] = q3.x; r[7] = q3.y;
  // MA_ACTION_INTERNALUPHIGHERNT: only NT_ON puts anything there (macroatom.cc:171); adding the 0 leaves cum[8] = cum[7]
  r[8] = ARTIS_OPT_NT_ON ? rates[8] : 0.;
  cum[0] = r[0];
#pragma unroll
  for (int i = 1; i < MA_N; i++) cum[i] = cum[i - 1] + r[i];
}
// A search the filter could not decide: how many of the direction's first nsearch cumulative sums are <= targetval, the sums
// RE-ADDED from the transitions' terms in the reference's order (macroatom.cc:64-140; the same matrans_terms() the
// population ran, added in the same order: the values the round-3 records held, bit for bit). Sums are non-decreasing, so
// the count is the index of the first sum above targetval. Out of line: rare (5e-4 per decision) and register-hungry.
AHD int ma_exact_search(const Env &env, int c, int ats0, int dir, int nsearch, double targetval) {
#if defined(ARTIS_MA_FAKE_EXACT) && defined(__HIP_DEVICE_COMPILE__)
  return (int)(targetval * 0.) + (nsearch > 1 ? 1 : 0);  // (timing experiment only: WRONG results -- what would k_thermal cost without the re-adding code in it?)
#endif
  double s = 0.;
  int j = 0;
  for (; j < nsearch; j++) {
    s += matrans_term_of(matrans_terms(env, c, ats0 + j), dir);
    if (!(s <= targetval)) break;
  }
  return j;
}
// the search of one direction of the record `rec` (level with nd / nu transitions, first entry of alltrans `ats`) with the
// 24-bit draw u: number of the direction's cumulative sums (the last one left out) <= (u * 2^-24) * (the direction's rate).
// On the filters; *amb: they cannot decide (the result is then meaningless).
// first0: the direction's first filter line when the caller has read it already (with the action filter: -DARTIS_MA_SPEC_DIR)
AHD int ma_search_filters(const Env &env, const MACtx &k, const U4 *rec, int dir, uint32_t u, bool *amb, const U4 *first0 = nullptr) {
  const int nsearch = ((dir != MADIR_UP) ? k.nd : k.nu) - 1;
  *amb = false;
  if (nsearch <= 0) return 0;
  bool a = env.ma_filters_off != 0;
  int ti = 0;
  for (int b0 = 0; b0 < nsearch && !a; b0 += MAREC_PER) {
    const U4 f = (b0 == 0 && first0 != nullptr) ? *first0 : rec[marec_slot(dir, b0 / MAREC_PER, k.nd, k.nu)];
    const int cnt = mafilt_count(f, u, &a);
    a = a || (f.w[3] >> 16) != MAFILT_NONE;
    if (a) break;
    ti += cnt;
    if (cnt < MAREC_PER) break;
  }
  *amb = a;
  return ti;
}
// The same search for the transition loop (round 6: one result instead of a count and a flag, the first line given): the number of the
// direction's cumulative sums <= z * (its rate), or -1 where the 15-bit entries cannot decide (a draw within their resolution of an entry,
// a line that is not usable, filters switched off). f0: the direction's first line, read with the action filter.
AHD int ma_search_filters_first(const Env &env, const MACtx &k, const U4 *rec, int dir, uint32_t u, const U4 &f0) {
  const int nsearch = ((dir != MADIR_UP) ? k.nd : k.nu) - 1;
  bool a;
  int ti = mafilt_count(f0, u, &a);
  // (a line's mark is 0x7FFF when it is usable; a direction of one transition has nothing to search: its line holds padding only, count 0)
  const bool bad = (int)a | (int)((f0.w[3] >> 16) != MAFILT_NONE) | (int)(env.ma_filters_off != 0);
  ti = bad ? -1 : ti;
  if (__builtin_expect(ti == MAREC_PER && nsearch > MAREC_PER, 0)) {
    // the first line's seven sums all lie below the draw and the direction has more: its further lines, one at a time
    const U4 *line = rec + marec_slot(dir, 1, k.nd, k.nu);
    int b0 = MAREC_PER;
    int cnt;
    do {
      const U4 f = *line++;
      cnt = mafilt_count(f, u, &a);
      const bool badl = (int)a | (int)((f.w[3] >> 16) != MAFILT_NONE);
      ti = badl ? -1 : ti + cnt;
      cnt = badl ? 0 : cnt;
      b0 += MAREC_PER;
    } while (cnt == MAREC_PER && b0 < nsearch);
  }
  return (nsearch <= 0) ? 0 : ti;
}
// ... again with the fine bytes of the lines whose 15-bit entries cannot decide (internal-down / internal-up; tables.h "FINE BYTES"): for the
// draws ma_search_filters() left. *amb: still undecided (3 draws of 2^24 per entry, or a line that is not usable).
AHD int ma_search_filters_fine(const Env &env, const MACtx &k, const U4 *rec, int dir, uint32_t u, bool *amb) {
  const int nsearch = ((dir != MADIR_UP) ? k.nd : k.nu) - 1;
  *amb = false;
  if (nsearch <= 0) return 0;
  if (env.ma_filters_off != 0 || dir == MADIR_RAD) {
    *amb = true;
    return 0;
  }
  int ti = 0;
  for (int b0 = 0; b0 < nsearch; b0 += MAREC_PER) {
    const int l = b0 / MAREC_PER;
    const U4 f = rec[marec_slot(dir, l, k.nd, k.nu)];
    bool a;
    int cnt = mafilt_count(f, u, &a);
    if ((f.w[3] >> 16) != MAFILT_NONE) {  // (the line is not usable: its sums were not finite fractions)
      *amb = true;
      return 0;
    }
    if (a) {
      const uint64_t fine = *(const uint64_t *)((const uint8_t *)rec + marec_fine_byte0(dir, l, k.nd, k.nu));
      cnt = mafilt_count_fine(f, fine, u, &a);
      if (a) {
        *amb = true;
        return 0;
      }
    }
    ti += cnt;
    if (cnt < MAREC_PER) break;
  }
  return ti;
}
// ... on the re-added sums (a draw the filters could not decide, fine bytes included)
AHD int ma_search_exact(const Env &env, const MACtx &k, const U4 *rec, int dir, uint32_t u) {
  if (dir != MADIR_RAD) {
    bool amb;
    const int ti = ma_search_filters_fine(env, k, rec, dir, u, &amb);
    if (!amb) return ti;
  }
  const bool down = dir != MADIR_UP;
  const int action = dir == MADIR_DOWN ? ARTIS_MA_ACTION_INTERNALDOWNSAME : (dir == MADIR_UP ? ARTIS_MA_ACTION_INTERNALUPSAME : ARTIS_MA_ACTION_RADDEEXC);
  const double targetval = rng_u24_value(u) * ma_rates_of(rec, k.nd, k.nu)[action];
  return ma_exact_search(env, k.c, k.ats + (down ? 0 : k.nd), dir, (down ? k.nd : k.nu) - 1, targetval);
}
AHD int ma_search_dir(const Env &env, const MACtx &k, const U4 *rec, int dir, uint32_t u) {
  bool amb;
  const int ti = ma_search_filters(env, k, rec, dir, u, &amb);
  if (__builtin_expect(amb, 0)) return ma_search_exact(env, k, rec, dir, u);
  return ti;
}
// the internal transition to the ti-th downward / upward transition's level: the walk goes on in that level's record
template <bool COLD = true>
AHD void ma_take_transition(const Env &env, Pkt &p, MACtx &k, bool down, int ti) {
  if (env.ma_tables_in_lds) {
    // the same information from two small static tables that the kernel has copied into LDS (k_thermal<.., true>): the
    // transition's target level (2 bytes), then that level's LevelPack -- two LDS reads instead of one trip to L2
    const int tl = env.M.alltrans_tlevel16[k.ats + (down ? 0 : k.nd) + ti];
    const LevelPack lp = env.M.level_pack[k.start + tl];
    p.ma_level = tl;
    k.rec = ma_resolve<COLD>(env, k.c, lp.rec_off);
    k.ats = lp.alltrans_startdown;
    k.nd = lp.ndown;
    k.nu = lp.nup;
    return;
  }
  const MaTarget tg = env.M.alltrans_target[k.ats + (down ? 0 : k.nd) + ti];
  MA_PROF_WAIT();
  p.ma_level = tg.level;
  k.rec = ma_resolve<COLD>(env, k.c, tg.rec);
  k.ats = tg.ats;
  k.nd = (int)(tg.ndnu & 0xFFFFu);
  k.nu = (int)(tg.ndnu >> 16);
}
// DEFER: an internal transition whose search the filters cannot decide (5e-4 of them) is not made here: MA_EXIT_DEFER is
// returned with the draw in k.defer, and the caller finishes it with ma_jump_deferred() once its transition loop is over --
// the re-adding of the sums (exp() and divisions of the rate coefficients) stays out of the loop of a kernel that runs
// at the edge of its registers, like the processes that end a walk.
constexpr int MA_EXIT_DEFER = 98;
// the walk stands in a cold level that has no record in this cell yet (tables.h "ON-DEMAND RECORDS"): nothing was drawn; the caller hands
// the packet to the slow path (PEND_MA_FILL)
constexpr int MA_EXIT_FILL = 97;
#ifndef ARTIS_MA_SPEC_DIR
#define ARTIS_MA_SPEC_DIR 1  // measured: k_thermal 466 -> 446 ms (the three slots are one 64-byte sector; the same idea lost in round 3, when they were three lines)
#endif
template <bool DEFER = false, bool COLD = true>
AHD int ma_jump_internal(const Env &env, Pkt &p, MACtx &k, const U4 *rec) {
  // index_upperbound (sn3d.h:85) over the 9 cumulative rates: action = number of cumulative values <= zrand * total,
  // clamped to the last one. Decided on the record's 16-byte filter (tables.h "FILTERS") unless the random number lies
  // within the filter's resolution of one of its entries; then on the f64 rates, with the same random number.
  if (COLD && __builtin_expect(k.rec == MA_REC_NONE, 0)) return MA_EXIT_FILL;
  MA_PROF_BEGIN();
  MA_PROF_MARK(env, 63);  // (clocks between the marks themselves: the cost of one mark)
#if defined(ARTIS_VISIT_COUNTS) && defined(__HIP_DEVICE_COMPILE__)
  if (env.visit_counts != nullptr) atomicAdd(&env.visit_counts[((int64_t)k.c * env.M.nlevels) + k.start + p.ma_level], 1u);
#endif
  int action;
#if ARTIS_MA_SPEC_DIR
  U4 fdir[2];
#endif
  {
    const U4 f = rec[0];
#if ARTIS_MA_SPEC_DIR
    // (measurement) the first filter lines of both directions with it: the three slots are one 64-byte sector, and the search
    // then waits for no second read
    fdir[0] = rec[1];
    fdir[1] = rec[2];
#endif
    MA_PROF_WAIT();
    MA_PROF_MARK(env, 59);
    // first entry above the last: the record has no usable filter (populate_mafilter_level: its total is not a positive finite
    // number). Round 6: ONE test leaves the common path -- "the filter cannot decide, or is not usable, or filters are off" -- and the
    // random number is drawn before it (the reference asserts on the total first, macroatom.cc:425: a record that fails ends the
    // call with an error, and what its generator holds then is never looked at).
    const uint32_t u1 = rng_u24(p);
    bool amb;
    action = mafilt_count(f, u1, &amb);
    // ("cum[8] = total <= zrand * total" never holds: zrand <= 1 - 2^-24, and the product of that with total is below total)
    const bool unusable = (f.w[0] & 0xFFFFu) > (f.w[3] >> 16);
    if (__builtin_expect((int)amb | (int)unusable | (int)(env.ma_filters_off != 0), 0)) {
      double r[MA_N], cum[MA_N];
      ma_load_rates(ma_rates_of(rec, k.nd, k.nu), r, cum);
      if (unusable && !(cum[MA_N - 1] > 0.)) {
        fail(env, 40);
        p.ma_level = -1;
        ARTIS_STAT(env, ARTIS_STAT_X_MA_JUMPS);
        return MA_EXIT_FAILED;
      }
      const double randomrate = rng_u24_value(u1) * cum[MA_N - 1];
      action = 0;
#pragma unroll
      for (int i = 0; i < MA_N; i++) action += (cum[i] <= randomrate) ? 1 : 0;  // cum is non-decreasing
      if (action > MA_N - 1) action = MA_N - 1;
    }
  }
  k.njumps++;  // stats::increment(INTERACTIONS) macroatom.cc:430 and the engine's transition counter: ma_flush_stats()
  const bool down = (action == ARTIS_MA_ACTION_INTERNALDOWNSAME);
  static_assert(ARTIS_MA_ACTION_INTERNALDOWNSAME == 4 && ARTIS_MA_ACTION_INTERNALUPSAME == 6, "the test below");
  if ((action | 2) == 6) {  // down or up, as ONE comparison (the compiler turns "== 4 || == 6" into a tree of branches)
    // macroatom.cc:433-447 and 536-550: one search for both directions, so that a wave runs it once. The target is the
    // number of the direction's cumulative sums <= zrand * (the direction's rate), the last sum (= the rate) left out.
    const uint32_t u2 = rng_u24(p);
    MA_PROF_MARK(env, 60);
#if ARTIS_MA_SPEC_DIR
    const U4 fsel = down ? fdir[0] : fdir[1];
    int ti = ma_search_filters_first(env, k, rec, down ? MADIR_DOWN : MADIR_UP, u2, fsel);
#else
    bool amb2;
    int ti = ma_search_filters(env, k, rec, down ? MADIR_DOWN : MADIR_UP, u2, &amb2);
    if (amb2) ti = -1;
#endif
    if (__builtin_expect(ti < 0, 0)) {
      if (DEFER) {
        k.defer = (int)(u2 | (down ? 0x1000000u : 0u));
        return MA_EXIT_DEFER;
      }
      ti = ma_search_exact(env, k, rec, down ? MADIR_DOWN : MADIR_UP, u2);
    }
    MA_PROF_MARK(env, 61);
    ma_take_transition<COLD>(env, p, k, down, ti);
    MA_PROF_MARK(env, 62);
    return -1;
  }
  return action;
}
// the second half of a transition that ma_jump_internal<true>() left undecided (MA_EXIT_DEFER): `rec` and k still describe
// the level the draw was made in
AHD void ma_jump_deferred(const Env &env, Pkt &p, MACtx &k, const U4 *rec) {
  const bool down = (k.defer & 0x1000000) != 0;
  const int ti = ma_search_exact(env, k, rec, down ? MADIR_DOWN : MADIR_UP, (uint32_t)k.defer & 0xFFFFFFu);
  ma_take_transition(env, p, k, down, ti);
}
// ... tried on the record's fine bytes first, where the caller would otherwise hand the packet to the slow-path kernel (k_thermal): true = the
// transition is made and the walk goes on in the caller's next phase; false = still undecided (k.defer keeps the draw)
template <bool COLD = true>
AHD bool ma_jump_deferred_fine(const Env &env, Pkt &p, MACtx &k, const U4 *rec) {
  const bool down = (k.defer & 0x1000000) != 0;
  bool amb;
  const int ti = ma_search_filters_fine(env, k, rec, down ? MADIR_DOWN : MADIR_UP, (uint32_t)k.defer & 0xFFFFFFu, &amb);
  if (amb) return false;
  ma_take_transition<COLD>(env, p, k, down, ti);
  return true;
}
// do_macroatom_raddeexcitation macroatom.cc:204 once the transition dti is known
AHD void ma_exit_raddeexc(const Env &env, Pkt &p, int64_t pi, const MACtx &k, int dti) {
  const DevModel &M = env.M;
  const int activatingline = p.ma_line;
  const int lineindex = M.alltrans_lineindex[k.ats + dti];
  if (lineindex == activatingline) ARTIS_STAT(env, ARTIS_STAT_RESONANCESCATTERINGS);
  const int ul = k.start + p.ma_level;
  const int lul = k.start + M.alltrans_targetlevelindex[k.ats + dti];
  const double e_trans = eps(M, ul) - eps(M, lul);
  const double oldnucmf = p.nu_cmf;
  p.nu_cmf = e_trans / HPLANCK;
  if (activatingline >= 0) ARTIS_STAT(env, (oldnucmf < p.nu_cmf) ? ARTIS_STAT_UPSCATTER : ARTIS_STAT_DOWNSCATTER);
  ARTIS_STAT(env, ARTIS_STAT_MA_DEACTIVATION_BB);
  emit_rpkt(env, p, pi);
  p.next_trans = lineindex + 1;
  p.emissiontype = lineindex;
  p.nscatterings = 0;
  ma_finish(env, p, pi);
}
// `rec`: the record the action was drawn from (the packet's current level: k still describes it).
// SPLIT: a search the filters cannot decide is left to the slow-path kernel (PEND_MA_RADSEARCH) instead of re-adding the sums here.
template <bool SPLIT = false>
AHD void ma_jump_exit(const Env &env, Pkt &p, int64_t pi, MACtx &k, const U4 *rec, int action) {
  const int c = k.c;
  if (action == ARTIS_MA_ACTION_RADDEEXC) {
    // targetval = zrand * (the rate); the transition is the number of cumulative radiative de-excitation sums <= targetval
    // among the first ndown - 1
    const uint32_t u = rng_u24(p);
    bool amb;
    int dti = ma_search_filters(env, k, rec, MADIR_RAD, u, &amb);
    if (amb) {
      if (SPLIT) {
        p.pend = PEND_MA_RADSEARCH;
        p.pend_arg = (int)u;
        return;
      }
      dti = ma_search_exact(env, k, rec, MADIR_RAD, u);
    }
    ma_exit_raddeexc(env, p, pi, k, dti);
  } else if (action == ARTIS_MA_ACTION_COLDEEXC || action == ARTIS_MA_ACTION_COLRECOMB) {
    ARTIS_STAT(env, action == ARTIS_MA_ACTION_COLDEEXC ? ARTIS_STAT_MA_DEACTIVATION_COLLDEEXC : ARTIS_STAT_MA_DEACTIVATION_COLLRECOMB);
    p.type = ARTIS_TYPE_KPKT;
#if !ARTIS_OPT_DIRECT_COL_HEAT
#if defined(__HIP_DEVICE_COMPILE__)
    if (env.estcache_nv == 1)
      est_cache_add_one(env, env.E.colheatingestimator, c, p.e_cmf);
    else
#endif
      cellest_add(env, env.E.colheatingestimator, CELLEST_COLHEAT, c, p.e_cmf);
#endif
    ma_finish(env, p, pi);
  } else if (action != MA_EXIT_FAILED) {
    // the rare bound-free channels need rate coefficients with exp() and, for a radiative recombination, an adaptive
    // quadrature: they are executed by the slow-path kernel (ma_slow_action) so that this loop stays small
    p.pend = PEND_MA_ACTION;
    p.pend_arg = action;
  }
}
AHD void ma_flush_stats(const Env &env, MACtx &k) {
  if (k.njumps != 0) {
    ARTIS_STAT_ADD(env, ARTIS_STAT_X_MA_JUMPS, k.njumps);
    ARTIS_STAT_ADD(env, ARTIS_STAT_INTERACTIONS, k.njumps);
    k.njumps = 0;
  }
}
// SPLIT: the form k_thermal runs -- a search the filters cannot decide becomes a pending slow-path action (PEND_MA_SEARCH / _RADSEARCH)
template <bool SPLIT = false>
AHD void ma_jump(const Env &env, Pkt &p, int64_t pi, MACtx &k) {
  ma_prepare(env, p, k);
  const U4 *rec = ma_record(env, k);
  const int action = ma_jump_internal<SPLIT>(env, p, k, rec);
  if (action == MA_EXIT_FILL) {
    p.pend = PEND_MA_FILL;
  } else if (SPLIT && action == MA_EXIT_DEFER) {
    p.pend = PEND_MA_SEARCH;
    p.pend_arg = k.defer;
  } else if (action >= 0) {
    ma_jump_exit<SPLIT>(env, p, pi, k, rec, action);
  }
  ma_flush_stats(env, k);
}

// A cold level's record in cell c, filled at the first visit (tables.h "ON-DEMAND RECORDS"; the reference: calc_rates_if_needed
// macroatom.cc:398-417): the sequential forms of the population, which add the same terms in the same order as its kernels (the host
// emulation fills every record with them; artis_amd_debug_cellcache() checks the kernels' records against them).
AHD void ma_fill_record(const Env &env, int c, int ul) {
  const LevelPack lpk = env.M.level_pack[ul];
  populate_mainit_at(ma_rec_of(env, c, lpk), lpk);
  populate_level_bb<true>(env, c, ul, nullptr);
  populate_macroatom<false>(env, c, ul);
  populate_coolfilter_level_seq(env, c, ul);
}
// The record of cold level ul in cell c exists (true), or another lane is filling it right now (false: ask again). A level without one gets it
// here: the lane claims the level's place in the cell's table, takes units of the pool, fills the record with the sequential forms and
// publishes it. (The pool used up: error flag 46, *failed.)
AHD bool ma_pool_known_full(const Env &env, uint32_t nunits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint64_t)__hip_atomic_load(env.K.ma_pool_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + nunits > (uint64_t)env.ma_pool_cap;
#else
  (void)env;
  (void)nunits;
  return false;
#endif
}
// A packet's macro-atom stands in a cold level that has NO record in its cell and none on its way (the row table says -1): with the pool used
// up nothing in the running kernel can give it one (k_tail hands such a packet to the slow-path list, before whose next launch the host
// empties the pool). A record another wave is filling right now (<= -2) will be published: not absent.
AHD bool ma_record_absent(const Env &env, const Pkt &p) {
  const DevModel &M = env.M;
  if (M.ncold == 0 || p.ma_level < 0) return false;
  const LevelPack lpk = M.level_pack[M.ion_uniquelevelindexstart[uion(M, p.ma_element, p.ma_ion)] + p.ma_level];
  if (lpk.rec_off >= 0) return false;
  return ma_rowtab_load(env.K.ma_rowtab + (krow(env, M.propcell_nonemptymgi[p.cellindex]) * M.ncold) + (-lpk.rec_off - 1)) == -1;
}
AHD bool ma_ensure_record(const Env &env, int c, int ul, bool *failed, bool *full = nullptr) {
  const DevModel &M = env.M;
  const LevelPack lpk = M.level_pack[ul];
  if (lpk.rec_off >= 0) return true;
  int32_t *tab = env.K.ma_rowtab + (krow(env, c) * M.ncold) + (-lpk.rec_off - 1);
  const uint32_t nunits = (uint32_t)((marec_slots(lpk.ndown, lpk.nup) + MAPOOL_UNIT - 1) / MAPOOL_UNIT);
  if (nunits > env.ma_pool_cap) {  // a pool that cannot hold this one record: ARTIS_AMD_MA_POOLFRAC (artis_engine.hip names the remedy)
    fail(env, 46);
    *failed = true;
    return false;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  const int32_t v = ma_rowtab_load(tab);
  ma_rowtab_acquire(env, v);
  if (v >= 0) return true;
  if (v != -1) return false;                       // another lane is at it
  if (atomicCAS(tab, -1, -2) != -1) return false;  // claimed: exactly one lane goes on
  // (a pool already known to be used up is not asked again: the counter only ever grows by requests that can still be served plus the few in
  // flight when it fills, so it cannot wrap however long packets wait -- ADVICE r05; the sums are compared in 64 bits)
  const bool known_full = ma_pool_known_full(env, nunits);
  const uint32_t unit = known_full ? env.ma_pool_cap : atomicAdd(env.K.ma_pool_used, nunits);
  if ((uint64_t)unit + nunits > (uint64_t)env.ma_pool_cap) {
    // the pool is used up: the level stays without a record, the packet waits on the slow-path list; the host empties the pool before that
    // list's next launch (what the pool held is filled again when next needed, as after a tile's refill)
    __hip_atomic_store(tab, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(env.ma_pool_full, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (full) *full = true;
    return false;
  }
  __hip_atomic_store(tab, -((int32_t)unit + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (where ma_rec_of() finds it while it is filled)
  ma_fill_record(env, c, ul);
  __threadfence();
  __hip_atomic_store(tab, (int32_t)unit, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  return true;
#else
  if (*tab >= 0) return true;
  uint32_t unit = *env.K.ma_pool_used;
  if (unit + nunits > env.ma_pool_cap) {
    // (the host emulation is one thread: it empties the pool on the spot -- every cold level of every cell is without a record again)
    for (int64_t i = 0; i < (int64_t)M.npts_nonempty * M.ncold; i++) env.K.ma_rowtab[i] = -1;
    unit = 0;
    if (env.ma_pool_full) *env.ma_pool_full += 1;
  }
  *env.K.ma_pool_used = unit + nunits;
  *tab = -((int32_t)unit + 3);
  ma_fill_record(env, c, ul);
  *tab = (int32_t)unit;
  return true;
#endif
}
// PEND_MA_FILL in the slow-path kernel (the sequential form: the tail kernel's and the host emulation's; k_slow lets the wave fill,
// ma_slow_fill_claim() below): whether this lane filled the record or another is at it, the packet goes back to its list -- the record is
// complete before the next launch of the thermal kernel (the tail kernel's waves come here again until it is).
AHD void ma_slow_fill(const Env &env, Pkt &p) {
  const DevModel &M = env.M;
  p.pend = PEND_NONE;
  bool failed = false, full = false;
  (void)ma_ensure_record(env, M.propcell_nonemptymgi[p.cellindex], M.ion_uniquelevelindexstart[uion(M, p.ma_element, p.ma_ion)] + p.ma_level, &failed, &full);
  if (failed) p.ma_level = -1;
  if (full) p.pend = PEND_MA_FILL;  // (the pool is used up: the packet waits on the slow-path list for the emptied pool)
}
// A slow-path action of the active macro-atom reads its level's record. The record of a cold level may be gone by now: a tiled run refills
// a tile (and empties the pool) while packets wait for it. true: it is there (again).
AHD bool ma_slow_record_ready(const Env &env, Pkt &p) {
  const DevModel &M = env.M;
  if (M.ncold == 0) return true;
  bool failed = false;
  const bool ready = ma_ensure_record(env, M.propcell_nonemptymgi[p.cellindex], M.ion_uniquelevelindexstart[uion(M, p.ma_element, p.ma_ion)] + p.ma_level, &failed);
  if (failed) {
    p.pend = PEND_NONE;
    p.ma_level = -1;
  }
  return ready;
}

#if defined(__HIPCC__) && !defined(ARTIS_HOST_EMU)
// ma_slow_fill() in three parts for the kernels that let the WAVE fill the record (artis_engine.hip ma_fill_record_wave): the lane claims the
// level's place and takes units of the pool (true: this lane's record is to be filled at *unit; the packet's pend is cleared either way),
// the wave fills, the lane publishes.
__device__ inline bool ma_slow_fill_claim(const Env &env, Pkt &p, int *c_out, int *ul_out, int32_t *unit_out) {
  const DevModel &M = env.M;
  p.pend = PEND_NONE;
  const int c = M.propcell_nonemptymgi[p.cellindex];
  const int ul = M.ion_uniquelevelindexstart[uion(M, p.ma_element, p.ma_ion)] + p.ma_level;
  const LevelPack lpk = M.level_pack[ul];
  if (lpk.rec_off >= 0) return false;
  int32_t *tab = env.K.ma_rowtab + (krow(env, c) * M.ncold) + (-lpk.rec_off - 1);
  if (ma_rowtab_load(tab) != -1) return false;
  if (atomicCAS(tab, -1, -2) != -1) return false;
  const uint32_t nunits = (uint32_t)((marec_slots(lpk.ndown, lpk.nup) + MAPOOL_UNIT - 1) / MAPOOL_UNIT);
  if (nunits > env.ma_pool_cap) {
    __hip_atomic_store(tab, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fail(env, 46);
    p.ma_level = -1;
    return false;
  }
  const bool known_full = ma_pool_known_full(env, nunits);
  const uint32_t unit = known_full ? env.ma_pool_cap : atomicAdd(env.K.ma_pool_used, nunits);
  if ((uint64_t)unit + nunits > (uint64_t)env.ma_pool_cap) {  // (used up: ma_ensure_record())
    __hip_atomic_store(tab, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(env.ma_pool_full, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p.pend = PEND_MA_FILL;
    return false;
  }
  __hip_atomic_store(tab, -((int32_t)unit + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *c_out = c;
  *ul_out = ul;
  *unit_out = (int32_t)unit;
  return true;
}
__device__ inline void ma_slow_fill_publish(const Env &env, int c, int ul, int32_t unit) {
  __threadfence();
  __hip_atomic_store(env.K.ma_rowtab + (krow(env, c) * env.M.ncold) + (-env.M.level_pack[ul].rec_off - 1), unit, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// the searches k_thermal left undecided (PEND_MA_SEARCH, PEND_MA_RADSEARCH), in the slow-path kernel
AHD void ma_slow_search(const Env &env, Pkt &p, int64_t pi) {
  MACtx k = ma_ctx(env, p);
  ma_prepare(env, p, k);
  const U4 *rec = ma_record(env, k);
  const uint32_t u = (uint32_t)p.pend_arg & 0xFFFFFFu;
  if (p.pend == PEND_MA_SEARCH) {
    const bool down = ((uint32_t)p.pend_arg & 0x1000000u) != 0;
    p.pend = PEND_NONE;
    const int ti = ma_search_exact(env, k, rec, down ? MADIR_DOWN : MADIR_UP, u);
    ma_take_transition(env, p, k, down, ti);  // (the transition was counted when its process was drawn)
  } else {
    p.pend = PEND_NONE;
    const int dti = ma_search_exact(env, k, rec, MADIR_RAD, u);
    ma_exit_raddeexc(env, p, pi, k, dti);
  }
}

// the bound-free transitions of do_macroatom(): macroatom.cc:481-488, 501-533, 552-560
AHD void ma_slow_action(const Env &env, Pkt &p, int64_t pi, FbSel *sel = nullptr) {
  const DevModel &M = env.M;
  const int c = M.propcell_nonemptymgi[p.cellindex];
  const int element = p.ma_element;
  const int ion = p.ma_ion;
  const int level = p.ma_level;
  const int action = p.pend_arg;
  p.pend = PEND_NONE;
  const int ui = uion(M, element, ion);
  const int ul = M.ion_uniquelevelindexstart[ui] + level;
  const double e_cur = eps(M, ul);
  const LevelPack lpk = M.level_pack[ul];
  const double rate_sel = ma_rates_of(ma_rec_of(env, c, lpk), lpk.ndown, lpk.nup)[action];
  const float T_e = env.C.Te[c];
  const float cnne = clumpednne(env.C, c);
  if (action == ARTIS_MA_ACTION_RADRECOMB) {
    // do_macroatom_radrecomb macroatom.cc:248
    const double targetval = rng_uniform(p) * rate_sel;
    double rate = 0;
    const int ls = M.ion_uniquelevelindexstart[ui - 1];
    int lowerlevel = -1, sel_t = -1;
    for (int r = M.level_recomb_start[ul]; r < M.level_recomb_start[ul + 1]; r++) {
      const int l = M.recomb_lower[r];
      const int t = M.recomb_target[r];
      const double e_trans = e_cur - eps(M, ls + l);
      const double R = rad_recomb(M, T_e, cnne, element, ion, l, t);
      rate += R * e_trans;
      if (targetval < rate) {
        lowerlevel = l;
        sel_t = t;
        break;
      }
    }
    if (lowerlevel < 0) {
      fail(env, 41);
      p.ma_level = -1;
      return;
    }
    p.nu_cmf = select_continuum_nu_sel(env, element, ion - 1, lowerlevel, sel_t, T_e, p, sel);
    if (sel != nullptr && sel->mode == 1) return;  // (recorded: the wave selects the frequency, then the action runs again)
    ARTIS_STAT(env, ARTIS_STAT_MA_DEACTIVATION_FB);
    emit_rpkt(env, p, pi);
    p.next_trans = -1;
    p.emissiontype = emtype_continuum(M, ls + lowerlevel, sel_t);
    p.nscatterings = 0;
    ma_finish(env, p, pi);
  } else if (action == ARTIS_MA_ACTION_INTERNALDOWNLOWER) {
    ARTIS_STAT(env, ARTIS_STAT_MA_INTERNALDOWNLOWER);
    const double targetrate = rng_uniform(p) * rate_sel;
    double rate = 0.;
    const int ls = M.ion_uniquelevelindexstart[ui - 1];
    int lower = -1;
    for (int r = M.level_recomb_start[ul]; r < M.level_recomb_start[ul + 1]; r++) {
      const int l = M.recomb_lower[r];
      const int t = M.recomb_target[r];
      const double e_target = eps(M, ls + l);
      const double e_trans = e_cur - e_target;
      const double R = rad_recomb(M, T_e, cnne, element, ion, l, t);
      const double Cc = col_recomb(M, T_e, cnne, element, ion, l, t, e_trans);
      rate += (R + Cc) * e_target;
      if (rate > targetrate) {
        lower = l;
        break;
      }
    }
    if (lower < 0) {
      fail(env, 42);
      p.ma_level = -1;
      return;
    }
    p.ma_ion = ion - 1;
    p.ma_level = lower;
  } else if (action == ARTIS_MA_ACTION_INTERNALUPHIGHER) {
    ARTIS_STAT(env, ARTIS_STAT_MA_INTERNALUPHIGHER);
    // do_macroatom_ionisation macroatom.cc:298
    const double targetrate = rng_uniform(p) * rate_sel;
    double rate = 0.;
    const int nt = M.level_nphixstargets[ul];
    const double *cpc = env.K.corrphotoioncoeff + (krow(env, c) * M.nphixstargets_total) + M.level_phixstargetstart[ul];
    int newlevel = -1;
    for (int t = 0; t < nt; t++) {
      const double e_trans = phixs_threshold(M, element, ion, level, t);
      const double R = cpc[t];
      const double Cc = col_ion(M, T_e, cnne, element, ion, level, t, e_trans);
      rate += (R + Cc) * e_cur;
      if (rate > targetrate) {
        newlevel = phixs_upperlevel(M, ul, t);
        break;
      }
    }
    if (newlevel < 0) {
      fail(env, 43);
      p.ma_level = -1;
      return;
    }
    p.ma_level = newlevel;
    p.ma_ion = ion + 1;
#if ARTIS_OPT_NT_ON
  } else if (action == ARTIS_MA_ACTION_INTERNALUPHIGHERNT) {  // macroatom.cc:562
    p.ma_ion = nt_random_upperion(env, p, c, element, ion, false);
    p.ma_level = 0;
    ARTIS_STAT(env, ARTIS_STAT_MA_INTERNALUPHIGHERNT);
#endif
  } else {
    fail(env, 44);  // MA_ACTION_INTERNALUPHIGHERNT needs NT_ON
    p.ma_level = -1;
  }
}

// rpkt_event_continuum rpkt.cc:422
// SPLIT: an absorption is left to the slow-path kernel (PEND_RPKT_ABSORB; U24 >= 0: that kernel, with the event's draw handed in)
template <bool SPLIT = false>
AHD void rpkt_event_continuum(const Env &env, Pkt &p, int64_t pi, Chi &x, int64_t slot, int32_t U24 = -1) {
  const DevModel &M = env.M;
  const double nu = p.nu_cmf;
  const double dop = doppler(p);
  const double chi_cont = chi_total(x) * dop;
  const double chi_es = x.chi_escatter * dop;
  const double chi_ff = x.chi_freefree_heat * dop;
  const double chi_bf = x.chi_boundfree * dop;
  const uint32_t u_event = (U24 >= 0) ? (uint32_t)U24 : rng_u24(p);
  const double chi_rnd = rng_u24_value(u_event) * chi_cont;  // (= rng_uniform(p) * chi_cont)
  if (SPLIT && !(chi_rnd < chi_es)) {
    p.pend = PEND_RPKT_ABSORB;
    p.pend_arg = (int32_t)u_event;
    return;
  }
  if (chi_rnd < chi_es) {
    p.nscatterings++;
    ARTIS_STAT(env, ARTIS_STAT_ELECTRON_SCATTERINGS);
#if ARTIS_OPT_VPKT_ON
    trace_vpkts(env, p, pi, ARTIS_TYPE_RPKT);  // rpkt.cc:441
#endif
    electron_scatter(p);
    set_em_here(env, p, pi);
  } else if (chi_rnd < chi_es + chi_ff) {
    ARTIS_STAT(env, ARTIS_STAT_K_FROM_FF);
    p.type = ARTIS_TYPE_KPKT;
    p.absorptiontype = ARTIS_ABSTYPE_FREEFREE;
  } else if (chi_rnd < chi_es + chi_ff + chi_bf) {
    p.absorptiontype = ARTIS_ABSTYPE_BOUNDFREE;
    const double chi_bf_rand = rng_uniform(p) * x.chi_boundfree;
    int ci = -1;
    chi_bf_gammacontr<true>(env, x.nonemptymgi, x.nu, slot, chi_bf_rand, &ci);
    const double nu_edge = M.allcont_nu_edge[ci];
    const int element = M.allcont_element[ci];
    const int ion = M.allcont_ion[ci];
    const int level = M.allcont_level[ci];
    const int t = M.allcont_phixstargetindex[ci];
    if (rng_uniform(p) < nu_edge / nu) {
      ARTIS_STAT(env, ARTIS_STAT_MA_ACTIVATION_BF);
      const MAState ma = {element, ion + 1, phixs_upperlevel(M, lstart(M, element, ion) + level, t), -99};
      ma_activate(p, ma, 1);
    } else {
      ARTIS_STAT(env, ARTIS_STAT_K_FROM_BF);
      p.type = ARTIS_TYPE_KPKT;
    }
  } else {
    fail(env, 50);
  }
}

// update_estimators rpkt.cc:502 + radfield::update_estimators radfield.cc:745
AHD void update_estimators(const Env &env, double e_cmf, double nu_cmf, double distance, int c, const Chi &x, bool thick, int64_t slot) {
  const double de = distance * e_cmf;
#if defined(__HIP_DEVICE_COMPILE__)
  const bool cached = env.estcache_nv == 3;  // (k_rpkt on a model with many cells: the wave's cache of accumulators)
  if (cached && de != 0) est_cache_add(env, c, de, de * nu_cmf, thick ? 0. : de * x.chi_freefree_heat);
#else
  const bool cached = false;
#endif
  if (de != 0 && !cached) {
    cellest_add(env, env.E.J, CELLEST_J, c, de);
    cellest_add(env, env.E.nuJ, CELLEST_NUJ, c, de * nu_cmf);
  }
  if (thick) return;
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
  if (de != 0) update_bfestimators(env, c, de, nu_cmf, x);
#endif
#if ARTIS_OPT_MULTIBIN_RADFIELD_MODEL_ON
  if (de != 0) {  // radfield.cc:762-770
    const int b = radbin_select(nu_cmf);
    if (b >= 0) {
      const int64_t k = ((int64_t)c * ARTIS_OPT_RADFIELDBINCOUNT) + b;
      ARTIS_EST_ADD(&env.E.radfieldbin_J[k * env.pair_stride], de);  // ({J, nuJ} of a bin: neighbours on the device)
      ARTIS_EST_ADD(&env.E.radfieldbin_nuJ[k * env.pair_stride], de * nu_cmf);
    }
  }
#endif
  if (!cached) cellest_add(env, env.E.ffheatingestimator, CELLEST_FFHEAT, c, de * x.chi_freefree_heat);
#if ARTIS_OPT_USE_LUT_PHOTOION || ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS
  // update_bfestimators rpkt.cc:519: the loop runs over the ground continua in rising nu_edge until nu_cmf <= nu_edge;
  // entries whose groundcont_gamma_contr is zero add nothing and are not in the packet's list
  const int nbfg = env.M.nbfcontinua_ground;
  const int ng = env.gamma_n[slot];
  const double *wsv = env.gamma_ws + (slot * nbfg);
  const int32_t *wsi = env.gamma_gi + (slot * nbfg);
  for (int j = 0; j < ng; j++) {
    const int i = wsi[j];
    const double nu_edge = env.M.groundcont_nu_edge[i];
    if (nu_cmf <= nu_edge) return;
    const int64_t k = ((int64_t)c * nbfg) + i;
    const double contr = wsv[j];
#if ARTIS_OPT_USE_LUT_PHOTOION
    ARTIS_EST_ADD(&env.E.gammaestimator[k * env.pair_stride], contr * (de / nu_cmf));
#endif
#if ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS
    ARTIS_EST_ADD(&env.E.bfheatingestimator[k * env.pair_stride], contr * de * (1. - (nu_edge / nu_cmf)));
#endif
  }
#endif
}

// do_rpkt_step rpkt.cc:542
template <bool SPLIT = false>
AHD bool do_rpkt_step(const Env &env, Pkt &p, int64_t pi, Chi &x, int64_t slot) {
  const DevModel &M = env.M;
  const double t2 = env.S.ts_end;
  ARTIS_STAT(env, ARTIS_STAT_X_RPKT_STEPS);
  PROF_BEGIN();
  const int c = M.propcell_nonemptymgi[p.cellindex];
  MAState ma = {-1, -1, -1, -99};
  const double tau_rnd = -log((double)rng_uniform_pos(p));
  int next_cell = -1;
  const double bdist = boundary_distance(env, p, &next_cell);
  PROF_MARK(env, 48);
  if (bdist == 0) {
    change_cell_or_escape(env, p, pi, next_cell);
    if (p.type != ARTIS_TYPE_RPKT) return false;
    const int nc = M.propcell_nonemptymgi[p.cellindex];
    return (nc < 0 || nc == c);
  }
  const double tdist = (t2 - p.prop_time) * CLIGHT_PROP;
  if (!(tdist >= 0)) fail(env, 60);
  const double abort_dist = dmin(tdist, bdist);
  double edist = -1;
  bool is_bb = true;
  const bool thick = (c >= 0) && (env.C.thick[c] == ARTIS_CELL_THICK);
  if (c < 0) {
    edist = DBLMAX;
    p.next_trans = -1;
  } else if (thick) {
    const double chi_grey = env.C.kappagrey[c] * env.C.rho[c] * doppler(p);
    edist = tau_rnd / chi_grey;
    p.next_trans = -1;
  } else {
    chi_rpkt_cont(env, p.nu_cmf, x, c, slot);
    PROF_MARK(env, 49);
    // get_nu_cmf_abort rpkt.cc:54
    const double half = abort_dist / 2.;
    const double abort_time = p.prop_time + (half / CLIGHT_PROP) + (half / CLIGHT_PROP);
    const double nu_cmf_abort = p.nu_rf * doppler_at(p.px + (p.dx * half) + (p.dx * half), p.py + (p.dy * half) + (p.dy * half),
                                                     p.pz + (p.dz * half) + (p.dz * half), p.dx, p.dy, p.dz, abort_time);
    const double dop = doppler(p);
    const double dnu_on_dl = (nu_cmf_abort - p.nu_cmf) / abort_dist;  // rpkt.cc:591
#if ARTIS_OPT_RPKT_USE_EXPANSION_OPACITIES
    edist = possible_event_expopac(env, c, p, x, ma, tau_rnd, nu_cmf_abort, dnu_on_dl, dop, &is_bb);  // rpkt.cc:594
#else
    int nt = p.next_trans;
    edist = possible_event(env, c, p, x, ma, tau_rnd, abort_dist, nu_cmf_abort, dnu_on_dl, dop, &nt, &is_bb);
    p.next_trans = nt;
#endif
    PROF_MARK(env, 50);
  }
  if (!(edist >= 0)) fail(env, 61);

  if ((edist < bdist) && (edist <= tdist)) {
    move_pkt(p, edist / 2.);
    update_estimators(env, p.e_cmf, p.nu_cmf, edist, c, x, thick, slot);
    move_pkt(p, edist / 2.);
    PROF_MARK(env, 51);
    ARTIS_STAT(env, ARTIS_STAT_INTERACTIONS);
    if (thick) {
      p.nscatterings++;
      ARTIS_STAT(env, ARTIS_STAT_ELECTRON_SCATTERINGS);
      emit_rpkt(env, p, pi);
    } else if (!is_bb) {
      rpkt_event_continuum<SPLIT>(env, p, pi, x, slot);
    } else {
#if !ARTIS_OPT_RPKT_BB_THERMALISATION
      ARTIS_STAT(env, ARTIS_STAT_MA_ACTIVATION_BB);
      p.absorptiontype = ma.activatingline;
      env.P.flight[pi].absorptionfreq = p.nu_rf;
      ma_activate(p, ma, 1);
#else
      // probability-based thermalisation (redistribution of the packet's frequency) or scattering, rpkt.cc:624-648
      if (ARTIS_OPT_RPKT_BB_THERMALISATION_PROBABILITY >= 1. || rng_uniform(p) < ARTIS_OPT_RPKT_BB_THERMALISATION_PROBABILITY) {
        p.absorptiontype = ma.activatingline;
        env.P.flight[pi].absorptionfreq = p.nu_rf;
        p.nu_cmf = sample_planck_times_expopac(env, c, p);
        p.next_trans = -1;
        p.emissiontype = ARTIS_EMTYPE_NOTSET;
        p.trueemissiontype = ARTIS_EMTYPE_NOTSET;
        p.flags |= PKT_FLAG_TRUEEM_NAN;
        env.P.cold[pi].trueem_time = -1.f;
        p.nscatterings = 0;
      } else {
        p.nscatterings++;
        ARTIS_STAT(env, ARTIS_STAT_ELECTRON_SCATTERINGS);
      }
      emit_rpkt(env, p, pi);
#endif
    }
    PROF_MARK(env, 52);
    return (p.type == ARTIS_TYPE_RPKT);
  }
  if ((bdist <= tdist) && (bdist <= edist)) {
    move_pkt(p, bdist / 2.);
    if (c >= 0) update_estimators(env, p.e_cmf, p.nu_cmf, bdist, c, x, thick, slot);
    move_pkt(p, bdist / 2.);
    PROF_MARK(env, 51);
    if (next_cell != p.cellindex) {
      change_cell_or_escape(env, p, pi, next_cell);
      if (next_cell < 0) return false;
      const int nc = M.propcell_nonemptymgi[p.cellindex];
      return ((nc < 0) || (nc == c));
    }
    return true;
  }
  if ((tdist < bdist) && (tdist <= edist)) {
    move_pkt(p, tdist / 2.);
    if (c >= 0) update_estimators(env, p.e_cmf, p.nu_cmf, tdist, c, x, thick, slot);
    move_pkt(p, tdist / 2.);
    p.prop_time = t2;
    return false;
  }
  fail(env, 62);
  return false;
}

// ---------------------------------------------------------------- kpkt.cc
#ifndef ARTIS_KPKT_BLOCKED_SEARCH
#define ARTIS_KPKT_BLOCKED_SEARCH 0  // 1: the two searches of do_kpkt() in blocked form. Fewer dependent reads (8 -> 4 and
                                     // 6 -> 2 stages) but 3x the loads: measured +12 % on k_thermal (MI355X, round 2)
#endif
AHD double sample_planck_montecarlo(double T, Pkt &p) {  // kpkt.cc:266
  const double nu_peak = 5.879e10 * T;
  const double B_peak = planck(nu_peak, T);
  while (true) {
    const double nu = ARTIS_OPT_NU_MIN_R + (rng_uniform(p) * (ARTIS_OPT_NU_MAX_R - ARTIS_OPT_NU_MIN_R));
    if (rng_uniform(p) * B_peak <= planck(nu, T)) return nu;
  }
}
AHD void thermal_emission_flags(const Env &env, Pkt &p, int64_t pi, int emtype) {
  p.next_trans = -1;
  p.emissiontype = emtype;
  p.trueemissiontype = emtype;
  set_trueem_from_em(env, p, pi);
  p.nscatterings = 0;
}
AHD void do_kpkt_blackbody(const Env &env, Pkt &p, int64_t pi) {  // kpkt.cc:399
  ARTIS_STAT(env, ARTIS_STAT_X_KPKT_STEPS);
  const int c = env.M.propcell_nonemptymgi[p.cellindex];
#if ARTIS_OPT_RPKT_BB_THERMALISATION
  if (env.C.thick[c] != ARTIS_CELL_THICK) {  // kpkt.cc:402
    p.nu_cmf = sample_planck_times_expopac(env, c, p);
  } else
#endif
  p.nu_cmf = sample_planck_montecarlo(env.C.Te[c], p);
  emit_rpkt(env, p, pi);
  ARTIS_STAT(env, ARTIS_STAT_K_TO_R_BB);
  ARTIS_STAT(env, ARTIS_STAT_INTERACTIONS);
  thermal_emission_flags(env, p, pi, ARTIS_EMTYPE_FREEFREE);
}
// kpkt.cc:461-476 on the terms themselves: the number of the level's first nsearch running cooling sums <= rnd_process, the
// sums re-added from `lo` (the ion's running sum before the level) term by term, as calculate_cooling_rates_ion() adds them
AHD int kpkt_collexc_exact(const Env &env, int c, int ats_up0, int nsearch, double lo, double rnd_process) {
#if defined(ARTIS_MA_FAKE_EXACT) && defined(__HIP_DEVICE_COMPILE__)
  return 0;
#endif
  double s = lo;
  int j = 0;
  for (; j < nsearch; j++) {
    s += matrans_terms(env, c, ats_up0 + j).kterm;
    if (s > rnd_process) break;
  }
  return j;
}
// kpkt.cc:477-490: the k-packet activates the macro-atom in the upper level of the chosen collisional excitation
AHD void kpkt_collexc_activate(const Env &env, Pkt &p, int element, int ion, const LevelPack &lpk, int first) {
  const int upper = env.M.alltrans_targetlevelindex[lpk.alltrans_startdown + lpk.ndown + first];
  ARTIS_STAT(env, ARTIS_STAT_MA_ACTIVATION_COLLEXC);
  ARTIS_STAT(env, ARTIS_STAT_K_TO_MA_COLLEXC);
  p.trueemissiontype = ARTIS_EMTYPE_NOTSET;
  p.flags |= PKT_FLAG_TRUEEM_NAN;  // trueem_pos = NaN (kpkt.cc:483)
  const MAState ma = {element, ion, upper, -99};
  ma_activate(p, ma, 0);
}
// ... in the slow-path kernel, for a k-packet whose cooling-term search k_thermal left undecided (PEND_KPKT_COLLEXC: the ion in
// ma_element / ma_ion, the draw of rnd_process in pend_arg): the term is found again from the draw, the sums re-added
AHD void kpkt_slow_collexc(const Env &env, Pkt &p) {
  const DevModel &M = env.M;
  const int c = M.propcell_nonemptymgi[p.cellindex];
  const int element = p.ma_element, ion = p.ma_ion;
  const uint32_t u = (uint32_t)p.pend_arg & 0xFFFFFFu;
  p.pend = PEND_NONE;
  p.ma_element = -1;
  p.ma_ion = -1;
  const int ui = uion(M, element, ion);
  const int ionstart = M.ion_coolingoffset[ui];
  const int nterms = M.ion_ncoolingterms[ui];
  const double *contribs = env.K.cooling_contrib + (krow(env, c) * M.ncoolingterms) + ionstart;
  const double rnd_process = rng_u24_value(u) * contribs[nterms - 1];
  const int ionoffset = upper_bound_d(contribs, nterms, rnd_process);
  const int ul = M.ion_uniquelevelindexstart[ui] + M.coolinglist_level[ionstart + ionoffset];
  const LevelPack lpk = M.level_pack[ul];
  const double lo = (ionoffset > 0) ? contribs[ionoffset - 1] : 0.;
  const int first = kpkt_collexc_exact(env, c, lpk.alltrans_startdown + lpk.ndown, lpk.nup - 1, lo, rnd_process);
  kpkt_collexc_activate(env, p, element, ion, lpk, first);
}
// do_kpkt kpkt.cc:425. SPLIT: a collisional-excitation search the level's filter cannot decide is left to the slow-path kernel.
template <bool SPLIT = false>
AHD void do_kpkt(const Env &env, Pkt &p, int64_t pi) {
  const DevModel &M = env.M;
  const double t2 = env.S.ts_end;
  ARTIS_STAT(env, ARTIS_STAT_X_KPKT_STEPS);
  const double deltat = ARTIS_KPKTDIFFUSION_TIMESTEP_FRACTION * env.S.width;
  const double t_current = dmin(p.prop_time + deltat, t2);
  const double sf = t_current / p.prop_time;
  p.px = p.px * sf;
  p.py = p.py * sf;
  p.pz = p.pz * sf;
  p.e_cmf *= p.prop_time / t_current;
  p.prop_time = t_current;
  if (t_current >= t2) return;
  ARTIS_STAT(env, ARTIS_STAT_INTERACTIONS);
  PROF_BEGIN();
  const int c = M.propcell_nonemptymgi[p.cellindex];
  const double *ioncontribs = env.K.ion_cooling_contribs + (krow(env, c) * M.nions);
  const uint16_t *guide = env.K.cool_guide + (krow(env, c) * M.nguide);  // (tables.h "COOLING GUIDES"; nguide == 0: none)
  const uint32_t u_ion = rng_u24(p);
  const double rndcool_ion = rng_u24_value(u_ion) * ioncontribs[M.nions - 1];  // (= rng_uniform(p) * ...)
  const int ui = (M.nguide > 0) ? guided_upper_bound(ioncontribs, M.nions, rndcool_ion, guide, M.guide_ion_shift, u_ion)
                 : ARTIS_KPKT_BLOCKED_SEARCH ? upper_bound_blocked<6>(ioncontribs, M.nions, rndcool_ion)
                                             : upper_bound_d(ioncontribs, M.nions, rndcool_ion);
  if (!(ui < M.nions)) {
    fail(env, 70);
    return;
  }
  PROF_MARK(env, 56);  // (-DARTIS_PROFILE) the ion drawn
  const int element = M.ion_element[ui];
  const int ion = ui - M.elem_uniqueionindexstart[element];
  const int ionstart = M.ion_coolingoffset[ui];
  const int nterms = M.ion_ncoolingterms[ui];
  const double *cellcontrib = env.K.cooling_contrib + (krow(env, c) * M.ncoolingterms);
  const double *contribs = cellcontrib + ionstart;
  const uint32_t u_process = rng_u24(p);
  const double rnd_process = rng_u24_value(u_process) * contribs[nterms - 1];
  const int ionoffset = (M.nguide > 0) ? guided_upper_bound(contribs, nterms, rnd_process, guide + M.ion_guideoff[ui], M.ion_guideshift[ui], u_process)
                        : ARTIS_KPKT_BLOCKED_SEARCH ? upper_bound_blocked<16>(contribs, nterms, rnd_process)
                                                    : upper_bound_d(contribs, nterms, rnd_process);
  if (!(ionoffset < nterms)) {
    fail(env, 71);
    return;
  }
  const int i = ionstart + ionoffset;
  const int ctype = M.coolinglist_type[i];
  const float T_e = env.C.Te[c];
  PROF_MARK(env, 57);  // ... the cooling term drawn
  if (ctype == ARTIS_COOLING_FREEFREE) {
    p.nu_cmf = -KB * T_e / HPLANCK * log((double)rng_uniform_pos(p));
    emit_rpkt(env, p, pi);
    ARTIS_STAT(env, ARTIS_STAT_K_TO_R_FF);
    thermal_emission_flags(env, p, pi, ARTIS_EMTYPE_FREEFREE);
#if ARTIS_OPT_VPKT_ON
    trace_vpkts(env, p, pi, ARTIS_TYPE_KPKT);  // kpkt.cc:515
#endif
  } else if (ctype == ARTIS_COOLING_FREEBOUND) {
    // the frequency sampling (an adaptive quadrature) runs in the slow-path kernel: kpkt_fb_emission()
    p.pend = PEND_KPKT_FB;
    p.ma_element = element;
    p.ma_ion = ion;
    p.ma_line = M.coolinglist_phixstargetindex[i];
    p.pend_arg = M.coolinglist_level[i];
  } else if (ctype == ARTIS_COOLING_COLLEXC) {
    // kpkt.cc:455-476: the reference adds the level's collisional-excitation terms to contrib_low one by one until the
    // sum exceeds rnd_process. The running sums are not kept: the level's record holds them as 15-bit fractions of the
    // level's span [lo, hi) of the ion's list (populate_coolfilter_line); a draw within the filter's resolution of an entry
    // re-adds the terms (kpkt_collexc_exact).
    const int start = M.ion_uniquelevelindexstart[ui];
    const int ul = start + M.coolinglist_level[i];
    const LevelPack lpk = M.level_pack[ul];
    const int nup = lpk.nup;
    const double lo = (ionoffset > 0) ? contribs[ionoffset - 1] : 0., hi = contribs[ionoffset];
    const int nsearch = nup - 1;  // the last sum is hi itself: > rnd_process
    int first = 0;                // first transition whose running sum is greater than rnd_process = the number of sums <= it
    if (nsearch > 0) {
      // (a cold level without a record in this cell: decided on the re-added sums like a draw the filter cannot decide)
      const int rslot = ma_resolve(env, c, lpk.rec_off);
      const U4 *rec = (rslot >= 0) ? env.K.macache + (krow(env, c) * M.nmacache) + rslot : env.K.ma_pool + ((int64_t)(rslot < -1 ? -(rslot + 2) : 0) * MAPOOL_UNIT);
      const double y = ((rnd_process - lo) / (hi - lo)) * MAFILT_SCALE;
      bool amb = env.ma_filters_off != 0 || rslot == MA_REC_NONE || !(y >= 0. && y < MAFILT_SCALE);
      const int yi = amb ? 0 : (int)y;
      for (int b0 = 0; b0 < nsearch && !amb; b0 += MAREC_PER) {
#if defined(ARTIS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) ARTIS_STAT(env, 58);  // wave-level rounds of this scan
#endif
        const U4 f = rec[marec_slot(MADIR_COOL, b0 / MAREC_PER, lpk.ndown, lpk.nup)];
        // (y is not a 24-bit draw: an entry within one unit of it either way is left to the f64 path)
        const int cnt = mafilt_count_between(f, (yi - 2 > -1) ? yi - 2 : -1, (yi + 1 < 32767) ? yi + 1 : 32767, &amb);
        amb = amb || (f.w[3] >> 16) != MAFILT_NONE;
        if (amb) break;
        first += cnt;
        if (cnt < MAREC_PER) break;
      }
      if (__builtin_expect(amb, 0)) {
        if (SPLIT) {
          p.pend = PEND_KPKT_COLLEXC;
          p.ma_element = element;
          p.ma_ion = ion;
          p.pend_arg = (int)u_process;
          return;
        }
        first = kpkt_collexc_exact(env, c, lpk.alltrans_startdown + lpk.ndown, nsearch, lo, rnd_process);
      }
    }
    kpkt_collexc_activate(env, p, element, ion, lpk, first);
  } else if (ctype == ARTIS_COOLING_COLLION) {
    const int upper = phixs_upperlevel(M, lstart(M, element, ion) + M.coolinglist_level[i], M.coolinglist_phixstargetindex[i]);
    ARTIS_STAT(env, ARTIS_STAT_MA_ACTIVATION_COLLION);
    ARTIS_STAT(env, ARTIS_STAT_K_TO_MA_COLLION);
    p.trueemissiontype = ARTIS_EMTYPE_NOTSET;
    p.flags |= PKT_FLAG_TRUEEM_NAN;  // trueem_pos = NaN (kpkt.cc:483)
    const MAState ma = {element, ion + 1, upper, -99};
    ma_activate(p, ma, 0);
  } else {
    fail(env, 73);
  }
  PROF_MARK(env, 41);  // ... the term's process carried out (slot 59 is the macro-atom stage clock of -DARTIS_PROFILE_MA, 58 counts scan rounds)
}

// free-bound emission of a k-packet, kpkt.cc:518-542
AHD void kpkt_fb_emission(const Env &env, Pkt &p, int64_t pi, FbSel *sel = nullptr) {
  const DevModel &M = env.M;
  const int c = M.propcell_nonemptymgi[p.cellindex];
  const int element = p.ma_element, ion = p.ma_ion, lowerlevel = p.pend_arg, t = p.ma_line;
  p.pend = PEND_NONE;
  p.ma_line = -99;
  p.nu_cmf = select_continuum_nu_sel(env, element, ion, lowerlevel, t, env.C.Te[c], p, sel);
  if (sel != nullptr && sel->mode == 1) return;
  emit_rpkt(env, p, pi);
  ARTIS_STAT(env, ARTIS_STAT_K_TO_R_FB);
  thermal_emission_flags(env, p, pi, emtype_continuum(M, lstart(M, element, ion) + lowerlevel, t));
#if ARTIS_OPT_VPKT_ON
  trace_vpkts(env, p, pi, ARTIS_TYPE_KPKT);  // kpkt.cc:541
#endif
}

// ---------------------------------------------------------------- gammapkt.cc / gammapkt.h, classic preset:
// GAMMA_THERMALISATION_SCHEME FREQUENCYDEPENDENT, no grey opacity, USE_XCOM_GAMMAPHOTOION off,
// PARTICLE_THERMALISATION_SCHEME INSTANTFULLDEPOSITION (artisoptions_classic.h:144-150)
AHD double sigma_compton_partial(double x, double f_max) {  // gammapkt.h:28
  const double term1 = ((x * x) - (2 * x) - 2) * log(f_max) / x / x;
  const double term2 = (((f_max * f_max) - 1) / (f_max * f_max)) / 2;
  const double term3 = ((f_max - 1) / x) * ((1 / x) + (2 / f_max) + (1 / (x * f_max)));
  return (3 * SIGMA_T * (term1 + term2 + term3) / (8 * x));
}
AHD double choose_f(double xx, double zrand) {  // gammapkt.h:38
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
AHD double meanf_sigma(double x) {  // gammapkt.h:68
  if (x < THOMSON_LIMIT) {
    double series = -409088. / 165.;  // Horner evaluation of the eight Taylor coefficients, highest order first
    series = (14588. / 15.) + (x * series);
    series = (-2584. / 7.) + (x * series);
    series = (940. / 7.) + (x * series);
    series = (-1616. / 35.) + (x * series);
    series = (147. / 10.) + (x * series);
    series = (-21. / 5.) + (x * series);
    series = 1. + (x * series);
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
AHD double chi_compton_cmf(const Env &env, int c, double nu_cmf) {  // gammapkt.cc:265
  if (ARTIS_OPT_GAMMA_USE_KAPPA_GREY) return 0.;
  const double xx = HPLANCK * nu_cmf / ME / CLIGHT / CLIGHT;
  const double sigma_cmf = (xx < THOMSON_LIMIT) ? SIGMA_T : sigma_compton_partial(xx, 1 + (2 * xx));
  return sigma_cmf * env.C.nnetot[c];
}
AHD double chi_photo_electric_cmf(const Env &env, int c, double ffegrp, double nu_cmf) {  // gammapkt.cc:416
  const double rho = env.C.rho[c];
  if (ARTIS_OPT_GAMMA_USE_KAPPA_GREY) return ARTIS_OPT_GAMMA_KAPPA_GREY * rho;
#if ARTIS_OPT_USE_XCOM_GAMMAPHOTOION
  {  // gammapkt.cc:443-495: the tabulated XCOM cross sections of every element, linear in log10-log10
    (void)ffegrp;
    const DevModel &M = env.M;
    const double hnu_over_1MeV = nu_cmf / NU_1MEV;
    const double log10_hnu_over_1MeV = log10(hnu_over_1MeV);
    double chi_cmf = 0.;
    for (int e = 0; e < M.nelements; e++) {
      const int s0 = M.xcom_elem_start[e], numb_energies = M.xcom_elem_start[e + 1] - s0;
      if (numb_energies == 0) continue;
      const double n_i = elem_numberdens(M, env.C, c, e);
      if (n_i == 0) continue;
      const double *E = M.xcom_energy + s0;
      const double *S = M.xcom_sigma + s0;
      int idx_above = -1;
      for (int j = 0; j < numb_energies; j++) {
        if (E[j] > hnu_over_1MeV) {
          idx_above = j;
          break;
        }
      }
      if (idx_above == 0) {
        chi_cmf += S[0] * n_i;
        continue;
      }
      if (idx_above == -1) {
        chi_cmf += S[numb_energies - 1] * n_i;
        continue;
      }
      const int idx_below = idx_above - 1;
      const double log10_E_above = log10(E[idx_above]);
      const double log10_E_below = log10(E[idx_below]);
      const double log10_sigma_below = log10(S[idx_below]);
      const double log10_sigma_above = log10(S[idx_above]);
      const double log10_sigma_interp =
          log10_sigma_below + ((log10_sigma_above - log10_sigma_below) / (log10_E_above - log10_E_below) * (log10_hnu_over_1MeV - log10_E_below));
      const double sigma_interp = pow(10., log10_sigma_interp);
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
AHD double sigma_pair_prod_factor(double nu_cmf) {  // gammapkt.cc:501
  const double hnu_over_1MeV = nu_cmf / NU_1MEV;
  if (nu_cmf > NU_1P5MEV) return 0.0481 + (0.301 * (hnu_over_1MeV - 1.5));
  return 0.10063 * (hnu_over_1MeV - 1.022);
}
AHD double chi_pair_prod_cmf(const Env &env, int c, double ffegrp, double nu_cmf) {  // gammapkt.cc:516
  if (ARTIS_OPT_GAMMA_USE_KAPPA_GREY) return 0.;
  const double rho = env.C.rho[c];
  if (nu_cmf <= NU_1P022MEV) return 0.;
  const double sigma_factor = sigma_pair_prod_factor(nu_cmf);
  const double sigma_cmf_si = sigma_factor * 196.e-27;
  const double sigma_cmf_fe = sigma_factor * 784.e-27;
  const double chi_cmf_si = sigma_cmf_si * (rho / MH / 28);
  const double chi_cmf_fe = sigma_cmf_fe * (rho / MH / 56);
  return dmax((chi_cmf_fe * ffegrp) + (chi_cmf_si * (1. - ffegrp)), 0.);
}
AHD double chi_cmf_loss_weighted(const Env &env, int c, double nu_cmf) {  // gammapkt.cc:548
  const double ffegrp = env.C.ffegrp[c];
  const double chi_pe = chi_photo_electric_cmf(env, c, ffegrp, nu_cmf);
  if (ARTIS_OPT_GAMMA_USE_KAPPA_GREY) return chi_pe;  // every interaction deposits the whole packet, gammapkt.cc:553
  const double xx = HPLANCK * nu_cmf / ME / CLIGHT / CLIGHT;
  const double chi_pp = chi_pair_prod_cmf(env, c, ffegrp, nu_cmf);
  return ((meanf_sigma(xx) * env.C.nnetot[c]) + chi_pe + (chi_pp * (1. - (NU_1P022MEV / nu_cmf))));
}
AHD void update_gamma_dep(const Env &env, const Pkt &p, int c, double dist) {  // gammapkt.cc:568
  if (!(dist > 0)) return;
  if (ARTIS_GAMMAPRODUCTS) return;  // the particles the gamma rays produce deposit instead, gammapkt.cc:572
  if (c < 0) return;
  const double doppler_sq = pow2(doppler(p));
  const double heating_cont = chi_cmf_loss_weighted(env, c, p.nu_cmf) * p.e_rf * dist * doppler_sq;
  cellest_add(env, env.E.dep_estimator_gamma, CELLEST_DEPGAMMA, c, heating_cont);
}
AHD double thomson_angle(Pkt &p) {  // gammapkt.cc:284
  const double B_coeff = (8. * rng_uniform(p)) - 4.;
  const double t_coeff = cbrt((sqrt(pow2(B_coeff) + 4) - B_coeff) / 2);
  return (1 / t_coeff) - t_coeff;
}
AHD void scatter_dir(const double dir_in[3], double cos_theta, Pkt &p, double dir_out[3]) {  // gammapkt.cc:297
  const double phi = rng_uniform(p) * 2 * PI;
  const double sin_theta_sq = 1. - pow2(cos_theta);
  const double sin_theta = sqrt(sin_theta_sq);
  const double zprime = cos_theta;
  double sphi, cphi;
  sin_cos(phi, &sphi, &cphi);
  const double xprime = sin_theta * cphi;
  const double yprime = sin_theta * sphi;
  if (fabs(dir_in[2]) > 0.999999999) {
    dir_out[0] = xprime;
    dir_out[1] = yprime;
    dir_out[2] = (dir_in[2] > 0) ? zprime : -zprime;
    return;
  }
  const double norm1 = 1. / sqrt(pow2(dir_in[0]) + pow2(dir_in[1]));
  const double norm2 = 1. / vlen(dir_in);
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
AHD void compton_scatter(const Env &env, Pkt &p, int64_t pi) {  // gammapkt.cc:346
  const double xx = HPLANCK * p.nu_cmf / ME / CLIGHT / CLIGHT;
  double f = 1.;
  bool stay_gamma = true;
  if (xx >= THOMSON_LIMIT) {
    f = choose_f(xx, rng_uniform(p));
    const double prob_gamma = 1. / f;
    stay_gamma = (rng_uniform(p) < prob_gamma);
  }
  if (stay_gamma) {
    p.nu_cmf = p.nu_cmf / f;
    const double pos[3] = {p.px, p.py, p.pz};
    const double dir[3] = {p.dx, p.dy, p.dz};
    const double vel_vec[3] = {pos[0] / p.prop_time, pos[1] / p.prop_time, pos[2] / p.prop_time};  // get_velocity vectors.h:50
    double cmf_dir[3], new_dir[3], out[3];
    angle_ab(dir, vel_vec, cmf_dir);
    const double cos_theta = (xx < THOMSON_LIMIT) ? thomson_angle(p) : 1. - ((f - 1) / xx);
    scatter_dir(cmf_dir, cos_theta, p, new_dir);
    const double negvel[3] = {vel_vec[0] * -1., vel_vec[1] * -1., vel_vec[2] * -1.};
    angle_ab(new_dir, negvel, out);
    p.dx = out[0];
    p.dy = out[1];
    p.dz = out[2];
    set_restframe_from_cmf(p);
  } else {
#if ARTIS_GAMMAPRODUCTS
    p.nu_cmf = p.nu_cmf * (1 - (1 / f));  // the gamma's energy loss is the electron's energy, gammapkt.cc:404
    p.type = ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS;
#else
    p.type = ARTIS_TYPE_NTLEPTON_DEPOSITED;
#endif
    p.absorptiontype = ARTIS_ABSTYPE_GAMMA_COMPTON;
    ARTIS_STAT(env, ARTIS_STAT_NT_FROM_GAMMA);
  }
}
AHD void emit_gamma_isotropic(Pkt &p) {  // gammapkt.cc:603
  double dir_cmf[3], out[3];
  rand_isotropic(p, dir_cmf);
  const double mt = -p.prop_time;
  const double vel_vec[3] = {p.px / mt, p.py / mt, p.pz / mt};
  angle_ab(dir_cmf, vel_vec, out);
  p.dx = out[0];
  p.dy = out[1];
  p.dz = out[2];
  set_restframe_from_cmf(p);
  p.type = ARTIS_TYPE_GAMMA;
}
AHD void pair_production(const Env &env, Pkt &p, int64_t pi) {  // gammapkt.cc:618
  const double pair_rest_mass_energy = 1.022 * MEV;
  const double gamma_energy = HPLANCK * p.nu_cmf;
  const double prob_gamma = pair_rest_mass_energy / gamma_energy;
  if (rng_uniform(p) > prob_gamma) {
#if ARTIS_GAMMAPRODUCTS
    const double particle_kinetic_energy = (gamma_energy - pair_rest_mass_energy) / 2;  // gammapkt.cc:630
    p.nu_cmf = particle_kinetic_energy / HPLANCK;
    p.type = (rng_uniform(p) > 0.5) ? ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS : ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAPLUS;
#else
    p.type = ARTIS_TYPE_NTLEPTON_DEPOSITED;
#endif
    p.absorptiontype = ARTIS_ABSTYPE_GAMMA_PAIRPRODUCTION;
    ARTIS_STAT(env, ARTIS_STAT_NT_FROM_GAMMA);
  } else {
    p.nu_cmf = 0.511 * MEV / HPLANCK;
    emit_gamma_isotropic(p);
  }
}
// transport_gamma gammapkt.cc:655 + do_gamma gammapkt.cc:911
#if ARTIS_OPT_GAMMA_THERMALISATION_SCHEME != ARTIS_GAMMA_FREQUENCYDEPENDENT
// column density times kappa along the ray of q (a copy of the packet) out of the grid: the loop of
// wollaeger_thermalisation() / guttman_thermalisation() (gammapkt.cc:805-828, :846-858)
AHD double gamma_ray_tau(const Env &env, Pkt q, double mean_gamma_opac) {
  const DevModel &M = env.M;
  double tau = 0.;
  int guard = 0;
  while (q.type != ARTIS_TYPE_ESCAPE) {
    int next_cell = -1;
    const double boundarydist = boundary_distance(env, q, &next_cell);
    const int c = M.propcell_nonemptymgi[q.cellindex];
    if (c >= 0) {
      const double rho = M.rho_tmin[c] * pow3(M.tmin / q.prop_time);  // the density when the ray reaches the cell
      tau += mean_gamma_opac * rho * boundarydist;
    }
    move_pkt(q, boundarydist);
    // change_cell_or_escape(pkt_copy, next_cellindex, false): no counters, nothing recorded
    if (next_cell >= 0) {
      if (next_cell != q.cellindex && M.gridtype == ARTIS_GRID_CARTESIAN3D) {
        double *pos[3] = {&q.px, &q.py, &q.pz};
        for (int d = 0; d < 3; d++) {
          const int idx = coordidx(M, next_cell, d);
          const double lo = M.coord_pos_min_tmin[d][idx] / M.tmin * q.prop_time;
          const double hi = (idx < (M.ncoordgrid[d] - 1)) ? M.coord_pos_min_tmin[d][idx + 1] / M.tmin * q.prop_time
                                                           : M.rmax / M.tmin * q.prop_time;
          *pos[d] = dclamp(*pos[d], lo, hi);
        }
      }
      q.cellindex = next_cell;
    } else {
      q.type = ARTIS_TYPE_ESCAPE;
    }
    if (++guard > 1000000) {
      fail(env, 97);
      break;
    }
  }
  return tau;
}
// do_gamma gammapkt.cc:911 with a parameterised thermalisation scheme: no transport, the packet is absorbed where it is
// with the scheme's probability (absorb_or_escape_gamma :754) or leaves the grid
AHD void do_gamma(const Env &env, Pkt &p, int64_t pi) {
  ARTIS_STAT(env, ARTIS_STAT_X_GAMMA_STEPS);
  double f_gamma;
#if ARTIS_OPT_GAMMA_THERMALISATION_SCHEME == ARTIS_GAMMA_BARNES
  {  // barnes_thermalisation gammapkt.cc:779
    const double E_kin = env.M.ejecta_kinetic_energy;
    const double v_ej = sqrt(E_kin * 2 / env.M.mtot_input);
    const double t_ineff = 1.4 * DAY * sqrt(env.M.mtot_input / (5.e-3 * MSUN)) * ((0.2 * CLIGHT) / v_ej);
    const double tau = pow2(t_ineff / p.prop_time);
    f_gamma = 1. - exp(-tau);
  }
#elif ARTIS_OPT_GAMMA_THERMALISATION_SCHEME == ARTIS_GAMMA_WOLLAEGER
  {  // wollaeger_thermalisation gammapkt.cc:797: the optical depth radially outwards
    Pkt q = p;
    const double pos[3] = {p.px, p.py, p.pz};
    const double mag = vlen(pos);
    q.dx = pos[0] / mag;  // vec_norm vectors.h:31
    q.dy = pos[1] / mag;
    q.dz = pos[2] / mag;
    f_gamma = 1. - exp(-gamma_ray_tau(env, q, 0.1));
  }
#else
  {  // guttman_thermalisation gammapkt.cc:831: the deposition probability averaged over 100 random directions
    double deposition_probability_sum = 0.;
    for (int i = 0; i < 100; i++) {
      Pkt q = p;
      double dir[3];
      rand_isotropic(p, dir);  // drawn from the packet's own generator
      q.dx = dir[0];
      q.dy = dir[1];
      q.dz = dir[2];
      deposition_probability_sum -= expm1(-gamma_ray_tau(env, q, 0.03));
    }
    f_gamma = deposition_probability_sum / 100;
  }
#endif
  if (!(f_gamma >= 0.) || !(f_gamma <= 1.)) fail(env, 98);
  if (rng_uniform(p) < f_gamma) {
    p.type = ARTIS_TYPE_NTLEPTON_DEPOSITED;
    p.absorptiontype = ARTIS_ABSTYPE_GAMMA_PHOTOELECTRIC;  // the schemes do not resolve the process (gammapkt.cc:761)
  } else {
    change_cell_or_escape(env, p, pi, -99);  // escape_type stays TYPE_GAMMA
  }
  if (p.type != ARTIS_TYPE_GAMMA && p.type != ARTIS_TYPE_ESCAPE) {
    if (!ARTIS_GAMMAPRODUCTS) scalar_add(env, ARTIS_SCALAR_GAMMA_DEP_DISCRETE, p.e_cmf);
    const int c = env.M.propcell_nonemptymgi[p.cellindex];  // no transport: the path estimator is fed here (gammapkt.cc:930)
    if (c >= 0) ARTIS_EST_ADD(&env.E.dep_estimator_gamma[(int64_t)c * env.est_stride], p.e_cmf);
  }
}
#else
AHD void do_gamma(const Env &env, Pkt &p, int64_t pi) {
  const double t2 = env.S.ts_end;
  ARTIS_STAT(env, ARTIS_STAT_X_GAMMA_STEPS);
  const double tau_next = -log((double)rng_uniform_pos(p));
  int next_cell = -1;
  const double boundarydist = boundary_distance(env, p, &next_cell);
  const int c = env.M.propcell_nonemptymgi[p.cellindex];
  const double dop = doppler(p);
  const double ffegrp = (c >= 0) ? (double)env.C.ffegrp[c] : 0.;
  const double chi_compton = (c >= 0) ? chi_compton_cmf(env, c, p.nu_cmf) * dop : 0.;
  const double chi_pe = (c >= 0) ? chi_photo_electric_cmf(env, c, ffegrp, p.nu_cmf) * dop : 0.;
  const double chi_pp = (c >= 0) ? chi_pair_prod_cmf(env, c, ffegrp, p.nu_cmf) * dop : 0.;
  const double chi_tot = chi_compton + chi_pe + chi_pp;
  const double edist = chi_tot > 0. ? tau_next / chi_tot : DBLMAX;
  if (!(edist >= 0)) fail(env, 80);
  const double tdist = (t2 - p.prop_time) * CLIGHT_PROP;
  if (!(tdist >= 0)) fail(env, 81);
  if ((boundarydist <= tdist) && (boundarydist <= edist)) {
    move_pkt(p, boundarydist / 2.);
    if (chi_tot > 0) update_gamma_dep(env, p, c, boundarydist);
    move_pkt(p, boundarydist / 2.);
    if (next_cell != p.cellindex) change_cell_or_escape(env, p, pi, next_cell);
  } else if ((tdist < boundarydist) && (tdist <= edist)) {
    move_pkt(p, tdist / 2.);
    if (chi_tot > 0) update_gamma_dep(env, p, c, tdist);
    move_pkt(p, tdist / 2.);
    p.prop_time = t2;
  } else if ((edist < boundarydist) && (edist <= tdist)) {
    move_pkt(p, edist / 2.);
    if (chi_tot > 0) update_gamma_dep(env, p, c, edist);
    move_pkt(p, edist / 2.);
    const double chi_rnd = rng_uniform(p) * chi_tot;
    if (chi_compton > chi_rnd) {
      compton_scatter(env, p, pi);
    } else if ((chi_compton + chi_pe) > chi_rnd) {
      p.type = ARTIS_GAMMAPRODUCTS ? ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS : ARTIS_TYPE_NTLEPTON_DEPOSITED;  // gammapkt.cc:734
      p.absorptiontype = ARTIS_ABSTYPE_GAMMA_PHOTOELECTRIC;
      ARTIS_STAT(env, ARTIS_STAT_NT_FROM_GAMMA);
    } else {
      pair_production(env, p, pi);
    }
  } else {
    fail(env, 82);
  }
  if (p.type != ARTIS_TYPE_GAMMA && p.type != ARTIS_TYPE_ESCAPE && !ARTIS_GAMMAPRODUCTS)
    scalar_add(env, ARTIS_SCALAR_GAMMA_DEP_DISCRETE, p.e_cmf);  // gammapkt.cc:926
}
#endif
// nonthermal::do_ntlepton_deposit nonthermal.cc:2529. NT_ON == false (artisoptions_classic.h:95): every deposit is heat.
// NT_ON with a Spencer-Fano solution (artisoptions_nltenebular.h:102-104): the deposit ionises or excites a macro-atom
// with the solution's fractions; the activation is recorded in the packet (ma_activate) and the walk runs in the thermal
// kernel, the packet keeping its deposit type until the macro-atom deactivates (as in the reference's do_macroatom()).
AHD void do_ntlepton_deposit(const Env &env, Pkt &p) {
  scalar_add(env, ARTIS_SCALAR_NT_ENERGY_DEPOSITED, p.e_cmf);
#if ARTIS_OPT_NT_ON
  const DevModel &M = env.M;
  const int c = M.propcell_nonemptymgi[p.cellindex];
  if (env.C.thick[c] != ARTIS_CELL_THICK) {
    double zrand = rng_uniform(p);
    const double frac_ionisation = env.C.nt_frac_ionisation[c];
    if (zrand < frac_ionisation) {
      // select_nt_ionisation nonthermal.cc:1537 over the running sums of populate_nt_cell()
      const double *cum = env.C.nt_ionenrate_cum + ((int64_t)c * M.nions);
      const double ratetotal = cum[M.nions - 1];
      if (ratetotal > 0.) {
        const double target = rng_uniform(p) * ratetotal;
        int element = -1, lowerion = -1;
        for (int e = 0; e < M.nelements && element < 0; e++) {
          for (int ion = 0; ion < M.elem_nions[e] - 1; ion++) {
            if (cum[uion(M, e, ion)] > target) {
              element = e;
              lowerion = ion;
              break;
            }
          }
        }
        if (element < 0) {
          fail(env, 93);
          return;
        }
        const int upperion = nt_random_upperion(env, p, c, element, lowerion, true);
        ARTIS_STAT(env, ARTIS_STAT_MA_ACTIVATION_NTCOLLION);
        ARTIS_STAT(env, ARTIS_STAT_INTERACTIONS);
        p.trueemissiontype = ARTIS_EMTYPE_NOTSET;
        p.flags |= PKT_FLAG_TRUEEM_NAN;
        ARTIS_STAT(env, ARTIS_STAT_NT_TO_IONISATION);
        const MAState ma = {element, upperion, 0, -99};
        ma_activate(p, ma, 0);
        return;
      }
      p.type = ARTIS_TYPE_KPKT;
      ARTIS_STAT(env, ARTIS_STAT_NT_TO_KPKT);
      return;
    }
    const double frac_excitation = ARTIS_OPT_NT_EXCITATION_ON ? env.C.nt_frac_excitation[c] : 0.;
    if (zrand < (frac_ionisation + frac_excitation)) {
      zrand -= frac_ionisation;
      const int64_t base = (int64_t)c * env.C.nt_excitations_stored;
      const int n = env.C.nt_exc_count[c];
      for (int i = 0; i < n; i++) {
        const double frac_deposition_exc = env.C.nt_exc_frac_deposition[base + i];
        if (zrand < frac_deposition_exc) {
          const int lineindex = M.alltrans_lineindex[env.C.nt_exc_alltransindex[base + i]];
          const int element = M.line_elementindex[lineindex];
          const int ion = M.line_ionindex[lineindex];
          const int upper = M.line_pack[lineindex].upper - lstart(M, element, ion);  // get_levelfromuniquelevelindex
          ARTIS_STAT(env, ARTIS_STAT_MA_ACTIVATION_NTCOLLEXC);
          ARTIS_STAT(env, ARTIS_STAT_INTERACTIONS);
          p.trueemissiontype = ARTIS_EMTYPE_NOTSET;
          p.flags |= PKT_FLAG_TRUEEM_NAN;
          ARTIS_STAT(env, ARTIS_STAT_NT_TO_EXCITATION);
          const MAState ma = {element, ion, upper, -99};
          ma_activate(p, ma, 0);
          return;
        }
        zrand -= frac_deposition_exc;
      }
    }
  }
#endif
  p.type = ARTIS_TYPE_KPKT;
  ARTIS_STAT(env, ARTIS_STAT_NT_TO_KPKT);
}

// nonthermal::do_ntalpha_fisprod_deposit nonthermal.cc:2520
AHD void do_ntalpha_fisprod_deposit(const Env &env, Pkt &p) {
  scalar_add(env, ARTIS_SCALAR_NT_ENERGY_DEPOSITED, p.e_cmf);
  p.type = ARTIS_TYPE_KPKT;
  ARTIS_STAT(env, ARTIS_STAT_NT_TO_KPKT);
}
// do_nonthermal_predeposit update_packets.cc:42 (INSTANTFULLDEPOSITION, TIMEDEPENDENT, TIMEDEPENDENT_WITH_ADIABATIC_LOSS)
AHD void do_nonthermal_predeposit(const Env &env, Pkt &p, int64_t pi) {
  double e_cmf_deposited = p.e_cmf;
  const int c = env.M.propcell_nonemptymgi[p.cellindex];
  const int priortype = p.type;
  const int deposit_type = (priortype == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA) ? ARTIS_TYPE_NTALPHA_FISPROD_DEPOSITED : ARTIS_TYPE_NTLEPTON_DEPOSITED;
#if ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_INSTANTFULLDEPOSITION
  p.type = deposit_type;  // absorption happens
#elif ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_BARNES || ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_WOLLAEGER
  {  // analytic thermalisation efficiency f_p: deposit with probability f_p, else the particle escapes (update_packets.cc:53-88)
    const double ts = p.prop_time;
    double f_p;
    if (ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_BARNES) {
      const double E_kin = env.M.ejecta_kinetic_energy;
      const double v_ej = sqrt(E_kin * 2 / env.M.mtot_input);
      const double prefactor = (priortype == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA) ? 7.74 : 7.4;
      const double tau_ineff = prefactor * DAY * sqrt(env.M.mtot_input / (5.e-3 * MSUN)) * pow((0.2 * CLIGHT) / v_ej, 3. / 2.);
      f_p = log1p(2. * ts * ts / tau_ineff / tau_ineff) / (2. * ts * ts / tau_ineff / tau_ineff);
    } else {
      const double A = (priortype == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA) ? 1.2 * 1.e-11 : 1.3 * 1.e-11;
      const double aux_term = 2 * A / (ts * env.C.rho[c]);
      f_p = log1p(aux_term) / aux_term;
    }
    if (!(f_p >= 0.) || !(f_p <= 1.)) fail(env, 94);
    if (rng_uniform(p) < f_p) {
      p.type = deposit_type;
    } else {
      e_cmf_deposited = 0.;
      change_cell_or_escape(env, p, pi, -99);  // escape_type keeps the particle type (grid.h:130)
    }
  }
#else
  {  // local time-dependent absorption, update_packets.cc:90-150
    const double ts = p.prop_time;
    const double ts_end = env.S.ts_end;
    const double rho = env.C.rho[c];
    const double particle_en = HPLANCK * p.nu_cmf;
    const double endot_collisional = (priortype == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA) ? 5.e11 * MEV * rho : 4.e10 * MEV * rho;
    const double endot_adiabatic =
        (ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_TIMEDEPENDENT_WITH_ADIABATIC_LOSS) ? particle_en / ts : 0.;
    const double endot = endot_collisional + endot_adiabatic;
    e_cmf_deposited = p.e_cmf * endot_collisional * dmin(ts_end - ts, particle_en / endot) / particle_en;
    const double rnd_en_absorb = rng_uniform(p) * particle_en;
    const double t_absorb = ts + (rnd_en_absorb / endot);
    const double t_new = dmin(t_absorb, ts_end);
    const bool absorbed = (t_absorb <= ts_end);
    if (absorbed) {
      p.type = deposit_type;
    } else {
      p.nu_cmf -= (endot * (ts_end - ts)) / HPLANCK;
    }
    const double scale = t_new / ts;
    p.px = p.px * scale;
    p.py = p.py * scale;
    p.pz = p.pz * scale;
    p.prop_time = t_new;
    if (ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_TIMEDEPENDENT_WITH_ADIABATIC_LOSS && absorbed)
      p.e_cmf *= endot_collisional / endot;
  }
#endif
  if (env.P.cold[pi].originated_particle != 0) {
    if (priortype == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS) {
      ARTIS_EST_ADD(&env.E.dep_estimator_electron[(int64_t)c * env.est_stride], e_cmf_deposited);
      if (p.type == deposit_type) scalar_add(env, ARTIS_SCALAR_ELECTRON_DEP_DISCRETE, p.e_cmf);
    } else if (priortype == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAPLUS) {
      ARTIS_EST_ADD(&env.E.dep_estimator_positron[(int64_t)c * env.est_stride], e_cmf_deposited);
      if (p.type == deposit_type) scalar_add(env, ARTIS_SCALAR_POSITRON_DEP_DISCRETE, p.e_cmf);
    } else if (priortype == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA) {
      ARTIS_EST_ADD(&env.E.dep_estimator_alpha[(int64_t)c * env.est_stride], e_cmf_deposited);
      if (p.type == deposit_type) scalar_add(env, ARTIS_SCALAR_ALPHA_DEP_DISCRETE, p.e_cmf);
    }
  } else if (ARTIS_GAMMAPRODUCTS) {  // update_packets.cc:174: products of gamma rays count as gamma deposition
    ARTIS_EST_ADD(&env.E.dep_estimator_gamma[(int64_t)c * env.est_stride], e_cmf_deposited);
    if (p.type == ARTIS_TYPE_NTLEPTON_DEPOSITED) scalar_add(env, ARTIS_SCALAR_GAMMA_DEP_DISCRETE, p.e_cmf);
  }
}
// update_pellet update_packets.cc:185 with pellet_gamma_decay gammapkt.cc:894
AHD void update_pellet(const Env &env, Pkt &p, int64_t pi) {
  const double t2 = env.S.ts_end;
  const double ts = p.prop_time;
  const double tdecay = env.P.cold[pi].tdecay;
  if (tdecay > t2) {
    const double scale = t2 / ts;
    p.px = p.px * scale;
    p.py = p.py * scale;
    p.pz = p.pz * scale;
    p.prop_time = t2;
  } else if (tdecay > ts) {
    scalar_add(env, ARTIS_SCALAR_PELLET_DECAYS, 1.);
    p.prop_time = tdecay;
    const double scale = tdecay / ts;
    p.px = p.px * scale;
    p.py = p.py * scale;
    p.pz = p.pz * scale;
    if (env.P.cold[pi].originated_particle != 0) {
      const int decaytype = env.P.cold[pi].pellet_decaytype;
      if (decaytype == ARTIS_DECAYTYPE_BETAPLUS) {
        p.type = ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAPLUS;
        scalar_add(env, ARTIS_SCALAR_POSITRON_EMISSION, p.e_cmf);
      } else if (decaytype == ARTIS_DECAYTYPE_BETAMINUS) {
        p.type = ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS;
        scalar_add(env, ARTIS_SCALAR_ELECTRON_EMISSION, p.e_cmf);
      } else if (decaytype == ARTIS_DECAYTYPE_ALPHA) {
        scalar_add(env, ARTIS_SCALAR_ALPHA_EMISSION, p.e_cmf);
        p.type = ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA;
      } else if (decaytype == ARTIS_DECAYTYPE_SPONTFISSION) {
        scalar_add(env, ARTIS_SCALAR_SPFISSION_DEP_DISCRETE, p.e_cmf);
        p.type = ARTIS_TYPE_NTALPHA_FISPROD_DEPOSITED;
      } else {
        fail(env, 90);
        p.type = ARTIS_TYPE_ESCAPE;  // leave the work lists
        return;
      }
      env.P.flight[pi].em_time = (float)p.prop_time;
      p.absorptiontype = ARTIS_ABSTYPE_PELLET_PARTICLEDECAY;
    } else {
      scalar_add(env, ARTIS_SCALAR_GAMMA_EMISSION, p.e_cmf);
      if (p.nu_cmf < 0) {  // no gamma spectrum known for the nuclide: straight to a k-packet
        p.type = ARTIS_TYPE_KPKT;
        p.absorptiontype = ARTIS_ABSTYPE_PELLET_NOGAMMASPEC;
      } else {
        emit_gamma_isotropic(p);
      }
    }
  } else if ((tdecay > 0) && (env.S.nts == 0)) {
    p.e_cmf *= tdecay / env.M.tmin;
    p.type = ARTIS_TYPE_PRE_KPKT;
    p.absorptiontype = ARTIS_ABSTYPE_PELLET_BEFORESIMSTART;
    ARTIS_STAT(env, ARTIS_STAT_K_FROM_EARLIERDECAY);
    p.prop_time = env.M.tmin;
  } else {
    fail(env, 91);
    p.type = ARTIS_TYPE_ESCAPE;
  }
}

// ---------------------------------------------------------------- packet load/store and the per-thread driver
// the packet types that do not use the cell cache (get_packet_cellcachegroupid update_packets.cc:340): handled by k_gamma
AHD bool type_gamma(int type) {
  return type == ARTIS_TYPE_GAMMA || type == ARTIS_TYPE_NTLEPTON_DEPOSITED || type == ARTIS_TYPE_NTALPHA_FISPROD_DEPOSITED ||
         type == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAMINUS || type == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_BETAPLUS ||
         type == ARTIS_TYPE_NONTHERMAL_PREDEPOSIT_ALPHA || type == ARTIS_TYPE_RADIOACTIVE_PELLET;
}
AHD bool type_handled(int type) {
  return type == ARTIS_TYPE_RPKT || type == ARTIS_TYPE_KPKT || type == ARTIS_TYPE_PRE_KPKT || type_gamma(type);
}
AHD bool pkt_active(const Pkt &p, double ts_end) { return type_handled(p.type) && p.prop_time < ts_end; }  // update_packets.cc:321

// -DARTIS_PKT_NT=1 (measurement): the packet lines are read and written with non-temporal accesses -- each is touched once
// per visit, and every line of them that stays in the CU's 32 KB L1 displaces a macro-atom record's
#ifndef ARTIS_PKT_NT
#define ARTIS_PKT_NT 0
#endif
template <typename T>
AHD T pkt_line_load(const T *src) {
#if ARTIS_PKT_NT && defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned int u4v __attribute__((ext_vector_type(4)));
  T out;
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 16; k++) ((u4v *)&out)[k] = __builtin_nontemporal_load(((const u4v *)src) + k);
  return out;
#else
  return *src;
#endif
}
template <typename T>
AHD void pkt_line_store(T *dst, const T &v) {
#if ARTIS_PKT_NT && defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned int u4v __attribute__((ext_vector_type(4)));
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 16; k++) __builtin_nontemporal_store(((const u4v *)&v)[k], ((u4v *)dst) + k);
#else
  *dst = v;
#endif
}
// Hot line <-> registers: everything a thermal packet needs; 128 B, one cache line, eight 16-byte accesses.
AHD void pkt_load_hot(const PktStore &P, int64_t i, Pkt &p) {
  const PktHot h = pkt_line_load(&P.hot[i]);
  p.s0 = h.rng[0]; p.s1 = h.rng[1]; p.s2 = h.rng[2]; p.s3 = h.rng[3];
  p.prop_time = h.prop_time;
  p.px = h.pos_x; p.py = h.pos_y; p.pz = h.pos_z;
  p.e_cmf = h.e_cmf; p.nu_cmf = h.nu_cmf;
  p.type = h.type; p.cellindex = h.cellindex; p.next_trans = h.next_trans; p.nscatterings = h.nscatterings;
  p.ma_element = h.ma_element; p.ma_ion = h.ma_ion; p.ma_level = h.ma_level; p.ma_line = h.ma_line;
  p.ma_origin = h.ma_origin; p.pend = h.pend; p.pend_arg = h.pend_arg; p.chi_mgi = h.chi_mgi;
  p.emissiontype = h.emissiontype; p.trueemissiontype = h.trueemissiontype; p.absorptiontype = h.absorptiontype;
  p.flags = h.flags;
}
AHD void pkt_store_hot(const PktStore &P, int64_t i, const Pkt &p) {
  PktHot h;
  h.rng[0] = p.s0; h.rng[1] = p.s1; h.rng[2] = p.s2; h.rng[3] = p.s3;
  h.prop_time = p.prop_time;
  h.pos_x = p.px; h.pos_y = p.py; h.pos_z = p.pz;
  h.e_cmf = p.e_cmf; h.nu_cmf = p.nu_cmf;
  h.type = p.type; h.cellindex = p.cellindex; h.next_trans = p.next_trans; h.nscatterings = p.nscatterings;
  h.ma_element = p.ma_element; h.ma_ion = p.ma_ion; h.ma_level = p.ma_level; h.ma_line = p.ma_line;
  h.ma_origin = p.ma_origin; h.pend = p.pend; h.pend_arg = p.pend_arg; h.chi_mgi = p.chi_mgi;
  h.emissiontype = p.emissiontype; h.trueemissiontype = p.trueemissiontype; h.absorptiontype = p.absorptiontype;
  h.flags = p.flags & ~PKT_FLAG_EMITTED;
  pkt_line_store(&P.hot[i], h);
}
// Flight line: direction, rest-frame quantities, polarisation ...
AHD void pkt_load_flight(const PktStore &P, int64_t i, Pkt &p) {
  const PktFlight &f = P.flight[i];
  p.dx = f.dir_x; p.dy = f.dir_y; p.dz = f.dir_z;
  p.nu_rf = f.nu_rf; p.e_rf = f.e_rf;
  p.stokes_q = f.stokes_q; p.stokes_u = f.stokes_u;
}
AHD void pkt_store_flight(const PktStore &P, int64_t i, const Pkt &p) {
  PktFlight &f = P.flight[i];
  f.dir_x = p.dx; f.dir_y = p.dy; f.dir_z = p.dz;
  f.nu_rf = p.nu_rf; f.e_rf = p.e_rf;
  f.stokes_q = p.stokes_q; f.stokes_u = p.stokes_u;
}
// ... and the packet's ContinuumOpacity (its cell index travels in the hot line)
AHD void chi_load(const PktStore &P, int64_t i, const Pkt &p, Chi &x) {
  const PktFlight &f = P.flight[i];
  x.nu = f.chi_nu; x.chi_escatter = f.chi_es; x.chi_freefree_heat = f.chi_ff; x.chi_boundfree = f.chi_bf;
  x.nonemptymgi = p.chi_mgi;
#if ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
  x.bf_end = -1;
  x.bf_begin = 0;
#endif
}
AHD void chi_store(const PktStore &P, int64_t i, Pkt &p, const Chi &x) {
  PktFlight &f = P.flight[i];
  f.chi_nu = x.nu; f.chi_es = x.chi_escatter; f.chi_ff = x.chi_freefree_heat; f.chi_bf = x.chi_boundfree;
  p.chi_mgi = x.nonemptymgi;
}
// all register state of a packet (r-packet, gamma and slow-path kernels)
AHD void pkt_load(const PktStore &P, int64_t i, Pkt &p) {
  pkt_load_hot(P, i, p);
  pkt_load_flight(P, i, p);
}
AHD void pkt_store(const PktStore &P, int64_t i, const Pkt &p) {
  pkt_store_flight(P, i, p);
  pkt_store_hot(P, i, p);
}
// Thermal kernel: only the hot line is loaded. The flight fields of the register struct are written by emit_rpkt() when
// a k-packet or macro-atom ends in an r-packet (the only way they change here), and then -- and only then -- stored.
AHD void pkt_clear_flight(Pkt &p) {
  p.dx = 0.; p.dy = 0.; p.dz = 0.; p.nu_rf = 0.; p.e_rf = 0.; p.stokes_q = 0.; p.stokes_u = 0.;
}
AHD void pkt_load_thermal(const PktStore &P, int64_t i, Pkt &p) {
  pkt_load_hot(P, i, p);
  pkt_clear_flight(p);
}
AHD void pkt_store_thermal(const PktStore &P, int64_t i, Pkt &p) {
  if (p.flags & PKT_FLAG_EMITTED) pkt_store_flight(P, i, p);
  pkt_store_hot(P, i, p);
  p.flags &= ~PKT_FLAG_EMITTED;
}

// Work lists: a packet that still needs updating is "in flight" (an r-packet inside or about to enter do_rpkt()),
// walking a macro-atom, a k-packet (or pre-k-packet) due for its next step, or waiting for a slow-path action.
// NEXT_BB: a (pre-)k-packet whose next step is do_kpkt_blackbody() (update_packets.cc:291-300) -- it always ends in an
// r-packet, so it gets a small kernel of its own instead of widening the thermal kernel's register footprint.
enum { NEXT_DONE = 0, NEXT_RPKT = 1, NEXT_MA = 2, NEXT_SLOW = 3, NEXT_KPKT = 4, NEXT_GAMMA = 5, NEXT_BB = 6, NEXT_NKINDS = 7 };
// Sort key of a work-list entry: the packet's propagation cell and, for r-packets, a coarse comoving-frequency bin
// (four per octave over 2^46 .. 2^54 Hz = 7e13 .. 1.8e16 Hz, bluest first, like the reference's own packet order
// compare_packet_order update_packets.cc:363). Read off the exponent and top mantissa bit: placement only, never results.
#ifndef ARTIS_SORT_NUBINS
#define ARTIS_SORT_NUBINS 32  // frequency bins of an r-packet's sort key: 8 octaves x 2, 4, 8, 16 per octave = 16, 32, 64, 128 (round 5: 32, with the bin as the key's major part)
#endif
constexpr int SORT_NUBINS = ARTIS_SORT_NUBINS;
constexpr int SORT_NU_LOG2PEROCTAVE = (SORT_NUBINS == 16) ? 1 : ((SORT_NUBINS == 32) ? 2 : ((SORT_NUBINS == 64) ? 3 : ((SORT_NUBINS == 128) ? 4 : -1)));
static_assert(SORT_NU_LOG2PEROCTAVE > 0, "ARTIS_SORT_NUBINS: 16, 32, 64 or 128");
constexpr int SORT_MABINS = 16;  // sub-keys of a thermal-list entry below its cell
AHD int32_t list_sort_key(int32_t cellindex, double nu_cmf, int nbins) {
  if (nbins <= 1) return cellindex;
  union { double d; uint64_t u; } v;
  v.d = nu_cmf;
  constexpr int m = SORT_NU_LOG2PEROCTAVE;
  const int e2 = (int)((v.u >> (52 - m)) & ((1u << (11 + m)) - 1u)) - ((1023 + 46) << m);  // 2^m * (exponent - 46) + the top m mantissa bits
  const int b = (nu_cmf > 0.) ? (e2 < 0 ? 0 : (e2 > SORT_NUBINS - 1 ? SORT_NUBINS - 1 : e2)) : 0;
  return (cellindex * SORT_NUBINS) + (SORT_NUBINS - 1 - b);
}
// do_packet() sends a pre-k-packet, and a k-packet in an optically thick cell, to do_kpkt_blackbody() (update_packets.cc:291-300)
AHD bool kpkt_blackbody_case(const Env &env, int type, int cellindex) {
  if (type == ARTIS_TYPE_PRE_KPKT) return true;
  const int c = env.M.propcell_nonemptymgi[cellindex];
  return c >= 0 && env.C.thick[c] == ARTIS_CELL_THICK;
}
AHD int classify(const Env &env, const Pkt &p, double ts_end) {
  const bool active = pkt_active(p, ts_end);
  if (active && type_gamma(p.type) && p.pend == PEND_NONE && !ma_pending(p)) return NEXT_GAMMA;  // no cell cache needed
  if ((p.pend != PEND_NONE || ma_pending(p) || active) && !in_tile(env, p.cellindex)) return NEXT_DONE;  // waits for its tile
  if (p.pend != PEND_NONE) return NEXT_SLOW;
  if (ma_pending(p)) return NEXT_MA;
  if (!active) return NEXT_DONE;
  if (p.type == ARTIS_TYPE_RPKT) return NEXT_RPKT;
  return kpkt_blackbody_case(env, p.type, p.cellindex) ? NEXT_BB : NEXT_KPKT;
}

// ---- r-packet kernel body. One iteration = one call of do_rpkt_step() (rpkt.cc:542). The packet's ContinuumOpacity x is
// persistent (loaded/stored by the caller); it is invalidated whenever the reference's do_rpkt() loop would be left
// (do_rpkt_step() returned false: new model cell, type change, escape, end of timestep), which is the same as
// resetting it on entry of do_rpkt() (oracle header, note 2). A macro-atom activation leaves the loop with the state
// recorded in the packet; if the packet is an r-packet again afterwards it continues the same do_rpkt() loop.
// Returns true while the packet can take another iteration in the r-packet kernel.
AHD bool rpkt_can_continue(const Pkt &p, double ts_end) {
  return !ma_pending(p) && p.pend == PEND_NONE && p.type == ARTIS_TYPE_RPKT && p.prop_time < ts_end;
}
template <bool SPLIT = false>
AHD bool rpkt_iter(const Env &env, Pkt &p, int64_t pi, Chi &x) {
  const bool cont = do_rpkt_step<SPLIT>(env, p, pi, x, pi);
  if (!ma_pending(p) && !cont) x.nonemptymgi = -1;
  return rpkt_can_continue(p, env.S.ts_end) && in_tile(env, p.cellindex);  // a new cell may belong to another tile
}
AHD int advance_rpkt(const Env &env, Pkt &p, int64_t pi, Chi &x, int budget) {
  int steps = 0;
  bool go = rpkt_can_continue(p, env.S.ts_end) && in_tile(env, p.cellindex);
  while (go && steps < budget) {
    go = rpkt_iter(env, p, pi, x);
    steps++;
  }
  return classify(env, p, env.S.ts_end);
}

// a macro-atom has just deactivated: a packet that was an r-packet before and after continues its do_rpkt() loop with
// its ContinuumOpacity; every other outcome ends or precedes a do_rpkt() call
AHD void chi_after_ma(Pkt &p) {
  if (!ma_pending(p) && !(p.type == ARTIS_TYPE_RPKT && p.ma_origin == 1)) p.chi_mgi = -1;
}

// ---- macro-atom kernel body: one iteration = one transition of the walk (ma_jump). Returns true while the walk goes on
// in this kernel (not deactivated, not handed to the slow path).
template <bool SPLIT = false>
AHD bool ma_iter(const Env &env, Pkt &p, int64_t pi, MACtx &k) {
  ma_jump<SPLIT>(env, p, pi, k);
  const bool go = ma_pending(p) && p.pend == PEND_NONE;
  if (!ma_pending(p)) chi_after_ma(p);
  return go;
}
template <bool SPLIT = false>
AHD int advance_ma(const Env &env, Pkt &p, int64_t pi, int budget) {
  MACtx k = ma_ctx(env, p);
  int units = 0;
  bool go = ma_pending(p) && p.pend == PEND_NONE;
  while (go && units < budget) {
    go = ma_iter<SPLIT>(env, p, pi, k);
    units++;
  }
  return classify(env, p, env.S.ts_end);
}

// ---- fused thermal body, phase form: one iteration = a macro-atom phase of up to ARTIS_MA_PHASE transitions, then ONE
// k-packet step for every lane whose macro-atom has deactivated. The phases make the lanes of a wave run the same code
// at the same time; they only order the work of different packets.
#ifndef ARTIS_MA_PHASE
#define ARTIS_MA_PHASE 40  // measured optimum on MI355X (round 6, after the loop's instruction diet -- a round costs less, so idle lanes cost less against the per-phase work: 24 / 32 / 40 / 48 / 64 rounds: k_thermal 376 / 358-362 / 356 / 363 / 386 ms, profiles/r06/sweep.txt; round 4: 32 of 16 / 24 / 32 / 40: 488 / 446 / 435 / 438 ms; round 3: 24): short enough to keep the lanes busy, long enough to amortise the per-phase work
#endif
AHD bool kpkt_eligible(const Pkt &p, double ts_end);
AHD bool thermal_can_continue(const Pkt &p, double ts_end) {
  if (p.pend != PEND_NONE) return false;
  if (ma_pending(p)) return true;
  return pkt_active(p, ts_end) && p.type != ARTIS_TYPE_RPKT && !type_gamma(p.type);
}
// returns the units of work done (transitions + k-packet steps); *go = the packet can take another iteration.
// (artis_engine.hip k_thermal spells the two phases out so that the wave reconverges between them.)
AHD int thermal_iter(const Env &env, Pkt &p, int64_t pi, MACtx &k, bool *go) {
  const double ts_end = env.S.ts_end;
  int j = 0;
  while (j < ARTIS_MA_PHASE && ma_pending(p) && p.pend == PEND_NONE) {
    ma_jump(env, p, pi, k);
    j++;
  }
  if (j > 0) chi_after_ma(p);
  if (kpkt_eligible(p, ts_end) && !kpkt_blackbody_case(env, p.type, p.cellindex)) {
    do_kpkt(env, p, pi);
    p.chi_mgi = -1;
    j++;
  }
  *go = thermal_can_continue(p, ts_end) && !(kpkt_eligible(p, ts_end) && kpkt_blackbody_case(env, p.type, p.cellindex));
  return j;
}

// ---- k-packet kernel body: ONE do_kpkt()/do_kpkt_blackbody() call (update_packets.cc:291-305); it ends in an emission,
// a macro-atom activation, a deferred free-bound emission, or at the end of the timestep.
AHD bool kpkt_eligible(const Pkt &p, double ts_end) {
  return p.pend == PEND_NONE && !ma_pending(p) && pkt_active(p, ts_end) && p.type != ARTIS_TYPE_RPKT && !type_gamma(p.type);
}
template <bool SPLIT = false>
AHD int advance_kpkt(const Env &env, Pkt &p, int64_t pi) {
  if (kpkt_eligible(p, env.S.ts_end) && !kpkt_blackbody_case(env, p.type, p.cellindex)) {
    do_kpkt<SPLIT>(env, p, pi);
    p.chi_mgi = -1;
  }
  return classify(env, p, env.S.ts_end);
}
// ---- blackbody kernel body: ONE do_kpkt_blackbody() call; the packet leaves as an r-packet
AHD int advance_blackbody(const Env &env, Pkt &p, int64_t pi) {
  if (kpkt_eligible(p, env.S.ts_end) && kpkt_blackbody_case(env, p.type, p.cellindex)) {
    do_kpkt_blackbody(env, p, pi);
    p.chi_mgi = -1;
  }
  return classify(env, p, env.S.ts_end);
}

// ---- gamma kernel body: one iteration = one do_packet() call (update_packets.cc:257) for a type that does not use the
// cell cache: pellet, gamma packet, non-thermal pre-deposit and deposit types. Returns true while the packet stays with this kernel.
AHD bool gamma_can_continue(const Pkt &p, double ts_end) {
  return type_gamma(p.type) && p.prop_time < ts_end && !ma_pending(p);  // (a deposit may have activated a macro-atom)
}
AHD bool gamma_iter(const Env &env, Pkt &p, int64_t pi) {
  if (p.type == ARTIS_TYPE_GAMMA) {
    do_gamma(env, p, pi);
  } else if (p.type == ARTIS_TYPE_RADIOACTIVE_PELLET) {
    update_pellet(env, p, pi);
  } else if (p.type == ARTIS_TYPE_NTLEPTON_DEPOSITED) {
    do_ntlepton_deposit(env, p);
  } else if (p.type == ARTIS_TYPE_NTALPHA_FISPROD_DEPOSITED) {
    do_ntalpha_fisprod_deposit(env, p);
  } else {
    do_nonthermal_predeposit(env, p, pi);
  }
  p.chi_mgi = -1;
  return gamma_can_continue(p, env.S.ts_end);
}
AHD int advance_gamma(const Env &env, Pkt &p, int64_t pi, int budget) {
  int steps = 0;
  bool go = gamma_can_continue(p, env.S.ts_end);
  while (go && steps < budget) {
    go = gamma_iter(env, p, pi);
    steps++;
  }
  return classify(env, p, env.S.ts_end);
}

// slow-path kernel body: the one deferred action of the packet
// PEND_RPKT_ABSORB in the slow-path kernel: the free-free / bound-free branch of rpkt_event_continuum() with the draw the r-packet kernel
// made, then what do_rpkt_step() and rpkt_iter() do after the event
AHD void rpkt_slow_absorption(const Env &env, Pkt &p, int64_t pi) {
  Chi x;
  chi_load(env.P, pi, p, x);
  const int32_t u = p.pend_arg;
  p.pend = PEND_NONE;
  p.pend_arg = 0;
  rpkt_event_continuum<false>(env, p, pi, x, pi, u);
  if (!ma_pending(p) && p.type != ARTIS_TYPE_RPKT) x.nonemptymgi = -1;
  chi_store(env.P, pi, p, x);
}
// the slow-path actions that end in select_continuum_nu(): a free-bound emission of a k-packet, a radiative recombination of a macro-atom
AHD bool slow_selects_continuum_nu(const Pkt &p) {
  return p.pend == PEND_KPKT_FB || (p.pend == PEND_MA_ACTION && p.pend_arg == ARTIS_MA_ACTION_RADRECOMB);
}
AHD int advance_slow(const Env &env, Pkt &p, int64_t pi, FbSel *sel = nullptr) {
  // (the actions of the active macro-atom need its level's record: a cold level's may have to be filled again first -- or be waited for)
  if ((p.pend == PEND_MA_ACTION || p.pend == PEND_MA_SEARCH || p.pend == PEND_MA_RADSEARCH) && !ma_slow_record_ready(env, p))
    return classify(env, p, env.S.ts_end);
  if (p.pend == PEND_MA_ACTION) {
    ma_slow_action(env, p, pi, sel);
    chi_after_ma(p);
  } else if (p.pend == PEND_KPKT_FB) {
    kpkt_fb_emission(env, p, pi, sel);
    p.chi_mgi = -1;
  } else if (p.pend == PEND_MA_SEARCH || p.pend == PEND_MA_RADSEARCH) {
    ma_slow_search(env, p, pi);
    chi_after_ma(p);
  } else if (p.pend == PEND_KPKT_COLLEXC) {
    kpkt_slow_collexc(env, p);
    p.chi_mgi = -1;
  } else if (p.pend == PEND_MA_FILL) {
    ma_slow_fill(env, p);
  } else if (p.pend == PEND_RPKT_ABSORB) {
    rpkt_slow_absorption(env, p, pi);
  }
  return classify(env, p, env.S.ts_end);
}

// caller's array (reference struct Packet) <-> resident records, one packet
AHD void aos_to_rec(const artis_packet &a, const PktStore &P, int64_t i) {
  PktHot h;
  for (int k = 0; k < 4; k++) h.rng[k] = a.rngstate[k];
  h.prop_time = a.prop_time;
  h.pos_x = a.pos[0]; h.pos_y = a.pos[1]; h.pos_z = a.pos[2];
  h.e_cmf = a.e_cmf; h.nu_cmf = a.nu_cmf;
  h.type = a.type; h.cellindex = a.cellindex; h.next_trans = a.next_trans; h.nscatterings = a.nscatterings;
  h.ma_element = -1; h.ma_ion = -1; h.ma_level = -1; h.ma_line = -99;
  h.ma_origin = 0; h.pend = PEND_NONE; h.pend_arg = 0; h.chi_mgi = -1;
  h.emissiontype = a.emissiontype; h.trueemissiontype = a.trueemissiontype; h.absorptiontype = a.absorptiontype;
  h.flags = 0;
  P.hot[i] = h;
  PktFlight f;
  f.dir_x = a.dir[0]; f.dir_y = a.dir[1]; f.dir_z = a.dir[2];
  f.nu_rf = a.nu_rf; f.e_rf = a.e_rf; f.stokes_q = a.stokes_q; f.stokes_u = a.stokes_u;
  f.absorptionfreq = a.absorptionfreq;
  f.chi_nu = -1.; f.chi_es = 0.; f.chi_ff = 0.; f.chi_bf = 0.;
  f.em_pos_x = a.em_pos[0]; f.em_pos_y = a.em_pos[1]; f.em_pos_z = a.em_pos[2];
  f.em_time = a.em_time; f.pad0 = 0.f;
  P.flight[i] = f;
  PktCold c;
  c.trueem_pos_x = a.trueem_pos[0]; c.trueem_pos_y = a.trueem_pos[1]; c.trueem_pos_z = a.trueem_pos[2];
  c.tdecay = a.tdecay;
  c.escape_time = a.escape_time; c.trueem_time = a.trueem_time;
  c.escape_type = a.escape_type; c.pellet_decaytype = a.pellet_decaytype;
  c.originated_particle = a.originated_from_particlenotgamma ? 1 : 0;
  c.pad[0] = c.pad[1] = c.pad[2] = 0;
  P.cold[i] = c;
}
AHD void rec_to_aos(const PktStore &P, int64_t i, artis_packet &a) {
  const PktHot h = P.hot[i];
  const PktFlight f = P.flight[i];
  const PktCold c = P.cold[i];
  for (int k = 0; k < 4; k++) a.rngstate[k] = h.rng[k];
  a.prop_time = h.prop_time;
  a.pos[0] = h.pos_x; a.pos[1] = h.pos_y; a.pos[2] = h.pos_z;
  a.dir[0] = f.dir_x; a.dir[1] = f.dir_y; a.dir[2] = f.dir_z;
  a.nu_cmf = h.nu_cmf; a.e_cmf = h.e_cmf; a.nu_rf = f.nu_rf; a.e_rf = f.e_rf;
  a.stokes_q = f.stokes_q; a.stokes_u = f.stokes_u;
  a.next_trans = h.next_trans; a.nscatterings = h.nscatterings; a.type = h.type; a.cellindex = h.cellindex;
  a.emissiontype = h.emissiontype; a.absorptiontype = h.absorptiontype; a.trueemissiontype = h.trueemissiontype;
  a.escape_type = c.escape_type;
  a.em_pos[0] = f.em_pos_x; a.em_pos[1] = f.em_pos_y; a.em_pos[2] = f.em_pos_z;
  const bool nan_trueem = (h.flags & PKT_FLAG_TRUEEM_NAN) != 0;  // kpkt.cc:483, recorded as a flag
  a.trueem_pos[0] = nan_trueem ? NAN : c.trueem_pos_x;
  a.trueem_pos[1] = nan_trueem ? NAN : c.trueem_pos_y;
  a.trueem_pos[2] = nan_trueem ? NAN : c.trueem_pos_z;
  a.absorptionfreq = f.absorptionfreq;
  a.em_time = f.em_time; a.trueem_time = c.trueem_time; a.escape_time = c.escape_time;
}

}  // namespace artis
