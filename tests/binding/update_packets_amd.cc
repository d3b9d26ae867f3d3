// The reference-side binding of the MI355X packet engine, COMPILED: what a maintainer adds next to update_packets.cc.
//
// This translation unit includes the REFERENCE's own packet.h, constants.h and stats.h where they lie (-I/root/reference,
// -DGPU_ON: the per-packet generator state is a member of struct Packet, packet.h:118), checks at compile time that the
// reference's struct Packet is the C-ABI's artis_packet member by member, owns the packets as std::span<Packet> like
// update_packets() does (update_packets.cc:530), and hands them to artis_amd_update_packets() through the C-ABI alone
// (include/artis_amd.h; no Python, no torch). Where the reference would fill artis_model / artis_cellstate from its
// globals:: arrays (INTEGRATION.md section 2 names each one), this test harness fills them from a dump of the same flat
// arrays written by tests/binding/dump.py, so that the program runs without the rest of the reference (which does not
// build in this image: it needs <print>/<format>/<mdspan>).
//
// Built by `make -C oracle ref` into oracle/_ref/update_packets_amd (the binary travels to the GPU box; the reference's
// headers do not). tests/test_gpu_parity.py::test_compiled_reference_side_binding runs it and compares its packets with
// the ctypes path byte for byte.
//
//   update_packets_amd <libartis_amd.so> <dump file> <output file>
#include <dlfcn.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <span>
#include <string>
#include <vector>

#include "constants.h"  // the reference's
#include "packet.h"
#include "stats.h"

#include "artis_amd.h"  // the C-ABI (include/)

// ---- the reference's struct Packet IS the C-ABI's artis_packet (GPU_ON layout)
static_assert(sizeof(Packet) == sizeof(artis_packet), "struct Packet and artis_packet differ in size");
#define SAME_MEMBER(f) static_assert(offsetof(Packet, f) == offsetof(artis_packet, f) && sizeof(Packet::f) == sizeof(artis_packet::f), #f)
SAME_MEMBER(rngstate); SAME_MEMBER(prop_time); SAME_MEMBER(pos); SAME_MEMBER(dir); SAME_MEMBER(nu_cmf); SAME_MEMBER(e_cmf);
SAME_MEMBER(nu_rf); SAME_MEMBER(e_rf); SAME_MEMBER(next_trans); SAME_MEMBER(nscatterings); SAME_MEMBER(emissiontype);
SAME_MEMBER(em_pos); SAME_MEMBER(em_time); SAME_MEMBER(absorptiontype); SAME_MEMBER(absorptionfreq); SAME_MEMBER(stokes_q);
SAME_MEMBER(stokes_u); SAME_MEMBER(trueemissiontype); SAME_MEMBER(trueem_pos); SAME_MEMBER(trueem_time); SAME_MEMBER(type);
SAME_MEMBER(cellindex); SAME_MEMBER(escape_type); SAME_MEMBER(escape_time); SAME_MEMBER(tdecay); SAME_MEMBER(number);
SAME_MEMBER(originated_from_particlenotgamma); SAME_MEMBER(pellet_decaytype); SAME_MEMBER(pellet_nucindex);
// ... the packet types and event counters the engine numbers like the reference
#define SAME_VALUE(a, b) static_assert(static_cast<int>(a) == static_cast<int>(b), #a)
SAME_VALUE(TYPE_ESCAPE, ARTIS_TYPE_ESCAPE); SAME_VALUE(TYPE_RPKT, ARTIS_TYPE_RPKT); SAME_VALUE(TYPE_KPKT, ARTIS_TYPE_KPKT);
SAME_VALUE(TYPE_MA, ARTIS_TYPE_MA); SAME_VALUE(TYPE_GAMMA, ARTIS_TYPE_GAMMA); SAME_VALUE(TYPE_RADIOACTIVE_PELLET, ARTIS_TYPE_RADIOACTIVE_PELLET);
SAME_VALUE(TYPE_PRE_KPKT, ARTIS_TYPE_PRE_KPKT);
SAME_VALUE(stats::Counter::COUNT, ARTIS_STAT_COUNT); SAME_VALUE(stats::Counter::ELECTRON_SCATTERINGS, ARTIS_STAT_ELECTRON_SCATTERINGS);
SAME_VALUE(stats::Counter::INTERACTIONS, ARTIS_STAT_INTERACTIONS); SAME_VALUE(stats::Counter::MA_STAT_DEACTIVATION_BB, ARTIS_STAT_MA_DEACTIVATION_BB);
static_assert(static_cast<int>(stats::Counter::COUNT) <= ARTIS_NSTATS);
static_assert(CLIGHT == 2.99792458e+10 && H == 6.6260755e-27);

// ---- the engine, bound at run time through its C-ABI only
struct EngineApi {
  decltype(&artis_amd_engine_create) create;
  decltype(&artis_amd_engine_destroy) destroy;
  decltype(&artis_amd_set_cellstate) set_cellstate;
  decltype(&artis_amd_update_packets) update_packets;
  decltype(&artis_amd_last_error) last_error;
  decltype(&artis_amd_abi_version) abi_version;
};
static EngineApi bind(const char *so) {
  void *h = dlopen(so, RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    std::fprintf(stderr, "dlopen %s: %s\n", so, dlerror());
    std::exit(2);
  }
  EngineApi a{};
#define SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, #name))
  SYM(create, artis_amd_engine_create); SYM(destroy, artis_amd_engine_destroy); SYM(set_cellstate, artis_amd_set_cellstate);
  SYM(update_packets, artis_amd_update_packets); SYM(last_error, artis_amd_last_error); SYM(abi_version, artis_amd_abi_version);
#undef SYM
  if (!a.create || !a.destroy || !a.set_cellstate || !a.update_packets || !a.last_error || !a.abi_version) std::exit(3);
  return a;
}

// what replaces the body of update_packets() (update_packets.cc:530): one call for the whole pass loop
static int update_packets_amd(const EngineApi &api, artis_amd_engine *eng, std::span<Packet> packets, artis_estimators &est) {
  return api.update_packets(eng, reinterpret_cast<artis_packet *>(packets.data()), static_cast<int64_t>(packets.size()), &est);
}

// ---- the flat arrays the reference holds in globals:: / grid::, here read from a dump (tests/binding/dump.py)
struct Dump {
  struct Entry { int64_t count; std::vector<unsigned char> bytes; };
  std::map<std::string, Entry> e;
  explicit Dump(const char *path) {
    FILE *f = std::fopen(path, "rb");
    if (!f) std::exit(4);
    char name[64];
    while (std::fread(name, 1, 64, f) == 64) {
      int64_t count = 0, nbytes = 0;
      if (std::fread(&count, 8, 1, f) != 1 || std::fread(&nbytes, 8, 1, f) != 1) std::exit(5);
      Entry en{count, std::vector<unsigned char>(static_cast<size_t>(nbytes) + 8)};
      if (nbytes > 0 && std::fread(en.bytes.data(), 1, static_cast<size_t>(nbytes), f) != static_cast<size_t>(nbytes)) std::exit(6);
      e.emplace(std::string(name), std::move(en));
    }
    std::fclose(f);
  }
  template <typename T> const T *array(const char *name) const {
    const auto it = e.find(name);
    return it == e.end() ? nullptr : reinterpret_cast<const T *>(it->second.bytes.data());  // absent: an optional field, NULL
  }
  template <typename T> T scalar(const char *name) const {
    const T *p = array<T>(name);
    return p ? *p : T{};
  }
  int64_t count(const char *name) const {
    const auto it = e.find(name);
    return it == e.end() ? 0 : it->second.count;
  }
};

int main(int argc, char **argv) {
  if (argc != 4) {
    std::fprintf(stderr, "usage: %s <libartis_amd.so> <dump> <output>\n", argv[0]);
    return 1;
  }
  const EngineApi api = bind(argv[1]);
  const Dump d(argv[2]);
  artis_model m{};
  m.nelements = d.scalar<int32_t>("nelements");
  m.nions = d.scalar<int32_t>("nions");
  m.nlevels = d.scalar<int32_t>("nlevels");
  m.nlines = d.scalar<int32_t>("nlines");
  m.nalltrans = d.scalar<int32_t>("nalltrans");
  m.nphixstargets_total = d.scalar<int32_t>("nphixstargets_total");
  m.nphixslevels = d.scalar<int32_t>("nphixslevels");
  m.nbfcontinua = d.scalar<int32_t>("nbfcontinua");
  m.nbfcontinua_ground = d.scalar<int32_t>("nbfcontinua_ground");
  m.ncoolingterms = d.scalar<int32_t>("ncoolingterms");
  m.nmatransblock = d.scalar<int32_t>("nmatransblock");
  m.NPHIXSPOINTS = d.scalar<int32_t>("NPHIXSPOINTS");
  m.NPHIXSNUINCREMENT = d.scalar<double>("NPHIXSNUINCREMENT");
  m.elem_nions = d.array<int32_t>("elem_nions");
  m.elem_uniqueionindexstart = d.array<int32_t>("elem_uniqueionindexstart");
  m.elem_anumber = d.array<int32_t>("elem_anumber");
  m.elem_lowest_ionstage = d.array<int32_t>("elem_lowest_ionstage");
  m.ion_element = d.array<int32_t>("ion_element");
  m.ion_nlevels = d.array<int32_t>("ion_nlevels");
  m.ion_nlevels_ionising = d.array<int32_t>("ion_nlevels_ionising");
  m.ion_maxrecombininglevel = d.array<int32_t>("ion_maxrecombininglevel");
  m.ion_uniquelevelindexstart = d.array<int32_t>("ion_uniquelevelindexstart");
  m.ion_coolingoffset = d.array<int32_t>("ion_coolingoffset");
  m.ion_ncoolingterms = d.array<int32_t>("ion_ncoolingterms");
  m.level_epsilon = d.array<double>("level_epsilon");
  m.level_statweight = d.array<float>("level_statweight");
  m.level_alltrans_startdown = d.array<int32_t>("level_alltrans_startdown");
  m.level_ndowntrans = d.array<int32_t>("level_ndowntrans");
  m.level_nuptrans = d.array<int32_t>("level_nuptrans");
  m.level_closestgroundlevelcont = d.array<int32_t>("level_closestgroundlevelcont");
  m.level_phixsstart = d.array<int32_t>("level_phixsstart");
  m.level_nphixstargets = d.array<int32_t>("level_nphixstargets");
  m.level_phixstargetstart = d.array<int32_t>("level_phixstargetstart");
  m.level_bflist_start = d.array<int32_t>("level_bflist_start");
  m.level_matransblock_start = d.array<int32_t>("level_matransblock_start");
  m.alltrans_lineindex = d.array<int32_t>("alltrans_lineindex");
  m.alltrans_targetlevelindex = d.array<int32_t>("alltrans_targetlevelindex");
  m.alltrans_einstein_A = d.array<float>("alltrans_einstein_A");
  m.alltrans_coll_str = d.array<float>("alltrans_coll_str");
  m.alltrans_osc_strength = d.array<float>("alltrans_osc_strength");
  m.alltrans_forbidden = d.array<uint8_t>("alltrans_forbidden");
  m.line_nu = d.array<double>("line_nu");
  m.line_elementindex = d.array<int32_t>("line_elementindex");
  m.line_ionindex = d.array<int32_t>("line_ionindex");
  m.line_uniquelevelindex_lower = d.array<int32_t>("line_uniquelevelindex_lower");
  m.line_uniquelevelindex_upper = d.array<int32_t>("line_uniquelevelindex_upper");
  m.line_B_ul = d.array<float>("line_B_ul");
  m.line_B_lu = d.array<float>("line_B_lu");
  m.allphixs = d.array<float>("allphixs");
  m.allphixstargets_levelindex = d.array<int32_t>("allphixstargets_levelindex");
  m.allphixstargets_probability = d.array<double>("allphixstargets_probability");
  m.allcont_nu_edge = d.array<double>("allcont_nu_edge");
  m.allcont_element = d.array<int32_t>("allcont_element");
  m.allcont_ion = d.array<int32_t>("allcont_ion");
  m.allcont_level = d.array<int32_t>("allcont_level");
  m.allcont_phixstargetindex = d.array<int32_t>("allcont_phixstargetindex");
  m.allcont_upperlevel = d.array<int32_t>("allcont_upperlevel");
  m.allcont_uniquelevelindex = d.array<int32_t>("allcont_uniquelevelindex");
  m.allcont_probability = d.array<double>("allcont_probability");
  m.allcont_groundcontestimindex = d.array<int32_t>("allcont_groundcontestimindex");
  m.groundcont_nu_edge = d.array<double>("groundcont_nu_edge");
  m.spontrecombcoeffs = d.array<double>("spontrecombcoeffs");
  m.corrphotoioncoeffs = d.array<double>("corrphotoioncoeffs");
  m.bfcooling_coeffs = d.array<double>("bfcooling_coeffs");
  m.coolinglist_type = d.array<uint8_t>("coolinglist_type");
  m.coolinglist_level = d.array<int32_t>("coolinglist_level");
  m.coolinglist_phixstargetindex = d.array<int32_t>("coolinglist_phixstargetindex");
  m.gridtype = d.scalar<int32_t>("gridtype");
  for (int a = 0; a < 3; a++) m.ncoordgrid[a] = d.array<int32_t>("ncoordgrid")[a];
  m.ngrid = d.scalar<int32_t>("ngrid");
  m.npts_nonempty = d.scalar<int32_t>("npts_nonempty");
  m.tmin = d.scalar<double>("tmin");
  m.vmax = d.scalar<double>("vmax");
  m.rmax = d.scalar<double>("rmax");
  m.coord_pos_min_tmin[0] = d.array<double>("coord_pos_min_tmin0");
  m.coord_pos_min_tmin[1] = d.array<double>("coord_pos_min_tmin1");
  m.coord_pos_min_tmin[2] = d.array<double>("coord_pos_min_tmin2");
  m.propcell_nonemptymgi = d.array<int32_t>("propcell_nonemptymgi");
  m.elem_meannucmass = d.array<float>("elem_meannucmass");
  m.ion_nt_sum_q_over_binding = d.array<double>("ion_nt_sum_q_over_binding");
  m.ejecta_kinetic_energy = d.scalar<double>("ejecta_kinetic_energy");
  m.mtot_input = d.scalar<double>("mtot_input");
  m.allcont_bfestimindex = d.array<int32_t>("allcont_bfestimindex");
  m.nbfestim = d.scalar<int32_t>("nbfestim");
  m.rho_tmin = d.array<float>("rho_tmin");
  m.xcom_elem_start = d.array<int32_t>("xcom_elem_start");
  m.xcom_energy = d.array<double>("xcom_energy");
  m.xcom_sigma = d.array<double>("xcom_sigma");
  m.detailed_lineindices = d.array<int32_t>("detailed_lineindices");
  m.detailed_linecount = d.scalar<int32_t>("detailed_linecount");
  artis_cellstate cs{};
  cs.rho = d.array<float>("cell.rho");
  cs.Te = d.array<float>("cell.Te");
  cs.TJ = d.array<float>("cell.TJ");
  cs.TR = d.array<float>("cell.TR");
  cs.W = d.array<float>("cell.W");
  cs.nne = d.array<float>("cell.nne");
  cs.nnetot = d.array<float>("cell.nnetot");
  cs.kappagrey = d.array<float>("cell.kappagrey");
  cs.thick = d.array<int32_t>("cell.thick");
  cs.clumpfactor = d.array<float>("cell.clumpfactor");
  cs.ion_groundlevelpops = d.array<float>("cell.ion_groundlevelpops");
  cs.ion_partfuncts = d.array<float>("cell.ion_partfuncts");
  cs.elem_massfracs = d.array<float>("cell.elem_massfracs");
  cs.corrphotoionrenorm = d.array<double>("cell.corrphotoionrenorm");
  cs.ffegrp = d.array<float>("cell.ffegrp");
  cs.levelpops = d.array<double>("cell.levelpops");
  cs.corrphotoioncoeff = d.array<double>("cell.corrphotoioncoeff");
  cs.radfieldbin_W = d.array<float>("cell.radfieldbin_W");
  cs.radfieldbin_T_R = d.array<float>("cell.radfieldbin_T_R");
  cs.nt_frac_ionisation = d.array<float>("cell.nt_frac_ionisation");
  cs.nt_frac_excitation = d.array<float>("cell.nt_frac_excitation");
  cs.nt_deposition_rate_density = d.array<double>("cell.nt_deposition_rate_density");
  cs.nt_eff_ionpot = d.array<float>("cell.nt_eff_ionpot");
  cs.nt_prob_num_auger = d.array<float>("cell.nt_prob_num_auger");
  cs.nt_ionenfrac_num_auger = d.array<float>("cell.nt_ionenfrac_num_auger");
  cs.nt_exc_count = d.array<int32_t>("cell.nt_exc_count");
  cs.nt_exc_frac_deposition = d.array<double>("cell.nt_exc_frac_deposition");
  cs.nt_exc_ratecoeffperdeposition = d.array<double>("cell.nt_exc_ratecoeffperdeposition");
  cs.nt_exc_alltransindex = d.array<int32_t>("cell.nt_exc_alltransindex");
  cs.nt_excitations_stored = d.scalar<int32_t>("cell.nt_excitations_stored");
  cs.expansionopacities = d.array<float>("cell.expansionopacities");
  cs.expansionopacity_planck_cumulative = d.array<double>("cell.expansionopacity_planck_cumulative");
  cs.Jb_lu_normed = d.array<double>("cell.Jb_lu_normed");
  cs.elem_meanweight = d.array<float>("cell.elem_meanweight");
  const artis_timestep ts{d.scalar<int32_t>("ts.nts"), d.scalar<double>("ts.start"), d.scalar<double>("ts.width"),
                          d.scalar<double>("ts.mid"), d.scalar<double>("ts.max_path_step")};
  // the packets: owned here as the reference owns them
  const int64_t npackets = d.count("packets") ;
  std::vector<Packet> pkts(static_cast<size_t>(npackets));
  std::memcpy(static_cast<void *>(pkts.data()), d.array<unsigned char>("packets"), sizeof(Packet) * pkts.size());

  // the estimator arrays of the reference (radfield.cc J / nuJ, globals.h:128-134, stats.cc), zero at the start of a timestep
  const size_t n = static_cast<size_t>(m.npts_nonempty), g = static_cast<size_t>(m.nbfcontinua_ground > 0 ? m.nbfcontinua_ground : 1);
  std::vector<double> J(n), nuJ(n), ffheat(n), colheat(n), gammaest(n * g), bfheat(n * g), depgamma(n), scalars(ARTIS_NSCALARS), depe(n),
      depp(n), depa(n);
  std::vector<int64_t> counters(ARTIS_NSTATS);
  artis_estimators est{};
  est.J = J.data(); est.nuJ = nuJ.data(); est.ffheatingestimator = ffheat.data(); est.colheatingestimator = colheat.data();
  est.gammaestimator = gammaest.data(); est.bfheatingestimator = bfheat.data(); est.stats = counters.data();
  est.dep_estimator_gamma = depgamma.data(); est.scalars = scalars.data(); est.dep_estimator_electron = depe.data();
  est.dep_estimator_positron = depp.data(); est.dep_estimator_alpha = depa.data();

  artis_amd_engine *eng = nullptr;
  if (api.create(&m, /*device*/ 0, &eng) != ARTIS_OK || api.set_cellstate(eng, &cs, &ts) != ARTIS_OK ||
      update_packets_amd(api, eng, std::span<Packet>(pkts), est) != ARTIS_OK) {
    std::fprintf(stderr, "engine: %s\n", api.last_error());
    return 7;
  }
  api.destroy(eng);

  FILE *out = std::fopen(argv[3], "wb");
  if (!out) return 8;
  std::fwrite(pkts.data(), sizeof(Packet), pkts.size(), out);
  for (const std::vector<double> *v : {&J, &nuJ, &ffheat, &colheat, &gammaest, &bfheat, &depgamma, &scalars, &depe, &depp, &depa})
    std::fwrite(v->data(), sizeof(double), v->size(), out);
  std::fwrite(counters.data(), sizeof(int64_t), counters.size(), out);
  std::fclose(out);
  std::printf("update_packets_amd: ABI %d, %lld packets, %lld packet-steps\n", api.abi_version(), static_cast<long long>(npackets),
              static_cast<long long>(counters[ARTIS_STAT_X_RPKT_STEPS] + counters[ARTIS_STAT_X_KPKT_STEPS]));
  return 0;
}
