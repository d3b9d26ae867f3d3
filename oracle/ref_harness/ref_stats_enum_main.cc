// Known-answer generator: compiles the REFERENCE's stats.h (included from /root/reference, never copied) and prints the
// integer value of every stats::Counter enumerator, so that the event-counter array of the C-ABI
// (artis_estimators.stats, ARTIS_STAT_* in include/artis_amd.h) can be pinned to the reference's indexing.
// Also prints decay::DecayType-independent constants the packet path shares with the reference: none.
#include <cstdio>

#include "stats.h"

#define C(name) std::printf("%s %d\n", #name, static_cast<int>(stats::Counter::name))

int main() {
  C(MA_STAT_ACTIVATION_COLLEXC);
  C(MA_STAT_ACTIVATION_COLLION);
  C(MA_STAT_ACTIVATION_NTCOLLEXC);
  C(MA_STAT_ACTIVATION_NTCOLLION);
  C(MA_STAT_ACTIVATION_BB);
  C(MA_STAT_ACTIVATION_BF);
  C(MA_STAT_ACTIVATION_FB);
  C(MA_STAT_DEACTIVATION_COLLDEEXC);
  C(MA_STAT_DEACTIVATION_COLLRECOMB);
  C(MA_STAT_DEACTIVATION_BB);
  C(MA_STAT_DEACTIVATION_FB);
  C(MA_STAT_INTERNALUPHIGHER);
  C(MA_STAT_INTERNALUPHIGHERNT);
  C(MA_STAT_INTERNALDOWNLOWER);
  C(K_STAT_TO_MA_COLLEXC);
  C(K_STAT_TO_MA_COLLION);
  C(K_STAT_TO_R_FF);
  C(K_STAT_TO_R_FB);
  C(K_STAT_TO_R_BB);
  C(K_STAT_FROM_FF);
  C(K_STAT_FROM_BF);
  C(NT_STAT_FROM_GAMMA);
  C(NT_STAT_TO_IONISATION);
  C(NT_STAT_TO_EXCITATION);
  C(NT_STAT_TO_KPKT);
  C(K_STAT_FROM_EARLIERDECAY);
  C(INTERACTIONS);
  C(ELECTRON_SCATTERINGS);
  C(RESONANCESCATTERINGS);
  C(CELLCROSSINGS);
  C(UPSCATTER);
  C(DOWNSCATTER);
  C(UPDATECELL);
  C(PKTESCAPES);
  C(COUNT);
  return 0;
}
