// Known-answer generator: compiles the REFERENCE's constants.h (included from /root/reference, never copied) and prints
// the physical constants the packet path uses as hex floats, so that the constants restated in oracle/artis_oracle.c
// and artis_amd/csrc/physics.h can be pinned bit for bit. Output: "<name> <hexfloat>" per line.
#include <cstdio>

#include "constants.h"

#define P(name) std::printf("%s %a\n", #name, static_cast<double>(name))

int main() {
  P(CLIGHT);
  P(CLIGHT_PROP);
  P(H);
  P(MH);
  P(ME);
  P(PI);
  P(EV);
  P(MEV);
  P(SIGMA_T);
  P(THOMSON_LIMIT);
  P(KB);
  P(SAHACONST);
  P(EULERGAMMA);
  P(CLIGHTSQUARED);
  P(CLIGHTSQUAREDOVERTWOH);
  P(HOVERKB);
  P(HCLIGHTOVERFOURPI);
  P(H_ionpot);
  P(C_0);
  return 0;
}
