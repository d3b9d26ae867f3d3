// Known-answer generator: compiles the REFERENCE's macroatom.h (included from /root/reference, never copied) and prints
// rad_deexcitation_ratecoeff() (macroatom.h:61, the Sobolev-escape radiative de-excitation rate used when the
// macro-atom transition rates of a cell are calculated) as a hex float.
// Usage: ref_macroatom <epsilon_trans> <A_ul> <g_upper> <g_lower> <nn_upper> <nn_lower> <t_current>
#include <cstdio>
#include <cstdlib>

#include "macroatom.h"

int main(int argc, char** argv) {
  if (argc < 8) return 2;
  const double eps = std::strtod(argv[1], nullptr);
  const float A_ul = std::strtof(argv[2], nullptr);
  const double gu = std::strtod(argv[3], nullptr);
  const double gl = std::strtod(argv[4], nullptr);
  const double nnu = std::strtod(argv[5], nullptr);
  const double nnl = std::strtod(argv[6], nullptr);
  const double t = std::strtod(argv[7], nullptr);
  std::printf("%a\n", rad_deexcitation_ratecoeff(eps, A_ul, gu, gl, nnu, nnl, t));
  return 0;
}
