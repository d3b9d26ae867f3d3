// Known-answer generator: compiles the REFERENCE's gausskronrod.h (included from /root/reference, never copied) and
// prints gauss_kronrod_integrate<31>(f, a, b, 15, tol, &error) for three analytic integrands, so that the oracle's
// restatement of the integrator (tables, summation order, adaptive bisection) can be pinned bit for bit.
// Usage: ref_gk31 <mode 1|2|3> <p0> <p1> <a> <b> <tol>   -> "<result hexfloat> <error hexfloat>"
// The integrands are the same three formulas as gk31_test_integrand() in oracle/artis_oracle.c.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "gausskronrod.h"

int main(int argc, char** argv) {
  if (argc < 7) return 2;
  const int mode = std::atoi(argv[1]);
  const double p0 = std::strtod(argv[2], nullptr);
  const double p1 = std::strtod(argv[3], nullptr);
  const double a = std::strtod(argv[4], nullptr);
  const double b = std::strtod(argv[5], nullptr);
  const double tol = std::strtod(argv[6], nullptr);
  double error = 0.;
  double result = 0.;
  if (mode == 1) {
    result = gauss_kronrod_integrate<31>([=](double x) { return std::exp(-p0 * x) * (1. + std::floor(x * p1)); }, a, b, 15, tol, &error);
  } else if (mode == 2) {
    result = gauss_kronrod_integrate<31>([=](double x) { return x * x * std::exp(-p0 * x) / (1. + (p1 * x * x * x)); }, a, b, 15, tol, &error);
  } else {
    result = gauss_kronrod_integrate<31>([=](double x) { return std::sqrt(std::fabs(x - p0)) + p1; }, a, b, 15, tol, &error);
  }
  std::printf("%a %a\n", result, error);
  return 0;
}
