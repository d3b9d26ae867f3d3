// Known-answer generator: compiles the REFERENCE's random.h (included from
// /root/reference, never copied) and prints streams of its generators, so the
// oracle's restatement of the RNG can be pinned bit for bit.
// Usage: ref_random <seed> <count>   -> one line per draw: "<u32 hex> <float hex bits>"
//        for rngstate_type{seed} (Xoshiro128PP seeded through SplitMix32) and
//        rng_uniform() in its GPU_ON form.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "random.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const auto seed = static_cast<std::uint32_t>(std::strtoul(argv[1], nullptr, 10));
  const int count = std::atoi(argv[2]);
  rngstate_type a{seed};
  rngstate_type b{seed};
  for (int i = 0; i < count; i++) {
    const std::uint32_t raw = a();
    const float z = rng_uniform(b);
    std::uint32_t zbits = 0;
    std::memcpy(&zbits, &z, 4);
    std::printf("%08x %08x\n", raw, zbits);
  }
  return 0;
}
