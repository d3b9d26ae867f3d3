// Microbenchmark: is the unit of an L1 miss a 64-byte sector or a 128-byte line? Per round a lane picks a random
// 128-byte record of an L2-resident (or larger) table and reads
//   mode 0: 16 B at offset 0                      mode 1: 16 B at offsets 0 and 16 (same 64-B half)
//   mode 2: 16 B at offsets 0 and 64 (other half) mode 3: 16 B at offset 0 of two different records
//   mode 4: 4 x 16 B at offsets 0..48 (one half)  mode 5: 4 x 16 B at offsets 16..64 (straddles the halves)
// Rounds are dependent (like a walk). Build: hipcc -O3 --offload-arch=gfx950 tools/sector_bench.hip -o /tmp/sector_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct D2 { double x, y; };
__device__ inline uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(const double *tab, uint32_t mask, int iters, double *out) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  double acc = 0;
  for (int it = 0; it < iters; it++) {
    const D2 *r = (const D2 *)(tab + (size_t)(lcg(s) & mask) * 16);
    if (MODE == 0) { const D2 a = r[0]; acc += a.x + a.y; }
    if (MODE == 1) { const D2 a = r[0], b = r[1]; acc += a.x + b.y; }
    if (MODE == 2) { const D2 a = r[0], b = r[4]; acc += a.x + b.y; }
    if (MODE == 3) { const D2 *r2 = (const D2 *)(tab + (size_t)(lcg(s) & mask) * 16); const D2 a = r[0], b = r2[0]; acc += a.x + b.y; }
    if (MODE == 4) { const D2 a = r[0], b = r[1], c = r[2], d = r[3]; acc += a.x + b.y + c.x + d.y; }
    if (MODE == 5) { const D2 a = r[1], b = r[2], c = r[3], d = r[4]; acc += a.x + b.y + c.x + d.y; }
    s += (uint32_t)(acc != 1.234);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  const int blocks = 256 * 4, iters = 2000;
  double *out; hipMalloc(&out, blocks * 256 * 8);
  for (size_t bytes : {size_t(2) << 20, size_t(256) << 20}) {
    double *tab; hipMalloc(&tab, bytes); hipMemset(tab, 0, bytes);
    const uint32_t mask = (uint32_t)(bytes / 128 - 1);
    for (int mode = 0; mode < 6; mode++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0) k<0><<<blocks, 256>>>(tab, mask, iters, out);
        if (mode == 1) k<1><<<blocks, 256>>>(tab, mask, iters, out);
        if (mode == 2) k<2><<<blocks, 256>>>(tab, mask, iters, out);
        if (mode == 3) k<3><<<blocks, 256>>>(tab, mask, iters, out);
        if (mode == 4) k<4><<<blocks, 256>>>(tab, mask, iters, out);
        if (mode == 5) k<5><<<blocks, 256>>>(tab, mask, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("table %7zu KB mode %d: %8.3f ms  %7.1f clocks of CU time per wave-round (2.4 GHz)\n", bytes >> 10, mode, ms,
             ms * 1e6 / ((double)blocks * 4 * iters / 256) * 2.4);
    }
    hipFree(tab);
  }
  return 0;
}
