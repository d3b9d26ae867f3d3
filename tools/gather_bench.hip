// Microbenchmark: cost of gathering 64-byte records with 16-byte loads, (A) one record per lane, four loads per lane,
// against (B) one record per quad of lanes per instruction (lane l reads piece l%4 of the record of packet 16j + l/4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
struct D2 { double x, y; };
__device__ inline uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(const double *tab, uint32_t nrec_mask, int iters, double *out) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  const int lane = threadIdx.x & 63;
  double acc = 0;
  for (int it = 0; it < iters; it++) {
    const uint32_t mine = lcg(s) & nrec_mask;
    if (MODE == 0) {
      const D2 *r = (const D2 *)(tab + (size_t)mine * 8);
      const D2 a = r[0], b = r[1], c = r[2], d = r[3];
      acc += a.x + b.y + c.x + d.y;
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t other = __builtin_amdgcn_ds_bpermute(((16 * j) + (lane >> 2)) << 2, mine);
        const D2 v = *((const D2 *)(tab + (size_t)other * 8) + (lane & 3));
        acc += v.x + v.y;
      }
    } else if (MODE == 2) {  // 8-byte loads, own record
      const double *r = tab + (size_t)mine * 8;
      double t = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) t += r[j];
      acc += t;
    } else if (MODE == 3) {  // one 16-byte load per lane own record (quarter of the data)
      const D2 *r = (const D2 *)(tab + (size_t)mine * 8);
      const D2 a = r[0];
      acc += a.x + a.y;
    }
    s += (uint32_t)(acc != 1.234);  // data dependence between rounds like a walk
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  const int blocks = 256 * 4, iters = 2000;
  double *out; hipMalloc(&out, blocks * 256 * 8);
  for (size_t bytes : {size_t(8) << 10, size_t(1) << 20, size_t(32) << 20, size_t(1) << 30}) {
    double *tab; hipMalloc(&tab, bytes); hipMemset(tab, 0, bytes);
    const uint32_t mask = (uint32_t)(bytes / 64 - 1);
    for (int mode = 0; mode < 4; mode++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0) k<0><<<blocks, 256>>>(tab, mask, iters, out);
        if (mode == 1) k<1><<<blocks, 256>>>(tab, mask, iters, out);
        if (mode == 2) k<2><<<blocks, 256>>>(tab, mask, iters, out);
        if (mode == 3) k<3><<<blocks, 256>>>(tab, mask, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double recs = (double)blocks * 256 * iters;
      printf("table %8zu KB mode %d: %8.3f ms  %7.2f G records/s  %6.2f ns per wave-round per CU\n", bytes >> 10, mode, ms,
             recs / ms * 1e-6, ms * 1e6 / ((double)blocks * 4 * iters / 256));
    }
    hipFree(tab);
  }
  return 0;
}
