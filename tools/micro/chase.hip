// Latency of dependent random reads on MI355X as a function of the footprint and of how the reads of one wave are spread:
//   mode 0: every lane chases its own chain anywhere in the buffer (one 8-byte read per hop)
//   mode 1: the wave's lanes stay within one 1 MB window (a "cell row") that moves every 64 hops
// build: hipcc --offload-arch=gfx950 -O3 -o chase tools/micro/chase.hip ; run: ./chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void k_fill(uint64_t *buf, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) buf[i] = mix(i + 12345);
}
__global__ void k_chase(const uint64_t *buf, uint64_t n, int hops, int mode, uint64_t window, long long *clocks, uint64_t *sink) {
  uint64_t x = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 7919 + 1);
  const uint64_t waveid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long t0 = clock64();
  for (int h = 0; h < hops; h++) {
    uint64_t idx;
    if (mode == 0) idx = x & (n - 1);  // (footprints and windows are powers of two: no division in the chain)
    else {
      const uint64_t wbase = (mix(waveid * 1315423911ULL + (uint64_t)(h >> 6)) & ((n / window) - 1)) * window;
      idx = wbase + (x & (window - 1));
    }
    x = mix(x ^ buf[idx]);
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)clocks, (unsigned long long)(t1 - t0));
  if (x == 42) *sink = x;
}
int main() {
  long long *d_clk; uint64_t *d_sink;
  CK(hipMalloc(&d_clk, 8)); CK(hipMalloc(&d_sink, 8));
  const double gbs[] = {0.0625, 1, 8, 32, 128};
  for (double gb : gbs) {
    const uint64_t n = (uint64_t)(gb * (1ull << 30)) / 8;
    uint64_t *buf;
    if (hipMalloc(&buf, n * 8) != hipSuccess) { printf("%.3f GB: allocation failed\n", gb); continue; }
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, n);
    CK(hipDeviceSynchronize());
    for (int mode = 0; mode < 2; mode++)
      for (int wpc : {1, 4, 16}) {  // waves per CU
        const int hops = 2048, nwaves = 256 * wpc;
        CK(hipMemset(d_clk, 0, 8));
        hipLaunchKernelGGL(k_chase, dim3(nwaves / 4), dim3(256), 0, 0, buf, n, hops, mode, (uint64_t)(1 << 17), d_clk, d_sink);
        CK(hipDeviceSynchronize());
        CK(hipMemset(d_clk, 0, 8));
        hipLaunchKernelGGL(k_chase, dim3(nwaves / 4), dim3(256), 0, 0, buf, n, hops, mode, (uint64_t)(1 << 17), d_clk, d_sink);
        CK(hipDeviceSynchronize());
        long long clk = 0;
        CK(hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost));
        printf("%7.3f GB  mode %d (%s)  %2d waves/CU: %7.0f clocks per hop (64 lanes, 64 different lines)\n", gb, mode, mode ? "wave in a 1 MB window" : "anywhere", wpc,
               (double)clk / nwaves / hops);
      }
    CK(hipFree(buf));
  }
  return 0;
}
