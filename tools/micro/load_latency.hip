// Latency of a chain of dependent loads (cycles per load, one lane, one wave) over tables of several sizes: global memory through
// the vector L1 / L2 (random chain, 16-byte records as the hull graphs of hb_mpr.hpp) against LDS.  s_memtime ticks ~ shader cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o build/load_latency tools/micro/load_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase_global(const int4* tab, int steps, unsigned long long* ticks, int* sink) {
  int i = 0;
  // warm the caches with one pass
  for (int k = 0; k < steps; k++) i = tab[i].x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < steps; k++) i = tab[i].x;
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { ticks[0] = t1 - t0; sink[0] = i; }
}
__global__ void chase_lds(const int4* tab, int n, int steps, unsigned long long* ticks, int* sink) {
  extern __shared__ int4 sh[];
  for (int k = threadIdx.x; k < n; k += blockDim.x) sh[k] = tab[k];
  __syncthreads();
  int i = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < steps; k++) i = sh[i].x;
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { ticks[0] = t1 - t0; sink[0] = i; }
}
int main() {
  unsigned long long* ticks; int* sink;
  hipMalloc(&ticks, 8); hipMalloc(&sink, 4);
  for (int kb : {8, 32, 150, 1024, 8192, 65536}) {
    const int n = kb * 1024 / 16;
    std::vector<int> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 rng(1);
    std::shuffle(perm.begin() + 1, perm.end(), rng);
    std::vector<int4> tab(n);
    for (int k = 0; k < n; k++) tab[perm[k]] = {perm[(k + 1) % n], 0, 0, 0};  // one cycle through all records
    int4* d; hipMalloc(&d, (size_t)n * 16);
    hipMemcpy(d, tab.data(), (size_t)n * 16, hipMemcpyHostToDevice);
    const int steps = std::min(n, 20000);
    hipLaunchKernelGGL(chase_global, dim3(1), dim3(64), 0, 0, d, steps, ticks, sink);
    hipDeviceSynchronize();
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("global, %6d KB table: %7.1f cycles per dependent 16-byte load\n", kb, (double)t / steps);
    if (kb <= 32) {
      hipLaunchKernelGGL(chase_lds, dim3(1), dim3(64), (size_t)n * 16, 0, d, n, steps, ticks, sink);
      hipDeviceSynchronize();
      hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      printf("LDS,    %6d KB table: %7.1f cycles per dependent 16-byte load\n", kb, (double)t / steps);
    }
    hipFree(d);
  }
  return 0;
}
