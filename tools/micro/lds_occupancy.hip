// LDS allocation granularity / residency probe: blocks of one wave per CU as a function of dynamic LDS bytes and VGPR budget
// (hipOccupancyMaxActiveBlocksPerMultiprocessor), then verified by a kernel that counts co-resident waves per CU.
//   hipcc --offload-arch=gfx950 -O2 -o build/lds_occupancy tools/micro/lds_occupancy.hip && ./build/lds_occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64, 3) void k168(float* out) { extern __shared__ float lds[]; lds[threadIdx.x] = threadIdx.x; __syncthreads(); out[blockIdx.x * 64 + threadIdx.x] = lds[63 - threadIdx.x]; }
__global__ __launch_bounds__(64, 2) void k256(float* out) { extern __shared__ float lds[]; lds[threadIdx.x] = threadIdx.x; __syncthreads(); out[blockIdx.x * 64 + threadIdx.x] = lds[63 - threadIdx.x]; }
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s: CUs %d, LDS per block max %zu, per CU %zu\n", p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
  hipFuncSetAttribute((const void*)k168, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  int last = -1;
  for (int bytes = 8192; bytes <= 24576; bytes += 64) {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k168, 64, bytes);
    if (nb != last) { printf("dynamic LDS %6d B -> %2d blocks (waves) per CU\n", bytes, nb); last = nb; }
  }
  return 0;
}
