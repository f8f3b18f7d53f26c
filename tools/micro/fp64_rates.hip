// Issue cost (cycles per wave64 instruction, one wave per SIMD, dependent chain vs 8 independent chains) of the fp64 operations the
// portal search is made of, measured with s_memtime around unrolled loops.  Build: hipcc --offload-arch=gfx950 -O3 -o build/fp64_rates tools/micro/fp64_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
template <int OP> __device__ __forceinline__ double op(double a, double b, float f) {
  if (OP == 0) return __builtin_fma(a, b, a);
  if (OP == 1) return a + (double)f * 1e-30 + (double)(f + (float)a);  // two cvt_f64_f32 + cvt_f32_f64 + adds
  if (OP == 2) return a > b ? a : b * 1.0000001;
  if (OP == 3) return 1.0 / a;
  if (OP == 4) return sqrt(a);
  if (OP == 5) return (double)((float)a * f);  // cvt_f32_f64, mul_f32, cvt_f64_f32
  if (OP == 6) return a * b;
  return a;
}
template <int CHAINS> __global__ void kf(float* out, unsigned long long* ticks, float seed) {
  float v[CHAINS];
  for (int c = 0; c < CHAINS; c++) v[c] = seed + threadIdx.x * 1e-3f + c;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) v[c] = __builtin_fmaf(v[c], 1.0000001f, v[c]);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < CHAINS; c++) s += v[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
template <int CHAINS> void runf(float* out, unsigned long long* ticks) {
  hipLaunchKernelGGL((kf<CHAINS>), dim3(1), dim3(64), 0, 0, out, ticks, 1.5f);
  hipDeviceSynchronize();
  unsigned long long t;
  hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-34s chains %d: %7.1f ticks per op\n", "fma_f32", CHAINS, (double)t / (16.0 * N * CHAINS));
}
template <int OP, int CHAINS> __global__ void k(double* out, unsigned long long* ticks, double seed, float f) {
  double v[CHAINS];
  for (int c = 0; c < CHAINS; c++) v[c] = seed + threadIdx.x * 1e-3 + c;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) v[c] = op<OP>(v[c], 1.0000001, f);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int c = 0; c < CHAINS; c++) s += v[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
template <int OP, int CHAINS> void run(const char* name, double* out, unsigned long long* ticks) {
  hipLaunchKernelGGL((k<OP, CHAINS>), dim3(1), dim3(64), 0, 0, out, ticks, 1.5, 0.75f);
  hipDeviceSynchronize();
  unsigned long long t;
  hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-34s chains %d: %7.1f ticks per op\n", name, CHAINS, (double)t / (16.0 * N * CHAINS));
}
int main() {
  double* out; unsigned long long* ticks;
  hipMalloc(&out, 64 * 8); hipMalloc(&ticks, 8);
  // tick calibration: s_memtime vs wall clock
  run<0, 1>("fma_f64 (dependent)", out, ticks); run<0, 8>("fma_f64", out, ticks);
  run<6, 1>("mul_f64 (dependent)", out, ticks); run<6, 8>("mul_f64", out, ticks);
  run<1, 1>("2 cvt_f64_f32 + cvt_f32_f64 + 3 add", out, ticks); run<1, 8>("2 cvt_f64_f32 + cvt_f32_f64 + 3 add", out, ticks);
  run<5, 1>("cvt f64->f32, mul_f32, cvt f32->f64", out, ticks); run<5, 8>("cvt f64->f32, mul_f32, cvt f32->f64", out, ticks);
  run<2, 1>("cmp_gt_f64 + 2 cndmask + mul", out, ticks); run<2, 8>("cmp_gt_f64 + 2 cndmask + mul", out, ticks);
  run<3, 1>("1.0 / x (f64)", out, ticks); run<3, 8>("1.0 / x (f64)", out, ticks);
  run<4, 1>("sqrt (f64)", out, ticks); run<4, 8>("sqrt (f64)", out, ticks);
  runf<1>((float*)out, ticks); runf<8>((float*)out, ticks);
  // s_memtime against the wall clock
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<0, 1>), dim3(1), dim3(64), 0, 0, out, ticks, 1.5, 0.75f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("one launch of the dependent fma_f64 loop: %llu ticks inside the kernel, %.1f us between the events\n", t, ms * 1e3);
  }
  return 0;
}
