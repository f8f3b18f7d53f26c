// What would the Gauss-Seidel row step cost with TWO envs per wave (32 lanes each)?  The row step of hb_step_kernel as it is
// (lane = row, AR row in registers; per row: mul, max, v_readlane, fma, v_writelane) against the half-wave form, in which the
// turn-holder's step is broadcast inside each half: two v_readlane (lanes i and 32 + i), a v_cndmask that picks the half's own, the
// fma, two v_writelane.  Rows 0 .. NROW-1 per env, S sweeps, one wave alone and two waves on one SIMD (blocks of 512 threads = 8 waves
// on one CU: two per SIMD).  s_memtime counts core clocks here.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/pgs_two_env tools/micro/pgs_two_env.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
constexpr int NROW = 12, S = 4000;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
extern "C" __device__ int wl(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
template <int TWO, int I>
__device__ __forceinline__ void rows(const float (&ar)[NROW], float& res, int& dl, float nAinv, float nforce, bool upper, int hl) {
  if constexpr (I < NROW) {
    const float d = fmaxf(res * nAinv, nforce);
    if constexpr (TWO == 2) {
      // row_newbcast: every row of 16 lanes gets its own lane I % 16; v_permlane16_swap of the result with a copy of itself leaves one
      // register with rows (0, 0, 2, 2) and the other with rows (1, 1, 3, 3): the turn-holder's row of each half, no SGPR round trip
      const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x150 + (I & 15), 0xf, 0xf, false);
      const u32x2 sw = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
      const float di = __int_as_float((int)((I & 16) ? sw.y : sw.x));
      res = __builtin_fmaf(ar[I], di, res);
      dl = (hl == I) ? __float_as_int(di) : dl;
    } else if constexpr (TWO == 1) {
      const int a = __builtin_amdgcn_readlane(__float_as_int(d), I), b = __builtin_amdgcn_readlane(__float_as_int(d), 32 + I);
      const float di = __int_as_float(upper ? b : a);
      res = __builtin_fmaf(ar[I], di, res);
      dl = wl(a, I, dl); dl = wl(b, 32 + I, dl);
    } else {
      const int a = __builtin_amdgcn_readlane(__float_as_int(d), I);
      res = __builtin_fmaf(ar[I], __int_as_float(a), res);
      dl = wl(a, I, dl);
    }
    rows<TWO, I + 1>(ar, res, dl, nAinv, nforce, upper, hl);
  }
}
template <int TWO> __global__ void k(float* out, const float* in, unsigned long long* ticks) {
  const int lane = threadIdx.x & 63, hl = lane & 31;
  float ar[NROW];
#pragma unroll
  for (int i = 0; i < NROW; i++) ar[i] = in[(i * 64 + lane) % 1024] * (i == (TWO ? hl : lane) ? 1.f : 0.01f) + (i == (TWO ? hl : lane) ? 1.f : 0.f);
  float res = in[lane] - 0.5f, force = 0.f;
  const float nAinv = -1.f / ar[(TWO ? hl : lane) % NROW];
  const bool upper = lane >= 32;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int s = 0; s < S; s++) {
    const float nforce = -force;
    int dl = 0;
    rows<TWO, 0>(ar, res, dl, nAinv, nforce, upper, hl);
    force += __int_as_float(dl);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = res + force;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int TWO> void run(const char* name, int threads, float* out, float* in, unsigned long long* ticks) {
  hipLaunchKernelGGL((k<TWO>), dim3(1), dim3(threads), 0, 0, out, in, ticks);  // warm-up
  (void)hipDeviceSynchronize();
  auto w0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL((k<TWO>), dim3(1), dim3(threads), 0, 0, out, in, ticks);
  (void)hipDeviceSynchronize();
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
  unsigned long long t;
  (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double rows = (double)S * NROW, envs = TWO ? 2.0 : 1.0;
  printf("%-52s %d wave(s) per SIMD: %6.1f cycles per row step of a wave (s_memtime), %6.1f per row update per env; launch %.0f us = %.2f ns of the CU per row update per env\n", name, threads / 256 ? threads / 256 : 1,
         (double)t / rows, (double)t / rows / envs, us, 1e3 * us / (rows * envs * (threads / 64)));
}
int main() {
  float *out, *in; unsigned long long* ticks;
  (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&in, 1024 * 4); (void)hipMalloc(&ticks, 8);
  float h[1024]; for (int i = 0; i < 1024; i++) h[i] = 0.25f + 0.5f * ((i * 37) % 101) / 101.f;
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<0>("one env per wave (as hb_step_kernel)", 64, out, in, ticks);
  run<1>("two envs per wave (half-wave broadcast)", 64, out, in, ticks);
  run<2>("two envs per wave (row_newbcast + permlane16_swap)", 64, out, in, ticks);
  run<0>("one env per wave (as hb_step_kernel)", 512, out, in, ticks);
  run<1>("two envs per wave (half-wave broadcast)", 512, out, in, ticks);
  run<2>("two envs per wave (row_newbcast + permlane16_swap)", 512, out, in, ticks);
  return 0;
}
