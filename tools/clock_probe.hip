// What core clock does the GPU sustain under a chip-wide fp64 VALU load, and how many cycles does a
// dependent / an independent v_fma_f64 take?  clock64() = s_memtime (core clock), wall_clock64() = 100 MHz.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_probe(double *out, long long *clk, int iters, int chains) {
  double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, b = 1e-9;
  const long long w0 = wall_clock64(), c0 = clock64();
  if (chains == 1) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a0 = __builtin_fma(a0, m, b);
    }
  } else {
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_fma(a0, m, b); a1 = __builtin_fma(a1, m, b); a2 = __builtin_fma(a2, m, b); a3 = __builtin_fma(a3, m, b);
      a4 = __builtin_fma(a4, m, b); a5 = __builtin_fma(a5, m, b); a6 = __builtin_fma(a6, m, b); a7 = __builtin_fma(a7, m, b);
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
  double *out; long long *clk, h[2];
  hipMalloc(&out, 8 * 256 * 65536); hipMalloc(&clk, 16);
  const int iters = 200000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpsimd = 1; wpsimd <= 8; wpsimd *= 2)
    for (int chains = 1; chains <= 8; chains *= 8) {
      // blocks of 256 threads (1 wave per SIMD); wpsimd blocks per CU on 256 CUs
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_probe, dim3(256 * wpsimd), dim3(256), 0, 0, out, clk, iters, chains);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      printf("  kernel %.3f ms -> %.1f TFLOP/s fp64 ", ms, 2.0 * 64 * 8.0 * iters * 4.0 * 256 * wpsimd / (ms * 1e-3) / 1e12);
      hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double ghz = (double)h[0] / ((double)h[1] * 10.0);   // core cycles per ns (100 MHz wall clock = 10 ns)
      printf("waves/SIMD %d  %s chains: %.2f core cycles per v_fma_f64 per wave, core clock %.2f GHz (%lld cycles in %.3f ms)\n",
             wpsimd, chains == 1 ? "1 dependent" : "8 independent", (double)h[0] / (8.0 * iters), ghz, h[0], h[1] * 1e-5);
    }
  return 0;
}
