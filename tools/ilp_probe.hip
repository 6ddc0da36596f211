// Cycles per fp64 VALU instruction per SIMD as a function of the waves resident on the SIMD and of the
// instruction-level parallelism inside each wave (independent chains), with and without a DPP move pair (+ s_nop) on the
// chain, as the column kernels have for every neighbour exchange.  Answers: how many waves does a SIMD need when
// every wave is one dependent chain?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP, int MIX>
__global__ void k(double *out, int iters) {
  double a[4], b = 1.0000001, c = 1e-9;
  for (int u = 0; u < 4; ++u) a[u] = threadIdx.x * 1e-3 + u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int u = 0; u < ILP; ++u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
      if (MIX == 1) {   // a DPP pair on the chain, as a neighbour exchange of a double
        asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(((int *)&a[0])[0]), "+v"(((int *)&a[0])[1]));
      }
    }
  }
  double s = 0; for (int u = 0; u < 4; ++u) s += a[u];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP, int MIX>
static void run(double *out, int wps, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 20000;
  hipLaunchKernelGGL((k<ILP, MIX>), dim3(256 * wps), dim3(256), 0, 0, out, 200);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<ILP, MIX>), dim3(256 * wps), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double nfma = 8.0 * ILP * iters * wps;   // fp64 instructions per SIMD
  printf("ILP %d mix %d waves/SIMD %d: %.2f cycles per v_fma_f64 per SIMD (%.2f per wave's own instruction)\n", ILP, MIX, wps,
         ms * 1e-3 * 2.3e9 / nfma, ms * 1e-3 * 2.3e9 / (8.0 * ILP * iters));
}
int main() {
  double *out; (void)hipMalloc(&out, 8 * 256 * 2048 * sizeof(double));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 1; w <= 4; ++w) { run<1, 0>(out, w, e0, e1); run<2, 0>(out, w, e0, e1); run<4, 0>(out, w, e0, e1); }
  for (int w = 1; w <= 4; ++w) { run<1, 1>(out, w, e0, e1); run<2, 1>(out, w, e0, e1); }
  return 0;
}
