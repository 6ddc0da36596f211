// Issue cost (core cycles per wave64 instruction, SIMD saturated with 8 resident waves) of the VALU
// instructions the transport kernels are made of.  One instruction kind per kernel, 8 independent
// register chains, inline asm so that the compiler cannot fold anything.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(name, ASM)                                                                   \
  __global__ void name(double *out, int iters) {                                            \
    double a[8], b = 1.0000001, c = 1e-9;                                                   \
    for (int u = 0; u < 8; ++u) a[u] = threadIdx.x * 1e-3 + u;                              \
    for (int i = 0; i < iters; ++i) {                                                       \
      REP8(ASM)                                                                             \
    }                                                                                       \
    double s = 0; for (int u = 0; u < 8; ++u) s += a[u];                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + b + c;                                 \
  }
#define A_FMA(u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
#define A_ADD(u) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[u]) : "v"(c));
#define A_MUL(u) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[u]) : "v"(b));
#define A_MAX(u) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[u]) : "v"(c));
#define A_RCP(u) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[u]));
#define A_FIX(u) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
#define A_SCL(u) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a[u]) : "v"(b) : "vcc");
#define A_FMS(u) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c) : "vcc");
#define A_CND(u) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(((int *)&a[u])[0]) : "v"(((int *)&b)[0]) : "vcc");
#define A_MOV(u) asm volatile("v_mov_b32 %0, %1" : "+v"(((int *)&a[u])[0]) : "v"(((int *)&b)[0]));
#define A_DPP(u) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(((int *)&a[u])[0]));
#define A_ADD64(u) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[u]) : "v"(b));
#define A_CMP(u) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[u]), "v"(b) : "vcc");
#define A_F32(u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(((float *)&a[u])[0]) : "v"(((float *)&b)[0]), "v"(((float *)&c)[0]));
#define A_CND64(u) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(((int *)&a[u])[0]) : "v"(((int *)&b)[0]) : "s20", "s21");
#define A_CNDNN(u) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(((int *)&a[u])[0]) : "v"(((int *)&c)[0]), "v"(((int *)&b)[0]) : "vcc");
#define A_AND(u) asm volatile("v_and_b32 %0, %0, %1" : "+v"(((int *)&a[u])[0]) : "v"(((int *)&b)[0]));
#define A_ADDU(u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(((int *)&a[u])[0]) : "v"(((int *)&b)[0]));
#define A_MOV64(u) asm volatile("v_mov_b64 %0, %1" : "+v"(a[u]) : "v"(b));
#define A_MIX(u) asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a[u]) : "v"(c), "v"(((int *)&b)[0]), "v"(((int *)&c)[0]) : "vcc");
#define A_MIX64(u) asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]" : "+v"(a[u]) : "v"(c), "v"(((int *)&b)[0]), "v"(((int *)&c)[0]) : "s20", "s21");
#define A_CMPCND(u) asm volatile("v_cmp_gt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a[u]), "v"(c), "v"(((int *)&b)[0]), "v"(((int *)&c)[0]) : "vcc");
#define A_CMPCND64(u) asm volatile("v_cmp_gt_f64 s[20:21], %0, %1\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]" : : "v"(a[u]), "v"(c), "v"(((int *)&b)[0]), "v"(((int *)&c)[0]) : "s20", "s21");
#define A_RDL(u) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(((int *)&a[u])[0]) : "s20");
KERNEL(k_fma, A_FMA) KERNEL(k_add, A_ADD) KERNEL(k_mul, A_MUL) KERNEL(k_max, A_MAX) KERNEL(k_rcp, A_RCP) KERNEL(k_fix, A_FIX)
KERNEL(k_scl, A_SCL) KERNEL(k_fms, A_FMS) KERNEL(k_cnd, A_CND) KERNEL(k_mov, A_MOV) KERNEL(k_dpp, A_DPP) KERNEL(k_add64, A_ADD64)
KERNEL(k_cmp, A_CMP) KERNEL(k_f32, A_F32) KERNEL(k_rdl, A_RDL)
KERNEL(k_mix, A_MIX) KERNEL(k_mix64, A_MIX64) KERNEL(k_cmpcnd, A_CMPCND) KERNEL(k_cmpcnd64, A_CMPCND64) KERNEL(k_cnd64, A_CND64) KERNEL(k_cndnn, A_CNDNN) KERNEL(k_and, A_AND) KERNEL(k_addu, A_ADDU) KERNEL(k_mov64, A_MOV64)
int main() {
  double *out; (void)hipMalloc(&out, 8 * 256 * 2048 * sizeof(double));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 100000, wps = 8;
  struct { const char *n; void (*k)(double *, int); } ks[] = {
      {"v_fma_f64", k_fma}, {"v_add_f64", k_add}, {"v_mul_f64", k_mul}, {"v_max_f64", k_max}, {"v_rcp_f64", k_rcp},
      {"v_div_fixup_f64", k_fix}, {"v_div_scale_f64", k_scl}, {"v_div_fmas_f64", k_fms}, {"v_cndmask_b32", k_cnd},
      {"v_mov_b32", k_mov}, {"v_mov_b32_dpp wave_shr", k_dpp}, {"v_lshl_add_u64", k_add64}, {"v_cmp_gt_f64", k_cmp},
      {"v_fma_f32", k_f32}, {"v_readlane_b32", k_rdl}, {"v_cndmask_b32_e64 sgpr mask", k_cnd64}, {"3 add_f64 + cndmask vcc (per 4)", k_mix}, {"3 add_f64 + cndmask sgpr (per 4)", k_mix64},
      {"cmp vcc + cndmask vcc (per 2)", k_cmpcnd}, {"cmp sgpr + cndmask sgpr (per 2)", k_cmpcnd64},
      {"v_cndmask_b32 no dst dep", k_cndnn}, {"v_and_b32", k_and}, {"v_add_u32", k_addu}, {"v_mov_b64", k_mov64}};
  for (auto &q : ks) {
    hipLaunchKernelGGL(q.k, dim3(256 * wps), dim3(256), 0, 0, out, 1000);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(q.k, dim3(256 * wps), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wps waves x 8*iters instructions in ms; cycles at 2.3 GHz
    printf("%-24s %6.2f ns per wave-instruction per SIMD = %5.2f cycles at 2.3 GHz\n", q.n, ms * 1e6 / (8.0 * iters * wps),
           ms * 1e6 / (8.0 * iters * wps) * 2.3);
  }
  return 0;
}
