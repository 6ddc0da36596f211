// Host->device transfer of the sizes the resident overlay moves per step, two ways: hipMemcpyAsync from page-locked memory
// (copy engine) and a kernel that reads the page-locked host memory itself (16 B per lane, coalesced).  GB/s and us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_pull(const double2 *src, double2 *dst, long long n2) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n2; q += (long long)gridDim.x * blockDim.x) dst[q] = src[q];
}
int main() {
  const size_t sizes[4] = {166464, 1581408, 4660992, 9800000};
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t bytes : sizes) {
    bytes = (bytes + 15) / 16 * 16;
    void *hostraw = malloc(bytes + 4096), *dev;
    void *host = (void *)(((size_t)hostraw + 4095) & ~(size_t)4095);
    for (size_t q = 0; q < bytes / 8; ++q) ((double *)host)[q] = (double)q;
    CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
    void *hd; CK(hipHostGetDevicePointer(&hd, host, 0));
    CK(hipMalloc(&dev, bytes));
    for (int blocks : {64, 256, 1024}) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, st, (const double2 *)hd, (double2 *)dev, (long long)(bytes / 16));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
      }
      printf("%8zu B  kernel pull %4d blocks: %7.1f us  %5.1f GB/s\n", bytes, blocks, best * 1e3, bytes / (best * 1e-3) / 1e9);
    }
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, st));
      CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    printf("%8zu B  hipMemcpyAsync         : %7.1f us  %5.1f GB/s\n", bytes, best * 1e3, bytes / (best * 1e-3) / 1e9);
    double chk; CK(hipMemcpy(&chk, (char *)dev + bytes - 8, 8, hipMemcpyDeviceToHost));
    if (chk != (double)(bytes / 8 - 1)) printf("  MISMATCH\n");
    CK(hipFree(dev)); CK(hipHostUnregister(host)); free(hostraw);
  }
  return 0;
}
