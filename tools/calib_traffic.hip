// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access width the transport kernels use
// (8 bytes per lane, coalesced along i): MI355X_MICROARCH.md calibrates only 16 B/lane and says
// other widths must be calibrated on a known byte count.  k_calib_read8 reads N doubles and writes
// N/64 doubles; k_calib_copy8 reads N and writes N.  Run under `rocprofv3 --pmc FETCH_SIZE` and
// `--pmc WRITE_SIZE` (separate passes); tools/pmc_summary.py divides the known bytes by the counters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_calib_read8(const double *__restrict__ x, double *__restrict__ y, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += x[i];
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) y[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = acc;
}
__global__ void k_calib_copy8(const double *__restrict__ x, double *__restrict__ y, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i] * 1.0000001;
}
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r)); return 1; } } while (0)
int main(int argc, char **argv) {
  const size_t n = (argc > 1 ? (size_t)atol(argv[1]) : (size_t)64) << 17;  // argv[1] MiB of doubles, default 64 MiB
  double *x, *y;
  CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8));
  CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_calib_read8, dim3(4096), dim3(256), 0, 0, x, y, n);
    hipLaunchKernelGGL(k_calib_copy8, dim3(4096), dim3(256), 0, 0, x, y, n);
  }
  CK(hipDeviceSynchronize());
  printf("calib bytes_read=%zu bytes_copy_read=%zu bytes_copy_write=%zu\n", n * 8, n * 8, n * 8);
  return 0;
}
