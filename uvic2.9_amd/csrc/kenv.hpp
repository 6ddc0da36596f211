// kenv.hpp -- how a workgroup-level routine is written once and run in two places.
//
// Every kernel of this package is a `*_block` / `*_cell` / `*_column` function
// template over an execution environment `Env`:
//
//   env.par([&](int tid) { ... });   one barrier-delimited phase: the body runs
//                                    for every thread of the workgroup, then the
//                                    workgroup synchronises.
//
// On the GPU (`GpuEnv`, hipcc, gfx950) `par` is `f(threadIdx.x); __syncthreads()`.
// In the build container there is no GPU, so tests/hostemu compiles the very
// same sources with `HostEnv`, whose `par` runs the phase body for tid =
// 0..nthreads-1 in a loop.  Because no per-thread state may live across phases
// (everything goes through the workgroup's shared tile), both executions
// perform the same arithmetic in the same order: the host emulation checks the
// kernel LOGIC bit-for-bit against the oracle on the CPU.  It is test
// infrastructure only; the shipped library (libuvic_gpu.so) contains GPU code
// only and has no CPU path.
#ifndef UVIC_KENV_HPP
#define UVIC_KENV_HPP

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define UVIC_DEV __device__ __forceinline__
struct GpuEnv {
  __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
  template <class F>
  __device__ __forceinline__ void par(F &&f) {
    f((int)threadIdx.x);
    __syncthreads();
  }
};
#else
#define UVIC_DEV inline
#endif

struct HostEnv {
  int nth;
  int nthreads() const { return nth; }
  template <class F>
  void par(F &&f) {
    for (int t = 0; t < nth; ++t) f(t);
  }
};

namespace uvic {
UVIC_DEV double dmax(double a, double b) { return a > b ? a : b; }
UVIC_DEV double dmin(double a, double b) { return a < b ? a : b; }
UVIC_DEV int imax(int a, int b) { return a > b ? a : b; }
UVIC_DEV int imin(int a, int b) { return a < b ? a : b; }
UVIC_DEV double dabs(double a) { return __builtin_fabs(a); }
}  // namespace uvic

#endif
