// kernels_filter.hpp -- polar Fourier filter of t(tau+1).
//
// Replaces /root/reference/source/common/filt.F:85-107 (gather a strip, filter, scatter) and the
// application part of /root/reference/source/common/filtr.F:172-223, 392-428 for one strip of one
// level of one row and one tracer: s' = mean-corrected F s with the operator F built once by
// filter_host.hpp.  One workgroup per (strip, tracer); thread p owns position p of the strip.  The two
// sums over the strip (ssum, ssm) are taken by one thread in the reference's order, the matrix-vector
// product accumulates over the source index in the reference's order: bit-identical.
#ifndef UVIC_KERNELS_FILTER_HPP
#define UVIC_KERNELS_FILTER_HPP

#include "filter_item.h"
#include "kernels_isopyc.hpp"

namespace uvic {

// lds: 2*nthreads + 4 doubles
template <class Env>
UVIC_DEV void filt_block(Env &env, const uvic_ctx &c, const FilterItem &it, int n1, const double *mats, double *lds) {
  UV_DIMS(c);
  const int NT = env.nthreads();
  double *s = lds, *sp = lds + NT, *scal = lds + 2 * NT;
  double *t = c.t_taup1 + (size_t)(n1 - 1) * N3;
  const int im = it.im, j = it.j, k = it.k;
  // position p (1-based) of the strip -> column; strips that run past imt-1 continue at column 2 (filt.F:88-99)
  auto col = [&](int p) {
    int i = it.is + p - 1;
    if (i > imt - 1) i -= imt - 2;
    return i;
  };
  env.par([&](int tid) {
    if (tid < im) s[tid] = t[X3(col(tid + 1), k, j)];
  });
  env.par([&](int tid) {
    if (tid == 0) {
      double ssum = 0.0;
      for (int p = 0; p < im; ++p) ssum = ssum + s[p];
      scal[0] = ssum;
      scal[1] = ssum * it.fimr;   // stemp
    }
  });
  if (it.mode == 0) {   // n <= 1: the strip mean, filtr.F:196-203
    env.par([&](int tid) {
      if (tid < im) {
        const int i = col(tid + 1);
        const double v = scal[1];
        t[X3(i, k, j)] = v;
        if (i == 2) t[X3(imt, k, j)] = v;
        if (i == imt - 1) t[X3(1, k, j)] = v;
      }
    });
    return;
  }
  env.par([&](int tid) {
    if (tid < im) s[tid] = s[tid] - scal[1];
  });
  const double *F = mats + it.mat;
  env.par([&](int tid) {
    if (tid < im) {
      double acc = 0.0;
      for (int q = 0; q < im; ++q) acc = acc + s[q] * F[(size_t)q * im + tid];
      sp[tid] = it.fnorm * acc;
    }
  });
  env.par([&](int tid) {
    if (tid == 0) {
      double ssm = 0.0;
      for (int p = 0; p < im; ++p) ssm = ssm + sp[p];
      scal[2] = (scal[0] - ssm) * it.fimr;
    }
  });
  env.par([&](int tid) {
    if (tid < im) {
      const int i = col(tid + 1);
      const double v = scal[2] + sp[tid];
      t[X3(i, k, j)] = v;
      if (i == 2) t[X3(imt, k, j)] = v;       // the setbcx after filt, tracer.F:1252-1256
      if (i == imt - 1) t[X3(1, k, j)] = v;
    }
  });
}

}  // namespace uvic
#endif
