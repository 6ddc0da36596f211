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

// sum of v[0..n) in index order (filtr.F:172-175, 414-417).  (Reading eight elements ahead of the additions was
// measured and is slower than what the compiler makes of the plain loop: 26 -> 30 us for the tracers' filter.)
UVIC_DEV double filt_ordered_sum(const double *v, int n) {
  double acc = 0.0;
  for (int p = 0; p < n; ++p) acc = acc + v[p];
  return acc;
}
// element `col` of F applied to s: sum over the source index in the reference's order (filtr.F:392-402)
UVIC_DEV double filt_matvec(const double *F, const double *s, int im, int col) {
  double acc = 0.0;
  for (int q = 0; q < im; ++q) acc = acc + s[q] * F[(size_t)q * im + col];
  return acc;
}

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
      const double ssum = filt_ordered_sum(s, im);
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
    if (tid < im) sp[tid] = it.fnorm * filt_matvec(F, s, im, tid);
  });
  env.par([&](int tid) {
    if (tid == 0) scal[2] = (scal[0] - filt_ordered_sum(sp, im)) * it.fimr;
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

// The application part of filtr.F:172-223, 392-428 on NSET strips of equal shape that already sit in the workgroup's
// tile, side by side: set h occupies s[h*H .. h*H+im), thread tid serves position tid % H of set tid / H (the
// workgroup has NSET*H threads).  Each is replaced by its filtered values.  sp: NSET*H doubles, scal: 4*NSET doubles.
template <class Env>
UVIC_DEV void filt_strips(Env &env, const FilterItem &it, const double *mats, double *s, double *sp, double *scal, int H) {
  const int im = it.im;
  if (it.mode == 2) {   // filter type 2 with n = 0
    env.par([&](int tid) { if (tid % H < im) s[tid] = 0.0; });
    return;
  }
  env.par([&](int tid) {
    if (tid % H == 0) {
      const int h = tid / H;
      const double ssum = filt_ordered_sum(s + h * H, im);
      scal[4 * h] = ssum;
      scal[4 * h + 1] = ssum * it.fimr;
    }
  });
  if (it.mode == 0) {
    env.par([&](int tid) { if (tid % H < im) s[tid] = scal[4 * (tid / H) + 1]; });
    return;
  }
  if (it.mode == 1) env.par([&](int tid) { if (tid % H < im) s[tid] = s[tid] - scal[4 * (tid / H) + 1]; });
  const double *F = mats + it.mat;
  env.par([&](int tid) {
    if (tid % H < im) sp[tid] = it.fnorm * filt_matvec(F, s + (tid / H) * H, im, tid % H);
  });
  if (it.mode == 3) {
    env.par([&](int tid) { if (tid % H < im) s[tid] = sp[tid]; });
    return;
  }
  env.par([&](int tid) {
    if (tid % H == 0) {
      const int h = tid / H;
      scal[4 * h + 2] = (scal[4 * h] - filt_ordered_sum(sp + h * H, im)) * it.fimr;
    }
  });
  env.par([&](int tid) { if (tid % H < im) s[tid] = scal[4 * (tid / H) + 2] + sp[tid]; });
}

// filuv.F:56-152 for one strip of one level of one row: rotate (u,v) to polar-stereographic components, filter both
// (side by side: the workgroup has 2*H threads, H >= im), rotate back.  u1, u2 = u(:,:,:,1:2,taup1).
// lds: 4*H + 8 doubles.
template <class Env>
UVIC_DEV void filuv_block(Env &env, int imt, int km, const FilterItem &it, const double *mats, const double *spsin,
                          const double *spcos, double *u1, double *u2, double *lds) {
  const int H = env.nthreads() / 2;
  double *t = lds, *sp = lds + 2 * H, *scal = lds + 4 * H;   // t[0..H): first component, t[H..2H): second
  const int im = it.im, j = it.j, k = it.k;
  const double fx = it.fx;
  auto col = [&](int p) {
    int i = it.is + p - 1;
    if (i > imt - 1) i -= imt - 2;
    return i;
  };
  env.par([&](int tid) {
    if (tid < im) {
      const int i = col(tid + 1);
      const double a = u1[X3(i, k, j)], b = u2[X3(i, k, j)];
      t[tid] = -fx * a * spsin[i - 1] - b * spcos[i - 1];
      t[H + tid] = fx * a * spcos[i - 1] - b * spsin[i - 1];
    }
  });
  filt_strips(env, it, mats, t, sp, scal, H);
  env.par([&](int tid) {
    if (tid < im) {
      const int i = col(tid + 1);
      u1[X3(i, k, j)] = fx * (-t[tid] * spsin[i - 1] + t[H + tid] * spcos[i - 1]);
      u2[X3(i, k, j)] = -t[tid] * spcos[i - 1] - t[H + tid] * spsin[i - 1];
    }
  });
}

// filuv.F:155-181 for one column of a filtered row: the vertical mean is removed again (from every level), then the
// land mask; with the cyclic images of clinic.F:506-509.  Both components march together, eight levels per batch.
UVIC_DEV void filuv_mean_column(int imt, int km, int i, int j, const int *kmu, const double *hr, const double *dzt, double *u1,
                                double *u2) {
  const int kb = kmu[X2(i, j)];
  double acc1 = 0.0, acc2 = 0.0;
  for (int k0 = 1; k0 <= km; k0 += 8) {
    double a[8], b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const size_t x = X3(i, (k0 + q <= km) ? k0 + q : km, j);
      a[q] = u1[x]; b[q] = u2[x];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (k0 + q <= km) {
        acc1 = acc1 + a[q] * dzt[k0 + q - 1];
        acc2 = acc2 + b[q] * dzt[k0 + q - 1];
      }
  }
  acc1 = acc1 * hr[X2(i, j)];
  acc2 = acc2 * hr[X2(i, j)];
  for (int k0 = 1; k0 <= km; k0 += 8) {
    double a[8], b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const size_t x = X3(i, (k0 + q <= km) ? k0 + q : km, j);
      a[q] = u1[x]; b[q] = u2[x];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = k0 + q;
      if (k <= km) {
        const double mask = (k <= kb) ? 1.0 : 0.0;
        const double v1 = (a[q] - acc1) * mask, v2 = (b[q] - acc2) * mask;
        u1[X3(i, k, j)] = v1; u2[X3(i, k, j)] = v2;
        if (i == 2) { u1[X3(imt, k, j)] = v1; u2[X3(imt, k, j)] = v2; }
        if (i == imt - 1) { u1[X3(1, k, j)] = v1; u2[X3(1, k, j)] = v2; }
      }
    }
  }
}

}  // namespace uvic
#endif
