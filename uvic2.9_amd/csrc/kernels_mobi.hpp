// kernels_mobi.hpp -- MOBI biogeochemistry source terms, one ocean column per thread.
//
// Replaces, for the option set "c30" (BASELINE config 4 = SURVEY.md option set C:
// O_mobi O_mobi_o2 O_mobi_iron O_carbon O_mobi_alk O_mobi_nitrogen O_carbon_13
// O_carbon_14 O_mobi_nitrogen_15),
//   /root/reference/updates/09/source/mom/tracer.F:311-545, 853-867  (column set-up, iron inputs, 14C)
//   /root/reference/updates/09/source/mom/mobi.F:519-1482            (mobi_driver)
//   /root/reference/updates/09/source/mom/mobi.F:1485-3313           (mobi_src, nbio Euler sub-steps)
//   /root/reference/updates/09/source/common/co2calc.F:1-526         (co2calc_SWS, drtsafe, ta_iter_SWS)
// Not bandwidth-bound: ~1e4 fp64 flops and ~1e2 transcendentals per wet cell,
// levels strictly sequential (export of level k is the import of level k+1).
// Column state the reference keeps in COMMON and rewrites per grid box (ptn_P,
// k1n, the carbonate constants ...) is thread-private.  The column of tracers is
// not copied: level k is read from t(tau-1) (coalesced along i) when needed, and
// the clamp max(.,trcmin) that mobi_src applies to the caller's column
// (mobi.F:1894) is re-applied where mobi_driver reads the clamped values.
// Arithmetic order is the reference's; exp/log/pow/tanh come from the device
// math library, so results agree with the CPU oracle to rounding (not bitwise).
#ifndef UVIC_KERNELS_MOBI_HPP
#define UVIC_KERNELS_MOBI_HPP

#include <math.h>

#include "../../include/uvic_gpu.h"
#include "kernels_isopyc.hpp"

#define UV_TRCMIN 5e-12      /* updates/09/source/mom/mobi.h:199-212 */
#define UV_RN15STD 0.0036765
#define UV_RC13STD 0.0112372
#define UV_RC14STD 1.176e-12
#define UV_MOBI_MAXT 40

// The parameter block is written once by uvic_gpu_set_mobi and never during a kernel: device code
// reads it through the constant address space, so that the compiler may keep or re-order those
// (scalar) loads across the workgroup barriers of the team kernel instead of re-issuing them
// after every barrier.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) uvic_mobi_params *mobi_params_cp;
typedef const __attribute__((address_space(4))) uvic_mobi_index *mobi_index_cp;
#define UV_CONST_AS(p) ((mobi_params_cp)(p))
#else
typedef const uvic_mobi_params *mobi_params_cp;
typedef const uvic_mobi_index *mobi_index_cp;
#define UV_CONST_AS(p) (p)
#endif
// per-step scalars of tracer.F:311-343
struct mobi_step {
  int nbio, month;
  double dtbio, rdtts, rnbio, declin;
};
// device view of the MOBI inputs
struct mobi_dev {
  const uvic_mobi_params *P;   // device copy
  const uvic_mobi_options *O;  // device copy; null: option set C (the kernels of this file)
  const double *tlat, *dnswr, *aice, *hice, *hsno, *sg_bathy, *fe_atmdep, *fe_hydr;
  double pi, radian, relyr, co2ccn;
  // work planes of one MOBI pass (one set per stream that may run it):
  //   pre  MP_COUNT x (imt,km,jmt)  what mobi_pre_cell hands to the column kernel
  //   aux  MA_COUNT x (imt,km,jmt)  what the column kernel hands to mobi_post_cell
  //   col  2 x (imt,jmt)            calcite production of the column (prca, prca13)
  double *pre, *aux, *col;
  mobi_step S;
  // 1: the alkalinity function of the carbonate solve shares reciprocals (ta_iter_shared); 0: the reference's
  // expression, division by division (bit-exact on the host against the oracle)
  int carb_shared;
};
enum { MP_BCT, MP_BCTZ, MP_NUD, MP_AOUT, MP_O2F, MP_AVEJ, MP_AVEJD, MP_AC13B, MP_COUNT };
enum { MA_EXPO, MA_EXPOP, MA_RN15, MA_RC13, MA_CALPRO, MA_NFIX, MA_COUNT };
#define UV_MOBI_WORK_PLANES 20   /* the most any option set's three passes use (kernels_mobi_gt.hpp: 11 + 9) */
static_assert(MP_COUNT + MA_COUNT <= UV_MOBI_WORK_PLANES, "work planes of mobi_store");
static inline void mobi_set_work(mobi_dev *m, double *w, int imt, int jmt, int km) {
  const size_t n3 = (size_t)imt * km * jmt;
  m->pre = w;
  m->aux = w + MP_COUNT * n3;
  m->col = w + (size_t)UV_MOBI_WORK_PLANES * n3;
}
static inline size_t mobi_work_doubles(int imt, int jmt, int km) {
  return (size_t)UV_MOBI_WORK_PLANES * imt * km * jmt + (size_t)2 * imt * jmt;
}

// positions of the MOBI column tracers for option set C, fixed at compile time so that the
// column vectors live in registers (mobi.F:440-504 assigns them in this order);
// uvic_gpu_set_mobi rejects a parameter block whose `im` differs.
struct MI {
  enum : int { po4 = 1, phyt = 2, phyt_phos = 3, zoop = 4, detr = 5, detr_phos = 6, dic = 7, dic13 = 8, phytc13 = 9, zoopc13 = 10, detrc13 = 11, doc13 = 12, diazc13 = 13, dop = 14, no3 = 15, don = 16, diaz = 17, din15 = 18, don15 = 19, phytn15 = 20, zoopn15 = 21, detrn15 = 22, diazn15 = 23, dfe = 24, detrfe = 25, count = 25 };
};

// which wave of a team advances (and stores the tendency of) which pool: 0 nutrients and producers,
// 1 zooplankton, detritus, iron, 2 the 15N pools, 3 the 13C pools
static constexpr int MOBI_OWNER[MI::count] = {0, 0, 0, 1, 1, 1, 0, 3, 3, 3, 3, 3, 3, 0, 0, 0, 0, 2, 2, 2, 2, 2, 2, 1, 1};

// The device MOBI path is tolerance-tested (1e-11 on the sources): mul+add pairs may fuse there.  The host build of the
// same source (tests/hostemu) stays uncontracted and bit-identical to the oracle.  -DUV_NO_CONTRACT: measurement only.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(UV_NO_CONTRACT)
#pragma clang fp contract(fast)
#endif
namespace uvic {
UVIC_DEV double flag01(double x) { return 0.5 + copysign(0.5, x); }
UVIC_DEV double sq(double x) { return x * x; }

#if defined(__HIP_DEVICE_COMPILE__)
#define UV_DIVC(x, cst) ((x) * (1.0 / (cst)))   /* division by a literal: its reciprocal is folded at compile time */
/* x**y for x > 0 on the sub-step's critical path: exp(y*log(x)) is about half the instructions of the correctly
 * rounded pow and differs from it by |y log x| ulp at most (a few 1e-15 here) */
#define UV_POWP(x, y) exp((y) * log(x))
#define UV_POW10(x) exp((x) * 2.302585092994045684)
#define UV_POW15(x, sqrtx) ((x) * (sqrtx))   /* x**1.5 with sqrt(x) at hand */
/* 0.5 + 0.5*tanh(y) = 1/(1 + exp(-2y)): one exp and one reciprocal instead of the library tanh (about half its
 * instructions on the sub-step's longest phase); exp(-2y) = inf gives 0, the limit */
#define UV_HALF_TANH(y) (1.0 / (1.0 + exp(-2.0 * (y))))
#else
#define UV_DIVC(x, cst) ((x) / (cst))
#define UV_POWP(x, y) pow(x, y)
#define UV_POW10(x) pow(10., x)
#define UV_POW15(x, sqrtx) pow(x, 1.5)
#define UV_HALF_TANH(y) (0.5 + 0.5 * tanh(y))
#endif
typedef struct {
  double k1, k2, k1p, k2p, k3p, ksi, kw, ks, kf, kb, bt, st, ft, pt, sit, ta, dic;
} carb_t;

/* co2calc.F:455-526 */
UVIC_DEV void ta_iter_SWS(const carb_t *q, double x, double *fn, double *df) {
  const double x2 = x * x, x3 = x2 * x;
  const double k12 = q->k1 * q->k2, k12p = q->k1p * q->k2p, k123p = k12p * q->k3p;
  const double c = 1.0 + q->st / q->ks + q->ft / q->kf;
  const double a = x3 + q->k1p * x2 + k12p * x + k123p;
  const double a2 = a * a;
  const double da = 3.0 * x2 + 2.0 * q->k1p * x + k12p;
  const double b = x2 + q->k1 * x + k12;
  const double b2 = b * b;
  const double db = 2.0 * x + q->k1;
  const double dic = q->dic, pt = q->pt, bt = q->bt, st = q->st, ft = q->ft, sit = q->sit;
  *fn = q->k1 * x * dic / b + 2.0 * dic * k12 / b + bt / (1.0 + x / q->kb) + q->kw / x + pt * k12p * x / a +
        2.0 * pt * k123p / a + sit / (1.0 + x / q->ksi) - x / c - st / (1.0 + q->ks / (x / c)) -
        ft / (1.0 + q->kf / (x / c)) - pt * x3 / a - q->ta;
  *df = ((q->k1 * dic * b) - q->k1 * x * dic * db) / b2 - 2.0 * dic * k12 * db / b2 - bt / q->kb / sq(1.0 + x / q->kb) -
        q->kw / x2 + (pt * k12p * (a - x * da)) / a2 - 2.0 * pt * k123p * da / a2 - sit / q->ksi / sq(1.0 + x / q->ksi) -
        1.0 / c - st * (1.0 / sq(1.0 + q->ks / (x / c))) * (q->ks * c / x2) -
        ft * (1.0 / sq(1.0 + q->kf / (x / c))) * (q->kf * c / x2) - pt * x2 * (3.0 * a - x * da) / a2;
}

// 1/y for y well inside the normal range: v_rcp_f64 and two Newton steps on the device (5 instructions; the IEEE
// division sequence takes 12, four of them for range handling that cannot trigger here)
UVIC_DEV double recip_nr(double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(y);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  return __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
#else
  return 1.0 / y;
#endif
}
// Quotients of the ecosystem sub-step whose denominator is a clamped pool (>= trcmin), a pool plus a positive
// half-saturation constant, or 1 + a positive ratio: finite, far from the ends of the exponent range, never zero.  On
// the device they skip the range handling of the IEEE sequence (8 instructions instead of 12, on the dependent chain
// of every sub-step); the host build divides, so that the emulation stays bit-identical to the oracle.  Quotients whose
// denominator is a difference of pools (it may vanish: the reference relies on +-inf being clamped) keep `/`.
UVIC_DEV double div_safe(double x, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(y);                       // relative error 2^-23
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);       // one Newton step: 2^-46
  const double q = x * r;
  return __builtin_fma(__builtin_fma(-y, q, x), r, q);      // corrected quotient: error (2^-46)^2, below the rounding
#else
  return x / y;
#endif
}
// The same function and derivative as ta_iter_SWS with every denominator inverted once: 7 reciprocals per
// evaluation instead of 27 divisions (the solve evaluates it ~13 times per cell; divisions were more than half of
// mobi_pre's instructions).  x in [1e-10, 1e-6], every denominator between 1e-36 and 1e3: no range handling needed.
// Terms keep the reference's order, so values differ from ta_iter_SWS by a few ulp of the largest term.
typedef struct {
  double rc, ckx_s, ckx_f, rkb, rksi;   // 1/c, ks*c, kf*c, 1/kb, 1/ksi: the same for every evaluation of a cell
} carb_inv_t;
UVIC_DEV void ta_iter_shared(const carb_t *q, const carb_inv_t *v, double x, double *fn, double *df) {
  const double x2 = x * x, x3 = x2 * x;
  const double k12 = q->k1 * q->k2, k12p = q->k1p * q->k2p, k123p = k12p * q->k3p;
  const double a = x3 + q->k1p * x2 + k12p * x + k123p;
  const double da = 3.0 * x2 + 2.0 * q->k1p * x + k12p;
  const double b = x2 + q->k1 * x + k12;
  const double db = 2.0 * x + q->k1;
  const double dic = q->dic, pt = q->pt, bt = q->bt, st = q->st, ft = q->ft, sit = q->sit;
  const double rx = recip_nr(x), rx2 = rx * rx;
  const double ra = recip_nr(a), ra2 = ra * ra, rb = recip_nr(b), rb2 = rb * rb;
  const double rub = recip_nr(1.0 + x * v->rkb), rus = recip_nr(1.0 + x * v->rksi);
  const double rvs = recip_nr(1.0 + v->ckx_s * rx), rvf = recip_nr(1.0 + v->ckx_f * rx);
  *fn = q->k1 * x * dic * rb + 2.0 * dic * k12 * rb + bt * rub + q->kw * rx + pt * k12p * x * ra + 2.0 * pt * k123p * ra +
        sit * rus - x * v->rc - st * rvs - ft * rvf - pt * x3 * ra - q->ta;
  *df = ((q->k1 * dic * b) - q->k1 * x * dic * db) * rb2 - 2.0 * dic * k12 * db * rb2 - bt * v->rkb * (rub * rub) - q->kw * rx2 +
        (pt * k12p * (a - x * da)) * ra2 - 2.0 * pt * k123p * da * ra2 - sit * v->rksi * (rus * rus) - v->rc -
        st * (rvs * rvs) * (v->ckx_s * rx2) - ft * (rvf * rvf) * (v->ckx_f * rx2) - pt * x2 * (3.0 * a - x * da) * ra2;
}

/* co2calc.F:401-453: bracketed Newton (Numerical Recipes rtsafe) */
template <bool SHARED>
UVIC_DEV double drtsafe_t(const carb_t *q, double x1, double x2, double xacc) {
  carb_inv_t v;
  if (SHARED) {
    const double c = 1.0 + q->st / q->ks + q->ft / q->kf;
    v.rc = 1.0 / c; v.ckx_s = q->ks * c; v.ckx_f = q->kf * c; v.rkb = 1.0 / q->kb; v.rksi = 1.0 / q->ksi;
  }
#define ta_iter_SWS(q, x, f, d) do { if (SHARED) ta_iter_shared(q, &v, x, f, d); else (ta_iter_SWS)(q, x, f, d); } while (0)
  const int maxit = 100;
  double fl, fh, df, f, xl, xh, swap, r, dxold, dx, temp;
  ta_iter_SWS(q, x1, &fl, &df);
  ta_iter_SWS(q, x2, &fh, &df);
  if (fl < 0.0) {
    xl = x1; xh = x2;
  } else {
    xh = x1; xl = x2;
    swap = fl; fl = fh; fh = swap;
  }
  r = 0.5 * (x1 + x2);
  dxold = fabs(x2 - x1);
  dx = dxold;
  ta_iter_SWS(q, r, &f, &df);
  for (int j = 1; j <= maxit; ++j) {
    if (((r - xh) * df - f) * ((r - xl) * df - f) >= 0. || fabs(2.0 * f) > fabs(dxold * df)) {
      dxold = dx;
      dx = 0.5 * (xh - xl);
      r = xl + dx;
      if (xl == r) return r;
    } else {
      dxold = dx;
      dx = f / df;
      temp = r;
      r = r - dx;
      if (temp == r) return r;
    }
    if (fabs(dx) < xacc) return r;
    ta_iter_SWS(q, r, &f, &df);
    if (f < 0.0) {
      xl = r; fl = f;
    } else {
      xh = r; fh = f;
    }
  }
  (void)fl; (void)fh;
  return r;
#undef ta_iter_SWS
}
UVIC_DEV double drtsafe(const carb_t *q, double x1, double x2, double xacc, int shared) {
  return shared ? drtsafe_t<true>(q, x1, x2, xacc) : drtsafe_t<false>(q, x1, x2, xacc);
}

/* co2calc.F:1-399; only the outputs mobi_driver uses are returned */
UVIC_DEV void mobi_co2calc_SWS(double t, double s, double dic_in, double ta_in, double co2_in, double atmpres, double depth,
                     double *ph, double *co2star_o, double *dco2star_o, double *pCO2_o, double *dpco2_o, double *CO3_o,
                     double *Omega_c, double *Omega_a, int carb_shared = 0) {
  carb_t q;
  const double phhi = 6., phlo = 10.;
  const double sit_in = 7.6875e-03, pt_in = 0.5125e-3;
  const double permil = 1.0 / 1024.5;
  q.pt = pt_in * permil;
  q.sit = sit_in * permil;
  q.ta = ta_in * permil;
  q.dic = dic_in * permil;
  const double C2K = 273.15;
  const double pres = depth * 0.1;
  const double permeg = 1.e-6;
  const double co2 = co2_in * permeg;
  const double tk = C2K + t;
  const double tk100 = tk / 100.0;
  const double tk1002 = tk100 * tk100;
  const double invtk = 1.0 / tk;
  const double dlogtk = log(tk);
  const double is = 19.924 * s / (1000. - 1.005 * s);
  const double is2 = is * is;
  const double sqrtis = sqrt(is);
  const double s2 = s * s;
  const double t2 = t * t;
  const double sqrts = sqrt(s);
  const double s15 = UV_POW15(s, sqrts);
  const double scl = s / 1.80655;
  const double pitkR = pres / tk / 83.15;
  const double p2itkR = pres * pitkR;
  q.bt = 0.000232 * scl / 10.811;
  q.st = 0.14 * scl / 96.062;
  q.ft = 0.000067 * scl / 18.9984;
  const double ff = exp(-162.8301 + 218.2968 / tk100 + 90.9241 * log(tk100) - 1.47696 * tk1002 +
                        s * (.025695 - .025225 * tk100 + 0.0049867 * tk1002));
  const double k0 = exp(93.4517 / tk100 - 60.2409 + 23.3585 * log(tk100) + s * (.023517 - 0.023656 * tk100 + 0.0047036 * tk1002));
  const double rt_x = 83.1451 * tk;
  const double delta_x = (57.7 - 0.118 * tk);
  double b_x = -1636.75 + 12.0408 * tk - 0.0327957 * tk * tk;
  b_x = b_x + 3.16528 * 1e-5 * tk * tk * tk;
  const double FugFac = exp((b_x + 2 * delta_x) * 1 / rt_x);
  q.k1 = UV_POW10(-1. * (3670.7 * invtk - 62.008 + 9.7944 * dlogtk - 0.0118 * s + 0.000116 * s2)) *
         exp((25.5 - 0.1271 * t) * pitkR + 0.5 * (-3.08e-3 + 8.77e-5 * t) * p2itkR);
  q.k2 = UV_POW10(-1 * (1394.7 * invtk + 4.777 - 0.0184 * s + 0.000118 * s2)) *
         exp((15.82 + 0.0219 * t) * pitkR + 0.5 * (1.13e-3 - 1.475e-4 * t) * p2itkR);
  q.k1p = exp(-4576.752 * invtk + 115.540 - 18.453 * dlogtk + (-106.736 * invtk + 0.69171) * sqrts +
              (-0.65643 * invtk - 0.01844) * s) *
          exp((14.51 - 0.1211 * t + 3.21e-4 * t2) * pitkR + 0.5 * (-2.67e-3 + 4.27e-5 * t) * p2itkR);
  q.k2p = exp(-8814.715 * invtk + 172.1033 - 27.927 * dlogtk + (-160.340 * invtk + 1.3566) * sqrts +
              (0.37335 * invtk - 0.05778) * s) *
          exp((23.12 - 0.1758 * t + 2.647e-3 * t2) * pitkR + 0.5 * (-5.15e-3 + 9.0e-5 * t) * p2itkR);
  q.k3p = exp(-3070.75 * invtk - 18.126 + (17.27039 * invtk + 2.81197) * sqrts + (-44.99486 * invtk - 0.09984) * s) *
          exp((26.57 - 0.202 * t + 3.042e-3 * t2) * pitkR + 0.5 * (-4.08e-3 + 7.14e-5 * t) * p2itkR);
  q.ksi = exp(-8904.2 * invtk + 117.400 - 19.334 * dlogtk + (-458.79 * invtk + 3.5913) * sqrtis +
              (188.74 * invtk - 1.5998) * is + (-12.1652 * invtk + 0.07871) * is2 + log(1.0 - 0.001005 * s)) *
          exp((29.48 - 0.1622 * t - 2.608e-3 * t2) * pitkR + 0.5 * (-2.84e-3) * p2itkR);
  q.kw = exp(-13847.26 * invtk + 148.9802 - 23.6521 * dlogtk + (118.67 * invtk - 5.977 + 1.0495 * dlogtk) * sqrts -
             0.01615 * s) *
         exp((20.02 - 0.1119 * t + 1.409e-3 * t2) * pitkR + 0.5 * (-5.13e-3 + 7.94e-5 * t) * p2itkR);
  q.ks = exp(-4276.1 * invtk + 141.328 - 23.093 * dlogtk + (-13856 * invtk + 324.57 - 47.986 * dlogtk) * sqrtis +
             (35474 * invtk - 771.54 + 114.723 * dlogtk) * is - 2698 * invtk * UV_POW15(is, sqrtis) + 1776 * invtk * is2 +
             log(1.0 - 0.001005 * s)) *
         exp((18.03 - .0466 * t - 3.16e-4 * t2) * pitkR + 0.5 * (-4.53e-3 + 9.0e-5 * t) * p2itkR);
  q.kf = exp(1590.2 * invtk - 12.641 + 1.525 * sqrtis + log(1.0 - 0.001005 * s)) *
         exp((9.78 + 9.0e-3 * t + 9.42e-4 * t2) * pitkR + 0.5 * (-3.91e-3 + 5.4e-5 * t) * p2itkR);
  q.kb = exp((-8966.90 - 2890.53 * sqrts - 77.942 * s + 1.728 * s15 - 0.0996 * s2) * invtk +
             (148.0248 + 137.1942 * sqrts + 1.62142 * s) + (-24.4344 - 25.085 * sqrts - 0.2474 * s) * dlogtk +
             0.053105 * sqrts * tk + log((1 + (q.st / q.ks) + (q.ft / q.kf)) / (1 + (q.st / q.ks)))) *
         exp((29.48 - 0.1622 * t - 2.608e-3 * t2) * pitkR + 0.5 * (-2.84e-3) * p2itkR);
  const double x1 = pow(10.0, -phhi);
  const double x2 = pow(10.0, -phlo);
  const double xacc = 1.e-10;
  const double hSWS = drtsafe(&q, x1, x2, xacc, carb_shared);
  const double hSWS2 = hSWS * hSWS;
  double co2star = q.dic * hSWS2 / (hSWS2 + q.k1 * hSWS + q.k1 * q.k2);
  const double co2starair = co2 * ff * atmpres;
  double dco2star = co2starair - co2star;
  *ph = -log10(hSWS);
  double pCO2 = co2star / (k0 * FugFac);
  double dpCO2 = pCO2 - co2starair;
  double CO3 = q.k1 * q.k2 * co2star / hSWS2;
  const double sqs = pow(s, 0.5), sq35 = pow(s / 35., 0.5);
  double Kspc = exp(-395.8293 + (6537.773 / tk) + 71.595 * log(tk) - 0.17959 * tk +
                    (-1.78938 + (410.64 / tk) + 0.0065453 * tk) * sqs - 0.17755 * s + 0.0094979 * s15);
  double Kspa = exp(-395.9180 + (6685.079 / tk) + 71.595 * log(tk) - 0.17959 * tk +
                    (-0.157481 + (202.938 / tk) + 0.0039780 * tk) * sqs - 0.23067 * s + 0.0136808 * s15);
  const double DVc = -65.28 + 0.397 * t - 0.005155 * (t * t) + (19.816 - 0.0441 * t - 0.00017 * (t * t)) * sq35;
  const double DVa = -65.50 + 0.397 * t - 0.005155 * (t * t) + (19.82 - 0.0441 * t - 0.00017 * (t * t)) * sq35;
  const double DK = 0.01847 + 0.0001956 * t - 0.000002212 * (t * t) + (-0.03217 - 0.0000711 * t + 0.000002212) * sq35;
  Kspc = Kspc * exp(-DVc * pitkR + 0.5 * DK * p2itkR);
  Kspa = Kspa * exp(-DVa * pitkR + 0.5 * DK * p2itkR);
  const double Ca = 10.28E-3;
  *Omega_c = Ca * CO3 / Kspc;
  *Omega_a = Ca * CO3 / Kspa;
  *co2star_o = co2star / permil;
  *dco2star_o = dco2star / permil;
  *CO3_o = CO3 / permil;
  *pCO2_o = pCO2 / permeg;
  *dpco2_o = dpCO2 / permeg;
}


typedef struct {
  double expo, expo_phos, calpro, nfix, rn15expo, rc13expo, expofe, remife;
} src_out_t;

/* Rayleigh-type fractionation factor: r + eps*(1-u)/u*log(1-u)*r/1000 (e.g. mobi.F:2589-2600) */
UVIC_DEV double rayleigh(double r, double eps, double u) { return r + UV_DIVC(div_safe(eps * (1 - u), u) * log(1 - u) * r, 1000.); }   /* u in [trcmin, 0.999] */
UVIC_DEV double clamp_ratio(double r, double hi, double lo) {
  r = dmin(r, hi);
  r = dmax(r, lo);
  return r;
}

// ---------------------------------------------------------------------------
// Team execution of the sub-step loop.  One ocean column is a strictly sequential chain of
// km levels x nbio Euler sub-steps (~2000 fp64 instructions each) and there are only ~7000
// columns, so a one-thread-per-column kernel is bound by the latency of that chain, not by
// throughput.  A team of four waves (one per SIMD of a CU) therefore works on the SAME 64
// columns: inside a sub-step each wave evaluates one independent group of rates
//   role 0 growth and nutrient limitation, 15N assimilation fractionation
//   role 1 grazing, mortality, remineralisation, export, 15N recycling fractionation
//   role 2 iron speciation and scavenging
//   role 3 isotope ratios (15N, 13C)
// publishes them in LDS, and after one workgroup barrier every wave applies the identical
// pool update, so all four hold the same state again.  Every quantity is still computed by
// the same expression as in the one-thread form (bit-identical results); only wave 0 stores.
// `NoTeam` runs all roles in one thread (host oracle comparison, single-wave kernel).
// ---------------------------------------------------------------------------
#define UV_MOBI_XN 37  /* rates exchanged per column and sub-step */
#define UV_MOBI_YN 51  /* pools, flags and P:N ratios exchanged per column and sub-step */
#define UV_MOBI_LDS_DOUBLES ((size_t)2 * (UV_MOBI_XN + UV_MOBI_YN) * 64)
struct NoTeam {
  static constexpr bool team = false;
  static constexpr int role = 0;
  int lane = 0;
  double *xs = nullptr;
  unsigned xc = 0;
#ifdef UV_MOBI_TIMING
  long long tq[8];
#endif
  UVIC_DEV void sync() const {}
};

template <class Team>
UVIC_DEV void mobi_src(Team &T, mobi_params_cp P, const mobi_step &S, double *bioin, double bct, double impo, double impo_phos,
                     double wwd, double nud, double nudop, double nudon, double *bioout, double bctz,
                     double rn15impo, double rc13impo, double ac13b, double impofe, double o2flag, double aou_term,
                     double avej, double avej_D, src_out_t *out) {
#define BIN(m) bioin[(m)-1]
  double biopo4 = BIN(MI::po4), biophyt = BIN(MI::phyt), biophyt_phos = BIN(MI::phyt_phos), biozoop = BIN(MI::zoop);
  double biodetr = BIN(MI::detr), biodetr_phos = BIN(MI::detr_phos);
  double ptn_P = biophyt_phos / biophyt;
  double ptn_detr = biodetr_phos / biodetr;
  double biodic = BIN(MI::dic), biodop = BIN(MI::dop), biono3 = BIN(MI::no3), biodon = BIN(MI::don), biodiaz = BIN(MI::diaz);
  double biodin15 = BIN(MI::din15), biodon15 = BIN(MI::don15), biophytn15 = BIN(MI::phytn15), biozoopn15 = BIN(MI::zoopn15);
  double biodetrn15 = BIN(MI::detrn15), biodiazn15 = BIN(MI::diazn15);
  double biodic13 = BIN(MI::dic13), biophytc13 = BIN(MI::phytc13), biozoopc13 = BIN(MI::zoopc13), biodetrc13 = BIN(MI::detrc13);
  double biodoc13 = BIN(MI::doc13), biodiazc13 = BIN(MI::diazc13), biodfe = BIN(MI::dfe), biodetrfe = BIN(MI::detrfe);
  /* negative-prevention flags from the unclamped input, mobi.F:1814-1891 */
  double po4flag = flag01(biopo4 - UV_TRCMIN), phytflag = flag01(biophyt - UV_TRCMIN), zoopflag = flag01(biozoop - UV_TRCMIN);
  double detrflag = flag01(biodetr - UV_TRCMIN), phyt_phosflag = flag01(biophyt_phos - UV_TRCMIN);
  double detr_phosflag = flag01(biodetr_phos - UV_TRCMIN);
  const double sf_P_phosflag = flag01(ptn_P - P->gamma1 * P->redptn);
  const double sf_detr_phosflag = flag01(ptn_detr - P->gamma1 * P->redptn);
  double dopflag = flag01(biodop - UV_TRCMIN), no3flag = flag01(biono3 - UV_TRCMIN), donflag = flag01(biodon - UV_TRCMIN);
  double diazflag = flag01(biodiaz - UV_TRCMIN), din15flag = flag01(biodin15 - UV_TRCMIN), don15flag = flag01(biodon15 - UV_TRCMIN);
  double phytn15flag = flag01(biophytn15 - UV_TRCMIN), zoopn15flag = flag01(biozoopn15 - UV_TRCMIN);
  double detrn15flag = flag01(biodetrn15 - UV_TRCMIN), diazn15flag = flag01(biodiazn15 - UV_TRCMIN);
  double dic13flag = flag01(biodic13 - UV_TRCMIN), phytc13flag = flag01(biophytc13 - UV_TRCMIN);
  double zoopc13flag = flag01(biozoopc13 - UV_TRCMIN), detrc13flag = flag01(biodetrc13 - UV_TRCMIN);
  double doc13flag = flag01(biodoc13 - UV_TRCMIN), diazc13flag = flag01(biodiazc13 - UV_TRCMIN);
  double dfeflag = flag01(biodfe - UV_TRCMIN), detrfeflag = flag01(biodetrfe - UV_TRCMIN);
  /* clamp the caller's column and the working copies, mobi.F:1894-1960 */
  _Pragma("unroll") for (int m = 0; m < MI::count; ++m) bioin[m] = dmax(bioin[m], UV_TRCMIN);
  biopo4 = dmax(biopo4, UV_TRCMIN); biophyt = dmax(biophyt, UV_TRCMIN); biozoop = dmax(biozoop, UV_TRCMIN);
  biodetr = dmax(biodetr, UV_TRCMIN); biophyt_phos = dmax(biophyt_phos, UV_TRCMIN); biodetr_phos = dmax(biodetr_phos, UV_TRCMIN);
  biodic = dmax(biodic, UV_TRCMIN); biono3 = dmax(biono3, UV_TRCMIN); biodop = dmax(biodop, UV_TRCMIN);
  biodon = dmax(biodon, UV_TRCMIN); biodiaz = dmax(biodiaz, UV_TRCMIN); biodin15 = dmax(biodin15, UV_TRCMIN);
  biodon15 = dmax(biodon15, UV_TRCMIN); biophytn15 = dmax(biophytn15, UV_TRCMIN); biozoopn15 = dmax(biozoopn15, UV_TRCMIN);
  biodetrn15 = dmax(biodetrn15, UV_TRCMIN); biodiazn15 = dmax(biodiazn15, UV_TRCMIN); biodic13 = dmax(biodic13, UV_TRCMIN);
  biophytc13 = dmax(biophytc13, UV_TRCMIN); biozoopc13 = dmax(biozoopc13, UV_TRCMIN); biodetrc13 = dmax(biodetrc13, UV_TRCMIN);
  biodoc13 = dmax(biodoc13, UV_TRCMIN); biodiazc13 = dmax(biodiazc13, UV_TRCMIN); biodfe = dmax(biodfe, UV_TRCMIN);
  biodetrfe = dmax(biodetrfe, UV_TRCMIN);
  /* the light-limited growth rates avej, avej_D (mobi.F:1984-2061) depend on the inputs only: mobi_pre_cell */
  const double gmax = P->gbio * bctz;
  const double nupt = P->nupt0 * bct;
  const double nupt_D = P->nupt0_D * bct;
  double nfixout = 0.0, expoout = 0.0, expo_phosout = 0.0, rn15expoout = 0.0, rc13expoout = 0.0, calproout = 0.0;
  double expofeout = 0.0, remifeout = 0.0;
  const double dtbio = S.dtbio, redctn = P->redctn, redptn = P->redptn, gamma1 = P->gamma1, geZ = P->geZ;
  const double dfr = P->dfr, dfrt = P->dfrt, pfr = P->pfr, rnd = P->redntp / P->diazntp; /* (redntp/diazntp) */
  const double nr_excr_P = 0.0, nr_excr_detr = 0.0;
  const double rn15hi = 2. * UV_RN15STD / (1 + UV_RN15STD), rn15lo = UV_RN15STD / (1 + UV_RN15STD) / 2.;
  const double rc13hi = 2. * UV_RC13STD / (1 + UV_RC13STD), rc13lo = 0.5 * UV_RC13STD / (1 + UV_RC13STD);

  for (int n = 1; n <= S.nbio; ++n) { /* mobi.F:2148-3252 */
#ifdef UV_MOBI_TIMING
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); long long tq0 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#define TQ(q) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const long long tq1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); T.tq[q] += tq1 - tq0; tq0 = tq1; }
#else
#define TQ(q)
#endif
#define ROLE(r) (!Team::team || Team::role == (r))
    // outputs of the four roles
    double npp = 0., npp_D = 0., no3upt_D = 0., dopupt = 0., dopupt_D = 0., fcassim = 0.;
    // th_no3, the nitrate switch of diazotroph uptake (a tanh, ~650 cycles), is formed by the isotope-ratio wave.
    // Cycles of this phase per sub-step with fenced counters (-DUV_MOBI_TIMING): growth 2000, grazing 1050, iron 2300,
    // ratios 870 before; growth 1350, grazing 1050, iron 1500 (its two pow() as exp(y*log(x))), ratios 1530 after.
    double th_no3 = 0.;
    double graz = 0., graz_Z = 0., graz_Det = 0., graz_D = 0., morp = 0., morpt = 0., morz = 0., remi = 0., expo = 0.;
    double expo_phos = 0., recy_dop = 0., recy_don = 0., morp_D = 0., morpt_D = 0., fcrecy = 0.;
    double feorgads = 0., fecol = 0., expofe = 0., remife = 0.;
    double fcexcr = 0., rtphytn15 = 0., rtzoopn15 = 0., rtdetrn15 = 0., rtdiazn15 = 0., fcnpp = 0.;
    double rtphytc13 = 0., rtzoopc13 = 0., rtdetrc13 = 0., rtdoc13 = 0., rtdiazc13 = 0.;
    if (ROLE(0)) {  // ---- growth and nutrient limitation (mobi.F:2150-2236), 15N assimilation (:2589-2600)
      const double p1 = dmin(biophyt, P->pmax);
      const double p2 = dmax(0.0, biophyt - P->pmax);
      const double k1n = div_safe(P->knmin * p1 + P->knmax * p2, p1 + p2);
      const double k1p_P = k1n * ptn_P;
      const double kfevar = div_safe(P->kfemin * p1 + P->kfemax * p2, p1 + p2);
      const double deffe = div_safe(biodfe, kfevar + biodfe);
      const double jmax = P->abio_P * bct * deffe;
      const double deffe_D = div_safe(biodfe, P->kfe_D + biodfe);
      const double jmax_D = dmax(0., P->abio_P * (bct - P->dbct_D) * deffe_D) * P->jdiar;
      const double limP_dop = div_safe(P->hdop * biodop, k1p_P + biodop);
      const double limP_po4 = div_safe(biopo4, k1p_P + biopo4);
      const double dopupt_flag = flag01(limP_dop - limP_po4);
      const double limP = limP_dop * dopupt_flag + limP_po4 * (1. - dopupt_flag);
      double u_P = dmin(avej, jmax * limP);
      u_P = dmin(u_P, div_safe(jmax * biono3, k1n + biono3));
      const double u_D = dmin(avej_D, jmax_D * limP);
      const double dopupt_D_flag = dopupt_flag;
      npp = u_P * biophyt;
      dopupt = npp * dopupt_flag; /* NB: from the unflagged npp, mobi.F:2236 */
      npp_D = dmax(0., u_D * biodiaz);
      no3upt_D = npp_D;   /* times th_no3 = 0.5 + 0.5*tanh(biono3 - 5), applied after the exchange (mobi.F:2233) */
      dopupt_D = npp_D * dopupt_D_flag;
      npp = npp * no3flag * (dopupt_flag * dopflag + (1. - dopupt_flag) * po4flag) * din15flag;
      npp_D = npp_D * (dopupt_D_flag * dopflag + (1. - dopupt_D_flag) * po4flag) * din15flag;
      no3upt_D = no3upt_D * no3flag * din15flag;
      double uno3 = div_safe(npp * dtbio, biono3);
      uno3 = dmin(uno3, 0.999);
      uno3 = dmax(uno3, UV_TRCMIN);
      const double rno3 = clamp_ratio(biodin15 / (biono3 - biodin15), 2 * UV_RN15STD, UV_RN15STD / 2.);
      const double bassim = rayleigh(rno3, P->eps_assim, uno3);
      fcassim = div_safe(bassim, 1 + bassim);
    }
    if (ROLE(1)) {  // ---- grazing, mortality, remineralisation, export (mobi.F:2223-2312), 15N recycling
      const double thetaZ = P->zprefP * biophyt + P->zprefDet * biodetr + P->zprefZ * biozoop + P->zprefDiaz * biodiaz + P->kzoo;
      const double ing_P = div_safe(P->zprefP, thetaZ), ing_Det = div_safe(P->zprefDet, thetaZ), ing_Z = div_safe(P->zprefZ, thetaZ);
      const double ing_D = div_safe(P->zprefDiaz, thetaZ);   /* thetaZ >= kzoo > 0 */
      const double g_D = gmax * ing_D * biodiaz;
      graz_D = g_D * biozoop;
      morpt_D = nupt_D * biodiaz;
      morp_D = P->nup_D * biodiaz * biodiaz;
      const double g_P = gmax * ing_P * biophyt;
      graz = g_P * biozoop;
      const double g_Z = gmax * ing_Z * biozoop;
      graz_Z = g_Z * biozoop;
      const double g_Det = gmax * ing_Det * biodetr;
      graz_Det = g_Det * biozoop;
      morp = P->nup * biophyt;
      morpt = nupt * biophyt;
      recy_don = nudon * bct * biodon;
      recy_dop = nudop * bct * biodop;
      morz = P->nuz * biozoop * biozoop;
      remi = nud * bct * biodetr;
      expo = wwd * biodetr;
      expo_phos = wwd * biodetr_phos;
      /* negative prevention, mobi.F:2343-2445 */
      graz = graz * phytflag * phyt_phosflag * sf_P_phosflag * phytn15flag;
      graz_Z = graz_Z * zoopflag * zoopn15flag;
      graz_Det = graz_Det * detrflag * detr_phosflag * sf_detr_phosflag * detrn15flag;
      morp = morp * phytflag * phyt_phosflag * phytn15flag;
      morpt = morpt * phytflag * phyt_phosflag * phytn15flag;
      morz = morz * zoopflag * zoopn15flag;
      remi = remi * detrflag * detr_phosflag * detrn15flag;
      expo = expo * detrflag * detrn15flag;
      expo_phos = expo_phos * detr_phosflag;
      recy_dop = recy_dop * dopflag;
      graz_D = graz_D * diazflag * diazn15flag;
      morpt_D = morpt_D * diazflag * diazn15flag;
      morp_D = morp_D * diazflag * diazn15flag;
      recy_don = recy_don * donflag * don15flag;
      double udon = div_safe(recy_don * dtbio, biodon);
      udon = dmin(udon, 0.999);
      udon = dmax(udon, UV_TRCMIN);
      const double rdon = clamp_ratio(biodon15 / (biodon - biodon15), 2 * UV_RN15STD, UV_RN15STD / 2.);
      const double brecy = rayleigh(rdon, P->eps_recy, udon);
      fcrecy = div_safe(brecy, 1 + brecy);
    }
    if (ROLE(2)) {  // ---- iron speciation and scavenging, mobi.F:2313-2342
      remife = nud * bct * biodetrfe;
      const double ligand = UV_DIVC(dmax(aou_term + UV_DIVC(UV_POWP(biodon, 0.8), 4.8), 0.5), 1000.);
      const double fepa = (1.0 + P->kfeleq * (ligand - biodfe)) * o2flag;
      const double feprime = div_safe(-fepa + sqrt(fepa * fepa + 4.0 * P->kfeleq * biodfe), 2.0 * P->kfeleq) * o2flag;
      feorgads = (P->kfeorg * (UV_POWP((biodetr * detrflag) * P->mc * redctn, 0.58)) * feprime) * o2flag;
      fecol = P->kfecol * (feprime * feprime) * o2flag;
      expofe = wwd * biodetrfe;
      remife = remife * detrfeflag;
      feorgads = feorgads * dfeflag;
      expofe = expofe * detrfeflag;
      fecol = fecol * dfeflag;
    }
    if (ROLE(3)) {  // ---- isotope ratios, mobi.F:2601-2695; the nitrate switch (see above)
      th_no3 = UV_HALF_TANH(biono3 - 5.);
      const double rzoop = clamp_ratio(biozoopn15 / (biozoop - biozoopn15), 2. * UV_RN15STD, UV_RN15STD / 2.);
      const double bexcr = rzoop - UV_DIVC(P->eps_excr * rzoop, 1000.);
      fcexcr = div_safe(bexcr, 1 + bexcr);
      rtphytn15 = clamp_ratio(div_safe(biophytn15, biophyt), rn15hi, rn15lo);
      rtzoopn15 = clamp_ratio(div_safe(biozoopn15, biozoop), rn15hi, rn15lo);
      rtdetrn15 = clamp_ratio(div_safe(biodetrn15, biodetr), rn15hi, rn15lo);
      rtdiazn15 = clamp_ratio(div_safe(biodiazn15, biodiaz), rn15hi, rn15lo);
      const double rdic13 = clamp_ratio(biodic13 / (biodic - biodic13), 2. * UV_RC13STD, 0.5 * UV_RC13STD);
      const double bc13npp = ac13b * rdic13;
      fcnpp = div_safe(bc13npp, 1 + bc13npp);
      rtphytc13 = clamp_ratio(div_safe(biophytc13, biophyt * redctn), rc13hi, rc13lo);
      rtzoopc13 = clamp_ratio(div_safe(biozoopc13, biozoop * redctn), rc13hi, rc13lo);
      rtdetrc13 = clamp_ratio(div_safe(biodetrc13, biodetr * redctn), rc13hi, rc13lo);
      rtdoc13 = clamp_ratio(div_safe(biodoc13, biodon * redctn), rc13hi, rc13lo);
      rtdiazc13 = clamp_ratio(div_safe(biodiazc13, biodiaz * redctn), rc13hi, rc13lo);
    }
    TQ(0)
    if (Team::team) {  // publish own group, one barrier, fetch the other three
      double *xb = T.xs + (size_t)(T.xc & 1u) * UV_MOBI_XN * 64 + T.lane;
      ++T.xc;
#define XA(X) X(0, npp) X(1, npp_D) X(2, no3upt_D) X(3, dopupt) X(4, dopupt_D) X(5, fcassim)
#define XB(X) X(6, graz) X(7, graz_Z) X(8, graz_Det) X(9, graz_D) X(10, morp) X(11, morpt) X(12, morz) X(13, remi) \
  X(14, expo) X(15, expo_phos) X(16, recy_dop) X(17, recy_don) X(18, morp_D) X(19, morpt_D) X(20, fcrecy)
#define XC(X) X(21, feorgads) X(22, fecol) X(23, expofe) X(24, remife)
#define XD(X) X(25, fcexcr) X(26, rtphytn15) X(27, rtzoopn15) X(28, rtdetrn15) X(29, rtdiazn15) X(30, fcnpp) \
  X(31, rtphytc13) X(32, rtzoopc13) X(33, rtdetrc13) X(34, rtdoc13) X(35, rtdiazc13) X(36, th_no3)
#define XPUT(sl, v) xb[(size_t)(sl) * 64] = v;
#define XGET(sl, v) v = xb[(size_t)(sl) * 64];
      if (Team::role == 0) { XA(XPUT) } else if (Team::role == 1) { XB(XPUT) } else if (Team::role == 2) { XC(XPUT) } else { XD(XPUT) }
      TQ(1)
      T.sync();
      TQ(2)
      if (Team::role != 0) { XA(XGET) }
      if (Team::role != 1) { XB(XGET) }
      if (Team::role != 2) { XC(XGET) }
      if (Team::role != 3) { XD(XGET) }
#undef XA
#undef XB
#undef XC
#undef XD
#undef XPUT
#undef XGET
    }
#undef ROLE
    TQ(3)
    no3upt_D = th_no3 * no3upt_D;   /* mobi.F:2233; the flags (0 or 1) are already in, which leaves the product unchanged */
    /* zooplankton budget, mobi.F:2446-2530 */
    const double dig_P = gamma1 * graz, dig_Z = gamma1 * graz_Z, dig_Det = gamma1 * graz_Det;
    double dig = dig_Z + dig_P + dig_Det;
    const double excr_P = gamma1 * (1 - geZ) * graz, excr_Z = gamma1 * (1 - geZ) * graz_Z;
    const double excr_Det = gamma1 * (1 - geZ) * graz_Det;
    double excr = excr_Z + excr_P + excr_Det;
    const double sf_P = (1. - gamma1) * graz, sf_Z = (1. - gamma1) * graz_Z, sf_Det = (1. - gamma1) * graz_Det;
    double sf = sf_P + sf_Z + sf_Det;
    const double sf_P_phos = (graz * ptn_P - dig_P * redptn);
    const double sf_Det_phos = (graz_Det * ptn_detr - dig_Det * redptn);
    double sf_phos = sf_P_phos + sf_Z * redptn + sf_Det_phos;
    const double dig_D = gamma1 * graz_D * rnd;
    dig = dig + dig_D;
    const double excr_D = gamma1 * (1 - geZ) * graz_D * rnd;
    excr = excr + excr_D;
    const double nr_excr_D = gamma1 * graz_D * (1 - rnd) + (1 - gamma1) * graz_D * (1 - rnd);
    const double sf_D = (1 - gamma1) * graz_D * rnd;
    sf = sf + sf_D;
    sf_phos = sf_phos + sf_D * redptn;
    const double bnfix = UV_RN15STD - P->eps_nfix * UV_RN15STD / 1000.;
    const double fcnfix = bnfix / (1 + bnfix);
    const double calpro = (morp + morz + (graz + graz_Z) * (1. - gamma1)) * P->capr * redctn * 1.e3;
    /* variable P:C of new production (Galbraith & Martiny 2015), mobi.F:2721-2724 */
    const double GM15ptc = 0.0060 + 0.0069 * biopo4;
    const double GM15ptn = GM15ptc * redctn * 1.e3;
    const double diazptn = P->diazptn, rfeton = P->rfeton;
    /* prognostic updates, mobi.F:2738-3085; every right-hand side uses the OLD state.  In a team each
       wave advances the pools it owns (0: nutrients and producers, 1: zooplankton, detritus, iron,
       2: 15N, 3: 13C), refreshes their flags and publishes both; every wave then holds the full state. */
#define OWN(r) (!Team::team || Team::role == (r))
    if (OWN(0)) {
      const double n_po4 = biopo4 + dtbio * (dopupt * ptn_P - GM15ptn * npp + (1. - dfrt) * morpt * ptn_P +
                                           (1. - pfr) * remi * ptn_detr + diazptn * (morpt_D - (npp_D - dopupt_D)) +
                                           recy_dop + redptn * (excr));
      const double n_dop = biodop + dtbio * (dfr * morp * ptn_P + dfrt * morpt * ptn_P + pfr * remi * ptn_detr -
                                           ptn_P * dopupt - diazptn * dopupt_D - recy_dop);
      const double n_phyt = biophyt + dtbio * (npp - morp - graz - morpt);
      const double n_phyt_phos = biophyt_phos + dtbio * (npp * GM15ptn - morp * ptn_P - graz * ptn_P - morpt * ptn_P);
      const double n_dic = biodic + dtbio * redctn * (excr + (1. - pfr) * remi + (1. - dfrt) * morpt - npp + morpt_D - npp_D +
                                                    recy_don + nr_excr_D + nr_excr_P + nr_excr_detr + morp_D * (1. - rnd));
      const double n_no3 = biono3 + dtbio * (excr + (1. - pfr) * remi + (1. - dfrt) * morpt - npp + morpt_D - no3upt_D +
                                           recy_don + nr_excr_D + nr_excr_P + nr_excr_detr + morp_D * (1. - rnd));
      const double n_don = biodon + dtbio * (dfr * morp + dfrt * morpt + pfr * remi - recy_don);
      const double n_diaz = biodiaz + dtbio * (npp_D - morp_D - morpt_D - graz_D);
      biopo4 = n_po4; biodop = n_dop; biophyt = n_phyt; biophyt_phos = n_phyt_phos; biodic = n_dic; biono3 = n_no3;
      biodon = n_don; biodiaz = n_diaz;
    }
    if (OWN(1)) {
      const double n_zoop = biozoop + dtbio * (dig - morz - graz_Z - excr);
      const double n_detr = biodetr + dtbio * ((1. - dfr) * morp + sf + morz - remi - graz_Det - expo + impo + morp_D * rnd);
      const double n_detr_phos = biodetr_phos + dtbio * ((1. - dfr) * morp * ptn_P + sf_phos + morz * redptn - remi * ptn_detr -
                                                       graz_Det * ptn_detr - expo_phos + impo_phos + morp_D * rnd * redptn);
      const double n_dfe = biodfe + dtbio * (rfeton * (excr + (1. - dfrt) * morpt - npp + morpt_D - npp_D + recy_don +
                                                     nr_excr_D + nr_excr_P + nr_excr_detr + morp_D * (1. - rnd)) -
                                           feorgads + remife - fecol);
      const double n_detrfe = biodetrfe + dtbio * (rfeton * (sf + (1. - dfr) * morp + morp_D * rnd + morz - graz_Det) +
                                                 feorgads + P->iscr * fecol - remife - expofe + impofe);
      biozoop = n_zoop; biodetr = n_detr; biodetr_phos = n_detr_phos; biodfe = n_dfe; biodetrfe = n_detrfe;
    }
    if (OWN(2)) {
      const double n_din15 =
        biodin15 + dtbio * (rtphytn15 * (1. - dfrt) * morpt + rtphytn15 * nr_excr_P + fcexcr * excr + rtdiazn15 * morpt_D +
                            rtdiazn15 * nr_excr_D + rtdiazn15 * morp_D * (1. - rnd) + rtdetrn15 * (1. - pfr) * remi +
                            rtdetrn15 * nr_excr_detr + fcrecy * recy_don - fcassim * npp - fcassim * no3upt_D);
      const double n_don15 =
        biodon15 + dtbio * (dfr * rtphytn15 * morp + dfrt * rtphytn15 * morpt + rtdetrn15 * pfr * remi - fcrecy * recy_don);
      const double n_phytn15 = biophytn15 + dtbio * (fcassim * npp - rtphytn15 * morp - rtphytn15 * graz - rtphytn15 * morpt);
      const double n_zoopn15 = biozoopn15 + dtbio * (rtphytn15 * dig_P + rtzoopn15 * dig_Z + rtdetrn15 * dig_Det +
                                                   rtdiazn15 * dig_D - rtzoopn15 * morz - rtzoopn15 * graz_Z - fcexcr * excr);
      const double n_detrn15 =
        biodetrn15 + dtbio * (rtphytn15 * (1. - dfr) * morp + rtphytn15 * sf_P + rtzoopn15 * sf_Z + rtdetrn15 * sf_Det +
                              rtdiazn15 * sf_D + rtzoopn15 * morz - rtdetrn15 * remi - rtdetrn15 * graz_Det -
                              rtdetrn15 * expo + rn15impo * impo + rtdiazn15 * morp_D * rnd);
      const double n_diazn15 = biodiazn15 + dtbio * (fcnfix * (npp_D - no3upt_D) + fcassim * no3upt_D - rtdiazn15 * morp_D -
                                                   rtdiazn15 * graz_D - rtdiazn15 * morpt_D);
      biodin15 = n_din15; biodon15 = n_don15; biophytn15 = n_phytn15; biozoopn15 = n_zoopn15; biodetrn15 = n_detrn15;
      biodiazn15 = n_diazn15;
    }
    if (OWN(3)) {
      const double n_dic13 =
        biodic13 + dtbio * redctn * (rtphytc13 * (1. - dfrt) * morpt + rtphytc13 * nr_excr_P + rtzoopc13 * excr +
                                     rtdiazc13 * morpt_D + rtdiazc13 * nr_excr_D + rtdiazc13 * morp_D * (1 - rnd) +
                                     rtdetrc13 * (1. - pfr) * remi + rtdetrc13 * nr_excr_detr + rtdoc13 * recy_don -
                                     fcnpp * npp - fcnpp * npp_D);
      const double n_doc13 = biodoc13 + dtbio * redctn * (dfr * rtphytc13 * morp + rtphytc13 * dfrt * morpt +
                                                        rtdetrc13 * pfr * remi - rtdoc13 * recy_don);
      const double n_phytc13 =
        biophytc13 + dtbio * redctn * (fcnpp * npp - rtphytc13 * morp - rtphytc13 * graz - rtphytc13 * morpt);
      const double n_zoopc13 =
        biozoopc13 + dtbio * redctn * (rtphytc13 * dig_P + rtzoopc13 * dig_Z + rtdetrc13 * dig_Det + rtdiazc13 * dig_D -
                                       rtzoopc13 * morz - rtzoopc13 * graz_Z - rtzoopc13 * excr);
      const double n_detrc13 =
        biodetrc13 + dtbio * redctn * (rtphytc13 * (1. - dfr) * morp + rtphytc13 * sf_P + rtzoopc13 * sf_Z + rtdetrc13 * sf_Det +
                                       rtdiazc13 * sf_D + rtzoopc13 * morz - rtdetrc13 * remi - rtdetrc13 * graz_Det -
                                       rtdetrc13 * expo + rc13impo + rtdiazc13 * morp_D * rnd);
      const double n_diazc13 = biodiazc13 + dtbio * redctn * (fcnpp * npp_D - rtdiazc13 * (morp_D + graz_D + morpt_D));
      biodic13 = n_dic13; biodoc13 = n_doc13; biophytc13 = n_phytc13; biozoopc13 = n_zoopc13; biodetrc13 = n_detrc13;
      biodiazc13 = n_diazc13;
    }
    /* accumulate, mobi.F:3088-3172 */
    expoout = expoout + expo;
    expo_phosout = expo_phosout + expo_phos;
    rn15expoout = rn15expoout + rtdetrn15;
    rc13expoout = rc13expoout + rtdetrc13 * expo;
    calproout = calproout + calpro;
    nfixout = nfixout + npp_D - no3upt_D;
    expofeout = expofeout + expofe;
    remifeout = remifeout + remife;
    /* the P:N ratios are refreshed from the new pools (mobi.F:2846-2849); flags that are still set are
       refreshed (mobi.F:3175-3251) */
    if (OWN(0)) {
      ptn_P = biophyt_phos / biophyt;
      if (po4flag == 1) po4flag = flag01(biopo4 - UV_TRCMIN);
      if (phytflag == 1) phytflag = flag01(biophyt - UV_TRCMIN);
      if (phyt_phosflag == 1) phyt_phosflag = flag01(biophyt_phos - UV_TRCMIN);
      if (no3flag == 1) no3flag = flag01(biono3 - UV_TRCMIN);
      if (dopflag == 1) dopflag = flag01(biodop - UV_TRCMIN);
      if (donflag == 1) donflag = flag01(biodon - UV_TRCMIN);
      if (diazflag == 1) diazflag = flag01(biodiaz - UV_TRCMIN);
    }
    if (OWN(1)) {
      ptn_detr = biodetr_phos / biodetr;
      if (zoopflag == 1) zoopflag = flag01(biozoop - UV_TRCMIN);
      if (detrflag == 1) detrflag = flag01(biodetr - UV_TRCMIN);
      if (detr_phosflag == 1) detr_phosflag = flag01(biodetr_phos - UV_TRCMIN);
      if (dfeflag == 1) dfeflag = flag01(biodfe - UV_TRCMIN);
      if (detrfeflag == 1) detrfeflag = flag01(biodetrfe - UV_TRCMIN);
    }
    if (OWN(2)) {
      if (din15flag == 1) din15flag = flag01(biodin15 - UV_TRCMIN);
      if (don15flag == 1) don15flag = flag01(biodon15 - UV_TRCMIN);
      if (phytn15flag == 1) phytn15flag = flag01(biophytn15 - UV_TRCMIN);
      if (zoopn15flag == 1) zoopn15flag = flag01(biozoopn15 - UV_TRCMIN);
      if (detrn15flag == 1) detrn15flag = flag01(biodetrn15 - UV_TRCMIN);
      if (diazn15flag == 1) diazn15flag = flag01(biodiazn15 - UV_TRCMIN);
    }
    if (OWN(3)) {
      if (dic13flag == 1) dic13flag = flag01(biodic13 - UV_TRCMIN);
      if (phytc13flag == 1) phytc13flag = flag01(biophytc13 - UV_TRCMIN);
      if (zoopc13flag == 1) zoopc13flag = flag01(biozoopc13 - UV_TRCMIN);
      if (detrc13flag == 1) detrc13flag = flag01(biodetrc13 - UV_TRCMIN);
      if (doc13flag == 1) doc13flag = flag01(biodoc13 - UV_TRCMIN);
      if (diazc13flag == 1) diazc13flag = flag01(biodiazc13 - UV_TRCMIN);
    }
    TQ(4)
    if (Team::team) {  // second exchange: new pools, their flags, the P:N ratios
      double *yb = T.xs + (size_t)2 * UV_MOBI_XN * 64 + (size_t)((T.xc - 1u) & 1u) * UV_MOBI_YN * 64 + T.lane;
#define YA(X) X(0, biopo4) X(1, biodop) X(2, biophyt) X(3, biophyt_phos) X(4, biodic) X(5, biono3) X(6, biodon) X(7, biodiaz) \
  X(8, ptn_P) X(9, po4flag) X(10, phytflag) X(11, phyt_phosflag) X(12, no3flag) X(13, dopflag) X(14, donflag) X(15, diazflag)
#define YB(X) X(16, biozoop) X(17, biodetr) X(18, biodetr_phos) X(19, biodfe) X(20, biodetrfe) X(21, ptn_detr) \
  X(22, zoopflag) X(23, detrflag) X(24, detr_phosflag) X(25, dfeflag) X(26, detrfeflag)
#define YC(X) X(27, biodin15) X(28, biodon15) X(29, biophytn15) X(30, biozoopn15) X(31, biodetrn15) X(32, biodiazn15) \
  X(33, din15flag) X(34, don15flag) X(35, phytn15flag) X(36, zoopn15flag) X(37, detrn15flag) X(38, diazn15flag)
#define YD(X) X(39, biodic13) X(40, biodoc13) X(41, biophytc13) X(42, biozoopc13) X(43, biodetrc13) X(44, biodiazc13) \
  X(45, dic13flag) X(46, phytc13flag) X(47, zoopc13flag) X(48, detrc13flag) X(49, doc13flag) X(50, diazc13flag)
#define YPUT(sl, v) yb[(size_t)(sl) * 64] = v;
#define YGET(sl, v) v = yb[(size_t)(sl) * 64];
      if (Team::role == 0) { YA(YPUT) } else if (Team::role == 1) { YB(YPUT) } else if (Team::role == 2) { YC(YPUT) } else { YD(YPUT) }
      TQ(5)
      T.sync();
      TQ(6)
      if (Team::role != 0) { YA(YGET) }
      if (Team::role != 1) { YB(YGET) }
      if (Team::role != 2) { YC(YGET) }
      if (Team::role != 3) { YD(YGET) }   // (what a role never reads is dropped by the compiler: roles are compile-time)
    }
#undef OWN
    TQ(7)
#undef TQ
  }
#undef YA
#undef YB
#undef YC
#undef YD
#undef YPUT
#undef YGET
  (void)dic13flag; (void)doc13flag; (void)phytc13flag; (void)zoopc13flag; (void)detrc13flag; (void)diazc13flag;
#define BOUT(m, v) bioout[(m)-1] = (v) /* the new pools; the caller forms the tendency (mobi.F:3255-3313) */
  BOUT(MI::po4, biopo4); BOUT(MI::phyt, biophyt); BOUT(MI::phyt_phos, biophyt_phos); BOUT(MI::zoop, biozoop);
  BOUT(MI::detr, biodetr); BOUT(MI::detr_phos, biodetr_phos); BOUT(MI::dic, biodic); BOUT(MI::dop, biodop);
  BOUT(MI::no3, biono3); BOUT(MI::don, biodon); BOUT(MI::diaz, biodiaz); BOUT(MI::din15, biodin15);
  BOUT(MI::don15, biodon15); BOUT(MI::phytn15, biophytn15); BOUT(MI::zoopn15, biozoopn15); BOUT(MI::detrn15, biodetrn15);
  BOUT(MI::diazn15, biodiazn15); BOUT(MI::dfe, biodfe); BOUT(MI::detrfe, biodetrfe); BOUT(MI::dic13, biodic13);
  BOUT(MI::phytc13, biophytc13); BOUT(MI::zoopc13, biozoopc13); BOUT(MI::detrc13, biodetrc13); BOUT(MI::doc13, biodoc13);
  BOUT(MI::diazc13, biodiazc13);
  out->expo = expoout; out->expo_phos = expo_phosout; out->calpro = calproout; out->nfix = nfixout;
  out->rn15expo = rn15expoout; out->rc13expo = rc13expoout; out->expofe = expofeout; out->remife = remifeout;
#undef BIN
#undef BOUT
}


// ---------------------------------------------------------------------------
// MOBI runs in three passes.  mobi_driver walks a column top-down because the export of
// level k is the import of level k+1, but most of what it evaluates per level depends only
// on the level's own inputs at tau-1 (or on the light that reaches it, a function of the
// inputs above).  Only the nbio Euler sub-steps and the hand-down of the export are a true
// vertical sequence, so
//   mobi_pre_cell    one thread per cell: carbonate chemistry, light, temperature and oxygen
//                    functions, light-limited growth rates            -> `pre` planes
//   mobi_column_body one thread (or team of waves) per column: the sub-steps -> src, `aux`, `col`
//   mobi_post_cell   one thread per cell: benthic and water-column denitrification, sedimentary
//                    iron, bottom remineralisation, DIC/alkalinity/13C/14C bookkeeping -> src
// Every quantity keeps the reference's expression and operation order.
// ---------------------------------------------------------------------------
#define UV_MOBI_LOCALS(c, M)                                   \
  UV_DIMS(c);                                                  \
  mobi_params_cp P = UV_CONST_AS(M.P);                         \
  const mobi_step &S = M.S;                                    \
  mobi_index_cp Q = &P->is;                                    \
  const size_t ij = X2(i, j), NS = (size_t)imt * jmt;          \
  (void)S; (void)Q; (void)ij; (void)NS
#define TM(k, n) c.t_taum1[X3(i, k, j) + (size_t)((n)-1) * N3]
#define TNC(k, m) dmax(TM(k, P->tracer_of_mobi[(m)-1]), UV_TRCMIN) /* clamped column value, mobi.F:1894 */
#define PRE(q) M.pre[(size_t)(q) * N3 + X3(i, k, j)]
#define AUX(q) M.aux[(size_t)(q) * N3 + X3(i, k, j)]

// co2calc_SWS (called at mobi.F:772 for every level) needs T, S, DIC and alkalinity of the
// cell only; of its outputs option set C uses CO2* through the 13C fractionation factor
// ac13b = ac13_aq_POC / ac13_DIC_aq (mobi.F:775-789).  Light: tracer.F:381-390, mobi.F:735-760.
// `part`: 0 = everything (host emulation, one thread per cell), 1 = the carbonate chemistry alone, 2 = the rest.  The
// two halves share nothing but their inputs, so the device gives each cell two threads: on a small latitude slab,
// where the pass is one round of waves, its latency (the head of the MOBI chain that bounds multi-GPU scaling) halves.
UVIC_DEV void mobi_pre_cell(const uvic_ctx &c, const mobi_dev &M, int i, int k, int j, int part = 0) {
  UV_MOBI_LOCALS(c, M);
  if (k > c.kmt[ij]) return;
  const double t_in = TM(k, P->itemp);
  const double s_in = 1.e3 * TM(k, P->isalt) + 35.0;
  const double dic_in = TM(k, P->idic), alk_in = TM(k, P->ialk);
  const double o2_in = TM(k, P->io2) * 1000.;
  if (part != 2) {
    const double atmpres = 1.0, depth = P->zt[k - 1] / 100.;
    double pH, co2star, dco2star, pCO2, dpco2, CO3, Omega_c, Omega_a;
    mobi_co2calc_SWS(t_in, s_in, dic_in, alk_in, M.co2ccn, atmpres, depth, &pH, &co2star, &dco2star, &pCO2, &dpco2, &CO3,
                     &Omega_c, &Omega_a, M.carb_shared);
    const double ac13_DIC_aq = -1.0512994e-4 * t_in + 1.011765;
    const double ac13_aq_POC = -0.017 * log10(dmin(dmax(co2star * 1000., 2.), 74.)) + 1.0034;
    PRE(MP_AC13B) = ac13_aq_POC / ac13_DIC_aq;
  }
  if (part == 1) return;
  // light geometry, tracer.F:381-390
  const double ai = M.aice[ij], hi = M.hice[ij], hs = M.hsno[ij];
  double rctheta = dmax(-1.5, dmin(1.5, M.tlat[ij] / M.radian - S.declin));
  rctheta = P->kw / sqrt(1. - (1. - sq(cos(rctheta))) / sq(1.33));
  double dayfrac = dmin(1., -tan(M.tlat[ij] / M.radian) * tan(S.declin));
  dayfrac = dmax(1e-12, acos(dmax(-1., dayfrac)) / M.pi);
  double swr = P->tap * M.dnswr[ij] * 1e-3 * (1. + ai * (exp(-P->ki * (hi + hs)) - 1.));
  // attenuation by the phytoplankton above, the same running product as mobi.F:735-740
  double phin = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
  // (device: the product of the exponentials as the exponential of the sum -- one exp instead of k per cell, within
  // the rounding of the MOBI source tolerance; the host build keeps the running product and stays bit-identical)
  double att = 0.0;
  for (int m = 1; m <= k; ++m) {
    att = att + P->kc * phin;
    phin = TNC(m, MI::phyt) * P->dzt[m - 1] + TNC(m, MI::diaz) * P->dzt[m - 1];
  }
  swr = swr * exp(-att);
#else
  for (int m = 1; m <= k; ++m) {
    swr = swr * exp(-P->kc * phin);
    phin = TNC(m, MI::phyt) * P->dzt[m - 1] + TNC(m, MI::diaz) * P->dzt[m - 1];
  }
#endif
  const double gl = swr * exp(P->ztt[k - 1] * rctheta);
  // oxygen saturation -> apparent oxygen utilisation, tracer.F:456-476
  double aou_in;
  {
    const double f1 = log((298.15 - t_in) / (273.15 + t_in));
    const double f2 = f1 * f1, f3 = f2 * f1, f4 = f3 * f1, f5 = f4 * f1;
    double o2sat = exp(2.00907 + 3.22014 * f1 + 4.05010 * f2 + 4.94457 * f3 - 2.56847E-1 * f4 + 3.88767 * f5 +
                       s_in * (-6.24523e-3 - 7.37614e-3 * f1 - 1.03410e-2 * f2 - 8.17083E-3 * f3) - 4.88682E-7 * s_in * s_in);
    o2sat = o2sat / 22391.6 * 1000.0 * 1000.;
    aou_in = o2sat - o2_in;
  }
  const double bct = UV_POWP(P->bbio, P->cbio * t_in);
  const double bctz = (0.5 * (tanh(o2_in - 8.) + 1)) * UV_POWP(P->bbio, P->cbio * t_in);
  PRE(MP_BCT) = bct;
  PRE(MP_BCTZ) = bctz;
  PRE(MP_NUD) = P->nud0 * (0.6 + 0.4 * tanh(0.22 * dmax(o2_in, 0.)));
  PRE(MP_O2F) = tanh(dmax(o2_in, 0.));                 // o2flag, mobi.F:2313
  PRE(MP_AOUT) = UV_DIVC(UV_POWP(dmax(aou_in, 40.), 0.8), 66.);    // the AOU term of the ligand concentration, mobi.F:2316
  /* light-limited growth, Evans & Parslow, with iron-dependent Chl:C, mobi.F:1984-2061 */
  const double biophyt = TNC(k, MI::phyt), biodiaz = TNC(k, MI::diaz), biodfe = TNC(k, MI::dfe), dzt = P->dzt[k - 1];
  const double p1 = dmin(biophyt, P->pmax);
  const double p2 = dmax(0.0, biophyt - P->pmax);
  const double kfevar = (P->kfemin * p1 + P->kfemax * p2) / (p1 + p2);
  const double deffe = biodfe / (kfevar + biodfe);
  const double thetamax = P->thetamaxlo + (P->thetamaxhi - P->thetamaxlo) * deffe;
  const double alpha_O = P->alphamin + (P->alphamax - P->alphamin) * deffe;
  const double gl_O = gl * thetamax * alpha_O;
  const double deffe_D = biodfe / (P->kfe_D + biodfe);
  const double thetamax_D = P->thetamaxlo + (P->thetamaxhi - P->thetamaxlo) * deffe_D;
  const double alpha_D = P->alphamin + (P->alphamax - P->alphamin) * deffe_D;
  const double gl_D = gl * thetamax_D * alpha_D;
  const double kirr = -P->kw - P->kc * (biophyt + biodiaz);
  const double f1 = exp(kirr * dzt);
  const double jmax = P->abio_P * bct * deffe;
  const double gd = jmax * dayfrac;
  double u1 = dmax(gl_O / gd, 1.e-6);
  double u2 = u1 * f1;
  double phi1 = log(u1 + sqrt(1. + u1 * u1)) - (sqrt(1. + u1 * u1) - 1.) / u1;
  double phi2 = log(u2 + sqrt(1. + u2 * u2)) - (sqrt(1. + u2 * u2) - 1.) / u2;
  PRE(MP_AVEJ) = gd * (phi1 - phi2) / (-kirr * dzt);
  const double jmax_D = dmax(0., P->abio_P * (bct - P->dbct_D) * deffe_D) * P->jdiar;
  const double gd_D = dmax(1.e-14, jmax_D * dayfrac);
  u1 = dmax(gl_D / gd_D, 1.e-6);
  u2 = u1 * f1;
  phi1 = log(u1 + sqrt(1. + u1 * u1)) - (sqrt(1. + u1 * u1) - 1.) / u1;
  phi2 = log(u2 + sqrt(1. + u2 * u2)) - (sqrt(1. + u2 * u2) - 1.) / u2;
  PRE(MP_AVEJD) = gd_D * (phi1 - phi2) / (-kirr * dzt);
}

// ---------------------------------------------------------------------------
// the vertical sequence of mobi_driver (mobi.F:519-1482) for the column (i,j): per level the
// nbio sub-steps of mobi_src, the raw source terms (mobi.F:1149-1205), and the hand-down of
// the export (mobi.F:1124-1134, 1268-1287).
// ---------------------------------------------------------------------------
template <class Team>
UVIC_DEV void mobi_column_body(Team &T, const uvic_ctx &c, const mobi_dev &M, int i, int j, bool live, int kmax) {
  UV_MOBI_LOCALS(c, M);
  const int kmx = live ? c.kmt[ij] : 0;
  double *src = const_cast<double *>(c.src);
  double sink = 0.0;
  // every thread of a team walks the same loops (workgroup barriers inside); threads with nothing to
  // store (land, levels below the sea floor) write into `sink`.  In a team each wave stores the
  // tendencies of the pools it owns and a share of the hand-over planes.
#define MINE(r) (!Team::team || Team::role == (r))
  double expo = 0.0, impo, expo_phos = 0.0, impo_phos, prca = 0.0;
  double rn15impo, rn15expo = 0.0, rc13impo, rc13expo = 0.0, prca13 = 0.0, expofe = 0.0, impofe;
  double snpzd[MI::count], bioin[MI::count];
  for (int k = 1; k <= kmax; ++k) {
    const bool store = live && k <= kmx;
#define OUT(ptr) (*(store ? (ptr) : &sink))
    rn15impo = rn15expo;
    const double dic_in = TM(k, P->idic);
    const double sgb = M.sg_bathy[ij + NS * (k - 1)];
    const double dztk = P->dzt[k - 1];
    rc13impo = rc13expo * P->dztr[k - 1];
    impo = expo * P->dztr[k - 1];
    impo_phos = expo_phos * P->dztr[k - 1];
    impofe = expofe * P->dztr[k - 1];
    _Pragma("unroll") for (int m = 1; m <= MI::count; ++m) bioin[m - 1] = TM(k, P->tracer_of_mobi[m - 1]);
    src_out_t so;
    mobi_src(T, P, S, bioin, PRE(MP_BCT), impo, impo_phos, P->wd[k - 1], PRE(MP_NUD), P->nudop0, P->nudon0, snpzd, PRE(MP_BCTZ),
             rn15impo, rc13impo, PRE(MP_AC13B), impofe, PRE(MP_O2F), PRE(MP_AOUT), PRE(MP_AVEJ), PRE(MP_AVEJD), &so);
    expo = so.expo; expo_phos = so.expo_phos; rn15expo = so.rn15expo; rc13expo = so.rc13expo; expofe = so.expofe;
    // tendency = (new pool - clamped input) / twodt; mobi_src left the clamped inputs in bioin, and a wave keeps
    // only those of the pools it owns
    _Pragma("unroll") for (int m = 0; m < MI::count; ++m)
      if (MINE(MOBI_OWNER[m])) snpzd[m] = (snpzd[m] - bioin[m]) * S.rdtts;
    expofe = expofe * S.rnbio;
    expo = expo * S.rnbio;
    expo_phos = expo_phos * S.rnbio;
    rn15expo = rn15expo * S.rnbio;
    rc13expo = rc13expo * S.rnbio;
    const double rcalpro_k = so.calpro * S.rnbio;
    // raw source terms into the slot of each MOBI tracer (mobi.F:1149-1205) and what the cell pass needs
    _Pragma("unroll") for (int m = 1; m <= MI::count; ++m)
      if (MINE(MOBI_OWNER[m - 1])) OUT(src + X3(i, k, j) + (size_t)(P->slot_of_mobi[m - 1] - 1) * N3) = snpzd[m - 1];
    if (MINE(1)) { OUT(&AUX(MA_EXPO)) = expo; OUT(&AUX(MA_EXPOP)) = expo_phos; OUT(&AUX(MA_CALPRO)) = rcalpro_k; OUT(&AUX(MA_NFIX)) = so.nfix; }
    if (MINE(2)) OUT(&AUX(MA_RN15)) = rn15expo;
    if (MINE(3)) OUT(&AUX(MA_RC13)) = rc13expo;
    // calcite production of the column, mobi.F:1228-1266 (bioin is clamped now)
    const double dprca = rcalpro_k * 1e-3;
    const double r13min = UV_TRCMIN * UV_RC13STD / (1 + UV_RC13STD);
    double rtdic13 = dmax(bioin[MI::dic13 - 1], r13min) / dmax(dic_in, UV_TRCMIN);
    rtdic13 = dmin(rtdic13, 2. * UV_RC13STD / (1 + UV_RC13STD));
    rtdic13 = dmax(rtdic13, 0.5 * UV_RC13STD / (1 + UV_RC13STD));
    if (k <= kmx) {   // levels below the sea floor are walked only for the team's barriers
      prca = prca + dprca * dztk;
      prca13 = prca13 + dprca * dztk * rtdic13;
    }
    // bottom remineralisation takes its share (mobi.F:1124-1134); the rest is the import of
    // the next level, mobi.F:1268-1287
    rc13expo = rc13expo - sgb * rc13expo;
    expo = expo - sgb * expo;
    expo_phos = expo_phos - sgb * expo_phos;
    expo = expo * dztk;
    expo_phos = expo_phos * dztk;
    rc13expo = rc13expo * dztk;
    expofe = expofe * dztk;
#undef OUT
  }
  if (live && MINE(3)) {
    M.col[ij] = prca;
    M.col[NS + ij] = prca13;
  }
#undef MINE
}

// one thread per column, all roles in the thread
UVIC_DEV void mobi_column_kernel(const uvic_ctx &c, const mobi_dev &M, int i, int j) {
  NoTeam T;
  mobi_column_body(T, c, M, i, j, true, c.kmt[(size_t)(i - 1) + (size_t)c.imt * (j - 1)]);
}

// ---------------------------------------------------------------------------
// the part of mobi_driver after mobi_src that stays inside the cell (mobi.F:1033-1134,
// 1228-1266, 1302-1436) and the caller's iron inputs and 14C (tracer.F:538-545, 853-867).
// Land and the levels below the sea floor get zero sources (static local src, tracer.F:121).
// ---------------------------------------------------------------------------
UVIC_DEV void mobi_post_cell(const uvic_ctx &c, const mobi_dev &M, int i, int k, int j) {
  UV_MOBI_LOCALS(c, M);
  double *src = const_cast<double *>(c.src);
#define SRC(s) src[X3(i, k, j) + (size_t)((s)-1) * N3]
#define SN(m) SRC(P->slot_of_mobi[(m)-1])
  const int kmx = c.kmt[ij];
  if (k > kmx) {
    for (int s = 1; s <= c.nsrc; ++s) SRC(s) = 0.0;
    return;
  }
  const double twodt = c.c2dtts, redctn = P->redctn;
  const double o2_in = TM(k, P->io2) * 1000.;
  const double dic_in = TM(k, P->idic);
  const double sgb = M.sg_bathy[ij + NS * (k - 1)];
  const double dztk = P->dzt[k - 1];
  const double expo = AUX(MA_EXPO), expo_phos = AUX(MA_EXPOP), rn15expo = AUX(MA_RN15), rc13expo = AUX(MA_RC13);
  const double rcalpro_k = AUX(MA_CALPRO), nfix_k = AUX(MA_NFIX);
  const double prca = M.col[ij], prca13 = M.col[NS + ij];
  const double r15min = UV_TRCMIN * UV_RN15STD / (1 + UV_RN15STD);
  // benthic denitrification on the sub-grid bathymetry, mobi.F:1033-1085
  const double tn_no3 = TNC(k, MI::no3), tn_din15 = TNC(k, MI::din15);
  const double no3flag = flag01(tn_no3 - UV_TRCMIN);
  const double din15flag = flag01(tn_din15 - UV_TRCMIN);
  const double lno3 = 0.5 * tanh(tn_no3 * 10 - 5.0);
  double sg_bdeni =
      (0.06 + 0.19 * UV_POWP(0.99, dmax(o2_in, UV_TRCMIN) - dmax(tn_no3, UV_TRCMIN))) * dmax(expo * sgb, UV_TRCMIN) * redctn * 1.e3;
  sg_bdeni = dmin(sg_bdeni, sgb * expo);
  sg_bdeni = dmax(sg_bdeni, 0.);
  sg_bdeni = sg_bdeni * (0.5 + lno3) * no3flag * din15flag;
  const double sn_no3 = SN(MI::no3) + sgb * expo - sg_bdeni;
  double rno3 = dmax(tn_din15, r15min) / dmax(tn_no3 - tn_din15, r15min);
  rno3 = dmin(rno3, 2. * UV_RN15STD);
  rno3 = dmax(rno3, UV_RN15STD / 2.);
  const double eps_bdeni = P->eps_bdeni0 * exp(-2.5e-6 * (P->zt[k - 1]));
  const double bbdeni = rno3 - eps_bdeni * rno3 / 1000.;
  const double sn_din15 = SN(MI::din15) + rn15expo * sgb * expo - bbdeni / (1 + bbdeni) * sg_bdeni;
  // sedimentary iron release, mobi.F:1086-1123
  const double coxdepth = dmin(dmax(P->zt[k - 1], 50000.), 150000.);
  const double oblinc = -1.26e-6 * coxdepth + 0.203;
  const double obexpc = -6.e-7 * coxdepth + 1.14;
  const double nburial = (oblinc * UV_POWP(expo * sgb * dztk / 100 * 86400. * 365. * redctn * 1000., obexpc)) /   /* base >= 0, exponent > 0: exp(y log 0) = 0 as pow */
                         (86400. * 365. * dztk / 100 * redctn * 1000.);
  const double coxsed = expo * sgb - nburial;
  const double fesedmax = 85.;
  const double fesed = fesedmax * tanh(coxsed * redctn * 1000 * dztk / 100 * 86400. / o2_in) / (dztk / 100 * 86400 * 1000);
  double fe = SN(MI::dfe) + fesed;
  // bottom remineralisation, mobi.F:1124-1134
  SN(MI::po4) = SN(MI::po4) + sgb * expo_phos;
  const double sn_dic = SN(MI::dic) + sgb * expo * redctn;
  const double sn_dic13 = SN(MI::dic13) + rc13expo * sgb * redctn;
  // DIC / alkalinity / 13C bookkeeping, mobi.F:1228-1266
  const double dic_sms = sn_dic;
  const double dprca = rcalpro_k * 1e-3;
  double s_dic = sn_dic - dprca;
  const double r13min = UV_TRCMIN * UV_RC13STD / (1 + UV_RC13STD);
  double rtdic13 = dmax(TNC(k, MI::dic13), r13min) / dmax(dic_in, UV_TRCMIN);
  rtdic13 = dmin(rtdic13, 2. * UV_RC13STD / (1 + UV_RC13STD));
  rtdic13 = dmax(rtdic13, 0.5 * UV_RC13STD / (1 + UV_RC13STD));
  double s_dic13 = sn_dic13 - rtdic13 * dprca;
  double a = -sn_dic * P->redntc * 1.e-3 - 2. * dprca;
  // water-column denitrification and oxygen, the reference's second pass (mobi.F:1302-1365)
  const double fo2 = tanh(0.22 * dmax(o2_in, 0.));
  const double so2 = dic_sms * P->redotc + nfix_k * S.rnbio * 1.25e-3;
  const double lno3b = 0.5 * tanh(tn_no3 - 2.5);
  double wcdeni = 800. * no3flag * so2 * (1.0 - fo2) * (0.5 + lno3b) * din15flag;
  wcdeni = dmax(wcdeni, 0.);
  SRC(Q->no3) = sn_no3 - wcdeni;
  double uno3 = wcdeni * twodt / tn_no3;
  uno3 = dmin(uno3, 0.999);
  uno3 = dmax(uno3, UV_TRCMIN);
  const double bwcdeni = rayleigh(rno3, P->eps_wcdeni, uno3);
  SRC(Q->din15) = sn_din15 - (bwcdeni / (1 + bwcdeni)) * wcdeni;
  a = a + wcdeni * 1.e-3;
  a = a + sg_bdeni * 1.e-3;
  a = a - nfix_k * S.rnbio * 1.e-3;
  SRC(Q->o2) = -so2 * fo2;
  // calcite dissolution profile (mobi.F:1373-1436), iron inputs (tracer.F:538-545), 14C (tracer.F:853-867)
  const double rc = (k < kmx) ? P->rcak[k - 1] : P->rcab[k - 1];
  s_dic = s_dic + prca * rc;
  s_dic13 = s_dic13 + prca13 * rc;
  a = a + 2. * prca * rc;
  SRC(Q->dic) = s_dic;
  SRC(Q->dic13) = s_dic13;
  SRC(Q->alk) = a;
  if (k == 1) fe = fe + M.fe_atmdep[ij + NS * (S.month - 1)] * 1000 / (P->dzt[0] / 100.);
  fe = fe + M.fe_hydr[ij + NS * (k - 1)];
  SRC(Q->dfe) = fe;
  if (Q->c14 > 0) SRC(Q->c14) = s_dic * UV_RC14STD - 3.836e-12 * TM(k, P->ic14);
#undef SRC
#undef SN
}
#undef TM
#undef TNC
#undef PRE
#undef AUX
#undef UV_MOBI_LOCALS

}  // namespace uvic
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang fp contract(off)
#endif

#if defined(__HIPCC__)
#include <string>
// upload parameters and forcing; called by uvic_gpu_set_mobi
struct mobi_store {
  void *params;
  double *f[8];
  void *opts;
  double *work, *work_side[2];  // work planes, one set per stream: the two side streams work ahead in turn
};
static inline int mobi_bind(int imt, int jmt, int km, const uvic_mobi_params *hp, const uvic_mobi_forcing *hf, mobi_dev *dev,
                            mobi_store *st, hipStream_t stream, std::string &err, const uvic_mobi_options *ho = nullptr) {
  if (hp->km != km) { err = "uvic_gpu_set_mobi: params.km differs from the model's km"; return 2; }
  if (!ho) {
    const int32_t want[MI::count] = {MI::po4, MI::phyt, MI::phyt_phos, MI::zoop, MI::detr, MI::detr_phos, MI::dic, MI::dic13,
                                     MI::phytc13, MI::zoopc13, MI::detrc13, MI::doc13, MI::diazc13, MI::dop, MI::no3, MI::don,
                                     MI::diaz, MI::din15, MI::don15, MI::phytn15, MI::zoopn15, MI::detrn15, MI::diazn15,
                                     MI::dfe, MI::detrfe};
    const int32_t *got = &hp->im.po4;
    if (hp->ntnpzd != MI::count) { err = "uvic_gpu_set_mobi: this build implements option set C (ntnpzd = 25)"; return 2; }
    for (int q = 0; q < MI::count; ++q)
      if (got[q] != want[q]) { err = "uvic_gpu_set_mobi: MOBI tracer order differs from option set C"; return 2; }
  }
  if (hp->ntnpzd > UV_MOBI_MAXT || km > 64) { err = "uvic_gpu_set_mobi: ntnpzd > 40 or km > 64 not supported"; return 2; }
  const size_t NS = (size_t)imt * jmt;
  const size_t sz[8] = {NS, NS, NS, NS, NS, NS * km, NS * 12, NS * km};
  const double *hsrc[8] = {hf->tlat, hf->dnswr, hf->aice, hf->hice, hf->hsno, hf->sg_bathy, hf->fe_atmdep, hf->fe_hydr};
  hipError_t e;
  if (!st->params) {
    if ((e = hipMalloc(&st->params, sizeof(uvic_mobi_params))) != hipSuccess) { err = hipGetErrorString(e); return 1; }
    for (int q = 0; q < 8; ++q)
      if ((e = hipMalloc((void **)&st->f[q], sz[q] * 8)) != hipSuccess) { err = hipGetErrorString(e); return 1; }
    const size_t wb = mobi_work_doubles(imt, jmt, km) * 8;
    if ((e = hipMalloc((void **)&st->work, wb)) != hipSuccess) { err = hipGetErrorString(e); return 1; }
    if ((e = hipMemsetAsync(st->work, 0, wb, stream)) != hipSuccess) { err = hipGetErrorString(e); return 1; }
    for (int q = 0; q < 2; ++q) {
      if ((e = hipMalloc((void **)&st->work_side[q], wb)) != hipSuccess) { err = hipGetErrorString(e); return 1; }
      if ((e = hipMemsetAsync(st->work_side[q], 0, wb, stream)) != hipSuccess) { err = hipGetErrorString(e); return 1; }
    }
  }
  if ((e = hipMemcpyAsync(st->params, hp, sizeof(uvic_mobi_params), hipMemcpyHostToDevice, stream)) != hipSuccess) { err = hipGetErrorString(e); return 1; }
  for (int q = 0; q < 8; ++q) {
    if (!hsrc[q]) { err = "uvic_gpu_set_mobi: null forcing array"; return 2; }
    if ((e = hipMemcpyAsync(st->f[q], hsrc[q], sz[q] * 8, hipMemcpyHostToDevice, stream)) != hipSuccess) { err = hipGetErrorString(e); return 1; }
  }
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) { err = hipGetErrorString(e); return 1; }
  dev->P = (const uvic_mobi_params *)st->params;
  dev->O = nullptr;
  if (ho) {   // another option set: the general column kernel reads the flags and the extra parameters from here
    if (!st->opts && (e = hipMalloc(&st->opts, sizeof(uvic_mobi_options))) != hipSuccess) { err = hipGetErrorString(e); return 1; }
    if ((e = hipMemcpy(st->opts, ho, sizeof(uvic_mobi_options), hipMemcpyHostToDevice)) != hipSuccess) { err = hipGetErrorString(e); return 1; }
    dev->O = (const uvic_mobi_options *)st->opts;
  }
  dev->tlat = st->f[0]; dev->dnswr = st->f[1]; dev->aice = st->f[2]; dev->hice = st->f[3]; dev->hsno = st->f[4];
  dev->sg_bathy = st->f[5]; dev->fe_atmdep = st->f[6]; dev->fe_hydr = st->f[7];
  mobi_set_work(dev, st->work, imt, jmt, km);
  dev->pi = hf->pi; dev->radian = hf->radian; dev->relyr = hf->relyr; dev->co2ccn = hf->co2ccn;
  dev->carb_shared = 1;

  return 0;
}
#endif
#endif
