/* oracle/uvic_oracle.c -- CPU restatement of the UVic 2.9 tracer transport.
 *
 * TEST INFRASTRUCTURE ONLY (see uvic_oracle.h).  Plain C, scalar, one core,
 * compiled with -O2 -ffp-contract=off so that every expression is evaluated
 * in the reference's order without fused multiply-adds.  Each function cites
 * the reference lines it follows (paths relative to /root/reference;
 * "u09/" = updates/09/source/).  Indices are 1-based through the macros below
 * so that every loop bound can be read against the Fortran.
 */
#include "uvic_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Threads for the courtesy N-core timing of bench.py (cpu_baseline_ncore): the tracers of orc_tracer_transport, the rows of
 * convct2 and the rows of orc_mobi_sources are independent, so they are shared out; every value is computed by the same
 * expressions as with one thread (the default, which the parity tests use). */
int orc_threads = 1;
void orc_set_threads(int n) { orc_threads = n > 1 ? n : 1; }

#define EPSLN 1.0e-20 /* source/common/pconst.h:20 */
#define P5 0.5
#define C0 0.0
#define C1 1.0
#define C2 2.0

/* -------- index helpers (1-based i,j,k; vertical faces 0..km) -------------- */
#define DIMS                                                                   \
  const int imt = c->imt, jmt = c->jmt, km = c->km;                            \
  const size_t N3 = (size_t)imt * km * jmt, NF = (size_t)imt * (km + 1) * jmt; \
  (void)N3; (void)NF; (void)jmt
#define X3(i, k, j) ((size_t)((i)-1) + (size_t)imt * ((size_t)((k)-1) + (size_t)km * ((j)-1)))
#define XF(i, k, j) ((size_t)((i)-1) + (size_t)imt * ((size_t)(k) + (size_t)(km + 1) * ((j)-1)))
#define X2(i, j) ((size_t)((i)-1) + (size_t)imt * ((j)-1))
#define XFIS(i, j, k) ((size_t)((i)-1) + (size_t)imt * ((size_t)((j)-1) + (size_t)jmt * ((k)-1)))

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* source/common/util.F:789-814 (O_cyclic) */
void orc_setbcx(double *a, int imt, int n) {
  for (int k = 0; k < n; ++k) {
    a[(size_t)k * imt] = a[(size_t)k * imt + imt - 2];
    a[(size_t)k * imt + imt - 1] = a[(size_t)k * imt + 1];
  }
}

/* source/mom/dens.h:13-22 */
static inline double dens(const double *cc, int km, double tq, double sq, int k) {
#define CK(m) cc[(k - 1) + (size_t)km * ((m)-1)]
  return (CK(1) + (CK(4) + CK(7) * sq) * sq + (CK(3) + CK(8) * sq + CK(6) * tq) * tq) * tq +
         (CK(2) + (CK(5) + CK(9) * sq) * sq) * sq;
}
static inline double drodt(const double *cc, int km, double tq, double sq, int k) {
  return CK(1) + (CK(4) + CK(7) * sq) * sq + (2.0 * CK(3) + 2.0 * CK(8) * sq + 3.0 * CK(6) * tq) * tq;
}
static inline double drods(const double *cc, int km, double tq, double sq, int k) {
  return (CK(4) + 2.0 * CK(7) * sq + CK(8) * tq) * tq + CK(2) + (2.0 * CK(5) + 3.0 * CK(9) * sq) * sq;
#undef CK
}

/* ---- statement functions of u09/common/isopyc.h:121-136 ------------------- */
#define ALPHA(i, k, j) c->alphai[X3(i, k, j)]
#define BETA(i, k, j) c->betai[X3(i, k, j)]
#define DDXT(i, k, j, n) c->ddxt[X3(i, k, j) + (size_t)((n)-1) * N3]
#define DDYT(i, k, j, n) c->ddyt[X3(i, k, j) + (size_t)((n)-1) * N3]
#define DDZT(i, k, j, n) c->ddzt[XF(i, k, j) + (size_t)((n)-1) * NF]
#define AI4(a, i, k, j, p, q) c->a[X3(i, k, j) + (size_t)((p) + 2 * (q)) * N3]
#define TMASK(i, k, j) c->tmask[X3(i, k, j)]
#define drodxe(i, k, j, ip) (ALPHA((i) + (ip), k, j) * DDXT(i, k, j, 1) + BETA((i) + (ip), k, j) * DDXT(i, k, j, 2))
#define drodze(i, k, j, ip, kr)                                        \
  (ALPHA((i) + (ip), k, j) * DDZT((i) + (ip), (k)-1 + (kr), j, 1) +    \
   BETA((i) + (ip), k, j) * DDZT((i) + (ip), (k)-1 + (kr), j, 2))
#define drodyn(i, k, j, jq) (ALPHA(i, k, (j) + (jq)) * DDYT(i, k, j, 1) + BETA(i, k, (j) + (jq)) * DDYT(i, k, j, 2))
#define drodzn(i, k, j, jq, kr)                                        \
  (ALPHA(i, k, (j) + (jq)) * DDZT(i, (k)-1 + (kr), (j) + (jq), 1) +    \
   BETA(i, k, (j) + (jq)) * DDZT(i, (k)-1 + (kr), (j) + (jq), 2))
#define drodxb(i, k, j, ip, kr)                                        \
  (ALPHA(i, (k) + (kr), j) * DDXT((i)-1 + (ip), (k) + (kr), j, 1) +    \
   BETA(i, (k) + (kr), j) * DDXT((i)-1 + (ip), (k) + (kr), j, 2))
#define drodyb(i, k, j, jq, kr)                                        \
  (ALPHA(i, (k) + (kr), j) * DDYT(i, (k) + (kr), (j)-1 + (jq), 1) +    \
   BETA(i, (k) + (kr), j) * DDYT(i, (k) + (kr), (j)-1 + (jq), 2))
#define drodzb(i, k, j, kr) (ALPHA(i, (k) + (kr), j) * DDZT(i, k, j, 1) + BETA(i, (k) + (kr), j) * DDZT(i, k, j, 2))

/* ===========================================================================
 * isopyc: u09/mom/isopyc.F:363-921 (elements, ai_east, ai_north, ai_bottom)
 * and :1140-1575 (isopyc_adv), called as isopyc(joff=0, js=1, je=jmt, is=2,
 * ie=imt-1) from source/mom/mom.F:340.
 * =========================================================================== */
static void elements(orc_ctx *c) { /* isopyc.F:363-464 */
  DIMS;
  const double *t1 = c->t_taum1, *t2 = c->t_taum1 + N3; /* T, S at tau-1 */
  const int is = 2, ie = imt - 1, js = 1, je = jmt;
  for (int j = js; j <= je; ++j) {
    for (int k = 1; k <= km; ++k)
      for (int i = is; i <= ie; ++i) {
        double tprime = t1[X3(i, k, j)] - c->to[k - 1];
        double sprime = t2[X3(i, k, j)] - c->so[k - 1];
        ALPHA(i, k, j) = drodt(c->c, km, tprime, sprime, k);
        BETA(i, k, j) = drods(c->c, km, tprime, sprime, k);
      }
    orc_setbcx(&ALPHA(1, 1, j), imt, km);
    orc_setbcx(&BETA(1, 1, j), imt, km);
  }
  for (int j = js; j <= je; ++j)
    for (int n = 1; n <= 2; ++n) {
      const double *t = c->t_taum1 + (size_t)(n - 1) * N3;
      for (int k = 1; k <= km; ++k) {
        int kp1 = imin(k + 1, km);
        for (int i = is; i <= ie; ++i)
          DDZT(i, k, j, n) = TMASK(i, kp1, j) * c->dzwr[k] * (t[X3(i, k, j)] - t[X3(i, kp1, j)]);
      }
      for (int i = is; i <= ie; ++i) DDZT(i, 0, j, n) = C0;
      orc_setbcx(&DDZT(1, 0, j, n), imt, km + 1);
    }
  for (int j = imax(js - 1, 2); j <= je - 1; ++j)
    for (int n = 1; n <= 2; ++n) {
      const double *t = c->t_taum1 + (size_t)(n - 1) * N3;
      for (int k = 1; k <= km; ++k)
        for (int i = is; i <= ie; ++i)
          DDXT(i, k, j, n) = TMASK(i, k, j) * TMASK(i + 1, k, j) * c->cstr[j - 1] * c->dxur[i - 1] *
                             (t[X3(i + 1, k, j)] - t[X3(i, k, j)]);
      orc_setbcx(&DDXT(1, 1, j, n), imt, km);
    }
  for (int j = imax(js - 1, 1); j <= je - 1; ++j)
    for (int n = 1; n <= 2; ++n) {
      const double *t = c->t_taum1 + (size_t)(n - 1) * N3;
      for (int k = 1; k <= km; ++k)
        for (int i = is; i <= ie; ++i)
          DDYT(i, k, j, n) =
              TMASK(i, k, j) * TMASK(i, k, j + 1) * c->dyur[j - 1] * (t[X3(i, k, j + 1)] - t[X3(i, k, j)]);
      orc_setbcx(&DDYT(1, 1, j, n), imt, km);
    }
}

static void ai_east(orc_ctx *c, int js, int je) { /* isopyc.F:559-665 */
  DIMS;
  for (int j = js; j <= je; ++j) {
    for (int k = 1; k <= km; ++k) {
      double sc = C1 / (c->slmxr * c->dtxsqr[k - 1]);
      double dzt4r = P5 * c->dzt2r[k - 1];
      for (int i = 2; i <= imt - 1; ++i) {
        double Ai0 = .5 * (c->fisop[XFIS(i, j, k)] + c->fisop[XFIS(i + 1, j, k)]) * c->ahisop +
                     c->addisop[X3(i, k, j)];
        double sumz = C0;
        for (int kr = 0; kr <= 1; ++kr)
          for (int ip = 0; ip <= 1; ++ip) {
            double sxe = fabs(drodxe(i, k, j, ip) / (drodze(i, k, j, ip, kr) + EPSLN));
            double a;
            if (sxe > sc) {
              double r = sc / (sxe + EPSLN);
              a = Ai0 * TMASK(i, k, j) * TMASK(i + 1, k, j) * (r * r);
            } else {
              a = Ai0 * TMASK(i, k, j) * TMASK(i + 1, k, j);
            }
            AI4(Ai_ez, i, k, j, ip, kr) = a;
            sumz = sumz + c->dzw[k - 1 + kr] * a;
          }
        c->K11[X3(i, k, j)] = dzt4r * sumz;
      }
    }
    for (int q = 0; q < 4; ++q) orc_setbcx(&c->Ai_ez[X3(1, 1, j) + (size_t)q * N3], imt, km);
    orc_setbcx(&c->K11[X3(1, 1, j)], imt, km);
  }
}

static void ai_north(orc_ctx *c, int js, int je) { /* isopyc.F:667-771 */
  DIMS;
  for (int j = js; j <= je; ++j) {
    for (int k = 1; k <= km; ++k) {
      double sc = C1 / (c->slmxr * c->dtxsqr[k - 1]);
      double dzt4r = P5 * c->dzt2r[k - 1];
      for (int i = 2; i <= imt - 1; ++i) {
        double Ai0 = P5 * (c->fisop[XFIS(i, j, k)] + c->fisop[XFIS(i, j + 1, k)]) * c->ahisop;
        double sumz = C0;
        for (int kr = 0; kr <= 1; ++kr)
          for (int jq = 0; jq <= 1; ++jq) {
            double syn = fabs(drodyn(i, k, j, jq) / (drodzn(i, k, j, jq, kr) + EPSLN));
            double a;
            if (syn > sc) {
              double r = sc / (syn + EPSLN);
              a = Ai0 * TMASK(i, k, j) * TMASK(i, k, j + 1) * (r * r);
            } else {
              a = Ai0 * TMASK(i, k, j) * TMASK(i, k, j + 1);
            }
            AI4(Ai_nz, i, k, j, jq, kr) = a;
            sumz = sumz + c->dzw[k - 1 + kr] * a;
          }
        c->K22[X3(i, k, j)] = dzt4r * sumz;
      }
    }
    for (int q = 0; q < 4; ++q) orc_setbcx(&c->Ai_nz[X3(1, 1, j) + (size_t)q * N3], imt, km);
    orc_setbcx(&c->K22[X3(1, 1, j)], imt, km);
  }
}

static void ai_bottom(orc_ctx *c, int js, int je) { /* isopyc.F:773-921 */
  DIMS;
  for (int j = js; j <= je; ++j) {
    for (int k = 1; k <= km - 1; ++k) {
      double sc = C1 / (c->slmxr * c->dtxsqr[k - 1]);
      for (int i = 2; i <= imt - 1; ++i) {
        double Ai0 = P5 * (c->fisop[XFIS(i, j, k + 1)] + c->fisop[XFIS(i, j, k)]) * c->ahisop;
        double sumx = C0;
        for (int ip = 0; ip <= 1; ++ip)
          for (int kr = 0; kr <= 1; ++kr) {
            double sxb = fabs(drodxb(i, k, j, ip, kr) / (drodzb(i, k, j, kr) + EPSLN));
            double a;
            if (sxb > sc) {
              double r = sc / (sxb + EPSLN);
              a = Ai0 * TMASK(i, k + 1, j) * (r * r);
            } else {
              a = Ai0 * TMASK(i, k + 1, j);
            }
            AI4(Ai_bx, i, k, j, ip, kr) = a;
            sumx = sumx + c->dxu[i - 1 + ip - 1] * a * (sxb * sxb);
          }
        double sumy = C0;
        for (int jq = 0; jq <= 1; ++jq) {
          double facty = c->csu[j - 1 + jq - 1] * c->dyu[j - 1 + jq - 1];
          for (int kr = 0; kr <= 1; ++kr) {
            double syb = fabs(drodyb(i, k, j, jq, kr) / (drodzb(i, k, j, kr) + EPSLN));
            double a;
            if (syb > sc) {
              double r = sc / (syb + EPSLN);
              a = Ai0 * TMASK(i, k + 1, j) * (r * r);
            } else {
              a = Ai0 * TMASK(i, k + 1, j);
            }
            AI4(Ai_by, i, k, j, jq, kr) = a;
            sumy = sumy + facty * a * (syb * syb);
          }
        }
        c->K33[X3(i, k, j)] = c->dxt4r[i - 1] * sumx + c->dyt4r[j - 1] * c->cstr[j - 1] * sumy;
      }
    }
    for (int q = 0; q < 4; ++q) {
      orc_setbcx(&c->Ai_bx[X3(1, 1, j) + (size_t)q * N3], imt, km);
      orc_setbcx(&c->Ai_by[X3(1, 1, j) + (size_t)q * N3], imt, km);
    }
    orc_setbcx(&c->K33[X3(1, 1, j)], imt, km);
  }
}

static void isopyc_adv(orc_ctx *c, int js, int je) { /* isopyc.F:1140-1575 */
  DIMS;
  /* full-grid face density gradients (isopyc.h: drodxte ... drodzbn); zero as
     the reference's COMMON storage is at start-up */
  double *w = (double *)calloc(8 * N3, sizeof(double));
  double *xte = w, *xbe = w + N3, *ytn = w + 2 * N3, *ybn = w + 3 * N3;
  double *zte = w + 4 * N3, *zbe = w + 5 * N3, *ztn = w + 6 * N3, *zbn = w + 7 * N3;
  for (int j = js; j <= je; ++j)
    for (int i = 1; i <= imt - 1; ++i) {
      double at = P5 * (ALPHA(i, 1, j) + ALPHA(i, 1, j + 1));
      double bt = P5 * (BETA(i, 1, j) + BETA(i, 1, j + 1));
      ytn[X3(i, 1, j)] = at * DDYT(i, 1, j, 1) + bt * DDYT(i, 1, j, 2);
      ztn[X3(i, 1, j)] = at * (DDZT(i, 1, j, 1) + DDZT(i, 1, j + 1, 1)) * P5 +
                         bt * (DDZT(i, 1, j, 2) + DDZT(i, 1, j + 1, 2)) * P5;
      at = P5 * (ALPHA(i, 1, j) + ALPHA(i + 1, 1, j));
      bt = P5 * (BETA(i, 1, j) + BETA(i + 1, 1, j));
      xte[X3(i, 1, j)] = at * DDXT(i, 1, j, 1) + bt * DDXT(i, 1, j, 2);
      zte[X3(i, 1, j)] = at * (DDZT(i, 1, j, 1) + DDZT(i + 1, 1, j, 1)) * P5 +
                         bt * (DDZT(i, 1, j, 2) + DDZT(i + 1, 1, j, 2)) * P5;
      for (int k = 1; k <= km; ++k) {
        int km1 = imax(k - 1, 1), kp1 = imin(k + 1, km);
        double ab = (ALPHA(i, k, j) + ALPHA(i, k, j + 1) + ALPHA(i, kp1, j) + ALPHA(i, kp1, j + 1)) * 0.25;
        double bb = (BETA(i, k, j) + BETA(i, k, j + 1) + BETA(i, kp1, j) + BETA(i, kp1, j + 1)) * 0.25;
        ybn[X3(i, k, j)] = ab * P5 * (DDYT(i, k, j, 1) + DDYT(i, kp1, j, 1)) +
                           bb * P5 * (DDYT(i, k, j, 2) + DDYT(i, kp1, j, 2));
        zbn[X3(i, k, j)] = ab * P5 * (DDZT(i, k, j, 1) + DDZT(i, k, j + 1, 1)) +
                           bb * P5 * (DDZT(i, k, j, 2) + DDZT(i, k, j + 1, 2));
        if (k > 1) {
          ytn[X3(i, k, j)] = ybn[X3(i, km1, j)];
          ztn[X3(i, k, j)] = zbn[X3(i, km1, j)];
        }
        ab = (ALPHA(i, k, j) + ALPHA(i + 1, k, j) + ALPHA(i, kp1, j) + ALPHA(i + 1, kp1, j)) * 0.25;
        bb = (BETA(i, k, j) + BETA(i + 1, k, j) + BETA(i, kp1, j) + BETA(i + 1, kp1, j)) * 0.25;
        xbe[X3(i, k, j)] = ab * P5 * (DDXT(i, k, j, 1) + DDXT(i, kp1, j, 1)) +
                           bb * P5 * (DDXT(i, k, j, 2) + DDXT(i, kp1, j, 2));
        zbe[X3(i, k, j)] = ab * P5 * (DDZT(i, k, j, 1) + DDZT(i + 1, k, j, 1)) +
                           bb * P5 * (DDZT(i, k, j, 2) + DDZT(i + 1, k, j, 2));
        if (k > 1) {
          xte[X3(i, k, j)] = xbe[X3(i, km1, j)];
          zte[X3(i, k, j)] = zbe[X3(i, km1, j)];
        }
      }
    }
  /* meridional component, isopyc.F:1381-1430 */
  for (int j = js; j <= je; ++j)
    for (int k = 1; k <= km; ++k) {
      double sc = C1 / (c->slmxr * c->dtxsqr[k - 1]);
      int kp1 = imin(k + 1, km);
      double top_bc = (k == 1) ? C0 : C1, bot_bc = (k == km) ? C0 : C1;
      for (int i = 1; i <= imt; ++i) {
        double Ath0 = c->athkdf * P5 * (c->fisop[XFIS(i, j, k)] + c->fisop[XFIS(i, j + 1, k)]);
        double stn = -ytn[X3(i, k, j)] / (ztn[X3(i, k, j)] + 0.125 * EPSLN);
        double sbn = -ybn[X3(i, k, j)] / (zbn[X3(i, k, j)] + 0.125 * EPSLN);
        double absstn = fabs(stn), abssbn = fabs(sbn), ath_t, ath_b;
        if (absstn > sc) {
          double r = sc / (absstn + EPSLN);
          ath_t = Ath0 * TMASK(i, k, j) * TMASK(i, k, j + 1) * (r * r);
        } else
          ath_t = Ath0 * TMASK(i, k, j) * TMASK(i, k, j + 1);
        if (abssbn > sc) {
          double r = sc / (abssbn + EPSLN);
          ath_b = Ath0 * TMASK(i, kp1, j) * TMASK(i, kp1, j + 1) * (r * r);
        } else
          ath_b = Ath0 * TMASK(i, kp1, j) * TMASK(i, kp1, j + 1);
        c->adv_vntiso[X3(i, k, j)] = -(ath_t * stn * top_bc - ath_b * sbn * bot_bc) * c->dztr[k - 1] * c->csu[j - 1];
      }
    }
  /* zonal component, isopyc.F:1436-1482 */
  int jstrt = imax(js, 2);
  for (int j = jstrt; j <= je; ++j) {
    for (int k = 1; k <= km; ++k) {
      double sc = C1 / (c->slmxr * c->dtxsqr[k - 1]);
      int kp1 = imin(k + 1, km);
      double top_bc = (k == 1) ? C0 : C1, bot_bc = (k == km) ? C0 : C1;
      for (int i = 1; i <= imt - 1; ++i) {
        double Ath0 = c->athkdf * P5 * (c->fisop[XFIS(i, j, k)] + c->fisop[XFIS(i + 1, j, k)]);
        double ste = -xte[X3(i, k, j)] / (zte[X3(i, k, j)] + 0.125 * EPSLN);
        double sbe = -xbe[X3(i, k, j)] / (zbe[X3(i, k, j)] + 0.125 * EPSLN);
        double absste = fabs(ste), abssbe = fabs(sbe), ath_t, ath_b;
        if (absste > sc) {
          double r = sc / (absste + EPSLN);
          ath_t = Ath0 * TMASK(i, k, j) * TMASK(i + 1, k, j) * (r * r);
        } else
          ath_t = Ath0 * TMASK(i, k, j) * TMASK(i + 1, k, j);
        if (abssbe > sc) {
          double r = sc / (abssbe + EPSLN);
          ath_b = Ath0 * TMASK(i, kp1, j) * TMASK(i + 1, kp1, j) * (r * r);
        } else
          ath_b = Ath0 * TMASK(i, kp1, j) * TMASK(i + 1, kp1, j);
        c->adv_vetiso[X3(i, k, j)] = -(ath_t * ste * top_bc - ath_b * sbe * bot_bc) * c->dztr[k - 1];
      }
    }
    orc_setbcx(&c->adv_vetiso[X3(1, 1, j)], imt, km);
  }
  /* vertical component by continuity + prefix sum in k, isopyc.F:1496-1526 */
  for (int j = jstrt; j <= je; ++j) {
    for (int i = 1; i <= imt; ++i) c->adv_vbtiso[XF(i, 0, j)] = C0;
    for (int k = 1; k <= km - 1; ++k)
      for (int i = 2; i <= imt; ++i)
        c->adv_vbtiso[XF(i, k, j)] =
            c->dzt[k - 1] * c->cstr[j - 1] *
            ((c->adv_vetiso[X3(i, k, j)] - c->adv_vetiso[X3(i - 1, k, j)]) * c->dxtr[i - 1] +
             (c->adv_vntiso[X3(i, k, j)] - c->adv_vntiso[X3(i, k, j - 1)]) * c->dytr[j - 1]);
    for (int k = 1; k <= km - 1; ++k)
      for (int i = 2; i <= imt; ++i)
        c->adv_vbtiso[XF(i, k, j)] = c->adv_vbtiso[XF(i, k, j)] + c->adv_vbtiso[XF(i, k - 1, j)];
    for (int i = 2; i <= imt; ++i) c->adv_vbtiso[XF(i, c->kmt[X2(i, j)], j)] = C0;
    orc_setbcx(&c->adv_vbtiso[XF(1, 0, j)], imt, km + 1);
  }
  free(w);
}

void orc_isopyc(orc_ctx *c) { /* isopyc.F:466-557 with js=1, je=jmt */
  const int jmt = c->jmt;
  elements(c);
  ai_east(c, 2, jmt - 1);
  ai_north(c, 1, jmt - 1);
  ai_bottom(c, 2, jmt - 1);
  isopyc_adv(c, 1, jmt - 1);
}

/* ===========================================================================
 * adv_flux, FCT branch: u09/mom/tracer_adv_flx.F:381-1028, called as
 * adv_flux(joff=0, js=2, je=jmt-1, is=2, ie=imt-1, n) from tracer.F:918.
 * =========================================================================== */
void orc_adv_flux(orc_ctx *c, int n) {
  DIMS;
  const double *tm = c->t_taum1 + (size_t)(n - 1) * N3; /* t(:,:,:,n,taum1) */
  const double *tt = c->t_tau + (size_t)(n - 1) * N3;   /* t(:,:,:,n,tau)   */
  const int istrt = 2, iend = imt - 1, istrtm1 = 1, iendp1 = imt;
  const int js = 2, je = jmt - 1;
  const int jstrt = js - 1, jend = imin(je, jmt - 1); /* first (only) window, :456-461 */
  const double c2dtts = c->c2dtts;
  double *adv_fe = c->adv_fe, *adv_fn = c->adv_fn, *adv_fb = c->adv_fb;
  /* per-tracer persistent scratch of mw.h:389-393; row 1 zeroed (:467-481) */
  double *anti_fe = (double *)calloc(N3, sizeof(double));
  double *anti_fn = (double *)calloc(N3, sizeof(double));
  double *anti_fb = (double *)calloc(NF, sizeof(double));
  double *RpY = (double *)calloc(N3, sizeof(double));
  double *RmY = (double *)calloc(N3, sizeof(double));
  double *t_lo = (double *)calloc((size_t)imt * km, sizeof(double));
  double *Rpl = (double *)calloc((size_t)imt * km, sizeof(double));
  double *Rmn = (double *)calloc((size_t)imt * km, sizeof(double));
  double *twodt = (double *)malloc(sizeof(double) * km);
  double *dcf = (double *)malloc(sizeof(double) * (imt + 2)), *Trmin = (double *)malloc(sizeof(double) * (imt + 2));
  double *Trmax = (double *)malloc(sizeof(double) * (imt + 2)), *Cpos = (double *)malloc(sizeof(double) * (imt + 2));
  double *Cneg = (double *)malloc(sizeof(double) * (imt + 2)), *flxlft = (double *)malloc(sizeof(double) * (imt + 2));
  double *flxrgt = (double *)malloc(sizeof(double) * (imt + 2));
#define TLO(i, k) t_lo[((i)-1) + (size_t)imt * ((k)-1)]
#define RPL(i, k) Rpl[((i)-1) + (size_t)imt * ((k)-1)]
#define RMN(i, k) Rmn[((i)-1) + (size_t)imt * ((k)-1)]
#define TMASKI(i, k, j) (C1 - TMASK(i, k, j))
#define CSTDXT2R(i, j) (c->cstr[(j)-1] * c->dxtr[(i)-1] * P5) /* tracer.F:243 */

  /* low order (upstream) fluxes, :500-547 */
  int jlast = imin(jend + 1, jmt - 1);
  for (int j = js - 1; j <= jlast; ++j)
    for (int k = 1; k <= km; ++k)
      for (int i = istrt; i <= iend; ++i) {
        double totadv = c->adv_vnt[X3(i, k, j)] + c->adv_vntiso[X3(i, k, j)];
        adv_fn[X3(i, k, j)] = totadv * (tm[X3(i, k, j)] + tm[X3(i, k, j + 1)]) +
                              fabs(totadv) * (tm[X3(i, k, j)] - tm[X3(i, k, j + 1)]);
      }
  for (int j = js; j <= jlast; ++j) {
    for (int k = 1; k <= km; ++k)
      for (int i = istrtm1; i <= iend; ++i) {
        double totadv = c->adv_vet[X3(i, k, j)] + c->adv_vetiso[X3(i, k, j)];
        adv_fe[X3(i, k, j)] = totadv * (tm[X3(i, k, j)] + tm[X3(i + 1, k, j)]) +
                              fabs(totadv) * (tm[X3(i, k, j)] - tm[X3(i + 1, k, j)]);
      }
    for (int k = 1; k <= km - 1; ++k)
      for (int i = istrt; i <= iend; ++i) {
        double totadv = c->adv_vbt[XF(i, k, j)] + c->adv_vbtiso[XF(i, k, j)];
        adv_fb[XF(i, k, j)] = totadv * (tm[X3(i, k + 1, j)] + tm[X3(i, k, j)]) +
                              fabs(totadv) * (tm[X3(i, k + 1, j)] - tm[X3(i, k, j)]);
      }
    for (int i = istrt; i <= iend; ++i) {
      adv_fb[XF(i, 0, j)] = c->adv_vbt[XF(i, 0, j)] * C2 * tm[X3(i, 1, j)];
      adv_fb[XF(i, km, j)] = C0;
    }
  }

  /* main j loop, :553-1003: everything is produced at row j+1 */
  for (int j = jstrt; j <= jend; ++j) {
    const int jrow = j + 1;
    const int jp2 = imin(j + 2, jmt);
    const int jp1 = imin(j + 1, jmt - 1);
    /* low order solution, :563-579 */
    for (int k = 1; k <= km; ++k) {
      twodt[k - 1] = c2dtts * c->dtxcel[k - 1];
      for (int i = istrt; i <= iend; ++i) {
        double advx = (adv_fe[X3(i, k, jp1)] - adv_fe[X3(i - 1, k, jp1)]) * CSTDXT2R(i, jp1);
        double advy = (adv_fn[X3(i, k, jp1)] - adv_fn[X3(i, k, jp1 - 1)]) * c->cstdyt2r[jrow - 1];
        double advz = (adv_fb[XF(i, k - 1, jp1)] - adv_fb[XF(i, k, jp1)]) * c->dzt2r[k - 1];
        TLO(i, k) = (tm[X3(i, k, j + 1)] - twodt[k - 1] * (advx + advy + advz) * TMASK(i, k, j + 1));
      }
    }
    orc_setbcx(t_lo, imt, km);
    /* raw antidiffusive fluxes, :586-619 */
    for (int k = 1; k <= km; ++k) {
      for (int i = istrtm1; i <= iend; ++i) {
        double totadv = c->adv_vet[X3(i, k, jp1)] + c->adv_vetiso[X3(i, k, jp1)];
        anti_fe[X3(i, k, j + 1)] = totadv * (tt[X3(i, k, j + 1)] + tt[X3(i + 1, k, j + 1)]) - adv_fe[X3(i, k, jp1)];
      }
      for (int i = istrt; i <= iend; ++i) {
        double totadv = c->adv_vnt[X3(i, k, jp1)] + c->adv_vntiso[X3(i, k, jp1)];
        anti_fn[X3(i, k, j + 1)] = totadv * (tt[X3(i, k, j + 1)] + tt[X3(i, k, jp2)]) - adv_fn[X3(i, k, jp1)];
      }
    }
    for (int k = 1; k <= km - 1; ++k)
      for (int i = istrt; i <= iend; ++i) {
        double totadv = c->adv_vbt[XF(i, k, jp1)] + c->adv_vbtiso[XF(i, k, jp1)];
        anti_fb[XF(i, k, j + 1)] = totadv * (tt[X3(i, k, j + 1)] + tt[X3(i, k + 1, j + 1)]) -
                                   adv_fb[XF(i, k, jp1)] * TMASK(i, k, j + 1);
      }
    for (int i = istrt; i <= iend; ++i) {
      anti_fb[XF(i, 0, j + 1)] = c->adv_vbt[XF(i, 0, j + 1)] * C2 * tm[X3(i, 1, j + 1)];
      anti_fb[XF(i, km, j + 1)] = C0;
    }
    /* delimit x, :638-711 */
    for (int k = 1; k <= km; ++k) {
      for (int i = istrt; i <= iendp1; ++i) Trmax[i] = P5 * (tt[X3(i - 1, k, j + 1)] + tt[X3(i, k, j + 1)]);
      for (int i = istrt; i <= iend; ++i) {
        double fxa = TMASK(i - 1, k, j + 1) * Trmax[i] + TMASKI(i - 1, k, j + 1) * TLO(i, k);
        double fxb = TMASK(i + 1, k, j + 1) * Trmax[i + 1] + TMASKI(i + 1, k, j + 1) * TLO(i, k);
        Trmax[i] = dmax(dmax(fxa, fxb), TLO(i, k));
        Trmin[i] = dmin(dmin(fxa, fxb), TLO(i, k));
        dcf[i] = CSTDXT2R(i, j + 1);
        flxlft[i] = anti_fe[X3(i - 1, k, j + 1)];
        flxrgt[i] = anti_fe[X3(i, k, j + 1)];
      }
      for (int i = istrt; i <= iend; ++i) {
        double Pplus = c2dtts * dcf[i] * (dmax(C0, flxlft[i]) - dmin(C0, flxrgt[i]));
        double Pminus = c2dtts * dcf[i] * (dmax(C0, flxrgt[i]) - dmin(C0, flxlft[i]));
        double Qplus = Trmax[i] - TLO(i, k);
        double Qminus = TLO(i, k) - Trmin[i];
        RPL(i, k) = dmin(1., TMASK(i, k, j + 1) * Qplus / (Pplus + EPSLN));
        RMN(i, k) = dmin(1., TMASK(i, k, j + 1) * Qminus / (Pminus + EPSLN));
      }
      orc_setbcx(Rpl, imt, km);
      orc_setbcx(Rmn, imt, km);
      for (int i = istrt; i <= iendp1; ++i) {
        Cpos[i - 1] = dmin(RPL(i, k), RMN(i - 1, k));
        Cneg[i - 1] = dmin(RPL(i - 1, k), RMN(i, k));
      }
      for (int i = istrtm1; i <= iend; ++i) {
        double f = anti_fe[X3(i, k, j + 1)];
        anti_fe[X3(i, k, j + 1)] = P5 * ((Cpos[i] + Cneg[i]) * f + (Cpos[i] - Cneg[i]) * fabs(f));
      }
    }
    /* delimit y, :717-783 */
    for (int k = 1; k <= km; ++k) {
      for (int i = istrt; i <= iend; ++i) {
        double fxa = P5 * TMASK(i, k, j) * (tt[X3(i, k, j)] + tt[X3(i, k, j + 1)]) + TMASKI(i, k, j) * TLO(i, k);
        double fxb =
            P5 * TMASK(i, k, jp2) * (tt[X3(i, k, j + 1)] + tt[X3(i, k, jp2)]) + TMASKI(i, k, jp2) * TLO(i, k);
        Trmax[i] = dmax(dmax(fxa, fxb), TLO(i, k));
        Trmin[i] = dmin(dmin(fxa, fxb), TLO(i, k));
        dcf[i] = c->cstdyt2r[jrow - 1];
        flxlft[i] = anti_fn[X3(i, k, j)];
        flxrgt[i] = anti_fn[X3(i, k, j + 1)];
      }
      for (int i = istrt; i <= iend; ++i) {
        double Pplus = c2dtts * dcf[i] * (dmax(C0, flxlft[i]) - dmin(C0, flxrgt[i]));
        double Pminus = c2dtts * dcf[i] * (dmax(C0, flxrgt[i]) - dmin(C0, flxlft[i]));
        double Qplus = Trmax[i] - TLO(i, k);
        double Qminus = TLO(i, k) - Trmin[i];
        RpY[X3(i, k, j + 1)] = dmin(1., TMASK(i, k, j + 1) * Qplus / (Pplus + EPSLN));
        RmY[X3(i, k, j + 1)] = dmin(1., TMASK(i, k, j + 1) * Qminus / (Pminus + EPSLN));
      }
      for (int i = istrt; i <= iend; ++i) {
        Cpos[i] = dmin(RpY[X3(i, k, j + 1)], RmY[X3(i, k, j)]);
        Cneg[i] = dmin(RpY[X3(i, k, j)], RmY[X3(i, k, j + 1)]);
      }
      for (int i = istrt; i <= iend; ++i) {
        double f = anti_fn[X3(i, k, j)];
        anti_fn[X3(i, k, j)] = P5 * ((Cpos[i] + Cneg[i]) * f + (Cpos[i] - Cneg[i]) * fabs(f));
      }
    }
    /* delimit z, :789-887 */
    for (int k = 1; k <= km; ++k) {
      for (int i = istrt; i <= iend; ++i) {
        double fxa, fxb;
        dcf[i] = c->dzt2r[k - 1];
        flxlft[i] = anti_fb[XF(i, k, j + 1)];
        flxrgt[i] = anti_fb[XF(i, k - 1, j + 1)];
        if (k > 1)
          fxa = P5 * TMASK(i, k - 1, j + 1) * (tt[X3(i, k - 1, j + 1)] + tt[X3(i, k, j + 1)]) +
                TMASKI(i, k - 1, j + 1) * TLO(i, k);
        else
          fxa = TLO(i, k);
        if (k < km)
          fxb = P5 * TMASK(i, k + 1, j + 1) * (tt[X3(i, k, j + 1)] + tt[X3(i, k + 1, j + 1)]) +
                TMASKI(i, k + 1, j + 1) * TLO(i, k);
        else
          fxb = TLO(i, k);
        Trmax[i] = dmax(dmax(fxa, fxb), TLO(i, k));
        Trmin[i] = dmin(dmin(fxa, fxb), TLO(i, k));
      }
      for (int i = istrt; i <= iend; ++i) {
        double Pplus = c2dtts * dcf[i] * (dmax(C0, flxlft[i]) - dmin(C0, flxrgt[i]));
        double Pminus = c2dtts * dcf[i] * (dmax(C0, flxrgt[i]) - dmin(C0, flxlft[i]));
        double Qplus = Trmax[i] - TLO(i, k);
        double Qminus = TLO(i, k) - Trmin[i];
        RPL(i, k) = dmin(1., TMASK(i, k, j + 1) * Qplus / (Pplus + EPSLN));
        RMN(i, k) = dmin(1., TMASK(i, k, j + 1) * Qminus / (Pminus + EPSLN));
      }
    }
    for (int k = 1; k <= km - 1; ++k) {
      for (int i = istrt; i <= iend; ++i) {
        Cneg[i] = dmin(RPL(i, k + 1), RMN(i, k));
        Cpos[i] = dmin(RPL(i, k), RMN(i, k + 1));
      }
      for (int i = istrt; i <= iend; ++i) {
        double f = anti_fb[XF(i, k, j + 1)];
        anti_fb[XF(i, k, j + 1)] = P5 * ((Cpos[i] + Cneg[i]) * f + (Cpos[i] - Cneg[i]) * fabs(f));
      }
    }
    for (int i = istrt; i <= iend; ++i) {
      anti_fb[XF(i, 0, j + 1)] = C0;
      anti_fb[XF(i, km, j + 1)] = C0;
    }
    /* add back low order, :989-999 */
    for (int k = 1; k <= km; ++k) {
      for (int i = istrtm1; i <= iend; ++i) anti_fe[X3(i, k, j + 1)] = anti_fe[X3(i, k, j + 1)] + adv_fe[X3(i, k, jp1)];
      for (int i = istrt; i <= iend; ++i) {
        anti_fn[X3(i, k, j)] = (anti_fn[X3(i, k, j)] + adv_fn[X3(i, k, j)]) * TMASK(i, k, j);
        anti_fb[XF(i, k, j + 1)] = (anti_fb[XF(i, k, j + 1)] + adv_fb[XF(i, k, jp1)]) * TMASK(i, k, j + 1);
      }
    }
  }
  /* copy out, :1008-1028 */
  for (int j = js - 1; j <= jend; ++j)
    for (int k = 1; k <= km; ++k)
      for (int i = istrt; i <= iend; ++i) adv_fn[X3(i, k, j)] = anti_fn[X3(i, k, j)];
  for (int j = js; j <= jend; ++j) {
    for (int k = 1; k <= km; ++k)
      for (int i = istrtm1; i <= iend; ++i) adv_fe[X3(i, k, j)] = anti_fe[X3(i, k, j)];
    for (int k = 1; k <= km - 1; ++k)
      for (int i = istrt; i <= iend; ++i) adv_fb[XF(i, k, j)] = anti_fb[XF(i, k, j)];
  }
  free(anti_fe); free(anti_fn); free(anti_fb); free(RpY); free(RmY); free(t_lo); free(Rpl); free(Rmn);
  free(twodt); free(dcf); free(Trmin); free(Trmax); free(Cpos); free(Cneg); free(flxlft); free(flxrgt);
}

/* tracer.F:925-1032: background diffusive fluxes */
void orc_diff_flux(orc_ctx *c, int n) {
  DIMS;
  const double *tm = c->t_taum1 + (size_t)(n - 1) * N3;
  const int js = 2, je = jmt - 1, istrt = 2, iend = imt - 1;
  for (int j = js; j <= je; ++j)
    for (int k = 1; k <= km; ++k)
      for (int i = istrt - 1; i <= iend; ++i) {
        double ah_cstdxur = c->diff_cet * c->cstr[j - 1] * c->dxur[i - 1]; /* tracer.F:246 */
        c->diff_fe[X3(i, k, j)] = ah_cstdxur * (tm[X3(i + 1, k, j)] - tm[X3(i, k, j)]);
      }
  for (int j = js - 1; j <= je; ++j)
    for (int k = 1; k <= km; ++k)
      for (int i = istrt; i <= iend; ++i)
        c->diff_fn[X3(i, k, j)] = c->diff_cnt * c->csu_dyur[j - 1] * (tm[X3(i, k, j + 1)] - tm[X3(i, k, j)]);
  for (int j = js; j <= je; ++j)
    for (int k = 1; k <= km - 1; ++k)
      for (int i = istrt; i <= iend; ++i)
        c->diff_fb[XF(i, k, j)] = c->diff_cbt[X3(i, k, j)] * c->dzwr[k] * (tm[X3(i, k, j)] - tm[X3(i, k + 1, j)]);
}

/* isopyc.F:923-1137 */
void orc_isoflux(orc_ctx *c, int n) {
  DIMS;
  const double *tm = c->t_taum1 + (size_t)(n - 1) * N3;
  const int js = 2, je = jmt - 1;
  for (int j = js; j <= je; ++j) {
    for (int k = 1; k <= km; ++k) {
      double dzt4r = P5 * c->dzt2r[k - 1];
      for (int i = 2; i <= imt - 1; ++i) {
        double sumz = C0;
        for (int kr = 0; kr <= 1; ++kr) {
          int km1kr = imax(k - 1 + kr, 1), kpkr = imin(k + kr, km);
          for (int ip = 0; ip <= 1; ++ip)
            sumz = sumz - AI4(Ai_ez, i, k, j, ip, kr) * (tm[X3(i + ip, km1kr, j)] - tm[X3(i + ip, kpkr, j)]) *
                              drodxe(i, k, j, ip) / (drodze(i, k, j, ip, kr) + EPSLN);
        }
        double flux_x = dzt4r * sumz;
        double cstdxur = c->cstr[j - 1] * c->dxur[i - 1];
        c->diff_fe[X3(i, k, j)] =
            c->diff_fe[X3(i, k, j)] + c->K11[X3(i, k, j)] * cstdxur * (tm[X3(i + 1, k, j)] - tm[X3(i, k, j)]) + flux_x;
      }
    }
    orc_setbcx(&c->diff_fe[X3(1, 1, j)], imt, km);
  }
  for (int j = js - 1; j <= je; ++j) {
    for (int k = 1; k <= km; ++k) {
      double csu_dzt4r = c->csu[j - 1] * P5 * c->dzt2r[k - 1];
      for (int i = 2; i <= imt - 1; ++i) {
        double sumz = C0;
        for (int kr = 0; kr <= 1; ++kr) {
          int km1kr = imax(k - 1 + kr, 1), kpkr = imin(k + kr, km);
          for (int jq = 0; jq <= 1; ++jq)
            sumz = sumz - AI4(Ai_nz, i, k, j, jq, kr) * (tm[X3(i, km1kr, j + jq)] - tm[X3(i, kpkr, j + jq)]) *
                              drodyn(i, k, j, jq) / (drodzn(i, k, j, jq, kr) + EPSLN);
        }
        double flux_y = csu_dzt4r * sumz;
        c->diff_fn[X3(i, k, j)] = c->diff_fn[X3(i, k, j)] +
                                  c->K22[X3(i, k, j)] * c->csu_dyur[j - 1] * (tm[X3(i, k, j + 1)] - tm[X3(i, k, j)]) +
                                  flux_y;
      }
    }
    orc_setbcx(&c->diff_fn[X3(1, 1, j)], imt, km);
  }
  for (int j = js; j <= je; ++j) {
    for (int k = 1; k <= km - 1; ++k)
      for (int i = 2; i <= imt - 1; ++i) {
        double sumx = C0;
        for (int ip = 0; ip <= 1; ++ip)
          for (int kr = 0; kr <= 1; ++kr)
            sumx = sumx - AI4(Ai_bx, i, k, j, ip, kr) * c->cstr[j - 1] *
                              (tm[X3(i + ip, k + kr, j)] - tm[X3(i - 1 + ip, k + kr, j)]) * drodxb(i, k, j, ip, kr) /
                              (drodzb(i, k, j, kr) + EPSLN);
        double sumy = C0;
        for (int jq = 0; jq <= 1; ++jq)
          for (int kr = 0; kr <= 1; ++kr)
            sumy = sumy - AI4(Ai_by, i, k, j, jq, kr) * c->csu[j - 1 + jq - 1] *
                              (tm[X3(i, k + kr, j + jq)] - tm[X3(i, k + kr, j - 1 + jq)]) * drodyb(i, k, j, jq, kr) /
                              (drodzb(i, k, j, kr) + EPSLN);
        c->diff_fbiso[XF(i, k, j)] = c->dxt4r[i - 1] * sumx + c->dyt4r[j - 1] * c->cstr[j - 1] * sumy;
      }
    for (int i = 2; i <= imt - 1; ++i) {
      c->diff_fbiso[XF(i, 0, j)] = C0;
      c->diff_fbiso[XF(i, km, j)] = C0;
    }
    orc_setbcx(&c->diff_fbiso[XF(1, 0, j)], imt, km + 1);
  }
}

/* tracer.F:1053-1130 with source/mom/fdift.h:25-88 */
void orc_explicit_update(orc_ctx *c, int n) {
  DIMS;
  const double *tm = c->t_taum1 + (size_t)(n - 1) * N3;
  const double *tt = c->t_tau + (size_t)(n - 1) * N3;
  double *tp = c->t_taup1 + (size_t)(n - 1) * N3;
  const double *stf = c->stf + (size_t)(n - 1) * imt * jmt, *btf = c->btf + (size_t)(n - 1) * imt * jmt;
  const int js = 2, je = jmt - 1, istrt = 2, iend = imt - 1;
  const double *source = NULL;
  if (c->src && c->itrc && c->itrc[n - 1] != 0) source = c->src + (size_t)(c->itrc[n - 1] - 1) * N3;
  for (int j = js; j <= je; ++j)
    for (int i = istrt; i <= iend; ++i) {
      int kb = c->kmt[X2(i, j)];
      c->diff_fb[XF(i, 0, j)] = stf[X2(i, j)];
      c->diff_fb[XF(i, kb, j)] = btf[X2(i, j)];
      c->adv_fb[XF(i, 0, j)] = c->adv_vbt[XF(i, 0, j)] * (tt[X3(i, 1, j)] + tt[X3(i, 1, j)]);
      c->adv_fb[XF(i, km, j)] = c->adv_vbt[XF(i, km, j)] * tt[X3(i, km, j)];
    }
  for (int j = js; j <= je; ++j)
    for (int k = 1; k <= km; ++k) {
      double twodt = c->c2dtts * c->dtxcel[k - 1];
      for (int i = istrt; i <= iend; ++i) {
        double cstdxtr = c->cstr[j - 1] * c->dxtr[i - 1];
        double cstdxt2r = c->cstr[j - 1] * c->dxtr[i - 1] * P5;
        double DIFF_Tx = (c->diff_fe[X3(i, k, j)] * TMASK(i + 1, k, j) - c->diff_fe[X3(i - 1, k, j)] * TMASK(i - 1, k, j)) *
                         cstdxtr;
        double DIFF_Ty = (c->diff_fn[X3(i, k, j)] * TMASK(i, k, j + 1) - c->diff_fn[X3(i, k, j - 1)] * TMASK(i, k, j - 1)) *
                         c->cstdytr[j - 1];
        double DIFF_Tz = (c->diff_fb[XF(i, k - 1, j)] - c->diff_fb[XF(i, k, j)]) * c->dztr[k - 1] * (C1 - c->aidif) +
                         (c->diff_fbiso[XF(i, k - 1, j)] - c->diff_fbiso[XF(i, k, j)]) * c->dztr[k - 1];
        double ADV_Tx = (c->adv_fe[X3(i, k, j)] - c->adv_fe[X3(i - 1, k, j)]) * cstdxt2r;
        double ADV_Ty = (c->adv_fn[X3(i, k, j)] - c->adv_fn[X3(i, k, j - 1)]) * c->cstdyt2r[j - 1];
        double ADV_Tz = (c->adv_fb[XF(i, k - 1, j)] - c->adv_fb[XF(i, k, j)]) * c->dzt2r[k - 1];
        double s = source ? source[X3(i, k, j)] : C0;
        tp[X3(i, k, j)] =
            tm[X3(i, k, j)] + twodt * (DIFF_Tx + DIFF_Ty + DIFF_Tz - ADV_Tx - ADV_Ty - ADV_Tz + s) * TMASK(i, k, j);
      }
    }
}

/* source/mom/invtri.F:1-115; dcb has all jmt rows here */
void orc_invtri(const orc_ctx *c, double *z, const double *topbc, const double *botbc, const double *dcb,
                const double *tdt, int is, int ie, int js, int je) {
  DIMS;
  const double eps = 1.e-30;
  double *a = (double *)malloc(sizeof(double) * km), *cc = (double *)malloc(sizeof(double) * (km + 1));
  double *b = (double *)malloc(sizeof(double) * km), *f = (double *)malloc(sizeof(double) * (km + 1));
  double *e = (double *)malloc(sizeof(double) * km);
  for (int j = js; j <= je; ++j)
    for (int i = is; i <= ie; ++i) {
      for (int k = 1; k <= km; ++k) {
        int km1 = imax(1, k - 1), kp1 = imin(k + 1, km);
        double factu = c->dztur[k - 1] * tdt[k - 1] * c->aidif;
        double factl = c->dztlr[k - 1] * tdt[k - 1] * c->aidif;
        a[k - 1] = -dcb[X3(i, km1, j)] * factu * TMASK(i, k, j);
        cc[k] = -dcb[X3(i, k, j)] * factl * TMASK(i, kp1, j);
        f[k] = z[X3(i, k, j)] * TMASK(i, k, j);
        b[k - 1] = C1 - a[k - 1] - cc[k];
      }
      a[0] = C0;
      cc[km] = C0;
      b[0] = C1 - a[0] - cc[1];
      b[km - 1] = C1 - a[km - 1] - cc[km];
      f[1] = z[X3(i, 1, j)] + topbc[X2(i, j)] * tdt[0] * c->dztr[0] * c->aidif * TMASK(i, 1, j);
      int k = imax(2, c->kmt[X2(i, j)]);
      f[k] = z[X3(i, k, j)] - botbc[X2(i, j)] * tdt[k - 1] * c->dztr[k - 1] * c->aidif * TMASK(i, k, j);
      double bet = TMASK(i, 1, j) / (b[0] + eps);
      z[X3(i, 1, j)] = f[1] * bet;
      for (k = 2; k <= km; ++k) {
        e[k - 1] = cc[k - 1] * bet;
        bet = TMASK(i, k, j) / (b[k - 1] - a[k - 1] * e[k - 1] + eps);
        z[X3(i, k, j)] = (f[k] - a[k - 1] * z[X3(i, k - 1, j)]) * bet;
      }
      for (k = km - 1; k >= 1; --k) z[X3(i, k, j)] = z[X3(i, k, j)] - e[k] * z[X3(i, k + 1, j)];
    }
  free(a); free(cc); free(b); free(f); free(e);
}

/* source/mom/convect.F:99-311 (O_fullconvect), diagnostics omitted */
void orc_convct2(const orc_ctx *c, double *ts, int is, int ie, int js, int je) {
  DIMS;
  const int nt = c->nt;
#define TS(i, k, j, n) ts[X3(i, k, j) + (size_t)((n)-1) * N3]
#define DENS(tq, sq, k) dens(c->c, km, tq, sq, k)
  const double *to = c->to, *so = c->so, *dz = c->dztxcl;
  for (int j = js; j <= je; ++j)
    for (int i = is; i <= ie; ++i) {
      int kbo = c->kmt[X2(i, j)];
      int kt = 1, kb = 2;
      while (kt < kbo) {
        double ru = DENS(TS(i, kt, j, 1) - to[kb - 1], TS(i, kt, j, 2) - so[kb - 1], kb);
        double rl = DENS(TS(i, kb, j, 1) - to[kb - 1], TS(i, kb, j, 2) - so[kb - 1], kb);
        if (ru > rl) {
          int chk_la = 1, chk_lb = 1;
          double zsm = dz[kt - 1] + dz[kb - 1];
          double tsm1 = TS(i, kt, j, 1) * dz[kt - 1] + TS(i, kb, j, 1) * dz[kb - 1];
          double tmx1 = tsm1 / zsm;
          double tsm2 = TS(i, kt, j, 2) * dz[kt - 1] + TS(i, kb, j, 2) * dz[kb - 1];
          double tmx2 = tsm2 / zsm;
          while (chk_lb || chk_la) {
            if (kb >= kbo) chk_lb = 0;
            while (chk_lb) {
              chk_lb = 0;
              int lb = kb + 1;
              ru = DENS(tmx1 - to[lb - 1], tmx2 - so[lb - 1], lb);
              rl = DENS(TS(i, lb, j, 1) - to[lb - 1], TS(i, lb, j, 2) - so[lb - 1], lb);
              if (ru > rl) {
                kb = lb;
                zsm = zsm + dz[kb - 1];
                tsm1 = tsm1 + TS(i, kb, j, 1) * dz[kb - 1];
                tmx1 = tsm1 / zsm;
                tsm2 = tsm2 + TS(i, kb, j, 2) * dz[kb - 1];
                tmx2 = tsm2 / zsm;
                chk_la = 1;
                if (kb < kbo) chk_lb = 1;
              }
            }
            chk_la = 1;
            if (kt <= 1) chk_la = 0;
            while (chk_la) {
              chk_la = 0;
              int la = kt - 1;
              ru = DENS(TS(i, la, j, 1) - to[kt - 1], TS(i, la, j, 2) - so[kt - 1], kt);
              rl = DENS(tmx1 - to[kt - 1], tmx2 - so[kt - 1], kt);
              if (ru > rl) {
                kt = la;
                zsm = zsm + dz[kt - 1];
                tsm1 = tsm1 + TS(i, kt, j, 1) * dz[kt - 1];
                tmx1 = tsm1 / zsm;
                tsm2 = tsm2 + TS(i, kt, j, 2) * dz[kt - 1];
                tmx2 = tsm2 / zsm;
                chk_lb = 1;
              }
            }
          }
          for (int k = kt; k <= kb; ++k) {
            TS(i, k, j, 1) = tmx1;
            TS(i, k, j, 2) = tmx2;
          }
          for (int n = 3; n <= nt; ++n) {
            double tsm3 = C0;
            for (int k = kt; k <= kb; ++k) tsm3 = tsm3 + TS(i, k, j, n) * dz[k - 1];
            double tmx3 = tsm3 / zsm;
            for (int k = kt; k <= kb; ++k) TS(i, k, j, n) = tmx3;
          }
          kt = kb + 1;
        } else {
          kt = kb;
        }
        kb = kt + 1;
      }
    }
}

/* tracer.F:902-1209: the per-tracer loop, convection and cyclic conditions */
void orc_tracer_transport(orc_ctx *c) {
  DIMS;
  double *twodt = (double *)malloc(sizeof(double) * km);
  for (int k = 1; k <= km; ++k) twodt[k - 1] = c->c2dtts * c->dtxcel[k - 1];
#pragma omp parallel if (orc_threads > 1) num_threads(orc_threads)
  {
    orc_ctx lc = *c; /* with several threads: the per-tracer work arrays are each thread's own */
    int mine = 0;
#ifdef _OPENMP
    mine = omp_in_parallel();
#endif
    if (mine) {
      lc.adv_fe = (double *)calloc(N3, sizeof(double)); lc.adv_fn = (double *)calloc(N3, sizeof(double));
      lc.adv_fb = (double *)calloc(NF, sizeof(double)); lc.diff_fe = (double *)calloc(N3, sizeof(double));
      lc.diff_fn = (double *)calloc(N3, sizeof(double)); lc.diff_fb = (double *)calloc(NF, sizeof(double));
      lc.diff_fbiso = (double *)calloc(NF, sizeof(double));
    }
#pragma omp for schedule(dynamic)
    for (int n = 1; n <= c->nt; ++n) {
      orc_adv_flux(&lc, n);
      orc_diff_flux(&lc, n);
      orc_isoflux(&lc, n);
      orc_explicit_update(&lc, n);
      double *tp = c->t_taup1 + (size_t)(n - 1) * N3;
      orc_invtri(&lc, tp, c->stf + (size_t)(n - 1) * imt * jmt, c->btf + (size_t)(n - 1) * imt * jmt, c->diff_cbt, twodt, 2,
                 imt - 1, 2, jmt - 1);
      for (int j = 2; j <= jmt - 1; ++j) orc_setbcx(&tp[X3(1, 1, j)], imt, km);
    }
    if (mine) {
      free(lc.adv_fe); free(lc.adv_fn); free(lc.adv_fb); free(lc.diff_fe); free(lc.diff_fn); free(lc.diff_fb); free(lc.diff_fbiso);
    }
  }
  if (orc_threads > 1) {
#pragma omp parallel for num_threads(orc_threads) schedule(dynamic)
    for (int j = 2; j <= jmt - 1; ++j) orc_convct2(c, c->t_taup1, 2, imt - 1, j, j);
  } else
    orc_convct2(c, c->t_taup1, 2, imt - 1, 2, jmt - 1);
  for (int n = 1; n <= c->nt; ++n)
    for (int j = 2; j <= jmt - 1; ++j) orc_setbcx(&c->t_taup1[(size_t)(n - 1) * N3 + X3(1, 1, j)], imt, km);
  free(twodt);
}
