// kernels_isopyc.hpp -- isopycnal mixing tensor and Gent-McWilliams velocities.
//
// Replaces, once per ocean step and from T,S(tau-1) only,
//   /root/reference/updates/09/source/mom/isopyc.F:363-464   (elements)
//   :559-665 (ai_east)  :667-771 (ai_north)  :773-921 (ai_bottom)
//   :1140-1575 (isopyc_adv)
// as called by `isopyc(joff=0, js=1, je=jmt, is=2, ie=imt-1)` from
// source/mom/mom.F:340.  HBM-bound streaming stencils in fp64; one thread per
// (i,k,j) cell, i fastest so that every load is coalesced along longitude.
// The reference's `setbcx` cyclic copies (source/common/util.F:789-814) are
// done by the threads that own columns 2 and imt-1 (they also store columns
// imt and 1), so no extra pass is needed.  Expression order follows the
// reference exactly (compiled with -ffp-contract=off): results are bit-identical
// to the oracle.
#ifndef UVIC_KERNELS_ISOPYC_HPP
#define UVIC_KERNELS_ISOPYC_HPP

#include "kenv.hpp"
#include "uvic_ctx.h"

namespace uvic {

#define UV_EPSLN 1.0e-20 /* source/common/pconst.h:20 */

#define UV_DIMS(c)                                                                         \
  const int imt = (c).imt, jmt = (c).jmt, km = (c).km;                                     \
  const size_t N3 = (size_t)imt * km * jmt, NF = (size_t)imt * (km + 1) * jmt;             \
  (void)N3; (void)NF; (void)jmt
#define X3(i, k, j) ((size_t)((i)-1) + (size_t)imt * ((size_t)((k)-1) + (size_t)km * ((j)-1)))
#define XF(i, k, j) ((size_t)((i)-1) + (size_t)imt * ((size_t)(k) + (size_t)(km + 1) * ((j)-1)))
#define X2(i, j) ((size_t)((i)-1) + (size_t)imt * ((j)-1))
#define XFIS(i, j, k) ((size_t)((i)-1) + (size_t)imt * ((size_t)((j)-1) + (size_t)jmt * ((k)-1)))

// store `v` at column i and at its cyclic image (util.F:803-811, O_cyclic)
#define UV_CYC_STORE(arr, IDX, i, v)              \
  do {                                            \
    const double _v = (v);                        \
    (arr)[IDX(i)] = _v;                           \
    if ((i) == 2) (arr)[IDX(imt)] = _v;           \
    if ((i) == imt - 1) (arr)[IDX(1)] = _v;       \
  } while (0)

// source/mom/dens.h:18-22
UVIC_DEV double eos_drodt(const double *cc, int km, double tq, double sq, int k) {
#define CK(m) cc[(k - 1) + (size_t)km * ((m)-1)]
  return CK(1) + (CK(4) + CK(7) * sq) * sq + (2.0 * CK(3) + 2.0 * CK(8) * sq + 3.0 * CK(6) * tq) * tq;
}
UVIC_DEV double eos_drods(const double *cc, int km, double tq, double sq, int k) {
  return (CK(4) + 2.0 * CK(7) * sq + CK(8) * tq) * tq + CK(2) + (2.0 * CK(5) + 3.0 * CK(9) * sq) * sq;
}
// source/mom/dens.h:13-15
UVIC_DEV double eos_dens(const double *cc, int km, double tq, double sq, int k) {
  return (CK(1) + (CK(4) + CK(7) * sq) * sq + (CK(3) + CK(8) * sq + CK(6) * tq) * tq) * tq +
         (CK(2) + (CK(5) + CK(9) * sq) * sq) * sq;
#undef CK
}

// statement functions of updates/09/source/common/isopyc.h:121-136
#define ALPHA(i, k, j) c.alphai[X3(i, k, j)]
#define BETA(i, k, j) c.betai[X3(i, k, j)]
#define DDXT(i, k, j, n) c.ddxt[X3(i, k, j) + (size_t)((n)-1) * N3]
#define DDYT(i, k, j, n) c.ddyt[X3(i, k, j) + (size_t)((n)-1) * N3]
#define DDZT(i, k, j, n) c.ddzt[XF(i, k, j) + (size_t)((n)-1) * NF]
#define TMASK(i, k, j) c.tmask[X3(i, k, j)]
#define drodxe(i, k, j, ip) (ALPHA((i) + (ip), k, j) * DDXT(i, k, j, 1) + BETA((i) + (ip), k, j) * DDXT(i, k, j, 2))
#define drodze(i, k, j, ip, kr) \
  (ALPHA((i) + (ip), k, j) * DDZT((i) + (ip), (k)-1 + (kr), j, 1) + BETA((i) + (ip), k, j) * DDZT((i) + (ip), (k)-1 + (kr), j, 2))
#define drodyn(i, k, j, jq) (ALPHA(i, k, (j) + (jq)) * DDYT(i, k, j, 1) + BETA(i, k, (j) + (jq)) * DDYT(i, k, j, 2))
#define drodzn(i, k, j, jq, kr) \
  (ALPHA(i, k, (j) + (jq)) * DDZT(i, (k)-1 + (kr), (j) + (jq), 1) + BETA(i, k, (j) + (jq)) * DDZT(i, (k)-1 + (kr), (j) + (jq), 2))
#define drodxb(i, k, j, ip, kr) \
  (ALPHA(i, (k) + (kr), j) * DDXT((i)-1 + (ip), (k) + (kr), j, 1) + BETA(i, (k) + (kr), j) * DDXT((i)-1 + (ip), (k) + (kr), j, 2))
#define drodyb(i, k, j, jq, kr) \
  (ALPHA(i, (k) + (kr), j) * DDYT(i, (k) + (kr), (j)-1 + (jq), 1) + BETA(i, (k) + (kr), j) * DDYT(i, (k) + (kr), (j)-1 + (jq), 2))
#define drodzb(i, k, j, kr) (ALPHA(i, (k) + (kr), j) * DDZT(i, k, j, 1) + BETA(i, (k) + (kr), j) * DDZT(i, k, j, 2))

// ---------------------------------------------------------------------------
// elements (isopyc.F:363-464): one thread per cell, i = 2..imt-1, all k, all j
// ---------------------------------------------------------------------------
UVIC_DEV void isopyc_elements_cell(const uvic_ctx &c, int i, int k, int j) {
  UV_DIMS(c);
  const double *t1 = c.t_taum1, *t2 = c.t_taum1 + N3;  // T and S at tau-1
  {
    const double tprime = t1[X3(i, k, j)] - c.to[k - 1];
    const double sprime = t2[X3(i, k, j)] - c.so[k - 1];
#define IDX(ii) X3(ii, k, j)
    UV_CYC_STORE(c.alphai, IDX, i, eos_drodt(c.c, km, tprime, sprime, k));
    UV_CYC_STORE(c.betai, IDX, i, eos_drods(c.c, km, tprime, sprime, k));
#undef IDX
  }
  const int kp1 = imin(k + 1, km);
  for (int n = 1; n <= 2; ++n) {
    const double *t = c.t_taum1 + (size_t)(n - 1) * N3;
    double *ddz = c.ddzt + (size_t)(n - 1) * NF;
    double *ddx = c.ddxt + (size_t)(n - 1) * N3;
    double *ddy = c.ddyt + (size_t)(n - 1) * N3;
#define IDXF(ii) XF(ii, k, j)
    UV_CYC_STORE(ddz, IDXF, i, TMASK(i, kp1, j) * c.dzwr[k] * (t[X3(i, k, j)] - t[X3(i, kp1, j)]));
#undef IDXF
    if (k == 1) {
#define IDXF(ii) XF(ii, 0, j)
      UV_CYC_STORE(ddz, IDXF, i, 0.0);
#undef IDXF
    }
#define IDX(ii) X3(ii, k, j)
    if (j >= 2 && j <= jmt - 1)
      UV_CYC_STORE(ddx, IDX, i,
                   TMASK(i, k, j) * TMASK(i + 1, k, j) * c.cstr[j - 1] * c.dxur[i - 1] * (t[X3(i + 1, k, j)] - t[X3(i, k, j)]));
    if (j <= jmt - 1)
      UV_CYC_STORE(ddy, IDX, i, TMASK(i, k, j) * TMASK(i, k, j + 1) * c.dyur[j - 1] * (t[X3(i, k, j + 1)] - t[X3(i, k, j)]));
#undef IDX
  }
}

// ---------------------------------------------------------------------------
// ai_east / ai_north / ai_bottom (isopyc.F:559-921): i = 2..imt-1, k = 1..km,
// j = 1..jmt-1 (east and bottom faces only for j >= 2)
// ---------------------------------------------------------------------------
UVIC_DEV void isopyc_ai_cell(const uvic_ctx &c, int i, int k, int j) {
  UV_DIMS(c);
  const double sc = 1.0 / (c.slmxr * c.dtxsqr[k - 1]);
  const double dzt4r = 0.5 * c.dzt2r[k - 1];
#define IDX(ii) X3(ii, k, j)
  if (j >= 2) {  // east face, isopyc.F:589-660
    const double Ai0 = .5 * (c.fisop[XFIS(i, j, k)] + c.fisop[XFIS(i + 1, j, k)]) * c.ahisop + c.addisop[X3(i, k, j)];
    double sumz = 0.0;
    for (int kr = 0; kr <= 1; ++kr)
      for (int ip = 0; ip <= 1; ++ip) {
        const double sxe = dabs(drodxe(i, k, j, ip) / (drodze(i, k, j, ip, kr) + UV_EPSLN));
        double a;
        if (sxe > sc) {
          const double r = sc / (sxe + UV_EPSLN);
          a = Ai0 * TMASK(i, k, j) * TMASK(i + 1, k, j) * (r * r);
        } else {
          a = Ai0 * TMASK(i, k, j) * TMASK(i + 1, k, j);
        }
        double *p = c.Ai_ez + (size_t)(ip + 2 * kr) * N3;
        UV_CYC_STORE(p, IDX, i, a);
        sumz = sumz + c.dzw[k - 1 + kr] * a;
      }
    UV_CYC_STORE(c.K11, IDX, i, dzt4r * sumz);
  }
  {  // north face, isopyc.F:697-766
    const double Ai0 = 0.5 * (c.fisop[XFIS(i, j, k)] + c.fisop[XFIS(i, j + 1, k)]) * c.ahisop;
    double sumz = 0.0;
    for (int kr = 0; kr <= 1; ++kr)
      for (int jq = 0; jq <= 1; ++jq) {
        const double syn = dabs(drodyn(i, k, j, jq) / (drodzn(i, k, j, jq, kr) + UV_EPSLN));
        double a;
        if (syn > sc) {
          const double r = sc / (syn + UV_EPSLN);
          a = Ai0 * TMASK(i, k, j) * TMASK(i, k, j + 1) * (r * r);
        } else {
          a = Ai0 * TMASK(i, k, j) * TMASK(i, k, j + 1);
        }
        double *p = c.Ai_nz + (size_t)(jq + 2 * kr) * N3;
        UV_CYC_STORE(p, IDX, i, a);
        sumz = sumz + c.dzw[k - 1 + kr] * a;
      }
    UV_CYC_STORE(c.K22, IDX, i, dzt4r * sumz);
  }
  if (j >= 2 && k <= km - 1) {  // bottom face, isopyc.F:803-916
    const double Ai0 = 0.5 * (c.fisop[XFIS(i, j, k + 1)] + c.fisop[XFIS(i, j, k)]) * c.ahisop;
    double sumx = 0.0;
    for (int ip = 0; ip <= 1; ++ip)
      for (int kr = 0; kr <= 1; ++kr) {
        const double sxb = dabs(drodxb(i, k, j, ip, kr) / (drodzb(i, k, j, kr) + UV_EPSLN));
        double a;
        if (sxb > sc) {
          const double r = sc / (sxb + UV_EPSLN);
          a = Ai0 * TMASK(i, k + 1, j) * (r * r);
        } else {
          a = Ai0 * TMASK(i, k + 1, j);
        }
        double *p = c.Ai_bx + (size_t)(ip + 2 * kr) * N3;
        UV_CYC_STORE(p, IDX, i, a);
        sumx = sumx + c.dxu[i - 1 + ip - 1] * a * (sxb * sxb);
      }
    double sumy = 0.0;
    for (int jq = 0; jq <= 1; ++jq) {
      const double facty = c.csu[j - 1 + jq - 1] * c.dyu[j - 1 + jq - 1];
      for (int kr = 0; kr <= 1; ++kr) {
        const double syb = dabs(drodyb(i, k, j, jq, kr) / (drodzb(i, k, j, kr) + UV_EPSLN));
        double a;
        if (syb > sc) {
          const double r = sc / (syb + UV_EPSLN);
          a = Ai0 * TMASK(i, k + 1, j) * (r * r);
        } else {
          a = Ai0 * TMASK(i, k + 1, j);
        }
        double *p = c.Ai_by + (size_t)(jq + 2 * kr) * N3;
        UV_CYC_STORE(p, IDX, i, a);
        sumy = sumy + facty * a * (syb * syb);
      }
    }
    UV_CYC_STORE(c.K33, IDX, i, c.dxt4r[i - 1] * sumx + c.dyt4r[j - 1] * c.cstr[j - 1] * sumy);
  }
#undef IDX
}

// ---------------------------------------------------------------------------
// isopyc_adv, horizontal GM velocities (isopyc.F:1187-1482): i = 1..imt,
// k = 1..km, j = 1..jmt-1.  The reference stores the face density gradients in
// eight full-grid arrays (drodxte ... drodzbn); here they are formed in
// registers: "top" gradients of level k are the "bottom" ones of level k-1.
// ---------------------------------------------------------------------------
struct GmGrad {
  double dy, dzn, dx, dze;
};

// bottom-face gradients of level kk (isopyc.F:1209-1244), valid for i <= imt-1
UVIC_DEV GmGrad gm_bottom(const uvic_ctx &c, int i, int kk, int j) {
  UV_DIMS(c);
  const int kp1 = imin(kk + 1, km);
  GmGrad g;
  double ab = (ALPHA(i, kk, j) + ALPHA(i, kk, j + 1) + ALPHA(i, kp1, j) + ALPHA(i, kp1, j + 1)) * 0.25;
  double bb = (BETA(i, kk, j) + BETA(i, kk, j + 1) + BETA(i, kp1, j) + BETA(i, kp1, j + 1)) * 0.25;
  g.dy = ab * 0.5 * (DDYT(i, kk, j, 1) + DDYT(i, kp1, j, 1)) + bb * 0.5 * (DDYT(i, kk, j, 2) + DDYT(i, kp1, j, 2));
  g.dzn = ab * 0.5 * (DDZT(i, kk, j, 1) + DDZT(i, kk, j + 1, 1)) + bb * 0.5 * (DDZT(i, kk, j, 2) + DDZT(i, kk, j + 1, 2));
  ab = (ALPHA(i, kk, j) + ALPHA(i + 1, kk, j) + ALPHA(i, kp1, j) + ALPHA(i + 1, kp1, j)) * 0.25;
  bb = (BETA(i, kk, j) + BETA(i + 1, kk, j) + BETA(i, kp1, j) + BETA(i + 1, kp1, j)) * 0.25;
  g.dx = ab * 0.5 * (DDXT(i, kk, j, 1) + DDXT(i, kp1, j, 1)) + bb * 0.5 * (DDXT(i, kk, j, 2) + DDXT(i, kp1, j, 2));
  g.dze = ab * 0.5 * (DDZT(i, kk, j, 1) + DDZT(i + 1, kk, j, 1)) + bb * 0.5 * (DDZT(i, kk, j, 2) + DDZT(i + 1, kk, j, 2));
  return g;
}

// top-face gradients of level 1 (isopyc.F:1189-1205)
UVIC_DEV GmGrad gm_top1(const uvic_ctx &c, int i, int j) {
  UV_DIMS(c);
  GmGrad g;
  double at = 0.5 * (ALPHA(i, 1, j) + ALPHA(i, 1, j + 1));
  double bt = 0.5 * (BETA(i, 1, j) + BETA(i, 1, j + 1));
  g.dy = at * DDYT(i, 1, j, 1) + bt * DDYT(i, 1, j, 2);
  g.dzn = at * (DDZT(i, 1, j, 1) + DDZT(i, 1, j + 1, 1)) * 0.5 + bt * (DDZT(i, 1, j, 2) + DDZT(i, 1, j + 1, 2)) * 0.5;
  at = 0.5 * (ALPHA(i, 1, j) + ALPHA(i + 1, 1, j));
  bt = 0.5 * (BETA(i, 1, j) + BETA(i + 1, 1, j));
  g.dx = at * DDXT(i, 1, j, 1) + bt * DDXT(i, 1, j, 2);
  g.dze = at * (DDZT(i, 1, j, 1) + DDZT(i + 1, 1, j, 1)) * 0.5 + bt * (DDZT(i, 1, j, 2) + DDZT(i + 1, 1, j, 2)) * 0.5;
  return g;
}

UVIC_DEV double gm_taper(double ath0, double m1, double m2, double abss, double sc) {
  if (abss > sc) {
    const double r = sc / (abss + UV_EPSLN);
    return ath0 * m1 * m2 * (r * r);
  }
  return ath0 * m1 * m2;
}

// `ven` (column-kernel path, else null): the plane of (tot_e, tot_n) pairs that pass A reads (kernels_col.hpp: CF_VE)
UVIC_DEV void isopyc_adv_cell(const uvic_ctx &c, int i, int k, int j, double *ven = nullptr) {
  UV_DIMS(c);
  const double sc = 1.0 / (c.slmxr * c.dtxsqr[k - 1]);
  const int kp1 = imin(k + 1, km);
  const double top_bc = (k == 1) ? 0.0 : 1.0, bot_bc = (k == km) ? 0.0 : 1.0;
  GmGrad top = {0.0, 0.0, 0.0, 0.0}, bot = {0.0, 0.0, 0.0, 0.0};
  if (i <= imt - 1) {  // column imt of the gradient arrays is never written by the reference
    bot = gm_bottom(c, i, k, j);
    top = (k == 1) ? gm_top1(c, i, j) : gm_bottom(c, i, k - 1, j);
  }
  {  // meridional, isopyc.F:1381-1430, i = 1..imt
    const double Ath0 = c.athkdf * 0.5 * (c.fisop[XFIS(i, j, k)] + c.fisop[XFIS(i, j + 1, k)]);
    const double stn = -top.dy / (top.dzn + 0.125 * UV_EPSLN);
    const double sbn = -bot.dy / (bot.dzn + 0.125 * UV_EPSLN);
    const double ath_t = gm_taper(Ath0, TMASK(i, k, j), TMASK(i, k, j + 1), dabs(stn), sc);
    const double ath_b = gm_taper(Ath0, TMASK(i, kp1, j), TMASK(i, kp1, j + 1), dabs(sbn), sc);
    const double v = -(ath_t * stn * top_bc - ath_b * sbn * bot_bc) * c.dztr[k - 1] * c.csu[j - 1];
    c.adv_vntiso[X3(i, k, j)] = v;
    c.tot_n[X3(i, k, j)] = c.adv_vnt[X3(i, k, j)] + v;
    if (ven) ven[2 * X3(i, k, j) + 1] = c.adv_vnt[X3(i, k, j)] + v;
  }
  if (j >= 2 && i >= 2 && i <= imt - 1) {  // zonal, isopyc.F:1436-1488 (+ setbcx)
    const double Ath0 = c.athkdf * 0.5 * (c.fisop[XFIS(i, j, k)] + c.fisop[XFIS(i + 1, j, k)]);
    const double ste = -top.dx / (top.dze + 0.125 * UV_EPSLN);
    const double sbe = -bot.dx / (bot.dze + 0.125 * UV_EPSLN);
    const double ath_t = gm_taper(Ath0, TMASK(i, k, j), TMASK(i + 1, k, j), dabs(ste), sc);
    const double ath_b = gm_taper(Ath0, TMASK(i, kp1, j), TMASK(i + 1, kp1, j), dabs(sbe), sc);
    const double v = -(ath_t * ste * top_bc - ath_b * sbe * bot_bc) * c.dztr[k - 1];
    c.adv_vetiso[X3(i, k, j)] = v;
    c.tot_e[X3(i, k, j)] = c.adv_vet[X3(i, k, j)] + v;
    if (ven) ven[2 * X3(i, k, j)] = c.adv_vet[X3(i, k, j)] + v;
    if (i == 2) {
      c.adv_vetiso[X3(imt, k, j)] = v;
      c.tot_e[X3(imt, k, j)] = c.adv_vet[X3(imt, k, j)] + v;
    }
    if (i == imt - 1) {
      c.adv_vetiso[X3(1, k, j)] = v;
      c.tot_e[X3(1, k, j)] = c.adv_vet[X3(1, k, j)] + v;
    }
  }
}

// ---------------------------------------------------------------------------
// vertical GM velocity by continuity and a prefix sum over k
// (isopyc.F:1496-1526), plus diff_cbt = background + K33
// (updates/09/source/mom/vmixc.F:182-188) and the total vertical advective
// velocity.  One thread per column, i = 2..imt-1, j = 2..jmt-1.
// ---------------------------------------------------------------------------
// `vbs` (column-kernel path, else null): the plane of (tot_b of face k, tot_n of row j-1) pairs of pass A (kernels_col.hpp: CF_VB)
UVIC_DEV void isopyc_column(const uvic_ctx &c, int i, int j, double *vbs = nullptr) {
  UV_DIMS(c);
  double run = 0.0;
  const int kz = c.kmt[X2(i, j)];
  // The running sum is the only level-to-level dependency.  The operands of eight levels are read
  // before their eight sums are stored: stores may alias the next loads as far as the compiler knows,
  // so a level-by-level loop would pay one memory round trip per level.
#define IDXF(ii) XF(ii, k, j)
  for (int k0 = 0; k0 <= km; k0 += 8) {
    double d[8], vb[8], vs[8];
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      d[u] = 0.0; vb[u] = 0.0; vs[u] = 0.0;
      if (k <= km) {
        vb[u] = c.adv_vbt[XF(i, k, j)];
        if (vbs && k >= 1) vs[u] = c.tot_n[X3(i, k, j - 1)];
        if (k >= 1 && k <= km - 1)
          d[u] = c.dzt[k - 1] * c.cstr[j - 1] *
                 ((c.adv_vetiso[X3(i, k, j)] - c.adv_vetiso[X3(i - 1, k, j)]) * c.dxtr[i - 1] +
                  (c.adv_vntiso[X3(i, k, j)] - c.adv_vntiso[X3(i, k, j - 1)]) * c.dytr[j - 1]);
      }
    }
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      if (k > km) break;
      double v = 0.0;
      if (k >= 1 && k <= km - 1) {
        run = d[u] + run;
        v = run;
      }
      if (k == kz) v = 0.0;
      UV_CYC_STORE(c.adv_vbtiso, IDXF, i, v);
      UV_CYC_STORE(c.tot_b, IDXF, i, vb[u] + v);
      if (vbs && k >= 1) {   // (at k = km v is zero: adv_vbt itself, the flux through the bottom face, tracer.F:1065)
        vbs[2 * X3(i, k, j)] = vb[u] + v;
        vbs[2 * X3(i, k, j) + 1] = vs[u];
      }
    }
  }
#undef IDXF
  if (!c.diff_cbt_given)
    for (int k0 = 1; k0 <= km; k0 += 8) {
      double s[8];
      _Pragma("unroll") for (int u = 0; u < 8; ++u)
        s[u] = (k0 + u <= km) ? c.diff_cbt_bg[X3(i, k0 + u, j)] + c.K33[X3(i, k0 + u, j)] : 0.0;
      _Pragma("unroll") for (int u = 0; u < 8; ++u)
        if (k0 + u <= km) c.diff_cbt[X3(i, k0 + u, j)] = s[u];
    }
}

}  // namespace uvic
#endif
