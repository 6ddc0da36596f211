// kernels_prep.hpp -- producers of the tracer step's shared inputs:
//   adv_vel   /root/reference/source/mom/adv_vel.F:63-131        (T-cell part, rigid lid)
//   vmixc     /root/reference/updates/09/source/mom/vmixc.F:62-190 (O_constvmix O_tidal_kv O_isopycmix)
// Same expressions and evaluation order as the reference; one memory window, joff = 0.  vmixc's exponentials depend on
// the level pair only: the library tabulates them on the host with the C library's exp -- the one the compiled
// reference calls -- so diff_cbt comes out bit-identical (uvic_ctx.vmix_e/vmix_d; without the tables the kernel calls
// exp itself and differs by its rounding).
#ifndef UVIC_KERNELS_PREP_HPP
#define UVIC_KERNELS_PREP_HPP

#include "kernels_isopyc.hpp"

namespace uvic {

// adv_vnt (rows 1..jmt) and adv_vet (rows 2..jmt); one thread per cell, i = 1..imt.
// Columns 1 and imt of adv_vnt are the cyclic images the reference makes with setbcx.
UVIC_DEV void adv_vel_hor_cell(const uvic_ctx &c, int i, int k, int j) {
  UV_DIMS(c);
  double *vnt = const_cast<double *>(c.adv_vnt), *vet = const_cast<double *>(c.adv_vet);
  const int iw = (i == 1) ? imt - 1 : ((i == imt) ? 2 : i);  // cyclic image columns
  vnt[X3(i, k, j)] = (c.u2[X3(iw, k, j)] * c.dxu[iw - 1] + c.u2[X3(iw - 1, k, j)] * c.dxu[iw - 2]) * c.csu[j - 1] * c.dxt2r[iw - 1];
  if (j >= 2) vet[X3(i, k, j)] = (c.u1[X3(i, k, j)] * c.dyu[j - 1] + c.u1[X3(i, k, j - 1)] * c.dyu[j - 2]) * c.dyt2r[j - 1];
}

// adv_vbt by continuity, integrated downward; one thread per column, rows 2..jmt, i = 2..imt-1
UVIC_DEV void adv_vel_vert_column(const uvic_ctx &c, int i, int j) {
  UV_DIMS(c);
  double *vbt = const_cast<double *>(c.adv_vbt);
  const int ic = (i == 2) ? imt : ((i == imt - 1) ? 1 : 0);
  double acc = 0.0;
  vbt[XF(i, 0, j)] = acc;
  if (ic) vbt[XF(ic, 0, j)] = acc;
  for (int k = 1; k <= km; ++k) {
    const double div = ((c.adv_vet[X3(i, k, j)] - c.adv_vet[X3(i - 1, k, j)]) * c.dxtr[i - 1] +
                        (c.adv_vnt[X3(i, k, j)] - c.adv_vnt[X3(i, k, j - 1)]) * c.dytr[j - 1]) *
                       c.cstr[j - 1] * c.dzt[k - 1];
    acc = div + acc;
    vbt[XF(i, k, j)] = acc;
    if (ic) vbt[XF(ic, k, j)] = acc;
  }
}

// diff_cbt: tidal mixing above the bottom level (an O(km) sum per cell), previous value elsewhere,
// plus K33; one thread per cell, rows 2..jmt-1, i = 2..imt-1
UVIC_DEV double vmixc_cell(const uvic_ctx &c, int i, int k, int j) {
  UV_DIMS(c);
  const int kz = c.kmt[X2(i, j)];
  double d = c.diff_cbt[X3(i, k, j)];
  if (k <= kz - 1) {
    const double at = dabs(c.tlat[X2(i, j)]);
    const double qk1 = (at < 30.) ? 0.33 : 1., qo1 = qk1;
    const double q2 = (at < 70.) ? 0.33 : 1.;
    const double zn2 = dmax(-c.gravrho0r * drodzb(i, k, j, 0), 1e-8);
    double edr = 0.;
    for (int k1 = k + 1; k1 <= kz; ++k1) {
      const double hab = c.zw[k - 1] - c.zw[k1 - 1];
      const double e = c.vmix_e ? c.vmix_e[(k - 1) * km + k1 - 1] : exp(hab * c.zetar);
      const double dn = c.vmix_d ? c.vmix_d[k1 - 1] : 1 - exp(-c.zetar * c.zw[k1 - 1]);
      edr = edr + (q2 * (c.edrm2[X3(i, k1, j)] + c.edrs2[X3(i, k1, j)]) + qk1 * c.edrk1[X3(i, k1, j)] + qo1 * c.edro1[X3(i, k1, j)]) *
                      e / dn;
    }
    const double zkappa = c.ogamma * edr / zn2;
    d = dmax(c.kappa_h, dmin(100., zkappa + c.kappa_h));
  }
  d = d + c.K33[X3(i, k, j)];
  c.diff_cbt[X3(i, k, j)] = d;
  return d;
}

}  // namespace uvic
#endif
