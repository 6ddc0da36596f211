/* clinic_oracle.c -- CPU restatement of the baroclinic momentum step.
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md and clinic_oracle.h).
 *
 * Follows, loop by loop,
 *   /root/reference/source/mom/state.F:1-41                  density at T-cell centres
 *   /root/reference/source/mom/adv_vel.F:150-231             advective velocities on U-cell faces
 *   /root/reference/updates/09/source/mom/setvbc.F:170-194   bottom drag
 *   /root/reference/updates/09/source/mom/clinic.F:24-560    internal-mode velocities at tau+1 and the
 *                                                            vertically averaged forcing zu
 *   /root/reference/updates/09/source/mom/clinic.F:729-895   asbcu, isbcu
 *   /root/reference/updates/09/source/mom/loadmw.F:590-667   add_ext_mode (O_stream_function)
 * with the statement functions of /root/reference/updates/09/source/mom/fdifm.h, for the options
 * O_consthmix O_constvmix O_anisotropic_viscosity O_stream_function O_cyclic (no O_implicitvmix,
 * O_damp_inertial_oscillation, O_biharmonic, O_neptune, O_pressure_gradient_average), one memory window
 * (joff = 0, js = 2, je = jmt-1, istrt = 2, iend = imt-1).  The polar filter (`filuv`) is orc_filuv in filter_oracle.c.
 * Compile: gcc -O2 -ffp-contract=off -std=gnu99 (oracle_c.py).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "clinic_oracle.h"

#define X3(i, k, j) ((size_t)((i)-1) + (size_t)imt * ((size_t)((k)-1) + (size_t)km * ((j)-1)))
#define XF(i, k, j) ((size_t)((i)-1) + (size_t)imt * ((size_t)(k) + (size_t)(km + 1) * ((j)-1))) /* k = 0..km */
#define X2(i, j) ((size_t)((i)-1) + (size_t)imt * ((j)-1))

static void setbcx(double *a, int imt, int n) { /* source/common/util.F:789-814 */
  for (int k = 0; k < n; ++k) {
    a[(size_t)k * imt] = a[(size_t)k * imt + imt - 2];
    a[(size_t)k * imt + imt - 1] = a[(size_t)k * imt + 1];
  }
}

/* source/mom/dens.h:13-16 */
static inline double dens(const double *cc, int km, double tq, double sq, int k) {
#define C(kk, n) cc[((kk)-1) + (size_t)km * ((n)-1)]
  return (C(k, 1) + (C(k, 4) + C(k, 7) * sq) * sq + (C(k, 3) + C(k, 8) * sq + C(k, 6) * tq) * tq) * tq +
         (C(k, 2) + (C(k, 5) + C(k, 9) * sq) * sq) * sq;
#undef C
}

void orc_state(int imt, int jmt, int km, const double *t, const double *s, const double *to, const double *so,
               const double *c, double *rho, int js, int je) {
  (void)jmt;
  for (int j = js; j <= je; ++j)
    for (int k = 1; k <= km; ++k)
      for (int i = 1; i <= imt; ++i)
        rho[X3(i, k, j)] = dens(c, km, t[X3(i, k, j)] - to[k - 1], s[X3(i, k, j)] - so[k - 1], k);
}

/* adv_vel.F:150-231 as called at mom.F:332 (js = 1, je = jmt): adv_vnu rows 1..jmt-1, adv_veu and adv_vbu rows 2..jmt-1 */
void orc_adv_vel_u(orc_mom *m) {
  const int imt = m->imt, jmt = m->jmt, km = m->km;
  const int istrt = 2, iend = imt - 1;
  const double *vnt = m->adv_vnt, *vet = m->adv_vet, *vbt = m->adv_vbt;
  for (int j = 1; j <= jmt - 1; ++j) {
    const double dyr = m->dytr[j];   /* dytr(jrow+1) */
    for (int k = 1; k <= km; ++k)
      for (int i = istrt; i <= iend; ++i)
        m->adv_vnu[X3(i, k, j)] = ((vnt[X3(i, k, j)] * m->duw[i - 1] + vnt[X3(i + 1, k, j)] * m->due[i - 1]) * m->dus[j] +
                                   (vnt[X3(i, k, j + 1)] * m->duw[i - 1] + vnt[X3(i + 1, k, j + 1)] * m->due[i - 1]) * m->dun[j - 1]) *
                                  dyr * m->dxur[i - 1];
    setbcx(m->adv_vnu + X3(1, 1, j), imt, km);
  }
  for (int j = 2; j <= jmt - 1; ++j) {
    const double dyr = m->dyur[j - 1];
    for (int k = 1; k <= km; ++k)
      for (int i = istrt - 1; i <= iend; ++i)
        m->adv_veu[X3(i, k, j)] = ((vet[X3(i, k, j)] * m->dus[j - 1] + vet[X3(i, k, j + 1)] * m->dun[j - 1]) * m->duw[i] +
                                   (vet[X3(i + 1, k, j)] * m->dus[j - 1] + vet[X3(i + 1, k, j + 1)] * m->dun[j - 1]) * m->due[i - 1]) *
                                  dyr * m->dxtr[i];
    setbcx(m->adv_veu + X3(1, 1, j), imt, km);
  }
  for (int j = 2; j <= jmt - 1; ++j) {
    const double dyn = m->dun[j - 1] * m->cst[j];
    const double dys = m->dus[j - 1] * m->cst[j - 1];
    const double dyr = m->dyur[j - 1] * m->csur[j - 1];
    for (int k = 0; k <= km; ++k)
      for (int i = istrt; i <= iend; ++i) {
        const double asw = m->duw[i - 1] * dys, anw = m->duw[i - 1] * dyn;
        const double ase = m->due[i - 1] * dys, ane = m->due[i - 1] * dyn;
        m->adv_vbu[XF(i, k, j)] = dyr * m->dxur[i - 1] *
                                  (vbt[XF(i, k, j)] * asw + vbt[XF(i + 1, k, j)] * ase + vbt[XF(i, k, j + 1)] * anw +
                                   vbt[XF(i + 1, k, j + 1)] * ane);
      }
    setbcx(m->adv_vbu + XF(1, 0, j), imt, km + 1);
  }
}

/* setvbc.F:170-209, rows 2..jmt-1 */
void orc_bmf(orc_mom *m) {
  const int imt = m->imt, jmt = m->jmt, km = m->km;
  const size_t N2 = (size_t)imt * jmt;
  for (int n = 1; n <= 2; ++n)
    for (int j = 2; j <= jmt - 1; ++j)
      for (int i = 2; i <= imt - 1; ++i) {
        double v = 0.0;
        const int kz = m->kmu[X2(i, j)];
        if (m->cdbot != 0.0 && kz != 0) {
          const double a = m->u_taum1[0][X3(i, kz, j)], b = m->u_taum1[1][X3(i, kz, j)];
          const double uvmag = sqrt(a * a + b * b);
          v = m->cdbot * m->u_taum1[n - 1][X3(i, kz, j)] * uvmag;
        }
        m->bmf[X2(i, j) + (size_t)(n - 1) * N2] = v;
      }
  for (int n = 1; n <= 2; ++n) setbcx(m->bmf + X2(1, 2) + (size_t)(n - 1) * N2, imt, jmt - 2); /* setvbc.F:206-209 */
}

void orc_clinic(orc_mom *m) {
  const int imt = m->imt, jmt = m->jmt, km = m->km;
  const int istrt = 2, iend = imt - 1, js = 2, je = jmt - 1, kmm1 = km - 1;
  const size_t N3 = (size_t)imt * km * jmt, NF = (size_t)imt * (km + 1) * jmt, N2 = (size_t)imt * jmt;
  const double p5 = 0.5;
  double *csudxur = calloc(N2, 8), *csudxu2r = calloc(N2, 8), *am_csudxtr = calloc(N3, 8);
  double *tempik = calloc(N3, 8);
  double *adv_fe = calloc(N3, 8), *diff_fe = calloc(N3, 8), *adv_fb = calloc(NF, 8), *diff_fb = calloc(NF, 8);
  double *baru = calloc(2 * N2, 8);
  double *gp = m->grad_p;
  const double *rho = m->rho;

  /* clinic.F:72-92 */
  for (int j = js; j <= je; ++j)
    for (int k = 1; k <= km; ++k)
      for (int i = istrt - 1; i <= iend; ++i) {
        csudxur[X2(i, j)] = m->csur[j - 1] * m->dxur[i - 1];
        csudxu2r[X2(i, j)] = m->csur[j - 1] * m->dxur[i - 1] * p5;
        am_csudxtr[X3(i, k, j)] = m->visc_ceu[X3(i, k, j)] * m->csur[j - 1] * m->dxtr[i];
      }

  /* hydrostatic pressure gradients, clinic.F:119-186 */
  const double grav_rho0r = m->grav * m->rho0r;
  for (int j = js; j <= je; ++j) {
    const double fxa = grav_rho0r * m->dzw[0] * m->csur[j - 1];
    const double fxb = grav_rho0r * m->dzw[0] * m->dyu2r[j - 1];
    for (int i = istrt - 1; i <= iend; ++i) {
      const double t1 = rho[X3(i + 1, 1, j + 1)] - rho[X3(i, 1, j)];
      const double t2 = rho[X3(i, 1, j + 1)] - rho[X3(i + 1, 1, j)];
      gp[X3(i, 1, j)] = (t1 - t2) * fxa * m->dxu2r[i - 1];
      gp[X3(i, 1, j) + N3] = (t1 + t2) * fxb;
    }
  }
  for (int j = js; j <= je + 1; ++j)
    for (int k = 2; k <= km; ++k)
      for (int i = istrt - 1; i <= iend + 1; ++i) tempik[X3(i, k, j)] = rho[X3(i, k - 1, j)] + rho[X3(i, k, j)];
  for (int j = js; j <= je; ++j) {
    const double fxa = grav_rho0r * m->csur[j - 1] * p5;
    const double fxb = grav_rho0r * m->dyu4r[j - 1];
    for (int k = 2; k <= km; ++k)
      for (int i = istrt - 1; i <= iend; ++i) {
        const double t1 = tempik[X3(i + 1, k, j + 1)] - tempik[X3(i, k, j)];
        const double t2 = tempik[X3(i, k, j + 1)] - tempik[X3(i + 1, k, j)];
        gp[X3(i, k, j)] = fxa * (t1 - t2) * m->dzw[k - 1] * m->dxu2r[i - 1];
        gp[X3(i, k, j) + N3] = fxb * (t1 + t2) * m->dzw[k - 1];
      }
  }
  for (int j = js; j <= je; ++j)
    for (int k = 1; k <= kmm1; ++k)
      for (int i = istrt - 1; i <= iend; ++i) {
        gp[X3(i, k + 1, j)] = gp[X3(i, k, j)] + gp[X3(i, k + 1, j)];
        gp[X3(i, k + 1, j) + N3] = gp[X3(i, k, j) + N3] + gp[X3(i, k + 1, j) + N3];
      }
  for (int j = js; j <= je; ++j) {
    setbcx(gp + X3(1, 1, j), imt, km);
    setbcx(gp + X3(1, 1, j) + N3, imt, km);
  }

  /* clinic.F:188-412: the two velocity components */
  for (int n = 1; n <= 2; ++n) {
    const double *ut = m->u_tau[n - 1], *um = m->u_taum1[n - 1];
    const double *uto = m->u_tau[2 - n], *umo = m->u_taum1[2 - n]; /* component 3-n */
    const double *ut1 = m->u_tau[0];
    double *up = m->u_taup1[n - 1];
    for (int j = js; j <= je; ++j)
      for (int k = 1; k <= km; ++k)
        for (int i = istrt - 1; i <= iend; ++i) adv_fe[X3(i, k, j)] = m->adv_veu[X3(i, k, j)] * (ut[X3(i, k, j)] + ut[X3(i + 1, k, j)]);
    for (int j = js; j <= je; ++j)
      for (int k = 1; k <= km; ++k)
        for (int i = istrt - 1; i <= iend; ++i) diff_fe[X3(i, k, j)] = am_csudxtr[X3(i, k, j)] * (um[X3(i + 1, k, j)] - um[X3(i, k, j)]);
    for (int j = js; j <= je; ++j)
      for (int k = 1; k <= kmm1; ++k)
        for (int i = istrt; i <= iend; ++i) {
          adv_fb[XF(i, k, j)] = m->adv_vbu[XF(i, k, j)] * (ut[X3(i, k, j)] + ut[X3(i, k + 1, j)]);
          diff_fb[XF(i, k, j)] = m->kappa_m * m->dzwr[k] * (um[X3(i, k, j)] - um[X3(i, k + 1, j)]);
        }
    for (int j = js; j <= je; ++j)
      for (int i = istrt; i <= iend; ++i) {
        const int kb = m->kmu[X2(i, j)];
        diff_fb[XF(i, 0, j)] = m->smf[X2(i, j) + (size_t)(n - 1) * N2];
        diff_fb[XF(i, kb, j)] = m->bmf[X2(i, j) + (size_t)(n - 1) * N2];
        adv_fb[XF(i, 0, j)] = m->adv_vbu[XF(i, 0, j)] * (ut[X3(i, 1, j)] + ut[X3(i, 1, j)]);
        adv_fb[XF(i, km, j)] = m->adv_vbu[XF(i, km, j)] * ut[X3(i, km, j)];
      }
    /* the tendency, clinic.F:334-356 with fdifm.h */
    for (int j = js; j <= je; ++j)
      for (int k = 1; k <= km; ++k)
        for (int i = istrt; i <= iend; ++i) {
          const double DIFF_Ux = (diff_fe[X3(i, k, j)] - diff_fe[X3(i - 1, k, j)]) * csudxur[X2(i, j)];
          const double DIFF_Uy = m->amc_north[X3(i, k, j)] * (um[X3(i, k, j + 1)] - um[X3(i, k, j)]) -
                                 m->amc_south[X3(i, k, j)] * (um[X3(i, k, j)] - um[X3(i, k, j - 1)]);
          const double DIFF_Uz = (diff_fb[XF(i, k - 1, j)] - diff_fb[XF(i, k, j)]) * m->dztr[k - 1];
          const double DIFF_metric = m->am3[j - 1] * um[X3(i, k, j)] +
                                     m->am4[(j - 1) + (size_t)(n - 1) * jmt] * m->dxmetr[i - 1] * (umo[X3(i + 1, k, j)] - umo[X3(i - 1, k, j)]);
          const double ADV_Ux = (adv_fe[X3(i, k, j)] - adv_fe[X3(i - 1, k, j)]) * csudxu2r[X2(i, j)];
          const double ADV_Uy = (m->adv_vnu[X3(i, k, j)] * (ut[X3(i, k, j)] + ut[X3(i, k, j + 1)]) -
                                 m->adv_vnu[X3(i, k, j - 1)] * (ut[X3(i, k, j - 1)] + ut[X3(i, k, j)])) *
                                m->csudyu2r[j - 1];
          const double ADV_Uz = (adv_fb[XF(i, k - 1, j)] - adv_fb[XF(i, k, j)]) * m->dzt2r[k - 1];
          const double ADV_metric = m->advmet[(j - 1) + (size_t)(n - 1) * jmt] * ut1[X3(i, k, j)] * uto[X3(i, k, j)];
          const double CORIOLIS = m->cori[X2(i, j) + (size_t)(n - 1) * N2] * uto[X3(i, k, j)];
          const double source = 0.0;
          up[X3(i, k, j)] = (DIFF_Ux + DIFF_Uy + DIFF_Uz + DIFF_metric - ADV_Ux - ADV_Uy - ADV_Uz + ADV_metric -
                             gp[X3(i, k, j) + (size_t)(n - 1) * N3] + CORIOLIS + source) *
                            m->umask[X3(i, k, j)];
        }
    /* vertical average of the forcing, clinic.F:376-399 */
    double *zu = m->zu + (size_t)(n - 1) * N2;
    for (int j = js; j <= je; ++j)
      for (int i = istrt; i <= iend; ++i) zu[X2(i, j)] = 0.0;
    for (int j = js; j <= je; ++j)
      for (int k = 1; k <= km; ++k) {
        const double fx = m->dzt[k - 1];
        for (int i = istrt; i <= iend; ++i) zu[X2(i, j)] = zu[X2(i, j)] + up[X3(i, k, j)] * fx;
      }
    for (int j = js; j <= je; ++j)
      for (int i = istrt; i <= iend; ++i) zu[X2(i, j)] = zu[X2(i, j)] * m->hr[X2(i, j)];
  }

  /* tau+1 velocities (explicit Coriolis), clinic.F:441-450 */
  for (int n = 1; n <= 2; ++n)
    for (int j = js; j <= je; ++j)
      for (int k = 1; k <= km; ++k)
        for (int i = istrt; i <= iend; ++i)
          m->u_taup1[n - 1][X3(i, k, j)] = m->u_taum1[n - 1][X3(i, k, j)] + m->c2dtuv * m->u_taup1[n - 1][X3(i, k, j)];

  /* subtract the vertical means, clinic.F:458-485 */
  for (int n = 1; n <= 2; ++n) {
    double *up = m->u_taup1[n - 1], *bu = baru + (size_t)(n - 1) * N2;
    for (int j = js; j <= je; ++j)
      for (int i = istrt; i <= iend; ++i) bu[X2(i, j)] = 0.0;
    for (int j = js; j <= je; ++j)
      for (int k = 1; k <= km; ++k)
        for (int i = istrt; i <= iend; ++i) bu[X2(i, j)] = bu[X2(i, j)] + up[X3(i, k, j)] * m->dzt[k - 1];
    for (int j = js; j <= je; ++j)
      for (int i = istrt; i <= iend; ++i) bu[X2(i, j)] = bu[X2(i, j)] * m->hr[X2(i, j)];
    for (int j = js; j <= je; ++j) {
      for (int k = 1; k <= km; ++k)
        for (int i = istrt; i <= iend; ++i) up[X3(i, k, j)] = up[X3(i, k, j)] - m->umask[X3(i, k, j)] * bu[X2(i, j)];
      setbcx(up + X3(1, 1, j), imt, km);
    }
  }
  free(csudxur); free(csudxu2r); free(am_csudxtr); free(tempik);
  free(adv_fe); free(diff_fe); free(adv_fb); free(diff_fb); free(baru);
}

/* u09/mom/loadmw.F:627-667 (add_ext_mode, O_stream_function, one time level): psi (imt,jmt), u1/u2 in place, rows 1..jmt-1 */
void orc_add_ext_mode(const orc_mom *m, const double *psi, double *u1, double *u2) {
  const int imt = m->imt, jmt = m->jmt, km = m->km;
  const int is = 2, ie = imt - 1;
  double *ext1 = calloc(imt + 1, 8), *ext2 = calloc(imt + 1, 8);
  for (int j = 1; j <= jmt - 1; ++j) {
    for (int i = is; i <= ie; ++i) {
      const double diag1 = psi[X2(i + 1, j + 1)] - psi[X2(i, j)];
      const double diag0 = psi[X2(i, j + 1)] - psi[X2(i + 1, j)];
      ext1[i] = -(diag1 + diag0) * m->dyu2r[j - 1] * m->hr[X2(i, j)];
      ext2[i] = (diag1 - diag0) * m->dxu2r[i - 1] * m->hr[X2(i, j)] * m->csur[j - 1];
    }
    for (int k = 1; k <= km; ++k)
      for (int i = is; i <= ie; ++i) {
        u1[X3(i, k, j)] = (u1[X3(i, k, j)] + ext1[i]) * m->umask[X3(i, k, j)];
        u2[X3(i, k, j)] = (u2[X3(i, k, j)] + ext2[i]) * m->umask[X3(i, k, j)];
      }
    setbcx(u1 + X3(1, 1, j), imt, km);
    setbcx(u2 + X3(1, 1, j), imt, km);
  }
  free(ext1); free(ext2);
}

/* clinic.F:853-892: geostrophic (level 2) currents for the ice model */
void orc_isbcu(const orc_mom *m, double *sbc_u, double *sbc_v, int osegs, int osege, double rts, const int *kmt) {
  const int imt = m->imt, jmt = m->jmt, km = m->km;
  const int is = 2, ie = imt - 1, js = 2, je = jmt - 1;
  if (osegs)
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie; ++i)
        if (kmt[X2(i, j)] != 0) {
          sbc_u[X2(i, j)] = 0.0;
          sbc_v[X2(i, j)] = 0.0;
        }
  for (int j = js; j <= je; ++j)
    for (int i = is; i <= ie; ++i) {
      sbc_u[X2(i, j)] = sbc_u[X2(i, j)] + m->u_tau[0][X3(i, 2, j)];
      sbc_v[X2(i, j)] = sbc_v[X2(i, j)] + m->u_tau[1][X3(i, 2, j)];
    }
  if (osege)
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie; ++i)
        if (kmt[X2(i, j)] != 0) {
          sbc_u[X2(i, j)] = rts * sbc_u[X2(i, j)];
          sbc_v[X2(i, j)] = rts * sbc_v[X2(i, j)];
        }
}

/* clinic.F:765-810: surface currents on T cells for the atmosphere/ice coupling */
void orc_asbcu(const orc_mom *m, double *sbc_u, double *sbc_v, int osegs, int osege, double rts, const int *kmt) {
  const int imt = m->imt, jmt = m->jmt, km = m->km;
  const int is = 2, ie = imt - 1, js = 2, je = jmt - 1;
  const double p25 = 0.25;
  const double *u1 = m->u_tau[0], *u2 = m->u_tau[1];
  if (osegs)
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie; ++i)
        if (kmt[X2(i, j)] != 0) {
          sbc_u[X2(i, j)] = 0.0;
          sbc_v[X2(i, j)] = 0.0;
        }
  for (int j = js; j <= je; ++j)
    for (int i = is; i <= ie; ++i) {
      sbc_u[X2(i, j)] = sbc_u[X2(i, j)] + p25 * (u1[X3(i, 1, j)] + u1[X3(i - 1, 1, j)] + u1[X3(i, 1, j - 1)] + u1[X3(i - 1, 1, j - 1)]);
      sbc_v[X2(i, j)] = sbc_v[X2(i, j)] + p25 * (u2[X3(i, 1, j)] + u2[X3(i - 1, 1, j)] + u2[X3(i, 1, j - 1)] + u2[X3(i - 1, 1, j - 1)]);
    }
  if (osege)
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie; ++i)
        if (kmt[X2(i, j)] != 0) {
          sbc_u[X2(i, j)] = rts * sbc_u[X2(i, j)];
          sbc_v[X2(i, j)] = rts * sbc_v[X2(i, j)];
        }
}
