/* prep_oracle.c -- CPU restatement of the producers of the tracer step's shared inputs
 * (SURVEY.md §8f rank 1).  TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Follows, line by line,
 *   /root/reference/source/mom/adv_vel.F:63-131             advective velocities on T-cell faces
 *   /root/reference/updates/09/source/mom/vmixc.F:62-190    vertical diffusivity: tidal mixing + K33
 * for the option sets of oracle/build_ref.py (O_constvmix O_tidal_kv O_isopycmix, rigid lid, one memory
 * window: joff = 0, js = 1, je = jmt).  Arrays are Fortran order, i fastest, all jmt rows.
 * Compile: gcc -O2 -ffp-contract=off -std=gnu99 (oracle_c.py).
 */
#include <math.h>
#include <stddef.h>

#define X3(i, k, j) ((size_t)((i)-1) + (size_t)imt * ((size_t)((k)-1) + (size_t)km * ((j)-1)))
#define XF(i, k, j) ((size_t)((i)-1) + (size_t)imt * ((size_t)(k) + (size_t)(km + 1) * ((j)-1))) /* k = 0..km */
#define X2(i, j) ((size_t)((i)-1) + (size_t)imt * ((j)-1))

static void setbcx(double *a, int imt, int n) { /* source/common/util.F:789-814 */
  for (int k = 0; k < n; ++k) {
    a[(size_t)k * imt] = a[(size_t)k * imt + imt - 2];
    a[(size_t)k * imt + imt - 1] = a[(size_t)k * imt + 1];
  }
}

/* adv_vel.F:63-131.  u1,u2 = u(:,:,:,1:2,tau) on (imt,km,jmt). */
void orc_adv_vel(int imt, int jmt, int km, const double *u1, const double *u2, const double *dxu, const double *dyu,
                 const double *dxt2r, const double *dyt2r, const double *dxtr, const double *dytr, const double *cstr,
                 const double *csu, const double *dzt, double *adv_vet, double *adv_vnt, double *adv_vbt) {
  const int istrt = 2, iend = imt - 1;
  /* north face, rows js..je = 1..jmt (adv_vel.F:63-72) */
  for (int j = 1; j <= jmt; ++j) {
    for (int k = 1; k <= km; ++k)
      for (int i = istrt; i <= iend; ++i)
        adv_vnt[X3(i, k, j)] = (u2[X3(i, k, j)] * dxu[i - 1] + u2[X3(i - 1, k, j)] * dxu[i - 2]) * csu[j - 1] * dxt2r[i - 1];
    setbcx(adv_vnt + X3(1, 1, j), imt, km);
  }
  /* east face, rows max(js,jsmw)..je = 2..jmt, i = istrt-1..iend+1 (adv_vel.F:79-88) */
  for (int j = 2; j <= jmt; ++j)
    for (int k = 1; k <= km; ++k)
      for (int i = istrt - 1; i <= iend + 1; ++i)
        adv_vet[X3(i, k, j)] = (u1[X3(i, k, j)] * dyu[j - 1] + u1[X3(i, k, j - 1)] * dyu[j - 2]) * dyt2r[j - 1];
  /* bottom face by continuity (adv_vel.F:94-131), rigid lid */
  for (int j = 2; j <= jmt; ++j) {
    for (int i = istrt; i <= iend; ++i) adv_vbt[XF(i, 0, j)] = 0.0;
    for (int k = 1; k <= km; ++k)
      for (int i = istrt; i <= iend; ++i)
        adv_vbt[XF(i, k, j)] = ((adv_vet[X3(i, k, j)] - adv_vet[X3(i - 1, k, j)]) * dxtr[i - 1] +
                                (adv_vnt[X3(i, k, j)] - adv_vnt[X3(i, k, j - 1)]) * dytr[j - 1]) *
                               cstr[j - 1] * dzt[k - 1];
    for (int k = 1; k <= km; ++k)
      for (int i = istrt; i <= iend; ++i) adv_vbt[XF(i, k, j)] = adv_vbt[XF(i, k, j)] + adv_vbt[XF(i, k - 1, j)];
    setbcx(adv_vbt + XF(1, 0, j), imt, km + 1);
  }
}

/* vmixc.F:62-190, tracer part: diff_cbt(i,k,j) for rows 2..jmt-1.
 * ddzt is (imt,0:km,jmt,2) as in isopyc.h (k = 0..km), alphai/betai/K33 (imt,km,jmt).
 * Levels k >= kmt keep their previous diff_cbt before K33 is added, as in the reference. */
void orc_vmixc(int imt, int jmt, int km, const int *kmt, const double *tlat, const double *zw, const double *alphai,
               const double *betai, const double *ddzt, const double *K33, const double *edrm2, const double *edrs2,
               const double *edrk1, const double *edro1, double kappa_h, double zetar, double ogamma, double gravrho0r,
               double *diff_cbt) {
  const size_t NF = (size_t)imt * (km + 1) * jmt;
  for (int j = 2; j <= jmt - 1; ++j)
    for (int i = 2; i <= imt - 1; ++i) {
      double qk1, qo1, q2;
      if (fabs(tlat[X2(i, j)]) < 30.) {
        qk1 = 0.33; qo1 = 0.33;
      } else {
        qk1 = 1.; qo1 = 1.;
      }
      if (fabs(tlat[X2(i, j)]) < 70.) q2 = 0.33; else q2 = 1.;
      const int kz = kmt[X2(i, j)];
      for (int k = 1; k <= kz - 1; ++k) {
        /* drodzb(i,k,j,0), isopyc.h:135-136 */
        const double drodzb = alphai[X3(i, k, j)] * ddzt[XF(i, k, j)] + betai[X3(i, k, j)] * ddzt[XF(i, k, j) + NF];
        const double zn2 = fmax(-gravrho0r * drodzb, 1e-8);
        double edr = 0.;
        for (int k1 = k + 1; k1 <= kz; ++k1) {
          const double hab = zw[k - 1] - zw[k1 - 1];
          edr = edr + (q2 * (edrm2[X3(i, k1, j)] + edrs2[X3(i, k1, j)]) + qk1 * edrk1[X3(i, k1, j)] + qo1 * edro1[X3(i, k1, j)]) *
                          exp(hab * zetar) / (1 - exp(-zetar * zw[k1 - 1]));
        }
        const double zkappa = ogamma * edr / zn2;
        diff_cbt[X3(i, k, j)] = fmax(kappa_h, fmin(100., zkappa + kappa_h));
      }
    }
  /* add the K33 component, vmixc.F:182-188 */
  for (int j = 2; j <= jmt - 1; ++j)
    for (int i = 2; i <= imt - 1; ++i)
      for (int k = 1; k <= km; ++k) diff_cbt[X3(i, k, j)] = diff_cbt[X3(i, k, j)] + K33[X3(i, k, j)];
}
