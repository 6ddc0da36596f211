/* filter_oracle.c -- CPU restatement of the polar Fourier filter of the tracers.
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Follows, line by line,
 *   /root/reference/source/common/findex.F:1-101    ocean strips per filtered row and level (O_cyclic)
 *   /root/reference/source/common/filt.F:36-115     gather a strip, filter, scatter (O_fourfil branch)
 *   /root/reference/source/common/filtr.F:1-430     the filter itself: a dense (im x im) operator built
 *                                                   from tabulated cosines, applied as s' = F s
 * Arrays are Fortran order; 1-based indices are kept through the IX macros.
 * Compile: gcc -O2 -ffp-contract=off -std=gnu99 (oracle_c.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- findex.F: isf/ief (jjmax, lsegf, kmax), kxx (imt,jmt) ---------------------------------- */
int orc_findex(const int *kxx, int imt, int jmt, int kmax, int jfrst, int jf1, int jf2, int lsegf, int jjmax, int *isf,
               int *ief) {
#define KXX(i, j) kxx[(size_t)((i)-1) + (size_t)imt * ((j)-1)]
#define ISF(jj, l, k) isf[(size_t)((jj)-1) + (size_t)jjmax * ((size_t)((l)-1) + (size_t)lsegf * ((k)-1))]
#define IEF(jj, l, k) ief[(size_t)((jj)-1) + (size_t)jjmax * ((size_t)((l)-1) + (size_t)lsegf * ((k)-1))]
  const int imax = imt;
  int *iis = (int *)calloc(lsegf + 2, sizeof(int)), *iie = (int *)calloc(lsegf + 2, sizeof(int));
  int jj = 0, rc = 0;
  for (int jrow = jfrst; jrow <= jmt - 1; ++jrow) {
    if (jrow <= jf1 || jrow >= jf2) {
      jj = jj + 1;
      if (jj > jjmax) { rc = 2; break; }
      for (int k = 1; k <= kmax; ++k) {
        for (int l = 1; l <= lsegf + 1; ++l) { iis[l] = 0; iie[l] = 0; }
        int l = 1;
        if (KXX(2, jrow) >= k) iis[1] = 2;
        for (int i = 2; i <= imax - 1; ++i) {
          if (l > lsegf + 1) { rc = 1; break; }
          if (KXX(i - 1, jrow) < k && KXX(i, jrow) >= k) iis[l] = i;
          if (KXX(i, jrow) >= k && KXX(i + 1, jrow) < k) {
            if (i != iis[l] || (i == 2 && KXX(1, jrow) >= k)) {
              iie[l] = i;
              l = l + 1;
            } else {
              iis[l] = 0;
            }
          }
        }
        if (rc) break;
        if (KXX(imax - 1, jrow) >= k && KXX(imax, jrow) >= k) {
          iie[l] = imax - 1;
          l = l + 1;
        }
        int lm = l - 1;
        if (lm > 1) { /* O_cyclic */
          if (iis[1] == 2 && iie[lm] == imax - 1 && KXX(1, jrow) >= k) {
            iis[1] = iis[lm];
            iie[1] = iie[1] + imax - 2;
            iis[lm] = 0;
            iie[lm] = 0;
            lm = lm - 1;
          }
        }
        if (lm > lsegf) { rc = 1; break; }
        for (l = 1; l <= lsegf; ++l) {
          ISF(jj, l, k) = iis[l];
          IEF(jj, l, k) = iie[l];
        }
      }
      if (rc) break;
    }
  }
  free(iis); free(iie);
  return rc;
}

/* ---- filtr.F -------------------------------------------------------------------------------------- */
typedef struct {
  int imt;
  double pi;
  double *cossav, *denmsv, *cosnpi, *ftarr; /* COMMON /cfilt_d/, /cfilt_r/ */
  int *icbase, *idbase;                      /* COMMON /cfilt_i/ */
  double *cof, *cosine, *denom, *temp, *sprime;
  int *indx;
} filtr_state;

static filtr_state *filtr_new(int imt, double pi) {
  filtr_state *f = (filtr_state *)calloc(1, sizeof *f);
  const int imtp1 = imt + 1, imtd2 = imt / 2;
  f->imt = imt; f->pi = pi;
  f->cossav = (double *)calloc((size_t)imtd2 * (imt - imtd2) + imt + 2, 8);
  f->denmsv = (double *)calloc((size_t)imt * imtp1 / 2 + imt + 2, 8);
  f->cosnpi = (double *)calloc(imt + 2, 8);
  f->ftarr = (double *)calloc((size_t)imt * imt + 1, 8);
  f->icbase = (int *)calloc(imtp1 + 2, sizeof(int));
  f->idbase = (int *)calloc(imtp1 + 2, sizeof(int));
  f->cof = (double *)calloc((size_t)imt * 8 + 2, 8);
  f->cosine = (double *)calloc((size_t)imt * 8 + 2, 8);
  f->denom = (double *)calloc((size_t)imt * 4 + 2, 8);
  f->temp = (double *)calloc((size_t)imt * 4 + 2, 8);
  f->sprime = (double *)calloc(imt + 2, 8);
  f->indx = (int *)calloc((size_t)imt * 8 + 2, sizeof(int));
  /* the `if (first)` block, filtr.F:121-170 */
  static const double circle[5] = {0, 0.0, -1.0, 0.0, 1.0};
  int ibase = 0, jbase = 0;
  for (int im = 1; im <= imtp1; ++im) {
    const double fimr = 1.0 / (double)im;
    const int imm1 = im - 1;
    for (int i = 1; i <= imm1; ++i) f->denmsv[ibase + i] = 1.0 / (1.0 - cos(pi * (double)i * fimr));
    f->idbase[im] = ibase;
    ibase = ibase + imm1;
    const int imqc = (im - 1) / 2;
    for (int i = 1; i <= imqc; ++i) f->cossav[jbase + i] = cos(pi * (double)i * fimr);
    f->icbase[im] = jbase;
    jbase = jbase + imqc;
  }
  for (int im = 1; im <= imt; ++im) f->cosnpi[im] = circle[(im - 1) % 4 + 1];
  return f;
}
static void filtr_free(filtr_state *f) {
  free(f->cossav); free(f->denmsv); free(f->cosnpi); free(f->ftarr); free(f->icbase); free(f->idbase); free(f->cof);
  free(f->cosine); free(f->denom); free(f->temp); free(f->sprime); free(f->indx); free(f);
}

/* s(1:im) is s[1..im].  Returns 0, or 1 on the reference's `stop` conditions. */
static int filtr(filtr_state *f, double *s, int im, int mm, int n, int iss) {
  const int imt = f->imt, imtp1 = imt + 1;
  double *cof = f->cof, *cosine = f->cosine, *denom = f->denom, *temp = f->temp, *ftarr = f->ftarr, *sprime = f->sprime;
  int *indx = f->indx;
  if (im < 1 || mm < 1 || mm > 3 || n < 0 || iss < 0) return 1;
  if (mm == 2 && n == 0) {
    for (int i = 1; i <= im; ++i) s[i] = 0.0;
    return 0;
  }
  const int nmax = (mm == 1) ? n - 1 : n;
  const int nmaxp1 = nmax + 1;
  const double cc1 = 0.5 * (double)nmax + 0.25, cc2 = (double)nmax + 0.5;
  int lcy;
  double fnorm;
  if (mm == 2) {
    lcy = 2 * (im + 1);
    fnorm = 2.0 / (double)(im + 1);
  } else {
    lcy = 2 * im;
    fnorm = 2.0 / (double)im;
  }
  const int lh = lcy / 2, lhm1 = lh - 1, lqm = (lh - 1) / 2, lcyp1 = lcy + 1, imx4 = im * 4, imx8 = im * 8;
  double ssum = 0.0;
  for (int i = 1; i <= im; ++i) ssum = ssum + s[i];
  const double fim = (double)im, fimr = 1.0 / fim, stemp = ssum * fimr;
  if (!(n > 1 || mm != 1)) {
    for (int i = 1; i <= im; ++i) s[i] = stemp;
    return 0;
  }
  if (mm != 2)
    for (int i = 1; i <= im; ++i) s[i] = s[i] - stemp;
  if (iss == 0) { /* build the operator, filtr.F:226-390 */
    int jbase = f->icbase[lh];
    for (int i = 1; i <= lqm; ++i) cosine[i] = f->cossav[jbase + i];
    for (int i = 1; i <= lqm; ++i) cosine[lh - i] = -f->cossav[jbase + i];
    if (2 * (lqm + 1) == lh) cosine[lqm + 1] = 0.0;
    cosine[lh] = -1.0;
    for (int i = 1; i <= lh; ++i) cosine[lh + i] = -cosine[i];
    int ibase = f->idbase[lh];
    for (int i = 1; i <= lhm1; ++i) denom[i] = 0.25 * f->denmsv[ibase + i];
    denom[lh] = 0.125;
    for (int i = 1; i <= lhm1; ++i) temp[i] = denom[lh - i];
    for (int i = 1; i <= lhm1; ++i) denom[lh + i] = temp[i];
    denom[lcy] = 0.0;
    for (int i = lcyp1; i <= imx4; ++i) denom[i] = denom[i - lcy];
    double fact1, fact2;
    if (mm == 3) {
      fact1 = 2 * nmax; fact2 = 2 * nmaxp1;
    } else {
      fact1 = nmax; fact2 = nmaxp1;
    }
    for (int i = 1; i <= imx4; ++i) indx[i] = (int)(i * fact1);
    for (int i = 1; i <= imx4; ++i) indx[imx4 + i] = (int)(i * fact2);
    const int maxind = (int)(imx4 * fact2);
    const int ncyc = (maxind - 1) / lcy + 1;
    int maxndx = lcy;
    if (!(maxndx >= maxind)) {
      int npwr, found = 0;
      for (npwr = 1; npwr <= ncyc + 2; ++npwr) {
        maxndx = 2 * maxndx;
        if (maxndx >= maxind) { found = 1; break; }
      }
      if (!found) return 1;
      for (int np = 1; np <= npwr; ++np) {
        maxndx = maxndx / 2;
        for (int i = 1; i <= imx8; ++i)
          if (indx[i] > maxndx) indx[i] = indx[i] - maxndx;
      }
    }
    for (int j = 1; j <= imx8; ++j) cof[j] = cosine[indx[j]];
    const int ioff1 = lcy, ioff2 = lcy + imx4;
    if (mm == 1) {
      for (int j = 1; j <= im; ++j) {
        const int joff = (j - 1) * imt;
        for (int i = 1; i <= im; ++i)
          ftarr[joff + i] = (cof[i - j + ioff1] - cof[i - j + ioff2]) * denom[i - j + ioff1] +
                            (cof[i + j - 1] - cof[imx4 + i + j - 1]) * denom[i + j - 1] - 0.5;
      }
      for (int j = 1; j <= im; ++j) ftarr[j * imtp1 - imt] = ftarr[j * imtp1 - imt] + cc1;
    } else if (mm == 2) {
      for (int j = 1; j <= im; ++j) {
        const int joff = (j - 1) * imt;
        for (int i = 1; i <= im; ++i)
          ftarr[joff + i] = (cof[i - j + ioff1] - cof[i - j + ioff2]) * denom[i - j + ioff1] -
                            (cof[i + j] - cof[imx4 + i + j]) * denom[i + j];
      }
      for (int j = 1; j <= im; ++j) ftarr[j * imtp1 - imt] = ftarr[j * imtp1 - imt] + cc1;
    } else {
      const double genadj = (2 * n == im) ? 0.5 : 0.0;
      for (int j = 1; j <= im; ++j) {
        const int joff = (j - 1) * imt;
        for (int i = 1; i <= im; ++i)
          ftarr[joff + i] = (2.0 * (cof[i - j + ioff1] - cof[i - j + ioff2])) * denom[2 * i - 2 * j + ioff1] - 0.5 -
                            genadj * f->cosnpi[i] * f->cosnpi[j];
      }
      for (int j = 1; j <= im; ++j) ftarr[j * imtp1 - imt] = ftarr[j * imtp1 - imt] + cc2;
    }
  }
  /* apply, filtr.F:392-428 */
  for (int i = 1; i <= im; ++i) sprime[i] = 0.0;
  for (int i = 1; i <= im; ++i) {
    const int ioff = (i - 1) * imt;
    for (int j = 1; j <= im; ++j) sprime[j] = sprime[j] + s[i] * ftarr[ioff + j];
  }
  for (int i = 1; i <= im; ++i) sprime[i] = fnorm * sprime[i];
  if (mm == 2) {
    for (int i = 1; i <= im; ++i) s[i] = sprime[i];
    return 0;
  }
  double ssm = 0.0;
  for (int i = 1; i <= im; ++i) ssm = ssm + sprime[i];
  ssm = (ssum - ssm) * fimr;
  for (int i = 1; i <= im; ++i) s[i] = ssm + sprime[i];
  return 0;
}

/* ---- filt.F:36-115 on t(imt,km,jmt,nt) = t(:,:,:,:,taup1), rows js..je (joff = 0) ---------------- */
int orc_filt(double *t, int imt, int km, int jmt, int nt, const int *kmt, const double *cst, const double *cstr, double pi,
             int jfrst, int jft0, int jft1, int jft2, int lsegf, int jmtfil, const int *istf, const int *ietf, int js,
             int je) {
#define T(i, k, j, n) t[(size_t)((i)-1) + (size_t)imt * ((size_t)((k)-1) + (size_t)km * ((size_t)((j)-1) + (size_t)jmt * ((n)-1)))]
#define ISTF(jj, l, k) istf[(size_t)((jj)-1) + (size_t)jmtfil * ((size_t)((l)-1) + (size_t)lsegf * ((k)-1))]
#define IETF(jj, l, k) ietf[(size_t)((jj)-1) + (size_t)jmtfil * ((size_t)((l)-1) + (size_t)lsegf * ((k)-1))]
  const int imtm1 = imt - 1, imtm2 = imt - 2, jskpt = jft2 - jft1;
  filtr_state *f = filtr_new(imt, pi);
  double *tempik = (double *)calloc(imt + 2, 8);
  int rc = 0;
  /* setbcx of every row, filt.F:36-40 */
  for (int n = 1; n <= nt; ++n)
    for (int j = js; j <= je; ++j)
      for (int k = 1; k <= km; ++k) {
        T(1, k, j, n) = T(imtm1, k, j, n);
        T(imt, k, j, n) = T(2, k, j, n);
      }
  int m = 1, nn = 0; /* `m` and `n` of filt.F keep their values between strips */
  for (int j = js; j <= je && !rc; ++j) {
    const int jrow = j;
    if ((jrow > jft1 && jrow < jft2) || jrow < jfrst) continue;
    int jj = jrow - jfrst + 1;
    if (jrow >= jft2) jj = jj - jskpt + 1;
    int isave = 0, ieave = 0;
    for (int l = 1; l <= lsegf && !rc; ++l)
      for (int k = 1; k <= km && !rc; ++k) {
        if (ISTF(jj, l, k) != 0) {
          const int is = ISTF(jj, l, k), ie = IETF(jj, l, k);
          int iredo = 0;
          const int im = ie - is + 1;
          if (is != isave || ie != ieave) {
            iredo = -1;
            isave = is;
            ieave = ie;
            if (im != imtm2 || kmt[(size_t)(1 - 1) + (size_t)imt * (jrow - 1)] < k) {
              m = 1;
              nn = (int)lround(im * cst[jrow - 1] * cstr[jft0 - 1]);
            } else {
              m = 3;
              nn = (int)lround(im * cst[jrow - 1] * cstr[jft0 - 1] * 0.5);
            }
          }
          for (int mm = 1; mm <= nt; ++mm) {
            const int idx = iredo + mm, ism1 = is - 1;
            int iea = ie;
            if (ie >= imt) iea = imtm1;
            for (int i = is; i <= iea; ++i) tempik[i - ism1] = T(i, k, j, mm);
            int ieb = 0, ii = 0;
            if (ie >= imt) {
              ieb = ie - imtm2;
              ii = imtm1 - is;
              for (int i = 2; i <= ieb; ++i) tempik[i + ii] = T(i, k, j, mm);
            }
            if (filtr(f, tempik, im, m, nn, idx)) { rc = 1; break; }
            for (int i = is; i <= iea; ++i) T(i, k, j, mm) = tempik[i - ism1];
            if (ie >= imt)
              for (int i = 2; i <= ieb; ++i) T(i, k, j, mm) = tempik[i + ii];
          }
        }
      }
  }
  free(tempik);
  filtr_free(f);
  return rc;
}

/* ---- filuv.F:1-196 on u(:,:,:,1:2,taup1), rows js..je (joff = 0), O_fourfil O_cyclic ---------------
 * u1, u2 (imt,km,jmt) in place.  isuf, ieuf (jmtfil,lsegf,km) from orc_findex on kmu with jfu1, jfu2. */
int orc_filuv(double *u1, double *u2, int imt, int km, int jmt, const int *kmu, const double *csu, const double *csur,
              const double *phi, const double *spsin, const double *spcos, const double *dzt, const double *hr, double pi,
              int jfrst, int jfu0, int jfu1, int jfu2, int lsegf, int jmtfil, const int *isuf, const int *ieuf, int js,
              int je) {
#define U1(i, k, j) u1[(size_t)((i)-1) + (size_t)imt * ((size_t)((k)-1) + (size_t)km * ((j)-1))]
#define U2(i, k, j) u2[(size_t)((i)-1) + (size_t)imt * ((size_t)((k)-1) + (size_t)km * ((j)-1))]
#define ISUF(jj, l, k) isuf[(size_t)((jj)-1) + (size_t)jmtfil * ((size_t)((l)-1) + (size_t)lsegf * ((k)-1))]
#define IEUF(jj, l, k) ieuf[(size_t)((jj)-1) + (size_t)jmtfil * ((size_t)((l)-1) + (size_t)lsegf * ((k)-1))]
  const int imtm1 = imt - 1, imtm2 = imt - 2, jskpu = jfu2 - jfu1;
  filtr_state *f = filtr_new(imt, pi);
  double *t1 = (double *)calloc(imt + 2, 8), *t2 = (double *)calloc(imt + 2, 8);
  int rc = 0;
  /* filuv.F:45-49 */
  for (int j = js; j <= je; ++j)
    for (int k = 1; k <= km; ++k) {
      U1(1, k, j) = U1(imtm1, k, j); U1(imt, k, j) = U1(2, k, j);
      U2(1, k, j) = U2(imtm1, k, j); U2(imt, k, j) = U2(2, k, j);
    }
  int m = 2, n = 0;
  for (int j = js; j <= je && !rc; ++j) {
    const int jrow = j;
    if ((jrow > jfu1 && jrow < jfu2) || jrow < jfrst) continue;
    int jj = jrow - jfrst + 1;
    if (jrow >= jfu2) jj = jj - jskpu + 1;
    double fx = -1.0;
    if (phi[jrow - 1] > 0.0) fx = 1.0;
    int isave = 0, ieave = 0;
    for (int l = 1; l <= lsegf && !rc; ++l)
      for (int k = 1; k <= km && !rc; ++k) {
        if (ISUF(jj, l, k) == 0) continue;
        const int is = ISUF(jj, l, k), ie = IEUF(jj, l, k);
        int iredo = 1;
        const int im = ie - is + 1;
        if (is != isave || ie != ieave) {
          iredo = 0;
          isave = is;
          ieave = ie;
          if (im != imtm2) {
            m = 2;
            n = (int)lround(im * csu[jrow - 1] * csur[jfu0 - 1]);
          } else {
            m = 3;
            n = (int)lround(im * csu[jrow - 1] * csur[jfu0 - 1] * 0.5);
          }
        }
        const int ism1 = is - 1;
        int iea = ie;
        if (ie >= imt) iea = imtm1;
        for (int i = is; i <= iea; ++i) {
          t1[i - ism1] = -fx * U1(i, k, j) * spsin[i - 1] - U2(i, k, j) * spcos[i - 1];
          t2[i - ism1] = fx * U1(i, k, j) * spcos[i - 1] - U2(i, k, j) * spsin[i - 1];
        }
        int ieb = 0, ii = 0;
        if (ie >= imt) {
          ieb = ie - imtm2;
          ii = imtm1 - is;
          for (int i = 2; i <= ieb; ++i) {
            t1[i + ii] = -fx * U1(i, k, j) * spsin[i - 1] - U2(i, k, j) * spcos[i - 1];
            t2[i + ii] = fx * U1(i, k, j) * spcos[i - 1] - U2(i, k, j) * spsin[i - 1];
          }
        }
        if (filtr(f, t1, im, m, n, iredo) || filtr(f, t2, im, m, n, 1)) { rc = 1; break; }
        for (int i = is; i <= iea; ++i) {
          U1(i, k, j) = fx * (-t1[i - ism1] * spsin[i - 1] + t2[i - ism1] * spcos[i - 1]);
          U2(i, k, j) = -t1[i - ism1] * spcos[i - 1] - t2[i - ism1] * spsin[i - 1];
        }
        if (ie >= imt)
          for (int i = 2; i <= ieb; ++i) {
            U1(i, k, j) = fx * (-t1[i + ii] * spsin[i - 1] + t2[i + ii] * spcos[i - 1]);
            U2(i, k, j) = -t1[i + ii] * spcos[i - 1] - t2[i + ii] * spsin[i - 1];
          }
      }
    if (isave != 0 && ieave != 0) { /* filuv.F:155-181: the vertical mean again, then the mask */
      for (int i = 1; i <= imt; ++i) { t1[i] = 0.0; t2[i] = 0.0; }
      for (int k = 1; k <= km; ++k)
        for (int i = 1; i <= imt; ++i) {
          t1[i] = t1[i] + U1(i, k, j) * dzt[k - 1];
          t2[i] = t2[i] + U2(i, k, j) * dzt[k - 1];
        }
      for (int i = 1; i <= imt; ++i) {
        t1[i] = t1[i] * hr[(size_t)(i - 1) + (size_t)imt * (jrow - 1)];
        t2[i] = t2[i] * hr[(size_t)(i - 1) + (size_t)imt * (jrow - 1)];
      }
      for (int k = 1; k <= km; ++k)
        for (int i = 1; i <= imt; ++i) {
          U1(i, k, j) = U1(i, k, j) - t1[i];
          U2(i, k, j) = U2(i, k, j) - t2[i];
        }
      for (int k = 1; k <= km; ++k)
        for (int i = 1; i <= imt; ++i) {
          const double mask = (kmu[(size_t)(i - 1) + (size_t)imt * (jrow - 1)] >= k) ? 1.0 : 0.0;
          U1(i, k, j) = U1(i, k, j) * mask;
          U2(i, k, j) = U2(i, k, j) * mask;
        }
    }
  }
  /* clinic.F:506-509 */
  for (int j = js; j <= je; ++j)
    for (int k = 1; k <= km; ++k) {
      U1(1, k, j) = U1(imtm1, k, j); U1(imt, k, j) = U1(2, k, j);
      U2(1, k, j) = U2(imtm1, k, j); U2(imt, k, j) = U2(2, k, j);
    }
  free(t1); free(t2);
  filtr_free(f);
  return rc;
}
