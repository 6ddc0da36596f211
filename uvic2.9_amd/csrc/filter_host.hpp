// filter_host.hpp -- set-up of the polar Fourier filter of the tracers (host side of the library).
//
// Replaces what the reference does once at start-up and, redundantly, inside every call:
//   /root/reference/source/common/findex.F:1-101   ocean strips of every filtered row and level
//   /root/reference/source/common/filt.F:48-83     which filter (m, n) a strip gets; strips of equal
//                                                  extent that follow one another share the operator
//   /root/reference/source/common/filtr.F:121-390  the operator: an (im x im) matrix from tabulated cosines
// The strips and operators depend on kmt and the grid only, so they are built once by
// uvic_gpu_set_filter and kept on the device; k_filt then applies s' = F s per strip, level and tracer
// (filtr.F:392-428).  Same expressions and summation order as the reference: bit-identical.
#ifndef UVIC_FILTER_HOST_HPP
#define UVIC_FILTER_HOST_HPP

#include <cmath>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "filter_item.h"

struct FilterSetup {
  std::vector<FilterItem> items;
  std::vector<double> mats;
};

namespace uvic_filter {

// the tables of filtr.F's `if (first)` block for strip lengths up to imt+1
struct Tables {
  std::vector<double> cossav, denmsv, cosnpi;
  std::vector<int> icbase, idbase;
  Tables(int imt, double pi) {
    const int imtp1 = imt + 1, imtd2 = imt / 2;
    cossav.assign((size_t)imtd2 * (imt - imtd2) + imt + 2, 0.0);
    denmsv.assign((size_t)imt * imtp1 / 2 + imt + 2, 0.0);
    cosnpi.assign(imt + 2, 0.0);
    icbase.assign(imtp1 + 2, 0);
    idbase.assign(imtp1 + 2, 0);
    static const double circle[5] = {0, 0.0, -1.0, 0.0, 1.0};
    int ibase = 0, jbase = 0;
    for (int im = 1; im <= imtp1; ++im) {
      const double fimr = 1.0 / (double)im;
      for (int i = 1; i <= im - 1; ++i) denmsv[ibase + i] = 1.0 / (1.0 - std::cos(pi * (double)i * fimr));
      idbase[im] = ibase;
      ibase += im - 1;
      const int imqc = (im - 1) / 2;
      for (int i = 1; i <= imqc; ++i) cossav[jbase + i] = std::cos(pi * (double)i * fimr);
      icbase[im] = jbase;
      jbase += imqc;
    }
    for (int im = 1; im <= imt; ++im) cosnpi[im] = circle[(im - 1) % 4 + 1];
  }
};

// filtr.F:226-390 for mm = 1, 2 or 3; F is compact: F[(i-1)*im + (j-1)] = ftarr((i-1)*imt + j)
inline bool build_operator(const Tables &T, int im, int mm, int n, std::vector<double> &F) {
  const int nmax = (mm == 1) ? n - 1 : n, nmaxp1 = nmax + 1;
  const double cc1 = 0.5 * (double)nmax + 0.25, cc2 = (double)nmax + 0.5;
  const int lcy = (mm == 2) ? 2 * (im + 1) : 2 * im, lh = lcy / 2, lhm1 = lh - 1, lqm = (lh - 1) / 2, lcyp1 = lcy + 1, imx4 = im * 4, imx8 = im * 8;
  std::vector<double> cosine(imx8 + 2, 0.0), denom(imx4 + 2, 0.0), temp(imx4 + 2, 0.0), cof(imx8 + 2, 0.0);
  std::vector<int> indx(imx8 + 2, 0);
  const int jbase = T.icbase[lh];
  for (int i = 1; i <= lqm; ++i) cosine[i] = T.cossav[jbase + i];
  for (int i = 1; i <= lqm; ++i) cosine[lh - i] = -T.cossav[jbase + i];
  if (2 * (lqm + 1) == lh) cosine[lqm + 1] = 0.0;
  cosine[lh] = -1.0;
  for (int i = 1; i <= lh; ++i) cosine[lh + i] = -cosine[i];
  const int ibase = T.idbase[lh];
  for (int i = 1; i <= lhm1; ++i) denom[i] = 0.25 * T.denmsv[ibase + i];
  denom[lh] = 0.125;
  for (int i = 1; i <= lhm1; ++i) temp[i] = denom[lh - i];
  for (int i = 1; i <= lhm1; ++i) denom[lh + i] = temp[i];
  denom[lcy] = 0.0;
  for (int i = lcyp1; i <= imx4; ++i) denom[i] = denom[i - lcy];
  const double fact1 = (mm == 3) ? 2 * nmax : nmax, fact2 = (mm == 3) ? 2 * nmaxp1 : nmaxp1;
  for (int i = 1; i <= imx4; ++i) indx[i] = (int)(i * fact1);
  for (int i = 1; i <= imx4; ++i) indx[imx4 + i] = (int)(i * fact2);
  const int maxind = (int)(imx4 * fact2), ncyc = (maxind - 1) / lcy + 1;
  int maxndx = lcy;
  if (!(maxndx >= maxind)) {
    int npwr;
    bool found = false;
    for (npwr = 1; npwr <= ncyc + 2; ++npwr) {
      maxndx = 2 * maxndx;
      if (maxndx >= maxind) { found = true; break; }
    }
    if (!found) return false;
    for (int np = 1; np <= npwr; ++np) {
      maxndx = maxndx / 2;
      for (int i = 1; i <= imx8; ++i)
        if (indx[i] > maxndx) indx[i] -= maxndx;
    }
  }
  for (int j = 1; j <= imx8; ++j) {
    if (indx[j] < 1 || indx[j] > imx8) return false;   // the reference would read outside `cosine` here
    cof[j] = cosine[indx[j]];
  }
  const int ioff1 = lcy, ioff2 = lcy + imx4;
  F.assign((size_t)im * im, 0.0);
#define FT(jrow, icol) F[(size_t)((jrow)-1) * im + ((icol)-1)]   /* ftarr((jrow-1)*imt + icol) */
  if (mm == 1) {
    for (int j = 1; j <= im; ++j)
      for (int i = 1; i <= im; ++i)
        FT(j, i) = (cof[i - j + ioff1] - cof[i - j + ioff2]) * denom[i - j + ioff1] +
                   (cof[i + j - 1] - cof[imx4 + i + j - 1]) * denom[i + j - 1] - 0.5;
    for (int j = 1; j <= im; ++j) FT(j, j) = FT(j, j) + cc1;
  } else if (mm == 2) {
    for (int j = 1; j <= im; ++j)
      for (int i = 1; i <= im; ++i)
        FT(j, i) = (cof[i - j + ioff1] - cof[i - j + ioff2]) * denom[i - j + ioff1] - (cof[i + j] - cof[imx4 + i + j]) * denom[i + j];
    for (int j = 1; j <= im; ++j) FT(j, j) = FT(j, j) + cc1;
  } else {
    const double genadj = (2 * n == im) ? 0.5 : 0.0;
    for (int j = 1; j <= im; ++j)
      for (int i = 1; i <= im; ++i)
        FT(j, i) = (2.0 * (cof[i - j + ioff1] - cof[i - j + ioff2])) * denom[2 * i - 2 * j + ioff1] - 0.5 -
                   genadj * T.cosnpi[i] * T.cosnpi[j];
    for (int j = 1; j <= im; ++j) FT(j, j) = FT(j, j) + cc2;
  }
#undef FT
  return true;
}

}  // namespace uvic_filter

// findex.F:22-66 (O_cyclic) for one row: ocean strips of every level, kept per (l,k); kxx = kmt or kmu
inline bool find_strips(int imt, int km, const int *kxx, int jrow, int lsegf, std::vector<int> &isf, std::vector<int> &ief) {
#define KXX(i, j) kxx[(size_t)((i)-1) + (size_t)imt * ((j)-1)]
  const int imax = imt;
  std::vector<int> iis(lsegf + 2), iie(lsegf + 2);
  isf.assign((size_t)lsegf * km, 0);
  ief.assign((size_t)lsegf * km, 0);
  for (int k = 1; k <= km; ++k) {
    for (int l = 1; l <= lsegf + 1; ++l) { iis[l] = 0; iie[l] = 0; }
    int l = 1;
    if (KXX(2, jrow) >= k) iis[1] = 2;
    for (int i = 2; i <= imax - 1; ++i) {
      if (l > lsegf + 1) return false;
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
    if (KXX(imax - 1, jrow) >= k && KXX(imax, jrow) >= k) {
      if (l > lsegf + 1) return false;
      iie[l] = imax - 1;
      l = l + 1;
    }
    int lm = l - 1;
    if (lm > 1 && iis[1] == 2 && iie[lm] == imax - 1 && KXX(1, jrow) >= k) {
      iis[1] = iis[lm];
      iie[1] = iie[1] + imax - 2;
      iis[lm] = 0;
      iie[lm] = 0;
      lm = lm - 1;
    }
    if (lm > lsegf) return false;
    for (l = 1; l <= lsegf; ++l) {
      isf[(size_t)(l - 1) * km + (k - 1)] = iis[l];
      ief[(size_t)(l - 1) * km + (k - 1)] = iie[l];
    }
  }
#undef KXX
  return true;
}

// the velocity filter: strips of kmu for rows jfrst..jmt-1 outside (jfu1, jfu2), filter type and wavenumber as
// filuv.F:69-98 (type 2 for a strip between coasts, type 3 for a full circle), operators, and the rows that hold a strip
inline int filter_build_u(int imt, int jmt, int km, const int *kmu, const double *csu, const double *csur, const double *phi,
                          double pi, int jfrst, int jfu0, int jfu1, int jfu2, int lsegf, FilterSetup &out, std::vector<int> &rows,
                          std::string &err) {
  using namespace uvic_filter;
  if (jfrst < 2 || jfu0 < 1 || jfu0 > jmt || jfu1 >= jfu2 || jfu2 > jmt) { err = "uvic_gpu_set_filter_u: rows out of range"; return 2; }
  const int imtm2 = imt - 2;
  Tables T(imt, pi);
  std::map<std::tuple<int, int, int>, int> known;
  out.items.clear();
  out.mats.clear();
  rows.clear();
  std::vector<int> isf, ief;
  for (int jrow = jfrst; jrow <= jmt - 1; ++jrow) {
    if (!(jrow <= jfu1 || jrow >= jfu2)) continue;
    if (!find_strips(imt, km, kmu, jrow, lsegf, isf, ief)) { err = "uvic_gpu_set_filter_u: more ocean strips in a row than lsegf"; return 2; }
    const double fx = (phi[jrow - 1] > 0.0) ? 1.0 : -1.0;
    int isave = 0, ieave = 0, m = 2, n = 0;
    for (int l = 1; l <= lsegf; ++l)
      for (int k = 1; k <= km; ++k) {
        const int is = isf[(size_t)(l - 1) * km + (k - 1)], ie = ief[(size_t)(l - 1) * km + (k - 1)];
        if (is == 0) continue;
        const int im = ie - is + 1;
        if (is != isave || ie != ieave) {
          isave = is;
          ieave = ie;
          if (im != imtm2) {
            m = 2;
            n = (int)std::lround(im * csu[jrow - 1] * csur[jfu0 - 1]);
          } else {
            m = 3;
            n = (int)std::lround(im * csu[jrow - 1] * csur[jfu0 - 1] * 0.5);
          }
        }
        if (im < 1 || n < 0) { err = "uvic_gpu_set_filter_u: bad strip (filtr would stop)"; return 2; }
        FilterItem it;
        it.j = jrow; it.k = k; it.is = is; it.im = im;
        it.fnorm = (m == 2) ? 2.0 / (double)(im + 1) : 2.0 / (double)im;
        it.fimr = 1.0 / (double)im;
        it.fx = fx;
        it.mat = 0;
        if (m == 2 && n == 0) {
          it.mode = 2;
        } else {
          it.mode = (m == 2) ? 3 : 1;
          const auto key = std::make_tuple(im, m, n);
          auto f = known.find(key);
          if (f == known.end()) {
            std::vector<double> F;
            if (!build_operator(T, im, m, n, F)) { err = "uvic_gpu_set_filter_u: cannot build the filter operator"; return 2; }
            const int off = (int)out.mats.size();
            out.mats.insert(out.mats.end(), F.begin(), F.end());
            f = known.emplace(key, off).first;
          }
          it.mat = f->second;
        }
        out.items.push_back(it);
      }
    if (isave != 0 && ieave != 0) rows.push_back(jrow);
  }
  return 0;
}

// strips (findex.F, O_cyclic), their filters (filt.F:48-83) and operators for rows jfrst..jmt-1 outside (jft1, jft2)
inline int filter_build(int imt, int jmt, int km, const int *kmt, const double *cst, const double *cstr, double pi, int jfrst,
                        int jft0, int jft1, int jft2, int lsegf, FilterSetup &out, std::string &err) {
  using namespace uvic_filter;
  if (jfrst < 2 || jft0 < 1 || jft0 > jmt || jft1 >= jft2 || jft2 > jmt) { err = "uvic_gpu_set_filter: rows out of range"; return 2; }
#define KXX(i, j) kmt[(size_t)((i)-1) + (size_t)imt * ((j)-1)]
  const int imax = imt, imtm2 = imt - 2;
  Tables T(imt, pi);
  std::map<std::tuple<int, int, int>, int> known;   // (im, m, n) -> operator offset
  out.items.clear();
  out.mats.clear();
  std::vector<int> iis(lsegf + 2), iie(lsegf + 2);
  for (int jrow = jfrst; jrow <= jmt - 1; ++jrow) {
    if (!(jrow <= jft1 || jrow >= jft2)) continue;
    // findex.F:22-66 for every level, kept per (l,k)
    std::vector<int> isf((size_t)lsegf * km, 0), ief((size_t)lsegf * km, 0);
    for (int k = 1; k <= km; ++k) {
      for (int l = 1; l <= lsegf + 1; ++l) { iis[l] = 0; iie[l] = 0; }
      int l = 1;
      if (KXX(2, jrow) >= k) iis[1] = 2;
      for (int i = 2; i <= imax - 1; ++i) {
        if (l > lsegf + 1) { err = "uvic_gpu_set_filter: more ocean strips in a row than lsegf"; return 2; }
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
      if (KXX(imax - 1, jrow) >= k && KXX(imax, jrow) >= k) {
        if (l > lsegf + 1) { err = "uvic_gpu_set_filter: more ocean strips in a row than lsegf"; return 2; }
        iie[l] = imax - 1;
        l = l + 1;
      }
      int lm = l - 1;
      if (lm > 1 && iis[1] == 2 && iie[lm] == imax - 1 && KXX(1, jrow) >= k) {
        iis[1] = iis[lm];
        iie[1] = iie[1] + imax - 2;
        iis[lm] = 0;
        iie[lm] = 0;
        lm = lm - 1;
      }
      if (lm > lsegf) { err = "uvic_gpu_set_filter: more ocean strips in a row than lsegf"; return 2; }
      for (l = 1; l <= lsegf; ++l) {
        isf[(size_t)(l - 1) * km + (k - 1)] = iis[l];
        ief[(size_t)(l - 1) * km + (k - 1)] = iie[l];
      }
    }
    // filt.F:56-83: strips in the order (l, k); m and n are recomputed only when the extent changes
    int isave = 0, ieave = 0, m = 1, n = 0;
    for (int l = 1; l <= lsegf; ++l)
      for (int k = 1; k <= km; ++k) {
        const int is = isf[(size_t)(l - 1) * km + (k - 1)], ie = ief[(size_t)(l - 1) * km + (k - 1)];
        if (is == 0) continue;
        const int im = ie - is + 1;
        if (is != isave || ie != ieave) {
          isave = is;
          ieave = ie;
          if (im != imtm2 || KXX(1, jrow) < k) {
            m = 1;
            n = (int)std::lround(im * cst[jrow - 1] * cstr[jft0 - 1]);
          } else {
            m = 3;
            n = (int)std::lround(im * cst[jrow - 1] * cstr[jft0 - 1] * 0.5);
          }
        }
        if (im < 1 || n < 0) { err = "uvic_gpu_set_filter: bad strip (filtr would stop)"; return 2; }
        FilterItem it;
        it.j = jrow; it.k = k; it.is = is; it.im = im;
        it.fnorm = 2.0 / (double)im;
        it.fimr = 1.0 / (double)im;
        it.fx = 0.0;
        it.mat = 0;
        if (!(n > 1 || m != 1)) {
          it.mode = 0;
        } else {
          it.mode = 1;
          const auto key = std::make_tuple(im, m, n);
          auto f = known.find(key);
          if (f == known.end()) {
            std::vector<double> F;
            if (!build_operator(T, im, m, n, F)) { err = "uvic_gpu_set_filter: cannot build the filter operator"; return 2; }
            const int off = (int)out.mats.size();
            out.mats.insert(out.mats.end(), F.begin(), F.end());
            f = known.emplace(key, off).first;
          }
          it.mat = f->second;
        }
        out.items.push_back(it);
      }
  }
#undef KXX
  return 0;
}
#endif
