// kernels_fct.hpp -- flux-corrected-transport advection, isopycnal diffusion,
// explicit update and implicit vertical solve of one (tracer, latitude row,
// longitude chunk) tile per workgroup.
//
// Replaces, for all tracers in ONE launch each,
//   /root/reference/updates/09/source/mom/tracer_adv_flx.F:381-1028  (adv_flux, FCT)
//   /root/reference/updates/09/source/mom/tracer.F:925-1032          (background diffusive fluxes)
//   /root/reference/updates/09/source/mom/isopyc.F:923-1137          (isoflux)
//   /root/reference/updates/09/source/mom/tracer.F:1053-1130 with source/mom/fdift.h:25-88
//   /root/reference/source/mom/invtri.F:1-115 (through ivdift, tracer.F:1938-2032)
//
// The reference's j loop of adv_flux is not a recurrence: row
// r needs the y-limiter ratios R+-Y of rows r-1, r, r+1 only.  It is split into
//   fct_rows_block   : per row: low-order fluxes, low-order solution t_lo, raw
//                      antidiffusive fluxes, all six limiter ratios, the limited
//                      x and z fluxes and their divergences ADV_Tx, ADV_Tz; the
//                      ratios R+-Y go to global memory,
//   update_rows_block: per row: limited y fluxes from R+-Y(r-1..r+1), ADV_Ty,
//                      all diffusive fluxes, the explicit update and the
//                      tridiagonal solve, writing t(tau+1).
// A workgroup owns an (i,k) tile of one row: W = (owned columns + 2x2 halo)
// columns times km levels, i contiguous, staged in LDS; neighbours in i and k
// come from LDS, rows r-1 and r+1 are read at the thread's own (i,k) from
// global memory (coalesced along i).  Longitude is cyclic: tile column l maps to
// model column wrap(i0-2+l) in 2..imt-1, so the stored cyclic images (columns 1
// and imt) of the inputs are never read; the kernels write them on output.
// Every expression keeps the reference's evaluation order (-ffp-contract=off).
#ifndef UVIC_KERNELS_FCT_HPP
#define UVIC_KERNELS_FCT_HPP

#include "kernels_isopyc.hpp"

namespace uvic {

struct Tile {
  int i0, i1;  // owned model columns, 2 <= i0 <= i1 <= imt-1
  int W;       // tile width = i1-i0+1+4
  int imt, km;
  UVIC_DEV int gi(int l) const {  // model column of tile column l (cyclic)
    int x = i0 - 2 + l - 2;
    const int p = imt - 2;
    x %= p;
    if (x < 0) x += p;
    return x + 2;
  }
};

UVIC_DEV Tile make_tile(const uvic_ctx &c, int chunk, int nchunk) {
  Tile t;
  const int owned = c.imt - 2;
  const int per = (owned + nchunk - 1) / nchunk;
  t.i0 = 2 + chunk * per;
  t.i1 = imin(t.i0 + per - 1, c.imt - 1);
  t.W = t.i1 - t.i0 + 1 + 4;
  t.imt = c.imt;
  t.km = c.km;
  return t;
}

// LDS doubles needed by the two block routines for a tile of width W
UVIC_DEV size_t fct_lds_doubles(int W, int km) { return (size_t)W * km * 6 + (size_t)W * (km + 1) * 2; }
UVIC_DEV size_t upd_lds_doubles(int W, int km) { return (size_t)W * km * 3 + (size_t)W * (km + 1) * 2; }

#define LC(l, k) ((size_t)(l) + (size_t)W * ((k)-1))  /* cell (l, k=1..km) */
#define LF(l, kf) ((size_t)(l) + (size_t)W * (kf))    /* face (l, kf=0..km) */

// ===========================================================================
// fct_rows_block: FCT quantities of row `r` (2 <= r <= jmt-1) for tracer n1
// (1-based).  Outputs: c.adv_x, c.adv_z (owned columns), c.RpY, c.RmY.
// ===========================================================================
template <class Env>
UVIC_DEV void fct_rows_block(Env &env, const uvic_ctx &c, int n1, int r, int chunk, int nchunk, double *lds) {
  UV_DIMS(c);
  const Tile T = make_tile(c, chunk, nchunk);
  const int W = T.W;
  const int NCW = W * km, NFW = W * (km + 1);
  const int nth = env.nthreads();
  const double *tm = c.t_taum1 + (size_t)(n1 - 1) * N3;
  const double *tt = c.t_tau + (size_t)(n1 - 1) * N3;
  double *s_tm = lds;            // t(tau-1) row r        (later: limiter ratio R+)
  double *s_tt = s_tm + NCW;     // t(tau)   row r
  double *s_felo = s_tt + NCW;   // low-order east flux
  double *s_tlo = s_felo + NCW;  // low-order solution
  double *s_afe = s_tlo + NCW;   // antidiffusive east flux (raw, then limited+low)
  double *s_Rm = s_afe + NCW;    // limiter ratio R-
  double *s_fblo = s_Rm + NCW;   // low-order bottom flux, faces 0..km
  double *s_afb = s_fblo + NFW;  // antidiffusive bottom flux, faces 0..km
  double *s_Rp = s_tm;
  const double c2dtts = c.c2dtts;
  const double cstr_r = c.cstr[r - 1];

  // P1: stage rows of both time levels ---------------------------------------
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      const int i = T.gi(l);
      s_tm[q] = tm[X3(i, k, r)];
      s_tt[q] = tt[X3(i, k, r)];
    }
  });
  // P2: low-order (upstream) fluxes through east and bottom faces, adv_flx:517-546
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      const int i = T.gi(l);
      if (l <= W - 2) {
        const double totadv = c.tot_e[X3(i, k, r)];
        const double a = s_tm[LC(l, k)], b = s_tm[LC(l + 1, k)];
        s_felo[q] = totadv * (a + b) + dabs(totadv) * (a - b);
      }
      if (k <= km - 1) {
        const double totadv = c.tot_b[XF(i, k, r)];
        const double up = s_tm[LC(l, k)], dn = s_tm[LC(l, k + 1)];
        s_fblo[LF(l, k)] = totadv * (dn + up) + dabs(totadv) * (dn - up);
      } else {
        s_fblo[LF(l, km)] = 0.0;
      }
      if (k == 1) s_fblo[LF(l, 0)] = c.adv_vbt[XF(i, 0, r)] * 2.0 * s_tm[LC(l, 1)];
    }
  });
  // P3: low-order solution (adv_flx:563-579) and raw antidiffusive fluxes (:586-619)
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      const int i = T.gi(l);
      const double tmask_c = TMASK(i, k, r);
      if (l >= 1 && l <= W - 2) {
        const double t_c = s_tm[q];
        const double t_s = tm[X3(i, k, r - 1)], t_n = tm[X3(i, k, r + 1)];
        const double vn = c.tot_n[X3(i, k, r)], vs = c.tot_n[X3(i, k, r - 1)];
        const double fn_n = vn * (t_c + t_n) + dabs(vn) * (t_c - t_n);  // adv_fn(i,k,r),   :501-514
        const double fn_s = vs * (t_s + t_c) + dabs(vs) * (t_s - t_c);  // adv_fn(i,k,r-1)
        const double twodt = c2dtts * c.dtxcel[k - 1];
        const double cstdxt2r = cstr_r * c.dxtr[i - 1] * 0.5;
        const double advx = (s_felo[LC(l, k)] - s_felo[LC(l - 1, k)]) * cstdxt2r;
        const double advy = (fn_n - fn_s) * c.cstdyt2r[r - 1];
        const double advz = (s_fblo[LF(l, k - 1)] - s_fblo[LF(l, k)]) * c.dzt2r[k - 1];
        s_tlo[q] = (t_c - twodt * (advx + advy + advz) * tmask_c);
        if (k <= km - 1) {
          const double totadv = c.tot_b[XF(i, k, r)];
          s_afb[LF(l, k)] = totadv * (s_tt[LC(l, k)] + s_tt[LC(l, k + 1)]) - s_fblo[LF(l, k)] * tmask_c;
        } else {
          s_afb[LF(l, km)] = 0.0;
        }
        if (k == 1) s_afb[LF(l, 0)] = c.adv_vbt[XF(i, 0, r)] * 2.0 * s_tm[LC(l, 1)];
      }
      if (l <= W - 2) {
        const double totadv = c.tot_e[X3(i, k, r)];
        s_afe[q] = totadv * (s_tt[LC(l, k)] + s_tt[LC(l + 1, k)]) - s_felo[q];
      }
    }
  });
  // P4: limiter ratios in x (adv_flx:638-690) into s_Rp/s_Rm, and in y (:717-757) to global
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      if (l < 1 || l > W - 2) continue;
      const int i = T.gi(l), iw = T.gi(l - 1), ie = T.gi(l + 1);
      const double tlo = s_tlo[q];
      const double tmask_c = TMASK(i, k, r);
      const double tc = s_tt[q];
      {
        const double mw = 0.5 * (s_tt[LC(l - 1, k)] + tc);  // running mean of adjacent points
        const double me = 0.5 * (tc + s_tt[LC(l + 1, k)]);
        const double tmw = TMASK(iw, k, r), tme = TMASK(ie, k, r);
        const double fxa = tmw * mw + (1.0 - tmw) * tlo;
        const double fxb = tme * me + (1.0 - tme) * tlo;
        const double Trmax = dmax(dmax(fxa, fxb), tlo), Trmin = dmin(dmin(fxa, fxb), tlo);
        const double dcf = cstr_r * c.dxtr[i - 1] * 0.5;
        const double flxlft = s_afe[LC(l - 1, k)], flxrgt = s_afe[LC(l, k)];
        const double Pplus = c2dtts * dcf * (dmax(0.0, flxlft) - dmin(0.0, flxrgt));
        const double Pminus = c2dtts * dcf * (dmax(0.0, flxrgt) - dmin(0.0, flxlft));
        const double Qplus = Trmax - tlo, Qminus = tlo - Trmin;
        s_Rp[q] = dmin(1., tmask_c * Qplus / (Pplus + UV_EPSLN));
        s_Rm[q] = dmin(1., tmask_c * Qminus / (Pminus + UV_EPSLN));
      }
      if (l >= 2 && l <= W - 3) {  // y ratios only for owned columns
        const double t_s = tt[X3(i, k, r - 1)], t_n = tt[X3(i, k, r + 1)];
        const double tms = TMASK(i, k, r - 1), tmn = TMASK(i, k, r + 1);
        const double fxa = 0.5 * tms * (t_s + tc) + (1.0 - tms) * tlo;
        const double fxb = 0.5 * tmn * (tc + t_n) + (1.0 - tmn) * tlo;
        const double Trmax = dmax(dmax(fxa, fxb), tlo), Trmin = dmin(dmin(fxa, fxb), tlo);
        const double dcf = c.cstdyt2r[r - 1];
        // raw antidiffusive north fluxes of rows r-1 and r (row 1 is zero, adv_flx:467-481)
        const double m_c = tm[X3(i, k, r)], m_s = tm[X3(i, k, r - 1)], m_n = tm[X3(i, k, r + 1)];
        const double vn = c.tot_n[X3(i, k, r)], vs = c.tot_n[X3(i, k, r - 1)];
        const double lo_n = vn * (m_c + m_n) + dabs(vn) * (m_c - m_n);
        const double lo_s = vs * (m_s + m_c) + dabs(vs) * (m_s - m_c);
        const double flxrgt = vn * (tc + t_n) - lo_n;
        const double flxlft = (r - 1 == 1) ? 0.0 : vs * (t_s + tc) - lo_s;
        const double Pplus = c2dtts * dcf * (dmax(0.0, flxlft) - dmin(0.0, flxrgt));
        const double Pminus = c2dtts * dcf * (dmax(0.0, flxrgt) - dmin(0.0, flxlft));
        const double Qplus = Trmax - tlo, Qminus = tlo - Trmin;
        const size_t g = X3(i, k, r) + (size_t)(n1 - 1 - c.n0) * N3;
        c.RpY[g] = dmin(1., tmask_c * Qplus / (Pplus + UV_EPSLN));
        c.RmY[g] = dmin(1., tmask_c * Qminus / (Pminus + UV_EPSLN));
      }
    }
  });
  // P5: apply the x delimiter and add the low-order flux (adv_flx:695-711, :989-992)
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      if (l < 1 || l > W - 3) continue;
      const double Cpos = dmin(s_Rp[LC(l + 1, k)], s_Rm[LC(l, k)]);
      const double Cneg = dmin(s_Rp[LC(l, k)], s_Rm[LC(l + 1, k)]);
      const double f = s_afe[q];
      s_afe[q] = 0.5 * ((Cpos + Cneg) * f + (Cpos - Cneg) * dabs(f)) + s_felo[q];
    }
  });
  // P6: ADV_Tx to global; limiter ratios in z (adv_flx:789-852) overwrite s_Rp/s_Rm
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      if (l < 2 || l > W - 3) continue;
      const int i = T.gi(l);
      const double cstdxt2r = cstr_r * c.dxtr[i - 1] * 0.5;
      c.adv_x[X3(i, k, r) + (size_t)(n1 - 1 - c.n0) * N3] = (s_afe[LC(l, k)] - s_afe[LC(l - 1, k)]) * cstdxt2r;
    }
  });
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      if (l < 2 || l > W - 3) continue;
      const int i = T.gi(l);
      const double tlo = s_tlo[q], tc = s_tt[q];
      const double tmask_c = TMASK(i, k, r);
      double fxa, fxb;
      if (k > 1) {
        const double tmu = TMASK(i, k - 1, r);
        fxa = 0.5 * tmu * (s_tt[LC(l, k - 1)] + tc) + (1.0 - tmu) * tlo;
      } else {
        fxa = tlo;
      }
      if (k < km) {
        const double tmd = TMASK(i, k + 1, r);
        fxb = 0.5 * tmd * (tc + s_tt[LC(l, k + 1)]) + (1.0 - tmd) * tlo;
      } else {
        fxb = tlo;
      }
      const double Trmax = dmax(dmax(fxa, fxb), tlo), Trmin = dmin(dmin(fxa, fxb), tlo);
      const double dcf = c.dzt2r[k - 1];
      const double flxlft = s_afb[LF(l, k)], flxrgt = s_afb[LF(l, k - 1)];
      const double Pplus = c2dtts * dcf * (dmax(0.0, flxlft) - dmin(0.0, flxrgt));
      const double Pminus = c2dtts * dcf * (dmax(0.0, flxrgt) - dmin(0.0, flxlft));
      const double Qplus = Trmax - tlo, Qminus = tlo - Trmin;
      s_Rp[q] = dmin(1., tmask_c * Qplus / (Pplus + UV_EPSLN));
      s_Rm[q] = dmin(1., tmask_c * Qminus / (Pminus + UV_EPSLN));
    }
  });
  // P7: apply the z delimiter, add low order, mask (adv_flx:857-887, :994-999);
  //     surface and bottom faces as tracer.F:1063-1065
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      if (l < 2 || l > W - 3) continue;
      const int i = T.gi(l);
      if (k <= km - 1) {
        const double Cneg = dmin(s_Rp[LC(l, k + 1)], s_Rm[LC(l, k)]);
        const double Cpos = dmin(s_Rp[LC(l, k)], s_Rm[LC(l, k + 1)]);
        const double f = s_afb[LF(l, k)];
        const double lim = 0.5 * ((Cpos + Cneg) * f + (Cpos - Cneg) * dabs(f));
        s_afb[LF(l, k)] = (lim + s_fblo[LF(l, k)]) * TMASK(i, k, r);
      } else {
        s_afb[LF(l, km)] = c.adv_vbt[XF(i, km, r)] * s_tt[LC(l, km)];
      }
      if (k == 1) s_afb[LF(l, 0)] = c.adv_vbt[XF(i, 0, r)] * (s_tt[LC(l, 1)] + s_tt[LC(l, 1)]);
    }
  });
  // P8: ADV_Tz to global (fdift.h:39)
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      if (l < 2 || l > W - 3) continue;
      const int i = T.gi(l);
      c.adv_z[X3(i, k, r) + (size_t)(n1 - 1 - c.n0) * N3] = (s_afb[LF(l, k - 1)] - s_afb[LF(l, k)]) * c.dzt2r[k - 1];
    }
  });
}

// final (limited + low-order, masked) north flux of row rr for one cell; rr in 1..jmt-1
UVIC_DEV double fct_north_flux(const uvic_ctx &c, const double *tm, const double *tt, const double *RpY,
                               const double *RmY, int i, int k, int rr) {
  UV_DIMS(c);
  const double v = c.tot_n[X3(i, k, rr)];
  const double m_c = tm[X3(i, k, rr)], m_n = tm[X3(i, k, rr + 1)];
  const double lo = v * (m_c + m_n) + dabs(v) * (m_c - m_n);
  const double f = (rr == 1) ? 0.0 : v * (tt[X3(i, k, rr)] + tt[X3(i, k, rr + 1)]) - lo;
  // R+-Y of rows 1 and jmt are zero (adv_flx:467-481; tmask(:,:,jmt) = 0)
  const double rp0 = RpY[X3(i, k, rr)], rm0 = RmY[X3(i, k, rr)];
  const double rp1 = RpY[X3(i, k, rr + 1)], rm1 = RmY[X3(i, k, rr + 1)];
  const double Cpos = dmin(rp1, rm0), Cneg = dmin(rp0, rm1);
  const double lim = 0.5 * ((Cpos + Cneg) * f + (Cpos - Cneg) * dabs(f));
  return (lim + lo) * TMASK(i, k, rr);
}

// total (background + isopycnal) diffusive north flux of row rr, rr in 1..jmt-1
// tracer.F:948-961 and isopyc.F:1008-1053
UVIC_DEV double diff_north_flux(const uvic_ctx &c, const double *tm, int i, int k, int rr) {
  UV_DIMS(c);
  const double dT = tm[X3(i, k, rr + 1)] - tm[X3(i, k, rr)];
  const double bg = c.diff_cnt * c.csu_dyur[rr - 1] * dT;
  const double csu_dzt4r = c.csu[rr - 1] * 0.5 * c.dzt2r[k - 1];
  double sumz = 0.0;
  for (int kr = 0; kr <= 1; ++kr) {
    const int km1kr = imax(k - 1 + kr, 1), kpkr = imin(k + kr, km);
    for (int jq = 0; jq <= 1; ++jq)
      sumz = sumz - c.Ai_nz[X3(i, k, rr) + (size_t)(jq + 2 * kr) * N3] *
                        (tm[X3(i, km1kr, rr + jq)] - tm[X3(i, kpkr, rr + jq)]) * drodyn(i, k, rr, jq) /
                        (drodzn(i, k, rr, jq, kr) + UV_EPSLN);
  }
  const double flux_y = csu_dzt4r * sumz;
  return bg + c.K22[X3(i, k, rr)] * c.csu_dyur[rr - 1] * dT + flux_y;
}

// ===========================================================================
// update_rows_block: t(tau+1) of row j (js <= j <= je) for tracer n1.
// ===========================================================================
template <class Env>
UVIC_DEV void update_rows_block(Env &env, const uvic_ctx &c, int n1, int j, int chunk, int nchunk, double *lds) {
  UV_DIMS(c);
  const Tile T = make_tile(c, chunk, nchunk);
  const int W = T.W;
  const int NCW = W * km, NFW = W * (km + 1);
  const int nth = env.nthreads();
  const size_t nloc = (size_t)(n1 - 1 - c.n0);
  const double *tm = c.t_taum1 + (size_t)(n1 - 1) * N3;
  const double *tt = c.t_tau + (size_t)(n1 - 1) * N3;
  double *tp = c.t_taup1 + (size_t)(n1 - 1) * N3;
  const double *RpY = c.RpY + nloc * N3, *RmY = c.RmY + nloc * N3;
  const double *stf = c.stf + (size_t)(n1 - 1) * imt * jmt, *btf = c.btf + (size_t)(n1 - 1) * imt * jmt;
  const double *source = 0;
  if (c.src && c.itrc[n1 - 1] != 0) source = c.src + (size_t)(c.itrc[n1 - 1] - 1) * N3;
  double *s_tm = lds;             // t(tau-1) row j
  double *s_dfe = s_tm + NCW;     // diffusive east flux
  double *s_tp = s_dfe + NCW;     // explicit t(tau+1)
  double *s_dfb = s_tp + NCW;     // vertical diffusive flux (explicit part), faces 0..km
  double *s_dfbi = s_dfb + NFW;   // K31/K32 isopycnal vertical flux, faces 0..km
  const double cstr_j = c.cstr[j - 1];

  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      s_tm[q] = tm[X3(T.gi(l), k, j)];
    }
  });
  // diffusive fluxes through east and bottom faces -----------------------------
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      const int i = T.gi(l);
      if (l >= 1 && l <= W - 3) {  // east face: tracer.F:930-942, isopyc.F:953-1002
        const int ie = T.gi(l + 1);
        const double dT = s_tm[LC(l + 1, k)] - s_tm[LC(l, k)];
        const double bg = c.diff_cet * cstr_j * c.dxur[i - 1] * dT;
        const double dzt4r = 0.5 * c.dzt2r[k - 1];
        double sumz = 0.0;
        for (int kr = 0; kr <= 1; ++kr) {
          const int km1kr = imax(k - 1 + kr, 1), kpkr = imin(k + kr, km);
          for (int ip = 0; ip <= 1; ++ip) {
            const int ii = ip ? ie : i;
            const double dro_x = c.alphai[X3(ii, k, j)] * c.ddxt[X3(i, k, j)] + c.betai[X3(ii, k, j)] * c.ddxt[X3(i, k, j) + N3];
            const double dro_z = c.alphai[X3(ii, k, j)] * c.ddzt[XF(ii, k - 1 + kr, j)] +
                                 c.betai[X3(ii, k, j)] * c.ddzt[XF(ii, k - 1 + kr, j) + NF];
            sumz = sumz - c.Ai_ez[X3(i, k, j) + (size_t)(ip + 2 * kr) * N3] *
                              (s_tm[LC(l + ip, km1kr)] - s_tm[LC(l + ip, kpkr)]) * dro_x / (dro_z + UV_EPSLN);
          }
        }
        const double flux_x = dzt4r * sumz;
        const double cstdxur = cstr_j * c.dxur[i - 1];
        s_dfe[q] = bg + c.K11[X3(i, k, j)] * cstdxur * dT + flux_x;
      }
      if (l >= 2 && l <= W - 3) {
        const int iw = T.gi(l - 1), ie = T.gi(l + 1);
        if (k <= km - 1) {
          // tracer.F:1025-1032
          s_dfb[LF(l, k)] = c.diff_cbt[X3(i, k, j)] * c.dzwr[k] * (s_tm[LC(l, k)] - s_tm[LC(l, k + 1)]);
          // isopyc.F:1062-1100
          double sumx = 0.0;
          for (int ip = 0; ip <= 1; ++ip)
            for (int kr = 0; kr <= 1; ++kr) {
              const int ix = ip ? i : iw;  // ddxt column i-1+ip
              const double dro_x = c.alphai[X3(i, k + kr, j)] * c.ddxt[X3(ix, k + kr, j)] +
                                   c.betai[X3(i, k + kr, j)] * c.ddxt[X3(ix, k + kr, j) + N3];
              const double dro_z = c.alphai[X3(i, k + kr, j)] * c.ddzt[XF(i, k, j)] + c.betai[X3(i, k + kr, j)] * c.ddzt[XF(i, k, j) + NF];
              sumx = sumx - c.Ai_bx[X3(i, k, j) + (size_t)(ip + 2 * kr) * N3] * cstr_j *
                                (s_tm[LC(l + ip, k + kr)] - s_tm[LC(l - 1 + ip, k + kr)]) * dro_x / (dro_z + UV_EPSLN);
            }
          double sumy = 0.0;
          for (int jq = 0; jq <= 1; ++jq)
            for (int kr = 0; kr <= 1; ++kr) {
              const double dro_y = c.alphai[X3(i, k + kr, j)] * c.ddyt[X3(i, k + kr, j - 1 + jq)] +
                                   c.betai[X3(i, k + kr, j)] * c.ddyt[X3(i, k + kr, j - 1 + jq) + N3];
              const double dro_z = c.alphai[X3(i, k + kr, j)] * c.ddzt[XF(i, k, j)] + c.betai[X3(i, k + kr, j)] * c.ddzt[XF(i, k, j) + NF];
              sumy = sumy - c.Ai_by[X3(i, k, j) + (size_t)(jq + 2 * kr) * N3] * c.csu[j - 1 + jq - 1] *
                                (tm[X3(i, k + kr, j + jq)] - tm[X3(i, k + kr, j - 1 + jq)]) * dro_y / (dro_z + UV_EPSLN);
            }
          s_dfbi[LF(l, k)] = c.dxt4r[i - 1] * sumx + c.dyt4r[j - 1] * cstr_j * sumy;
        } else {
          s_dfb[LF(l, km)] = 0.0;
          s_dfbi[LF(l, km)] = 0.0;
        }
        if (k == 1) {
          s_dfb[LF(l, 0)] = stf[X2(i, j)];
          s_dfbi[LF(l, 0)] = 0.0;
        }
        (void)ie;
      }
    }
  });
  // bottom boundary condition of the explicit vertical flux, tracer.F:1060-1062
  env.par([&](int tid) {
    for (int l = 2 + tid; l <= W - 3; l += nth) {
      const int i = T.gi(l);
      s_dfb[LF(l, c.kmt[X2(i, j)])] = btf[X2(i, j)];
    }
  });
  // explicit update, tracer.F:1109-1130 with fdift.h -----------------------------
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      if (l < 2 || l > W - 3) continue;
      const int i = T.gi(l), iw = T.gi(l - 1), ie = T.gi(l + 1);
      const double cstdxtr = cstr_j * c.dxtr[i - 1];
      const double DIFF_Tx = (s_dfe[LC(l, k)] * TMASK(ie, k, j) - s_dfe[LC(l - 1, k)] * TMASK(iw, k, j)) * cstdxtr;
      const double dfn_n = diff_north_flux(c, tm, i, k, j), dfn_s = diff_north_flux(c, tm, i, k, j - 1);
      const double DIFF_Ty = (dfn_n * TMASK(i, k, j + 1) - dfn_s * TMASK(i, k, j - 1)) * c.cstdytr[j - 1];
      const double DIFF_Tz = (s_dfb[LF(l, k - 1)] - s_dfb[LF(l, k)]) * c.dztr[k - 1] * (1.0 - c.aidif) +
                             (s_dfbi[LF(l, k - 1)] - s_dfbi[LF(l, k)]) * c.dztr[k - 1];
      const double ADV_Tx = c.adv_x[X3(i, k, j) + nloc * N3];
      const double fn_n = fct_north_flux(c, tm, tt, RpY, RmY, i, k, j);
      const double fn_s = fct_north_flux(c, tm, tt, RpY, RmY, i, k, j - 1);
      const double ADV_Ty = (fn_n - fn_s) * c.cstdyt2r[j - 1];
      const double ADV_Tz = c.adv_z[X3(i, k, j) + nloc * N3];
      const double s = source ? source[X3(i, k, j)] : 0.0;
      const double twodt = c.c2dtts * c.dtxcel[k - 1];
      s_tp[q] = s_tm[q] + twodt * (DIFF_Tx + DIFF_Ty + DIFF_Tz - ADV_Tx - ADV_Ty - ADV_Tz + s) * TMASK(i, k, j);
    }
  });
  // implicit vertical diffusion: Thomas algorithm per column, invtri.F:57-110.
  // The forward sweep keeps e(k) in the (now free) s_dfe column.
  env.par([&](int tid) {
    for (int l = 2 + tid; l <= W - 3; l += nth) {
      const int i = T.gi(l);
      const double eps = 1.e-30;
      const double aidif = c.aidif;
      const int kz = imax(2, c.kmt[X2(i, j)]);
      const double topbc = stf[X2(i, j)], botbc = btf[X2(i, j)];
      double bet = 0.0, zprev = 0.0, cprev = 0.0;
      for (int k = 1; k <= km; ++k) {
        const int km1 = imax(1, k - 1), kp1 = imin(k + 1, km);
        const double tdt = c.c2dtts * c.dtxcel[k - 1];
        const double factu = c.dztur[k - 1] * tdt * aidif;
        const double factl = c.dztlr[k - 1] * tdt * aidif;
        const double mk = TMASK(i, k, j);
        double a = -c.diff_cbt[X3(i, km1, j)] * factu * mk;
        double cc = -c.diff_cbt[X3(i, k, j)] * factl * TMASK(i, kp1, j);
        const double z = s_tp[LC(l, k)];
        double f = z * mk;
        if (k == 1) a = 0.0;
        if (k == km) cc = 0.0;
        const double b = 1.0 - a - cc;
        if (k == 1) f = z + topbc * tdt * c.dztr[0] * aidif * mk;
        if (k == kz) f = z - botbc * tdt * c.dztr[k - 1] * aidif * mk;
        double znew;
        if (k == 1) {
          bet = mk / (b + eps);
          znew = f * bet;
        } else {
          const double e = cprev * bet;
          s_dfe[LC(l, k)] = e;
          bet = mk / (b - a * e + eps);
          znew = (f - a * zprev) * bet;
        }
        s_tp[LC(l, k)] = znew;
        zprev = znew;
        cprev = cc;
      }
      for (int k = km - 1; k >= 1; --k) s_tp[LC(l, k)] = s_tp[LC(l, k)] - s_dfe[LC(l, k + 1)] * s_tp[LC(l, k + 1)];
    }
  });
  // store row j of t(tau+1) including the cyclic images (tracer.F:1153-1155)
  env.par([&](int tid) {
    for (int q = tid; q < NCW; q += nth) {
      const int l = q % W, k = q / W + 1;
      if (l < 2 || l > W - 3) continue;
      const int i = T.gi(l);
      const double v = s_tp[q];
      tp[X3(i, k, j)] = v;
      if (i == 2) tp[X3(imt, k, j)] = v;
      if (i == imt - 1) tp[X3(1, k, j)] = v;
    }
  });
}

// ===========================================================================
// convct2 (source/mom/convect.F:99-311, O_fullconvect) for one column, all
// tracers; then the cyclic images.  One thread per (i,j), i = 2..imt-1.
// ===========================================================================
UVIC_DEV void convect_column(const uvic_ctx &c, int i, int j) {
  UV_DIMS(c);
  double *ts = c.t_taup1;
  const int nt = c.nt;
#define TS(k, n) ts[X3(i, k, j) + (size_t)((n)-1) * N3]
#define DENS(tq, sq, k) eos_dens(c.c, km, tq, sq, k)
  const double *to = c.to, *so = c.so, *dz = c.dztxcl;
  const int kbo = c.kmt[X2(i, j)];
  int kt = 1, kb = 2;
  bool mixed = false;
  while (kt < kbo) {
    double ru = DENS(TS(kt, 1) - to[kb - 1], TS(kt, 2) - so[kb - 1], kb);
    double rl = DENS(TS(kb, 1) - to[kb - 1], TS(kb, 2) - so[kb - 1], kb);
    if (ru > rl) {
      bool chk_la = true, chk_lb = true;
      double zsm = dz[kt - 1] + dz[kb - 1];
      double tsm1 = TS(kt, 1) * dz[kt - 1] + TS(kb, 1) * dz[kb - 1];
      double tmx1 = tsm1 / zsm;
      double tsm2 = TS(kt, 2) * dz[kt - 1] + TS(kb, 2) * dz[kb - 1];
      double tmx2 = tsm2 / zsm;
      while (chk_lb || chk_la) {
        if (kb >= kbo) chk_lb = false;
        while (chk_lb) {
          chk_lb = false;
          const int lb = kb + 1;
          ru = DENS(tmx1 - to[lb - 1], tmx2 - so[lb - 1], lb);
          rl = DENS(TS(lb, 1) - to[lb - 1], TS(lb, 2) - so[lb - 1], lb);
          if (ru > rl) {
            kb = lb;
            zsm = zsm + dz[kb - 1];
            tsm1 = tsm1 + TS(kb, 1) * dz[kb - 1];
            tmx1 = tsm1 / zsm;
            tsm2 = tsm2 + TS(kb, 2) * dz[kb - 1];
            tmx2 = tsm2 / zsm;
            chk_la = true;
            if (kb < kbo) chk_lb = true;
          }
        }
        chk_la = true;
        if (kt <= 1) chk_la = false;
        while (chk_la) {
          chk_la = false;
          const int la = kt - 1;
          ru = DENS(TS(la, 1) - to[kt - 1], TS(la, 2) - so[kt - 1], kt);
          rl = DENS(tmx1 - to[kt - 1], tmx2 - so[kt - 1], kt);
          if (ru > rl) {
            kt = la;
            zsm = zsm + dz[kt - 1];
            tsm1 = tsm1 + TS(kt, 1) * dz[kt - 1];
            tmx1 = tsm1 / zsm;
            tsm2 = tsm2 + TS(kt, 2) * dz[kt - 1];
            tmx2 = tsm2 / zsm;
            chk_lb = true;
          }
        }
      }
      for (int k = kt; k <= kb; ++k) {
        TS(k, 1) = tmx1;
        TS(k, 2) = tmx2;
      }
      for (int n = 3; n <= nt; ++n) {
        double tsm3 = 0.0;
        for (int k = kt; k <= kb; ++k) tsm3 = tsm3 + TS(k, n) * dz[k - 1];
        const double tmx3 = tsm3 / zsm;
        for (int k = kt; k <= kb; ++k) TS(k, n) = tmx3;
      }
      mixed = true;
      kt = kb + 1;
    } else {
      kt = kb;
    }
    kb = kt + 1;
  }
  if (mixed && (i == 2 || i == imt - 1)) {  // cyclic images, tracer.F:1199-1203
    const int ic = (i == 2) ? imt : 1;
    for (int n = 1; n <= nt; ++n)
      for (int k = 1; k <= km; ++k) ts[X3(ic, k, j) + (size_t)(n - 1) * N3] = TS(k, n);
  }
#undef TS
#undef DENS
}

// ===========================================================================
// convct2 split in two (same arithmetic, same order as convect_column above):
//   convect_ts_column : one thread per column walks T and S (staged in a small
//                       per-thread scratch, LDS on the GPU), mixes them and records
//                       every mixed segment (kt, kb, total thickness zsm) in order;
//   convect_apply_cell: one thread per (column, tracer n >= 3) replays the segments
//                       (tsm = sum_{k=kt..kb} t*dztxcl in that order, / zsm).
// The mixing ranges depend on T and S only (convect.F:189-255), the other tracers
// are mixed over them afterwards (:257-271), so the replay is exact and exposes
// nt-2 times more parallelism; most columns have no unstable segment at all.
// ===========================================================================
// `tab` (optional): the per-level tables of the walk -- c(km,9), to, so, dztxcl, in this order -- where the caller has staged
// them (LDS): the walk evaluates two densities per step with the level depending on the data, and every evaluation
// through global memory is a round trip on a chain others wait for
// `preloaded`: colT/colS hold the column already (pass B of the same workgroup has left it there)
UVIC_DEV void convect_ts_column(const uvic_ctx &c, int i, int j, double *colT, double *colS, int stride, const double *tab = nullptr,
                                bool preloaded = false) {
  UV_DIMS(c);
  double *ts = c.t_taup1;
  const double *eosc = tab ? tab : c.c;
#define DENS(tq, sq, k) eos_dens(eosc, km, tq, sq, k)
#define CT(k) colT[(size_t)((k)-1) * stride]
#define CS(k) colS[(size_t)((k)-1) * stride]
  const double *to = tab ? tab + (size_t)9 * km : c.to, *so = tab ? tab + (size_t)10 * km : c.so;
  const double *dz = tab ? tab + (size_t)11 * km : c.dztxcl;
  const int kbo = c.kmt[X2(i, j)];
  if (!preloaded)
  for (int k0 = 1; k0 <= km; k0 += 8) {   // (eight levels per memory round trip, not one)
    double a[8], b[8];
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {
      const int k = k0 + u <= km ? k0 + u : km;
      a[u] = ts[X3(i, k, j)];
      b[u] = ts[X3(i, k, j) + N3];
    }
    _Pragma("unroll") for (int u = 0; u < 8; ++u)
      if (k0 + u <= km) {
        CT(k0 + u) = a[u];
        CS(k0 + u) = b[u];
      }
  }
  int nseg = 0;
  int kt = 1, kb = 2;
  while (kt < kbo) {
    double ru = DENS(CT(kt) - to[kb - 1], CS(kt) - so[kb - 1], kb);
    double rl = DENS(CT(kb) - to[kb - 1], CS(kb) - so[kb - 1], kb);
    if (ru > rl) {
      bool chk_la = true, chk_lb = true;
      double zsm = dz[kt - 1] + dz[kb - 1];
      double tsm1 = CT(kt) * dz[kt - 1] + CT(kb) * dz[kb - 1];
      double tmx1 = tsm1 / zsm;
      double tsm2 = CS(kt) * dz[kt - 1] + CS(kb) * dz[kb - 1];
      double tmx2 = tsm2 / zsm;
      while (chk_lb || chk_la) {
        if (kb >= kbo) chk_lb = false;
        while (chk_lb) {
          chk_lb = false;
          const int lb = kb + 1;
          ru = DENS(tmx1 - to[lb - 1], tmx2 - so[lb - 1], lb);
          rl = DENS(CT(lb) - to[lb - 1], CS(lb) - so[lb - 1], lb);
          if (ru > rl) {
            kb = lb;
            zsm = zsm + dz[kb - 1];
            tsm1 = tsm1 + CT(kb) * dz[kb - 1];
            tmx1 = tsm1 / zsm;
            tsm2 = tsm2 + CS(kb) * dz[kb - 1];
            tmx2 = tsm2 / zsm;
            chk_la = true;
            if (kb < kbo) chk_lb = true;
          }
        }
        chk_la = true;
        if (kt <= 1) chk_la = false;
        while (chk_la) {
          chk_la = false;
          const int la = kt - 1;
          ru = DENS(CT(la) - to[kt - 1], CS(la) - so[kt - 1], kt);
          rl = DENS(tmx1 - to[kt - 1], tmx2 - so[kt - 1], kt);
          if (ru > rl) {
            kt = la;
            zsm = zsm + dz[kt - 1];
            tsm1 = tsm1 + CT(kt) * dz[kt - 1];
            tmx1 = tsm1 / zsm;
            tsm2 = tsm2 + CS(kt) * dz[kt - 1];
            tmx2 = tsm2 / zsm;
            chk_lb = true;
          }
        }
      }
      for (int k = kt; k <= kb; ++k) {
        CT(k) = tmx1;
        CS(k) = tmx2;
      }
      c.cv_kt[X3(i, nseg + 1, j)] = kt;
      c.cv_kb[X3(i, nseg + 1, j)] = kb;
      c.cv_z[X3(i, nseg + 1, j)] = zsm;
      ++nseg;
      kt = kb + 1;
    } else {
      kt = kb;
    }
    kb = kt + 1;
  }
  c.cv_nseg[X2(i, j)] = nseg;
  if (nseg > 0) {
    const int ic = (i == 2) ? imt : ((i == imt - 1) ? 1 : 0);
    for (int k = 1; k <= km; ++k) {
      const double a = CT(k), b = CS(k);
      ts[X3(i, k, j)] = a;
      ts[X3(i, k, j) + N3] = b;
      if (ic) {
        ts[X3(ic, k, j)] = a;
        ts[X3(ic, k, j) + N3] = b;
      }
    }
  }
#undef DENS
#undef CT
#undef CS
}

// tracer n (>= 3) of column (i,j): replay the recorded segments, convect.F:263-271
UVIC_DEV void convect_apply_cell(const uvic_ctx &c, int i, int j, int n) {
  UV_DIMS(c);
  const int nseg = c.cv_nseg[X2(i, j)];
  if (nseg == 0) return;
  double *t = c.t_taup1 + (size_t)(n - 1) * N3;
  const double *dz = c.dztxcl;
  for (int s = 1; s <= nseg; ++s) {
    const int kt = c.cv_kt[X3(i, s, j)], kb = c.cv_kb[X3(i, s, j)];
    const double zsm = c.cv_z[X3(i, s, j)];
    double tsm3 = 0.0;
    for (int k = kt; k <= kb; ++k) tsm3 = tsm3 + t[X3(i, k, j)] * dz[k - 1];
    const double tmx3 = tsm3 / zsm;
    for (int k = kt; k <= kb; ++k) t[X3(i, k, j)] = tmx3;
  }
  const int ic = (i == 2) ? imt : ((i == imt - 1) ? 1 : 0);
  if (ic)
    for (int k = 1; k <= km; ++k) t[X3(ic, k, j)] = t[X3(i, k, j)];
}

}  // namespace uvic
#endif
