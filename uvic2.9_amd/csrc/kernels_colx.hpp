// kernels_colx.hpp -- the column kernels in the REFERENCE'S OWN ORDER OF OPERATIONS, for T and S.
//
// Every convective adjustment is decided on the bits of T and S (source/mom/convect.F:189-255: two densities are
// compared), and the isopycnal tensor, K33 and the GM velocities of the next steps are functions of them.  A T or S
// that differs from the reference by rounding flips a marginally stable column sooner or later (one
// column of the 100-step run, 2e-10 of max|t|).  So the production step sends these two tracers -- 2 of nt -- through
// kernels that evaluate every expression exactly as the reference does (no contraction, IEEE division, the isopycnal
// fluxes as ((Ai*dT)*drodx)/(drodz+epsln) and not through folded coefficients); the result equals kernels_fct.hpp's and
// the compiled reference's bit for bit.  The other tracers keep the folded, contracted column kernels (kernels_col.hpp).
//
// The row kernels of kernels_fct.hpp are exact as well, but a (row, tracer) tile of 1024 threads and 66 KB of LDS finds no
// room beside the bulk passes (fct_rows + update_rows of T,S: 0.32 ms in the loop).  Here the work is laid out like the
// column kernels: a lane is an ocean column of the pass-A lane map marching down k, x-neighbours by DPP, rows r-1, r+1,
// r+2 from the lane's own loads.  A workgroup is four waves on the same 64 lanes, one ROLE each:
//     wave 0: T, advective half   wave 1: S, advective half   wave 2: T, diffusive half   wave 3: S, diffusive half
// so that the chain every step waits for is as long as ONE half (reference-order isoflux costs 24 IEEE divisions per
// cell and tracer: the two halves weigh about the same).
//   advective half (tracer_adv_flx.F:500-887, 989-999): low-order fluxes, t_lo, the six limiter ratios, limited x and z
//       fluxes -> ADV_Tx, ADV_Tz; and, as in kernels_col.hpp (YFIN), t_lo and the y ratios of the row to the north and
//       with them the FINAL flux through the north face (adv_flx:770-783)
//   diffusive half (tracer.F:925-1032, isopyc.F:953-1108, fdift.h:61-88): diff_fe, diff_fn of rows r and r-1, diff_fb,
//       diff_fbiso -> D = DIFF_Tx + DIFF_Ty + DIFF_Tz
//   pass B (`colx_upd_wave`): t(tau-1) + twodt*(D - ADV_Tx - ADV_Ty - ADV_Tz + source)*tmask in the reference's order
//       (tracer.F:1114-1127), then invtri (source/mom/invtri.F:57-110).
// This header is compiled with contraction off (the library's default; kernels_col.hpp switches it back off at its end).
#ifndef UVIC_KERNELS_COLX_HPP
#define UVIC_KERNELS_COLX_HPP

#include "kernels_col.hpp"

namespace uvic {
#if defined(__HIPCC__)
#pragma clang fp contract(off)

// adv_flx:501-514: totadv*(a+b) + |totadv|*(a-b)
__device__ __forceinline__ double x_up(double v, double a, double b) { return v * (a + b) + dabs(v) * (a - b); }
// adv_flx:703-705: 0.5*((Cpos+Cneg)*f + (Cpos-Cneg)*|f|)
__device__ __forceinline__ double x_lim(double cpos, double cneg, double f) { return 0.5 * ((cpos + cneg) * f + (cpos - cneg) * dabs(f)); }
// adv_flx:672-690 (the land blend fxa = tmask*mean + (1-tmask)*t_lo is a select: tmask is 0 or 1)
__device__ __forceinline__ void x_ratio(double fxa, double fxb, double tlo, double scale, double flxlft, double flxrgt, double mask,
                                        double &rp, double &rm) {
  const double trmax = fmx(fmx(fxa, fxb), tlo), trmin = fmn(fmn(fxa, fxb), tlo);
  const double pplus = scale * (fmx(0.0, flxlft) - fmn(0.0, flxrgt));
  const double pminus = scale * (fmx(0.0, flxrgt) - fmn(0.0, flxlft));
  rp = fmn(1., mask * (trmax - tlo) / (pplus + UV_EPSLN));
  rm = fmn(1., mask * (tlo - trmin) / (pminus + UV_EPSLN));
}

// Division with a shared reciprocal.  The compiler's sequence for x / y in fp64 is: r = v_rcp_f64(y), two Newton steps on r,
// q = x*r, the residual x - y*q, one correction of q (plus v_div_scale / v_div_fixup, which act only on operands near the
// ends of the exponent range: not on slopes and tracer differences).  The reciprocal depends on y alone, and the twenty
// quotients of a cell's isopycnal fluxes have nine denominators between them, most of which a neighbouring lane or the
// level above has formed already: x_rcp once per denominator, x_div per quotient, the same bits as x / y.
__device__ __forceinline__ double x_rcp(double y) {
  double r = __builtin_amdgcn_rcp(y);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double x_div(double x, double y, double r) {
#ifdef UV_COLX_PLAIN_DIV
  (void)r;
  return x / y;
#else
  const double q = x * r;
  return __builtin_fma(__builtin_fma(-y, q, x), r, q);
#endif
}

// what the two halves leave for pass B, one plane per local tracer each (views of the FCT work arrays, which the
// column path does not use for T and S): ADV_Tx, ADV_Tz, the final north-face flux, D
struct ColxOut {
  double *adv_x, *adv_z, *fn, *dif;
};

#define XOC(k, dj) (((((k)-1) * imt + ((dj) + 1) * rowstride)) * 8)   /* byte offset of level k of row r+dj from the lane's offset (row r-1) */
#define XOF(kf) (((kf) * imt) * 8)

// ===========================================================================
// advective half, one tracer (global number n1, local slot nloc)
// ===========================================================================
__device__ __forceinline__ void colx_adv_wave(const uvic_ctx &c, const ColxOut &o, int code, int n1, int nloc) {
  UV_DIMS(c);
  const int i = COL_LANE_I(code), r = COL_LANE_R(code);
  const bool owned = COL_LANE_OWNED(code) != 0;
  LaneTable t_dzt2r, t_dtxcel;
  t_dzt2r.load(c.dzt2r, km); t_dtxcel.load(c.dtxcel, km);
  const int kz = c.kmt[X2(i, r)], kz_s = c.kmt[X2(i, r - 1)], kz_n = c.kmt[X2(i, r + 1)];
  const int kz_w = dpp_i<DPP_WAVE_SHR1>(kz), kz_e = dpp_i<DPP_WAVE_SHL1>(kz);
  const int rnn = imin(r + 2, jmt);
  const int kz_nn = c.kmt[X2(i, rnn)];
  const double cstr_r = c.cstr[r - 1];
  const double cstdxt2r = cstr_r * c.dxtr[i - 1] * 0.5, cstdyt2r = c.cstdyt2r[r - 1];
  const double cstdxt2r_N = c.cstr[r] * c.dxtr[i - 1] * 0.5, cstdyt2r_N = c.cstdyt2r[r];
  const bool south_wall = (r - 1 == 1);
  const double c2dtts = c.c2dtts;
  const int rowstride = imt * km;
  const unsigned lb = (unsigned)((r - 2) * rowstride + (i - 1)) * 8u;
  const unsigned lbf = (unsigned)((r - 1) * imt * (km + 1) + (i - 1)) * 8u;
  const unsigned lb_nn = (unsigned)((rnn - 1) * rowstride + (i - 1)) * 8u;
  const unsigned lbf_N = lbf + (unsigned)(imt * (km + 1)) * 8u;
  const brsrc b_te = mkbuf(c.tot_e, N3 * 8), b_tn = mkbuf(c.tot_n, N3 * 8);
  const brsrc b_tb = mkbuf(c.tot_b, NF * 8), b_vb = mkbuf(c.adv_vbt, NF * 8);
  const brsrc b_tm = mkbuf(c.t_taum1 + (size_t)(n1 - 1) * N3, N3 * 8), b_tt = mkbuf(c.t_tau + (size_t)(n1 - 1) * N3, N3 * 8);
  const brsrc b_ax = mkbuf(o.adv_x + (size_t)nloc * N3, N3 * 8), b_az = mkbuf(o.adv_z + (size_t)nloc * N3, N3 * 8);
  const brsrc b_fn = mkbuf(o.fn + (size_t)nloc * N3, N3 * 8);
  const double vb0 = bld(b_vb, lbf, XOF(0)), vb0N = bld(b_vb, lbf_N, XOF(0));
  // state carried from level to level
  double mc1 = bld(b_tm, lb, XOC(1, 0)), ms1 = bld(b_tm, lb, XOC(1, -1)), mn1 = bld(b_tm, lb, XOC(1, 1));
  double tc1 = bld(b_tt, lb, XOC(1, 0)), tc0 = tc1;
  double fblo_up = vb0 * 2.0 * mc1, afb_up = fblo_up;        // adv_flx:543, 617 (both from t(tau-1))
  double fbfin_up = vb0 * (tc1 + tc1);                       // tracer.F:1063
  double fbloN_up = vb0N * 2.0 * mn1;
  double rzp_prev = 0.0, rzm_prev = 0.0, mk_prev = 0.0, adz_last = 0.0;
  double me1 = shfl_e(mc1);
  for (int s = 1; s <= km; ++s) {
    const bool last = (s == km);
    const int sp = last ? km : s + 1;
    // loads of the level
    const double ve = bld(b_te, lb, XOC(s, 0)), vn = bld(b_tn, lb, XOC(s, 0)), vs = bld(b_tn, lb, XOC(s, -1));
    const double vb = last ? bld(b_vb, lbf, XOF(km)) : bld(b_tb, lbf, XOF(s));
    const double veN = bld(b_te, lb, XOC(s, 1)), vnN = bld(b_tn, lb, XOC(s, 1));
    const double vbN = last ? 0.0 : bld(b_tb, lbf_N, XOF(s));
    const double mc2 = bld(b_tm, lb, XOC(sp, 0)), ms2 = bld(b_tm, lb, XOC(sp, -1)), mn2 = bld(b_tm, lb, XOC(sp, 1));
    const double tc2 = bld(b_tt, lb, XOC(sp, 0)), t_s = bld(b_tt, lb, XOC(s, -1)), t_n = bld(b_tt, lb, XOC(s, 1));
    const double m_nn = bld(b_tm, lb_nn, XOC(s, -1)), t_nn = bld(b_tt, lb_nn, XOC(s, -1));
    const double dzt2r_s = t_dzt2r.at(s - 1), dzt2r_up = (s >= 2) ? t_dzt2r.at(s - 2) : 0.0;
    const double twodt = c2dtts * t_dtxcel.at(s - 1);
    const double mk = (s <= kz) ? 1.0 : 0.0;
    const bool wet_w = s <= kz_w, wet_e = s <= kz_e, wet_s = s <= kz_s, wet_n = s <= kz_n;
    const bool wet_up = s - 1 >= 1 && s - 1 <= kz, wet_dn = s + 1 <= kz;
    const double m_c = mc1, tt_c = tc1;
    const double m_e = me1, tt_e = shfl_e(tt_c), tt_w = shfl_w(tt_c);
    const double mc2_e = shfl_e(mc2);
    // low order and raw antidiffusive fluxes (adv_flx:500-619)
    const double felo = x_up(ve, m_c, m_e);
    const double afe = ve * (tt_c + tt_e) - felo;
    const double felo_w = shfl_w(felo), afe_w = shfl_w(afe);
    const double fnlo_n = x_up(vn, m_c, mn1), fnlo_s = x_up(vs, ms1, m_c);
    double fblo = 0.0, afb = 0.0;
    if (!last) {
      fblo = vb * (mc2 + m_c) + dabs(vb) * (mc2 - m_c);
      afb = vb * (tt_c + tc2) - fblo * mk;
    }
    const double advx = (felo - felo_w) * cstdxt2r, advy = (fnlo_n - fnlo_s) * cstdyt2r;
    const double advz = (fblo_up - fblo) * dzt2r_s;
    const double tlo = m_c - twodt * (advx + advy + advz) * mk;
    double rxp, rxm, ryp, rym, rzp, rzm;
    {
      const double mw = 0.5 * (tt_w + tt_c), me = 0.5 * (tt_c + tt_e);
      x_ratio(wet_w ? mw : tlo, wet_e ? me : tlo, tlo, c2dtts * cstdxt2r, afe_w, afe, mk, rxp, rxm);
    }
    const double afn_n = vn * (tt_c + t_n) - fnlo_n;
    {
      const double afn_s = south_wall ? 0.0 : vs * (t_s + tt_c) - fnlo_s;
      x_ratio(wet_s ? 0.5 * (t_s + tt_c) : tlo, wet_n ? 0.5 * (tt_c + t_n) : tlo, tlo, c2dtts * cstdyt2r, afn_s, afn_n, mk, ryp, rym);
    }
    {
      const double fxa = wet_up ? 0.5 * (tc0 + tt_c) : tlo;
      const double fxb = (!last && wet_dn) ? 0.5 * (tt_c + tc2) : tlo;
      x_ratio(fxa, fxb, tlo, c2dtts * dzt2r_s, afb, afb_up, mk, rzp, rzm);
    }
    // the row to the north: its t_lo and y ratios, the same expressions one row up
    {
      const double m_N = mn1;
      const double mkN = (s <= kz_n) ? 1.0 : 0.0;
      const double feloN = x_up(veN, m_N, shfl_e(m_N));
      const double fnloN_n = x_up(vnN, m_N, m_nn);
      double fbloN = 0.0;
      if (!last) fbloN = vbN * (mn2 + m_N) + dabs(vbN) * (mn2 - m_N);
      const double advxN = (feloN - shfl_w(feloN)) * cstdxt2r_N, advyN = (fnloN_n - fnlo_n) * cstdyt2r_N;
      const double advzN = (fbloN_up - fbloN) * dzt2r_s;
      const double tloN = m_N - twodt * (advxN + advyN + advzN) * mkN;
      const double afn_nn = vnN * (t_n + t_nn) - fnloN_n;
      double rypN, rymN;
      x_ratio(mk != 0.0 ? 0.5 * (tt_c + t_n) : tloN, (s <= kz_nn) ? 0.5 * (t_n + t_nn) : tloN, tloN, c2dtts * cstdyt2r_N, afn_n, afn_nn, mkN,
              rypN, rymN);
      // the limited flux through the north face, final (adv_flx:770-783, 994-996)
      const double fn_fin = (x_lim(fmn(rypN, rym), fmn(ryp, rymN), afn_n) + fnlo_n) * mk;
      if (owned) bst(b_fn, lb, XOC(s, 0), fn_fin);
      fbloN_up = fbloN;
    }
    // limited x flux and its divergence (adv_flx:695-711, 989-992; fdift.h:25)
    const double rxp_e = shfl_e(rxp), rxm_e = shfl_e(rxm);
    const double fefin = x_lim(fmn(rxp_e, rxm), fmn(rxp, rxm_e), afe) + felo;
    const double ADV_Tx = (fefin - shfl_w(fefin)) * cstdxt2r;
    if (owned) bst(b_ax, lb, XOC(s, 0), ADV_Tx);
    // level s-1: limited z flux through the face between s-1 and s (adv_flx:857-887, 994-999; fdift.h:39)
    if (s >= 2) {
      const double fbfin = (x_lim(fmn(rzp_prev, rzm), fmn(rzp, rzm_prev), afb_up) + fblo_up) * mk_prev;
      const double ADV_Tz = (fbfin_up - fbfin) * dzt2r_up;
      if (owned) bst(b_az, lb, XOC(s - 1, 0), ADV_Tz);
      fbfin_up = fbfin;
    }
    if (last) adz_last = (fbfin_up - vb * tt_c) * dzt2r_s;   // bottom face of the column (tracer.F:1065); vb holds adv_vbt there
    rzp_prev = rzp; rzm_prev = rzm; mk_prev = mk;
    fblo_up = fblo; afb_up = afb;
    me1 = mc2_e;
    tc0 = tc1; tc1 = tc2;
    mc1 = mc2; ms1 = ms2; mn1 = mn2;
  }
  if (owned) bst(b_az, lb, XOC(km, 0), adz_last);
}

// ===========================================================================
// diffusive half, one tracer: the isopycnal fluxes in the reference's own association (isopyc.F:953-1108 through the
// statement functions of isopyc.h:121-136), from the Ai planes, alphai, betai and the T,S gradients of isopyc_elements
// ===========================================================================
__device__ __forceinline__ void colx_dif_wave(const uvic_ctx &c, const ColxOut &o, int code, int n1, int nloc) {
  UV_DIMS(c);
  const int i = COL_LANE_I(code), r = COL_LANE_R(code);
  const bool owned = COL_LANE_OWNED(code) != 0;
  LaneTable t_dzt2r, t_dztr, t_dzwr;
  t_dzt2r.load(c.dzt2r, km); t_dztr.load(c.dztr, km); t_dzwr.load(c.dzwr + 1, km);   // dzwr(1..km)
  const int kz = c.kmt[X2(i, r)], kz_s = c.kmt[X2(i, r - 1)], kz_n = c.kmt[X2(i, r + 1)];
  const int kz_w = dpp_i<DPP_WAVE_SHR1>(kz), kz_e = dpp_i<DPP_WAVE_SHL1>(kz);
  const double cstr_j = c.cstr[r - 1];
  const double cstdxur = cstr_j * c.dxur[i - 1];                 // isopyc.F:997
  const double bg_e = c.diff_cet * cstr_j * c.dxur[i - 1];        // tracer.F:930-942
  const double cstdxtr = cstr_j * c.dxtr[i - 1], cstdytr = c.cstdytr[r - 1];
  const double csu_dyur_n = c.csu_dyur[r - 1], csu_dyur_s = c.csu_dyur[r - 2];
  const double bg_n = c.diff_cnt * csu_dyur_n, bg_s = c.diff_cnt * csu_dyur_s;   // tracer.F:948-961
  const double csu_n = c.csu[r - 1], csu_s = c.csu[r - 2];
  const double dxt4r = c.dxt4r[i - 1], dyt4r_cstr = c.dyt4r[r - 1] * cstr_j;
  const double aidif1 = 1.0 - c.aidif;
  const double stf = c.stf[X2(i, r) + (size_t)(n1 - 1) * imt * jmt], btf = c.btf[X2(i, r) + (size_t)(n1 - 1) * imt * jmt];
  const int rowstride = imt * km, rowstride_f = imt * (km + 1);
  const unsigned lb = (unsigned)((r - 2) * rowstride + (i - 1)) * 8u;            // level 1 of column i of row r-1, cell fields
  const unsigned lbf = (unsigned)((r - 2) * rowstride_f + (i - 1)) * 8u;         // face 0 of column i of row r-1, face fields
#define XOFJ(kf, dj) ((((kf) * imt) + ((dj) + 1) * rowstride_f) * 8)
  const brsrc b_tm = mkbuf(c.t_taum1 + (size_t)(n1 - 1) * N3, N3 * 8);
  const brsrc b_al = mkbuf(c.alphai, N3 * 8), b_be = mkbuf(c.betai, N3 * 8);
  const brsrc b_dx = mkbuf(c.ddxt, N3 * 16), b_dy = mkbuf(c.ddyt, N3 * 16), b_dz = mkbuf(c.ddzt, NF * 16);
  const brsrc b_ez = mkbuf(c.Ai_ez, N3 * 32), b_nz = mkbuf(c.Ai_nz, N3 * 32), b_bx = mkbuf(c.Ai_bx, N3 * 32), b_by = mkbuf(c.Ai_by, N3 * 32);
  const brsrc b_k11 = mkbuf(c.K11, N3 * 8), b_k22 = mkbuf(c.K22, N3 * 8), b_dcb = mkbuf(c.diff_cbt, N3 * 8);
  const brsrc b_D = mkbuf(o.dif + (size_t)nloc * N3, N3 * 8);
  const int P1 = (int)(N3 * 8), PF1 = (int)(NF * 8);   // byte distance between the two components (T, S) of a gradient field
  // level s of: t(tau-1) (own, south, north), alphai, betai (own), ddxt and ddyt (both components; ddyt of rows r and r-1)
  double mc1 = bld(b_tm, lb, XOC(1, 0)), ms1 = bld(b_tm, lb, XOC(1, -1)), mn1 = bld(b_tm, lb, XOC(1, 1));
  double al1 = bld(b_al, lb, XOC(1, 0)), be1 = bld(b_be, lb, XOC(1, 0));
  double dxT1 = bld(b_dx, lb, XOC(1, 0)), dxS1 = bld(b_dx, lb, XOC(1, 0) + P1);
  double dyT1 = bld(b_dy, lb, XOC(1, 0)), dyS1 = bld(b_dy, lb, XOC(1, 0) + P1);
  double dyTs1 = bld(b_dy, lb, XOC(1, -1)), dySs1 = bld(b_dy, lb, XOC(1, -1) + P1);
  // face s-1 of ddzt (own, south, north columns): zero at the surface (isopyc.F:417)
  double dzT0 = 0.0, dzS0 = 0.0, dzTs0 = 0.0, dzSs0 = 0.0, dzTn0 = 0.0, dzSn0 = 0.0;
  // T(s-1) - T(s) of the own, south and north columns (zero at the top: km1kr = max(k-1+kr, 1))
  double dz_c = 0.0, dz_s = 0.0, dz_n = 0.0;
  double dfb_up = stf, dfbi_up = 0.0;     // tracer.F:1060, isopyc.F:1062
  if (kz == 0) dfb_up = btf;              // (face 0 is the bottom face of a land column, tracer.F:1061)
  // drodz + epsln of the own column with the gradient of the face ABOVE level s, and its reciprocal: at level s+1 this is
  // what level s formed for its bottom face with alphai, betai of level s+1 (isopyc.h:135-136), so it is handed down
  double yu_c = al1 * dzT0 + be1 * dzS0 + UV_EPSLN, ru_c = x_rcp(yu_c);
  for (int s = 1; s <= km; ++s) {
    const bool last = (s == km);
    const int sp = last ? km : s + 1;
    const double mc2 = bld(b_tm, lb, XOC(sp, 0)), ms2 = bld(b_tm, lb, XOC(sp, -1)), mn2 = bld(b_tm, lb, XOC(sp, 1));
    const double al2 = bld(b_al, lb, XOC(sp, 0)), be2 = bld(b_be, lb, XOC(sp, 0));
    const double al_n = bld(b_al, lb, XOC(s, 1)), be_n = bld(b_be, lb, XOC(s, 1));
    const double al_s = bld(b_al, lb, XOC(s, -1)), be_s = bld(b_be, lb, XOC(s, -1));
    const double dxT2 = bld(b_dx, lb, XOC(sp, 0)), dxS2 = bld(b_dx, lb, XOC(sp, 0) + P1);
    const double dyT2 = bld(b_dy, lb, XOC(sp, 0)), dyS2 = bld(b_dy, lb, XOC(sp, 0) + P1);
    const double dyTs2 = bld(b_dy, lb, XOC(sp, -1)), dySs2 = bld(b_dy, lb, XOC(sp, -1) + P1);
    const double dzT1 = bld(b_dz, lbf, XOFJ(s, 0)), dzS1 = bld(b_dz, lbf, XOFJ(s, 0) + PF1);
    const double dzTs1 = bld(b_dz, lbf, XOFJ(s, -1)), dzSs1 = bld(b_dz, lbf, XOFJ(s, -1) + PF1);
    const double dzTn1 = bld(b_dz, lbf, XOFJ(s, 1)), dzSn1 = bld(b_dz, lbf, XOFJ(s, 1) + PF1);
    double aez[4], anz[4], asz[4], abx[4], aby[4];
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {
      aez[p] = bld(b_ez, lb, XOC(s, 0) + p * P1);
      anz[p] = bld(b_nz, lb, XOC(s, 0) + p * P1);
      asz[p] = bld(b_nz, lb, XOC(s, -1) + p * P1);
      abx[p] = bld(b_bx, lb, XOC(s, 0) + p * P1);
      aby[p] = bld(b_by, lb, XOC(s, 0) + p * P1);
    }
    const double k11 = bld(b_k11, lb, XOC(s, 0)), k22n = bld(b_k22, lb, XOC(s, 0)), k22s = bld(b_k22, lb, XOC(s, -1));
    const double dcb = bld(b_dcb, lb, XOC(s, 0));
    const double dzt4r = 0.5 * t_dzt2r.at(s - 1), ddztr = t_dztr.at(s - 1);
    const double m_c = mc1;
    // differences of t(tau-1) (zero below the last level: kpkr = min(k+kr, km))
    const double dz_dn = (!last) ? m_c - mc2 : 0.0, dzs_dn = (!last) ? ms1 - ms2 : 0.0, dzn_dn = (!last) ? mn1 - mn2 : 0.0;
    const double dze_up = shfl_e(dz_c), dze_dn = shfl_e(dz_dn);
    const double m_e = shfl_e(m_c), mc2_e = shfl_e(mc2);
    const double dx_c = m_e - m_c, dx_d = mc2_e - mc2;            // T(i+1) - T(i) at levels s, s+1
    const double dxw_c = shfl_w(dx_c), dxw_d = shfl_w(dx_d);      // T(i) - T(i-1)
    // drodz + epsln (isopyc.h:125-136) of the own column below level s, of the north and south columns above and below it,
    // and their reciprocals; those of the east column are the east lane's own (the same expression on the same operands)
    const double yd_c = al1 * dzT1 + be1 * dzS1 + UV_EPSLN, rd_c = x_rcp(yd_c);
    const double yu_n = al_n * dzTn0 + be_n * dzSn0 + UV_EPSLN, ru_n = x_rcp(yu_n);
    const double yd_n = al_n * dzTn1 + be_n * dzSn1 + UV_EPSLN, rd_n = x_rcp(yd_n);
    const double yu_s = al_s * dzTs0 + be_s * dzSs0 + UV_EPSLN, ru_s = x_rcp(yu_s);
    const double yd_s = al_s * dzTs1 + be_s * dzSs1 + UV_EPSLN, rd_s = x_rcp(yd_s);
    const double yu_e = shfl_e(yu_c), ru_e = shfl_e(ru_c), yd_e = shfl_e(yd_c), rd_e = shfl_e(rd_c);
    // ---- east face: tracer.F:930-942, isopyc.F:953-1002 -------------------------------------------------
    double dfe;
    {
      const double al_e = shfl_e(al1), be_e = shfl_e(be1);
      const double dro_x0 = al1 * dxT1 + be1 * dxS1, dro_x1 = al_e * dxT1 + be_e * dxS1;
      double sumz = 0.0;
      sumz = sumz - x_div(aez[0] * dz_c * dro_x0, yu_c, ru_c);
      sumz = sumz - x_div(aez[1] * dze_up * dro_x1, yu_e, ru_e);
      sumz = sumz - x_div(aez[2] * dz_dn * dro_x0, yd_c, rd_c);
      sumz = sumz - x_div(aez[3] * dze_dn * dro_x1, yd_e, rd_e);
      const double flux_x = dzt4r * sumz;
      dfe = bg_e * dx_c + k11 * cstdxur * dx_c + flux_x;
    }
    const double me_mask = (s <= kz_e) ? 1.0 : 0.0, mw_mask = (s <= kz_w) ? 1.0 : 0.0;
    const double DIFF_Tx = (dfe * me_mask - shfl_w(dfe) * mw_mask) * cstdxtr;
    // ---- north faces of rows r and r-1: tracer.F:948-961, isopyc.F:1008-1053 -------------------------------
    double dfn_n, dfn_s;
    {
      const double dT = mn1 - m_c;
      const double dro_y0 = al1 * dyT1 + be1 * dyS1, dro_y1 = al_n * dyT1 + be_n * dyS1;
      double sumz = 0.0;
      sumz = sumz - x_div(anz[0] * dz_c * dro_y0, yu_c, ru_c);
      sumz = sumz - x_div(anz[1] * dz_n * dro_y1, yu_n, ru_n);
      sumz = sumz - x_div(anz[2] * dz_dn * dro_y0, yd_c, rd_c);
      sumz = sumz - x_div(anz[3] * dzn_dn * dro_y1, yd_n, rd_n);
      const double flux_y = csu_n * dzt4r * sumz;
      dfn_n = bg_n * dT + k22n * csu_dyur_n * dT + flux_y;
    }
    {
      const double dT = m_c - ms1;
      const double dro_y0 = al_s * dyTs1 + be_s * dySs1, dro_y1 = al1 * dyTs1 + be1 * dySs1;
      double sumz = 0.0;
      sumz = sumz - x_div(asz[0] * dz_s * dro_y0, yu_s, ru_s);
      sumz = sumz - x_div(asz[1] * dz_c * dro_y1, yu_c, ru_c);
      sumz = sumz - x_div(asz[2] * dzs_dn * dro_y0, yd_s, rd_s);
      sumz = sumz - x_div(asz[3] * dz_dn * dro_y1, yd_c, rd_c);
      const double flux_y = csu_s * dzt4r * sumz;
      dfn_s = bg_s * dT + k22s * csu_dyur_s * dT + flux_y;
    }
    const double mn_mask = (s <= kz_n) ? 1.0 : 0.0, ms_mask = (s <= kz_s) ? 1.0 : 0.0;
    const double DIFF_Ty = (dfn_n * mn_mask - dfn_s * ms_mask) * cstdytr;
    // ---- bottom face: tracer.F:1025-1032, isopyc.F:1062-1108 ----------------------------------------------
    double dfb = 0.0, dfbi = 0.0;
    if (!last) {
      dfb = dcb * t_dzwr.at(s - 1) * (m_c - mc2);
      const double dxTw1 = shfl_w(dxT1), dxSw1 = shfl_w(dxS1), dxTw2 = shfl_w(dxT2), dxSw2 = shfl_w(dxS2);
      const double den1 = al2 * dzT1 + be2 * dzS1 + UV_EPSLN, rcp1 = x_rcp(den1);
      double sumx = 0.0;
      sumx = sumx - x_div(abx[0] * cstr_j * dxw_c * (al1 * dxTw1 + be1 * dxSw1), yd_c, rd_c);     // ip = 0, kr = 0
      sumx = sumx - x_div(abx[2] * cstr_j * dxw_d * (al2 * dxTw2 + be2 * dxSw2), den1, rcp1);     // ip = 0, kr = 1
      sumx = sumx - x_div(abx[1] * cstr_j * dx_c * (al1 * dxT1 + be1 * dxS1), yd_c, rd_c);        // ip = 1, kr = 0
      sumx = sumx - x_div(abx[3] * cstr_j * dx_d * (al2 * dxT2 + be2 * dxS2), den1, rcp1);        // ip = 1, kr = 1
      double sumy = 0.0;
      sumy = sumy - x_div(aby[0] * csu_s * (m_c - ms1) * (al1 * dyTs1 + be1 * dySs1), yd_c, rd_c);   // jq = 0, kr = 0
      sumy = sumy - x_div(aby[2] * csu_s * (mc2 - ms2) * (al2 * dyTs2 + be2 * dySs2), den1, rcp1);   // jq = 0, kr = 1
      sumy = sumy - x_div(aby[1] * csu_n * (mn1 - m_c) * (al1 * dyT1 + be1 * dyS1), yd_c, rd_c);     // jq = 1, kr = 0
      sumy = sumy - x_div(aby[3] * csu_n * (mn2 - mc2) * (al2 * dyT2 + be2 * dyS2), den1, rcp1);     // jq = 1, kr = 1
      dfbi = dxt4r * sumx + dyt4r_cstr * sumy;
      yu_c = den1; ru_c = rcp1;   // the own column's drodz above level s+1
    }
    if (s == kz) dfb = btf;   // tracer.F:1061 (after the interior values are formed)
    const double DIFF_Tz = (dfb_up - dfb) * ddztr * aidif1 + (dfbi_up - dfbi) * ddztr;
    const double D = DIFF_Tx + DIFF_Ty + DIFF_Tz;
    if (owned) bst(b_D, lb, XOC(s, 0), D);
    dfb_up = dfb; dfbi_up = dfbi;
    dz_c = dz_dn; dz_s = dzs_dn; dz_n = dzn_dn;
    mc1 = mc2; ms1 = ms2; mn1 = mn2;
    al1 = al2; be1 = be2;
    dxT1 = dxT2; dxS1 = dxS2; dyT1 = dyT2; dyS1 = dyS2; dyTs1 = dyTs2; dySs1 = dySs2;
    dzT0 = dzT1; dzS0 = dzS1; dzTs0 = dzTs1; dzSs0 = dzSs1; dzTn0 = dzTn1; dzSn0 = dzSn1;
  }
#undef XOFJ
}

// ===========================================================================
// pass B in the reference's order: explicit update (tracer.F:1114-1127), invtri (invtri.F:57-110).  `ework`: the wave's LDS
// scratch, e(k) and z(k) as in colupd_wave; t(tau+1) of the column is left in zwork[k][lane] (k = 1..km) for the convective
// walk of the same workgroup.
// ===========================================================================
__device__ __forceinline__ void colx_upd_wave(const uvic_ctx &c, const ColxOut &o, double *ework, int code, int n1, int nloc) {
  UV_DIMS(c);
  const int lane = threadIdx.x;
  const int i = COL_LANE_I(code), r = COL_LANE_R(code);
  LaneTable t_dtxcel, t_dztur, t_dztlr, t_dztr;
  t_dtxcel.load(c.dtxcel, km); t_dztur.load(c.dztur, km); t_dztlr.load(c.dztlr, km); t_dztr.load(c.dztr, km);
  if (!COL_LANE_OWNED(code)) return;
  double *tp = c.t_taup1 + (size_t)(n1 - 1) * N3;
  const double *source = 0;
  if (c.src && c.itrc[n1 - 1] != 0) source = c.src + (size_t)(c.itrc[n1 - 1] - 1) * N3;
  const bool has_src = source != 0;
  const int kz = c.kmt[X2(i, r)];
  const double cstdyt2r = c.cstdyt2r[r - 1];
  const int rowstride = imt * km;
  const unsigned lb = (unsigned)((r - 2) * rowstride + (i - 1)) * 8u;
  const double topbc = c.stf[X2(i, r) + (size_t)(n1 - 1) * imt * jmt], botbc = c.btf[X2(i, r) + (size_t)(n1 - 1) * imt * jmt];
  const double aidif = c.aidif, eps = 1.e-30;
  const int kb = imax(2, kz);
  const int ic = (i == 2) ? imt : ((i == imt - 1) ? 1 : 0);
  double *zwork = ework + (size_t)(km + 1) * 64;
  const brsrc b_tm = mkbuf(c.t_taum1 + (size_t)(n1 - 1) * N3, N3 * 8), b_tp = mkbuf(tp, N3 * 8);
  const brsrc b_ax = mkbuf(o.adv_x + (size_t)nloc * N3, N3 * 8), b_az = mkbuf(o.adv_z + (size_t)nloc * N3, N3 * 8);
  const brsrc b_fn = mkbuf(o.fn + (size_t)nloc * N3, N3 * 8), b_D = mkbuf(o.dif + (size_t)nloc * N3, N3 * 8);
  const brsrc b_src = mkbuf(has_src ? source : (const double *)(o.dif + (size_t)nloc * N3), N3 * 8), b_dcb = mkbuf(c.diff_cbt, N3 * 8);
  double bet = 0.0, zprev = 0.0, cprev = 0.0, dcb_up = 0.0;
  for (int k = 1; k <= km; ++k) {
    const double m_c = bld(b_tm, lb, XOC(k, 0));
    const double D = bld(b_D, lb, XOC(k, 0)), ax = bld(b_ax, lb, XOC(k, 0)), az = bld(b_az, lb, XOC(k, 0));
    const double fn_n = bld(b_fn, lb, XOC(k, 0)), fn_s = bld(b_fn, lb, XOC(k, -1));
    const double sv = bld(b_src, lb, XOC(k, 0)), dcb = bld(b_dcb, lb, XOC(k, 0));
    const double mk = (k <= kz) ? 1.0 : 0.0;
    const double ADV_Ty = (fn_n - fn_s) * cstdyt2r;                      // fdift.h:31-32
    const double tdt = c.c2dtts * t_dtxcel.at(k - 1);
    const double z = m_c + tdt * (D - ax - ADV_Ty - az + (has_src ? sv : 0.0)) * mk;
    const int kp1 = imin(k + 1, km);
    const double factu = t_dztur.at(k - 1) * tdt * aidif, factl = t_dztlr.at(k - 1) * tdt * aidif;
    double a = -((k == 1) ? dcb : dcb_up) * factu * mk;
    double cc = -dcb * factl * ((kp1 <= kz) ? 1.0 : 0.0);
    double f = z * mk;
    if (k == 1) a = 0.0;
    if (k == km) cc = 0.0;
    const double b = 1.0 - a - cc;
    if (k == 1) f = z + topbc * tdt * t_dztr.at(0) * aidif * mk;
    if (k == kb) f = z - botbc * tdt * t_dztr.at(k - 1) * aidif * mk;
    double znew;
    if (k == 1) {
      bet = mk / (b + eps);
      znew = f * bet;
    } else {
      const double e = cprev * bet;
      ework[(size_t)k * 64 + lane] = e;
      bet = mk / (b - a * e + eps);
      znew = (f - a * zprev) * bet;
    }
    zwork[(size_t)k * 64 + lane] = znew;
    zprev = znew;
    cprev = cc;
    dcb_up = dcb;
  }
  double znext = zprev;
  bst(b_tp, lb, XOC(km, 0), znext);
  if (ic) tp[X3(ic, km, r)] = znext;
  for (int k = km - 1; k >= 1; --k) {
    const double zk = zwork[(size_t)k * 64 + lane] - ework[(size_t)(k + 1) * 64 + lane] * znext;
    bst(b_tp, lb, XOC(k, 0), zk);
    if (ic) tp[X3(ic, k, r)] = zk;
    zwork[(size_t)k * 64 + lane] = zk;
    znext = zk;
  }
}
#undef XOC
#undef XOF
#endif  // __HIPCC__
}  // namespace uvic
#endif
