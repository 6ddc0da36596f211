// kernels_clinic.hpp -- the baroclinic momentum step:
//   state    /root/reference/source/mom/state.F:1-41 (as called at u09/mom/loadmw.F:154)
//   clinic   /root/reference/updates/09/source/mom/clinic.F:24-560 with the statement functions of
//            /root/reference/updates/09/source/mom/fdifm.h, for the option set of run/mk.in:
//            O_consthmix O_constvmix O_anisotropic_viscosity O_stream_function O_cyclic (explicit Coriolis
//            term, explicit vertical friction, no O_pressure_gradient_average), one memory window
//   plus what `clinic` takes from its neighbours in mom.F's loop and no other routine needs:
//            the U-cell advective velocities (source/mom/adv_vel.F:150-231), the bottom drag
//            (u09/mom/setvbc.F:170-194), and isbcu/asbcu (clinic.F:729-895)
//   add_ext_mode  /root/reference/updates/09/source/mom/loadmw.F:590-667 (O_stream_function): the external mode added
//            to the internal-mode velocities at the start of a step (device-resident velocities)
//
// Same expressions and evaluation order as the reference (the library is built -ffp-contract=off): results
// are bit-identical.  Where the reference builds whole arrays first (adv_veu, adv_fe, diff_fb, ...), a cell
// here evaluates the same expressions for the faces it needs; a face shared by two cells is evaluated twice
// with the same operands.
//
// The three passes of clinic (the launch geometry is in uvic_gpu.hip; the polar filter filuv that follows them is in
// kernels_filter.hpp):
//   clinic_gradp_column   lane per column: hydrostatic pressure gradient, summed downward
//   clinic_tend_cell      thread per cell: the tendency of both components into u(tau+1)
//   clinic_finish_column  lane per column: zu, the leapfrog update, removal of the vertical mean, cyclic images
#ifndef UVIC_KERNELS_CLINIC_HPP
#define UVIC_KERNELS_CLINIC_HPP

#include "kernels_isopyc.hpp"
#include "uvic_mom_ctx.h"

namespace uvic {

// the column marches read UV_KB levels at a time: the loads of a batch are independent, the sums that use them keep
// the reference's order
#define UV_KB 8

// rho(i,k,j), state.F:22-28; rows 2..jmt, all columns
UVIC_DEV void state_cell(const uvic_mom_ctx &m, int i, int k, int j) {
  UV_DIMS(m);
  m.rho[X3(i, k, j)] = eos_dens(m.c, km, m.t_tau[X3(i, k, j)] - m.to[k - 1], m.s_tau[X3(i, k, j)] - m.so[k - 1], k);
}

// setvbc.F:170-194 for one column and component n (1,2)
UVIC_DEV double mom_bmf(const uvic_mom_ctx &m, int i, int j, int n) {
  UV_DIMS(m);
  const int kz = m.kmu[X2(i, j)];
  if (m.cdbot == 0.0 || kz == 0) return 0.0;
  const double a = m.um1[X3(i, kz, j)], b = m.um2[X3(i, kz, j)];
  const double uvmag = sqrt(a * a + b * b);
  return m.cdbot * (n == 1 ? a : b) * uvmag;
}

// adv_vel.F:168-177 / :190-199 / :221-225
UVIC_DEV double mom_vnu(const uvic_mom_ctx &m, int i, int k, int j) {
  UV_DIMS(m);
  const double *vnt = m.adv_vnt;
  return ((vnt[X3(i, k, j)] * m.duw[i - 1] + vnt[X3(i + 1, k, j)] * m.due[i - 1]) * m.dus[j] +
          (vnt[X3(i, k, j + 1)] * m.duw[i - 1] + vnt[X3(i + 1, k, j + 1)] * m.due[i - 1]) * m.dun[j - 1]) *
         m.dytr[j] * m.dxur[i - 1];
}
UVIC_DEV double mom_veu(const uvic_mom_ctx &m, int i, int k, int j) {   // i = 1 is the image of imt-1 (setbcx, adv_vel.F:201)
  UV_DIMS(m);
  if (i == 1) i = imt - 1;
  const double *vet = m.adv_vet;
  return ((vet[X3(i, k, j)] * m.dus[j - 1] + vet[X3(i, k, j + 1)] * m.dun[j - 1]) * m.duw[i] +
          (vet[X3(i + 1, k, j)] * m.dus[j - 1] + vet[X3(i + 1, k, j + 1)] * m.dun[j - 1]) * m.due[i - 1]) *
         m.dyur[j - 1] * m.dxtr[i];
}
UVIC_DEV double mom_vbu(const uvic_mom_ctx &m, int i, int k, int j) {   // k = 0..km
  UV_DIMS(m);
  const double *vbt = m.adv_vbt;
  const double dyn = m.dun[j - 1] * m.cst[j], dys = m.dus[j - 1] * m.cst[j - 1], dyr = m.dyur[j - 1] * m.csur[j - 1];
  const double asw = m.duw[i - 1] * dys, anw = m.duw[i - 1] * dyn, ase = m.due[i - 1] * dys, ane = m.due[i - 1] * dyn;
  return dyr * m.dxur[i - 1] *
         (vbt[XF(i, k, j)] * asw + vbt[XF(i + 1, k, j)] * ase + vbt[XF(i, k, j + 1)] * anw + vbt[XF(i + 1, k, j + 1)] * ane);
}

// clinic.F:119-186: grad_p(i,1:km,j,1:2) for one column, i = 2..imt-1, and the cyclic images setbcx makes of
// columns 2 and imt-1 (the value the reference first computes at i = 1 is overwritten by them, :182-185)
UVIC_DEV void clinic_gradp_column(const uvic_mom_ctx &m, int i, int j) {
  UV_DIMS(m);
  const double *rho = m.rho;
  double *gp = m.grad_p;
  const double p5 = 0.5;
  double g1, g2;
  {
    const double fxa = m.grav_rho0r * m.dzw[0] * m.csur[j - 1];
    const double fxb = m.grav_rho0r * m.dzw[0] * m.dyu2r[j - 1];
    const double t1 = rho[X3(i + 1, 1, j + 1)] - rho[X3(i, 1, j)];
    const double t2 = rho[X3(i, 1, j + 1)] - rho[X3(i + 1, 1, j)];
    g1 = (t1 - t2) * fxa * m.dxu2r[i - 1];
    g2 = (t1 + t2) * fxb;
  }
  const double fxa = m.grav_rho0r * m.csur[j - 1] * p5;
  const double fxb = m.grav_rho0r * m.dyu4r[j - 1];
  // tempik(.,k,.) = rho(.,k-1,.) + rho(.,k,.) at the four corners; the loads of UV_KB levels are issued together
  double r00 = rho[X3(i, 1, j)], r10 = rho[X3(i + 1, 1, j)], r01 = rho[X3(i, 1, j + 1)], r11 = rho[X3(i + 1, 1, j + 1)];
  auto put = [&](int k) {
    gp[X3(i, k, j)] = g1;
    gp[X3(i, k, j) + N3] = g2;
    if (i == 2) { gp[X3(imt, k, j)] = g1; gp[X3(imt, k, j) + N3] = g2; }
    if (i == imt - 1) { gp[X3(1, k, j)] = g1; gp[X3(1, k, j) + N3] = g2; }
  };
  put(1);
  for (int k0 = 2; k0 <= km; k0 += UV_KB) {
    double n00[UV_KB], n10[UV_KB], n01[UV_KB], n11[UV_KB];
#pragma unroll
    for (int q = 0; q < UV_KB; ++q) {
      const int k = (k0 + q <= km) ? k0 + q : km;
      n00[q] = rho[X3(i, k, j)]; n10[q] = rho[X3(i + 1, k, j)]; n01[q] = rho[X3(i, k, j + 1)]; n11[q] = rho[X3(i + 1, k, j + 1)];
    }
#pragma unroll
    for (int q = 0; q < UV_KB; ++q) {
      const int k = k0 + q;
      if (k <= km) {
        const double t1 = (r11 + n11[q]) - (r00 + n00[q]);
        const double t2 = (r01 + n01[q]) - (r10 + n10[q]);
        g1 = g1 + fxa * (t1 - t2) * m.dzw[k - 1] * m.dxu2r[i - 1];
        g2 = g2 + fxb * (t1 + t2) * m.dzw[k - 1];
        r00 = n00[q]; r10 = n10[q]; r01 = n01[q]; r11 = n11[q];
        put(k);
      }
    }
  }
}

// clinic.F:188-356: the tendency of u and v of one cell, stored in u(tau+1) as the reference does
UVIC_DEV void clinic_tend_cell(const uvic_mom_ctx &m, int i, int k, int j) {
  UV_DIMS(m);
  const size_t N2 = (size_t)imt * jmt;
  const int kb = m.kmu[X2(i, j)];
  if (k > kb) {   // umask = 0: the reference multiplies a finite sum by zero
    m.up1[X3(i, k, j)] = 0.0;
    m.up2[X3(i, k, j)] = 0.0;
    return;
  }
  const double p5 = 0.5;
  const double csudxur = m.csur[j - 1] * m.dxur[i - 1];
  const double csudxu2r = m.csur[j - 1] * m.dxur[i - 1] * p5;
  const double am_e = m.visc_ceu[X3(i, k, j)] * m.csur[j - 1] * m.dxtr[i];
  const double am_w = m.visc_ceu[X3(i - 1, k, j)] * m.csur[j - 1] * m.dxtr[i - 1];
  const double veu_e = mom_veu(m, i, k, j), veu_w = mom_veu(m, i - 1, k, j);
  const double vnu_n = mom_vnu(m, i, k, j), vnu_s = mom_vnu(m, i, k, j - 1);
  const double vbu_t = mom_vbu(m, i, k - 1, j), vbu_b = mom_vbu(m, i, k, j);
  const double amn = m.amc_north[X3(i, k, j)], ams = m.amc_south[X3(i, k, j)];
  const double ut1c = m.ut1[X3(i, k, j)], ut2c = m.ut2[X3(i, k, j)];
  for (int n = 1; n <= 2; ++n) {
    const double *ut = (n == 1) ? m.ut1 : m.ut2, *um = (n == 1) ? m.um1 : m.um2;
    const double *umo = (n == 1) ? m.um2 : m.um1;
    const double utc = (n == 1) ? ut1c : ut2c, uto = (n == 1) ? ut2c : ut1c;
    const double umc = um[X3(i, k, j)];
    // horizontal fluxes through the east and west faces (clinic.F:196-236)
    const double adv_fe_e = veu_e * (utc + ut[X3(i + 1, k, j)]);
    const double adv_fe_w = veu_w * (ut[X3(i - 1, k, j)] + utc);
    const double diff_fe_e = am_e * (um[X3(i + 1, k, j)] - umc);
    const double diff_fe_w = am_w * (umc - um[X3(i - 1, k, j)]);
    // vertical fluxes through the top (k-1) and bottom (k) faces (clinic.F:278-314)
    double adv_fb_t, adv_fb_b, diff_fb_t, diff_fb_b;
    if (k == 1) {
      adv_fb_t = vbu_t * (utc + utc);
      diff_fb_t = m.smf[X2(i, j) + (size_t)(n - 1) * N2];
    } else {
      adv_fb_t = vbu_t * (ut[X3(i, k - 1, j)] + utc);
      diff_fb_t = m.kappa_m * m.dzwr[k - 1] * (um[X3(i, k - 1, j)] - umc);
    }
    if (k == km) {
      adv_fb_b = vbu_b * utc;
      diff_fb_b = 0.0;
    } else {
      adv_fb_b = vbu_b * (utc + ut[X3(i, k + 1, j)]);
      diff_fb_b = m.kappa_m * m.dzwr[k] * (umc - um[X3(i, k + 1, j)]);
    }
    if (k == kb) diff_fb_b = mom_bmf(m, i, j, n);   // diff_fb(i,kb,j) = bmf, clinic.F:307
    const double DIFF_Ux = (diff_fe_e - diff_fe_w) * csudxur;
    const double DIFF_Uy = amn * (um[X3(i, k, j + 1)] - umc) - ams * (umc - um[X3(i, k, j - 1)]);
    const double DIFF_Uz = (diff_fb_t - diff_fb_b) * m.dztr[k - 1];
    const double DIFF_metric = m.am3[j - 1] * umc + m.am4[(j - 1) + (size_t)(n - 1) * jmt] * m.dxmetr[i - 1] *
                                                      (umo[X3(i + 1, k, j)] - umo[X3(i - 1, k, j)]);
    const double ADV_Ux = (adv_fe_e - adv_fe_w) * csudxu2r;
    const double ADV_Uy = (vnu_n * (utc + ut[X3(i, k, j + 1)]) - vnu_s * (ut[X3(i, k, j - 1)] + utc)) * m.csudyu2r[j - 1];
    const double ADV_Uz = (adv_fb_t - adv_fb_b) * m.dzt2r[k - 1];
    const double ADV_metric = m.advmet[(j - 1) + (size_t)(n - 1) * jmt] * ut1c * uto;
    const double CORIOLIS = m.cori[X2(i, j) + (size_t)(n - 1) * N2] * uto;
    const double source = 0.0;
    const double tend = (DIFF_Ux + DIFF_Uy + DIFF_Uz + DIFF_metric - ADV_Ux - ADV_Uy - ADV_Uz + ADV_metric -
                         m.grad_p[X3(i, k, j) + (size_t)(n - 1) * N3] + CORIOLIS + source);
    ((n == 1) ? m.up1 : m.up2)[X3(i, k, j)] = tend;
  }
}

// clinic.F:376-485 for one column: zu, u(tau+1) = u(tau-1) + c2dtuv*tendency, minus its vertical mean, cyclic images.
// Both components march together (their loads share a batch); each sum keeps the reference's order.
UVIC_DEV void clinic_finish_column(const uvic_mom_ctx &m, int i, int j) {
  UV_DIMS(m);
  const size_t N2 = (size_t)imt * jmt;
  const int kb = m.kmu[X2(i, j)];
  const double hr = m.hr[X2(i, j)];
  double zu1 = 0.0, zu2 = 0.0, baru1 = 0.0, baru2 = 0.0;
  for (int k0 = 1; k0 <= km; k0 += UV_KB) {
    double td1[UV_KB], td2[UV_KB], a1[UV_KB], a2[UV_KB];
#pragma unroll
    for (int q = 0; q < UV_KB; ++q) {
      const size_t x = X3(i, (k0 + q <= km) ? k0 + q : km, j);
      td1[q] = m.up1[x]; td2[q] = m.up2[x];
      a1[q] = m.um1[x]; a2[q] = m.um2[x];
    }
#pragma unroll
    for (int q = 0; q < UV_KB; ++q) {
      const int k = k0 + q;
      if (k <= km) {
        const double dz = m.dzt[k - 1];
        zu1 = zu1 + td1[q] * dz;
        zu2 = zu2 + td2[q] * dz;
        const double v1 = a1[q] + m.c2dtuv * td1[q], v2 = a2[q] + m.c2dtuv * td2[q];
        baru1 = baru1 + v1 * dz;
        baru2 = baru2 + v2 * dz;
        m.up1[X3(i, k, j)] = v1;
        m.up2[X3(i, k, j)] = v2;
      }
    }
  }
  m.zu[X2(i, j)] = zu1 * hr;
  m.zu[X2(i, j) + N2] = zu2 * hr;
  baru1 = baru1 * hr;
  baru2 = baru2 * hr;
  for (int k0 = 1; k0 <= km; k0 += UV_KB) {
    double w1[UV_KB], w2[UV_KB];
#pragma unroll
    for (int q = 0; q < UV_KB; ++q) {
      const size_t x = X3(i, (k0 + q <= km) ? k0 + q : km, j);
      w1[q] = m.up1[x]; w2[q] = m.up2[x];
    }
#pragma unroll
    for (int q = 0; q < UV_KB; ++q) {
      const int k = k0 + q;
      if (k <= km) {
        const double mask = (k <= kb) ? 1.0 : 0.0;
        const double v1 = w1[q] - mask * baru1, v2 = w2[q] - mask * baru2;
        m.up1[X3(i, k, j)] = v1;
        m.up2[X3(i, k, j)] = v2;
        if (i == 2) { m.up1[X3(imt, k, j)] = v1; m.up2[X3(imt, k, j)] = v2; }
        if (i == imt - 1) { m.up1[X3(1, k, j)] = v1; m.up2[X3(1, k, j)] = v2; }
      }
    }
  }
}

// add_ext_mode, u09/mom/loadmw.F:627-667 (O_stream_function): the external mode of one U column from the stream function
// around it, added to every level, land masked, cyclic images (setbcx, :664-666).  Rows 1..jmt-1.
UVIC_DEV void add_ext_mode_column(const uvic_mom_ctx &m, int i, int j, const double *psi, double *u1, double *u2) {
  UV_DIMS(m);
  const double diag1 = psi[X2(i + 1, j + 1)] - psi[X2(i, j)];
  const double diag0 = psi[X2(i, j + 1)] - psi[X2(i + 1, j)];
  const double hr = m.hr[X2(i, j)];
  const double ext1 = -(diag1 + diag0) * m.dyu2r[j - 1] * hr;
  const double ext2 = (diag1 - diag0) * m.dxu2r[i - 1] * hr * m.csur[j - 1];
  const int kb = m.kmu[X2(i, j)];
  for (int k0 = 1; k0 <= km; k0 += UV_KB) {
    double a[UV_KB], b[UV_KB];
#pragma unroll
    for (int q = 0; q < UV_KB; ++q) {
      const size_t x = X3(i, (k0 + q <= km) ? k0 + q : km, j);
      a[q] = u1[x]; b[q] = u2[x];
    }
#pragma unroll
    for (int q = 0; q < UV_KB; ++q) {
      const int k = k0 + q;
      if (k <= km) {
        const double mask = (k <= kb) ? 1.0 : 0.0;
        const double v1 = (a[q] + ext1) * mask, v2 = (b[q] + ext2) * mask;
        u1[X3(i, k, j)] = v1; u2[X3(i, k, j)] = v2;
        if (i == 2) { u1[X3(imt, k, j)] = v1; u2[X3(imt, k, j)] = v2; }
        if (i == imt - 1) { u1[X3(1, k, j)] = v1; u2[X3(1, k, j)] = v2; }
      }
    }
  }
}

// isbcu (clinic.F:853-892) and asbcu (:765-810) for one column of one row; flags: bit 0 osegs, bit 1 osege
UVIC_DEV void clinic_sbcu_cell(const uvic_mom_ctx &m, int i, int j, int flags, double rts) {
  UV_DIMS(m);
  const bool wet = m.kmt[X2(i, j)] != 0;
  const double p25 = 0.25;
  double gu = m.sbc_gu[X2(i, j)], gv = m.sbc_gv[X2(i, j)], su = m.sbc_su[X2(i, j)], sv = m.sbc_sv[X2(i, j)];
  if ((flags & 1) && wet) { gu = 0.0; gv = 0.0; su = 0.0; sv = 0.0; }
  gu = gu + m.ut1[X3(i, 2, j)];
  gv = gv + m.ut2[X3(i, 2, j)];
  su = su + p25 * (m.ut1[X3(i, 1, j)] + m.ut1[X3(i - 1, 1, j)] + m.ut1[X3(i, 1, j - 1)] + m.ut1[X3(i - 1, 1, j - 1)]);
  sv = sv + p25 * (m.ut2[X3(i, 1, j)] + m.ut2[X3(i - 1, 1, j)] + m.ut2[X3(i, 1, j - 1)] + m.ut2[X3(i - 1, 1, j - 1)]);
  if ((flags & 2) && wet) { gu = rts * gu; gv = rts * gv; su = rts * su; sv = rts * sv; }
  m.sbc_gu[X2(i, j)] = gu; m.sbc_gv[X2(i, j)] = gv; m.sbc_su[X2(i, j)] = su; m.sbc_sv[X2(i, j)] = sv;
}

}  // namespace uvic
#endif
