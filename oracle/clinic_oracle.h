/* oracle/clinic_oracle.h -- CPU restatement of the baroclinic momentum step.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call it, and only as the checker.  Pinned against the reference's own
 * `state`, `adv_vel`, `setvbc` and `clinic` compiled into oracle/_ref (configuration "m2" of
 * oracle/build_ref.py) by tests/test_clinic.py.
 *
 * Layout as in uvic_oracle.h: Fortran order, i fastest; cell fields (imt,km,jmt), vertical-face fields
 * (imt,km+1,jmt) with face index 0..km, 2-D fields (imt,jmt); every field carries all jmt rows.  The two
 * velocity components are separate arrays: u[0] = zonal, u[1] = meridional.
 */
#ifndef UVIC_CLINIC_ORACLE_H
#define UVIC_CLINIC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_mom {
  int imt, jmt, km;
  /* scalars: scalar.h c2dtuv, grav, rho0r; vmixc.h kappa_m (O_constvmix: visc_cbu, u09/mom/vmixc.F:85); scalar.h cdbot */
  double c2dtuv, grav, rho0r, kappa_m, cdbot;
  /* metrics (grdvar.h) */
  const double *dxur, *dxu2r, *dxtr, *dxmetr, *duw, *due;          /* (imt) */
  const double *dyur, *dyu2r, *dyu4r, *dytr, *csur, *cst, *dus, *dun, *csudyu2r; /* (jmt) */
  const double *advmet, *am3, *am4;                                /* (jmt,2), (jmt), (jmt,2) */
  const double *dzt, *dztr, *dzt2r;                                /* (km) */
  const double *dzw, *dzwr;                                        /* (0:km) */
  const int *kmu;                                                  /* (imt,jmt) */
  const double *umask;                                             /* (imt,km,jmt) */
  const double *hr;                                                /* (imt,jmt) emode.h */
  const double *cori;                                              /* (imt,jmt,2) */
  const double *visc_ceu, *amc_north, *amc_south;                  /* (imt,km,jmt) hmixc.h, O_anisotropic_viscosity */
  const double *adv_vet, *adv_vnt;                                 /* (imt,km,jmt) */
  const double *adv_vbt;                                           /* (imt,km+1,jmt) */
  const double *smf;                                               /* (imt,jmt,2) */
  const double *rho;                                               /* (imt,km,jmt) */
  const double *u_tau[2], *u_taum1[2];                             /* (imt,km,jmt) */
  double *u_taup1[2];
  double *zu;                                                      /* (imt,jmt,2) */
  /* intermediates, exposed for stage-by-stage comparison */
  double *bmf;                                                     /* (imt,jmt,2) */
  double *adv_veu, *adv_vnu;                                       /* (imt,km,jmt) */
  double *adv_vbu;                                                 /* (imt,km+1,jmt) */
  double *grad_p;                                                  /* (imt,km,jmt,2) */
} orc_mom;

/* source/mom/state.F:1-41 as called at u09/mom/loadmw.F:154: rho on rows js..je, i = 1..imt */
void orc_state(int imt, int jmt, int km, const double *t, const double *s, const double *to, const double *so,
               const double *c, double *rho, int js, int je);
/* source/mom/adv_vel.F:150-231: adv_vnu, adv_veu, adv_vbu from the T-cell face velocities */
void orc_adv_vel_u(orc_mom *m);
/* u09/mom/setvbc.F:170-194: bottom drag bmf from u(taum1) at level kmu */
void orc_bmf(orc_mom *m);
/* u09/mom/clinic.F:24-560 for the option set of build_ref.py's "m2" without the polar filter */
void orc_clinic(orc_mom *m);
/* u09/mom/loadmw.F:590-667 add_ext_mode for one time level: psi (imt,jmt); u1, u2 (imt,km,jmt) in place */
void orc_add_ext_mode(const orc_mom *m, const double *psi, double *u1, double *u2);
/* u09/mom/clinic.F:816-895 (isbcu) and :729-811 (asbcu): accumulate u(tau) into two sbc planes (imt,jmt) each */
void orc_isbcu(const orc_mom *m, double *sbc_u, double *sbc_v, int osegs, int osege, double rts, const int *kmt);
void orc_asbcu(const orc_mom *m, double *sbc_u, double *sbc_v, int osegs, int osege, double rts, const int *kmt);

#ifdef __cplusplus
}
#endif
#endif
