/* uvic_mom_ctx.h -- the data contract of the baroclinic momentum step (`state` + `clinic`, SURVEY.md §8f
 * rank 4) on the device: dimensions, scalars and device pointers, passed by value to the kernels of
 * kernels_clinic.hpp.  Layout conventions as uvic_ctx.h (Fortran order, i fastest, all jmt rows); the two
 * velocity components are separate (imt,km,jmt) fields.
 */
#ifndef UVIC_MOM_CTX_H
#define UVIC_MOM_CTX_H

typedef struct uvic_mom_ctx {
  int imt, jmt, km;
  int js, je;                                                      /* U rows computed (2..jmt-1 for the whole grid) */
  double c2dtuv, grav_rho0r, kappa_m, cdbot;
  const double *dxur, *dxu2r, *dxtr, *dxmetr, *duw, *due;          /* (imt) grdvar.h */
  const double *dyur, *dyu2r, *dyu4r, *dytr, *csur, *cst, *dus, *dun, *csudyu2r; /* (jmt) */
  const double *advmet, *am3, *am4;                                /* (jmt,2), (jmt), (jmt,2) */
  const double *dzt, *dztr, *dzt2r;                                /* (km) */
  const double *dzw, *dzwr;                                        /* (0:km) */
  const double *to, *so, *c;                                       /* (km), (km), (km,9) state.h */
  const int *kmt, *kmu;                                            /* (imt,jmt) */
  const double *hr;                                                /* (imt,jmt) */
  const double *cori;                                              /* (imt,jmt,2) */
  const double *visc_ceu, *amc_north, *amc_south;                  /* (imt,km,jmt) */
  const double *adv_vet, *adv_vnt;                                 /* (imt,km,jmt) */
  const double *adv_vbt;                                           /* (imt,km+1,jmt) */
  const double *smf;                                               /* (imt,jmt,2) */
  const double *t_tau, *s_tau;                                     /* (imt,km,jmt): T and S at tau */
  double *rho;                                                     /* (imt,km,jmt) */
  const double *ut1, *ut2, *um1, *um2;                             /* u(tau), u(tau-1) */
  double *up1, *up2;                                               /* u(tau+1) */
  double *zu;                                                      /* (imt,jmt,2) */
  double *grad_p;                                                  /* (imt,km,jmt,2) */
  double *sbc_gu, *sbc_gv, *sbc_su, *sbc_sv;                       /* (imt,jmt): isbcu / asbcu accumulators */
} uvic_mom_ctx;

#endif
