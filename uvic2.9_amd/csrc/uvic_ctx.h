/* uvic_ctx.h -- the data contract of the tracer step on the device.
 *
 * One plain struct of dimensions, scalars and (device) pointers, passed by
 * value to every kernel.  Layout conventions (package-wide):
 * Fortran order, i fastest;  cell fields (imt,km,jmt);  vertical-face fields
 * (imt,km+1,jmt) with face index 0..km;  2-D fields (imt,jmt);  tracers
 * (imt,km,jmt,nt).  The reference dimensions many arrays over a sub-range of
 * rows (updates/09/source/mom/mw.h:246-316); here every field carries all jmt
 * rows and the C ABI maps the reference's row ranges onto them.
 */
#ifndef UVIC_CTX_H
#define UVIC_CTX_H

typedef struct uvic_ctx {
  int imt, jmt, km, nt, nsrc;
  /* scalars of the step */
  double c2dtts, aidif, diff_cet, diff_cnt;
  double slmxr, ahisop, athkdf;
  /* metrics */
  const double *dxt, *dxtr, *dxu, *dxur, *dxt4r;                  /* (imt) */
  const double *dyt, *dytr, *dyu, *dyur, *dyt4r;                  /* (jmt) */
  const double *cst, *cstr, *csu, *cstdytr, *cstdyt2r, *csu_dyur; /* (jmt) */
  const double *dzt, *dztr, *dzt2r, *dztur, *dztlr;               /* (km) */
  const double *dzw, *dzwr;                                       /* (0:km) */
  const double *dtxcel, *dtxsqr, *dztxcl;                         /* (km) */
  const double *to, *so, *c;                                      /* (km),(km),(km,9) */
  const int *kmt;                                                 /* (imt,jmt) */
  const double *tmask;                                            /* (imt,km,jmt) */
  const double *fisop;                                            /* (imt,jmt,km) */
  const double *addisop;                                          /* (imt,km,jmt) */
  const double *t_taum1, *t_tau;                                  /* (imt,km,jmt,nt) */
  double *t_taup1;
  const double *adv_vet, *adv_vnt;                                /* (imt,km,jmt) */
  const double *adv_vbt;                                          /* (imt,km+1,jmt) */
  double *diff_cbt;                                               /* (imt,km,jmt) incl. K33 */
  const double *stf, *btf;                                        /* (imt,jmt,nt) */
  const double *src;                                              /* (imt,km,jmt,nsrc) */
  const int *itrc;                                                /* (nt) */
  /* isopyc products */
  double *alphai, *betai;
  double *ddxt, *ddyt;                                            /* (imt,km,jmt,2) */
  double *ddzt;                                                   /* (imt,km+1,jmt,2) */
  double *Ai_ez, *Ai_nz, *Ai_bx, *Ai_by;                          /* (imt,km,jmt,2,2) */
  double *K11, *K22, *K33;
  double *adv_vetiso, *adv_vntiso;
  double *adv_vbtiso;                                             /* (imt,km+1,jmt) */
  /* device-only work space */
  const double *diff_cbt_bg;                                      /* (imt,km,jmt) before K33 */
  double *tot_e, *tot_n;                                          /* adv_v?t + adv_v?tiso */
  double *tot_b;                                                  /* (imt,km+1,jmt) */
  double *adv_x, *adv_z;                                          /* ADV_Tx, ADV_Tz (imt,km,jmt,nt) */
  double *RpY, *RmY;                                              /* y-limiter ratios (imt,km,jmt,nt) */
  double *fny;                                                    /* column kernels: HALF of the final limited flux through the north face (imt,km,jmt,nt) */
  /* convection: mixed segments found from T,S by convect_ts_column, applied to the other
   * tracers by convect_apply_cell.  cv_nseg (imt,jmt); cv_kt, cv_kb, cv_z (imt,km,jmt) */
  int *cv_nseg, *cv_kt, *cv_kb;
  double *cv_z;
  /* tracer-index shard handled by this context: global tracers n0+1 .. n0+nt_local */
  int n0, nt_local;
  /* latitude slab handled by this context (rows js..je are computed) */
  int js, je;
  /* 1: diff_cbt was uploaded with K33 already added (host vmixc); isopyc leaves it alone */
  int diff_cbt_given;
  /* inputs of adv_vel and vmixc (kernels_prep.hpp) */
  const double *u1, *u2;                                          /* (imt,km,jmt) u(:,:,:,1:2,tau) */
  const double *dxt2r, *dyt2r, *zw;                               /* (imt), (jmt), (km) */
  const double *tlat;                                             /* (imt,jmt) */
  const double *edrm2, *edrs2, *edrk1, *edro1;                    /* (imt,km,jmt) */
  double kappa_h, zetar, ogamma, gravrho0r;
  /* vmixc's two exponentials, tabulated on the host with the C library's exp (the one the reference calls):
   * vmix_e[(k-1)*km + k1-1] = exp((zw(k)-zw(k1))*zetar), vmix_d[k1-1] = 1 - exp(-zetar*zw(k1)); null: exp in the kernel */
  const double *vmix_e, *vmix_d;
  int vmix_dev;      /* 1: this step's diff_cbt is formed on the device (vmixc_cell) where its inputs are prepared */
  int no_landskip;   /* measurement: 1 = segments without ocean are marched like the others (UVIC_NO_LANDSKIP) */
  int prio;          /* bit 0: MOBI team waves at normal issue priority (UVIC_TEAM_PRIO0); bit 1: this launch at raised priority (the T,S passes) */
} uvic_ctx;

#endif
