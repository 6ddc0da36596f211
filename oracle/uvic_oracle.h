/* oracle/uvic_oracle.h -- CPU restatement of the UVic 2.9 tracer time-step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call it, and only as the checker.  Parity pinning: the reference
 * has no tests or golden vectors of its own; this restatement
 * is pinned against the reference itself compiled into oracle/_ref (see
 * oracle/build_ref.py) and against fixtures under tests/golden/ generated from
 * that build (tests/golden/make_golden.py).
 *
 * Layout (package-wide): every array is Fortran order, i fastest.
 *   3-D cell fields      (imt, km,   jmt)
 *   vertical-face fields (imt, km+1, jmt)   level index 0..km (reference 0:km)
 *   2-D fields           (imt, jmt)
 *   tracers              (imt, km, jmt, nt)
 * The reference dimensions several arrays over a sub-range of rows
 * (jsmw:jemw, 1:jemw, ...; updates/09/source/mom/mw.h:246-316); here every
 * field carries all jmt rows and rows outside the reference's range are unused.
 */
#ifndef UVIC_ORACLE_H
#define UVIC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx {
  int imt, jmt, km, nt, nsrc;
  /* scalars */
  double c2dtts, aidif, diff_cet, diff_cnt;
  double slmxr, ahisop, athkdf;
  /* horizontal / vertical metrics (1-D) */
  const double *dxt, *dxtr, *dxu, *dxur, *dxt4r;          /* (imt) */
  const double *dyt, *dytr, *dyu, *dyur, *dyt4r;          /* (jmt) */
  const double *cst, *cstr, *csu, *cstdytr, *cstdyt2r, *csu_dyur; /* (jmt) */
  const double *dzt, *dztr, *dzt2r, *dztur, *dztlr;       /* (km) */
  const double *dzw, *dzwr;                               /* (0:km) */
  const double *dtxcel, *dtxsqr, *dztxcl;                 /* (km) */
  /* equation of state (source/mom/state.h:38) */
  const double *to, *so, *c;                              /* (km), (km), (km,9) */
  /* topography */
  const int *kmt;                                         /* (imt,jmt) */
  const double *tmask;                                    /* (imt,km,jmt) */
  /* isopycnal mixing inputs */
  const double *fisop;                                    /* (imt,jmt,km) reference layout */
  const double *addisop;                                  /* (imt,km,jmt) */
  /* state */
  const double *t_taum1, *t_tau;                          /* (imt,km,jmt,nt) */
  double *t_taup1;                                        /* (imt,km,jmt,nt) */
  /* advective velocities on T-cell faces */
  const double *adv_vet, *adv_vnt;                        /* (imt,km,jmt) */
  const double *adv_vbt;                                  /* (imt,km+1,jmt) */
  /* vertical diffusivity at cell bottoms, K33 already added */
  const double *diff_cbt;                                 /* (imt,km,jmt) */
  const double *stf, *btf;                                /* (imt,jmt,nt) */
  const double *src;                                      /* (imt,km,jmt,nsrc) or NULL */
  const int *itrc;                                        /* (nt) 1-based slot, 0 = none */
  /* isopyc products (outputs of orc_isopyc, inputs of orc_isoflux/orc_adv_flux) */
  double *alphai, *betai;                                 /* (imt,km,jmt) */
  double *ddxt, *ddyt;                                    /* (imt,km,jmt,2) */
  double *ddzt;                                           /* (imt,km+1,jmt,2) */
  double *Ai_ez, *Ai_nz, *Ai_bx, *Ai_by;                  /* (imt,km,jmt,2,2) */
  double *K11, *K22, *K33;                                /* (imt,km,jmt) */
  double *adv_vetiso, *adv_vntiso;                        /* (imt,km,jmt) */
  double *adv_vbtiso;                                     /* (imt,km+1,jmt) */
  /* per-tracer work arrays exposed for stage-by-stage comparison */
  double *adv_fe, *adv_fn;                                /* (imt,km,jmt) */
  double *adv_fb;                                         /* (imt,km+1,jmt) */
  double *diff_fe, *diff_fn;                              /* (imt,km,jmt) */
  double *diff_fb, *diff_fbiso;                           /* (imt,km+1,jmt) */
} orc_ctx;

/* isopyc.F:363-921,1140-1575: alpha/beta, gradients, Ai_*, K11/K22/K33, GM velocities */
void orc_isopyc(orc_ctx *c);
/* tracer_adv_flx.F:381-1028 (FCT branch) for tracer n (1-based) */
void orc_adv_flux(orc_ctx *c, int n);
/* tracer.F:925-1032: background diffusive fluxes diff_fe, diff_fn, diff_fb for tracer n */
void orc_diff_flux(orc_ctx *c, int n);
/* isopyc.F:923-1137: isopycnal fluxes added to diff_fe/diff_fn, diff_fbiso */
void orc_isoflux(orc_ctx *c, int n);
/* tracer.F:1053-1130: vertical b.c., source, explicit update into t_taup1(:,:,:,n) */
void orc_explicit_update(orc_ctx *c, int n);
/* invtri.F:1-115 on z(imt,km,jmt) rows js..je, columns is..ie (1-based, inclusive) */
void orc_invtri(const orc_ctx *c, double *z, const double *topbc, const double *botbc,
                const double *dcb, const double *tdt, int is, int ie, int js, int je);
/* convect.F:99-311 on ts(imt,km,jmt,nt) */
void orc_convct2(const orc_ctx *c, double *ts, int is, int ie, int js, int je);
/* cyclic boundary, util.F:789-814 */
void orc_setbcx(double *a, int imt, int n);
/* the transport part of `tracer` for all tracers: tracer.F:902-1209 */
void orc_tracer_transport(orc_ctx *c);

#ifdef __cplusplus
}
#endif
#endif
