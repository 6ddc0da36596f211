/* uvic_gpu.h -- C ABI of the MI355X-native UVic 2.9 ocean tracer time-step.
 *
 * This is the drop-in boundary of SURVEY.md §8(b).  The reference has no FFI:
 * its tracer path consists of Fortran-77 external procedures with implicit
 * interfaces that exchange all bulk data through COMMON blocks.  The entry
 * points below are what a Fortran `bind(C)` interface in an overlay `tracer.F`
 * binds (uvic2.9_amd/fortran/); each one cites the reference
 * interface it replaces (paths relative to /root/reference, "u09/" =
 * updates/09/source/).
 *
 * Conventions
 *   - plain C: pointers, sizes, int status (0 = ok); no C++ or torch types.
 *   - `integer` = int32_t, `real` = double (the reference is built -r8,
 *     run/mk.ver:51), `logical` = int32_t.
 *   - arrays are Fortran order, i fastest.  The device keeps every field over
 *     all jmt rows; `uvic_gpu_upload_rows` maps the reference's row ranges
 *     (jsmw:jemw, 1:jemw, jsmw:jmw; u09/mom/mw.h:246-316) onto them.
 *   - ownership: the caller owns every host array; device buffers belong to
 *     the handle between uvic_gpu_create and uvic_gpu_destroy.
 *   - all calls are synchronous with respect to the host unless stated.
 */
#ifndef UVIC_GPU_H
#define UVIC_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uvic_gpu uvic_gpu;

/* compile-time `parameter`s of the reference (u09/common/size.h:27-144,
 * u09/mom/mobi.h:104-142) become run-time dimensions */
typedef struct uvic_dims {
  int32_t imt, jmt, km, nt, nsrc, ntnpzd;
} uvic_dims;

/* device fields.  Shapes: C = (imt,km,jmt), F = (imt,km+1,jmt), S = (imt,jmt) */
enum uvic_field {
  /* 1-D metrics, u09 & source/common: grdvar.h, coord.h, accel.h */
  UVIC_F_DXT = 0, UVIC_F_DXTR, UVIC_F_DXU, UVIC_F_DXUR, UVIC_F_DXT4R,          /* (imt) */
  UVIC_F_DYT, UVIC_F_DYTR, UVIC_F_DYU, UVIC_F_DYUR, UVIC_F_DYT4R,              /* (jmt) */
  UVIC_F_CST, UVIC_F_CSTR, UVIC_F_CSU, UVIC_F_CSTDYTR, UVIC_F_CSTDYT2R, UVIC_F_CSU_DYUR,
  UVIC_F_DZT, UVIC_F_DZTR, UVIC_F_DZT2R, UVIC_F_DZTUR, UVIC_F_DZTLR,           /* (km) */
  UVIC_F_DZW, UVIC_F_DZWR,                                                    /* (0:km) */
  UVIC_F_DTXCEL, UVIC_F_DTXSQR, UVIC_F_DZTXCL,                                /* (km) */
  UVIC_F_TO, UVIC_F_SO, UVIC_F_C,             /* source/mom/state.h:38: (km),(km),(km,9) */
  UVIC_F_KMT,                                 /* int32 (imt,jmt), u09/common/levind.h:9 */
  UVIC_F_FISOP,                               /* (imt,jmt,km), u09/common/isopyc.h */
  UVIC_F_ADDISOP,                             /* C */
  UVIC_F_T_TAUM1, UVIC_F_T_TAU, UVIC_F_T_TAUP1, /* (imt,km,jmt,nt), u09/mom/mw.h:77 */
  UVIC_F_ADV_VET, UVIC_F_ADV_VNT,             /* C,  mw.h adv_vet(imt,km,jsmw:jmw), adv_vnt(imt,km,1:jmw) */
  UVIC_F_ADV_VBT,                             /* F,  mw.h adv_vbt(imt,0:km,jsmw:jmw) */
  UVIC_F_DIFF_CBT_BG,                         /* C,  vmixc.h diff_cbt BEFORE "+K33" (u09/mom/vmixc.F:182-188) */
  UVIC_F_STF, UVIC_F_BTF,                     /* (imt,jmt,nt), mw.h stf/btf(imt,1:jmw,nt) */
  UVIC_F_SRC,                                 /* (imt,km,jmt,nsrc), tracer.F:121 */
  UVIC_F_ITRC,                                /* int32 (nt), mw.h itrc */
  /* products of uvic_gpu_isopyc (downloadable; u09/common/isopyc.h:20-83) */
  UVIC_F_ALPHAI, UVIC_F_BETAI, UVIC_F_DDXT, UVIC_F_DDYT, UVIC_F_DDZT,
  UVIC_F_AI_EZ, UVIC_F_AI_NZ, UVIC_F_AI_BX, UVIC_F_AI_BY, UVIC_F_K11, UVIC_F_K22, UVIC_F_K33,
  UVIC_F_ADV_VETISO, UVIC_F_ADV_VNTISO, UVIC_F_ADV_VBTISO,
  UVIC_F_DIFF_CBT,                            /* C, background + K33 */
  /* inputs of uvic_gpu_adv_vel / uvic_gpu_vmixc */
  UVIC_F_U1, UVIC_F_U2,                       /* C, mw.h u(imt,km,jmw,1:2,tau): zonal, meridional velocity on U cells */
  UVIC_F_DXT2R, UVIC_F_DYT2R,                 /* (imt), (jmt) grdvar.h */
  UVIC_F_ZW,                                  /* (km) coord.h: depth of T-cell bottoms */
  UVIC_F_TLAT,                                /* S, grdvar.h tlat(imt,jmt) */
  UVIC_F_EDRM2, UVIC_F_EDRS2, UVIC_F_EDRK1, UVIC_F_EDRO1, /* C, u09/mom/tidal_kv.h edr*(imt,km,jmt) */
  /* baroclinic momentum step, uvic_gpu_state / uvic_gpu_clinic */
  UVIC_F_RHO,                                 /* C, mw.h rho(imt,km,jsmw:jmw): product of uvic_gpu_state */
  UVIC_F_UM1, UVIC_F_UM2,                     /* C, u(:,:,:,1:2,taum1) */
  UVIC_F_UP1, UVIC_F_UP2,                     /* C, u(:,:,:,1:2,taup1): internal-mode velocities, product of uvic_gpu_clinic */
  UVIC_F_ZU,                                  /* (imt,jmt,2) emode.h zu: vertically averaged forcing, product */
  UVIC_F_GRAD_P,                              /* (imt,km,jmt,2) mw.h grad_p, product */
  UVIC_F_SMF,                                 /* (imt,jmt,2) mw.h smf: wind stress (setvbc.F:163-164) */
  UVIC_F_KMU,                                 /* int32 (imt,jmt) levind.h */
  UVIC_F_HR,                                  /* S, emode.h hr: reciprocal depth of U columns */
  UVIC_F_CORI,                                /* (imt,jmt,2) grdvar.h */
  UVIC_F_VISC_CEU, UVIC_F_AMC_NORTH, UVIC_F_AMC_SOUTH, /* C, u09/common/hmixc.h (O_anisotropic_viscosity; a model without
                                               * that option fills them with am, amc_north(jrow), amc_south(jrow)) */
  UVIC_F_DXU2R, UVIC_F_DXMETR, UVIC_F_DUW, UVIC_F_DUE,                 /* (imt) grdvar.h */
  UVIC_F_DYU2R, UVIC_F_DYU4R, UVIC_F_CSUR, UVIC_F_DUS, UVIC_F_DUN, UVIC_F_CSUDYU2R, /* (jmt) grdvar.h */
  UVIC_F_ADVMET,                              /* (jmt,2) grdvar.h */
  UVIC_F_AM3, UVIC_F_AM4,                     /* (jmt), (jmt,2) hmixc.h */
  UVIC_F_SBC_GU, UVIC_F_SBC_GV, UVIC_F_SBC_SU, UVIC_F_SBC_SV, /* S: the sbc planes igu, igv (isbcu) and isu, isv (asbcu) */
  UVIC_F_SPSIN, UVIC_F_SPCOS,                 /* (imt) cpolar.h: rotation to polar-stereographic components in filuv */
  UVIC_F_PHI,                                 /* (jmt) coord.h: latitude of the U rows in radians (its sign, filuv.F:66-67) */
  UVIC_F_PSI,                                 /* (imt,jmt,2) emode.h psi: stream function at tau (,,1) and tau-1 (,,2) from `tropic` */
  UVIC_F_COUNT
};

/* scalars that change per step or per run (scalar.h, vmixc.h, hmixc.h, isopyc.h) */
typedef struct uvic_params {
  double c2dtts;   /* 2*dtts on leapfrog steps, dtts on mixing steps (source/mom/mom.F:108-148) */
  double aidif;    /* implicit fraction of vertical diffusion, control.in &mixing */
  double diff_cet, diff_cnt; /* u09/mom/hmixc.F:177-200 */
  double slmxr;    /* 1/slmx, u09/mom/isopyc.F:105 */
  double ahisop, athkdf; /* isopyc.F:79-83 */
  /* 0: UVIC_F_DIFF_CBT_BG holds diff_cbt before "+K33" and uvic_gpu_isopyc adds K33
   *    (device-resident stepping);
   * 1: the caller uploaded UVIC_F_DIFF_CBT as host vmixc left it, K33 included
   *    (u09/mom/vmixc.F:182-188; the host ran isopyc -> vmixc itself, mom.F:340-347) */
  int32_t diff_cbt_has_k33;
  int32_t pad_;
} uvic_params;

/* constants of the tidal-mixing scheme (u09/mom/tidal_kv.h, set in u09/mom/setmom.F:80-82) and the
 * background diffusivity (vmixc.h) */
typedef struct uvic_vmix_params {
  double kappa_h, zetar, ogamma, gravrho0r;
} uvic_vmix_params;

/* ---- MOBI biogeochemistry (option set C) --------------------- */
/* 1-based positions, 0 = absent.  `im`: position in the MOBI column vector
 * (imobi*, u09/mom/mobi.F:440-504); `is`: source slot (is*, u09/common/UVic_ESCM.F:1376-1483) */
typedef struct uvic_mobi_index {
  int32_t po4, phyt, phyt_phos, zoop, detr, detr_phos, dic, dic13, phytc13, zoopc13, detrc13, doc13, diazc13;
  int32_t dop, no3, don, diaz, din15, don15, phytn15, zoopn15, detrn15, diazn15, dfe, detrfe, alk, o2, c14;
} uvic_mobi_index;

/* COMMON /npzd_r/ after mobi_init (u09/mom/mobi.h:144-262, unit conversions
 * u09/mom/mobi.F:209-290, 432-438) and the vertical grid (coord.h, grdvar.h) */
typedef struct uvic_mobi_params {
  int32_t km, ntnpzd, nsrc, pad_;
  uvic_mobi_index im, is;
  int32_t tracer_of_mobi[40]; /* prognostic tracer index (1-based) of MOBI tracer m */
  int32_t slot_of_mobi[40];   /* source slot (1-based) of MOBI tracer m */
  int32_t itemp, isalt, idic, ialk, io2, ic14;
  double dtnpzd;
  double kw, kc, ki, tap, abio_P, bbio, cbio, nup, nup_D, nupt0, nupt0_D, gamma1, gbio, nuz, nud0, nudon0, nudop0;
  double redptn, redctn, redntp, redotc, redntc, diazntp, diazptn, kzoo, geZ;
  double zprefP, zprefDet, zprefZ, zprefDiaz;
  double kfe_D, kfemin, kfemax, knmin, knmax, pmax, thetamaxlo, thetamaxhi, alphamin, alphamax;
  double kfeleq, kfeorg, kfecol, mc, rfeton, iscr, jdiar, dbct_D, hdop, dfr, dfrt, pfr;
  double eps_assim, eps_recy, eps_excr, eps_nfix, eps_wcdeni, eps_bdeni0, capr;
  double wd[64], ztt[64], rcak[64], rcab[64];
  double zt[64], dzt[64], dztr[64];
} uvic_mobi_params;

/* 2-D/3-D inputs MOBI reads from other components' COMMON blocks
 * (u09/mom/tracer.F:370-390, 538-545): host arrays, copied by the call */
typedef struct uvic_mobi_forcing {
  double pi, radian;       /* ndcon, scalar.h */
  double relyr;            /* tmngr.h: year fraction -> month index and solar declination */
  double co2ccn;           /* cembm.h */
  const double *tlat;      /* (imt,jmt) grdvar.h */
  const double *dnswr;     /* (imt,jmt) u09/embm/atm.h:86 */
  const double *aice, *hice, *hsno; /* (imt,jmt) ice.h, time level 2 */
  const double *sg_bathy;  /* (imt,jmt,km) levind.h */
  const double *fe_atmdep; /* (imt,jmt,12) mobi.h fe_atmdep(imt,jmt,1,12) */
  const double *fe_hydr;   /* (imt,jmt,km) mobi.h */
} uvic_mobi_forcing;

/* The cpp options of u09/mom/mobi.F that differ between the option sets the reference can build, and
 * what they add to COMMON /npzd_r/.  Always on: O_mobi_o2, O_mobi_iron, O_carbon, O_mobi_alk, O_mobi_nitrogen.
 * `im`/`is`: position in tnpzd (imobi*) and source slot (is*) of every MOBI column tracer, 0 = not in this set, in the
 * order  po4 phyt phyt_phos zoop detr detr_phos dic dic13 phytc13 zoopc13 detrc13 doc13 diazc13 dop no3 don diaz
 *        din15 don15 phytn15 zoopn15 detrn15 diazn15 dfe detrfe caco3 diat sil opl diatn15 diatc13 caco3c13 */
#define UVIC_MOBI_NX 32
typedef struct uvic_mobi_options {
  int32_t n15, c13, caco3, silicon;   /* O_mobi_nitrogen_15, O_carbon_13, O_mobi_caco3, O_mobi_silicon */
  int32_t im[UVIC_MOBI_NX], is[UVIC_MOBI_NX];
  int32_t is_alk, is_o2, is_c14, pad_;
  double kc_c, dissk0, caprmax, kcapr;                                           /* O_mobi_caco3 (mobi.h) */
  double abiodiat, kfemin_Diat, kfemax_Diat, knmin_Diat, knmax_Diat, pmax_Diat;  /* O_mobi_silicon */
  double zprefDiat, nu_diat, nudt0, opl_disk0;
  double wc[64], wo[64];                                                         /* sinking of CaCO3 and opal per level */
} uvic_mobi_options;
/* uvic_gpu_set_mobi for any of those option sets.  With the flags of set C (n15 = c13 = 1, caco3 = silicon = 0) it is
 * uvic_gpu_set_mobi; otherwise the sources come from the general column kernel (csrc/kernels_mobi_gen.hpp). */
int uvic_gpu_set_mobi_opt(uvic_gpu *h, const uvic_mobi_params *p, const uvic_mobi_options *o, const uvic_mobi_forcing *f);

/* the same for callers without C structs (Fortran), in two calls: this one names the option set and is followed by
 * uvic_gpu_set_mobi_flat, which then goes through uvic_gpu_set_mobi_opt.  flags = n15, c13, caco3, silicon;
 * im, is: UVIC_MOBI_NX entries each in the order above; isx = is_alk, is_o2, is_c14; oscal = the 14 doubles of
 * uvic_mobi_options from kc_c to opl_disk0; wc, wo: km values each (ignored without the option) */
int uvic_gpu_mobi_options_flat(uvic_gpu *h, const int32_t *flags, const int32_t *im, const int32_t *is, const int32_t *isx,
                               const double *oscal, const double *wc, const double *wo, int km);

/* upload MOBI parameters and forcing; after this uvic_gpu_tracer computes the
 * source terms itself (replaces the column loop of u09/mom/tracer.F:355-545 with
 * mobi_driver/mobi_src/co2calc_SWS and the 14C source, tracer.F:853-867) instead
 * of reading UVIC_F_SRC as given */
int uvic_gpu_set_mobi(uvic_gpu *h, const uvic_mobi_params *p, const uvic_mobi_forcing *f);
/* the same for callers without C structs (Fortran): `idx` = im(28), is(28) in the
 * member order of uvic_mobi_index; `tracer_of_mobi`, `slot_of_mobi` (ntnpzd);
 * `itr` = itemp,isalt,idic,ialk,io2,ic14; `scal` = dtnpzd followed by the double
 * members of uvic_mobi_params from `kw` to `capr` in declaration order (1 + 59 values);
 * `prof` = wd,ztt,rcak,rcab,zt,dzt,dztr, km values each; `fsc` = pi,radian,relyr,co2ccn */
int uvic_gpu_set_mobi_flat(uvic_gpu *h, int km, int ntnpzd, int nsrc, const int32_t *idx, const int32_t *tracer_of_mobi,
                           const int32_t *slot_of_mobi, const int32_t *itr, const double *scal, const double *prof,
                           const double *fsc, const double *tlat, const double *dnswr, const double *aice,
                           const double *hice, const double *hsno, const double *sg_bathy, const double *fe_atmdep,
                           const double *fe_hydr);
/* what of the MOBI forcing changes every step, after uvic_gpu_set_mobi[_flat] has set the rest once: the fields
 * tracer.F:355-390 reads (downward shortwave, ice fraction and thickness, snow), relyr and the atmospheric CO2.
 * All four fields null: they stay (they change once per ocean segment), only relyr and co2ccn move on.  New fields
 * void what a look-ahead chain computed from the old ones. */
int uvic_gpu_set_mobi_step(uvic_gpu *h, double relyr, double co2ccn, const double *dnswr, const double *aice,
                           const double *hice, const double *hsno);
/* page-lock a host range that is uploaded from or downloaded into every step (the reference's COMMON blocks live as
 * long as the process); the memory stays the caller's */
int uvic_gpu_pin_host(uvic_gpu *h, void *ptr, int64_t bytes);
/* ends the registration (uvic_gpu_destroy does so for every range the handle registered) */
int uvic_gpu_unpin_host(uvic_gpu *h, void *ptr);
/* the source-term kernel alone (for parity tests): fills UVIC_F_SRC */
int uvic_gpu_mobi(uvic_gpu *h);

const char *uvic_gpu_last_error(void);
int uvic_gpu_abi_version(void);

/* allocate all device state for one model instance on `device` */
int uvic_gpu_create(uvic_gpu **h, const uvic_dims *dims, int device);
int uvic_gpu_destroy(uvic_gpu *h);

/* whole-field transfers; `count` elements starting at element `offset` */
int uvic_gpu_upload(uvic_gpu *h, int field, const void *host, int64_t offset, int64_t count);
int uvic_gpu_download(uvic_gpu *h, int field, void *host, int64_t offset, int64_t count);
/* host array dimensioned (imt, kdim, jlo:jhi [, extra]) -> device rows jlo..jhi */
int uvic_gpu_upload_rows(uvic_gpu *h, int field, const double *host, int jlo, int jhi);
int uvic_gpu_download_rows(uvic_gpu *h, int field, double *host, int jlo, int jhi);
/* level k (1-based) of tracer n (1-based; 1 for a plain cell field; 0 = every tracer, host (imt, jmt, nt)) of a cell field as contiguous (imt, jmt) planes.
 * With the state resident on the device the host needs T,S whole but of the other tracers only the surface level:
 * `set_sbc` reads t(i,1,j,n,taup1) (u09/mom/set_sbc.F:36-72) */
int uvic_gpu_download_level(uvic_gpu *h, int field, int n, int k, double *host);
/* number of elements and device address of a field (for zero-copy plumbing) */
int64_t uvic_gpu_field_elems(uvic_gpu *h, int field);
void *uvic_gpu_field_devptr(uvic_gpu *h, int field);
void *uvic_gpu_stream(uvic_gpu *h); /* hipStream_t the kernels are launched on */

int uvic_gpu_set_params(uvic_gpu *h, const uvic_params *p);
/* arithmetic of the transport kernels (env UVIC_EXACT names the mode at create):
 *   0 (default): T and S -- on whose bits every convective adjustment is decided, convect.F:189-255 -- in the
 *      reference's order of operations (bit-identical to the reference Fortran built without FMA contraction:
 *      tracer_adv_flx.F:500-999, isopyc.F:953-1108, invtri.F), the other tracers with the isopycnal coefficients
 *      folded once per step (agreement to rounding: <= 1e-13 relative after one step, <= 1e-12 after 100, tests);
 *   1: every tracer in the reference's order of operations (bit-identical);
 *   2: every tracer through the folded column kernels (T and S then agree to rounding only);
 *   3: as 0 with T and S through the row kernels instead of the exact column kernels (cross-check). */
int uvic_gpu_set_exact(uvic_gpu *h, int exact);
/* cross-check and tuning switches for tests and tools; the defaults are what the library is measured with.
 *   "nchunk" n: longitude chunks of the row kernels (0 = automatic);  "mobi_generic" 1: option set C through the
 *   general MOBI column kernel (call before uvic_gpu_set_mobi_opt);  "mobi_team" 0: one thread per column instead
 *   of four-wave teams;  "convect_onepass" 1: convct2 (convect.F:99-311) as one kernel over all tracers;
 *   "mobi_streams" 1: every look-ahead MOBI chain on one side stream;  "push_wait_ms" n: how long
 *   uvic_gpu_push_exchange waits for a peer.  Non-zero status for an unknown name.
 * The shipped library reads no other environment variable than UVIC_EXACT. */
int uvic_gpu_set_option(uvic_gpu *h, const char *name, int value);
/* work decomposition: this instance computes tracers n0+1..n0+nt_local and rows
 * js..je (1-based, inclusive); defaults: all tracers, rows 2..jmt-1 */
int uvic_gpu_set_shard(uvic_gpu *h, int n0, int nt_local, int js, int je);

/* replaces `call isopyc (joff, js, je, is, ie)`, source/mom/mom.F:340
 * (u09/mom/isopyc.F:466-557), plus the "+K33" of u09/mom/vmixc.F:182-188 */
int uvic_gpu_isopyc(uvic_gpu *h);
/* replaces the per-tracer loop of `tracer` (u09/mom/tracer.F:902-1167):
 * adv_flux (FCT), diffusive fluxes, isoflux, explicit update, ivdift/invtri,
 * cyclic conditions -- for all tracers of the shard; reads UVIC_F_SRC */
int uvic_gpu_transport(uvic_gpu *h);
/* replaces `call convct2 (t(1,1,1,1,taup1), joff, js, je, is, ie, kmt)`,
 * u09/mom/tracer.F:1198-1203 (source/mom/convect.F:99-311) */
int uvic_gpu_convect(uvic_gpu *h);
/* replaces `call tracer (joff, js, je, is, ie)`, source/mom/mom.F:389:
 * sources, transport, convection.  Knowing that convct2 follows, it sends T and S through the
 * transport first and replays their mixed segments inside the update of the other tracers; the
 * result equals uvic_gpu_transport followed by uvic_gpu_convect bit for bit (tests/test_gpu_fast.py) */
int uvic_gpu_tracer(uvic_gpu *h);
/* time-level rotation done by putmw/getvar through the ramdrive
 * (u09/mom/loadmw.F:528-588,717-744): taum1 <- tau, tau <- taup1 */
int uvic_gpu_rotate(uvic_gpu *h);
int uvic_gpu_sync(uvic_gpu *h);

/* asynchronous forms for a device-resident time loop (no host synchronisation;
 * pair with uvic_gpu_sync): one whole step; or, for tracer-index sharding,
 * the part before the exchange of t(tau+1) and the convection after it */
int uvic_gpu_step_async(uvic_gpu *h);
/* start computing the MOBI sources of the NEXT step from t(tau) on a side stream,
 * overlapped with this step's transport (the source terms of a leapfrog step depend only
 * on t(tau-1), which is this step's t(tau)); call before uvic_gpu_step_async of this step (preferred:
 * the chain is then queued ahead of the step's own side-stream work) or between it and
 * uvic_gpu_rotate, only when the next step is a leapfrog step with `c2dtts_next` */
int uvic_gpu_prefetch_sources(uvic_gpu *h, double c2dtts_next);
/* forward ("mixing") time step: t(tau-1) := t(tau) (u09/mom/loadmw.F:107-111, 569-584) by
 * aliasing the device buffers instead of copying; switch off again before the next step */
int uvic_gpu_set_mixing(uvic_gpu *h, int on);
/* the T,S-derived fields (isopyc products, folded coefficients) of the NEXT step on a second side stream; same calling
 * rule as uvic_gpu_prefetch_sources: before uvic_gpu_rotate of this step, next step leapfrog */
int uvic_gpu_prefetch_isopyc(uvic_gpu *h);
/* one call per step of a device-resident loop: forward-step aliasing (mixing), c2dtts, the whole step, and the
 * look-ahead chains of the next step (mobi_ahead with c2dtts_next, iso_ahead); then exchange halo rows if the
 * decomposition has neighbours, then uvic_gpu_rotate (which ends a forward step's aliasing).
 * iso_ahead: bit 0 = the T,S-derived fields of the next step (a leapfrog step) from this step's t(tau); bit 1 = those
 * of the step after next (a leapfrog step) from this step's t(tau+1), as soon as T and S of it are final -- single
 * rank only (a latitude slab receives the halo rows of t(tau+1) with the exchange that follows) */
int uvic_gpu_step_lookahead(uvic_gpu *h, double c2dtts, int mixing, int mobi_ahead, double c2dtts_next, int iso_ahead);
/* the same with relyr and co2ccn of the NEXT step named (the reference's clock advances every ocean step and tracer.F:311-338
 * takes the month of the dust field and the declination from it): the look-ahead MOBI chain computes with them; should the
 * next step then be given other values (uvic_gpu_set_mobi_step), it discards the chain's sources and computes its own */
int uvic_gpu_step_lookahead_at(uvic_gpu *h, double c2dtts, int mixing, int mobi_ahead, double c2dtts_next, double relyr_next,
                               double co2ccn_next, int iso_ahead);
int uvic_gpu_prefetch_sources_at(uvic_gpu *h, double c2dtts_next, double relyr_next, double co2ccn_next);
int uvic_gpu_step_pre_async(uvic_gpu *h);
int uvic_gpu_convect_async(uvic_gpu *h);

/* ---- the resident Fortran overlay (uvic2.9_amd/fortran/tracer_gpu.F, UVIC_RESIDENT=1) -----------------------------
 * 0: uvic_gpu_upload, _upload_rows and _set_mobi_step queue their copies on the main stream and return; the host
 * buffers (COMMON blocks, page-locked with uvic_gpu_pin_host) must stay unchanged until the next call that waits
 * (any download, uvic_gpu_overlay_step, uvic_gpu_sync).  Default 1. */
int uvic_gpu_set_host_sync(uvic_gpu *h, int on);
/* `set_sbc` (u09/mom/set_sbc.F:36-72, called at tracer.F:1273-1286) sums t(i,1,j,n,taup1) over the steps of an ocean
 * segment for the atmosphere.  With the state on the device the sums of the listed tracers (1-based) are kept there --
 * same additions, same order -- and fetched once per segment: count planes (imt, jmt), tracer order as listed */
int uvic_gpu_sbc_config(uvic_gpu *h, int count, const int32_t *tracers);
int uvic_gpu_sbc_transfer(uvic_gpu *h, double *host, int upload);
/* one resident step: forward-step aliasing, the step, its look-ahead chains (see uvic_gpu_step_lookahead_at), the
 * surface sums (sbc_zero: a segment's first step, set_sbc.F:40-48; sbc_accumulate), T,S of t(tau+1) to `ts_host`
 * (imt, km, jmt, 2; what clinic and the next loadmw read) and the rotation of the time levels.  Returns as soon as T,S
 * are on the host; pass B of the other tracers may still be running, the next call queues behind it. */
/* ---- time-step integrals of O_time_step_monitor on the device ----------------------------------------------
 * With tsiint = tsiper (the shipped run/control.in: 10 days each) tsiperts (source/common/switch.F:458-459) is true on
 * EVERY ocean step, and `tracer` then also forms tbar, travar, dtabs (diagt1, u09/mom/tracer.F:1516-1537, from t(tau),
 * t(tau-1) and t(tau+1) before convection) and the volume sum of delta 14C (:1329-1353), `clinic` ektot (u09/mom/
 * clinic.F:616-630).  uvic_gpu_set_tsi(h, 1, ic14, idic) names the step that follows such a step (until the next
 * uvic_gpu_rotate; ic14 = idic = 0: no delta-14C sum); uvic_gpu_tsi_read waits for the step and fills the arrays as
 * source/common/diag.h declares them, (0:km, nt, jmt) each, every sum added along i in the reference's order (bit-identical
 * with bit-identical tracers), and dc14bar (the rows' sums added in row order: equal to the reference's single running
 * sum to rounding).  uvic_gpu_tsi_ektot does the same for ektot (0:km, jmt) from UVIC_F_U1, UVIC_F_U2. */
int uvic_gpu_set_tsi(uvic_gpu *h, int on, int ic14, int idic);
int uvic_gpu_tsi_read(uvic_gpu *h, double *tbar, double *travar, double *dtabs, double *dc14bar);
int uvic_gpu_tsi_ektot(uvic_gpu *h, double rho0, double *ektot);
/* Time-average steps (O_time_averages: timavgperts, one year in ten with the shipped run/control.in).  What `tracer`
 * accumulates inside its own loops on such a step, with the options of run/mk.in: the convection diagnostics of convct2
 * (O_save_convection; source/mom/convect.F:183-191, 279-283, 295-301: totalk, vdepth, pe -- u09/mom/tracer.F:1211-1222 adds
 * them to ta_totalk, ta_vdepth, ta_pe) and the delta-14C field (O_carbon_14; tracer.F:1329-1340, :1355-1364 -> ta_dc14).
 * uvic_gpu_set_tavg names the coming step as one (grav: pconst; zt (km): coord.h; ic14, idic 1-based or 0); the step then
 * forms them on the device -- pe as the reference's one running sum over the column before and after convection, every
 * interior column -- and uvic_gpu_tavg_read fetches totalk, vdepth, pe (imt,jmt each) and dc14 (imt,km,jmt; null if not
 * wanted) after it.  Everything else a time-average step needs is outside `tracer` (diag -> avgvar reads t(tau), u(tau),
 * adv_vbt from the memory window: a caller that keeps them on the device brings them down on such steps). */
int uvic_gpu_set_tavg(uvic_gpu *h, int on, double grav, const double *zt, int ic14, int idic);
int uvic_gpu_tavg_read(uvic_gpu *h, double *totalk, double *vdepth, double *pe, double *dc14);

typedef struct uvic_overlay_step {
  double c2dtts, c2dtts_next, relyr_next, co2ccn_next;
  int32_t mixing, mobi_ahead, iso_ahead, sbc_zero, sbc_accumulate, pad;
} uvic_overlay_step;
int uvic_gpu_overlay_step(uvic_gpu *h, const uvic_overlay_step *s, double *ts_host);
/* The inputs `tracer` reads from the memory window on every step (mw.h: adv_vet, adv_vnt, adv_vbt; vmixc.h: diff_cbt;
 * csbc/mw.h: stf, btf) in one call, for a caller that keeps t on the device: replaces six uvic_gpu_upload_rows.  Host
 * arrays as the window holds them -- adv_vet (imt,km,jsmw:jmt), adv_vnt (imt,km,1:jmt), adv_vbt (imt,0:km,jsmw:jmt),
 * diff_cbt (imt,km,jsmw:jemw), stf and btf (imt,jmt,nt) -- page-locked (uvic_gpu_pin_host) and left alone until the
 * step's uvic_gpu_overlay_step returns.  The copies run beside the main stream into the device copy the previous step
 * does not read, what T and S need first; the step waits for each group where it reads it.  adv_vbt may be null: it is
 * then formed on the device from adv_vet and adv_vnt as source/mom/adv_vel.F:98-127 does (rigid lid, zero at the surface).
 * diff_cbt may be null (after uvic_gpu_set_vmix_params, with uvic_params.diff_cbt_has_k33 = 1): the step then forms it on
 * the device as updates/09/source/mom/vmixc.F:62-190 does -- tidal mixing above the bottom level from the stratification
 * of the step's isopyc fields and UVIC_F_EDR*, the previous value below, plus K33 -- bit-identically (the scheme's two
 * exponentials depend on the level pair only and are tabulated on the host with the C library's exp); a caller that
 * does so need not run the host's `isopyc` and `vmixc` on that step at all (uvic2.9_amd/fortran/mixing_gpu.F). */
int uvic_gpu_overlay_inputs(uvic_gpu *h, int jsmw, int jemw, const double *adv_vet, const double *adv_vnt, const double *adv_vbt,
                            const double *diff_cbt, const double *stf, const double *btf);
/* (adv_vet and adv_vnt may both be null after uvic_gpu_overlay_velocities, which has formed the step's velocities on the device.)
 *
 * For a caller that keeps u on the device, on the momentum stream beside the main stream; uvic_gpu_momentum_wait
 * returns when what they send to the host has arrived.  Host arrays page-locked and left alone until then.
 *
 * uvic_gpu_overlay_velocities -- the start of a leapfrog step, before `tracer`: what `loadmw` does to u with the memory
 *   window wide open (u09/mom/loadmw.F:86-99: rotation of the time levels, add_ext_mode from psi(,,1) on tau, and from
 *   psi(,,2) on tau-1 if `ext_taum1`), then `adv_vel` (source/mom/adv_vel.F:63-131) into the inputs of the tracer step.
 * uvic_gpu_overlay_momentum -- where the reference calls `clinic` (after `tracer`, source/mom/mom.F:389-395): the
 *   time-step monitor's kinetic energy of u(tau) (clinic.F:616-630; ektot (0:km, jmt) if ektot_host is given), `state`
 *   (loadmw.F:154) from T,S of UVIC_F_T_TAU (t_level 0) or UVIC_F_T_TAUM1 (-1: the tracer step has rotated the levels
 *   already) unless rho_host (imt,km,2:jmt) is given, `clinic` with sbc_flags/rts as uvic_gpu_clinic, zu to zu_host.
 *   `fresh`: u(tau), u(tau-1) were uploaded on the main stream just now.  u(tau+1) stays in UVIC_F_UP1/UP2. */
int uvic_gpu_overlay_velocities(uvic_gpu *h, int ext_taum1, const double *psi);
int uvic_gpu_overlay_momentum(uvic_gpu *h, int fresh, int t_level, int sbc_flags, double rts, double rho0, const double *smf,
                              const double *rho_host, double *zu_host, double *ektot_host);
/* latitude-slab decomposition (uvic_gpu_set_shard js..je): the two outermost owned rows of t(tau+1) of every tracer
 * go to the neighbour's halo after each step (reach of the FCT stencil, u09/mom/tracer_adv_flx.F:553-555).  The library
 * packs them into contiguous staging buffers and unpacks what was received, both on its main stream; the caller moves
 * the buffers between ranks (RCCL send/recv on that stream).  which: 0 send south, 1 send north, 2 receive south,
 * 3 receive north; each holds uvic_gpu_halo_elems doubles, laid out (nt, 2, imt*km). */
int64_t uvic_gpu_halo_elems(uvic_gpu *h);
void *uvic_gpu_halo_buffer(uvic_gpu *h, int which);
int uvic_gpu_halo_pack(uvic_gpu *h, int south, int north);
int uvic_gpu_halo_unpack(uvic_gpu *h, int south, int north);
/* The same two exchanges -- the halo rows of a latitude slab, the all-gather of t(:,:,:,slice,tau+1) of tracer shards
 * (the exchange where the reference's single process has none: tracer.F:902-1167 loops over n on one memory) -- as a
 * direct push between the ranks of one node: every rank owns a receive window and arrival counters in device memory,
 * its peers map them (hipIpcMemHandle_t, 64 bytes each) and write them with their own kernels; the data crosses each
 * xGMI link once and no collective library is on the path.
 *   uvic_gpu_push_setup     once per handle; mode 1: tracer shards (nt a multiple of world), 2: latitude slabs
 *   uvic_gpu_push_export    128 bytes (window handle, counter handle) for the caller to pass to the peers
 *   uvic_gpu_push_open      what rank `peer` exported (peer == own rank: no mapping, for a rank that neighbours itself)
 *   uvic_gpu_push_exchange  per step, queued on the main stream after the step: push, signal, wait (on the device,
 *                           at most "push_wait_ms" -- uvic_gpu_set_option, default 2000; a peer that never arrives
 *                           makes the next uvic_gpu_sync fail), take.  south/north: neighbour ranks of a slab or -1. */
int uvic_gpu_push_setup(uvic_gpu *h, int world, int rank, int mode);
int uvic_gpu_push_export(uvic_gpu *h, void *handles128);
int uvic_gpu_push_open(uvic_gpu *h, int peer, const void *handles128);
int uvic_gpu_push_exchange(uvic_gpu *h, int south, int north);

/* ---- producers of the step's shared inputs ------------------------------
 * replaces `call adv_vel (joff, js, je, is, ie)` (source/mom/mom.F:332; source/mom/adv_vel.F:63-131, the
 * T-cell part, rigid lid): UVIC_F_ADV_VET/VNT/VBT from UVIC_F_U1/U2 */
int uvic_gpu_adv_vel(uvic_gpu *h);
int uvic_gpu_adv_vel_async(uvic_gpu *h);   /* queued on the main stream, no wait */
int uvic_gpu_set_vmix_params(uvic_gpu *h, const uvic_vmix_params *p);
/* replaces the tracer part of `call vmixc (joff, js, je, is, ie)` (mom.F:347; u09/mom/vmixc.F:62-190 with
 * O_constvmix O_tidal_kv O_isopycmix): UVIC_F_DIFF_CBT = max(kappa_h, min(100, tidal + kappa_h)) above the
 * bottom level, the previous value elsewhere, plus K33.  Call after uvic_gpu_isopyc of the same step with
 * uvic_params.diff_cbt_has_k33 = 1 (isopyc then leaves diff_cbt to this call).  Bit-identical with the reference: exp
 * of the level-pair terms comes from a host-made table (built at the first call after uvic_gpu_set_vmix_params from
 * UVIC_F_ZW, which must have been uploaded). */
int uvic_gpu_vmixc(uvic_gpu *h);

/* ---- the reference's second, coarser boundary: the O_TMM column-batch source operator --------
 * With -DO_TMM u09/common/size.h:26-30 makes imt the batch size and jmt = 1, and `tracer` keeps only the MOBI source
 * loop and the 14C source (u09/mom/tracer.F:109-124, 214/304, 894/1288): the external Transport-Matrix-Method driver
 * fills t(1:imt,:,1,:,taum1), kmt(1:imt,1) and the forcing of the batch, calls `tracer (0, 1, 1, 1, imt)` and reads
 * src(imt,km,1,nsrc) from COMMON /mobicomm/.  These three calls are that operator: columns in, sources out, no
 * horizontal structure.  (The handle is an ordinary one whose grid holds the batch in one row; only the calls below
 * and uvic_gpu_destroy are meant for it.)
 *   kmt (ncols); forcing `f` with per-column arrays: tlat, dnswr, aice, hice, hsno (ncols), sg_bathy and fe_hydr
 *   (ncols,km), fe_atmdep (ncols,12) -- the reference's (imt,1,..) arrays as they lie in memory;
 *   `o` null = option set C (uvic_gpu_set_mobi), else as uvic_gpu_set_mobi_opt. */
int uvic_gpu_tmm_create(uvic_gpu **h, int ncols, int km, int nt, int nsrc, int ntnpzd, int device);
int uvic_gpu_tmm_set_mobi(uvic_gpu *h, const int32_t *kmt, const uvic_mobi_params *p, const uvic_mobi_options *o,
                          const uvic_mobi_forcing *f);
/* one call of the O_TMM `tracer`: t_taum1 (ncols,km,nt) in, src (ncols,km,nsrc) out; c2dtts, relyr, co2ccn as the
 * reference reads them from COMMON (scalar.h, tmngr.h, cembm.h); the four forcing fields (ncols) may be null = unchanged */
int uvic_gpu_tmm_sources(uvic_gpu *h, double c2dtts, double relyr, double co2ccn, const double *t_taum1, const double *dnswr,
                         const double *aice, const double *hice, const double *hsno, double *src);

/* ---- baroclinic momentum step -------------------------------------------
 * scalars of `clinic`: scalar.h c2dtuv, grav, rho0r, cdbot; vmixc.h kappa_m (O_constvmix: visc_cbu = kappa_m,
 * u09/mom/vmixc.F:85) */
typedef struct uvic_clinic_params {
  double c2dtuv, grav, rho0r, kappa_m, cdbot;
} uvic_clinic_params;
int uvic_gpu_set_clinic_params(uvic_gpu *h, const uvic_clinic_params *p);
/* replaces `call state (t(1,1,1,1,tau), t(1,1,1,2,tau), rho(1,1,jsmw), max(jsmw,js), je, istrt-1, iend+1)`
 * (u09/mom/loadmw.F:154; source/mom/state.F:1-41): UVIC_F_RHO on rows 2..jmt from T and S of UVIC_F_T_TAU */
int uvic_gpu_state(uvic_gpu *h);
/* replaces `call clinic (joff, js, je, is, ie)` (source/mom/mom.F:395; u09/mom/clinic.F:24-560 with fdifm.h,
 * options O_consthmix O_constvmix O_anisotropic_viscosity O_stream_function O_cyclic O_fourfil O_ice_evp) for one
 * memory window, rows 2..jmt-1: UVIC_F_UP1/UP2 (internal-mode velocities at tau+1, vertical mean removed, polar
 * filter applied when uvic_gpu_set_filter_u was called) and UVIC_F_ZU from UVIC_F_RHO, UVIC_F_U1/U2 (tau),
 * UVIC_F_UM1/UM2 (tau-1), UVIC_F_ADV_VET/VNT/VBT and UVIC_F_SMF.  What `clinic` reads from its neighbours in mom.F's
 * loop is evaluated inside: the U-cell advective velocities (source/mom/adv_vel.F:150-231) and the bottom drag
 * (u09/mom/setvbc.F:170-194).  sbc_flags: bit 0 = accumulate the ice/atmosphere surface velocities (isbcu, asbcu,
 * clinic.F:729-895; `eots`), bit 1 = osegs, bit 2 = osege; rts = 1/ntspos. */
int uvic_gpu_clinic(uvic_gpu *h, int sbc_flags, double rts);
/* the same two, queued on the handle's main stream without waiting (uvic_gpu_sync waits): for a caller that keeps
 * the momentum step device-resident between the tracer steps */
int uvic_gpu_state_async(uvic_gpu *h);
int uvic_gpu_clinic_async(uvic_gpu *h, int sbc_flags, double rts);
/* state + clinic of this time step on a stream of their own, beside the tracer step the caller queues next on the main
 * stream (`clinic` reads rho, u and the advective velocities, none of which the tracer step writes).  zu is copied to
 * zu_host (imt,jmt,2; null: not) and uvic_gpu_momentum_wait returns when it has arrived, so that the host's `tropic` can
 * run beside the tracer step.  Later calls that touch u, the advective velocities or these fields wait for it themselves. */
int uvic_gpu_momentum_async(uvic_gpu *h, int sbc_flags, double rts, double *zu_host);
int uvic_gpu_momentum_wait(uvic_gpu *h);
/* device-resident velocities: what u09/mom/loadmw.F does to u at the start of a time step with the memory window wide
 * open (:86-99): the time levels rotate (tau-1 <- tau <- tau+1, by pointer: UVIC_F_UM1/UM2, U1/U2, UP1/UP2) ... */
int uvic_gpu_rotate_u(uvic_gpu *h);
/* ... and `call add_ext_mode (joff, js, je, istrt, iend, 'tau')` (loadmw.F:590-714, O_stream_function) adds the external
 * mode from UVIC_F_PSI to the internal-mode velocities `clinic` left: level 0 = tau (from psi(,,1)), -1 = tau-1 (from
 * psi(,,2); the reference does that on the first time step only).  Rows 1..jmt-1, land masked, cyclic images. */
int uvic_gpu_add_ext_mode(uvic_gpu *h, int level);

/* polar filter of the velocities: replaces `call filuv (joff, js, je)` (clinic.F:500; source/common/filuv.F with
 * O_fourfil O_cyclic).  Strips come from UVIC_F_KMU by findex's rule, rows jfrst..jfu1 and jfu2..jmt-1, reference
 * row jfu0 (index.h; u09/common/setcom.F:80-86).  jfrst > jmt switches it off. */
int uvic_gpu_set_filter_u(uvic_gpu *h, double pi, int jfrst, int jfu0, int jfu1, int jfu2, int lsegf);

/* ---- polar Fourier filter of the tracers ---------------------------------
 * replaces `call filt (joff, js, je)` inside tracer (u09/mom/tracer.F:1245; source/common/filt.F,
 * filtr.F, findex.F with O_fourfil O_cyclic).  Called once after kmt and the grid metrics are uploaded:
 * the ocean strips of the rows jfrst..jft1 and jft2..jmt-1 (index.h, set in u09/common/setcom.F:75-85)
 * and their filter operators are built on the host and kept on the device; from then on every
 * uvic_gpu_tracer / uvic_gpu_convect filters t(tau+1) after convection.  jfrst > jmt switches it off.
 * `pi` is the reference's constant (scalar.h), `lsegf` the strip limit of index.h:34. */
int uvic_gpu_set_filter(uvic_gpu *h, double pi, int jfrst, int jft0, int jft1, int jft2, int lsegf);

/* run `nrep` x [isopyc, tracer] back to back and return the mean duration in
 * milliseconds of every kernel, measured with HIP events on the launch stream;
 * names[i] points to static strings.  Used by bench.py for the roofline. */
int uvic_gpu_profile(uvic_gpu *h, int nrep, int max_kernels, const char **names, double *mean_ms, int *nkernels);
/* the same measurement inside the caller's own time loop: between uvic_gpu_profile_live(h, 1) and
 * uvic_gpu_profile_read every kernel launch of the handle (main and side stream) is bracketed by
 * HIP events on its stream; _read synchronises, returns the mean per kernel and switches it off. */
int uvic_gpu_profile_live(uvic_gpu *h, int on);
int uvic_gpu_profile_read(uvic_gpu *h, int max_kernels, const char **names, double *mean_ms, int *nkernels);

#ifdef __cplusplus
}
#endif
#endif
