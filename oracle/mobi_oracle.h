/* oracle/mobi_oracle.h -- TEST INFRASTRUCTURE ONLY (see uvic_oracle.h). */
#ifndef MOBI_ORACLE_H
#define MOBI_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MOBI_MAXT 40
#define ORC_MOBI_MAXK 64

/* 1-based positions; 0 = not present */
typedef struct orc_mobi_index {
  int po4, phyt, phyt_phos, zoop, detr, detr_phos, dic, dic13, phytc13, zoopc13, detrc13, doc13, diazc13;
  int dop, no3, don, diaz, din15, don15, phytn15, zoopn15, detrn15, diazn15, dfe, detrfe, alk, o2, c14;
} orc_mobi_index;

/* parameters of /npzd_r/ after mobi_init (u09/mom/mobi.h), per-step scalars of
 * tracer.F:340-343 and the vertical grid */
typedef struct orc_mobi {
  int km, ntnpzd, nsrc, nbio;
  double dtbio, rdtts, rnbio;
  orc_mobi_index im;       /* position of each tracer in tnpzd(:, m)   (imobi*) */
  orc_mobi_index is;       /* source slot of each tracer                (is*)    */
  int tracer_of_mobi[ORC_MOBI_MAXT]; /* prognostic tracer index of mobi tracer m */
  int itemp, isalt, idic, ialk, io2, ic14;
  double kw, kc, ki, tap, abio_P, bbio, cbio, nup, nup_D, nupt0, nupt0_D, gamma1, gbio, nuz, nud0, nudon0, nudop0;
  double redptn, redctn, redntp, redotc, redntc, diazntp, diazptn, kzoo, geZ;
  double zprefP, zprefDet, zprefZ, zprefDiaz;
  double kfe_D, kfemin, kfemax, knmin, knmax, pmax, thetamaxlo, thetamaxhi, alphamin, alphamax;
  double kfeleq, kfeorg, kfecol, mc, rfeton, iscr, jdiar, dbct_D, hdop, dfr, dfrt, pfr;
  double eps_assim, eps_recy, eps_excr, eps_nfix, eps_wcdeni, eps_bdeni0, capr;
  double wd[ORC_MOBI_MAXK], ztt[ORC_MOBI_MAXK], rcak[ORC_MOBI_MAXK], rcab[ORC_MOBI_MAXK];
  double zt[ORC_MOBI_MAXK], dzt[ORC_MOBI_MAXK], dztr[ORC_MOBI_MAXK];
} orc_mobi;

typedef struct orc_mobi_forcing {
  double pi, radian, relyr, co2ccn;
  const double *tlat, *dnswr, *aice, *hice, *hsno;  /* (imt,jmt) */
  const double *sg_bathy;                           /* (imt,jmt,km) */
  const double *fe_atmdep;                          /* (imt,jmt,12) */
  const double *fe_hydr;                            /* (imt,jmt,km) */
} orc_mobi_forcing;

void orc_co2calc_SWS(double t, double s, double dic_in, double ta_in, double co2_in, double atmpres, double depth,
                     double *ph, double *co2star, double *dco2star, double *pCO2, double *dpco2, double *CO3,
                     double *Omega_c, double *Omega_a);
void orc_mobi_driver(const orc_mobi *P, int kmx, double twodt, double rctheta, double dayfrac, double swr, double *tnpzd,
                     const double *t_in, const double *o2_in, const double *aou_in, const double *s_in, const double *dic_in,
                     const double *alk_in, double co2_in, const double *sgb_in, double *src);
void orc_mobi_sources(const orc_mobi *P, const orc_mobi_forcing *F, int imt, int jmt, const int *kmt, const double *t_taum1,
                      int nt, double c2dtts, double *src);
#ifdef __cplusplus
}
#endif
#endif
