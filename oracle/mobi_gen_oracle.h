/* oracle/mobi_gen_oracle.h -- TEST INFRASTRUCTURE ONLY (see uvic_oracle.h). */
#ifndef MOBI_GEN_ORACLE_H
#define MOBI_GEN_ORACLE_H
#include "mobi_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif

/* the MOBI column tracers of every option set; im[]/is[] hold the 1-based position in tnpzd / the source slot, 0 = absent */
enum {
  X_po4, X_phyt, X_phyt_phos, X_zoop, X_detr, X_detr_phos, X_dic, X_dic13, X_phytc13, X_zoopc13, X_detrc13, X_doc13, X_diazc13,
  X_dop, X_no3, X_don, X_diaz, X_din15, X_don15, X_phytn15, X_zoopn15, X_detrn15, X_diazn15, X_dfe, X_detrfe,
  X_caco3, X_diat, X_sil, X_opl, X_diatn15, X_diatc13, X_caco3c13, X_COUNT
};

typedef struct orc_mobig {
  int km, ntnpzd, nsrc, nbio;
  int opt_n15, opt_c13, opt_caco3, opt_silicon;
  double dtbio, rdtts, rnbio;
  int im[X_COUNT], is[X_COUNT];
  int tracer_of_mobi[ORC_MOBI_MAXT];
  int itemp, isalt, idic, ialk, io2, ic14, is_alk, is_o2, is_c14, pad_;
  double kw, kc, ki, tap, abio_P, bbio, cbio, nup, nup_D, nupt0, nupt0_D, gamma1, gbio, nuz, nud0, nudon0, nudop0;
  double redptn, redctn, redntp, redotc, redntc, diazntp, diazptn, kzoo, geZ;
  double zprefP, zprefDet, zprefZ, zprefDiaz;
  double kfe_D, kfemin, kfemax, knmin, knmax, pmax, thetamaxlo, thetamaxhi, alphamin, alphamax;
  double kfeleq, kfeorg, kfecol, mc, rfeton, iscr, jdiar, dbct_D, hdop, dfr, dfrt, pfr;
  double eps_assim, eps_recy, eps_excr, eps_nfix, eps_wcdeni, eps_bdeni0, capr;
  /* O_mobi_caco3 */
  double kc_c, dissk0, caprmax, kcapr;
  /* O_mobi_silicon */
  double abiodiat, kfemin_Diat, kfemax_Diat, knmin_Diat, knmax_Diat, pmax_Diat, zprefDiat, nu_diat, nudt0, opl_disk0;
  double wd[ORC_MOBI_MAXK], ztt[ORC_MOBI_MAXK], rcak[ORC_MOBI_MAXK], rcab[ORC_MOBI_MAXK], wc[ORC_MOBI_MAXK], wo[ORC_MOBI_MAXK];
  double zt[ORC_MOBI_MAXK], dzt[ORC_MOBI_MAXK], dztr[ORC_MOBI_MAXK];
} orc_mobig;

int orc_mobig_sizeof(void);
void orc_mobig_driver(const orc_mobig *P, int kmx, double twodt, double rctheta, double dayfrac, double swr, double *tnpzd,
                      const double *t_in, const double *o2_in, const double *aou_in, const double *s_in, const double *dic_in,
                      const double *alk_in, double co2_in, const double *sgb_in, double *src);
void orc_mobig_sources(const orc_mobig *P, const orc_mobi_forcing *F, int imt, int jmt, const int *kmt, const double *t_taum1,
                       int nt, double c2dtts, double *src);
#ifdef __cplusplus
}
#endif
#endif
