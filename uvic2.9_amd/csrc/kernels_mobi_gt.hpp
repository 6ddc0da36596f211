// kernels_mobi_gt.hpp -- MOBI option sets F (nt = 18) and the shipped run/mk.in set (nt = 37: O_mobi_caco3, O_mobi_silicon)
// in the three-pass, four-wave-team form of option set C (kernels_mobi.hpp):
//   mobig_pre_cell   (kernels_mobi_gen.hpp) one thread per cell: carbonate chemistry, light, temperature and oxygen functions,
//                    light-limited growth rates
//   mobigt_column    a team of four waves per 64 ocean columns walks the levels: the nbio Euler sub-steps of mobi_src
//                    (u09/mom/mobi.F:2148-3252) with one ROLE per wave, the raw tendencies and the hand-down of the exports
//   mobigt_post_cell one thread per cell: what mobi_driver does to a level after mobi_src (mobi.F:1033-1134, 1228-1266,
//                    the second and third loop :1302-1436 -- with prognostic CaCO3 both touch their own level only) and the
//                    caller's iron inputs and 14C (tracer.F:538-545, 853-867)
// Equations and their order are those of kernels_mobi_gen.hpp (`mobig_src`, `mobig_column`: one thread per column, the
// reference's own structure), which stays as the cross-check (uvic_gpu_set_option "mobi_team" 0): a column of the general
// kernel is one chain of km x nbio sub-steps of ~2200 fp64 instructions (1.03 ms per launch for the nt = 37 set); a team
// splits a sub-step four ways.  Roles (wave r evaluates, publishes in LDS, one workgroup barrier, the others fetch):
//   0 growth and nutrient limitation of phytoplankton, diatoms and diazotrophs; 15N assimilation
//   1 grazing, mortality, remineralisation, export, CaCO3 dissolution, opal; 15N recycling
//   2 iron speciation and scavenging
//   3 isotope ratios (15N, 13C) and the nitrate switch of diazotroph uptake
// then every wave advances the pools it owns (0 nutrients and producers, 1 zooplankton, detritus, iron, CaCO3, silicate,
// opal, 2 the 15N pools, 3 the 13C pools), refreshes their flags and publishes both (second barrier).  The exchange is
// single-buffered: a wave overwrites a rate only behind the second barrier of the sub-step, which every wave reaches
// after it has consumed the rates; likewise for the pools.
// Only the sets with prognostic CaCO3 come here (F and nt = 37): without it the calcite production of the whole column
// returns through a fixed profile (mobi.F:1373-1436), which option set C's own kernels handle.
#ifndef UVIC_KERNELS_MOBI_GT_HPP
#define UVIC_KERNELS_MOBI_GT_HPP

#include "kernels_mobi_gen.hpp"

namespace uvic {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(UV_NO_CONTRACT)
#pragma clang fp contract(fast)
#endif

// what the column pass hands to the cell pass, (imt,km,jmt) each, behind the MPG_* planes in the work buffer
enum { MAG_EXPO, MAG_EXPOP, MAG_RN15, MAG_RC13, MAG_CALPRO, MAG_NFIX, MAG_DISSL, MAG_EXPOC, MAG_EXPOOPL, MAG_COUNT };
static_assert(MPG_COUNT + MAG_COUNT <= UV_MOBI_WORK_PLANES, "work planes of mobi_store");
#define UV_MOBIGT_XN 51  /* rates exchanged per column and sub-step */
#define UV_MOBIGT_YN 65  /* pools, flags and P:N ratios exchanged per column and sub-step */
#define UV_MOBIGT_LDS_DOUBLES ((size_t)(UV_MOBIGT_XN + UV_MOBIGT_YN) * 64)

// owner of pool X_*: the wave that advances it and stores its tendency
static constexpr int MOBIGT_OWNER[X_COUNT] = {
    /* po4 phyt phyt_phos */ 0, 0, 0, /* zoop detr detr_phos */ 1, 1, 1, /* dic */ 0,
    /* dic13 phytc13 zoopc13 detrc13 doc13 diazc13 */ 3, 3, 3, 3, 3, 3, /* dop no3 don diaz */ 0, 0, 0, 0,
    /* din15 don15 phytn15 zoopn15 detrn15 diazn15 */ 2, 2, 2, 2, 2, 2, /* dfe detrfe */ 1, 1,
    /* caco3 diat sil opl */ 1, 0, 1, 1, /* diatn15 diatc13 caco3c13 */ 2, 3, 3};

/* mobi_src (mobi.F:1485-3313) for a team; bioin: the level's pools (unclamped), clamped on return; bioout: the new pools */
template <class Team, int N15, int C13, int CACO3, int SIL>
UVIC_DEV void mobigt_src(Team &T, mobi_params_cp P, mobi_options_cp O, const mobi_step &St, double capr, double (&bioin)[X_COUNT],
                         double avej, double avej_D, double avej_Diat, double bct, double impo, double impo_phos, double wwd, double nud,
                         double impocaco3, double wwc, double dissk1, double impoopl, double wwo, double opl_disk1, double nudop,
                         double nudon, double (&bioout)[X_COUNT], double bctz, double rn15impo, double rc13impo, double ac13b,
                         double rcaco3c13impo, double impofe, double o2flag, double aou_term, gsrc_out *out) {
#define BIN(x) bioin[x]
  double biopo4 = BIN(X_po4), biophyt = BIN(X_phyt), biophyt_phos = BIN(X_phyt_phos), biozoop = BIN(X_zoop);
  double biodetr = BIN(X_detr), biodetr_phos = BIN(X_detr_phos);
  double ptn_P = biophyt_phos / biophyt;
  double ptn_detr = biodetr_phos / biodetr;
  double biodic = BIN(X_dic), biodop = BIN(X_dop), biono3 = BIN(X_no3), biodon = BIN(X_don), biodiaz = BIN(X_diaz);
  double biodin15 = BIN(X_din15), biodon15 = BIN(X_don15), biophytn15 = BIN(X_phytn15), biozoopn15 = BIN(X_zoopn15);
  double biodetrn15 = BIN(X_detrn15), biodiazn15 = BIN(X_diazn15), biodiatn15 = BIN(X_diatn15);
  double biodic13 = BIN(X_dic13), biophytc13 = BIN(X_phytc13), biozoopc13 = BIN(X_zoopc13), biodetrc13 = BIN(X_detrc13);
  double biodoc13 = BIN(X_doc13), biodiazc13 = BIN(X_diazc13), biodiatc13 = BIN(X_diatc13), biocaco3c13 = BIN(X_caco3c13);
  double biocaco3 = BIN(X_caco3), biodiat = BIN(X_diat), biosil = BIN(X_sil), bioopl = BIN(X_opl);
  double biodfe = BIN(X_dfe), biodetrfe = BIN(X_detrfe);
  /* flags from the unclamped input, mobi.F:1814-1891; defaults 1.  (The flags of the 13C pools, of diatn15 and of
     caco3c13 are set and refreshed by the reference but enter no rate: not carried.) */
  double po4flag = g_flag01(biopo4 - UV_TRCMIN), phytflag = g_flag01(biophyt - UV_TRCMIN), zoopflag = g_flag01(biozoop - UV_TRCMIN);
  double detrflag = g_flag01(biodetr - UV_TRCMIN), phyt_phosflag = g_flag01(biophyt_phos - UV_TRCMIN);
  double detr_phosflag = g_flag01(biodetr_phos - UV_TRCMIN);
  const double sf_P_phosflag = g_flag01(ptn_P - P->gamma1 * P->redptn);
  const double sf_detr_phosflag = g_flag01(ptn_detr - P->gamma1 * P->redptn);
  double din15flag = 1., don15flag = 1., phytn15flag = 1., zoopn15flag = 1., detrn15flag = 1., diazn15flag = 1.;
  double dopflag = g_flag01(biodop - UV_TRCMIN), no3flag = g_flag01(biono3 - UV_TRCMIN), donflag = g_flag01(biodon - UV_TRCMIN);
  double diazflag = g_flag01(biodiaz - UV_TRCMIN);
  if (N15) {
    din15flag = g_flag01(biodin15 - UV_TRCMIN); don15flag = g_flag01(biodon15 - UV_TRCMIN); phytn15flag = g_flag01(biophytn15 - UV_TRCMIN);
    zoopn15flag = g_flag01(biozoopn15 - UV_TRCMIN); detrn15flag = g_flag01(biodetrn15 - UV_TRCMIN);
    diazn15flag = g_flag01(biodiazn15 - UV_TRCMIN);
  }
  double dfeflag = g_flag01(biodfe - UV_TRCMIN), detrfeflag = g_flag01(biodetrfe - UV_TRCMIN);
  double caco3flag = 1., diatflag = 1., silflag = 1., oplflag = 1.;
  if (CACO3) caco3flag = g_flag01(biocaco3 - UV_TRCMIN);
  if (SIL) { diatflag = g_flag01(biodiat - UV_TRCMIN); silflag = g_flag01(biosil - UV_TRCMIN); oplflag = g_flag01(bioopl - UV_TRCMIN); }
  /* clamp the caller's column and the working copies, mobi.F:1894-1960 */
  _Pragma("unroll") for (int x = 0; x < X_COUNT; ++x) bioin[x] = g_max(bioin[x], UV_TRCMIN);
  biopo4 = g_max(biopo4, UV_TRCMIN); biophyt = g_max(biophyt, UV_TRCMIN); biozoop = g_max(biozoop, UV_TRCMIN);
  biodetr = g_max(biodetr, UV_TRCMIN); biophyt_phos = g_max(biophyt_phos, UV_TRCMIN); biodetr_phos = g_max(biodetr_phos, UV_TRCMIN);
  biodic = g_max(biodic, UV_TRCMIN); biono3 = g_max(biono3, UV_TRCMIN); biodop = g_max(biodop, UV_TRCMIN);
  biodon = g_max(biodon, UV_TRCMIN); biodiaz = g_max(biodiaz, UV_TRCMIN);
  biodin15 = g_max(biodin15, UV_TRCMIN); biodon15 = g_max(biodon15, UV_TRCMIN); biophytn15 = g_max(biophytn15, UV_TRCMIN);
  biodiatn15 = g_max(biodiatn15, UV_TRCMIN); biozoopn15 = g_max(biozoopn15, UV_TRCMIN); biodetrn15 = g_max(biodetrn15, UV_TRCMIN);
  biodiazn15 = g_max(biodiazn15, UV_TRCMIN);
  biodic13 = g_max(biodic13, UV_TRCMIN); biophytc13 = g_max(biophytc13, UV_TRCMIN); biodiatc13 = g_max(biodiatc13, UV_TRCMIN);
  biocaco3c13 = g_max(biocaco3c13, UV_TRCMIN); biozoopc13 = g_max(biozoopc13, UV_TRCMIN); biodetrc13 = g_max(biodetrc13, UV_TRCMIN);
  biodoc13 = g_max(biodoc13, UV_TRCMIN); biodiazc13 = g_max(biodiazc13, UV_TRCMIN);
  biocaco3 = g_max(biocaco3, UV_TRCMIN); biodiat = g_max(biodiat, UV_TRCMIN); biosil = g_max(biosil, UV_TRCMIN);
  bioopl = g_max(bioopl, UV_TRCMIN); biodfe = g_max(biodfe, UV_TRCMIN); biodetrfe = g_max(biodetrfe, UV_TRCMIN);
  const double gmax = P->gbio * bctz;
  const double nupt = P->nupt0 * bct;
  const double nupt_D = P->nupt0_D * bct;
  const double nudt = O->nudt0 * bct;
  double nfixout = 0.0, expoout = 0.0, expo_phosout = 0.0, rn15expoout = 0.0, rc13expoout = 0.0, calproout = 0.0;
  double expofeout = 0.0, remifeout = 0.0, rcaco3c13expoout = 0.0, disslout = 0.0, expocaco3out = 0.0, expooplout = 0.0;
  const double dtbio = St.dtbio, redctn = P->redctn, redptn = P->redptn, gamma1 = P->gamma1, geZ = P->geZ;
  const double dfr = P->dfr, dfrt = P->dfrt, pfr = P->pfr, rnd = P->redntp / P->diazntp;
  const double nr_excr_P = 0.0, nr_excr_detr = 0.0;
  const double rn15hi = 2. * UV_RN15STD / (1 + UV_RN15STD), rn15lo = UV_RN15STD / (1 + UV_RN15STD) / 2.;
  const double rc13hi = 2. * UV_RC13STD / (1 + UV_RC13STD), rc13lo = 0.5 * UV_RC13STD / (1 + UV_RC13STD);
  const double diazptn = P->diazptn, rfeton = P->rfeton;

  for (int n = 1; n <= St.nbio; ++n) { /* mobi.F:2148-3252 */
#define ROLE(r) (Team::role == (r))
    double npp = 0., npp_D = 0., no3upt_D = 0., dopupt = 0., dopupt_D = 0., fcassim = 0., npp_Diat = 0., dopupt_Diat = 0.;
    double th_no3 = 0.;
    double graz = 0., graz_Z = 0., graz_Det = 0., graz_D = 0., morp = 0., morpt = 0., morz = 0., remi = 0., expo = 0.;
    double expo_phos = 0., recy_dop = 0., recy_don = 0., morp_D = 0., morpt_D = 0., fcrecy = 0.;
    double dissl = 0., expocaco3 = 0., graz_Diat = 0., morp_Diat = 0., morpt_Diat = 0., opldis = 0., expoopl = 0., sipr0 = 0.;
    double feorgads = 0., fecol = 0., expofe = 0., remife = 0.;
    double fcexcr = 0., rtphytn15 = 0., rtzoopn15 = 0., rtdetrn15 = 0., rtdiazn15 = 0., rtdiatn15 = 0., fcnpp = 0.;
    double rtphytc13 = 0., rtzoopc13 = 0., rtdetrc13 = 0., rtdoc13 = 0., rtdiazc13 = 0., rtdiatc13 = 0., rtcaco3c13 = 0., rtdic13 = 0.;
    if (ROLE(0)) {  // ---- growth and nutrient limitation (mobi.F:2150-2236), 15N assimilation (:2589-2600)
      double p1 = g_min(biophyt, P->pmax);
      double p2 = g_max(0.0, biophyt - P->pmax);
      const double k1n = div_safe(P->knmin * p1 + P->knmax * p2, p1 + p2);
      const double k1p_P = k1n * ptn_P;
      const double kfevar = div_safe(P->kfemin * p1 + P->kfemax * p2, p1 + p2);
      const double deffe = div_safe(biodfe, kfevar + biodfe);
      const double jmax = P->abio_P * bct * deffe;
      double k1n_Diat = 0., k1p_Diat = 0., jmax_Diat = 0.;
      if (SIL) {
        p1 = g_min(biodiat, O->pmax_Diat);
        p2 = g_max(0.0, biodiat - O->pmax_Diat);
        const double kfevar_Diat = div_safe(O->kfemin_Diat * p1 + O->kfemax_Diat * p2, p1 + p2);
        k1n_Diat = div_safe(O->knmin_Diat * p1 + O->knmax_Diat * p2, p1 + p2);
        k1p_Diat = k1n_Diat * redptn;
        const double deffe_Diat = div_safe(biodfe, kfevar_Diat + biodfe);
        jmax_Diat = O->abiodiat * bct * deffe_Diat;
      }
      const double deffe_D = div_safe(biodfe, P->kfe_D + biodfe);
      const double jmax_D = g_max(0., P->abio_P * (bct - P->dbct_D) * deffe_D) * P->jdiar;
      double limP_dop = div_safe(P->hdop * biodop, k1p_P + biodop);
      double limP_po4 = div_safe(biopo4, k1p_P + biopo4);
      const double dopupt_flag = g_flag01(limP_dop - limP_po4);
      const double limP = limP_dop * dopupt_flag + limP_po4 * (1. - dopupt_flag);
      double u_P = g_min(avej, jmax * limP);
      double u_Diat = 0., dopupt_Diat_flag = 0.;
      if (SIL) {
        const double k1si = 5.e-3;
        const double limSi = div_safe(biosil, k1si + biosil);
        limP_dop = div_safe(P->hdop * biodop, k1p_Diat + biodop);
        limP_po4 = div_safe(biopo4, k1p_Diat + biopo4);
        dopupt_Diat_flag = g_flag01(limP_dop - limP_po4);
        const double limP_Diat = limP_dop * dopupt_Diat_flag + limP_po4 * (1. - dopupt_Diat_flag);
        u_Diat = g_min(avej_Diat, jmax_Diat * limSi);
        u_Diat = g_min(u_Diat, jmax_Diat * limP_Diat);
      }
      u_P = g_min(u_P, div_safe(jmax * biono3, k1n + biono3));
      if (SIL) u_Diat = g_min(u_Diat, div_safe(jmax_Diat * biono3, k1n_Diat + biono3));
      const double u_D = g_min(avej_D, jmax_D * limP);
      const double dopupt_D_flag = dopupt_flag;
      npp = u_P * biophyt;
      npp_Diat = SIL ? u_Diat * biodiat : 0.;
      dopupt = npp * dopupt_flag; /* from the unflagged npp, mobi.F:2236 */
      dopupt_Diat = SIL ? npp_Diat * dopupt_Diat_flag : 0.;
      npp_D = g_max(0., u_D * biodiaz);
      no3upt_D = npp_D;   /* times th_no3 = 0.5 + 0.5*tanh(biono3 - 5), applied after the exchange (mobi.F:2233) */
      dopupt_D = npp_D * dopupt_D_flag;
      npp = npp * no3flag * (dopupt_flag * dopflag + (1. - dopupt_flag) * po4flag) * din15flag;
      if (SIL) npp_Diat = npp_Diat * no3flag * (dopupt_Diat_flag * dopflag + (1. - dopupt_Diat_flag) * po4flag) * din15flag;
      npp_D = npp_D * (dopupt_D_flag * dopflag + (1. - dopupt_D_flag) * po4flag) * din15flag;
      no3upt_D = no3upt_D * no3flag * din15flag;
      if (N15) {
        double uno3 = div_safe(npp * dtbio, biono3);
        uno3 = g_min(uno3, 0.999);
        uno3 = g_max(uno3, UV_TRCMIN);
        const double rno3 = g_clamp(biodin15 / (biono3 - biodin15), 2 * UV_RN15STD, UV_RN15STD / 2.);
        const double bassim = rayleigh(rno3, P->eps_assim, uno3);
        fcassim = div_safe(bassim, 1 + bassim);
      }
    }
    if (ROLE(1)) {  // ---- grazing, mortality, remineralisation, export (mobi.F:2223-2312), CaCO3, opal, 15N recycling
      double thetaZ = P->zprefP * biophyt + P->zprefDet * biodetr + P->zprefZ * biozoop + P->zprefDiaz * biodiaz + P->kzoo;
      if (SIL) thetaZ = thetaZ + O->zprefDiat * biodiat;
      const double ing_P = div_safe(P->zprefP, thetaZ), ing_Det = div_safe(P->zprefDet, thetaZ), ing_Z = div_safe(P->zprefZ, thetaZ);
      const double ing_D = div_safe(P->zprefDiaz, thetaZ);
      const double ing_Diat = SIL ? div_safe(O->zprefDiat, thetaZ) : 0.;
      const double g_D = gmax * ing_D * biodiaz;
      graz_D = g_D * biozoop;
      morpt_D = nupt_D * biodiaz;
      morp_D = P->nup_D * biodiaz * biodiaz;
      const double g_P = gmax * ing_P * biophyt;
      graz = g_P * biozoop;
      const double g_Z = gmax * ing_Z * biozoop;
      graz_Z = g_Z * biozoop;
      const double g_Det = gmax * ing_Det * biodetr;
      graz_Det = g_Det * biozoop;
      morp = P->nup * biophyt;
      morpt = nupt * biophyt;
      recy_don = nudon * bct * biodon;
      recy_dop = nudop * bct * biodop;
      morz = P->nuz * biozoop * biozoop;
      remi = nud * bct * biodetr;
      expo = wwd * biodetr;
      expo_phos = wwd * biodetr_phos;
      if (CACO3) {
        dissl = biocaco3 * dissk1;
        expocaco3 = wwc * biocaco3;
      }
      if (SIL) {
        const double g_Diat = gmax * ing_Diat * biodiat;
        graz_Diat = g_Diat * biozoop;
        morp_Diat = O->nu_diat * biodiat;
        morpt_Diat = nudt * biodiat;
        opldis = bioopl * opl_disk1;
        expoopl = wwo * bioopl;
      }
      /* negative prevention, mobi.F:2343-2445 */
      graz = graz * phytflag * phyt_phosflag * sf_P_phosflag * phytn15flag;
      graz_Z = graz_Z * zoopflag * zoopn15flag;
      graz_Det = graz_Det * detrflag * detr_phosflag * sf_detr_phosflag * detrn15flag;
      morp = morp * phytflag * phyt_phosflag * phytn15flag;
      morpt = morpt * phytflag * phyt_phosflag * phytn15flag;
      morz = morz * zoopflag * zoopn15flag;
      remi = remi * detrflag * detr_phosflag * detrn15flag;
      expo = expo * detrflag * detrn15flag;
      expo_phos = expo_phos * detr_phosflag;
      recy_dop = recy_dop * dopflag;
      graz_D = graz_D * diazflag * diazn15flag;
      morpt_D = morpt_D * diazflag * diazn15flag;
      morp_D = morp_D * diazflag * diazn15flag;
      recy_don = recy_don * donflag * don15flag;
      if (CACO3) {
        dissl = dissl * caco3flag;
        expocaco3 = expocaco3 * caco3flag;
      }
      if (SIL) {
        graz_Diat = graz_Diat * diatflag;
        morp_Diat = morp_Diat * diatflag;
        morpt_Diat = morpt_Diat * diatflag;
        /* opal production depends on iron (mobi.F:2683-2697): the tanh of the sub-step sits here, the lightest role */
        const double negcoeff = -0.46204044117647, VTP = 1.60266544117647, tanh_m = 6.9, tanh_b = -3.673092;
        sipr0 = (negcoeff * (2.0 * UV_HALF_TANH(tanh_m * biodfe * 1.e3 + tanh_b) - 1.0) + VTP);   /* tanh(y) = 2/(1 + exp(-2y)) - 1: one exp on the device */
        opldis = opldis * oplflag;
        expoopl = expoopl * oplflag;
      }
      if (N15) {
        double udon = div_safe(recy_don * dtbio, biodon);
        udon = g_min(udon, 0.999);
        udon = g_max(udon, UV_TRCMIN);
        const double rdon = g_clamp(biodon15 / (biodon - biodon15), 2 * UV_RN15STD, UV_RN15STD / 2.);
        const double brecy = rayleigh(rdon, P->eps_recy, udon);
        fcrecy = div_safe(brecy, 1 + brecy);
      }
    }
    if (ROLE(2)) {  // ---- iron speciation and scavenging, mobi.F:2313-2342
      remife = nud * bct * biodetrfe;
      const double ligand = UV_DIVC(g_max(aou_term + UV_DIVC(UV_POWP(biodon, 0.8), 4.8), 0.5), 1000.);
      const double fepa = (1.0 + P->kfeleq * (ligand - biodfe)) * o2flag;
      const double feprime = div_safe(-fepa + sqrt(fepa * fepa + 4.0 * P->kfeleq * biodfe), 2.0 * P->kfeleq) * o2flag;
      feorgads = (P->kfeorg * (UV_POWP((biodetr * detrflag) * P->mc * redctn, 0.58)) * feprime) * o2flag;
      fecol = P->kfecol * (feprime * feprime) * o2flag;
      expofe = wwd * biodetrfe;
      remife = remife * detrfeflag;
      feorgads = feorgads * dfeflag;
      expofe = expofe * detrfeflag;
      fecol = fecol * dfeflag;
    }
    if (ROLE(3)) {  // ---- isotope ratios, mobi.F:2601-2695; the nitrate switch
      th_no3 = UV_HALF_TANH(biono3 - 5.);
      if (N15) {
        const double rzoop = g_clamp(biozoopn15 / (biozoop - biozoopn15), 2. * UV_RN15STD, UV_RN15STD / 2.);
        const double bexcr = rzoop - UV_DIVC(P->eps_excr * rzoop, 1000.);
        fcexcr = div_safe(bexcr, 1 + bexcr);
        rtphytn15 = g_clamp(div_safe(biophytn15, biophyt), rn15hi, rn15lo);
        if (SIL) rtdiatn15 = g_clamp(div_safe(biodiatn15, biodiat), rn15hi, rn15lo);
        rtzoopn15 = g_clamp(div_safe(biozoopn15, biozoop), rn15hi, rn15lo);
        rtdetrn15 = g_clamp(div_safe(biodetrn15, biodetr), rn15hi, rn15lo);
        rtdiazn15 = g_clamp(div_safe(biodiazn15, biodiaz), rn15hi, rn15lo);
      }
      if (C13) {
        const double rdic13 = g_clamp(biodic13 / (biodic - biodic13), 2. * UV_RC13STD, 0.5 * UV_RC13STD);
        const double bc13npp = ac13b * rdic13;
        fcnpp = div_safe(bc13npp, 1 + bc13npp);
        rtdic13 = g_clamp(div_safe(biodic13, biodic), rc13hi, rc13lo);
        rtphytc13 = g_clamp(div_safe(biophytc13, biophyt * redctn), rc13hi, rc13lo);
        if (SIL) rtdiatc13 = g_clamp(div_safe(biodiatc13, biodiat * redctn), rc13hi, rc13lo);
        if (CACO3) rtcaco3c13 = g_clamp(div_safe(biocaco3c13, biocaco3), rc13hi, rc13lo);
        rtzoopc13 = g_clamp(div_safe(biozoopc13, biozoop * redctn), rc13hi, rc13lo);
        rtdetrc13 = g_clamp(div_safe(biodetrc13, biodetr * redctn), rc13hi, rc13lo);
        rtdoc13 = g_clamp(div_safe(biodoc13, biodon * redctn), rc13hi, rc13lo);
        rtdiazc13 = g_clamp(div_safe(biodiazc13, biodiaz * redctn), rc13hi, rc13lo);
      }
    }
    {  // publish own group, one barrier, fetch the other three (X(slot, variable, present))
      double *xb = T.xs + T.lane;
#define XA(X) X(0, npp, 1) X(1, npp_D, 1) X(2, no3upt_D, 1) X(3, dopupt, 1) X(4, dopupt_D, 1) X(5, fcassim, N15) \
  X(6, npp_Diat, SIL) X(7, dopupt_Diat, SIL)
#define XB(X) X(8, graz, 1) X(9, graz_Z, 1) X(10, graz_Det, 1) X(11, graz_D, 1) X(12, morp, 1) X(13, morpt, 1) X(14, morz, 1) \
  X(15, remi, 1) X(16, expo, 1) X(17, expo_phos, 1) X(18, recy_dop, 1) X(19, recy_don, 1) X(20, morp_D, 1) X(21, morpt_D, 1) \
  X(22, fcrecy, N15) X(23, dissl, CACO3) X(24, expocaco3, CACO3) X(25, graz_Diat, SIL) X(26, morp_Diat, SIL) \
  X(27, morpt_Diat, SIL) X(28, opldis, SIL) X(29, expoopl, SIL) X(30, sipr0, SIL)
#define XC(X) X(31, feorgads, 1) X(32, fecol, 1) X(33, expofe, 1) X(34, remife, 1)
#define XD(X) X(35, fcexcr, N15) X(36, rtphytn15, N15) X(37, rtzoopn15, N15) X(38, rtdetrn15, N15) X(39, rtdiazn15, N15) \
  X(40, rtdiatn15, N15 && SIL) X(41, fcnpp, C13) X(42, rtphytc13, C13) X(43, rtzoopc13, C13) X(44, rtdetrc13, C13) \
  X(45, rtdoc13, C13) X(46, rtdiazc13, C13) X(47, rtdiatc13, C13 && SIL) X(48, rtcaco3c13, C13 && CACO3) X(49, rtdic13, C13) \
  X(50, th_no3, 1)
#define XPUT(sl, v, on) if (on) xb[(size_t)(sl) * 64] = v;
#define XGET(sl, v, on) if (on) v = xb[(size_t)(sl) * 64];
      if (Team::role == 0) { XA(XPUT) } else if (Team::role == 1) { XB(XPUT) } else if (Team::role == 2) { XC(XPUT) } else { XD(XPUT) }
      T.sync();
      if (Team::role != 0) { XA(XGET) }
      if (Team::role != 1) { XB(XGET) }
      if (Team::role != 2) { XC(XGET) }
      if (Team::role != 3) { XD(XGET) }
#undef XA
#undef XB
#undef XC
#undef XD
#undef XPUT
#undef XGET
    }
#undef ROLE
    no3upt_D = th_no3 * no3upt_D;   /* mobi.F:2233; the flags (0 or 1) are already in, which leaves the product unchanged */
    /* zooplankton budget, mobi.F:2446-2575 */
    const double dig_P = gamma1 * graz, dig_Z = gamma1 * graz_Z, dig_Det = gamma1 * graz_Det;
    const double dig_Diat = gamma1 * graz_Diat;
    double dig = dig_Z + dig_P + dig_Det;
    if (SIL) dig = dig + dig_Diat;
    const double excr_P = gamma1 * (1 - geZ) * graz, excr_Z = gamma1 * (1 - geZ) * graz_Z;
    const double excr_Det = gamma1 * (1 - geZ) * graz_Det, excr_Diat = gamma1 * (1 - geZ) * graz_Diat;
    double excr = excr_Z + excr_P + excr_Det;
    if (SIL) excr = excr + excr_Diat;
    const double sf_P = (1. - gamma1) * graz, sf_Z = (1. - gamma1) * graz_Z, sf_Det = (1. - gamma1) * graz_Det;
    const double sf_Diat = (1. - gamma1) * graz_Diat;
    double sf = sf_P + sf_Z + sf_Det;
    if (SIL) sf = sf + sf_Diat;
    const double sf_P_phos = (graz * ptn_P - dig_P * redptn);
    const double sf_Det_phos = (graz_Det * ptn_detr - dig_Det * redptn);
    double sf_phos = sf_P_phos + sf_Z * redptn + sf_Det_phos;
    if (SIL) sf_phos = sf_phos + sf_Diat * redptn;
    const double dig_D = gamma1 * graz_D * rnd;
    dig = dig + dig_D;
    const double excr_D = gamma1 * (1 - geZ) * graz_D * rnd;
    excr = excr + excr_D;
    const double nr_excr_D = gamma1 * graz_D * (1 - rnd) + (1 - gamma1) * graz_D * (1 - rnd);
    const double sf_D = (1 - gamma1) * graz_D * rnd;
    sf = sf + sf_D;
    sf_phos = sf_phos + sf_D * redptn;
    double fcnfix = 0.;
    if (N15) {
      const double bnfix = UV_RN15STD - P->eps_nfix * UV_RN15STD / 1000.;
      fcnfix = bnfix / (1 + bnfix);
    }
    double calpro;
    if (CACO3) calpro = ((sf_Z + morz) * capr + (sf_P + morp) * capr) * redctn * 1.e3;
    else calpro = (morp + morz + (graz + graz_Z) * (1. - gamma1)) * capr * redctn * 1.e3;
    /* variable P:C of new production (Galbraith & Martiny 2015), mobi.F:2699-2702 */
    const double GM15ptc = 0.0060 + 0.0069 * biopo4;
    const double GM15ptn = GM15ptc * redctn * 1.e3;
    /* prognostic updates, mobi.F:2712-3085; every right-hand side uses the OLD state; each wave advances the pools it owns */
#define OWN(r) (Team::role == (r))
    if (OWN(0)) {
      double t_po4 = excr;
      if (SIL) t_po4 = excr + (1. - dfrt) * morpt_Diat - (npp_Diat - dopupt_Diat);
      const double n_po4 = biopo4 + dtbio * (dopupt * ptn_P - GM15ptn * npp + (1. - dfrt) * morpt * ptn_P +
                                             (1. - pfr) * remi * ptn_detr + diazptn * (morpt_D - (npp_D - dopupt_D)) +
                                             recy_dop + redptn * (t_po4));
      double n_dop;
      if (SIL)
        n_dop = biodop + dtbio * (dfr * morp * ptn_P + redptn * (dfr * morp_Diat + dfrt * morpt_Diat - dopupt_Diat) +
                                  dfrt * morpt * ptn_P + pfr * remi * ptn_detr - ptn_P * dopupt - diazptn * dopupt_D - recy_dop);
      else
        n_dop = biodop + dtbio * (dfr * morp * ptn_P + dfrt * morpt * ptn_P + pfr * remi * ptn_detr - ptn_P * dopupt -
                                  diazptn * dopupt_D - recy_dop);
      const double n_phyt = biophyt + dtbio * (npp - morp - graz - morpt);
      const double n_phyt_phos = biophyt_phos + dtbio * (npp * GM15ptn - morp * ptn_P - graz * ptn_P - morpt * ptn_P);
      double n_dic, n_no3, n_don;
      if (SIL) {
        n_dic = biodic + dtbio * redctn * (excr + (1. - pfr) * remi + (1. - dfrt) * morpt - npp + (1. - dfrt) * morpt_Diat -
                                           npp_Diat + morpt_D - npp_D + recy_don + nr_excr_D + nr_excr_P + nr_excr_detr +
                                           morp_D * (1. - rnd));
        n_no3 = biono3 + dtbio * (excr + (1. - pfr) * remi + (1. - dfrt) * morpt - npp + (1. - dfrt) * morpt_Diat - npp_Diat +
                                  morpt_D - no3upt_D + recy_don + nr_excr_D + nr_excr_P + nr_excr_detr + morp_D * (1. - rnd));
        n_don = biodon + dtbio * (dfr * morp + dfrt * morpt + pfr * remi - recy_don + dfr * morp_Diat + dfrt * morpt_Diat);
      } else {
        n_dic = biodic + dtbio * redctn * (excr + (1. - pfr) * remi + (1. - dfrt) * morpt - npp + morpt_D - npp_D + recy_don +
                                           nr_excr_D + nr_excr_P + nr_excr_detr + morp_D * (1. - rnd));
        n_no3 = biono3 + dtbio * (excr + (1. - pfr) * remi + (1. - dfrt) * morpt - npp + morpt_D - no3upt_D + recy_don +
                                  nr_excr_D + nr_excr_P + nr_excr_detr + morp_D * (1. - rnd));
        n_don = biodon + dtbio * (dfr * morp + dfrt * morpt + pfr * remi - recy_don);
      }
      const double n_diaz = biodiaz + dtbio * (npp_D - morp_D - morpt_D - graz_D);
      const double n_diat = biodiat + dtbio * (npp_Diat - morp_Diat - graz_Diat - morpt_Diat);
      biopo4 = n_po4; biodop = n_dop; biophyt = n_phyt; biophyt_phos = n_phyt_phos; biodic = n_dic; biono3 = n_no3;
      biodon = n_don; biodiaz = n_diaz;
      if (SIL) biodiat = n_diat;
    }
    if (OWN(1)) {
      const double n_zoop = biozoop + dtbio * (dig - morz - graz_Z - excr);
      double n_detr, n_detr_phos, n_dfe, n_detrfe;
      if (SIL) {
        n_detr = biodetr + dtbio * ((1. - dfr) * morp + sf + morz - remi - graz_Det - expo + impo + morp_D * rnd +
                                    (1. - dfr) * morp_Diat);
        n_detr_phos = biodetr_phos + dtbio * ((1. - dfr) * morp * ptn_P + sf_phos + morz * redptn - remi * ptn_detr -
                                              graz_Det * ptn_detr - expo_phos + impo_phos + morp_D * rnd * redptn +
                                              (1. - dfr) * morp_Diat * redptn);
        n_dfe = biodfe + dtbio * (rfeton * (excr + (1. - dfrt) * morpt - npp + morpt_D - npp_D + recy_don + nr_excr_D +
                                            nr_excr_P + nr_excr_detr + morp_D * (1. - rnd)) -
                                  feorgads + remife - fecol + rfeton * ((1. - dfrt) * morpt_Diat - npp_Diat));
        n_detrfe = biodetrfe + dtbio * (rfeton * (sf + (1. - dfr) * morp + morp_D * rnd + morz - graz_Det) + feorgads +
                                        P->iscr * fecol - remife - expofe + impofe + rfeton * (1. - dfr) * morp_Diat);
      } else {
        n_detr = biodetr + dtbio * ((1. - dfr) * morp + sf + morz - remi - graz_Det - expo + impo + morp_D * rnd);
        n_detr_phos = biodetr_phos + dtbio * ((1. - dfr) * morp * ptn_P + sf_phos + morz * redptn - remi * ptn_detr -
                                              graz_Det * ptn_detr - expo_phos + impo_phos + morp_D * rnd * redptn);
        n_dfe = biodfe + dtbio * (rfeton * (excr + (1. - dfrt) * morpt - npp + morpt_D - npp_D + recy_don + nr_excr_D +
                                            nr_excr_P + nr_excr_detr + morp_D * (1. - rnd)) -
                                  feorgads + remife - fecol);
        n_detrfe = biodetrfe + dtbio * (rfeton * (sf + (1. - dfr) * morp + morp_D * rnd + morz - graz_Det) + feorgads +
                                        P->iscr * fecol - remife - expofe + impofe);
      }
      /* opal production, mobi.F:2683-2697 (O_mobi_iron) */
      const double oplpro = SIL ? (morp_Diat + sf_Diat) * sipr0 * silflag * (1.e-3) : 0.;
      const double n_caco3 = biocaco3 + dtbio * (calpro - dissl - expocaco3 + impocaco3);
      const double n_sil = biosil + dtbio * (opldis - oplpro);
      const double n_opl = bioopl + dtbio * (oplpro - opldis - expoopl + impoopl);
      biozoop = n_zoop; biodetr = n_detr; biodetr_phos = n_detr_phos; biodfe = n_dfe; biodetrfe = n_detrfe;
      if (CACO3) biocaco3 = n_caco3;
      if (SIL) { biosil = n_sil; bioopl = n_opl; }
    }
    if (N15 && OWN(2)) {
      double n_din15, n_don15, n_zoopn15, n_detrn15, n_diatn15 = biodiatn15;
      if (SIL) {
        n_din15 = biodin15 + dtbio * (rtphytn15 * (1. - dfrt) * morpt + rtphytn15 * nr_excr_P +
                                      rtdiatn15 * (1. - dfrt) * morpt_Diat - fcassim * npp_Diat + fcexcr * excr +
                                      rtdiazn15 * morpt_D + rtdiazn15 * nr_excr_D + rtdiazn15 * morp_D * (1. - rnd) +
                                      rtdetrn15 * (1. - pfr) * remi + rtdetrn15 * nr_excr_detr + fcrecy * recy_don -
                                      fcassim * npp - fcassim * no3upt_D);
        n_don15 = biodon15 + dtbio * (dfr * rtphytn15 * morp + dfr * rtdiatn15 * morp_Diat + dfrt * rtdiatn15 * morpt_Diat +
                                      dfrt * rtphytn15 * morpt + rtdetrn15 * pfr * remi - fcrecy * recy_don);
        n_diatn15 = biodiatn15 + dtbio * (fcassim * npp_Diat - rtdiatn15 * morp_Diat - rtdiatn15 * graz_Diat -
                                          rtdiatn15 * morpt_Diat);
        n_zoopn15 = biozoopn15 + dtbio * (rtphytn15 * dig_P + rtdiatn15 * dig_Diat + rtzoopn15 * dig_Z + rtdetrn15 * dig_Det +
                                          rtdiazn15 * dig_D - rtzoopn15 * morz - rtzoopn15 * graz_Z - fcexcr * excr);
        n_detrn15 = biodetrn15 + dtbio * (rtphytn15 * (1. - dfr) * morp + rtdiatn15 * (1. - dfr) * morp_Diat +
                                          rtdiatn15 * sf_Diat + rtphytn15 * sf_P + rtzoopn15 * sf_Z + rtdetrn15 * sf_Det +
                                          rtdiazn15 * sf_D + rtzoopn15 * morz - rtdetrn15 * remi - rtdetrn15 * graz_Det -
                                          rtdetrn15 * expo + rn15impo * impo + rtdiazn15 * morp_D * rnd);
      } else {
        n_din15 = biodin15 + dtbio * (rtphytn15 * (1. - dfrt) * morpt + rtphytn15 * nr_excr_P + fcexcr * excr +
                                      rtdiazn15 * morpt_D + rtdiazn15 * nr_excr_D + rtdiazn15 * morp_D * (1. - rnd) +
                                      rtdetrn15 * (1. - pfr) * remi + rtdetrn15 * nr_excr_detr + fcrecy * recy_don -
                                      fcassim * npp - fcassim * no3upt_D);
        n_don15 = biodon15 + dtbio * (dfr * rtphytn15 * morp + dfrt * rtphytn15 * morpt + rtdetrn15 * pfr * remi -
                                      fcrecy * recy_don);
        n_zoopn15 = biozoopn15 + dtbio * (rtphytn15 * dig_P + rtzoopn15 * dig_Z + rtdetrn15 * dig_Det + rtdiazn15 * dig_D -
                                          rtzoopn15 * morz - rtzoopn15 * graz_Z - fcexcr * excr);
        n_detrn15 = biodetrn15 + dtbio * (rtphytn15 * (1. - dfr) * morp + rtphytn15 * sf_P + rtzoopn15 * sf_Z +
                                          rtdetrn15 * sf_Det + rtdiazn15 * sf_D + rtzoopn15 * morz - rtdetrn15 * remi -
                                          rtdetrn15 * graz_Det - rtdetrn15 * expo + rn15impo * impo +
                                          rtdiazn15 * morp_D * rnd);
      }
      const double n_phytn15 = biophytn15 + dtbio * (fcassim * npp - rtphytn15 * morp - rtphytn15 * graz - rtphytn15 * morpt);
      const double n_diazn15 = biodiazn15 + dtbio * (fcnfix * (npp_D - no3upt_D) + fcassim * no3upt_D - rtdiazn15 * morp_D -
                                                   rtdiazn15 * graz_D - rtdiazn15 * morpt_D);
      biodin15 = n_din15; biodon15 = n_don15; biophytn15 = n_phytn15; biodiatn15 = n_diatn15; biozoopn15 = n_zoopn15;
      biodetrn15 = n_detrn15; biodiazn15 = n_diazn15;
    }
    if (C13 && OWN(3)) {
      double n_dic13, n_doc13, n_zoopc13, n_detrc13, n_diatc13 = biodiatc13, n_caco3c13 = biocaco3c13;
      if (SIL) {
        n_dic13 = biodic13 + dtbio * redctn * (rtphytc13 * (1. - dfrt) * morpt + rtphytc13 * nr_excr_P + rtzoopc13 * excr +
                                               rtdiazc13 * morpt_D + rtdiazc13 * nr_excr_D + rtdiazc13 * morp_D * (1 - rnd) +
                                               rtdetrc13 * (1. - pfr) * remi + rtdetrc13 * nr_excr_detr +
                                               rtdiatc13 * (1. - dfrt) * morpt_Diat - fcnpp * npp_Diat + rtdoc13 * recy_don -
                                               fcnpp * npp - fcnpp * npp_D);
        n_doc13 = biodoc13 + dtbio * redctn * (dfr * rtphytc13 * morp + rtdiatc13 * (dfr * morp_Diat + dfrt * morpt_Diat) +
                                               rtphytc13 * dfrt * morpt + rtdetrc13 * pfr * remi - rtdoc13 * recy_don);
        n_zoopc13 = biozoopc13 + dtbio * redctn * (rtphytc13 * dig_P + rtdiatc13 * dig_Diat + rtzoopc13 * dig_Z +
                                                   rtdetrc13 * dig_Det + rtdiazc13 * dig_D - rtzoopc13 * morz -
                                                   rtzoopc13 * graz_Z - rtzoopc13 * excr);
        n_detrc13 = biodetrc13 + dtbio * redctn * (rtphytc13 * (1. - dfr) * morp + rtdiatc13 * (1. - dfr) * morp_Diat +
                                                   rtdiatc13 * sf_Diat + rtphytc13 * sf_P + rtzoopc13 * sf_Z +
                                                   rtdetrc13 * sf_Det + rtdiazc13 * sf_D + rtzoopc13 * morz -
                                                   rtdetrc13 * remi - rtdetrc13 * graz_Det - rtdetrc13 * expo + rc13impo +
                                                   rtdiazc13 * morp_D * rnd);
        n_diatc13 = biodiatc13 + dtbio * redctn * (fcnpp * npp_Diat - rtdiatc13 * (morp_Diat + graz_Diat + morpt_Diat));
      } else {
        n_dic13 = biodic13 + dtbio * redctn * (rtphytc13 * (1. - dfrt) * morpt + rtphytc13 * nr_excr_P + rtzoopc13 * excr +
                                               rtdiazc13 * morpt_D + rtdiazc13 * nr_excr_D + rtdiazc13 * morp_D * (1 - rnd) +
                                               rtdetrc13 * (1. - pfr) * remi + rtdetrc13 * nr_excr_detr + rtdoc13 * recy_don -
                                               fcnpp * npp - fcnpp * npp_D);
        n_doc13 = biodoc13 + dtbio * redctn * (dfr * rtphytc13 * morp + rtphytc13 * dfrt * morpt + rtdetrc13 * pfr * remi -
                                               rtdoc13 * recy_don);
        n_zoopc13 = biozoopc13 + dtbio * redctn * (rtphytc13 * dig_P + rtzoopc13 * dig_Z + rtdetrc13 * dig_Det +
                                                   rtdiazc13 * dig_D - rtzoopc13 * morz - rtzoopc13 * graz_Z -
                                                   rtzoopc13 * excr);
        n_detrc13 = biodetrc13 + dtbio * redctn * (rtphytc13 * (1. - dfr) * morp + rtphytc13 * sf_P + rtzoopc13 * sf_Z +
                                                   rtdetrc13 * sf_Det + rtdiazc13 * sf_D + rtzoopc13 * morz -
                                                   rtdetrc13 * remi - rtdetrc13 * graz_Det - rtdetrc13 * expo + rc13impo +
                                                   rtdiazc13 * morp_D * rnd);
      }
      const double n_phytc13 = biophytc13 + dtbio * redctn * (fcnpp * npp - rtphytc13 * morp - rtphytc13 * graz - rtphytc13 * morpt);
      const double n_diazc13 = biodiazc13 + dtbio * redctn * (fcnpp * npp_D - rtdiazc13 * (morp_D + graz_D + morpt_D));
      if (CACO3)
        n_caco3c13 = biocaco3c13 + dtbio * (rtdic13 * calpro - rtcaco3c13 * dissl - rtcaco3c13 * expocaco3 + rcaco3c13impo);
      biodic13 = n_dic13; biodoc13 = n_doc13; biophytc13 = n_phytc13; biozoopc13 = n_zoopc13; biodetrc13 = n_detrc13;
      biodiazc13 = n_diazc13; biocaco3c13 = n_caco3c13; biodiatc13 = n_diatc13;
    }
    /* accumulate, mobi.F:3088-3172 */
    expoout = expoout + expo;
    expo_phosout = expo_phosout + expo_phos;
    if (N15) rn15expoout = rn15expoout + rtdetrn15;
    if (C13) {
      rc13expoout = rc13expoout + rtdetrc13 * expo;
      if (CACO3) rcaco3c13expoout = rcaco3c13expoout + rtcaco3c13 * expocaco3;
    }
    calproout = calproout + calpro;
    if (CACO3) { disslout = disslout + dissl; expocaco3out = expocaco3out + expocaco3; }
    if (SIL) expooplout = expooplout + expoopl;
    nfixout = nfixout + npp_D - no3upt_D;
    expofeout = expofeout + expofe;
    remifeout = remifeout + remife;
    /* the P:N ratios are refreshed from the new pools; flags that are still set are refreshed, mobi.F:3175-3251 */
    if (OWN(0)) {
      ptn_P = biophyt_phos / biophyt;
      if (po4flag == 1) po4flag = g_flag01(biopo4 - UV_TRCMIN);
      if (phytflag == 1) phytflag = g_flag01(biophyt - UV_TRCMIN);
      if (phyt_phosflag == 1) phyt_phosflag = g_flag01(biophyt_phos - UV_TRCMIN);
      if (no3flag == 1) no3flag = g_flag01(biono3 - UV_TRCMIN);
      if (dopflag == 1) dopflag = g_flag01(biodop - UV_TRCMIN);
      if (donflag == 1) donflag = g_flag01(biodon - UV_TRCMIN);
      if (diazflag == 1) diazflag = g_flag01(biodiaz - UV_TRCMIN);
      if (CACO3 && SIL && diatflag == 1) diatflag = g_flag01(biodiat - UV_TRCMIN);   /* (nested under O_mobi_caco3, mobi.F:3212-3222) */
    }
    if (OWN(1)) {
      ptn_detr = biodetr_phos / biodetr;
      if (zoopflag == 1) zoopflag = g_flag01(biozoop - UV_TRCMIN);
      if (detrflag == 1) detrflag = g_flag01(biodetr - UV_TRCMIN);
      if (detr_phosflag == 1) detr_phosflag = g_flag01(biodetr_phos - UV_TRCMIN);
      if (dfeflag == 1) dfeflag = g_flag01(biodfe - UV_TRCMIN);
      if (detrfeflag == 1) detrfeflag = g_flag01(biodetrfe - UV_TRCMIN);
      if (CACO3) {
        if (caco3flag == 1) caco3flag = g_flag01(biocaco3 - UV_TRCMIN);
        if (SIL) {
          if (silflag == 1) silflag = g_flag01(biosil - UV_TRCMIN);
          if (oplflag == 1) oplflag = g_flag01(bioopl - UV_TRCMIN);
        }
      }
    }
    if (N15 && OWN(2)) {
      if (din15flag == 1) din15flag = g_flag01(biodin15 - UV_TRCMIN);
      if (don15flag == 1) don15flag = g_flag01(biodon15 - UV_TRCMIN);
      if (phytn15flag == 1) phytn15flag = g_flag01(biophytn15 - UV_TRCMIN);
      if (zoopn15flag == 1) zoopn15flag = g_flag01(biozoopn15 - UV_TRCMIN);
      if (detrn15flag == 1) detrn15flag = g_flag01(biodetrn15 - UV_TRCMIN);
      if (diazn15flag == 1) diazn15flag = g_flag01(biodiazn15 - UV_TRCMIN);
    }
    {  // second exchange: new pools, their flags, the P:N ratios
      double *yb = T.xs + (size_t)UV_MOBIGT_XN * 64 + T.lane;
#define YA(X) X(0, biopo4, 1) X(1, biodop, 1) X(2, biophyt, 1) X(3, biophyt_phos, 1) X(4, biodic, 1) X(5, biono3, 1) X(6, biodon, 1) \
  X(7, biodiaz, 1) X(8, ptn_P, 1) X(9, po4flag, 1) X(10, phytflag, 1) X(11, phyt_phosflag, 1) X(12, no3flag, 1) X(13, dopflag, 1) \
  X(14, donflag, 1) X(15, diazflag, 1) X(16, biodiat, SIL) X(17, diatflag, SIL)
#define YB(X) X(18, biozoop, 1) X(19, biodetr, 1) X(20, biodetr_phos, 1) X(21, biodfe, 1) X(22, biodetrfe, 1) X(23, ptn_detr, 1) \
  X(24, zoopflag, 1) X(25, detrflag, 1) X(26, detr_phosflag, 1) X(27, dfeflag, 1) X(28, detrfeflag, 1) X(29, biocaco3, CACO3) \
  X(30, caco3flag, CACO3) X(31, biosil, SIL) X(32, bioopl, SIL) X(33, silflag, SIL) X(34, oplflag, SIL)
#define YC(X) X(35, biodin15, N15) X(36, biodon15, N15) X(37, biophytn15, N15) X(38, biozoopn15, N15) X(39, biodetrn15, N15) \
  X(40, biodiazn15, N15) X(41, din15flag, N15) X(42, don15flag, N15) X(43, phytn15flag, N15) X(44, zoopn15flag, N15) \
  X(45, detrn15flag, N15) X(46, diazn15flag, N15) X(47, biodiatn15, N15 && SIL)
#define YD(X) X(48, biodic13, C13) X(49, biodoc13, C13) X(50, biophytc13, C13) X(51, biozoopc13, C13) X(52, biodetrc13, C13) \
  X(53, biodiazc13, C13) X(54, biodiatc13, C13 && SIL) X(55, biocaco3c13, C13 && CACO3)
#define YPUT(sl, v, on) if (on) yb[(size_t)(sl) * 64] = v;
#define YGET(sl, v, on) if (on) v = yb[(size_t)(sl) * 64];
      if (Team::role == 0) { YA(YPUT) } else if (Team::role == 1) { YB(YPUT) } else if (Team::role == 2) { YC(YPUT) } else { YD(YPUT) }
      T.sync();
      if (Team::role != 0) { YA(YGET) }
      if (Team::role != 1) { YB(YGET) }
      if (Team::role != 2) { YC(YGET) }
      if (Team::role != 3) { YD(YGET) }   // (what a role never reads is dropped by the compiler: roles are compile-time)
#undef YA
#undef YB
#undef YC
#undef YD
#undef YPUT
#undef YGET
    }
#undef OWN
  }
  /* the new pools; the caller forms the tendency of those it owns (mobi.F:3255-3313) */
  _Pragma("unroll") for (int x = 0; x < X_COUNT; ++x) bioout[x] = 0.0;
#define BOUT(x, v) bioout[x] = (v)
  BOUT(X_po4, biopo4); BOUT(X_phyt, biophyt); BOUT(X_phyt_phos, biophyt_phos); BOUT(X_zoop, biozoop);
  BOUT(X_detr, biodetr); BOUT(X_detr_phos, biodetr_phos); BOUT(X_dic, biodic); BOUT(X_dop, biodop);
  BOUT(X_no3, biono3); BOUT(X_don, biodon); BOUT(X_diaz, biodiaz); BOUT(X_din15, biodin15);
  BOUT(X_don15, biodon15); BOUT(X_phytn15, biophytn15); BOUT(X_zoopn15, biozoopn15); BOUT(X_detrn15, biodetrn15);
  BOUT(X_diazn15, biodiazn15); BOUT(X_diatn15, biodiatn15); BOUT(X_caco3, biocaco3); BOUT(X_diat, biodiat);
  BOUT(X_sil, biosil); BOUT(X_opl, bioopl); BOUT(X_dfe, biodfe); BOUT(X_detrfe, biodetrfe); BOUT(X_dic13, biodic13);
  BOUT(X_phytc13, biophytc13); BOUT(X_zoopc13, biozoopc13); BOUT(X_detrc13, biodetrc13); BOUT(X_doc13, biodoc13);
  BOUT(X_diazc13, biodiazc13); BOUT(X_diatc13, biodiatc13); BOUT(X_caco3c13, biocaco3c13);
  out->expo = expoout; out->expo_phos = expo_phosout; out->calpro = calproout; out->nfix = nfixout;
  out->rn15expo = rn15expoout; out->rc13expo = rc13expoout; out->expofe = expofeout; out->remife = remifeout;
  out->expocaco3 = expocaco3out; out->dissl = disslout; out->rcaco3c13expo = rcaco3c13expoout; out->expoopl = expooplout;
#undef BIN
#undef BOUT
}

// ---------------------------------------------------------------------------
// the vertical sequence of mobi_driver for the 64 columns of a team: per level the sub-steps, the raw tendencies into the
// source slots (each wave those of the pools it owns), what the cell pass needs, and the hand-down of the exports
// (mobi.F:1124-1134, 1268-1287).  `live`: the lane holds a column; `kmax`: the deepest column of the team.
// ---------------------------------------------------------------------------
template <class Team, int N15, int C13, int CACO3, int SIL>
UVIC_DEV void mobigt_column(Team &T, const uvic_ctx &c, const mobi_dev &M, int i, int j, bool live, int kmax) {
  UV_DIMS(c);
  mobi_params_cp P = UV_CONST_AS(M.P);
  mobi_options_cp O = UV_CONST_OPT(M.O);
  const mobi_step &St = M.S;
  const size_t ij = X2(i, j), NS = (size_t)imt * jmt;
  double *src = const_cast<double *>(c.src);
  const int kmx = live ? c.kmt[ij] : 0;
  const int *I = O->im, *S = O->is;
  double *aux = M.pre + (size_t)MPG_COUNT * N3;
  double sink = 0.0;
#define TNR(k, x) c.t_taum1[X3(i, k, j) + (size_t)(P->tracer_of_mobi[I[x] - 1] - 1) * N3]
#define PREG(q) M.pre[(size_t)(q) * N3 + X3(i, k, j)]
#define AUXG(q) aux[(size_t)(q) * N3 + X3(i, k, j)]
#define MINE(r) (Team::role == (r))
  double expo = 0.0, impo, expo_phos = 0.0, impo_phos;
  double rn15impo = 0.0, rn15expo = 0.0, rc13impo = 0.0, rc13expo = 0.0, expofe = 0.0, impofe;
  double rcaco3c13impo = 0.0, rcaco3c13expo = 0.0, impocaco3 = 0.0, expocaco3 = 0.0, dissk1 = 0.0;
  double expoopl = 0.0, impoopl = 0.0, opl_disk1 = 0.0;
  double capr = P->capr;
  double snpzd[X_COUNT], bioin[X_COUNT];
  for (int k = 1; k <= kmax; ++k) {
    const bool store = live && k <= kmx;
#define OUT(ptr) (*(store ? (ptr) : &sink))
    if (N15) rn15impo = rn15expo;
    double ac13b = 0.0;
    {
      const double Omega_c = PREG(MPG_OMEGAC);
      if (C13) {
        ac13b = PREG(MPG_AC13B);
        rc13impo = rc13expo * P->dztr[k - 1];
        if (CACO3) rcaco3c13impo = rcaco3c13expo * P->dztr[k - 1];
      }
      if (CACO3) {
        dissk1 = O->dissk0 * g_max(0., (1. - Omega_c));
        capr = O->caprmax * g_max(0., (Omega_c - 1.) / (O->kcapr + Omega_c - 1.));
      }
      if (SIL) opl_disk1 = O->opl_disk0;
    }
    if (CACO3) impocaco3 = expocaco3 * P->dztr[k - 1];
    impo = expo * P->dztr[k - 1];
    impo_phos = expo_phos * P->dztr[k - 1];
    impofe = expofe * P->dztr[k - 1];
    if (SIL) impoopl = expoopl * P->dztr[k - 1];
    _Pragma("unroll") for (int x = 0; x < X_COUNT; ++x) bioin[x] = I[x] > 0 ? TNR(k, x) : 0.0;
    gsrc_out so;
    mobigt_src<Team, N15, C13, CACO3, SIL>(T, P, O, St, capr, bioin, PREG(MPG_AVEJ), PREG(MPG_AVEJD), PREG(MPG_AVEJDIAT), PREG(MPG_BCT), impo,
                                           impo_phos, P->wd[k - 1], PREG(MPG_NUD), impocaco3, O->wc[k - 1], dissk1, impoopl, O->wo[k - 1],
                                           opl_disk1, P->nudop0, P->nudon0, snpzd, PREG(MPG_BCTZ), rn15impo, rc13impo, ac13b, rcaco3c13impo,
                                           impofe, PREG(MPG_O2F), PREG(MPG_AOUT), &so);
    expo = so.expo; expo_phos = so.expo_phos; expofe = so.expofe;
    if (N15) rn15expo = so.rn15expo;
    if (C13) rc13expo = so.rc13expo;
    if (C13 && CACO3) rcaco3c13expo = so.rcaco3c13expo;
    if (CACO3) expocaco3 = so.expocaco3;
    if (SIL) expoopl = so.expoopl;
    // tendency = (new pool - clamped input) / twodt, mobi.F:3255-3313, 922-947: each wave those of the pools it owns,
    // straight into their source slots (mobi.F:1149-1205)
    _Pragma("unroll") for (int x = 0; x < X_COUNT; ++x)
      if (MINE(MOBIGT_OWNER[x]) && I[x] > 0 && S[x] > 0)
        OUT(src + X3(i, k, j) + (size_t)(S[x] - 1) * N3) = (snpzd[x] - bioin[x]) * St.rdtts;
    expofe = expofe * St.rnbio;
    if (CACO3) expocaco3 = expocaco3 * St.rnbio;
    if (SIL) expoopl = expoopl * St.rnbio;
    expo = expo * St.rnbio;
    expo_phos = expo_phos * St.rnbio;
    if (N15) rn15expo = rn15expo * St.rnbio;
    if (C13) {
      rc13expo = rc13expo * St.rnbio;
      if (CACO3) rcaco3c13expo = rcaco3c13expo * St.rnbio;
    }
    if (MINE(1)) {
      OUT(&AUXG(MAG_EXPO)) = expo; OUT(&AUXG(MAG_EXPOP)) = expo_phos; OUT(&AUXG(MAG_CALPRO)) = so.calpro * St.rnbio;
      OUT(&AUXG(MAG_NFIX)) = so.nfix;
      if (CACO3) { OUT(&AUXG(MAG_DISSL)) = so.dissl * St.rnbio; OUT(&AUXG(MAG_EXPOC)) = expocaco3; }
      if (SIL) OUT(&AUXG(MAG_EXPOOPL)) = expoopl;
    }
    if (N15 && MINE(2)) OUT(&AUXG(MAG_RN15)) = rn15expo;
    if (C13 && MINE(3)) OUT(&AUXG(MAG_RC13)) = rc13expo;
    // bottom remineralisation takes its share (mobi.F:1124-1134); the rest is the import of the next level, mobi.F:1268-1287
    const double sgb = M.sg_bathy[ij + NS * (k - 1)];
    const double dztk = P->dzt[k - 1];
    if (C13) rc13expo = rc13expo - sgb * rc13expo;
    expo = expo - sgb * expo;
    expo_phos = expo_phos - sgb * expo_phos;
    expo = expo * dztk;
    expo_phos = expo_phos * dztk;
    if (C13) {
      rc13expo = rc13expo * dztk;
      if (CACO3) rcaco3c13expo = rcaco3c13expo * dztk;
    }
    expofe = expofe * dztk;
    if (CACO3) expocaco3 = expocaco3 * dztk;
    if (SIL) expoopl = expoopl * dztk;
#undef OUT
  }
#undef MINE
#undef AUXG
#undef PREG
#undef TNR
}

// ---------------------------------------------------------------------------
// the part of mobi_driver after mobi_src that stays inside the cell, for the sets with prognostic CaCO3 (the equations of
// mobig_column, kernels_mobi_gen.hpp, in its order); land and the levels below the sea floor get zero sources
// ---------------------------------------------------------------------------
template <int N15, int C13, int CACO3, int SIL>
UVIC_DEV void mobigt_post_cell(const uvic_ctx &c, const mobi_dev &M, int i, int k, int j) {
  static_assert(CACO3 == 1, "without prognostic CaCO3 the calcite production of the column returns through a profile: option set C's kernels");
  UV_DIMS(c);
  mobi_params_cp P = UV_CONST_AS(M.P);
  mobi_options_cp O = UV_CONST_OPT(M.O);
  const mobi_step &St = M.S;
  const size_t ij = X2(i, j), NS = (size_t)imt * jmt;
  double *src = const_cast<double *>(c.src);
  const int kmx = c.kmt[ij];
  const int *I = O->im, *S = O->is;
  const double *aux = M.pre + (size_t)MPG_COUNT * N3;
#define SRC(s) src[X3(i, k, j) + (size_t)((s)-1) * N3]
#define SX(x) SRC(S[x])
#define TIN(n) c.t_taum1[X3(i, k, j) + (size_t)((n)-1) * N3]
#define TN(x) g_max(c.t_taum1[X3(i, k, j) + (size_t)(P->tracer_of_mobi[I[x] - 1] - 1) * N3], UV_TRCMIN)
#define AUXG(q) aux[(size_t)(q) * N3 + X3(i, k, j)]
  if (k > kmx) {
    for (int s = 1; s <= P->nsrc; ++s) SRC(s) = 0.0;
    return;
  }
  const double twodt = c.c2dtts, redctn = P->redctn;
  const double o2_in = TIN(P->io2) * 1000., dic_in = TIN(P->idic);
  const double sgb = M.sg_bathy[ij + NS * (k - 1)], dztk = P->dzt[k - 1];
  const double expo = AUXG(MAG_EXPO), expo_phos = AUXG(MAG_EXPOP);
  const double rn15expo = N15 ? AUXG(MAG_RN15) : 0., rc13expo = C13 ? AUXG(MAG_RC13) : 0.;
  const double rcalpro_k = AUXG(MAG_CALPRO), nfix_k = AUXG(MAG_NFIX), rdissl_k = AUXG(MAG_DISSL), rexpocaco3_k = AUXG(MAG_EXPOC);
  const double bio_no3 = TN(X_no3), bio_din15 = N15 ? TN(X_din15) : 0.;
  /* benthic denitrification, mobi.F:1033-1085 */
  const double no3flag = g_flag01(bio_no3 - UV_TRCMIN);
  const double din15flag = N15 ? g_flag01(bio_din15 - UV_TRCMIN) : 1.;
  const double lno3 = 0.5 * tanh(bio_no3 * 10 - 5.0);
  double sg_bdeni = (0.06 + 0.19 * UV_POWP(0.99, g_max(o2_in, UV_TRCMIN) - g_max(bio_no3, UV_TRCMIN))) *
                    g_max(expo * sgb, UV_TRCMIN) * redctn * 1.e3;
  sg_bdeni = g_min(sg_bdeni, sgb * expo);
  sg_bdeni = g_max(sg_bdeni, 0.);
  sg_bdeni = sg_bdeni * (0.5 + lno3) * no3flag * din15flag;
  double sn_no3 = SX(X_no3) + sgb * expo - sg_bdeni;
  double sn_din15 = 0.;
  const double r15min = UV_TRCMIN * UV_RN15STD / (1 + UV_RN15STD);
  double rno3 = 0.;
  if (N15) {
    rno3 = g_max(bio_din15, r15min) / g_max(bio_no3 - bio_din15, r15min);
    rno3 = g_min(rno3, 2. * UV_RN15STD);
    rno3 = g_max(rno3, UV_RN15STD / 2.);
    const double eps_bdeni = P->eps_bdeni0 * exp(-2.5e-6 * (P->zt[k - 1]));
    const double bbdeni = rno3 - eps_bdeni * rno3 / 1000.;
    sn_din15 = SX(X_din15) + rn15expo * sgb * expo - bbdeni / (1 + bbdeni) * sg_bdeni;
  }
  /* sedimentary iron release, mobi.F:1086-1123 */
  const double coxdepth = g_min(g_max(P->zt[k - 1], 50000.), 150000.);
  const double oblinc = -1.26e-6 * coxdepth + 0.203;
  const double obexpc = -6.e-7 * coxdepth + 1.14;
  const double nburial = (oblinc * UV_POWP(expo * sgb * dztk / 100 * 86400. * 365. * redctn * 1000., obexpc)) /
                         (86400. * 365. * dztk / 100 * redctn * 1000.);
  const double coxsed = expo * sgb - nburial;
  const double fesedmax = 85.;
  const double fesed = fesedmax * tanh(coxsed * redctn * 1000 * dztk / 100 * 86400. / o2_in) / (dztk / 100 * 86400 * 1000);
  double fe = SX(X_dfe) + fesed;
  /* bottom remineralisation, mobi.F:1124-1134 */
  SX(X_po4) = SX(X_po4) + sgb * expo_phos;
  const double sn_dic = SX(X_dic) + sgb * expo * redctn;
  double s_dic13 = C13 ? SX(X_dic13) + rc13expo * sgb * redctn : 0.;
  /* DIC / alkalinity / 13C bookkeeping, mobi.F:1228-1266 */
  const double dic_sms = sn_dic;
  double s_dic = sn_dic;
  double rtdic13_k = 0., rtcaco3c13_k = 0.;
  if (C13) {
    const double r13min = UV_TRCMIN * UV_RC13STD / (1 + UV_RC13STD);
    double r = g_max(TN(X_dic13), r13min) / g_max(dic_in, UV_TRCMIN);
    r = g_min(r, 2. * UV_RC13STD / (1 + UV_RC13STD));
    r = g_max(r, 0.5 * UV_RC13STD / (1 + UV_RC13STD));
    rtdic13_k = r;
    double rc = g_max(TN(X_caco3c13), r13min) / g_max(TN(X_caco3), UV_TRCMIN);
    rc = g_min(rc, 2. * UV_RC13STD / (1 + UV_RC13STD));
    rc = g_max(rc, 0.5 * UV_RC13STD / (1 + UV_RC13STD));
    rtcaco3c13_k = rc;
  }
  double a = -sn_dic * P->redntc * 1.e-3;
  /* oxygen, water-column denitrification, 15N: the reference's second loop (mobi.F:1302-1365) */
  const double fo2 = tanh(0.22 * g_max(o2_in, 0.));
  const double so2 = dic_sms * P->redotc + nfix_k * St.rnbio * 1.25e-3;
  const double lno3b = 0.5 * tanh(bio_no3 - 2.5);
  double wcdeni = 800. * no3flag * so2 * (1.0 - fo2) * (0.5 + lno3b);
  if (N15) wcdeni = wcdeni * din15flag;
  wcdeni = g_max(wcdeni, 0.);
  SX(X_no3) = sn_no3 - wcdeni;
  if (N15) {
    double uno3 = wcdeni * twodt / bio_no3;
    uno3 = g_min(uno3, 0.999);
    uno3 = g_max(uno3, UV_TRCMIN);
    const double bwcdeni = rayleigh(rno3, P->eps_wcdeni, uno3);
    SX(X_din15) = sn_din15 - (bwcdeni / (1 + bwcdeni)) * wcdeni;
  }
  a = a + wcdeni * 1.e-3;
  a = a + sg_bdeni * 1.e-3;
  a = a - nfix_k * St.rnbio * 1.e-3;
  SRC(O->is_o2) = -so2 * fo2;
  /* prognostic CaCO3: dissolution and production in the level, what reaches the sea floor dissolves there (mobi.F:1373-1436) */
  const double rexp = (k == kmx) ? rexpocaco3_k : 0.0;
  if (k < kmx) {
    s_dic = s_dic + rdissl_k * 1.e-3 - rcalpro_k * 1.e-3;
    if (C13) s_dic13 = s_dic13 + rdissl_k * 1.e-3 * rtcaco3c13_k - rcalpro_k * 1.e-3 * rtdic13_k;
    a = a + 2. * rdissl_k * 1.e-3 - 2. * rcalpro_k * 1.e-3;
  } else {
    s_dic = s_dic + rdissl_k * 1.e-3 - rcalpro_k * 1.e-3 + rexp * 1.e-3;
    if (C13) s_dic13 = s_dic13 + rdissl_k * 1.e-3 * rtcaco3c13_k - rcalpro_k * 1.e-3 * rtdic13_k + rexp * 1.e-3 * rtcaco3c13_k;
    a = a + 2. * rdissl_k * 1.e-3 - 2. * rcalpro_k * 1.e-3 + 2. * rexp * 1.e-3;
  }
  SX(X_dic) = s_dic;
  if (C13) SX(X_dic13) = s_dic13;
  SRC(O->is_alk) = a;
  if (SIL && k == kmx) SX(X_sil) = SX(X_sil) + AUXG(MAG_EXPOOPL);
  /* iron inputs, tracer.F:538-545 */
  if (k == 1) fe = fe + M.fe_atmdep[ij + NS * (St.month - 1)] * 1000 / (P->dzt[0] / 100.);
  fe = fe + M.fe_hydr[ij + NS * (k - 1)];
  SX(X_dfe) = fe;
  /* carbon-14, tracer.F:853-867 */
  if (O->is_c14 > 0) SRC(O->is_c14) = s_dic * UV_RC14STD - 3.836e-12 * TIN(P->ic14);
#undef SRC
#undef SX
#undef TIN
#undef TN
#undef AUXG
}

#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang fp contract(off)
#endif
}  // namespace uvic
#endif
