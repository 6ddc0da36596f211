// MOBI with the cpp options of u09/mom/mobi.F as RUN-TIME flags, for the option sets other than C:
//   F (nt=18):  no isotopes, prognostic CaCO3 (O_mobi_caco3)
//   the shipped run/mk.in set (nt=37): C + O_mobi_caco3 + O_mobi_silicon (diatoms, silicate, opal)
// Always on, as in every set the reference can build: O_mobi_o2, O_mobi_iron, O_carbon, O_mobi_alk, O_mobi_nitrogen.
// One thread per ocean column walks mobi_driver's three loops as the reference does (mobi.F:519-1482, mobi_src
// :1485-3313, the caller's part tracer.F:355-545, 853-867), expression by expression in the reference's order, no
// reassociation: the results differ from the CPU through the device's exp/log/tanh, fused mul+add pairs and powers taken
// as exp(y log x) and quotients whose denominator cannot vanish without range handling (UV_POWP, div_safe, as in the
// kernels of set C; the quotients the reference lets run to +-inf and clamps keep `/`) -- 3e-12 of the oracle at most, tested at 1e-11.  Option set C keeps its own
// three-pass, four-wave-team kernels (kernels_mobi.hpp); this form is the general one, not the fast one: the flags are
// wave-uniform branches, the column state lives in local memory.
// Set E (no O_mobi_alk) is not served: the reference itself reads t(i,:,j,ialk,taum1) with ialk = 0 there (tracer.F:491).
#ifndef UVIC_KERNELS_MOBI_GEN_HPP
#define UVIC_KERNELS_MOBI_GEN_HPP

#include "kernels_mobi.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) uvic_mobi_options *mobi_options_cp;
#define UV_CONST_OPT(p) ((mobi_options_cp)(p))
#else
typedef const uvic_mobi_options *mobi_options_cp;
#define UV_CONST_OPT(p) (p)
#endif

namespace uvic {

// tolerance-tested like the kernels of set C (1e-11 on the sources): mul+add pairs may fuse on the device
#if defined(__HIP_DEVICE_COMPILE__) && !defined(UV_NO_CONTRACT)
#pragma clang fp contract(fast)
#endif

// positions in uvic_mobi_options::im / ::is (include/uvic_gpu.h)
enum {
  X_po4, X_phyt, X_phyt_phos, X_zoop, X_detr, X_detr_phos, X_dic, X_dic13, X_phytc13, X_zoopc13, X_detrc13, X_doc13, X_diazc13,
  X_dop, X_no3, X_don, X_diaz, X_din15, X_don15, X_phytn15, X_zoopn15, X_detrn15, X_diazn15, X_dfe, X_detrfe,
  X_caco3, X_diat, X_sil, X_opl, X_diatn15, X_diatc13, X_caco3c13, X_COUNT
};
static_assert(X_COUNT == UVIC_MOBI_NX, "pool table of include/uvic_gpu.h");

UVIC_DEV double g_max(double a, double b) { return a > b ? a : b; }
UVIC_DEV double g_min(double a, double b) { return a < b ? a : b; }
UVIC_DEV double g_sq(double x) { return x * x; }
UVIC_DEV double g_flag01(double x) { return 0.5 + copysign(0.5, x); }
UVIC_DEV double g_clamp(double r, double hi, double lo) {
  r = g_min(r, hi);
  r = g_max(r, lo);
  return r;
}
/* r + eps*(1-u)/u*log(1-u)*r/1000 (mobi.F:2589-2600) */
UVIC_DEV double g_rayleigh(double r, double eps, double u) { return r + eps * (1 - u) / u * log(1 - u) * r / 1000.; }

typedef struct {
  double expo, expo_phos, calpro, nfix, rn15expo, rc13expo, expofe, remife;
  double expocaco3, dissl, rcaco3c13expo, expoopl;
} gsrc_out;

/* mobi_src, mobi.F:1485-3313 */
// TN15 .. TSIL: 1/0 = the option fixed at compile time (the sets of SURVEY.md §2c are instantiated so: dead branches and
// their state vanish), -1 = read from the flags at run time (any other combination)
template <int TN15, int TC13, int TCACO3, int TSIL>
UVIC_DEV void mobig_src(mobi_params_cp P, mobi_options_cp O, const mobi_step &St, double capr, double (&bioin)[X_COUNT], double avej, double avej_D, double avej_Diat, double bct, double impo,
                         double impo_phos, double wwd, double nud, double impocaco3, double wwc, double dissk1,
                         double impoopl, double wwo, double opl_disk1, double nudop, double nudon, double (&bioout)[X_COUNT], double bctz,
                         double rn15impo, double rc13impo, double ac13b, double rcaco3c13impo, double impofe, double o2,
                         double aou, gsrc_out *out) {
  const int *I = O->im;
  const int N15 = TN15 < 0 ? O->n15 : TN15, C13 = TC13 < 0 ? O->c13 : TC13, CACO3 = TCACO3 < 0 ? O->caco3 : TCACO3;
  const int SIL = TSIL < 0 ? O->silicon : TSIL;
/* the pools travel indexed by their identity (X_*), not by their position in tnpzd: every index is a compile-time
   constant and the vectors live in registers; a pool the set does not have arrives as 0 */
#define BIN(x) bioin[x]
#define BGET(x) bioin[x]
  double biopo4 = BIN(X_po4), biophyt = BIN(X_phyt), biophyt_phos = BIN(X_phyt_phos), biozoop = BIN(X_zoop);
  double biodetr = BIN(X_detr), biodetr_phos = BIN(X_detr_phos);
  double ptn_P = biophyt_phos / biophyt;
  double ptn_detr = biodetr_phos / biodetr;
  double biodic = BIN(X_dic), biodop = BIN(X_dop), biono3 = BIN(X_no3), biodon = BIN(X_don), biodiaz = BIN(X_diaz);
  double biodin15 = BGET(X_din15), biodon15 = BGET(X_don15), biophytn15 = BGET(X_phytn15), biozoopn15 = BGET(X_zoopn15);
  double biodetrn15 = BGET(X_detrn15), biodiazn15 = BGET(X_diazn15), biodiatn15 = BGET(X_diatn15);
  double biodic13 = BGET(X_dic13), biophytc13 = BGET(X_phytc13), biozoopc13 = BGET(X_zoopc13), biodetrc13 = BGET(X_detrc13);
  double biodoc13 = BGET(X_doc13), biodiazc13 = BGET(X_diazc13), biodiatc13 = BGET(X_diatc13), biocaco3c13 = BGET(X_caco3c13);
  double biocaco3 = BGET(X_caco3), biodiat = BGET(X_diat), biosil = BGET(X_sil), bioopl = BGET(X_opl);
  double biodfe = BIN(X_dfe), biodetrfe = BIN(X_detrfe);
  /* flags from the unclamped input, mobi.F:1814-1891; defaults 1 */
  double po4flag = g_flag01(biopo4 - UV_TRCMIN), phytflag = g_flag01(biophyt - UV_TRCMIN), zoopflag = g_flag01(biozoop - UV_TRCMIN);
  double detrflag = g_flag01(biodetr - UV_TRCMIN), phyt_phosflag = g_flag01(biophyt_phos - UV_TRCMIN);
  double detr_phosflag = g_flag01(biodetr_phos - UV_TRCMIN);
  const double sf_P_phosflag = g_flag01(ptn_P - P->gamma1 * P->redptn);
  const double sf_detr_phosflag = g_flag01(ptn_detr - P->gamma1 * P->redptn);
  double din15flag = 1., don15flag = 1., phytn15flag = 1., zoopn15flag = 1., detrn15flag = 1., diazn15flag = 1.;
  double dic13flag = 1., doc13flag = 1., phytc13flag = 1., zoopc13flag = 1., detrc13flag = 1., diazc13flag = 1.;
  double diatn15flag = 1., diatc13flag = 1., caco3c13flag = 1.;
  double dopflag = g_flag01(biodop - UV_TRCMIN), no3flag = g_flag01(biono3 - UV_TRCMIN), donflag = g_flag01(biodon - UV_TRCMIN);
  double diazflag = g_flag01(biodiaz - UV_TRCMIN);
  if (N15) {
    din15flag = g_flag01(biodin15 - UV_TRCMIN); don15flag = g_flag01(biodon15 - UV_TRCMIN); phytn15flag = g_flag01(biophytn15 - UV_TRCMIN);
    if (SIL) diatn15flag = g_flag01(biodiatn15 - UV_TRCMIN);
    zoopn15flag = g_flag01(biozoopn15 - UV_TRCMIN); detrn15flag = g_flag01(biodetrn15 - UV_TRCMIN);
    diazn15flag = g_flag01(biodiazn15 - UV_TRCMIN);
  }
  if (C13) {
    dic13flag = g_flag01(biodic13 - UV_TRCMIN); phytc13flag = g_flag01(biophytc13 - UV_TRCMIN);
    if (SIL) diatc13flag = g_flag01(biodiatc13 - UV_TRCMIN);
    if (CACO3) caco3c13flag = g_flag01(biocaco3c13 - UV_TRCMIN);
    zoopc13flag = g_flag01(biozoopc13 - UV_TRCMIN); detrc13flag = g_flag01(biodetrc13 - UV_TRCMIN);
    doc13flag = g_flag01(biodoc13 - UV_TRCMIN); diazc13flag = g_flag01(biodiazc13 - UV_TRCMIN);
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
  /* the light-limited growth rates avej, avej_D, avej_Diat (mobi.F:1961-2061) depend on the level's inputs only: mobig_pre_cell */
  double p1, p2, kfevar, deffe, deffe_D, jmax, jmax_D, kfevar_Diat = 0., deffe_Diat = 0., jmax_Diat = 0.;
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
    p1 = g_min(biophyt, P->pmax);
    p2 = g_max(0.0, biophyt - P->pmax);
    const double k1n = div_safe(P->knmin * p1 + P->knmax * p2, p1 + p2);
    const double k1p_P = k1n * ptn_P;
    kfevar = div_safe(P->kfemin * p1 + P->kfemax * p2, p1 + p2);
    deffe = div_safe(biodfe, kfevar + biodfe);
    jmax = P->abio_P * bct * deffe;
    double k1n_Diat = 0., k1p_Diat = 0.;
    if (SIL) {
      p1 = g_min(biodiat, O->pmax_Diat);
      p2 = g_max(0.0, biodiat - O->pmax_Diat);
      kfevar_Diat = div_safe(O->kfemin_Diat * p1 + O->kfemax_Diat * p2, p1 + p2);
      k1n_Diat = div_safe(O->knmin_Diat * p1 + O->knmax_Diat * p2, p1 + p2);
      k1p_Diat = k1n_Diat * redptn;
      deffe_Diat = div_safe(biodfe, kfevar_Diat + biodfe);
      jmax_Diat = O->abiodiat * bct * deffe_Diat;
    }
    deffe_D = div_safe(biodfe, P->kfe_D + biodfe);
    jmax_D = g_max(0., P->abio_P * (bct - P->dbct_D) * deffe_D) * P->jdiar;
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
    double thetaZ = P->zprefP * biophyt + P->zprefDet * biodetr + P->zprefZ * biozoop + P->zprefDiaz * biodiaz + P->kzoo;
    if (SIL) thetaZ = thetaZ + O->zprefDiat * biodiat;
    const double ing_P = div_safe(P->zprefP, thetaZ), ing_Det = div_safe(P->zprefDet, thetaZ), ing_Z = div_safe(P->zprefZ, thetaZ);
    const double ing_D = div_safe(P->zprefDiaz, thetaZ);
    const double ing_Diat = SIL ? div_safe(O->zprefDiat, thetaZ) : 0.;
    double npp = u_P * biophyt;
    double npp_Diat = SIL ? u_Diat * biodiat : 0.;
    const double dopupt = npp * dopupt_flag; /* from the unflagged npp, mobi.F:2236 */
    const double dopupt_Diat = SIL ? npp_Diat * dopupt_Diat_flag : 0.;
    double npp_D = g_max(0., u_D * biodiaz);
    const double g_D = gmax * ing_D * biodiaz;
    double graz_D = g_D * biozoop;
    double morpt_D = nupt_D * biodiaz;
    double morp_D = P->nup_D * biodiaz * biodiaz;
    double no3upt_D = UV_HALF_TANH(biono3 - 5.) * npp_D;
    const double dopupt_D = npp_D * dopupt_D_flag;
    const double g_P = gmax * ing_P * biophyt;
    double graz = g_P * biozoop;
    const double g_Z = gmax * ing_Z * biozoop;
    double graz_Z = g_Z * biozoop;
    const double g_Det = gmax * ing_Det * biodetr;
    double graz_Det = g_Det * biozoop;
    double morp = P->nup * biophyt;
    double morpt = nupt * biophyt;
    double recy_don = nudon * bct * biodon;
    double recy_dop = nudop * bct * biodop;
    double morz = P->nuz * biozoop * biozoop;
    double remi = nud * bct * biodetr;
    double expo = wwd * biodetr;
    double expo_phos = wwd * biodetr_phos;
    double dissl = 0., expocaco3 = 0.;
    if (CACO3) {
      dissl = biocaco3 * dissk1;
      expocaco3 = wwc * biocaco3;
    }
    double graz_Diat = 0., morp_Diat = 0., morpt_Diat = 0., opldis = 0., expoopl = 0.;
    if (SIL) {
      const double g_Diat = gmax * ing_Diat * biodiat;
      graz_Diat = g_Diat * biozoop;
      morp_Diat = O->nu_diat * biodiat;
      morpt_Diat = nudt * biodiat;
      opldis = bioopl * opl_disk1;
      expoopl = wwo * bioopl;
    }
    double remife = nud * bct * biodetrfe;
    /* iron scavenging, mobi.F:2313-2342 */
    const double o2flag = tanh(g_max(o2, 0.));
    const double ligand = g_max(UV_POWP(g_max(aou, 40.), 0.8) / 66. + UV_POWP(biodon, 0.8) / 4.8, 0.5) / 1000.;
    const double fepa = (1.0 + P->kfeleq * (ligand - biodfe)) * o2flag;
    const double feprime = ((-fepa + sqrt(fepa * fepa + 4.0 * P->kfeleq * biodfe)) / (2.0 * P->kfeleq)) * o2flag;
    double feorgads = (P->kfeorg * (UV_POWP((biodetr * detrflag) * P->mc * redctn, 0.58)) * feprime) * o2flag;
    double fecol = P->kfecol * (feprime * feprime) * o2flag;
    double expofe = wwd * biodetrfe;
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
    npp = npp * no3flag * (dopupt_flag * dopflag + (1. - dopupt_flag) * po4flag) * din15flag;
    if (SIL) npp_Diat = npp_Diat * no3flag * (dopupt_Diat_flag * dopflag + (1. - dopupt_Diat_flag) * po4flag) * din15flag;
    npp_D = npp_D * (dopupt_D_flag * dopflag + (1. - dopupt_D_flag) * po4flag) * din15flag;
    graz_D = graz_D * diazflag * diazn15flag;
    morpt_D = morpt_D * diazflag * diazn15flag;
    morp_D = morp_D * diazflag * diazn15flag;
    no3upt_D = no3upt_D * no3flag * din15flag;
    recy_don = recy_don * donflag * don15flag;
    if (CACO3) {
      dissl = dissl * caco3flag;
      expocaco3 = expocaco3 * caco3flag;
    }
    if (SIL) {
      graz_Diat = graz_Diat * diatflag;
      morp_Diat = morp_Diat * diatflag;
      morpt_Diat = morpt_Diat * diatflag;
    }
    remife = remife * detrfeflag;
    feorgads = feorgads * dfeflag;
    expofe = expofe * detrfeflag;
    fecol = fecol * dfeflag;
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
    /* nitrogen-15 fractionation, mobi.F:2576-2636 */
    double fcassim = 0., fcrecy = 0., fcexcr = 0., fcnfix = 0.;
    double rtphytn15 = 0., rtdiatn15 = 0., rtzoopn15 = 0., rtdetrn15 = 0., rtdiazn15 = 0.;
    if (N15) {
      double uno3 = div_safe(npp * dtbio, biono3);
      uno3 = g_min(uno3, 0.999);
      uno3 = g_max(uno3, UV_TRCMIN);
      const double rno3 = g_clamp(biodin15 / (biono3 - biodin15), 2 * UV_RN15STD, UV_RN15STD / 2.);
      const double bassim = g_rayleigh(rno3, P->eps_assim, uno3);
      fcassim = div_safe(bassim, 1 + bassim);
      double udon = div_safe(recy_don * dtbio, biodon);
      udon = g_min(udon, 0.999);
      udon = g_max(udon, UV_TRCMIN);
      const double rdon = g_clamp(biodon15 / (biodon - biodon15), 2 * UV_RN15STD, UV_RN15STD / 2.);
      const double brecy = g_rayleigh(rdon, P->eps_recy, udon);
      fcrecy = div_safe(brecy, 1 + brecy);
      const double rzoop = g_clamp(biozoopn15 / (biozoop - biozoopn15), 2. * UV_RN15STD, UV_RN15STD / 2.);
      const double bexcr = rzoop - P->eps_excr * rzoop / 1000.;
      fcexcr = div_safe(bexcr, 1 + bexcr);
      const double bnfix = UV_RN15STD - P->eps_nfix * UV_RN15STD / 1000.;
      fcnfix = div_safe(bnfix, 1 + bnfix);
      rtphytn15 = g_clamp(div_safe(biophytn15, biophyt), rn15hi, rn15lo);
      if (SIL) rtdiatn15 = g_clamp(div_safe(biodiatn15, biodiat), rn15hi, rn15lo);
      rtzoopn15 = g_clamp(div_safe(biozoopn15, biozoop), rn15hi, rn15lo);
      rtdetrn15 = g_clamp(div_safe(biodetrn15, biodetr), rn15hi, rn15lo);
      rtdiazn15 = g_clamp(div_safe(biodiazn15, biodiaz), rn15hi, rn15lo);
    }
    /* carbon-13 fractionation, mobi.F:2637-2676 */
    double fcnpp = 0., rtdic13 = 0., rtphytc13 = 0., rtdiatc13 = 0., rtcaco3c13 = 0., rtzoopc13 = 0., rtdetrc13 = 0.;
    double rtdoc13 = 0., rtdiazc13 = 0.;
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
    double calpro;
    if (CACO3) calpro = ((sf_Z + morz) * capr + (sf_P + morp) * capr) * redctn * 1.e3;
    else calpro = (morp + morz + (graz + graz_Z) * (1. - gamma1)) * capr * redctn * 1.e3;
    double oplpro = 0.;
    if (SIL) { /* mobi.F:2683-2697 (O_mobi_iron) */
      const double negcoeff = -0.46204044117647, VTP = 1.60266544117647, tanh_m = 6.9, tanh_b = -3.673092;
      const double sipr0 = (negcoeff * tanh(tanh_m * biodfe * 1.e3 + tanh_b) + VTP);
      oplpro = (morp_Diat + sf_Diat) * sipr0 * silflag * (1.e-3);
      opldis = opldis * oplflag;
      expoopl = expoopl * oplflag;
    }
    /* variable P:C of new production (Galbraith & Martiny 2015), mobi.F:2699-2702 */
    const double GM15ptc = 0.0060 + 0.0069 * biopo4;
    const double GM15ptn = GM15ptc * redctn * 1.e3;
    /* prognostic updates, mobi.F:2712-3085; every right-hand side uses the OLD state */
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
    const double n_zoop = biozoop + dtbio * (dig - morz - graz_Z - excr);
    double n_detr, n_detr_phos;
    if (SIL) {
      n_detr = biodetr + dtbio * ((1. - dfr) * morp + sf + morz - remi - graz_Det - expo + impo + morp_D * rnd +
                                  (1. - dfr) * morp_Diat);
      n_detr_phos = biodetr_phos + dtbio * ((1. - dfr) * morp * ptn_P + sf_phos + morz * redptn - remi * ptn_detr -
                                            graz_Det * ptn_detr - expo_phos + impo_phos + morp_D * rnd * redptn +
                                            (1. - dfr) * morp_Diat * redptn);
    } else {
      n_detr = biodetr + dtbio * ((1. - dfr) * morp + sf + morz - remi - graz_Det - expo + impo + morp_D * rnd);
      n_detr_phos = biodetr_phos + dtbio * ((1. - dfr) * morp * ptn_P + sf_phos + morz * redptn - remi * ptn_detr -
                                            graz_Det * ptn_detr - expo_phos + impo_phos + morp_D * rnd * redptn);
    }
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
    const double n_caco3 = biocaco3 + dtbio * (calpro - dissl - expocaco3 + impocaco3);
    const double n_diat = biodiat + dtbio * (npp_Diat - morp_Diat - graz_Diat - morpt_Diat);
    const double n_sil = biosil + dtbio * (opldis - oplpro);
    const double n_opl = bioopl + dtbio * (oplpro - opldis - expoopl + impoopl);
    double n_dfe, n_detrfe;
    if (SIL) {
      n_dfe = biodfe + dtbio * (rfeton * (excr + (1. - dfrt) * morpt - npp + morpt_D - npp_D + recy_don + nr_excr_D +
                                          nr_excr_P + nr_excr_detr + morp_D * (1. - rnd)) -
                                feorgads + remife - fecol + rfeton * ((1. - dfrt) * morpt_Diat - npp_Diat));
      n_detrfe = biodetrfe + dtbio * (rfeton * (sf + (1. - dfr) * morp + morp_D * rnd + morz - graz_Det) + feorgads +
                                      P->iscr * fecol - remife - expofe + impofe + rfeton * (1. - dfr) * morp_Diat);
    } else {
      n_dfe = biodfe + dtbio * (rfeton * (excr + (1. - dfrt) * morpt - npp + morpt_D - npp_D + recy_don + nr_excr_D +
                                          nr_excr_P + nr_excr_detr + morp_D * (1. - rnd)) -
                                feorgads + remife - fecol);
      n_detrfe = biodetrfe + dtbio * (rfeton * (sf + (1. - dfr) * morp + morp_D * rnd + morz - graz_Det) + feorgads +
                                      P->iscr * fecol - remife - expofe + impofe);
    }
    double n_din15 = biodin15, n_don15 = biodon15, n_phytn15 = biophytn15, n_diatn15 = biodiatn15, n_zoopn15 = biozoopn15;
    double n_detrn15 = biodetrn15, n_diazn15 = biodiazn15;
    if (N15) {
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
      n_phytn15 = biophytn15 + dtbio * (fcassim * npp - rtphytn15 * morp - rtphytn15 * graz - rtphytn15 * morpt);
      n_diazn15 = biodiazn15 + dtbio * (fcnfix * (npp_D - no3upt_D) + fcassim * no3upt_D - rtdiazn15 * morp_D -
                                        rtdiazn15 * graz_D - rtdiazn15 * morpt_D);
    }
    double n_dic13 = biodic13, n_doc13 = biodoc13, n_phytc13 = biophytc13, n_zoopc13 = biozoopc13, n_detrc13 = biodetrc13;
    double n_diazc13 = biodiazc13, n_caco3c13 = biocaco3c13, n_diatc13 = biodiatc13;
    if (C13) {
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
      n_phytc13 = biophytc13 + dtbio * redctn * (fcnpp * npp - rtphytc13 * morp - rtphytc13 * graz - rtphytc13 * morpt);
      n_diazc13 = biodiazc13 + dtbio * redctn * (fcnpp * npp_D - rtdiazc13 * (morp_D + graz_D + morpt_D));
      if (CACO3)
        n_caco3c13 = biocaco3c13 + dtbio * (rtdic13 * calpro - rtcaco3c13 * dissl - rtcaco3c13 * expocaco3 + rcaco3c13impo);
    }
    biopo4 = n_po4; biodop = n_dop; biophyt = n_phyt; biophyt_phos = n_phyt_phos; biozoop = n_zoop; biodetr = n_detr;
    biodetr_phos = n_detr_phos; biodic = n_dic; biono3 = n_no3; biodon = n_don; biodiaz = n_diaz;
    ptn_P = biophyt_phos / biophyt;
    ptn_detr = biodetr_phos / biodetr;
    if (CACO3) biocaco3 = n_caco3;
    if (SIL) { biodiat = n_diat; biosil = n_sil; bioopl = n_opl; }
    biodfe = n_dfe; biodetrfe = n_detrfe;
    biodin15 = n_din15; biodon15 = n_don15; biophytn15 = n_phytn15; biodiatn15 = n_diatn15; biozoopn15 = n_zoopn15;
    biodetrn15 = n_detrn15; biodiazn15 = n_diazn15;
    biodic13 = n_dic13; biodoc13 = n_doc13; biophytc13 = n_phytc13; biozoopc13 = n_zoopc13; biodetrc13 = n_detrc13;
    biodiazc13 = n_diazc13; biocaco3c13 = n_caco3c13; biodiatc13 = n_diatc13;
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
    /* refresh flags that are still set, mobi.F:3175-3251 */
    if (po4flag == 1) po4flag = g_flag01(biopo4 - UV_TRCMIN);
    if (phytflag == 1) phytflag = g_flag01(biophyt - UV_TRCMIN);
    if (zoopflag == 1) zoopflag = g_flag01(biozoop - UV_TRCMIN);
    if (detrflag == 1) detrflag = g_flag01(biodetr - UV_TRCMIN);
    if (phyt_phosflag == 1) phyt_phosflag = g_flag01(biophyt_phos - UV_TRCMIN);
    if (detr_phosflag == 1) detr_phosflag = g_flag01(biodetr_phos - UV_TRCMIN);
    if (no3flag == 1) no3flag = g_flag01(biono3 - UV_TRCMIN);
    if (dopflag == 1) dopflag = g_flag01(biodop - UV_TRCMIN);
    if (donflag == 1) donflag = g_flag01(biodon - UV_TRCMIN);
    if (diazflag == 1) diazflag = g_flag01(biodiaz - UV_TRCMIN);
    if (N15) {
      if (din15flag == 1) din15flag = g_flag01(biodin15 - UV_TRCMIN);
      if (don15flag == 1) don15flag = g_flag01(biodon15 - UV_TRCMIN);
      if (phytn15flag == 1) phytn15flag = g_flag01(biophytn15 - UV_TRCMIN);
      if (SIL && diatn15flag == 1) diatn15flag = g_flag01(biodiatn15 - UV_TRCMIN);
      if (zoopn15flag == 1) zoopn15flag = g_flag01(biozoopn15 - UV_TRCMIN);
      if (detrn15flag == 1) detrn15flag = g_flag01(biodetrn15 - UV_TRCMIN);
      if (diazn15flag == 1) diazn15flag = g_flag01(biodiazn15 - UV_TRCMIN);
    }
    if (CACO3) {
      if (caco3flag == 1) caco3flag = g_flag01(biocaco3 - UV_TRCMIN);
      if (SIL) { /* (nested under O_mobi_caco3 in the reference, mobi.F:3212-3222) */
        if (diatflag == 1) diatflag = g_flag01(biodiat - UV_TRCMIN);
        if (silflag == 1) silflag = g_flag01(biosil - UV_TRCMIN);
        if (oplflag == 1) oplflag = g_flag01(bioopl - UV_TRCMIN);
      }
    }
    if (dfeflag == 1) dfeflag = g_flag01(biodfe - UV_TRCMIN);
    if (detrfeflag == 1) detrfeflag = g_flag01(biodetrfe - UV_TRCMIN);
    if (C13) {
      if (dic13flag == 1) dic13flag = g_flag01(biodic13 - UV_TRCMIN);
      if (phytc13flag == 1) phytc13flag = g_flag01(biophytc13 - UV_TRCMIN);
      if (SIL && diatc13flag == 1) diatc13flag = g_flag01(biodiatc13 - UV_TRCMIN);
      if (CACO3 && caco3c13flag == 1) caco3c13flag = g_flag01(biocaco3c13 - UV_TRCMIN);
      if (zoopc13flag == 1) zoopc13flag = g_flag01(biozoopc13 - UV_TRCMIN);
      if (detrc13flag == 1) detrc13flag = g_flag01(biodetrc13 - UV_TRCMIN);
      if (doc13flag == 1) doc13flag = g_flag01(biodoc13 - UV_TRCMIN);
      if (diazc13flag == 1) diazc13flag = g_flag01(biodiazc13 - UV_TRCMIN);
    }
  }
  (void)dic13flag; (void)doc13flag; (void)phytc13flag; (void)zoopc13flag; (void)detrc13flag; (void)diazc13flag;
  (void)diatn15flag; (void)diatc13flag; (void)caco3c13flag;
  _Pragma("unroll") for (int x = 0; x < X_COUNT; ++x) bioout[x] = 0.0;
#define BOUT(x, v) do { if (I[x] > 0) bioout[x] = (v) - bioin[x]; } while (0)
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
#undef BGET
#undef BOUT
}

// work planes of the general kernel: (imt,km,jmt) each, in the space of the MP_* / MA_* planes of kernels_mobi.hpp
enum { MPG_OMEGAC, MPG_AC13B, MPG_BCT, MPG_BCTZ, MPG_NUD, MPG_AOU, MPG_AVEJ, MPG_AVEJD, MPG_AVEJDIAT, MPG_O2F, MPG_AOUT, MPG_COUNT };
static_assert(MPG_COUNT <= UV_MOBI_WORK_PLANES, "work planes of mobi_store");
/* What a level needs that depends on the level's own inputs (or, for the light, on the inputs above it) only -- one thread
   per (ocean column, level), as mobi_pre_cell does for set C: the carbonate chemistry (mobi.F:766-789: co2calc_SWS, the 13C
   fractionation factor of photosynthesis), the light that reaches the level (tracer.F:381-390, mobi.F:799-820: the running
   product over the levels above), temperature and oxygen functions (:827-848), the oxygen saturation (tracer.F:456-476) and
   the light-limited growth rates (mobi_src, mobi.F:1961-2061). */
UVIC_DEV void mobig_pre_cell(const uvic_ctx &c, const mobi_dev &M, int i, int k, int j) {
  UV_DIMS(c);
  mobi_params_cp P = UV_CONST_AS(M.P);
  mobi_options_cp O = UV_CONST_OPT(M.O);
  const mobi_step &St = M.S;
  const size_t ij = X2(i, j);
  if (k > c.kmt[ij]) return;
  const int *I = O->im;
  const int CACO3 = O->caco3, SIL = O->silicon;
#define TIN(k, n) c.t_taum1[X3(i, k, j) + (size_t)((n)-1) * N3]
#define TNR(k, x) c.t_taum1[X3(i, k, j) + (size_t)(P->tracer_of_mobi[I[x] - 1] - 1) * N3]
#define TN(k, x) g_max(TNR(k, x), UV_TRCMIN)
#define PREG(q) M.pre[(size_t)(q) * N3 + X3(i, k, j)]
  const double t_in = TIN(k, P->itemp), s_in = 1.e3 * TIN(k, P->isalt) + 35.0, o2_in = TIN(k, P->io2) * 1000.;
  const double dic_in = TIN(k, P->idic), alk_in = TIN(k, P->ialk);
  {
    const double atmpres = 1.0, depth = P->zt[k - 1] / 100.;
    double pH, co2star, dco2star, pCO2, dpco2, CO3, Omega_c, Omega_a;
    mobi_co2calc_SWS(t_in, s_in, dic_in, alk_in, M.co2ccn, atmpres, depth, &pH, &co2star, &dco2star, &pCO2, &dpco2, &CO3, &Omega_c,
                     &Omega_a, M.carb_shared);
    const double ac13_DIC_aq = -1.0512994e-4 * t_in + 1.011765;
    const double ac13_aq_POC = -0.017 * log10(g_min(g_max(co2star * 1000., 2.), 74.)) + 1.0034;
    PREG(MPG_AC13B) = ac13_aq_POC / ac13_DIC_aq;
    PREG(MPG_OMEGAC) = Omega_c;
  }
  /* the caller's light geometry, tracer.F:381-390 */
  const double ai = M.aice[ij], hi = M.hice[ij], hs = M.hsno[ij];
  double rctheta = g_max(-1.5, g_min(1.5, M.tlat[ij] / M.radian - St.declin));
  rctheta = P->kw / sqrt(1. - (1. - g_sq(cos(rctheta))) / g_sq(1.33));
  double dayfrac = g_min(1., -tan(M.tlat[ij] / M.radian) * tan(St.declin));
  dayfrac = g_max(1e-12, acos(g_max(-1., dayfrac)) / M.pi);
  double swr = P->tap * M.dnswr[ij] * 1e-3 * (1. + ai * (exp(-P->ki * (hi + hs)) - 1.));
  /* attenuation by what lies above: the running product of mobi_driver's level loop, mobi.F:799-816 */
  double phin = 0.0, caco3in = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
  double att = 0.0;   // (device: one exp of the summed exponents instead of k, as in mobi_pre_cell)
#endif
  for (int m = 1; m <= k; ++m) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (CACO3) att = att + (P->kc * phin + O->kc_c * caco3in);
    else att = att + P->kc * phin;
#else
    if (CACO3) swr = swr * exp(-P->kc * phin - O->kc_c * caco3in);
    else swr = swr * exp(-P->kc * phin);
#endif
    phin = g_max(TN(m, X_phyt), UV_TRCMIN) * P->dzt[m - 1] + g_max(TN(m, X_diaz), UV_TRCMIN) * P->dzt[m - 1];
    if (SIL) phin = phin + g_max(TN(m, X_diat), UV_TRCMIN) * P->dzt[m - 1];
    if (CACO3) caco3in = caco3in + TNR(m, X_caco3) * P->dzt[m - 1];
  }
#if defined(__HIP_DEVICE_COMPILE__)
  swr = swr * exp(-att);
#endif
  const double gl = swr * exp(P->ztt[k - 1] * rctheta);
  const double bct = UV_POWP(P->bbio, P->cbio * t_in);
  const double bctz = (0.5 * (tanh(o2_in - 8.) + 1)) * UV_POWP(P->bbio, P->cbio * t_in);
  PREG(MPG_BCT) = bct;
  PREG(MPG_BCTZ) = bctz;
  PREG(MPG_NUD) = P->nud0 * (0.6 + 0.4 * tanh(0.22 * g_max(o2_in, 0.)));
  { /* oxygen saturation, tracer.F:456-476 */
    const double f1 = log((298.15 - t_in) / (273.15 + t_in));
    const double f2 = f1 * f1, f3 = f2 * f1, f4 = f3 * f1, f5 = f4 * f1;
    double o2sat = exp(2.00907 + 3.22014 * f1 + 4.05010 * f2 + 4.94457 * f3 - 2.56847E-1 * f4 + 3.88767 * f5 +
                       s_in * (-6.24523e-3 - 7.37614e-3 * f1 - 1.03410e-2 * f2 - 8.17083E-3 * f3) - 4.88682E-7 * s_in * s_in);
    o2sat = o2sat / 22391.6 * 1000.0 * 1000.;
    PREG(MPG_AOU) = o2sat - o2_in;
    // (for the team form, kernels_mobi_gt.hpp: two functions of the level's inputs that mobi_src would form in every sub-step)
    PREG(MPG_O2F) = tanh(g_max(o2_in, 0.));                                      // o2flag, mobi.F:2313
    PREG(MPG_AOUT) = UV_DIVC(UV_POWP(g_max(o2sat - o2_in, 40.), 0.8), 66.);      // the AOU term of the ligand concentration, mobi.F:2316
  }
  /* the light-limited growth rates from the level's (clamped) pools, mobi.F:1961-2061 */
  const double biophyt = TN(k, X_phyt), biodiaz = TN(k, X_diaz), biodfe = TN(k, X_dfe), dzt = P->dzt[k - 1];
  const double biodiat = SIL ? TN(k, X_diat) : 0.0, biocaco3 = CACO3 ? TN(k, X_caco3) : 0.0;
  double p1, p2, kfevar, deffe, avej, avej_D, avej_Diat = 0.0;
  /* iron-dependent Chl:C and initial slope, mobi.F:1961-1996 */
  p1 = g_min(biophyt, P->pmax);
  p2 = g_max(0.0, biophyt - P->pmax);
  kfevar = (P->kfemin * p1 + P->kfemax * p2) / (p1 + p2);
  deffe = biodfe / (kfevar + biodfe);
  const double thetamax = P->thetamaxlo + (P->thetamaxhi - P->thetamaxlo) * deffe;
  const double alpha_O = P->alphamin + (P->alphamax - P->alphamin) * deffe;
  const double gl_O = gl * thetamax * alpha_O;
  double kfevar_Diat = 0., deffe_Diat = 0., gl_Diat = 0.;
  if (SIL) {
    p1 = g_min(biodiat, O->pmax_Diat);
    p2 = g_max(0.0, biodiat - O->pmax_Diat);
    kfevar_Diat = (O->kfemin_Diat * p1 + O->kfemax_Diat * p2) / (p1 + p2);
    deffe_Diat = biodfe / (kfevar_Diat + biodfe);
    const double thetamax_Diat = P->thetamaxlo + (P->thetamaxhi - P->thetamaxlo) * deffe_Diat;
    const double alpha_Diat = P->alphamin + (P->alphamax - P->alphamin) * deffe_Diat;
    gl_Diat = gl * thetamax_Diat * alpha_Diat;
  }
  const double deffe_D = biodfe / (P->kfe_D + biodfe);
  const double thetamax_D = P->thetamaxlo + (P->thetamaxhi - P->thetamaxlo) * deffe_D;
  const double alpha_D = P->alphamin + (P->alphamax - P->alphamin) * deffe_D;
  const double gl_D = gl * thetamax_D * alpha_D;
  /* light, Evans & Parslow, mobi.F:1997-2061 */
  double psum = biophyt + biodiaz;
  if (SIL) psum = psum + biodiat;
  double kirr = -P->kw - P->kc * psum;
  if (CACO3) kirr = kirr - O->kc_c * biocaco3;
  const double f1 = exp(kirr * dzt);
  const double jmax = P->abio_P * bct * deffe;
  const double gd = jmax * dayfrac;
  double u1 = g_max(gl_O / gd, 1.e-6);
  double u2 = u1 * f1;
  double phi1 = log(u1 + sqrt(1. + u1 * u1)) - (sqrt(1. + u1 * u1) - 1.) / u1;
  double phi2 = log(u2 + sqrt(1. + u2 * u2)) - (sqrt(1. + u2 * u2) - 1.) / u2;
  avej = gd * (phi1 - phi2) / (-kirr * dzt);
  const double jmax_D = g_max(0., P->abio_P * (bct - P->dbct_D) * deffe_D) * P->jdiar;
  const double gd_D = g_max(1.e-14, jmax_D * dayfrac);
  u1 = g_max(gl_D / gd_D, 1.e-6);
  u2 = u1 * f1;
  phi1 = log(u1 + sqrt(1. + u1 * u1)) - (sqrt(1. + u1 * u1) - 1.) / u1;
  phi2 = log(u2 + sqrt(1. + u2 * u2)) - (sqrt(1. + u2 * u2) - 1.) / u2;
  avej_D = gd_D * (phi1 - phi2) / (-kirr * dzt);
  double jmax_Diat = 0.;
  if (SIL) {
    jmax_Diat = O->abiodiat * bct * deffe_Diat;
    const double gd_Diat = jmax_Diat * dayfrac;
    u1 = g_max(gl_Diat / gd_Diat, 1.e-6);
    u2 = u1 * f1;
    phi1 = log(u1 + sqrt(1. + u1 * u1)) - (sqrt(1. + u1 * u1) - 1.) / u1;
    phi2 = log(u2 + sqrt(1. + u2 * u2)) - (sqrt(1. + u2 * u2) - 1.) / u2;
    avej_Diat = gd_Diat * (phi1 - phi2) / (-kirr * dzt);
  }
  PREG(MPG_AVEJ) = avej;
  PREG(MPG_AVEJD) = avej_D;
  PREG(MPG_AVEJDIAT) = avej_Diat;
#undef TIN
#undef TNR
#undef TN
#undef PREG
}

/* mobi_driver (mobi.F:519-1482) with the caller's column set-up, iron inputs and 14C (tracer.F:355-545, 853-867) */
template <int TN15, int TC13, int TCACO3, int TSIL>
UVIC_DEV void mobig_column(const uvic_ctx &c, const mobi_dev &M, int i, int j) {
  UV_DIMS(c);
  mobi_params_cp P = UV_CONST_AS(M.P);
  mobi_options_cp O = UV_CONST_OPT(M.O);
  const mobi_step &St = M.S;
  const size_t ij = X2(i, j), NS = (size_t)imt * jmt;
  double *src = const_cast<double *>(c.src);
  const int kmx = c.kmt[ij];
  const double twodt = c.c2dtts;
  const int *I = O->im, *S = O->is;
  const int N15 = TN15 < 0 ? O->n15 : TN15, C13 = TC13 < 0 ? O->c13 : TC13, CACO3 = TCACO3 < 0 ? O->caco3 : TCACO3;
  const int SIL = TSIL < 0 ? O->silicon : TSIL;
/* tnpzd(k, x): the column is not copied -- a level is clamped (mobi.F:1894, through mobi_src's copy-out) before anything
   reads it except the light terms, which clamp themselves or (caco3in) want the raw value: TNR */
#define TIN(k, n) c.t_taum1[X3(i, k, j) + (size_t)((n)-1) * N3]
#define TNR(k, x) c.t_taum1[X3(i, k, j) + (size_t)(P->tracer_of_mobi[I[x] - 1] - 1) * N3]
#define TN(k, x) g_max(TNR(k, x), UV_TRCMIN)
#define SRC(k, s) src[X3(i, k, j) + (size_t)((s)-1) * N3]
#define PREG(q) M.pre[(size_t)(q) * N3 + X3(i, k, j)]
#define SX(k, x) SRC(k, S[x])
  double expo = 0.0, impo, expo_phos = 0.0, impo_phos, prca = 0.0;
  double rn15impo = 0.0, rn15expo = 0.0, rc13impo = 0.0, rc13expo = 0.0, prca13 = 0.0, expofe = 0.0, impofe;
  double rcaco3c13impo = 0.0, rcaco3c13expo = 0.0, impocaco3 = 0.0, expocaco3 = 0.0, dissk1 = 0.0;
  double expoopl = 0.0, impoopl = 0.0, opl_disk1 = 0.0;
  double capr = P->capr;
  double snpzd[X_COUNT], bioin[X_COUNT];
  for (int s = 1; s <= P->nsrc; ++s)
    for (int k = 1; k <= km; ++k) SRC(k, s) = 0.0;
  const double redctn = P->redctn;
  for (int k = 1; k <= kmx; ++k) {
    const double o2_in = TIN(k, P->io2) * 1000., dic_in = TIN(k, P->idic);
    const double aou_in = PREG(MPG_AOU), bct = PREG(MPG_BCT), bctz = PREG(MPG_BCTZ), nud = PREG(MPG_NUD);
    double rcalpro_k = 0.0, rdissl_k = 0.0, rexpocaco3_k = 0.0, rexpoopl_k = 0.0, bdeni_k = 0.0, nfix_k = 0.0;
    double dic_npzd_sms_k = 0.0, rtdic13_k = 0.0, rtcaco3c13_k = 0.0;
    if (N15) rn15impo = rn15expo;
    double ac13b = 0.0;
    if (C13 || CACO3) {
      /* co2calc_SWS (mobi.F:772) needs the cell's own T, S, DIC and alkalinity only: mobig_pre_cell has solved it for every
         cell at once and left what this loop uses of it */
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
    const double nudon = P->nudon0, nudop = P->nudop0;
    /* tnpzd(k,:) is passed as a strided section: copy in, clamp, copy out -- bioin[] holds the clamped level from here on */
    _Pragma("unroll") for (int x = 0; x < X_COUNT; ++x) bioin[x] = I[x] > 0 ? TNR(k, x) : 0.0;
    gsrc_out so;
    mobig_src<TN15, TC13, TCACO3, TSIL>(P, O, St, capr, bioin, PREG(MPG_AVEJ), PREG(MPG_AVEJD), PREG(MPG_AVEJDIAT), bct, impo, impo_phos, P->wd[k - 1], nud, impocaco3, O->wc[k - 1],
                 dissk1, impoopl, O->wo[k - 1], opl_disk1, nudop, nudon, snpzd, bctz, rn15impo, rc13impo, ac13b, rcaco3c13impo,
                 impofe, o2_in, aou_in, &so);
    expo = so.expo; expo_phos = so.expo_phos; expofe = so.expofe;
    if (N15) rn15expo = so.rn15expo;
    if (C13) rc13expo = so.rc13expo;
    if (C13 && CACO3) rcaco3c13expo = so.rcaco3c13expo;
    if (CACO3) expocaco3 = so.expocaco3;
    if (SIL) expoopl = so.expoopl;
    nfix_k = so.nfix;
    _Pragma("unroll") for (int x = 0; x < X_COUNT; ++x) snpzd[x] = snpzd[x] * St.rdtts;
    expofe = expofe * St.rnbio;
    if (CACO3) expocaco3 = expocaco3 * St.rnbio;
    if (SIL) {
      expoopl = expoopl * St.rnbio;
      rexpoopl_k = expoopl;
    }
    expo = expo * St.rnbio;
    expo_phos = expo_phos * St.rnbio;
    if (N15) rn15expo = rn15expo * St.rnbio;
    if (C13) {
      rc13expo = rc13expo * St.rnbio;
      if (CACO3) rcaco3c13expo = rcaco3c13expo * St.rnbio;
    }
    rcalpro_k = so.calpro * St.rnbio;
    if (CACO3) {
      rdissl_k = so.dissl * St.rnbio;
      rexpocaco3_k = expocaco3;
    }
    const double sgb = M.sg_bathy[ij + NS * (k - 1)];
#define SN(x) snpzd[x]
    /* benthic denitrification, mobi.F:1033-1085 */
    const double no3flag = g_flag01(bioin[X_no3] - UV_TRCMIN);
    const double din15flag = N15 ? g_flag01(bioin[X_din15] - UV_TRCMIN) : 1.;
    const double lno3 = 0.5 * tanh(bioin[X_no3] * 10 - 5.0);
    double sg_bdeni = (0.06 + 0.19 * UV_POWP(0.99, g_max(o2_in, UV_TRCMIN) - g_max(bioin[X_no3], UV_TRCMIN))) *
                      g_max(expo * sgb, UV_TRCMIN) * redctn * 1.e3;
    sg_bdeni = g_min(sg_bdeni, sgb * expo);
    sg_bdeni = g_max(sg_bdeni, 0.);
    sg_bdeni = sg_bdeni * (0.5 + lno3) * no3flag * din15flag;
    bdeni_k = sg_bdeni;
    SN(X_no3) = SN(X_no3) + sgb * expo - sg_bdeni;
    if (N15) {
      const double r15min = UV_TRCMIN * UV_RN15STD / (1 + UV_RN15STD);
      double rno3 = g_max(bioin[X_din15], r15min) / g_max(bioin[X_no3] - bioin[X_din15], r15min);
      rno3 = g_min(rno3, 2. * UV_RN15STD);
      rno3 = g_max(rno3, UV_RN15STD / 2.);
      const double eps_bdeni = P->eps_bdeni0 * exp(-2.5e-6 * (P->zt[k - 1]));
      const double bbdeni = rno3 - eps_bdeni * rno3 / 1000.;
      SN(X_din15) = SN(X_din15) + rn15expo * sgb * expo - bbdeni / (1 + bbdeni) * sg_bdeni;
    }
    /* sedimentary iron release, mobi.F:1086-1123 */
    const double coxdepth = g_min(g_max(P->zt[k - 1], 50000.), 150000.);
    const double oblinc = -1.26e-6 * coxdepth + 0.203;
    const double obexpc = -6.e-7 * coxdepth + 1.14;
    const double dztk = P->dzt[k - 1];
    const double nburial = (oblinc * UV_POWP(expo * sgb * dztk / 100 * 86400. * 365. * redctn * 1000., obexpc)) /
                           (86400. * 365. * dztk / 100 * redctn * 1000.);
    const double coxsed = expo * sgb - nburial;
    const double fesedmax = 85.;
    const double fesed = fesedmax * tanh(coxsed * redctn * 1000 * dztk / 100 * 86400. / o2_in) / (dztk / 100 * 86400 * 1000);
    SN(X_dfe) = SN(X_dfe) + fesed;
    /* bottom remineralisation, mobi.F:1124-1134 */
    SN(X_po4) = SN(X_po4) + sgb * expo_phos;
    SN(X_dic) = SN(X_dic) + sgb * expo * redctn;
    if (C13) {
      SN(X_dic13) = SN(X_dic13) + rc13expo * sgb * redctn;
      rc13expo = rc13expo - sgb * rc13expo;
    }
    expo = expo - sgb * expo;
    expo_phos = expo_phos - sgb * expo_phos;
    /* scatter into the source slots, mobi.F:1149-1205 */
    _Pragma("unroll") for (int x = 0; x < X_COUNT; ++x)
      if (I[x] > 0 && S[x] > 0) SRC(k, S[x]) = snpzd[x];
    /* DIC / alkalinity / 13C bookkeeping, mobi.F:1228-1266 */
    dic_npzd_sms_k = SN(X_dic);
    const double dprca = rcalpro_k * 1e-3;
    prca = prca + dprca * dztk;
    if (!CACO3) SX(k, X_dic) = SN(X_dic) - dprca;
    if (C13) {
      const double r13min = UV_TRCMIN * UV_RC13STD / (1 + UV_RC13STD);
      double r = g_max(bioin[X_dic13], r13min) / g_max(dic_in, UV_TRCMIN);
      r = g_min(r, 2. * UV_RC13STD / (1 + UV_RC13STD));
      r = g_max(r, 0.5 * UV_RC13STD / (1 + UV_RC13STD));
      rtdic13_k = r;
      prca13 = prca13 + dprca * dztk * r;
      if (CACO3) {
        double rc = g_max(bioin[X_caco3c13], r13min) / g_max(bioin[X_caco3], UV_TRCMIN);
        rc = g_min(rc, 2. * UV_RC13STD / (1 + UV_RC13STD));
        rc = g_max(rc, 0.5 * UV_RC13STD / (1 + UV_RC13STD));
        rtcaco3c13_k = rc;
      } else {
        SX(k, X_dic13) = SX(k, X_dic13) - r * dprca;
      }
    }
    if (CACO3) SRC(k, O->is_alk) = -SN(X_dic) * P->redntc * 1.e-3;
    else SRC(k, O->is_alk) = -SN(X_dic) * P->redntc * 1.e-3 - 2. * dprca;
    /* export of this level becomes the import of the next, mobi.F:1268-1287 */
    expo = expo * dztk;
    expo_phos = expo_phos * dztk;
    if (C13) {
      rc13expo = rc13expo * dztk;
      if (CACO3) rcaco3c13expo = rcaco3c13expo * dztk;
    }
    expofe = expofe * dztk;
    if (CACO3) expocaco3 = expocaco3 * dztk;
    if (SIL) expoopl = expoopl * dztk;
    { /* mobi_driver's second loop (oxygen, water-column denitrification, 15N: mobi.F:1302-1365) touches level k only */
      const double fo2 = tanh(0.22 * g_max(o2_in, 0.));
      const double so2 = dic_npzd_sms_k * P->redotc + nfix_k * St.rnbio * 1.25e-3;
      const double no3flag = g_flag01(bioin[X_no3] - UV_TRCMIN);
      const double din15flag = N15 ? g_flag01(bioin[X_din15] - UV_TRCMIN) : 1.;
      const double lno3 = 0.5 * tanh(bioin[X_no3] - 2.5);
      double wcdeni = 800. * no3flag * so2 * (1.0 - fo2) * (0.5 + lno3);
      if (N15) wcdeni = wcdeni * din15flag;
      wcdeni = g_max(wcdeni, 0.);
      SX(k, X_no3) = SX(k, X_no3) - wcdeni;
      if (N15) {
        double uno3 = wcdeni * twodt / bioin[X_no3];
        uno3 = g_min(uno3, 0.999);
        uno3 = g_max(uno3, UV_TRCMIN);
        const double r15min = UV_TRCMIN * UV_RN15STD / (1 + UV_RN15STD);
        double rno3 = g_max(bioin[X_din15], r15min) / g_max(bioin[X_no3] - bioin[X_din15], r15min);
        rno3 = g_min(rno3, 2. * UV_RN15STD);
        rno3 = g_max(rno3, UV_RN15STD / 2.);
        const double bwcdeni = g_rayleigh(rno3, P->eps_wcdeni, uno3);
        SX(k, X_din15) = SX(k, X_din15) - (bwcdeni / (1 + bwcdeni)) * wcdeni;
      }
      SRC(k, O->is_alk) = SRC(k, O->is_alk) + wcdeni * 1.e-3;
      SRC(k, O->is_alk) = SRC(k, O->is_alk) + bdeni_k * 1.e-3;
      SRC(k, O->is_alk) = SRC(k, O->is_alk) - nfix_k * St.rnbio * 1.e-3;
      SRC(k, O->is_o2) = -so2 * fo2;
      }
    if (CACO3) { /* ... and so does the third with prognostic CaCO3 (mobi.F:1373-1436) */
      const double rexp = (k == kmx) ? rexpocaco3_k : 0.0;
      if (k < kmx) {
        SX(k, X_dic) = SX(k, X_dic) + rdissl_k * 1.e-3 - rcalpro_k * 1.e-3;
        if (C13) SX(k, X_dic13) = SX(k, X_dic13) + rdissl_k * 1.e-3 * rtcaco3c13_k - rcalpro_k * 1.e-3 * rtdic13_k;
        SRC(k, O->is_alk) = SRC(k, O->is_alk) + 2. * rdissl_k * 1.e-3 - 2. * rcalpro_k * 1.e-3;
      } else {
        SX(k, X_dic) = SX(k, X_dic) + rdissl_k * 1.e-3 - rcalpro_k * 1.e-3 + rexp * 1.e-3;
        if (C13) SX(k, X_dic13) = SX(k, X_dic13) + rdissl_k * 1.e-3 * rtcaco3c13_k - rcalpro_k * 1.e-3 * rtdic13_k + rexp * 1.e-3 * rtcaco3c13_k;
        SRC(k, O->is_alk) = SRC(k, O->is_alk) + 2. * rdissl_k * 1.e-3 - 2. * rcalpro_k * 1.e-3 + 2. * rexp * 1.e-3;
      }
    }
    if (SIL && k == kmx) SX(k, X_sil) = SX(k, X_sil) + rexpoopl_k;
  }
  if (!CACO3) { /* the third loop without prognostic CaCO3: the column's calcite production comes back by a fixed profile */
    for (int k = 1; k <= kmx - 1; ++k) {
      SX(k, X_dic) = SX(k, X_dic) + prca * P->rcak[k - 1];
      if (C13) SX(k, X_dic13) = SX(k, X_dic13) + prca13 * P->rcak[k - 1];
      SRC(k, O->is_alk) = SRC(k, O->is_alk) + 2. * prca * P->rcak[k - 1];
    }
    SX(kmx, X_dic) = SX(kmx, X_dic) + prca * P->rcab[kmx - 1];
    if (C13) SX(kmx, X_dic13) = SX(kmx, X_dic13) + prca13 * P->rcab[kmx - 1];
    SRC(kmx, O->is_alk) = SRC(kmx, O->is_alk) + 2. * prca * P->rcab[kmx - 1];
  }
  /* iron inputs, tracer.F:538-545 */
  const int isdfe = S[X_dfe];
  SRC(1, isdfe) = SRC(1, isdfe) + M.fe_atmdep[ij + NS * (St.month - 1)] * 1000 / (P->dzt[0] / 100.);
  for (int k = 1; k <= kmx; ++k) SRC(k, isdfe) = SRC(k, isdfe) + M.fe_hydr[ij + NS * (k - 1)];
  /* carbon-14, tracer.F:853-867 */
  if (O->is_c14 > 0)
    for (int k = 1; k <= kmx; ++k) SRC(k, O->is_c14) = SRC(k, S[X_dic]) * UV_RC14STD - 3.836e-12 * TIN(k, P->ic14);
#undef PREG
#undef SN
#undef TN
#undef TNR
#undef TIN
#undef SRC
#undef SX
}


#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang fp contract(off)
#endif
}  // namespace uvic
#endif
