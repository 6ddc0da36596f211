/* oracle/mobi_oracle.c -- CPU restatement of the MOBI source-term path for the
 * option set "c30" (BASELINE config 4 = SURVEY.md option set C: O_mobi O_mobi_o2
 * O_mobi_iron O_carbon O_mobi_alk O_mobi_nitrogen O_carbon_13 O_carbon_14
 * O_mobi_nitrogen_15; nt=30, nsrc=28, ntnpzd=25).
 *
 * TEST INFRASTRUCTURE ONLY (see uvic_oracle.h).  Follows, with the cpp options
 * of that set resolved,
 *   u09/mom/tracer.F:311-545   column set-up, light geometry, O2 saturation, call, iron inputs
 *   u09/mom/tracer.F:853-867   carbon-14 source
 *   u09/mom/mobi.F:519-1482    mobi_driver (three passes over the column)
 *   u09/mom/mobi.F:1485-3313   mobi_src (nbio forward-Euler sub-steps of the ecosystem)
 *   u09/common/co2calc.F:1-526 co2calc_SWS, drtsafe, ta_iter_SWS
 * ("u09/" = /root/reference/updates/09/source/).  Quantities the reference keeps
 * in COMMON and overwrites per grid box (ptn_P, ntp_P, ptn_detr, k1n, k1p_P and
 * the carbonate constants) are locals here; the clamp `bioin = max(bioin,trcmin)`
 * that mobi_src applies to the CALLER's column (mobi.F:1894, copy-in/copy-out of
 * tnpzd(k,:)) is kept because mobi_driver reads the clamped values afterwards.
 * Pinned against oracle/_ref (tests/test_mobi_oracle.py) and tests/golden.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

#include "mobi_oracle.h"

#define TRCMIN 5e-12      /* u09/mom/mobi.h:199-212 */
#define RN15STD 0.0036765
#define RC13STD 0.0112372
#define RC14STD 1.176e-12

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }
/* 0.5 + sign(0.5, x): 1 when x >= 0 (or +0), 0 otherwise */
static inline double flag01(double x) { return 0.5 + copysign(0.5, x); }
static inline double sq(double x) { return x * x; }

/* ------------------------------------------------------------------------- */
/* carbonate chemistry, co2calc.F                                             */
/* ------------------------------------------------------------------------- */
typedef struct {
  double k1, k2, k1p, k2p, k3p, ksi, kw, ks, kf, kb, bt, st, ft, pt, sit, ta, dic;
} carb_t;

/* co2calc.F:455-526 */
static void ta_iter_SWS(const carb_t *q, double x, double *fn, double *df) {
  const double x2 = x * x, x3 = x2 * x;
  const double k12 = q->k1 * q->k2, k12p = q->k1p * q->k2p, k123p = k12p * q->k3p;
  const double c = 1.0 + q->st / q->ks + q->ft / q->kf;
  const double a = x3 + q->k1p * x2 + k12p * x + k123p;
  const double a2 = a * a;
  const double da = 3.0 * x2 + 2.0 * q->k1p * x + k12p;
  const double b = x2 + q->k1 * x + k12;
  const double b2 = b * b;
  const double db = 2.0 * x + q->k1;
  const double dic = q->dic, pt = q->pt, bt = q->bt, st = q->st, ft = q->ft, sit = q->sit;
  *fn = q->k1 * x * dic / b + 2.0 * dic * k12 / b + bt / (1.0 + x / q->kb) + q->kw / x + pt * k12p * x / a +
        2.0 * pt * k123p / a + sit / (1.0 + x / q->ksi) - x / c - st / (1.0 + q->ks / (x / c)) -
        ft / (1.0 + q->kf / (x / c)) - pt * x3 / a - q->ta;
  *df = ((q->k1 * dic * b) - q->k1 * x * dic * db) / b2 - 2.0 * dic * k12 * db / b2 - bt / q->kb / sq(1.0 + x / q->kb) -
        q->kw / x2 + (pt * k12p * (a - x * da)) / a2 - 2.0 * pt * k123p * da / a2 - sit / q->ksi / sq(1.0 + x / q->ksi) -
        1.0 / c - st * (1.0 / sq(1.0 + q->ks / (x / c))) * (q->ks * c / x2) -
        ft * (1.0 / sq(1.0 + q->kf / (x / c))) * (q->kf * c / x2) - pt * x2 * (3.0 * a - x * da) / a2;
}

/* co2calc.F:401-453: bracketed Newton (Numerical Recipes rtsafe) */
static double drtsafe(const carb_t *q, double x1, double x2, double xacc) {
  const int maxit = 100;
  double fl, fh, df, f, xl, xh, swap, r, dxold, dx, temp;
  ta_iter_SWS(q, x1, &fl, &df);
  ta_iter_SWS(q, x2, &fh, &df);
  if (fl < 0.0) {
    xl = x1; xh = x2;
  } else {
    xh = x1; xl = x2;
    swap = fl; fl = fh; fh = swap;
  }
  r = 0.5 * (x1 + x2);
  dxold = fabs(x2 - x1);
  dx = dxold;
  ta_iter_SWS(q, r, &f, &df);
  for (int j = 1; j <= maxit; ++j) {
    if (((r - xh) * df - f) * ((r - xl) * df - f) >= 0. || fabs(2.0 * f) > fabs(dxold * df)) {
      dxold = dx;
      dx = 0.5 * (xh - xl);
      r = xl + dx;
      if (xl == r) return r;
    } else {
      dxold = dx;
      dx = f / df;
      temp = r;
      r = r - dx;
      if (temp == r) return r;
    }
    if (fabs(dx) < xacc) return r;
    ta_iter_SWS(q, r, &f, &df);
    if (f < 0.0) {
      xl = r; fl = f;
    } else {
      xh = r; fh = f;
    }
  }
  (void)fl; (void)fh;
  return r;
}

/* co2calc.F:1-399; only the outputs mobi_driver uses are returned */
void orc_co2calc_SWS(double t, double s, double dic_in, double ta_in, double co2_in, double atmpres, double depth,
                     double *ph, double *co2star_o, double *dco2star_o, double *pCO2_o, double *dpco2_o, double *CO3_o,
                     double *Omega_c, double *Omega_a) {
  carb_t q;
  const double phhi = 6., phlo = 10.;
  const double sit_in = 7.6875e-03, pt_in = 0.5125e-3;
  const double permil = 1.0 / 1024.5;
  q.pt = pt_in * permil;
  q.sit = sit_in * permil;
  q.ta = ta_in * permil;
  q.dic = dic_in * permil;
  const double C2K = 273.15;
  const double pres = depth * 0.1;
  const double permeg = 1.e-6;
  const double co2 = co2_in * permeg;
  const double tk = C2K + t;
  const double tk100 = tk / 100.0;
  const double tk1002 = tk100 * tk100;
  const double invtk = 1.0 / tk;
  const double dlogtk = log(tk);
  const double is = 19.924 * s / (1000. - 1.005 * s);
  const double is2 = is * is;
  const double sqrtis = sqrt(is);
  const double s2 = s * s;
  const double t2 = t * t;
  const double sqrts = sqrt(s);
  const double s15 = pow(s, 1.5);
  const double scl = s / 1.80655;
  const double pitkR = pres / tk / 83.15;
  const double p2itkR = pres * pitkR;
  q.bt = 0.000232 * scl / 10.811;
  q.st = 0.14 * scl / 96.062;
  q.ft = 0.000067 * scl / 18.9984;
  const double ff = exp(-162.8301 + 218.2968 / tk100 + 90.9241 * log(tk100) - 1.47696 * tk1002 +
                        s * (.025695 - .025225 * tk100 + 0.0049867 * tk1002));
  const double k0 = exp(93.4517 / tk100 - 60.2409 + 23.3585 * log(tk100) + s * (.023517 - 0.023656 * tk100 + 0.0047036 * tk1002));
  const double rt_x = 83.1451 * tk;
  const double delta_x = (57.7 - 0.118 * tk);
  double b_x = -1636.75 + 12.0408 * tk - 0.0327957 * tk * tk;
  b_x = b_x + 3.16528 * 1e-5 * tk * tk * tk;
  const double FugFac = exp((b_x + 2 * delta_x) * 1 / rt_x);
  q.k1 = pow(10., -1. * (3670.7 * invtk - 62.008 + 9.7944 * dlogtk - 0.0118 * s + 0.000116 * s2)) *
         exp((25.5 - 0.1271 * t) * pitkR + 0.5 * (-3.08e-3 + 8.77e-5 * t) * p2itkR);
  q.k2 = pow(10., -1 * (1394.7 * invtk + 4.777 - 0.0184 * s + 0.000118 * s2)) *
         exp((15.82 + 0.0219 * t) * pitkR + 0.5 * (1.13e-3 - 1.475e-4 * t) * p2itkR);
  q.k1p = exp(-4576.752 * invtk + 115.540 - 18.453 * dlogtk + (-106.736 * invtk + 0.69171) * sqrts +
              (-0.65643 * invtk - 0.01844) * s) *
          exp((14.51 - 0.1211 * t + 3.21e-4 * t2) * pitkR + 0.5 * (-2.67e-3 + 4.27e-5 * t) * p2itkR);
  q.k2p = exp(-8814.715 * invtk + 172.1033 - 27.927 * dlogtk + (-160.340 * invtk + 1.3566) * sqrts +
              (0.37335 * invtk - 0.05778) * s) *
          exp((23.12 - 0.1758 * t + 2.647e-3 * t2) * pitkR + 0.5 * (-5.15e-3 + 9.0e-5 * t) * p2itkR);
  q.k3p = exp(-3070.75 * invtk - 18.126 + (17.27039 * invtk + 2.81197) * sqrts + (-44.99486 * invtk - 0.09984) * s) *
          exp((26.57 - 0.202 * t + 3.042e-3 * t2) * pitkR + 0.5 * (-4.08e-3 + 7.14e-5 * t) * p2itkR);
  q.ksi = exp(-8904.2 * invtk + 117.400 - 19.334 * dlogtk + (-458.79 * invtk + 3.5913) * sqrtis +
              (188.74 * invtk - 1.5998) * is + (-12.1652 * invtk + 0.07871) * is2 + log(1.0 - 0.001005 * s)) *
          exp((29.48 - 0.1622 * t - 2.608e-3 * t2) * pitkR + 0.5 * (-2.84e-3) * p2itkR);
  q.kw = exp(-13847.26 * invtk + 148.9802 - 23.6521 * dlogtk + (118.67 * invtk - 5.977 + 1.0495 * dlogtk) * sqrts -
             0.01615 * s) *
         exp((20.02 - 0.1119 * t + 1.409e-3 * t2) * pitkR + 0.5 * (-5.13e-3 + 7.94e-5 * t) * p2itkR);
  q.ks = exp(-4276.1 * invtk + 141.328 - 23.093 * dlogtk + (-13856 * invtk + 324.57 - 47.986 * dlogtk) * sqrtis +
             (35474 * invtk - 771.54 + 114.723 * dlogtk) * is - 2698 * invtk * pow(is, 1.5) + 1776 * invtk * is2 +
             log(1.0 - 0.001005 * s)) *
         exp((18.03 - .0466 * t - 3.16e-4 * t2) * pitkR + 0.5 * (-4.53e-3 + 9.0e-5 * t) * p2itkR);
  q.kf = exp(1590.2 * invtk - 12.641 + 1.525 * sqrtis + log(1.0 - 0.001005 * s)) *
         exp((9.78 + 9.0e-3 * t + 9.42e-4 * t2) * pitkR + 0.5 * (-3.91e-3 + 5.4e-5 * t) * p2itkR);
  q.kb = exp((-8966.90 - 2890.53 * sqrts - 77.942 * s + 1.728 * s15 - 0.0996 * s2) * invtk +
             (148.0248 + 137.1942 * sqrts + 1.62142 * s) + (-24.4344 - 25.085 * sqrts - 0.2474 * s) * dlogtk +
             0.053105 * sqrts * tk + log((1 + (q.st / q.ks) + (q.ft / q.kf)) / (1 + (q.st / q.ks)))) *
         exp((29.48 - 0.1622 * t - 2.608e-3 * t2) * pitkR + 0.5 * (-2.84e-3) * p2itkR);
  const double x1 = pow(10.0, -phhi);
  const double x2 = pow(10.0, -phlo);
  const double xacc = 1.e-10;
  const double hSWS = drtsafe(&q, x1, x2, xacc);
  const double hSWS2 = hSWS * hSWS;
  double co2star = q.dic * hSWS2 / (hSWS2 + q.k1 * hSWS + q.k1 * q.k2);
  const double co2starair = co2 * ff * atmpres;
  double dco2star = co2starair - co2star;
  *ph = -log10(hSWS);
  double pCO2 = co2star / (k0 * FugFac);
  double dpCO2 = pCO2 - co2starair;
  double CO3 = q.k1 * q.k2 * co2star / hSWS2;
  const double sqs = pow(s, 0.5), sq35 = pow(s / 35., 0.5);
  double Kspc = exp(-395.8293 + (6537.773 / tk) + 71.595 * log(tk) - 0.17959 * tk +
                    (-1.78938 + (410.64 / tk) + 0.0065453 * tk) * sqs - 0.17755 * s + 0.0094979 * s15);
  double Kspa = exp(-395.9180 + (6685.079 / tk) + 71.595 * log(tk) - 0.17959 * tk +
                    (-0.157481 + (202.938 / tk) + 0.0039780 * tk) * sqs - 0.23067 * s + 0.0136808 * s15);
  const double DVc = -65.28 + 0.397 * t - 0.005155 * (t * t) + (19.816 - 0.0441 * t - 0.00017 * (t * t)) * sq35;
  const double DVa = -65.50 + 0.397 * t - 0.005155 * (t * t) + (19.82 - 0.0441 * t - 0.00017 * (t * t)) * sq35;
  const double DK = 0.01847 + 0.0001956 * t - 0.000002212 * (t * t) + (-0.03217 - 0.0000711 * t + 0.000002212) * sq35;
  Kspc = Kspc * exp(-DVc * pitkR + 0.5 * DK * p2itkR);
  Kspa = Kspa * exp(-DVa * pitkR + 0.5 * DK * p2itkR);
  const double Ca = 10.28E-3;
  *Omega_c = Ca * CO3 / Kspc;
  *Omega_a = Ca * CO3 / Kspa;
  *co2star_o = co2star / permil;
  *dco2star_o = dco2star / permil;
  *CO3_o = CO3 / permil;
  *pCO2_o = pCO2 / permeg;
  *dpco2_o = dpCO2 / permeg;
}

/* ------------------------------------------------------------------------- */
/* mobi_src, mobi.F:1485-3313: one grid box, nbio sub-steps                    */
/* ------------------------------------------------------------------------- */
typedef struct {
  double expo, expo_phos, calpro, nfix, rn15expo, rc13expo, expofe, remife;
} src_out_t;

/* Rayleigh-type fractionation factor: r + eps*(1-u)/u*log(1-u)*r/1000 (e.g. mobi.F:2589-2600) */
static inline double rayleigh(double r, double eps, double u) { return r + eps * (1 - u) / u * log(1 - u) * r / 1000.; }
static inline double clamp_ratio(double r, double hi, double lo) {
  r = dmin(r, hi);
  r = dmax(r, lo);
  return r;
}

static void mobi_src(const orc_mobi *P, double *bioin, double gl, double bct, double impo, double dzt, double impo_phos,
                     double dayfrac, double wwd, double nud, double nudop, double nudon, double *bioout, double bctz,
                     double rn15impo, double rc13impo, double ac13b, double impofe, double o2, double aou,
                     src_out_t *out) {
  const orc_mobi_index *I = &P->im;
#define BIN(m) bioin[(m)-1]
  double biopo4 = BIN(I->po4), biophyt = BIN(I->phyt), biophyt_phos = BIN(I->phyt_phos), biozoop = BIN(I->zoop);
  double biodetr = BIN(I->detr), biodetr_phos = BIN(I->detr_phos);
  double ptn_P = biophyt_phos / biophyt;
  double ptn_detr = biodetr_phos / biodetr;
  double biodic = BIN(I->dic), biodop = BIN(I->dop), biono3 = BIN(I->no3), biodon = BIN(I->don), biodiaz = BIN(I->diaz);
  double biodin15 = BIN(I->din15), biodon15 = BIN(I->don15), biophytn15 = BIN(I->phytn15), biozoopn15 = BIN(I->zoopn15);
  double biodetrn15 = BIN(I->detrn15), biodiazn15 = BIN(I->diazn15);
  double biodic13 = BIN(I->dic13), biophytc13 = BIN(I->phytc13), biozoopc13 = BIN(I->zoopc13), biodetrc13 = BIN(I->detrc13);
  double biodoc13 = BIN(I->doc13), biodiazc13 = BIN(I->diazc13), biodfe = BIN(I->dfe), biodetrfe = BIN(I->detrfe);
  /* negative-prevention flags from the unclamped input, mobi.F:1814-1891 */
  double po4flag = flag01(biopo4 - TRCMIN), phytflag = flag01(biophyt - TRCMIN), zoopflag = flag01(biozoop - TRCMIN);
  double detrflag = flag01(biodetr - TRCMIN), phyt_phosflag = flag01(biophyt_phos - TRCMIN);
  double detr_phosflag = flag01(biodetr_phos - TRCMIN);
  const double sf_P_phosflag = flag01(ptn_P - P->gamma1 * P->redptn);
  const double sf_detr_phosflag = flag01(ptn_detr - P->gamma1 * P->redptn);
  double dopflag = flag01(biodop - TRCMIN), no3flag = flag01(biono3 - TRCMIN), donflag = flag01(biodon - TRCMIN);
  double diazflag = flag01(biodiaz - TRCMIN), din15flag = flag01(biodin15 - TRCMIN), don15flag = flag01(biodon15 - TRCMIN);
  double phytn15flag = flag01(biophytn15 - TRCMIN), zoopn15flag = flag01(biozoopn15 - TRCMIN);
  double detrn15flag = flag01(biodetrn15 - TRCMIN), diazn15flag = flag01(biodiazn15 - TRCMIN);
  double dic13flag = flag01(biodic13 - TRCMIN), phytc13flag = flag01(biophytc13 - TRCMIN);
  double zoopc13flag = flag01(biozoopc13 - TRCMIN), detrc13flag = flag01(biodetrc13 - TRCMIN);
  double doc13flag = flag01(biodoc13 - TRCMIN), diazc13flag = flag01(biodiazc13 - TRCMIN);
  double dfeflag = flag01(biodfe - TRCMIN), detrfeflag = flag01(biodetrfe - TRCMIN);
  /* clamp the caller's column and the working copies, mobi.F:1894-1960 */
  for (int m = 0; m < P->ntnpzd; ++m) bioin[m] = dmax(bioin[m], TRCMIN);
  biopo4 = dmax(biopo4, TRCMIN); biophyt = dmax(biophyt, TRCMIN); biozoop = dmax(biozoop, TRCMIN);
  biodetr = dmax(biodetr, TRCMIN); biophyt_phos = dmax(biophyt_phos, TRCMIN); biodetr_phos = dmax(biodetr_phos, TRCMIN);
  biodic = dmax(biodic, TRCMIN); biono3 = dmax(biono3, TRCMIN); biodop = dmax(biodop, TRCMIN);
  biodon = dmax(biodon, TRCMIN); biodiaz = dmax(biodiaz, TRCMIN); biodin15 = dmax(biodin15, TRCMIN);
  biodon15 = dmax(biodon15, TRCMIN); biophytn15 = dmax(biophytn15, TRCMIN); biozoopn15 = dmax(biozoopn15, TRCMIN);
  biodetrn15 = dmax(biodetrn15, TRCMIN); biodiazn15 = dmax(biodiazn15, TRCMIN); biodic13 = dmax(biodic13, TRCMIN);
  biophytc13 = dmax(biophytc13, TRCMIN); biozoopc13 = dmax(biozoopc13, TRCMIN); biodetrc13 = dmax(biodetrc13, TRCMIN);
  biodoc13 = dmax(biodoc13, TRCMIN); biodiazc13 = dmax(biodiazc13, TRCMIN); biodfe = dmax(biodfe, TRCMIN);
  biodetrfe = dmax(biodetrfe, TRCMIN);
  /* light-limited growth, Evans & Parslow, with iron-dependent Chl:C, mobi.F:1984-2061 */
  double p1 = dmin(biophyt, P->pmax);
  double p2 = dmax(0.0, biophyt - P->pmax);
  double kfevar = (P->kfemin * p1 + P->kfemax * p2) / (p1 + p2);
  double deffe = biodfe / (kfevar + biodfe);
  const double thetamax = P->thetamaxlo + (P->thetamaxhi - P->thetamaxlo) * deffe;
  const double alpha_O = P->alphamin + (P->alphamax - P->alphamin) * deffe;
  const double gl_O = gl * thetamax * alpha_O;
  double deffe_D = biodfe / (P->kfe_D + biodfe);
  const double thetamax_D = P->thetamaxlo + (P->thetamaxhi - P->thetamaxlo) * deffe_D;
  const double alpha_D = P->alphamin + (P->alphamax - P->alphamin) * deffe_D;
  const double gl_D = gl * thetamax_D * alpha_D;
  const double kirr = -P->kw - P->kc * (biophyt + biodiaz);
  const double f1 = exp(kirr * dzt);
  double jmax = P->abio_P * bct * deffe;
  const double gd = jmax * dayfrac;
  double u1 = dmax(gl_O / gd, 1.e-6);
  double u2 = u1 * f1;
  double phi1 = log(u1 + sqrt(1. + u1 * u1)) - (sqrt(1. + u1 * u1) - 1.) / u1;
  double phi2 = log(u2 + sqrt(1. + u2 * u2)) - (sqrt(1. + u2 * u2) - 1.) / u2;
  const double avej = gd * (phi1 - phi2) / (-kirr * dzt);
  const double gmax = P->gbio * bctz;
  double jmax_D = dmax(0., P->abio_P * (bct - P->dbct_D) * deffe_D) * P->jdiar;
  const double gd_D = dmax(1.e-14, jmax_D * dayfrac);
  u1 = dmax(gl_D / gd_D, 1.e-6);
  u2 = u1 * f1;
  phi1 = log(u1 + sqrt(1. + u1 * u1)) - (sqrt(1. + u1 * u1) - 1.) / u1;
  phi2 = log(u2 + sqrt(1. + u2 * u2)) - (sqrt(1. + u2 * u2) - 1.) / u2;
  const double avej_D = gd_D * (phi1 - phi2) / (-kirr * dzt);
  const double nupt = P->nupt0 * bct;
  const double nupt_D = P->nupt0_D * bct;
  double nfixout = 0.0, expoout = 0.0, expo_phosout = 0.0, rn15expoout = 0.0, rc13expoout = 0.0, calproout = 0.0;
  double expofeout = 0.0, remifeout = 0.0;
  const double dtbio = P->dtbio, redctn = P->redctn, redptn = P->redptn, gamma1 = P->gamma1, geZ = P->geZ;
  const double dfr = P->dfr, dfrt = P->dfrt, pfr = P->pfr, rnd = P->redntp / P->diazntp; /* (redntp/diazntp) */
  const double nr_excr_P = 0.0, nr_excr_detr = 0.0;
  const double rn15hi = 2. * RN15STD / (1 + RN15STD), rn15lo = RN15STD / (1 + RN15STD) / 2.;
  const double rc13hi = 2. * RC13STD / (1 + RC13STD), rc13lo = 0.5 * RC13STD / (1 + RC13STD);

  for (int n = 1; n <= P->nbio; ++n) { /* mobi.F:2148-3252 */
    p1 = dmin(biophyt, P->pmax);
    p2 = dmax(0.0, biophyt - P->pmax);
    const double k1n = (P->knmin * p1 + P->knmax * p2) / (p1 + p2);
    const double k1p_P = k1n * ptn_P;
    kfevar = (P->kfemin * p1 + P->kfemax * p2) / (p1 + p2);
    deffe = biodfe / (kfevar + biodfe);
    jmax = P->abio_P * bct * deffe;
    deffe_D = biodfe / (P->kfe_D + biodfe);
    jmax_D = dmax(0., P->abio_P * (bct - P->dbct_D) * deffe_D) * P->jdiar;
    const double limP_dop = P->hdop * biodop / (k1p_P + biodop);
    const double limP_po4 = biopo4 / (k1p_P + biopo4);
    const double dopupt_flag = flag01(limP_dop - limP_po4);
    const double limP = limP_dop * dopupt_flag + limP_po4 * (1. - dopupt_flag);
    double u_P = dmin(avej, jmax * limP);
    u_P = dmin(u_P, jmax * biono3 / (k1n + biono3));
    const double u_D = dmin(avej_D, jmax_D * limP);
    const double dopupt_D_flag = dopupt_flag;
    const double thetaZ = P->zprefP * biophyt + P->zprefDet * biodetr + P->zprefZ * biozoop + P->zprefDiaz * biodiaz + P->kzoo;
    const double ing_P = P->zprefP / thetaZ, ing_Det = P->zprefDet / thetaZ, ing_Z = P->zprefZ / thetaZ;
    const double ing_D = P->zprefDiaz / thetaZ;
    double npp = u_P * biophyt;
    const double dopupt = npp * dopupt_flag; /* NB: from the unflagged npp, mobi.F:2236 */
    double npp_D = dmax(0., u_D * biodiaz);
    const double g_D = gmax * ing_D * biodiaz;
    double graz_D = g_D * biozoop;
    double morpt_D = nupt_D * biodiaz;
    double morp_D = P->nup_D * biodiaz * biodiaz;
    double no3upt_D = (0.5 + 0.5 * tanh(biono3 - 5.)) * npp_D;
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
    double remife = nud * bct * biodetrfe;
    /* iron scavenging, mobi.F:2313-2342 */
    const double o2flag = tanh(dmax(o2, 0.));
    const double ligand = dmax(pow(dmax(aou, 40.), 0.8) / 66. + pow(biodon, 0.8) / 4.8, 0.5) / 1000.;
    const double fepa = (1.0 + P->kfeleq * (ligand - biodfe)) * o2flag;
    const double feprime = ((-fepa + sqrt(fepa * fepa + 4.0 * P->kfeleq * biodfe)) / (2.0 * P->kfeleq)) * o2flag;
    double feorgads = (P->kfeorg * (pow((biodetr * detrflag) * P->mc * redctn, 0.58)) * feprime) * o2flag;
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
    npp_D = npp_D * (dopupt_D_flag * dopflag + (1. - dopupt_D_flag) * po4flag) * din15flag;
    graz_D = graz_D * diazflag * diazn15flag;
    morpt_D = morpt_D * diazflag * diazn15flag;
    morp_D = morp_D * diazflag * diazn15flag;
    no3upt_D = no3upt_D * no3flag * din15flag;
    recy_don = recy_don * donflag * don15flag;
    remife = remife * detrfeflag;
    feorgads = feorgads * dfeflag;
    expofe = expofe * detrfeflag;
    fecol = fecol * dfeflag;
    /* zooplankton budget, mobi.F:2446-2530 */
    const double dig_P = gamma1 * graz, dig_Z = gamma1 * graz_Z, dig_Det = gamma1 * graz_Det;
    double dig = dig_Z + dig_P + dig_Det;
    const double excr_P = gamma1 * (1 - geZ) * graz, excr_Z = gamma1 * (1 - geZ) * graz_Z;
    const double excr_Det = gamma1 * (1 - geZ) * graz_Det;
    double excr = excr_Z + excr_P + excr_Det;
    const double sf_P = (1. - gamma1) * graz, sf_Z = (1. - gamma1) * graz_Z, sf_Det = (1. - gamma1) * graz_Det;
    double sf = sf_P + sf_Z + sf_Det;
    const double sf_P_phos = (graz * ptn_P - dig_P * redptn);
    const double sf_Det_phos = (graz_Det * ptn_detr - dig_Det * redptn);
    double sf_phos = sf_P_phos + sf_Z * redptn + sf_Det_phos;
    const double dig_D = gamma1 * graz_D * rnd;
    dig = dig + dig_D;
    const double excr_D = gamma1 * (1 - geZ) * graz_D * rnd;
    excr = excr + excr_D;
    const double nr_excr_D = gamma1 * graz_D * (1 - rnd) + (1 - gamma1) * graz_D * (1 - rnd);
    const double sf_D = (1 - gamma1) * graz_D * rnd;
    sf = sf + sf_D;
    sf_phos = sf_phos + sf_D * redptn;
    /* nitrogen-15 fractionation, mobi.F:2589-2650 */
    double uno3 = npp * dtbio / biono3;
    uno3 = dmin(uno3, 0.999);
    uno3 = dmax(uno3, TRCMIN);
    double rno3 = clamp_ratio(biodin15 / (biono3 - biodin15), 2 * RN15STD, RN15STD / 2.);
    const double bassim = rayleigh(rno3, P->eps_assim, uno3);
    const double fcassim = bassim / (1 + bassim);
    double udon = recy_don * dtbio / biodon;
    udon = dmin(udon, 0.999);
    udon = dmax(udon, TRCMIN);
    const double rdon = clamp_ratio(biodon15 / (biodon - biodon15), 2 * RN15STD, RN15STD / 2.);
    const double brecy = rayleigh(rdon, P->eps_recy, udon);
    const double fcrecy = brecy / (1 + brecy);
    const double rzoop = clamp_ratio(biozoopn15 / (biozoop - biozoopn15), 2. * RN15STD, RN15STD / 2.);
    const double bexcr = rzoop - P->eps_excr * rzoop / 1000.;
    const double fcexcr = bexcr / (1 + bexcr);
    const double bnfix = RN15STD - P->eps_nfix * RN15STD / 1000.;
    const double fcnfix = bnfix / (1 + bnfix);
    const double rtphytn15 = clamp_ratio(biophytn15 / biophyt, rn15hi, rn15lo);
    const double rtzoopn15 = clamp_ratio(biozoopn15 / biozoop, rn15hi, rn15lo);
    const double rtdetrn15 = clamp_ratio(biodetrn15 / biodetr, rn15hi, rn15lo);
    const double rtdiazn15 = clamp_ratio(biodiazn15 / biodiaz, rn15hi, rn15lo);
    /* carbon-13 fractionation, mobi.F:2651-2695 */
    const double rdic13 = clamp_ratio(biodic13 / (biodic - biodic13), 2. * RC13STD, 0.5 * RC13STD);
    const double bc13npp = ac13b * rdic13;
    const double fcnpp = bc13npp / (1 + bc13npp);
    const double rtphytc13 = clamp_ratio(biophytc13 / (biophyt * redctn), rc13hi, rc13lo);
    const double rtzoopc13 = clamp_ratio(biozoopc13 / (biozoop * redctn), rc13hi, rc13lo);
    const double rtdetrc13 = clamp_ratio(biodetrc13 / (biodetr * redctn), rc13hi, rc13lo);
    const double rtdoc13 = clamp_ratio(biodoc13 / (biodon * redctn), rc13hi, rc13lo);
    const double rtdiazc13 = clamp_ratio(biodiazc13 / (biodiaz * redctn), rc13hi, rc13lo);
    const double calpro = (morp + morz + (graz + graz_Z) * (1. - gamma1)) * P->capr * redctn * 1.e3;
    /* variable P:C of new production (Galbraith & Martiny 2015), mobi.F:2721-2724 */
    const double GM15ptc = 0.0060 + 0.0069 * biopo4;
    const double GM15ptn = GM15ptc * redctn * 1.e3;
    const double diazptn = P->diazptn, rfeton = P->rfeton;
    /* prognostic updates, mobi.F:2738-3085; every right-hand side uses the OLD state */
    const double n_po4 = biopo4 + dtbio * (dopupt * ptn_P - GM15ptn * npp + (1. - dfrt) * morpt * ptn_P +
                                           (1. - pfr) * remi * ptn_detr + diazptn * (morpt_D - (npp_D - dopupt_D)) +
                                           recy_dop + redptn * (excr));
    const double n_dop = biodop + dtbio * (dfr * morp * ptn_P + dfrt * morpt * ptn_P + pfr * remi * ptn_detr -
                                           ptn_P * dopupt - diazptn * dopupt_D - recy_dop);
    const double n_phyt = biophyt + dtbio * (npp - morp - graz - morpt);
    const double n_phyt_phos = biophyt_phos + dtbio * (npp * GM15ptn - morp * ptn_P - graz * ptn_P - morpt * ptn_P);
    const double n_zoop = biozoop + dtbio * (dig - morz - graz_Z - excr);
    const double n_detr = biodetr + dtbio * ((1. - dfr) * morp + sf + morz - remi - graz_Det - expo + impo + morp_D * rnd);
    const double n_detr_phos = biodetr_phos + dtbio * ((1. - dfr) * morp * ptn_P + sf_phos + morz * redptn - remi * ptn_detr -
                                                       graz_Det * ptn_detr - expo_phos + impo_phos + morp_D * rnd * redptn);
    const double n_dic = biodic + dtbio * redctn * (excr + (1. - pfr) * remi + (1. - dfrt) * morpt - npp + morpt_D - npp_D +
                                                    recy_don + nr_excr_D + nr_excr_P + nr_excr_detr + morp_D * (1. - rnd));
    const double n_no3 = biono3 + dtbio * (excr + (1. - pfr) * remi + (1. - dfrt) * morpt - npp + morpt_D - no3upt_D +
                                           recy_don + nr_excr_D + nr_excr_P + nr_excr_detr + morp_D * (1. - rnd));
    const double n_don = biodon + dtbio * (dfr * morp + dfrt * morpt + pfr * remi - recy_don);
    const double n_diaz = biodiaz + dtbio * (npp_D - morp_D - morpt_D - graz_D);
    /* the P:N ratios are refreshed HERE, from the new phyt/detr (mobi.F:2846-2849), and the
       remaining updates still use the old pools but see no ptn_* */
    const double n_dfe = biodfe + dtbio * (rfeton * (excr + (1. - dfrt) * morpt - npp + morpt_D - npp_D + recy_don +
                                                     nr_excr_D + nr_excr_P + nr_excr_detr + morp_D * (1. - rnd)) -
                                           feorgads + remife - fecol);
    const double n_detrfe = biodetrfe + dtbio * (rfeton * (sf + (1. - dfr) * morp + morp_D * rnd + morz - graz_Det) +
                                                 feorgads + P->iscr * fecol - remife - expofe + impofe);
    const double n_din15 =
        biodin15 + dtbio * (rtphytn15 * (1. - dfrt) * morpt + rtphytn15 * nr_excr_P + fcexcr * excr + rtdiazn15 * morpt_D +
                            rtdiazn15 * nr_excr_D + rtdiazn15 * morp_D * (1. - rnd) + rtdetrn15 * (1. - pfr) * remi +
                            rtdetrn15 * nr_excr_detr + fcrecy * recy_don - fcassim * npp - fcassim * no3upt_D);
    const double n_don15 =
        biodon15 + dtbio * (dfr * rtphytn15 * morp + dfrt * rtphytn15 * morpt + rtdetrn15 * pfr * remi - fcrecy * recy_don);
    const double n_phytn15 = biophytn15 + dtbio * (fcassim * npp - rtphytn15 * morp - rtphytn15 * graz - rtphytn15 * morpt);
    const double n_zoopn15 = biozoopn15 + dtbio * (rtphytn15 * dig_P + rtzoopn15 * dig_Z + rtdetrn15 * dig_Det +
                                                   rtdiazn15 * dig_D - rtzoopn15 * morz - rtzoopn15 * graz_Z - fcexcr * excr);
    const double n_detrn15 =
        biodetrn15 + dtbio * (rtphytn15 * (1. - dfr) * morp + rtphytn15 * sf_P + rtzoopn15 * sf_Z + rtdetrn15 * sf_Det +
                              rtdiazn15 * sf_D + rtzoopn15 * morz - rtdetrn15 * remi - rtdetrn15 * graz_Det -
                              rtdetrn15 * expo + rn15impo * impo + rtdiazn15 * morp_D * rnd);
    const double n_diazn15 = biodiazn15 + dtbio * (fcnfix * (npp_D - no3upt_D) + fcassim * no3upt_D - rtdiazn15 * morp_D -
                                                   rtdiazn15 * graz_D - rtdiazn15 * morpt_D);
    const double n_dic13 =
        biodic13 + dtbio * redctn * (rtphytc13 * (1. - dfrt) * morpt + rtphytc13 * nr_excr_P + rtzoopc13 * excr +
                                     rtdiazc13 * morpt_D + rtdiazc13 * nr_excr_D + rtdiazc13 * morp_D * (1 - rnd) +
                                     rtdetrc13 * (1. - pfr) * remi + rtdetrc13 * nr_excr_detr + rtdoc13 * recy_don -
                                     fcnpp * npp - fcnpp * npp_D);
    const double n_doc13 = biodoc13 + dtbio * redctn * (dfr * rtphytc13 * morp + rtphytc13 * dfrt * morpt +
                                                        rtdetrc13 * pfr * remi - rtdoc13 * recy_don);
    const double n_phytc13 =
        biophytc13 + dtbio * redctn * (fcnpp * npp - rtphytc13 * morp - rtphytc13 * graz - rtphytc13 * morpt);
    const double n_zoopc13 =
        biozoopc13 + dtbio * redctn * (rtphytc13 * dig_P + rtzoopc13 * dig_Z + rtdetrc13 * dig_Det + rtdiazc13 * dig_D -
                                       rtzoopc13 * morz - rtzoopc13 * graz_Z - rtzoopc13 * excr);
    const double n_detrc13 =
        biodetrc13 + dtbio * redctn * (rtphytc13 * (1. - dfr) * morp + rtphytc13 * sf_P + rtzoopc13 * sf_Z + rtdetrc13 * sf_Det +
                                       rtdiazc13 * sf_D + rtzoopc13 * morz - rtdetrc13 * remi - rtdetrc13 * graz_Det -
                                       rtdetrc13 * expo + rc13impo + rtdiazc13 * morp_D * rnd);
    const double n_diazc13 = biodiazc13 + dtbio * redctn * (fcnpp * npp_D - rtdiazc13 * (morp_D + graz_D + morpt_D));
    biopo4 = n_po4; biodop = n_dop; biophyt = n_phyt; biophyt_phos = n_phyt_phos; biozoop = n_zoop; biodetr = n_detr;
    biodetr_phos = n_detr_phos; biodic = n_dic; biono3 = n_no3; biodon = n_don; biodiaz = n_diaz;
    ptn_P = biophyt_phos / biophyt;
    ptn_detr = biodetr_phos / biodetr;
    biodfe = n_dfe; biodetrfe = n_detrfe; biodin15 = n_din15; biodon15 = n_don15; biophytn15 = n_phytn15;
    biozoopn15 = n_zoopn15; biodetrn15 = n_detrn15; biodiazn15 = n_diazn15; biodic13 = n_dic13; biodoc13 = n_doc13;
    biophytc13 = n_phytc13; biozoopc13 = n_zoopc13; biodetrc13 = n_detrc13; biodiazc13 = n_diazc13;
    /* accumulate, mobi.F:3088-3172 */
    expoout = expoout + expo;
    expo_phosout = expo_phosout + expo_phos;
    rn15expoout = rn15expoout + rtdetrn15;
    rc13expoout = rc13expoout + rtdetrc13 * expo;
    calproout = calproout + calpro;
    nfixout = nfixout + npp_D - no3upt_D;
    expofeout = expofeout + expofe;
    remifeout = remifeout + remife;
    /* refresh flags that are still set, mobi.F:3175-3251 */
    if (po4flag == 1) po4flag = flag01(biopo4 - TRCMIN);
    if (phytflag == 1) phytflag = flag01(biophyt - TRCMIN);
    if (zoopflag == 1) zoopflag = flag01(biozoop - TRCMIN);
    if (detrflag == 1) detrflag = flag01(biodetr - TRCMIN);
    if (phyt_phosflag == 1) phyt_phosflag = flag01(biophyt_phos - TRCMIN);
    if (detr_phosflag == 1) detr_phosflag = flag01(biodetr_phos - TRCMIN);
    if (no3flag == 1) no3flag = flag01(biono3 - TRCMIN);
    if (dopflag == 1) dopflag = flag01(biodop - TRCMIN);
    if (donflag == 1) donflag = flag01(biodon - TRCMIN);
    if (diazflag == 1) diazflag = flag01(biodiaz - TRCMIN);
    if (din15flag == 1) din15flag = flag01(biodin15 - TRCMIN);
    if (don15flag == 1) don15flag = flag01(biodon15 - TRCMIN);
    if (phytn15flag == 1) phytn15flag = flag01(biophytn15 - TRCMIN);
    if (zoopn15flag == 1) zoopn15flag = flag01(biozoopn15 - TRCMIN);
    if (detrn15flag == 1) detrn15flag = flag01(biodetrn15 - TRCMIN);
    if (diazn15flag == 1) diazn15flag = flag01(biodiazn15 - TRCMIN);
    if (dfeflag == 1) dfeflag = flag01(biodfe - TRCMIN);
    if (detrfeflag == 1) detrfeflag = flag01(biodetrfe - TRCMIN);
    if (dic13flag == 1) dic13flag = flag01(biodic13 - TRCMIN);
    if (phytc13flag == 1) phytc13flag = flag01(biophytc13 - TRCMIN);
    if (zoopc13flag == 1) zoopc13flag = flag01(biozoopc13 - TRCMIN);
    if (detrc13flag == 1) detrc13flag = flag01(biodetrc13 - TRCMIN);
    if (doc13flag == 1) doc13flag = flag01(biodoc13 - TRCMIN);
    if (diazc13flag == 1) diazc13flag = flag01(biodiazc13 - TRCMIN);
  }
  (void)dic13flag; (void)doc13flag; (void)phytc13flag; (void)zoopc13flag; (void)detrc13flag; (void)diazc13flag;
#define BOUT(m, v) bioout[(m)-1] = (v)-BIN(m)
  BOUT(I->po4, biopo4); BOUT(I->phyt, biophyt); BOUT(I->phyt_phos, biophyt_phos); BOUT(I->zoop, biozoop);
  BOUT(I->detr, biodetr); BOUT(I->detr_phos, biodetr_phos); BOUT(I->dic, biodic); BOUT(I->dop, biodop);
  BOUT(I->no3, biono3); BOUT(I->don, biodon); BOUT(I->diaz, biodiaz); BOUT(I->din15, biodin15);
  BOUT(I->don15, biodon15); BOUT(I->phytn15, biophytn15); BOUT(I->zoopn15, biozoopn15); BOUT(I->detrn15, biodetrn15);
  BOUT(I->diazn15, biodiazn15); BOUT(I->dfe, biodfe); BOUT(I->detrfe, biodetrfe); BOUT(I->dic13, biodic13);
  BOUT(I->phytc13, biophytc13); BOUT(I->zoopc13, biozoopc13); BOUT(I->detrc13, biodetrc13); BOUT(I->doc13, biodoc13);
  BOUT(I->diazc13, biodiazc13);
  out->expo = expoout; out->expo_phos = expo_phosout; out->calpro = calproout; out->nfix = nfixout;
  out->rn15expo = rn15expoout; out->rc13expo = rc13expoout; out->expofe = expofeout; out->remife = remifeout;
#undef BIN
#undef BOUT
}

/* ------------------------------------------------------------------------- */
/* mobi_driver, mobi.F:519-1482                                               */
/* ------------------------------------------------------------------------- */
void orc_mobi_driver(const orc_mobi *P, int kmx, double twodt, double rctheta, double dayfrac, double swr, double *tnpzd,
                     const double *t_in, const double *o2_in, const double *aou_in, const double *s_in, const double *dic_in,
                     const double *alk_in, double co2_in, const double *sgb_in, double *src) {
  const int km = P->km;
  const orc_mobi_index *I = &P->im;
  const orc_mobi_index *S = &P->is;
#define TN(k, m) tnpzd[((k)-1) + (size_t)km * ((m)-1)]
#define SRC(k, s) src[((k)-1) + (size_t)km * ((s)-1)]
  double expo = 0.0, impo, expo_phos = 0.0, impo_phos, phin = 0.0, prca = 0.0;
  double rn15impo, rn15expo = 0.0, rc13impo, rc13expo = 0.0, prca13 = 0.0, expofe = 0.0, impofe;
  double snpzd[ORC_MOBI_MAXT], bioin[ORC_MOBI_MAXT];
  double rcalpro[ORC_MOBI_MAXK], bdeni[ORC_MOBI_MAXK], nfix[ORC_MOBI_MAXK], dic_npzd_sms[ORC_MOBI_MAXK];
  for (size_t q = 0; q < (size_t)km * P->nsrc; ++q) src[q] = 0.0;
  for (int k = 0; k < km; ++k) rcalpro[k] = bdeni[k] = nfix[k] = dic_npzd_sms[k] = 0.0;
  const double redctn = P->redctn;
  for (int k = 1; k <= kmx; ++k) {
    rn15impo = rn15expo;
    const double atmpres = 1.0, depth = P->zt[k - 1] / 100.;
    double pH, co2star, dco2star, pCO2, dpco2, CO3, Omega_c, Omega_a;
    orc_co2calc_SWS(t_in[k - 1], s_in[k - 1], dic_in[k - 1], alk_in[k - 1], co2_in, atmpres, depth, &pH, &co2star, &dco2star,
                    &pCO2, &dpco2, &CO3, &Omega_c, &Omega_a);
    const double ac13_DIC_aq = -1.0512994e-4 * t_in[k - 1] + 1.011765;
    const double ac13_aq_POC = -0.017 * log10(dmin(dmax(co2star * 1000., 2.), 74.)) + 1.0034;
    const double ac13b = ac13_aq_POC / ac13_DIC_aq;
    rc13impo = rc13expo * P->dztr[k - 1];
    swr = swr * exp(-P->kc * phin);
    phin = dmax(TN(k, I->phyt), TRCMIN) * P->dzt[k - 1] + dmax(TN(k, I->diaz), TRCMIN) * P->dzt[k - 1];
    const double gl = swr * exp(P->ztt[k - 1] * rctheta);
    impo = expo * P->dztr[k - 1];
    impo_phos = expo_phos * P->dztr[k - 1];
    impofe = expofe * P->dztr[k - 1];
    const double bct = pow(P->bbio, P->cbio * t_in[k - 1]);
    const double bctz = (0.5 * (tanh(o2_in[k - 1] - 8.) + 1)) * pow(P->bbio, P->cbio * t_in[k - 1]);
    const double nud = P->nud0 * (0.6 + 0.4 * tanh(0.22 * dmax(o2_in[k - 1], 0.)));
    const double nudon = P->nudon0, nudop = P->nudop0;
    /* tnpzd(k,:) is passed as a strided section: copy in, clamp, copy out (mobi.F:853, 1894) */
    for (int m = 1; m <= P->ntnpzd; ++m) bioin[m - 1] = TN(k, m);
    src_out_t so;
    mobi_src(P, bioin, gl, bct, impo, P->dzt[k - 1], impo_phos, dayfrac, P->wd[k - 1], nud, nudop, nudon, snpzd, bctz, rn15impo,
             rc13impo, ac13b, impofe, o2_in[k - 1], aou_in[k - 1], &so);
    for (int m = 1; m <= P->ntnpzd; ++m) TN(k, m) = bioin[m - 1];
    expo = so.expo; expo_phos = so.expo_phos; rn15expo = so.rn15expo; rc13expo = so.rc13expo; expofe = so.expofe;
    nfix[k - 1] = so.nfix;
    for (int m = 0; m < P->ntnpzd; ++m) snpzd[m] = snpzd[m] * P->rdtts;
    expofe = expofe * P->rnbio;
    expo = expo * P->rnbio;
    expo_phos = expo_phos * P->rnbio;
    rn15expo = rn15expo * P->rnbio;
    rc13expo = rc13expo * P->rnbio;
    rcalpro[k - 1] = so.calpro * P->rnbio;
    const double sgb = sgb_in[k - 1];
    /* benthic denitrification on the sub-grid bathymetry, mobi.F:1033-1085 */
    const double no3flag = flag01(TN(k, I->no3) - TRCMIN);
    const double din15flag = flag01(TN(k, I->din15) - TRCMIN);
    const double lno3 = 0.5 * tanh(TN(k, I->no3) * 10 - 5.0);
    double sg_bdeni = (0.06 + 0.19 * pow(0.99, dmax(o2_in[k - 1], TRCMIN) - dmax(TN(k, I->no3), TRCMIN))) *
                      dmax(expo * sgb, TRCMIN) * redctn * 1.e3;
    sg_bdeni = dmin(sg_bdeni, sgb * expo);
    sg_bdeni = dmax(sg_bdeni, 0.);
    sg_bdeni = sg_bdeni * (0.5 + lno3) * no3flag * din15flag;
    bdeni[k - 1] = sg_bdeni;
#define SN(m) snpzd[(m)-1]
    SN(I->no3) = SN(I->no3) + sgb * expo - sg_bdeni;
    const double r15min = TRCMIN * RN15STD / (1 + RN15STD);
    double rno3 = dmax(TN(k, I->din15), r15min) / dmax(TN(k, I->no3) - TN(k, I->din15), r15min);
    rno3 = dmin(rno3, 2. * RN15STD);
    rno3 = dmax(rno3, RN15STD / 2.);
    const double eps_bdeni = P->eps_bdeni0 * exp(-2.5e-6 * (P->zt[k - 1]));
    const double bbdeni = rno3 - eps_bdeni * rno3 / 1000.;
    SN(I->din15) = SN(I->din15) + rn15expo * sgb * expo - bbdeni / (1 + bbdeni) * sg_bdeni;
    /* sedimentary iron release, mobi.F:1086-1123 */
    const double coxdepth = dmin(dmax(P->zt[k - 1], 50000.), 150000.);
    const double oblinc = -1.26e-6 * coxdepth + 0.203;
    const double obexpc = -6.e-7 * coxdepth + 1.14;
    const double dztk = P->dzt[k - 1];
    const double nburial = (oblinc * pow(expo * sgb * dztk / 100 * 86400. * 365. * redctn * 1000., obexpc)) /
                           (86400. * 365. * dztk / 100 * redctn * 1000.);
    const double coxsed = expo * sgb - nburial;
    const double fesedmax = 85.;
    const double fesed = fesedmax * tanh(coxsed * redctn * 1000 * dztk / 100 * 86400. / o2_in[k - 1]) / (dztk / 100 * 86400 * 1000);
    SN(I->dfe) = SN(I->dfe) + fesed;
    /* bottom remineralisation, mobi.F:1124-1134 */
    SN(I->po4) = SN(I->po4) + sgb * expo_phos;
    SN(I->dic) = SN(I->dic) + sgb * expo * redctn;
    SN(I->dic13) = SN(I->dic13) + rc13expo * sgb * redctn;
    rc13expo = rc13expo - sgb * rc13expo;
    expo = expo - sgb * expo;
    expo_phos = expo_phos - sgb * expo_phos;
    /* scatter into the source slots, mobi.F:1149-1205 */
    SRC(k, S->po4) = SN(I->po4); SRC(k, S->phyt) = SN(I->phyt); SRC(k, S->zoop) = SN(I->zoop); SRC(k, S->detr) = SN(I->detr);
    SRC(k, S->phyt_phos) = SN(I->phyt_phos); SRC(k, S->detr_phos) = SN(I->detr_phos); SRC(k, S->dic) = SN(I->dic);
    SRC(k, S->dop) = SN(I->dop); SRC(k, S->no3) = SN(I->no3); SRC(k, S->don) = SN(I->don); SRC(k, S->diaz) = SN(I->diaz);
    SRC(k, S->din15) = SN(I->din15); SRC(k, S->don15) = SN(I->don15); SRC(k, S->phytn15) = SN(I->phytn15);
    SRC(k, S->zoopn15) = SN(I->zoopn15); SRC(k, S->detrn15) = SN(I->detrn15); SRC(k, S->diazn15) = SN(I->diazn15);
    SRC(k, S->dic13) = SN(I->dic13); SRC(k, S->phytc13) = SN(I->phytc13); SRC(k, S->zoopc13) = SN(I->zoopc13);
    SRC(k, S->detrc13) = SN(I->detrc13); SRC(k, S->doc13) = SN(I->doc13); SRC(k, S->diazc13) = SN(I->diazc13);
    SRC(k, S->dfe) = SN(I->dfe); SRC(k, S->detrfe) = SN(I->detrfe);
    /* DIC / alkalinity / 13C bookkeeping, mobi.F:1228-1266 */
    dic_npzd_sms[k - 1] = SN(I->dic);
    const double dprca = rcalpro[k - 1] * 1e-3;
    prca = prca + dprca * dztk;
    SRC(k, S->dic) = SN(I->dic) - dprca;
    const double r13min = TRCMIN * RC13STD / (1 + RC13STD);
    double rtdic13 = dmax(TN(k, I->dic13), r13min) / dmax(dic_in[k - 1], TRCMIN);
    rtdic13 = dmin(rtdic13, 2. * RC13STD / (1 + RC13STD));
    rtdic13 = dmax(rtdic13, 0.5 * RC13STD / (1 + RC13STD));
    prca13 = prca13 + dprca * dztk * rtdic13;
    SRC(k, S->dic13) = SRC(k, S->dic13) - rtdic13 * dprca;
    SRC(k, S->alk) = -SN(I->dic) * P->redntc * 1.e-3 - 2. * dprca;
    /* export of this level becomes the import of the next, mobi.F:1268-1287 */
    expo = expo * dztk;
    expo_phos = expo_phos * dztk;
    rc13expo = rc13expo * dztk;
    expofe = expofe * dztk;
  }
  /* second pass: oxygen, water-column denitrification, 15N (mobi.F:1302-1365) */
  for (int k = 1; k <= kmx; ++k) {
    const double fo2 = tanh(0.22 * dmax(o2_in[k - 1], 0.));
    const double so2 = dic_npzd_sms[k - 1] * P->redotc + nfix[k - 1] * P->rnbio * 1.25e-3;
    const double no3flag = flag01(TN(k, I->no3) - TRCMIN);
    const double din15flag = flag01(TN(k, I->din15) - TRCMIN);
    const double lno3 = 0.5 * tanh(TN(k, I->no3) - 2.5);
    double wcdeni = 800. * no3flag * so2 * (1.0 - fo2) * (0.5 + lno3) * din15flag;
    wcdeni = dmax(wcdeni, 0.);
    SRC(k, S->no3) = SRC(k, S->no3) - wcdeni;
    double uno3 = wcdeni * twodt / TN(k, I->no3);
    uno3 = dmin(uno3, 0.999);
    uno3 = dmax(uno3, TRCMIN);
    const double r15min = TRCMIN * RN15STD / (1 + RN15STD);
    double rno3 = dmax(TN(k, I->din15), r15min) / dmax(TN(k, I->no3) - TN(k, I->din15), r15min);
    rno3 = dmin(rno3, 2. * RN15STD);
    rno3 = dmax(rno3, RN15STD / 2.);
    const double bwcdeni = rayleigh(rno3, P->eps_wcdeni, uno3);
    SRC(k, S->din15) = SRC(k, S->din15) - (bwcdeni / (1 + bwcdeni)) * wcdeni;
    SRC(k, S->alk) = SRC(k, S->alk) + wcdeni * 1.e-3;
    SRC(k, S->alk) = SRC(k, S->alk) + bdeni[k - 1] * 1.e-3;
    SRC(k, S->alk) = SRC(k, S->alk) - nfix[k - 1] * P->rnbio * 1.e-3;
    SRC(k, S->o2) = -so2 * fo2;
  }
  /* third pass: calcite dissolution profile (mobi.F:1373-1436) */
  for (int k = 1; k <= kmx - 1; ++k) {
    SRC(k, S->dic) = SRC(k, S->dic) + prca * P->rcak[k - 1];
    SRC(k, S->dic13) = SRC(k, S->dic13) + prca13 * P->rcak[k - 1];
    SRC(k, S->alk) = SRC(k, S->alk) + 2. * prca * P->rcak[k - 1];
  }
  SRC(kmx, S->dic) = SRC(kmx, S->dic) + prca * P->rcab[kmx - 1];
  SRC(kmx, S->dic13) = SRC(kmx, S->dic13) + prca13 * P->rcab[kmx - 1];
  SRC(kmx, S->alk) = SRC(kmx, S->alk) + 2. * prca * P->rcab[kmx - 1];
#undef SN
#undef TN
#undef SRC
}

/* ------------------------------------------------------------------------- */
/* the caller side in `tracer`: tracer.F:311-545 and :853-867                  */
/* ------------------------------------------------------------------------- */
void orc_mobi_sources(const orc_mobi *P, const orc_mobi_forcing *F, int imt, int jmt, const int *kmt, const double *t_taum1,
                      int nt, double c2dtts, double *src) {
  const int km = P->km;
  const size_t N3 = (size_t)imt * km * jmt;
  const double pi = F->pi, radian = F->radian;
  /* month of the year for the iron deposition, tracer.F:311-336 */
  const double yrtime = fmod(F->relyr, 1.);
  int mi = 12;
  for (int m = 1; m <= 12; ++m)
    if (yrtime <= m / 12.) { mi = m; break; }
  const double declin = sin((fmod(F->relyr, 1.) - 0.22) * 2. * pi) * 0.4;
#define T4(i, k, j, n) t_taum1[(size_t)((i)-1) + (size_t)imt * ((size_t)((k)-1) + (size_t)km * ((j)-1)) + (size_t)((n)-1) * N3]
#define SRC4(i, k, j, s) src[(size_t)((i)-1) + (size_t)imt * ((size_t)((k)-1) + (size_t)km * ((j)-1)) + (size_t)((s)-1) * N3]
  extern int orc_threads; /* uvic_oracle.c: > 1 for the N-core timing of bench.py only (columns are independent) */
#pragma omp parallel for if (orc_threads > 1) num_threads(orc_threads) schedule(dynamic)
  for (int j = 2; j <= jmt - 1; ++j)
    for (int i = 2; i <= imt - 1; ++i) {
      double tnpzd[ORC_MOBI_MAXK * ORC_MOBI_MAXT], col_src[ORC_MOBI_MAXK * ORC_MOBI_MAXT];
      double t_in[ORC_MOBI_MAXK], o2_in[ORC_MOBI_MAXK], aou_in[ORC_MOBI_MAXK], s_in[ORC_MOBI_MAXK], dic_in[ORC_MOBI_MAXK];
      double alk_in[ORC_MOBI_MAXK], sgb_in[ORC_MOBI_MAXK];
      const int kmx = kmt[(i - 1) + (size_t)imt * (j - 1)];
      if (kmx <= 0) continue;
      const size_t ij = (size_t)(i - 1) + (size_t)imt * (j - 1);
      const double ai = F->aice[ij], hi = F->hice[ij], hs = F->hsno[ij];
      double rctheta = dmax(-1.5, dmin(1.5, F->tlat[ij] / radian - declin));
      rctheta = P->kw / sqrt(1. - (1. - sq(cos(rctheta))) / sq(1.33));
      double dayfrac = dmin(1., -tan(F->tlat[ij] / radian) * tan(declin));
      dayfrac = dmax(1e-12, acos(dmax(-1., dayfrac)) / pi);
      const double swr = P->tap * F->dnswr[ij] * 1e-3 * (1. + ai * (exp(-P->ki * (hi + hs)) - 1.));
      for (int m = 1; m <= P->ntnpzd; ++m)
        for (int k = 1; k <= km; ++k) tnpzd[(k - 1) + km * (m - 1)] = T4(i, k, j, P->tracer_of_mobi[m - 1]);
      for (int k = 1; k <= km; ++k) {
        t_in[k - 1] = T4(i, k, j, P->itemp);
        o2_in[k - 1] = T4(i, k, j, P->io2) * 1000.;
        s_in[k - 1] = 1.e3 * T4(i, k, j, P->isalt) + 35.0;
        dic_in[k - 1] = T4(i, k, j, P->idic);
        alk_in[k - 1] = T4(i, k, j, P->ialk);
        sgb_in[k - 1] = F->sg_bathy[ij + (size_t)imt * jmt * (k - 1)];
        aou_in[k - 1] = 0.0;
      }
      for (int k = 1; k <= kmx; ++k) { /* oxygen saturation, tracer.F:456-476 */
        const double f1 = log((298.15 - t_in[k - 1]) / (273.15 + t_in[k - 1]));
        const double f2 = f1 * f1, f3 = f2 * f1, f4 = f3 * f1, f5 = f4 * f1;
        double o2sat = exp(2.00907 + 3.22014 * f1 + 4.05010 * f2 + 4.94457 * f3 - 2.56847E-1 * f4 + 3.88767 * f5 +
                           s_in[k - 1] * (-6.24523e-3 - 7.37614e-3 * f1 - 1.03410e-2 * f2 - 8.17083E-3 * f3) -
                           4.88682E-7 * s_in[k - 1] * s_in[k - 1]);
        o2sat = o2sat / 22391.6 * 1000.0 * 1000.;
        aou_in[k - 1] = o2sat - o2_in[k - 1];
      }
      orc_mobi_driver(P, kmx, c2dtts, rctheta, dayfrac, swr, tnpzd, t_in, o2_in, aou_in, s_in, dic_in, alk_in, F->co2ccn,
                      sgb_in, col_src);
      for (int s = 1; s <= P->nsrc; ++s)
        for (int k = 1; k <= km; ++k) SRC4(i, k, j, s) = col_src[(k - 1) + km * (s - 1)];
      /* iron inputs, tracer.F:538-545 */
      SRC4(i, 1, j, P->is.dfe) = SRC4(i, 1, j, P->is.dfe) + F->fe_atmdep[ij + (size_t)imt * jmt * (mi - 1)] * 1000 / (P->dzt[0] / 100.);
      for (int k = 1; k <= kmx; ++k) SRC4(i, k, j, P->is.dfe) = SRC4(i, k, j, P->is.dfe) + F->fe_hydr[ij + (size_t)imt * jmt * (k - 1)];
    }
  /* carbon-14, tracer.F:853-867 */
  if (P->is.c14 > 0)
    for (int j = 2; j <= jmt - 1; ++j)
      for (int i = 2; i <= imt - 1; ++i) {
        const int kmx = kmt[(i - 1) + (size_t)imt * (j - 1)];
        for (int k = 1; k <= kmx; ++k)
          SRC4(i, k, j, P->is.c14) = SRC4(i, k, j, P->is.dic) * RC14STD - 3.836e-12 * T4(i, k, j, P->ic14);
      }
  (void)nt;
#undef T4
#undef SRC4
}
