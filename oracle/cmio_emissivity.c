/*
 * cmio_emissivity.c - ORACLE (test infrastructure): line and continuum
 * emissivities of a cell from its converged state - EmissivityCalculator
 * (src/EmissivityCalculator.cpp:42-116 tables and Balmer jump, :126-430
 * calculate_emissivities) on top of LineCoolingData::get_line_strengths
 * (cmio_line_strengths in cmio_linecooling.c).
 *
 * The reference writes the 42 emissivities out one by one; here the forbidden
 * and fine-structure lines are a TABLE - emission line <- sum of (ion,
 * transition) line strengths - and only the recombination lines and averages
 * are formulas. Pinned by the reference's bjump_testdata.txt
 * (test/testEmissivityCalculator.cpp:50-86) and linestr_testdata.txt
 * (test/testLineCoolingData.cpp:151-330); its third fixture,
 * hiilines_testdata.txt, is not in the reference tree.
 */
#include "cmio_internal.h"

#include <math.h>

/* ions of the line-cooling data, src/LineCoolingData.hpp */
enum { NI = 0, NII, OI, OII, OIII, NeIII, SII, SIII, CII, CIII, NIII, NeII,
       SIV };
/* transitions of a five-level ion, src/LineCoolingData.hpp:87-114 */
enum { T01 = 0, T02, T03, T04, T12, T13, T14, T23, T24, T34 };
#define FIVE(ion, t) (10 * (ion) + (t))
#define TWO(ion) (100 + (ion)-NIII)

/* src/EmissivityValues.hpp:36-81 */
enum {
  EL_HAlpha = 0, EL_HBeta, EL_HII, EL_BALMER_JUMP_LOW, EL_BALMER_JUMP_HIGH,
  EL_OI_6300, EL_OI_6364, EL_OII_3727, EL_OIII_5007, EL_OIII_4959,
  EL_OIII_4363, EL_OIII_52mu, EL_OIII_88mu, EL_NII_5755, EL_NII_6548,
  EL_NII_6584, EL_NeIII_3869, EL_NeIII_3968, EL_SII_6725, EL_SII_4072,
  EL_SIII_9405, EL_SIII_6312, EL_SIII_19mu, EL_SIII_33mu, EL_avg_T,
  EL_avg_T_count, EL_avg_nH_nHe, EL_avg_nH_nHe_count, EL_NeII_12mu,
  EL_NIII_57mu, EL_NeIII_15mu, EL_NII_122mu, EL_CII_158mu, EL_CII_2325,
  EL_CIII_1908, EL_OII_7325, EL_SIV_10mu, EL_HeI_5876, EL_Hrec_s,
  EL_WFC2_F439W, EL_WFC2_F555W, EL_WFC2_F675W, EL_COUNT
};

/* emission line <- line strengths that add up to it
 * (src/EmissivityCalculator.cpp:247-372, :395-423) */
static const struct {
  int line, strength;
} line_terms[] = {
    {EL_NII_5755, FIVE(NII, T34)},    {EL_NII_6548, FIVE(NII, T13)},
    {EL_NII_6584, FIVE(NII, T23)},    {EL_NII_122mu, FIVE(NII, T12)},
    {EL_OI_6300, FIVE(OI, T03)},      {EL_OI_6364, FIVE(OI, T13)},
    {EL_OII_3727, FIVE(OII, T01)},    {EL_OII_3727, FIVE(OII, T02)},
    {EL_OII_7325, FIVE(OII, T14)},    {EL_OII_7325, FIVE(OII, T24)},
    {EL_OII_7325, FIVE(OII, T13)},    {EL_OII_7325, FIVE(OII, T23)},
    {EL_OIII_4363, FIVE(OIII, T34)},  {EL_OIII_4959, FIVE(OIII, T13)},
    {EL_OIII_5007, FIVE(OIII, T23)},  {EL_OIII_52mu, FIVE(OIII, T12)},
    {EL_OIII_88mu, FIVE(OIII, T01)},  {EL_NeIII_3869, FIVE(NeIII, T03)},
    {EL_NeIII_3968, FIVE(NeIII, T13)}, {EL_NeIII_15mu, FIVE(NeIII, T01)},
    {EL_SII_4072, FIVE(SII, T03)},    {EL_SII_4072, FIVE(SII, T04)},
    {EL_SII_6725, FIVE(SII, T01)},    {EL_SII_6725, FIVE(SII, T02)},
    {EL_SIII_9405, FIVE(SIII, T13)},  {EL_SIII_9405, FIVE(SIII, T23)},
    {EL_SIII_6312, FIVE(SIII, T34)},  {EL_SIII_19mu, FIVE(SIII, T12)},
    {EL_SIII_33mu, FIVE(SIII, T01)},  {EL_CII_158mu, FIVE(CII, T01)},
    {EL_CII_2325, FIVE(CII, T02)},    {EL_CII_2325, FIVE(CII, T12)},
    {EL_CII_2325, FIVE(CII, T03)},    {EL_CII_2325, FIVE(CII, T13)},
    {EL_CII_2325, FIVE(CII, T04)},    {EL_CII_2325, FIVE(CII, T14)},
    {EL_CIII_1908, FIVE(CIII, T01)},  {EL_CIII_1908, FIVE(CIII, T02)},
    {EL_CIII_1908, FIVE(CIII, T03)},  {EL_NIII_57mu, TWO(NIII)},
    {EL_NeII_12mu, TWO(NeII)},        {EL_SIV_10mu, TWO(SIV)},
    /* the three WFC2 filters: lines inside the pass bands (H beta / H alpha
     * are added below) */
    {EL_WFC2_F439W, FIVE(OIII, T34)}, {EL_WFC2_F439W, FIVE(SIII, T03)},
    {EL_WFC2_F439W, FIVE(SIII, T04)}, {EL_WFC2_F555W, FIVE(NI, T01)},
    {EL_WFC2_F555W, FIVE(NI, T02)},   {EL_WFC2_F555W, FIVE(NII, T34)},
    {EL_WFC2_F555W, FIVE(OI, T34)},   {EL_WFC2_F555W, FIVE(OIII, T03)},
    {EL_WFC2_F555W, FIVE(OIII, T13)}, {EL_WFC2_F555W, FIVE(OIII, T23)},
    {EL_WFC2_F675W, FIVE(NII, T03)},  {EL_WFC2_F675W, FIVE(NII, T13)},
    {EL_WFC2_F675W, FIVE(NII, T23)},  {EL_WFC2_F675W, FIVE(OI, T03)},
    {EL_WFC2_F675W, FIVE(OI, T13)},   {EL_WFC2_F675W, FIVE(OI, T23)},
    {EL_WFC2_F675W, FIVE(SII, T01)},  {EL_WFC2_F675W, FIVE(SII, T02)},
    {EL_WFC2_F675W, FIVE(SIII, T34)},
};

/* EmissivityCalculator::get_balmer_jump_emission,
 * src/EmissivityCalculator.cpp:42-116: Brown & Mathews (1970) continuum
 * coefficients of H and He above and below the Balmer jump, log-log
 * interpolated in T; out = {H high (3681 A), H low (3643 A), He high, He low}
 * in J m^3 s^-1 angstrom^-1 */
void cmio_balmer_jump(double T, double out[4]) {
  static const double ttab[8] = {4.e3,  6.e3,  8.e3,  1.e4,
                                 1.2e4, 1.4e4, 1.6e4, 1.8e4};
  static const double coefficient[4][8] = {
      {0.162, 0.584, 1.046, 1.437, 1.742, 1.977, 2.159, 2.297},   /* hplt */
      {92.6, 50.9, 33.8, 24.8, 19.53, 16.09, 13.7, 11.96},        /* hmit */
      {0.189, 0.622, 1.076, 1.45, 1.74, 1.963, 2.14, 2.27},       /* heplt */
      {15.7, 9.23, 6.71, 5.49, 4.83, 4.41, 4.135, 3.94}};         /* hemit */
  static const double wavelength[4] = {3681., 3643., 3681., 3643.};
  double logttab[8];
  for (int i = 0; i < 8; ++i)
    logttab[i] = log(ttab[i]);
  const double logt = log(T);
  int i = (int)cmio_locate(logt, logttab, 8);
  i = i < 0 ? 0 : (i > 6 ? 6 : i);
  const double lightspeed = 299792458.;
  for (int k = 0; k < 4; ++k) {
    const double lo = log(coefficient[k][i]), hi = log(coefficient[k][i + 1]);
    const double v =
        exp(lo + (logt - logttab[i]) * (hi - lo) / (logttab[i + 1] - logttab[i]));
    /* 1e-40 erg cm^3 s^-1 Hz^-1 -> J m^3 s^-1 angstrom^-1 */
    out[k] = v * (1.e-43 * lightspeed / (wavelength[k] * wavelength[k]));
  }
}

/* EmissivityCalculator::calculate_emissivities for one cell,
 * src/EmissivityCalculator.cpp:126-430: n total number density (m^-3), T,
 * x[14] the ionic fractions; out[42] in the order of EmissivityValues.hpp
 * (all zero for a cell with x_H >= 0.2 or T <= 3000 K) */
void cmio_emissivities(const cmio_model *model, double ntot, double T,
                       const double *x, double *out) {
  for (int l = 0; l < EL_COUNT; ++l)
    out[l] = 0.;
  if (!(x[CMIO_ION_H_n] < 0.2 && T > 3000.))
    return;
  const double AHe = model->abundance[CMIO_EL_He];
  const double nhp = ntot * (1. - x[CMIO_ION_H_n]);
  const double nhep = ntot * (1. - x[CMIO_ION_He_n]) * AHe;
  const double ne = nhp + nhep;
  /* the ions the line cooling data know, :152-223 (as in the thermal
   * balance) */
  const double AC = model->abundance[CMIO_EL_C], AN = model->abundance[CMIO_EL_N],
               AO = model->abundance[CMIO_EL_O],
               ANe = model->abundance[CMIO_EL_Ne],
               AS = model->abundance[CMIO_EL_S];
  double abund[13];
  abund[CII] = AC * (1. - x[CMIO_ION_C_p1] - x[CMIO_ION_C_p2]);
  abund[CIII] = AC * x[CMIO_ION_C_p1];
  abund[NI] = AN * (1. - x[CMIO_ION_N_n] - x[CMIO_ION_N_p1] - x[CMIO_ION_N_p2]);
  abund[NII] = AN * x[CMIO_ION_N_n];
  abund[NIII] = AN * x[CMIO_ION_N_p1];
  abund[OI] = AO * (1. - x[CMIO_ION_O_n] - x[CMIO_ION_O_p1]);
  abund[OII] = AO * x[CMIO_ION_O_n];
  abund[OIII] = AO * x[CMIO_ION_O_p1];
  abund[NeII] = ANe * x[CMIO_ION_Ne_n];
  abund[NeIII] = ANe * x[CMIO_ION_Ne_p1];
  abund[SII] = AS * (1. - x[CMIO_ION_S_p1] - x[CMIO_ION_S_p2] - x[CMIO_ION_S_p3]);
  abund[SIII] = AS * x[CMIO_ION_S_p1];
  abund[SIV] = AS * x[CMIO_ION_S_p2];

  double strength[103];
  cmio_line_strengths(T, ne, abund, strength);
  for (size_t k = 0; k < sizeof line_terms / sizeof line_terms[0]; ++k)
    out[line_terms[k].line] += strength[line_terms[k].strength];
  for (int l = 0; l < EL_COUNT; ++l)
    out[l] *= ntot;

  /* recombination lines (Osterbrock & Ferland 2006 table 4.1, fits to Storey
   * & Hummer 1995), :229-238, :381-388 */
  const double T4 = T * 1.e-4;
  out[EL_HAlpha] = ne * nhp * 2.87 * 1.24e-38 * pow(T4, -0.938);
  out[EL_HBeta] = ne * nhp * 1.24e-38 * pow(T4, -0.878);
  out[EL_HII] = nhp * ne * 4.9e-40 * pow(T4, -0.848);
  out[EL_HeI_5876] = ne * nhep * 1.69e-38 * pow(T4, -1.065);
  out[EL_Hrec_s] =
      ne * nhp * 7.982e-23 /
      (sqrt(T / 3.148) * pow(1. + sqrt(T / 3.148), 0.252) *
       pow(1. + sqrt(T / 7.036e5), 1.748));
  double jump[4];
  cmio_balmer_jump(T, jump);
  out[EL_BALMER_JUMP_LOW] = ne * (nhp * jump[1] + nhep * jump[3]);
  out[EL_BALMER_JUMP_HIGH] = ne * (nhp * jump[0] + nhep * jump[2]);
  /* weights for emission-weighted averages, :374-380 */
  out[EL_avg_T] = ne * nhp * T;
  out[EL_avg_T_count] = ne * nhp;
  out[EL_avg_nH_nHe] = ne * (1. - x[CMIO_ION_He_n]);
  out[EL_avg_nH_nHe_count] = ne * (1. - x[CMIO_ION_H_n]);
  /* the filters see H beta resp. H alpha too, :400, :410 */
  out[EL_WFC2_F555W] += out[EL_HBeta];
  out[EL_WFC2_F675W] += out[EL_HAlpha];
}
