/*
 * device_emissivity.h - line and continuum emissivities of a cell from its
 * converged state: EmissivityCalculator::calculate_emissivities
 * (src/EmissivityCalculator.cpp:126-430) on LineCoolingData::get_line_strengths
 * (src/LineCoolingData.cpp:1859-1952). Post-processing: runs once on the final
 * grid, one cell per lane; kept apart from the thermal balance's
 * line_cooling() so that the temperature kernel's code does not change.
 */
#ifndef CMI_DEVICE_EMISSIVITY_H
#define CMI_DEVICE_EMISSIVITY_H

#include "device_thermal.h"

#define CMI_NEMISSIONLINE 42
#define CMI_NLINESTRENGTH (10 * CMI_LC_NFIVE_DEV + CMI_LC_NTWO_DEV)

/* ions of the line-cooling data (src/LineCoolingData.hpp) and transitions of
 * a five-level ion (:87-114) */
namespace cmi_em {
enum { NI = 0, NII, OI, OII, OIII, NeIII, SII, SIII, CII, CIII, NIII, NeII,
       SIV };
enum { T01 = 0, T02, T03, T04, T12, T13, T14, T23, T24, T34 };
/* src/EmissivityValues.hpp:36-81 */
enum {
  HAlpha = 0, HBeta, HII, BALMER_JUMP_LOW, BALMER_JUMP_HIGH, OI_6300, OI_6364,
  OII_3727, OIII_5007, OIII_4959, OIII_4363, OIII_52mu, OIII_88mu, NII_5755,
  NII_6548, NII_6584, NeIII_3869, NeIII_3968, SII_6725, SII_4072, SIII_9405,
  SIII_6312, SIII_19mu, SIII_33mu, avg_T, avg_T_count, avg_nH_nHe,
  avg_nH_nHe_count, NeII_12mu, NIII_57mu, NeIII_15mu, NII_122mu, CII_158mu,
  CII_2325, CIII_1908, OII_7325, SIV_10mu, HeI_5876, Hrec_s, WFC2_F439W,
  WFC2_F555W, WFC2_F675W
};
} // namespace cmi_em

/* which line strengths add up to which emission line
 * (src/EmissivityCalculator.cpp:247-372, :395-423): {line, 10 ion +
 * transition | 100 + two-level ion} */
struct EmissionTerm {
  uint8_t line, strength;
};
#define CMI_EM5(ion, t) (uint8_t)(10 * cmi_em::ion + cmi_em::t)
#define CMI_EM2(ion) (uint8_t)(100 + cmi_em::ion - cmi_em::NIII)
__device__ __constant__ const EmissionTerm cmi_emission_terms[] = {
    {cmi_em::NII_5755, CMI_EM5(NII, T34)},
    {cmi_em::NII_6548, CMI_EM5(NII, T13)},
    {cmi_em::NII_6584, CMI_EM5(NII, T23)},
    {cmi_em::NII_122mu, CMI_EM5(NII, T12)},
    {cmi_em::OI_6300, CMI_EM5(OI, T03)},
    {cmi_em::OI_6364, CMI_EM5(OI, T13)},
    {cmi_em::OII_3727, CMI_EM5(OII, T01)},
    {cmi_em::OII_3727, CMI_EM5(OII, T02)},
    {cmi_em::OII_7325, CMI_EM5(OII, T14)},
    {cmi_em::OII_7325, CMI_EM5(OII, T24)},
    {cmi_em::OII_7325, CMI_EM5(OII, T13)},
    {cmi_em::OII_7325, CMI_EM5(OII, T23)},
    {cmi_em::OIII_4363, CMI_EM5(OIII, T34)},
    {cmi_em::OIII_4959, CMI_EM5(OIII, T13)},
    {cmi_em::OIII_5007, CMI_EM5(OIII, T23)},
    {cmi_em::OIII_52mu, CMI_EM5(OIII, T12)},
    {cmi_em::OIII_88mu, CMI_EM5(OIII, T01)},
    {cmi_em::NeIII_3869, CMI_EM5(NeIII, T03)},
    {cmi_em::NeIII_3968, CMI_EM5(NeIII, T13)},
    {cmi_em::NeIII_15mu, CMI_EM5(NeIII, T01)},
    {cmi_em::SII_4072, CMI_EM5(SII, T03)},
    {cmi_em::SII_4072, CMI_EM5(SII, T04)},
    {cmi_em::SII_6725, CMI_EM5(SII, T01)},
    {cmi_em::SII_6725, CMI_EM5(SII, T02)},
    {cmi_em::SIII_9405, CMI_EM5(SIII, T13)},
    {cmi_em::SIII_9405, CMI_EM5(SIII, T23)},
    {cmi_em::SIII_6312, CMI_EM5(SIII, T34)},
    {cmi_em::SIII_19mu, CMI_EM5(SIII, T12)},
    {cmi_em::SIII_33mu, CMI_EM5(SIII, T01)},
    {cmi_em::CII_158mu, CMI_EM5(CII, T01)},
    {cmi_em::CII_2325, CMI_EM5(CII, T02)},
    {cmi_em::CII_2325, CMI_EM5(CII, T12)},
    {cmi_em::CII_2325, CMI_EM5(CII, T03)},
    {cmi_em::CII_2325, CMI_EM5(CII, T13)},
    {cmi_em::CII_2325, CMI_EM5(CII, T04)},
    {cmi_em::CII_2325, CMI_EM5(CII, T14)},
    {cmi_em::CIII_1908, CMI_EM5(CIII, T01)},
    {cmi_em::CIII_1908, CMI_EM5(CIII, T02)},
    {cmi_em::CIII_1908, CMI_EM5(CIII, T03)},
    {cmi_em::NIII_57mu, CMI_EM2(NIII)},
    {cmi_em::NeII_12mu, CMI_EM2(NeII)},
    {cmi_em::SIV_10mu, CMI_EM2(SIV)},
    /* the three WFC2 filters (H beta / H alpha are added separately) */
    {cmi_em::WFC2_F439W, CMI_EM5(OIII, T34)},
    {cmi_em::WFC2_F439W, CMI_EM5(SIII, T03)},
    {cmi_em::WFC2_F439W, CMI_EM5(SIII, T04)},
    {cmi_em::WFC2_F555W, CMI_EM5(NI, T01)},
    {cmi_em::WFC2_F555W, CMI_EM5(NI, T02)},
    {cmi_em::WFC2_F555W, CMI_EM5(NII, T34)},
    {cmi_em::WFC2_F555W, CMI_EM5(OI, T34)},
    {cmi_em::WFC2_F555W, CMI_EM5(OIII, T03)},
    {cmi_em::WFC2_F555W, CMI_EM5(OIII, T13)},
    {cmi_em::WFC2_F555W, CMI_EM5(OIII, T23)},
    {cmi_em::WFC2_F675W, CMI_EM5(NII, T03)},
    {cmi_em::WFC2_F675W, CMI_EM5(NII, T13)},
    {cmi_em::WFC2_F675W, CMI_EM5(NII, T23)},
    {cmi_em::WFC2_F675W, CMI_EM5(OI, T03)},
    {cmi_em::WFC2_F675W, CMI_EM5(OI, T13)},
    {cmi_em::WFC2_F675W, CMI_EM5(OI, T23)},
    {cmi_em::WFC2_F675W, CMI_EM5(SII, T01)},
    {cmi_em::WFC2_F675W, CMI_EM5(SII, T02)},
    {cmi_em::WFC2_F675W, CMI_EM5(SIII, T34)},
};
#undef CMI_EM5
#undef CMI_EM2
#define CMI_NEMISSIONTERM                                                      \
  (int)(sizeof(cmi_emission_terms) / sizeof(cmi_emission_terms[0]))

/* LineCoolingData::get_line_strengths: strength[10 e + t], strength[100 + i]
 * (J s^-1 per hydrogen atom). The level populations as in line_cooling(). */
__device__ inline void line_strengths(const LineCoolingDev &lc,
                                      double temperature,
                                      double electron_density,
                                      const double (&abund)[13],
                                      double (&strength)[CMI_NLINESTRENGTH]) {
  const double kb = CMI_BOLTZMANN;
  const double prefactor = lc.prefactor * electron_density / sqrt(temperature);
  const double Tinv = 1. / temperature;
  const double logT = log(temperature);
#pragma unroll 1
  for (int e = 0; e < CMI_LC_NFIVE_DEV; ++e) {
    const double *A = lc.A[e];
    const double *w = lc.inv_weight[e];
    const double *E = lc.energy[e];
    double M[5][5];
    double pop[5] = {1., 0., 0., 0., 0.};
    double sumC[5] = {0., 0., 0., 0., 0.};
#pragma unroll
    for (int lo = 0; lo < 4; ++lo) {
#pragma unroll
      for (int hi = lo + 1; hi < 5; ++hi) {
        const int t = lc_tr(lo, hi);
        const double down = lc_collision_strength(lc.cs[e][t], prefactor,
                                                  temperature, Tinv, logT);
        const double up = down * exp(-E[t] * Tinv);
        M[hi][lo] = up * w[lo];
        if (lo > 0)
          M[lo][hi] = A[t] + w[hi] * down;
        sumC[hi] += down;
        sumC[lo] += up;
      }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
      M[0][k] = 1.;
#pragma unroll
    for (int i = 1; i < 5; ++i) {
      double sumA = A[lc_tr(0, i)];
#pragma unroll
      for (int j = 1; j < i; ++j)
        sumA += A[lc_tr(j, i)];
      M[i][i] = -(sumA + w[i] * sumC[i]);
    }
    (void)solve_5x5(M, pop);
    const double pre = abund[e] * kb;
#pragma unroll
    for (int lo = 0; lo < 4; ++lo) {
#pragma unroll
      for (int hi = lo + 1; hi < 5; ++hi) {
        const int t = lc_tr(lo, hi);
        strength[10 * e + t] = pre * pop[hi] * A[t] * E[t];
      }
    }
  }
#pragma unroll 1
  for (int i = 0; i < CMI_LC_NTWO_DEV; ++i) {
    const double ksi = lc.two_energy[i];
    const double cs = lc_collision_strength(lc.two_cs[i], prefactor,
                                            temperature, Tinv, logT);
    const double Texp = exp(-ksi * Tinv);
    const double pop =
        cs * Texp * lc.two_inv_weight[i][0] /
        (lc.two_A[i] +
         cs * (lc.two_inv_weight[i][1] + Texp * lc.two_inv_weight[i][0]));
    strength[10 * CMI_LC_NFIVE_DEV + i] =
        abund[CMI_LC_NFIVE_DEV + i] * kb * pop * ksi * lc.two_A[i];
  }
}

/* EmissivityCalculator::get_balmer_jump_emission,
 * src/EmissivityCalculator.cpp:42-116: {H high, H low, He high, He low} */
__device__ inline void balmer_jump(double T, double (&out)[4]) {
  const double ttab[8] = {4.e3, 6.e3, 8.e3, 1.e4, 1.2e4, 1.4e4, 1.6e4, 1.8e4};
  const double coefficient[4][8] = {
      {0.162, 0.584, 1.046, 1.437, 1.742, 1.977, 2.159, 2.297},
      {92.6, 50.9, 33.8, 24.8, 19.53, 16.09, 13.7, 11.96},
      {0.189, 0.622, 1.076, 1.45, 1.74, 1.963, 2.14, 2.27},
      {15.7, 9.23, 6.71, 5.49, 4.83, 4.41, 4.135, 3.94}};
  const double wavelength[4] = {3681., 3643., 3681., 3643.};
  const double logt = log(T);
  /* Utilities::locate on the logarithms of ttab, clamped to [0, 6] */
  int i = 0;
  while (i < 6 && logt > log(ttab[i + 1]))
    ++i;
  const double lt0 = log(ttab[i]), lt1 = log(ttab[i + 1]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double lo = log(coefficient[k][i]), hi = log(coefficient[k][i + 1]);
    const double v = exp(lo + (logt - lt0) * (hi - lo) / (lt1 - lt0));
    out[k] = v * (1.e-43 * 299792458. / (wavelength[k] * wavelength[k]));
  }
}

/* calculate_emissivities for one cell, src/EmissivityCalculator.cpp:126-430 */
__device__ inline void cell_emissivities(const ModelDev &m, double ntot,
                                         double T, const double (&x)[CMI_NION],
                                         double (&out)[CMI_NEMISSIONLINE]) {
  using namespace cmi_em;
#pragma unroll
  for (int l = 0; l < CMI_NEMISSIONLINE; ++l)
    out[l] = 0.;
  if (!(x[ION_H_n] < 0.2 && T > 3000.))
    return;
  const double AHe = m.abundance[0];
  const double nhp = ntot * (1. - x[ION_H_n]);
  const double nhep = ntot * (1. - x[ION_He_n]) * AHe;
  const double ne = nhp + nhep;
  const double AC = m.abundance[1], AN = m.abundance[2], AO = m.abundance[3],
               ANe = m.abundance[4], AS = m.abundance[5];
  double abund[13];
  abund[CII] = AC * (1. - x[ION_C_p1] - x[ION_C_p2]);
  abund[CIII] = AC * x[ION_C_p1];
  abund[NI] = AN * (1. - x[ION_N_n] - x[ION_N_p1] - x[ION_N_p2]);
  abund[NII] = AN * x[ION_N_n];
  abund[NIII] = AN * x[ION_N_p1];
  abund[OI] = AO * (1. - x[ION_O_n] - x[ION_O_p1]);
  abund[OII] = AO * x[ION_O_n];
  abund[OIII] = AO * x[ION_O_p1];
  abund[NeII] = ANe * x[ION_Ne_n];
  abund[NeIII] = ANe * x[ION_Ne_p1];
  abund[SII] = AS * (1. - x[ION_S_p1] - x[ION_S_p2] - x[ION_S_p3]);
  abund[SIII] = AS * x[ION_S_p1];
  abund[SIV] = AS * x[ION_S_p2];

  double strength[CMI_NLINESTRENGTH];
  line_strengths(m.tables->lc, T, ne, abund, strength);
  for (int k = 0; k < CMI_NEMISSIONTERM; ++k)
    out[cmi_emission_terms[k].line] += strength[cmi_emission_terms[k].strength];
  for (int l = 0; l < CMI_NEMISSIONLINE; ++l)
    out[l] *= ntot;

  const double T4 = T * 1.e-4;
  out[HAlpha] = ne * nhp * 2.87 * 1.24e-38 * pow(T4, -0.938);
  out[HBeta] = ne * nhp * 1.24e-38 * pow(T4, -0.878);
  out[HII] = nhp * ne * 4.9e-40 * pow(T4, -0.848);
  out[HeI_5876] = ne * nhep * 1.69e-38 * pow(T4, -1.065);
  out[Hrec_s] = ne * nhp * 7.982e-23 /
                (sqrt(T / 3.148) * pow(1. + sqrt(T / 3.148), 0.252) *
                 pow(1. + sqrt(T / 7.036e5), 1.748));
  double jump[4];
  balmer_jump(T, jump);
  out[BALMER_JUMP_LOW] = ne * (nhp * jump[1] + nhep * jump[3]);
  out[BALMER_JUMP_HIGH] = ne * (nhp * jump[0] + nhep * jump[2]);
  out[avg_T] = ne * nhp * T;
  out[avg_T_count] = ne * nhp;
  out[avg_nH_nHe] = ne * (1. - x[ION_He_n]);
  out[avg_nH_nHe_count] = ne * (1. - x[ION_H_n]);
  out[WFC2_F555W] += out[HBeta];
  out[WFC2_F675W] += out[HAlpha];
}

#endif
