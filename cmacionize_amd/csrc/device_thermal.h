/*
 * device_thermal.h - line cooling and the thermal balance of one cell.
 */
#ifndef CMI_DEVICE_THERMAL_H
#define CMI_DEVICE_THERMAL_H

#include "device_physics.h"

/* transition (lower i, upper j) -> index of the 10-entry transition tables */
__device__ __forceinline__ constexpr int lc_tr(int i, int j) {
  return i == 0 ? j - 1 : (i == 1 ? j + 2 : (i == 2 ? j + 4 : 9));
}

/* LineCoolingData::solve_system_of_linear_equations,
 * src/LineCoolingData.cpp:1492-1555: Gaussian elimination with partial
 * pivoting, fully unrolled with static indices so that the 5x5 system stays
 * in registers. Returns false for a singular matrix. */
__device__ inline bool solve_5x5(double (&A)[5][5], double (&B)[5]) {
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    int imax = 0;
    double Amax = 0.;
#pragma unroll
    for (int i = j; i < 5; ++i) {
      if (fabs(A[i][j]) > fabs(Amax)) {
        Amax = A[i][j];
        imax = i;
      }
    }
    if (Amax == 0.)
      return false;
    const double Amax_inv = 1. / Amax;
    /* swap rows j and imax (select instead of dynamic indexing) */
#pragma unroll
    for (int i = j + 1; i < 5; ++i) {
      if (imax == i) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const double t = A[j][k];
          A[j][k] = A[i][k];
          A[i][k] = t;
        }
        const double t = B[j];
        B[j] = B[i];
        B[i] = t;
      }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
      A[j][k] *= Amax_inv;
    B[j] *= Amax_inv;
#pragma unroll
    for (int i = j + 1; i < 5; ++i) {
#pragma unroll
      for (int k = j + 1; k < 5; ++k)
        A[i][k] -= A[i][j] * A[j][k];
      B[i] -= A[i][j] * B[j];
    }
  }
#pragma unroll
  for (int i = 3; i >= 0; --i) {
#pragma unroll
    for (int j = 4; j > i; --j)
      B[i] -= B[j] * A[i][j];
  }
  return true;
}

/* collision strength fit, src/LineCoolingData.cpp:1590-1601. The two powers
 * of T are taken as exp(y ln T) with the logarithm the fit needs anyway: a
 * balance evaluation has 206 of them, and the general pow() is 4-5x the
 * instructions of exp(). Relative difference to pow(): ~|y ln T| ulp ~ 2e-15,
 * far inside the 1e-6 of the reference's own line-cooling test. */
__device__ __forceinline__ double lc_collision_strength(const double *a,
                                                        double prefactor,
                                                        double T, double Tinv,
                                                        double logT) {
  return prefactor * exp((1. + a[0]) * logT) *
         (a[1] + a[2] * Tinv + a[3] * logT +
          a[4] * T * (1. + (a[5] - 1.) * exp(a[6] * logT)));
}

/* LineCoolingData::get_cooling, src/LineCoolingData.cpp:1767-1847 with
 * compute_level_populations (:1569-1701) and compute_level_population
 * (:1714-1738). abund[13]: NI NII OI OII OIII NeIII SII SIII CII CIII NIII NeII
 * SIV number fractions relative to H. */
__device__ inline double line_cooling(const LineCoolingDev &lc,
                                      double temperature,
                                      double electron_density,
                                      const double abund[13]) {
  if (electron_density == 0.)
    return 1.e-99;
  const double kb = CMI_BOLTZMANN;
  const double prefactor = lc.prefactor * electron_density / sqrt(temperature);
  const double Tinv = 1. / temperature;
  const double logT = log(temperature);

  double cooling = 0.;
  for (int e = 0; e < CMI_LC_NFIVE_DEV; ++e) {
    double down[CMI_LC_NTRANS_DEV], up[CMI_LC_NTRANS_DEV];
#pragma unroll
    for (int t = 0; t < CMI_LC_NTRANS_DEV; ++t) {
      const double cs =
          lc_collision_strength(lc.cs[e][t], prefactor, temperature, Tinv, logT);
      down[t] = cs;
      up[t] = cs * exp(-lc.energy[e][t] * Tinv);
    }
    const double *A = lc.A[e];
    const double *w = lc.inv_weight[e];
    double M[5][5];
    double pop[5] = {1., 0., 0., 0., 0.};
#pragma unroll
    for (int k = 0; k < 5; ++k)
      M[0][k] = 1.; /* populations sum to 1 */
#pragma unroll
    for (int i = 1; i < 5; ++i) {
#pragma unroll
      for (int j = 0; j < i; ++j)
        M[i][j] = up[lc_tr(j, i)] * w[j];
      double sumA = A[lc_tr(0, i)];
#pragma unroll
      for (int j = 1; j < i; ++j)
        sumA += A[lc_tr(j, i)];
      double sumC = down[lc_tr(0, i)];
#pragma unroll
      for (int j = 1; j < i; ++j)
        sumC += down[lc_tr(j, i)];
#pragma unroll
      for (int k = i + 1; k < 5; ++k)
        sumC += up[lc_tr(i, k)];
      M[i][i] = -(sumA + w[i] * sumC);
#pragma unroll
      for (int k = i + 1; k < 5; ++k)
        M[i][k] = A[lc_tr(i, k)] + w[k] * down[lc_tr(i, k)];
    }
    /* a singular matrix aborts the reference (cmac_error); here the ion then
     * contributes the populations of the unsolved right-hand side */
    (void)solve_5x5(M, pop);
    const double *E = lc.energy[e];
    double cl[5];
    cl[1] = pop[1] * A[lc_tr(0, 1)] * E[lc_tr(0, 1)];
#pragma unroll
    for (int i = 2; i < 5; ++i) {
      double s = A[lc_tr(0, i)] * E[lc_tr(0, i)];
#pragma unroll
      for (int j = 1; j < i; ++j)
        s += A[lc_tr(j, i)] * E[lc_tr(j, i)];
      cl[i] = pop[i] * s;
    }
    cooling += abund[e] * kb * (cl[1] + cl[2] + cl[3] + cl[4]);
  }
#pragma unroll
  for (int i = 0; i < CMI_LC_NTWO_DEV; ++i) {
    const double ksi = lc.two_energy[i];
    const double cs = lc_collision_strength(lc.two_cs[i], prefactor,
                                            temperature, Tinv, logT);
    const double Texp = exp(-ksi * Tinv);
    const double pop =
        cs * Texp * lc.two_inv_weight[i][0] /
        (lc.two_A[i] +
         cs * (lc.two_inv_weight[i][1] + Texp * lc.two_inv_weight[i][0]));
    cooling += abund[CMI_LC_NFIVE_DEV + i] * kb * ksi * lc.two_A[i] * pop;
  }
  return cooling;
}

/* TemperatureCalculator::compute_cooling_and_heating_balance,
 * src/TemperatureCalculator.cpp:207-501. j[14], h[2] normalised integrals;
 * x[2..13] receive the metal fractions at temperature T. */
__device__ inline void cooling_and_heating_balance(
    const ModelDev &m, double &h0, double &he0, double &gain, double &loss,
    double T, double n, double midpoint_z, const double j[CMI_NION],
    const double h[2], double x[CMI_NION]) {
  enum { NI = 0, NII, OI, OII, OIII, NeIII, SII, SIII, CII, CIII, NIII, NeII,
         SIV };
  const double alphaH = cmi_recombination_rate(m, ION_H_n, T);
  const double alphaHe = cmi_recombination_rate(m, ION_He_n, T);
  const double T4 = T * 1.e-4;
  const double sqrtT = sqrt(T);
  const double logT = log(T);
  const double AHe = m.abundance[0];

  cmi_ionization_states_hydrogen_helium(alphaH, alphaHe, j[ION_H_n],
                                        j[ION_He_n], n, AHe, T, h0, he0);
  const double ne = n * (1. - h0 + AHe * (1. - he0));
  const double nhp = n * (1. - h0);
  const double nhep = (1. - he0) * n * AHe;
  const double nenhp = ne * nhp;
  const double nenhep = ne * nhep;

  gain = n * (h[0] * h0 + h[1] * AHe * he0);
  const double alpha_e_2sP = 4.17e-20 * pow(T4, -0.861);
  const double pHots = 1. / (1. + 77. * he0 / (sqrtT * h0));
  gain += pHots * 1.21765423e-18 * alpha_e_2sP * nenhep;
  gain += 1.5e-37 * n * ne * m.pahfac;
  double heatcr = 0.;
  if (m.crfac > 0.) {
    heatcr = m.crfac * 1.2e-25 / sqrt(ne);
    if (m.crscale > 0.)
      heatcr *= exp(-fabs(midpoint_z) / m.crscale);
  }
  gain += heatcr;

  const double nh0 = n * h0;
  const double nhe0 = n * he0 * AHe;
  cmi_ionization_states_metals(m, &j[2], ne, T, T4, nh0, nhe0, nhp, x);

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

  loss = line_cooling(m.tables->lc, T, ne, abund) * n;
  const double c = 5.5 - logT;
  const double gff = 1.1 + 0.34 * exp(-c * c / 3.);
  loss += 1.42e-40 * gff * sqrtT * (nenhp + nenhep);
  const double Lhp =
      2.85e-40 * nenhp * sqrtT * (5.914 - 0.5 * logT + 0.01184 * cbrt(T));
  const double Lhep = 1.55e-39 * nenhep * pow(T, 0.3647);
  loss += Lhp + Lhep;
  loss = fmax(loss, 0.);
  gain = fmax(gain, 0.);
}

/* TemperatureCalculator::calculate_temperature(vars, jfac, hfac, midpoint),
 * src/TemperatureCalculator.cpp:567-931. J[14], heating[2] un-normalised;
 * temperature in/out; x[14] in/out. */
__device__ inline void temperature_cell(const ModelDev &m, double jfac,
                                        double hfac, double ntot,
                                        double midpoint_z, double &temperature,
                                        const double J[CMI_NION],
                                        double heating[2],
                                        double x[CMI_NION]) {
  const double jH = jfac * J[ION_H_n];
  const double jHe = jfac * J[ION_He_n];
  bool neutral = (jH == 0. && jHe == 0.) || ntot == 0.;
  const double crfac = m.crfac;
  double h0 = 0., he0 = 0.;
  if (!neutral && crfac > 0.) {
    const double alphaH = cmi_recombination_rate(m, ION_H_n, 8000.);
    const double alphaHe = cmi_recombination_rate(m, ION_He_n, 8000.);
    cmi_ionization_states_hydrogen_helium(alphaH, alphaHe, jH, jHe, ntot,
                                          m.abundance[0], 8000., h0, he0);
    neutral = h0 > m.crlim;
  }
  if (neutral) {
    /* no radiation, vacuum, or too neutral for cosmic ray heating: 500 K,
     * H and He neutral, all metal fractions (including N0, O0, Ne0) zero */
    temperature = 500.;
#pragma unroll
    for (int i = 0; i < CMI_NION; ++i)
      x[i] = 0.;
    x[ION_H_n] = 1.;
    x[ION_He_n] = 1.;
    heating[0] = 0.;
    heating[1] = 0.;
    return;
  }

  double T0 = temperature;
  if (temperature <= 4000.)
    T0 = 8000.;
  double j[CMI_NION];
#pragma unroll
  for (int i = 0; i < CMI_NION; ++i)
    j[i] = jfac * J[i];
  double h[2] = {hfac * heating[0], hfac * heating[1]};

  int niter = 0;
  double gain0 = 1., loss0 = 0.;
  h0 = 0.;
  he0 = 0.;
  const double logtt = log(1.1 / 0.9);
  while (fabs(gain0 - loss0) > m.t_epsilon * gain0 &&
         niter < m.t_max_iterations) {
    ++niter;
    const double T1 = 1.1 * T0;
    double h01, he01, gain1, loss1;
    cooling_and_heating_balance(m, h01, he01, gain1, loss1, T1, ntot,
                                midpoint_z, j, h, x);
    const double T2 = 0.9 * T0;
    double h02, he02, gain2, loss2;
    cooling_and_heating_balance(m, h02, he02, gain2, loss2, T2, ntot,
                                midpoint_z, j, h, x);
    cooling_and_heating_balance(m, h0, he0, gain0, loss0, T0, ntot, midpoint_z,
                                j, h, x);
    double expgain;
    if (gain2 > 0.)
      expgain = (gain1 > 0.) ? log(gain1 / gain2) : -99.;
    else
      expgain = (gain1 > 0.) ? 99. : 0.;
    double exploss;
    if (loss2 > 0.)
      exploss = (loss1 > 0.) ? log(loss1 / loss2) : -99.;
    else
      exploss = (loss1 > 0.) ? 99. : 0.;
    const double expdiff = expgain - exploss;
    if (gain0 > 0. && expdiff != 0.)
      T0 *= pow(loss0 / gain0, logtt / expdiff);
    else
      T0 = T1;
    if (T0 < m.t_min_ionized) {
      T0 = 500.;
      h0 = 1.;
      he0 = 1.;
      gain0 = 1.;
      loss0 = 1.;
    }
    if (T0 > 1.e10) {
      T0 = 1.e10;
      h0 = 1.e-10;
      he0 = 1.e-10;
      gain0 = 1.;
      loss0 = 1.;
    }
  }
  T0 = fmin(30000., T0);
  temperature = T0;
  if (J[ION_H_n] == 0.)
    h0 = 1.;
  if (J[ION_He_n] == 0.)
    he0 = 1.;
  x[ION_H_n] = h0;
  x[ION_He_n] = he0;
  if (h0 == 1. || h0 <= 1.e-10) {
#pragma unroll
    for (int i = ION_C_p1; i < CMI_NION; ++i)
      x[i] = 0.;
  }
  heating[0] = h[0];
  heating[1] = h[1];
}

#endif
