/*
 * cmio_temperature.c - ORACLE (test infrastructure): thermal balance.
 *
 * Restates src/TemperatureCalculator.cpp:
 *   :207-501  compute_cooling_and_heating_balance
 *   :567-931  calculate_temperature (one cell)
 *   :944-970  grid loop
 */
#include "cmio_internal.h"

#include <math.h>

/* line cooling element order, src/LineCoolingData.hpp:38-80 */
enum { NI = 0, NII, OI, OII, OIII, NeIII, SII, SIII, CII, CIII, NIII, NeII,
       SIV };

void cmio_cooling_and_heating_balance(const cmio_model *model, double *h0,
                                      double *he0, double *gain, double *loss,
                                      double T, double n, double midpoint_z,
                                      const double j[CMIO_NION],
                                      const double h[2], double pahfac,
                                      double crfac, double crscale,
                                      double x[CMIO_NION]) {
  const double alphaH = cmio_recombination_rate(model, CMIO_ION_H_n, T);
  const double alphaHe = cmio_recombination_rate(model, CMIO_ION_He_n, T);
  const double jH = j[CMIO_ION_H_n];
  const double jHe = j[CMIO_ION_He_n];
  const double hH = h[0];
  const double hHe = h[1];
  const double T4 = T * 1.e-4;
  const double sqrtT = sqrt(T);
  const double logT = log(T);
  const double *A = model->abundance;
  const double AHe = A[CMIO_EL_He];

  cmio_ionization_states_hydrogen_helium(alphaH, alphaHe, jH, jHe, n, AHe, T,
                                         h0, he0);
  const double ne = n * (1. - *h0 + AHe * (1. - *he0));
  const double nhp = n * (1. - *h0);
  const double nhep = (1. - *he0) * n * AHe;
  const double nenhp = ne * nhp;
  const double nenhep = ne * nhep;

  /* heating: photoionization of H and He */
  *gain = n * (hH * *h0 + hHe * AHe * *he0);
  /* on-the-spot absorption of He Ly-alpha */
  const double alpha_e_2sP = 4.17e-20 * pow(T4, -0.861);
  const double pHots = 1. / (1. + 77. * *he0 / (sqrtT * *h0));
  *gain += pHots * 1.21765423e-18 * alpha_e_2sP * nenhep;
  /* PAH heating */
  *gain += 1.5e-37 * n * ne * pahfac;
  /* cosmic rays */
  double heatcr = 0.;
  if (crfac > 0.) {
    heatcr = crfac * 1.2e-25 / sqrt(ne);
    if (crscale > 0.) {
      heatcr *= exp(-fabs(midpoint_z) / crscale);
    }
  }
  *gain += heatcr;

  /* metal ionization balance at this temperature (overwrites x[2..13]) */
  const double nh0 = n * *h0;
  const double nhe0 = n * *he0 * AHe;
  cmio_ionization_states_metals(model, &j[2], ne, T, T4, nh0, nhe0, nhp, x);

  /* coolant number fractions: element abundance times the stage that emits
   * the lines (the tracked fraction of an ion is that of the stage ABOVE the
   * one named by the line cooling element) */
  double abund[13];
  abund[CII] = A[CMIO_EL_C] * (1. - x[CMIO_ION_C_p1] - x[CMIO_ION_C_p2]);
  abund[CIII] = A[CMIO_EL_C] * x[CMIO_ION_C_p1];
  abund[NI] = A[CMIO_EL_N] *
              (1. - x[CMIO_ION_N_n] - x[CMIO_ION_N_p1] - x[CMIO_ION_N_p2]);
  abund[NII] = A[CMIO_EL_N] * x[CMIO_ION_N_n];
  abund[NIII] = A[CMIO_EL_N] * x[CMIO_ION_N_p1];
  abund[OI] = A[CMIO_EL_O] * (1. - x[CMIO_ION_O_n] - x[CMIO_ION_O_p1]);
  abund[OII] = A[CMIO_EL_O] * x[CMIO_ION_O_n];
  abund[OIII] = A[CMIO_EL_O] * x[CMIO_ION_O_p1];
  abund[NeII] = A[CMIO_EL_Ne] * x[CMIO_ION_Ne_n];
  abund[NeIII] = A[CMIO_EL_Ne] * x[CMIO_ION_Ne_p1];
  abund[SII] = A[CMIO_EL_S] *
               (1. - x[CMIO_ION_S_p1] - x[CMIO_ION_S_p2] - x[CMIO_ION_S_p3]);
  abund[SIII] = A[CMIO_EL_S] * x[CMIO_ION_S_p1];
  abund[SIV] = A[CMIO_EL_S] * x[CMIO_ION_S_p2];

  *loss = cmio_line_cooling(T, ne, abund) * n;

  /* free-free cooling */
  const double c = 5.5 - logT;
  const double gff = 1.1 + 0.34 * exp(-c * c / 3.);
  *loss += 1.42e-40 * gff * sqrtT * (nenhp + nenhep);
  /* recombination cooling */
  const double Lhp =
      2.85e-40 * nenhp * sqrtT * (5.914 - 0.5 * logT + 0.01184 * cbrt(T));
  const double Lhep = 1.55e-39 * nenhep * pow(T, 0.3647);
  *loss += Lhp + Lhep;

  *loss = fmax(*loss, 0.);
  *gain = fmax(*gain, 0.);
}

static void set_neutral(double x[CMIO_NION], double heating[2]) {
  /* unlike the ionization-only path, N0, O0 and Ne0 are set to 0 here
   * (src/TemperatureCalculator.cpp:586-618) */
  for (int ion = 0; ion < CMIO_NION; ++ion)
    x[ion] = 0.;
  x[CMIO_ION_H_n] = 1.;
  x[CMIO_ION_He_n] = 1.;
  heating[0] = 0.;
  heating[1] = 0.;
}

void cmio_temperature_cell(const cmio_model *model, double jfac, double hfac,
                           double ntot, double midpoint_z, double *temperature,
                           const double J[CMIO_NION], double heating[2],
                           double x[CMIO_NION]) {
  const double jH = jfac * J[CMIO_ION_H_n];
  const double jHe = jfac * J[CMIO_ION_He_n];
  if ((jH == 0. && jHe == 0.) || ntot == 0.) {
    *temperature = 500.;
    set_neutral(x, heating);
    return;
  }
  /* cosmic ray factor of the cell: DensityValues default, i.e. the global
   * factor is used unchanged */
  double crfac = model->crfac;
  double h0, he0;
  if (crfac > 0.) {
    const double alphaH = cmio_recombination_rate(model, CMIO_ION_H_n, 8000.);
    const double alphaHe =
        cmio_recombination_rate(model, CMIO_ION_He_n, 8000.);
    cmio_ionization_states_hydrogen_helium(alphaH, alphaHe, jH, jHe, ntot,
                                           model->abundance[CMIO_EL_He],
                                           8000., &h0, &he0);
    if (h0 > model->crlim) {
      *temperature = 500.;
      set_neutral(x, heating);
      return;
    }
  }

  double T0 = *temperature;
  if (*temperature <= 4000.) {
    T0 = 8000.;
  }
  double j[CMIO_NION];
  for (int ion = 0; ion < CMIO_NION; ++ion)
    j[ion] = jfac * J[ion];
  double h[2];
  h[0] = hfac * heating[0];
  h[1] = hfac * heating[1];

  uint_fast32_t niter = 0;
  double gain0 = 1.;
  double loss0 = 0.;
  h0 = 0.;
  he0 = 0.;
  const double logtt = log(1.1 / 0.9);
  while (fabs(gain0 - loss0) > model->t_epsilon * gain0 &&
         niter < (uint_fast32_t)model->t_max_iterations) {
    ++niter;
    const double T1 = 1.1 * T0;
    double h01, he01, gain1, loss1;
    cmio_cooling_and_heating_balance(model, &h01, &he01, &gain1, &loss1, T1,
                                     ntot, midpoint_z, j, h, model->pahfac,
                                     crfac, model->crscale, x);
    const double T2 = 0.9 * T0;
    double h02, he02, gain2, loss2;
    cmio_cooling_and_heating_balance(model, &h02, &he02, &gain2, &loss2, T2,
                                     ntot, midpoint_z, j, h, model->pahfac,
                                     crfac, model->crscale, x);
    cmio_cooling_and_heating_balance(model, &h0, &he0, &gain0, &loss0, T0,
                                     ntot, midpoint_z, j, h, model->pahfac,
                                     crfac, model->crscale, x);
    /* logarithmic slopes of gain and loss between 0.9 T0 and 1.1 T0 */
    double expgain;
    if (gain2 > 0.) {
      expgain = (gain1 > 0.) ? log(gain1 / gain2) : -99.;
    } else {
      expgain = (gain1 > 0.) ? 99. : 0.;
    }
    double exploss;
    if (loss2 > 0.) {
      exploss = (loss1 > 0.) ? log(loss1 / loss2) : -99.;
    } else {
      exploss = (loss1 > 0.) ? 99. : 0.;
    }
    const double expdiff = expgain - exploss;
    if (gain0 > 0. && expdiff != 0.) {
      T0 *= pow(loss0 / gain0, logtt / expdiff);
    } else {
      T0 = T1;
    }
    if (T0 < model->t_min_ionized) {
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
  *temperature = T0;
  if (J[CMIO_ION_H_n] == 0.) {
    h0 = 1.;
  }
  if (J[CMIO_ION_He_n] == 0.) {
    he0 = 1.;
  }
  x[CMIO_ION_H_n] = h0;
  x[CMIO_ION_He_n] = he0;
  if (h0 == 1. || h0 <= 1.e-10) {
    for (int ion = CMIO_ION_C_p1; ion < CMIO_NION; ++ion)
      x[ion] = 0.;
  }
  heating[0] = h[0];
  heating[1] = h[1];
}

void cmio_calculate_temperature(const cmio_grid *grid, const cmio_model *model,
                                cmio_cells *cells, double totweight) {
  cmio_calculate_temperature_range(
      grid, model, cells, totweight, 0,
      (int64_t)grid->ncell[0] * grid->ncell[1] * grid->ncell[2]);
}

/* the `block` argument of TemperatureCalculator::calculate_temperature
 * (src/TemperatureCalculator.cpp:944-970): cells [first, first + count) */
void cmio_calculate_temperature_range(const cmio_grid *grid,
                                      const cmio_model *model,
                                      cmio_cells *cells, double totweight,
                                      int64_t first, int64_t count) {
  const double jfac = model->total_luminosity / totweight;
  const double hfac = jfac * CMIO_PLANCK;
  const double cellside_z = grid->sides[2] / grid->ncell[2];
  const double volume = (grid->sides[0] / grid->ncell[0]) *
                        (grid->sides[1] / grid->ncell[1]) * cellside_z;
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = first; i < first + count; ++i) {
    double J[CMIO_NION], heating[2], x[CMIO_NION];
    for (int ion = 0; ion < CMIO_NION; ++ion) {
      J[ion] = cells->mean_intensity[ion][i];
      x[ion] = cells->ionic_fraction[ion][i];
    }
    heating[0] = cells->heating[0][i];
    heating[1] = cells->heating[1][i];
    /* cell midpoint z: anchor + cellside * iz + 0.5 * cellside
     * (src/CartesianDensityGrid.hpp:85-89) */
    const int64_t iz = i % grid->ncell[2];
    const double zmid = (grid->anchor[2] + cellside_z * iz) + 0.5 * cellside_z;
    double T = cells->temperature[i];
    cmio_temperature_cell(model, jfac / volume, hfac / volume,
                          cells->number_density[i], zmid, &T, J, heating, x);
    cells->temperature[i] = T;
    for (int ion = 0; ion < CMIO_NION; ++ion)
      cells->ionic_fraction[ion][i] = x[ion];
    cells->heating[0][i] = heating[0];
    cells->heating[1][i] = heating[1];
  }
}
