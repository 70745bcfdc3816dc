/*
 * cmio_ionization.c - ORACLE (test infrastructure): per-cell ionization
 * balance.
 *
 * Restates src/IonizationStateCalculator.cpp:
 *   :70-272   calculate_ionization_state(jfac, hfac, vars)
 *   :323-501  compute_ionization_states_metals
 *   :511-530  grid loop (jfac = L / totweight, per cell / volume)
 *   :649-753  compute_ionization_states_hydrogen_helium
 *   :802-820  compute_ionization_state_hydrogen
 */
#include "cmio_internal.h"

#include <math.h>

double cmio_ionization_state_hydrogen(double alphaH, double jH, double nH) {
  if (jH > 0. && nH > 0.) {
    const double aa = 0.5 * jH / (nH * alphaH);
    const double bb = 2. / aa;
    if (bb < 1.e-10) {
      return fmax(1.e-14, 0.25 * bb);
    } else {
      const double cc = sqrt(bb + 1.);
      return fmax(1.e-14, 1. + aa * (1. - cc));
    }
  } else {
    return 1.;
  }
}

int cmio_ionization_states_hydrogen_helium(double alphaH, double alphaHe,
                                           double jH, double jHe, double nH,
                                           double AHe, double T, double *h0_out,
                                           double *he0_out) {
  if (jH < 1.e-20) {
    *h0_out = 1.;
    *he0_out = 1.;
    return 0;
  }
  const double alpha_e_2sP = 4.17e-20 * pow(T * 1.e-4, -0.861);
  const double ch1 = alphaH * nH / jH;
  const double ch2 = AHe * alpha_e_2sP * nH / jH;
  double che = 0.;
  if (jHe > 0.) {
    che = alphaHe * nH / jHe;
  }
  double h0old = 0.99 * (1. - exp(-0.5 / ch1));
  double h0 = 0.9 * h0old;
  double he0old = 1.;
  if (che > 0.) {
    he0old = 0.5 / che;
    he0old = fmin(he0old, 1.);
  }
  double he0 = 0.;
  int niter = 0;
  int status = 0;
  /* note the && : the loop stops as soon as EITHER fraction has converged */
  while (fabs(h0 - h0old) > 1.e-4 * h0old &&
         fabs(he0 - he0old) > 1.e-4 * he0old) {
    ++niter;
    h0old = h0;
    if (he0 > 0.) {
      he0old = he0;
    } else {
      he0old = 0.;
    }
    const double pHots = 1. / (1. + 77. * he0old / sqrt(T) / h0old);
    const double ch = ch1 - ch2 * AHe * (1. - he0old) * pHots / (1. - h0old);

    he0 = 1.;
    if (che) {
      const double bhe = (1. + 2. * AHe - h0) * che + 1.;
      const double che_bhe = che / bhe;
      const double opAHeh0 = 1. + AHe - h0;
      const double t1he = 4. * AHe * opAHeh0 * che_bhe * che_bhe;
      if (t1he < 1.e-3) {
        he0 = opAHeh0 * che_bhe;
      } else {
        he0 = (bhe - sqrt(bhe * bhe - 4. * AHe * opAHeh0 * che * che)) /
              (2. * AHe * che);
      }
    }
    const double b = ch * (2. + AHe - he0 * AHe) + 1.;
    const double ch_b = ch / b;
    const double opAHeh0AHe = 1. + AHe - he0 * AHe;
    const double t1 = 4. * ch_b * ch_b * opAHeh0AHe;
    if (t1 < 1.e-3) {
      h0 = ch_b * opAHeh0AHe;
    } else {
      h0 = (b - sqrt(b * b - 4. * ch * ch * opAHeh0AHe)) / (2. * ch);
    }
    if (niter > 10) {
      h0 = 0.5 * (h0 + h0old);
      he0 = 0.5 * (he0 + he0old);
    }
    if (niter > 20) {
      /* cmac_error("Too many iterations in ionization loop!") */
      status = 1;
      break;
    }
  }
  *h0_out = h0;
  *he0_out = he0;
  return status;
}

/* src/IonizationStateCalculator.cpp:323-501 */
void cmio_ionization_states_metals(const cmio_model *model,
                                   const double j_metals[12], double ne,
                                   double T, double T4, double nh0, double nhe0,
                                   double nhp, double x[CMIO_NION]) {
  const double jCp1 = j_metals[0], jCp2 = j_metals[1];
  const double jNn = j_metals[2], jNp1 = j_metals[3], jNp2 = j_metals[4];
  const double jOn = j_metals[5], jOp1 = j_metals[6];
  const double jNen = j_metals[7], jNep1 = j_metals[8];
  const double jSp1 = j_metals[9], jSp2 = j_metals[10], jSp3 = j_metals[11];

  double alpha[CMIO_NION];
  for (int ion = CMIO_ION_C_p1; ion < CMIO_NION; ++ion) {
    alpha[ion] = cmio_recombination_rate(model, ion, T);
  }

  /* carbon */
  {
    const double C21 = jCp1 / (ne * alpha[CMIO_ION_C_p1]);
    const double C32 =
        jCp2 / (ne * alpha[CMIO_ION_C_p2] +
                nh0 * cmio_ct_recombination_rate_H(CMIO_ION_C_p2, T4) +
                nhe0 * cmio_ct_recombination_rate_He(CMIO_ION_C_p2, T4));
    const double C31 = C32 * C21;
    const double sumC_inv = 1. / (1. + C21 + C31);
    x[CMIO_ION_C_p1] = C21 * sumC_inv;
    x[CMIO_ION_C_p2] = C31 * sumC_inv;
  }
  /* nitrogen */
  {
    const double N21 =
        (jNn + nhp * cmio_ct_ionization_rate_H(CMIO_ION_N_n, T4)) /
        (ne * alpha[CMIO_ION_N_n] +
         nh0 * cmio_ct_recombination_rate_H(CMIO_ION_N_n, T4));
    const double N32 =
        jNp1 / (ne * alpha[CMIO_ION_N_p1] +
                nh0 * cmio_ct_recombination_rate_H(CMIO_ION_N_p1, T4) +
                nhe0 * cmio_ct_recombination_rate_He(CMIO_ION_N_p1, T4));
    const double N43 =
        jNp2 / (ne * alpha[CMIO_ION_N_p2] +
                nh0 * cmio_ct_recombination_rate_H(CMIO_ION_N_p2, T4) +
                nhe0 * cmio_ct_recombination_rate_He(CMIO_ION_N_p2, T4));
    const double N31 = N32 * N21;
    const double N41 = N43 * N31;
    const double sumN_inv = 1. / (1. + N21 + N31 + N41);
    x[CMIO_ION_N_n] = N21 * sumN_inv;
    x[CMIO_ION_N_p1] = N31 * sumN_inv;
    x[CMIO_ION_N_p2] = N41 * sumN_inv;
  }
  /* oxygen */
  {
    const double O21 =
        (jOn + nhp * cmio_ct_ionization_rate_H(CMIO_ION_O_n, T4)) /
        (ne * alpha[CMIO_ION_O_n] +
         nh0 * cmio_ct_recombination_rate_H(CMIO_ION_O_n, T4));
    const double O32 =
        jOp1 / (ne * alpha[CMIO_ION_O_p1] +
                nh0 * cmio_ct_recombination_rate_H(CMIO_ION_O_p1, T4) +
                nhe0 * cmio_ct_recombination_rate_He(CMIO_ION_O_p1, T4));
    const double O31 = O32 * O21;
    const double sumO_inv = 1. / (1. + O21 + O31);
    x[CMIO_ION_O_n] = O21 * sumO_inv;
    x[CMIO_ION_O_p1] = O31 * sumO_inv;
  }
  /* neon */
  {
    const double Ne21 = jNen / (ne * alpha[CMIO_ION_Ne_n]);
    const double Ne32 =
        jNep1 / (ne * alpha[CMIO_ION_Ne_p1] +
                 nh0 * cmio_ct_recombination_rate_H(CMIO_ION_Ne_p1, T4) +
                 nhe0 * cmio_ct_recombination_rate_He(CMIO_ION_Ne_p1, T4));
    const double Ne31 = Ne32 * Ne21;
    const double sumNe_inv = 1. / (1. + Ne21 + Ne31);
    x[CMIO_ION_Ne_n] = Ne21 * sumNe_inv;
    x[CMIO_ION_Ne_p1] = Ne31 * sumNe_inv;
  }
  /* sulphur */
  {
    const double S21 =
        jSp1 / (ne * alpha[CMIO_ION_S_p1] +
                nh0 * cmio_ct_recombination_rate_H(CMIO_ION_S_p1, T4));
    const double S32 =
        jSp2 / (ne * alpha[CMIO_ION_S_p2] +
                nh0 * cmio_ct_recombination_rate_H(CMIO_ION_S_p2, T4) +
                nhe0 * cmio_ct_recombination_rate_He(CMIO_ION_S_p2, T4));
    const double S43 =
        jSp3 / (ne * alpha[CMIO_ION_S_p3] +
                nh0 * cmio_ct_recombination_rate_H(CMIO_ION_S_p3, T4) +
                nhe0 * cmio_ct_recombination_rate_He(CMIO_ION_S_p3, T4));
    const double S31 = S32 * S21;
    const double S41 = S43 * S31;
    const double sumS_inv = 1. / (1. + S21 + S31 + S41);
    x[CMIO_ION_S_p1] = S21 * sumS_inv;
    x[CMIO_ION_S_p2] = S31 * sumS_inv;
    x[CMIO_ION_S_p3] = S41 * sumS_inv;
  }
}

/* src/IonizationStateCalculator.cpp:70-272 for one cell given as scalars.
 * J[14] un-normalised integrals, heating[2] in/out (normalised in place),
 * x[14] out. */
void cmio_ionization_state_cell(const cmio_model *model, double jfac,
                                double hfac, double ntot, double T,
                                const double J[CMIO_NION], double heating[2],
                                double x[CMIO_NION]) {
  const double jH = jfac * J[CMIO_ION_H_n];
  const double jHe = jfac * J[CMIO_ION_He_n];
  heating[0] = hfac * heating[0];
  heating[1] = hfac * heating[1];

  if (jH > 0. && ntot > 0.) {
    const double alphaH = cmio_recombination_rate(model, CMIO_ION_H_n, T);
    const double AHe = model->abundance[CMIO_EL_He];
    double h0, he0 = 0.;
    if (AHe != 0.) {
      const double alphaHe = cmio_recombination_rate(model, CMIO_ION_He_n, T);
      cmio_ionization_states_hydrogen_helium(alphaH, alphaHe, jH, jHe, ntot,
                                             AHe, T, &h0, &he0);
    } else {
      h0 = cmio_ionization_state_hydrogen(alphaH, jH, ntot);
    }
    x[CMIO_ION_H_n] = h0;
    x[CMIO_ION_He_n] = he0;

    const double nhp = ntot * (1. - h0);
    const double ne = ntot * (1. - h0 + AHe * (1. - he0));
    const double T4 = T * 1.e-4;
    double j_metals[12];
    for (int i = 0; i < 12; ++i) {
      j_metals[i] = jfac * J[CMIO_ION_C_p1 + i];
    }
    const double nh0 = ntot * h0;
    const double nhe0 = ntot * he0 * AHe;
    cmio_ionization_states_metals(model, j_metals, ne, T, T4, nh0, nhe0, nhp,
                                  x);
  } else {
    if (ntot > 0.) {
      /* no ionizing radiation: neutral. N, O and Ne have their neutral stage
       * as a tracked ion, so those fractions are 1 (:190-224) */
      for (int ion = 0; ion < CMIO_NION; ++ion)
        x[ion] = 0.;
      x[CMIO_ION_H_n] = 1.;
      x[CMIO_ION_He_n] = 1.;
      x[CMIO_ION_N_n] = 1.;
      x[CMIO_ION_O_n] = 1.;
      x[CMIO_ION_Ne_n] = 1.;
    } else {
      for (int ion = 0; ion < CMIO_NION; ++ion)
        x[ion] = 0.;
    }
  }
}

/* src/IonizationStateCalculator.cpp:511-530 + :135-139 (per-cell volume) */
void cmio_calculate_ionization_state(const cmio_grid *grid,
                                     const cmio_model *model,
                                     cmio_cells *cells, double totweight) {
  cmio_calculate_ionization_state_range(
      grid, model, cells, totweight, 0,
      (int64_t)grid->ncell[0] * grid->ncell[1] * grid->ncell[2]);
}

/* the `block` argument of
 * IonizationStateCalculator::calculate_ionization_state
 * (src/IonizationStateCalculator.cpp:511-530): cells [first, first + count) */
void cmio_calculate_ionization_state_range(const cmio_grid *grid,
                                           const cmio_model *model,
                                           cmio_cells *cells, double totweight,
                                           int64_t first, int64_t count) {
  const double jfac = model->total_luminosity / totweight;
  const double hfac = jfac * CMIO_PLANCK;
  /* src/CartesianDensityGrid.hpp:98-100 */
  const double volume = (grid->sides[0] / grid->ncell[0]) *
                        (grid->sides[1] / grid->ncell[1]) *
                        (grid->sides[2] / grid->ncell[2]);
#pragma omp parallel for
  for (int64_t i = first; i < first + count; ++i) {
    double J[CMIO_NION], heating[2], x[CMIO_NION];
    for (int ion = 0; ion < CMIO_NION; ++ion)
      J[ion] = cells->mean_intensity[ion][i];
    heating[0] = cells->heating[0][i];
    heating[1] = cells->heating[1][i];
    cmio_ionization_state_cell(model, jfac / volume, hfac / volume,
                               cells->number_density[i],
                               cells->temperature[i], J, heating, x);
    for (int ion = 0; ion < CMIO_NION; ++ion)
      cells->ionic_fraction[ion][i] = x[ion];
    cells->heating[0][i] = heating[0];
    cells->heating[1][i] = heating[1];
  }
}
