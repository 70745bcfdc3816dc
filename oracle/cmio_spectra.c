/*
 * cmio_spectra.c - ORACLE (test infrastructure): source spectra, diffuse
 * re-emission spectra and the re-emission decision tree.
 *
 * Restates
 *   src/PlanckPhotonSourceSpectrum.cpp:53-113 (table), :149-165 (sample)
 *   src/HydrogenLymanContinuumSpectrum.cpp:40-122, :136-153
 *   src/HeliumLymanContinuumSpectrum.cpp:45-133, :147-164
 *   src/HeliumTwoPhotonContinuumSpectrum.cpp:44-101, :167-180
 *   src/PhysicalDiffuseReemissionHandler.hpp:66-105 (probabilities)
 *   src/PhysicalDiffuseReemissionHandler.cpp:219-370 (reemit)
 *   src/FixedValueDiffuseReemissionHandler.hpp:73-86
 *   src/MonochromaticPhotonSourceSpectrum.hpp:97-100
 */
#include "cmio_atomic_data.h"
#include "cmio_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

cmio_tables *cmio_tables_create(const cmio_model *model) {
  cmio_tables *t = (cmio_tables *)calloc(1, sizeof(cmio_tables));
  const double h = CMIO_PLANCK;
  const double k = CMIO_BOLTZMANN;

  /* ---- Planck source spectra (src/PlanckPhotonSourceSpectrum.cpp:53-113):
   * the discrete sources', the continuous source's ---- */
  for (int which = 0; which < 2; ++which) {
    const int wanted =
        which == 0 ? model->spectrum_type == CMIO_SPECTRUM_PLANCK
                   : (model->continuous_type != 0 &&
                      model->continuous_spectrum_type == CMIO_SPECTRUM_PLANCK);
    if (!wanted)
      continue;
    const double temperature = which == 0
                                   ? model->planck_temperature
                                   : model->continuous_planck_temperature;
    double *planck_cdf = which == 0 ? t->planck_cdf : t->planck2_cdf;
    double *planck_logcdf = which == 0 ? t->planck_logcdf : t->planck2_logcdf;
    double *planck_logfreq =
        which == 0 ? t->planck_logfreq : t->planck2_logfreq;
    const double max_frequency = 4.;
    const double min_frequency = 3.289e15;
    static double frequency[CMIO_NFREQ], luminosity[CMIO_NFREQ];
    for (int i = 0; i < CMIO_NFREQ; ++i) {
      frequency[i] = 1. + i * (max_frequency - 1.) / (CMIO_NFREQ - 1.);
      luminosity[i] =
          frequency[i] * frequency[i] * frequency[i] /
          (exp(h * frequency[i] * min_frequency / (k * temperature)) - 1.);
    }
    planck_cdf[0] = 0.;
    for (int i = 1; i < CMIO_NFREQ; ++i) {
      planck_cdf[i] = planck_cdf[i - 1] +
                      0.5 *
                          (luminosity[i] / frequency[i] +
                           luminosity[i - 1] / frequency[i - 1]) *
                          (frequency[i] - frequency[i - 1]);
    }
    planck_logcdf[0] = -10.;
    planck_logfreq[0] = 0.;
    for (int i = 1; i < CMIO_NFREQ; ++i) {
      planck_cdf[i] /= planck_cdf[CMIO_NFREQ - 1];
      planck_logcdf[i] = log10(planck_cdf[i]);
      planck_logfreq[i] = log10(frequency[i]);
    }
  }

  /* ---- H and He Lyman continua (depend on the cross sections) ---- */
  for (int s = 0; s < 2; ++s) {
    const int ion = (s == 0) ? CMIO_ION_H_n : CMIO_ION_He_n;
    const double min_frequency =
        (s == 0) ? 3.289e15 : 1.81 * 3.288465385e15;
    const double max_frequency =
        (s == 0) ? 4. * min_frequency : 4. * 3.288465385e15;
    double *nu = t->lyc_freq[s];
    for (int i = 0; i < CMIO_NFREQ; ++i) {
      nu[i] = min_frequency +
              i * (max_frequency - min_frequency) / (CMIO_NFREQ - 1.);
    }
    for (int iT = 0; iT < CMIO_NTEMP; ++iT) {
      double *cdf = t->lyc_cdf[s][iT];
      cdf[0] = 0.;
      t->lyc_T[iT] = 1500. + (iT + 0.5) * 13500. / CMIO_NTEMP;
      for (int inu = 1; inu < CMIO_NFREQ; ++inu) {
        double xsec = cmio_cross_section(model, ion, nu[inu - 1]);
        const double j1 = nu[inu - 1] * nu[inu - 1] * nu[inu - 1] * xsec *
                          exp(-(h * (nu[inu - 1] - min_frequency)) /
                              (k * t->lyc_T[iT]));
        xsec = cmio_cross_section(model, ion, nu[inu]);
        const double j2 =
            nu[inu] * nu[inu] * nu[inu] * xsec *
            exp(-(h * (nu[inu] - min_frequency)) / (k * t->lyc_T[iT]));
        cdf[inu] =
            0.5 * (j1 / nu[inu] + j2 / nu[inu - 1]) * (nu[inu] - nu[inu - 1]);
      }
      for (int inu = 1; inu < CMIO_NFREQ; ++inu) {
        cdf[inu] = cdf[inu - 1] + cdf[inu];
      }
      const double total = cdf[CMIO_NFREQ - 1];
      for (int inu = 0; inu < CMIO_NFREQ; ++inu) {
        cdf[inu] /= total; /* NaN table if the cross section is zero */
      }
    }
  }

  /* ---- He two-photon continuum ---- */
  {
    const double min_frequency = 3.288465385e15;
    const double max_frequency = 1.6 * min_frequency;
    const double nu0 = 4.98e15;
    double *nu = t->he2pc_freq;
    double *cdf = t->he2pc_cdf;
    for (int i = 0; i < CMIO_NFREQ; ++i) {
      nu[i] = min_frequency +
              i * (max_frequency - min_frequency) / (CMIO_NFREQ - 1.);
    }
    cdf[0] = 0.;
    for (int i = 1; i < CMIO_NFREQ; ++i) {
      double A[2] = {0., 0.};
      for (int e = 0; e < 2; ++e) {
        const double y = nu[i - 1 + e] / nu0;
        if (y < 1.) {
          const uint_fast32_t j = cmio_locate(y, cmi_he2q_y, CMI_HE2Q_N);
          const double f =
              (y - cmi_he2q_y[j]) / (cmi_he2q_y[j + 1] - cmi_he2q_y[j]);
          A[e] = cmi_he2q_A[j] + f * (cmi_he2q_A[j + 1] - cmi_he2q_A[j]);
        }
      }
      cdf[i] = 0.5 * (A[0] + A[1]) * (nu[i] - nu[i - 1]);
    }
    for (int i = 1; i < CMIO_NFREQ; ++i) {
      cdf[i] = cdf[i - 1] + cdf[i];
    }
    const double total = cdf[CMIO_NFREQ - 1];
    for (int i = 0; i < CMIO_NFREQ; ++i) {
      cdf[i] /= total;
    }
  }
  return t;
}

void cmio_tables_free(cmio_tables *tables) { free(tables); }

/* src/HeliumTwoPhotonContinuumSpectrum.cpp:144-160 (used by the test) */
double cmio_he2pc_integral(void) {
  const double miny = 3.289e15 / 4.98e15;
  double integral = 0.;
  for (int i = 1; i < CMI_HE2Q_N; ++i) {
    if (cmi_he2q_y[i - 1] > miny) {
      integral += 0.5 * (cmi_he2q_A[i - 1] + cmi_he2q_A[i]) *
                  (cmi_he2q_y[i] - cmi_he2q_y[i - 1]);
    } else if (cmi_he2q_y[i] > miny) {
      integral += cmi_he2q_A[i] * (cmi_he2q_y[i] - miny);
    }
  }
  return integral * 4.98e15 / 3.289e15;
}

static double sample_planck_table(const double *cdf, const double *logcdf,
                                  const double *logfreq, cmio_rng *rng) {
  const double x = cmio_rng_next(rng);
  const uint_fast32_t ix = cmio_locate(x, cdf, CMIO_NFREQ);
  const double log_random_frequency =
      (log10(x) - logcdf[ix]) / (logcdf[ix + 1] - logcdf[ix]) *
          (logfreq[ix + 1] - logfreq[ix]) +
      logfreq[ix];
  const double frequency = pow(10., log_random_frequency);
  return frequency * 3.288465385e15;
}
static double sample_planck(const cmio_tables *t, cmio_rng *rng) {
  return sample_planck_table(t->planck_cdf, t->planck_logcdf,
                             t->planck_logfreq, rng);
}

static double sample_lyc(const cmio_tables *t, int s, double temperature,
                         cmio_rng *rng) {
  const uint_fast32_t iT = cmio_locate(temperature, t->lyc_T, CMIO_NTEMP);
  const double x = cmio_rng_next(rng);
  const uint_fast32_t inu1 = cmio_locate(x, t->lyc_cdf[s][iT], CMIO_NFREQ);
  const uint_fast32_t inu2 = cmio_locate(x, t->lyc_cdf[s][iT + 1], CMIO_NFREQ);
  const double *nu = t->lyc_freq[s];
  return nu[inu1] + (temperature - t->lyc_T[iT]) * (nu[inu2] - nu[inu1]) /
                        (t->lyc_T[iT + 1] - t->lyc_T[iT]);
}

static double sample_he2pc(const cmio_tables *t, cmio_rng *rng) {
  const double x = cmio_rng_next(rng);
  const uint_fast32_t inu = cmio_locate(x, t->he2pc_cdf, CMIO_NFREQ);
  return t->he2pc_freq[inu] + (t->he2pc_freq[inu + 1] - t->he2pc_freq[inu]) *
                                  (x - t->he2pc_cdf[inu]) /
                                  (t->he2pc_cdf[inu + 1] - t->he2pc_cdf[inu]);
}

double cmio_spectrum_sample(const cmio_model *model, cmio_rng *rng) {
  if (model->spectrum_type == CMIO_SPECTRUM_MONOCHROMATIC) {
    return model->mono_frequency; /* no random number drawn */
  }
  if (model->spectrum_type == CMIO_SPECTRUM_TABLE) {
    return cmio_table_value(&model->spectrum_table[0], 0, cmio_rng_next(rng));
  }
  if (!model->tables) {
    cmio_set_error("cmio: Planck spectrum needs cmio_tables_create");
    return NAN;
  }
  return sample_planck(model->tables, rng);
}

/* the continuous source's spectrum */
double cmio_continuous_spectrum_sample(const cmio_model *model,
                                       cmio_rng *rng) {
  if (model->continuous_spectrum_type == CMIO_SPECTRUM_MONOCHROMATIC) {
    return model->continuous_mono_frequency;
  }
  if (model->continuous_spectrum_type == CMIO_SPECTRUM_TABLE) {
    return cmio_table_value(&model->spectrum_table[1], 0, cmio_rng_next(rng));
  }
  if (!model->tables) {
    cmio_set_error("cmio: Planck spectrum needs cmio_tables_create");
    return NAN;
  }
  const cmio_tables *t = model->tables;
  return sample_planck_table(t->planck2_cdf, t->planck2_logcdf,
                             t->planck2_logfreq, rng);
}

void cmio_sample_spectrum(const cmio_model *model, int kind,
                          double temperature, uint32_t seed, uint64_t n,
                          double *out) {
  const cmio_tables *t = model->tables;
  for (uint64_t i = 0; i < n; ++i) {
    cmio_rng rng = {seed, 0u, i, 0u, NULL};
    switch (kind) {
    case 0:
      out[i] = sample_planck(t, &rng);
      break;
    case 1:
      out[i] = sample_lyc(t, 0, temperature, &rng);
      break;
    case 2:
      out[i] = sample_lyc(t, 1, temperature, &rng);
      break;
    default:
      out[i] = sample_he2pc(t, &rng);
      break;
    }
  }
}

void cmio_reemission_probabilities(double temperature, double p[5]) {
  const double T4 = temperature * 1.e-4;
  const double alpha_1_H = 1.58e-13 * pow(T4, -0.53);
  const double alpha_A_agn = 4.18e-13 * pow(T4, -0.7);
  p[0] = alpha_1_H / alpha_A_agn;
  const double alpha_1_He = 1.54e-13 * pow(T4, -0.486);
  const double alpha_e_2tS = 2.1e-13 * pow(T4, -0.381);
  const double alpha_e_2sS = 2.06e-14 * pow(T4, -0.451);
  const double alpha_e_2sP = 4.17e-14 * pow(T4, -0.695);
  const double alphaHe = alpha_1_He + alpha_e_2tS + alpha_e_2sS + alpha_e_2sP;
  const double He_LyC = alpha_1_He / alphaHe;
  const double He_NpEEv = He_LyC + alpha_e_2tS / alphaHe;
  const double He_TPC = He_NpEEv + alpha_e_2sS / alphaHe;
  const double He_LyA = He_TPC + alpha_e_2sP / alphaHe;
  p[1] = He_LyC;
  p[2] = He_NpEEv;
  p[3] = He_TPC;
  p[4] = He_LyA;
}

double cmio_reemit_frequency(const cmio_model *model, const cmio_photon *photon,
                             double AHe, double T, double xH, double xHe,
                             cmio_rng *rng, int32_t *type) {
  if (model->reemit_type == CMIO_REEMIT_FIXED) {
    const double u = cmio_rng_next(rng);
    if (u < model->reemit_fixed_probability) {
      *type = CMIO_TYPE_DIFFUSE_HI;
      return model->reemit_fixed_frequency;
    }
    *type = CMIO_TYPE_ABSORBED;
    return 0.;
  }
  const cmio_tables *t = model->tables;
  if (!t) {
    cmio_set_error("cmio: physical re-emission needs cmio_tables_create");
    *type = CMIO_TYPE_ABSORBED;
    return 0.;
  }
  /* the reference stores these per cell at the start of the iteration
   * (src/IonizationSimulation.cpp:380-383); T does not change while packets
   * fly, so evaluating them here gives the same numbers */
  double p[5];
  cmio_reemission_probabilities(T, p);

  double new_frequency = 0.;
  const double nH0anuH0 = xH * photon->cross_section[CMIO_ION_H_n];
  const double nHe0anuHe0 = xHe * AHe * photon->cross_section[CMIO_ION_He_n];
  const double pHabs = nH0anuH0 / (nH0anuH0 + nHe0anuHe0);

  double x = cmio_rng_next(rng);
  if (x <= pHabs) {
    x = cmio_rng_next(rng);
    if (x <= p[0]) {
      new_frequency = sample_lyc(t, 0, T, rng);
      *type = CMIO_TYPE_DIFFUSE_HI;
    } else {
      *type = CMIO_TYPE_ABSORBED;
    }
  } else {
    x = cmio_rng_next(rng);
    if (x <= p[1]) {
      new_frequency = sample_lyc(t, 1, T, rng);
      *type = CMIO_TYPE_DIFFUSE_HeI;
    } else if (x <= p[2]) {
      new_frequency = 4.788e15; /* 19.8 eV line */
      *type = CMIO_TYPE_DIFFUSE_HeI;
    } else if (x <= p[3]) {
      x = cmio_rng_next(rng);
      if (x < 0.56) {
        new_frequency = sample_he2pc(t, rng);
        *type = CMIO_TYPE_DIFFUSE_HeI;
      } else {
        *type = CMIO_TYPE_ABSORBED;
      }
    } else if (x <= p[4]) {
      const double sqrtTnH0 = sqrt(T) * xH;
      const double pHots = sqrtTnH0 / (sqrtTnH0 + 77. * xHe);
      x = cmio_rng_next(rng);
      if (x < pHots) {
        x = cmio_rng_next(rng);
        if (x <= p[0]) {
          new_frequency = sample_lyc(t, 0, T, rng);
          *type = CMIO_TYPE_DIFFUSE_HI;
        } else {
          *type = CMIO_TYPE_ABSORBED;
        }
      } else {
        x = cmio_rng_next(rng);
        if (x < 0.56) {
          new_frequency = sample_he2pc(t, rng);
          *type = CMIO_TYPE_DIFFUSE_HeI;
        } else {
          *type = CMIO_TYPE_ABSORBED;
        }
      }
    } else {
      *type = CMIO_TYPE_ABSORBED;
    }
  }
  return new_frequency;
}

double cmio_reemit_scripted(const cmio_model *model, double sigma_H,
                            double sigma_He, double AHe, double T, double xH,
                            double xHe, const double *uniforms,
                            int32_t *type, uint32_t *draws) {
  cmio_photon photon;
  for (int ion = 0; ion < CMIO_NION; ++ion)
    photon.cross_section[ion] = 0.;
  photon.cross_section[CMIO_ION_H_n] = sigma_H;
  photon.cross_section[CMIO_ION_He_n] = sigma_He;
  cmio_rng rng = {0u, 0u, 0u, 0u, uniforms};
  const double nu =
      cmio_reemit_frequency(model, &photon, AHe, T, xH, xHe, &rng, type);
  if (draws)
    *draws = rng.draw;
  return nu;
}
