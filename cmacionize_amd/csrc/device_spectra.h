/*
 * device_spectra.h - device samplers for the source spectrum and the diffuse
 * re-emission spectra. The CDF tables (SpectraDev) are built on the host at
 * initialisation - the lowering of the reference's PhotonSourceSpectrum
 * plugins into device descriptors.
 */
#ifndef CMI_DEVICE_SPECTRA_H
#define CMI_DEVICE_SPECTRA_H

#include "device_physics.h"

/* PlanckPhotonSourceSpectrum::get_random_frequency,
 * src/PlanckPhotonSourceSpectrum.cpp:149-165: log-log interpolation */
__device__ inline double sample_planck_table(const double *cdf,
                                             const double *logcdf,
                                             const double *logfreq,
                                             const uint16_t *guide,
                                             PacketRng &rng) {
  const double x = rng.next();
  const uint32_t ix = cmi_locate_guided(x, cdf, guide, CMI_NFREQ);
  const double log_random_frequency =
      (log10(x) - logcdf[ix]) / (logcdf[ix + 1] - logcdf[ix]) *
          (logfreq[ix + 1] - logfreq[ix]) +
      logfreq[ix];
  const double frequency = exp10(log_random_frequency);
  return frequency * 3.288465385e15;
}
__device__ inline double sample_planck(const SpectraDev *s, PacketRng &rng) {
  return sample_planck_table(s->planck_cdf, s->planck_logcdf,
                             s->planck_logfreq, s->planck_guide, rng);
}

/* Hydrogen/HeliumLymanContinuumSpectrum::get_random_frequency,
 * src/HydrogenLymanContinuumSpectrum.cpp:136-153 (which = 0) and
 * src/HeliumLymanContinuumSpectrum.cpp:147-164 (which = 1): the same uniform
 * is located in the two bracketing temperature rows, the frequencies are
 * interpolated linearly in T */
__device__ inline double sample_lyman_continuum(const SpectraDev *s, int which,
                                                double temperature,
                                                PacketRng &rng) {
  const uint32_t iT = cmi_locate_linear(temperature, s->lyc_T, CMI_NTEMP);
  const double x = rng.next();
  const uint32_t inu1 = cmi_locate_guided(x, s->lyc_cdf[which][iT],
                                          s->lyc_guide[which][iT], CMI_NFREQ);
  const uint32_t inu2 =
      cmi_locate_guided(x, s->lyc_cdf[which][iT + 1],
                        s->lyc_guide[which][iT + 1], CMI_NFREQ);
  const double *nu = s->lyc_freq[which];
  return nu[inu1] + (temperature - s->lyc_T[iT]) * (nu[inu2] - nu[inu1]) /
                        (s->lyc_T[iT + 1] - s->lyc_T[iT]);
}

/* HeliumTwoPhotonContinuumSpectrum::get_random_frequency,
 * src/HeliumTwoPhotonContinuumSpectrum.cpp:167-180 */
__device__ inline double sample_he_two_photon(const SpectraDev *s,
                                              PacketRng &rng) {
  const double x = rng.next();
  const uint32_t inu =
      cmi_locate_guided(x, s->he2pc_cdf, s->he2pc_guide, CMI_NFREQ);
  return s->he2pc_freq[inu] + (s->he2pc_freq[inu + 1] - s->he2pc_freq[inu]) *
                                  (x - s->he2pc_cdf[inu]) /
                                  (s->he2pc_cdf[inu + 1] - s->he2pc_cdf[inu]);
}

/* A spectrum known only through PhotonSourceSpectrum::get_random_frequency
 * (src/PhotonSourceSpectrum.hpp:48-50), lowered by the host into its quantile
 * function: one uniform, located in the cumulative distribution x[], the
 * frequency interpolated between y[i] and y[i + 1] - linearly (the form of
 * src/HeliumTwoPhotonContinuumSpectrum.cpp:167-180) or log-log (the form of
 * src/PlanckPhotonSourceSpectrum.cpp:149-165). */
__device__ inline double sample_spectrum_table(const TableDev &t,
                                               PacketRng &rng) {
  return cmi_table_value(t, 0, rng.next());
}

/* PhotonSourceSpectrum::get_random_frequency of the discrete sources
 * (origin 0) or of the continuous source (origin 1) */
__device__ inline double sample_source_spectrum(const ModelDev &m,
                                                PacketRng &rng,
                                                uint32_t origin = 0) {
  if (origin != 0) {
    if (m.continuous_spectrum_type == 0)
      return m.continuous_mono_frequency;
    if (CMI_UNLIKELY(m.continuous_spectrum_type == 2))
      return sample_spectrum_table(m.spectrum_table[1], rng);
    return sample_planck_table(m.spectra->planck2_cdf, m.spectra->planck2_logcdf,
                               m.spectra->planck2_logfreq,
                               m.spectra->planck2_guide, rng);
  }
  if (m.spectrum_type == 0) {
    /* MonochromaticPhotonSourceSpectrum: no random number is drawn
     * (src/MonochromaticPhotonSourceSpectrum.hpp:97-100) */
    return m.mono_frequency;
  }
  if (CMI_UNLIKELY(m.spectrum_type == 2))
    return sample_spectrum_table(m.spectrum_table[0], rng);
  return sample_planck(m.spectra, rng);
}

#endif
